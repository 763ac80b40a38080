"""Pieces shared by the three entry scripts: process-group / device set-up, output directories, the
optional tensorboard logger, synthetic data loaders and rank-0 checkpointing."""
import os

import torch

from datasets import ConcatDataset, get_dataset
from mcdseg import dist as mdist
from util import mkdir_if_not_exist, save_checkpoint, save_dic_to_json, check_if_done


class Run:
    def __init__(self, args):
        self.rank, self.world, self.local = mdist.init_from_env()
        if not torch.cuda.is_available():
            raise SystemExit("this trainer runs on an MI355X: the HIP kernels are the only implementation (no CPU fallback)")
        self.device = torch.device("cuda", self.local if self.world > 1 else torch.cuda.current_device())
        torch.cuda.set_device(self.device)
        if getattr(args, "no_pretrained", False):
            os.environ["MCDSEG_PRETRAINED"] = "0"
        if getattr(args, "dtype", "f32") == "f16":  # reduced precision (BASELINE config 5 "bf16"): one fp16 term per product and, inside the
            from mcdseg import ops                   # trunk, ONE 16-bit value per activation / z / gradient element (ops.HALF_STORAGE) --
            ops.CONV_MATH = "f16x1"                  # unless MCDSEG_ACT_STORAGE says how activations are to be kept
            if "MCDSEG_ACT_STORAGE" not in os.environ:
                ops.ACT_STORAGE = "compact"
        torch.manual_seed(getattr(args, "seed", 1234))
        self._log = None
        self._pipe = None
        self._args = args

    def images(self, t):
        """batch -> device; raw uint8 HWC batches (``--synthetic_raw``) go through the device input pipeline"""
        if t.dtype == torch.uint8:
            return self._pipeline().images(t)
        return t.to(self.device, non_blocking=True)

    def labels(self, t):
        if t.dtype == torch.uint8:
            return self._pipeline().labels(t)
        return t.to(self.device, non_blocking=True)

    def _pipeline(self):
        if self._pipe is None:
            from datasets import DeviceInputPipeline
            lists = getattr(self._args, "src_file_list", None) or getattr(self._args, "tgt_file_list", None)
            self._pipe = DeviceInputPipeline(self._args.input_ch, self._args.n_class, self.device,
                                             background_id=getattr(self._args, "background_id", 255),
                                             img_shape=self._args.train_img_shape if lists else None)  # real files: Scale on the GPU
        return self._pipe

    @property
    def is_main(self):
        return self.rank == 0

    def configure_logger(self, tflog_dir, args):
        if not self.is_main:
            return
        mkdir_if_not_exist(tflog_dir)
        try:
            from tensorboard_logger import configure, log_value
            configure(tflog_dir, flush_secs=5)
            self._log = log_value
        except ImportError:
            if not getattr(args, "no_tflog", False):
                print("tensorboard_logger is not installed: scalar logs go to stdout only")

    def log_value(self, name, value, step):
        if self.is_main:
            if self._log is not None:
                self._log(name, value, step)
            print("  [%d] %s = %s" % (step, name, value))

    def save_params(self, args, json_fn):
        if self.is_main:
            check_if_done(json_fn)
            save_dic_to_json(dict(vars(args)), json_fn, verbose=False)
        mdist.barrier()

    def save(self, save_dic, filename):
        """rank 0 writes the checkpoint (its BatchNorm running statistics are the ones kept, as replica 0's are
        under nn.DataParallel)."""
        if self.is_main:
            save_checkpoint(save_dic, is_best=False, filename=filename)
            print("saved %s" % filename)
        mdist.barrier()

    def sync_replicas(self, modules):
        """identical start on every rank (same seed already gives that; the broadcast makes it unconditional)"""
        if self.world > 1:
            for m in modules:
                mdist.broadcast_([p.data for p in m.parameters()] + [b for b in m.buffers()])


def synthetic_spec(args, seed_offset, rank):
    shape = [int(x) for x in args.train_img_shape]
    return dict(length=args.synthetic_len, img_shape=shape, n_class=args.n_class, seed=args.seed + seed_offset + 101 * rank,
                raw=getattr(args, "synthetic_raw", False), background_id=getattr(args, "background_id", 255))


def make_loader(args, run, names_splits):
    """DataLoader over one dataset or a ConcatDataset of (source, target); per-rank shard by seed."""
    sets = []
    lists = [getattr(args, "src_file_list", None), getattr(args, "tgt_file_list", None)]
    for i, (name, split) in enumerate(names_splits):
        spec = synthetic_spec(args, 7 * i, run.rank) if (args.synthetic or getattr(args, "synthetic_raw", False)) else None
        sets.append(get_dataset(dataset_name=name, split=split, img_transform=None, label_transform=None, test=False,
                                input_ch=args.input_ch, synthetic=spec, file_list=lists[i] if i < 2 else None))
    ds = sets[0] if len(sets) == 1 else ConcatDataset(*sets)
    return torch.utils.data.DataLoader(ds, batch_size=args.batch_size, shuffle=True, pin_memory=True, drop_last=True)
