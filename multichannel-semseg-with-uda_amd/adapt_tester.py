#!/usr/bin/env python3
"""MCD inference -- the reference's ``adapt_tester.py`` (:17-146): load a trained checkpoint, eval-mode G -> F1
(optionally averaged with F2), argmax over the non-background classes, write uint8 label PNGs (resized NEAREST to the
test shape), report the mean prediction entropy.

On the HIP path eval-mode BatchNorm is folded into the convolution epilogue (``mcdseg_conv_fprop_affine``) and the
argmax / entropy tail is one kernel (``mcdseg_predict_labels``).  The palette visualisation and ``eval.py`` of the
reference are outside this build.

    python adapt_tester.py nyu train_output/.../pth/MCD-normal-drn_d_38-1.pth.tar --synthetic
"""
import os

import numpy as np
import torch
from PIL import Image

from argmyparse import add_additional_params_to_args, get_da_mcd_testing_parser
from datasets import get_dataset
from models.model_util import get_models
from util import check_if_done, load_checkpoint, mkdir_if_not_exist, save_dic_to_json
from mcdseg import ops
from eval import ConfusionMeter


def main(argv=None, mfnet=False):
    """``mfnet=True`` is the two-encoder variant (adapt_mfnet_tester.py): checkpoints hold ``g_3ch_state_dict`` /
    ``g_1ch_state_dict``, the classifier takes both feature maps and only F1 is evaluated (:102-105)."""
    args = get_da_mcd_testing_parser().parse_args(argv)
    args = add_additional_params_to_args(args)
    if not torch.cuda.is_available():
        raise SystemExit("this tester runs on an MI355X: the HIP kernels are the only implementation (no CPU fallback)")
    dev = torch.device("cuda", torch.cuda.current_device())
    indir, infn = os.path.split(args.trained_checkpoint)
    trained_mode = indir.split(os.path.sep)[-2]
    args.mode = "%s---%s-%s" % (trained_mode, args.tgt_dataset, args.split)
    model_name = (infn.replace(".pth", "").replace(".tar", "") if mfnet else infn.replace(".pth", "")) + ("-use_f2" if args.use_f2 else "")
    if not os.path.exists(args.trained_checkpoint):
        raise OSError("%s does not exist!" % args.trained_checkpoint)
    checkpoint = load_checkpoint(args.trained_checkpoint)
    train_args = checkpoint["args"]
    args.start_epoch = checkpoint["epoch"]
    base_outdir = os.path.join(args.outdir, args.mode, model_name)
    mkdir_if_not_exist(base_outdir)
    json_fn = os.path.join(base_outdir, "param.json")
    check_if_done(json_fn)
    save_dic_to_json(dict(vars(args)), json_fn, verbose=False)

    train_img_shape = [int(x) for x in train_args.train_img_shape]
    test_img_shape = tuple(int(x) for x in args.test_img_shape)
    spec = dict(length=args.synthetic_len, img_shape=train_img_shape, n_class=train_args.n_class, seed=args.seed) if args.synthetic else None
    tgt_dataset = get_dataset(dataset_name=args.tgt_dataset, split=args.split, img_transform=None, label_transform=None, test=True,
                              input_ch=train_args.input_ch, synthetic=spec)
    loader = torch.utils.data.DataLoader(tgt_dataset, batch_size=args.batch_size, pin_memory=True)

    os.environ["MCDSEG_PRETRAINED"] = "0"  # weights come from the checkpoint
    method = getattr(train_args, "method", "MCD")
    if mfnet:  # the trainer stores the full "MFNet-<fusion>" name in method_detail (adapt_mfnet_trainer.py:29)
        method = train_args.method_detail if "MFNet" in train_args.method_detail else method + "-" + train_args.method_detail
    models = get_models(net_name=train_args.net, res=train_args.res, input_ch=train_args.input_ch, n_class=train_args.n_class,
                        method=method, is_data_parallel=getattr(train_args, "is_data_parallel", False))
    if mfnet:
        G3, G1, F1, F2 = models
        G3.load_state_dict(checkpoint["g_3ch_state_dict"])
        G1.load_state_dict(checkpoint["g_1ch_state_dict"])
        encoders = (G3, G1)
    else:
        G, F1, F2 = models
        G.load_state_dict(checkpoint["g_state_dict"])
        encoders = (G,)
    F1.load_state_dict(checkpoint["f1_state_dict"])
    if args.use_f2:
        F2.load_state_dict(checkpoint["f2_state_dict"])
    for m in encoders + (F1, F2):
        m.eval()
        m.to(dev)
    n_used = args.n_class if getattr(train_args, "add_bg_loss", False) else args.n_class - 1

    label_outdir = os.path.join(base_outdir, "label")
    mkdir_if_not_exist(label_outdir)
    total_ent, batches = 0.0, 0
    # the reference shells out to eval.py over the written PNGs (util.py:36-41); here the confusion matrix is accumulated
    # on the device while the label maps are still there (background = 255 in label PNGs, n_class-1 in training labels)
    meter = ConfusionMeter(train_args.n_class, background_id=255, device=dev)
    with torch.no_grad():
        for imgs, gts, paths in loader:
            imgs = imgs.to(dev, non_blocking=True)
            if mfnet:
                out1 = F1(encoders[0](imgs[:, :3, :, :]), encoders[1](imgs[:, 3:, :, :]))
                out2 = None  # adapt_mfnet_tester.py:105 evaluates F1 alone, with or without --use_f2
            else:
                feature = encoders[0](imgs)
                out1 = F1(feature)
                out2 = F2(feature) if args.use_f2 else None
            labels, ent = ops.predict_labels(out1, out2, n_used)
            total_ent += float(ent)
            batches += 1
            if torch.is_tensor(gts) and gts.dim() == 3 and tuple(gts.shape) == tuple(labels.shape):
                gts = gts.to(dev)
                meter.update(labels, torch.where(gts == train_args.n_class - 1, torch.full_like(gts, 255), gts))
            if args.saves_prob:
                prob_outdir = os.path.join(base_outdir, "prob")
                mkdir_if_not_exist(prob_outdir)
                avg = out1 if out2 is None else (out1 + out2) / 2
                for k, path in enumerate(paths):
                    np.save(os.path.join(prob_outdir, os.path.basename(path).replace("png", "npy")), avg[k].cpu().numpy())
            lab = labels.cpu().numpy()
            for k, path in enumerate(paths):
                img = Image.fromarray(lab[k]).resize(test_img_shape, Image.NEAREST)
                img.save(os.path.join(label_outdir, os.path.basename(path)))
    ave_ent = total_ent / max(batches, 1)
    print("average entropy: %s" % ave_ent)
    with open(os.path.join(base_outdir, "ave_ent_%s.txt" % ave_ent), "w") as f:
        f.write(str(ave_ent))
    if int(meter.hist.sum()) > 0:
        summary = meter.summary()
        save_dic_to_json(summary, os.path.join(base_outdir, "eval_result.json"), verbose=False)
        print("pixAcc %.2f  mAcc %.2f  fwIoU %.2f  mIoU %.2f" % (summary["pixAcc"], summary["mAcc"], summary["fwIoU"], summary["mIoU"]))
    return label_outdir, ave_ent


if __name__ == "__main__":
    main()
