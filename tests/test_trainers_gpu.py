"""GPU: the entry points (adapt_trainer / adapt_mfnet_trainer / source_trainer) end to end on synthetic data:
CLI, output layout, checkpoint dict layout of the reference, resume."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


COMMON = ["--input_ch", "6", "-b", "2", "--train_img_shape", "96", "64", "--synthetic", "--synthetic_len", "4",
          "--no_pretrained", "--no_tflog", "--epochs", "1", "--max_iter", "10"]


@pytest.mark.parametrize("solver", ["fused", "dropin"])
def test_adapt_trainer_checkpoint_and_resume(tmp_path, solver):
    _need_gpu()
    import adapt_trainer
    import util
    out = str(tmp_path / "out")
    assert adapt_trainer.main(["suncg", "nyu", "--base_outdir", out, "--solver", solver] + COMMON) == 0
    d = os.path.join(out, "suncg-train2nyu-train_6ch")
    ck_fn = os.path.join(d, "pth", "MCD-normal-drn_d_38-1.pth.tar")
    assert os.path.exists(ck_fn) and os.path.exists(os.path.join(d, "param-MCD-normal-drn_d_38.json"))
    ck = util.load_checkpoint(ck_fn)
    assert sorted(ck.keys()) == ["args", "epoch", "f1_state_dict", "f2_state_dict", "g_state_dict", "optimizer_f", "optimizer_g"]
    assert ck["epoch"] == 1 and ck["args"].n_class == 41 and ck["args"].start_epoch == 1
    assert len(ck["g_state_dict"]) == 248 and list(ck["f1_state_dict"].keys()) == ["up.weight"]
    assert int(ck["g_state_dict"]["base.0.1.num_batches_tracked"]) == 14  # 2 iterations x 7 forwards
    assert len(ck["optimizer_g"]["state"]) == len(ck["optimizer_g"]["param_groups"][0]["params"])
    assert all(torch.isfinite(v).all() for v in ck["g_state_dict"].values())
    # resume for one more epoch: the pickled args replace the CLI ones, epochs comes from the checkpoint
    ck["args"].epochs = 2
    util.save_checkpoint(ck, False, ck_fn)
    assert adapt_trainer.main(["suncg", "nyu", "--resume", ck_fn] + COMMON) == 0
    ck2 = util.load_checkpoint(os.path.join(d, "pth", "MCD-normal-drn_d_38-2.pth.tar"))
    assert ck2["epoch"] == 2 and int(ck2["g_state_dict"]["base.0.1.num_batches_tracked"]) == 28


def test_adapt_trainer_in_the_two_byte_chain(tmp_path, monkeypatch):
    """``adapt_trainer.py --dtype f16`` (BASELINE config 5's "bf16"): the one-term arithmetic with 2-byte activation storage inside the trunk,
    end to end through the CLI -- the chain's kernels run, the checkpoint has the reference's layout and finite fp32 state, and a resumed
    epoch continues from it."""
    _need_gpu()
    import adapt_trainer
    import util
    from mcdseg import ops
    monkeypatch.delenv("MCDSEG_ACT_STORAGE", raising=False)
    monkeypatch.setattr(ops, "CONV_MATH", ops.CONV_MATH)      # (the trainer sets both: restored when the test ends)
    monkeypatch.setattr(ops, "ACT_STORAGE", ops.ACT_STORAGE)
    names = []

    class _Names:
        def wants(self, name):
            names.append(name)
            return False
    monkeypatch.setattr(ops, "LAUNCH_TIMER", _Names())
    out = str(tmp_path / "out")
    assert adapt_trainer.main(["suncg", "nyu", "--base_outdir", out, "--dtype", "f16"] + COMMON) == 0
    assert (ops.CONV_MATH, ops.ACT_STORAGE) == ("f16x1", "compact")
    assert {"bn_apply_half", "bn_bwd_reduce_half", "bn_bwd_apply_half"} <= set(names), sorted(set(names))
    assert any("SplitF16x1" in nm for nm in names) and not any("SplitF16x3" in nm for nm in names), sorted(set(names))
    ck = util.load_checkpoint(os.path.join(out, "suncg-train2nyu-train_6ch", "pth", "MCD-normal-drn_d_38-1.pth.tar"))
    assert len(ck["g_state_dict"]) == 248 and all(v.dtype != torch.bfloat16 and torch.isfinite(v.float()).all() for v in ck["g_state_dict"].values())
    assert int(ck["g_state_dict"]["base.0.1.num_batches_tracked"]) == 14


def test_adapt_mfnet_trainer(tmp_path):
    _need_gpu()
    import adapt_mfnet_trainer
    import util
    out = str(tmp_path / "out")
    assert adapt_mfnet_trainer.main(["suncg", "nyu", "--base_outdir", out, "--method_detail", "MFNet-ScoreAddFusion"] + COMMON) == 0
    ck = util.load_checkpoint(os.path.join(out, "suncg-train2nyu-train_6ch_MFNet", "pth", "MFNet-ScoreAddFusion-normal-drn_d_38-1.pth.tar"))
    assert sorted(ck.keys()) == ["args", "epoch", "f1_state_dict", "f2_state_dict", "g_1ch_state_dict", "g_3ch_state_dict",
                                 "optimizer_f", "optimizer_g"]
    assert sorted(ck["f1_state_dict"].keys()) == ["up1.weight", "up2.weight"]
    assert list(ck["g_3ch_state_dict"]["base.0.0.weight"].shape) == [16, 3, 7, 7]


def test_source_trainer_cfg1(tmp_path):
    _need_gpu()
    import source_trainer
    import util
    out = str(tmp_path / "out")
    args = ["suncg", "--base_outdir", out, "--input_ch", "6", "-b", "2", "--train_img_shape", "320", "240", "--synthetic",
            "--synthetic_len", "2", "--no_pretrained", "--no_tflog", "--epochs", "1"]
    assert source_trainer.main(args) == 0
    ck = util.load_checkpoint(os.path.join(out, "suncg-train_only_6ch", "pth", "normal-drn_d_38-1.pth.tar"))
    assert sorted(ck.keys()) == ["args", "epoch", "optimizer", "state_dict"]
    assert all(k.startswith("module.") for k in ck["state_dict"])  # DataParallel prefix (model_util.py:36-37)
    assert "module.up.weight" in ck["state_dict"] and "module.seg.bias" in ck["state_dict"]


def test_adapt_multitask_trainer(tmp_path):
    _need_gpu()
    import adapt_multitask_trainer
    import util
    out = str(tmp_path / "out")
    assert adapt_multitask_trainer.main(["suncg", "nyu", "--base_outdir", out] + COMMON) == 0
    ck = util.load_checkpoint(os.path.join(out, "suncg-train2nyu-train_6ch_MCDmultitask", "pth", "MCD-normal-drn_d_38-1.pth.tar"))
    assert sorted(ck.keys()) == ["args", "dec_state_dict", "enc_state_dict", "epoch", "optimizer_dec", "optimizer_enc"]
    assert len(ck["dec_state_dict"]) == 51 and "s_semsegcls" in ck["dec_state_dict"]
    assert list(ck["dec_state_dict"]["deprgr_dec.conv3.weight"].shape) == [3, 512, 1, 1]


def test_adapt_tester_label_maps_match_oracle(tmp_path):
    """Train one synthetic epoch, run the tester (folded-BN inference + argmax/entropy kernel) and compare the written
    label PNGs and the entropy with the CPU oracle evaluating the same checkpoint (adapt_tester.py:87-126)."""
    _need_gpu()
    import numpy as np
    from PIL import Image
    import adapt_tester
    import adapt_trainer
    import util
    from datasets import SyntheticRGBD
    from oracle import ref_models
    out = str(tmp_path / "out")
    assert adapt_trainer.main(["suncg", "nyu", "--base_outdir", out] + COMMON) == 0
    ck_fn = os.path.join(out, "suncg-train2nyu-train_6ch", "pth", "MCD-normal-drn_d_38-1.pth.tar")
    label_dir, ent = adapt_tester.main(["nyu", ck_fn, "--outdir", str(tmp_path / "test"), "--synthetic", "--synthetic_len", "3",
                                        "--use_f2", "--test_img_shape", "96", "64"])
    ck = util.load_checkpoint(ck_fn)
    g, f1, f2 = ref_models.get_models("drn_d_38", 6, 41)
    g.load_state_dict(ck["g_state_dict"]), f1.load_state_dict(ck["f1_state_dict"]), f2.load_state_dict(ck["f2_state_dict"])
    for m in (g, f1, f2):
        m.eval()
    ds = SyntheticRGBD(3, 6, [96, 64], 41, seed=4321, test=True)
    ents, agree, safe_total = [], 0, 0
    for i in range(3):
        img, _, name = ds[i]
        with torch.no_grad():
            o = (f1(g(img[None])) + f2(g(img[None]))) / 2
        p = torch.softmax(o, dim=1)
        ents.append(float(-(p * torch.log(p + 1e-6)).mean()))
        top2 = o[0, :40].topk(2, dim=0).values
        safe = ((top2[0] - top2[1]) > 2e-3).numpy()
        got = np.array(Image.open(os.path.join(label_dir, name)))
        ref = o[0, :40].argmax(0).numpy().astype(np.uint8)
        assert got.shape == ref.shape == (64, 96) and got.max() <= 39
        assert (got == ref)[safe].all()
        agree += int((got == ref).sum())
        safe_total += got.size
    assert agree / safe_total > 0.995
    assert abs(ent - sum(ents) / 3) <= 1e-4 * abs(sum(ents) / 3)
    # device-side evaluation written next to the label maps: same numbers as the oracle's fast_hist over the PNGs
    import json
    from oracle import ref_io
    res = json.load(open(os.path.join(os.path.dirname(label_dir), "eval_result.json")))
    hist = np.zeros((41, 41), dtype=np.int64)
    for i in range(3):
        _, lbl, name = ds[i]
        gt = lbl.numpy().copy()
        gt[gt == 40] = 255
        pred = np.array(Image.open(os.path.join(label_dir, name)).resize((96, 64), Image.NEAREST)).astype(np.int64)
        hist += ref_io.fast_hist(gt.flatten(), pred.flatten(), 41)
    used = np.where(hist.sum(1) != 0)[0]
    sub = hist[used][:, used].astype(np.float64)
    assert res["used_class_ids"] == used.tolist()
    assert abs(res["pixAcc"] - 100 * ref_io.calc_pixel_accuracy(sub)) < 1e-9
    assert abs(res["mIoU"] - 100 * np.mean(ref_io.per_class_iu(sub))) < 1e-9


def test_adapt_trainer_raw_uint8_input_pipeline(tmp_path):
    """--synthetic_raw: uint8 HWC images and uint8 labels (background 255) go through the device-side
    ToTensor/Normalize/ReLabel kernels; the step must equal feeding the oracle-preprocessed fp32 batch."""
    _need_gpu()
    import numpy as np
    from datasets import DeviceInputPipeline, SyntheticRGBD
    from oracle import ref_io
    dev = torch.device("cuda:0")
    ds = SyntheticRGBD(2, 6, [48, 32], 41, seed=11, raw=True)
    img, lbl = ds[0]
    assert img.dtype == torch.uint8 and tuple(img.shape) == (32, 48, 6) and lbl.dtype == torch.uint8 and int(lbl.max()) == 255
    pipe = DeviceInputPipeline(6, 41, dev)
    x = pipe.images(img[None])
    y = pipe.labels(lbl[None])
    assert np.array_equal(x.cpu().numpy(), ref_io.normalize_u8(img[None].numpy(), ref_io.IMAGENET_MEAN6, ref_io.IMAGENET_STD6))
    assert np.array_equal(y.cpu().numpy(), ref_io.relabel(lbl[None].numpy(), 255, 40))
    import adapt_trainer
    out = str(tmp_path / "out")
    args = [a for a in COMMON if a != "--synthetic"] + ["--synthetic_raw"]
    assert adapt_trainer.main(["suncg", "nyu", "--base_outdir", out] + args) == 0
    assert os.path.exists(os.path.join(out, "suncg-train2nyu-train_6ch", "pth", "MCD-normal-drn_d_38-1.pth.tar"))


def test_adapt_trainer_from_image_files(tmp_path):
    """--src_file_list / --tgt_file_list: PNG files on disk (RGB, HHA, label; other sizes than --train_img_shape) are opened
    with PIL and resized / normalised / relabelled on the device; the trainer runs end to end on them"""
    _need_gpu()
    import numpy as np
    from PIL import Image
    rng = np.random.RandomState(4)
    for dom in ("src", "tgt"):
        lines = []
        for i in range(4):
            for kind in ("rgb", "hha"):
                Image.fromarray(rng.randint(0, 256, size=(80, 120, 3)).astype(np.uint8)).save(tmp_path / ("%s_%s%d.png" % (dom, kind, i)))
            lbl = rng.randint(0, 40, size=(80, 120)).astype(np.uint8)
            lbl[rng.rand(80, 120) < 0.1] = 255
            Image.fromarray(lbl).save(tmp_path / ("%s_lbl%d.png" % (dom, i)))
            lines.append("%s_rgb%d.png %s_hha%d.png %s_lbl%d.png" % (dom, i, dom, i, dom, i))
        (tmp_path / (dom + ".txt")).write_text("\n".join(lines) + "\n")
    import adapt_trainer
    out = str(tmp_path / "out")
    args = [a for a in COMMON if a != "--synthetic"] + ["--src_file_list", str(tmp_path / "src.txt"), "--tgt_file_list", str(tmp_path / "tgt.txt")]
    assert adapt_trainer.main(["suncg", "nyu", "--base_outdir", out] + args) == 0
    ck = torch.load(os.path.join(out, "suncg-train2nyu-train_6ch", "pth", "MCD-normal-drn_d_38-1.pth.tar"), weights_only=False)
    assert all(torch.isfinite(v).all() for v in ck["g_state_dict"].values())
    assert int(ck["g_state_dict"]["base.0.1.num_batches_tracked"]) == 14  # 4 files, batch 2: two iterations x 7 forwards


def test_adapt_mfnet_trainer_score_gate_fusion_matches_oracle(tmp_path):
    """MFNet-ScoreGateFusion with ProbCrossEntropyLoss2d (adapt_mfnet_trainer.py:149): one full three-step update on the
    HIP path against the CPU oracle running the reference's statements from the same initial state."""
    _need_gpu()
    from loss import ProbCrossEntropyLoss2d, get_prob_distance_criterion
    from models.model_util import get_models, get_optimizer
    from oracle import ref_loss, ref_mcd, ref_models
    from recipe import fill_state_, make_batch
    from solvers.solver import MFNetMCDSolver
    dev = torch.device("cuda:0")
    os.environ["MCDSEG_PRETRAINED"] = "0"
    method = "MFNet-ScoreGateFusion"
    hip = get_models("drn_d_38", input_ch=6, n_class=41, method=method)
    ora = ref_models.get_models("drn_d_38", 6, 41, method=method)
    for i, (h, o) in enumerate(zip(hip, ora)):
        fill_state_(o, 500 + i)
        h.load_state_dict(o.state_dict())
        h.to(dev).train(), o.train()
    src, lbl, tgt = make_batch(77, 2, 6, 64, 96, 41)
    w = ref_loss.class_weights(41)
    o_og = torch.optim.SGD(list(ora[0].parameters()) + list(ora[1].parameters()), lr=1e-3, momentum=0.9, weight_decay=2e-5)
    o_of = torch.optim.SGD(list(ora[2].parameters()) + list(ora[3].parameters()), lr=1e-3, momentum=0.9, weight_decay=2e-5)
    oc, od = ref_mcd.mfnet_mcd_step(ora[0], ora[1], ora[2], ora[3], o_og, o_of, ref_loss.ProbCrossEntropyLoss2d(w), ref_loss.Diff2d(),
                                    src, lbl, tgt, num_k=2)
    h_og = get_optimizer(list(hip[0].parameters()) + list(hip[1].parameters()), "sgd", 1e-3, 0.9, 2e-5)
    h_of = get_optimizer(list(hip[2].parameters()) + list(hip[3].parameters()), "sgd", 1e-3, 0.9, 2e-5)
    solver = MFNetMCDSolver(hip[0], hip[1], hip[2], hip[3], h_og, h_of, ProbCrossEntropyLoss2d(w.to(dev)),
                            get_prob_distance_criterion("diff"), num_k=2)
    hc, hd = solver.step(src.to(dev), lbl.to(dev), tgt.to(dev))
    assert abs(float(hc) - oc) <= 2e-4 * abs(oc), (float(hc), oc)
    assert abs(float(hd) - od) <= 1e-2 * abs(od) + 1e-9, (float(hd), od)
    for h, o in zip(hip[2:], ora[2:]):  # classifier parameters after the update (gate conv + both up-samplers)
        for (k, a), (_, b) in zip(h.state_dict().items(), o.state_dict().items()):
            assert float((a.cpu() - b).norm()) <= 3e-4 * float(b.norm()) + 1e-7, k
    import adapt_mfnet_trainer
    out = str(tmp_path / "out")
    assert adapt_mfnet_trainer.main(["suncg", "nyu", "--base_outdir", out, "--method_detail", "MFNet-GateFusion"] + COMMON) == 0


def test_adapt_mfnet_tester_label_maps_match_oracle(tmp_path):
    """two-encoder tester (adapt_mfnet_tester.py:85-132): label PNGs and entropy against the CPU oracle on the same
    checkpoint -- F1 alone is evaluated, on the RGB / HHA feature pair"""
    _need_gpu()
    import numpy as np
    from PIL import Image
    import adapt_mfnet_tester
    import adapt_mfnet_trainer
    import util
    from datasets import SyntheticRGBD
    from oracle import ref_models
    out = str(tmp_path / "out")
    assert adapt_mfnet_trainer.main(["suncg", "nyu", "--base_outdir", out, "--method_detail", "MFNet-ScoreAddFusion"] + COMMON) == 0
    ck_fn = os.path.join(out, "suncg-train2nyu-train_6ch_MFNet", "pth", "MFNet-ScoreAddFusion-normal-drn_d_38-1.pth.tar")
    label_dir, ent = adapt_mfnet_tester.main(["nyu", ck_fn, "--outdir", str(tmp_path / "test"), "--synthetic", "--synthetic_len", "2",
                                              "--test_img_shape", "96", "64"])
    ck = util.load_checkpoint(ck_fn)
    g3, g1, f1, _ = ref_models.get_models("drn_d_38", 6, 41, method="MFNet-ScoreAddFusion")
    g3.load_state_dict(ck["g_3ch_state_dict"]), g1.load_state_dict(ck["g_1ch_state_dict"]), f1.load_state_dict(ck["f1_state_dict"])
    for m in (g3, g1, f1):
        m.eval()
    ds = SyntheticRGBD(2, 6, [96, 64], 41, seed=4321, test=True)
    ents = []
    for i in range(2):
        img, _, name = ds[i]
        with torch.no_grad():
            o = f1(g3(img[None, :3]), g1(img[None, 3:]))
        p = torch.softmax(o, dim=1)
        ents.append(float(-(p * torch.log(p + 1e-6)).mean()))
        top2 = o[0, :40].topk(2, dim=0).values
        safe = ((top2[0] - top2[1]) > 2e-3).numpy()
        got = np.array(Image.open(os.path.join(label_dir, name)))
        ref = o[0, :40].argmax(0).numpy().astype(np.uint8)
        assert got.shape == ref.shape == (64, 96)
        assert (got == ref)[safe].all() and (got == ref).mean() > 0.995
    assert abs(ent - sum(ents) / 2) <= 1e-4 * abs(sum(ents) / 2)


def _run_ranks(world, script, args, extra_env=None, timeout=600):
    import socket
    import subprocess
    import sys
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MCDSEG_SINGLE_DEVICE="1", MCDSEG_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4",
               MCDSEG_PRETRAINED="0")
    env.update(extra_env or {})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(port), script] + args
    return subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)


def test_two_ranks_on_one_gpu_equal_the_single_process_step(tmp_path):
    """Data parallelism through the real kernels: two ranks (both on cuda:0, gloo -- RCCL refuses two ranks on one device) run one
    MCD step on the SAME batch; the summed gradients times 1/world and the all-reduced cross-entropy normaliser over world then
    equal the single-process quantities exactly, so every parameter and buffer matches the one-rank run bit for bit."""
    _need_gpu()
    import json
    here = os.path.dirname(os.path.abspath(__file__))
    worker = os.path.join(here, "dp_worker.py")
    out = {}
    for world in (1, 2):
        fn = str(tmp_path / ("fp%d.json" % world))
        r = _run_ranks(world, worker, [fn])
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
        out[world] = json.load(open(fn))
    assert out[1].pop("world") == 1 and out[2].pop("world") == 2
    assert out[1].keys() == out[2].keys() and len(out[1]) > 200
    for k in out[1]:
        assert out[1][k] == out[2][k], k


def test_two_ranks_bucketed_overlap_with_rank_local_arrival_orders(tmp_path):
    """ADVICE r4 (medium) through the real kernels: MCDSEG_DP_OVERLAP=1 with small buckets on two ranks whose gradients reach the
    buckets at DIFFERENT times -- rank 0 defers its weight gradients to the side stream (early deliveries through the sink), rank 1
    keeps them on the main stream (DP_SKEW).  Started "when complete", the buckets' collectives would be issued in different orders
    on the two ranks (gloo pairs them by order: wrong sums or a hang); started in bucket order they match, and -- same batch on both
    ranks -- the step equals the single-process step bit for bit."""
    _need_gpu()
    import json
    here = os.path.dirname(os.path.abspath(__file__))
    worker = os.path.join(here, "dp_worker.py")
    out = {}
    for world in (1, 2):
        fn = str(tmp_path / ("fp%d.json" % world))
        r = _run_ranks(world, worker, [fn], extra_env={"MCDSEG_DP_OVERLAP": "1", "MCDSEG_DP_BUCKET_MB": "4", "DP_SKEW": "1",
                                                       "MCDSEG_DIST_FORCE": "1"}, timeout=300)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
        out[world] = json.load(open(fn))
    assert out[1].pop("world") == 1 and out[2].pop("world") == 2
    assert out[1].pop("deferred") > 0 and out[2].pop("deferred") > 0  # (rank 0 of either run deferred: the early path was exercised)
    assert out[1].keys() == out[2].keys() and len(out[1]) > 200
    for k in out[1]:
        assert out[1][k] == out[2][k], k


def test_bench_with_two_ranks_on_one_gpu():
    """``bench.py --gpus 2`` under a launcher, both ranks on cuda:0 over gloo: the multi-rank branch of the benchmark (per-rank
    batches, barrier + MAX-over-ranks timing, rank 0's single JSON line with the whole-job rate) on the real step."""
    _need_gpu()
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = _run_ranks(2, os.path.join(root, "bench.py"), ["--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "2", "--height", "64",
                                                       "--width", "96", "--no_cpu_baseline"])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["scaling"] == "weak" and line["cpu_baseline"] is None
    assert abs(line["value"] - 2 * 2 / (line["ms_per_step"] * 1e-3)) <= 1e-2 * line["value"]
    assert line["config"]["global_pairs"] == 4 and line["config"]["parallelism"] == "dp2"


_PLAIN_BENCH = {}


@pytest.mark.parametrize("overlap", ["0", "1"])
def test_bench_with_one_rank_through_rccl(overlap):
    """``bench.py`` under ``torch.distributed.run`` with ONE rank and MCDSEG_DIST_FORCE=1: the process group is RCCL ("nccl") and every
    collective of the step -- the flat-gradient all-reduces (in one piece, or bucketed from the backward hooks with
    MCDSEG_DP_OVERLAP=1), the cross-entropy normaliser, the MAX over ranks of the timing -- really goes through it on the GPU.
    What a one-GPU box can prove of the multi-GPU path: same losses as the plain single-process run, the collectives timed, and
    (overlap on or off) the trunk's weight gradients still on the side stream."""
    _need_gpu()
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    args = ["--gpus", "1", "--steps", "2", "--warmup", "1", "--batch", "2", "--height", "64", "--width", "96", "--no_cpu_baseline",
            "--literal_steps", "1", "--batch_pool", "1"]
    r = _run_ranks(1, os.path.join(root, "bench.py"), args, extra_env={"MCDSEG_DIST_BACKEND": "nccl", "MCDSEG_DIST_FORCE": "1", "MCDSEG_DP_OVERLAP": overlap,
                                                                      "MCDSEG_DP_BUCKET_MB": "8"})
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    import subprocess
    import sys
    if "line" not in _PLAIN_BENCH:  # (the plain single-process run is the same for both parametrisations)
        plain = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, capture_output=True, text=True, timeout=600,
                               env=dict(os.environ, MCDSEG_PRETRAINED="0"))
        assert plain.returncode == 0, plain.stderr[-3000:]
        _PLAIN_BENCH["line"] = json.loads([ln for ln in plain.stdout.splitlines() if ln.startswith("{")][-1])
    ref = _PLAIN_BENCH["line"]
    assert line["n_gpus"] == 1 and line["config"]["parallelism"] == "dp1"
    assert line["config"]["c_loss"] == ref["config"]["c_loss"] and line["config"]["d_loss"] == ref["config"]["d_loss"]
    assert line["config"]["collectives"] == "rccl (forced, 1 rank)" and ref["config"]["collectives"] == "none (single process)"
    # the exchange is instrumented (HIP events around every all-reduce / around the wait for a bucketed one) ...
    coll = line["collectives"]
    assert coll is not None and ref["collectives"] is None
    assert coll["collectives_per_step"] >= 7 and coll["collective_ms_per_step"] >= 0.0 and coll["bytes_per_step"] > 5 * 4 * 26_000_000
    # ... and the bucketed exchange no longer pushes the weight gradients back onto the main stream: they stay on
    # the side stream and reach the buckets through FlatSGD's gradient sink
    assert line["wgrad_stream"]["deferred"] > 0, line["wgrad_stream"]
    # round 6: the same invocation also measures the OTHER setting of MCDSEG_DP_OVERLAP after the timed region
    other = line["dp_overlap_other_setting"]
    assert other is not None and other["MCDSEG_DP_OVERLAP"] == ("0" if overlap == "1" else "1"), other
    assert other["ms_per_step"] > 0 and other["collectives"]["collectives_per_step"] >= 7, other
    assert ref["dp_overlap_other_setting"] is None


def test_native_rccl_allreduce_one_rank():
    """``mcdseg_comm_unique_id`` / ``mcdseg_comm_init`` / ``mcdseg_allreduce`` / ``mcdseg_comm_destroy`` (include/mcdseg.h "Data
    parallelism", csrc/comm.hip): RCCL called from the C ABI on the caller's stream.  What one GPU can show: a communicator of one rank,
    the sum over one rank leaves the buffer as it is (bit for bit, stream-ordered behind the kernel that wrote it), errors are reported
    -- and ``bench.py`` with ``MCDSEG_NATIVE_RCCL=1`` runs the step's gradient exchange through it (both MCDSEG_DP_OVERLAP settings in
    the one invocation) with the losses of the plain single-process run."""
    _need_gpu()
    import ctypes
    import json
    import torch
    from mcdseg._lib import check, lib
    L = lib()
    dev = torch.device("cuda:0")
    ident = (ctypes.c_ubyte * 128)()
    check(L.mcdseg_comm_unique_id(ident), "comm_unique_id")
    assert any(ident)
    comm = ctypes.c_void_p()
    check(L.mcdseg_comm_init(ctypes.byref(comm), 1, ident, 0), "comm_init")
    s = torch.cuda.Stream(dev)
    with torch.cuda.stream(s):
        x = torch.randn(3 * 1000 * 1000 + 7, device=dev)
        ref = x.clone()
        check(L.mcdseg_allreduce(x.data_ptr(), x.numel(), comm, s.cuda_stream), "allreduce")
        check(L.mcdseg_allreduce(x.data_ptr(), 0, comm, s.cuda_stream), "allreduce of nothing")
    s.synchronize()
    assert torch.equal(x, ref)
    check(L.mcdseg_comm_destroy(comm), "comm_destroy")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    args = ["--gpus", "1", "--steps", "2", "--warmup", "1", "--batch", "2", "--height", "64", "--width", "96", "--no_cpu_baseline",
            "--literal_steps", "1", "--batch_pool", "1", "--other_configs", "", "--strict_steps", "0"]
    lines = {}
    for native in ("1", "0"):
        r = _run_ranks(1, os.path.join(root, "bench.py"), args, extra_env={"MCDSEG_DIST_BACKEND": "nccl", "MCDSEG_DIST_FORCE": "1",
                                                                          "MCDSEG_NATIVE_RCCL": native, "MCDSEG_DP_BUCKET_MB": "8"})
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
        lines[native] = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    a, b = lines["1"], lines["0"]
    assert a["config"]["collectives"].startswith("rccl through the C ABI (mcdseg_allreduce)") and b["config"]["collectives"] == "rccl (forced, 1 rank)"
    assert a["config"]["c_loss"] == b["config"]["c_loss"] and a["config"]["d_loss"] == b["config"]["d_loss"]
    assert a["collectives"]["bytes_per_step"] == b["collectives"]["bytes_per_step"] > 5 * 4 * 26_000_000
    assert a["dp_overlap_other_setting"]["collectives"]["bytes_per_step"] == a["collectives"]["bytes_per_step"]
