#!/usr/bin/env python3
"""Generator of the fp64 "truth" fixtures of the full-size parity tests (tests/golden/grad_truth_*.npz; test infrastructure: imports the
CPU oracle, never the product).

At BASELINE's full sizes both the HIP path and the CPU oracle compute in fp32 through 41 (drn_d_38) or 105 (drn_d_105) train-mode
BatchNorms, and at random initialisation that is worth 1-6 % on every trunk gradient WHOEVER computes it: a bound on |HIP - oracle32|
must be wide enough for two independent fp32 errors, and hides a systematic error of one layer behind them.  The yardstick that does
not: the oracle run a second time in fp64 on the same parameters and batch is the truth; the fp32 oracle's own distance from it says what
fp32 costs; the HIP path has to stay within a small multiple of THAT distance of the truth -- per tensor.  The fp64 gradients themselves
are hundreds of megabytes, so a fixture keeps of each gradient tensor

    <name>/n64   its L2 norm (fp64 oracle)
    <name>/d32   || oracle fp32 - oracle fp64 ||
    <name>/x64   the fp64 tensor itself when it has at most EXACT elements, else
    <name>/s64   a count-sketch of it: K buckets, element i goes to bucket h(i) with sign s(i), h and s drawn from a generator seeded by
                 (name, numel) -- linear, so sketch(a) - sketch(b) = sketch(a - b), and || sketch(d) ||^2 is an unbiased estimate of
                 || d ||^2 with relative standard deviation sqrt(2 / K) (K = 512: 6 % of the square, 3 % of the norm)

and of the forward outputs (encoder features, one head's logits) a fixed sub-sample of the fp64 values with the fp32 oracle's largest error.
The recipes (networks, seeds, batches) are those of tests/test_model_gpu.py's full-size tests.

    python tests/golden/make_grad_truth.py [--out DIR] cfg2 cfg3 cfg4 cfg5n2 cfg5n8

Run on the GPU box's host (128 cores: cfg2 takes 40 s in fp32 + 130 s in fp64); the fixtures are committed with this script.
"""
import argparse
import os
import sys
import time
import zlib

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (HERE, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from recipe import fill_state_, make_batch  # noqa: E402

NC = 41
K = 512       # sketch buckets
EXACT = 512   # tensors up to this many elements are kept whole


def physical_cores():
    cores, pid = set(), None
    try:
        for line in open("/proc/cpuinfo"):
            k, _, v = line.partition(":")
            if k.strip() == "physical id":
                pid = v.strip()
            elif k.strip() == "core id":
                cores.add((pid, v.strip()))
    except OSError:
        pass
    return len(cores) or os.cpu_count() or 1


def sketch(name, t):
    """count-sketch of a tensor (any device; fp64 result on the CPU), deterministic in (name, numel)"""
    flat = t.detach().double().cpu().reshape(-1)
    g = torch.Generator().manual_seed(zlib.crc32(("%s/%d" % (name, flat.numel())).encode()))
    idx = torch.randint(0, K, (flat.numel(),), generator=g)
    sign = torch.randint(0, 2, (flat.numel(),), generator=g).double() * 2 - 1
    return torch.zeros(K, dtype=torch.float64).index_add_(0, idx, flat * sign)


def record(out, name, g32, g64):
    g32, g64 = g32.detach().double().cpu(), g64.detach().double().cpu()
    out[name + "/n64"] = np.float64(float(g64.norm()))
    out[name + "/d32"] = np.float64(float((g32 - g64).norm()))
    if g64.numel() <= EXACT:
        out[name + "/x64"] = g64.numpy().astype(np.float32)  # (the fp64 values rounded once: 6e-8, far below the distances measured)
    else:
        out[name + "/s64"] = sketch(name, g64).numpy().astype(np.float32)  # (differences of interest are 1e-2 of a bucket: fp32 keeps them)


def record_output(out, name, o32, o64, stride):
    """a forward output: the fp32 oracle's largest error on every ``stride``-th pixel (``e32``, ``e32_stride``) and the fp64 values on a
    sub-sample of those pixels thin enough to commit (at most 100 000 values, kept as fp32: ``sub64``, ``stride``) -- the HIP path's
    largest error on the thinner set is held against the oracle's on the denser one"""
    a, b = o32.detach().double()[:, :, ::stride, ::stride], o64.detach().double()[:, :, ::stride, ::stride]
    out[name + "/e32"] = np.float64(float((a - b).abs().max()))
    out[name + "/e32_stride"] = np.int64(stride)
    out[name + "/scale"] = np.float64(float(b.abs().max()))
    f = 1
    while b[:, :, ::f, ::f].numel() > 100000 and f < 8:
        f *= 2
    out[name + "/sub64"] = b[:, :, ::f, ::f].contiguous().numpy().astype(np.float32)
    out[name + "/stride"] = np.int64(stride * f)


def run_mcd(net, seeds, n, h, w, batch_seed, double):
    """source cross-entropy pass of adapt_trainer.py:163-185 on the oracle: (features, logits of F1, {name: gradient})"""
    from oracle import ref_loss, ref_models
    og, of1, of2 = ref_models.get_models(net, 6, NC)
    for m, s in zip((og, of1, of2), seeds):
        fill_state_(m, s)
    src, lbl, _ = make_batch(batch_seed, n, 6, h, w, NC)
    cw = ref_loss.class_weights(NC)
    if double:
        og, of1, of2, src, cw = og.double(), of1.double(), of2.double(), src.double(), cw.double()
    og.train(), of1.train(), of2.train()
    feat = og(src)
    crit = ref_loss.CrossEntropyLoss2d(cw)
    logits = of1(feat)
    (crit(logits, lbl) + crit(of2(feat), lbl)).backward()
    gs = {"g." + k: v.grad for k, v in og.named_parameters()}
    gs.update({"f%d.%s" % (i + 1, k): v.grad for i, m in enumerate((of1, of2)) for k, v in m.named_parameters()})
    return feat.detach(), logits.detach(), gs


def run_mfnet(n, double):
    """tests/test_model_gpu.py::test_cfg3_full_batch_vs_oracle's pass"""
    from oracle import ref_loss, ref_models
    ora = ref_models.get_models("drn_d_38", 6, NC, method="MFNet-ScoreAddFusion")
    for i, m in enumerate(ora):
        fill_state_(m, 51 + i)
    src, lbl, _ = make_batch(79, n, 6, 480, 640, NC)
    cw = ref_loss.class_weights(NC)
    if double:
        ora = [m.double() for m in ora]
        src, cw = src.double(), cw.double()
    for m in ora:
        m.train()
    ra = ora[0](src[:, :3].contiguous())
    rb = ora[1](src[:, 3:].contiguous())
    ra.retain_grad(), rb.retain_grad()
    o1, o2 = ora[2](ra, rb), ora[3](ra, rb)
    crit = ref_loss.CrossEntropyLoss2d(cw)
    l1, l2 = crit(o1, lbl), crit(o2, lbl)
    (l1 + l2).backward()
    gs = {"%d.%s" % (i, k): v.grad for i in range(4) for k, v in ora[i].named_parameters()}
    gs["d/d(RGB score map)"], gs["d/d(HHA score map)"] = ra.grad, rb.grad
    return (ra.detach(), rb.detach(), o1.detach()), (float(l1.detach()), float(l2.detach())), gs


def run_multitask(n, double):
    """tests/test_model_gpu.py::test_cfg4_full_batch_vs_oracle's pass"""
    from oracle import ref_loss, ref_multitask
    cw = ref_loss.class_weights(NC)
    if double:
        cw = cw.double()
    renc, rdec = ref_multitask.get_multitask_models("drn_d_38", 6, NC, ref_loss.CrossEntropyLoss2d(cw), ref_loss.Diff2d())
    fill_state_(renc, 81), fill_state_(rdec, 82)
    src, lbl, _ = make_batch(80, n, 6, 480, 640, NC)
    if double:
        renc, rdec, src = renc.double(), rdec.double(), src.double()
    renc.train(), rdec.train()
    rgb, dep = src[:, :3].contiguous(), src[:, 3:].contiguous()
    fet = renc(rgb)
    fet.retain_grad()
    loss = rdec.get_loss(fet, lbl, dep)
    loss.backward()
    gs = {"enc." + k: v.grad for k, v in renc.named_parameters()}
    gs.update({"dec." + k: v.grad for k, v in rdec.named_parameters() if v.grad is not None})
    gs["d/d(encoder features)"] = fet.grad
    return fet.detach(), float(loss.detach()), gs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("configs", nargs="+", choices=["cfg2", "cfg3", "cfg4", "cfg5n2", "cfg5n8"])
    ap.add_argument("--out", default=HERE)
    ap.add_argument("--tiny", action="store_true", help="(smoke run of this script: 1 x 6 x 64 x 96 instead of the stated sizes)")
    args = ap.parse_args()
    if args.tiny:
        global make_batch
        full = make_batch
        make_batch = lambda seed, n, ch, h, w, nc: full(seed, 1, ch, 64, 96, nc)  # noqa: E731
    torch.set_num_threads(physical_cores())
    os.makedirs(args.out, exist_ok=True)
    for cfg in args.configs:
        t0 = time.time()
        out = {}
        if cfg in ("cfg2", "cfg5n2", "cfg5n8"):
            net, seeds, n, h, w, bs = {"cfg2": ("drn_d_38", (11, 12, 13), 16, 480, 640, 78), "cfg5n2": ("drn_d_105", (71, 72, 73), 2, 720, 1280, 78),
                                       "cfg5n8": ("drn_d_105", (71, 72, 73), 8, 720, 1280, 77)}[cfg]
            f32, l32, g32 = run_mcd(net, seeds, n, h, w, bs, False)
            t1 = time.time()
            f64, l64, g64 = run_mcd(net, seeds, n, h, w, bs, True)
            record_output(out, "feat", f32, f64, 4)
            record_output(out, "logits1", l32, l64, 32)
            out["recipe"] = np.array("%s seeds %s, make_batch(%d, %d, 6, %d, %d, %d): CE(F1(G(x))) + CE(F2(G(x))) backward" % (net, seeds, bs, n, h, w, NC))
        elif cfg == "cfg3":
            (a32, b32, o32), ls32, g32 = run_mfnet(16, False)
            t1 = time.time()
            (a64, b64, o64), ls64, g64 = run_mfnet(16, True)
            record_output(out, "score_rgb", a32, a64, 4)
            record_output(out, "score_hha", b32, b64, 4)
            record_output(out, "logits1", o32, o64, 32)
            out["losses64"], out["losses32"] = np.array(ls64), np.array(ls32)
            out["recipe"] = np.array("MFNet-ScoreAddFusion drn_d_38 seeds 51.., make_batch(79, 16, 6, 480, 640, %d)" % NC)
        else:
            f32, ls32, g32 = run_multitask(8, False)
            t1 = time.time()
            f64, ls64, g64 = run_multitask(8, True)
            record_output(out, "feat", f32, f64, 4)
            out["loss64"], out["loss32"] = np.float64(ls64), np.float64(ls32)
            out["recipe"] = np.array("multitask drn_d_38 seeds 81, 82, make_batch(80, 8, 6, 480, 640, %d): get_loss backward" % NC)
        assert set(g32) == set(g64)
        for k in sorted(g64):
            record(out, k, g32[k], g64[k])
        out["names"] = np.array(sorted(g64))
        path = os.path.join(args.out, "grad_truth_%s.npz" % cfg)
        np.savez_compressed(path, **out)
        num = sum(float(out[k + "/d32"]) ** 2 for k in g64)
        den = sum(float(out[k + "/n64"]) ** 2 for k in g64)
        worst = max((float(out[k + "/d32"]) / max(float(out[k + "/n64"]), 1e-300), k) for k in g64)
        print("%s: %d tensors, oracle fp32 %.0f s, fp64 %.0f s; oracle32 - fp64 over all tensors %.3e, worst %.3e (%s); %s %.2f MB"
              % (cfg, len(g64), t1 - t0, time.time() - t1, (num / den) ** 0.5, worst[0], worst[1], path, os.path.getsize(path) / 1e6), flush=True)
        del g32, g64, out


if __name__ == "__main__":
    main()
