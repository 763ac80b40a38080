"""Test infrastructure: the fp64 "truth" fixtures of the full-size parity tests (tests/golden/grad_truth_*.npz, written by
tests/golden/make_grad_truth.py from the CPU oracle in fp32 and fp64) and the yardstick they carry: how far the fp32 ORACLE sits from the
truth, per tensor.  ``distances`` measures the HIP path against the same truth; the tests bound the ratio."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(cfg):
    path = os.path.join(GOLDEN, "grad_truth_%s.npz" % cfg)
    if not os.path.exists(path):
        raise FileNotFoundError("%s is missing: run tests/golden/make_grad_truth.py %s (on a host with many cores) and commit the fixture" % (path, cfg))
    return np.load(path)


def distances(fx, grads):
    """{name: (|HIP - fp64|, |oracle32 - fp64|, |fp64|)} for every gradient tensor of the fixture; ``grads``: {name: tensor}.  Tensors
    kept whole in the fixture are compared exactly, the others through their count-sketch (an unbiased estimate of the distance, relative
    standard deviation 3 % at 512 buckets; make_grad_truth.py)"""
    from make_grad_truth import sketch  # (tests/golden is on sys.path: conftest.py)
    out = {}
    names = [str(n) for n in fx["names"]]
    missing = [n for n in names if n not in grads]
    assert not missing, "gradients missing on the HIP side: %s" % missing[:5]
    for name in names:
        g = grads[name].detach().double().cpu()
        if name + "/x64" in fx.files:
            d = float((g - torch.from_numpy(fx[name + "/x64"]).double().reshape(g.shape)).norm())
        else:
            d = float((sketch(name, g) - torch.from_numpy(fx[name + "/s64"]).double()).norm())
        out[name] = (d, float(fx[name + "/d32"]), float(fx[name + "/n64"]))
    return out


def summary(dist, floor=2e-5):
    """(overall ratio, worst per-tensor ratio, its name, overall HIP distance, overall oracle32 distance) over the tensors that carry
    signal.  A tensor's yardstick is max(|oracle32 - fp64|, floor * |fp64|): where the fp32 oracle is closer to the truth than the
    kernels' own accuracy bound (2e-5 of the scale: the up-sampling kernels, whose gradients never pass the trunk's BatchNorms) that
    bound is the yardstick.  Tensors that are rounding noise on BOTH sides (a bias in front of a train-mode BatchNorm: zero gradient up
    to rounding) are left out of the ratios and returned separately as (name, |HIP|, |oracle32 - fp64|, |fp64|)."""
    total = sum(n * n for _, _, n in dist.values()) ** 0.5
    num = den = 0.0
    worst = (0.0, None)
    noise = []
    for name, (d, d32, n64) in dist.items():
        if n64 <= 1e-9 * total:
            noise.append((name, d, d32, n64))
            continue
        yard = max(d32, floor * n64)
        num, den = num + d * d, den + yard * yard
        worst = max(worst, (d / yard, name))
    return (num / den) ** 0.5, worst[0], worst[1], num ** 0.5 / total, den ** 0.5 / total, noise


def output_error(fx, name, tensor):
    """(max |HIP - fp64| on the fixture's sub-sample, the fp32 oracle's own, the scale) of a forward output"""
    s = int(fx[name + "/stride"])
    ref = torch.from_numpy(fx[name + "/sub64"]).double()
    got = tensor.detach()[:, :, ::s, ::s].double().cpu()
    assert got.shape == ref.shape, (name, got.shape, ref.shape)
    return float((got - ref).abs().max()), float(fx[name + "/e32"]), float(fx[name + "/scale"])


# ---- the HIP side of the fixtures' recipes (the same networks, seeds and batches as make_grad_truth.py's oracle runs)
NC = 41


def hip_mcd(net, seeds, n, h, w, batch_seed, dev):
    """source cross-entropy pass (adapt_trainer.py:163-185) through the HIP kernels: ({output name: tensor}, {gradient name: tensor})"""
    from loss import CrossEntropyLoss2d
    from models.model_util import get_models
    from recipe import fill_state_, make_batch
    g, f1, f2 = get_models(net, 6, NC)
    for m, s in zip((g, f1, f2), seeds):
        fill_state_(m, s)
        m.to(dev).train()
    src, lbl, _ = make_batch(batch_seed, n, 6, h, w, NC)
    cw = torch.ones(NC)
    cw[NC - 1] = 0
    crit = CrossEntropyLoss2d(cw.to(dev))
    feat = g(src.to(dev))
    logits = f1(feat)
    (crit(logits, lbl.to(dev)) + crit(f2(feat), lbl.to(dev))).backward()
    torch.cuda.synchronize()
    gs = {"g." + k: v.grad for k, v in g.named_parameters()}
    gs.update({"f%d.%s" % (i + 1, k): v.grad for i, m in enumerate((f1, f2)) for k, v in m.named_parameters()})
    return {"feat": feat.detach(), "logits1": logits.detach()}, gs


def hip_mfnet(n, dev):
    from loss import CrossEntropyLoss2d
    from models.model_util import get_models
    from recipe import fill_state_, make_batch
    hip = get_models("drn_d_38", 6, NC, method="MFNet-ScoreAddFusion")
    for i, m in enumerate(hip):
        fill_state_(m, 51 + i)
        m.to(dev).train()
    src, lbl, _ = make_batch(79, n, 6, 480, 640, NC)
    cw = torch.ones(NC)
    cw[NC - 1] = 0
    s = src.to(dev)
    a, b = hip[0](s[:, :3].contiguous()), hip[1](s[:, 3:].contiguous())
    a.retain_grad(), b.retain_grad()
    o1, o2 = hip[2](a, b), hip[3](a, b)
    crit = CrossEntropyLoss2d(cw.to(dev))
    l1, l2 = crit(o1, lbl.to(dev)), crit(o2, lbl.to(dev))
    (l1 + l2).backward()
    torch.cuda.synchronize()
    gs = {"%d.%s" % (i, k): v.grad for i in range(4) for k, v in hip[i].named_parameters()}
    gs["d/d(RGB score map)"], gs["d/d(HHA score map)"] = a.grad, b.grad
    return {"score_rgb": a.detach(), "score_hha": b.detach(), "logits1": o1.detach(), "losses": (float(l1.detach()), float(l2.detach()))}, gs


def hip_multitask(n, dev):
    from loss import CrossEntropyLoss2d, Diff2d
    from models.model_util import get_multitask_models
    from recipe import fill_state_, make_batch
    cw = torch.ones(NC)
    cw[NC - 1] = 0
    enc, dec = get_multitask_models("drn_d_38", 6, NC, CrossEntropyLoss2d(cw), Diff2d())
    fill_state_(enc, 81), fill_state_(dec, 82)
    enc.to(dev).train(), dec.to(dev).train()
    src, lbl, _ = make_batch(80, n, 6, 480, 640, NC)
    rgb, dep = src[:, :3].contiguous(), src[:, 3:].contiguous()
    fet = enc(rgb.to(dev))
    fet.retain_grad()
    loss = dec.get_loss(fet, lbl.to(dev), dep.to(dev))
    loss.backward()
    torch.cuda.synchronize()
    gs = {"enc." + k: v.grad for k, v in enc.named_parameters()}
    gs.update({"dec." + k: v.grad for k, v in dec.named_parameters() if v.grad is not None})
    gs["d/d(encoder features)"] = fet.grad
    return {"feat": fet.detach(), "loss": float(loss.detach())}, gs


RECIPES = {"cfg2": ("drn_d_38", (11, 12, 13), 16, 480, 640, 78), "cfg5n2": ("drn_d_105", (71, 72, 73), 2, 720, 1280, 78),
           "cfg5n8": ("drn_d_105", (71, 72, 73), 8, 720, 1280, 77)}


def hip_run(cfg, dev):
    if cfg in RECIPES:
        return hip_mcd(*RECIPES[cfg], dev)
    return hip_mfnet(16, dev) if cfg == "cfg3" else hip_multitask(8, dev)


def report(cfg, dev, fx=None):
    """one line per question: the HIP path's distance from the truth in units of the fp32 oracle's own"""
    fx = fx if fx is not None else load(cfg)
    outs, gs = hip_run(cfg, dev)
    lines = []
    for name, t in outs.items():
        if torch.is_tensor(t) and name + "/sub64" in fx.files:
            e, e32, sc = output_error(fx, name, t)
            lines.append("%s %-10s max |HIP - fp64| %.3e  oracle32 %.3e  (scale %.3e): %.2fx" % (cfg, name, e, e32, sc, e / max(e32, 1e-30)))
    dist = distances(fx, gs)
    overall, worst, wname, dh, d32, noise = summary(dist)
    ratios = sorted(d / max(d32_, 2e-5 * n64) for d, d32_, n64 in dist.values() if n64 > 0)
    q = lambda f: ratios[min(len(ratios) - 1, int(f * len(ratios)))]  # noqa: E731
    lines.append("%s gradients: %d tensors, HIP - fp64 %.3e, oracle32 - fp64 %.3e of the overall norm: %.2fx; per tensor median %.2fx, 90 %% %.2fx, "
                 "worst %.2fx (%s); %d noise-only tensors" % (cfg, len(dist), dh, d32, overall, q(0.5), q(0.9), worst, wname, len(noise)))
    top = sorted(((d / max(d32_, 2e-5 * n64), k, d, d32_, n64) for k, (d, d32_, n64) in dist.items() if n64 > 0), reverse=True)[:6]
    for r, k, d, d32_, n64 in top:
        lines.append("    %-40s %.2fx   HIP %.3e  oracle32 %.3e  of its norm" % (k, r, d / n64, d32_ / n64))
    return lines


if __name__ == "__main__":
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    root = os.path.dirname(here)
    for p in (os.path.join(here, "golden"), root, os.path.join(root, "multichannel-semseg-with-uda_amd")):
        sys.path.insert(0, p)
    os.environ.setdefault("MCDSEG_PRETRAINED", "0")
    import mcdseg
    from mcdseg import ops
    for cfg in sys.argv[1:]:
        with_opts = {}
        if cfg.startswith("cfg5"):
            ops.ACT_STORAGE = "compact"
            if cfg == "cfg5n2":
                with_opts = dict(PP_CUS=16)  # 256 CUs at N = 32: the same rounds of tiles, hence the same launch plan
                ops.MAX_CONV_BYTES = 150 << 20
        with mcdseg.options(**with_opts):
            for line in report(cfg, torch.device("cuda:0")):
                print(line, flush=True)
        ops.ACT_STORAGE, ops.MAX_CONV_BYTES = "fp32", (1 << 31) - (1 << 26)
        torch.cuda.empty_cache()
