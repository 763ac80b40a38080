"""Development probe: drn_d_105 source-step gradients at 2 x 6 x 720 x 1280 -- uncut default vs (a) cut batches, (b) bf16x6 arithmetic, (c) f32
arithmetic: per-layer relative differences, to tell amplified rounding noise from an addressing bug in the sliced companions."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "multichannel-semseg-with-uda_amd"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
os.environ["MCDSEG_PRETRAINED"] = "0"
from recipe import fill_state_, make_batch  # noqa: E402

NC = 41


def run(math, limit):
    from loss import CrossEntropyLoss2d
    from mcdseg import ops
    from models.model_util import get_models
    ops.CONV_MATH = math
    ops.bump_weight_epoch()
    if limit:
        ops.MAX_CONV_BYTES = limit
    dev = torch.device("cuda:0")
    s, l, _ = (v.to(dev) for v in make_batch(6, 2, 6, 720, 1280, NC))
    cw = torch.ones(NC)
    cw[NC - 1] = 0
    g, f1, f2 = get_models("drn_d_105", 6, NC)
    for m, seed in ((g, 71), (f1, 72), (f2, 73)):
        fill_state_(m, seed)
        m.to(dev).train()
    feat = g(s)
    crit = CrossEntropyLoss2d(cw.to(dev))
    loss = crit(f1(feat), l) + crit(f2(feat), l)
    loss.backward()
    return float(loss.detach()), {k: p.grad.detach().clone() for k, p in g.named_parameters()}


def main():
    base_limit = None
    l0, g0 = run("f16x3", None)
    from mcdseg import ops
    keep = ops.MAX_CONV_BYTES
    out = {}
    out["cut"] = run("f16x3", 4 * 1024 * 90 * 160 * 2 - 1)
    ops.MAX_CONV_BYTES = keep
    out["bf16x6"] = run("bf16x6", None)
    out["f32"] = run("f32", None)
    keys = [k for k in g0 if k.endswith("conv1.weight") or k.endswith("conv3.weight") or k.startswith(("seg", "base.8", "base.7"))]
    print("%-34s %10s %10s %10s" % ("tensor", "cut", "bf16x6", "f32"))
    for k in keys[::3] + [k for k in keys if k.startswith(("base.6.2", "base.7", "base.8", "seg"))]:
        print("%-34s " % k + " ".join("%10.2e" % float((out[m][1][k] - g0[k]).double().norm() / (g0[k].double().norm() + 1e-30)) for m in ("cut", "bf16x6", "f32")))
    print("loss", l0, {m: out[m][0] for m in out})


if __name__ == "__main__":
    main()
