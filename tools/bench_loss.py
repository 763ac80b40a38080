#!/usr/bin/env python3
"""Times the loss kernels at the benchmark shape (N=16, 41 classes, 60x80 scores -> 480x640): the fused up-sampler + loss
kernel against up8_fwd x 2 + the plain loss kernel.  Development tool."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multichannel-semseg-with-uda_amd"))
import torch  # noqa: E402

from mcdseg import ops  # noqa: E402


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    dev = torch.device("cuda:0")
    n, c, hi, wi = 16, int(os.environ.get("NC", "41")), 60, 80
    g = torch.Generator().manual_seed(0)
    s = torch.randn(n, c, hi, wi, generator=g).to(dev)
    w1 = (torch.randn(c, 1, 16, 16, generator=g) * 0.1).to(dev)
    w2 = (torch.randn(c, 1, 16, 16, generator=g) * 0.1).to(dev)
    lab = torch.randint(0, c, (n, 8 * hi, 8 * wi), generator=g).to(dev)
    cw = torch.ones(c, device=dev)
    for name, labels, kw in (("CE+CE", lab, dict(ce_coef=1.0)), ("Diff", None, dict(diff_coef=1.0))):
        t_f = timed(lambda: ops.up8_mcd_losses(s, w1, s, w2, labels, cw if labels is not None else None, **kw))
        t_2 = timed(lambda: ops.mcd_losses(ops.up8(s, w1), ops.up8(s, w2), labels, cw if labels is not None else None, **kw))
        t_v = timed(lambda: ops.up8_mcd_losses(s, w1, s, w2, labels, cw if labels is not None else None, want_grad=False, **kw))
        print("%-6s fused %.3f ms (values only, no gradient stores: %.3f ms)   two-pass %.3f ms" % (name, t_f, t_v, t_2), flush=True)


if __name__ == "__main__":
    main()
