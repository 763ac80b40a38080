#!/usr/bin/env python3
"""Probe, on the GPU: the fused up-sample + loss kernel and the up-sampler's backward at the benchmark's shape (N = 16, 41 classes,
60 x 80 scores -> 480 x 640 logits), each alone: LDS-DMA form against the register-staged form (MCDSEG_UP8_LOSS_DMA), with and
without the gradient stores.

    python tools/probes/up8_loss_probe.py
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "multichannel-semseg-with-uda_amd"))
import torch  # noqa: E402

from mcdseg import ops  # noqa: E402


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(reps):
        fn()
    t1.record()
    torch.cuda.synchronize()
    return t0.elapsed_time(t1) / reps


def main():
    dev = torch.device("cuda:0")
    n, c, hi, wi = 16, 41, 60, 80
    g = torch.Generator().manual_seed(3)
    s1 = (2 * torch.randn(n, c, hi, wi, generator=g)).to(dev)
    s2 = (2 * torch.randn(n, c, hi, wi, generator=g)).to(dev)
    w1 = (torch.randn(c, 1, 16, 16, generator=g) * 0.2).to(dev)
    w2 = (torch.randn(c, 1, 16, 16, generator=g) * 0.2).to(dev)
    lab = torch.randint(0, c, (n, 8 * hi, 8 * wi), generator=g).to(dev)
    cw = torch.ones(c, device=dev)
    wsum = ops.ce_normaliser(lab, cw, c, -100)
    gb = 2 * 4 * n * c * 64 * hi * wi / 1e9
    for dma in ("0", "1"):
        os.environ["MCDSEG_UP8_LOSS_DMA"] = dma
        for name, kw in (("CE + Diff, gradients", dict(labels=lab, ce=1.0, diff=-1.0, grad=True)),
                         ("Diff, gradients", dict(labels=None, ce=0.0, diff=1.0, grad=True)),
                         ("CE + Diff, values only", dict(labels=lab, ce=1.0, diff=-1.0, grad=False))):
            t = timeit(lambda: ops.up8_mcd_losses(s1, w1, s2, w2, kw["labels"], cw if kw["labels"] is not None else None, ce_coef=kw["ce"],
                                                  diff_coef=kw["diff"], want_grad=kw["grad"], wsum=wsum if kw["labels"] is not None else None))
            print("loss kernel, %-9s %-24s %.3f ms%s" % ("LDS-DMA" if dma == "1" else "registers", name, t,
                                                         "  (%.0f GB/s of gradient stores)" % (gb / t * 1e3) if kw["grad"] else ""))
    gy = torch.randn(n, c, 8 * hi, 8 * wi, device=dev)
    for name, dx, dw in (("dx", True, False), ("dw", False, True), ("dx + dw", True, True)):
        t = timeit(lambda: ops._up8_bwd(gy, w1, s1, dx, dw))
        print("up8 backward %-8s %.3f ms  (%.0f GB/s)" % (name, t, gb / 2 / t * 1e3))


if __name__ == "__main__":
    main()
