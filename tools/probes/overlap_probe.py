#!/usr/bin/env python3
"""Does an HBM-bound pass hide behind a power-metered MFMA kernel when both are in flight on two HIP streams?
Times K launches of the 512 -> 512 d4 forward (pre-split operands), K' launches of bn_apply on a tensor of the same size, and both
together on two streams.  Development probe."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "multichannel-semseg-with-uda_amd"))
import torch  # noqa: E402

from mcdseg import ops  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    n, c, h, w, d = 16, 512, 60, 80, 4
    x = torch.randn(n, c, h, w, device=dev)
    wt = torch.randn(c, c, 3, 3, device=dev) * 0.05
    desc = ops.conv_desc(x.shape, wt.shape, 1, d, d)
    pk = ops.PackedWeights()
    wf, wd, mpf = pk.get(wt, desc)
    xb = ops._bound_or_measure(x, None)
    wb = pk.w_bound
    (x_cb, _) = ops.split_companion(x, xb)
    L = ops.lib()
    mean = torch.zeros(c, device=dev)
    rstd = torch.ones(c, device=dev)
    gamma = torch.ones(c, device=dev)
    beta = torch.zeros(c, device=dev)
    z = torch.randn(n, c, h, w, device=dev)
    out = torch.empty_like(z)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

    def conv():
        ops._conv_fprop(desc, x, wf, None, True, mpf, x_cb, xb, wb)

    def bn():
        ops.check(L.mcdseg_bn_apply(ops._p(z), ops._p(mean), ops._p(rstd), ops._p(gamma), ops._p(beta), None, ops._p(out), n, c, h * w, 1,
                                    ops._stream()), "bn_apply")

    def run(kc, kb):
        torch.cuda.synchronize()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        s1.wait_stream(torch.cuda.current_stream())
        s2.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s1):
            for _ in range(kc):
                conv()
        with torch.cuda.stream(s2):
            for _ in range(kb):
                bn()
        torch.cuda.current_stream().wait_stream(s1)
        torch.cuda.current_stream().wait_stream(s2)
        t1.record()
        torch.cuda.synchronize()
        return t0.elapsed_time(t1)

    for _ in range(2):
        run(3, 3)
    k = 20
    tc = run(k, 0)
    tb = run(0, k)
    print("conv alone   %7.3f ms / launch" % (tc / k))
    print("bn alone     %7.3f ms / launch" % (tb / k))
    for kb in (k, 2 * k, 4 * k, 8 * k):
        t = run(k, kb)
        print("conv x%d + bn x%d on two streams: %7.2f ms  (serial sum %7.2f, max %7.2f)" % (k, kb, t, tc + tb * kb / k, max(tc, tb * kb / k)))


if __name__ == "__main__":
    main()
