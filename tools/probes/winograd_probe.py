#!/usr/bin/env python3
"""Probe, on the GPU: what would Winograd F(2x2, 3x3) buy on the stride-1 3x3 layers of L5-L8?

F(2x2, 3x3) turns one 3x3 convolution over P pixels into 16 independent GEMMs [Cout x Cin] x [Cin x P/4] (one per position of the
4 x 4 transformed tile; a dilated layer is d^2 interleaved dense problems: the same counts): 16/4 = 4 multiplications per output
and input channel where the direct form has 9.  What the GEMM side of that costs is measured EXACTLY by a kernel that exists: the
1x1 convolution Cin -> Cout over 4 P pixels -- the same FLOPs (4 P Cin Cout MACs), the same operand stream (a companion of 4 P
pixels = the 16 transformed planes, 16 B per input element), the same output stream (4 P Cout fp32 = the 16 planes of M) -- run by the
benchmark's own ping-pong forward kernel.  The transforms on either side are HBM-bound passes priced at the rate the BatchNorm apply
kernels reach on the same tensors (measured here):
  input  : the BN-apply producer writes 16 B per element of transformed companion where it writes 4 B now  (+12 B per element);
  output : a pass that reads the 16 planes (16 B per output element), forms z = A^T M A and the BatchNorm statistics, writes z (4 B).
Accuracy is tools/probes/winograd_numerics.py's subject (CPU).

    python tools/probes/winograd_probe.py
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


def conv_ms(n, cin, cout, k, dil, h, w, dgrad=False):
    dev = torch.device("cuda:0")
    x = torch.randn(n, cin, h, w, device=dev)
    wt = torch.randn(cout, cin, k, k, device=dev) * 0.05
    desc = ops.conv_desc(x.shape, wt.shape, 1, dil * (k // 2), dil)
    pk = ops.PackedWeights()
    wf, wd, mpf = pk.get(wt, desc)
    xb = ops._bound_or_measure(x, None)
    if dgrad:
        gy = torch.randn(n, cout, h, w, device=dev)
        gb = ops._bound_or_measure(gy, None)
        gy_cb, _ = ops.split_companion(gy, gb)
        return timeit(lambda: ops._conv_dgrad(desc, gy, wd, gy_cb, gb, pk.w_bound))
    x_cb, _ = ops.split_companion(x, xb)
    return timeit(lambda: ops._conv_fprop(desc, x, wf, None, True, mpf, x_cb, xb, pk.w_bound))


def main():
    dev = torch.device("cuda:0")
    n, h, w = 16, 60, 80
    # the HBM rate of a producer pass on these tensors: split_cb reads 4 B and writes 4 B per element
    rates = {}
    for c in (256, 512):
        x = torch.randn(n, c, h, w, device=dev)
        b = ops._bound_or_measure(x, None)
        t = timeit(lambda: ops.split_companion(x, b))
        rates[c] = 8.0 * x.numel() / t / 1e6  # GB/s
    print("producer-pass rate (split_cb, 4 B read + 4 B written per element): %s GB/s" % {k: round(v) for k, v in rates.items()})
    print("%-22s %9s %9s | %9s %9s %9s %9s | %7s" % ("layer (N=16, 60x80)", "direct", "dgrad", "GEMM x16", "in +12B", "out 20B", "Winograd", "speedup"))
    for name, c, dil in (("256 -> 256 3x3 d2", 256, 2), ("512 -> 512 3x3 d4", 512, 4), ("512 -> 512 3x3 d1", 512, 1)):
        t_dir = conv_ms(n, c, c, 3, dil, h, w)
        t_dg = conv_ms(n, c, c, 3, dil, h, w, dgrad=True)
        t_gemm = conv_ms(n, c, c, 1, 1, 2 * h, 2 * w)          # 4 P pixels: the 16 planes of P / 4 tiles
        elems = n * c * h * w
        t_in = 12.0 * elems / rates[c] / 1e6                   # ms
        t_out = 20.0 * elems / rates[c] / 1e6
        t_w = t_gemm + t_in + t_out
        print("%-22s %9.3f %9.3f | %9.3f %9.3f %9.3f %9.3f | %6.2fx" % (name, t_dir, t_dg, t_gemm, t_in, t_out, t_w, t_dir / t_w))


if __name__ == "__main__":
    main()
