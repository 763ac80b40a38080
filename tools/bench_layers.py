#!/usr/bin/env python3
"""Per-layer micro-benchmark of the HIP kernels on the BASELINE cfg2 shapes (N=16, 6x480x640, drn_d_38).
Prints achieved TFLOP/s (algorithmic 2*MACs) for fprop / dgrad / wgrad of every distinct conv geometry and
achieved algorithmic GB/s for the streaming kernels.  Development tool; bench.py is the judged benchmark."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multichannel-semseg-with-uda_amd"))
import torch  # noqa: E402

from mcdseg import ops  # noqa: E402

# (name, Cin, Cout, k, stride, dil, H, W, count per forward)
LAYERS = [
    ("L0 7x7 6->16", 6, 16, 7, 1, 1, 480, 640, 1),
    ("L1 16->16", 16, 16, 3, 1, 1, 480, 640, 1),
    ("L2 16->32 s2", 16, 32, 3, 2, 1, 480, 640, 1),
    ("L3 32->64 s2", 32, 64, 3, 2, 1, 240, 320, 1),
    ("L3 64->64", 64, 64, 3, 1, 1, 120, 160, 5),
    ("L3 ds 1x1 s2", 32, 64, 1, 2, 1, 240, 320, 1),
    ("L4 64->128 s2", 64, 128, 3, 2, 1, 120, 160, 1),
    ("L4 128->128", 128, 128, 3, 1, 1, 60, 80, 7),
    ("L5 128->256 d2", 128, 256, 3, 1, 2, 60, 80, 1),
    ("L5 256->256 d2", 256, 256, 3, 1, 2, 60, 80, 11),
    ("L5 ds 1x1", 128, 256, 1, 1, 1, 60, 80, 1),
    ("L6 256->512 d4", 256, 512, 3, 1, 4, 60, 80, 1),
    ("L6 512->512 d4", 512, 512, 3, 1, 4, 60, 80, 5),
    ("L7 512->512 d2", 512, 512, 3, 1, 2, 60, 80, 1),
    ("L8 512->512 d1", 512, 512, 3, 1, 1, 60, 80, 1),
    ("seg 512->41", 512, 41, 1, 1, 1, 60, 80, 1),
]


def timeit(fn, reps):
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
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--only", default="")
    ap.add_argument("--math", default=None, help="f32 | bf16x6 | f16x3")
    ap.add_argument("--hw", type=int, nargs=2, default=None, help="override the feature-map size of every selected layer (tile-count experiments)")
    args = ap.parse_args()
    if args.math:
        ops.CONV_MATH = args.math
    dev = torch.device("cuda:0")
    n = args.batch
    tot = {"fprop": 0.0, "dgrad": 0.0, "wgrad": 0.0}
    print("%-18s %8s | %9s %7s | %9s %7s | %9s %7s" % ("layer", "GFLOP", "fprop ms", "TF", "dgrad ms", "TF", "wgrad ms", "TF"))
    for name, cin, cout, k, s, d, h, w, cnt in LAYERS:
        if args.only and args.only not in name:
            continue
        pad = d * (k // 2)
        if args.hw:
            h, w = args.hw
        x = torch.randn(n, cin, h, w, device=dev)
        wt = torch.randn(cout, cin, k, k, device=dev) * 0.05
        desc = ops.conv_desc(x.shape, wt.shape, s, pad, d)
        pk = ops.PackedWeights()
        wf, wd, mpf = pk.get(wt, desc)
        gy = torch.randn(n, cout, desc.Ho, desc.Wo, device=dev)
        gf = 2.0 * n * desc.Ho * desc.Wo * cout * cin * k * k / 1e9
        xb = ops._bound_or_measure(x, None)
        gb = ops._bound_or_measure(gy, None)
        wb = pk.w_bound
        tf = timeit(lambda: ops._conv_fprop(desc, x, wf, None, True, mpf, None, xb, wb), args.reps)
        td = timeit(lambda: ops._conv_dgrad(desc, gy, wd, None, gb, wb), args.reps)
        tw = timeit(lambda: ops._conv_wgrad(desc, x, gy, None, None, xb, gb), args.reps)
        extra = ""
        if ops.CONV_MATH in ops.MATH_ID and min(cin, cout) >= 16 and cin % 8 == 0 and cout % 8 == 0:
            (x_cb, _), (gy_cb, _) = ops.split_companion(x, xb), ops.split_companion(gy, gb)
            tf2 = timeit(lambda: ops._conv_fprop(desc, x, wf, None, True, mpf, x_cb, xb, wb), args.reps)
            td2 = timeit(lambda: ops._conv_dgrad(desc, gy, wd, gy_cb, gb, wb), args.reps)
            tw2 = timeit(lambda: ops._conv_wgrad(desc, x, gy, x_cb, gy_cb, xb, gb), args.reps)
            extra = " | pre-split: fprop %.3f ms %.1f TF, dgrad %.3f ms %.1f TF, wgrad %.3f ms %.1f TF" % (
                tf2, gf / tf2, td2, gf / td2, tw2, gf / tw2)
        elif ops.CONV_MATH == "f16x3" and cin % 8 != 0 and ops._wgrad_thin_tr(desc):  # the stem: padded input companion
            (x_cb, _), (gy_cb, _) = ops.split_companion_padded(x, xb), ops.split_companion(gy, gb)
            tw2 = timeit(lambda: ops._conv_wgrad(desc, x, gy, x_cb, gy_cb, xb, gb), args.reps)
            tf2 = timeit(lambda: ops._conv_fprop(desc, x, wf, None, True, mpf, x_cb, xb, wb), args.reps)
            extra = " | pre-split: fprop %.3f ms %.1f TF, wgrad %.3f ms %.1f TF" % (tf2, gf / tf2, tw2, gf / tw2)
        tot["fprop"] += tf * cnt
        tot["dgrad"] += td * cnt
        tot["wgrad"] += tw * cnt
        print("%-18s %8.1f | %9.3f %7.1f | %9.3f %7.1f | %9.3f %7.1f%s" % (name, gf, tf, gf / tf, td, gf / td, tw, gf / tw, extra))
    print("per pass (weighted by layer count): fprop %.1f ms  dgrad %.1f ms  wgrad %.1f ms" % (tot["fprop"], tot["dgrad"], tot["wgrad"]))
    if not args.only or args.only == "stream":  # --only stream: just these
        # streaming kernels
        c = 41
        feat = torch.randn(n, c, 60, 80, device=dev)
        upw = torch.randn(c, 1, 16, 16, device=dev) * 0.05
        z1 = ops.up8(feat, upw)
        z2 = ops.up8(feat * 0.9, upw)
        lab = torch.randint(0, c, (n, 480, 640), device=dev)
        cw = torch.ones(c, device=dev)
        nb = z1.numel() * 4
        t = timeit(lambda: ops.up8(feat, upw), args.reps)
        print("up8_fwd            %.3f ms  %.0f GB/s (write)" % (t, nb / t / 1e6))
        t = timeit(lambda: ops._up8_bwd_input(z1, upw, n, c, 60, 80), args.reps)
        print("up8_bwd_input      %.3f ms  %.0f GB/s (read)" % (t, nb / t / 1e6))
        t = timeit(lambda: ops._up8_bwd_weight(z1, feat, n, c, 60, 80), args.reps)
        print("up8_bwd_weight     %.3f ms  %.0f GB/s (read)" % (t, nb / t / 1e6))
        for dxw in ((True, True), (True, False), (False, True)):
            t = timeit(lambda: ops._up8_bwd(z1, upw, feat, *dxw), args.reps)
            print("up8_bwd dx=%d dw=%d  %.3f ms  %.0f GB/s (read)" % (dxw[0], dxw[1], t, nb / t / 1e6))
        t = timeit(lambda: ops.mcd_losses(z1, z2, lab, cw, ce_coef=1.0, diff_coef=-1.0), args.reps)
        print("softmax_ce_l1      %.3f ms  %.0f GB/s (2 reads + 2 writes)" % (t, 4 * nb / t / 1e6))
        x = torch.randn(n, 64, 120, 160, device=dev)
        from models.drn import BatchNorm2d, Conv2d
        conv, bn = Conv2d(64, 64, 3, padding=1, bias=False).to(dev), BatchNorm2d(64).to(dev)
        y = ops.conv_bn_act(x, conv, bn)
        L = ops.lib()
        import ctypes
        mean = torch.zeros(64, device=dev); rstd = torch.ones(64, device=dev)
        out = torch.empty_like(x)
        t = timeit(lambda: L.mcdseg_bn_apply(ops._p(x), ops._p(mean), ops._p(rstd), ops._p(bn.weight), ops._p(bn.bias), None,
                                             ops._p(out), n, 64, 120 * 160, 1, ops._stream()), args.reps)
        print("bn_apply           %.3f ms  %.0f GB/s (read+write)" % (t, 2 * x.numel() * 4 / t / 1e6))


if __name__ == "__main__":
    main()
