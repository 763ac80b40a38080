#!/usr/bin/env python3
"""Timing-only ablations of the f32-MFMA implicit-GEMM conv kernel (development tool; the table in profiles/README.md).
The hook (mcdseg_debug_ablate) exists in conv_gemm.hip only, so the tool pins MCDSEG_CONV_MATH to f32."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multichannel-semseg-with-uda_amd"))
import torch
from mcdseg import ops
from mcdseg._lib import lib
L = lib()
dev = torch.device("cuda:0")
n, cin, cout, k, d, h, w = 16, 512, 512, 3, 4, 60, 80
x = torch.randn(n, cin, h, w, device=dev)
wt = torch.randn(cout, cin, k, k, device=dev) * 0.05
desc = ops.conv_desc(x.shape, wt.shape, 1, d, d)
ops.CONV_MATH = "f32"
wf, wd, mpf = ops.PackedWeights().get(wt, desc)
gf = 2.0 * n * h * w * cout * cin * 9 / 1e9
def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(reps): fn()
    t1.record(); torch.cuda.synchronize()
    return t0.elapsed_time(t1) / reps
for bits, what in ((0, "full"), (1, "no global loads"), (2, "no LDS stores"), (3, "no loads, no stores"), (7, "no loads/stores/barriers"), (4, "no barriers")):
    L.mcdseg_debug_ablate(bits)
    t = timeit(lambda: ops._conv_fprop(desc, x, wf, None, False, mpf))
    t2 = timeit(lambda: ops._conv_dgrad(desc, x, wd))
    print("%-28s fprop(no stats) %.3f ms %.1f TF | dgrad %.3f ms %.1f TF" % (what, t, gf / t, t2, gf / t2))
L.mcdseg_debug_ablate(0)
t = timeit(lambda: ops._conv_fprop(desc, x, wf, None, True, mpf))
print("fprop with stats epilogue    %.3f ms %.1f TF" % (t, gf / t))
