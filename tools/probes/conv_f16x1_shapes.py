"""Development probe (GPU): forward convolution and data gradient of drn_d_105's Bottleneck layers in the 2-byte chain at BASELINE config
5's size (N = 32, 90 x 160 maps), each launch alone: TFLOP/s against the pipe and the bytes each launch must move (one-piece operand in,
16-bit result out) per second against HBM -- which layers are bound by which.    python tools/probes/conv_f16x1_shapes.py [N]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "multichannel-semseg-with-uda_amd"))
from mcdseg import ops  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
ONLY = sys.argv[2] if len(sys.argv) > 2 else ""  # e.g. "256-1024"
dev = torch.device("cuda:0")
H, W = 90, 160
ops.CONV_MATH = "f16x1"


def timed(fn, reps=6):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for cin, cout, k, d in ((1024, 256, 1, 1), (256, 256, 3, 2), (256, 1024, 1, 1), (2048, 512, 1, 1), (512, 512, 3, 4), (512, 2048, 1, 1)):
    if ONLY and ONLY != "%d-%d" % (cin, cout):
        continue
    g = torch.Generator().manual_seed(5)
    wt = (torch.randn(cout, cin, k, k, generator=g) * (2.0 / (k * k * cout)) ** 0.5).to(dev)
    desc = ops.conv_desc((N, cin, H, W), wt.shape, 1, d * (k // 2), d)
    pk = ops.PackedWeights()
    wf, wd, mpf = pk.get(wt, desc)
    x = torch.randn(N, cin, H, W, device=dev)
    x_cb, x_bound = ops.split_companion(x)
    del x
    gy = torch.randn(N, cout, H, W, device=dev)
    gy_cb, gy_bound = ops.split_companion(gy)
    del gy
    tf = timed(lambda: ops._conv_fprop_half(desc, x_cb, x_bound, wf, pk.w_bound, mpf))
    td = timed(lambda: ops._conv_dgrad_half(desc, gy_cb, gy_bound, wd, pk.w_bound))
    gf = 2.0 * N * H * W * cout * cin * k * k / 1e9
    gb = N * H * W * (cin + cout) * 2 / 1e9  # one 16-bit value per element in, one out
    print("%4d -> %4d %dx%d d%d: forward %.3f ms = %6.1f TFLOP/s (%.2f of the pipe), %.2f GB -> %.2f TB/s;  data gradient %.3f ms = %6.1f TFLOP/s (%.2f), %.2f TB/s;  batch pieces %s"
          % (cin, cout, k, k, d, tf, gf / tf, gf / tf / 2500.0, gb, gb / tf, td, gf / td, gf / td / 2500.0, gb / td, ops._batch_pieces_half(desc)), flush=True)
    del x_cb, gy_cb
