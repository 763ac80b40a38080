"""Development probe (GPU): the weight gradient of the Bottleneck layers of drn_d_105 at BASELINE config 5's size (N = 32, 90 x 160 maps)
in the one-term arithmetic, each launch alone: TFLOP/s against the 2.5 PFLOP/s pipe and the operand bytes per second -- which of the
layers are bound by the operand stream (the 1 x 1 convolutions) and which by the K loop (3 x 3).    python tools/probes/wgrad_f16x1_shapes.py [N]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "multichannel-semseg-with-uda_amd"))
from mcdseg import ops  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda:0")
H, W = 90, 160
for math in ("f16x1", "f16x3"):
    ops.CONV_MATH = math
    for cin, cout, k, d in ((1024, 256, 1, 1), (256, 256, 3, 2), (256, 1024, 1, 1), (2048, 512, 1, 1), (512, 512, 3, 4), (512, 2048, 1, 1)):
        x = torch.randn(N, cin, H, W, device=dev)
        gy = torch.randn(N, cout, H, W, device=dev)
        desc = ops.conv_desc(x.shape, (cout, cin, k, k), 1, d * (k // 2), d)
        xb, gb = ops._bound_or_measure(x, None), ops._bound_or_measure(gy, None)
        (x_cb, _), (gy_cb, _) = ops.split_companion(x, xb), ops.split_companion(gy, gb)
        for _ in range(2):
            ops._conv_wgrad(desc, x, gy, x_cb, gy_cb, xb, gb)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 6
        e0.record()
        for _ in range(reps):
            ops._conv_wgrad(desc, x, gy, x_cb, gy_cb, xb, gb)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        gf = 2.0 * N * H * W * cout * cin * k * k / 1e9
        pieces = 1 if math == "f16x1" else 2
        gbytes = N * H * W * (cin + cout) * 2 * pieces / 1e9
        print("%s  %4d -> %4d %dx%d d%d: %.3f ms  %7.1f TFLOP/s alg (%.2f of the pipe at %d term(s))  operands %.2f GB -> %.2f TB/s, pieces of the batch %s"
              % (math, cin, cout, k, k, d, ms, gf / ms, gf / ms * (1 if math == "f16x1" else 3) / 2500.0, 1 if math == "f16x1" else 3, gbytes, gbytes / ms,
                 ops._batch_pieces(desc, wgrad_cb=True)), flush=True)
        del x, gy, x_cb, gy_cb
