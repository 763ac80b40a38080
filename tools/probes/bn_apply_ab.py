"""Development probe (GPU): the forward BatchNorm-apply kernels alone at BASELINE config 2's shapes (N = 16), events around each launch;
"cold" = a 1 GB buffer streamed between launches, "tail" = z written front to back right before the launch (what the convolution leaves
in the Infinity Cache).    python tools/probes/bn_apply_ab.py"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "multichannel-semseg-with-uda_amd"))
from mcdseg import ops  # noqa: E402
from mcdseg._lib import lib  # noqa: E402

dev = torch.device("cuda:0")
L = lib()
N = 16
flush = torch.empty(1 << 28, dtype=torch.float32, device=dev)
p = ops._p
for c, hw in ((64, 19200), (128, 4800), (256, 4800), (512, 4800)):
    z = torch.randn(N, c, hw, device=dev)
    res = torch.randn(N, c, hw, device=dev)
    y = torch.empty_like(z)
    mean, rstd, gamma, beta = (torch.randn(c, device=dev) * 0.1 for _ in range(4))
    rstd = rstd.abs() + 0.5
    y_bound = torch.full((1,), 8.0, device=dev)
    y_cb = ops._cb_alloc(N, c, hw, dev)
    nb = L.mcdseg_bn_relu_mask_bytes(N, c, hw)
    rmask = torch.empty(max(nb, 8) // 8, dtype=torch.int64, device=dev)
    math = ops.MATH_ID[ops.CONV_MATH]
    forms = {
        "plain (z -> pieces)": (8, lambda: L.mcdseg_bn_apply_cb(p(z), p(mean), p(rstd), p(gamma), p(beta), None, None, None, None, p(y_cb), p(y_bound), math, N, c, hw, 1, ops._stream())),
        "plain + fp32 y": (12, lambda: L.mcdseg_bn_apply_cb(p(z), p(mean), p(rstd), p(gamma), p(beta), None, None, None, p(y), p(y_cb), p(y_bound), math, N, c, hw, 1, ops._stream())),
        "residual + fp32 y + bit-plane": (16, lambda: L.mcdseg_bn_apply_cb_mask(p(z), p(mean), p(rstd), p(gamma), p(beta), p(res), p(y), p(y_cb), p(y_bound), p(rmask), math, N, c, hw, ops._stream())),
    }
    for name, (bpe, call) in forms.items():
        line = "C %3d HW %5d  %-30s" % (c, hw, name)
        for mode in ("cold", "tail"):
            tot, reps = 0.0, 10
            for r in range(reps + 2):
                if mode == "cold":
                    flush.add_(1.0)
                else:
                    flush.add_(1.0)
                    z.mul_(1.0)  # rewritten front to back: its tail stays in the cache
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                rc = call()
                e1.record()
                torch.cuda.synchronize()
                assert rc == 0, L.mcdseg_last_error()
                if r >= 2:
                    tot += e0.elapsed_time(e1)
            ms = tot / reps
            line += "   %s %.4f ms = %.2f TB/s" % (mode, ms, N * c * hw * bpe / ms / 1e9)
        print(line, flush=True)
