#!/usr/bin/env python3
"""development: is the step host-bound?  Enqueue time (host, no synchronisation) against GPU time (events) of one generator forward
pass and of one forward + backward pass at BASELINE config 2's size."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "multichannel-semseg-with-uda_amd"))
os.environ.setdefault("MCDSEG_PRETRAINED", "0")
import torch  # noqa: E402

from models.model_util import get_models  # noqa: E402

dev = torch.device("cuda:0")
g, f1, f2 = get_models("drn_d_38", 6, 41)
g.to(dev).train()
x = torch.randn(16, 6, 480, 640, device=dev)


def measure(fn, reps=8):
    fn()
    torch.cuda.synchronize()
    host, gpu = [], []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e0.record()
        fn()
        e1.record()
        host.append((time.perf_counter() - t0) * 1e3)
        torch.cuda.synchronize()
        gpu.append(e0.elapsed_time(e1))
    return sorted(host)[len(host) // 2], sorted(gpu)[len(gpu) // 2]


def fwd():
    with torch.no_grad():
        g(x)


def fwd_bwd():
    y = g(x)
    y.backward(torch.ones_like(y))
    for p in g.parameters():
        p.grad = None


for name, fn in (("forward (no tape)", fwd), ("forward + backward", fwd_bwd)):
    h, d = measure(fn)
    print("%-20s host enqueue %.2f ms, GPU %.2f ms" % (name, h, d))
