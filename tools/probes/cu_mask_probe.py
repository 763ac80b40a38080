#!/usr/bin/env python3
"""Probe: what do the bits of hipExtStreamCreateWithCUMask select on this chip?  The benchmark's 512 -> 512 forward convolution (480
workgroups, one per CU) on streams with different masks; the time says how many CUs a mask left."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "multichannel-semseg-with-uda_amd"))
import torch  # noqa: E402

from mcdseg import ops  # noqa: E402


def stream_with(words):
    hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
    mask = (ctypes.c_uint32 * len(words))(*words)
    h = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(h), ctypes.c_uint32(len(words)), mask)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(h.value)


def main():
    dev = torch.device("cuda:0")
    n, c, h, w = 16, 512, 60, 80
    x = torch.randn(n, c, h, w, device=dev)
    wt = torch.randn(c, c, 3, 3, device=dev) * 0.05
    desc = ops.conv_desc(x.shape, wt.shape, 1, 4, 4)
    pk = ops.PackedWeights()
    wf, wd, mpf = pk.get(wt, desc)
    xb = ops._bound_or_measure(x, None)
    x_cb, _ = ops.split_companion(x, xb)
    torch.cuda.synchronize()
    F = 0xFFFFFFFF
    cases = [("torch stream", None), ("all 256 bits", [F] * 8), ("words 0-3", [F] * 4 + [0] * 4), ("words 0-5", [F] * 6 + [0] * 2),
             ("every other bit", [0x55555555] * 8), ("low 16 bits of each word", [0xFFFF] * 8), ("low 24 bits of each word", [0xFFFFFF] * 8),
             ("word 0 only", [F] + [0] * 7), ("words 0-1 (2 words passed)", [F, F]), ("8 words, words 0-1 set", [F, F] + [0] * 6)]
    for name, words in cases:
        s = torch.cuda.Stream() if words is None else stream_with(words)
        with torch.cuda.stream(s):
            for _ in range(3):
                ops._conv_fprop(desc, x, wf, None, True, mpf, x_cb, xb, pk.w_bound)
            t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0.record()
            for _ in range(10):
                ops._conv_fprop(desc, x, wf, None, True, mpf, x_cb, xb, pk.w_bound)
            t1.record()
        torch.cuda.synchronize()
        print("%-32s %.3f ms" % (name, t0.elapsed_time(t1) / 10), flush=True)


if __name__ == "__main__":
    main()
