#!/usr/bin/env python3
"""CPU experiment (no GPU): how accurate are the candidate operand splits for the matrix pipe, measured on the SAME
yardstick as tests/test_model_gpu.py::test_backward_small_vs_reference -- the reference's fp64 parameter gradients and
its own fp32 noise floor (tests/golden/bwd_small.npz)?

Every nn.Conv2d of the CPU oracle network is replaced by an emulation of the kernel arithmetic: both operands are
split into low-precision pieces, the selected cross terms are convolved (fp32 accumulate, as the MFMA does; products
of two pieces are exact in fp32), and forward, data-gradient and weight-gradient all use split operands.

    bf16x6   three bf16 pieces, six cross terms                         (round 1's kernels)
    bf16x3   two bf16 pieces, three cross terms                         (known to fail)
    f16x3    two fp16 pieces, a1b1 + a1b2 + a2b1, power-of-two scale from the tensor's max
    f16x4    the same plus a2b2

Usage: python tools/split_numerics.py [scheme ...]
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
from recipe import fill_state_, make_batch  # noqa: E402

from oracle import ref_loss, ref_models  # noqa: E402

NC = 41


def split_bf16(x, n):
    out, r = [], x
    for _ in range(n):
        p = r.to(torch.bfloat16).float()
        out.append(p)
        r = r - p
    return out, 1.0


def split_f16(x, n, headroom_log2=15):
    """x = s * (h1 + h2): s a power of two chosen from max|x| so that max|x|/s sits below 2^headroom."""
    m = float(x.abs().max())
    if m == 0.0 or not np.isfinite(m):
        return [torch.zeros_like(x)] * n, 1.0
    e = int(np.ceil(np.log2(m))) - headroom_log2
    s = 2.0 ** e
    r = x / s
    out = []
    for _ in range(n):
        p = r.to(torch.float16).float()
        out.append(p)
        r = r - p
    return out, s


SCHEMES = {
    "bf16x6": (lambda t: split_bf16(t, 3), [(2, 0), (0, 2), (1, 1), (1, 0), (0, 1), (0, 0)]),
    "bf16x3": (lambda t: split_bf16(t, 2), [(1, 0), (0, 1), (0, 0)]),
    "f16x3": (lambda t: split_f16(t, 2), [(1, 0), (0, 1), (0, 0)]),
    "f16x4": (lambda t: split_f16(t, 2), [(1, 1), (1, 0), (0, 1), (0, 0)]),
    "f16x3_h8": (lambda t: split_f16(t, 2, 8), [(1, 0), (0, 1), (0, 0)]),
}
ACTIVE = None


def _terms(fn, a, b):
    split, terms = SCHEMES[ACTIVE]
    pa, sa = split(a)
    pb, sb = split(b)
    acc = None
    for i, j in terms:
        t = fn(pa[i], pb[j])
        acc = t if acc is None else acc + t
    return acc * (sa * sb)


class SplitConv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, stride, pad, dil):
        ctx.save_for_backward(x, w)
        ctx.geom = (stride, pad, dil)
        return _terms(lambda a, b: F.conv2d(a, b, None, stride, pad, dil), x, w)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        stride, pad, dil = ctx.geom
        dx = _terms(lambda a, b: torch.nn.grad.conv2d_input(x.shape, b, a, stride, pad, dil), dy, w)
        dw = _terms(lambda a, b: torch.nn.grad.conv2d_weight(b, w.shape, a, stride, pad, dil), dy, x)
        return dx, dw, None, None, None


def patched_forward(self, x):
    if ACTIVE is None or self.in_channels < 16 and False:
        return F.conv2d(x, self.weight, self.bias, self.stride, self.padding, self.dilation)
    y = SplitConv.apply(x, self.weight, self.stride, self.padding, self.dilation)
    if self.bias is not None:
        y = y + self.bias.view(1, -1, 1, 1)
    return y


def run(which, scheme):
    global ACTIVE
    ACTIVE = scheme
    torch.nn.Conv2d.forward = patched_forward
    g, f1, f2 = ref_models.get_models("drn_d_38", 6, NC)
    for m, seed in ((g, 11), (f1, 12), (f2, 13)):
        fill_state_(m, seed)
        m.train()
    src, lbl, tgt = make_batch(21, 2, 6, 64, 96, NC)
    feat = g(src if which == "ce" else tgt)
    a, b = f1(feat), f2(feat)
    if which == "ce":
        crit = ref_loss.CrossEntropyLoss2d(ref_loss.class_weights(NC))
        loss = crit(a, lbl) + crit(b, lbl)
    else:
        loss = ref_loss.Diff2d()(a, b)
    loss.backward()
    fx = np.load(os.path.join(ROOT, "tests", "golden", "bwd_small.npz"))
    named = dict(g.named_parameters())
    worst = 0.0
    rows = []
    for key in fx.files:
        if not key.startswith(which + "/f64/"):
            continue
        name = key.split("/", 2)[2]
        if name in named:
            got = named[name].grad
            got = got if got.numel() <= 40000 else got.reshape(got.shape[0], -1)[:16, :288]
        elif name == "up1":
            got = f1.up.weight.grad
        elif name == "up2":
            got = f2.up.weight.grad
        elif name == "bn_gamma_all":
            got = torch.cat([p.grad.reshape(-1) for k, p in named.items() if p.dim() == 1 and k.endswith("weight")])
        elif name == "bn_beta_all":
            got = torch.cat([p.grad.reshape(-1) for k, p in named.items() if p.dim() == 1 and k.endswith("bias") and not k.startswith("seg")])
        else:
            continue
        g64 = fx[key]
        noise = np.abs(fx[key.replace("/f64/", "/f32/")] - g64).max()
        err = np.abs(got.double().numpy() - g64).max()
        scale = np.abs(g64).max()
        ratio = err / max(noise, 1e-30)
        ok = err <= max(1e-3 * max(scale, 1e-3), 8 * noise)
        rows.append((name, err, noise, scale, ratio, ok))
        worst = max(worst, ratio)
    return float(loss), float(fx[which + "/loss64"]), worst, rows


if __name__ == "__main__":
    torch.set_num_threads(8)
    schemes = sys.argv[1:] or [None, "bf16x6", "f16x3", "f16x4", "bf16x3"]
    for sch in schemes:
        sch = None if sch in (None, "none") else sch
        for which in ("ce", "diff"):
            loss, l64, worst, rows = run(which, sch)
            bad = [r[0] for r in rows if not r[5]]
            print("%-9s %-4s loss rel err %.2e   worst err/ref-noise ratio %.2f   failing(k=8): %s" %
                  (sch or "fp32", which, abs(loss - l64) / abs(l64), worst, bad or "-"), flush=True)
