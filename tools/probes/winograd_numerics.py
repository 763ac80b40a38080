#!/usr/bin/env python3
"""Probe, CPU only: what would Winograd F(2x2, 3x3) cost in ACCURACY on the stride-1 3x3 layers of L5-L8 when its
transformed operands go through the same f16x3 split arithmetic as the direct kernel (csrc/split.h: x = s (h1 + h2), terms
h1 h1' + h1 h2' + h2 h1', fp32 accumulation)?  One 512 -> Cout layer at one image, dilation 1 (a dilated layer is d^2 interleaved
dense problems: same arithmetic), operands as the network has them (post-ReLU activations, He-scaled weights).  Everything is emulated
with torch on the CPU: pieces are fp16 values held in fp32, every product of two pieces is exact in fp32, sums are fp32 matmuls.

    python tools/probes/winograd_numerics.py [--cin 512 --cout 64 --hw 48]
"""
import argparse

import torch
import torch.nn.functional as F


def split2(x, bound):
    """(h1, h2, s): x ~= s (h1 + h2), fp16 pieces held in fp32; s = 2^(ceil(log2 bound) - 15) as csrc/split.h derives it"""
    s = 2.0 ** (torch.ceil(torch.log2(torch.tensor(float(bound)))) - 15)
    t = x / s
    h1 = t.half().float()
    h2 = (t - h1).half().float()
    return h1, h2, float(s)


def mm3(a, b, bound_a, bound_b):
    """a [M, K] @ b [K, N] in the three-term split arithmetic, fp32 accumulation"""
    a1, a2, sa = split2(a, bound_a)
    b1, b2, sb = split2(b, bound_b)
    acc = a2 @ b1
    acc = acc + a1 @ b2
    acc = acc + a1 @ b1
    return acc * (sa * sb)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cin", type=int, default=512)
    ap.add_argument("--cout", type=int, default=64)
    ap.add_argument("--hw", type=int, default=48)
    args = ap.parse_args()
    torch.manual_seed(0)
    C, Co, H = args.cin, args.cout, args.hw
    x = torch.relu(torch.randn(1, C, H, H))
    w = torch.randn(Co, C, 3, 3) * (2.0 / (9 * C)) ** 0.5
    truth = F.conv2d(x.double(), w.double(), padding=1)[0]           # [Co, H, H]
    scale = float(truth.abs().max())
    # ---- direct: im2col, K = 9 C
    cols = F.unfold(x, 3, padding=1)[0]                              # [9 C, H H]
    direct = mm3(w.reshape(Co, -1), cols, w.abs().max(), x.abs().max()).reshape(Co, H, H)
    plain = (w.reshape(Co, -1) @ cols).reshape(Co, H, H)             # an fp32 chain of the same depth, for scale
    # ---- Winograd F(2x2, 3x3): 16 planes of K = C
    Bt = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float32)
    G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float32)
    At = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float32)
    xp = F.pad(x[0], (1, 1, 1, 1))                                   # [C, H + 2, H + 2]
    T = H // 2
    d = xp.unfold(1, 4, 2).unfold(2, 4, 2)                           # [C, T, T, 4, 4] tiles with overlap 2
    V = torch.einsum("ia,ctuab,jb->ctuij", Bt, d, Bt)                # input transform in fp32 (what a BN-apply producer would emit)
    U = torch.einsum("ia,ocab,jb->ocij", G, w, G)                    # weight transform (fp32; per optimizer step)
    M = torch.empty(Co, T, T, 4, 4)
    for i in range(4):
        for j in range(4):
            m = mm3(U[:, :, i, j], V[:, :, :, i, j].reshape(C, T * T), U[:, :, i, j].abs().max(), 4.0 * x.abs().max())
            M[:, :, :, i, j] = m.reshape(Co, T, T)
    Y = torch.einsum("ia,otuab,jb->otuij", At, M, At)                # [Co, T, T, 2, 2], fp32
    wino = Y.permute(0, 1, 3, 2, 4).reshape(Co, H, H)
    # the same with exact (fp64) products, to separate the transform's own rounding from the split arithmetic's
    V64 = torch.einsum("ia,ctuab,jb->ctuij", Bt.double(), d.double(), Bt.double())
    U64 = torch.einsum("ia,ocab,jb->ocij", G.double(), w.double(), G.double())
    M64 = torch.einsum("ocij,ctuij->otuij", U64, V64)
    Y64 = torch.einsum("ia,otuab,jb->otuij", At.double(), M64, At.double()).permute(0, 1, 3, 2, 4).reshape(Co, H, H)
    for name, got in (("fp32 chain (im2col matmul)", plain), ("direct, f16x3", direct), ("Winograd F(2x2,3x3), f16x3 on transformed operands", wino),
                      ("Winograd in fp64 (sanity: the algebra)", Y64)):
        err = (got.double() - truth).abs()
        print("%-58s max err %.3e = %.2e of scale   rms %.3e of scale" % (name, float(err.max()), float(err.max()) / scale,
                                                                         float((err ** 2).mean().sqrt()) / scale))


if __name__ == "__main__":
    main()
