#!/usr/bin/env python3
"""Development tool (GPU): the small three-step trace under each conv arithmetic -- relative L2 distance of every stored
parameter delta from the reference's fp64 delta, next to the reference's own fp32 noise (tests/golden/trace_deltas.npz)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "multichannel-semseg-with-uda_amd"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
os.environ["MCDSEG_PRETRAINED"] = "0"
from recipe import fill_state_, make_batch  # noqa: E402


def run(math):
    from mcdseg import ops
    ops.CONV_MATH = math
    from loss import CrossEntropyLoss2d, get_prob_distance_criterion
    from models.model_util import get_models, get_optimizer
    from solvers.solver import MCDSolver
    dev = torch.device("cuda:0")
    g, f1, f2 = get_models("drn_d_38", 6, 41)
    for m, seed in ((g, 11), (f1, 12), (f2, 13)):
        fill_state_(m, seed)
        m.to(dev).train()
    flat = lambda: dict(list(g.state_dict().items()) + [("f1." + k, v) for k, v in f1.state_dict().items()] +  # noqa: E731
                        [("f2." + k, v) for k, v in f2.state_dict().items()])
    before = {k: v.detach().clone() for k, v in flat().items()}
    s, l, t = (v.to(dev) for v in make_batch(41, 2, 6, 64, 96, 41))
    og = get_optimizer(g.parameters(), "sgd", 1e-3, 0.9, 2e-5)
    of = get_optimizer(list(f1.parameters()) + list(f2.parameters()), "sgd", 1e-3, 0.9, 2e-5)
    cw = torch.ones(41)
    cw[40] = 0
    solver = MCDSolver(g, f1, f2, og, of, CrossEntropyLoss2d(cw.to(dev)), get_prob_distance_criterion("diff"), num_k=4)
    for _ in range(2):
        solver.step(s, l, t)
    after = flat()
    fx = np.load(os.path.join(ROOT, "tests", "golden", "trace_deltas.npz"))
    out = {}
    for key in fx.files:
        if not key.startswith("f64/"):
            continue
        kind, name = key[4:].split("/", 1)
        r64, r32 = fx[key], fx["f32/" + key[4:]].astype(np.float64)
        cur = after[name].detach().double().cpu()
        got = cur - before[name].double().cpu() if kind == "delta" else cur
        got = (got if got.numel() <= 40000 else got.reshape(got.shape[0], -1)[:16, :288]).numpy()
        out[key[4:]] = (np.linalg.norm(got - r64) / np.linalg.norm(r64), np.linalg.norm(r32 - r64) / np.linalg.norm(r64))
    return out


if __name__ == "__main__":
    res = {m: run(m) for m in (sys.argv[1:] or ["f32", "bf16x6", "f16x3"])}
    keys = list(next(iter(res.values())).keys())
    print("%-40s %10s " % ("tensor", "ref noise") + " ".join("%10s" % m for m in res))
    for k in keys:
        print("%-40s %10.2e " % (k, next(iter(res.values()))[k][1]) + " ".join("%10.2e" % res[m][k][0] for m in res))
