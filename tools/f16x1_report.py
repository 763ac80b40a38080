#!/usr/bin/env python3
"""Development tool (GPU): how far the reduced-precision arithmetic (MCDSEG_CONV_MATH=f16x1) sits from the reference's golden vectors:
forward features / logits / arg-max maps (tests/golden/fwd_small.npz), loss values and update directions of the small three-step trace."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "multichannel-semseg-with-uda_amd"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
os.environ["MCDSEG_PRETRAINED"] = "0"
from recipe import fill_state_, make_batch  # noqa: E402

NC = 41


def main():
    from mcdseg import ops
    from models.model_util import get_models
    dev = torch.device("cuda:0")
    fx = np.load(os.path.join(ROOT, "tests", "golden", "fwd_small.npz"))
    for math in sys.argv[1:] or ["f16x3", "f16x1"]:
        ops.CONV_MATH = math
        for mode in ("train", "eval"):
            g, f1, f2 = get_models("drn_d_38", 6, NC)
            for m, seed in ((g, 11), (f1, 12), (f2, 13)):
                fill_state_(m, seed)
                m.to(dev).train(mode == "train")
            src, _, _ = make_batch(21, 2, 6, 64, 96, NC)
            with torch.no_grad():
                feat = g(src.to(dev))
                o1 = f1(feat)
            ref = fx["feat_" + mode]
            err = np.abs(feat.cpu().numpy() - ref).max()
            sub = np.abs(o1[:, :, ::4, ::4].cpu().numpy() - fx["logits1_sub_" + mode]).max()
            pred = o1[:, :NC - 1].argmax(1).cpu().numpy()
            mism = pred != fx["argmax1_" + mode]
            marg = fx["margin1_" + mode]
            print("%-6s %-5s feat max abs err %.3e (scale %.3e, rel %.2e)  logits err %.3e (scale %.3e)  argmax mismatch %.4f; worst margin of a mismatch %.3e; "
                  "mismatch where margin > 1e-2: %d, > 2e-2: %d, > 5e-2: %d" % (math, mode, err, np.abs(ref).max(), err / np.abs(ref).max(), sub,
                                                     np.abs(fx["logits1_sub_" + mode]).max(), mism.mean(), float(marg[mism].max()) if mism.any() else 0.0,
                                                     int((mism & (marg > 1e-2)).sum()), int((mism & (marg > 2e-2)).sum()), int((mism & (marg > 5e-2)).sum())))
    # three-step trace: losses and update directions
    from loss import CrossEntropyLoss2d, get_prob_distance_criterion
    from models.model_util import get_optimizer
    from solvers.solver import MCDSolver
    import json
    tr = json.load(open(os.path.join(ROOT, "tests", "golden", "traces.json")))["mcd_small"]
    dl = np.load(os.path.join(ROOT, "tests", "golden", "trace_deltas.npz"))
    for math in sys.argv[1:] or ["f16x3", "f16x1"]:
        ops.CONV_MATH = math
        g, f1, f2 = get_models("drn_d_38", 6, NC)
        for m, seed in ((g, 11), (f1, 12), (f2, 13)):
            fill_state_(m, seed)
            m.to(dev).train()
        flat = lambda: dict(list(g.state_dict().items()) + [("f1." + k, v) for k, v in f1.state_dict().items()] +  # noqa: E731
                            [("f2." + k, v) for k, v in f2.state_dict().items()])
        before = {k: v.detach().clone() for k, v in flat().items()}
        n, ch, h, w = tr["shape"]
        s, l, t = (v.to(dev) for v in make_batch(tr["seed_batch"], n, ch, h, w, NC))
        og = get_optimizer(g.parameters(), "sgd", 1e-3, 0.9, 2e-5)
        of = get_optimizer(list(f1.parameters()) + list(f2.parameters()), "sgd", 1e-3, 0.9, 2e-5)
        cw = torch.ones(NC)
        cw[NC - 1] = 0
        solver = MCDSolver(g, f1, f2, og, of, CrossEntropyLoss2d(cw.to(dev)), get_prob_distance_criterion("diff"), num_k=4)
        out = [solver.step(s, l, t) for _ in range(2)]
        print(math, "losses", [(float(a), float(b)) for a, b in out], "reference", tr.get("losses") or [(tr.get("c_loss"), tr.get("d_loss"))])
        after = flat()
        cos = {}
        for key in dl.files:
            if key.startswith("f64/delta/"):
                name = key[len("f64/delta/"):]
                r = dl[key].ravel()
                d = after[name].double().cpu() - before[name].double().cpu()
                d = (d if d.numel() <= 40000 else d.reshape(d.shape[0], -1)[:16, :288]).numpy().ravel()
                cos[name] = float(np.dot(d, r) / (np.linalg.norm(d) * np.linalg.norm(r)))
        print(math, "cosine of the parameter updates with the reference's fp64 updates: min %.4f  " % min(cos.values()), {k: round(v, 4) for k, v in cos.items()})


if __name__ == "__main__":
    main()
