#!/usr/bin/env python3
"""Development tool (GPU): what the HOST spends on one MCD step -- the time to enqueue a step (no synchronisation inside) against the
time the device needs for it, and a cProfile of the enqueue.  python tools/host_profile.py [cfg2|cfg4] [N]"""
import cProfile
import io
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import torch  # noqa: E402


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
    dev = torch.device("cuda:0")
    if cfg == "cfg2":
        import argparse
        n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
        args = argparse.Namespace(net="drn_d_38", input_ch=6, n_class=41, no_forward_reuse=False)
        solver, _ = bench.build_hip(args, dev)
        batch = [t.to(dev) for t in bench.synthetic_batch(n, 6, 480, 640, 41, 1234)]
    else:
        from loss import CrossEntropyLoss2d, Diff2d
        from models.model_util import get_multitask_models, get_optimizer
        from solvers.solver import MultiTaskMCDSolver
        n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
        cw = torch.ones(41)
        cw[40] = 0
        enc, dec = get_multitask_models("drn_d_38", 6, 41, CrossEntropyLoss2d(cw), Diff2d())
        enc.to(dev).train(), dec.to(dev).train()
        solver = MultiTaskMCDSolver(enc, dec, get_optimizer(enc.parameters(), "sgd", 1e-3, 0.9, 2e-5), get_optimizer(dec.parameters(), "sgd", 1e-3, 0.9, 2e-5), num_k=4)
        batch = [t.to(dev) for t in bench.synthetic_batch(n, 6, 480, 640, 41, 1234)]
    for _ in range(3):
        solver.step(*batch)
    torch.cuda.synchronize()
    enq, tot = [], []
    for _ in range(5):
        t0 = time.perf_counter()
        solver.step(*batch)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        enq.append(1e3 * (t1 - t0))
        tot.append(1e3 * (t2 - t0))
    print("%s N=%d: host enqueue %.1f ms per step (min %.1f), step wall %.1f ms (min %.1f): the host is %s"
          % (cfg, n, sorted(enq)[2], min(enq), sorted(tot)[2], min(tot), "AHEAD of the device" if sorted(enq)[2] < 0.9 * sorted(tot)[2] else "the bottleneck"))
    # back-to-back steps without synchronisation: the steady-state rate
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        solver.step(*batch)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("5 steps back to back: enqueued in %.1f ms per step, done in %.1f ms per step" % (1e3 * (t1 - t0) / 5, 1e3 * (t2 - t0) / 5))
    pr = cProfile.Profile()
    pr.enable()
    solver.step(*batch)
    pr.disable()
    torch.cuda.synchronize()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
    print(s.getvalue()[:6000])


if __name__ == "__main__":
    main()
