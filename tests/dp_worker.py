"""Helper of tests/test_trainers_gpu.py::test_two_ranks_on_one_gpu_equal_the_single_process_step (not a test module).

Started once per rank by ``torch.distributed.run`` with MCDSEG_SINGLE_DEVICE=1 MCDSEG_DIST_BACKEND=gloo: every rank builds the same
models, runs ONE MCD step on the SAME batch through the real HIP kernels on cuda:0 -- the data-parallel plumbing included: the
optimizer's flat-gradient all-reduce with the 1/world scale folded into the SGD kernel, the all-reduced cross-entropy normaliser --
and rank 0 writes a fingerprint of the result to the path given as argv[1]."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multichannel-semseg-with-uda_amd"))
sys.path.insert(0, ROOT)
os.environ["MCDSEG_PRETRAINED"] = "0"
if os.environ.get("DP_SKEW") == "1" and os.environ.get("RANK") == "1":
    # rank-local state that changes WHEN gradients reach the optimizer's buckets: this rank keeps every weight gradient on the main
    # stream (they arrive one by one during the pass), rank 0 defers them (they arrive early through the gradient sink)
    os.environ["MCDSEG_OVERLAP_WGRAD"] = "0"
import torch  # noqa: E402


def run_step(dev):
    from loss import CrossEntropyLoss2d, get_prob_distance_criterion
    from models.model_util import get_models, get_optimizer
    from solvers.solver import MCDSolver
    from tests.golden.recipe import fill_state_, make_batch
    nc = 41
    g, f1, f2 = get_models("drn_d_38", 6, nc)
    for m, seed in ((g, 11), (f1, 12), (f2, 13)):
        fill_state_(m, seed)
        m.to(dev).train()
    s, l, t = (v.to(dev) for v in make_batch(77, 2, 6, 64, 96, nc))
    og = get_optimizer(g.parameters(), "sgd", 1e-3, 0.9, 2e-5)
    of = get_optimizer(list(f1.parameters()) + list(f2.parameters()), "sgd", 1e-3, 0.9, 2e-5)
    cw = torch.ones(nc)
    cw[nc - 1] = 0
    solver = MCDSolver(g, f1, f2, og, of, CrossEntropyLoss2d(cw.to(dev)), get_prob_distance_criterion("diff"), num_k=2)
    c_loss, d_loss = solver.step(s, l, t)
    fp = {"c_loss": float(c_loss), "d_loss": float(d_loss)}
    for name, m in (("g", g), ("f1", f1), ("f2", f2)):
        for k, v in m.state_dict().items():
            if v.dtype.is_floating_point:
                fp["%s.%s" % (name, k)] = [float(v.double().sum()), float(v.double().abs().sum())]
    return fp


def main():
    from mcdseg import dist as mdist
    rank, world, _ = mdist.init_from_env()
    dev = torch.device("cuda:0")
    fp = run_step(dev)
    fp["world"] = world
    if os.environ.get("MCDSEG_DP_OVERLAP") == "1":  # (what the bucketed exchange did on this rank: the test wants it to have been used)
        from mcdseg import ops
        fp["deferred"] = ops.WGRAD_STREAM_STATS["deferred"]
    if rank == 0:
        with open(sys.argv[1], "w") as f:
            json.dump(fp, f)
    mdist.barrier()
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
