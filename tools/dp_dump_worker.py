"""Determinism probe (kernel development, not a test): one rank of an MCD step through the real kernels that dumps FULL tensors.

Started per rank by ``torch.distributed.run`` (MCDSEG_SINGLE_DEVICE=1 MCDSEG_DIST_BACKEND=gloo) or plainly for world 1.
argv: <out.pt> [mode]   mode = "step" (one whole three-step update: final state of G/F1/F2), "grads" (phase A only: the feature
map, the four loss values and every parameter gradient before any optimizer step).  DP_SIZE="N,H,W" (default 2,64,96).
Every rank writes <out.pt>.rank<r>; ``tools/dp_compare.py`` diffs any number of such files tensor by tensor."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multichannel-semseg-with-uda_amd"))
sys.path.insert(0, ROOT)
os.environ["MCDSEG_PRETRAINED"] = "0"
import torch  # noqa: E402


def poison_allocations():
    """DP_POISON=1: every torch.empty / empty_like / new_empty comes back filled with 0xFF bytes (NaN in fp16 / fp32 / fp64), so a kernel that
    reads memory nobody wrote shows up as NaN instead of as whatever the caching allocator last held there"""
    real_empty, real_like = torch.empty, torch.empty_like

    def fill(t):
        if t.is_cuda and t.numel() > 0 and t.is_contiguous():
            t.view(torch.uint8).fill_(0xFF)
        return t

    torch.empty = lambda *a, **k: fill(real_empty(*a, **k))
    torch.empty_like = lambda *a, **k: fill(real_like(*a, **k))


def main():
    if os.environ.get("DP_POISON") == "1":
        poison_allocations()
    from mcdseg import dist as mdist
    from loss import CrossEntropyLoss2d, get_prob_distance_criterion
    from models.model_util import get_models, get_optimizer
    from solvers.solver import MCDSolver
    from tests.golden.recipe import fill_state_, make_batch
    out, mode = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "step")
    rank, world, _ = mdist.init_from_env()
    dev = torch.device("cuda:0")
    n, h, w = (int(v) for v in os.environ.get("DP_SIZE", "2,64,96").split(","))
    net = os.environ.get("DP_NET", "drn_d_38")
    nc = 41
    g, f1, f2 = get_models(net, 6, nc)
    for m, seed in ((g, 11), (f1, 12), (f2, 13)):
        fill_state_(m, seed)
        m.to(dev).train()
    s, l, t = (v.to(dev) for v in make_batch(77, n, 6, h, w, nc))
    og = get_optimizer(g.parameters(), "sgd", 1e-3, 0.9, 2e-5)
    of = get_optimizer(list(f1.parameters()) + list(f2.parameters()), "sgd", 1e-3, 0.9, 2e-5)
    cw = torch.ones(nc)
    cw[nc - 1] = 0
    solver = MCDSolver(g, f1, f2, og, of, CrossEntropyLoss2d(cw.to(dev)), get_prob_distance_criterion("diff"),
                       num_k=int(os.environ.get("DP_NUM_K", "2")))
    dump = {}
    if mode == "grads":
        feats = solver._features(s)
        losses = solver._loss_backward(feats, l, ce_coef=1.0)
        dump["feat"] = feats[0].detach()
        dump["losses"] = losses
        for name, m in (("g", g), ("f1", f1), ("f2", f2)):
            for k, p in m.named_parameters():
                if p.grad is not None:
                    dump["grad:%s.%s" % (name, k)] = p.grad
            for k, b in m.named_buffers():
                dump["buf:%s.%s" % (name, k)] = b
    else:
        c_loss, d_loss = solver.step(s, l, t)
        dump["c_loss"], dump["d_loss"] = c_loss, d_loss
        for name, m in (("g", g), ("f1", f1), ("f2", f2)):
            for k, v in m.state_dict().items():
                dump["%s.%s" % (name, k)] = v
    torch.cuda.synchronize()
    torch.save({k: v.detach().cpu() for k, v in dump.items()}, "%s.rank%d" % (out, rank))
    mdist.barrier()
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
