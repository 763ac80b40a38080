"""CPU, world_size 2 over gloo: the data-parallel path of the flat optimizer (one all-reduce of the flat gradient
buffer, 1/world folded into the update) and the global CE normaliser.  The HIP update kernel is replaced by a
reference formula here -- what is under test is the host-side distributed logic."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _ref_sgd_(p, g, v, lr, mu, wd, gs=1.0):
    v.mul_(mu).add_(g * gs + wd * p)
    p.sub_(lr * v)


def _worker(rank, world, port, tmp, overlap=False):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    if overlap:  # bucketed all-reduce from post-accumulate hooks: tiny buckets so that the three parameters fall into several
        os.environ.update(MCDSEG_DP_OVERLAP="1", MCDSEG_DP_BUCKET_MB="0.0005")
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "multichannel-semseg-with-uda_amd"))
    from mcdseg import dist as mdist
    from mcdseg import ops
    from mcdseg.optim import FlatSGD
    r, w, _ = mdist.init_from_env(backend="gloo")
    assert (r, w) == (rank, world) and mdist.is_distributed() and mdist.world_size() == world
    ops.sgd_momentum_flat_ = lambda p, g, v, lr, mu, wd, gs=1.0, params=None: _ref_sgd_(p, g, v, lr, mu, wd, gs)
    FlatSGD._require_gpu = False
    gen = torch.Generator().manual_seed(0)
    shapes = [(8, 3, 3, 3), (8,), (5, 7)]
    params = [torch.nn.Parameter(torch.randn(s, generator=gen)) for s in shapes]
    ref = [p.detach().clone() for p in params]
    refv = [torch.zeros_like(p) for p in ref]
    opt = FlatSGD(params, lr=0.1, momentum=0.9, weight_decay=0.01)
    for step in range(3):
        grads_all = [[torch.randn(s, generator=torch.Generator().manual_seed(100 * step + 10 * k + i)) for i, s in enumerate(shapes)]
                     for k in range(world)]
        opt.zero_grad()
        if overlap == "twice" and step > 0:
            # TWO backward passes before one step() (MCDSolver's step B) -- the second pass finds the buckets' collectives
            # started by the first; the update must use the ACCUMULATED gradient averaged over the ranks.  And before that a pass whose
            # gradients are dropped by zero_grad() without a step (the reference's literal loop): its collectives must not leak.
            junk = sum((p * 7.0).sum() for p in params)
            junk.backward()
            assert all(b["work"] is not None for b in opt._flat["buckets"])
            opt.zero_grad()
            assert all(b["work"] is None and b["arrived"] == 0 for b in opt._flat["buckets"])
            extra = [[torch.randn(s, generator=torch.Generator().manual_seed(5000 + 100 * step + 10 * k + i)) for i, s in enumerate(shapes)]
                     for k in range(world)]
            sum((p * g).sum() for p, g in zip(params, grads_all[rank])).backward()
            sum((p * g).sum() for p, g in zip(params, extra[rank])).backward()
            assert all(b["dirty"] and b["work"] is None for b in opt._flat["buckets"])
            grads_all = [[a + b for a, b in zip(grads_all[k], extra[k])] for k in range(world)]
        elif overlap and step > 0:  # through autograd, so that the hooks see the gradients arrive (step 0: plain assignment -> fallback path)
            loss = sum((p * g).sum() for p, g in zip(params, grads_all[rank]))
            loss.backward()
            fl = opt._flat
            assert len(fl["buckets"]) >= 2 and all(b["work"] is not None for b in fl["buckets"])
        else:
            for p, g in zip(params, grads_all[rank]):
                p.grad = g.clone()
        opt.step()
        for i in range(len(ref)):
            gavg = sum(grads_all[k][i] for k in range(world)) / world
            _ref_sgd_(ref[i], gavg, refv[i], 0.1, 0.9, 0.01)
    for p, q in zip(params, ref):
        assert torch.allclose(p.detach(), q, rtol=1e-5, atol=1e-6), float((p.detach() - q).abs().max())
    # every rank holds the same replica
    flat = opt.flat_buffers()[0].clone()
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    torch.distributed.all_gather(gathered, flat)
    assert all(torch.equal(gathered[0], t) for t in gathered)
    # global CE normaliser: sum over ranks / world
    local = torch.tensor([10.0 + rank])
    mdist.all_reduce_sum_(local)
    assert float(local) == sum(10.0 + k for k in range(world))
    mdist.barrier()
    torch.distributed.destroy_process_group()
    open(os.path.join(tmp, "ok%d" % rank), "w").write("ok")


def test_flat_sgd_data_parallel_gloo(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / ("ok%d" % r)).exists() for r in range(world))


def test_flat_sgd_bucketed_overlap_gloo(tmp_path):
    """MCDSEG_DP_OVERLAP=1: gradients copied into the flat buffer by post-accumulate hooks, buckets all-reduced asynchronously as they
    complete during backward -- same parameters as the reference update of the averaged gradients (and the un-hooked first step
    takes the one-collective path)"""
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), True), nprocs=world, join=True)
    assert all((tmp_path / ("ok%d" % r)).exists() for r in range(world))


def test_flat_sgd_bucketed_overlap_two_backward_passes_gloo(tmp_path):
    """MCDSEG_DP_OVERLAP=1 with two backward passes before one step() and with a zero_grad() that drops a pass: the
    accumulated gradient is averaged over the ranks, nothing of the dropped pass survives"""
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), "twice"), nprocs=world, join=True)
    assert all((tmp_path / ("ok%d" % r)).exists() for r in range(world))


def test_single_process_helpers():
    from mcdseg import dist as mdist
    assert mdist.world_size() == 1 and mdist.rank() == 0 and not mdist.is_distributed()
    t = torch.ones(3)
    assert mdist.all_reduce_sum_(t) is t and float(t.sum()) == 3
    mdist.barrier()
    env = {k: os.environ.pop(k, None) for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    try:
        assert mdist.init_from_env() == (0, 1, 0)
    finally:
        for k, v in env.items():
            if v is not None:
                os.environ[k] = v
