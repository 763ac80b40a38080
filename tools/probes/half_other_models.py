"""Development probe (GPU): the MFNet and multitask model families in the reduced-precision mode with 2-byte activation storage
(``--dtype f16``: CONV_MATH f16x1 + compact storage + the 2-byte chain) -- one forward + backward of each at a small size, gradients
against the default arithmetic's.    python tools/probes/half_other_models.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "multichannel-semseg-with-uda_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
os.environ.setdefault("MCDSEG_PRETRAINED", "0")
from mcdseg import ops  # noqa: E402

NC = 41
dev = torch.device("cuda:0")
H, W, N = 96, 128, 4


def mfnet():
    from loss import CrossEntropyLoss2d
    from models.model_util import get_models
    from recipe import fill_state_, make_batch
    ms = get_models("drn_d_38", 6, NC, method="MFNet-ScoreAddFusion")
    for i, m in enumerate(ms):
        fill_state_(m, 51 + i)
        m.to(dev).train()
    src, lbl, _ = make_batch(79, N, 6, H, W, NC)
    s = src.to(dev)
    a, b = ms[0](s[:, :3].contiguous()), ms[1](s[:, 3:].contiguous())
    o1, o2 = ms[2](a, b), ms[3](a, b)
    crit = CrossEntropyLoss2d(torch.ones(NC).to(dev))
    (crit(o1, lbl.to(dev)) + crit(o2, lbl.to(dev))).backward()
    torch.cuda.synchronize()
    return {"%d.%s" % (i, k): v.grad.float().clone() for i in range(4) for k, v in ms[i].named_parameters() if v.grad is not None}, o1.detach().float()


def multitask():
    from loss import CrossEntropyLoss2d, Diff2d
    from models.model_util import get_multitask_models
    from recipe import fill_state_, make_batch
    enc, dec = get_multitask_models("drn_d_38", 6, NC, CrossEntropyLoss2d(torch.ones(NC)), Diff2d())
    fill_state_(enc, 81), fill_state_(dec, 82)
    enc.to(dev).train(), dec.to(dev).train()
    src, lbl, _ = make_batch(80, N, 6, H, W, NC)
    fet = enc(src[:, :3].contiguous().to(dev))
    loss = dec.get_loss(fet, lbl.to(dev), src[:, 3:].contiguous().to(dev))
    loss.backward()
    torch.cuda.synchronize()
    g = {"enc." + k: v.grad.float().clone() for k, v in enc.named_parameters() if v.grad is not None}
    g.update({"dec." + k: v.grad.float().clone() for k, v in dec.named_parameters() if v.grad is not None})
    return g, fet.detach().float()


def cmp(tag, name, g0, o0, g1, o1):
    num = sum(float(((g1[k] - g0[k]).double() ** 2).sum()) for k in g0)
    den = sum(float((g0[k].double() ** 2).sum()) for k in g0)
    worst = max(((float((g1[k] - g0[k]).norm() / (g0[k].norm() + 1e-30)), k) for k in g0))
    cos = min(float((g1[k].double() * g0[k].double()).sum() / (g1[k].double().norm() * g0[k].double().norm() + 1e-300)) for k in g0)
    print("%s, %s vs the default arithmetic: %d gradient tensors, overall rel L2 %.3e, worst tensor %.3e (%s), smallest cosine %.4f, output rel %.3e"
          % (name, tag, len(g0), (num / den) ** 0.5, worst[0], worst[1], cos, float((o1 - o0).norm() / o0.norm())), flush=True)


for name, fn in (("MFNet-ScoreAddFusion", mfnet), ("multitask", multitask)):
    ops.CONV_MATH, ops.ACT_STORAGE, ops.HALF_STORAGE = "f16x3", "fp32", True
    g0, o0 = fn()
    ops.CONV_MATH, ops.ACT_STORAGE, ops.HALF_STORAGE = "f16x1", "compact", False
    cmp("f16x1 with two-piece storage (round 5)", name, g0, o0, *fn())
    ops.CONV_MATH, ops.ACT_STORAGE, ops.HALF_STORAGE = "f16x1", "compact", True
    cmp("f16x1 in the 2-byte chain", name, g0, o0, *fn())
ops.CONV_MATH, ops.ACT_STORAGE = "f16x3", "fp32"
