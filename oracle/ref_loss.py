"""Oracle restatement of the reference's losses on the hot path (plain torch, CPU).

TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py``.

* ``CrossEntropyLoss2d``          loss.py:7-13   log_softmax over dim 1 (the implicit dim the
                                  reference gets for 4-D input) + weighted-mean NLL, ignore_index -100
* ``Diff2d``                      loss.py:93-100 mean |softmax(o1) - softmax(o2)| over N*C*H*W
* ``get_prob_distance_criterion`` loss.py:192-210 (only "diff" is on the hot path)
* ``class_weights``               util.py:99-111 ones(n_class) with the background class zeroed

``ce_and_grad`` / ``diff_and_grad`` are closed-form numpy/fp64 versions of the same maths
(SURVEY.md Appendix C); they pin the formulas the fused HIP kernel implements.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


class CrossEntropyLoss2d(nn.Module):
    """The class weights live in the ``nll_loss`` sub-module as a buffer (key ``nll_loss.weight``), as in the
    reference -- that key shows up in ``MCDMultiTaskDecoder.state_dict()`` (51 tensors)."""

    def __init__(self, weight=None, size_average=True, ignore_index=-100):
        super().__init__()
        self.nll_loss = nn.NLLLoss(weight, ignore_index=ignore_index, reduction="mean" if size_average else "sum")

    @property
    def weight(self):
        return self.nll_loss.weight

    def forward(self, inputs, targets):
        return self.nll_loss(F.log_softmax(inputs, dim=1), targets)


class ProbCrossEntropyLoss2d(nn.Module):
    """loss.py:16-30: NLLLoss2d(weight, size_average)(log(inputs), targets) for probability maps (gated MFNet fusions)"""

    def __init__(self, weight=None, size_average=True):
        super().__init__()
        self.nll_loss = nn.NLLLoss(weight, reduction="mean" if size_average else "sum")

    def forward(self, inputs, targets):
        return self.nll_loss(torch.log(inputs), targets)


class Diff2d(nn.Module):
    def __init__(self, weight=None, size_average=True):
        super().__init__()
        self.weight = weight

    def forward(self, inputs1, inputs2):
        return torch.mean(torch.abs(F.softmax(inputs1, dim=1) - F.softmax(inputs2, dim=1)))


def get_prob_distance_criterion(criterion_name, n_class=None):
    if criterion_name == "diff":
        return Diff2d()
    raise NotImplementedError("oracle covers d_loss='diff' only (the default, argmyparse.py:131)")


def class_weights(n_class, add_bg_loss=False):
    w = torch.ones(n_class)
    if not add_bg_loss:
        w[n_class - 1] = 0
    return w


# ------------------------------------------------------------------ closed forms (fp64 numpy)
def _softmax64(z):
    z = np.asarray(z, dtype=np.float64)
    m = z.max(axis=1, keepdims=True)
    e = np.exp(z - m)
    s = e.sum(axis=1, keepdims=True)
    return e / s, m + np.log(s)


def ce_and_grad(z, y, w, ignore_index=-100):
    """L = sum_i w[y_i] (lse_i - z_i[y_i]) / sum_i w[y_i];  dL/dz = w[y_i] (p - onehot) / W."""
    p, lse = _softmax64(z)
    n, c, h, wd = p.shape
    y = np.asarray(y)
    valid = y != ignore_index
    ys = np.where(valid, y, 0)
    wy = np.asarray(w, dtype=np.float64)[ys] * valid
    zy = np.take_along_axis(np.asarray(z, np.float64), ys[:, None], axis=1)[:, 0]
    W = wy.sum()
    loss = (wy * (lse[:, 0] - zy)).sum() / W
    onehot = np.zeros_like(p)
    np.put_along_axis(onehot, ys[:, None], 1.0, axis=1)
    grad = wy[:, None] * (p - onehot) / W
    return loss, grad


def diff_and_grad(z1, z2):
    """D = mean|p1-p2|; dD/dz1_c = p1_c (s_c - sum_k s_k p1_k), dD/dz2_c = -p2_c (s_c - sum_k s_k p2_k), s = sign(p1-p2)/M."""
    p1, _ = _softmax64(z1)
    p2, _ = _softmax64(z2)
    M = p1.size
    d = np.abs(p1 - p2).sum() / M
    s = np.sign(p1 - p2) / M
    g1 = p1 * (s - (s * p1).sum(axis=1, keepdims=True))
    g2 = -p2 * (s - (s * p2).sum(axis=1, keepdims=True))
    return d, g1, g2
