"""Loss criteria of the MCD hot path on the fused HIP kernel (reference: loss.py).

  CrossEntropyLoss2d           loss.py:7-13    log_softmax(dim 1) + weighted NLL (mean over sum of weights)
  ProbCrossEntropyLoss2d       loss.py:16-30   weighted NLL of log(p) for probability maps (gated MFNet fusions)
  Diff2d                       loss.py:93-100  mean |softmax(o1) - softmax(o2)|
  get_prob_distance_criterion  loss.py:192-210 ("diff" is the default ``--d_loss``, argmyparse.py:131)

Both criteria are thin ``nn.Module`` shells over ``mcdseg.ops`` (forward value and d/dlogits come out of
one streaming pass).  ``DiscrepancyLoss`` is an alias of ``Diff2d`` (BASELINE.json uses that name; the
reference has no such symbol).  The other distances of the reference (JSD, Symkl2d, MySymkl2d, SpatialJSD2d,
MisSymKLD; loss.py:70-189) are off the hot path: they are kept as plain-torch criteria over the logits the HIP
classifiers produce, so ``adapt_trainer.py --d_loss symkl`` keeps working through the
drop-in statement loop; only 'diff' runs on the fused kernel.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from mcdseg import ops


class _ClassWeights(nn.Module):
    """holds the class-weight vector where the reference's ``nn.NLLLoss2d`` holds it: buffer ``weight``"""

    def __init__(self, weight):
        super().__init__()
        self.register_buffer("weight", weight)


class CrossEntropyLoss2d(nn.Module):
    def __init__(self, weight=None, size_average=True, ignore_index=-100):
        super().__init__()
        self.nll_loss = _ClassWeights(weight)  # state-dict key ``nll_loss.weight`` as in the reference (loss.py:10)
        self.size_average = size_average
        self.ignore_index = ignore_index

    @property
    def weight(self):
        return self.nll_loss.weight

    def forward(self, inputs, targets):
        w = self.weight
        if w is not None and w.device != inputs.device:
            w = w.to(inputs.device)
        return ops.cross_entropy2d(inputs, targets, w, self.ignore_index, self.size_average)


class ProbCrossEntropyLoss2d(nn.Module):
    """cross entropy between a probability map (0..1) and the labels (loss.py:16-30): the criterion the reference pairs
    with the gated MFNet fusions (adapt_mfnet_trainer.py:149)"""

    def __init__(self, weight=None, size_average=True):
        super().__init__()
        self.nll_loss = _ClassWeights(weight)
        self.size_average = size_average

    def forward(self, inputs, targets):
        w = self.nll_loss.weight
        if w is not None and w.device != inputs.device:
            w = w.to(inputs.device)
        return ops.prob_cross_entropy2d(inputs, targets, w, -100, self.size_average)


class Diff2d(nn.Module):
    def __init__(self, weight=None, size_average=True):
        super().__init__()
        self.weight = weight

    def forward(self, inputs1, inputs2):
        return ops.diff2d(inputs1, inputs2)


DiscrepancyLoss = Diff2d


# ---- the non-default probability distances (plain torch; torch 0.4's implicit softmax dim of a 4-D tensor is 1, and
# F.kl_div(input, target, size_average=True) is the element-wise mean of target * (log(target) - input))
def _kl_mean(log_q, p, size_average=True):
    return F.kl_div(log_q, p, reduction="mean" if size_average else "sum")


class JSD(nn.Module):
    """loss.py:78-89: 0.5 * (KL(p1 || softmax(m)) + KL(p2 || softmax(m))) with m the mean of the LOGITS"""

    def __init__(self, weight=None, size_average=True):
        super().__init__()
        self.weight, self.size_average = weight, size_average

    def forward(self, inputs1, inputs2):
        log_m = F.log_softmax(0.5 * (inputs1 + inputs2), dim=1)
        return 0.5 * (_kl_mean(log_m, F.softmax(inputs1, dim=1), self.size_average) +
                      _kl_mean(log_m, F.softmax(inputs2, dim=1), self.size_average))


class Symkl2d(nn.Module):
    """loss.py:103-118: symmetric KL over rows of ``n_target_ch`` entries (a plain ``view`` of NCHW, as the reference does)"""

    def __init__(self, weight=None, n_target_ch=None, size_average=True):
        super().__init__()
        self.weight, self.n_target_ch, self.size_average = weight, n_target_ch, size_average

    def forward(self, inputs1, inputs2):
        rows = lambda t: t.reshape(-1, self.n_target_ch)  # noqa: E731
        p1, p2 = rows(F.softmax(inputs1, dim=1)), rows(F.softmax(inputs2, dim=1))
        l1, l2 = rows(F.log_softmax(inputs1, dim=1)), rows(F.log_softmax(inputs2, dim=1))
        return 0.5 * (_kl_mean(l1, p2, self.size_average) + _kl_mean(l2, p1, self.size_average))


class MySymkl2d(nn.Module):
    """loss.py:144-154: mean over all elements of 0.5 * (p1 log(p1/p2) + p2 log(p2/p1))"""

    def __init__(self, weight=None, size_average=True):
        super().__init__()
        self.weight = weight

    def forward(self, inputs1, inputs2):
        p1, p2 = F.softmax(inputs1, dim=1), F.softmax(inputs2, dim=1)
        return torch.mean(0.5 * (p1 * torch.log(p1 / p2) + p2 * torch.log(p2 / p1)))


class MisSymKLD(nn.Module):
    """loss.py:66-75 (and SpatialJSD2d, loss.py:157-173, which evaluates the same expression): kl_div is handed
    PROBABILITIES where it expects log-probabilities -- 'strange but somehow works well' in the reference's words"""

    def __init__(self, weight=None, size_average=True):
        super().__init__()
        self.weight = weight

    def forward(self, inputs1, inputs2):
        p1, p2 = F.softmax(inputs1, dim=1), F.softmax(inputs2, dim=1)
        return 0.5 * (_kl_mean(p1, p2) + _kl_mean(p2, p1))


SpatialJSD2d = MisSymKLD


def get_prob_distance_criterion(criterion_name, n_class=None):
    """loss.py:192-210; 'diff' is the fused HIP kernel, the rest are the torch criteria above"""
    if criterion_name == "diff":
        return Diff2d()
    if criterion_name == "jsd":
        return JSD()
    if criterion_name in ("symkl", "nmlsymkl"):
        return Symkl2d(n_target_ch=n_class, size_average=True)
    if criterion_name == "mysymkl":
        return MySymkl2d()
    if criterion_name == "spatial_jsd":
        return SpatialJSD2d()
    if criterion_name == "mis_symkl":
        return MisSymKLD()
    raise NotImplementedError()
