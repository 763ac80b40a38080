"""Loss criteria of the MCD hot path on the fused HIP kernel (reference: loss.py).

  CrossEntropyLoss2d           loss.py:7-13    log_softmax(dim 1) + weighted NLL (mean over sum of weights)
  ProbCrossEntropyLoss2d       loss.py:16-30   weighted NLL of log(p) for probability maps (gated MFNet fusions)
  Diff2d                       loss.py:93-100  mean |softmax(o1) - softmax(o2)|
  get_prob_distance_criterion  loss.py:192-210 ("diff" is the default ``--d_loss``, argmyparse.py:131)

Both criteria are thin ``nn.Module`` shells over ``mcdseg.ops`` (forward value and d/dlogits come out of
one streaming pass).  ``DiscrepancyLoss`` is an alias of ``Diff2d`` (BASELINE.json uses that name; the
reference has no such symbol).  The other distances of the reference (JSD, Symkl2d, ...) are not on the
hot path and are not implemented.
"""
import torch.nn as nn

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


def get_prob_distance_criterion(criterion_name, n_class=None):
    if criterion_name == "diff":
        return Diff2d()
    if criterion_name in ("jsd", "symkl", "nmlsymkl", "mysymkl", "spatial_jsd", "mis_symkl"):
        raise NotImplementedError("d_loss=%r is outside the MI355X hot path; only 'diff' (the default) is built" % criterion_name)
    raise NotImplementedError()
