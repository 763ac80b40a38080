"""Late-fusion operators of the MFNet classifiers (reference: models/fusion.py:6-65), on the HIP kernels.

* ``AddFusion`` (the BASELINE MFNet-ScoreAddFusion config) never materialises its operands when it sits behind the
  up-samplers: the owning classifier asks for ``up(x1) + up(x2)`` in one kernel (``ops.up8_dual``).
* ``GateFusion``: the gate logits are a 1x1 convolution (HIP implicit GEMM, bias in the epilogue) of the concatenated
  inputs; sigmoid, mix and the whole backward of the mix are one streaming kernel each (``ops.gate_mix``); the optional
  channel softmax in front is ``ops.softmax_channels``.
* ``ConcatFusion`` / ``ConcatConvFusion``: the concatenation is a copy (``torch.cat``, no arithmetic); the 3x3
  convolution is the HIP kernel.

Parameter names (``conv.weight`` / ``conv.bias``) and the substring dispatch of ``get_fusion_model`` are the reference's.
"""
import torch
import torch.nn as nn

from mcdseg import ops

from .drn import Conv2d


def _stack(x1, x2):
    return torch.cat([x1, x2], 1)


class AddFusion(nn.Module):
    def forward(self, x1, x2):
        return x1 + x2


class ConcatFusion(nn.Module):
    def forward(self, x1, x2):
        return _stack(x1, x2)


class ConcatConvFusion(nn.Module):
    """3x3 convolution (2C -> C, padding 1, bias) over the stacked inputs"""

    def __init__(self, inplanes):
        super().__init__()
        self.conv = Conv2d(2 * inplanes, inplanes, kernel_size=3, padding=1)

    def forward(self, x1, x2):
        return self.conv(_stack(x1, x2))


class GateFusion(nn.Module):
    """per-element convex mix of the two inputs, gate = sigmoid(1x1 conv of both); ``apply_softmax`` (the Score
    variant) turns both inputs into class probabilities first, so the output is a probability map too"""

    def __init__(self, inplanes, apply_softmax=False):
        super().__init__()
        self.apply_softmax = apply_softmax
        self.conv = Conv2d(2 * inplanes, inplanes, kernel_size=1, stride=1)

    def forward(self, x1, x2):
        if self.apply_softmax:
            x1, x2 = ops.softmax_channels(x1), ops.softmax_channels(x2)
        return ops.gate_mix(x1, x2, self.conv(_stack(x1, x2)))


_DISPATCH = (  # first substring hit wins -- "ConcatFusion" therefore shadows nothing, "ConcatConvFusion" is reached last
    ("ScoreGateFusion", lambda n_ch: GateFusion(n_ch, apply_softmax=True)),
    ("GateFusion", lambda n_ch: GateFusion(n_ch)),
    ("AddFusion", lambda n_ch: AddFusion()),
    ("ConcatFusion", lambda n_ch: ConcatFusion()),
    ("ConcatConvFusion", lambda n_ch: ConcatConvFusion(n_ch)),
)


def get_fusion_model(fusion_type, n_ch):
    for key, make in _DISPATCH:
        if key in fusion_type:
            return make(n_ch)
    raise NotImplementedError()
