"""Late-fusion operators of the MFNet classifiers (reference: models/fusion.py:6-65).

``AddFusion`` (the BASELINE MFNet-ScoreAddFusion config) never materialises its operands: the
classifier that owns it asks the up-sampling kernel for ``up(x1) + up(x2)`` directly
(``mcdseg.ops.up8_dual``).  The gate / concat variants are host-level compositions kept for API
completeness; their 1x1 / 3x3 convolutions run on the HIP convolution kernel.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .drn import Conv2d


class GateFusion(nn.Module):
    def __init__(self, inplanes, apply_softmax=False):
        super().__init__()
        self.conv = Conv2d(inplanes * 2, inplanes, kernel_size=1, stride=1)
        self.apply_softmax = apply_softmax

    def forward(self, x1, x2):
        if self.apply_softmax:
            x1, x2 = F.softmax(x1, dim=1), F.softmax(x2, dim=1)
        gate = torch.sigmoid(self.conv(torch.cat([x1, x2], 1)))
        return x1 * gate + x2 * (1 - gate)


class AddFusion(nn.Module):
    def forward(self, x1, x2):
        return x1 + x2


class ConcatFusion(nn.Module):
    def forward(self, x1, x2):
        return torch.cat([x1, x2], 1)


class ConcatConvFusion(nn.Module):
    def __init__(self, inplanes):
        super().__init__()
        self.conv = Conv2d(inplanes * 2, inplanes, kernel_size=3, padding=1)

    def forward(self, x1, x2):
        return self.conv(torch.cat([x1, x2], 1))


def get_fusion_model(fusion_type, n_ch):
    # substring dispatch; the first match wins, in the reference's order (models/fusion.py:53-65)
    if "ScoreGateFusion" in fusion_type:
        return GateFusion(n_ch, apply_softmax=True)
    if "GateFusion" in fusion_type:
        return GateFusion(n_ch)
    elif "AddFusion" in fusion_type:
        return AddFusion()
    elif "ConcatFusion" in fusion_type:
        return ConcatFusion()
    elif "ConcatConvFusion" in fusion_type:
        return ConcatConvFusion(n_ch)
    raise NotImplementedError()
