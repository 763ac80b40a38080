"""Segmentation wrappers around the DRN trunk (reference: models/dilated_fcn.py).

  DRNSeg                              :68-110   trunk + seg + up (source-only training, cfg1)
  DRNSegBase                = G       :217-250  trunk + 1x1 ``seg`` (ver1) / trunk only (ver2)
  DRNSegPixelClassifier     = F1/F2   :340-366  learned x8 depthwise transposed-conv up-sampler
  FusionDRNSegPixelClassifier         :431-470  fuse features, one up-sampler
  ScoreFusionDRNSegPixelClassifier    :473-491  one up-sampler per modality, fuse scores

State-dict keys follow the reference: ``base.<stage>...``, ``seg.{weight,bias}``,
``up.weight`` / ``up1.weight`` / ``up2.weight``.
"""
import math

import torch
import torch.nn as nn
from torch.nn import Parameter

from mcdseg import ops

from . import drn
from .drn import BatchNorm2d, Conv2d, FusedSequential, run_fused
from .fusion import AddFusion, ConcatFusion, get_fusion_model


def _trunk(model_name, pretrained, input_ch):
    ctor = drn.__dict__.get(model_name)
    if ctor is None or not model_name.startswith("drn_"):
        raise NotImplementedError("unknown DRN variant %r" % (model_name,))
    model = ctor(pretrained=pretrained, num_classes=0, input_ch=input_ch)
    return Trunk(*model.trunk()), model.out_dim


class Trunk(FusedSequential):
    """The DRN stages as ``nn.Sequential(*children[:-2])`` (models/dilated_fcn.py:223); same state_dict keys.  Every stage but
    the last runs inside ``ops.trunk_internal()``: with MCDSEG_ACT_STORAGE=compact those layers keep their activations only
    as the pre-split companions (BASELINE config 5); the last stage writes fp32, so what leaves the trunk is an ordinary tensor."""

    def forward(self, x):
        mods = list(self.children())
        if not mods:
            return x
        # the fusing walk of FusedSequential, so that the DRN-C stem (conv1, bn1, relu as top-level children, models/drn.py:118-121)
        # runs as one fused group like everywhere else
        with ops.late_weight_grads(self):  # (the weight gradients of these layers may stay on the side stream, mcdseg/ops.py)
            with ops.trunk_internal():
                x = run_fused(mods[:-1], x, following=mods[-1])
            return run_fused(mods[-1:], x)


def _seg_head(cin, n_class):
    seg = Conv2d(cin, n_class, kernel_size=1, bias=True)
    n = seg.kernel_size[0] * seg.kernel_size[1] * seg.out_channels
    seg.weight.data.normal_(0, math.sqrt(2.0 / n))
    seg.bias.data.zero_()
    return seg


class Up8(nn.ConvTranspose2d):
    """ConvTranspose2d(C, C, 16, stride 8, padding 4, groups=C, bias=False) -- parameters in torch's layout and
    default initialisation (the reference leaves ``fill_up_weights`` commented out, :93), HIP kernels for the maths."""

    def __init__(self, n_class):
        super().__init__(n_class, n_class, 16, stride=8, padding=4, output_padding=0, groups=n_class, bias=False)

    def forward(self, x, output_size=None):
        return ops.up8(x, self.weight)


class Up8Pairs(nn.ConvTranspose2d):
    """ConvTranspose2d(2C, C, 16, stride 8, padding 4, groups=C, bias=False) behind ``ConcatFusion``
    (models/dilated_fcn.py:445-448): output channel g sums the up-sampled input channels 2g and 2g+1 of the stacked
    tensor, each with its own 16x16 kernel -- the two-input form of the up-sampling kernel on the even / odd channels."""

    def __init__(self, n_class):
        super().__init__(2 * n_class, n_class, 16, stride=8, padding=4, output_padding=0, groups=n_class, bias=False)

    def forward(self, x, output_size=None):
        return ops.up8_dual(x[:, 0::2].contiguous(), self.weight[0::2].contiguous(), x[:, 1::2].contiguous(),
                            self.weight[1::2].contiguous())


def _up(cin, n_class, use_torch_up=False):
    if use_torch_up:
        raise NotImplementedError("use_torch_up (bilinear) is not on the MCD hot path")
    return Up8(n_class) if cin == n_class else Up8Pairs(n_class)


class DRNSeg(nn.Module):
    def __init__(self, model_name, n_class, input_ch=3, pretrained_model=None, pretrained=True, use_torch_up=False):
        super().__init__()
        self.base, out_dim = _trunk(model_name, pretrained, input_ch)
        if pretrained_model is not None:
            self.base.load_state_dict({k.replace("module.", ""): v for k, v in pretrained_model.items()}, strict=False)
        self.seg = _seg_head(out_dim, n_class)
        self.up = _up(n_class, n_class, use_torch_up)

    def forward(self, x):
        return self.up(self.seg(self.base(x)))

    def optim_parameters(self, memo=None):
        yield from self.base.parameters()
        yield from self.seg.parameters()


class DRNSegBase(nn.Module):
    def __init__(self, model_name, n_class, pretrained=True, input_ch=3, ver="ver1"):
        super().__init__()
        self.base, out_dim = _trunk(model_name, pretrained, input_ch)
        self.ver = ver
        if ver == "ver1":
            self.seg = _seg_head(out_dim, n_class)
        elif ver == "ver2":
            print("ver2 will be used")

    def forward(self, x):
        x = self.base(x)
        return x if self.ver == "ver2" else self.seg(x)

    def optim_parameters(self, memo=None):
        yield from self.base.parameters()
        yield from self.seg.parameters()


class DRNSegPixelClassifier(nn.Module):
    def __init__(self, n_class, use_torch_up=False, dropout=False, ver="ver1"):
        super().__init__()
        self.dropout = dropout
        self.ver = ver
        if ver == "ver2":
            self.seg = _seg_head(512, n_class)
        self.up = _up(n_class, n_class, use_torch_up)

    def forward(self, x):
        if self.ver == "ver2":
            x = self.seg(x)
        return self.up(x)


class FusionDRNSegPixelClassifier(nn.Module):
    def __init__(self, fusion_type, n_class, use_torch_up=False, ver="ver1"):
        super().__init__()
        self.fusion = get_fusion_model(fusion_type, n_class if ver == "ver1" else 512)
        self.ver = ver
        self.up = _up(2 * n_class if isinstance(self.fusion, ConcatFusion) else n_class, n_class, use_torch_up)
        if ver == "ver2":
            self.seg = _seg_head(512, n_class)

    def forward(self, x1, x2):
        if isinstance(self.fusion, AddFusion) and self.ver == "ver1":
            # up(x1 + x2) = up(x1) + up(x2): the sum is never materialised, one pass over the full-resolution output
            return ops.up8_dual(x1, self.up.weight, x2, self.up.weight)
        h = self.fusion(x1, x2)
        if self.ver == "ver2":
            h = self.seg(h)
        return self.up(h)


class ScoreFusionDRNSegPixelClassifier(nn.Module):
    def __init__(self, fusion_type, n_class):
        super().__init__()
        self.fusion = get_fusion_model(fusion_type, n_class)
        self.up1 = Up8(n_class)
        self.up2 = Up8(n_class)

    def forward(self, x1, x2):
        if isinstance(self.fusion, AddFusion):
            # up1(x1) + up2(x2) in one pass over the full-resolution tensor
            return ops.up8_dual(x1, self.up1.weight, x2, self.up2.weight)
        return self.fusion(self.up1(x1), self.up2(x2))


# ------------------------------------------------------------------------------------------------ multitask (cfg4)
class MultiTaskEncoder(nn.Module):
    """DRN trunk on the RGB channels (models/dilated_fcn.py:554-566)."""

    def __init__(self, model_name, pretrained=True, input_ch=3):
        super().__init__()
        self.base, _ = _trunk(model_name, pretrained, input_ch)

    def forward(self, x):
        return self.base(x)


class CBR(nn.Module):
    """conv (with bias) - BN - ReLU as one fused HIP group (models/dilated_fcn.py:632-644)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True):
        super().__init__()
        self.conv = Conv2d(in_channels, out_channels, kernel_size, stride=stride, padding=padding, dilation=dilation,
                           groups=groups, bias=bias)
        self.bn = BatchNorm2d(out_channels)

    def forward(self, x):
        return ops.conv_bn_act(x, self.conv, self.bn, relu=True)


class ThreeLayerDecoder(nn.Module):
    def __init__(self, output_ch, input_ch=512):
        super().__init__()
        self.cbr1 = CBR(input_ch, 512, kernel_size=3, padding=1)
        self.cbr2 = CBR(512, 512, kernel_size=1)
        self.conv3 = Conv2d(512, output_ch, kernel_size=1)

    def forward(self, x):
        return self.conv3(self.cbr2(self.cbr1(x)))


class _Bilinear8(nn.Module):
    """nn.Upsample(scale_factor=8, mode='bilinear') (align_corners=False) on the HIP kernel; holds no parameters, so
    the decoder's state_dict is unchanged."""

    def forward(self, x):
        return ops.bilinear8(x)


class MCDMultiTaskDecoder(nn.Module):
    """Two segmentation heads + one HHA-regression head on 512-ch features, x8 bilinear up-sampling, learned
    log-variance task weights ``exp(-s) * L + s`` (models/dilated_fcn.py:661-739)."""

    def __init__(self, n_class, depth_ch, semseg_criterion=None, discrepancy_criterion=None):
        super().__init__()
        self.s_semsegcls = Parameter(torch.ones(1))
        self.s_deprgr = Parameter(torch.ones(1))
        self.semsegcls_dec1 = ThreeLayerDecoder(n_class)
        self.semsegcls_dec2 = ThreeLayerDecoder(n_class)
        self.deprgr_dec = ThreeLayerDecoder(depth_ch)
        self.semseg_criterion = semseg_criterion
        self.discrepancy_criterion = discrepancy_criterion
        self.upsample = _Bilinear8()

    def semseg_forward(self, x):
        return self.upsample(self.semsegcls_dec1(x)), self.upsample(self.semsegcls_dec2(x))

    def depth_forward(self, x):
        return self.upsample(self.deprgr_dec(x))

    def forward(self, x):
        pred_semseg1, pred_semseg2 = self.semseg_forward(x)
        return pred_semseg1, pred_semseg2, self.depth_forward(x)

    def get_cls_descrepancy(self, x):
        pred_semseg1, pred_semseg2 = self.semseg_forward(x)
        return self.discrepancy_criterion(pred_semseg1, pred_semseg2)

    def get_semseg_loss(self, x, gt_semseg, separately_returning=False):
        pred_semseg1, pred_semseg2 = self.semseg_forward(x)
        loss1 = self.semseg_criterion(pred_semseg1, gt_semseg)
        loss2 = self.semseg_criterion(pred_semseg2, gt_semseg)
        return (loss1, loss2) if separately_returning else loss1 + loss2

    def get_depth_loss(self, x, gt_dep):
        return ops.mse_loss(self.depth_forward(x), gt_dep)

    def get_loss(self, x, gt_semseg, gt_dep, separately_returning=False):
        loss1, loss2 = self.get_semseg_loss(x, gt_semseg, separately_returning=True)
        s = self.s_semsegcls
        semseg_loss = ((torch.exp(-s) * loss1 + s) + (torch.exp(-s) * loss2 + s)) / 2
        depreg_loss = torch.exp(-self.s_deprgr) * self.get_depth_loss(x, gt_dep) + self.s_deprgr
        return (semseg_loss, depreg_loss) if separately_returning else semseg_loss + depreg_loss

    def get_task_weights(self):
        import numpy as np
        return (np.sqrt(np.exp(2 * self.s_semsegcls.data.cpu().numpy())), np.sqrt(np.exp(2 * self.s_deprgr.data.cpu().numpy())))
