"""Oracle restatement of the reference's DRN segmentation models (plain torch, CPU).

TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py``.

What is restated, and where it lives in the reference:

* DRN-D trunk (layer0..layer8 of ``DRN``)            models/drn.py:103-205
* ``BasicBlock`` / ``Bottleneck`` residual units       models/drn.py:26-100
* He-normal conv init, BN gamma=1 beta=0               models/drn.py:163-169
* first-conv surgery for 1/4/5/6 input channels        models/drn.py:256-299
* ``DRNSegBase`` (G), ``DRNSegPixelClassifier`` (F)    models/dilated_fcn.py:217-250, 340-366
* ``DRNSeg`` (full model used by source_trainer)       models/dilated_fcn.py:68-110
* fusion ops + the two MFNet classifiers               models/fusion.py:6-65, models/dilated_fcn.py:431-491
* factories ``get_models`` / ``get_full_model``        models/model_util.py:6-39, 160-286

The module tree is arranged so that ``state_dict()`` yields exactly the key
names and shapes of the reference; the golden fixture
``keys_shapes.json`` pins that.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

# stage widths of every DRN variant (models/drn.py:105)
WIDTHS = (16, 32, 64, 128, 256, 512, 512, 512)

# (block kind, repeats per stage) -- models/drn.py:302-348
ARCH = {
    "drn_d_22": ("basic", (1, 1, 2, 2, 2, 2, 1, 1)),
    "drn_d_38": ("basic", (1, 1, 3, 4, 6, 3, 1, 1)),
    "drn_d_54": ("bottleneck", (1, 1, 3, 4, 6, 3, 1, 1)),
    "drn_d_105": ("bottleneck", (1, 1, 3, 4, 23, 3, 1, 1)),
}


def _conv(cin, cout, k, stride=1, dilation=1, bias=False):
    pad = dilation * (k // 2)
    return nn.Conv2d(cin, cout, k, stride=stride, padding=pad, dilation=dilation, bias=bias)


class BasicBlock(nn.Module):
    """conv3x3-BN-ReLU-conv3x3-BN (+shortcut) -ReLU   (models/drn.py:26-59)."""
    expansion = 1

    def __init__(self, cin, planes, stride, dil, shortcut):
        super().__init__()
        self.conv1 = _conv(cin, planes, 3, stride, dil[0])
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = _conv(planes, planes, 3, 1, dil[1])
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = shortcut

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        h = F.relu(self.bn1(self.conv1(x)))
        h = self.bn2(self.conv2(h))
        return F.relu(h + idt)


class Bottleneck(nn.Module):
    """1x1-BN-ReLU-3x3(dil)-BN-ReLU-1x1(x4)-BN + shortcut, ReLU (models/drn.py:62-100)."""
    expansion = 4

    def __init__(self, cin, planes, stride, dil, shortcut):
        super().__init__()
        self.conv1 = _conv(cin, planes, 1)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = _conv(planes, planes, 3, stride, dil[1])
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = _conv(planes, planes * 4, 1)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.downsample = shortcut

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        h = F.relu(self.bn1(self.conv1(x)))
        h = F.relu(self.bn2(self.conv2(h)))
        h = self.bn3(self.conv3(h))
        return F.relu(h + idt)


def _plain_stage(cin, cout, n, stride=1, dilation=1):
    """``_make_conv_layers`` (models/drn.py:195-205): n x (conv3x3, BN, ReLU)."""
    mods = []
    for i in range(n):
        mods += [_conv(cin, cout, 3, stride if i == 0 else 1, dilation), nn.BatchNorm2d(cout), nn.ReLU(inplace=True)]
        cin = cout
    return nn.Sequential(*mods), cout


def _res_stage(kind, cin, planes, n, stride=1, dilation=1, new_level=True):
    """``_make_layer`` (models/drn.py:171-193)."""
    block = BasicBlock if kind == "basic" else Bottleneck
    cout = planes * block.expansion
    shortcut = None
    if stride != 1 or cin != cout:
        shortcut = nn.Sequential(nn.Conv2d(cin, cout, 1, stride=stride, bias=False), nn.BatchNorm2d(cout))
    if dilation == 1:
        first = (1, 1)
    else:
        first = (dilation // 2 if new_level else dilation, dilation)
    blocks = [block(cin, planes, stride, first, shortcut)]
    for _ in range(1, n):
        blocks.append(block(cout, planes, 1, (dilation, dilation), None))
    return nn.Sequential(*blocks), cout


def he_normal_(module):
    """Conv: N(0, sqrt(2/(kh*kw*Cout))); BN: gamma 1, beta 0 (models/drn.py:163-169)."""
    for m in module.modules():
        if isinstance(m, nn.Conv2d):
            fan = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
            m.weight.data.normal_(0, math.sqrt(2.0 / fan))
            if m.bias is not None:
                m.bias.data.zero_()
        elif isinstance(m, nn.BatchNorm2d):
            m.weight.data.fill_(1)
            m.bias.data.zero_()


def widen_first_conv(w3, input_ch):
    """First-conv surgery of ``replace_first_conv`` (models/drn.py:256-299).

    ``w3`` is the 3-channel 7x7 kernel [16,3,7,7].  1 channel keeps the R
    slice; 4..6 channels keep RGB and re-use the first ``input_ch-3`` RGB
    slices for the extra channels; 3 is the identity; anything else is
    rejected, as in the reference.
    """
    if input_ch == 3:
        return w3.clone()
    if input_ch == 1:
        return w3[:, 0:1].clone()
    if 3 < input_ch <= 6:
        extra = input_ch - 3
        return torch.cat([w3, w3[:, :extra]], dim=1).clone()
    raise NotImplementedError("input_ch must be 1, 3, 4, 5 or 6")


def drn_trunk(model_name, input_ch=3):
    """layer0..layer8 as one ``nn.Sequential`` (== ``Sequential(*children()[:-2])``,
    models/dilated_fcn.py:223).  Returns (trunk, out_channels)."""
    kind, reps = ARCH[model_name]
    stages = []
    stem = nn.Sequential(nn.Conv2d(3, WIDTHS[0], 7, padding=3, bias=False), nn.BatchNorm2d(WIDTHS[0]),
                         nn.ReLU(inplace=True))
    stages.append(stem)
    c = WIDTHS[0]
    s, c = _plain_stage(c, WIDTHS[0], reps[0]); stages.append(s)
    s, c = _plain_stage(c, WIDTHS[1], reps[1], stride=2); stages.append(s)
    s, c = _res_stage(kind, c, WIDTHS[2], reps[2], stride=2); stages.append(s)
    s, c = _res_stage(kind, c, WIDTHS[3], reps[3], stride=2); stages.append(s)
    s, c = _res_stage(kind, c, WIDTHS[4], reps[4], dilation=2, new_level=False); stages.append(s)
    s, c = _res_stage(kind, c, WIDTHS[5], reps[5], dilation=4, new_level=False); stages.append(s)
    s, c = _plain_stage(c, WIDTHS[6], reps[6], dilation=2); stages.append(s)
    s, c = _plain_stage(c, WIDTHS[7], reps[7], dilation=1); stages.append(s)
    trunk = nn.Sequential(*stages)
    he_normal_(trunk)
    if input_ch != 3:
        old = trunk[0][0]
        new = nn.Conv2d(input_ch, WIDTHS[0], 7, padding=3, bias=False)
        new.weight.data = widen_first_conv(old.weight.data, input_ch)
        trunk[0] = nn.Sequential(new, trunk[0][1], nn.ReLU(inplace=True))
    return trunk, c


def _seg_head(cin, n_class):
    seg = nn.Conv2d(cin, n_class, 1, bias=True)
    seg.weight.data.normal_(0, math.sqrt(2.0 / n_class))  # fan = 1*1*out_channels (dilated_fcn.py:229-232)
    seg.bias.data.zero_()
    return seg


def _up8(cin, n_class):
    """Learned x8 up-sampler: ConvTranspose2d k16 s8 p4, one group per class
    (models/dilated_fcn.py:357-360; default torch init, no bilinear fill)."""
    return nn.ConvTranspose2d(cin, n_class, 16, stride=8, padding=4, output_padding=0, groups=n_class, bias=False)


class DRNSegBase(nn.Module):
    """Generator G (models/dilated_fcn.py:217-250)."""

    def __init__(self, model_name, n_class, input_ch=3, ver="ver1"):
        super().__init__()
        self.base, c = drn_trunk(model_name, input_ch)
        self.ver = ver
        if ver == "ver1":
            self.seg = _seg_head(c, n_class)

    def forward(self, x):
        x = self.base(x)
        return x if self.ver == "ver2" else self.seg(x)


class DRNSegPixelClassifier(nn.Module):
    """Classifier F1/F2 (models/dilated_fcn.py:340-366)."""

    def __init__(self, n_class, ver="ver1"):
        super().__init__()
        self.ver = ver
        if ver == "ver2":
            self.seg = _seg_head(512, n_class)
        self.up = _up8(n_class, n_class)

    def forward(self, x):
        if self.ver == "ver2":
            x = self.seg(x)
        return self.up(x)


class DRNSeg(nn.Module):
    """trunk + seg + up in one module (models/dilated_fcn.py:68-110)."""

    def __init__(self, model_name, n_class, input_ch=3):
        super().__init__()
        self.base, c = drn_trunk(model_name, input_ch)
        self.seg = _seg_head(c, n_class)
        self.up = _up8(n_class, n_class)

    def forward(self, x):
        return self.up(self.seg(self.base(x)))


# ---------------------------------------------------------------- fusion (models/fusion.py)
class GateFusion(nn.Module):
    def __init__(self, ch, apply_softmax=False):
        super().__init__()
        self.conv = nn.Conv2d(2 * ch, ch, 1)
        self.apply_softmax = apply_softmax

    def forward(self, a, b):
        if self.apply_softmax:
            a, b = F.softmax(a, dim=1), F.softmax(b, dim=1)
        g = torch.sigmoid(self.conv(torch.cat([a, b], 1)))
        return a * g + b * (1 - g)


class AddFusion(nn.Module):
    def forward(self, a, b):
        return a + b


class ConcatFusion(nn.Module):
    def forward(self, a, b):
        return torch.cat([a, b], 1)


class ConcatConvFusion(nn.Module):
    def __init__(self, ch):
        super().__init__()
        self.conv = nn.Conv2d(2 * ch, ch, 3, padding=1)

    def forward(self, a, b):
        return self.conv(torch.cat([a, b], 1))


def get_fusion_model(fusion_type, n_ch):
    # substring dispatch, first hit wins -- order as in models/fusion.py:53-65
    if "ScoreGateFusion" in fusion_type:
        return GateFusion(n_ch, apply_softmax=True)
    if "GateFusion" in fusion_type:
        return GateFusion(n_ch)
    if "AddFusion" in fusion_type:
        return AddFusion()
    if "ConcatFusion" in fusion_type:
        return ConcatFusion()
    if "ConcatConvFusion" in fusion_type:
        return ConcatConvFusion(n_ch)
    raise NotImplementedError(fusion_type)


class FusionDRNSegPixelClassifier(nn.Module):
    """fuse features, then one up-sampler (models/dilated_fcn.py:431-470)."""

    def __init__(self, fusion_type, n_class, ver="ver1"):
        super().__init__()
        self.ver = ver
        self.fusion = get_fusion_model(fusion_type, n_class if ver == "ver1" else 512)
        self.up = _up8(2 * n_class if isinstance(self.fusion, ConcatFusion) else n_class, n_class)
        if ver == "ver2":
            self.seg = _seg_head(512, n_class)

    def forward(self, a, b):
        h = self.fusion(a, b)
        if self.ver == "ver2":
            h = self.seg(h)
        return self.up(h)


class ScoreFusionDRNSegPixelClassifier(nn.Module):
    """one up-sampler per modality, then fuse (models/dilated_fcn.py:473-491)."""

    def __init__(self, fusion_type, n_class):
        super().__init__()
        self.fusion = get_fusion_model(fusion_type, n_class)
        self.up1 = _up8(n_class, n_class)
        self.up2 = _up8(n_class, n_class)

    def forward(self, a, b):
        return self.fusion(self.up1(a), self.up2(b))


# ---------------------------------------------------------------- factories (models/model_util.py)
def get_models(net_name, input_ch, n_class, res="50", method="MCD", is_data_parallel=False):
    if "drn" not in net_name or "fusenet" in net_name:
        raise NotImplementedError("oracle covers the DRN hot path only")
    ver = "ver2" if "ver2" in net_name else "ver1"
    drn_name = net_name.replace("_ver2", "")
    if method == "MCD":
        models = [DRNSegBase(drn_name, n_class, input_ch, ver), DRNSegPixelClassifier(n_class, ver),
                  DRNSegPixelClassifier(n_class, ver)]
    elif "MFNet" in method:
        assert input_ch in (4, 6)
        fusion_type = method.split("-")[-1]
        g3 = DRNSegBase(drn_name, n_class, 3, ver)
        g1 = DRNSegBase(drn_name, n_class, input_ch - 3, ver)
        if "score" in method.lower():
            fs = [ScoreFusionDRNSegPixelClassifier(fusion_type, n_class) for _ in range(2)]
        else:
            fs = [FusionDRNSegPixelClassifier(fusion_type, n_class, ver) for _ in range(2)]
        models = [g3, g1] + fs
    else:
        return NotImplementedError("Sorry... Only MCD is supported!")  # returned, not raised (model_util.py:281)
    if is_data_parallel:
        return [nn.DataParallel(m) for m in models]
    return models


def get_full_model(net, res, n_class, input_ch, is_data_parallel=True):
    if "drn" not in net:
        raise NotImplementedError("oracle covers the DRN hot path only")
    model = DRNSeg(net, n_class, input_ch)
    return nn.DataParallel(model) if is_data_parallel else model


def get_optimizer(model_parameters, opt, lr, momentum, weight_decay):
    params = [p for p in model_parameters if p.requires_grad]
    if opt == "sgd":
        return torch.optim.SGD(params, lr=lr, momentum=momentum, weight_decay=weight_decay)
    if opt == "adadelta":
        return torch.optim.Adadelta(params, lr=lr, weight_decay=weight_decay)
    if opt == "adam":
        return torch.optim.Adam(params, lr=lr, betas=(0.5, 0.999), weight_decay=weight_decay)
    raise NotImplementedError(opt)
