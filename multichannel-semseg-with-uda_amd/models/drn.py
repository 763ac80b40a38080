"""Dilated Residual Networks (DRN-C / DRN-D trunks) on the MI355X HIP kernels.

Counterpart of the reference's ``models/drn.py`` (``DRN`` :103-253, ``BasicBlock`` :26-59,
``Bottleneck`` :62-100, ``replace_first_conv`` :256-299, constructors :302-348).  The module tree --
and therefore every ``state_dict`` key and shape -- is the reference's; the arithmetic is not torch's:
each conv+BN(+ReLU)(+residual) group is executed by ``mcdseg.ops.conv_bn_act`` (implicit-GEMM MFMA
convolution with the BatchNorm statistics fused into its epilogue, then one fused normalise /
residual / ReLU pass).  ``Conv2d`` / ``BatchNorm2d`` below are parameter holders whose parents do the
fusing; they keep ``nn.Conv2d`` / ``nn.BatchNorm2d`` as base classes so that ``isinstance`` checks
such as ``fix_batchnorm_when_training`` (models/model_util.py:305-310) keep working.
"""
import glob
import math
import os
import warnings

import torch
import torch.nn as nn

from mcdseg import ops

__all__ = ["DRN", "BasicBlock", "Bottleneck", "drn_c_26", "drn_c_42", "drn_c_58", "drn_d_22", "drn_d_38", "drn_d_54",
           "drn_d_105"]

CHANNELS = (16, 32, 64, 128, 256, 512, 512, 512)


class Conv2d(nn.Conv2d):
    """Convolution parameters + the packed GEMM images the HIP kernels read."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        if self.groups != 1 or self.kernel_size[0] != self.kernel_size[1] or self.stride[0] != self.stride[1] \
                or self.padding[0] != self.padding[1] or self.dilation[0] != self.dilation[1]:
            raise NotImplementedError("mcdseg convolutions are square, ungrouped and symmetric")
        self._packed = ops.PackedWeights()

    def forward(self, x):  # stand-alone use (no BatchNorm behind it), e.g. the 1x1 ``seg`` head
        return ops.conv2d_bias(x, self)


class BatchNorm2d(nn.BatchNorm2d):
    """BatchNorm parameters / running statistics.  Never runs alone: the owning block fuses it with the
    convolution in front of it (statistics come out of the conv epilogue)."""

    def forward(self, x):
        raise RuntimeError("mcdseg BatchNorm2d is fused into its convolution; call the parent block")


class FusedSequential(nn.Sequential):
    """``nn.Sequential`` that executes every (Conv2d, BatchNorm2d[, ReLU]) run of its children as one fused
    HIP group; any other child is called as usual.  Covers ``_make_conv_layers`` stages
    (models/drn.py:195-205), the 1x1 projection shortcuts (:175-180) and the DRN-C stem (:118-121)."""

    def forward(self, x):
        return run_fused(list(self.children()), x)


def run_fused(mods, x, internal_last=False, following=None):
    """the fusing walk over a list of modules: (Conv2d, BatchNorm2d[, ReLU]) runs become one fused group each.  A ReLU group followed by
    another fused group of the same list feeds that convolution only, and so does the last one when the caller says so
    (``internal_last``: the next stage's convolutions are its only readers): such a group need not write its fp32 output
    (``ops.conv_bn_act(..., internal=True)``; the same bits).  A child that is itself a convolution chain (the stages of a DRN as
    children of ``models.dilated_fcn.Trunk``) is walked the same way, its last group internal when the child behind it --
    ``following`` for the last of ``mods`` -- reads it through convolutions only."""
    i = 0
    while i < len(mods):
        m = mods[i]
        if isinstance(m, Conv2d) and i + 1 < len(mods) and isinstance(mods[i + 1], BatchNorm2d):
            relu = i + 2 < len(mods) and isinstance(mods[i + 2], nn.ReLU)
            nxt = i + (3 if relu else 2)
            feeds_conv = (nxt + 1 < len(mods) and isinstance(mods[nxt], Conv2d) and isinstance(mods[nxt + 1], BatchNorm2d)) \
                or (nxt >= len(mods) and internal_last)
            x = ops.conv_bn_act(x, m, mods[i + 1], relu=relu, internal=bool(relu and feeds_conv and ops.INTERNAL_STAGES), thin_ok=True)
            i = nxt
        elif type(m) is FusedSequential and not _has_hooks(m):  # a stage of plain convolutions (one with hooks is CALLED: below)
            nxt_stage = mods[i + 1] if i + 1 < len(mods) else following
            x = run_fused(list(m.children()), x, internal_last=nxt_stage is not None and _reads_companions_only(nxt_stage))
            i += 1
        else:
            x = m(x)
            i += 1
    return x


def _has_hooks(module):
    """forward (pre-)hooks registered on a stage: such a stage is called as a module -- its hooks fire and see a real fp32 output -- instead
    of being walked group by group with a companion-only hand-over"""
    return bool(module._forward_hooks or module._forward_pre_hooks)


def _reads_companions_only(stage):
    """the stage that follows a convolution chain reads the chain's output through convolutions only (never as an identity shortcut, never
    as a tensor it hands on): another convolution chain, or residual blocks whose first block projects its shortcut"""
    if isinstance(stage, FusedSequential):
        mods = list(stage.children())
        return len(mods) >= 2 and isinstance(mods[0], Conv2d) and isinstance(mods[1], BatchNorm2d) and _companion_conv(mods[0])
    if isinstance(stage, nn.Sequential) and len(stage) > 0 and isinstance(stage[0], (BasicBlock, Bottleneck)):
        b = stage[0]
        ds = list(b.downsample.children()) if isinstance(b.downsample, FusedSequential) else []
        return (len(ds) == 2 and isinstance(ds[0], Conv2d) and isinstance(ds[1], BatchNorm2d) and _companion_conv(ds[0])
                and _companion_conv(b.conv1))
    return False


def _companion_conv(conv):
    """forward, data and weight gradient of this convolution all read the pre-split companion of its input (16 channels on the thin
    layers' window kernels, 24 and more on the tap-pair / 128-wide / ping-pong plans)"""
    return conv.in_channels >= 16 and conv.in_channels % 8 == 0 and conv.out_channels % 8 == 0 and conv.groups == 1


ConvBNReLU = FusedSequential
ConvBN = FusedSequential


def conv3x3(cin, cout, stride=1, padding=1, dilation=1):
    return Conv2d(cin, cout, kernel_size=3, stride=stride, padding=padding, bias=False, dilation=dilation)


def _project(downsample, x, box):
    """the 1x1 projection shortcut of a block (a ``ConvBN``): as a fused group whose input gradient goes through the block's GradBox"""
    mods = list(downsample.children()) if isinstance(downsample, FusedSequential) else []
    if box is not None and len(mods) == 2 and isinstance(mods[0], Conv2d) and isinstance(mods[1], BatchNorm2d):
        return ops.conv_bn_act(x, mods[0], mods[1], relu=False, in_box=box, shortcut_only=True)
    return downsample(x)


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None, dilation=(1, 1), residual=True):
        super().__init__()
        self.conv1 = conv3x3(inplanes, planes, stride, padding=dilation[0], dilation=dilation[0])
        self.bn1 = BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = conv3x3(planes, planes, padding=dilation[1], dilation=dilation[1])
        self.bn2 = BatchNorm2d(planes)
        self.downsample = downsample
        self.stride = stride
        self.residual = residual

    def forward(self, x):
        # the block input's gradient has two producers, conv1's data gradient and the shortcut: summed through a GradBox (ops.GradBox)
        box = ops.grad_box(x) if self.residual else None
        h = ops.conv_bn_act(x, self.conv1, self.bn1, relu=True, internal=True, in_box=box)  # only conv2 below reads h
        shortcut = None
        if self.residual:
            shortcut = x if self.downsample is None else _project(self.downsample, x, box)
        return ops.conv_bn_act(h, self.conv2, self.bn2, relu=True, residual=shortcut,
                               res_box=box if self.downsample is None else None)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None, dilation=(1, 1), residual=True):
        super().__init__()
        self.conv1 = Conv2d(inplanes, planes, kernel_size=1, bias=False)
        self.bn1 = BatchNorm2d(planes)
        self.conv2 = Conv2d(planes, planes, kernel_size=3, stride=stride, padding=dilation[1], bias=False,
                            dilation=dilation[1])
        self.bn2 = BatchNorm2d(planes)
        self.conv3 = Conv2d(planes, planes * 4, kernel_size=1, bias=False)
        self.bn3 = BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        box = ops.grad_box(x)  # (as in BasicBlock)
        h = ops.conv_bn_act(x, self.conv1, self.bn1, relu=True, internal=True, in_box=box)  # only the next convolution reads these two
        h = ops.conv_bn_act(h, self.conv2, self.bn2, relu=True, internal=True)
        shortcut = x if self.downsample is None else _project(self.downsample, x, box)
        return ops.conv_bn_act(h, self.conv3, self.bn3, relu=True, residual=shortcut,
                               res_box=box if self.downsample is None else None)


class DRN(nn.Module):
    """Trunk layer0/conv1 .. layer8 (+ avgpool/fc when ``num_classes > 0``, kept so that
    ``list(model.children())[:-2]`` is the segmentation trunk as in models/dilated_fcn.py:223)."""

    def __init__(self, block, layers, num_classes=1000, channels=CHANNELS, out_map=False, out_middle=False, pool_size=28,
                 arch="D"):
        super().__init__()
        self.inplanes = channels[0]
        self.out_map = out_map
        self.out_dim = channels[-1]
        self.out_middle = out_middle
        self.arch = arch

        if arch == "C":
            self.conv1 = Conv2d(3, channels[0], kernel_size=7, stride=1, padding=3, bias=False)
            self.bn1 = BatchNorm2d(channels[0])
            self.relu = nn.ReLU(inplace=True)
            self.layer1 = self._make_layer(BasicBlock, channels[0], layers[0], stride=1)
            self.layer2 = self._make_layer(BasicBlock, channels[1], layers[1], stride=2)
        elif arch == "D":
            self.layer0 = ConvBNReLU(Conv2d(3, channels[0], kernel_size=7, stride=1, padding=3, bias=False),
                                     BatchNorm2d(channels[0]), nn.ReLU(inplace=True))
            self.layer1 = self._make_conv_layers(channels[0], layers[0], stride=1)
            self.layer2 = self._make_conv_layers(channels[1], layers[1], stride=2)
        else:
            raise ValueError("arch must be 'C' or 'D'")

        self.layer3 = self._make_layer(block, channels[2], layers[2], stride=2)
        self.layer4 = self._make_layer(block, channels[3], layers[3], stride=2)
        self.layer5 = self._make_layer(block, channels[4], layers[4], dilation=2, new_level=False)
        self.layer6 = None if layers[5] == 0 else self._make_layer(block, channels[5], layers[5], dilation=4, new_level=False)
        if arch == "C":
            self.layer7 = None if layers[6] == 0 else self._make_layer(BasicBlock, channels[6], layers[6], dilation=2,
                                                                       new_level=False, residual=False)
            self.layer8 = None if layers[7] == 0 else self._make_layer(BasicBlock, channels[7], layers[7], dilation=1,
                                                                       new_level=False, residual=False)
        else:
            self.layer7 = None if layers[6] == 0 else self._make_conv_layers(channels[6], layers[6], dilation=2)
            self.layer8 = None if layers[7] == 0 else self._make_conv_layers(channels[7], layers[7], dilation=1)

        if num_classes > 0:
            self.avgpool = nn.AvgPool2d(pool_size)
            self.fc = Conv2d(self.out_dim, num_classes, kernel_size=1, stride=1, padding=0, bias=True)
        init_he_normal_(self)

    def _make_layer(self, block, planes, blocks, stride=1, dilation=1, new_level=True, residual=True):
        assert dilation == 1 or dilation % 2 == 0
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = ConvBN(Conv2d(self.inplanes, planes * block.expansion, kernel_size=1, stride=stride, bias=False),
                                BatchNorm2d(planes * block.expansion))
        first = (1, 1) if dilation == 1 else (dilation // 2 if new_level else dilation, dilation)
        seq = [block(self.inplanes, planes, stride, downsample, dilation=first, residual=residual)]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            seq.append(block(self.inplanes, planes, residual=residual, dilation=(dilation, dilation)))
        return nn.Sequential(*seq)

    def _make_conv_layers(self, channels, convs, stride=1, dilation=1):
        mods = []
        for i in range(convs):
            mods += [Conv2d(self.inplanes, channels, kernel_size=3, stride=stride if i == 0 else 1, padding=dilation,
                            bias=False, dilation=dilation), BatchNorm2d(channels), nn.ReLU(inplace=True)]
            self.inplanes = channels
        return ConvBNReLU(*mods)

    def trunk(self):
        """The segmentation encoder: every child except avgpool / fc."""
        return [m for name, m in self.named_children() if name not in ("avgpool", "fc")]

    def forward(self, x):
        with ops.late_weight_grads(self):
            return self._forward(x)

    def _forward(self, x):
        feats = []
        if self.arch == "C":
            x = ops.conv_bn_act(x, self.conv1, self.bn1, relu=True)
            stages = (self.layer1, self.layer2, self.layer3, self.layer4, self.layer5, self.layer6, self.layer7, self.layer8)
        else:
            x = self.layer0(x)
            stages = (self.layer1, self.layer2, self.layer3, self.layer4, self.layer5, self.layer6, self.layer7, self.layer8)
        live = [st for st in stages if st is not None]
        for k, st in enumerate(live):
            if type(st) is FusedSequential and not _has_hooks(st):
                # a convolution chain whose output only the next stage's convolutions read hands on its companion alone
                # (a subclass, or a stage with forward hooks, is called like any module and writes its fp32 output)
                last_internal = (not self.out_middle and k + 1 < len(live) and _reads_companions_only(live[k + 1]))
                x = run_fused(list(st.children()), x, internal_last=last_internal)
            else:
                x = st(x)
            feats.append(x)
        if not hasattr(self, "fc"):
            return (x, feats) if self.out_middle else x
        if self.out_map:
            x = self.fc(x)
        else:
            raise NotImplementedError("ImageNet classification head (avgpool + fc) is outside the MCD hot path")
        return (x, feats) if self.out_middle else x


def init_he_normal_(module):
    """Conv ~ N(0, sqrt(2 / (kh*kw*Cout))), BN gamma 1 / beta 0 (models/drn.py:163-169)."""
    for m in module.modules():
        if isinstance(m, nn.Conv2d):
            n = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
            m.weight.data.normal_(0, math.sqrt(2.0 / n))
        elif isinstance(m, nn.BatchNorm2d):
            m.weight.data.fill_(1)
            m.bias.data.zero_()


def replace_first_conv(model, input_ch, arch):
    """Adapt the 7x7 stem to ``input_ch`` channels (models/drn.py:256-299): 3 -> unchanged; 1 -> the R slice;
    4..6 -> RGB kernel plus its first ``input_ch - 3`` slices again; anything else is rejected."""
    if input_ch == 3:
        return model
    old = model.conv1 if arch == "C" else model.layer0[0]
    new = Conv2d(input_ch, 16, kernel_size=7, stride=1, padding=3, bias=False)
    if input_ch == 1:
        new.weight.data = old.weight.data[:, 0:1].clone()
    elif 3 < input_ch <= 6:
        extra = input_ch - 3
        new.weight.data[:, :3] = old.weight.data
        new.weight.data[:, 3:3 + extra] = old.weight.data[:, :extra]
    else:
        raise NotImplementedError()
    if arch == "C":
        model.conv1 = new
    else:
        model.layer0 = ConvBNReLU(new, model.layer0[1], nn.ReLU(inplace=True))
    return model


def _load_pretrained(model, name):
    """The reference downloads ImageNet weights (models/drn.py:8-18).  There is no network on the GPU box: weights are
    taken from $MCDSEG_PRETRAINED_DIR/<name>-*.pth.  Missing weights are an error -- a run that silently starts from
    He-normal initialisation is not the reference's run -- unless MCDSEG_PRETRAINED=0 / --no_pretrained opts out."""
    if os.environ.get("MCDSEG_PRETRAINED", "1") == "0":
        return
    root = os.environ.get("MCDSEG_PRETRAINED_DIR", "")
    hits = sorted(glob.glob(os.path.join(root, name.replace("-", "_") + "-*.pth"))) if root else []
    if not hits:
        raise FileNotFoundError("mcdseg: pretrained weights for %s not found: put %s-*.pth under $MCDSEG_PRETRAINED_DIR, or set "
                                "MCDSEG_PRETRAINED=0 (trainers: --no_pretrained) to train from He-normal initialisation"
                                % (name, name.replace("-", "_")))
    sd = torch.load(hits[0], map_location="cpu")
    model.load_state_dict(sd, strict=False)


def _build(block, layers, arch, name, pretrained, input_ch, **kwargs):
    model = DRN(block, layers, arch=arch, **kwargs)
    if pretrained:
        _load_pretrained(model, name)
    return replace_first_conv(model, input_ch=input_ch, arch=arch)


def drn_c_26(pretrained=False, input_ch=3, **kw):
    return _build(BasicBlock, [1, 1, 2, 2, 2, 2, 1, 1], "C", "drn-c-26", pretrained, input_ch, **kw)


def drn_c_42(pretrained=False, input_ch=3, **kw):
    return _build(BasicBlock, [1, 1, 3, 4, 6, 3, 1, 1], "C", "drn-c-42", pretrained, input_ch, **kw)


def drn_c_58(pretrained=False, input_ch=3, **kw):
    return _build(Bottleneck, [1, 1, 3, 4, 6, 3, 1, 1], "C", "drn-c-58", pretrained, input_ch, **kw)


def drn_d_22(pretrained=False, input_ch=3, **kw):
    return _build(BasicBlock, [1, 1, 2, 2, 2, 2, 1, 1], "D", "drn-d-22", pretrained, input_ch, **kw)


def drn_d_38(pretrained=False, input_ch=3, **kw):
    return _build(BasicBlock, [1, 1, 3, 4, 6, 3, 1, 1], "D", "drn-d-38", pretrained, input_ch, **kw)


def drn_d_54(pretrained=False, input_ch=3, **kw):
    return _build(Bottleneck, [1, 1, 3, 4, 6, 3, 1, 1], "D", "drn-d-54", pretrained, input_ch, **kw)


def drn_d_105(pretrained=False, input_ch=3, **kw):
    return _build(Bottleneck, [1, 1, 3, 4, 23, 3, 1, 1], "D", "drn-d-105", pretrained, input_ch, **kw)
