"""Oracle restatement of the multitask (segmentation + HHA regression) MCD variant -- BASELINE config 4.

TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py``.

* ``MultiTaskEncoder``      models/dilated_fcn.py:554-566   DRN trunk on the RGB channels only
* ``CBR``                   :632-644                         conv (with bias) - BN - ReLU
* ``ThreeLayerDecoder``     :647-658                         CBR 3x3, CBR 1x1, conv 1x1
* ``MCDMultiTaskDecoder``   :661-739                         two segmentation heads + one depth head, bilinear x8,
                                                             learned log-variance task weights exp(-s) L + s
* ``get_multitask_models``  models/model_util.py:81-99
* ``multitask_mcd_step``    adapt_multitask_trainer.py:166-239 (statement for statement, including the unused
                            ``semseg_forward`` of step B that still moves BatchNorm running statistics)
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .ref_models import drn_trunk


class MultiTaskEncoder(nn.Module):
    def __init__(self, model_name, input_ch=3):
        super().__init__()
        self.base, _ = drn_trunk(model_name, input_ch)

    def forward(self, x):
        return self.base(x)


class CBR(nn.Module):
    def __init__(self, cin, cout, kernel_size, padding=0):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, kernel_size, padding=padding, bias=True)
        self.bn = nn.BatchNorm2d(cout)

    def forward(self, x):
        return F.relu(self.bn(self.conv(x)))


class ThreeLayerDecoder(nn.Module):
    def __init__(self, output_ch, input_ch=512):
        super().__init__()
        self.cbr1 = CBR(input_ch, 512, 3, padding=1)
        self.cbr2 = CBR(512, 512, 1)
        self.conv3 = nn.Conv2d(512, output_ch, 1)

    def forward(self, x):
        return self.conv3(self.cbr2(self.cbr1(x)))


class MCDMultiTaskDecoder(nn.Module):
    def __init__(self, n_class, depth_ch, semseg_criterion=None, discrepancy_criterion=None):
        super().__init__()
        self.s_semsegcls = nn.Parameter(torch.ones(1))  # log-variance of the segmentation task, initialised to 1
        self.s_deprgr = nn.Parameter(torch.ones(1))
        self.semsegcls_dec1 = ThreeLayerDecoder(n_class)
        self.semsegcls_dec2 = ThreeLayerDecoder(n_class)
        self.deprgr_dec = ThreeLayerDecoder(depth_ch)
        self.semseg_criterion = semseg_criterion
        self.discrepancy_criterion = discrepancy_criterion

    @staticmethod
    def upsample(x):
        return F.interpolate(x, scale_factor=8, mode="bilinear", align_corners=False)

    def semseg_forward(self, x):
        return self.upsample(self.semsegcls_dec1(x)), self.upsample(self.semsegcls_dec2(x))

    def depth_forward(self, x):
        return self.upsample(self.deprgr_dec(x))

    def forward(self, x):
        a, b = self.semseg_forward(x)
        return a, b, self.depth_forward(x)

    def get_cls_descrepancy(self, x):
        a, b = self.semseg_forward(x)
        return self.discrepancy_criterion(a, b)

    def get_semseg_loss(self, x, gt_semseg, separately_returning=False):
        a, b = self.semseg_forward(x)
        l1, l2 = self.semseg_criterion(a, gt_semseg), self.semseg_criterion(b, gt_semseg)
        return (l1, l2) if separately_returning else l1 + l2

    def get_depth_loss(self, x, gt_dep):
        return F.mse_loss(self.depth_forward(x), gt_dep)

    def get_loss(self, x, gt_semseg, gt_dep, separately_returning=False):
        l1, l2 = self.get_semseg_loss(x, gt_semseg, separately_returning=True)
        s = self.s_semsegcls
        semseg = ((torch.exp(-s) * l1 + s) + (torch.exp(-s) * l2 + s)) / 2
        dep = torch.exp(-self.s_deprgr) * self.get_depth_loss(x, gt_dep) + self.s_deprgr
        return (semseg, dep) if separately_returning else semseg + dep


def get_multitask_models(net_name, input_ch, n_class, semseg_criterion=None, discrepancy_criterion=None,
                         is_data_parallel=False, is_src_only=False):
    if "drn" not in net_name or is_src_only:
        raise NotImplementedError("oracle covers the MCD multitask DRN decoder only")
    enc = MultiTaskEncoder(net_name, input_ch=3)
    dec = MCDMultiTaskDecoder(n_class, input_ch - 3, semseg_criterion, discrepancy_criterion)
    if is_data_parallel:
        return nn.DataParallel(enc), nn.DataParallel(dec)
    return enc, dec


def multitask_mcd_step(model_enc, model_dec, optimizer_enc, optimizer_dec, src_imgs, src_gt_semseg, tgt_imgs, num_k=4,
                       num_multiply_d_loss=1):
    src_rgbs, src_depths = src_imgs[:, :3], src_imgs[:, 3:]
    tgt_rgbs, tgt_depths = tgt_imgs[:, :3], tgt_imgs[:, 3:]

    optimizer_enc.zero_grad()
    optimizer_dec.zero_grad()
    src_fet = model_enc(src_rgbs)
    tgt_fet = model_enc(tgt_rgbs)
    src_semseg_loss, src_depth_loss = model_dec.get_loss(src_fet, src_gt_semseg, src_depths, separately_returning=True)
    tgt_depth_loss = model_dec.get_depth_loss(tgt_fet, tgt_depths)
    loss = src_semseg_loss + src_depth_loss + tgt_depth_loss
    loss.backward()
    c_loss = float(loss.detach())
    optimizer_enc.step()
    optimizer_dec.step()

    optimizer_enc.zero_grad()
    optimizer_dec.zero_grad()
    src_fet = model_enc(src_rgbs)
    model_dec.semseg_forward(src_fet)  # result unused in the reference (adapt_multitask_trainer.py:208); BN stats move
    src_semseg_loss, src_depth_loss = model_dec.get_loss(src_fet, src_gt_semseg, src_depths, separately_returning=True)
    tgt_fet = model_enc(tgt_rgbs)
    tgt_depth_loss = model_dec.get_depth_loss(tgt_fet, tgt_depths)
    tgt_discrepancy = model_dec.get_cls_descrepancy(tgt_fet)
    loss = src_semseg_loss + src_depth_loss + tgt_depth_loss - tgt_discrepancy
    loss.backward()
    optimizer_dec.step()

    for _ in range(num_k):
        optimizer_enc.zero_grad()
        tgt_fet = model_enc(tgt_rgbs)
        loss = model_dec.get_cls_descrepancy(tgt_fet) * num_multiply_d_loss
        loss.backward()
        optimizer_enc.step()
    d_loss = float(loss.detach()) / num_k
    return c_loss, d_loss, (float(src_semseg_loss.detach()), float(src_depth_loss.detach()), float(tgt_depth_loss.detach()))
