"""Model / optimizer factory with the reference's names and signatures (models/model_util.py:6-39, 160-310).

Only the DRN branches -- the ones the MCD hot path uses -- are implemented; every other network name
raises ``NotImplementedError`` (the reference's message is kept).  ``get_optimizer('sgd')`` returns the
flat-buffer HIP SGD (``mcdseg.optim.FlatSGD``), state-dict compatible with ``torch.optim.SGD``.
"""
import os

import torch
from torch import nn
from torch.nn.modules.batchnorm import _BatchNorm

DRN_NAMES = ["drn_c_26", "drn_c_42", "drn_c_58", "drn_d_22", "drn_d_38", "drn_d_54", "drn_d_105"]


def _pretrained_default():
    # the reference always asks for ImageNet weights (a download); MCDSEG_PRETRAINED=0 / --no_pretrained turns it off
    return os.environ.get("MCDSEG_PRETRAINED", "1") != "0"


def _wrap(models, is_data_parallel):
    """``nn.DataParallel`` in the reference is single-process multi-GPU.  This build is one process per GPU
    (RCCL all-reduce inside the optimizer, ``mcdseg.dist``), so the wrapper only has to reproduce the
    ``module.`` prefix of DataParallel checkpoints."""
    if not is_data_parallel:
        return models
    wrap = lambda m: nn.DataParallel(m, device_ids=[torch.cuda.current_device()] if torch.cuda.is_available() else None)
    return [wrap(m) for m in models] if isinstance(models, (list, tuple)) else wrap(models)


def get_full_model(net, res, n_class, input_ch, is_data_parallel=True):
    if "drn" in net:
        from models.dilated_fcn import DRNSeg
        assert net in DRN_NAMES
        model = DRNSeg(net, n_class, input_ch=input_ch, pretrained=_pretrained_default())
    else:
        raise NotImplementedError("Only FCN, SegNet, PSPNet, DRNet, UNet are supported!")
    return _wrap(model, is_data_parallel)


def get_models(net_name, input_ch, n_class, res="50", method="MCD", is_data_parallel=False):
    if "drn" not in net_name or "fusenet" in net_name:
        raise NotImplementedError("Only FCN (Including Dilated FCN), SegNet, PSPNet UNet are supported!")
    from models.dilated_fcn import (DRNSegBase, DRNSegPixelClassifier, FusionDRNSegPixelClassifier,
                                    ScoreFusionDRNSegPixelClassifier)
    ver = "ver2" if "ver2" in net_name else "ver1"
    drn_name = net_name.replace("_ver2", "")
    pre = _pretrained_default()
    if method == "MCD":
        model_list = [DRNSegBase(model_name=drn_name, n_class=n_class, input_ch=input_ch, ver=ver, pretrained=pre),
                      DRNSegPixelClassifier(n_class=n_class, ver=ver), DRNSegPixelClassifier(n_class=n_class, ver=ver)]
    elif "MFNet" in method:
        assert input_ch in [4, 6]
        fusion_type = method.split("-")[-1]
        print("fusion type: %s" % fusion_type)
        g3 = DRNSegBase(model_name=drn_name, n_class=n_class, input_ch=3, ver=ver, pretrained=pre)
        g1 = DRNSegBase(model_name=drn_name, n_class=n_class, input_ch=input_ch - 3, ver=ver, pretrained=pre)
        if "score" in method.lower():
            print("Score Fusion!!!")
            fs = [ScoreFusionDRNSegPixelClassifier(fusion_type=fusion_type, n_class=n_class) for _ in range(2)]
        else:
            fs = [FusionDRNSegPixelClassifier(fusion_type=fusion_type, n_class=n_class, ver=ver) for _ in range(2)]
        model_list = [g3, g1] + fs
    else:
        return NotImplementedError("Sorry... Only MCD is supported!")  # returned, not raised: models/model_util.py:281
    return _wrap(model_list, is_data_parallel)


def get_multitask_models(net_name, input_ch, n_class, semseg_criterion=None, discrepancy_criterion=None,
                         is_data_parallel=False, is_src_only=False):
    """RGB encoder + MCD multitask decoder (models/model_util.py:81-99); the source-only decoder variant is not on
    the hot path."""
    if "drn" not in net_name:
        raise NotImplementedError("Only FCN (Including Dilated FCN), SegNet, PSPNet UNet are supported!")
    if is_src_only:
        raise NotImplementedError("MultiTaskDecoder (source-only) is outside the MCD hot path")
    from models.dilated_fcn import MCDMultiTaskDecoder, MultiTaskEncoder
    model_enc = MultiTaskEncoder(model_name=net_name, input_ch=3, pretrained=_pretrained_default())  # RGB is 3 channel
    model_dec = MCDMultiTaskDecoder(n_class=n_class, depth_ch=input_ch - 3, semseg_criterion=semseg_criterion,
                                    discrepancy_criterion=discrepancy_criterion)
    if is_data_parallel:
        return _wrap(model_enc, True), _wrap(model_dec, True)
    return model_enc, model_dec


def get_optimizer(model_parameters, opt, lr, momentum, weight_decay):
    params = [p for p in model_parameters if p.requires_grad]
    if opt == "sgd":
        from mcdseg.optim import FlatSGD
        return FlatSGD(params, lr=lr, momentum=momentum, weight_decay=weight_decay)
    elif opt == "adadelta":
        return torch.optim.Adadelta(params, lr=lr, weight_decay=weight_decay)
    elif opt == "adam":
        return torch.optim.Adam(params, lr=lr, betas=[0.5, 0.999], weight_decay=weight_decay)
    raise NotImplementedError("Only (Momentum) SGD, Adadelta, Adam are supported!")


def fix_batchnorm_when_training(model):
    if issubclass(type(model), _BatchNorm):
        model.training = False
    for module in model.children():
        fix_batchnorm_when_training(module)


def fix_dropout_when_training(model):
    if type(model) in [nn.Dropout, nn.Dropout2d, nn.Dropout3d, nn.AlphaDropout]:
        model.training = False
        print("Fixed one dropout layer")
    for module in model.children():
        fix_dropout_when_training(module)


def check_training(model):
    print(type(model))
    print(model.training)
    for module in model.children():
        check_training(module)
