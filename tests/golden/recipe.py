"""Deterministic, portable recipes for weights and inputs used by the golden fixtures.

numpy's legacy ``RandomState`` (MT19937) stream is stable across numpy versions, so the same
tensors can be regenerated wherever the tests run -- the 104 MB state_dict of drn_d_38 is
never stored, only the recipe and the reference's outputs.
"""
import math

import numpy as np
import torch


def fill_state_(module, seed):
    """Overwrite every parameter and buffer of ``module`` (in sorted key order) with
    recipe values: He-normal conv kernels, perturbed BN affine + running stats."""
    rs = np.random.RandomState(seed)
    sd = module.state_dict()
    for key in sorted(sd.keys()):
        t = sd[key]
        shape = tuple(t.shape)
        leaf = key.split(".")[-1]
        if leaf == "num_batches_tracked":
            t.zero_()
            continue
        x = rs.standard_normal(shape).astype(np.float64)
        if t.dim() == 4:  # conv / transposed-conv kernel
            if "up" in key.split(".")[-2]:
                x *= 0.06
            else:
                x *= math.sqrt(2.0 / (shape[0] * shape[2] * shape[3]))
        elif leaf == "weight":  # BN gamma
            x = 1.0 + 0.1 * x
        elif leaf == "running_var":
            x = 1.0 + 0.1 * np.abs(x)
        elif leaf == "running_mean":
            x = 0.05 * x
        else:  # BN beta, conv bias, log-variance scalars
            x = 0.05 * x
        t.copy_(torch.from_numpy(x).to(t.dtype))
    return module


def make_batch(seed, n, ch, h, w, n_class):
    """(src_imgs, src_lbls, tgt_imgs): N(0,1) images, labels uniform in [0, n_class) --
    class n_class-1 is the zero-weight background."""
    rs = np.random.RandomState(seed)
    src = torch.from_numpy(rs.standard_normal((n, ch, h, w)).astype(np.float32))
    lbl = torch.from_numpy(rs.randint(0, n_class, size=(n, h, w)).astype(np.int64))
    tgt = torch.from_numpy(rs.standard_normal((n, ch, h, w)).astype(np.float32))
    return src, lbl, tgt


def checksum(t):
    t = t.detach().double().reshape(-1)
    return [float(t.sum()), float(t.norm())]


def state_checksums(module):
    return {k: checksum(v) for k, v in module.state_dict().items()}


def fusion_inputs(seed, n_class):
    """(x1, x2, grad_output, labels) of the fusion-classifier fixtures (make_golden_fusion.py)"""
    rs = np.random.RandomState(seed)
    x1 = torch.from_numpy(rs.standard_normal((2, n_class, 2, 4)).astype(np.float32))
    x2 = torch.from_numpy(rs.standard_normal((2, n_class, 2, 4)).astype(np.float32))
    gy = torch.from_numpy(rs.standard_normal((2, n_class, 16, 32)).astype(np.float32))
    lbl = torch.from_numpy(rs.randint(0, n_class, size=(2, 16, 32)).astype(np.int64))
    return x1, x2, gy, lbl
