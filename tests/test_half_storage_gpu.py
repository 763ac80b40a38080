"""GPU: the 2-byte activation storage of the one-term arithmetic (BASELINE config 5 "bf16"; include/mcdseg.h "2-byte activation
storage", mcdseg/ops.py HALF_STORAGE) -- every kernel of the chain through the C ABI against what it is DEFINED to compute:

  * the convolutions' 16-bit epilogues against their own fp32 epilogues (same accumulators, so the 16-bit tensors must be the fp32
    ones rounded ONCE: bitwise), with the BatchNorm partial rows untouched;
  * the three BatchNorm kernels against an fp64 statement of the same maps on the same 16-bit inputs;
  * a Bottleneck block (models/drn.py:62-100 of the reference) through the chain against the same block in round 5's f16x1 storage.
"""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def _scale(bound):
    b = float(bound)
    return 2.0 ** (math.frexp(b)[1] - 15) if b > 0 else 1.0


def _units_to_nchw(t, n, c, h, w):
    """[N][C/8][HW][8] -> [N, C, H, W] (any dtype)"""
    return t.reshape(n, c // 8, h * w, 8).permute(0, 1, 3, 2).reshape(n, c, h, w)


def _nchw_to_units(t):
    n, c, h, w = t.shape
    return t.reshape(n, c // 8, 8, h * w).permute(0, 1, 3, 2).contiguous()


# (Cin, Cout, k, stride, dil, H, W, N): 4-wave tiles (64 / 128 rows), ping-pong tiles (256+ rows, several plans), stride 2 (parity classes)
HALF_CONVS = [(64, 64, 3, 1, 1, 12, 20, 3), (256, 64, 1, 1, 1, 11, 13, 2), (64, 256, 1, 1, 1, 11, 13, 2), (128, 128, 3, 1, 2, 13, 19, 2),
              (256, 512, 3, 1, 4, 30, 40, 6), (1024, 256, 1, 1, 1, 30, 40, 4), (256, 256, 3, 1, 2, 60, 80, 4), (64, 128, 3, 2, 1, 15, 17, 2),
              (32, 256, 1, 2, 1, 20, 28, 2)]


@pytest.mark.parametrize("case", HALF_CONVS, ids=lambda c: "x".join(map(str, c)))
def test_conv_half_epilogues_round_the_fp32_results_once(case, monkeypatch):
    dev = _dev()
    from mcdseg import ops
    from mcdseg._lib import lib
    import ctypes
    monkeypatch.setattr(ops, "CONV_MATH", "f16x1")
    cin, cout, k, s, d, h, w, n = case
    g = torch.Generator().manual_seed(61)
    x = torch.randn(n, cin, h, w, generator=g).to(dev)
    wt = (torch.randn(cout, cin, k, k, generator=g) * (2.0 / (k * k * cout)) ** 0.5).to(dev)
    pad = d * (k // 2)
    desc = ops.conv_desc(x.shape, wt.shape, s, pad, d)
    assert lib().mcdseg_conv_split_half_ok(ctypes.byref(desc), 1, 0) == 1 and lib().mcdseg_conv_split_half_ok(ctypes.byref(desc), 1, 1) == 1
    gy = torch.randn(n, cout, desc.Ho, desc.Wo, generator=g).to(dev)
    pk = ops.PackedWeights()
    wf, wd, mpf = pk.get(wt, desc)
    x_cb, x_bound = ops.split_companion(x)
    gy_cb, gy_bound = ops.split_companion(gy)
    # forward: fp32 epilogue vs 16-bit epilogue
    z, part, rows = ops._conv_fprop(desc, x, wf, None, True, mpf, x_cb, x_bound, pk.w_bound)
    z16, z_bound, part16, rows16 = ops._conv_fprop_half(desc, x_cb, x_bound, wf, pk.w_bound, mpf)
    torch.cuda.synchronize()
    assert rows == rows16 and torch.equal(part, part16), "the BatchNorm partial rows come from the fp32 accumulators either way"
    zb = float(z_bound)
    assert zb == float(np.float32(np.float32(k * k * cin) * np.float32(float(x_bound))) * np.float32(float(pk.w_bound)))
    assert float(z.abs().max()) <= zb
    zs = _scale(zb)
    want = (z / zs).to(torch.float16)
    got = _units_to_nchw(z16.view(torch.float16), n, cout, desc.Ho, desc.Wo)
    assert torch.equal(got, want), "z16 is the fp32 z rounded once to scaled fp16"
    assert float((got.float() * zs - z).abs().max()) <= 2.0 ** -11 * float(z.abs().max()) + 2.0 ** -24 * zs
    # data gradient: fp32 epilogue vs bf16 epilogue, without and with an addend
    dx = ops._conv_dgrad(desc, None, wd, gy_cb, gy_bound, pk.w_bound)
    dx16 = ops._conv_dgrad_half(desc, gy_cb, gy_bound, wd, pk.w_bound)
    assert dx16.dtype == torch.bfloat16 and dx16.shape == x.shape
    assert torch.equal(_units_to_nchw(dx16, n, cin, h, w), dx.to(torch.bfloat16))
    add = (torch.randn(n, cin, h, w, generator=g) * float(dx.abs().mean())).to(dev).to(torch.bfloat16)
    add_units = _nchw_to_units(add).reshape(n, cin, h, w)
    dx16a = ops._conv_dgrad_half(desc, gy_cb, gy_bound, wd, pk.w_bound, add_units)
    assert torch.equal(_units_to_nchw(dx16a, n, cin, h, w), (dx + add.float()).to(torch.bfloat16)), "one rounding after the fp32 sum"
    # a one-piece companion serves the consumers of f16x1 (piece stride 0): the same results from the leading piece alone
    lead = x_cb[:x.numel()].clone()
    z1, _, _ = ops._conv_fprop(desc, x, wf, None, True, mpf, lead, x_bound, pk.w_bound)
    dw2 = ops._conv_wgrad(desc, x, gy, x_cb, gy_cb, x_bound, gy_bound)
    dw1 = ops._conv_wgrad(desc, x, gy, lead, gy_cb[:gy.numel()].clone(), x_bound, gy_bound)
    assert torch.equal(z1, z) and torch.equal(dw1, dw2)


def _bn_case(n, c, h, w, seed, dev):
    g = torch.Generator().manual_seed(seed)
    z = torch.randn(n, c, h, w, generator=g) * (0.5 + torch.rand(1, c, 1, 1, generator=g) * 3) + torch.randn(1, c, 1, 1, generator=g)
    gamma = torch.rand(c, generator=g) + 0.5
    beta = torch.randn(c, generator=g) * 0.3
    res = torch.randn(n, c, h, w, generator=g).clamp_min(0)
    dy = torch.randn(n, c, h, w, generator=g) * 1e-6  # (the magnitude of a mean-reduced loss's gradients: far below fp16's normal range)
    return [t.to(dev) for t in (z, gamma, beta, res, dy)]


@pytest.mark.parametrize("shape,res,relu", [((3, 64, 12, 20), False, True), ((2, 256, 11, 13), True, True), ((2, 128, 30, 40), False, False),
                                            ((4, 1024, 30, 40), True, True), ((2, 64, 9, 7), True, False)])
def test_bn_half_kernels_against_fp64(shape, res, relu):
    dev = _dev()
    from mcdseg._lib import check, lib
    from mcdseg import ops
    import ctypes
    L = lib()
    n, c, h, w = shape
    hw = h * w
    z, gamma, beta, rr, dy = _bn_case(n, c, h, w, 71, dev)
    p = ops._p
    st = ops._stream()
    # the 16-bit tensors as the chain would hold them
    z_bound = (z.abs().max() * 37.0).reshape(1)  # (a loose bound, as K x_bound w_bound is)
    zs = _scale(z_bound)
    z16 = _nchw_to_units((z / zs).to(torch.float16))
    zq = _units_to_nchw(z16, n, c, h, w).double() * zs       # what every kernel below reads as z
    res_bound = (rr.abs().max() * 1.5).reshape(1)
    rs = _scale(res_bound)
    r16 = _nchw_to_units((rr / rs).to(torch.float16))
    rq = _units_to_nchw(r16, n, c, h, w).double() * rs
    mean = zq.mean((0, 2, 3)).float()
    var = zq.var((0, 2, 3), unbiased=False)
    rstd = (1.0 / torch.sqrt(var + 1e-5)).float()
    a = gamma.double() * rstd.double()
    y_ref = zq * a.view(1, c, 1, 1) + (beta.double() - mean.double() * a).view(1, c, 1, 1) + (rq if res else 0)
    pre = y_ref
    if relu:
        y_ref = y_ref.clamp_min(0)
    y_bound = (y_ref.abs().max().float() * 1.7).reshape(1)
    ys = _scale(y_bound)
    y_cb = torch.empty(n * c * hw, dtype=torch.int16, device=dev)
    check(L.mcdseg_bn_apply_half(p(z16), p(z_bound), p(mean), p(rstd), p(gamma), p(beta), p(r16) if res else None, p(res_bound) if res else None,
                                 p(y_cb), p(y_bound), n, c, hw, int(relu), st), "bn_apply_half")
    y_got = _units_to_nchw(y_cb.view(torch.float16), n, c, h, w).double() * ys
    err = float((y_got - y_ref).abs().max())
    assert err <= 2.0 ** -11 * float(y_ref.abs().max()) + 2.0 ** -24 * ys + 1e-6 * float(y_ref.abs().max()), err
    # backward: bf16 gradient units in, fp64 reference of BatchNorm's backward on the same inputs
    dy16 = _nchw_to_units(dy.to(torch.bfloat16))
    dyq = _units_to_nchw(dy16, n, c, h, w).double()
    # the ReLU mask: the kernels take it from the stored activation (with residual) or recompute fma(z16, a zs, b) > 0 in fp32 (without);
    # either differs from the fp64 y > 0 only within an ulp of zero -- those pixels are left out of the element-wise comparison
    mask = (pre > 0) if relu else torch.ones_like(dyq, dtype=torch.bool)
    margin = (pre.abs() > 1e-5 * float(pre.abs().max())) if relu else mask
    gm = dyq * mask
    xhat = (zq - mean.double().view(1, c, 1, 1)) * rstd.double().view(1, c, 1, 1)
    dbeta_ref = gm.sum((0, 2, 3))
    dgamma_ref = (gm * xhat).sum((0, 2, 3))
    m = n * hw
    dz_ref = a.view(1, c, 1, 1) * (gm - dbeta_ref.view(1, c, 1, 1) / m - xhat * dgamma_ref.view(1, c, 1, 1) / m)
    kind = 0 if not relu else (4 if res else 2)
    ws = torch.empty(L.mcdseg_bn_bwd_half_workspace_bytes(n, c, hw) // 4 + 1, dtype=torch.float32, device=dev)
    dgamma, dbeta = torch.empty(c, device=dev), torch.empty(c, device=dev)
    dz_bound = torch.empty(1, device=dev)
    check(L.mcdseg_bn_bwd_reduce_half(p(dy16), p(y_cb) if kind == 4 else None, p(z16), p(z_bound), p(mean), p(rstd), p(gamma), p(beta), p(dgamma),
                                      p(dbeta), p(dz_bound), kind, 1, n, c, hw, p(ws), ctypes.c_size_t(ws.numel() * 4), st), "bn_bwd_reduce_half")
    assert float((dbeta.double() - dbeta_ref).abs().max()) <= 1e-5 * float(dyq.abs().sum((0, 2, 3)).max())
    assert float((dgamma.double() - dgamma_ref).abs().max()) <= 1e-5 * float((dyq * xhat).abs().sum((0, 2, 3)).max())
    assert float(dz_ref.abs().max()) <= float(dz_bound), "dz_bound bounds dz"
    dz_cb = torch.empty(n * c * hw, dtype=torch.int16, device=dev)
    dres = torch.empty(n, c, h, w, dtype=torch.bfloat16, device=dev) if res else None
    check(L.mcdseg_bn_bwd_apply_half(p(dy16), p(y_cb) if kind == 4 else None, p(z16), p(z_bound), p(mean), p(rstd), p(gamma), p(beta), p(dgamma),
                                     p(dbeta), p(dz_cb), p(dz_bound), p(dres), kind, 1, n, c, hw, st), "bn_bwd_apply_half")
    dzs = _scale(dz_bound)
    dz_got = _units_to_nchw(dz_cb.view(torch.float16), n, c, h, w).double() * dzs
    dz_scale = float(dz_ref.abs().max())
    bad = ((dz_got - dz_ref).abs() > 2e-3 * dz_scale + 2.0 ** -24 * dzs) & margin
    assert not bool(bad.any()), (int(bad.sum()), dz_scale, float((dz_got - dz_ref).abs().max()))
    if res:
        want = torch.where(y_got > 0, dyq, torch.zeros_like(dyq)) if relu else dyq
        assert torch.equal(_units_to_nchw(dres, n, c, h, w).double(), want), "the residual's gradient is dy under the stored activation's mask: exact"


def test_pack_unpack_bf16_units():
    dev = _dev()
    from mcdseg import ops
    g = torch.randn(3, 40, 7, 9, generator=torch.Generator().manual_seed(5)).to(dev) * 1e-7
    u = ops.pack_bf16_units(g)
    assert u.dtype == torch.bfloat16 and torch.equal(_units_to_nchw(u, 3, 40, 7, 9), g.to(torch.bfloat16))
    assert torch.equal(ops.unpack_bf16_units(u), g.to(torch.bfloat16).float())


def _bottleneck_run(dev, half, seed=3, n=2, h=24, w=32):
    """two Bottleneck blocks (the first projecting its shortcut) + a plain tail group that leaves the trunk in fp32, forward and backward"""
    from mcdseg import ops
    from models.drn import Bottleneck, BatchNorm2d, Conv2d, ConvBN, ConvBNReLU
    import torch.nn as nn
    torch.manual_seed(seed)
    ds = ConvBN(Conv2d(64, 256, kernel_size=1, stride=1, bias=False), BatchNorm2d(256))
    # (the head of the chain as in a DRN-D trunk, models/drn.py:195-205: thin layers -- 16 input channels, their window kernels multiply both
    # pieces of an operand whatever the arithmetic -- stay OUTSIDE the 2-byte chain; the 32 -> 64 convolution is its first group)
    net = nn.ModuleList([ConvBNReLU(Conv2d(16, 16, kernel_size=3, padding=1, bias=False), BatchNorm2d(16), nn.ReLU(inplace=True),
                                    Conv2d(16, 32, kernel_size=3, stride=2, padding=1, bias=False), BatchNorm2d(32), nn.ReLU(inplace=True),
                                    Conv2d(32, 64, kernel_size=3, padding=1, bias=False), BatchNorm2d(64), nn.ReLU(inplace=True)),
                         Bottleneck(64, 64, 1, ds, dilation=(1, 1)), Bottleneck(256, 64, dilation=(2, 2)),
                         ConvBNReLU(Conv2d(256, 128, kernel_size=3, padding=1, bias=False), BatchNorm2d(128), nn.ReLU(inplace=True))]).to(dev).train()
    for m in net.modules():
        if isinstance(m, nn.BatchNorm2d):
            m.weight.data.uniform_(0.5, 1.5)
            m.bias.data.normal_(0, 0.2)
    x = torch.randn(n, 16, 2 * h, 2 * w, generator=torch.Generator().manual_seed(seed + 1)).to(dev).requires_grad_()
    prev = (ops.CONV_MATH, ops.ACT_STORAGE, ops.HALF_STORAGE)
    ops.CONV_MATH, ops.ACT_STORAGE, ops.HALF_STORAGE = "f16x1", "compact", half
    names = []

    class _Names:
        def wants(self, name):
            names.append(name)
            return False
    prev_t, ops.LAUNCH_TIMER = ops.LAUNCH_TIMER, _Names()
    try:
        with ops.late_weight_grads(net):
            with ops.trunk_internal():
                t = net[0](x)
                t = net[1](t)
                t = net[2](t)
                inner = (t.dtype, ops.is_virtual(t))
            y = net[3](t)
        gy = torch.randn(y.shape, generator=torch.Generator().manual_seed(seed + 2)).to(dev) * 1e-5
        y.backward(gy)
        torch.cuda.synchronize()
    finally:
        ops.LAUNCH_TIMER = prev_t
        ops.CONV_MATH, ops.ACT_STORAGE, ops.HALF_STORAGE = prev
    grads = {k: p.grad.detach().clone() for k, p in net.named_parameters()}
    return y.detach(), x.grad.detach().clone(), grads, inner, set(names)


def test_bottleneck_blocks_through_the_half_chain():
    """the 2-byte chain against round 5's storage of the same arithmetic (two fp16 pieces stored, fp32 z and gradients): the differences
    are the extra roundings of z (11 bits), of the residual stream (11 instead of 22 bits) and of the gradients between the groups (bf16:
    8 bits) -- stated here as bounds on the block's output and on every gradient"""
    dev = _dev()
    y0, gx0, g0, inner0, names0 = _bottleneck_run(dev, False)
    y1, gx1, g1, inner1, names1 = _bottleneck_run(dev, True)
    assert inner0 == (torch.float32, True) and inner1 == (torch.bfloat16, True)
    assert {"bn_apply_half", "bn_bwd_reduce_half", "bn_bwd_apply_half"} <= names1 and not ({"bn_apply_half"} & names0)
    assert y1.dtype == torch.float32 and gx1.dtype == torch.float32 and gx1.shape == gx0.shape

    def rel(a, b):
        return float((a.double() - b.double()).norm() / b.double().norm())

    worst = max((rel(g1[k], g0[k]), k) for k in g0)
    cos = min(float(torch.dot(g1[k].flatten().double(), g0[k].flatten().double()) / (g1[k].double().norm() * g0[k].double().norm())) for k in g0)
    print("half chain vs two-piece f16x1 storage: output rel L2 %.2e, input gradient %.2e, worst parameter gradient %.2e (%s), smallest cosine %.5f"
          % (rel(y1, y0), rel(gx1, gx0), worst[0], worst[1], cos))
    assert rel(y1, y0) <= 5e-3, rel(y1, y0)        # measured 1.6e-3
    assert rel(gx1, gx0) <= 0.12, rel(gx1, gx0)    # measured 5.5e-2: eight roundings to bf16 between the groups, and the ReLU masks of a z kept to 11 bits
    assert worst[0] <= 0.15, worst
    assert cos >= 0.99, cos


@pytest.mark.parametrize("case", [(256, 256, 3, 1, 2, 60, 80, 16), (1024, 256, 1, 1, 1, 90, 160, 4), (256, 1024, 1, 1, 1, 90, 160, 4), (48, 256, 1, 1, 1, 128, 129, 4)],
                         ids=lambda c: "x".join(map(str, c)))
def test_two_k_steps_per_interval_are_bitwise_the_one_step_kernel(case, monkeypatch, libopt):
    """SplitF16x1D (csrc/split.h: the one-term ping-pong convolution with two K-steps of 16 channels per barrier interval) against
    SplitF16x1's kernel (option PP_DEEP = 0): the same products in the same order -- forward output, BatchNorm partial rows, data gradient
    and the 16-bit epilogues bit for bit; an odd number of K-steps (48 channels, 1 x 1) stays on the one-step kernel."""
    dev = _dev()
    import ctypes
    from mcdseg import ops
    monkeypatch.setattr(ops, "CONV_MATH", "f16x1")
    cin, cout, k, s, d, h, w, n = case
    g = torch.Generator().manual_seed(67)
    x = torch.randn(n, cin, h, w, generator=g).to(dev)
    wt = (torch.randn(cout, cin, k, k, generator=g) * (2.0 / (k * k * cout)) ** 0.5).to(dev)
    desc = ops.conv_desc(x.shape, wt.shape, s, d * (k // 2), d)
    gy = torch.randn(n, cout, desc.Ho, desc.Wo, generator=g).to(dev)
    pk = ops.PackedWeights()
    wf, wd, mpf = pk.get(wt, desc)
    x_cb, x_bound = ops.split_companion(x)
    gy_cb, gy_bound = ops.split_companion(gy)
    L = ops.lib()
    even = (k * k * ((cin + 15) // 16)) % 2 == 0
    outs = {}
    for deep in (1, 0):
        libopt(PP_DEEP=deep)
        assert L.mcdseg_conv_split_pp_deep(ctypes.byref(desc), 1, 0) == (1 if (deep and even) else 0)
        names = []

        class _Names:
            def wants(self, name):
                names.append(name)
                return False
        prev, ops.LAUNCH_TIMER = ops.LAUNCH_TIMER, _Names()
        try:
            z, part, rows = ops._conv_fprop(desc, x, wf, None, True, mpf, x_cb, x_bound, pk.w_bound)
            dx = ops._conv_dgrad(desc, None, wd, gy_cb, gy_bound, pk.w_bound)
            z16, zb, part16, _ = ops._conv_fprop_half(desc, x_cb, x_bound, wf, pk.w_bound, mpf)
            dx16 = ops._conv_dgrad_half(desc, gy_cb, gy_bound, wd, pk.w_bound)
        finally:
            ops.LAUNCH_TIMER = prev
        outs[deep] = (z, part, dx, z16, part16, dx16, names)
    assert any("SplitF16x1D" in nm for nm in outs[1][6]) == even and not any("SplitF16x1D" in nm for nm in outs[0][6]), (outs[1][6], outs[0][6])
    assert any("conv_gemm_split_pp_kernel" in nm for nm in outs[1][6]), outs[1][6]
    for a, b, what in zip(outs[1][:6], outs[0][:6], ("forward", "partial rows", "data gradient", "z16", "partial rows (16-bit epilogue)", "dx16")):
        assert torch.equal(a, b), what


def test_mfnet_and_multitask_models_run_in_the_two_byte_chain():
    """the other two model families of the path (MFNet score fusion: two 3-channel encoders; multitask: one encoder, three decoders) in
    ``--dtype f16``: one forward + backward each at 4 x 96 x 128, default arithmetic / f16x1 with round 5's two-piece storage / f16x1 in
    the 2-byte chain (``tools/probes/half_other_models.py``).  The chain costs what the one-term arithmetic costs and little more:
    its distance from the default's gradients is within 1.25 x the two-piece storage's [measured 1.07 x and 1.04 x: 0.261 against 0.245,
    0.352 against 0.338], the outputs within 3e-2 [1.6e-2]"""
    import os
    import re
    import subprocess
    import sys
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "probes", "half_other_models.py")], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, MCDSEG_PRETRAINED="0"))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    rows = re.findall(r"^(\S+), f16x1 (with two-piece storage|in the 2-byte chain).*overall rel L2 (\S+),.*output rel (\S+)$", r.stdout, re.M)
    assert len(rows) == 4, r.stdout[-3000:]
    got = {(m, "chain" if "chain" in k else "two"): (float(g), float(o)) for m, k, g, o in rows}
    for model in ("MFNet-ScoreAddFusion", "multitask"):
        assert got[(model, "chain")][0] <= 1.25 * got[(model, "two")][0], got
        assert got[(model, "chain")][1] <= 3e-2, got


@pytest.mark.parametrize("case", [(256, 256, 3, 1, 2, 60, 80, 16), (1024, 256, 1, 1, 1, 90, 160, 4), (512, 512, 3, 1, 4, 24, 32, 6), (392, 504, 3, 1, 4, 29, 37, 8),
                                  (256, 400, 3, 2, 1, 63, 81, 6), (256, 256, 1, 1, 1, 33, 1000, 4)], ids=lambda c: "x".join(map(str, c)))
def test_six_stage_weight_gradient_is_bitwise_the_three_stage_kernel(case, monkeypatch, libopt):
    """the one-term ping-pong weight gradient with the idle piece-1 LDS slots used as three more stages (``conv_wgrad_split_pp_kernel<SplitF16x1D>``:
    the DMAs five K-steps ahead instead of two) against the three-stage kernel (option WGRAD_PP_DEEP = 0): the same stages in the same
    order, the same sums bit for bit -- slabs of an odd number of K-steps (the second case: 57; the last: 33) end on a single-step interval;
    and both are the fp64 weight gradient within the one-term arithmetic's error"""
    dev = _dev()
    import ctypes
    from mcdseg import ops
    monkeypatch.setattr(ops, "CONV_MATH", "f16x1")
    cin, cout, k, s, d, h, w, n = case
    g = torch.Generator().manual_seed(91)
    x = torch.randn(n, cin, h, w, generator=g).to(dev)
    desc = ops.conv_desc(x.shape, (cout, cin, k, k), s, d * (k // 2), d)
    gy = torch.randn(n, cout, desc.Ho, desc.Wo, generator=g).to(dev)
    x_cb, x_bound = ops.split_companion(x)
    gy_cb, gy_bound = ops.split_companion(gy)
    if ops.lib().mcdseg_conv_wgrad_variant(ctypes.byref(desc), ops.MATH_ID["f16x1"], 1) != 17:
        pytest.fail("not a ping-pong weight-gradient geometry")
    got = {}
    for deep in (1, 0):
        libopt(WGRAD_PP_DEEP=deep)
        assert ("SplitF16x1D" in ops.wgrad_split_kernel_name(desc, True)) == bool(deep)
        got[deep] = ops._conv_wgrad(desc, x, gy, x_cb, gy_cb, x_bound, gy_bound)
        again = ops._conv_wgrad(desc, x, gy, x_cb, gy_cb, x_bound, gy_bound)
        assert torch.equal(got[deep].view(torch.int32), again.view(torch.int32))
    assert torch.equal(got[1].view(torch.int32), got[0].view(torch.int32))
    x64 = x.double().cpu()
    w64 = torch.zeros(cout, cin, k, k, dtype=torch.float64, requires_grad=True)
    ref = torch.autograd.grad(torch.nn.functional.conv2d(x64, w64, None, stride=s, padding=d * (k // 2), dilation=d), w64, gy.double().cpu())[0]
    err = float((got[1].double().cpu() - ref).abs().max()) / float(ref.abs().max())
    assert err <= 2e-3, err  # (operands rounded to 11 bits)
