"""GPU: every HIP kernel (through the C ABI) against a plain fp64 PyTorch-CPU statement of the same op,
and the loss kernel against the oracle's closed forms and the golden vectors."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def _maxerr(got, ref):
    got = got.detach().double().cpu()
    ref = ref.detach().double().cpu()
    return float((got - ref).abs().max()), float(ref.abs().max()) + 1e-30


def _assert_close(got, ref, rtol, what):
    err, scale = _maxerr(got, ref)
    assert err <= rtol * scale, "%s: max err %.3e vs scale %.3e (rel %.2e > %.1e)" % (what, err, scale, err / scale, rtol)


# (Cin, Cout, k, stride, dil, H, W, N, bias)  -- every conv geometry of drn_d_38 / drn_d_105 / MFNet stems
CONV_CASES = [
    (6, 16, 7, 1, 1, 20, 28, 2, False),     # 6-ch stem (K padded 6 -> 8)
    (3, 16, 7, 1, 1, 18, 22, 1, False),     # MFNet stems
    (1, 16, 7, 1, 1, 9, 13, 2, False),
    (16, 16, 3, 1, 1, 20, 28, 2, False),    # layer1
    (16, 32, 3, 2, 1, 21, 27, 2, False),    # layer2, stride 2, odd size
    (32, 64, 3, 2, 1, 20, 28, 2, False),    # layer3 first block
    (32, 64, 1, 2, 1, 20, 28, 2, False),    # 1x1 stride-2 shortcut
    (64, 64, 3, 1, 1, 12, 20, 3, False),
    (64, 128, 3, 2, 1, 12, 20, 2, False),
    (128, 256, 3, 1, 2, 12, 15, 2, False),  # dilation 2
    (256, 256, 3, 1, 2, 9, 12, 2, False),
    (128, 256, 1, 1, 1, 9, 12, 2, False),   # 1x1 projection
    (256, 512, 3, 1, 4, 10, 12, 2, False),  # dilation 4
    (512, 512, 3, 1, 4, 8, 12, 2, False),
    (512, 512, 3, 1, 1, 8, 12, 1, False),
    (512, 41, 1, 1, 1, 8, 12, 2, True),     # seg head with bias (M padded 41 -> 64)
    (64, 256, 1, 1, 1, 7, 9, 2, False),     # bottleneck 1x1s
    (256, 64, 1, 1, 1, 7, 9, 2, False),
    (40, 24, 3, 1, 1, 11, 13, 2, True),     # ragged channel counts
]


def _conv_inputs(case, seed):
    cin, cout, k, s, d, h, w, n, bias = case
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, k, k, generator=g) * (2.0 / (k * k * cout)) ** 0.5
    b = torch.randn(cout, generator=g) * 0.1 if bias else None
    pad = d * (k // 2)
    return x, wt, b, s, pad, d


@pytest.mark.parametrize("case", CONV_CASES, ids=lambda c: "x".join(map(str, c[:5])))
def test_conv_fprop_dgrad_wgrad(case):
    dev = _dev()
    from mcdseg import ops
    x, wt, b, s, pad, d = _conv_inputs(case, 7)
    desc = ops.conv_desc(x.shape, wt.shape, s, pad, d)
    packed = ops.PackedWeights()
    xg, wg = x.to(dev), wt.to(dev)
    wf, wd, mpf = packed.get(wg, desc)
    y, _, _ = ops._conv_fprop(desc, xg, wf, b.to(dev) if b is not None else None, False, mpf, w_bound=packed.w_bound)
    x64 = x.double().requires_grad_()
    w64 = wt.double().requires_grad_()
    ref = F.conv2d(x64, w64, b.double() if b is not None else None, stride=s, padding=pad, dilation=d)
    _assert_close(y, ref, 2e-5, "fprop")
    gy = torch.randn(ref.shape, generator=torch.Generator().manual_seed(8))
    gx_ref, gw_ref = torch.autograd.grad(ref, [x64, w64], gy.double())
    dx = ops._conv_dgrad(desc, gy.to(dev), wd, w_bound=packed.w_bound)
    _assert_close(dx, gx_ref, 2e-5, "dgrad")
    dw = ops._conv_wgrad(desc, xg, gy.to(dev))
    _assert_close(dw, gw_ref, 2e-5, "wgrad")


def test_conv_rejects_bad_geometry():
    dev = _dev()
    from mcdseg import ops
    from mcdseg._lib import ConvDesc, lib
    import ctypes
    x = torch.zeros(1, 4, 8, 8, device=dev)
    bad = ConvDesc(1, 4, 8, 8, 8, 3, 3, 1, 1, 1, 9, 9)  # wrong Ho/Wo
    y = torch.zeros(1, 8, 9, 9, device=dev)
    rc = lib().mcdseg_conv_fprop(ctypes.byref(bad), ops._p(x), ops._p(x), None, ops._p(y), None, ops._stream())
    assert rc != 0 and b"Ho/Wo" in lib().mcdseg_last_error()
    with pytest.raises(RuntimeError, match="GPU"):
        ops._req(torch.zeros(2), "cpu tensor")


def test_new_entry_points_reject_what_they_cannot_do(monkeypatch):
    """error behaviour of this round's entry points: the LDS-window forward has no bias epilogue (rejected, not ignored), the
    padded companion needs its bound scalar, the z-mask BatchNorm backward its beta, a weight gradient on split operands both
    bound scalars -- each with a reason in mcdseg_last_error()"""
    dev = _dev()
    import ctypes
    from mcdseg import ops
    L = ops.lib()
    monkeypatch.setattr(ops, "CONV_MATH", "f16x3")
    x = torch.randn(1, 16, 8, 32, device=dev)
    wt = torch.randn(16, 16, 3, 3, device=dev) * 0.1
    desc = ops.conv_desc(x.shape, wt.shape, 1, 1, 1)
    pk = ops.PackedWeights()
    wf, wd, mpf = pk.get(wt, desc)
    x_cb, x_bound = ops.split_companion(x)
    y = torch.empty(1, 16, 8, 32, device=dev)
    bias = torch.zeros(16, device=dev)
    rc = L.mcdseg_conv_split_fprop(ctypes.byref(desc), 3, ops._p(x), ops._p(x_cb), ops._p(x_bound), ops._p(wf), ops._p(pk.w_bound),
                                   ops._p(bias), ops._p(y), None, ops._stream())
    assert rc != 0 and b"no bias" in L.mcdseg_last_error()
    x6 = torch.randn(1, 6, 8, 8, device=dev)
    cb = torch.empty(2 * 8 * 64, dtype=torch.int16, device=dev)
    rc = L.mcdseg_split_cb_padded(ops._p(x6), ops._p(cb), None, 3, 1, 6, 64, ops._stream())
    assert rc != 0 and b"bound" in L.mcdseg_last_error()
    z = torch.randn(1, 16, 8, 32, device=dev)
    v = torch.ones(16, device=dev)
    ws = torch.empty(L.mcdseg_bn_bwd_workspace_bytes(1, 16, 256) // 4 + 1, device=dev)
    rc = L.mcdseg_bn_bwd_reduce_zmask(ops._p(z), ops._p(z), ops._p(v), ops._p(v), ops._p(v), None, ops._p(v.clone()), ops._p(v.clone()), None,
                                      1, 1, 16, 256, ops._p(ws), ctypes.c_size_t(ws.numel() * 4), ops._stream())
    assert rc != 0 and b"null pointer" in L.mcdseg_last_error()
    dw = torch.empty_like(wt)
    wsz = torch.empty(L.mcdseg_conv_wgrad_workspace_bytes(ctypes.byref(desc)) // 4 + 1, device=dev)
    gy_cb, gy_bound = ops.split_companion(z)
    rc = L.mcdseg_conv_split_wgrad(ctypes.byref(desc), 3, None, ops._p(x_cb), None, None, ops._p(gy_cb), ops._p(gy_bound), ops._p(dw), ops._p(wsz),
                                   ctypes.c_size_t(wsz.numel() * 4), ops._stream())
    assert rc != 0  # companions without the operand's bound scalar, and no fp32 operand to fall back to


@pytest.mark.parametrize("cfg", [
    # Cin, Cout, k, stride, dil, H, W, N, relu, residual, train
    (16, 16, 3, 1, 1, 20, 28, 2, True, False, True),
    (16, 32, 3, 2, 1, 21, 27, 2, True, False, True),
    (64, 64, 3, 1, 1, 12, 20, 3, True, True, True),
    (128, 256, 1, 1, 1, 9, 12, 2, False, False, True),
    (256, 256, 3, 1, 2, 9, 12, 2, True, True, True),
    (512, 512, 3, 1, 4, 8, 12, 2, True, True, True),
    (64, 64, 3, 1, 1, 12, 20, 2, True, True, False),   # eval-mode BN (fix_bn / inference)
    (40, 24, 3, 1, 1, 11, 13, 2, False, True, True),   # HW % 4 != 0 -> scalar paths
    (6, 16, 7, 1, 1, 20, 28, 2, True, False, True),    # stem: direct (LDS-tiled) convolution, one ragged tile per image
    (6, 16, 7, 1, 1, 19, 141, 2, True, False, True),   # stem: 3 x 3 tiles, ragged in both directions
    (3, 16, 7, 1, 1, 17, 70, 1, True, False, True),    # MFNet RGB stem
    (1, 16, 7, 1, 1, 9, 13, 2, True, False, True),     # MFNet single-channel stem
    (6, 16, 7, 1, 1, 20, 28, 2, True, False, False),   # stem, eval-mode BN
], ids=lambda c: "x".join(map(str, c)))
def test_conv_bn_act_fwd_bwd(cfg):
    dev = _dev()
    from mcdseg import ops
    from models.drn import BatchNorm2d, Conv2d
    cin, cout, k, s, d, h, w, n, relu, use_res, train = cfg
    g = torch.Generator().manual_seed(11)
    pad = d * (k // 2)
    conv = Conv2d(cin, cout, k, stride=s, padding=pad, dilation=d, bias=False)
    bn = BatchNorm2d(cout)
    with torch.no_grad():
        conv.weight.copy_(torch.randn(conv.weight.shape, generator=g) * (2.0 / (k * k * cout)) ** 0.5)
        bn.weight.copy_(1 + 0.2 * torch.randn(cout, generator=g))
        bn.bias.copy_(0.1 * torch.randn(cout, generator=g))
        bn.running_mean.copy_(0.05 * torch.randn(cout, generator=g))
        bn.running_var.copy_(1 + 0.1 * torch.rand(cout, generator=g))
    x = torch.randn(n, cin, h, w, generator=g) + 0.3
    ho, wo = (h + 2 * pad - d * (k - 1) - 1) // s + 1, (w + 2 * pad - d * (k - 1) - 1) // s + 1
    res = torch.randn(n, cout, ho, wo, generator=g) if use_res else None
    gy = torch.randn(n, cout, ho, wo, generator=g)

    # fp64 reference on CPU
    x64 = x.double().requires_grad_()
    w64 = conv.weight.detach().double().requires_grad_()
    ga64 = bn.weight.detach().double().requires_grad_()
    be64 = bn.bias.detach().double().requires_grad_()
    r64 = res.double().requires_grad_() if use_res else None
    rm, rv = bn.running_mean.double().clone(), bn.running_var.double().clone()
    z = F.conv2d(x64, w64, None, s, pad, d)
    o = F.batch_norm(z, rm, rv, ga64, be64, training=train, momentum=0.1, eps=1e-5)
    if use_res:
        o = o + r64
    if relu:
        o = F.relu(o)
    ins = [x64, w64, ga64, be64] + ([r64] if use_res else [])
    grads = torch.autograd.grad(o, ins, gy.double())

    conv.to(dev), bn.to(dev)
    bn.train(train)
    xg = x.to(dev).requires_grad_()
    rg = res.to(dev).requires_grad_() if use_res else None
    y = ops.conv_bn_act(xg, conv, bn, relu=relu, residual=rg)
    _assert_close(y, o, 3e-5, "forward")
    if train:
        _assert_close(bn.running_mean, rm, 1e-5, "running_mean")
        _assert_close(bn.running_var, rv, 1e-5, "running_var")
        assert int(bn.num_batches_tracked) == 1
    else:
        assert int(bn.num_batches_tracked) == 0
    y.backward(gy.to(dev))
    got = [xg.grad, conv.weight.grad, bn.weight.grad, bn.bias.grad] + ([rg.grad] if use_res else [])
    for name, a, b in zip(["dx", "dw", "dgamma", "dbeta", "dres"], got, grads):
        _assert_close(a, b, 1e-4, name)


@pytest.mark.parametrize("shape", [(2, 41, 8, 12), (1, 5, 3, 7), (3, 16, 6, 10)])
def test_up8_fwd_bwd(shape):
    dev = _dev()
    from mcdseg import ops
    n, c, hi, wi = shape
    g = torch.Generator().manual_seed(5)
    x = torch.randn(n, c, hi, wi, generator=g)
    w = torch.randn(c, 1, 16, 16, generator=g) * 0.1
    x2 = torch.randn(n, c, hi, wi, generator=g)
    w2 = torch.randn(c, 1, 16, 16, generator=g) * 0.1
    gy = torch.randn(n, c, 8 * hi, 8 * wi, generator=g)
    t = [v.double().requires_grad_() for v in (x, w, x2, w2)]
    ref = F.conv_transpose2d(t[0], t[1], stride=8, padding=4, groups=c)
    ref2 = ref + F.conv_transpose2d(t[2], t[3], stride=8, padding=4, groups=c)
    gr = torch.autograd.grad(ref, t[:2], gy.double(), retain_graph=True)
    gr2 = torch.autograd.grad(ref2, t, gy.double())
    d = [v.to(dev).requires_grad_() for v in (x, w, x2, w2)]
    y = ops.up8(d[0], d[1])
    _assert_close(y, ref, 1e-5, "up8 fwd")
    y.backward(gy.to(dev))
    _assert_close(d[0].grad, gr[0], 2e-5, "up8 dx")
    _assert_close(d[1].grad, gr[1], 2e-5, "up8 dw")
    for v in d:
        v.grad = None
    y2 = ops.up8_dual(*d)
    _assert_close(y2, ref2, 1e-5, "up8_dual fwd")
    y2.backward(gy.to(dev))
    for name, a, b in zip(("dx1", "dw1", "dx2", "dw2"), [v.grad for v in d], gr2):
        _assert_close(a, b, 2e-5, "up8_dual " + name)


@pytest.mark.parametrize("shape", [(2, 41, 23, 80), (1, 3, 60, 37), (2, 4, 1, 1), (1, 2, 21, 160), (1, 1, 4, 200)])
def test_up8_backward_band_kernel(shape):
    """The single-pass backward of the up-sampler (mcdseg_up8_bwd: dy staged through an LDS ring, both gradients or either one)
    against fp64 and against the two separate kernels: several row bands per plane (Hi >= 20), ragged last band, a row that is
    not a whole number of DMA units, Wi = 1 (both window edges in one column), Wi = 200 (falls back to the two kernels)."""
    dev = _dev()
    from mcdseg import ops
    n, c, hi, wi = shape
    g = torch.Generator().manual_seed(6)
    x = torch.randn(n, c, hi, wi, generator=g)
    w = torch.randn(c, 1, 16, 16, generator=g) * 0.1
    gy = torch.randn(n, c, 8 * hi, 8 * wi, generator=g)
    x64, w64 = x.double().requires_grad_(), w.double().requires_grad_()
    ref = F.conv_transpose2d(x64, w64, stride=8, padding=4, groups=c)
    rx, rw = torch.autograd.grad(ref, [x64, w64], gy.double())
    xg, wg, gg = x.to(dev), w.to(dev), gy.to(dev)
    dx, dw = ops._up8_bwd(gg, wg, xg, True, True)
    _assert_close(dx, rx, 2e-5, "band kernel dx")
    _assert_close(dw, rw, 2e-5, "band kernel dw")
    dx1, none = ops._up8_bwd(gg, wg, xg, True, False)
    none2, dw1 = ops._up8_bwd(gg, wg, xg, False, True)
    assert none is None and none2 is None
    assert torch.equal(dx1, dx) and torch.equal(dw1, dw), "the one-gradient forms differ from the combined launch"
    _assert_close(ops._up8_bwd_input(gg, wg, n, c, hi, wi), rx, 2e-5, "separate dx kernel")
    _assert_close(ops._up8_bwd_weight(gg, xg, n, c, hi, wi), rw, 2e-5, "separate dw kernel")
    assert ops._up8_bwd(gg, wg, xg, False, False) == (None, None)


@pytest.mark.parametrize("shape", [(2, 41, 5, 7), (1, 12, 3, 20), (2, 20, 4, 33), (1, 41, 2, 80), (5, 41, 30, 24), (3, 30, 33, 17)])
@pytest.mark.parametrize("mode", ["ce+diff", "diff", "ce-single", "shared-scores"])
@pytest.mark.parametrize("dma", ["1", "0"])
def test_up8_loss_fused_equals_two_pass(shape, mode, dma, monkeypatch, libopt):
    """The loss kernel that forms the up-sampled logits on the fly (mcdseg_up8_softmax_ce_l1) against up8 followed by the
    plain loss kernel: identical logit gradients (bitwise), loss values up to the order of the block partial sums; ragged
    row segments (8 Wi not a multiple of 128), the image border and ignore_index pixels included.  Both forms of the kernel:
    inputs prefetched by LDS-DMA behind a counted wait (the default; the last two shapes give every workgroup several
    patches, i.e. gradient stores still in flight when the next patch's inputs are waited for) and staged through registers."""
    dev = _dev()
    from mcdseg import ops
    libopt(UP8_LOSS_DMA=int(dma))
    assert ("_dma_" in ops.up8_loss_kernel_name(*shape, mode != "ce-single", mode != "diff")) == (dma == "1")
    n, c, hi, wi = shape
    g = torch.Generator().manual_seed(11)
    s1 = (2 * torch.randn(n, c, hi, wi, generator=g)).to(dev)
    s2 = s1 if mode == "shared-scores" else (2 * torch.randn(n, c, hi, wi, generator=g)).to(dev)
    w1 = (torch.randn(c, 1, 16, 16, generator=g) * 0.2).to(dev)
    w2 = (torch.randn(c, 1, 16, 16, generator=g) * 0.2).to(dev)
    lab = torch.randint(0, c, (n, 8 * hi, 8 * wi), generator=g)
    lab[0, 0, :5] = -100
    lab = lab.to(dev)
    cw = (0.5 + torch.rand(c, generator=g)).to(dev)
    single = mode == "ce-single"
    kw = dict(ce_coef=0.0 if mode == "diff" else 1.0, diff_coef=0.0 if single else 0.7)
    labels = None if mode == "diff" else lab
    z1, z2 = ops.up8(s1, w1), (None if single else ops.up8(s2, w2))
    ref_l, ref_g1, ref_g2 = ops.mcd_losses(z1, z2, labels, cw if labels is not None else None, **kw)
    got_l, got_g1, got_g2 = ops.up8_mcd_losses(s1, w1, None if single else s2, None if single else w2, labels,
                                               cw if labels is not None else None, **kw)
    assert torch.equal(got_g1, ref_g1)
    if not single:
        assert torch.equal(got_g2, ref_g2)
    assert float((got_l - ref_l).abs().max()) <= 2e-6 * float(ref_l.abs().max())
    vals, _, _ = ops.up8_mcd_losses(s1, w1, None if single else s2, None if single else w2, labels,
                                    cw if labels is not None else None, want_grad=False, **kw)
    assert torch.equal(vals, got_l)
    if dma == "1":  # the two forms against each other: same arithmetic, same patch order, same sums -- the loss values bit for bit too
        libopt(UP8_LOSS_DMA=0)
        reg_l, reg_g1, reg_g2 = ops.up8_mcd_losses(s1, w1, None if single else s2, None if single else w2, labels,
                                                   cw if labels is not None else None, **kw)
        assert torch.equal(reg_l, got_l) and torch.equal(reg_g1, got_g1) and (single or torch.equal(reg_g2, got_g2))


def test_up8_loss_and_backward_at_the_benchmark_size(monkeypatch, libopt):
    """BASELINE config 2's own loss problem (16 x 41 x 60 x 80 scores -> 480 x 640 logits, 38 patches per workgroup): the LDS-DMA form of
    the fused kernel against the register-staged form, bit for bit in both gradients and in the loss values; size-independent properties
    of the result (a cross-entropy gradient sums to zero over the classes of a pixel, ignored pixels have none; the discrepancy's two
    gradients are orthogonal to the constant vector); and the up-sampler's one-pass backward of that gradient against the two separate
    kernels."""
    dev = _dev()
    from mcdseg import ops
    n, c, hi, wi = 16, 41, 60, 80
    g = torch.Generator().manual_seed(23)
    s1 = (2 * torch.randn(n, c, hi, wi, generator=g)).to(dev)
    s2 = (2 * torch.randn(n, c, hi, wi, generator=g)).to(dev)
    w1 = (torch.randn(c, 1, 16, 16, generator=g) * 0.2).to(dev)
    w2 = (torch.randn(c, 1, 16, 16, generator=g) * 0.2).to(dev)
    lab = torch.randint(0, c, (n, 8 * hi, 8 * wi), generator=g)
    lab[:, :7, :] = -100
    lab = lab.to(dev)
    cw = (0.5 + torch.rand(c, generator=g)).to(dev)
    out = {}
    for dma in ("1", "0"):
        libopt(UP8_LOSS_DMA=int(dma))
        out[dma] = (ops.up8_mcd_losses(s1, w1, s2, w2, lab, cw, ce_coef=1.0, diff_coef=0.0),
                    ops.up8_mcd_losses(s1, w1, s2, w2, None, None, ce_coef=0.0, diff_coef=-1.0))
    for a, b in zip(out["1"], out["0"]):
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
    (l_ce, g1, g2), (l_d, d1, d2) = out["1"]
    scale = float(g1.abs().max())
    assert float(g1.sum(dim=1).abs().max()) <= 2e-5 * scale and float(g2.sum(dim=1).abs().max()) <= 2e-5 * scale
    assert float(g1[:, :, :7, :].abs().max()) == 0.0  # ignore_index rows
    assert float(d1.sum(dim=1).abs().max()) <= 2e-5 * float(d1.abs().max()) and float(d2.sum(dim=1).abs().max()) <= 2e-5 * float(d2.abs().max())
    assert float(l_ce[0]) > 0 and float(l_d[2]) > 0
    del out, g2, d1, d2
    dx, dw = ops._up8_bwd(g1, w1, s1, True, True)
    _assert_close(dx, ops._up8_bwd_input(g1, w1, n, c, hi, wi), 2e-5, "band kernel dx at full size")
    _assert_close(dw, ops._up8_bwd_weight(g1, s1, n, c, hi, wi), 2e-5, "band kernel dw at full size")
    dx1, _ = ops._up8_bwd(g1, w1, s1, True, False)
    _, dw1 = ops._up8_bwd(g1, w1, s1, False, True)
    assert torch.equal(dx1, dx) and torch.equal(dw1, dw)


def test_loss_kernel_against_golden_and_closed_forms(golden):
    dev = _dev()
    from mcdseg import ops
    from oracle import ref_loss
    fx = golden.npz("loss_small.npz")
    z1 = torch.from_numpy(fx["z1"]).to(dev)
    z2 = torch.from_numpy(fx["z2"]).to(dev)
    y = torch.from_numpy(fx["y"]).to(dev)
    w = torch.from_numpy(fx["w"]).to(dev)
    # fused: CE on both heads + discrepancy, coefficients as in step B (CE - Diff)
    losses, g1, g2 = ops.mcd_losses(z1, z2, y, w, ce_coef=1.0, diff_coef=-1.0)
    ls = losses.cpu().numpy()
    ce2, gce2 = ref_loss.ce_and_grad(fx["z2"], fx["y"], fx["w"])
    assert abs(ls[0] - float(fx["ce"])) <= 1e-6 * float(fx["ce"])
    assert abs(ls[1] - ce2) <= 2e-6 * ce2
    assert abs(ls[2] - float(fx["diff"])) <= 2e-6 * float(fx["diff"])
    assert abs(ls[3] - float(fx["w"][fx["y"]].sum())) < 1e-3
    ref1 = fx["g_ce"].astype(np.float64) - fx["g_d1"]
    ref2 = gce2 - fx["g_d2"]
    assert np.abs(g1.cpu().numpy() - ref1).max() <= 1e-6 * np.abs(ref1).max()
    assert np.abs(g2.cpu().numpy() - ref2).max() <= 1e-6 * np.abs(ref2).max()
    # autograd-facing single-purpose paths
    a = z1.clone().requires_grad_()
    ce = ops.cross_entropy2d(a, y, torch.from_numpy(fx["wfull"]).to(dev))
    (3.0 * ce).backward()
    l64, g64 = ref_loss.ce_and_grad(fx["z1"], fx["y"], fx["wfull"])
    assert abs(float(ce) - float(fx["ce_w"])) <= 1e-6 * float(fx["ce_w"])
    assert np.abs(a.grad.cpu().numpy() - 3.0 * g64).max() <= 1e-6 * np.abs(3.0 * g64).max()
    a = z1.clone().requires_grad_()
    b = z2.clone().requires_grad_()
    dd = ops.diff2d(a, b)
    (-dd).backward()
    assert abs(float(dd) - float(fx["diff"])) <= 2e-6 * float(fx["diff"])
    # g_d1 / g_d2 are the reference's own fp32 autograd results: agree to a few fp32 ulps of the largest entry
    assert np.abs(a.grad.cpu().numpy() + fx["g_d1"]).max() <= 5e-6 * np.abs(fx["g_d1"]).max()
    assert np.abs(b.grad.cpu().numpy() + fx["g_d2"]).max() <= 5e-6 * np.abs(fx["g_d2"]).max()


@pytest.mark.parametrize("c,shape", [(41, (2, 9, 13)), (20, (1, 16, 16)), (14, (3, 5, 7)), (3, (2, 4, 4))])
def test_loss_kernel_edge_cases(c, shape):
    """ignore_index pixels, all-background batches (W = 0 in the reference gives nan), sum reduction."""
    dev = _dev()
    from mcdseg import ops
    from oracle import ref_loss
    n, h, w = shape
    rs = np.random.RandomState(c)
    z = (2 * rs.standard_normal((n, c, h, w))).astype(np.float32)
    y = rs.randint(0, c, size=(n, h, w)).astype(np.int64)
    y[0, 0, :2] = -100
    cw = rs.uniform(0.5, 1.5, c).astype(np.float32)
    l64, g64 = ref_loss.ce_and_grad(z, y, cw)
    zt = torch.from_numpy(z).to(dev).requires_grad_()
    loss = ops.cross_entropy2d(zt, torch.from_numpy(y).to(dev), torch.from_numpy(cw).to(dev))
    loss.backward()
    assert abs(float(loss) - l64) <= 2e-6 * abs(l64)
    assert np.abs(zt.grad.cpu().numpy() - g64).max() <= 2e-6 * np.abs(g64).max()
    assert float(zt.grad[0, :, 0, :2].abs().max()) == 0.0
    zt.grad = None
    # all-background batch: W = sum w[y] = 0, the reference's weighted mean is 0/0 = nan with nan gradients (nn.NLLLoss2d)
    cw0 = cw.copy()
    cw0[1] = 0.0
    y0 = np.full_like(y, 1)
    y0[0, 0, :2] = -100
    z0 = torch.from_numpy(z).to(dev).requires_grad_()
    l0 = ops.cross_entropy2d(z0, torch.from_numpy(y0).to(dev), torch.from_numpy(cw0).to(dev))
    l0.backward()
    tz = torch.from_numpy(z).requires_grad_()
    tl = F.nll_loss(F.log_softmax(tz, dim=1), torch.from_numpy(y0), torch.from_numpy(cw0))
    tl.backward()
    assert bool(torch.isnan(tl)) and bool(torch.isnan(l0))
    assert torch.equal(torch.isnan(z0.grad).cpu(), torch.isnan(tz.grad))
    s = ops.cross_entropy2d(zt, torch.from_numpy(y).to(dev), torch.from_numpy(cw).to(dev), size_average=False)
    W = float(cw[y[y >= 0]].sum())
    assert abs(float(s) - l64 * W) <= 3e-6 * abs(l64 * W)
    # identical heads: zero discrepancy, zero gradient
    d = ops.diff2d(zt.detach().clone().requires_grad_(), zt.detach().clone().requires_grad_())
    assert float(d) == 0.0
    with pytest.raises(RuntimeError, match="48 classes"):
        ops.mcd_losses(torch.zeros(1, 64, 2, 2, device=dev), None, None, None)
    with pytest.raises(ValueError):
        ops.mcd_losses(zt.detach(), None, torch.zeros(n, h + 1, w, dtype=torch.int64, device=dev), None, ce_coef=1.0)


def test_sgd_kernel_and_flat_optimizer():
    dev = _dev()
    from mcdseg.optim import FlatSGD
    g = torch.Generator().manual_seed(3)
    shapes = [(16, 6, 7, 7), (16,), (16,), (33, 5, 3, 3), (41, 1, 16, 16), (7,)]
    ref_params = [torch.nn.Parameter(torch.randn(s, generator=g)) for s in shapes]
    our_params = [torch.nn.Parameter(p.detach().clone().to(dev)) for p in ref_params]
    ref = torch.optim.SGD(ref_params, lr=1e-2, momentum=0.9, weight_decay=2e-3)
    ours = FlatSGD(our_params, lr=1e-2, momentum=0.9, weight_decay=2e-3)
    for step in range(4):
        ref.zero_grad(), ours.zero_grad()
        for i, (a, b) in enumerate(zip(ref_params, our_params)):
            if step == 2 and i == 3:
                continue  # a parameter without gradient is skipped by torch -- and by us
            gr = torch.randn(a.shape, generator=g)
            a.grad = gr.clone()
            b.grad = gr.to(dev)
        ref.step(), ours.step()
        if step == 1:  # checkpoint round trip in torch's layout (adapt_trainer.py:29-59, 232-245)
            sd = ours.state_dict()
            assert sorted(sd.keys()) == ["param_groups", "state"]
            assert set(sd["state"][0].keys()) == {"momentum_buffer"}
            ref.load_state_dict({"state": {k: {"momentum_buffer": v["momentum_buffer"].cpu()} for k, v in sd["state"].items()},
                                 "param_groups": ref.state_dict()["param_groups"]})
            ours.load_state_dict(sd)
    for a, b in zip(ref_params, our_params):
        _assert_close(b, a, 2e-6, "param")
        _assert_close(ours.state[b]["momentum_buffer"], ref.state[a]["momentum_buffer"], 2e-6, "momentum")
    fp, fg, fv = ours.flat_buffers()
    assert all(p.data_ptr() >= fp.data_ptr() and p.data_ptr() < fp.data_ptr() + 4 * fp.numel() for p in our_params)


@pytest.mark.parametrize("shape", [(2, 41, 8, 12), (1, 3, 5, 7), (2, 3, 1, 9)])
def test_bilinear8_fwd_bwd(shape):
    dev = _dev()
    from mcdseg import ops
    g = torch.Generator().manual_seed(9)
    x = torch.randn(shape, generator=g)
    x64 = x.double().requires_grad_()
    ref = F.interpolate(x64, scale_factor=8, mode="bilinear", align_corners=False)
    gy = torch.randn(ref.shape, generator=g)
    (gx,) = torch.autograd.grad(ref, x64, gy.double())
    xd = x.to(dev).requires_grad_()
    y = ops.bilinear8(xd)
    _assert_close(y, ref, 2e-6, "bilinear8 fwd")
    y.backward(gy.to(dev))
    _assert_close(xd.grad, gx, 1e-5, "bilinear8 bwd")


@pytest.mark.parametrize("n", [2 * 3 * 64 * 96, 1001])
def test_mse(n):
    dev = _dev()
    from mcdseg import ops
    g = torch.Generator().manual_seed(n)
    p, t = torch.randn(n, generator=g), torch.randn(n, generator=g)
    p64 = p.double().requires_grad_()
    ref = F.mse_loss(p64, t.double())
    (gp,) = torch.autograd.grad(3.0 * ref, p64)
    pd = p.to(dev).requires_grad_()
    loss = ops.mse_loss(pd, t.to(dev))
    (3.0 * loss).backward()
    assert abs(float(loss) - float(ref)) <= 2e-6 * float(ref)
    _assert_close(pd.grad, gp, 2e-6, "mse grad")


def test_conv_bias_bn_relu():
    """CBR of the multitask decoders (models/dilated_fcn.py:632-644): conv WITH bias, then BN, then ReLU."""
    dev = _dev()
    from mcdseg import ops
    from models.drn import BatchNorm2d, Conv2d
    g = torch.Generator().manual_seed(21)
    conv, bn = Conv2d(48, 40, 3, padding=1, bias=True), BatchNorm2d(40)
    with torch.no_grad():
        conv.weight.copy_(torch.randn(conv.weight.shape, generator=g) * 0.05)
        conv.bias.copy_(torch.randn(40, generator=g))
        bn.weight.copy_(1 + 0.2 * torch.randn(40, generator=g))
        bn.bias.copy_(0.1 * torch.randn(40, generator=g))
    x = torch.randn(2, 48, 10, 12, generator=g)
    gy = torch.randn(2, 40, 10, 12, generator=g)
    t = [v.detach().double().requires_grad_() for v in (x, conv.weight, conv.bias, bn.weight, bn.bias)]
    rm, rv = torch.zeros(40, dtype=torch.float64), torch.ones(40, dtype=torch.float64)
    o = F.relu(F.batch_norm(F.conv2d(t[0], t[1], t[2], padding=1), rm, rv, t[3], t[4], training=True, momentum=0.1, eps=1e-5))
    grads = torch.autograd.grad(o, t, gy.double())
    conv.to(dev), bn.to(dev)
    xd = x.to(dev).requires_grad_()
    y = ops.conv_bn_act(xd, conv, bn, relu=True)
    _assert_close(y, o, 3e-5, "forward")
    _assert_close(bn.running_mean, rm, 1e-5, "running_mean (includes the conv bias)")
    y.backward(gy.to(dev))
    for name, a, b in zip(["dx", "dw", "dgamma", "dbeta"], [xd.grad, conv.weight.grad, bn.weight.grad, bn.bias.grad],
                          [grads[0], grads[1], grads[3], grads[4]]):
        _assert_close(a, b, 1e-4, name)
    # d(bias) vanishes analytically (BN removes the channel mean): both are rounding noise around zero
    assert float(conv.bias.grad.abs().max()) <= 1e-4 * float(grads[1].abs().max()) * 120


def test_weight_update_between_forward_and_backward_is_refused():
    """The packed data-gradient image of a convolution is shared by every graph that used the weight and is re-packed in place after an
    optimizer step (lazily, by the next forward): from then on a backward pass through a graph recorded BEFORE the step would silently
    multiply by the new weights.  It raises instead -- torch's "modified by an inplace operation" for a saved tensor -- while stepping ANOTHER optimizer in between
    (the reuse of step B's target forward: only the classifiers' optimizer steps) leaves the graph valid."""
    dev = _dev()
    from mcdseg import ops
    from mcdseg.optim import FlatSGD
    from models.drn import BatchNorm2d, Conv2d
    g = torch.Generator().manual_seed(9)
    conv, bn = Conv2d(32, 64, 3, padding=1, bias=False).to(dev), BatchNorm2d(64).to(dev)
    other = Conv2d(16, 16, 3, padding=1, bias=False).to(dev)
    x = torch.randn(2, 32, 10, 12, generator=g).to(dev).requires_grad_()
    gy = torch.randn(2, 64, 10, 12, generator=g).to(dev)
    opt = FlatSGD(list(conv.parameters()) + list(bn.parameters()), lr=0.1, momentum=0.9)
    opt_other = FlatSGD(other.parameters(), lr=0.1, momentum=0.9)
    ops.conv_bn_act(x, conv, bn, relu=True).backward(gy)     # gradients for the step below
    other.weight.grad = torch.zeros_like(other.weight)
    y = ops.conv_bn_act(x, conv, bn, relu=True)               # graph recorded with the current weights
    opt_other.step()                                          # another optimizer: this graph stays valid
    y.backward(gy, retain_graph=True)
    opt.step()                                                # the graph's own weights move ...
    y.backward(gy, retain_graph=True)                         # ... but the packed image still holds the recorded ones: still exact
    ops.conv_bn_act(x.detach(), conv, bn, relu=True)          # the next forward re-packs the image in place
    with pytest.raises(RuntimeError, match="modified between a forward pass and its backward"):
        y.backward(gy)


def test_deferred_weight_gradients_equal_the_serial_schedule(monkeypatch):
    """MCDSEG_OVERLAP_WGRAD=2 (the default): inside ``ops.late_weight_grads`` the weight gradient of a fused group runs on a
    low-priority side stream and reaches autograd through a ``_LateGrad`` identity node that waits for that stream first.  Everything
    that could read the gradient early must therefore still see the finished tensor: a weight used twice in one graph (the engine
    sums the two gradients), a gradient already present (AccumulateGrad adds in place), a tensor hook, torch.autograd.grad.  All
    bit-equal to the one-stream schedule -- with every kernel group on the side stream delayed by ~10 ms, so that a missing wait
    shows at once instead of once in a thousand runs."""
    dev = _dev()
    from mcdseg import ops
    from models.drn import BatchNorm2d, Conv2d
    g = torch.Generator().manual_seed(31)
    convs = torch.nn.ModuleList([Conv2d(64, 64, 3, padding=d, dilation=d, bias=False) for d in (1, 2, 1)]).to(dev)
    bns = [BatchNorm2d(64).to(dev) for _ in convs]
    with torch.no_grad():
        for c in convs:
            c.weight.copy_((torch.randn(c.weight.shape, generator=g) * 0.05).to(dev))
    x = torch.randn(4, 64, 96, 128, generator=g).to(dev)
    gy = torch.randn(4, 64, 96, 128, generator=g).to(dev)
    params = [c.weight for c in convs] + [b.weight for b in bns] + [b.bias for b in bns]
    seen, on_side = [], []

    def chain(inp, twice):
        with ops.late_weight_grads(convs):
            h = inp
            for k, (c, b) in enumerate(zip(convs, bns)):
                h = ops.conv_bn_act(h, c, b, relu=True)
                if twice == "middle" and k == 1:
                    h = ops.conv_bn_act(h, c, b, relu=True)  # the same weight a second time in this graph (no alias left: main stream)
            if twice == "first":
                h = ops.conv_bn_act(h, convs[0], bns[0], relu=True)
            return h

    inner = ops._conv_wgrad

    def lagging(*a, **kw):
        if ops._LAUNCH.stream is not None:
            on_side.append(1)
            with torch.cuda.stream(ops._LAUNCH.stream):
                torch.cuda._sleep(20_000_000)
        return inner(*a, **kw)
    monkeypatch.setattr(ops, "_conv_wgrad", lagging)

    def run(mode, twice=None, accumulate=False, hook=False):
        monkeypatch.setattr(ops, "OVERLAP_WGRAD", mode)
        for p_ in params:
            p_.grad = None
        handle = convs[1].weight.register_hook(lambda gr: seen.append(float(gr.abs().sum())) or None) if hook else None
        xin = x.clone().requires_grad_()
        chain(xin, twice).backward(gy)
        if accumulate:
            chain(xin, twice).backward(gy)
        if handle is not None:
            handle.remove()
        assert not ops._PENDING and not ops._HELD, "a backward pass ended with an unjoined weight gradient"
        assert all(getattr(c, "_w_late", None) is None for c in convs)
        return [xin.grad.clone()] + [p_.grad.clone() for p_ in params]

    names = ["dx"] + ["dw%d" % k for k in range(3)] + ["dgamma%d" % k for k in range(3)] + ["dbeta%d" % k for k in range(3)]
    for kw in (dict(), dict(twice="middle"), dict(twice="first"), dict(accumulate=True), dict(hook=True), dict(twice="first", accumulate=True)):
        ref = run("0", **kw)
        assert not on_side
        del seen[:]
        got = run("2", **kw)
        assert len(on_side) >= 2, "the weight gradients of the groups with companions did not run on the side stream"
        del on_side[:]
        for name, a, b in zip(names, got, ref):
            assert torch.equal(a, b), "%s differs between the deferred and the serial schedule (%s)" % (name, kw)
        if kw.get("hook"):
            assert len(seen) == 1 and seen[0] == float(ref[2].abs().sum()), "the hook saw an unfinished weight gradient"
    # torch.autograd.grad (no AccumulateGrad at all), and a convolution outside the context (no alias: main stream)
    monkeypatch.setattr(ops, "OVERLAP_WGRAD", "0")
    xin = x.clone().requires_grad_()
    ref = torch.autograd.grad(chain(xin, None), [xin] + params[:3], gy)
    monkeypatch.setattr(ops, "OVERLAP_WGRAD", "2")
    got = torch.autograd.grad(chain(xin, None), [xin] + params[:3], gy)
    for a, b in zip(got, ref):
        assert torch.equal(a, b)
    del on_side[:]
    h = ops.conv_bn_act(ops.conv_bn_act(xin, convs[0], bns[0], relu=True), convs[1], bns[1], relu=True)
    h.backward(gy)
    assert not on_side and not ops._PENDING and not ops._HELD


def test_conv_batch_split_for_large_operands(monkeypatch):
    """Operands above the 2 GiB launch limit (cfg5's 2048-channel layers at N = 32) are cut along N on the host; exercised here
    with a tiny limit on a chain of two fused groups, so that the second group's forward, data gradient and weight gradient run in
    slices WITH their companions (mcdseg_conv_desc.Ncb: the pieces of a slice are not adjacent) -- asserted by kernel name."""
    dev = _dev()
    from mcdseg import ops
    from models.drn import BatchNorm2d, Conv2d
    g = torch.Generator().manual_seed(2)
    conv1, bn1 = Conv2d(24, 40, 3, padding=2, dilation=2, bias=False).to(dev), BatchNorm2d(40).to(dev)
    conv2, bn2 = Conv2d(40, 136, 3, padding=1, bias=False).to(dev), BatchNorm2d(136).to(dev)
    x = torch.randn(5, 24, 12, 14, generator=g).to(dev).requires_grad_()
    gy = torch.randn(5, 136, 12, 14, generator=g).to(dev)
    params = (conv1.weight, bn1.weight, bn1.bias, conv2.weight, bn2.weight, bn2.bias)

    def run():
        for bn in (bn1, bn2):
            bn.running_mean.zero_(), bn.running_var.fill_(1), bn.num_batches_tracked.zero_()
        for t in (x,) + params:
            t.grad = None
        names = []

        class _Names:
            def wants(self, name):
                names.append(name)
                return False
        prev, ops.LAUNCH_TIMER = ops.LAUNCH_TIMER, _Names()
        try:
            y = ops.conv_bn_act(ops.conv_bn_act(x, conv1, bn1, relu=True), conv2, bn2, relu=True)
            y.backward(gy)
        finally:
            ops.LAUNCH_TIMER = prev
        return names, [t.detach().clone() for t in (y, x.grad) + tuple(p.grad for p in params) + (bn1.running_var, bn2.running_var)]

    _, whole = run()
    hw = 12 * 14
    monkeypatch.setattr(ops, "MAX_CONV_BYTES", 2 * 4 * 136 * hw)  # two images of the widest tensor -> pieces of 2, 2, 1
    desc = ops.conv_desc((5, 40, 12, 14), conv2.weight.shape, 1, 1, 1)
    assert ops._batch_pieces(desc) == [(0, 2), (2, 4), (4, 5)]
    names, split = run()
    if ops.CONV_MATH in ops.MATH_ID:  # the second group's three passes ran from the companions, slice by slice
        pre = [nm for nm in names if (nm.startswith("conv_gemm_split_kernel") and nm.endswith("true>")) or nm.startswith("conv_gemm_split_pp_kernel")]
        assert len(pre) >= 6, names
        assert sum(nm.startswith("conv_wgrad_split_tr") for nm in names) >= 3, names
    for name, a, b in zip(["y", "dx", "dw1", "dgamma1", "dbeta1", "dw2", "dgamma2", "dbeta2", "running_var1", "running_var2"], split, whole):
        _assert_close(a, b, 5e-6, name)


def test_folded_bn_inference_and_predict_tail():
    dev = _dev()
    from mcdseg import ops
    from models.drn import BatchNorm2d, Conv2d
    g = torch.Generator().manual_seed(31)
    conv, bn = Conv2d(24, 40, 3, padding=2, dilation=2, bias=True), BatchNorm2d(40)
    with torch.no_grad():
        conv.weight.copy_(torch.randn(conv.weight.shape, generator=g) * 0.08)
        conv.bias.copy_(torch.randn(40, generator=g) * 0.2)
        bn.weight.copy_(1 + 0.2 * torch.randn(40, generator=g)), bn.bias.copy_(0.1 * torch.randn(40, generator=g))
        bn.running_mean.copy_(0.1 * torch.randn(40, generator=g)), bn.running_var.copy_(0.5 + torch.rand(40, generator=g))
    x = torch.randn(3, 24, 9, 11, generator=g)
    res = torch.randn(3, 40, 9, 11, generator=g)
    ref = F.relu(F.batch_norm(F.conv2d(x.double(), conv.weight.double(), conv.bias.double(), padding=2, dilation=2),
                              bn.running_mean.double(), bn.running_var.double(), bn.weight.double(), bn.bias.double(), training=False,
                              eps=1e-5) + res.double())
    conv.to(dev), bn.to(dev).eval()
    with torch.no_grad():
        y = ops.conv_bn_act(x.to(dev), conv, bn, relu=True, residual=res.to(dev))   # folded path (no grad, eval)
    _assert_close(y, ref, 2e-5, "folded conv+BN+residual+ReLU")
    xg = x.to(dev).requires_grad_()
    y2 = ops.conv_bn_act(xg, conv, bn, relu=True, residual=res.to(dev))              # eval BN with a tape (fix_bn training)
    _assert_close(y2, ref, 2e-5, "eval-mode BN, autograd path")
    # argmax / entropy tail
    z1 = torch.randn(2, 41, 7, 9, generator=g) * 2
    z2 = torch.randn(2, 41, 7, 9, generator=g) * 2
    lab, ent = ops.predict_labels(z1.to(dev), z2.to(dev), 40)
    o = (z1 + z2) / 2
    assert torch.equal(lab.cpu(), o[:, :40].argmax(1).to(torch.uint8))
    p = torch.softmax(o.double(), dim=1)
    assert abs(float(ent) - float(-(p * torch.log(p + 1e-6)).mean())) <= 1e-5
    lab1, _ = ops.predict_labels(z1.to(dev), None, 41)
    assert torch.equal(lab1.cpu(), z1.argmax(1).to(torch.uint8))


@pytest.mark.parametrize("case", [(16, 32, 3, 21, 27, 2), (32, 64, 3, 20, 28, 2), (32, 64, 1, 20, 28, 2), (32, 64, 1, 21, 27, 2), (64, 128, 3, 13, 19, 3),
                                  (64, 128, 1, 12, 20, 2), (128, 256, 3, 33, 40, 2), (32, 64, 3, 64, 96, 4)], ids=lambda c: "x".join(map(str, c)))
def test_stride2_data_gradient_forms_are_bitwise(case, libopt):
    """the three tilings of a stride-2 data gradient (library option DGRAD_INTERLEAVE: 0 = the four parity classes one after the other,
    1 = interleaved on neighbouring tiles, 2 = ROW classes: both x-parities of a row in one tile, dense stores) sum the same products in
    the same order: bit for bit one result -- from the fp32 operand, from the pre-split companion, and with the other gradient of the
    same tensor added in the epilogue; and that result is the fp64 data gradient within the split arithmetic's error"""
    dev = _dev()
    from mcdseg import ops
    cin, cout, k, h, w, n = case
    x, wt, _, s, pad, d = _conv_inputs((cin, cout, k, 2, 1, h, w, n, False), 21)
    desc = ops.conv_desc(x.shape, wt.shape, s, pad, d)
    packed = ops.PackedWeights()
    wf, wd, _ = packed.get(wt.to(dev), desc)
    gy = torch.randn(n, cout, desc.Ho, desc.Wo, generator=torch.Generator().manual_seed(22)).to(dev)
    other = torch.randn(x.shape, generator=torch.Generator().manual_seed(23)).to(dev)
    gy_cb, gy_bound = ops.split_companion(gy)
    got = {}
    for form in (0, 1, 2):
        libopt(DGRAD_INTERLEAVE=form)
        got[form] = (ops._conv_dgrad(desc, gy, wd, w_bound=packed.w_bound),
                     ops._conv_dgrad(desc, None, wd, dy_cb=gy_cb, dy_bound=gy_bound, w_bound=packed.w_bound) if gy_cb is not None else None,
                     ops._conv_dgrad(desc, gy, wd, dy_cb=gy_cb, dy_bound=gy_bound, w_bound=packed.w_bound, addend=other))
    for form in (1, 2):
        for a, b, what in zip(got[0], got[form], ("fp32 operand", "companion", "with addend")):
            if a is not None:
                assert torch.equal(a.view(torch.int32), b.view(torch.int32)), "form %d, %s" % (form, what)
    x64 = x.double().requires_grad_()
    ref = torch.autograd.grad(F.conv2d(x64, wt.double(), None, stride=2, padding=pad), x64, gy.double().cpu())[0]
    _assert_close(got[2][0], ref, 2e-5, "dgrad")
    _assert_close(got[2][2], ref + other.double().cpu(), 2e-5, "dgrad + addend")


@pytest.mark.parametrize("case", [c for c in CONV_CASES if min(c[0], c[1]) >= 16], ids=lambda c: "x".join(map(str, c[:5])))
def test_conv_split_accuracy(case, monkeypatch):
    """split-precision paths (f16x3: two scaled fp16 pieces, three cross terms; bf16x6: three bf16 pieces, six terms): against
    fp64 they must be as accurate as the exact-fp32 MFMA path (within 2x of its error, and within the same 2e-5-of-scale bound)."""
    dev = _dev()
    from mcdseg import ops
    x, wt, b, s, pad, d = _conv_inputs(case, 17)
    desc = ops.conv_desc(x.shape, wt.shape, s, pad, d)
    xg, wg = x.to(dev), wt.to(dev)
    x64, w64 = x.double().requires_grad_(), wt.double().requires_grad_()
    ref = F.conv2d(x64, w64, b.double() if b is not None else None, stride=s, padding=pad, dilation=d)
    gy = torch.randn(ref.shape, generator=torch.Generator().manual_seed(18))
    gx_ref, gw_ref = torch.autograd.grad(ref, [x64, w64], gy.double())
    errs = {}
    for math in ("f32", "bf16x6", "f16x3"):
        monkeypatch.setattr(ops, "CONV_MATH", math)
        pk = ops.PackedWeights()
        wf, wd, mpf = pk.get(wg, desc)
        assert ops._is_split(wf) == (math != "f32")
        y, part, rows = ops._conv_fprop(desc, xg, wf, b.to(dev) if b is not None else None, True, mpf, w_bound=pk.w_bound)
        dx = ops._conv_dgrad(desc, gy.to(dev), wd, w_bound=pk.w_bound)
        dw = ops._conv_wgrad(desc, xg, gy.to(dev))
        errs[math] = (_maxerr(y, ref), _maxerr(dx, gx_ref), _maxerr(dw, gw_ref))
        _assert_close(y, ref, 2e-5, math + " fprop")
        _assert_close(dx, gx_ref, 2e-5, math + " dgrad")
        _assert_close(dw, gw_ref, 2e-5, math + " wgrad")
    for math in ("bf16x6", "f16x3"):
        for k in (0, 1, 2):
            e32, es = errs["f32"][k][0], errs[math][k][0]
            assert es <= max(2.0 * e32, 2e-6 * errs["f32"][k][1]), "%s err %.3e vs f32-MFMA err %.3e" % (math, es, e32)


@pytest.mark.parametrize("cin,cout,stride", [(16, 16, 1), (16, 32, 2)])
def test_full_resolution_layers_of_cfg5_in_one_launch(cin, cout, stride, monkeypatch):
    """BASELINE config 5's thin layers at its stated batch -- 32 x 720 x 1280, 1.89 GB per 16-channel tensor, 3 % below the 2 GiB a
    launch can address -- are no longer cut along N (ops._batch_pieces: the split kernels need no tile of slack), so they keep their
    companions: forward, data gradient and weight gradient from the companions in ONE launch each.  Checked at that size:
    the last images' forward / data-gradient values equal the same images processed alone bit for bit (different pixel tiles,
    different offsets near the 2 GiB end), and the weight gradient equals the sum of the batch quarters' to fp32 rounding."""
    dev = _dev()
    from mcdseg import ops
    monkeypatch.setattr(ops, "CONV_MATH", "f16x3")
    n, h, w = 32, 720, 1280
    g = torch.Generator(device=dev).manual_seed(61)
    x = torch.randn(n, cin, h, w, device=dev, generator=g)
    wt = torch.randn(cout, cin, 3, 3, device=dev, generator=g) * (2.0 / (9 * cout)) ** 0.5
    desc = ops.conv_desc(x.shape, wt.shape, stride, 1, 1)
    assert ops._batch_pieces(desc) == [(0, n)]
    gy = torch.randn(n, cout, desc.Ho, desc.Wo, device=dev, generator=g)
    pk = ops.PackedWeights()
    wf, wd, mpf = pk.get(wt, desc)
    x_cb, x_bound = ops.split_companion(x)
    gy_cb, gy_bound = ops.split_companion(gy)
    y, part, rows = ops._conv_fprop(desc, x, wf, None, True, mpf, x_cb, x_bound, pk.w_bound)
    dx = ops._conv_dgrad(desc, None, wd, gy_cb, gy_bound, pk.w_bound)
    dw = ops._conv_wgrad(desc, x, gy, x_cb, gy_cb, x_bound, gy_bound)
    assert bool(torch.isfinite(dw).all())
    # the last three images alone (their own companions, the full tensors' bounds)
    tail = slice(n - 3, n)
    xs, gs = x[tail].contiguous(), gy[tail].contiguous()
    ds = ops.conv_desc(xs.shape, wt.shape, stride, 1, 1)
    xs_cb, _ = ops.split_companion(xs, x_bound)
    gs_cb, _ = ops.split_companion(gs, gy_bound)
    ys, _, _ = ops._conv_fprop(ds, xs, wf, None, True, mpf, xs_cb, x_bound, pk.w_bound)
    dxs = ops._conv_dgrad(ds, None, wd, gs_cb, gy_bound, pk.w_bound)
    assert torch.equal(y[tail], ys) and torch.equal(dx[tail], dxs)
    del ys, dxs, xs_cb, gs_cb, y, dx
    acc = torch.zeros_like(dw, dtype=torch.float64)
    for q in range(4):
        sl = slice(8 * q, 8 * q + 8)
        xq, gq = x[sl].contiguous(), gy[sl].contiguous()
        dq = ops.conv_desc(xq.shape, wt.shape, stride, 1, 1)
        xq_cb, _ = ops.split_companion(xq, x_bound)
        gq_cb, _ = ops.split_companion(gq, gy_bound)
        acc += ops._conv_wgrad(dq, xq, gq, xq_cb, gq_cb, x_bound, gy_bound).double()
    _assert_close(dw, acc, 1e-5, "weight gradient of the whole batch vs the sum over its quarters")


def _round_like_f16x1(t, bound):
    """what SplitF16x1 multiplies: the leading scaled fp16 piece, s * fp16(t / s), s = 2^(e - 15) with bound = m 2^e (csrc/split.h)"""
    import math
    b = float(bound)
    scale = 2.0 ** (math.frexp(b)[1] - 15) if b > 0 else 1.0
    return (t.double() / scale).to(torch.float16).double() * scale


@pytest.mark.parametrize("case", [(128, 128, 3, 1, 2, 13, 19, 2), (64, 96, 3, 1, 1, 12, 16, 2), (256, 512, 3, 1, 4, 30, 40, 6), (16, 16, 3, 1, 1, 21, 45, 2),
                                  (136, 200, 1, 1, 1, 11, 13, 3), (64, 128, 3, 2, 1, 15, 17, 2)], ids=lambda c: "x".join(map(str, c)))
def test_conv_reduced_precision_f16x1(case, monkeypatch):
    """``MCDSEG_CONV_MATH=f16x1`` (--dtype f16; BASELINE config 5's reduced-precision intent): f16x3's operands, leading term only.
    The kernels must compute EXACTLY that arithmetic: against an fp64 convolution of the operands rounded to their leading scaled
    fp16 piece the result is as tight as every other kernel test (2e-5 of the scale) -- forward, data gradient and weight
    gradient, from fp32 operands and from the companions (all tile shapes these cases select, the 256 x 128 weight-gradient tile
    included).  Against the UNROUNDED fp64 result the deviation is the operands' 11-bit rounding: measured 2e-4 .. 6e-4 of the
    scale, bounded here at 2e-3 (the thin 16-channel layers keep their three-term window kernels and stay at 2e-5)."""
    dev = _dev()
    from mcdseg import ops
    monkeypatch.setattr(ops, "CONV_MATH", "f16x1")
    cin, cout, k, s, d, h, w, n = case
    x, wt, _, s, pad, d = _conv_inputs((cin, cout, k, s, d, h, w, n, False), 51)
    desc = ops.conv_desc(x.shape, wt.shape, s, pad, d)
    gy = torch.randn(n, cout, desc.Ho, desc.Wo, generator=torch.Generator().manual_seed(52))
    xg, wg, gyg = x.to(dev), wt.to(dev), gy.to(dev)
    pk = ops.PackedWeights()
    wf, wd, mpf = pk.get(wg, desc)
    x_cb, x_bound = ops.split_companion(xg)
    gy_cb, gy_bound = ops.split_companion(gyg)
    thin = cin <= 16  # the window kernels (three terms) take these from the companions
    xr, wr, gr = _round_like_f16x1(x, x_bound), _round_like_f16x1(wt, pk.w_bound), _round_like_f16x1(gy, gy_bound)
    ref_y = F.conv2d(xr, wr, None, s, pad, d)
    x64, w64 = xr.clone().requires_grad_(), wr.clone().requires_grad_()
    (gx_w,) = torch.autograd.grad(F.conv2d(x64, wr, None, s, pad, d), [x64], gr)      # dgrad: rounded dy, rounded w
    (gw_x,) = torch.autograd.grad(F.conv2d(xr, w64, None, s, pad, d), [w64], gr)      # wgrad: rounded x, rounded dy
    xe, we = x.double().requires_grad_(), wt.double().requires_grad_()
    ye = F.conv2d(xe, we, None, s, pad, d)
    gxe, gwe = torch.autograd.grad(ye, [xe, we], gy.double())
    y0, _, _ = ops._conv_fprop(desc, xg, wf, None, True, mpf, None, x_bound, pk.w_bound)
    dx0 = ops._conv_dgrad(desc, gyg, wd, None, gy_bound, pk.w_bound)
    dw0 = ops._conv_wgrad(desc, xg, gyg, None, None, x_bound, gy_bound)
    y1, _, _ = ops._conv_fprop(desc, xg, wf, None, True, mpf, x_cb, x_bound, pk.w_bound)
    dx1 = ops._conv_dgrad(desc, None, wd, gy_cb, gy_bound, pk.w_bound)
    dw1 = ops._conv_wgrad(desc, xg, gyg, x_cb, gy_cb, x_bound, gy_bound)
    if not thin:
        for name, got, ref in (("fprop", y0, ref_y), ("dgrad", dx0, gx_w), ("fprop (companion)", y1, ref_y), ("dgrad (companion)", dx1, gx_w),
                               ("wgrad (companion)", dw1, gw_x)):
            _assert_close(got, ref, 2e-5, "f16x1 " + name + " vs the rounded-operand arithmetic")
        if min(cin, cout) > 64:  # the split plan from fp32 operands (thinner layers take the f32 weight-gradient kernels there)
            _assert_close(dw0, gw_x, 2e-5, "f16x1 wgrad vs the rounded-operand arithmetic")
        assert torch.equal(y0, y1) and torch.equal(dx0, dx1)
    for name, got, ref in (("fprop", y1, ye.detach()), ("dgrad", dx1, gxe), ("wgrad", dw1, gwe)):
        _assert_close(got, ref, 2e-5 if thin else 2e-3, "f16x1 " + name + " vs fp64")
    if not thin:
        assert _maxerr(y1, ye.detach())[0] > 2e-5 * _maxerr(y1, ye.detach())[1], "suspiciously exact: is the one-term arithmetic running?"


def test_f16x3_scaling_covers_the_fp32_range(monkeypatch):
    """The per-tensor power-of-two scale makes the fp16 pieces independent of the operand's magnitude: tensors scaled by
    2^-40 ... 2^+40 give the correspondingly scaled result to the same relative accuracy (no overflow to inf, no flush to 0),
    and a tensor with a 2^20 dynamic range keeps its small entries' contribution."""
    dev = _dev()
    from mcdseg import ops
    monkeypatch.setattr(ops, "CONV_MATH", "f16x3")
    x, wt, _, s, pad, d = _conv_inputs((64, 128, 3, 1, 1, 12, 16, 2, False), 19)
    desc = ops.conv_desc(x.shape, wt.shape, 1, pad, d)
    ref = F.conv2d(x.double(), wt.double(), None, 1, pad, d)
    for ex, ew in ((0, 0), (40, -30), (-40, 20), (60, 60), (-50, -50)):
        xs, ws = x * 2.0 ** ex, wt * 2.0 ** ew
        pk = ops.PackedWeights()
        wf, wd, mpf = pk.get(ws.to(dev), desc)
        y, _, _ = ops._conv_fprop(desc, xs.to(dev), wf, None, False, mpf, w_bound=pk.w_bound)
        _assert_close(y.double().cpu() * 2.0 ** (-ex - ew), ref, 2e-5, "fprop at 2^%d x 2^%d" % (ex, ew))
        cb, bound = ops.split_companion(xs.to(dev))
        y2, _, _ = ops._conv_fprop(desc, xs.to(dev), wf, None, False, mpf, cb, bound, pk.w_bound)
        assert torch.equal(y, y2)
    # wide dynamic range inside one tensor: channel 0 carries values 2^20 larger than the rest; outputs that only see the small
    # channels through zero weights on channel 0 must keep their accuracy
    xw = x.clone()
    xw[:, 0] *= 2.0 ** 20
    w0 = wt.clone()
    w0[:64, 0] = 0.0  # the first 64 output channels ignore the large input channel
    pk = ops.PackedWeights()
    wf, wd, mpf = pk.get(w0.to(dev), desc)
    y, _, _ = ops._conv_fprop(desc, xw.to(dev), wf, None, False, mpf, w_bound=pk.w_bound)
    refw = F.conv2d(xw.double(), w0.double(), None, 1, pad, d)
    _assert_close(y[:, :64], refw[:, :64], 2e-4, "small-magnitude channels beside a 2^20 larger one")
    _assert_close(y[:, 64:], refw[:, 64:], 2e-5, "large-magnitude channels")


@pytest.mark.parametrize("math", ["f16x3", "bf16x6"])
def test_presplit_operands_are_bitwise_equivalent(math, monkeypatch):
    """Splitting an activation into 16-bit pieces in its producer (bn_apply_cb / bn_bwd_apply_cb) or inside the consuming
    convolution's K loop is the same arithmetic: results must be bit-identical, forward and backward.  (f16x3: both paths must
    then use the same scale -- the in-loop path is handed the producer's bound.)"""
    dev = _dev()
    from mcdseg import ops
    from models.drn import BatchNorm2d, Conv2d
    monkeypatch.setattr(ops, "CONV_MATH", math)
    g = torch.Generator().manual_seed(41)
    c1, b1 = Conv2d(24, 40, 3, padding=1, bias=False).to(dev), BatchNorm2d(40).to(dev)     # 40 = 2.5 x 16: ragged last chunk
    c2, b2 = Conv2d(40, 64, 3, padding=2, dilation=2, bias=False).to(dev), BatchNorm2d(64).to(dev)
    c3, b3 = Conv2d(64, 32, 1, stride=2, bias=False).to(dev), BatchNorm2d(32).to(dev)
    x = torch.randn(3, 24, 14, 18, generator=g).to(dev)
    gy = torch.randn(3, 32, 7, 9, generator=g).to(dev)

    def run(presplit):
        monkeypatch.setattr(ops, "PRESPLIT", presplit)
        for m in (b1, b2, b3):
            m.running_mean.zero_(), m.running_var.fill_(1)
        xs = x.clone().requires_grad_()
        for p in list(c1.parameters()) + list(c2.parameters()) + list(c3.parameters()):
            p.grad = None
        y1 = ops.conv_bn_act(xs, c1, b1, relu=True)
        assert (ops._cb_of(y1)[0] is not None) == presplit
        y2 = ops.conv_bn_act(y1, c2, b2, relu=True, residual=None)
        y3 = ops.conv_bn_act(y2, c3, b3, relu=False)
        y3.backward(gy)
        return [t.detach().clone() for t in (y1, y2, y3, xs.grad, c1.weight.grad, c2.weight.grad, c3.weight.grad)]

    a, b = run(True), run(False)
    for name, u, v in zip(["y1", "y2", "y3", "dx", "dw1", "dw2", "dw3"], a, b):
        if name in ("dw2", "dw3") and math == "f16x3":
            # 40 -> 64 and (round 5: the 64-wide plan from 24 channels up) 64 -> 32 channels: with both companions the weight gradient
            # runs the tap-pair kernel in the split arithmetic, without them the f32-MFMA kernel -- same result to fp32 accuracy, not
            # the same bits
            _assert_close(u, v, 2e-5, "%s (split tap-pair kernel vs f32 kernel)" % name)
            continue
        assert torch.equal(u, v), "%s differs between pre-split and in-loop split (max %.3e)" % (name, float((u - v).abs().max()))


def test_compact_activation_storage_of_a_residual_block(monkeypatch):
    """MCDSEG_ACT_STORAGE=compact inside a trunk: the fused groups hand each other companions only (no fp32 activation is
    written, the residual add and the ReLU mask read the companion).  A residual block + projection shortcut, forward and
    backward, against the same block with fp32 activations (the companion carries 22 of fp32's 24 bits) and against fp64."""
    dev = _dev()
    from mcdseg import ops
    from models.drn import BasicBlock, BatchNorm2d, Conv2d, ConvBN, ConvBNReLU
    monkeypatch.setattr(ops, "CONV_MATH", "f16x3")
    g = torch.Generator().manual_seed(61)
    stem = ConvBNReLU(Conv2d(24, 64, 3, padding=1, bias=False), BatchNorm2d(64), torch.nn.ReLU(inplace=True)).to(dev)
    blk = BasicBlock(64, 128, stride=2, downsample=ConvBN(Conv2d(64, 128, 1, stride=2, bias=False), BatchNorm2d(128)),
                     dilation=(1, 1)).to(dev)
    blk2 = BasicBlock(128, 128, dilation=(2, 2)).to(dev)
    tail = ConvBNReLU(Conv2d(128, 72, 3, padding=1, bias=False), BatchNorm2d(72), torch.nn.ReLU(inplace=True)).to(dev)
    mods = [stem, blk, blk2, tail]
    with torch.no_grad():
        for m in mods:
            for p in m.parameters():
                if p.dim() == 1:
                    p.copy_(1 + 0.2 * torch.randn(p.shape, generator=g).to(dev) if p.mean() > 0.5 else 0.1 * torch.randn(p.shape, generator=g).to(dev))
    x = torch.randn(3, 24, 20, 24, generator=g).to(dev)
    gy = torch.randn(3, 72, 10, 12, generator=g).to(dev)

    def run(storage):
        monkeypatch.setattr(ops, "ACT_STORAGE", storage)
        for m in mods:
            m.zero_grad()
            for b in m.modules():
                if isinstance(b, BatchNorm2d):
                    b.running_mean.zero_(), b.running_var.fill_(1), b.num_batches_tracked.zero_()
        xs = x.clone().requires_grad_()
        seen = []
        with ops.trunk_internal():
            h = stem(xs)
            seen.append(ops.is_virtual(h))
            h = blk(h)
            seen.append(ops.is_virtual(h))
            h = blk2(h)
            seen.append(ops.is_virtual(h))
        y = tail(h)  # the trunk's last layer: fp32
        assert not ops.is_virtual(y)
        y.backward(gy)
        grads = [xs.grad] + [p.grad for m in mods for p in m.parameters()]
        stats = [b.running_var.clone() for m in mods for b in m.modules() if isinstance(b, BatchNorm2d)]
        return seen, [t.detach().clone() for t in [y] + grads + stats]

    seen_c, c = run("compact")
    seen_f, f = run("fp32")
    assert seen_c == [True, True, True] and seen_f == [False, False, False]
    for i, (a, b) in enumerate(zip(c, f)):
        err, scale = _maxerr(a, b)
        assert err <= 2e-4 * scale, "tensor %d: compact vs fp32 activations differ by %.3e of scale %.3e" % (i, err, scale)
    # and the virtual activation reconstructs to the fp32 one within 2^-22
    monkeypatch.setattr(ops, "ACT_STORAGE", "compact")
    with torch.no_grad(), ops.trunk_internal():
        hv = stem(x)
    monkeypatch.setattr(ops, "ACT_STORAGE", "fp32")
    with torch.no_grad():
        hr = stem(x)
    assert ops.is_virtual(hv) and float((ops.materialize(hv) - hr).abs().max()) <= 3e-7 * float(hr.abs().max())


def test_stale_presplit_companion_is_not_used(monkeypatch):
    """An in-place write to an activation between two fused groups (the reference's statements run unchanged over the
    drop-in modules may do that) must not leave the next convolution reading the old pre-split image."""
    dev = _dev()
    from mcdseg import ops
    from models.drn import BatchNorm2d, Conv2d
    g = torch.Generator().manual_seed(43)
    c1, b1 = Conv2d(24, 64, 3, padding=1, bias=False).to(dev), BatchNorm2d(64).to(dev)
    c2, b2 = Conv2d(64, 72, 3, padding=1, bias=False).to(dev), BatchNorm2d(72).to(dev)
    x = torch.randn(2, 24, 12, 16, generator=g).to(dev)
    mask = (torch.rand(2, 64, 12, 16, generator=g) > 0.5).float().to(dev)
    gy = torch.randn(2, 72, 12, 16, generator=g).to(dev)

    def run(presplit):
        monkeypatch.setattr(ops, "PRESPLIT", presplit)
        for p in list(c1.parameters()) + list(c2.parameters()):
            p.grad = None
        with torch.no_grad():
            y1 = ops.conv_bn_act(x, c1, b1, relu=True)
            had = ops._cb_of(y1)[0] is not None
            y1.mul_(mask)         # in-place: bumps y1._version
            y1.add_(0.25)
            assert ops._cb_of(y1) == (None, None)
        y1 = y1.requires_grad_()
        y2 = ops.conv_bn_act(y1, c2, b2, relu=True)
        y2.backward(gy)
        return had, [t.detach().clone() for t in (y2, y1.grad, c2.weight.grad)]

    had, a = run(True)
    assert had == (ops.CONV_MATH != "f32")
    _, b = run(False)
    for name, u, v in zip(["y2", "dy1", "dw2"], a, b):
        assert torch.equal(u, v), "%s read a stale companion (max diff %.3e)" % (name, float((u - v).abs().max()))


# the large pixel-tile instantiations only run at benchmark-size grids: (Cin, Cout, k, dil, N, H, W)
BIG_TILE_CASES = [
    (64, 512, 3, 4, 1, 256, 256),    # fprop 64 -> 512, dilation 4: P = 65 536 = 512 pixel tiles x 2 row tiles
    (256, 512, 1, 1, 2, 181, 182),   # 1x1, ragged pixel count (65 884 is not a multiple of 128)
    (512, 64, 3, 2, 1, 256, 257),    # dgrad towards 512 channels (M = Cin), ragged
]


@pytest.mark.parametrize("case", BIG_TILE_CASES, ids=lambda c: "x".join(map(str, c)))
def test_conv_large_tile_kernels(case):
    """The tiles chosen at BASELINE batch sizes (csrc launch rule: pre-split operand, 512 rows, >= 1024 tile slots) against
    fp64, and bit-for-bit against the 128 x 128 kernel the same shape takes without a pre-split operand (same K order)."""
    dev = _dev()
    from mcdseg import ops
    if ops.CONV_MATH == "f32":
        pytest.skip("split-precision kernels only")
    cin, cout, k, d, n, h, w = case
    x, wt, _, s, pad, d = _conv_inputs((cin, cout, k, 1, d, h, w, n, False), 29)
    desc = ops.conv_desc(x.shape, wt.shape, 1, pad, d)
    pk = ops.PackedWeights()
    wf, wd, mpf = pk.get(wt.to(dev), desc)
    xg = x.to(dev)
    gy = torch.randn(n, cout, desc.Ho, desc.Wo, generator=torch.Generator().manual_seed(30))
    gyg = gy.to(dev)
    x64, w64 = x.double().requires_grad_(), wt.double().requires_grad_()
    ref = F.conv2d(x64, w64, None, 1, pad, d)
    (gx_ref,) = torch.autograd.grad(ref, [x64], gy.double())
    names = []

    class _Names:
        def wants(self, name):
            names.append(name)
            return False
    prev, ops.LAUNCH_TIMER = ops.LAUNCH_TIMER, _Names()
    try:
        x_cb, x_bound = ops.split_companion(xg)
        gy_cb, gy_bound = ops.split_companion(gyg)
        y_cb, part_cb, rows = ops._conv_fprop(desc, xg, wf, None, True, mpf, x_cb, x_bound, pk.w_bound)
        y, part, rows2 = ops._conv_fprop(desc, xg, wf, None, True, mpf, None, x_bound, pk.w_bound)
        dx_cb = ops._conv_dgrad(desc, None, wd, gy_cb, gy_bound, pk.w_bound)
        dx = ops._conv_dgrad(desc, gyg, wd, None, gy_bound, pk.w_bound)
    finally:
        ops.LAUNCH_TIMER = prev
    big = [nm for nm in names if "4, 2, 2, 2" in nm or "4, 4, 2, 2" in nm or "conv_gemm_split_pp_kernel" in nm]
    assert big, "no large-tile kernel ran: %s" % names
    _assert_close(y_cb, ref, 2e-5, "fprop (large tile)")
    _assert_close(dx_cb, gx_ref, 2e-5, "dgrad (large tile)")
    assert torch.equal(y_cb, y) and torch.equal(dx_cb, dx), "large tile differs from the 128x128 kernel"
    if rows == rows2:
        assert torch.equal(part_cb, part), "fused BN partial statistics differ"
    # the fused statistics reproduce the channel means / variances of the output
    L = ops.lib()
    import ctypes
    mean = torch.empty(cout, device=dev)
    rstd = torch.empty(cout, device=dev)
    ws = torch.empty(L.mcdseg_bn_stats_workspace_bytes(rows, cout) // 8 + 1, dtype=torch.float64, device=dev)
    gam, bet, yb = torch.full((cout,), 1.5, device=dev), torch.full((cout,), -0.25, device=dev), torch.empty(1, device=dev)
    ops.check(L.mcdseg_bn_stats_finalize(ops._p(part_cb), rows, cout, mpf, ops._p(mean), ops._p(rstd), None, None, None, 0.1, 1e-5,
                                         ops._p(gam), ops._p(bet), None, ops._p(yb), 1, ops._p(ws), ctypes.c_size_t(ws.numel() * 8),
                                         ops._stream()), "bn_stats_finalize")
    npix = n * desc.Ho * desc.Wo  # Samuelson bound of the BN output: |gamma| sqrt(n-1) + |beta|
    assert abs(float(yb) - (1.5 * (npix - 1) ** 0.5 + 0.25)) <= 1e-3 * float(yb)
    # one launch standing for two forward passes (solvers/solver.py: step B's target forward is step C's first): the running
    # statistics receive the update twice, running = 0.81 old + 0.19 stat, and num_batches_tracked counts both
    rm, rv = torch.full((cout,), 0.25, device=dev), torch.full((cout,), 1.5, device=dev)
    nbt = torch.zeros(1, dtype=torch.int64, device=dev)
    ops.check(L.mcdseg_bn_stats_finalize(ops._p(part_cb), rows, cout, mpf, ops._p(mean), ops._p(rstd), ops._p(rm), ops._p(rv), ops._p(nbt),
                                         0.1, 1e-5, ops._p(gam), ops._p(bet), None, ops._p(yb), 2, ops._p(ws), ctypes.c_size_t(ws.numel() * 8),
                                         ops._stream()), "bn_stats_finalize")
    assert int(nbt) == 2 and float((rm - (0.81 * 0.25 + 0.19 * mean)).abs().max()) <= 1e-6
    r = ref.detach()
    merr = float((mean.double().cpu() - r.mean((0, 2, 3))).abs().max())
    assert merr <= 1e-5 * float(r.std()), "fused BN mean off by %.3e (output std %.3e)" % (merr, float(r.std()))
    _assert_close(rstd, (r.var((0, 2, 3), unbiased=False) + 1e-5).rsqrt(), 1e-5, "fused BN rstd")


@pytest.mark.parametrize("rows,c,mp", [(480, 512, 512), (480, 128, 128), (7, 16, 32), (1000, 41, 64), (1024, 2048, 2048), (333, 24, 32)])
def test_bn_statistics_in_one_launch_are_bitwise_the_two_stage_result(rows, c, mp, monkeypatch, libopt):
    """``bn_stats_one_kernel`` (csrc/bn.hip): the merge of the convolution epilogue's partial rows into mean / rstd /
    running statistics / output bound as ONE launch for layers with few partial rows (<= 1024: the 1/8-resolution maps) -- the same fp64
    sums in the same order as ``bn_stats_partial_kernel`` + ``bn_stats_finalize_kernel`` (MCDSEG_BN_STATS_ONE=0), so every output is
    bit for bit theirs; also with two running-statistics updates, a residual bound, and ragged channel counts."""
    dev = _dev()
    import ctypes
    from mcdseg import ops
    L = ops.lib()
    g = torch.Generator().manual_seed(rows * 131 + c)
    part = torch.zeros(rows, 3, mp)
    part[:, 0, :c] = torch.randint(1, 161, (rows, 1), generator=g).float()          # counts: the same for every channel of a row
    part[:, 1, :c] = torch.randn(rows, c, generator=g) * 3.0 + 1.5                  # means
    part[:, 2, :c] = torch.rand(rows, c, generator=g) * 100.0                       # M2
    part = part.to(dev)
    gam, bet = torch.randn(c, generator=g).to(dev), torch.randn(c, generator=g).to(dev)
    resb = torch.tensor([2.75], device=dev)
    out = {}
    for mode in ("0", None):
        if mode is None:
            libopt(BN_STATS_ONE=1024)  # (the default)
        else:
            libopt(BN_STATS_ONE=int(mode))
        mean, rstd = torch.empty(c, device=dev), torch.empty(c, device=dev)
        rm, rv = torch.full((c,), 0.25, device=dev), torch.full((c,), 1.5, device=dev)
        nbt = torch.full((1,), 5, dtype=torch.int64, device=dev)
        yb = torch.full((1,), 1e30 if mode is None else 0.0, device=dev)  # (the one-launch form must not depend on what the scalar held)
        ws = torch.empty(L.mcdseg_bn_stats_workspace_bytes(rows, c) // 8 + 1, dtype=torch.float64, device=dev)
        names = []
        ops.check(L.mcdseg_bn_stats_finalize(ops._p(part), rows, c, mp, ops._p(mean), ops._p(rstd), ops._p(rm), ops._p(rv), ops._p(nbt), 0.1, 1e-5,
                                             ops._p(gam), ops._p(bet), ops._p(resb), ops._p(yb), 2, ops._p(ws), ctypes.c_size_t(ws.numel() * 8),
                                             ops._stream()), "bn_stats_finalize")
        torch.cuda.synchronize()
        out[mode] = (mean, rstd, rm, rv, nbt, yb)
    for a, b, what in zip(out["0"], out[None], ("mean", "rstd", "running_mean", "running_var", "num_batches_tracked", "y_bound")):
        assert torch.equal(a, b), "%s differs between the one-launch and the two-stage statistics" % what
    assert int(out[None][4]) == 7
    cnt = part[:, 0, 0].double().cpu()
    mu = part[:, 1, :c].double().cpu()
    n = cnt.sum()
    ref_mean = (cnt[:, None] * mu).sum(0) / n
    assert float((out[None][0].double().cpu() - ref_mean).abs().max()) <= 1e-6 * float(ref_mean.abs().max())


# (Cin, Cout, k, stride, dil, N, H, W, math): one or two rounds of 256 x 256 tiles for the 8-wave ping-pong kernel plus a ragged rest
PINGPONG_CASES = [
    (64, 512, 3, 1, 2, 2, 160, 131, "f16x3"),   # forward M = 512: 326 tiles -> one round (128 pixel tiles) + 9 152 ragged pixels on 256 x 128
    (512, 64, 3, 1, 4, 2, 160, 131, "f16x3"),   # its mirror: the DATA gradient has M = Cin = 512
    (64, 256, 3, 1, 1, 3, 150, 160, "f16x3"),   # M = 256 (one row tile): 256 of 281 pixel tiles, the rest on 128 x 128
    (48, 256, 1, 1, 1, 4, 128, 129, "f16x3"),   # 1x1: three K-steps (prologue and tail of the K loop only); 258 full tiles
    (24, 256, 3, 2, 1, 1, 512, 514, "f16x3"),   # stride-2 forward, ragged contraction (24 channels = 1.5 K-steps per tap)
    (64, 512, 3, 1, 2, 2, 160, 131, "f16x1"),   # the reduced-precision arithmetic (one term, one staged piece)
]


@pytest.mark.parametrize("case", PINGPONG_CASES, ids=lambda c: "x".join(map(str, c)))
def test_conv_pingpong_tile(case, monkeypatch, libopt):
    """``conv_gemm_split_pp_kernel`` (csrc/conv_gemm_split_pp.hip: eight waves in two groups half a K-step apart) takes whole rounds of
    one 256 x 256 tile per CU and, with its 256 x 128 tile, the remaining pixels.  Forward (+ fused BatchNorm partial rows) and data
    gradient of such a two-launch convolution against fp64, and BIT FOR BIT against the same convolution on the 4-wave tiles
    (MCDSEG_PINGPONG=0: same K order, same term order, same 64-pixel statistic rows), on the 256 x 128 tile alone and with the 4-wave
    tiles for the rest; the parts API writes what the whole call writes."""
    dev = _dev()
    import ctypes
    from mcdseg import ops
    cin, cout, k, stride, d, n, h, w, math = case
    monkeypatch.setattr(ops, "CONV_MATH", math)
    libopt(PP_MIN_ROUNDS=1)  # (by default the kernels take a convolution from two whole rounds of tiles on)
    libopt(PP_WIDE_FILL=101)  # (... and the 256 x 320 tile takes it whole where that is cheaper: below, forced)
    x, wt, _, s, pad, d = _conv_inputs((cin, cout, k, stride, d, h, w, n, False), 41)
    desc = ops.conv_desc(x.shape, wt.shape, stride, pad, d)
    pk = ops.PackedWeights()
    wf, wd, mpf = pk.get(wt.to(dev), desc)
    xg = x.to(dev)
    gy = torch.randn(n, cout, desc.Ho, desc.Wo, generator=torch.Generator().manual_seed(42))
    gyg = gy.to(dev)
    x_cb, x_bound = ops.split_companion(xg)
    gy_cb, gy_bound = ops.split_companion(gyg)
    L, mid = ops.lib(), ops.MATH_ID[math]
    cus = torch.cuda.get_device_properties(dev).multi_processor_count
    outs = {}
    for tag in ("pp", "off"):
        libopt(PINGPONG=int("3" if tag == "pp" else "0"))
        names = []

        class _Names:
            def wants(self, name):
                names.append(name)
                return False
        prev, ops.LAUNCH_TIMER = ops.LAUNCH_TIMER, _Names()
        try:
            y, part, rows = ops._conv_fprop(desc, xg, wf, None, True, mpf, x_cb, x_bound, pk.w_bound)
            dx = ops._conv_dgrad(desc, None, wd, gy_cb, gy_bound, pk.w_bound) if stride == 1 else None
        finally:
            ops.LAUNCH_TIMER = prev
        outs[tag] = (y, dx, part, rows, names)
        for dgrad, m, pix in ((0, cout, n * desc.Ho * desc.Wo), (1, cin, n * h * w)):
            pp = L.mcdseg_conv_split_parts(ctypes.byref(desc), mid, 1, dgrad)
            tiles = (pix // 256) * (m // 256) if m % 256 == 0 and (dgrad == 0 or stride == 1) else 0
            want = (tiles // cus) * cus // max(1, m // 256) * 256 if tag == "pp" else 0
            assert pp == min(want, pix // 256 * 256), (tag, dgrad, pp, want)
    y, dx, part, rows, names = outs["pp"]
    dp = math == "f16x1"  # (the one-term arithmetic: two K-steps per barrier interval where their number is even -- every case here)
    assert any(nm in (ops.pingpong_kernel_name(False, deep=dp), ops.pingpong_kernel_name(True, deep=dp)) for nm in names), names
    assert any(nm in (ops.pingpong_kernel_name(False, small=True), ops.pingpong_kernel_name(True, small=True)) for nm in names), names
    assert not any("conv_gemm_split_pp_kernel" in nm for nm in outs["off"][4]), outs["off"][4]
    assert torch.equal(y, outs["off"][0]), "forward differs from the 4-wave tiles (max %.3e)" % float((y - outs["off"][0]).abs().max())
    assert rows == outs["off"][3] and torch.equal(part, outs["off"][2]), "fused BatchNorm partial rows differ"
    if dx is not None:
        assert torch.equal(dx, outs["off"][1]), "data gradient differs from the 4-wave tiles"
    if math == "f16x3":
        x64, w64 = x.double().requires_grad_(), wt.double()
        ref = F.conv2d(x64, w64, None, stride, pad, d)
        _assert_close(y, ref, 2e-5, "forward (ping-pong + rest)")
        if dx is not None:
            (gx_ref,) = torch.autograd.grad(ref, [x64], gy.double())
            _assert_close(dx, gx_ref, 2e-5, "data gradient (ping-pong + rest)")
    if dx is not None:  # data gradient + addend on the two-launch plan (256 x 256 + 256 x 128 tiles, ragged last tile)
        addend = torch.randn(dx.shape, generator=torch.Generator().manual_seed(46)).to(dev)
        dx_add = ops._conv_dgrad(desc, None, wd, gy_cb, gy_bound, pk.w_bound, addend=addend)
        assert torch.equal(dx_add, dx + addend), "dgrad + addend differs from the separate add (max %.3e)" % float((dx_add - dx - addend).abs().max())
    # the 256 x 128 ping-pong tile alone (mode 2) and the 256 x 256 tile with the 4-wave tiles for the rest (mode 1): the same bits
    for mode in ("2", "1"):
        libopt(PINGPONG=int(mode))
        y_m, part_m, rows_m = ops._conv_fprop(desc, xg, wf, None, True, mpf, x_cb, x_bound, pk.w_bound)
        assert torch.equal(y_m, y) and rows_m == rows and torch.equal(part_m, part), "mode %s differs" % mode
        if dx is not None:
            assert torch.equal(ops._conv_dgrad(desc, None, wd, gy_cb, gy_bound, pk.w_bound), dx), "data gradient, mode %s" % mode
    # the 256 x 320 tile (mode 4 forces it; by default it takes a convolution with fewer than two rounds of 256 x 256 tiles whose 320-pixel
    # tiles fill their rounds): the same output bits; its BatchNorm partial rows are one per 160 pixels -- the same moments, regrouped
    libopt(PINGPONG=4)
    names = []

    class _Names4:
        def wants(self, name):
            names.append(name)
            return False
    prev, ops.LAUNCH_TIMER = ops.LAUNCH_TIMER, _Names4()
    try:
        y_w, part_w, rows_w = ops._conv_fprop(desc, xg, wf, None, True, mpf, x_cb, x_bound, pk.w_bound)
        dx_w = ops._conv_dgrad(desc, None, wd, gy_cb, gy_bound, pk.w_bound) if dx is not None else None
    finally:
        ops.LAUNCH_TIMER = prev
    if cout % 256 == 0:
        assert ops.pingpong_kernel_name(False, wide=True, deep=dp) in names, names
    if dx is not None and cin % 256 == 0:  # (the data gradient has M = Cin output rows)
        assert L.mcdseg_conv_split_wide_pingpong(ctypes.byref(desc), mid, 1, 1) == 1 and ops.pingpong_kernel_name(True, wide=True, deep=dp) in names, names
    pixels = n * desc.Ho * desc.Wo
    if cout % 256 == 0:
        assert L.mcdseg_conv_split_wide_pingpong(ctypes.byref(desc), mid, 1, 0) == 1
        assert L.mcdseg_conv_split_parts(ctypes.byref(desc), mid, 1, 0) == pixels and rows_w == 2 * ((pixels + 319) // 320)
    else:
        assert L.mcdseg_conv_split_wide_pingpong(ctypes.byref(desc), mid, 1, 0) == 0 and rows_w == rows
    assert torch.equal(y_w, y), "forward on the 256 x 320 tile differs (max %.3e)" % float((y_w - y).abs().max())
    if dx is not None:
        assert torch.equal(dx_w, dx), "data gradient on the 256 x 320 tile differs"
    pw = part_w.double().view(rows_w, 3, -1)[:, :, :cout]
    cnt, mean_r, m2_r = pw[:, 0], pw[:, 1], pw[:, 2]
    assert float(cnt[:, 0].sum()) == pixels
    mean = (cnt * mean_r).sum(0) / pixels
    var = (m2_r + cnt * (mean_r - mean) ** 2).sum(0) / pixels
    y64 = y.double().transpose(0, 1).reshape(cout, -1)
    assert float((mean - y64.mean(1)).abs().max()) <= 1e-6 * float(y64.abs().max())
    assert float((var - y64.var(1, unbiased=False)).abs().max()) <= 1e-6 * float(y64.var(1, unbiased=False).max())
    # parts 1 + 2 of the C ABI write exactly what part 0 writes (poisoned output, two calls)
    libopt(PINGPONG=3)
    y2 = torch.full_like(y, float("nan"))
    part2 = torch.full_like(part, float("nan"))
    for prt in (2, 1):
        ops.check(L.mcdseg_conv_split_fprop_part(ctypes.byref(desc), mid, None, ops._p(x_cb), ops._p(x_bound), ops._p(wf), ops._p(pk.w_bound), None,
                                                 ops._p(y2), ops._p(part2), prt, ops._stream()), "conv_split_fprop_part")
    assert torch.equal(y2, y) and torch.equal(part2, part)


@pytest.mark.parametrize("case", [(128, 128, 3, 1, 2, 45, 67), (128, 256, 3, 2, 2, 33, 40), (256, 128, 3, 2, 1, 47, 30), (136, 200, 1, 1, 3, 29, 31)],
                         ids=lambda c: "x".join(map(str, c)))
def test_conv_wide_tile_kernels(case, monkeypatch, libopt):
    """The 128 x 256 tile (``conv_gemm_split_kernel<P, 4, 2, 1, 4, ...>``: the 128- and 256-channel layers at BASELINE batch sizes, all-DMA
    K loop with two k-halves per thread) forced onto small problems with MCDSEG_WIDETILE_MIN_SLOTS: forward (+ fused BatchNorm partial
    rows) and data gradient against fp64 and bit for bit against the 128 x 128 tile (same K order, same 64-pixel statistic rows)."""
    dev = _dev()
    from mcdseg import ops
    if ops.CONV_MATH == "f32":
        pytest.skip("split-precision kernels only")
    cin, cout, k, d, n, h, w = case
    x, wt, _, s, pad, d = _conv_inputs((cin, cout, k, 1, d, h, w, n, False), 33)
    desc = ops.conv_desc(x.shape, wt.shape, 1, pad, d)
    pk = ops.PackedWeights()
    wf, wd, mpf = pk.get(wt.to(dev), desc)
    xg = x.to(dev)
    gy = torch.randn(n, cout, desc.Ho, desc.Wo, generator=torch.Generator().manual_seed(34))
    gyg = gy.to(dev)
    x64, w64 = x.double().requires_grad_(), wt.double().requires_grad_()
    ref = F.conv2d(x64, w64, None, 1, pad, d)
    (gx_ref,) = torch.autograd.grad(ref, [x64], gy.double())
    x_cb, x_bound = ops.split_companion(xg)
    gy_cb, gy_bound = ops.split_companion(gyg)
    outs = {}
    for tag, slots in (("wide", "1"), ("square", "1000000000")):
        libopt(WIDETILE_MIN_SLOTS=int(slots))
        names = []

        class _Names:
            def wants(self, name):
                names.append(name)
                return False
        prev, ops.LAUNCH_TIMER = ops.LAUNCH_TIMER, _Names()
        try:
            y, part, rows = ops._conv_fprop(desc, xg, wf, None, True, mpf, x_cb, x_bound, pk.w_bound)
            dx = ops._conv_dgrad(desc, None, wd, gy_cb, gy_bound, pk.w_bound)
        finally:
            ops.LAUNCH_TIMER = prev
        want = "4, 2, 1, 4" if tag == "wide" else "2, 2, 2, 2"
        assert len(names) == 2 and all(want in nm for nm in names), names
        outs[tag] = (y, dx, part.view(rows, 3, mpf), rows)
    _assert_close(outs["wide"][0], ref, 2e-5, "fprop (128 x 256 tile)")
    _assert_close(outs["wide"][1], gx_ref, 2e-5, "dgrad (128 x 256 tile)")
    assert torch.equal(outs["wide"][0], outs["square"][0]) and torch.equal(outs["wide"][1], outs["square"][1])
    rw, rs = outs["wide"][3], outs["square"][3]
    assert rw >= rs and torch.equal(outs["wide"][2][:rs, :, :cout], outs["square"][2][:, :, :cout])
    assert float(outs["wide"][2][rs:, 0, :cout].abs().max()) == 0.0 if rw > rs else True  # rows past the last pixel count nothing


@pytest.mark.parametrize("cout,stride,h,w,n", [(16, 1, 16, 64, 1), (16, 1, 21, 45, 2), (32, 2, 23, 70, 2), (32, 1, 9, 33, 1),
                                               (16, 2, 12, 40, 2)])
def test_thin_layer_window_kernels(cout, stride, h, w, n, monkeypatch):
    """The LDS-window kernels of the thin 3x3 layers (16 contraction channels; conv_thin_window.hip): forward with fused
    BatchNorm partial statistics, and the stride-1 data gradient, from pre-split companions -- against fp64, against the
    implicit-GEMM kernel of the same arithmetic, and the statistics against the output's moments; ragged tiles included."""
    dev = _dev()
    import ctypes
    from mcdseg import ops
    monkeypatch.setattr(ops, "CONV_MATH", "f16x3")
    x, wt, _, s, pad, d = _conv_inputs((16, cout, 3, stride, 1, h, w, n, False), 41)
    desc = ops.conv_desc(x.shape, wt.shape, s, pad, d)
    L = ops.lib()
    assert L.mcdseg_conv_split_window_ok(ctypes.byref(desc), ops.MATH_ID["f16x3"], 1, 0) == (1 if stride == 1 else 0)
    x64, w64 = x.double().requires_grad_(), wt.double().requires_grad_()
    ref = F.conv2d(x64, w64, None, stride=s, padding=pad, dilation=d)
    gy = torch.randn(ref.shape, generator=torch.Generator().manual_seed(42))
    (gx_ref,) = torch.autograd.grad(ref, [x64], gy.double())
    xg, gyg = x.to(dev), gy.to(dev)
    pk = ops.PackedWeights()
    wf, wd, mpf = pk.get(wt.to(dev), desc)
    x_cb, x_bound = ops.split_companion(xg)
    y_w, part, rows = ops._conv_fprop(desc, xg, wf, None, True, mpf, x_cb, x_bound, pk.w_bound)
    y_g, _, _ = ops._conv_fprop(desc, xg, wf, None, True, mpf, None, x_bound, pk.w_bound)
    _assert_close(y_w, ref, 2e-5, "window forward")
    assert float((y_w - y_g).abs().max()) <= 4e-6 * float(ref.detach().abs().max()), "window kernel vs implicit GEMM"
    mean = torch.empty(cout, device=dev)
    rstd = torch.empty(cout, device=dev)
    ws = torch.empty(L.mcdseg_bn_stats_workspace_bytes(rows, cout) // 8 + 1, dtype=torch.float64, device=dev)
    ops.check(L.mcdseg_bn_stats_finalize(ops._p(part), rows, cout, mpf, ops._p(mean), ops._p(rstd), None, None, None, 0.1, 1e-5,
                                         None, None, None, None, 1, ops._p(ws), ctypes.c_size_t(ws.numel() * 8), ops._stream()),
              "bn_stats_finalize")
    r = ref.detach()
    assert float((mean.double().cpu() - r.mean((0, 2, 3))).abs().max()) <= 1e-5 * float(r.std())
    _assert_close(rstd, (r.var((0, 2, 3), unbiased=False) + 1e-5).rsqrt(), 1e-5, "fused BN rstd (window kernel)")
    gy_cb, gy_bound = ops.split_companion(gyg)
    took = L.mcdseg_conv_split_window_ok(ctypes.byref(desc), ops.MATH_ID["f16x3"], 1, 1)
    assert took == (1 if (stride == 1 and cout == 16) else 0)
    dx_w = ops._conv_dgrad(desc, None, wd, gy_cb, gy_bound, pk.w_bound)
    _assert_close(dx_w, gx_ref, 2e-5, "dgrad from the companion (window kernel where it applies)")


@pytest.mark.parametrize("cin,h,w,n", [(6, 37, 70, 2), (3, 16, 33, 1), (6, 64, 96, 3)])
def test_stem_forward_on_the_window_kernel(cin, h, w, n, monkeypatch):
    """The 7x7 stem (3 / 6 -> 16 channels) on the LDS-window kernel from the zero-padded 8-channel companion of the network
    input: against fp64, against the direct bf16x6 stem kernel, statistics against the output's moments, and the fused group
    (ops.conv_bn_act) takes it -- same outputs and gradients as with the direct kernel within the arithmetic's tolerance."""
    dev = _dev()
    import ctypes
    from mcdseg import ops
    monkeypatch.setattr(ops, "CONV_MATH", "f16x3")
    x, wt, _, s, pad, d = _conv_inputs((cin, 16, 7, 1, 3, h, w, n, False), 43)
    desc = ops.conv_desc(x.shape, wt.shape, s, pad, d)
    L = ops.lib()
    assert L.mcdseg_conv_split_window_ok(ctypes.byref(desc), ops.MATH_ID["f16x3"], 1, 0) == 1
    assert L.mcdseg_conv_split_window_ok(ctypes.byref(desc), ops.MATH_ID["f16x3"], 0, 0) == 0
    ref = F.conv2d(x.double(), wt.double(), None, stride=s, padding=pad, dilation=d)
    xg = x.to(dev)
    pk = ops.PackedWeights()
    wf, wd, mpf = pk.get(wt.to(dev), desc)
    x_cb, x_bound = ops.split_companion_padded(xg)
    y_w, part, rows = ops._conv_fprop(desc, xg, wf, None, True, mpf, x_cb, x_bound, pk.w_bound)
    y_d, part_d, rows_d = ops._conv_fprop(desc, xg, wf, None, True, mpf, None, None, pk.w_bound)
    _assert_close(y_w, ref, 2e-5, "stem window forward")
    _assert_close(y_d, ref, 2e-5, "stem direct forward (image behind the standard one)")
    mean = torch.empty(16, device=dev)
    rstd = torch.empty(16, device=dev)
    ws = torch.empty(L.mcdseg_bn_stats_workspace_bytes(rows, 16) // 8 + 1, dtype=torch.float64, device=dev)
    ops.check(L.mcdseg_bn_stats_finalize(ops._p(part), rows, 16, mpf, ops._p(mean), ops._p(rstd), None, None, None, 0.1, 1e-5,
                                         None, None, None, None, 1, ops._p(ws), ctypes.c_size_t(ws.numel() * 8), ops._stream()),
              "bn_stats_finalize")
    assert float((mean.double().cpu() - ref.mean((0, 2, 3))).abs().max()) <= 1e-5 * float(ref.std())
    _assert_close(rstd, (ref.var((0, 2, 3), unbiased=False) + 1e-5).rsqrt(), 1e-5, "fused BN rstd (stem window kernel)")

    # the fused group: window kernel by default, direct kernel with the knob off
    conv = torch.nn.Conv2d(cin, 16, 7, 1, 3, bias=False).to(dev)
    with torch.no_grad():
        conv.weight.copy_(wt)
    bn = torch.nn.BatchNorm2d(16).to(dev)
    conv._packed = ops.PackedWeights()
    gy = torch.randn(ref.shape, generator=torch.Generator().manual_seed(44)).to(dev)
    outs = []
    for window in (True, False):
        monkeypatch.setattr(ops, "STEM_WINDOW", window)
        conv.weight.grad = None
        y = ops.conv_bn_act(xg.clone(), conv, bn, relu=True)
        y.backward(gy)
        outs.append((y.detach().clone(), conv.weight.grad.clone()))
    _assert_close(outs[0][0], outs[1][0], 2e-5, "fused group output, window vs direct stem")
    _assert_close(outs[0][1], outs[1][1], 1e-4, "stem weight gradient, window vs direct stem forward")


def test_group_weight_pack_equals_the_per_layer_pack(monkeypatch):
    """After an optimizer step every stale weight image of the device is rebuilt by two table-driven launches
    (mcdseg_conv_split_pack_weights_multi); images and bound scalars are bit-identical to the per-layer pack, including the
    stem (its direct kernel's image sits behind the standard one), a layer with a ragged channel count and a 1x1 layer."""
    dev = _dev()
    from mcdseg import ops
    monkeypatch.setattr(ops, "CONV_MATH", "f16x3")
    shapes = [((16, 6, 7, 7), 1, 3, 1), ((16, 16, 3, 3), 1, 1, 1), ((64, 32, 3, 3), 2, 1, 1), ((41, 512, 1, 1), 1, 0, 1),
              ((256, 128, 3, 3), 1, 2, 2), ((24, 40, 3, 3), 1, 1, 1)]
    g = torch.Generator().manual_seed(21)
    ws = [(torch.randn(sh, generator=g) * 0.1).to(dev) for sh, _, _, _ in shapes]
    descs = [ops.conv_desc((1, sh[1], 16, 16), sh, st, pad, dil) for sh, st, pad, dil in shapes]
    pks = [ops.PackedWeights() for _ in shapes]
    for pk, w, d in zip(pks, ws, descs):
        pk.get(w, d)  # first use: per-layer pack (registers the layer for group packs)
    for w in ws:
        w.mul_(1.5).add_(0.01)  # an "optimizer step": every image is stale now
    pks[0].get(ws[0], descs[0])  # the first stale get re-packs all of them
    grouped = [(pk.wf.clone(), pk.wd.clone(), pk.w_bound.clone()) for pk in pks]
    assert all(pk.key == ops.PackedWeights._key_of(w) for pk, w in zip(pks, ws)), "the group pack did not refresh every layer"
    monkeypatch.setattr(ops, "GROUP_PACK", False)
    for (wf, wd, wb), w, d in zip(grouped, ws, descs):
        ref = ops.PackedWeights()
        ref.get(w, d)
        assert torch.equal(ref.wf, wf) and torch.equal(ref.wd, wd) and torch.equal(ref.w_bound, wb), tuple(w.shape)


@pytest.mark.parametrize("n,off", [(1, 0), (3, 0), (5, 1), (1027, 0), (4096 + 3, 3), (3 * 1000 * 1000 + 1, 0), (1 << 20, 2)])
def test_absmax_scalar_is_the_exact_maximum(n, off):
    """mcdseg_absmax (the bound scalar every fp16 scale is derived from): exactly max |x| for aligned and unaligned starts,
    vector body and scalar tail, with the maximum planted in the first / last element; NaN propagates as a non-finite bound."""
    dev = _dev()
    from mcdseg import ops
    g = torch.Generator().manual_seed(n)
    base = torch.randn(n + off, generator=g).to(dev)
    for where in (0, n - 1, n // 2):
        x = base.clone()[off:]
        x[where] = -37.5
        got = ops.absmax(x)
        assert float(got) == 37.5 and float(got) == float(x.abs().max())
    x = base.clone()[off:]
    assert float(ops.absmax(x)) == float(x.abs().max())
    x[n - 1] = float("nan")
    assert not np.isfinite(float(ops.absmax(x)))


def test_conv_nonfinite_operands():
    """Contract of the split-precision convolutions for non-finite data: an output that a NaN / inf operand reaches is
    non-finite (an fp32 FMA chain would give +-inf where the split gives NaN: inf - inf in the remainder), every other output
    is unaffected; denormal operands are harmless."""
    dev = _dev()
    from mcdseg import ops
    x, wt, _, s, pad, d = _conv_inputs((64, 128, 3, 1, 1, 12, 16, 1, False), 37)
    desc = ops.conv_desc(x.shape, wt.shape, 1, pad, d)
    pk = ops.PackedWeights()
    wf, wd, mpf = pk.get(wt.to(dev), desc)
    x[0, 9, 6, 1] = 0.0  # the position that will hold the denormal
    clean, _, _ = ops._conv_fprop(desc, x.to(dev), wf, None, False, mpf, w_bound=pk.w_bound)
    xb = x.clone()
    xb[0, 5, 2, 3] = float("inf")
    xb[0, 7, 9, 12] = float("nan")
    xb[0, 9, 6, 1] = 1e-41  # denormal
    for cb in (False, True):
        xg = xb.to(dev)
        x_cb, x_bound = ops.split_companion(xg) if cb else (None, None)
        y, _, _ = ops._conv_fprop(desc, xg, wf, None, False, mpf, x_cb, x_bound, pk.w_bound)
        bad = ~torch.isfinite(y)
        touched = torch.zeros(1, 1, 12, 16, dtype=torch.bool)
        touched[0, 0, 1:4, 2:5] = True
        touched[0, 0, 8:11, 11:14] = True
        assert bool((bad.cpu() == touched.expand_as(bad)).all()), "non-finite outputs are not exactly the reached ones"
        ok = ~touched.expand_as(bad)
        assert float((y.cpu()[ok] - clean.cpu()[ok]).abs().max()) <= 2e-5 * float(clean.abs().max())


@pytest.mark.parametrize("cin,cout,hw,n", [(64, 128, (13, 19), 2), (16, 16, (21, 45), 2), (32, 41, (9, 11), 3), (6, 16, (20, 28), 2)])
def test_bn_backward_mask_from_z_is_bitwise(cin, cout, hw, n, monkeypatch):
    """BatchNorm backward of a ReLU group without residual recomputes the mask from z (y > 0 <=> fma(z, a, b) > 0 with the
    forward kernels' own a, b) instead of reading y: every gradient is bit-identical to the y-reading form -- negative and
    zero gammas, exact zeros of the pre-activation included."""
    dev = _dev()
    from mcdseg import ops
    from models.drn import BatchNorm2d, Conv2d
    g = torch.Generator().manual_seed(9)
    k = 7 if cin <= 8 else 3
    x = torch.randn(n, cin, *hw, generator=g).to(dev)
    gy = torch.randn(n, cout, *hw, generator=g).to(dev)

    def run(zmask):
        monkeypatch.setattr(ops, "BN_ZMASK", zmask)
        torch.manual_seed(6)
        conv, bn = Conv2d(cin, cout, k, padding=k // 2, bias=False).to(dev), BatchNorm2d(cout).to(dev)
        with torch.no_grad():
            bn.weight.uniform_(-1.0, 1.5)
            bn.weight[0] = 0.0
            bn.bias.uniform_(-0.5, 0.5)
            bn.bias[0] = 0.0  # gamma = beta = 0: the pre-activation is exactly 0 everywhere in this channel
        bn.train()
        xin = x.clone().requires_grad_(cin > 8)
        y = ops.conv_bn_act(xin, conv, bn)
        y.backward(gy)
        return [xin.grad, conv.weight.grad, bn.weight.grad, bn.bias.grad]

    for a, b in zip(run(True), run(False)):
        assert (a is None and b is None) or torch.equal(a, b)


@pytest.mark.parametrize("block,planes,stride,dil,hw,n", [("basic", 64, 1, 1, (12, 20), 2), ("basic", 128, 2, 1, (28, 40), 2),
                                                          ("basic", 256, 1, 2, (60, 80), 3), ("bottleneck", 64, 1, 1, (10, 14), 2),
                                                          ("basic", 64, 1, 1, (120, 160), 3)])
def test_relu_bit_plane_is_bitwise_the_fp32_mask(block, planes, stride, dil, hw, n, monkeypatch):
    """BatchNorm backward of a ReLU group WITH residual (the block's last group) takes its mask from the bit-plane the forward apply
    kernel wrote (round 5: ``mcdseg_bn_apply_cb_mask`` / ``_bn_bwd_reduce_mask`` / ``_bn_bwd_apply_cb_mask``, 1 bit per element)
    instead of from the fp32 y: block output and every gradient are bit-identical to the y-reading form (MCDSEG_RELU_MASK=0) -- ragged
    256-pixel blocks, several 256-pixel blocks per reduce chunk and channel counts from 64 to 256 included -- and the mask entry
    points are the ones that ran."""
    dev = _dev()
    from mcdseg import ops
    from models import drn
    g = torch.Generator().manual_seed(15)
    inpl = planes * (4 if block == "bottleneck" else 1) if stride == 1 else planes // 2
    x = torch.randn(n, inpl, *hw, generator=g).to(dev)
    assert hw[0] * hw[1] % 4 == 0

    def run(on):
        monkeypatch.setattr(ops, "RELU_MASK", on)
        torch.manual_seed(4)
        cls = drn.BasicBlock if block == "basic" else drn.Bottleneck
        outpl = planes * cls.expansion
        down = None
        if stride != 1 or inpl != outpl:
            down = drn.ConvBN(drn.Conv2d(inpl, outpl, kernel_size=1, stride=stride, bias=False), drn.BatchNorm2d(outpl))
        blk = cls(inpl, planes, stride, down, dilation=(dil, dil)).to(dev).train()
        pre_c, pre_b = drn.Conv2d(inpl, inpl, 3, padding=1, bias=False).to(dev), drn.BatchNorm2d(inpl).to(dev).train()
        xin = x.clone().requires_grad_()
        h0 = ops.conv_bn_act(xin, pre_c, pre_b)
        calls = []
        L = ops.lib()
        real = {k: getattr(L, k) for k in ("mcdseg_bn_apply_cb_mask", "mcdseg_bn_bwd_reduce_mask", "mcdseg_bn_bwd_apply_cb_mask")}

        class Spy:  # (ctypes function objects cannot be patched in place: wrap the handle ops.lib() returns)
            def __getattr__(self, name):
                fn = getattr(L, name)
                if name in real:
                    def wrapped(*a, **kw):
                        calls.append(name)
                        return fn(*a, **kw)
                    return wrapped
                return fn
        monkeypatch.setattr(ops, "lib", lambda: Spy())
        out = blk(h0)
        gy = torch.randn(out.shape, generator=torch.Generator().manual_seed(16)).to(dev)
        out.backward(gy)
        monkeypatch.setattr(ops, "lib", lambda: L)
        grads = [xin.grad] + [p.grad for p in blk.parameters()] + [pre_c.weight.grad, pre_b.weight.grad]
        return out.detach(), grads, calls

    o1, g1, c1 = run(True)
    o0, g0, c0 = run(False)
    assert sorted(set(c1)) == ["mcdseg_bn_apply_cb_mask", "mcdseg_bn_bwd_apply_cb_mask", "mcdseg_bn_bwd_reduce_mask"] and c0 == [], (c1, c0)
    assert torch.equal(o1, o0)
    for a, b in zip(g1, g0):
        assert torch.equal(a, b)


@pytest.mark.parametrize("block,planes,stride,dil,hw", [("basic", 64, 1, 1, (12, 20)), ("basic", 128, 2, 1, (13, 19)),
                                                        ("basic", 256, 1, 2, (9, 12)), ("bottleneck", 64, 1, 1, (10, 14))])
def test_internal_groups_skip_the_fp32_output_bitwise(block, planes, stride, dil, hw, monkeypatch):
    """conv1 of a BasicBlock (conv1 / conv2 of a Bottleneck) writes only its companion: the block's output and every gradient
    are bit-identical to the form that also writes the fp32 activation (MCDSEG_INTERNAL_SKIP_Y=0), and the intermediate tensor
    handed on is the 4-byte stand-in."""
    dev = _dev()
    from mcdseg import ops
    from models import drn
    g = torch.Generator().manual_seed(13)
    inpl = planes * (4 if block == "bottleneck" else 1) if stride == 1 else planes // 2
    x = torch.randn(2, inpl, *hw, generator=g).to(dev)

    def run(skip):
        monkeypatch.setattr(ops, "INTERNAL_SKIP_Y", skip)
        torch.manual_seed(4)
        cls = drn.BasicBlock if block == "basic" else drn.Bottleneck
        outpl = planes * cls.expansion
        down = None
        if stride != 1 or inpl != outpl:
            down = drn.ConvBN(drn.Conv2d(inpl, outpl, kernel_size=1, stride=stride, bias=False), drn.BatchNorm2d(outpl))
        blk = cls(inpl, planes, stride, down, dilation=(dil, dil)).to(dev).train()
        pre_c, pre_b = drn.Conv2d(inpl, inpl, 3, padding=1, bias=False).to(dev), drn.BatchNorm2d(inpl).to(dev).train()
        xin = x.clone().requires_grad_()
        h0 = ops.conv_bn_act(xin, pre_c, pre_b)  # gives the block an input with a companion, as inside the network
        seen = []
        orig = ops.conv_bn_act

        def spy(*a, **kw):
            y = orig(*a, **kw)
            seen.append(ops.is_virtual(y))
            return y

        monkeypatch.setattr(ops, "conv_bn_act", spy)
        out = blk(h0)
        monkeypatch.setattr(ops, "conv_bn_act", orig)
        gy = torch.randn(out.shape, generator=torch.Generator().manual_seed(14)).to(dev)
        out.backward(gy)
        grads = [xin.grad] + [p.grad for p in blk.parameters()] + [pre_c.weight.grad, pre_b.weight.grad]
        return out.detach(), grads, seen

    o1, g1, seen1 = run(True)
    o0, g0, seen0 = run(False)
    assert any(seen1) and not any(seen0), (seen1, seen0)
    assert torch.equal(o1, o0)
    for a, b in zip(g1, g0):
        assert torch.equal(a, b)


# (Cin, Cout, k, stride, dil, H, W, N): 128x128-plan layers only (min(C) > 64, C % 8 == 0)
WGRAD_CB_CASES = [
    (128, 128, 3, 1, 1, 12, 16, 2),   # whole 8x4 tiles
    (136, 200, 3, 1, 2, 13, 19, 2),   # ragged tiles, channel counts that are not tile multiples
    (256, 128, 3, 1, 4, 9, 10, 1),    # dilation 4: most taps of the border tiles fall into the padding
    (128, 256, 1, 1, 1, 11, 13, 2),   # 1x1
    (128, 128, 3, 2, 1, 15, 17, 2),   # stride 2
    (72, 80, 3, 1, 1, 8, 8, 3),
    (512, 512, 3, 1, 4, 30, 40, 2),   # layer6 at 240x320: several workgroups per image along the pixels
    (64, 64, 3, 1, 1, 12, 16, 2),     # 64-channel layers: tap pairs share one staged dY tile (f16x3; the f32 plan otherwise)
    (64, 128, 3, 2, 1, 15, 17, 2),    # ... stride 2, two co tiles
    (48, 64, 3, 1, 2, 13, 19, 2),     # ... ragged channel groups, dilation 2
    (64, 128, 1, 2, 1, 11, 13, 2),    # ... 1x1 stride 2: a single tap (the pair's second half is empty)
    (32, 64, 3, 2, 1, 23, 31, 2),     # ... base.3.0 conv1: 32 input channels on the 64-wide tile (half of it padding; round 5)
    (32, 64, 1, 2, 1, 23, 31, 2),     # ... base.3.0 downsample
    (24, 40, 3, 1, 1, 9, 14, 1),      # ... ragged on both sides
    (16, 16, 3, 1, 1, 16, 64, 1),     # thin layers (layer1): the window kernel, whole 8 x 32 tiles
    (16, 16, 3, 1, 1, 21, 45, 2),     # ... ragged rows and columns
    (16, 32, 3, 2, 1, 23, 70, 2),     # ... layer2: stride 2, two row tiles of output channels
]


@pytest.mark.parametrize("math", ["f16x3", "bf16x6"])
@pytest.mark.parametrize("case", WGRAD_CB_CASES, ids=lambda c: "x".join(map(str, c[:7])))
def test_conv_wgrad_presplit_operands(case, math, monkeypatch):
    """weight gradient from the pre-split companions (8x8 register transposes) = the in-loop split kernel = fp64"""
    dev = _dev()
    from mcdseg import ops
    monkeypatch.setattr(ops, "CONV_MATH", math)
    cin, cout, k, s, d, h, w, n = case
    x, wt, _, s, pad, d = _conv_inputs((cin, cout, k, s, d, h, w, n, False), 23)
    desc = ops.conv_desc(x.shape, wt.shape, s, pad, d)
    x64, w64 = x.double(), wt.double().requires_grad_()
    ref = F.conv2d(x64, w64, None, stride=s, padding=pad, dilation=d)
    gy = torch.randn(ref.shape, generator=torch.Generator().manual_seed(24))
    (gw_ref,) = torch.autograd.grad(ref, [w64], gy.double())
    xg, gyg = x.to(dev), gy.to(dev)
    x_cb, x_bound = ops.split_companion(xg)
    gy_cb, gy_bound = ops.split_companion(gyg)
    assert x_cb is not None and gy_cb is not None
    if cin <= 16 and math == "f16x3":  # the thin-layer window kernel is what the companions select here
        import ctypes
        assert ops.lib().mcdseg_conv_wgrad_variant(ctypes.byref(desc), ops.MATH_ID[math], 1) == 15
    if 16 < min(cin, cout) <= 64 and math == "f16x3":  # the tap-pair kernel of the 64-wide plan, from 24 channels up
        import ctypes
        assert ops.lib().mcdseg_conv_wgrad_variant(ctypes.byref(desc), ops.MATH_ID[math], 1) == 14
    dw_loop = ops._conv_wgrad(desc, xg, gyg, None, None, x_bound, gy_bound)
    dw_cb = ops._conv_wgrad(desc, xg, gyg, x_cb, gy_cb, x_bound, gy_bound)
    _assert_close(dw_loop, gw_ref, 2e-5, "wgrad (in-loop split)")
    _assert_close(dw_cb, gw_ref, 2e-5, "wgrad (pre-split operands)")
    e_loop, e_cb = _maxerr(dw_loop, gw_ref)[0], _maxerr(dw_cb, gw_ref)[0]
    assert e_cb <= max(2.0 * e_loop, 2e-6 * _maxerr(dw_loop, gw_ref)[1])


@pytest.mark.parametrize("case", [(256, 512, 3, 1, 4, 30, 40, 6), (512, 512, 3, 1, 2, 24, 32, 6), (264, 512, 3, 1, 4, 29, 37, 6)],
                         ids=lambda c: "x".join(map(str, c)))
def test_conv_wgrad_large_tile_kernel(case, monkeypatch, libopt):
    """``conv_wgrad_split_tr_kernel<SplitF16x3, 4, 2, 3>`` -- the 256 x 128 weight-gradient tile, 14 % of the benchmark step -- on
    problems small enough for an fp64 reference but with enough split-K work that the launcher takes it (it was
    only reached at the benchmark's size): named through ``mcdseg_conv_wgrad_variant == 13``, <= 2e-5 of the scale against fp64, and
    bit for bit the 128 x 128 kernel's result (same stage tiles in the same order, same MFMA K order; MCDSEG_WGRAD_BIG=0 selects it)."""
    dev = _dev()
    import ctypes
    from mcdseg import ops
    monkeypatch.setattr(ops, "CONV_MATH", "f16x3")
    libopt(WGRAD_BIG=1)  # (the default)
    libopt(WGRAD_PP=0)  # (the ping-pong / stream-K kernel takes these layers when the problem is large enough: next test)
    cin, cout, k, s, d, h, w, n = case
    x, wt, _, s, pad, d = _conv_inputs((cin, cout, k, s, d, h, w, n, False), 41)
    desc = ops.conv_desc(x.shape, wt.shape, s, pad, d)
    assert ops.lib().mcdseg_conv_wgrad_variant(ctypes.byref(desc), ops.MATH_ID["f16x3"], 1) == 13
    x64, w64 = x.double(), wt.double().requires_grad_()
    ref = F.conv2d(x64, w64, None, stride=s, padding=pad, dilation=d)
    gy = torch.randn(ref.shape, generator=torch.Generator().manual_seed(42))
    (gw_ref,) = torch.autograd.grad(ref, [w64], gy.double())
    xg, gyg = x.to(dev), gy.to(dev)
    x_cb, x_bound = ops.split_companion(xg)
    gy_cb, gy_bound = ops.split_companion(gyg)
    dw_big = ops._conv_wgrad(desc, xg, gyg, x_cb, gy_cb, x_bound, gy_bound)
    _assert_close(dw_big, gw_ref, 2e-5, "wgrad (256 x 128 tiles)")
    libopt(WGRAD_BIG=0)
    assert ops.lib().mcdseg_conv_wgrad_variant(ctypes.byref(desc), ops.MATH_ID["f16x3"], 1) == 12
    dw_small = ops._conv_wgrad(desc, xg, gyg, x_cb, gy_cb, x_bound, gy_bound)
    _assert_close(dw_small, gw_ref, 2e-5, "wgrad (128 x 128 tiles)")
    assert torch.equal(dw_big, dw_small)


@pytest.mark.parametrize("math", ["f16x3", "f16x1"])
def test_conv_pingpong_wide_tile_by_default(math, monkeypatch, libopt):
    """What the launcher chooses BY DEFAULT at BASELINE config 2's pixel count (76800 = 240 tiles of 320 pixels on 256 CUs): the
    256 x 320 ping-pong tile for a 256-row problem, its 128 x 320 form for the 128-channel layers -- forward (+ partial rows of 160
    pixels) and data gradient bit for bit those of the 4-wave tiles (MCDSEG_PINGPONG=0), the statistics the same moments, fp64 parity
    of the default arithmetic."""
    dev = _dev()
    import ctypes
    from mcdseg import ops
    monkeypatch.setattr(ops, "CONV_MATH", math)
    libopt(PINGPONG=3)  # (the default)
    L, mid = ops.lib(), ops.MATH_ID[math]
    # (Cin, Cout, N, forward tile, data-gradient tile): 76800 pixels = 240 tiles of 320; 38400 pixels (half: BASELINE config 4's N = 8)
    # = 240 tiles of 160 for a 256-row problem, and nothing for a 128-row one (0: the 4-wave tiles keep it)
    for cin, cout, n, kind_f, kind_d in ((128, 256, 4, 1, 2), (128, 128, 4, 2, 2), (256, 256, 2, 3, 3), (128, 256, 2, 3, 0)):
        h, w, dil = 120, 160, 2
        x, wt, _, s, pad, d = _conv_inputs((cin, cout, 3, 1, dil, h, w, n, False), 43)
        desc = ops.conv_desc(x.shape, wt.shape, 1, pad, d)
        pk = ops.PackedWeights()
        wf, wd, mpf = pk.get(wt.to(dev), desc)
        xg = x.to(dev)
        gy = torch.randn(n, cout, desc.Ho, desc.Wo, generator=torch.Generator().manual_seed(44))
        x_cb, x_bound = ops.split_companion(xg)
        gy_cb, gy_bound = ops.split_companion(gy.to(dev))
        pixels = n * h * w
        assert L.mcdseg_conv_split_wide_pingpong(ctypes.byref(desc), mid, 1, 0) == kind_f
        assert L.mcdseg_conv_split_wide_pingpong(ctypes.byref(desc), mid, 1, 1) == kind_d
        names = []

        class _Names:
            def wants(self, name):
                names.append(name)
                return False
        prev, ops.LAUNCH_TIMER = ops.LAUNCH_TIMER, _Names()
        try:
            y, part, rows = ops._conv_fprop(desc, xg, wf, None, True, mpf, x_cb, x_bound, pk.w_bound)
            dx = ops._conv_dgrad(desc, None, wd, gy_cb, gy_bound, pk.w_bound)
        finally:
            ops.LAUNCH_TIMER = prev
        # (the one-term arithmetic runs these -- an even number of K-steps -- two K-steps per barrier interval: policy SplitF16x1D, the
        # same products in the same order, so the bitwise comparison with the 4-wave tiles below covers it)
        deep = [bool(L.mcdseg_conv_split_pp_deep(ctypes.byref(desc), mid, g)) for g in (0, 1)]
        assert deep == [math == "f16x1"] * 2
        assert names[0] == ops.pingpong_kernel_name(False, wide=kind_f, deep=deep[0]) and len(names) == 2, names
        assert names[1] == ops.pingpong_kernel_name(True, wide=kind_d, deep=deep[1]) if kind_d else "conv_gemm_split_pp_kernel" not in names[1], names
        assert ("SplitF16x1D" in names[0]) == (math == "f16x1")
        assert rows == (pixels // 160 if kind_f == 3 else 2 * (pixels // 320))
        libopt(PINGPONG=0)
        y0, part0, rows0 = ops._conv_fprop(desc, xg, wf, None, True, mpf, x_cb, x_bound, pk.w_bound)
        dx0 = ops._conv_dgrad(desc, None, wd, gy_cb, gy_bound, pk.w_bound)
        libopt(PINGPONG=3)  # (the default)
        assert torch.equal(y, y0) and torch.equal(dx, dx0), "the wide tile's results differ from the 4-wave tiles'"
        # data gradient + addend (a residual block's other gradient; the tile of the addend goes through LDS): the bits of dgrad-then-add
        addend = torch.randn(dx.shape, generator=torch.Generator().manual_seed(45)).to(dev)
        dx_add = ops._conv_dgrad(desc, None, wd, gy_cb, gy_bound, pk.w_bound, addend=addend)
        assert torch.equal(dx_add, dx + addend), "dgrad + addend differs from the separate add (max %.3e)" % float((dx_add - dx - addend).abs().max())
        pw = part.double().view(rows, 3, -1)[:, :, :cout]
        cnt, mean_r, m2_r = pw[:, 0], pw[:, 1], pw[:, 2]
        assert float(cnt[:, 0].sum()) == pixels
        mean = (cnt * mean_r).sum(0) / pixels
        var = (m2_r + cnt * (mean_r - mean) ** 2).sum(0) / pixels
        y64 = y.double().transpose(0, 1).reshape(cout, -1)
        assert float((mean - y64.mean(1)).abs().max()) <= 1e-6 * float(y64.abs().max())
        assert float((var - y64.var(1, unbiased=False)).abs().max()) <= 1e-6 * float(y64.var(1, unbiased=False).max())
        if math == "f16x3":
            x64 = x.double().requires_grad_()
            ref = F.conv2d(x64, wt.double(), None, 1, pad, d)
            _assert_close(y, ref, 2e-5, "forward (wide tile)")
            (gx_ref,) = torch.autograd.grad(ref, [x64], gy.double())
            _assert_close(dx, gx_ref, 2e-5, "data gradient (wide tile)")


# (Cin, Cout, k, stride, dil, H, W, N, math): enough 16-pixel K-steps for the stream-K plan (>= 32 per CU)
WGRAD_PP_CASES = [
    (512, 512, 3, 1, 2, 24, 32, 6, "f16x3"),   # 36 tiles x 288 K-steps: pieces of 41 K-steps, most of them inside one tile
    (256, 256, 3, 1, 2, 60, 80, 4, "f16x3"),   # 9 tiles x 1 200: a 256-channel layer of the benchmark (N = 4)
    (392, 504, 3, 1, 4, 29, 37, 8, "f16x3"),   # ragged channel groups on both sides (two tiles each way, the second partly empty), ragged pixel tiles
    (256, 512, 1, 1, 1, 61, 83, 16, "f16x3"),  # 1x1: two tiles, pieces far shorter than a tile
    (256, 400, 3, 2, 1, 63, 81, 6, "f16x3"),   # stride 2: X is gathered through the convolution geometry
    (512, 512, 3, 1, 2, 24, 32, 6, "f16x1"),   # the reduced-precision arithmetic (the piece-1 waves move nothing)
    # BASELINE config 2's own shapes (N = 16, 60 x 80 maps): the slab plan at exactly the 9 / 18 / 36 tiles the benchmark launches
    (256, 256, 3, 1, 2, 60, 80, 16, "f16x3"),  # base.5.{1..5}: 9 tiles
    (256, 512, 3, 1, 4, 60, 80, 16, "f16x3"),  # base.6.0 conv1: 18 tiles
    (512, 512, 3, 1, 4, 60, 80, 16, "f16x3"),  # base.6.{0,1,2}: 36 tiles
]


@pytest.mark.parametrize("case", WGRAD_PP_CASES, ids=lambda c: "x".join(map(str, c)))
def test_conv_wgrad_pingpong_stream_k(case, monkeypatch, libopt):
    """``conv_wgrad_split_pp_kernel`` (csrc/conv_wgrad_split_pp.hip): the weight gradient of the 256-channel-and-wider layers on
    256 x 256 tiles, eight waves in two groups half a K-step apart, over a split of the pixel range -- by default K slabs whose tiles run
    side by side on one XCD, or (MCDSEG_WGRAD_PP=1) stream-K: one equal piece of the flattened (tile, K-step) range per CU -- with one
    fp32 slab per segment, summed in K order in fp64.  Named through
    ``mcdseg_conv_wgrad_variant == 17``; within 2e-5 of fp64; two runs agree bit for bit; and it equals the 4-wave transposed-read
    kernels (MCDSEG_WGRAD_PP=0) to the last bits -- the same products, only grouped into other partial sums."""
    dev = _dev()
    import ctypes
    from mcdseg import ops
    cin, cout, k, s, d, h, w, n, math = case
    monkeypatch.setattr(ops, "CONV_MATH", math)
    libopt(WGRAD_PP=2)  # (the default)
    x, wt, _, s, pad, d = _conv_inputs((cin, cout, k, s, d, h, w, n, False), 43)
    desc = ops.conv_desc(x.shape, wt.shape, s, pad, d)
    L, mid = ops.lib(), ops.MATH_ID[math]
    assert L.mcdseg_conv_wgrad_variant(ctypes.byref(desc), mid, 1) == 17
    gy = torch.randn(n, cout, desc.Ho, desc.Wo, generator=torch.Generator().manual_seed(44))
    xg, gyg = x.to(dev), gy.to(dev)
    x_cb, x_bound = ops.split_companion(xg)
    gy_cb, gy_bound = ops.split_companion(gyg)
    names = []

    class _Names:
        def wants(self, name):
            names.append(name)
            return False
    prev, ops.LAUNCH_TIMER = ops.LAUNCH_TIMER, _Names()
    try:
        dw = ops._conv_wgrad(desc, xg, gyg, x_cb, gy_cb, x_bound, gy_bound)
    finally:
        ops.LAUNCH_TIMER = prev
    # (the one-term arithmetic runs the kernel with two K-steps per barrier interval: policy SplitF16x1D, tests/test_half_storage_gpu.py)
    assert names == ["conv_wgrad_split_pp_kernel<%s>" % ("SplitF16x1D" if math == "f16x1" else ops.POLICY[math])], names
    dw_b = ops._conv_wgrad(desc, xg, gyg, x_cb, gy_cb, x_bound, gy_bound)
    assert torch.equal(dw, dw_b), "two runs differ"
    # the default decomposition is the slab plan (K slabs whose tiles run side by side on one XCD); MCDSEG_WGRAD_PP=1 is stream-K
    libopt(WGRAD_PP=1)
    assert L.mcdseg_conv_wgrad_variant(ctypes.byref(desc), mid, 1) == 17
    dw_sk = ops._conv_wgrad(desc, xg, gyg, x_cb, gy_cb, x_bound, gy_bound)
    assert torch.equal(dw_sk, ops._conv_wgrad(desc, xg, gyg, x_cb, gy_cb, x_bound, gy_bound)), "two stream-K runs differ"
    # (fp32 accumulation inside a slab / a stream-K piece: the two plans cut K = N Ho Wo pixels differently, and at the benchmark's 76800
    # pixels a slab sums ~11 000 products per accumulator -- 2.1e-6 of the scale measured there, 1e-6 at the smaller cases)
    assert float((dw - dw_sk).abs().max()) <= 4e-6 * float(dw.abs().max()), "the two decompositions differ by more than their rounding"
    libopt(WGRAD_PP=0)
    assert L.mcdseg_conv_wgrad_variant(ctypes.byref(desc), mid, 1) in (12, 13)
    dw_tr = ops._conv_wgrad(desc, xg, gyg, x_cb, gy_cb, x_bound, gy_bound)
    scale = float(dw_tr.abs().max())
    assert float((dw - dw_tr).abs().max()) <= 4e-6 * scale, float((dw - dw_tr).abs().max()) / scale
    if math == "f16x3":
        x64, w64 = x.double(), wt.double().requires_grad_()
        ref = F.conv2d(x64, w64, None, stride=s, padding=pad, dilation=d)
        (gw_ref,) = torch.autograd.grad(ref, [w64], gy.double())
        _assert_close(dw, gw_ref, 2e-5, "wgrad (ping-pong, stream-K)")


# (Cin, Cout, k, stride, dil, H, W, N, math): 3 x 3 kernels with more than 64 channels on both sides and at most 128 on one
WGRAD_PP3_CASES = [
    (128, 128, 3, 1, 1, 60, 80, 16, "f16x3"),   # base.4.{0 conv2, 1, 2, 3}: BASELINE config 2's own shape, 3 tiles x 85 slabs
    (128, 256, 3, 1, 2, 60, 80, 6, "f16x3"),    # base.5.0 conv1 (two co tiles), dilation 2
    (96, 128, 3, 1, 1, 45, 50, 16, "f16x3"),    # ragged channel groups (the ci tile three quarters full), ragged 16-pixel stages
    (128, 128, 3, 2, 1, 121, 163, 16, "f16x3"),  # stride 2: X is gathered through the convolution geometry
    (128, 128, 3, 1, 1, 60, 80, 16, "f16x1"),   # the reduced-precision arithmetic (the piece-1 waves move nothing)
]


@pytest.mark.parametrize("case", WGRAD_PP3_CASES, ids=lambda c: "x".join(map(str, c)))
def test_conv_wgrad_row_of_taps(case, monkeypatch, libopt):
    """``conv_wgrad_split_pp3_kernel`` (csrc/conv_wgrad_split_pp.hip; round 5): the weight gradient of the 128-channel layers on tiles
    of 128 (co) x one kernel row of three taps x 128 (ci) -- the ping-pong kernel's structure, one staged dZ block serving three taps --
    over the slab plan, one fp32 slab per item, summed in K order in fp64.  Named through ``mcdseg_conv_wgrad_variant == 18``; within
    2e-5 of fp64; two runs agree bit for bit; and it equals the 4-wave transposed-read kernel (MCDSEG_WGRAD_PP3=0) to the last bits --
    the same products, grouped into other partial sums."""
    dev = _dev()
    import ctypes
    from mcdseg import ops
    cin, cout, k, s, d, h, w, n, math = case
    monkeypatch.setattr(ops, "CONV_MATH", math)
    libopt(WGRAD_PP3=1)  # (the default)
    x, wt, _, s, pad, d = _conv_inputs((cin, cout, k, s, d, h, w, n, False), 53)
    desc = ops.conv_desc(x.shape, wt.shape, s, pad, d)
    L, mid = ops.lib(), ops.MATH_ID[math]
    assert L.mcdseg_conv_wgrad_variant(ctypes.byref(desc), mid, 1) == 18
    gy = torch.randn(n, cout, desc.Ho, desc.Wo, generator=torch.Generator().manual_seed(54))
    xg, gyg = x.to(dev), gy.to(dev)
    x_cb, x_bound = ops.split_companion(xg)
    gy_cb, gy_bound = ops.split_companion(gyg)
    names = []

    class _Names:
        def wants(self, name):
            names.append(name)
            return False
    prev, ops.LAUNCH_TIMER = ops.LAUNCH_TIMER, _Names()
    try:
        dw = ops._conv_wgrad(desc, xg, gyg, x_cb, gy_cb, x_bound, gy_bound)
    finally:
        ops.LAUNCH_TIMER = prev
    assert names == ["conv_wgrad_split_pp3_kernel<%s>" % ops.POLICY[math]], names
    assert torch.equal(dw, ops._conv_wgrad(desc, xg, gyg, x_cb, gy_cb, x_bound, gy_bound)), "two runs differ"
    libopt(WGRAD_PP3=0)
    assert L.mcdseg_conv_wgrad_variant(ctypes.byref(desc), mid, 1) in (12, 13)
    dw_tr = ops._conv_wgrad(desc, xg, gyg, x_cb, gy_cb, x_bound, gy_bound)
    scale = float(dw_tr.abs().max())
    assert float((dw - dw_tr).abs().max()) <= 4e-6 * scale, float((dw - dw_tr).abs().max()) / scale
    if math == "f16x3":
        x64, w64 = x.double(), wt.double().requires_grad_()
        ref = F.conv2d(x64, w64, None, stride=s, padding=pad, dilation=d)
        (gw_ref,) = torch.autograd.grad(ref, [w64], gy.double())
        _assert_close(dw, gw_ref, 2e-5, "wgrad (row of taps)")


@pytest.mark.parametrize("case", [(128, 128, 3, 1, 1, 12, 16, 2), (136, 200, 3, 1, 2, 13, 19, 2), (256, 128, 3, 1, 4, 9, 10, 1), (128, 128, 3, 2, 1, 15, 17, 2),
                                  (72, 80, 3, 1, 1, 8, 8, 3), (128, 256, 3, 1, 2, 30, 40, 4), (128, 136, 5, 1, 1, 9, 11, 1)], ids=lambda c: "x".join(map(str, c)))
def test_conv_wgrad_two_taps_per_workgroup(case, monkeypatch, libopt):
    """``conv_wgrad_split_tr_kernel<SplitF16x3, 4, 2, 3, true>``: the 128-row weight-gradient layers with two taps per workgroup (X at both
    tap shifts on the "A" side, one shared dY tile on the "B" side, transposed output tile; odd tap counts leave the last pair half empty):
    named through ``mcdseg_conv_wgrad_variant == 16``, <= 2e-5 of the scale against fp64, and bit for bit the one-tap kernel
    (MCDSEG_WGRAD_TWOTAP=0) -- the cross terms' pieces are swapped with the roles, so products and their order are the same."""
    dev = _dev()
    import ctypes
    from mcdseg import ops
    monkeypatch.setattr(ops, "CONV_MATH", "f16x3")
    libopt(WGRAD_TWOTAP=1)  # (off by default: slower at the benchmark's sizes)
    cin, cout, k, s, d, h, w, n = case
    x, wt, _, s, pad, d = _conv_inputs((cin, cout, k, s, d, h, w, n, False), 43)
    desc = ops.conv_desc(x.shape, wt.shape, s, pad, d)
    if min(cin, cout) <= 64:  # (the 64-channel plan has its own two-tap kernel)
        assert ops.lib().mcdseg_conv_wgrad_variant(ctypes.byref(desc), ops.MATH_ID["f16x3"], 1) == 14
        return
    assert ops.lib().mcdseg_conv_wgrad_variant(ctypes.byref(desc), ops.MATH_ID["f16x3"], 1) == 16
    x64, w64 = x.double(), wt.double().requires_grad_()
    ref = F.conv2d(x64, w64, None, stride=s, padding=pad, dilation=d)
    gy = torch.randn(ref.shape, generator=torch.Generator().manual_seed(44))
    (gw_ref,) = torch.autograd.grad(ref, [w64], gy.double())
    xg, gyg = x.to(dev), gy.to(dev)
    x_cb, x_bound = ops.split_companion(xg)
    gy_cb, gy_bound = ops.split_companion(gyg)
    dw2 = ops._conv_wgrad(desc, xg, gyg, x_cb, gy_cb, x_bound, gy_bound)
    _assert_close(dw2, gw_ref, 2e-5, "wgrad (two taps per workgroup)")
    libopt(WGRAD_TWOTAP=0)
    assert ops.lib().mcdseg_conv_wgrad_variant(ctypes.byref(desc), ops.MATH_ID["f16x3"], 1) == 12
    dw1 = ops._conv_wgrad(desc, xg, gyg, x_cb, gy_cb, x_bound, gy_bound)
    assert torch.equal(dw2, dw1)


@pytest.mark.parametrize("cin,h,w,n", [(6, 21, 45, 2), (3, 16, 64, 1), (1, 9, 33, 2)])
def test_stem_wgrad_from_padded_companion(cin, h, w, n, monkeypatch):
    """7x7 stem weight gradient in the split arithmetic: the 6- (3-, 1-) channel input's zero-padded companion
    (mcdseg_split_cb_padded) and the companion of dz through the thin-layer window kernel, against fp64 and the f32 kernel"""
    dev = _dev()
    import ctypes
    from mcdseg import ops
    monkeypatch.setattr(ops, "CONV_MATH", "f16x3")
    x, wt, _, s, pad, d = _conv_inputs((cin, 16, 7, 1, 1, h, w, n, False), 31)
    desc = ops.conv_desc(x.shape, wt.shape, s, pad, d)
    assert ops.lib().mcdseg_conv_wgrad_variant(ctypes.byref(desc), ops.MATH_ID["f16x3"], 1) == 15
    x64, w64 = x.double(), wt.double().requires_grad_()
    ref = F.conv2d(x64, w64, None, stride=s, padding=pad, dilation=d)
    gy = torch.randn(ref.shape, generator=torch.Generator().manual_seed(32))
    (gw_ref,) = torch.autograd.grad(ref, [w64], gy.double())
    xg, gyg = x.to(dev), gy.to(dev)
    x_cb, x_bound = ops.split_companion_padded(xg)
    again, _ = ops.split_companion_padded(xg)
    assert again is x_cb  # cached on the tensor
    xg2 = xg.clone()
    xg2.mul_(2.0)
    assert ops.split_companion_padded(xg2)[0] is not x_cb
    gy_cb, gy_bound = ops.split_companion(gyg)
    dw_f32 = ops._conv_wgrad(desc, xg, gyg, None, None, None, None)
    dw_tr = ops._conv_wgrad(desc, xg, gyg, x_cb, gy_cb, x_bound, gy_bound)
    _assert_close(dw_f32, gw_ref, 2e-5, "stem wgrad (f32 kernel)")
    _assert_close(dw_tr, gw_ref, 2e-5, "stem wgrad (window kernel, split operands)")


# ------------------------------------------------------------------------------ input pipeline and evaluation kernels
def _io_golden():
    import os
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "io_small.npz"))


def test_normalize_u8_is_bit_exact():
    """ToTensor+Normalize on the device = the oracle's fp32 arithmetic, bit for bit, including the two-source (RGB + HHA)
    form and every byte value"""
    dev = _dev()
    from datasets import DeviceInputPipeline
    from oracle import ref_io
    rng = np.random.RandomState(6)
    rgb = rng.randint(0, 256, size=(3, 21, 35, 3)).astype(np.uint8)
    hha = rng.randint(0, 256, size=(3, 21, 35, 3)).astype(np.uint8)
    rgb[0, :, :, 0].flat[:256] = np.arange(256, dtype=np.uint8)  # all byte values
    pipe = DeviceInputPipeline(6, 41, dev)
    got = pipe.images(torch.from_numpy(rgb), torch.from_numpy(hha))
    ref = ref_io.normalize_u8(np.concatenate([rgb, hha], axis=3), ref_io.IMAGENET_MEAN6, ref_io.IMAGENET_STD6)
    assert got.shape == (3, 6, 21, 35) and np.array_equal(got.cpu().numpy(), ref)
    one = DeviceInputPipeline(1, 41, dev, normalize_way="none").images(torch.from_numpy(rgb[..., :1]))
    assert np.array_equal(one.cpu().numpy(), ref_io.normalize_u8(rgb[..., :1], [0.0], [1.0]))
    six = DeviceInputPipeline(6, 41, dev).images(torch.from_numpy(np.concatenate([rgb, hha], axis=3)))  # one 6-channel source
    assert torch.equal(six, got)


def test_device_resize_matches_pillow_vectors(golden):
    """Scale(img_shape, Image.BILINEAR / NEAREST) on the device = PIL.Image.resize, bit for bit (vectors made by the real
    Pillow, tests/golden/make_golden_resize.py), batched, and through the pipeline: resize -> normalise / relabel = the oracle's
    chain of the same steps"""
    dev = _dev()
    from datasets import DeviceInputPipeline
    from mcdseg import ops
    from oracle import ref_io
    fx = golden.npz("resize_small.npz")
    for tag in [k[4:] for k in fx.files if k.startswith("img_")]:
        size = tuple(int(v) for v in fx["size_" + tag])
        img = torch.from_numpy(np.stack([fx["img_" + tag], fx["img_" + tag][::-1].copy()])).to(dev)   # a batch of two
        got = ops.resize_u8(img, size).cpu().numpy()
        assert np.array_equal(got[0], fx["rimg_" + tag]), tag
        assert np.array_equal(got[1], ref_io.resize_bilinear_u8(fx["img_" + tag][::-1], size)), tag
        lbl = torch.from_numpy(fx["lbl_" + tag][None]).to(dev)
        assert np.array_equal(ops.resize_u8(lbl, size, nearest=True).cpu().numpy()[0], fx["rlbl_" + tag]), tag
    rng = np.random.RandomState(8)
    rgb = rng.randint(0, 256, size=(2, 37, 53, 3)).astype(np.uint8)
    hha = rng.randint(0, 256, size=(2, 37, 53, 3)).astype(np.uint8)
    lbl = rng.randint(0, 41, size=(2, 37, 53)).astype(np.uint8)
    lbl[:, :5, :5] = 255
    pipe = DeviceInputPipeline(6, 41, dev, img_shape=(32, 24))
    got = pipe.images(torch.from_numpy(rgb), torch.from_numpy(hha)).cpu().numpy()
    ref = np.stack([ref_io.normalize_u8(np.concatenate([ref_io.resize_bilinear_u8(rgb[i], (32, 24)), ref_io.resize_bilinear_u8(hha[i], (32, 24))],
                                                       axis=2), ref_io.IMAGENET_MEAN6, ref_io.IMAGENET_STD6) for i in range(2)])
    assert got.shape == (2, 6, 24, 32) and np.array_equal(got, ref)
    gl = pipe.labels(torch.from_numpy(lbl)).cpu().numpy()
    assert np.array_equal(gl, np.stack([ref_io.relabel(ref_io.resize_nearest_u8(lbl[i], (32, 24)), 255, 40) for i in range(2)]))


def test_relabel_u8_matches_reference_vector():
    dev = _dev()
    from mcdseg import ops
    g = _io_golden()
    got = ops.relabel_u8(torch.from_numpy(g["lbl_u8"]).to(dev), 255, 40)
    assert got.dtype == torch.int64 and np.array_equal(got.cpu().numpy(), g["lbl_i64"])


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_confusion_hist_matches_reference_vector(tag):
    """fast_hist on the device (LDS bins for n <= 64, global bins above) = the reference's bincount, exactly; the
    call accumulates, and the summary numbers of the meter equal eval.py's"""
    dev = _dev()
    import eval as mc_eval
    g = _io_golden()
    n = int(g["n_" + tag])
    gt, pred = torch.from_numpy(g["gt_" + tag]).to(dev), torch.from_numpy(g["pred_" + tag]).to(dev)
    hist = mc_eval.fast_hist(gt, pred, n)
    assert np.array_equal(hist.cpu().numpy(), g["hist_" + tag])
    mc_eval.fast_hist(gt, pred, n, out=hist)
    assert np.array_equal(hist.cpu().numpy(), 2 * g["hist_" + tag])
    meter = mc_eval.ConfusionMeter(n, device=dev)
    for i in range(gt.shape[0]):
        meter.update(pred[i], gt[i])
    s = meter.summary()
    assert np.allclose(np.array(s["IoU"]) / 100, g["iu_" + tag], rtol=1e-12, equal_nan=True)
    assert np.isclose(s["fwIoU"] / 100, float(g["fw_" + tag])) and np.isclose(s["pixAcc"] / 100, float(g["pa_" + tag]))
    assert np.isclose(s["mAcc"] / 100, float(g["ma_" + tag]))


def test_confusion_hist_full_size_checksum():
    """BASELINE-size property: 16 x 480 x 640 label maps -- the matrix sums to the number of non-background pixels, its
    row sums are the ground-truth class counts, its column sums the prediction counts over those pixels"""
    dev = _dev()
    import eval as mc_eval
    g = torch.Generator().manual_seed(9)
    gt = torch.randint(0, 41, (16, 480, 640), generator=g)
    gt[gt == 40] = 255
    pred = torch.randint(0, 40, (16, 480, 640), generator=g)
    hist = mc_eval.fast_hist(gt.to(dev), pred.to(dev), 41).cpu()
    keep = gt != 255
    assert int(hist.sum()) == int(keep.sum())
    assert torch.equal(hist.sum(1), torch.bincount(gt[keep], minlength=41))
    assert torch.equal(hist.sum(0), torch.bincount(pred[keep], minlength=41))


def test_stem_direct_conv_matches_generic_kernels(monkeypatch):
    """the stem's direct convolution (bf16x6) against fp64 and against the f32-MFMA implicit GEMM: plain, with bias, and the
    folded-BN inference epilogue (scale, shift, ReLU)"""
    dev = _dev()
    from mcdseg import ops
    from mcdseg._lib import lib
    from models.drn import BatchNorm2d, Conv2d
    import ctypes
    g = torch.Generator().manual_seed(51)
    x = torch.randn(2, 6, 23, 77, generator=g)
    wt = torch.randn(16, 6, 7, 7, generator=g) * 0.08
    bias = torch.randn(16, generator=g) * 0.1
    desc = ops.conv_desc(x.shape, wt.shape, 1, 3, 1)
    assert lib().mcdseg_conv_split_direct_ok(ctypes.byref(desc)) == 1
    ref = F.conv2d(x.double(), wt.double(), bias.double(), padding=3)
    outs = {}
    for math in ("f32", "bf16x6", "f16x3"):  # the direct kernel keeps the bf16x6 arithmetic under either split mode
        monkeypatch.setattr(ops, "CONV_MATH", math)
        wf, _, mpf = ops.PackedWeights().get(wt.to(dev), desc, need_dgrad=False)
        assert ops._is_split(wf) == (math != "f32")
        y, part, rows = ops._conv_fprop(desc, x.to(dev), wf, bias.to(dev), True, mpf)
        _assert_close(y, ref, 2e-5, math + " stem fprop")
        outs[math] = _maxerr(y, ref)[0]
    assert outs["bf16x6"] <= max(2.0 * outs["f32"], 2e-6 * float(ref.abs().max())) and outs["f16x3"] == outs["bf16x6"]
    # inference: eval-mode BN folded into the epilogue
    monkeypatch.setattr(ops, "CONV_MATH", "f16x3")
    conv, bn = Conv2d(6, 16, 7, padding=3, bias=False), BatchNorm2d(16)
    with torch.no_grad():
        conv.weight.copy_(wt)
        bn.weight.copy_(1 + 0.2 * torch.randn(16, generator=g)), bn.bias.copy_(0.1 * torch.randn(16, generator=g))
        bn.running_mean.copy_(0.1 * torch.randn(16, generator=g)), bn.running_var.copy_(0.5 + torch.rand(16, generator=g))
    bn.eval()
    o = F.relu(F.batch_norm(F.conv2d(x.double(), wt.double(), None, padding=3), bn.running_mean.double(), bn.running_var.double(),
                            bn.weight.double(), bn.bias.double(), training=False, eps=1e-5))
    conv.to(dev), bn.to(dev)
    with torch.no_grad():
        y = ops.conv_bn_act(x.to(dev), conv, bn, relu=True)
    _assert_close(y, o, 3e-5, "stem inference")
