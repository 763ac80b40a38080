"""CPU: the oracle (oracle/) against the golden vectors captured from the real reference."""
import numpy as np
import pytest
import torch

from oracle import ref_loss, ref_mcd, ref_models
from recipe import checksum, fill_state_, make_batch, state_checksums

NC = 41


def _keys_shapes(m):
    return {k: list(v.shape) for k, v in m.state_dict().items()}


def test_state_dict_layouts(golden):
    ks = golden.json("keys_shapes.json")
    g, f1, f2 = ref_models.get_models("drn_d_38", 6, NC)
    assert _keys_shapes(g) == ks["MCD/drn_d_38/6ch/G"]
    assert len(ks["MCD/drn_d_38/6ch/G"]) == 248
    assert _keys_shapes(f1) == ks["MCD/drn_d_38/6ch/F"] == {"up.weight": [41, 1, 16, 16]}
    g3, g1, mf1, _ = ref_models.get_models("drn_d_38", 6, NC, method="MFNet-ScoreAddFusion")
    assert _keys_shapes(g3) == ks["MFNet-ScoreAddFusion/drn_d_38/6ch/G_3ch"]
    assert _keys_shapes(g1) == ks["MFNet-ScoreAddFusion/drn_d_38/6ch/G_1ch"]
    assert _keys_shapes(mf1) == ks["MFNet-ScoreAddFusion/drn_d_38/6ch/F"]
    assert _keys_shapes(ref_models.get_models("drn_d_38", 6, NC, method="MFNet-AddFusion")[2]) == ks["MFNet-AddFusion/drn_d_38/6ch/F"]
    g, f, _ = ref_models.get_models("drn_d_38_ver2", 6, NC)
    assert _keys_shapes(g) == ks["MCD/drn_d_38_ver2/6ch/G"]
    assert _keys_shapes(f) == ks["MCD/drn_d_38_ver2/6ch/F"]
    assert _keys_shapes(ref_models.get_models("drn_d_105", 6, NC)[0]) == ks["MCD/drn_d_105/6ch/G"]
    assert _keys_shapes(ref_models.get_full_model("drn_d_38", "50", NC, 6)) == ks["full/drn_d_38/6ch/DataParallel"]


def test_unknown_method_is_returned_not_raised():
    # models/model_util.py:281 returns the exception object
    assert isinstance(ref_models.get_models("drn_d_38", 6, NC, method="DANN"), NotImplementedError)
    with pytest.raises(NotImplementedError):
        ref_models.get_models("fcn", 6, NC)


def test_init_and_first_conv(golden):
    torch.manual_seed(3)
    g = ref_models.DRNSegBase("drn_d_38", NC, input_ch=6)
    w = g.base[0][0].weight.data
    assert torch.equal(w[:, 3:6], w[:, :3])
    st = golden.json("traces.json")["init_stats"]
    std = float(g.base[6][1].conv1.weight.data.std())
    assert abs(std - st["expected"]) < 0.02 * st["expected"]
    assert abs(st["layer6.1.conv1.std"] - st["expected"]) < 0.02 * st["expected"]
    assert float(g.base[5][0].bn1.weight.detach().min()) == 1.0 and float(g.seg.bias.detach().abs().max()) == 0.0
    with pytest.raises(NotImplementedError):
        ref_models.widen_first_conv(w[:, :3], 7)


@pytest.mark.parametrize("mode", ["train", "eval"])
def test_forward_small(golden, mode):
    fx = golden.npz("fwd_small.npz")
    g, f1, f2 = ref_models.get_models("drn_d_38", 6, NC)
    for m, seed in ((g, 11), (f1, 12), (f2, 13)):
        fill_state_(m, seed)
        m.train() if mode == "train" else m.eval()
    src, _, _ = make_batch(21, 2, 6, 64, 96, NC)
    with torch.no_grad():
        feat = g(src)
        o1, o2 = f1(feat), f2(feat)
    np.testing.assert_allclose(feat.numpy(), fx["feat_" + mode], rtol=0, atol=1e-5 * np.abs(fx["feat_" + mode]).max())
    np.testing.assert_allclose(o1[:, :, ::4, ::4].numpy(), fx["logits1_sub_" + mode], rtol=0, atol=2e-5)
    np.testing.assert_allclose(checksum(o2), fx["logits2_cs_" + mode], rtol=1e-5)
    pred = o1[:, :NC - 1].argmax(1).numpy()
    safe = fx["margin1_" + mode] > 1e-4
    assert (pred == fx["argmax1_" + mode])[safe].all()
    if mode == "train":
        sd = g.state_dict()
        for k in fx.files:
            if k.startswith("rs/"):
                np.testing.assert_allclose(sd[k[3:]].numpy(), fx[k], rtol=1e-5, atol=1e-6)
        assert int(sd["base.8.1.num_batches_tracked"]) == int(fx["nbt"]) == 1


def test_losses(golden):
    fx = golden.npz("loss_small.npz")
    z1 = torch.from_numpy(fx["z1"]).requires_grad_()
    z2 = torch.from_numpy(fx["z2"]).requires_grad_()
    y = torch.from_numpy(fx["y"])
    ce = ref_loss.CrossEntropyLoss2d(torch.from_numpy(fx["w"]))(z1, y)
    g, = torch.autograd.grad(ce, z1)
    assert abs(float(ce) - float(fx["ce"])) < 1e-6 * float(fx["ce"])
    np.testing.assert_allclose(g.numpy(), fx["g_ce"], rtol=0, atol=1e-9)
    cew = ref_loss.CrossEntropyLoss2d(torch.from_numpy(fx["wfull"]))(z1, y)
    assert abs(float(cew) - float(fx["ce_w"])) < 1e-6 * float(fx["ce_w"])
    d = ref_loss.get_prob_distance_criterion("diff")(z1, z2)
    g1, g2 = torch.autograd.grad(d, [z1, z2])
    assert abs(float(d) - float(fx["diff"])) < 1e-6 * float(fx["diff"])
    np.testing.assert_allclose(g1.numpy(), fx["g_d1"], rtol=0, atol=1e-10)
    np.testing.assert_allclose(g2.numpy(), fx["g_d2"], rtol=0, atol=1e-10)
    # closed forms (Appendix C) used as the spec of the fused HIP kernel
    l64, g64 = ref_loss.ce_and_grad(fx["z1"], fx["y"], fx["w"])
    assert abs(l64 - float(fx["ce"])) < 1e-5
    assert np.abs(g64 - fx["g_ce"]).max() < 1e-7
    d64, a, b = ref_loss.diff_and_grad(fx["z1"], fx["z2"])
    assert abs(d64 - float(fx["diff"])) < 1e-7
    assert np.abs(a - fx["g_d1"]).max() < 1e-8 and np.abs(b - fx["g_d2"]).max() < 1e-8
    with pytest.raises(NotImplementedError):
        ref_loss.get_prob_distance_criterion("jsd")


@pytest.mark.parametrize("which", ["ce", "diff"])
def test_backward_small(golden, which):
    fx = golden.npz("bwd_small.npz")
    g, f1, f2 = ref_models.get_models("drn_d_38", 6, NC)
    for m, seed in ((g, 11), (f1, 12), (f2, 13)):
        fill_state_(m, seed)
        m.train()
    src, lbl, tgt = make_batch(21, 2, 6, 64, 96, NC)
    feat = g(src if which == "ce" else tgt)
    a, b = f1(feat), f2(feat)
    if which == "ce":
        crit = ref_loss.CrossEntropyLoss2d(ref_loss.class_weights(NC))
        loss = crit(a, lbl) + crit(b, lbl)
    else:
        loss = ref_loss.Diff2d()(a, b)
    loss.backward()
    assert abs(float(loss) - float(fx[which + "/loss32"])) < 2e-6 * abs(float(fx[which + "/loss32"]))
    named = dict(g.named_parameters())
    for key in fx.files:
        if not key.startswith(which + "/f64/"):
            continue
        name = key.split("/", 2)[2]
        if name in named:
            got = named[name].grad
            got = got if got.numel() <= 40000 else got.reshape(got.shape[0], -1)[:16, :288]
        elif name == "up1":
            got = f1.up.weight.grad
        elif name == "up2":
            got = f2.up.weight.grad
        else:
            continue
        g64 = fx[key]
        g32 = fx[key.replace("/f64/", "/f32/")]
        noise = np.abs(g32 - g64).max()
        err = np.abs(got.double().numpy() - g64).max()
        # the oracle has to sit inside the reference's own fp32 noise floor
        assert err <= max(4 * noise, 1e-5 * np.abs(g64).max(), 1e-9), (name, err, noise)


def _check_state(mod, ref_cs, rtol=2e-4):
    got = state_checksums(mod)
    assert got.keys() == ref_cs.keys()
    for k, (s, l2) in ref_cs.items():
        assert abs(got[k][1] - l2) <= rtol * max(abs(l2), 1e-6), (k, got[k], (s, l2))


def _flat_state(g, f1, f2):
    return dict(list(g.state_dict().items()) + [("f1." + k, v) for k, v in f1.state_dict().items()] +
                [("f2." + k, v) for k, v in f2.state_dict().items()])


def _mcd_trace(tr, deltas=None):
    g, f1, f2 = ref_models.get_models("drn_d_38", 6, NC)
    for m, seed in ((g, 11), (f1, 12), (f2, 13)):
        fill_state_(m, seed)
        m.train()
    before = {k: v.detach().clone() for k, v in _flat_state(g, f1, f2).items()}
    n, ch, h, w = tr["shape"]
    s, l, t = make_batch(tr["seed_batch"], n, ch, h, w, NC)
    og = ref_models.get_optimizer(g.parameters(), "sgd", 1e-3, 0.9, 2e-5)
    of = ref_models.get_optimizer(list(f1.parameters()) + list(f2.parameters()), "sgd", 1e-3, 0.9, 2e-5)
    crit = ref_loss.CrossEntropyLoss2d(ref_loss.class_weights(NC))
    for it in tr["iters"]:
        c, d = ref_mcd.mcd_step(g, f1, f2, og, of, crit, ref_loss.Diff2d(), s, l, t)
        assert abs(c - it["c_loss"]) < 1e-4 * it["c_loss"]
        assert abs(d - it["d_loss"]) < 1e-3 * it["d_loss"]
    _check_state(g, tr["g"]), _check_state(f1, tr["f1"]), _check_state(f2, tr["f2"])
    assert int(g.state_dict()["base.0.1.num_batches_tracked"]) == tr["nbt"] == 7 * len(tr["iters"])
    assert sorted(og.state_dict().keys()) == tr["opt_g_state_keys"]
    if deltas is not None:
        # the updates themselves (after - before) against the reference's, inside the reference's own fp32-vs-fp64 spread
        after = _flat_state(g, f1, f2)
        for key in deltas.files:
            if not key.startswith("f64/"):
                continue
            kind, name = key[4:].split("/", 1)
            r64, r32 = deltas[key], deltas["f32/" + key[4:]].astype(np.float64)
            noise = np.linalg.norm(r32 - r64) / np.linalg.norm(r64)
            cur = after[name].detach().double()
            got = cur - before[name].double() if kind == "delta" else cur
            got = (got if got.numel() <= 40000 else got.reshape(got.shape[0], -1)[:16, :288]).numpy()
            rel = np.linalg.norm(got - r64) / np.linalg.norm(r64)
            assert rel <= max(4.0 * noise, 1e-6), (key, rel, noise)


def test_mcd_three_step_small(golden):
    _mcd_trace(golden.json("traces.json")["mcd_small"], golden.npz("trace_deltas.npz"))


def test_mcd_three_step_240x320(golden):
    _mcd_trace(golden.json("traces.json")["mcd_240x320"])


def test_mfnet_three_step(golden):
    tr = golden.json("traces.json")["mfnet_small"]
    ms = ref_models.get_models("drn_d_38", 6, NC, method="MFNet-ScoreAddFusion")
    for m, seed in zip(ms, (51, 52, 53, 54)):
        fill_state_(m, seed)
        m.train()
    n, ch, h, w = tr["shape"]
    s, l, t = make_batch(tr["seed_batch"], n, ch, h, w, NC)
    fx = golden.npz("mfnet_small.npz")
    with torch.no_grad():
        a, b = ms[0](s[:, :3]), ms[1](s[:, 3:])
        o = ms[2](a, b)
    np.testing.assert_allclose(a.numpy(), fx["feat_rgb"], rtol=0, atol=1e-5 * np.abs(fx["feat_rgb"]).max())
    np.testing.assert_allclose(o[:, :, ::4, ::4].numpy(), fx["logits1_sub"], rtol=0, atol=2e-5)
    for m, seed in zip(ms, (51, 52, 53, 54)):
        fill_state_(m, seed)
    og = ref_models.get_optimizer(list(ms[0].parameters()) + list(ms[1].parameters()), "sgd", 1e-3, 0.9, 2e-5)
    of = ref_models.get_optimizer(list(ms[2].parameters()) + list(ms[3].parameters()), "sgd", 1e-3, 0.9, 2e-5)
    crit = ref_loss.CrossEntropyLoss2d(ref_loss.class_weights(NC))
    c, d = ref_mcd.mfnet_mcd_step(ms[0], ms[1], ms[2], ms[3], og, of, crit, ref_loss.Diff2d(), s, l, t)
    assert abs(c - tr["c_loss"]) < 1e-4 * tr["c_loss"] and abs(d - tr["d_loss"]) < 1e-3 * tr["d_loss"]
    _check_state(ms[0], tr["g_3ch"]), _check_state(ms[1], tr["g_1ch"]), _check_state(ms[2], tr["f1"])


def test_source_step_cfg1(golden):
    tr = golden.json("traces.json")["source_240x320"]
    m = ref_models.get_full_model("drn_d_38", "50", NC, 6)
    fill_state_(m, 61)
    m.train()
    n, ch, h, w = tr["shape"]
    s, l, _ = make_batch(tr["seed_batch"], n, ch, h, w, NC)
    opt = ref_models.get_optimizer(m.parameters(), "sgd", 1e-3, 0.9, 2e-5)
    loss = ref_mcd.source_step(m, opt, ref_loss.CrossEntropyLoss2d(ref_loss.class_weights(NC)), s, l)
    assert abs(loss - tr["loss"]) < 1e-5 * tr["loss"]
    _check_state(m, tr["state"])


def test_d105_bottleneck_small(golden):
    """drn_d_105 (Bottleneck blocks, the cfg5 trunk): forward features and selected gradients."""
    fx = golden.npz("d105_small.npz")
    tr = golden.json("traces.json")["d105_small"]
    g, f1, f2 = ref_models.get_models("drn_d_105", 6, NC)
    for m, seed in ((g, 71), (f1, 72), (f2, 73)):
        fill_state_(m, seed)
        m.train()
    n, ch, h, w = tr["shape"]
    s, l, _ = make_batch(tr["seed_batch"], n, ch, h, w, NC)
    crit = ref_loss.CrossEntropyLoss2d(ref_loss.class_weights(NC))
    feat = g(s)
    loss = crit(f1(feat), l) + crit(f2(feat), l)
    loss.backward()
    np.testing.assert_allclose(feat.detach().numpy(), fx["feat"], rtol=0, atol=1e-5 * np.abs(fx["feat"]).max())
    assert abs(float(loss) - tr["loss"]) < 1e-5 * tr["loss"]
    named = dict(g.named_parameters())
    np.testing.assert_allclose(named["seg.weight"].grad.numpy(), fx["g/seg.weight"], rtol=0, atol=1e-5 * np.abs(fx["g/seg.weight"]).max())


def test_multitask_cfg4(golden):
    from oracle import ref_multitask
    tr = golden.json("traces.json")["multitask_small"]
    fx = golden.npz("multitask_small.npz")
    crit = ref_loss.CrossEntropyLoss2d(ref_loss.class_weights(NC))
    enc, dec = ref_multitask.get_multitask_models("drn_d_38", 6, NC, crit, ref_loss.Diff2d())
    assert {k: list(v.shape) for k, v in enc.state_dict().items()} == tr["keys_shapes"]["enc"]
    assert {k: list(v.shape) for k, v in dec.state_dict().items()} == tr["keys_shapes"]["dec"]
    assert len(dec.state_dict()) == 51 and "semseg_criterion.nll_loss.weight" in dec.state_dict()
    fill_state_(enc, 81), fill_state_(dec, 82)
    enc.train(), dec.train()
    n, ch, h, w = tr["shape"]
    s, l, t = make_batch(tr["seed_batch"], n, ch, h, w, NC)
    with torch.no_grad():
        fet = enc(s[:, :3])
        a, b, d = dec(fet)
    np.testing.assert_allclose(fet.numpy(), fx["fet"], rtol=0, atol=1e-5 * np.abs(fx["fet"]).max())
    np.testing.assert_allclose(a[:, :, ::4, ::4].numpy(), fx["seg1_sub"], rtol=0, atol=1e-5 * np.abs(fx["seg1_sub"]).max())
    np.testing.assert_allclose(d.numpy(), fx["dep"], rtol=0, atol=1e-5 * np.abs(fx["dep"]).max())
    fill_state_(enc, 81), fill_state_(dec, 82)
    oe = ref_models.get_optimizer(enc.parameters(), "sgd", 1e-3, 0.9, 2e-5)
    od = ref_models.get_optimizer(dec.parameters(), "sgd", 1e-3, 0.9, 2e-5)
    c, dl, parts = ref_multitask.multitask_mcd_step(enc, dec, oe, od, s, l, t)
    assert abs(c - tr["c_loss"]) < 1e-4 * tr["c_loss"] and abs(dl - tr["d_loss"]) < 1e-3 * tr["d_loss"]
    assert all(abs(p - q) < 1e-3 * abs(q) for p, q in zip(parts, tr["parts"]))
    _check_state(enc, tr["enc"]), _check_state(dec, tr["dec"], rtol=1e-3)
    sd = dec.state_dict()
    assert int(enc.state_dict()["base.0.1.num_batches_tracked"]) == tr["nbt_enc"] == 8
    assert int(sd["semsegcls_dec1.cbr1.bn.num_batches_tracked"]) == tr["nbt_seg"] == 8
    assert int(sd["deprgr_dec.cbr1.bn.num_batches_tracked"]) == tr["nbt_dep"] == 4


# ------------------------------------------------------------------------------ either side of the step
def _io_golden():
    import os
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "io_small.npz"))


def test_eval_metrics_and_label_transform_match_reference_vectors():
    """oracle/ref_io.py against vectors produced by the reference's eval.py / transform.py (make_golden_io.py)"""
    from oracle import ref_io
    g = _io_golden()
    for tag in "abc":
        n = int(g["n_" + tag])
        gt, pred = g["gt_" + tag].astype(np.int64).flatten(), g["pred_" + tag].astype(np.int64).flatten()
        hist = ref_io.fast_hist(gt, pred, n)
        assert np.array_equal(hist, g["hist_" + tag])
        assert hist.sum() == int((gt < n).sum())                      # background (255) pixels are dropped
        used = np.where(hist.sum(1) != 0)[0]
        sub = hist[used][:, used].astype(np.float64)
        with np.errstate(divide="ignore", invalid="ignore"):
            assert np.allclose(ref_io.per_class_iu(sub), g["iu_" + tag], rtol=1e-12, equal_nan=True)
            assert np.isclose(ref_io.calc_fw_iu(sub), float(g["fw_" + tag]), rtol=1e-12)
            assert np.isclose(ref_io.calc_pixel_accuracy(sub), float(g["pa_" + tag]), rtol=1e-12)
            assert np.isclose(ref_io.calc_mean_accuracy(sub), float(g["ma_" + tag]), rtol=1e-12)
    assert np.array_equal(ref_io.relabel(g["lbl_u8"], 255, 40), g["lbl_i64"])


def test_resize_restatement_matches_pillow_vectors(golden):
    """Scale(img_shape, BILINEAR / NEAREST) = PIL.Image.resize: the oracle's restatement of Pillow's 8-bit arithmetic against
    vectors produced by the real Pillow (tests/golden/make_golden_resize.py), bit for bit -- and, where Pillow is installed,
    against Pillow itself on fresh data"""
    from oracle import ref_io
    fx = golden.npz("resize_small.npz")
    tags = [k[4:] for k in fx.files if k.startswith("img_")]
    assert len(tags) >= 6
    for tag in tags:
        size = tuple(int(v) for v in fx["size_" + tag])
        assert np.array_equal(ref_io.resize_bilinear_u8(fx["img_" + tag], size), fx["rimg_" + tag]), tag
        assert np.array_equal(ref_io.resize_nearest_u8(fx["lbl_" + tag], size), fx["rlbl_" + tag]), tag
    try:
        from PIL import Image
    except ImportError:
        return
    rng = np.random.RandomState(5)
    img = rng.randint(0, 256, size=(45, 61, 3)).astype(np.uint8)
    for size in ((30, 22), (100, 77), (61, 20)):
        assert np.array_equal(ref_io.resize_bilinear_u8(img, size), np.asarray(Image.fromarray(img).resize(size, Image.BILINEAR)))
        assert np.array_equal(ref_io.resize_nearest_u8(img[:, :, 0], size), np.asarray(Image.fromarray(img[:, :, 0]).resize(size, Image.NEAREST)))


def test_normalize_restatement_is_torch_totensor_normalize_arithmetic():
    """torchvision is absent, so ToTensor()+Normalize() are pinned to their published arithmetic executed by torch itself:
    byte -> float -> div(255), then sub_(mean).div_(std) per channel, all in fp32."""
    from oracle import ref_io
    rng = np.random.RandomState(5)
    img = rng.randint(0, 256, size=(2, 9, 11, 6)).astype(np.uint8)
    img[0, 0, 0] = [0, 255, 1, 254, 127, 128]
    got = ref_io.normalize_u8(img, ref_io.IMAGENET_MEAN6, ref_io.IMAGENET_STD6)
    t = torch.from_numpy(img).permute(0, 3, 1, 2).contiguous().float().div(255)
    mean = torch.tensor(ref_io.IMAGENET_MEAN6, dtype=torch.float32).view(1, -1, 1, 1)
    std = torch.tensor(ref_io.IMAGENET_STD6, dtype=torch.float32).view(1, -1, 1, 1)
    t.sub_(mean).div_(std)
    assert got.dtype == np.float32 and np.array_equal(got, t.numpy())


def test_host_metrics_module_matches_reference_vectors():
    """the product's eval.py (host formulas on the n x n matrix) against the same vectors"""
    import eval as mc_eval
    g = _io_golden()
    for tag in "abc":
        hist = g["hist_" + tag]
        used = np.where(hist.sum(1) != 0)[0]
        sub = hist[used][:, used]
        assert np.allclose(mc_eval.per_class_iu(sub), g["iu_" + tag], rtol=1e-12, equal_nan=True)
        assert np.isclose(mc_eval.calc_fw_iu(sub), float(g["fw_" + tag]), rtol=1e-12)
        assert np.isclose(mc_eval.calc_pixel_accuracy(sub), float(g["pa_" + tag]), rtol=1e-12)
        assert np.isclose(mc_eval.calc_mean_accuracy(sub), float(g["ma_" + tag]), rtol=1e-12)


FUSION_VARIANTS = [("gate", "FusionDRNSegPixelClassifier", "MFNet-GateFusion", 300, 400),
                   ("scoregate", "ScoreFusionDRNSegPixelClassifier", "MFNet-ScoreGateFusion", 301, 401),
                   ("concat", "FusionDRNSegPixelClassifier", "MFNet-ConcatFusion", 302, 402),
                   ("concatconv", "FusionDRNSegPixelClassifier", "MFNet-ConcatConvFusion", 303, 403)]


def _fusion_golden():
    import os
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fusion_small.npz"))


@pytest.mark.parametrize("variant", FUSION_VARIANTS, ids=lambda v: v[0])
def test_fusion_classifiers_match_reference_vectors(variant):
    """oracle fusion classifiers (gate / score-gate / concat / concat-conv) against the reference's fp64 outputs and
    gradients; ProbCrossEntropyLoss2d on the score-gate probabilities"""
    from recipe import fusion_inputs
    tag, cls, ftype, wseed, xseed = variant
    g = _fusion_golden()
    m = fill_state_(getattr(ref_models, cls)(ftype, 41), wseed)
    assert sorted(m.state_dict().keys()) == list(g[tag + "_keys"])
    x1, x2, gy, lbl = fusion_inputs(xseed, 41)
    a, b = x1.clone().requires_grad_(), x2.clone().requires_grad_()
    y = m(a, b)
    names = [k for k, _ in m.named_parameters()]
    grads = torch.autograd.grad(y, [a, b] + [p for _, p in m.named_parameters()], gy)
    tol = lambda ref: 2e-5 * float(np.abs(ref).max()) + 1e-7  # noqa: E731
    assert np.abs(y.detach().numpy() - g[tag + "_y"]).max() <= tol(g[tag + "_y"])
    assert np.abs(grads[0].numpy() - g[tag + "_dx1"]).max() <= tol(g[tag + "_dx1"])
    assert np.abs(grads[1].numpy() - g[tag + "_dx2"]).max() <= tol(g[tag + "_dx2"])
    for k, gv in zip(names, grads[2:]):
        ref = g["%s_grad_%s" % (tag, k)]
        assert np.abs(gv.numpy() - ref).max() <= 1e-4 * float(np.abs(ref).max()) + 1e-7, k
    if tag == "scoregate":
        w = ref_loss.class_weights(41)
        p = torch.from_numpy(g["scoregate_y"]).requires_grad_()
        for sfx, sa in (("mean", True), ("sum", False)):
            val = ref_loss.ProbCrossEntropyLoss2d(w.double(), sa)(p, torch.from_numpy(g["probce_lbl"]))
            (gp,) = torch.autograd.grad(val, [p])
            assert abs(float(val) - float(g["probce_" + sfx])) <= 1e-12 * abs(float(val))
            assert np.allclose(gp.numpy(), g["probce_grad_" + sfx], rtol=1e-12, atol=0)
