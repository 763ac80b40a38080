#!/usr/bin/env python3
"""Generate the golden vectors by running the REAL reference (build container only).

    python tests/golden/make_golden.py            # needs /root/reference

The reference is Python 2 / torch 0.4 source.  It is never copied: three of its files are read
as text, passed through lib2to3 *in memory* (plus the one ``async=True`` keyword rename) and
exec'd into fresh module objects; ``torchvision`` (absent here, unused by the hot classes) is
stubbed.  Every fixture is data: recipe seeds, the reference's outputs,
and checksums.  While generating, each oracle function is asserted against the reference --
the fixtures then let ``tests/test_oracle_golden.py`` re-check the oracle anywhere.
"""
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from recipe import checksum, fill_state_, make_batch, state_checksums  # noqa: E402


def load_reference():
    """Import loss, models.drn, models.dilated_fcn, models.fusion, models.model_util from REF."""
    from lib2to3 import refactor
    tool = refactor.RefactoringTool(refactor.get_fixers_from_package("lib2to3.fixes"))
    for name in ("torchvision", "torchvision.transforms"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    sys.path.insert(0, REF)
    import models as ref_models_pkg  # the reference's (empty) models/__init__.py
    assert ref_models_pkg.__file__.startswith(REF)

    def shim(modname, relpath, package=None, patch=None):
        with open(os.path.join(REF, relpath)) as fh:
            src = fh.read()
        if patch:
            src = patch(src)
        src = str(tool.refactor_string(src + "\n", os.path.join(REF, relpath)))
        mod = types.ModuleType(modname)
        mod.__file__ = os.path.join(REF, relpath)
        if package:
            mod.__package__ = package
        sys.modules[modname] = mod
        exec(compile(src, mod.__file__, "exec"), mod.__dict__)
        return mod

    ref_loss = shim("loss", "loss.py")
    drn = shim("models.drn", "models/drn.py", package="models")
    ref_models_pkg.drn = drn
    sys.modules.setdefault("drn", drn)  # `import drn` (py2 implicit relative import, dilated_fcn.py:23)
    dfcn = shim("models.dilated_fcn", "models/dilated_fcn.py", package="models",
                patch=lambda s: s.replace("async=True", "non_blocking=True"))
    ref_models_pkg.dilated_fcn = dfcn
    import models.model_util as ref_mu  # py3-clean, imported unmodified
    import models.fusion as ref_fusion
    return ref_loss, drn, dfcn, ref_fusion, ref_mu


def ref_get_models(ref_mu, dfcn, net, input_ch, n_class, method):
    """The reference factory hard-wires pretrained=True (a download); build the same classes with
    pretrained=False -- identical module trees (models/model_util.py:187-204, 225-251)."""
    ver = "ver2" if "ver2" in net else "ver1"
    drn_name = net.replace("_ver2", "")
    if method == "MCD":
        g = dfcn.DRNSegBase(drn_name, n_class, pretrained=False, input_ch=input_ch, ver=ver)
        return [g, dfcn.DRNSegPixelClassifier(n_class=n_class, ver=ver), dfcn.DRNSegPixelClassifier(n_class=n_class, ver=ver)]
    fusion_type = method.split("-")[-1]
    g3 = dfcn.DRNSegBase(drn_name, n_class, pretrained=False, input_ch=3, ver=ver)
    g1 = dfcn.DRNSegBase(drn_name, n_class, pretrained=False, input_ch=input_ch - 3, ver=ver)
    if "score" in method.lower():
        fs = [dfcn.ScoreFusionDRNSegPixelClassifier(fusion_type=fusion_type, n_class=n_class) for _ in range(2)]
    else:
        fs = [dfcn.FusionDRNSegPixelClassifier(fusion_type=fusion_type, n_class=n_class, ver=ver) for _ in range(2)]
    return [g3, g1] + fs


def keys_shapes(m):
    return {k: list(v.shape) for k, v in m.state_dict().items()}


def close(a, b, tol, what):
    err = float((a.double() - b.double()).abs().max())
    scale = float(b.double().abs().max()) + 1e-30
    assert err <= tol * max(1.0, scale), "%s: oracle vs reference max err %.3e (scale %.3e)" % (what, err, scale)
    return err


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    ref_loss, ref_drn, dfcn, ref_fusion, ref_mu = load_reference()
    from oracle import ref_loss as o_loss, ref_mcd as o_mcd, ref_models as o_models

    NC = 41
    out = {}

    # ---------------------------------------------------------------- A. state_dict layouts
    ks = {}
    ref_mcd_models = ref_get_models(ref_mu, dfcn, "drn_d_38", 6, NC, "MCD")
    ks["MCD/drn_d_38/6ch/G"] = keys_shapes(ref_mcd_models[0])
    ks["MCD/drn_d_38/6ch/F"] = keys_shapes(ref_mcd_models[1])
    r = ref_get_models(ref_mu, dfcn, "drn_d_38", 6, NC, "MFNet-ScoreAddFusion")
    ks["MFNet-ScoreAddFusion/drn_d_38/6ch/G_3ch"] = keys_shapes(r[0])
    ks["MFNet-ScoreAddFusion/drn_d_38/6ch/G_1ch"] = keys_shapes(r[1])
    ks["MFNet-ScoreAddFusion/drn_d_38/6ch/F"] = keys_shapes(r[2])
    r = ref_get_models(ref_mu, dfcn, "drn_d_38", 6, NC, "MFNet-AddFusion")
    ks["MFNet-AddFusion/drn_d_38/6ch/F"] = keys_shapes(r[2])
    r = ref_get_models(ref_mu, dfcn, "drn_d_38_ver2", 6, NC, "MCD")
    ks["MCD/drn_d_38_ver2/6ch/G"] = keys_shapes(r[0])
    ks["MCD/drn_d_38_ver2/6ch/F"] = keys_shapes(r[1])
    r = ref_get_models(ref_mu, dfcn, "drn_d_105", 6, NC, "MCD")
    ks["MCD/drn_d_105/6ch/G"] = keys_shapes(r[0])
    full = torch.nn.DataParallel(dfcn.DRNSeg("drn_d_38", NC, input_ch=6, pretrained=False))
    ks["full/drn_d_38/6ch/DataParallel"] = keys_shapes(full)
    for tag, make in (("MCD/drn_d_38/6ch/G", lambda: o_models.get_models("drn_d_38", 6, NC)[0]),
                      ("MCD/drn_d_105/6ch/G", lambda: o_models.get_models("drn_d_105", 6, NC)[0]),
                      ("MCD/drn_d_38_ver2/6ch/G", lambda: o_models.get_models("drn_d_38_ver2", 6, NC)[0]),
                      ("MCD/drn_d_38_ver2/6ch/F", lambda: o_models.get_models("drn_d_38_ver2", 6, NC)[1]),
                      ("MFNet-ScoreAddFusion/drn_d_38/6ch/F",
                       lambda: o_models.get_models("drn_d_38", 6, NC, method="MFNet-ScoreAddFusion")[2]),
                      ("full/drn_d_38/6ch/DataParallel", lambda: o_models.get_full_model("drn_d_38", "50", NC, 6))):
        assert keys_shapes(make()) == ks[tag], tag
    with open(os.path.join(HERE, "keys_shapes.json"), "w") as fh:
        json.dump(ks, fh, indent=0, sort_keys=True)
    print("keys_shapes.json: %d layouts" % len(ks))

    # first-conv surgery (models/drn.py:256-299): 6-ch kernel = RGB kernel + its first 3 slices again
    torch.manual_seed(5)
    m6 = ref_drn.drn_d_38(pretrained=False, input_ch=6)
    w6 = list(m6.layer0.children())[0].weight.data
    assert torch.equal(w6[:, 3:6], w6[:, 0:3])
    assert torch.equal(o_models.widen_first_conv(w6[:, :3], 6), w6)
    m4 = ref_drn.drn_d_38(pretrained=False, input_ch=4)
    w4 = list(m4.layer0.children())[0].weight.data
    assert torch.equal(o_models.widen_first_conv(w4[:, :3], 4), w4)
    m1 = ref_drn.drn_d_38(pretrained=False, input_ch=1)
    assert list(list(m1.layer0.children())[0].weight.shape) == [16, 1, 7, 7]
    # He-normal statistics of the reference initialiser (models/drn.py:163-169)
    wmid = m6.layer6[1].conv1.weight.data
    init_stats = {"layer6.1.conv1.std": float(wmid.std()), "expected": float(np.sqrt(2.0 / (9 * 512)))}

    # ---------------------------------------------------------------- B. forward, small
    H, W, N = 64, 96, 2
    rg, rf1, rf2 = ref_mcd_models
    og, of1, of2 = o_models.get_models("drn_d_38", 6, NC)
    for a, b, seed in ((rg, og, 11), (rf1, of1, 12), (rf2, of2, 13)):
        fill_state_(a, seed)
        fill_state_(b, seed)
        for (ka, va), (kb, vb) in zip(a.state_dict().items(), b.state_dict().items()):
            assert ka == kb and torch.equal(va, vb)
    src, lbl, tgt = make_batch(21, N, 6, H, W, NC)
    fwd = {}
    for mode in ("train", "eval"):
        for m in (rg, rf1, rf2, og, of1, of2):
            m.train() if mode == "train" else m.eval()
        fill_state_(rg, 11), fill_state_(og, 11)
        with torch.no_grad():
            rfeat = rg(src); r1 = rf1(rfeat); r2 = rf2(rfeat)
            ofeat = og(src); o1 = of1(ofeat); o2 = of2(ofeat)
        close(ofeat, rfeat, 1e-5, "feat/" + mode), close(o1, r1, 1e-5, "logits1/" + mode)
        fwd["feat_" + mode] = rfeat.numpy()
        fwd["logits1_sub_" + mode] = r1[:, :, ::4, ::4].numpy().copy()
        fwd["logits1_cs_" + mode] = np.array(checksum(r1))
        fwd["logits2_cs_" + mode] = np.array(checksum(r2))
        fwd["argmax1_" + mode] = r1[:, :NC - 1].argmax(1).to(torch.uint8).numpy()  # adapt_tester.py:121-124
        top2 = r1[:, :NC - 1].topk(2, dim=1).values
        fwd["margin1_" + mode] = (top2[:, 0] - top2[:, 1]).numpy()
        if mode == "train":
            sd = rg.state_dict()
            for k in ("base.0.1.running_mean", "base.0.1.running_var", "base.5.0.bn1.running_mean",
                      "base.5.0.bn1.running_var", "base.8.1.running_var", "base.6.0.downsample.1.running_mean"):
                fwd["rs/" + k] = sd[k].numpy().copy()
                close(og.state_dict()[k], sd[k], 1e-5, k)
            fwd["nbt"] = np.array(int(sd["base.8.1.num_batches_tracked"]))
    np.savez_compressed(os.path.join(HERE, "fwd_small.npz"), **fwd)
    print("fwd_small.npz")

    # ---------------------------------------------------------------- C. losses
    rs = np.random.RandomState(31)
    z1 = torch.from_numpy((3.0 * rs.standard_normal((2, NC, 12, 16))).astype(np.float32)).requires_grad_()
    z2 = torch.from_numpy((3.0 * rs.standard_normal((2, NC, 12, 16))).astype(np.float32)).requires_grad_()
    y = torch.from_numpy(rs.randint(0, NC, size=(2, 12, 16)).astype(np.int64))
    w = o_loss.class_weights(NC)
    ce_ref = ref_loss.CrossEntropyLoss2d(w)(z1, y)
    g_ce, = torch.autograd.grad(ce_ref, z1)
    d_ref = ref_loss.get_prob_distance_criterion("diff")(z1, z2)
    g_d1, g_d2 = torch.autograd.grad(d_ref, [z1, z2])
    wfull = torch.from_numpy(rs.uniform(0.5, 2.0, NC).astype(np.float32))
    ce_ref_w = ref_loss.CrossEntropyLoss2d(wfull)(z1, y)
    close(o_loss.CrossEntropyLoss2d(w)(z1, y), ce_ref, 1e-6, "CE")
    close(o_loss.CrossEntropyLoss2d(wfull)(z1, y), ce_ref_w, 1e-6, "CE(w)")
    close(o_loss.Diff2d()(z1, z2), d_ref, 1e-6, "Diff")
    l64, g64 = o_loss.ce_and_grad(z1.detach().numpy(), y.numpy(), w.numpy())
    assert abs(l64 - float(ce_ref)) < 1e-5 and np.abs(g64 - g_ce.numpy()).max() < 1e-7
    d64, gd1, gd2 = o_loss.diff_and_grad(z1.detach().numpy(), z2.detach().numpy())
    assert abs(d64 - float(d_ref)) < 1e-7 and np.abs(gd1 - g_d1.numpy()).max() < 1e-8
    assert np.abs(gd2 - g_d2.numpy()).max() < 1e-8
    np.savez_compressed(os.path.join(HERE, "loss_small.npz"), z1=z1.detach().numpy(), z2=z2.detach().numpy(),
                        y=y.numpy(), w=w.numpy(), wfull=wfull.numpy(), ce=float(ce_ref), ce_w=float(ce_ref_w),
                        g_ce=g_ce.numpy(), diff=float(d_ref), g_d1=g_d1.numpy(), g_d2=g_d2.numpy())
    print("loss_small.npz  CE %.6f Diff %.6f" % (float(ce_ref), float(d_ref)))

    # ---------------------------------------------------------------- D. backward, small (fp32 + fp64)
    crit = ref_loss.CrossEntropyLoss2d(w)
    critd = ref_loss.get_prob_distance_criterion("diff")
    picks = ["base.0.0.weight", "base.1.0.weight", "base.3.0.conv1.weight", "base.3.0.downsample.0.weight",
             "base.5.0.conv1.weight", "base.6.2.conv2.weight", "base.8.0.weight", "seg.weight", "seg.bias"]
    bwd = {}

    def run_bwd(models, dtype, which):
        g, f1, f2 = models
        for m, seed in ((g, 11), (f1, 12), (f2, 13)):
            fill_state_(m, seed)
            m.train()
            m.to(dtype)
            m.zero_grad()
        x = (src if which != "diff" else tgt).to(dtype)
        feat = g(x)
        a, b = f1(feat), f2(feat)
        cw = w.to(dtype)
        if which == "ce":
            loss = type(crit)(cw)(a, lbl) + type(crit)(cw)(b, lbl)
        else:
            loss = critd(a, b)
        loss.backward()
        res = {"loss": float(loss)}
        named = dict(g.named_parameters())
        for k in picks:
            gr = named[k].grad
            res[k] = gr if gr.numel() <= 40000 else gr.reshape(gr.shape[0], -1)[:16, :288]
        res["bn_gamma_all"] = torch.cat([p.grad.reshape(-1) for k, p in named.items() if p.dim() == 1 and k.endswith("weight")])
        res["bn_beta_all"] = torch.cat([p.grad.reshape(-1) for k, p in named.items() if p.dim() == 1 and k.endswith("bias") and not k.startswith("seg")])
        res["up1"] = f1.up.weight.grad
        res["up2"] = f2.up.weight.grad
        res["cs"] = {k: checksum(p.grad) for k, p in named.items()}
        return res

    for which in ("ce", "diff"):
        r32 = run_bwd(ref_mcd_models, torch.float32, which)
        o32 = run_bwd((og, of1, of2), torch.float32, which)
        r64 = run_bwd(ref_mcd_models, torch.float64, which)
        for k in r32:
            if k in ("loss", "cs"):
                continue
            # oracle must sit inside the reference's own fp32 noise floor
            noise = float((r32[k].double() - r64[k]).abs().max())
            err = float((o32[k].double() - r64[k]).abs().max())
            scale = float(r64[k].abs().max())
            assert err <= max(4 * noise, 1e-5 * scale, 1e-9), (which, k, err, noise, scale)
            bwd["%s/f32/%s" % (which, k)] = r32[k].numpy()
            bwd["%s/f64/%s" % (which, k)] = r64[k].numpy()
        bwd[which + "/loss32"] = np.array(r32["loss"])
        bwd[which + "/loss64"] = np.array(r64["loss"])
        out["bwd_cs_" + which] = r64["cs"]
        out["bwd_cs32_" + which] = r32["cs"]
    for m in list(ref_mcd_models) + [og, of1, of2]:
        m.float()
    np.savez_compressed(os.path.join(HERE, "bwd_small.npz"), **bwd)
    print("bwd_small.npz")

    # ---------------------------------------------------------------- E. three-step traces
    def mcd_trace(h, wd, iters, seed_batch):
        for m, seed in ((rg, 11), (rf1, 12), (rf2, 13), (og, 11), (of1, 12), (of2, 13)):
            fill_state_(m, seed)
            m.train()
        s, l, t = make_batch(seed_batch, N, 6, h, wd, NC)
        r_og = ref_mu.get_optimizer(rg.parameters(), "sgd", 1e-3, 0.9, 2e-5)
        r_of = ref_mu.get_optimizer(list(rf1.parameters()) + list(rf2.parameters()), "sgd", 1e-3, 0.9, 2e-5)
        o_og = o_models.get_optimizer(og.parameters(), "sgd", 1e-3, 0.9, 2e-5)
        o_of = o_models.get_optimizer(list(of1.parameters()) + list(of2.parameters()), "sgd", 1e-3, 0.9, 2e-5)
        tr = {"shape": [N, 6, h, wd], "seed_batch": seed_batch, "iters": []}
        for it in range(iters):
            # the reference loop, statement for statement (adapt_trainer.py:155-214), on the reference modules
            r_og.zero_grad(); r_of.zero_grad()
            f = rg(s); loss = crit(rf1(f), l) + crit(rf2(f), l); loss.backward(); c_loss = float(loss)
            r_og.step(); r_of.step()
            r_og.zero_grad(); r_of.zero_grad()
            f = rg(s); loss = crit(rf1(f), l) + crit(rf2(f), l)
            f = rg(t); loss = loss - critd(rf1(f), rf2(f)); loss.backward(); r_of.step()
            for _ in range(4):
                r_og.zero_grad()
                f = rg(t); loss = critd(rf1(f), rf2(f)) * 1; loss.backward(); r_og.step()
            d_loss = float(loss) / 4
            oc, od = o_mcd.mcd_step(og, of1, of2, o_og, o_of, o_loss.CrossEntropyLoss2d(w), o_loss.Diff2d(), s, l, t)
            assert abs(oc - c_loss) < 1e-4 * abs(c_loss) and abs(od - d_loss) < 1e-3 * abs(d_loss) + 1e-7, (oc, c_loss, od, d_loss)
            tr["iters"].append({"c_loss": c_loss, "d_loss": d_loss})
            print("  mcd %dx%d iter %d: c_loss %.6f d_loss %.8f (oracle %.6f %.8f)" % (h, wd, it, c_loss, d_loss, oc, od))
        tr["g"] = state_checksums(rg)
        tr["f1"] = state_checksums(rf1)
        tr["f2"] = state_checksums(rf2)
        tr["mom_g_first"] = checksum(r_og.state[next(iter(rg.parameters()))]["momentum_buffer"])
        tr["nbt"] = int(rg.state_dict()["base.0.1.num_batches_tracked"])
        tr["opt_g_state_keys"] = sorted(r_og.state_dict().keys())
        tr["opt_g_group_keys"] = sorted(r_og.state_dict()["param_groups"][0].keys())
        return tr

    out["mcd_small"] = mcd_trace(64, 96, 2, 41)
    out["mcd_240x320"] = mcd_trace(240, 320, 1, 42)

    # ---------------------------------------------------------------- F. MFNet-ScoreAddFusion
    rm = ref_get_models(ref_mu, dfcn, "drn_d_38", 6, NC, "MFNet-ScoreAddFusion")
    om = o_models.get_models("drn_d_38", 6, NC, method="MFNet-ScoreAddFusion")
    for i, seed in enumerate((51, 52, 53, 54)):
        fill_state_(rm[i], seed), fill_state_(om[i], seed)
        rm[i].train(), om[i].train()
    s, l, t = make_batch(43, N, 6, H, W, NC)
    with torch.no_grad():
        ra, rb = rm[0](s[:, :3]), rm[1](s[:, 3:])
        ro = rm[2](ra, rb)
        oo = om[2](om[0](s[:, :3]), om[1](s[:, 3:]))
    close(oo, ro, 1e-5, "mfnet logits")
    mf = {"feat_rgb": ra.numpy(), "feat_hha": rb.numpy(), "logits1_sub": ro[:, :, ::4, ::4].numpy().copy(),
          "logits1_cs": np.array(checksum(ro))}
    np.savez_compressed(os.path.join(HERE, "mfnet_small.npz"), **mf)
    for i, seed in enumerate((51, 52, 53, 54)):
        fill_state_(rm[i], seed), fill_state_(om[i], seed)
    r_og = ref_mu.get_optimizer(list(rm[0].parameters()) + list(rm[1].parameters()), "sgd", 1e-3, 0.9, 2e-5)
    r_of = ref_mu.get_optimizer(list(rm[2].parameters()) + list(rm[3].parameters()), "sgd", 1e-3, 0.9, 2e-5)
    o_og = o_models.get_optimizer(list(om[0].parameters()) + list(om[1].parameters()), "sgd", 1e-3, 0.9, 2e-5)
    o_of = o_models.get_optimizer(list(om[2].parameters()) + list(om[3].parameters()), "sgd", 1e-3, 0.9, 2e-5)

    def rheads(x):
        a, b = rm[0](x[:, :3, :, :]), rm[1](x[:, 3:, :, :])
        return rm[2](a, b), rm[3](a, b)

    # adapt_mfnet_trainer.py:174-244 on the reference modules
    r_og.zero_grad(); r_of.zero_grad()
    a, b = rheads(s); loss = crit(a, l) + crit(b, l); loss.backward(); c_loss = float(loss); r_og.step(); r_of.step()
    r_og.zero_grad(); r_of.zero_grad()
    a, b = rheads(s); loss = crit(a, l) + crit(b, l)
    a, b = rheads(t); loss = loss - critd(a, b); loss.backward(); r_of.step(); r_of.zero_grad()
    for _ in range(4):
        r_og.zero_grad(); a, b = rheads(t); loss = critd(a, b); loss.backward(); r_og.step()
    d_loss = float(loss) / 4
    oc, od = o_mcd.mfnet_mcd_step(om[0], om[1], om[2], om[3], o_og, o_of, o_loss.CrossEntropyLoss2d(w), o_loss.Diff2d(), s, l, t)
    assert abs(oc - c_loss) < 1e-4 * abs(c_loss) and abs(od - d_loss) < 1e-3 * abs(d_loss) + 1e-7
    out["mfnet_small"] = {"shape": [N, 6, H, W], "seed_batch": 43, "c_loss": c_loss, "d_loss": d_loss,
                          "g_3ch": state_checksums(rm[0]), "g_1ch": state_checksums(rm[1]),
                          "f1": state_checksums(rm[2]), "f2": state_checksums(rm[3])}
    print("  mfnet: c_loss %.6f d_loss %.8f" % (c_loss, d_loss))

    # ---------------------------------------------------------------- G. cfg1: source-only step, DRNSeg in DataParallel
    rfull = full
    ofull = o_models.get_full_model("drn_d_38", "50", NC, 6)
    fill_state_(rfull, 61), fill_state_(ofull, 61)
    rfull.train(), ofull.train()
    s, l, _ = make_batch(44, 2, 6, 240, 320, NC)
    r_opt = ref_mu.get_optimizer(rfull.parameters(), "sgd", 1e-3, 0.9, 2e-5)
    o_opt = o_models.get_optimizer(ofull.parameters(), "sgd", 1e-3, 0.9, 2e-5)
    r_opt.zero_grad(); preds = rfull(s); loss = crit(preds, l); loss.backward(); r_opt.step()
    ol = o_mcd.source_step(ofull, o_opt, o_loss.CrossEntropyLoss2d(w), s, l)
    assert abs(ol - float(loss)) < 1e-5 * abs(float(loss))
    out["source_240x320"] = {"shape": [2, 6, 240, 320], "seed_batch": 44, "loss": float(loss),
                             "logits_cs": checksum(preds), "state": state_checksums(rfull)}
    print("  source: loss %.6f" % float(loss))

    # ---------------------------------------------------------------- H. drn_d_105 (Bottleneck, cfg5 trunk), small
    r105 = ref_get_models(ref_mu, dfcn, "drn_d_105", 6, NC, "MCD")
    o105 = o_models.get_models("drn_d_105", 6, NC)
    for i, seed in enumerate((71, 72, 73)):
        fill_state_(r105[i], seed), fill_state_(o105[i], seed)
        r105[i].train(), o105[i].train()
    s, l, t = make_batch(45, 2, 6, H, W, NC)
    for m in list(r105) + list(o105):
        m.zero_grad()
    rfeat = r105[0](s); rl = crit(r105[1](rfeat), l) + crit(r105[2](rfeat), l); rl.backward()
    ofeat = o105[0](s); ol = o_loss.CrossEntropyLoss2d(w)(o105[1](ofeat), l) + o_loss.CrossEntropyLoss2d(w)(o105[2](ofeat), l); ol.backward()
    close(ofeat, rfeat, 1e-5, "d105 feat")
    assert abs(float(ol) - float(rl)) < 1e-5 * float(rl)
    named = dict(r105[0].named_parameters())
    d105 = {"feat": rfeat.detach().numpy(), "loss": np.array(float(rl)),
            "g/base.0.0.weight": named["base.0.0.weight"].grad.numpy(), "g/seg.weight": named["seg.weight"].grad.numpy(),
            "g/base.5.11.conv2.weight_sub": named["base.5.11.conv2.weight"].grad.reshape(256, -1)[:16, :288].numpy(),
            "g/base.7.0.weight_sub": named["base.7.0.weight"].grad.reshape(512, -1)[:16, :288].numpy()}
    np.savez_compressed(os.path.join(HERE, "d105_small.npz"), **d105)
    out["d105_small"] = {"seed_batch": 45, "shape": [2, 6, H, W], "loss": float(rl), "grad_cs": {k: checksum(p.grad) for k, p in named.items()}}
    print("  d105: loss %.6f" % float(rl))

    # ---------------------------------------------------------------- I. multitask (cfg4): encoder + MCD multitask decoder
    from oracle import ref_multitask as o_mt
    r_enc = dfcn.MultiTaskEncoder(model_name="drn_d_38", pretrained=False, input_ch=3)
    r_dec = dfcn.MCDMultiTaskDecoder(n_class=NC, depth_ch=3, semseg_criterion=crit, discrepancy_criterion=critd)
    o_enc, o_dec = o_mt.get_multitask_models("drn_d_38", 6, NC, o_loss.CrossEntropyLoss2d(w), o_loss.Diff2d())
    ks_mt = {"enc": keys_shapes(r_enc), "dec": keys_shapes(r_dec)}
    assert keys_shapes(o_enc) == ks_mt["enc"] and keys_shapes(o_dec) == ks_mt["dec"]
    for a, b, seed in ((r_enc, o_enc, 81), (r_dec, o_dec, 82)):
        fill_state_(a, seed), fill_state_(b, seed)
        a.train(), b.train()
    s, l, t = make_batch(46, 2, 6, H, W, NC)
    with torch.no_grad():
        rf = r_enc(s[:, :3]); ra, rb, rd = r_dec(rf)
        of_ = o_enc(s[:, :3]); oa, ob, od = o_dec(of_)
    close(oa, ra, 1e-5, "mt seg1"), close(od, rd, 1e-5, "mt depth")
    mt = {"fet": rf.numpy(), "seg1_sub": ra[:, :, ::4, ::4].numpy().copy(), "seg2_cs": np.array(checksum(rb)), "dep": rd.numpy()}
    np.savez_compressed(os.path.join(HERE, "multitask_small.npz"), **mt)
    for a, b, seed in ((r_enc, o_enc, 81), (r_dec, o_dec, 82)):
        fill_state_(a, seed), fill_state_(b, seed)
    r_oe = ref_mu.get_optimizer(r_enc.parameters(), "sgd", 1e-3, 0.9, 2e-5)
    r_od = ref_mu.get_optimizer(r_dec.parameters(), "sgd", 1e-3, 0.9, 2e-5)
    o_oe = o_models.get_optimizer(o_enc.parameters(), "sgd", 1e-3, 0.9, 2e-5)
    o_od = o_models.get_optimizer(o_dec.parameters(), "sgd", 1e-3, 0.9, 2e-5)
    # adapt_multitask_trainer.py:166-239 on the reference modules
    sr, sd, tr_, td = s[:, :3, :, :], s[:, 3:, :, :], t[:, :3, :, :], t[:, 3:, :, :]
    r_oe.zero_grad(); r_od.zero_grad()
    src_fet = r_enc(sr); tgt_fet = r_enc(tr_)
    ssl, sdl = r_dec.get_loss(src_fet, l, sd, separately_returning=True)
    tdl = r_dec.get_depth_loss(tgt_fet, td)
    loss = ssl + sdl + tdl; loss.backward(); c_loss = float(loss); r_oe.step(); r_od.step()
    r_oe.zero_grad(); r_od.zero_grad()
    src_fet = r_enc(sr); r_dec.semseg_forward(src_fet)
    ssl, sdl = r_dec.get_loss(src_fet, l, sd, separately_returning=True)
    tgt_fet = r_enc(tr_); tdl = r_dec.get_depth_loss(tgt_fet, td)
    disc = r_dec.get_cls_descrepancy(tgt_fet)
    loss = ssl + sdl + tdl - disc; loss.backward(); r_od.step()
    for _ in range(4):
        r_oe.zero_grad(); tgt_fet = r_enc(tr_); disc = r_dec.get_cls_descrepancy(tgt_fet); loss = disc * 1; loss.backward(); r_oe.step()
    d_loss = float(loss) / 4
    oc, od_, parts = o_mt.multitask_mcd_step(o_enc, o_dec, o_oe, o_od, s, l, t)
    assert abs(oc - c_loss) < 1e-4 * abs(c_loss) and abs(od_ - d_loss) < 1e-3 * abs(d_loss) + 1e-7, (oc, c_loss, od_, d_loss)
    out["multitask_small"] = {"shape": [2, 6, H, W], "seed_batch": 46, "c_loss": c_loss, "d_loss": d_loss,
                              "parts": [float(ssl), float(sdl), float(tdl)], "enc": state_checksums(r_enc),
                              "dec": state_checksums(r_dec), "keys_shapes": ks_mt,
                              "nbt_enc": int(r_enc.state_dict()["base.0.1.num_batches_tracked"]),
                              "nbt_seg": int(r_dec.state_dict()["semsegcls_dec1.cbr1.bn.num_batches_tracked"]),
                              "nbt_dep": int(r_dec.state_dict()["deprgr_dec.cbr1.bn.num_batches_tracked"])}
    print("  multitask: c_loss %.6f d_loss %.8f nbt enc/seg/dep %d/%d/%d" % (c_loss, d_loss, out["multitask_small"]["nbt_enc"],
          out["multitask_small"]["nbt_seg"], out["multitask_small"]["nbt_dep"]))

    out["init_stats"] = init_stats
    out["torch_version"] = torch.__version__
    with open(os.path.join(HERE, "traces.json"), "w") as fh:
        json.dump(out, fh, indent=0, sort_keys=True)
    print("traces.json written")


if __name__ == "__main__":
    main()
