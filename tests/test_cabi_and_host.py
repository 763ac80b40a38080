"""CPU: the C-ABI library loads and exports every symbol of include/mcdseg.h; host-side logic (flags, helpers,
factory, checkpoint layout) behaves like the reference's.  No kernel is launched here."""
import argparse
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(autouse=True)
def _no_pretrained(monkeypatch):
    monkeypatch.setenv("MCDSEG_PRETRAINED", "0")


def test_no_packed_fp32_instruction_with_op_sel_in_the_library():
    """the gfx950 quirk behind round 2's red two-rank test (mcdseg/_lib.py NO_PACKED_F32): no kernel of the built library may hold a
    packed-fp32 VALU instruction that routes a source's high dword to the low lane -- checked on the disassembly, no GPU needed"""
    import mcdseg
    from mcdseg import _lib
    mcdseg.build()
    if not os.path.exists(os.path.join(os.environ.get("ROCM_LLVM_BIN", "/opt/rocm/lib/llvm/bin"), "llvm-objdump")):
        pytest.skip("llvm-objdump of the ROCm toolchain not found")
    text = _lib.device_disassembly()
    assert text.count("v_mfma_f32_32x32x16_f16") > 500, "the disassembly does not look like libmcdseg's"
    assert _lib.packed_f32_opsel_sites() == []
    # the register spills of the 256 x 256 / 256 x 128 forward tiles (BASELINE config 5's plan) sit in the prologue and
    # the epilogue -- no scratch access between the first and the last matrix instruction of any ping-pong kernel
    assert _lib.spills_inside_matrix_loops() == []
    # round 5: loads, stores and LDS-DMA share one in-order counter, and a compiler-made `s_waitcnt vmcnt(0)` between an
    # LDS-DMA issue and the stores behind it drains the prefetch every iteration -- none in the fused loss kernel's item loop (whose
    # hand-written vmcnt(63) counts on it) nor in the up-sampler's backward
    assert text.count("up8_softmax_ce_l1_dma_kernel") >= 8, "the LDS-DMA loss kernels are missing from the library"
    assert _lib.drains_inside_store_loops() == []
    # mcd_hidden_dma writes M0 behind the compiler's back -- in the kernels that use it nothing else may touch M0 and no
    # builtin LDS-DMA may sit beside it; and the loss kernel's counted vmcnt(63) needs the item's heads x C stores in the loop's text
    assert _lib.hidden_dma_hazards() == []
    counts = _lib.loss_dma_store_counts()
    assert (41, 2) in counts and all(n >= c * h for (c, h), n in counts.items()), counts


def test_fused_loss_kernel_name_follows_the_library_rule(monkeypatch, libopt):
    """ops.up8_loss_kernel_name restates csrc/loss.hip's dispatch (the launch timer and bench.py's counter lookup go by this name)"""
    from mcdseg import ops
    libopt(UP8_LOSS_DMA=1)  # (the default)
    assert ops.up8_loss_kernel_name(16, 41, 60, 80, True, True) == "up8_softmax_ce_l1_dma_kernel<41, true, true>"
    assert ops.up8_loss_kernel_name(16, 41, 60, 80, False, True) == "up8_softmax_ce_l1_dma_kernel<41, false, true>"
    assert ops.up8_loss_kernel_name(2, 14, 8, 8, True, False) == "up8_softmax_ce_l1_dma_kernel<16, true, false>"
    assert ops.up8_loss_kernel_name(2, 24, 8, 8, True, False) == "up8_softmax_ce_l1_dma_kernel<24, true, true>"
    assert ops.up8_loss_kernel_name(2, 40, 8, 8, True, False) == "up8_softmax_ce_l1_dma_kernel<48, true, false>"
    # a tensor past a 32-bit buffer resource, or the switch: the register-staged kernel
    assert ops.up8_loss_kernel_name(64, 41, 360, 640, True, True) == "up8_softmax_ce_l1_kernel<48, true>"
    libopt(UP8_LOSS_DMA=0)
    assert ops.up8_loss_kernel_name(16, 41, 60, 80, True, True) == "up8_softmax_ce_l1_kernel<48, true>"


def test_library_exports_every_header_symbol():
    import mcdseg
    from mcdseg import _lib
    mcdseg.build()
    header = open(os.path.join(ROOT, "include", "mcdseg.h")).read()
    declared = set(re.findall(r"\b(mcdseg_[a-z0-9_]+)\s*\(", header))
    declared.discard("mcdseg_conv_desc")
    assert declared, "no declarations parsed"
    L = mcdseg.lib()
    for name in sorted(declared):
        assert hasattr(L, name), "libmcdseg.so does not export %s" % name
    assert declared == set(_lib.EXPORTS), (declared ^ set(_lib.EXPORTS))
    # ... and nothing else: every extern "C" symbol of the library is declared in the header
    import subprocess
    nm = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True)
    if nm.returncode == 0:
        exported = set(re.findall(r"\b(mcdseg_[a-z0-9_]+)$", nm.stdout, re.M))
        internal = {n for n in exported if n.startswith("mcdseg_internal_") or n == "mcdseg_set_error"}
        assert exported - internal == declared, (exported - internal) ^ declared
    assert L.mcdseg_version() == 101
    assert isinstance(L.mcdseg_last_error(), bytes)
    # every source is written for gfx950 directly: no CUDA shims / dual paths
    for src in _lib.sources():
        text = open(src).read()
        assert "__HIP_PLATFORM" not in text and "cuda_runtime" not in text and "hipify" not in text.lower()


@pytest.mark.parametrize("math", ["f16x3", "bf16x6"])
def test_weight_gradient_batch_pieces_respect_the_library_limit(monkeypatch, math):
    """every piece ``ops._batch_pieces`` hands to the weight gradient must pass the library's own operand-range rule
    (mcdseg_conv_wgrad_fits = the check inside mcdseg_conv_wgrad / mcdseg_conv_split_wgrad) -- also where the weight gradient still
    gathers fp32 values with a 128-channel tile of slack although forward and data gradient run slack-free: thin 16-channel
    layers cut along N (companions dropped), bf16x6's thin layers, 1x1 shortcuts.  Host arithmetic only."""
    import ctypes
    import mcdseg
    from mcdseg import ops
    mcdseg.build()
    monkeypatch.setattr(ops, "CONV_MATH", math)
    L, mid = mcdseg.lib(), ops.MATH_ID[math]
    cases = [  # (N, Cin, H, W, Cout, k, stride, dil)
        (40, 16, 720, 1280, 16, 3, 1, 1),    # cfg5's full-resolution 16 -> 16 layer past one launch: pieces lose their companions
        (36, 16, 720, 1280, 16, 3, 1, 1),    # 35-image pieces of it failed the (N*C + 128) rule in round 3
        (32, 16, 720, 1280, 32, 3, 2, 1),    # 16 -> 32 stride 2
        (64, 16, 720, 1280, 32, 1, 2, 1),    # DRN-C's 1x1 stride-2 shortcut
        (32, 2048, 90, 160, 512, 3, 1, 2),   # cfg5's widest tensor (3.8 GB): cut with companions
        (32, 256, 180, 320, 256, 3, 1, 1),
        (16, 512, 60, 80, 512, 3, 1, 4),     # cfg2: one launch
    ]
    for n, cin, h, w, cout, k, stride, dil in cases:
        pad = dil * (k // 2)
        desc = ops.conv_desc((n, cin, h, w), (cout, cin, k, k), stride, pad, dil)
        for have_cb in (True, False):
            pieces = ops._batch_pieces(desc, wgrad_cb=have_cb)
            if have_cb and len(pieces) > 1 and cin <= 16:  # (what ops._conv_wgrad does: the thin window kernel takes whole batches only)
                have_cb, pieces = False, ops._batch_pieces(desc, wgrad_cb=False)
            assert pieces[0][0] == 0 and pieces[-1][1] == n and all(a[1] == b[0] for a, b in zip(pieces, pieces[1:]))
            for a, b in pieces:
                d = desc if (a, b) == (0, n) else ops._sub_desc(desc, b - a, n if have_cb else 0)
                assert L.mcdseg_conv_wgrad_fits(ctypes.byref(d), mid, int(have_cb)) == 1, (math, n, cin, cout, h, w, have_cb, a, b)
    # the rule itself: fp32-gathering plans need the tile of slack, companion-reading plans do not
    d = ops.conv_desc((35, 16, 720, 1280), (16, 16, 3, 3), 1, 1, 1)
    assert (35 * 16 + 128) * 720 * 1280 * 4 >= 2 ** 31 > 35 * 16 * 720 * 1280 * 4
    assert L.mcdseg_conv_wgrad_fits(ctypes.byref(d), mid, 0) == 0
    assert L.mcdseg_conv_wgrad_fits(ctypes.byref(d), 0, 0) == 0
    if math == "f16x3":
        assert L.mcdseg_conv_wgrad_variant(ctypes.byref(d), mid, 1) == 15 and L.mcdseg_conv_wgrad_fits(ctypes.byref(d), mid, 1) == 1
    else:
        assert L.mcdseg_conv_wgrad_fits(ctypes.byref(d), mid, 1) == 0  # (bf16x6 has no thin window kernel)
    # a small limit still cuts (the GPU tests force pieces that way)
    monkeypatch.setattr(ops, "MAX_CONV_BYTES", 2 * 4 * 128 * 32 * 32)
    d = ops.conv_desc((5, 128, 32, 32), (128, 128, 3, 3), 1, 1, 1)
    assert ops._batch_pieces(d, wgrad_cb=True) == ops._batch_pieces(d) == [(0, 2), (2, 4), (4, 5)]


def test_weight_gradient_plan_of_every_config2_layer():
    """Which kernel ``mcdseg_conv_split_wgrad`` launches for each convolution of drn_d_38 at BASELINE config 2 (N = 16, 6 x 480 x 640;
    SURVEY.md appendix A) when both companions are passed -- host-side arithmetic of the library (``mcdseg_conv_wgrad_variant``), so the
    dispatch table is pinned without a GPU: 15 the thin-layer window kernel, 14 the 64-wide tap pairs (from 24 channels up: round 5),
    18 the row-of-taps ping-pong kernel of the 128-channel layers (round 5), 17 the 256 x 256 ping-pong kernel, 12 the 4-wave 128 x 128
    tiles (what is left: the 1 x 1 projection 128 -> 256), 1 the f32 64 x 64 plan (the 41-channel seg head)."""
    import ctypes
    from mcdseg import ops
    L = ops.lib()
    layers = [  # (Cin, Cout, k, stride, dil, H, W) -> variant
        ((16, 16, 3, 1, 1, 480, 640), 15), ((16, 32, 3, 2, 1, 480, 640), 15),
        ((32, 64, 3, 2, 1, 240, 320), 14), ((32, 64, 1, 2, 1, 240, 320), 14), ((64, 64, 3, 1, 1, 120, 160), 14),
        ((64, 128, 3, 2, 1, 120, 160), 14), ((64, 128, 1, 2, 1, 120, 160), 14),
        ((128, 128, 3, 1, 1, 60, 80), 18), ((128, 256, 3, 1, 2, 60, 80), 18), ((128, 256, 1, 1, 1, 60, 80), 12),
        ((256, 256, 3, 1, 2, 60, 80), 17), ((256, 512, 3, 1, 4, 60, 80), 17), ((256, 512, 1, 1, 1, 60, 80), 17),
        ((512, 512, 3, 1, 4, 60, 80), 17), ((512, 512, 3, 1, 2, 60, 80), 17), ((512, 512, 3, 1, 1, 60, 80), 17),
        ((512, 41, 1, 1, 1, 60, 80), 1),
    ]
    for (cin, cout, k, s, d, h, w), want in layers:
        desc = ops.conv_desc((16, cin, h, w), (cout, cin, k, k), s, d * (k // 2), d)
        got = L.mcdseg_conv_wgrad_variant(ctypes.byref(desc), ops.MATH_ID["f16x3"], 1)
        assert got == want, ((cin, cout, k, s, d, h, w), got, want)
        # the workspace the entry point asks for covers the plan's slabs, and one launch addresses these operands
        assert L.mcdseg_conv_wgrad_workspace_bytes(ctypes.byref(desc)) > 0
        assert L.mcdseg_conv_wgrad_fits(ctypes.byref(desc), ops.MATH_ID["f16x3"], 1) == 1


def test_forward_and_data_gradient_plan_of_every_config2_layer():
    """The same for forward and data gradient (``mcdseg_conv_split_wide_pingpong`` / ``_window_ok`` / ``_direct_ok`` /
    ``_tile_config``; host-side arithmetic, 256 CUs assumed off the GPU): the 256- and 512-row problems on the 256 x 320 ping-pong tile
    (1), the 128-row ones on its 128 x 320 form (2), the stem's forward on the direct kernel, the 16 -> 16 layer on the LDS-window
    kernels, everything else on the 4-wave tiles (64 x 256 for 64 rows, 32 x 256 below)."""
    import ctypes
    from mcdseg import ops
    L, mid = ops.lib(), ops.MATH_ID["f16x3"]
    # (Cin, Cout, k, stride, dil, H, W) -> (wide forward, wide dgrad, window forward, window dgrad, direct, 4-wave tile fwd, 4-wave tile dgrad)
    layers = [
        ((6, 16, 7, 1, 1, 480, 640), (0, 0, 1, 0, 1, 1214, 1214)), ((16, 16, 3, 1, 1, 480, 640), (0, 0, 1, 1, 0, 1214, 1214)),
        ((16, 32, 3, 2, 1, 480, 640), (0, 0, 0, 0, 0, 1214, 1214)), ((32, 64, 3, 2, 1, 240, 320), (0, 0, 0, 0, 0, 2214, 1214)),
        ((64, 64, 3, 1, 1, 120, 160), (0, 0, 0, 0, 0, 2214, 2214)), ((64, 128, 3, 2, 1, 120, 160), (2, 0, 0, 0, 0, 2222, 2214)),
        ((128, 128, 3, 1, 1, 60, 80), (2, 2, 0, 0, 0, 2222, 2222)), ((128, 256, 3, 1, 2, 60, 80), (1, 2, 0, 0, 0, 2222, 2222)),
        ((256, 256, 3, 1, 2, 60, 80), (1, 1, 0, 0, 0, 2222, 2222)), ((256, 512, 3, 1, 4, 60, 80), (1, 1, 0, 0, 0, 4222, 2222)),
        ((512, 512, 3, 1, 4, 60, 80), (1, 1, 0, 0, 0, 4222, 4222)), ((512, 41, 1, 1, 1, 60, 80), (0, 0, 0, 0, 0, 2214, 4222)),
    ]
    for (cin, cout, k, s, d, h, w), want in layers:
        desc = ops.conv_desc((16, cin, h, w), (cout, cin, k, k), s, d * (k // 2), d)
        got = (L.mcdseg_conv_split_wide_pingpong(ctypes.byref(desc), mid, 1, 0), L.mcdseg_conv_split_wide_pingpong(ctypes.byref(desc), mid, 1, 1),
               L.mcdseg_conv_split_window_ok(ctypes.byref(desc), mid, 1, 0), L.mcdseg_conv_split_window_ok(ctypes.byref(desc), mid, 1, 1),
               L.mcdseg_conv_split_direct_ok(ctypes.byref(desc)), L.mcdseg_conv_split_tile_config(cout, 16 * desc.Ho * desc.Wo, 1),
               L.mcdseg_conv_split_tile_config(cin, 16 * h * w, 1))
        assert got == want, ((cin, cout, k, s, d, h, w), got, want)
        if want[0]:  # a wide tile takes the whole convolution, and its BatchNorm partial rows are one per 160 pixels
            assert L.mcdseg_conv_split_parts(ctypes.byref(desc), mid, 1, 0) == 16 * desc.Ho * desc.Wo
            assert L.mcdseg_conv_split_stat_rows_for(ctypes.byref(desc), mid, 1) == 2 * ((16 * desc.Ho * desc.Wo + 319) // 320)


def test_grad_box_protocol_and_kernel_names(tmp_path, monkeypatch):
    """Host logic of round 4 that needs no GPU: ``ops.GradBox`` (every producer of a shared gradient but the last leaves its tensor and
    reports None, the last returns what was left -- over several backward passes through one graph), the ping-pong kernel names the
    profilers key on, ``forward_fork`` off the GPU, and ``bench.is_forward_conv``'s reading of the template arguments."""
    import importlib.util
    from mcdseg import ops
    box = ops.GradBox()
    assert box.attach() is box and box.attach() is box and box.n == 2
    for _ in range(2):  # two backward passes through the same graph
        g, last = box.arrive()
        assert g is None and not last
        box.leave("first")
        g, last = box.arrive()
        assert g == "first" and last and box.g is None and box.seen == 0
    one = ops.GradBox().attach()
    assert one.arrive() == (None, True)  # a single producer returns its own gradient at once
    # a PARTIAL backward pass (autograd.grad(inputs=...), an exception mid-pass) runs only some producers: what it leaves behind must
    # not enter the next pass's sum -- passes are told apart by the engine's graph-task id
    task = [7]
    real = ops._graph_task_id
    ops._graph_task_id = lambda: task[0]
    try:
        box = ops.GradBox()
        box.attach(), box.attach()
        assert box.arrive() == (None, False)
        box.leave("stale")          # ... and the second producer of pass 7 never runs
        task[0] = 8
        assert box.arrive() == (None, False)   # first arrival of pass 8: not taken for the last one, the stale tensor is gone
        box.leave("fresh")
        assert box.arrive() == ("fresh", True)
    finally:
        ops._graph_task_id = real
    assert ops._graph_task_id() == -1  # (outside a backward pass)
    x = torch.zeros(1, requires_grad=True)
    assert isinstance(ops.grad_box(x), ops.GradBox) == ops.FUSE_RES_ADD
    with torch.no_grad():
        assert ops.grad_box(x) is None
    assert ops.grad_box(torch.zeros(1)) is None
    assert ops.forward_fork(torch.device("cpu")) is None
    names = {ops.pingpong_kernel_name(False, "f16x3"): "4, 2, 1, 4", ops.pingpong_kernel_name(True, "f16x3", small=True): "2, 2, 2, 2",
             ops.pingpong_kernel_name(False, "f16x3", wide=1): "2, 5, 2, 2", ops.pingpong_kernel_name(True, "f16x3", wide=2): "1, 5, 2, 2",
             ops.pingpong_kernel_name(False, "f16x1", wide=3): "1, 5, 4, 1"}
    for name, tile in names.items():
        assert name.startswith("conv_gemm_split_pp_kernel<Split") and name.endswith(tile + ">"), name
    spec = importlib.util.spec_from_file_location("bench_for_test", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    fwd = ["conv_gemm_split_pp_kernel<SplitF16x3, false, 2, 5, 2, 2>", "conv_gemm_split_kernel<SplitF16x3, 2, 2, 2, 2, false, true>",
           "conv_gemm_kernel<2, 2, 2, 2, 16, false>"]
    bwd = ["conv_gemm_split_pp_kernel<SplitF16x3, true, 4, 2, 1, 4>", "conv_gemm_split_kernel<SplitF16x3, 2, 2, 2, 2, true, false>",
           "conv_gemm_kernel<2, 2, 2, 2, 16, true>", "conv_stem_x6_kernel", "bn_apply_cb"]
    assert all(bench.is_forward_conv(n) for n in fwd) and not any(bench.is_forward_conv(n) for n in bwd)
    # roofline.traffic is a lookup in a committed PMC table: refused unless the table was made from THIS build's kernel sources (their
    # fingerprint is in the table) and from this set of kernels
    import glob
    import json
    import shutil
    from mcdseg import _lib
    tables = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")))
    doc = json.load(open(tables[-1]))
    table = doc["kernels"]
    name = next(n for n in table if n.startswith("conv_gemm_split_pp_kernel<SplitF16x3, false"))
    run = {n: dict(launches=v["launches_per_step"]) for n, v in table.items() if n.startswith("conv_") and "<" in n}
    fake_root = str(tmp_path)
    os.makedirs(os.path.join(fake_root, "profiles"))
    monkeypatch.setattr(bench, "ROOT", fake_root)
    json.dump(dict(doc, source_fingerprint="0" * 40), open(os.path.join(fake_root, "profiles", "r98_pmc_traffic.json"), "w"))
    got, why = bench.pmc_traffic(name, run, 1)
    assert got is None and "kernel sources changed" in why, why   # (a table of another build: the newest by name, and refused)
    json.dump(dict(doc, source_fingerprint=_lib.source_fingerprint()), open(os.path.join(fake_root, "profiles", "r97_pmc_traffic.json"), "w"))
    got, source = bench.pmc_traffic(name, run, 1)
    assert got == table[name]["hbm_bytes_per_launch"] and source.endswith("r97_pmc_traffic.json")
    got, why = bench.pmc_traffic(name, dict(run, **{"conv_renamed_kernel<1, 2>": dict(launches=3)}), 1)
    assert got is None and "stale" in why and "conv_renamed_kernel<1, 2>" in why
    got, why = bench.pmc_traffic(name, dict(run, **{name: dict(launches=run[name]["launches"] + 6)}), 1)
    assert got is None and "times per step" in why
    shutil.rmtree(os.path.join(fake_root, "profiles"))
    r = bench.kernel_roofline(name, dict(flops=3e12, bytes=1e9, ms=3.0, launches=10), "f16x3", None)
    assert abs(r["frac"] - 3 * r["frac_algorithmic"]) < 1e-3 and abs(r["frac_algorithmic"] - 1000.0 / 2500.0) < 1e-3


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "multichannel-semseg-with-uda_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, re.M), os.path.join(dirpath, f)


def test_ops_refuse_cpu_tensors():
    from mcdseg import ops
    from models.model_util import get_models
    g, f1, _ = get_models("drn_d_38", 6, 41)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        g(torch.zeros(1, 6, 16, 16))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        f1(torch.zeros(1, 41, 2, 2))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.mcd_losses(torch.zeros(1, 3, 2, 2), None, None, None)


def test_parser_defaults_match_reference():
    # argmyparse.py:35-137 of the reference
    import argmyparse
    a = argmyparse.get_da_mcd_training_parser().parse_args(["suncg", "nyu"])
    assert (a.net, a.opt, a.lr, a.momentum, a.weight_decay, a.batch_size) == ("drn_d_38", "sgd", 1e-3, 0.9, 2e-5, 1)
    assert (a.num_k, a.num_multiply_d_loss, a.d_loss, a.method, a.input_ch) == (4, 1, "diff", "MCD", 3)
    assert (a.epochs, a.max_iter, a.savename, a.base_outdir, a.background_id) == (40, 5000, "normal", "train_output", 255)
    assert not a.uses_one_classifier and not a.fix_bn and not a.adjust_lr and a.train_img_shape is None
    a = argmyparse.add_additional_params_to_args(a)
    assert a.n_class == 41 and a.train_img_shape == [640, 480]  # W, H (datasets.py:561, 1021-1022)
    s = argmyparse.get_src_only_training_parser().parse_args(["suncg", "--input_ch", "6", "-b", "2", "--train_img_shape", "320", "240"])
    assert s.train_img_shape == [320, 240] and s.split == "train"
    assert "drn_d_38_ver2" in argmyparse.AVAILABLE_NET_LIST and "drn_d_105_ver2_fusenet" in argmyparse.AVAILABLE_NET_LIST
    with pytest.raises(SystemExit):
        argmyparse.get_da_mcd_training_parser().parse_args(["suncg", "nyu", "--input_ch", "5"])


def test_dataset_helpers():
    import datasets
    assert datasets.__file__.startswith(os.path.join(ROOT, "multichannel-semseg-with-uda_amd"))
    assert [datasets.get_n_class(d) for d in ("suncg", "nyu", "gta", "city16", "sun")] == [41, 41, 20, 16, 14]
    assert datasets.get_img_shape("city", True) == [1024, 512] and datasets.get_img_shape("nyu", True) == [640, 480]
    with pytest.raises(AssertionError):
        datasets.check_src_tgt_ok("synthia", "city")
    with pytest.raises(NotImplementedError):
        datasets.get_dataset("suncg", "train", None, None, False)
    a = datasets.SyntheticRGBD(4, 6, [32, 24], 41, seed=3)
    b = datasets.SyntheticRGBD(4, 6, [32, 24], 41, seed=3)
    img, lbl = a[2]
    assert img.shape == (6, 24, 32) and lbl.shape == (24, 32) and lbl.dtype == torch.int64
    assert torch.equal(img, b[2][0]) and int(lbl.max()) <= 40 and not torch.equal(img, a[1][0])
    cat = datasets.ConcatDataset(a, datasets.SyntheticRGBD(3, 6, [32, 24], 41, seed=4))
    assert len(cat) == 3 and len(cat[0]) == 2


def test_file_list_loader(tmp_path):
    """datasets.FileListRGBD: real files (PIL) -> the raw uint8 arrays the device pipeline takes (datasets.py:20-28 of the
    reference opens files the same way; its transforms then run on the GPU here)"""
    import numpy as np
    from PIL import Image
    import datasets
    rng = np.random.RandomState(3)
    lines = []
    truth = []
    for i in range(3):
        rgb = rng.randint(0, 256, size=(12, 16, 3)).astype(np.uint8)
        hha = rng.randint(0, 256, size=(12, 16, 3)).astype(np.uint8)
        lbl = rng.randint(0, 41, size=(12, 16)).astype(np.uint8)
        lbl[0, :3] = 255
        Image.fromarray(rgb).save(tmp_path / ("rgb%d.png" % i))
        Image.fromarray(hha).save(tmp_path / ("hha%d.png" % i))
        pal = Image.fromarray(lbl, mode="P") if i == 1 else Image.fromarray(lbl)  # palette and grey label PNGs
        if i == 1:
            pal.putpalette([v for k in range(256) for v in (k, 255 - k, (7 * k) % 256)])
        pal.save(tmp_path / ("lbl%d.png" % i))
        lines.append("rgb%d.png hha%d.png lbl%d.png" % (i, i, i))
        truth.append((rgb, hha, lbl))
    (tmp_path / "train.txt").write_text("# rgb hha label\n" + "\n".join(lines) + "\n")
    ds = datasets.get_dataset("nyu", "train", None, None, False, input_ch=6, file_list=str(tmp_path / "train.txt"))
    assert len(ds) == 3
    for i, (rgb, hha, lbl) in enumerate(truth):
        img, lab = ds[i]
        assert img.dtype == torch.uint8 and tuple(img.shape) == (12, 16, 6) and tuple(lab.shape) == (12, 16)
        assert np.array_equal(img.numpy()[:, :, :3], rgb) and np.array_equal(img.numpy()[:, :, 3:], hha)
        assert np.array_equal(lab.numpy(), lbl)
    ds3 = datasets.FileListRGBD(str(tmp_path / "train.txt"), input_ch=3, test=True)
    img, lab, name = ds3[2]
    assert tuple(img.shape) == (12, 16, 3) and name.endswith("lbl2.png")
    batch = next(iter(torch.utils.data.DataLoader(ds, batch_size=2)))
    assert tuple(batch[0].shape) == (2, 12, 16, 6) and batch[0].dtype == torch.uint8
    with pytest.raises(ValueError):
        (tmp_path / "bad.txt").write_text("only_one_path.png\n")
        datasets.FileListRGBD(str(tmp_path / "bad.txt"))


def test_util_helpers(tmp_path):
    import util
    w = util.get_class_weight_from_file(41)
    assert float(w.sum()) == 40 and float(w[40]) == 0 and float(util.get_class_weight_from_file(41, add_bg_loss=True)[40]) == 1
    csv = tmp_path / "w.csv"
    csv.write_text("class_id,weight\n1,2.0\n0,0.5\n2,1.0\n")
    assert util.get_class_weight_from_file(3, str(csv)).tolist() == [0.5, 2.0, 0.0]
    opt = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=1.0)
    assert util.adjust_learning_rate(opt, 1e-3, 0.1, 0, 40) == 1e-3
    assert util.adjust_learning_rate(opt, 1e-3, 0.1, 20, 40) == pytest.approx(1e-4)
    assert util.adjust_learning_rate(opt, 1e-3, 0.1, 30, 40) == pytest.approx(1e-5) and opt.param_groups[0]["lr"] == pytest.approx(1e-5)
    fn = tmp_path / "p.json"
    util.save_dic_to_json({"a": 1, "ns": argparse.Namespace(x=1)}, str(fn), verbose=False)
    assert '"a": 1' in fn.read_text()
    util.check_if_done(str(fn))  # stdin is not a TTY under pytest: must not block


def test_factory_errors_and_quirks():
    from models import model_util
    with pytest.raises(NotImplementedError):
        model_util.get_models("fcn", 6, 41)
    with pytest.raises(NotImplementedError):
        model_util.get_full_model("psp", "50", 41, 6)
    assert isinstance(model_util.get_models("drn_d_38", 6, 41, method="DANN"), NotImplementedError)  # model_util.py:281
    with pytest.raises(NotImplementedError):
        model_util.get_optimizer([torch.nn.Parameter(torch.zeros(1))], "rmsprop", 1e-3, 0.9, 0)
    import loss
    assert loss.DiscrepancyLoss is loss.Diff2d
    assert isinstance(loss.get_prob_distance_criterion("diff"), loss.Diff2d)
    with pytest.raises(NotImplementedError):
        loss.get_prob_distance_criterion("nope")
    g = model_util.get_models("drn_d_38", 6, 41)[0]
    w = g.base[0][0].weight.data
    assert torch.equal(w[:, 3:6], w[:, :3])  # models/drn.py:285-288
    g.train()
    model_util.fix_batchnorm_when_training(g)
    assert not g.base[5][0].bn1.training and g.base[5][0].training


@pytest.mark.parametrize("name", ["jsd", "symkl", "nmlsymkl", "mysymkl", "spatial_jsd", "mis_symkl"])
def test_non_default_distances_match_reference_vectors(golden, name):
    """--d_loss other than 'diff' (loss.py:192-210 of the reference): plain-torch criteria, value and d/dlogits against
    vectors produced by the reference's own classes (tests/golden/make_golden_dist.py)"""
    import numpy as np
    import loss
    fx = golden.npz("dist_small.npz")
    crit = loss.get_prob_distance_criterion(name, n_class=int(fx["n_class"]))
    a = torch.from_numpy(fx["z1"]).requires_grad_()
    b = torch.from_numpy(fx["z2"]).requires_grad_()
    val = crit(a, b)
    ga, gb = torch.autograd.grad(val, [a, b])
    assert abs(float(val) - float(fx[name + "_val"])) <= 1e-12 * abs(float(fx[name + "_val"]))
    assert np.abs(ga.numpy() - fx[name + "_g1"]).max() <= 1e-12 and np.abs(gb.numpy() - fx[name + "_g2"]).max() <= 1e-12


def test_missing_pretrained_weights_are_an_error(monkeypatch, tmp_path):
    """the reference's factories default to pretrained=True (a download, models/drn.py:8-18); without the weights on disk the
    drop-in must not silently train from scratch"""
    from models import drn
    monkeypatch.setenv("MCDSEG_PRETRAINED", "1")
    monkeypatch.setenv("MCDSEG_PRETRAINED_DIR", str(tmp_path))
    with pytest.raises(FileNotFoundError, match="MCDSEG_PRETRAINED"):
        drn.drn_d_22(pretrained=True)
    monkeypatch.setenv("MCDSEG_PRETRAINED", "0")
    drn.drn_d_22(pretrained=True)  # explicit opt-out: He-normal initialisation


def test_checkpoint_layout_roundtrips_with_torch_modules(golden, tmp_path):
    """state_dicts and the optimizer state use the reference's layout: they load into plain-torch modules of the
    same architecture (the oracle, pinned to the reference's key set) and back."""
    import util
    from models.model_util import get_models, get_optimizer
    from oracle import ref_models
    g, f1, f2 = get_models("drn_d_38", 6, 41)
    ks = golden.json("keys_shapes.json")
    assert {k: list(v.shape) for k, v in g.state_dict().items()} == ks["MCD/drn_d_38/6ch/G"]
    og = get_optimizer(g.parameters(), "sgd", 1e-3, 0.9, 2e-5)
    args = argparse.Namespace(net="drn_d_38", uses_one_classifier=False)
    save_dic = {"epoch": 1, "args": args, "g_state_dict": g.state_dict(), "f1_state_dict": f1.state_dict(),
                "f2_state_dict": f2.state_dict(), "optimizer_g": og.state_dict(), "optimizer_f": og.state_dict()}
    fn = str(tmp_path / "MCD-normal-drn_d_38-1.pth.tar")
    util.save_checkpoint(save_dic, False, fn)
    ck = util.load_checkpoint(fn)
    assert sorted(ck.keys()) == ["args", "epoch", "f1_state_dict", "f2_state_dict", "g_state_dict", "optimizer_f", "optimizer_g"]
    rg, rf1, _ = ref_models.get_models("drn_d_38", 6, 41)
    rg.load_state_dict(ck["g_state_dict"])  # strict
    rf1.load_state_dict(ck["f1_state_dict"])
    g.load_state_dict(rg.state_dict())
    ref_opt = ref_models.get_optimizer(rg.parameters(), "sgd", 1e-3, 0.9, 2e-5)
    ref_opt.load_state_dict(ck["optimizer_g"])
    assert ref_opt.state_dict()["param_groups"][0]["momentum"] == 0.9
    og.load_state_dict(ref_opt.state_dict())


def test_drn_c_generator_has_the_reference_state_dict_layout(golden):
    """``--net drn_c_26``: conv1 / bn1 / relu are top-level children of the trunk (models/drn.py:118-121), so the keys are
    base.0.weight, base.1.*, base.3.0.conv1.weight ... -- pinned to the reference's own key list (make_golden_drnc.py)"""
    import json
    from models.model_util import get_models
    want = json.loads(str(golden.npz("drnc_small.npz")["keys"]))
    g = get_models("drn_c_26", 6, 41)[0]
    assert [[k, list(v.shape)] for k, v in g.state_dict().items()] == want


def test_library_reads_no_environment_variable():
    """no hidden inputs.  Plan selection is an explicit table behind mcdseg_set_option (csrc/options.h); neither the
    sources nor the built library refer to getenv, and the table round-trips through the ABI."""
    import subprocess
    import mcdseg
    from mcdseg import _lib
    for src in _lib.sources() + [os.path.join(_lib.CSRC, h) for h in os.listdir(_lib.CSRC) if h.endswith(".h")]:
        assert "getenv" not in open(src).read(), src
    syms = subprocess.run(["nm", "-D", "--undefined-only", mcdseg.build()], capture_output=True, text=True, check=True).stdout
    assert "getenv" not in syms
    names = mcdseg.option_names()
    assert len(names) == len(set(names)) >= 25 and "PINGPONG" in names and "PP_CUS" in names
    for nm in names:
        assert mcdseg.get_option(nm) == mcdseg.option_default(nm), "a test left %s set" % nm
    with mcdseg.options(PP_CUS=16, PINGPONG=0):
        assert (mcdseg.get_option("PP_CUS"), mcdseg.get_option("MCDSEG_PINGPONG")) == (16, 0)
    assert (mcdseg.get_option("PP_CUS"), mcdseg.get_option("PINGPONG")) == (0, 3)
    with pytest.raises(RuntimeError, match="unknown option"):
        mcdseg.set_option("NO_SUCH_OPTION", 1)


def test_two_byte_chain_host_rules():
    """Round 6, no GPU: which convolutions of BASELINE config 5's network the 2-byte chain takes (``mcdseg_conv_split_half_ok``), which of
    them run with two K-steps per barrier interval (``mcdseg_conv_split_pp_deep``), how their batches are cut, and how the stand-in
    tensors announce themselves to autograd."""
    import ctypes
    from mcdseg import ops
    L = ops.lib()

    def d(n, cin, h, w, cout, k, s=1, dil=1):
        return ops.conv_desc((n, cin, h, w), (cout, cin, k, k), s, dil * (k // 2), dil)
    # the chain: both channel counts multiples of 8, at least 16 contraction channels, the one-term arithmetic only
    for desc, fwd, bwd in ((d(32, 1024, 90, 160, 256, 1), 1, 1), (d(32, 256, 90, 160, 256, 3, dil=2), 1, 1), (d(32, 32, 360, 640, 256, 1, s=2), 1, 1),
                           (d(32, 16, 720, 1280, 16, 3), 0, 0),          # the thin layers' window kernels: not the chain's
                           (d(32, 16, 720, 1280, 32, 3, s=2), 1, 1),     # (the kernels could; ops keeps Cin <= 16 out: its weight gradient multiplies both pieces)
                           (d(32, 6, 720, 1280, 16, 7), 0, 0),           # the stem
                           (d(2, 512, 60, 80, 41, 1), 0, 0)):            # 41 channels: no unit layout
        assert L.mcdseg_conv_split_half_ok(ctypes.byref(desc), 1, 0) == fwd and L.mcdseg_conv_split_half_ok(ctypes.byref(desc), 1, 1) == bwd, tuple(desc.__getattribute__(f) for f in ("Cin", "Cout", "KH"))
        assert L.mcdseg_conv_split_half_ok(ctypes.byref(desc), 3, 0) == 0  # never in the fp32-grade arithmetic
    # two K-steps per interval: an even number of K-steps, the one-term arithmetic
    assert L.mcdseg_conv_split_pp_deep(ctypes.byref(d(32, 1024, 90, 160, 256, 1)), 1, 0) == 1
    assert L.mcdseg_conv_split_pp_deep(ctypes.byref(d(4, 48, 128, 129, 256, 1)), 1, 0) == 0      # 3 K-steps
    assert L.mcdseg_conv_split_pp_deep(ctypes.byref(d(32, 1024, 90, 160, 256, 1)), 3, 0) == 0
    assert ops.pingpong_kernel_name(False, "f16x1", deep=True) == "conv_gemm_split_pp_kernel<SplitF16x1D, false, 4, 2, 1, 4>"
    assert ops.pingpong_kernel_name(True, "f16x1", small=True, deep=True) == "conv_gemm_split_pp_kernel<SplitF16x1, true, 2, 2, 2, 2>"
    assert ops.pingpong_kernel_name(False, "f16x3", deep=True) == "conv_gemm_split_pp_kernel<SplitF16x3, false, 4, 2, 1, 4>"
    # batch cuts: config 5's 2048-channel maps at N = 32 are 3.8 GB in fp32 (two launches) and 1.9 GB as 16-bit units (one)
    big = d(32, 2048, 90, 160, 512, 1)
    assert len(ops._batch_pieces(big)) == 2 and ops._batch_pieces_half(big) == [(0, 32)]
    assert ops._batch_pieces_half(d(64, 2048, 90, 160, 512, 1)) == [(0, 35), (35, 64)]
    # the stand-in of a chain activation: a bfloat16 tensor of the logical shape with 2 bytes of storage
    v = ops._virtual((2, 64, 8, 8), torch.device("cpu"), torch.bfloat16)
    v._mcd_virtual = True
    assert ops.is_half(v) and ops.is_virtual(v) and v.shape == (2, 64, 8, 8) and v.untyped_storage().nbytes() == 2
    f = ops._virtual((2, 64, 8, 8), torch.device("cpu"))
    f._mcd_virtual = True
    assert not ops.is_half(f) and not ops.is_half(torch.zeros(2, dtype=torch.bfloat16))


def test_comm_entry_points_validate_their_arguments():
    """include/mcdseg.h "Data parallelism": the argument checks of the communicator entry points happen before RCCL is looked up (no GPU
    here): a null communicator, a null buffer, a rank outside the job are -EINVAL with a message; destroying nothing is a no-op"""
    import ctypes
    from mcdseg._lib import lib
    L = lib()
    buf = (ctypes.c_float * 4)()
    assert L.mcdseg_allreduce(buf, 4, None, None) == -22 and b"null communicator" in L.mcdseg_last_error()
    comm = ctypes.c_void_p()
    ident = (ctypes.c_ubyte * 128)()
    assert L.mcdseg_comm_init(None, 1, ident, 0) == -22
    assert L.mcdseg_comm_init(ctypes.byref(comm), 2, ident, 2) == -22 and b"rank 2 of 2" in L.mcdseg_last_error()
    assert L.mcdseg_comm_init(ctypes.byref(comm), 0, ident, 0) == -22
    assert L.mcdseg_comm_unique_id(None) == -22
    assert L.mcdseg_comm_destroy(None) == 0
