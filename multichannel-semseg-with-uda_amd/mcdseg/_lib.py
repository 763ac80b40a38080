"""ctypes loader / builder for libmcdseg.so (C ABI: include/mcdseg.h)."""
import ctypes
import glob
import os
import subprocess
import threading

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
REPO = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
INCLUDE = os.path.join(REPO, "include")
# MCDSEG_LIB: load another build of the library (kernel development: A/B variants built with tools/build_variant.py)
LIB_PATH = os.environ.get("MCDSEG_LIB") or os.path.join(HERE, "libmcdseg.so")

c_void_p, c_int, c_i32, c_i64, c_float, c_size_t = (ctypes.c_void_p, ctypes.c_int, ctypes.c_int32, ctypes.c_int64,
                                                    ctypes.c_float, ctypes.c_size_t)


class ConvDesc(ctypes.Structure):
    """mirror of ``mcdseg_conv_desc``"""
    _fields_ = [(n, c_i32) for n in ("N", "Cin", "H", "W", "Cout", "KH", "KW", "stride", "pad", "dil", "Ho", "Wo", "Ncb")]


_P = ctypes.POINTER
_SIGNATURES = {
    # name: (restype, argtypes)   -- one entry per symbol declared in include/mcdseg.h
    "mcdseg_version": (c_int, []),
    "mcdseg_last_error": (ctypes.c_char_p, []),
    "mcdseg_option_count": (c_i32, []),
    "mcdseg_option_name": (ctypes.c_char_p, [c_i32]),
    "mcdseg_set_option": (c_int, [ctypes.c_char_p, c_i64]),
    "mcdseg_get_option": (c_int, [ctypes.c_char_p, _P(c_i64), _P(c_i64)]),
    "mcdseg_comm_unique_id": (c_int, [c_void_p]),
    "mcdseg_comm_init": (c_int, [_P(c_void_p), c_i32, c_void_p, c_i32]),
    "mcdseg_comm_destroy": (c_int, [c_void_p]),
    "mcdseg_allreduce": (c_int, [c_void_p, c_i64, c_void_p, c_void_p]),
    "mcdseg_conv_packed_dims": (c_int, [_P(ConvDesc), _P(c_i32), _P(c_i32), _P(c_i32), _P(c_i32)]),
    "mcdseg_conv_pack_weights": (c_int, [_P(ConvDesc), c_void_p, c_void_p, c_void_p, c_void_p]),
    "mcdseg_conv_stat_rows": (c_i64, [_P(ConvDesc)]),
    "mcdseg_conv_split_stat_rows": (c_i64, [_P(ConvDesc)]),
    "mcdseg_conv_split_stat_rows_for": (c_i64, [_P(ConvDesc), c_i32, c_i32]),
    "mcdseg_conv_split_window_ok": (c_i32, [_P(ConvDesc), c_i32, c_i32, c_i32]),
    "mcdseg_conv_split_direct_ok": (c_i32, [_P(ConvDesc)]),
    "mcdseg_conv_split_tile_config": (c_i32, [c_i32, c_i64, c_i32]),
    "mcdseg_conv_fprop": (c_int, [_P(ConvDesc), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mcdseg_conv_fprop_affine": (c_int, [_P(ConvDesc), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_i32, c_void_p, c_void_p]),
    "mcdseg_bn_eval_affine": (c_int, [c_void_p] * 5 + [c_i32, c_float, c_void_p, c_void_p, c_void_p]),
    "mcdseg_predict_workspace_bytes": (c_size_t, [c_i32, c_i32]),
    "mcdseg_predict_labels": (c_int, [c_void_p] * 4 + [c_i32] * 4 + [c_void_p, c_size_t, c_void_p]),
    "mcdseg_absmax": (c_int, [c_void_p, c_i64, c_void_p, c_void_p]),
    "mcdseg_conv_split_packed_bytes": (c_int, [_P(ConvDesc), c_i32, _P(c_i64), _P(c_i64)]),
    "mcdseg_conv_split_pack_weights": (c_int, [_P(ConvDesc), c_i32] + [c_void_p] * 5),
    "mcdseg_conv_split_pack_weights_multi": (c_int, [c_void_p, c_void_p, c_i32, c_i32, c_void_p, c_void_p]),
    "mcdseg_conv_split_fprop": (c_int, [_P(ConvDesc), c_i32] + [c_void_p] * 9),
    "mcdseg_conv_split_fprop_affine": (c_int, [_P(ConvDesc), c_i32] + [c_void_p] * 8 + [c_i32, c_void_p, c_void_p]),
    "mcdseg_conv_split_dgrad": (c_int, [_P(ConvDesc), c_i32] + [c_void_p] * 7),
    "mcdseg_conv_split_parts": (c_i64, [_P(ConvDesc), c_i32, c_i32, c_i32]),
    "mcdseg_conv_split_rest_pingpong": (c_i32, [_P(ConvDesc), c_i32, c_i32, c_i32]),
    "mcdseg_conv_split_wide_pingpong": (c_i32, [_P(ConvDesc), c_i32, c_i32, c_i32]),
    "mcdseg_conv_split_fprop_part": (c_int, [_P(ConvDesc), c_i32] + [c_void_p] * 8 + [c_i32, c_void_p]),
    "mcdseg_conv_split_dgrad_part": (c_int, [_P(ConvDesc), c_i32] + [c_void_p] * 6 + [c_i32, c_void_p]),
    "mcdseg_conv_split_dgrad_add": (c_int, [_P(ConvDesc), c_i32] + [c_void_p] * 7 + [c_i32, c_void_p]),
    "mcdseg_split_cb": (c_int, [c_void_p, c_void_p, c_void_p, c_i32, c_i32, c_i32, c_i32, c_void_p]),
    "mcdseg_split_cb_padded": (c_int, [c_void_p, c_void_p, c_void_p, c_i32, c_i32, c_i32, c_i32, c_void_p]),
    "mcdseg_unsplit_cb": (c_int, [c_void_p, c_void_p, c_i32, c_i32, c_i32, c_i32, c_void_p, c_void_p]),
    "mcdseg_bn_apply_cb": (c_int, [c_void_p] * 11 + [c_i32] * 5 + [c_void_p]),
    "mcdseg_bn_bwd_apply_cb": (c_int, [c_void_p] * 13 + [c_i32] * 6 + [c_void_p]),
    "mcdseg_bn_bwd_reduce_zmask": (c_int, [c_void_p] * 9 + [c_i32] * 4 + [c_void_p, c_size_t, c_void_p]),
    "mcdseg_bn_bwd_apply_cb_zmask": (c_int, [c_void_p] * 11 + [c_i32] * 5 + [c_void_p]),
    "mcdseg_bn_relu_mask_bytes": (c_size_t, [c_i32] * 3),
    "mcdseg_bn_apply_cb_mask": (c_int, [c_void_p] * 10 + [c_i32] * 4 + [c_void_p]),
    "mcdseg_bn_bwd_reduce_mask": (c_int, [c_void_p] * 9 + [c_i32] * 4 + [c_void_p, c_size_t, c_void_p]),
    "mcdseg_bn_bwd_apply_cb_mask": (c_int, [c_void_p] * 12 + [c_i32] * 5 + [c_void_p]),
    "mcdseg_conv_split_wgrad": (c_int, [_P(ConvDesc), c_i32] + [c_void_p] * 8 + [c_size_t, c_void_p]),
    "mcdseg_conv_dgrad": (c_int, [_P(ConvDesc), c_void_p, c_void_p, c_void_p, c_void_p]),
    "mcdseg_conv_wgrad_variant": (c_i32, [_P(ConvDesc), c_i32, c_i32]),
    "mcdseg_conv_wgrad_fits": (c_i32, [_P(ConvDesc), c_i32, c_i32]),
    "mcdseg_conv_wgrad_workspace_bytes": (c_size_t, [_P(ConvDesc)]),
    "mcdseg_conv_wgrad": (c_int, [_P(ConvDesc), c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "mcdseg_bn_stats_workspace_bytes": (c_size_t, [c_i64, c_i32]),
    "mcdseg_bn_stats_finalize": (c_int, [c_void_p, c_i64, c_i32, c_i32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                         c_float, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_i32, c_void_p, c_size_t, c_void_p]),
    "mcdseg_bn_eval_stats": (c_int, [c_void_p, c_void_p, c_i32, c_float, c_void_p, c_void_p, c_void_p]),
    "mcdseg_bn_apply": (c_int, [c_void_p] * 7 + [c_i32] * 4 + [c_void_p]),
    "mcdseg_bn_bwd_workspace_bytes": (c_size_t, [c_i32, c_i32, c_i32]),
    "mcdseg_bn_bwd_reduce": (c_int, [c_void_p] * 3 + [c_i32] + [c_void_p] * 7 + [c_i32] * 5 + [c_void_p, c_size_t, c_void_p]),
    "mcdseg_bn_bwd_apply": (c_int, [c_void_p] * 10 + [c_i32] * 5 + [c_void_p]),
    "mcdseg_up8_fwd": (c_int, [c_void_p] * 5 + [c_i32] * 4 + [c_void_p]),
    "mcdseg_up8_bwd_input": (c_int, [c_void_p] * 3 + [c_i32] * 4 + [c_void_p]),
    "mcdseg_up8_bwd_weight_workspace_bytes": (c_size_t, [c_i32] * 4),
    "mcdseg_up8_bwd_weight": (c_int, [c_void_p] * 3 + [c_i32] * 4 + [c_void_p, c_size_t, c_void_p]),
    "mcdseg_up8_bwd_workspace_bytes": (c_size_t, [c_i32] * 4),
    "mcdseg_up8_bwd": (c_int, [c_void_p] * 5 + [c_i32] * 4 + [c_void_p, c_size_t, c_void_p]),
    "mcdseg_loss_workspace_bytes": (c_size_t, [c_i32, c_i32]),
    "mcdseg_softmax_ce_l1": (c_int, [c_void_p] * 4 + [c_i64, c_float, c_float] + [c_void_p] * 4 + [c_i32] * 3 +
                             [c_void_p, c_size_t, c_void_p]),
    "mcdseg_up8_loss_workspace_bytes": (c_size_t, [c_i32] * 3),
    "mcdseg_up8_softmax_ce_l1": (c_int, [c_void_p] * 6 + [c_i64, c_float, c_float] + [c_void_p] * 4 + [c_i32] * 4 +
                                 [c_void_p, c_size_t, c_void_p]),
    "mcdseg_label_weight_sum_workspace_bytes": (c_size_t, [c_i64]),
    "mcdseg_label_weight_sum": (c_int, [c_void_p, c_void_p, c_i64, c_i32, c_i64, c_void_p, c_void_p, c_size_t, c_void_p]),
    "mcdseg_bilinear8_fwd": (c_int, [c_void_p, c_void_p] + [c_i32] * 4 + [c_void_p]),
    "mcdseg_bilinear8_bwd": (c_int, [c_void_p, c_void_p] + [c_i32] * 4 + [c_void_p]),
    "mcdseg_mse_workspace_bytes": (c_size_t, [c_i64]),
    "mcdseg_mse": (c_int, [c_void_p] * 4 + [c_i64, c_void_p, c_size_t, c_void_p]),
    "mcdseg_gate_mix_fwd": (c_int, [c_void_p] * 4 + [c_i64, c_void_p]),
    "mcdseg_gate_mix_bwd": (c_int, [c_void_p] * 7 + [c_i64, c_void_p]),
    "mcdseg_softmax_ch_fwd": (c_int, [c_void_p, c_void_p, c_i32, c_i32, c_i32, c_void_p]),
    "mcdseg_softmax_ch_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_i32, c_i32, c_i32, c_void_p]),
    "mcdseg_prob_nll_workspace_bytes": (c_size_t, [c_i32, c_i32]),
    "mcdseg_prob_nll": (c_int, [c_void_p, c_void_p, c_void_p, c_i64, c_i32, c_void_p, c_void_p, c_i32, c_i32, c_i32, c_void_p,
                                c_size_t, c_void_p]),
    "mcdseg_normalize_u8": (c_int, [c_void_p] * 4 + [c_i32] * 6 + [c_void_p]),
    "mcdseg_resize_workspace_bytes": (c_size_t, [c_i32] * 6),
    "mcdseg_resize_bilinear_u8": (c_int, [c_void_p, c_void_p] + [c_i32] * 6 + [c_void_p, c_size_t, c_void_p]),
    "mcdseg_resize_nearest_u8": (c_int, [c_void_p, c_void_p] + [c_i32] * 5 + [c_void_p, c_size_t, c_void_p]),
    "mcdseg_relabel_u8": (c_int, [c_void_p, c_void_p, c_i64, c_i32, c_i32, c_void_p]),
    "mcdseg_confusion_hist": (c_int, [c_void_p, c_void_p, c_i64, c_i32, c_void_p, c_void_p]),
    "mcdseg_scale_by_device_scalar": (c_int, [c_void_p, c_void_p, c_i64, c_void_p]),
    "mcdseg_sgd_momentum_flat": (c_int, [c_void_p, c_void_p, c_void_p, c_i64, c_float, c_float, c_float, c_float, c_void_p]),
    # 2-byte activation storage (round 6)
    "mcdseg_conv_split_half_ok": (c_i32, [_P(ConvDesc), c_i32, c_i32]),
    "mcdseg_conv_split_pp_deep": (c_i32, [_P(ConvDesc), c_i32, c_i32]),
    "mcdseg_up8_loss_variant": (c_i32, [c_i32] * 5),
    "mcdseg_conv_wgrad_thin_tr_config": (c_i32, [_P(ConvDesc)]),
    "mcdseg_conv_split_fprop_half": (c_int, [_P(ConvDesc), c_i32] + [c_void_p] * 7 + [c_i32, c_void_p]),
    "mcdseg_conv_split_dgrad_half": (c_int, [_P(ConvDesc), c_i32] + [c_void_p] * 6 + [c_i32, c_void_p]),
    "mcdseg_bn_apply_half": (c_int, [c_void_p] * 10 + [c_i32] * 4 + [c_void_p]),
    "mcdseg_bn_bwd_half_workspace_bytes": (c_size_t, [c_i32] * 3),
    "mcdseg_bn_bwd_reduce_half": (c_int, [c_void_p] * 11 + [c_i32] * 5 + [c_void_p, c_size_t, c_void_p]),
    "mcdseg_bn_bwd_apply_half": (c_int, [c_void_p] * 13 + [c_i32] * 5 + [c_void_p]),
    "mcdseg_pack_bf16_units": (c_int, [c_void_p, c_void_p, c_i32, c_i32, c_i32, c_void_p]),
    "mcdseg_unpack_bf16_units": (c_int, [c_void_p, c_void_p, c_i32, c_i32, c_i32, c_void_p]),
}
EXPORTS = tuple(_SIGNATURES)

_lib = None
_lock = threading.Lock()


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


# gfx950 erratum met in round 3: a packed-fp32 VALU instruction (v_pk_add_f32 / v_pk_mul_f32 /
# v_pk_fma_f32) whose OP_SEL routes the HIGH dword of a 64-bit VGPR source to the LOW result lane occasionally reads 0.0 for one
# 16-lane pass when another process loads the same CUs -- found as single-channel (c % 8 == 1) errors of the BatchNorm
# backward under two ranks per device.  The compiler forms these instructions on its own from scalar fp32 source code, so the
# files below are compiled with the packed-fp32 feature off, and ``packed_f32_opsel_sites`` (tests/test_cabi_and_host.py)
# disassembles the built library to prove that no such instruction is left in ANY kernel.
NO_PACKED_F32 = {"bn.hip", "loss.hip", "multitask.hip", "fusion.hip", "io.hip", "sgd.hip", "up8.hip"}
CFLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-fvisibility=hidden"]
NO_PK_FLAGS = ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]  # (the host pass ignores it with a warning)
OBJ_DIR = os.path.join(CSRC, "build")


def _compile_one(hipcc, src, obj, verbose):
    cmd = [hipcc] + CFLAGS + (NO_PK_FLAGS if os.path.basename(src) in NO_PACKED_F32 else []) + ["-I", INCLUDE, "-I", CSRC, "-c", src,
                                                                                                "-o", obj + ".tmp%d" % os.getpid()]
    if verbose:
        print(" ".join(cmd))
    r = subprocess.run(cmd, capture_output=True, text=True)
    noise = ("is not a recognized feature for this target", "warning generated")
    err = "\n".join(ln for ln in r.stderr.splitlines() if not any(t in ln for t in noise))
    if r.returncode != 0:
        raise RuntimeError("hipcc failed on %s:\n%s" % (src, err))
    if err.strip():
        print(err)
    os.replace(obj + ".tmp%d" % os.getpid(), obj)


def build(force=False, verbose=False):
    """Compile csrc/*.hip for gfx950 into mcdseg/libmcdseg.so (in-tree, so it travels with the repo): one object per source
    (csrc/build/*.o, compiled in parallel, re-used while its source and the headers are unchanged), then one link."""
    srcs = sources()
    hdrs = glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(INCLUDE, "*.h")) + [os.path.abspath(__file__)]
    deps = srcs + hdrs
    if not force and os.path.exists(LIB_PATH) and all(os.path.getmtime(LIB_PATH) >= os.path.getmtime(d) for d in deps):
        return LIB_PATH
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OBJ_DIR, exist_ok=True)
    hdr_time = max(os.path.getmtime(h) for h in hdrs)
    objs, todo = [], []
    for src in srcs:
        obj = os.path.join(OBJ_DIR, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_time):
            todo.append((src, obj))
    if todo:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=min(len(todo), max(1, (os.cpu_count() or 2) - 1))) as pool:
            for f in [pool.submit(_compile_one, hipcc, src, obj, verbose) for src, obj in todo]:
                f.result()
    cmd = [hipcc, "--offload-arch=gfx950", "-fPIC", "-shared", "-fvisibility=hidden", "-o", LIB_PATH + ".tmp%d" % os.getpid()] + objs
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)  # private temp + atomic rename: concurrent builders cannot corrupt the library
    os.replace(LIB_PATH + ".tmp%d" % os.getpid(), LIB_PATH)
    return LIB_PATH


def source_fingerprint():
    """sha1 over the kernel sources (csrc/*.hip, csrc/*.h, include/*.h, in name order): what a table of per-kernel measurements
    (profiles/*_pmc_traffic.json) records about the build it was made from, and what bench.py compares before it quotes the table"""
    import hashlib
    h = hashlib.sha1()
    for path in sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(INCLUDE, "*.h"))):
        h.update(os.path.basename(path).encode())
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


_DISASSEMBLY = {}


def device_disassembly(path=None):
    """ISA text of every gfx950 code object bundled in the library (llvm-objdump from the ROCm toolchain; no GPU needed)"""
    import tempfile
    path = path or LIB_PATH
    key = (path, os.path.getmtime(path))
    if _DISASSEMBLY.get("key") == key:  # (the guards below each walk the same text)
        return _DISASSEMBLY["text"]
    llvm = os.environ.get("ROCM_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
    out = []
    with tempfile.TemporaryDirectory() as tmp:
        copy = os.path.join(tmp, "lib.so")
        with open(path, "rb") as f, open(copy, "wb") as g:
            g.write(f.read())
        subprocess.run([os.path.join(llvm, "llvm-objdump"), "--offloading", copy], check=True, capture_output=True, cwd=tmp)
        for co in sorted(glob.glob(os.path.join(tmp, "lib.so.*gfx950*"))):
            out.append(subprocess.run([os.path.join(llvm, "llvm-objdump"), "-d", co], check=True, capture_output=True, text=True).stdout)
    _DISASSEMBLY.update(key=key, text="\n".join(out))
    return _DISASSEMBLY["text"]


def packed_f32_opsel_sites(path=None):
    """[(kernel symbol, instruction)] of every packed-fp32 VALU instruction with a non-default OP_SEL (the pattern of the erratum
    described at NO_PACKED_F32) in the built library; must be empty"""
    import re
    sites, sym = [], "?"
    pat = re.compile(r"\bv_pk_(add|mul|fma)_f32\b.*\bop_sel:\[")
    for line in device_disassembly(path).splitlines():
        m = re.match(r"^[0-9a-f]+ <(.*)>:$", line)
        if m:
            sym = m.group(1)
        elif pat.search(line):
            sites.append((sym, line.strip()))
    return sites


def spills_inside_matrix_loops(path=None, prefix=("conv_gemm_split_pp_kernel", "conv_wgrad_split_pp_kernel", "conv_wgrad_split_pp3_kernel")):
    """[(kernel symbol, line offset)] of every scratch (spill) access that lies BETWEEN the first and the last matrix instruction of a
    ping-pong kernel, i.e. inside or between its K loops; must be empty.  (The 256 x 256 and 256 x 128 forward tiles hold 10 / 13
    spilled registers -- values parked before the K loop and fetched back in the epilogue, which costs a few scratch accesses per
    wave and tile; a spill inside the loop would cost them per K-step.)"""
    import re
    out, sym, body = [], None, []

    def close():
        if sym is None or not any(p in sym for p in prefix):
            return
        mf = [i for i, ln in enumerate(body) if "v_mfma" in ln]
        if not mf:
            return
        out.extend((sym, i) for i, ln in enumerate(body) if "scratch_" in ln and mf[0] < i < mf[-1])

    for line in device_disassembly(path).splitlines():
        m = re.match(r"^[0-9a-f]+ <(.*)>:$", line)
        if m:
            close()
            sym, body = m.group(1), []
        else:
            body.append(line)
    close()
    return out


def drains_inside_store_loops(path=None, prefix=("up8_softmax_ce_l1_dma_kernel", "up8_bwd_band_kernel")):
    """[(kernel symbol, instruction)] of every ``s_waitcnt vmcnt(0)`` of these kernels that lies between an LDS-DMA issue and the last
    global store behind it; must be empty.  gfx950 counts loads, stores and LDS-DMA in ONE in-order counter: a ``vmcnt(0)`` there -- the
    compiler puts one in front of any load it tracks whose first use is inside the loop, and in front of LDS reads behind an LDS-DMA issued
    through the builtin -- makes every iteration wait for the prefetch it has just issued (and, in the loss kernel, would sit between the
    DMA and the stores whose count the hand-written ``vmcnt(63)`` relies on).  DESIGN.md section 4.1g."""
    import re
    out, sym, body = [], None, []

    def close():
        if sym is None or not any(p in sym for p in prefix):
            return
        dma = [i for i, ln in enumerate(body) if re.search(r"buffer_load_dword(x4)? .* lds", ln)]
        st = [i for i, ln in enumerate(body) if "global_store_dword" in ln]
        if not dma or not st:
            return
        out.extend((sym, ln.strip()) for i, ln in enumerate(body) if dma[-1] < i < st[-1] and re.search(r"s_waitcnt.*vmcnt\(0\)", ln))

    for line in device_disassembly(path).splitlines():
        m = re.match(r"^[0-9a-f]+ <(.*)>:$", line)
        if m:
            close()
            sym, body = m.group(1), []
        else:
            body.append(line)
    close()
    return out


def _kernel_bodies(path=None):
    """[(kernel symbol, [instruction lines])] of the built library"""
    import re
    out, sym, body = [], None, []
    for line in device_disassembly(path).splitlines():
        m = re.match(r"^[0-9a-f]+ <(.*)>:$", line)
        if m:
            if sym is not None:
                out.append((sym, body))
            sym, body = m.group(1), []
        else:
            body.append(line)
    if sym is not None:
        out.append((sym, body))
    return out


HIDDEN_DMA_KERNELS = ("up8_softmax_ce_l1_dma_kernel", "up8_bwd_band_kernel", "conv_wgrad_split_tr64_kernel")  # users of mcd_hidden_dma


def hidden_dma_hazards(path=None):
    """[(kernel symbol, instruction)] that break the contract of ``mcd_hidden_dma`` (csrc/common.h): that inline assembly writes M0 -- the
    LDS base of the DMA behind it -- without the compiler's knowledge, so a kernel that uses it must contain NO other use of M0: in the
    kernels of HIDDEN_DMA_KERNELS every instruction that names M0 and every LDS-DMA must belong to one triple
    `s_mov_b32 m0, sN; s_nop 0; buffer_load_dword[x4] ... lds` -- an LDS-DMA outside a triple is one whose M0 set-up the compiler hoisted
    (a builtin DMA beside the hidden ones), any other M0 instruction (`s_add_i32 m0`, `v_movrel*`, an `s_mov_b32 sN, m0` spill) is a
    compiler-made use the asm would clobber.  Must be empty.  (Whether a wait separates a DMA from the LDS reads of its buffer cannot be
    read off the text -- the wait sits at the head of the NEXT loop iteration, and the reads right behind a DMA are those of the other
    buffer: that is what the bitwise A/B tests against the register kernels and tools/op_contention.py's soak establish.)"""
    import re
    bad, seen = [], set()
    for sym, body in _kernel_bodies(path):
        if not any(k in sym for k in HIDDEN_DMA_KERNELS):
            continue
        ins = [ln.split("//")[0].strip() for ln in body]
        ins = [i for i in ins if i]
        trip = set()
        for i in range(len(ins) - 2):
            if re.match(r"s_mov_b32 m0, s\d+$", ins[i]) and ins[i + 1] == "s_nop 0" and re.match(r"buffer_load_dword(x4)? .* lds$", ins[i + 2]):
                trip.update((i, i + 2))
        if trip:
            seen.add([k for k in HIDDEN_DMA_KERNELS if k in sym][0])
        for i, t in enumerate(ins):
            if (re.search(r"\bm0\b", t) or re.match(r"buffer_load_dword(x4)? .* lds$", t)) and i not in trip:
                bad.append((sym, t))
    bad.extend((k, "no hidden DMA found in this kernel: is HIDDEN_DMA_KERNELS stale?") for k in HIDDEN_DMA_KERNELS if k not in seen)
    return bad


def loss_dma_store_counts(path=None):
    """{(classes, heads): global stores} of the LDS-DMA loss kernels: the hand-counted `s_waitcnt vmcnt(63)` of csrc/loss.hip relies on a
    wave issuing its item's heads x C gradient stores BEHIND the next item's DMAs (`behind = nst`)"""
    import re
    out = {}
    for sym, body in _kernel_bodies(path):
        m = re.search(r"up8_softmax_ce_l1_dma_kernelILi(\d+)ELb([01])ELb[01]E", sym)
        if m:
            n = sum(1 for ln in body if re.search(r"\bglobal_store_dword\b", ln))
            key = (int(m.group(1)), 2 if m.group(2) == "1" else 1)
            out[key] = min(n, out.get(key, n))
    return out


def lib():
    """The loaded library; raises if it has not been built (no fallback exists)."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                if not os.path.exists(LIB_PATH):
                    raise RuntimeError("libmcdseg.so is missing (%s): run `python -c 'import __graft_entry__ as g; g.build()'` "
                                       "-- the HIP kernels are the only implementation, there is no fallback" % LIB_PATH)
                # PyTorch-ROCm bundles its own libamdhip64.so; the kernels must run in THAT runtime instance (they
                # are launched on torch's streams), so it has to be resident before libmcdseg.so resolves its
                # libamdhip64.so.7 dependency -- otherwise /opt/rocm's copy is loaded as a second, device-less runtime.
                import torch
                bundled = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
                if os.path.exists(bundled):
                    ctypes.CDLL(bundled, mode=ctypes.RTLD_GLOBAL)
                handle = ctypes.CDLL(LIB_PATH)
                for name, (res, args) in _SIGNATURES.items():
                    fn = getattr(handle, name)
                    fn.restype = res
                    fn.argtypes = args
                if handle.mcdseg_version() != 101:
                    raise RuntimeError("libmcdseg.so version mismatch: %d" % handle.mcdseg_version())
                _options_from_env(handle)
                _lib = handle
    return _lib


def option_names():
    """names of the library's plan / development options (csrc/options.h)"""
    L = lib()
    return [L.mcdseg_option_name(i).decode() for i in range(L.mcdseg_option_count())]


def _options_from_env(handle):
    """The library never reads the environment (include/mcdseg.h, mcdseg_set_option): the MCDSEG_<OPTION> variables are translated HERE,
    once, when the library is loaded.  A value that is not an integer is an error, not a silent default."""
    for i in range(handle.mcdseg_option_count()):
        name = handle.mcdseg_option_name(i).decode()
        raw = os.environ.get("MCDSEG_" + name)
        if raw is None or raw.strip() == "":
            continue
        try:
            value = int(raw.strip())
        except ValueError:
            raise ValueError("MCDSEG_%s must be an integer, got %r" % (name, raw))
        if handle.mcdseg_set_option(name.encode(), value) != 0:
            raise RuntimeError("mcdseg_set_option(%s) failed" % name)


def get_option(name):
    """current value of a library option"""
    v = c_i64()
    check(lib().mcdseg_get_option(name.encode(), ctypes.byref(v), None), "get_option")
    return v.value


def option_default(name):
    d = c_i64()
    check(lib().mcdseg_get_option(name.encode(), None, ctypes.byref(d)), "get_option")
    return d.value


OPTION_EPOCH = 0  # bumped by every set_option: host-side caches of the library's plan answers (ops._batch_pieces) are keyed by it


def set_option(name, value):
    """set a library option (process-wide; read by the launchers at every call); returns the previous value"""
    global OPTION_EPOCH
    prev = get_option(name)
    check(lib().mcdseg_set_option(name.encode(), int(value)), "set_option")
    OPTION_EPOCH += 1
    return prev


class options:
    """``with mcdseg.options(PINGPONG=0, PP_CUS=16): ...`` -- library options for the duration of a block (also usable from a test
    fixture: ``enter()`` / ``restore()``)"""

    def __init__(self, **values):
        self.values, self.prev = values, {}

    def __enter__(self):
        for k, v in self.values.items():
            self.prev[k] = set_option(k, v)
        return self

    def __exit__(self, *exc):
        for k, v in self.prev.items():
            set_option(k, v)
        self.prev = {}


def check(rc, what):
    if rc != 0:
        msg = lib().mcdseg_last_error()
        raise RuntimeError("%s failed (%d): %s" % (what, rc, msg.decode() if msg else "?"))
