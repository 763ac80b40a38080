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
    _fields_ = [(n, c_i32) for n in ("N", "Cin", "H", "W", "Cout", "KH", "KW", "stride", "pad", "dil", "Ho", "Wo")]


_P = ctypes.POINTER
_SIGNATURES = {
    # name: (restype, argtypes)   -- one entry per symbol declared in include/mcdseg.h
    "mcdseg_version": (c_int, []),
    "mcdseg_last_error": (ctypes.c_char_p, []),
    "mcdseg_conv_packed_dims": (c_int, [_P(ConvDesc), _P(c_i32), _P(c_i32), _P(c_i32), _P(c_i32)]),
    "mcdseg_conv_pack_weights": (c_int, [_P(ConvDesc), c_void_p, c_void_p, c_void_p, c_void_p]),
    "mcdseg_conv_stat_rows": (c_i64, [_P(ConvDesc)]),
    "mcdseg_conv_split_stat_rows": (c_i64, [_P(ConvDesc)]),
    "mcdseg_conv_split_stat_rows_for": (c_i64, [_P(ConvDesc), c_i32, c_i32]),
    "mcdseg_conv_split_window_ok": (c_i32, [_P(ConvDesc), c_i32, c_i32, c_i32]),
    "mcdseg_conv_split_direct_ok": (c_i32, [_P(ConvDesc)]),
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
    "mcdseg_split_cb": (c_int, [c_void_p, c_void_p, c_void_p, c_i32, c_i32, c_i32, c_i32, c_void_p]),
    "mcdseg_split_cb_padded": (c_int, [c_void_p, c_void_p, c_void_p, c_i32, c_i32, c_i32, c_i32, c_void_p]),
    "mcdseg_unsplit_cb": (c_int, [c_void_p, c_void_p, c_i32, c_i32, c_i32, c_i32, c_void_p, c_void_p]),
    "mcdseg_bn_apply_cb": (c_int, [c_void_p] * 11 + [c_i32] * 5 + [c_void_p]),
    "mcdseg_bn_bwd_apply_cb": (c_int, [c_void_p] * 13 + [c_i32] * 6 + [c_void_p]),
    "mcdseg_bn_bwd_reduce_zmask": (c_int, [c_void_p] * 9 + [c_i32] * 4 + [c_void_p, c_size_t, c_void_p]),
    "mcdseg_bn_bwd_apply_cb_zmask": (c_int, [c_void_p] * 11 + [c_i32] * 5 + [c_void_p]),
    "mcdseg_conv_split_wgrad": (c_int, [_P(ConvDesc), c_i32] + [c_void_p] * 8 + [c_size_t, c_void_p]),
    "mcdseg_conv_dgrad": (c_int, [_P(ConvDesc), c_void_p, c_void_p, c_void_p, c_void_p]),
    "mcdseg_conv_wgrad_variant": (c_i32, [_P(ConvDesc), c_i32, c_i32]),
    "mcdseg_conv_wgrad_workspace_bytes": (c_size_t, [_P(ConvDesc)]),
    "mcdseg_conv_wgrad": (c_int, [_P(ConvDesc), c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "mcdseg_bn_stats_workspace_bytes": (c_size_t, [c_i64, c_i32]),
    "mcdseg_bn_stats_finalize": (c_int, [c_void_p, c_i64, c_i32, c_i32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                         c_float, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
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
}
EXPORTS = tuple(_SIGNATURES)

_lib = None
_lock = threading.Lock()


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def build(force=False, verbose=False):
    """Compile csrc/*.hip for gfx950 into mcdseg/libmcdseg.so (in-tree, so it travels with the repo)."""
    srcs = sources()
    deps = srcs + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(INCLUDE, "*.h"))
    if not force and os.path.exists(LIB_PATH) and all(os.path.getmtime(LIB_PATH) >= os.path.getmtime(d) for d in deps):
        return LIB_PATH
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-fvisibility=hidden", "-shared", "-I", INCLUDE, "-I", CSRC,
           "-o", LIB_PATH + ".tmp%d" % os.getpid()] + srcs  # private temp + atomic rename: concurrent builders cannot corrupt it
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    os.replace(LIB_PATH + ".tmp%d" % os.getpid(), LIB_PATH)
    return LIB_PATH


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
                if handle.mcdseg_version() != 100:
                    raise RuntimeError("libmcdseg.so version mismatch: %d" % handle.mcdseg_version())
                _lib = handle
    return _lib


def check(rc, what):
    if rc != 0:
        msg = lib().mcdseg_last_error()
        raise RuntimeError("%s failed (%d): %s" % (what, rc, msg.decode() if msg else "?"))
