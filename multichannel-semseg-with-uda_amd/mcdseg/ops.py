"""torch.autograd wrappers around the libmcdseg kernels.

PyTorch supplies device memory, the current HIP stream and the autograd tape; every arithmetic
pass over activations, gradients and parameters below is a hand-written HIP kernel.  The functions
mirror the ATen call sites of the reference's hot path:

  conv_bn_act      nn.Conv2d + nn.BatchNorm2d + ReLU (+ residual)   models/drn.py:43-59, 80-100, 195-205
  conv2d_bias      nn.Conv2d with bias (the 1x1 ``seg`` head)       models/dilated_fcn.py:226-232
  up8 / up8_dual   depthwise ConvTranspose2d k16 s8 p4              models/dilated_fcn.py:357-366, 479-491
  cross_entropy2d  log_softmax + weighted NLL                       loss.py:7-13
  diff2d           mean |softmax - softmax|                         loss.py:93-100
  mcd_losses       both of the above in one fused kernel            adapt_trainer.py:163-212
"""
import collections
import contextlib
import ctypes
import os
import threading
import weakref

import torch

from ._lib import ConvDesc, check, get_option, lib

# bumped by the optimizer after every in-place parameter update (the kernels write through raw
# pointers, so torch's own version counters do not move)
WEIGHT_EPOCH = 0


def bump_weight_epoch(params=None):
    """an in-place parameter update happened: of ``params`` (each carries its own counter, so that stepping one optimizer does not
    invalidate the packed images of another's parameters), or -- None -- of anything"""
    global WEIGHT_EPOCH
    if params is None:
        WEIGHT_EPOCH += 1
        return
    for p in params:
        p._mcd_epoch = getattr(p, "_mcd_epoch", 0) + 1


# Optional launch timer (bench.py): an object with ``wants(name) -> bool`` and ``add(name, work, start, stop)``.
# When set, the named kernel launches are bracketed by HIP events recorded on the launch stream.
LAUNCH_TIMER = None


class _timed:
    def __init__(self, name, work):
        self.on = LAUNCH_TIMER is not None and LAUNCH_TIMER.wants(name)
        self.name, self.work = name, work

    def __enter__(self):
        if self.on:
            self.t0 = torch.cuda.Event(enable_timing=True)
            self.t1 = torch.cuda.Event(enable_timing=True)
            self.t0.record()

    def __exit__(self, *exc):
        if self.on:
            self.t1.record()
            LAUNCH_TIMER.add(self.name, self.work, self.t0, self.t1)


def conv_work(desc):
    """(algorithmic FLOPs, algorithmic bytes) of one conv pass: 2*MACs; input read once + output written once."""
    macs = desc.N * desc.Ho * desc.Wo * desc.Cout * desc.Cin * desc.KH * desc.KW
    byts = 4 * (desc.N * desc.Cin * desc.H * desc.W + desc.N * desc.Cout * desc.Ho * desc.Wo)
    return 2 * macs, byts


POLICY = {"f16x3": "SplitF16x3", "bf16x6": "SplitBf16x6", "f16x1": "SplitF16x1"}


def gemm_kernel_name(m, k, dgrad, split=False, presplit=False, direct=False, pixels=0, math=None):
    """Template instantiation conv_fprop / conv_dgrad dispatch to (the launcher's own rule through
    ``mcdseg_conv_split_tile_config``; csrc/common.h mcd_bm / mcd_bk for the f32 kernels); the string equals the kernel name
    rocprofv3 prints, so profiles/*_pmc_traffic.json can be keyed by it."""
    bm = 32 if m <= 32 else (64 if m <= 64 else 128)
    cfg = {128: "2, 2, 2, 2", 64: "2, 2, 1, 4", 32: "1, 2, 1, 4"}[bm]
    if split and direct:
        return "conv_stem_x6_kernel"
    if split:
        code = "%04d" % lib().mcdseg_conv_split_tile_config(int(m), int(pixels), int(bool(presplit)))
        return "conv_gemm_split_kernel<%s, %s, %s, %s>" % (POLICY[math or CONV_MATH], ", ".join(code), "true" if dgrad else "false",
                                                           "true" if presplit else "false")
    return "conv_gemm_kernel<%s, %d, %s>" % (cfg, 8 if k <= 8 else 16, "true" if dgrad else "false")


def pingpong_kernel_name(dgrad, math=None, small=False, wide=0, deep=False):
    """rocprofv3's name of the 8-wave ping-pong kernel (csrc/conv_gemm_split_pp.hip): its 256 x 256 tile, the 256 x 128 one
    (``small``), the 256 x 320 one (``wide`` = 1), the 128 x 320 one (2) or the 256 x 160 one (3: what ``mcdseg_conv_split_wide_pingpong`` returns).
    ``deep`` (``mcdseg_conv_split_pp_deep``; never on the 256 x 128 tile): the one-term arithmetic with two K-steps per barrier interval"""
    tile = {0: "2, 2, 2, 2" if small else "4, 2, 1, 4", 1: "2, 5, 2, 2", 2: "1, 5, 2, 2", 3: "1, 5, 4, 1"}[int(wide)]
    policy = POLICY[math or CONV_MATH] + ("D" if (deep and not (small and not wide) and (math or CONV_MATH) == "f16x1") else "")
    return "conv_gemm_split_pp_kernel<%s, %s, %s>" % (policy, "true" if dgrad else "false", tile)


def _split_launches(d, presplit, dgrad, name, call, work=None):
    """One split-arithmetic convolution as the library would launch it, each kernel bracketed by its own timer: the pixels
    ``mcdseg_conv_split_parts`` gives to the ping-pong kernel (part 1) and the rest on the 4-wave tiles (part 2); ``call(part)``
    invokes the ``_part`` entry point.  Work is shared out by pixels."""
    if LAUNCH_TIMER is None:  # nobody asks which kernels ran: one call, the library launches both parts itself (and the host saves
        call(0)               # the four plan queries below -- 700 convolutions per MCD step)
        return
    if callable(name):
        name = name()
    pixels = d.N * (d.H * d.W if dgrad else d.Ho * d.Wo)
    pp = lib().mcdseg_conv_split_parts(ctypes.byref(d), MATH_ID[CONV_MATH], int(presplit), int(dgrad)) if presplit else 0
    flops, byts = (work or conv_work)(d)
    if pp > 0:
        wide = lib().mcdseg_conv_split_wide_pingpong(ctypes.byref(d), MATH_ID[CONV_MATH], 1, int(dgrad))
        deep = bool(lib().mcdseg_conv_split_pp_deep(ctypes.byref(d), MATH_ID[CONV_MATH], int(dgrad)))
        with _timed(pingpong_kernel_name(dgrad, wide=wide, deep=deep), (flops * pp / pixels, byts * pp / pixels)):
            call(1)
    if pp < pixels:
        if presplit and lib().mcdseg_conv_split_rest_pingpong(ctypes.byref(d), MATH_ID[CONV_MATH], 1, int(dgrad)):
            name = pingpong_kernel_name(dgrad, small=True)
        with _timed(name, (flops * (pixels - pp) / pixels, byts * (pixels - pp) / pixels)):
            call(2 if pp > 0 else 0)


def wgrad_kernel_name(cout, cin, taps=9):
    if cin <= 16 and taps > 1:
        return "conv_wgrad_thin_kernel<%d, %s>" % (8 if cin <= 8 else 16, "true" if cout <= 16 else "false")
    lo = min(cout, cin)
    mid = lo > 32 or (lo > 16 and cin % 8 == 0 and cout % 8 == 0)  # (make_plan of csrc/conv_wgrad.hip: the 64 x 64 plan)
    return "conv_wgrad_kernel<%s>" % ("2, 2, 2, 2, 16" if lo > 64 else ("1, 1, 2, 2, 32" if mid else "1, 1, 1, 1, 32"))


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


# the stream the next kernels are launched on when it is not torch's current one: the side stream while _conv_backward enqueues a
# weight gradient there.  torch's CURRENT stream stays the main one meanwhile, so every tensor still comes from the main stream's
# allocator pool, and whatever those kernels touch is collected in ``keep``: the caller holds the list until the main stream has
# waited for the side stream, which makes the memory reusable at once.  (Tensor.record_stream instead would return it only when
# the side stream has passed the point in WALL-CLOCK time; the host enqueues a whole backward pass ahead of the device, so the
# allocator kept growing -- 27 -> 116 GB reserved for drn_d_105 at N = 8 -- and cfg5 at N = 32, 232 of 288 GB, fell into its
# free-everything-and-retry path: 14-18 s per step instead of 3.8 s.)
class _LaunchStream(threading.local):  # (per thread: the autograd engine runs backward nodes on one worker thread per device)
    stream = None
    keep = None


_LAUNCH = _LaunchStream()


def _stream():
    s = _LAUNCH.stream
    return ctypes.c_void_p((s if s is not None else torch.cuda.current_stream()).cuda_stream)


def _on_launch_stream(*tensors):
    """tensors just handed to kernels on the launch stream: their memory must not be reused before those kernels have run"""
    if _LAUNCH.stream is not None:
        _LAUNCH.keep.extend(t for t in tensors if t is not None)


def _req(t, name, dtype=torch.float32):
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("mcdseg: %s must live on the GPU -- the HIP kernels are the only implementation "
                           "(no CPU fallback)" % name)
    if t.dtype != dtype:
        raise TypeError("mcdseg: %s must be %s, got %s" % (name, dtype, t.dtype))
    return t if t.is_contiguous() else t.contiguous()


def _ws(nbytes, device):
    return torch.empty(max(int(nbytes), 4) // 4 + 1, dtype=torch.float32, device=device)


def conv_desc(x_shape, w_shape, stride, pad, dil):
    n, cin, h, w = x_shape
    cout, cin_w, kh, kw = w_shape
    if cin_w != cin:
        raise ValueError("mcdseg: conv expects %d input channels, got %d" % (cin_w, cin))
    ho = (h + 2 * pad - dil * (kh - 1) - 1) // stride + 1
    wo = (w + 2 * pad - dil * (kw - 1) - 1) // stride + 1
    return ConvDesc(n, cin, h, w, cout, kh, kw, stride, pad, dil, ho, wo)


# Matrix-pipe arithmetic of the convolutions (MCDSEG_CONV_MATH), csrc/split.h:
#   "f16x3" (default)   fp32 operands as s * (h1 + h2): two fp16 pieces and a power-of-two scale per tensor, three cross terms
#                       on v_mfma_f32_32x32x16_f16 -- fp32-grade against the reference's fp64 gradients (same noise-floor
#                       test as the other modes) at 5.3x the f32 matrix rate;
#   "bf16x6"            three bf16 pieces, the six largest cross terms on v_mfma_f32_32x32x16_bf16 (2.67x the f32 rate);
#   "f32"               v_mfma_f32_32x32x2_f32, the exact k-ordered fp32 FMA chain.
#   "f16x1"             REDUCED precision (BASELINE config 5's "bf16"; trainers / bench: --dtype f16): f16x3's operands -- same
#                       companions, weight images and bounds -- multiplied with the leading term only: one MFMA instead of three,
#                       operands rounded to fp16's 11 significant bits (bf16 would keep 8); accumulation, BatchNorm, loss and
#                       optimizer stay fp32.  NOT within north_star's 1e-3 of the fp32 reference: tests/test_model_gpu.py states
#                       what it keeps.
CONV_MATH = os.environ.get("MCDSEG_CONV_MATH", "f16x3")
# producers (BN apply / BN backward apply) also emit the split of what they write, so the convolution that gathers it
# does not re-split every activation inside its K loop (MCDSEG_PRESPLIT=0 turns this off)
PRESPLIT = os.environ.get("MCDSEG_PRESPLIT", "1") != "0"
if CONV_MATH not in ("f16x3", "bf16x6", "f32", "f16x1"):
    raise ValueError("MCDSEG_CONV_MATH must be f16x3, bf16x6, f32 or f16x1, got %r" % CONV_MATH)
MATH_ID = {"f16x3": 3, "bf16x6": 6, "f16x1": 1}   # MCDSEG_MATH_F16X3 / _BF16X6 / _F16X1 of include/mcdseg.h
PIECES = {"f16x3": 2, "bf16x6": 3, "f16x1": 2}    # (f16x1 stores what f16x3 stores)


# the stem's training forward on the LDS-window kernel (f16x3, 0.20 instead of 0.35 ms per launch at 14 x 480 x 640).  OFF by
# default: the stem's rounding is amplified by every BatchNorm behind it -- with the 22-bit f16x3 stem the weight updates of the
# small three-step trace sit at 1.7-4x the reference's own fp32-vs-fp64 spread, with the 24-bit bf16x6 direct kernel at 0.8-1.0x
# (tools/delta_report.py) -- and the layer is 0.4 % of the step
STEM_WINDOW = os.environ.get("MCDSEG_STEM_WINDOW", "0") != "0"
STEM_DIRECT = os.environ.get("MCDSEG_STEM_DIRECT", "1") != "0"  # the stem's forward as the direct (bf16x6) convolution

# Activation storage inside a DRN trunk (MCDSEG_ACT_STORAGE):
#   "fp32" (default)  every fused group writes its fp32 output y next to the pre-split companion;
#   "compact"         between the layers of a trunk only the companion is written (4 B/element with f16x3: the 22 leading bits
#                     of y) -- residual adds and ReLU masks read it, no fp32 y exists.  8 instead of 12 B/element are kept for
#                     backward, which is what lets BASELINE config 5 (drn_d_105, 32 x 720 x 1280 per GPU) fit 288 GB, and the
#                     BatchNorm passes move a third fewer bytes.  The trunk's last layer, and everything outside a trunk, stays fp32.
ACT_STORAGE = os.environ.get("MCDSEG_ACT_STORAGE", "fp32")
if ACT_STORAGE not in ("fp32", "compact"):
    raise ValueError("MCDSEG_ACT_STORAGE must be fp32 or compact, got %r" % ACT_STORAGE)
_TRUNK_DEPTH = 0


class trunk_internal:
    """context of the layers INSIDE a trunk whose outputs only other fused groups consume (models/dilated_fcn.py Trunk)"""

    def __enter__(self):
        global _TRUNK_DEPTH
        _TRUNK_DEPTH += 1

    def __exit__(self, *exc):
        global _TRUNK_DEPTH
        _TRUNK_DEPTH -= 1


def _compact_now():
    return ACT_STORAGE == "compact" and _TRUNK_DEPTH > 0 and _scaled()


# 2-byte activation storage (round 6; BASELINE config 5 "bf16"): in the one-term arithmetic f16x1 a compact group keeps ONE 16-bit value per
# element of everything its BatchNorm passes move -- the convolution writes z as scaled fp16 units (after taking the BatchNorm partial sums
# from its fp32 accumulators), the activation is the leading piece of the companion alone, the gradients between the groups travel as bf16
# units (include/mcdseg.h, "2-byte activation storage").  Such a group's output is a VIRTUAL tensor of dtype bfloat16: the dtype is the
# contract with autograd -- whoever consumes it owes its gradient as a bf16 tensor of the same logical shape whose BYTES are in the unit
# layout [N][C/8][HW][8] (element-wise sums of two such tensors, which is all autograd ever does to them, do not care).
# MCDSEG_HALF_STORAGE=0: f16x1 stores what f16x3 stores (round 5's form).
HALF_STORAGE = os.environ.get("MCDSEG_HALF_STORAGE", "1") != "0"


def _half_now():
    return HALF_STORAGE and CONV_MATH == "f16x1" and _compact_now()


def is_half(t):
    """an activation of the 2-byte chain (see HALF_STORAGE): virtual, dtype bfloat16"""
    return t is not None and t.dtype == torch.bfloat16 and is_virtual(t)


def pack_bf16_units(g):
    """fp32 NCHW gradient -> the bf16 unit layout (a gradient entering the 2-byte chain from a kernel without the 16-bit epilogue)"""
    n, c, h, w = g.shape
    out = torch.empty((n, c, h, w), dtype=torch.bfloat16, device=g.device)
    check(lib().mcdseg_pack_bf16_units(_p(_req(g, "gradient")), _p(out), n, c, h * w, _stream()), "pack_bf16_units")
    return out


def unpack_bf16_units(g):
    """the inverse of ``pack_bf16_units``"""
    n, c, h, w = g.shape
    out = torch.empty((n, c, h, w), dtype=torch.float32, device=g.device)
    check(lib().mcdseg_unpack_bf16_units(_p(_req(g, "gradient", torch.bfloat16)), _p(out), n, c, h * w, _stream()), "unpack_bf16_units")
    return out


# A fused ReLU group WITHOUT residual whose output only the next split convolution consumes (conv1 of a BasicBlock, conv1 / conv2
# of a Bottleneck: ``conv_bn_act(..., internal=True)``) need not write its fp32 output at all, in ANY storage mode and without
# changing a bit: the consumer's forward and weight gradient read the companion either way, and the group's own backward
# takes its ReLU mask from z (BN_ZMASK).  8 instead of 12 bytes written per element, 4 bytes per element less kept for backward.
# (Taken for more than 32 channels and un-cut batches only: there the consumer's forward, data and weight gradient all run on the
# split kernels.  A consumer that needs fp32 after all reconstructs it from the companion, 22 bits, as in compact storage.)
INTERNAL_SKIP_Y = os.environ.get("MCDSEG_INTERNAL_SKIP_Y", "1") != "0"
# ... and so need the groups of a trunk's plain convolution chains (models/drn.py:195-205: the stem, layer1, layer2, layer7) whose output
# only the next stage's convolutions read (round 5; models/drn.py run_fused / _reads_companions_only decide, "0": they write fp32 too).
# These include the 16- and 32-channel layers: their consumers' forward, data and weight gradients all read companions since the thin
# layers' window kernels (round 2) and the tap-pair weight gradient from 24 channels up (round 5).
INTERNAL_STAGES = os.environ.get("MCDSEG_INTERNAL_STAGES", "1") != "0"
# a block's 1x1 projection shortcut is read as a residual only (fp32): its companion is not written ("0": it is, as before round 5)
SHORTCUT_NO_CB = os.environ.get("MCDSEG_SHORTCUT_NO_CB", "1") != "0"


def _virtual(shape, device, dtype=torch.float32):
    """stand-in for an activation that exists only as its companion: right shape / device / dtype, 4 bytes of storage"""
    return torch.empty(1, dtype=dtype, device=device).expand(shape)


# how many times a train-mode BatchNorm forward applies its running-statistics update (solvers/solver.py: one generator
# forward standing for two identical ones of the reference's schedule)
BN_RUNNING_REPEAT = 1


class bn_running_updates:
    def __init__(self, k):
        self.k = int(k)

    def __enter__(self):
        global BN_RUNNING_REPEAT
        self.old, BN_RUNNING_REPEAT = BN_RUNNING_REPEAT, self.k

    def __exit__(self, *exc):
        global BN_RUNNING_REPEAT
        BN_RUNNING_REPEAT = self.old


def is_virtual(t):
    return t is not None and getattr(t, "_mcd_virtual", False)


def materialize(x, cb=None, bound=None):
    """fp32 tensor of an activation kept as its companion (value = scale * sum of pieces); ``x`` itself when it is real"""
    if cb is None:
        if not is_virtual(x):
            return x
        cb, bound = _cb_of(x)
    if cb is None:
        raise RuntimeError("mcdseg: a compact activation lost its companion (was it modified in place?)")
    n, c, h, w = x.shape
    out = torch.empty((n, c, h, w), dtype=torch.float32, device=cb.device)
    check(lib().mcdseg_unsplit_cb(_p(cb), _p(bound), MATH_ID[CONV_MATH], n, c, h * w, _p(out), _stream()), "unsplit_cb")
    return out


def _use_split(contraction_channels):
    return contraction_channels >= 16 and CONV_MATH in MATH_ID


def _scaled():
    """the active arithmetic needs per-tensor bound scalars (f16x3 and its one-term form f16x1)"""
    return CONV_MATH in ("f16x3", "f16x1")


def absmax(x):
    """device scalar max |x| -- the bound of a tensor whose producer did not supply one"""
    x = _req(x, "tensor")
    b = torch.empty(1, dtype=torch.float32, device=x.device)
    check(lib().mcdseg_absmax(_p(x), x.numel(), _p(b), _stream()), "absmax")
    _on_launch_stream(b)
    return b


def _bound_or_measure(x, bound):
    """bound scalar of ``x`` for the scaled arithmetic: the producer's if known, else measured (cached on the tensor)"""
    if not _scaled():
        return None
    if bound is not None:
        return bound
    rec = getattr(x, "_mcd_bound", None)
    if rec is not None and rec[1] == x._version and rec[2] == x.data_ptr():
        return rec[0]
    b = absmax(x)
    try:
        x._mcd_bound = (b, x._version, x.data_ptr())
    except AttributeError:
        pass
    return b


_PACK_REGISTRY = weakref.WeakSet()   # every PackedWeights alive
_PACK_TABLES = {}                    # (device, math) -> cached device tables of the last group pack
GROUP_PACK = os.environ.get("MCDSEG_GROUP_PACK", "1") != "0"
FUSED_UP_LOSS = os.environ.get("MCDSEG_FUSED_UP_LOSS", "1") != "0"  # MCDSolver: up-sampler folded into the loss kernel
# BatchNorm backward of a ReLU group without residual: the mask y > 0 recomputed from z (bit-identical), y never read
BN_ZMASK = os.environ.get("MCDSEG_BN_ZMASK", "1") != "0"
# BatchNorm backward of a ReLU group WITH residual: the mask y > 0 from a bit-plane the forward apply kernel wrote (1 bit per element)
# instead of from the fp32 y (32): the same mask, bit for bit; "0": read y (round 4's form)
RELU_MASK = os.environ.get("MCDSEG_RELU_MASK", "1") != "0"
# the gradient of a residual block's input has two producers (the first convolution's data gradient and the shortcut); autograd would
# add them with an element-wise kernel -- 80 adds of 40-160 MB tensors per MCD step, 7.3 ms at BASELINE config 2.  With this on the
# one that runs second folds the other's tensor into its own epilogue (``GradBox``); "0": autograd's add (same bits, tests compare)
FUSE_RES_ADD = os.environ.get("MCDSEG_FUSE_RES_ADD", "1") != "0"


class GradBox:
    """Mailbox between the producers of ONE tensor's gradient inside a residual block (models/drn.py BasicBlock / Bottleneck: the block
    input feeds the first convolution and the shortcut -- the identity, or the 1x1 projection).  Each producer ``attach``es in the
    forward pass; in the backward pass every producer but the last leaves its gradient here and reports None to autograd, the last
    one returns the sum -- formed in its convolution's data-gradient epilogue when it is one (``_conv_dgrad``'s ``addend``).  The sum
    is the one autograd's own accumulation would form, bit for bit (fp32 addition commutes), one kernel and three passes over the
    tensor earlier.  A box lives as long as the graph of its forward pass; it may see several backward passes -- also PARTIAL ones
    (``torch.autograd.grad(..., inputs=[...])``, ``backward(inputs=...)``, a pass that died in an exception) in which only some of
    the producers run: what such a pass left behind is dropped when a producer of ANOTHER pass arrives (the passes are told apart by
    the autograd engine's graph-task id), so a stale count or tensor never enters a later sum.  (In a partial pass the gradient of the
    block input is complete only if all its producers run -- which they do whenever the engine needs that gradient at all, since every
    producer lies on a path to it; a producer that runs only for its own weight's sake parks a tensor nobody asked for.)  Tensor hooks
    on the block input see ONE call, with the sum."""
    __slots__ = ("n", "seen", "g", "task")

    def __init__(self):
        self.n, self.seen, self.g, self.task = 0, 0, None, None

    def attach(self):
        self.n += 1
        return self

    def arrive(self):
        """(gradient left by the earlier producers or None, whether the caller is the last producer of this backward pass)"""
        task = _graph_task_id()
        if task != self.task:
            self.task, self.seen, self.g = task, 0, None
        self.seen += 1
        g, self.g = self.g, None
        last = self.seen >= self.n
        if last:
            self.seen = 0
        return g, last

    def leave(self, g):
        self.g = g


# (a private torch hook, resolved once: on a build without it the residual sums go back to autograd's own add -- same bits)
_GRAPH_TASK_ID = getattr(torch._C, "_current_graph_task_id", None)
if _GRAPH_TASK_ID is None:
    FUSE_RES_ADD = False


def _graph_task_id():
    """id of the backward pass the calling autograd node runs in (-1 outside one)"""
    return _GRAPH_TASK_ID()


def grad_box(x):
    """a ``GradBox`` for the gradient of ``x`` when there will be one (and the folding is on), else None"""
    return GradBox() if (FUSE_RES_ADD and torch.is_grad_enabled() and x.requires_grad) else None


class PackedWeights:
    """GEMM images of one conv kernel, refreshed when the parameter changes.  ``wf`` / ``wd``: forward / data-gradient image
    (fp32 for the f32 kernels, 16-bit pieces for the split kernels); ``w_bound``: device scalar max |w| (f16x3).

    An optimizer step invalidates every image of the model at once, so the first stale ``get`` of a forward pass re-packs ALL
    known convolutions of that device in two launches (``mcdseg_conv_split_pack_weights_multi``) instead of ~3 per conv."""

    def __init__(self):
        self.key = None
        self.wf = None
        self.wd = None
        self.w_bound = None
        self.mpf = 0
        self._weight = None   # weakref to the parameter, for group packs
        self._dims = None     # (Cout, Cin, taps, direct-stem?, math)
        _PACK_REGISTRY.add(self)

    @staticmethod
    def _key_of(weight):
        return (weight.data_ptr(), weight._version, WEIGHT_EPOCH, getattr(weight, "_mcd_epoch", 0), weight.device, CONV_MATH)

    def get(self, weight, desc, need_dgrad=True):
        key = self._key_of(weight)
        if key != self.key or (need_dgrad and self.wd is None):
            if GROUP_PACK and self._dims is not None and self.wd is not None and self._dims[4] == CONV_MATH:
                _pack_group(weight.device)
            if self._key_of(weight) != self.key or (need_dgrad and self.wd is None):
                self._pack_single(weight, desc)
        return self.wf, self.wd, self.mpf

    def _pack_single(self, weight, desc):
        L = lib()
        mpf, kpf, mpd, kpd = (ctypes.c_int32() for _ in range(4))
        check(L.mcdseg_conv_packed_dims(ctypes.byref(desc), mpf, kpf, mpd, kpd), "conv_packed_dims")
        taps = desc.KH * desc.KW
        w = _req(weight.detach(), "conv weight")
        dev = w.device
        fsp, dsp = _use_split(desc.Cin), _use_split(desc.Cout)
        direct = STEM_DIRECT and CONV_MATH in MATH_ID and bool(L.mcdseg_conv_split_direct_ok(ctypes.byref(desc)))
        if direct:
            fsp = True  # the stem: direct convolution on the split path although it contracts < 16 channels
        # f32 images (kept for whichever direction does not run on the split path)
        self.wf = None if fsp else torch.empty(taps * kpf.value * mpf.value, dtype=torch.float32, device=dev)
        self.wd = None if dsp else torch.empty(taps * kpd.value * mpd.value, dtype=torch.float32, device=dev)
        if self.wf is not None or self.wd is not None:
            check(L.mcdseg_conv_pack_weights(ctypes.byref(desc), _p(w), _p(self.wf), _p(self.wd), _stream()), "conv_pack_weights")
        self._dims = None
        if fsp or dsp:
            mid = MATH_ID[CONV_MATH]
            fb, db = ctypes.c_int64(), ctypes.c_int64()
            check(L.mcdseg_conv_split_packed_bytes(ctypes.byref(desc), mid, fb, db), "conv_split_packed_bytes")
            if fsp:
                self.wf = torch.empty(fb.value // 2, dtype=torch.int16, device=dev)
            if dsp:
                self.wd = torch.empty(db.value // 2, dtype=torch.int16, device=dev)
            if _scaled() and self.w_bound is None:
                self.w_bound = torch.empty(1, dtype=torch.float32, device=dev)
            with _timed("pack_weights_split_kernel", (0, 4 * w.numel() + (fb.value if fsp else 0) + (db.value if dsp else 0))):
                check(L.mcdseg_conv_split_pack_weights(ctypes.byref(desc), mid, _p(w), _p(self.wf) if fsp else None,
                                                       _p(self.wd) if dsp else None, _p(self.w_bound) if _scaled() else None,
                                                       _stream()), "conv_split_pack_weights")
            if fsp and dsp:  # both images on the split path: eligible for the group pack from now on
                self._weight = weakref.ref(weight)
                self._dims = (desc.Cout, desc.Cin, taps, direct, CONV_MATH)
        self.mpf = mpf.value
        self.key = self._key_of(weight)


def _pack_group(device):
    """re-pack every known split-path convolution on ``device`` whose image is stale: one table-driven absmax launch and one
    table-driven pack launch (plus the stem's own forward image)"""
    L = lib()
    members = []
    for pw in list(_PACK_REGISTRY):
        w = pw._weight() if pw._weight is not None else None
        if (w is None or pw._dims is None or pw._dims[4] != CONV_MATH or w.device != device or pw.wd is None or pw.wf is None
                or pw._dims[2] > 72):  # (the table-driven pack kernel stages up to 72 taps per tile)
            continue
        members.append((pw, w))
    if len(members) < 2:
        return
    members.sort(key=lambda m: m[1].data_ptr())
    sig = tuple((w.data_ptr(), pw.wf.data_ptr(), pw.wd.data_ptr()) + pw._dims for pw, w in members)
    tab = _PACK_TABLES.get((device, CONV_MATH))
    if tab is None or tab["sig"] != sig:
        n = len(members)
        bounds = torch.zeros(n, dtype=torch.float32, device=device)
        ptrs, dims = [], []
        for i, (pw, w) in enumerate(members):
            ptrs += [w.data_ptr(), pw.wf.data_ptr(), pw.wd.data_ptr(), bounds.data_ptr() + 4 * i]
            dims += [pw._dims[0], pw._dims[1], pw._dims[2], 0]
        tab = dict(sig=sig, bounds=bounds, ptrs=torch.tensor(ptrs, dtype=torch.int64).to(device),
                   dims=torch.tensor(dims, dtype=torch.int32).to(device))
        _PACK_TABLES[(device, CONV_MATH)] = tab
    if _scaled():
        for i, (pw, _) in enumerate(members):
            pw.w_bound = tab["bounds"][i:i + 1]
    n = len(members)
    nbytes = sum(4 * w.numel() + 2 * pw.wf.numel() + 2 * pw.wd.numel() for pw, w in members)
    with _timed("pack_weights_multi_kernel", (0, nbytes)):
        check(L.mcdseg_conv_split_pack_weights_multi(_p(tab["ptrs"]), _p(tab["dims"]), n, MATH_ID[CONV_MATH],
                                                     _p(tab["bounds"]) if _scaled() else None, _stream()), "conv_split_pack_weights_multi")
    for pw, w in members:
        if pw._dims[3]:  # the stem's direct-kernel forward image (behind the standard one) has its own layout
            d = ConvDesc(1, pw._dims[1], 8, 8, pw._dims[0], 7, 7, 1, 3, 1, 8, 8)
            check(L.mcdseg_conv_split_pack_weights(ctypes.byref(d), MATH_ID[CONV_MATH], _p(w.detach()), _p(pw.wf), None,
                                                   _p(pw.w_bound) if _scaled() else None, _stream()), "conv_split_pack_weights")
        pw.key = PackedWeights._key_of(w)


def _is_split(w_image):
    """packed image of the split kernels (16-bit pieces) rather than of the f32 kernels"""
    return w_image.dtype == torch.int16


# ------------------------------------------------------------------------------------------------ raw launchers
# The kernels address one conv operand with 32-bit buffer offsets, so a single launch takes operands below 2 GiB
# (+ a tile of slack).  Larger batches (cfg5: 32 x 720 x 1280) are cut along N on the host: conv is independent per
# image, the BN partial rows of the pieces are simply concatenated, weight gradients of the pieces are summed.
MAX_CONV_BYTES = int(os.environ.get("MCDSEG_MAX_CONV_BYTES", str((1 << 31) - (1 << 26))))


def _sub_desc(desc, n, ncb=0):
    """descriptor of ``n`` images of the batch; ``ncb``: batch size of the tensor the companions were written for (Ncb of mcdseg.h)"""
    return ConvDesc(n, desc.Cin, desc.H, desc.W, desc.Cout, desc.KH, desc.KW, desc.stride, desc.pad, desc.dil, desc.Ho, desc.Wo, ncb)


def _cb_slice(cb, a, channels, hw):
    """pointer to image ``a`` in piece 0 of a companion [piece][N][C/8][HW][8 x 16 bit] (None stays None)"""
    return None if cb is None else ctypes.c_void_p(cb.data_ptr() + a * (channels // 8) * hw * 16)


_PIECES_CACHE = {}


def _batch_pieces(desc, wgrad_cb=None):
    """[(first image, end)] of the launches a convolution's batch is cut into.  ``wgrad_cb``: None for the forward pass and the data
    gradient; for the weight gradient, whether both pre-split companions will be passed.  (Memoised: a group asks several times per pass,
    and the weight gradient's rule queries the library.)"""
    from . import _lib
    key = (desc.N, desc.Cin, desc.H, desc.W, desc.Cout, desc.KH, desc.KW, desc.stride, desc.pad, desc.dil, wgrad_cb, MAX_CONV_BYTES, CONV_MATH,
           _lib.OPTION_EPOCH)
    hit = _PIECES_CACHE.get(key)
    if hit is None:
        if len(_PIECES_CACHE) > 4096:
            _PIECES_CACHE.clear()
        hit = _PIECES_CACHE[key] = _batch_pieces_uncached(desc, wgrad_cb)
    return hit


def _batch_pieces_uncached(desc, wgrad_cb=None):
    # the f32 kernels express padding and ragged channel tails as offsets the buffer range check rejects, up to a 128-channel tile
    # past the tensor: (N*C + 128) * H*W * 4 < 2 GiB per operand -- the tile of slack is per LAUNCH, not per image (charging it per
    # image cut the full-resolution 16-channel layers in two and lost their pre-split operands).  The split kernels mark such
    # accesses with an explicit out-of-range offset instead, so the forward pass and the data gradient of a layer that runs on them
    # (both channel counts multiples of 8, at least 16) need no slack: BASELINE config 5's 16- and 32-channel layers at
    # 32 x 720 x 1280 (1.89 GB per tensor) stay in one launch and keep their companions.  The WEIGHT gradient is slack-free only on
    # the plans that read both companions (mcdseg_conv_wgrad_fits states the library's own rule): the f32 plans of thin layers, the
    # split plan without companions and bf16x6's thin layers still gather fp32 values with the slack.
    if wgrad_cb and CONV_MATH in MATH_ID and desc.Cin <= 16:
        # the thin layers' window weight gradient (csrc/conv_wgrad_thin_tr.hip) reads both companions of the WHOLE batch through one
        # descriptor each and nothing else: the library's own rule decides (round 6: the stem at BASELINE config 5's N = 32 -- a 1.9 GB dz,
        # two 0.9 GB pieces of its companion -- was cut by the fp32 kernels' slack rule and fell back to the f32 tap-packed kernel: 37 ms)
        m = MATH_ID[CONV_MATH]
        if lib().mcdseg_conv_wgrad_variant(ctypes.byref(desc), m, 1) == 15 and lib().mcdseg_conv_wgrad_fits(ctypes.byref(desc), m, 1):
            return [(0, desc.N)]
    split_only = CONV_MATH in MATH_ID and desc.Cin % 8 == 0 and desc.Cout % 8 == 0 and min(desc.Cin, desc.Cout) >= 16
    if wgrad_cb is not None:
        math = MATH_ID.get(CONV_MATH, 0)
        split_only = split_only and bool(wgrad_cb) and lib().mcdseg_conv_wgrad_variant(ctypes.byref(desc), math, 1) >= 11
    slack = 0 if split_only else 128
    step = desc.N
    for c, hw in ((desc.Cin, desc.H * desc.W), (desc.Cout, desc.Ho * desc.Wo)):
        step = min(step, max(1, (MAX_CONV_BYTES - 4 * slack * hw) // (4 * c * hw)))
    if wgrad_cb is not None:  # (the variant can change with N: hold every piece to the library's rule)
        while step > 1 and not lib().mcdseg_conv_wgrad_fits(ctypes.byref(_sub_desc(desc, step, desc.N if wgrad_cb else 0)), math, int(wgrad_cb)):
            step -= 1
    if step >= desc.N:
        return [(0, desc.N)]
    return [(i, min(i + step, desc.N)) for i in range(0, desc.N, step)]


def _conv_fprop(desc, x, wf, bias, want_stats, mpf, x_cb=None, x_bound=None, w_bound=None):
    L = lib()
    y = torch.empty((desc.N, desc.Cout, desc.Ho, desc.Wo), dtype=torch.float32, device=x.device)
    pieces = _batch_pieces(desc)
    # a batch cut along N keeps its companion: piece p of a slice lies p * N * (C/8) * HW * 16 bytes behind its piece 0 (Ncb)
    descs = [desc if len(pieces) == 1 else _sub_desc(desc, b - a, desc.N if x_cb is not None else 0) for a, b in pieces]
    split = _is_split(wf)
    part, rows, row_off = None, 0, [0]
    if want_stats:
        for d in descs:
            row_off.append(row_off[-1] + (L.mcdseg_conv_split_stat_rows_for(ctypes.byref(d), MATH_ID[CONV_MATH], int(x_cb is not None))
                                          if split else L.mcdseg_conv_stat_rows(ctypes.byref(d))))
        rows = row_off[-1]
        part = torch.empty(rows * 3 * mpf, dtype=torch.float32, device=x.device)
    direct = split and bool(L.mcdseg_conv_split_direct_ok(ctypes.byref(desc)))
    if split and not direct:
        x_bound = _bound_or_measure(x, x_bound)
    for i, ((a, b), d) in enumerate(zip(pieces, descs)):
        pp = None if part is None else ctypes.c_void_p(part.data_ptr() + 4 * row_off[i] * 3 * mpf)
        def name(d=d):  # (formed only when a launch timer asks: it costs two queries of the library's plan)
            return (_window_name(d, x_cb is not None, False) if split else None) \
                or gemm_kernel_name(desc.Cout, desc.Cin, False, split, x_cb is not None, direct, d.N * d.Ho * d.Wo)
        if split:
            _split_launches(d, x_cb is not None, False, name, lambda part: check(L.mcdseg_conv_split_fprop_part(
                ctypes.byref(d), MATH_ID[CONV_MATH], _p(_sl(x, a, b)), _cb_slice(x_cb, a, desc.Cin, desc.H * desc.W), _p(x_bound), _p(wf),
                _p(w_bound), _p(bias), _p(y[a:b]), pp, part, _stream()), "conv_split_fprop"))
        else:
            with _timed(name() if LAUNCH_TIMER is not None else "", conv_work(d)):
                check(L.mcdseg_conv_fprop(ctypes.byref(d), _p(x[a:b]), _p(wf), _p(bias), _p(y[a:b]), pp, _stream()), "conv_fprop")
    return y, part, rows


def _sl(t, a, b):
    return None if t is None else t[a:b]


def _window_name(d, presplit, dgrad):
    """rocprofv3's name of the LDS-window kernel when it takes this thin 3x3 geometry (csrc/conv_thin_window.hip), else None"""
    if not lib().mcdseg_conv_split_window_ok(ctypes.byref(d), MATH_ID.get(CONV_MATH, 0), int(presplit), int(dgrad)):
        return None
    m = d.Cin if dgrad else d.Cout
    stem = d.KH * d.KW == 49
    return "conv_thin_window_kernel<%d, %d, %d, %s>" % (1 if stem else 2, m // 16, 13 if stem else 5, "true" if dgrad else "false")


def _conv_dgrad(desc, dy, wd, dy_cb=None, dy_bound=None, w_bound=None, addend=None):
    """``dy`` may be None when its pre-split companion is given and the batch is not cut (the kernel reads only ``dy_cb``).
    ``addend``: another gradient of the same input (``GradBox``); the result is data gradient + addend -- in the kernel's epilogue
    where the kernel can (``mcdseg_conv_split_dgrad_add``), by an element-wise add otherwise: the same bits either way."""
    L = lib()
    if addend is not None:
        addend = _req(addend, "gradient addend")
    dx = torch.empty((desc.N, desc.Cin, desc.H, desc.W), dtype=torch.float32, device=(dy if dy is not None else dy_cb).device)
    pieces = _batch_pieces(desc)
    split = _is_split(wd)
    if split and dy is not None:
        dy_bound = _bound_or_measure(dy, dy_bound)
    for a, b in pieces:
        d = desc if (a, b) == (0, desc.N) else _sub_desc(desc, b - a, desc.N if dy_cb is not None else 0)
        def name(d=d):
            return (_window_name(d, dy_cb is not None, True) if split else None) \
                or gemm_kernel_name(desc.Cin, desc.Cout, True, split, dy_cb is not None, False, d.N * d.H * d.W)
        if split and addend is not None and _window_name(d, dy_cb is not None, True) is None:
            _split_launches(d, dy_cb is not None, True, name, lambda part: check(L.mcdseg_conv_split_dgrad_add(
                ctypes.byref(d), MATH_ID[CONV_MATH], _p(_sl(dy, a, b)), _cb_slice(dy_cb, a, desc.Cout, desc.Ho * desc.Wo), _p(dy_bound),
                _p(wd), _p(w_bound), _p(addend[a:b]), _p(dx[a:b]), part, _stream()), "conv_split_dgrad_add"))
            continue
        if split:
            _split_launches(d, dy_cb is not None, True, name, lambda part: check(L.mcdseg_conv_split_dgrad_part(
                ctypes.byref(d), MATH_ID[CONV_MATH], _p(_sl(dy, a, b)), _cb_slice(dy_cb, a, desc.Cout, desc.Ho * desc.Wo), _p(dy_bound),
                _p(wd), _p(w_bound), _p(dx[a:b]), part, _stream()), "conv_split_dgrad"))
        else:
            with _timed(name() if LAUNCH_TIMER is not None else "", conv_work(d)):
                check(L.mcdseg_conv_dgrad(ctypes.byref(d), _p(dy[a:b]), _p(wd), _p(dx[a:b]), _stream()), "conv_dgrad")
        if addend is not None:
            dx[a:b].add_(addend[a:b])
    return dx


def _batch_pieces_half(desc):
    """[(first image, end)] of the launches of a convolution of the 2-byte chain: a launch addresses one piece of its pre-split operand
    through a 32-bit buffer resource (< 2 GiB, 2 bytes per element); the 16-bit output is addressed with 64-bit pointers"""
    step = desc.N
    for c, hw in ((desc.Cin, desc.H * desc.W), (desc.Cout, desc.Ho * desc.Wo)):
        step = min(step, max(1, MAX_CONV_BYTES // (2 * c * hw)))
    if step >= desc.N:
        return [(0, desc.N)]
    return [(i, min(i + step, desc.N)) for i in range(0, desc.N, step)]


def half_conv_work(d):
    """(algorithmic FLOPs, algorithmic bytes) of one conv pass of the 2-byte chain: 16-bit operand in, 16-bit result out"""
    flops, byts = conv_work(d)
    return flops, byts // 2


def _unit_slice(t, a, channels, hw):
    """pointer to image ``a`` of a 16-bit tensor in the unit layout [N][C/8][HW][8]"""
    return None if t is None else ctypes.c_void_p(t.data_ptr() + a * channels * hw * 2)


def _conv_fprop_half(desc, x_cb, x_bound, wf, w_bound, mpf):
    """forward convolution of the 2-byte chain: (z16 units [int16], z_bound scalar, BatchNorm partial rows, row count)"""
    L = lib()
    dev = x_cb.device
    z16 = torch.empty(desc.N * desc.Cout * desc.Ho * desc.Wo, dtype=torch.int16, device=dev)
    z_bound = torch.empty(1, dtype=torch.float32, device=dev)
    pieces = _batch_pieces_half(desc)
    descs = [desc if len(pieces) == 1 else _sub_desc(desc, b - a, desc.N) for a, b in pieces]
    row_off = [0]
    for d in descs:
        row_off.append(row_off[-1] + L.mcdseg_conv_split_stat_rows_for(ctypes.byref(d), MATH_ID[CONV_MATH], 1))
    rows = row_off[-1]
    part = torch.empty(rows * 3 * mpf, dtype=torch.float32, device=dev)
    for i, ((a, b), d) in enumerate(zip(pieces, descs)):
        pp = ctypes.c_void_p(part.data_ptr() + 4 * row_off[i] * 3 * mpf)
        def name(d=d):
            return gemm_kernel_name(desc.Cout, desc.Cin, False, True, True, False, d.N * d.Ho * d.Wo)
        _split_launches(d, True, False, name, lambda part_no: check(L.mcdseg_conv_split_fprop_half(
            ctypes.byref(d), MATH_ID[CONV_MATH], _cb_slice(x_cb, a, desc.Cin, desc.H * desc.W), _p(x_bound), _p(wf), _p(w_bound),
            _unit_slice(z16, a, desc.Cout, desc.Ho * desc.Wo), _p(z_bound), pp, part_no, _stream()), "conv_split_fprop_half"), work=half_conv_work)
    return z16, z_bound, part, rows


def _conv_dgrad_half(desc, dy_cb, dy_bound, wd, w_bound, addend16=None):
    """data gradient of the 2-byte chain: bf16 units (+ the other gradient of the same tensor, same layout, in the kernel's epilogue)"""
    L = lib()
    dx = torch.empty((desc.N, desc.Cin, desc.H, desc.W), dtype=torch.bfloat16, device=dy_cb.device)
    if addend16 is not None:
        addend16 = _req(addend16, "gradient addend", torch.bfloat16)
    for a, b in _batch_pieces_half(desc):
        d = desc if (a, b) == (0, desc.N) else _sub_desc(desc, b - a, desc.N)
        def name(d=d):
            return gemm_kernel_name(desc.Cin, desc.Cout, True, True, True, False, d.N * d.H * d.W)
        _split_launches(d, True, True, name, lambda part_no: check(L.mcdseg_conv_split_dgrad_half(
            ctypes.byref(d), MATH_ID[CONV_MATH], _cb_slice(dy_cb, a, desc.Cout, desc.Ho * desc.Wo), _p(dy_bound), _p(wd), _p(w_bound),
            _unit_slice(addend16, a, desc.Cin, desc.H * desc.W), _unit_slice(dx, a, desc.Cin, desc.H * desc.W), part_no, _stream()),
            "conv_split_dgrad_half"), work=half_conv_work)
    return dx


def _cb_wanted(channels):
    """Emit the pre-split (channel-blocked) companion of a tensor with this many channels?  Only when the conv that gathers
    it runs on the split path (contraction >= 16 channels) and the layout applies (multiple of 8)."""
    return PRESPLIT and _use_split(channels) and channels % 8 == 0


def _cb_alloc(n, c, hw, device):
    return torch.empty(PIECES[CONV_MATH] * n * c * hw, dtype=torch.int16, device=device)


def split_companion(x, bound=None):
    """pre-split companion of an fp32 NCHW tensor no fused BN group produced: (companion, bound scalar) -- (None, None) when
    the layout does not apply"""
    x = _req(x, "tensor to split")
    n, c, h, w = x.shape
    if not _cb_wanted(c) or n * (c // 8) > 65535:
        return None, None
    bound = _bound_or_measure(x, bound)
    cb = _cb_alloc(n, c, h * w, x.device)
    check(lib().mcdseg_split_cb(_p(x), _p(cb), _p(bound), MATH_ID[CONV_MATH], n, c, h * w, _stream()), "split_cb")
    return cb, bound


def split_companion_padded(x, bound=None):
    """companion of a tensor whose channel count is not a multiple of 8 (the 6-channel network input): ceil(C/8) channel groups,
    zeros in the missing channels -- read by the stem's window-forward and weight-gradient kernels.  Cached on the tensor object (same guard
    as ``_mcd_cb``): one MCD step back-propagates through the stem several times with the same batch."""
    rec = getattr(x, "_mcd_cbp", None)
    if rec is not None and rec[2] == x._version and rec[3] == x.data_ptr() and rec[4] == CONV_MATH:
        return rec[0], rec[1]
    x = _req(x, "tensor to split")
    n, c, h, w = x.shape
    bound = _bound_or_measure(x, bound)
    cb = torch.empty(PIECES[CONV_MATH] * n * ((c + 7) // 8) * 8 * h * w, dtype=torch.int16, device=x.device)
    check(lib().mcdseg_split_cb_padded(_p(x), _p(cb), _p(bound), MATH_ID[CONV_MATH], n, c, h * w, _stream()), "split_cb_padded")
    x._mcd_cbp = (cb, bound, x._version, x.data_ptr(), CONV_MATH)
    return cb, bound


# mcdseg_conv_wgrad_variant code -> the kernel name rocprofv3 prints
_WGRAD_NAMES = {10: "conv_wgrad_split_kernel<%s>", 11: "conv_wgrad_split_cb_kernel<%s>", 12: "conv_wgrad_split_tr_kernel<%s, 2, 2, 3, false>",
                13: "conv_wgrad_split_tr_kernel<%s, 4, 2, 3, false>", 14: "conv_wgrad_split_tr64_kernel<%s>",
                16: "conv_wgrad_split_tr_kernel<%s, 4, 2, 3, true>",
                17: "conv_wgrad_split_pp_kernel<%s>", 18: "conv_wgrad_split_pp3_kernel<%s>"}


def wgrad_split_kernel_name(d, have_cb):
    """rocprofv3's name of the split-arithmetic weight-gradient kernel the library launches for this geometry"""
    v = lib().mcdseg_conv_wgrad_variant(ctypes.byref(d), MATH_ID[CONV_MATH], int(have_cb))
    if v == 15:  # csrc/conv_wgrad_thin_tr.hip: <channel groups of the input, row tiles, column tiles, tile rows> (no policy argument)
        cfg = lib().mcdseg_conv_wgrad_thin_tr_config(ctypes.byref(d))  # (the library's own choice, not a restatement of it)
        return "conv_wgrad_thin_tr_kernel<%d, %d, %d, %d>" % (cfg // 1000000, cfg // 10000 % 100, cfg // 100 % 100, cfg % 100)
    policy = POLICY[CONV_MATH]
    if v == 17 and CONV_MATH == "f16x1" and get_option("WGRAD_PP_DEEP"):
        policy = "SplitF16x1D"  # six logical stages (csrc/conv_wgrad_split_pp.hip)
    return _WGRAD_NAMES.get(v, "conv_wgrad<%s>") % policy


def _wgrad_thin_tr(desc):
    """the thin-layer window kernel (csrc/conv_wgrad_thin_tr.hip) takes this geometry when both companions exist"""
    return _scaled() and desc.Cin <= 16 and lib().mcdseg_conv_wgrad_variant(ctypes.byref(desc), MATH_ID[CONV_MATH], 1) == 15


def _wgrad_split_plan(desc, have_cb=False):
    """a split-arithmetic plan of csrc/conv_wgrad.hip applies (else the f32 kernels run): the 128x128 plan for
    min(Cin, Cout) > 64, and -- from both pre-split companions only, f16x3 -- the 64-channel tap-pair plan for 32 < min <= 64
    and the window kernel of the thin 3x3 layers"""
    if CONV_MATH not in MATH_ID:
        return False
    if desc.Cin <= 16 and desc.KH * desc.KW > 1:
        return bool(have_cb and _wgrad_thin_tr(desc))
    lo = min(desc.Cout, desc.Cin)
    if lo > 64:
        return True
    return get_option("WGRAD_TR64") != 0 and have_cb and _scaled() and lo > 16 and desc.Cin % 8 == 0 and desc.Cout % 8 == 0


def _conv_wgrad(desc, x, dy, x_cb=None, dy_cb=None, x_bound=None, dy_bound=None):
    L = lib()
    total = None
    if x_cb is None or dy_cb is None:
        x_cb = dy_cb = None  # both companions or none
    pieces = _batch_pieces(desc, wgrad_cb=x_cb is not None)
    if x_cb is not None and len(pieces) > 1 and desc.Cin <= 16:
        x_cb = dy_cb = None  # the thin layers' window kernel takes whole batches only: the cut batch gathers fp32 values (with the slack)
        pieces = _batch_pieces(desc, wgrad_cb=False)
    split = _wgrad_split_plan(desc, x_cb is not None)
    if split and x_cb is None:
        x_bound, dy_bound = _bound_or_measure(x, x_bound), _bound_or_measure(dy, dy_bound)
    for a, b in pieces:
        d = desc if (a, b) == (0, desc.N) else _sub_desc(desc, b - a, desc.N if x_cb is not None else 0)
        ws = _ws(L.mcdseg_conv_wgrad_workspace_bytes(ctypes.byref(d)), (x if x is not None else x_cb).device)
        dw = torch.empty((desc.Cout, desc.Cin, desc.KH, desc.KW), dtype=torch.float32, device=ws.device)
        name = "" if LAUNCH_TIMER is None else (wgrad_split_kernel_name(d, x_cb is not None) if split else
                                                 wgrad_kernel_name(desc.Cout, desc.Cin, desc.KH * desc.KW))
        with _timed(name, conv_work(d)):
            if split:
                check(L.mcdseg_conv_split_wgrad(ctypes.byref(d), MATH_ID[CONV_MATH], _p(_sl(x, a, b)), _cb_slice(x_cb, a, desc.Cin, desc.H * desc.W),
                                                _p(x_bound), _p(_sl(dy, a, b)), _cb_slice(dy_cb, a, desc.Cout, desc.Ho * desc.Wo), _p(dy_bound),
                                                _p(dw), _p(ws), ctypes.c_size_t(ws.numel() * 4), _stream()), "conv_split_wgrad")
            else:
                check(L.mcdseg_conv_wgrad(ctypes.byref(d), _p(x[a:b]), _p(dy[a:b]), _p(dw), _p(ws), ctypes.c_size_t(ws.numel() * 4),
                                          _stream()), "conv_wgrad")
        _on_launch_stream(ws, dw)
        if total is None:
            total = dw
        else:
            with torch.cuda.stream(_LAUNCH.stream or torch.cuda.current_stream()):
                total.add_(dw)
    return total


# dgrad and wgrad of one layer are independent.  MCDSEG_OVERLAP_WGRAD:
#   "0"  both on the current stream;
#   "1"  wgrad on a second HIP stream, joined right behind dgrad (round 1; measured +1 %: both kernels are matrix-bound);
#   "2"  (default) wgrad on the second stream and NOT joined until the backward pass is nearly over: the weight gradient of layer k
#        then also runs beside the HBM-bound BatchNorm backward of layer k-1, which a power-metered matrix kernel hides
#        (tools/probes/overlap_probe.py: 20 forward launches + 20 bn_apply passes take as long as the 20 forward launches alone;
#        -4 % of the cfg2 step).  Correct by construction: a deferred gradient is handed to autograd through a ``_LateGrad``
#        identity node that sits between the parameter and the convolution and makes the main stream wait for the side stream
#        BEFORE it passes the gradient on -- nothing downstream (the engine's summation of several gradients of one weight,
#        AccumulateGrad's in-place add, hooks, torch.autograd.grad's capture) ever sees an unfinished tensor.  WHEN the wait
#        happens is a matter of the engine's ready queue (latest-created node first): ``late_weight_grads`` creates the identity
#        nodes at the start of a trunk's forward pass, so they run at the end of its backward pass; a convolution called outside
#        such a context gets no identity node and runs its weight gradient on the main stream.
OVERLAP_WGRAD = os.environ.get("MCDSEG_OVERLAP_WGRAD", "2")
if OVERLAP_WGRAD not in ("0", "1", "2"):
    raise ValueError("MCDSEG_OVERLAP_WGRAD must be 0, 1 or 2, got %r" % OVERLAP_WGRAD)
_SIDE = {}
_PENDING = {}  # device index -> (event behind the last weight gradient on the side stream, the stream that has to wait for it)


def _side_stream(device):
    # (a stream of HIP's lowest priority -- hipStreamCreateWithPriority through ctypes, wrapped as torch.cuda.ExternalStream -- was
    # measured too: 235.1 vs 236.2 ms on one box, 236.1 / 234.6 vs 234.8 / 235.0 on another; within the noise, not kept)
    key = device.index if device.index is not None else torch.cuda.current_device()
    if key not in _SIDE:
        _SIDE[key] = _masked_stream(device, SIDE_CUS) if SIDE_CUS > 0 else torch.cuda.Stream(device=device)
    return _SIDE[key]


# MCDSEG_SIDE_CUS=n (experiment, round 5): the side stream as a HIP stream restricted to the first n compute units
# (hipExtStreamCreateWithCUMask), so that the main stream's HBM-bound BatchNorm passes always find CUs no weight-gradient workgroup
# holds.  0 (default): an ordinary stream.
SIDE_CUS = int(os.environ.get("MCDSEG_SIDE_CUS", "0"))


def _masked_stream(device, n_cus):
    hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
    total = torch.cuda.get_device_properties(device).multi_processor_count
    words = (total + 31) // 32
    mask = (ctypes.c_uint32 * words)()
    for b in range(min(n_cus, total)):
        mask[b // 32] |= 1 << (b % 32)
    handle = ctypes.c_void_p()
    with torch.cuda.device(device):
        rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(handle), ctypes.c_uint32(words), mask)
    if rc != 0 or not handle.value:
        raise RuntimeError("hipExtStreamCreateWithCUMask failed (%d)" % rc)
    return torch.cuda.ExternalStream(handle.value, device=device)


def join_side_streams(device_index=None):
    """The consumer stream of a device -- the stream its backward pass ran on when the weight gradients were deferred, recorded with
    the event -- waits for the weight gradients still running on that device's side stream.  A ``_LateGrad`` node joins its own
    device only (under nn.DataParallel every device has its own autograd thread, whose current stream says nothing about another
    device's); the final callback of a backward pass that deferred one joins all of them.  Safe at any time."""
    for key in ([device_index] if device_index is not None else list(_PENDING)):
        rec = _PENDING.pop(key, None)
        if rec is not None:
            rec[1].wait_event(rec[0])
            # backward passes of two sub-graphs may run on two streams (MFNet's encoders, solvers/solver.py) and share the side stream:
            # whoever joins waits too, whichever stream recorded the event last
            cur = torch.cuda.current_stream(rec[1].device)
            if cur != rec[1]:
                cur.wait_event(rec[0])
            _HELD.pop(key, None)  # (after the wait: the operands of the deferred launches go back to the allocator)


# ---- two independent forward passes side by side (MCD step B: the generator on the source and on the target batch, same weights).
# A forward pass on ONE stream alternates matrix-bound convolutions with HBM-bound BatchNorm passes and nothing runs beside either;
# two passes on two streams fill each other's gaps.  The only state they share is each BatchNorm's running statistics, which the
# reference updates source-first (adapt_trainer.py:187-200): the pass that leads (source, side stream) records an event behind every
# statistics launch, the pass that follows (target, main stream) waits for the layer's event before it touches the layer -- the same
# updates in the same order (and the lazily re-packed weight images of the leading pass are complete before the follower reads
# them).  MCDSEG_OVERLAP_STEPB=0: one after the other.
OVERLAP_STEPB = os.environ.get("MCDSEG_OVERLAP_STEPB", "1") != "0"
_FWD_SYNC = None  # "lead" / "follow" inside ForwardFork's contexts
_FWD_LEAD_DONE = None  # event behind the last launch of the leading pass


class ForwardFork:
    def __init__(self, device):
        self.device = device
        self.main = torch.cuda.current_stream(device)
        self.side = _side_stream(device)

    @contextlib.contextmanager
    def lead(self):
        """the pass that goes first in the reference's order: on the side stream, without a tape"""
        global _FWD_SYNC, _FWD_LEAD_DONE
        self.side.wait_stream(self.main)
        prev, _FWD_SYNC = _FWD_SYNC, "lead"
        try:
            with torch.cuda.stream(self.side), torch.no_grad():
                yield
            _FWD_LEAD_DONE = torch.cuda.Event()
            _FWD_LEAD_DONE.record(self.side)
        finally:
            _FWD_SYNC = prev

    @contextlib.contextmanager
    def follow(self):
        global _FWD_SYNC
        prev, _FWD_SYNC = _FWD_SYNC, "follow"
        try:
            yield
        finally:
            _FWD_SYNC = prev

    def join(self, tensors):
        """the main stream waits for the leading pass; its results (allocated from the side stream's pool) now belong to both"""
        global _FWD_LEAD_DONE
        _FWD_LEAD_DONE = None
        self.main.wait_stream(self.side)
        for t in tensors:
            t.record_stream(self.main)
            for c in (getattr(t, "_mcd_cb", None) or ())[:2]:
                if torch.is_tensor(c):
                    c.record_stream(self.main)


def forward_fork(device):
    """a ``ForwardFork`` when two forward passes may run side by side on ``device`` now, else None"""
    if not OVERLAP_STEPB or device.type != "cuda" or _LAUNCH.stream is not None:
        return None
    if LAUNCH_TIMER is not None and LAUNCH_TIMER.wants("conv_wgrad"):
        return None  # a step whose launches are bracketed by HIP events runs every kernel alone
    if not _room_to_defer(device):
        return None
    return ForwardFork(device)


def _fwd_sync_wait(anchor):
    ev = getattr(anchor, "_mcd_fwd_ev", None)
    if ev is not None:
        torch.cuda.current_stream().wait_event(ev)
        anchor._mcd_fwd_ev = None


def _fwd_sync_record(anchor):
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream())
    anchor._mcd_fwd_ev = ev


class _LateGrad(torch.autograd.Function):
    """identity on a convolution weight; its backward makes the main stream wait for the side stream, then passes the gradient on"""

    @staticmethod
    def forward(ctx, w):
        return w.view_as(w)

    @staticmethod
    def backward(ctx, g):
        join_side_streams(g.device.index)
        return g


class late_weight_grads:
    """Context around the forward pass of a trunk (``root``: an nn.Module): one ``_LateGrad`` alias per convolution weight, created
    NOW -- before every other node of this forward pass, hence executed after all of them in the backward pass -- and consumed by
    the first ``conv_bn_act`` call on that convolution.  Nothing is prepared without grad mode, for weights that do not require a
    gradient, for weights with foreign post-accumulate hooks (they want their gradients DURING the pass; FlatSGD's bucketed
    all-reduce takes deferred gradients through its ``_mcd_grad_sink`` instead) or when MCDSEG_OVERLAP_WGRAD is not "2"."""

    def __init__(self, root):
        self.root, self.mods = root, []

    def __enter__(self):
        if OVERLAP_WGRAD == "2" and torch.is_grad_enabled():
            for m in self.root.modules():
                if isinstance(getattr(m, "_packed", None), PackedWeights) and getattr(m, "_w_late", None) is None:
                    w = m.weight
                    # (a weight with post-accumulate hooks wants its gradient DURING the pass -- unless the hook's owner also left a
                    # ``_mcd_grad_sink`` on it, which receives a deferred gradient on the side stream: _conv_backward)
                    if w.is_cuda and w.requires_grad and (not getattr(w, "_post_accumulate_grad_hooks", None) or
                                                          getattr(w, "_mcd_grad_sink", None) is not None):
                        alias = _LateGrad.apply(w)
                        alias._mcd_param = w  # (the packed images are keyed by the parameter, see PackedWeights._key_of)
                        m._w_late = alias
                        self.mods.append(m)
        return self

    def __exit__(self, *exc):
        for m in self.mods:
            m._w_late = None


def _take_late(conv):
    """the weight tensor a fused group hands to autograd: the prepared ``_LateGrad`` alias (once), else the parameter itself"""
    alias = getattr(conv, "_w_late", None)
    if alias is None or not torch.is_grad_enabled():
        return conv.weight
    conv._w_late = None
    return alias


# What a deferred weight gradient reads and writes stays allocated until the main stream has waited for it.  At most MAX_LAG
# launches are left behind: before the next one is enqueued the main stream waits for the oldest (the side stream is rarely more
# than a layer or two behind, so this wait is almost always already satisfied) and its operands are released.  And nothing is
# deferred while the allocator holds more than DEFER_MEM_FRACTION of the device's memory (cfg5 at N = 32: 232 of 288 GB).
MAX_LAG = int(os.environ.get("MCDSEG_OVERLAP_WGRAD_LAG", "4"))
DEFER_MEM_FRACTION = float(os.environ.get("MCDSEG_OVERLAP_WGRAD_MEM", "0.6"))
DEFER_RESERVED_FRACTION = float(os.environ.get("MCDSEG_OVERLAP_WGRAD_RESERVED", "0.85"))
WGRAD_STREAM_STATS = {"deferred": 0, "no_room": 0}  # launches left on the side stream / kept on the main stream for lack of memory
_TOTAL_MEM = {}
_HELD = {}  # device index -> deque of (event behind a deferred weight gradient, [tensors it uses])


def _room_to_defer(device):
    key = device.index if device.index is not None else torch.cuda.current_device()
    if key not in _TOTAL_MEM:
        _TOTAL_MEM[key] = torch.cuda.get_device_properties(key).total_memory
    # memory IN USE decides (with memory_reserved alone, cached-but-free blocks left by one large pass would latch deferral off for the
    # rest of the run) -- under a higher ceiling on what the allocator HOLDS: the side stream's pool (ForwardFork's leading pass) caches
    # blocks that cannot serve the main stream and that memory_allocated does not see; past the ceiling the allocator is one large
    # request away from its free-everything-and-retry path (14-18 s per step at BASELINE config 5)
    return (torch.cuda.memory_allocated(key) < DEFER_MEM_FRACTION * _TOTAL_MEM[key]
            and torch.cuda.memory_reserved(key) < DEFER_RESERVED_FRACTION * _TOTAL_MEM[key])


def _dgrad_any(desc, dy, wd, dy_cb, dy_bound, w_bound, addend, dx16):
    """the data gradient in the format its consumer is owed: fp32 NCHW, or -- ``dx16``: the convolution's input is an activation of the
    2-byte chain -- bf16 units, straight from the kernel's epilogue where it has one, converted otherwise"""
    if not dx16:
        return _conv_dgrad(desc, dy, wd, dy_cb, dy_bound, w_bound, addend)
    if dy_cb is not None and _is_split(wd) and lib().mcdseg_conv_split_half_ok(ctypes.byref(desc), MATH_ID.get(CONV_MATH, 0), 1):
        return _conv_dgrad_half(desc, dy_cb, dy_bound, wd, w_bound, addend)
    dx = pack_bf16_units(_conv_dgrad(desc, dy, wd, dy_cb, dy_bound, w_bound, None))
    return dx if addend is None else dx + addend


def _conv_backward(desc, x, dy, wd, need_dx, need_dw, dy_cb=None, x_cb=None, dy_bound=None, x_bound=None, w_bound=None, defer=False, param=None,
                   dx_addend=None, dx16=False):
    """(dx, dw).  ``defer``: dw goes to a ``_LateGrad`` node, so mode "2" may leave it on the side stream.  ``param``: the parameter
    behind that node; when its optimizer has left a ``_mcd_grad_sink`` on it (FlatSGD's bucketed all-reduce, MCDSEG_DP_OVERLAP=1) a
    deferred gradient is handed to the sink ON THE SIDE STREAM, right behind its kernels -- the exchange of a bucket then starts when
    its last weight gradient has been enqueued, long before the ``_LateGrad`` nodes pass the gradients on at the end of the pass."""
    mode = OVERLAP_WGRAD
    sink = getattr(param, "_mcd_grad_sink", None) if param is not None else None
    if mode == "2" and not (defer and x_cb is not None and dy_cb is not None):
        mode = "0"  # (without companions the weight gradient measures bounds and caches them on tensors the main stream reads)
    if mode != "0" and LAUNCH_TIMER is not None and LAUNCH_TIMER.wants("conv_wgrad"):
        mode = "0"  # a step whose launches are bracketed by HIP events runs every kernel alone, so that the pairs time kernels
    # (nothing is deferred while the allocator is close to its limit -- BASELINE config 5 -- also for a weight whose optimizer exchanges
    # gradients in buckets: its gradient then reaches the sink as "not early" (below), which spoils the bucket's early start on THIS rank
    # only; the collectives still start in bucket order on every rank, FlatSGD._drain)
    if mode == "2" and not _room_to_defer(x.device):
        mode = "0"
        WGRAD_STREAM_STATS["no_room"] += 1
    if not (need_dx and need_dw and mode != "0"):
        if sink is not None and need_dw:
            sink(param, None)  # this contribution to the weight's gradient reaches p.grad WITHOUT passing through the sink
        return ((_dgrad_any(desc, dy, wd, dy_cb, dy_bound, w_bound, dx_addend, dx16) if need_dx else None),
                (_conv_wgrad(desc, x, dy, x_cb, dy_cb, x_bound, dy_bound) if need_dw else None))
    main = torch.cuda.current_stream()
    side = _side_stream(x.device)
    key = side.device.index
    if _scaled() and dy is not None:
        dy_bound = _bound_or_measure(dy, dy_bound)  # measured once, on the main stream, for both consumers
    # (launching wgrad only once dgrad has finished -- so that it would run beside the next BatchNorm backward from its first
    # workgroup on -- is slower: 243 vs 237.6 ms per step; the weight gradient fills the data gradient's partial rounds as it is)
    side.wait_stream(main)
    _LAUNCH.stream, _LAUNCH.keep = side, [x, dy, x_cb, dy_cb, x_bound, dy_bound]
    try:
        dw = _conv_wgrad(desc, x, dy, x_cb, dy_cb, x_bound, dy_bound)
    finally:
        keep, _LAUNCH.stream, _LAUNCH.keep = _LAUNCH.keep, None, None
    dx = _dgrad_any(desc, dy, wd, dy_cb, dy_bound, w_bound, dx_addend, dx16)
    if mode == "1":
        main.wait_stream(side)
        if sink is not None:
            sink(param, None)
        return dx, dw
    if sink is not None:
        with torch.cuda.stream(side):
            sink(param, dw)
    WGRAD_STREAM_STATS["deferred"] += 1
    ev = torch.cuda.Event()
    ev.record(side)
    _PENDING[key] = (ev, main)
    held = _HELD.setdefault(key, collections.deque())
    held.append((ev, keep))
    while len(held) > MAX_LAG:
        main.wait_event(held.popleft()[0])  # (its tensors are dropped with the tuple: the main stream is behind their last use now)
    torch.autograd.Variable._execution_engine.queue_callback(join_side_streams)  # (backstop; the _LateGrad node has waited by then)
    return dx, dw


def _channel_reduce(dy, y, z, mean, rstd, relu, gamma=None, want_bound=False, train=True, y_cb=None, zmask_beta=None, rmask=None):
    """(dgamma, dbeta) of a BN (z given) or just the per-channel sum of dy (z None); with ``want_bound`` also the device
    scalar bounding |dz| of the tensor bn_bwd_apply will write from these sums (include/mcdseg.h).  ``zmask_beta``: the group
    has a ReLU and no residual, so the mask is recomputed from z and ``y`` is not read."""
    L = lib()
    n, c = dy.shape[0], dy.shape[1]
    hw = dy.shape[2] * dy.shape[3]
    ws = _ws(L.mcdseg_bn_bwd_workspace_bytes(n, c, hw), dy.device)
    dgamma = torch.empty(c, dtype=torch.float32, device=dy.device) if z is not None else None
    dbeta = torch.empty(c, dtype=torch.float32, device=dy.device)
    bound = torch.empty(1, dtype=torch.float32, device=dy.device) if want_bound else None
    if rmask is not None:  # a ReLU group with residual: the mask from its bit-plane (``RELU_MASK``), y is not read
        with _timed("bn_bwd_reduce", (0, 4 * n * c * hw * 2)):
            check(L.mcdseg_bn_bwd_reduce_mask(_p(dy), _p(rmask), _p(z), _p(mean), _p(rstd), _p(gamma) if want_bound else None, _p(dgamma),
                                              _p(dbeta), _p(bound), int(train), n, c, hw, _p(ws), ctypes.c_size_t(ws.numel() * 4), _stream()),
                  "bn_bwd_reduce_mask")
        return dgamma, dbeta, bound
    if zmask_beta is not None:
        with _timed("bn_bwd_reduce", (0, 4 * n * c * hw * 2)):
            check(L.mcdseg_bn_bwd_reduce_zmask(_p(dy), _p(z), _p(mean), _p(rstd), _p(gamma), _p(zmask_beta), _p(dgamma), _p(dbeta), _p(bound),
                                               int(train), n, c, hw, _p(ws), ctypes.c_size_t(ws.numel() * 4), _stream()),
                  "bn_bwd_reduce_zmask")
        return dgamma, dbeta, bound
    with _timed("bn_bwd_reduce", (0, 4 * n * c * hw * (1 + (y is not None) + (z is not None)))):
        check(L.mcdseg_bn_bwd_reduce(_p(dy), _p(y), _p(y_cb) if y is None else None, MATH_ID.get(CONV_MATH, 0), _p(z), _p(mean), _p(rstd),
                                     _p(dgamma), _p(dbeta), _p(gamma) if want_bound else None, _p(bound), int(train), n, c, hw,
                                     int(relu), _p(ws), ctypes.c_size_t(ws.numel() * 4), _stream()), "bn_bwd_reduce")
    return dgamma, dbeta, bound


DEBUG_TAPE = None  # a list: _ConvBNAct.backward appends its intermediate tensors (development only)


# ------------------------------------------------------------------------------------------------ conv + BN + act
class _ConvBNAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, gamma, beta, residual, conv_bias, running_mean, running_var, nbt, packed, geom, training,
                momentum, eps, relu, x_cb, x_bound, res_bound, aux):
        """aux: dict(x_virtual, res_virtual, res_cb, compact) -- compact activation storage (see ACT_STORAGE)"""
        L = lib()
        if _FWD_SYNC == "follow":
            _fwd_sync_wait(gamma)  # (ForwardFork: the leading pass is through with this layer)
        stride, pad, dil = geom
        desc = conv_desc(x.shape, weight.shape, stride, pad, dil)
        wf, wd, mpf = packed.get(getattr(weight, "_mcd_param", weight), desc)
        w_bound = packed.w_bound
        x_virtual = aux["x_virtual"]
        x_half = bool(aux.get("x_half"))  # (decided on the tensor autograd knows, before a consumer of fp32 materializes it below)
        # (a batch cut along N keeps its companions -- the split kernels take slices, mcdseg.h Ncb -- except on the thin layers'
        # window kernels, Cin <= 16)
        uncut = len(_batch_pieces(desc)) == 1
        if x_virtual and not (_is_split(wf) and x_cb is not None and (uncut or desc.Cin > 16)):
            x, x_virtual = materialize(x, x_cb, x_bound), False  # this consumer reads fp32
        if not x_virtual:
            x = _req(x, "conv input")
        if _is_split(wf) and _scaled():
            x_bound = _bound_or_measure(x, x_bound)
        f_cb = x_cb
        if (STEM_WINDOW and _is_split(wf) and x_cb is None and conv_bias is None and PRESPLIT and desc.Cin % 8 != 0 and not x_virtual
                and len(_batch_pieces(desc)) == 1 and L.mcdseg_conv_split_window_ok(ctypes.byref(desc), MATH_ID[CONV_MATH], 1, 0)):
            # the stem: forward on the LDS-window kernel from the zero-padded companion of the network input (cached on the
            # tensor: the same batch goes through the stem several times per MCD step, forward and weight gradient)
            f_cb, x_bound = split_companion_padded(x, x_bound)
        c = desc.Cout
        hw = desc.Ho * desc.Wo
        # the 2-byte chain (HALF_STORAGE): this group keeps z, y, dz and the gradients it hands on as one 16-bit value per element
        # (not the thin layers, Cin <= 16: their window weight gradient multiplies BOTH pieces of dz's companion whatever the arithmetic)
        half = bool(aux.get("half") and training and _is_split(wf) and x_cb is not None and conv_bias is None and _cb_wanted(c) and desc.Cin > 16
                    and desc.N * (c // 8) <= 65535 and not aux.get("no_cb")
                    and (residual is None or (aux["res_virtual"] and aux["res_cb"] is not None and residual.dtype == torch.bfloat16))
                    and L.mcdseg_conv_split_half_ok(ctypes.byref(desc), MATH_ID[CONV_MATH], 0) and _wgrad_split_plan(desc, True))
        if half:
            return _ConvBNAct._forward_half(ctx, L, desc, x, weight, gamma, beta, residual, running_mean, running_var, nbt, packed, wf, wd, mpf,
                                            w_bound, momentum, eps, relu, x_cb, x_bound, res_bound, aux)
        z, part, rows = _conv_fprop(desc, x, wf, _req(conv_bias, "conv bias"), training, mpf, f_cb, x_bound, w_bound)
        mean = torch.empty(c, dtype=torch.float32, device=z.device)
        rstd = torch.empty(c, dtype=torch.float32, device=z.device)
        # the pre-split companion of y: the scaled arithmetic needs |y|'s bound BEFORE y is written -- train-mode statistics
        # give one (Samuelson), eval-mode running statistics do not (the consumer then measures y)
        want_cb = _cb_wanted(c) and desc.N * (c // 8) <= 65535 and (training or not _scaled()) and not aux.get("no_cb")
        compact = aux["compact"] and want_cb and training
        if compact and aux.get("single_piece_only") and c <= 32 and not (aux.get("thin_ok") and c >= 16 and len(_batch_pieces(desc)) == 1):
            compact = False  # the consumer would run its weight gradient on the f32 kernels and read fp32
        res_cb = aux["res_cb"] if aux["res_virtual"] else None
        if aux["res_virtual"] and not want_cb:
            residual, res_cb = materialize(residual, aux["res_cb"], res_bound), None  # plain bn_apply reads fp32
        elif not aux["res_virtual"]:
            residual = _req(residual, "residual")
        has_res = residual is not None
        y_bound = None
        if training and _scaled() and _use_split(c):  # with or without a companion: consumers that split y themselves use it too
            y_bound = torch.empty(1, dtype=torch.float32, device=z.device)
            if has_res:
                res_bound = _bound_or_measure(residual, res_bound)
        if training:
            track = running_mean is not None
            ws = torch.empty(L.mcdseg_bn_stats_workspace_bytes(rows, c) // 8 + 1, dtype=torch.float64, device=z.device)
            with _timed("bn_stats_finalize", (0, 12 * rows * mpf)):  # (one launch also when it stands for BN_RUNNING_REPEAT forward passes)
                check(L.mcdseg_bn_stats_finalize(_p(part), rows, c, mpf, _p(mean), _p(rstd), _p(running_mean) if track else None,
                                                 _p(running_var) if track else None, _p(nbt) if track else None,
                                                 float(momentum), float(eps), _p(gamma) if y_bound is not None else None,
                                                 _p(beta) if y_bound is not None else None,
                                                 _p(res_bound) if (y_bound is not None and has_res) else None, _p(y_bound),
                                                 int(BN_RUNNING_REPEAT if track else 1), _p(ws), ctypes.c_size_t(ws.numel() * 8), _stream()),
                      "bn_stats_finalize")
            if _FWD_SYNC == "lead":
                _fwd_sync_record(gamma)
        else:
            check(L.mcdseg_bn_eval_stats(_p(running_mean), _p(running_var), c, float(eps), _p(mean), _p(rstd), _stream()),
                  "bn_eval_stats")
        y = _virtual(z.shape, z.device) if compact else torch.empty_like(z)
        y_cb = None
        elems = desc.N * c * hw
        rmask = None
        if (want_cb and RELU_MASK and relu and has_res and training and not compact and res_cb is None and any(ctx.needs_input_grad)
                and (z.data_ptr() | y.data_ptr() | residual.data_ptr()) % 16 == 0):
            nbytes = L.mcdseg_bn_relu_mask_bytes(desc.N, c, hw)
            if nbytes > 0:  # the ReLU bit-plane for this group's backward pass (see RELU_MASK)
                rmask = torch.empty(nbytes // 8, dtype=torch.int64, device=z.device)
        if rmask is not None:
            y_cb = _cb_alloc(desc.N, c, hw, z.device)
            with _timed("bn_apply_cb", (0, elems * (4 + 4 + 2 * PIECES[CONV_MATH] + 4) + rmask.numel() * 8)):
                check(L.mcdseg_bn_apply_cb_mask(_p(z), _p(mean), _p(rstd), _p(gamma), _p(beta), _p(residual), _p(y), _p(y_cb), _p(y_bound),
                                                _p(rmask), MATH_ID[CONV_MATH], desc.N, c, hw, _stream()), "bn_apply_cb_mask")
        elif want_cb:
            y_cb = _cb_alloc(desc.N, c, hw, z.device)
            with _timed("bn_apply_cb", (0, elems * (4 + (0 if compact else 4) + 2 * PIECES[CONV_MATH] + (4 if has_res else 0)))):
                check(L.mcdseg_bn_apply_cb(_p(z), _p(mean), _p(rstd), _p(gamma), _p(beta), _p(residual) if res_cb is None else None,
                                           _p(res_cb), _p(res_bound) if res_cb is not None else None, None if compact else _p(y),
                                           _p(y_cb), _p(y_bound), MATH_ID[CONV_MATH], desc.N, c, hw, int(relu), _stream()), "bn_apply_cb")
        else:
            with _timed("bn_apply", (0, elems * (8 + (4 if has_res else 0)))):
                check(L.mcdseg_bn_apply(_p(z), _p(mean), _p(rstd), _p(gamma), _p(beta), _p(residual), _p(y), desc.N, c, hw, int(relu),
                                        _stream()), "bn_apply")
        ctx.desc, ctx.wd, ctx.relu, ctx.training, ctx.has_res = desc, wd, relu, training, has_res
        ctx.w_bound = w_bound
        ctx.packed, ctx.pack_key = packed, packed.key  # the data-gradient image is shared and re-packed in place: see backward
        ctx.defer_ok = hasattr(weight, "_mcd_param")  # the weight came through a _LateGrad alias (late_weight_grads)
        ctx.w_param = getattr(weight, "_mcd_param", None)
        if ctx.w_param is None and getattr(weight, "_mcd_grad_sink", None) is not None:
            ctx.w_param = weight  # (no alias -- a second use of the module in one graph: its gradient is reported to the sink as "not early")
        ctx.in_box, ctx.res_box = aux.get("in_box"), aux.get("res_box")
        ctx.has_bias = conv_bias is not None
        ctx.x_cb, ctx.x_bound = x_cb, x_bound  # wgrad reads the input's split companion too (an input of this node: safe to hold)
        ctx.x_virtual, ctx.compact = x_virtual, compact
        ctx.rmask = rmask
        ctx.half = False
        ctx.x_half = x_half        # the gradients owed to activations of the 2-byte chain: bf16 units
        ctx.res_half = bool(aux.get("res_half"))
        ctx.save_for_backward(x, z, y, mean, rstd, gamma, y_cb if compact else None, y_bound if compact else None, beta)
        ctx.set_materialize_grads(False)  # no zero-filled "gradient" for the non-differentiable companions
        for t in (y_cb, y_bound):
            if t is not None:
                ctx.mark_non_differentiable(t)
        return y, y_cb, y_bound

    @staticmethod
    def _forward_half(ctx, L, desc, x, weight, gamma, beta, residual, running_mean, running_var, nbt, packed, wf, wd, mpf, w_bound, momentum,
                      eps, relu, x_cb, x_bound, res_bound, aux):
        """the group in the 2-byte chain (HALF_STORAGE; include/mcdseg.h "2-byte activation storage"): train mode, pre-split input"""
        c, hw = desc.Cout, desc.Ho * desc.Wo
        dev = x_cb.device
        has_res = residual is not None
        z16, z_bound, part, rows = _conv_fprop_half(desc, x_cb, x_bound, wf, w_bound, mpf)
        mean = torch.empty(c, dtype=torch.float32, device=dev)
        rstd = torch.empty(c, dtype=torch.float32, device=dev)
        y_bound = torch.empty(1, dtype=torch.float32, device=dev)
        track = running_mean is not None
        ws = torch.empty(L.mcdseg_bn_stats_workspace_bytes(rows, c) // 8 + 1, dtype=torch.float64, device=dev)
        with _timed("bn_stats_finalize", (0, 12 * rows * mpf)):
            check(L.mcdseg_bn_stats_finalize(_p(part), rows, c, mpf, _p(mean), _p(rstd), _p(running_mean) if track else None,
                                             _p(running_var) if track else None, _p(nbt) if track else None, float(momentum), float(eps),
                                             _p(gamma), _p(beta), _p(res_bound) if has_res else None, _p(y_bound),
                                             int(BN_RUNNING_REPEAT if track else 1), _p(ws), ctypes.c_size_t(ws.numel() * 8), _stream()),
                  "bn_stats_finalize")
        if _FWD_SYNC == "lead":
            _fwd_sync_record(gamma)
        y_cb = torch.empty(desc.N * c * hw, dtype=torch.int16, device=dev)   # ONE piece: the activation
        with _timed("bn_apply_half", (0, desc.N * c * hw * (4 + (2 if has_res else 0)))):
            check(L.mcdseg_bn_apply_half(_p(z16), _p(z_bound), _p(mean), _p(rstd), _p(gamma), _p(beta), _p(aux["res_cb"]) if has_res else None,
                                         _p(res_bound) if has_res else None, _p(y_cb), _p(y_bound), desc.N, c, hw, int(relu), _stream()),
                  "bn_apply_half")
        y = _virtual((desc.N, c, desc.Ho, desc.Wo), dev, torch.bfloat16)
        ctx.desc, ctx.wd, ctx.relu, ctx.training, ctx.has_res = desc, wd, relu, True, has_res
        ctx.w_bound = w_bound
        ctx.packed, ctx.pack_key = packed, packed.key
        ctx.defer_ok = hasattr(weight, "_mcd_param")
        ctx.w_param = getattr(weight, "_mcd_param", None)
        if ctx.w_param is None and getattr(weight, "_mcd_grad_sink", None) is not None:
            ctx.w_param = weight
        ctx.in_box, ctx.res_box = aux.get("in_box"), aux.get("res_box")
        ctx.has_bias = False
        ctx.x_cb, ctx.x_bound = x_cb, x_bound
        ctx.x_virtual, ctx.compact = aux["x_virtual"], True
        ctx.rmask = None
        ctx.half = True
        ctx.x_half = bool(aux.get("x_half"))
        ctx.res_half = has_res
        ctx.z_bound = z_bound
        ctx.save_for_backward(x, z16, y, mean, rstd, gamma, y_cb, y_bound, beta)
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(y_cb, y_bound)
        return y, y_cb, y_bound

    @staticmethod
    def _backward_half(ctx, dy):
        """backward of a group of the 2-byte chain: ``dy`` bf16 units -> dz as the leading companion piece, the residual's gradient as bf16
        units, then the convolution's gradients from companions alone"""
        L = lib()
        x, z16, y, mean, rstd, gamma, y_cb, y_bound, beta = ctx.saved_tensors
        desc = ctx.desc
        if ctx.packed.key != ctx.pack_key:
            raise RuntimeError("mcdseg: a convolution weight was modified between a forward pass and its backward pass "
                               "(an optimizer stepped the generator while its graph was still alive)")
        dy = _req(dy, "grad_output", torch.bfloat16)
        n, c, hw = desc.N, desc.Cout, desc.Ho * desc.Wo
        dev = dy.device
        mask = 0 if not ctx.relu else (4 if ctx.has_res else 2)
        ws = _ws(L.mcdseg_bn_bwd_half_workspace_bytes(n, c, hw), dev)
        dgamma = torch.empty(c, dtype=torch.float32, device=dev)
        dbeta = torch.empty(c, dtype=torch.float32, device=dev)
        dz_bound = torch.empty(1, dtype=torch.float32, device=dev)
        elems = n * c * hw
        with _timed("bn_bwd_reduce_half", (0, elems * (4 + (2 if mask == 4 else 0)))):
            check(L.mcdseg_bn_bwd_reduce_half(_p(dy), _p(y_cb) if mask == 4 else None, _p(z16), _p(ctx.z_bound), _p(mean), _p(rstd), _p(gamma),
                                              _p(beta), _p(dgamma), _p(dbeta), _p(dz_bound), mask, 1, n, c, hw, _p(ws),
                                              ctypes.c_size_t(ws.numel() * 4), _stream()), "bn_bwd_reduce_half")
        dz_cb = torch.empty(elems, dtype=torch.int16, device=dev)
        dres = None
        if ctx.has_res and ctx.needs_input_grad[4]:
            dres = torch.empty_like(dy) if ctx.relu else dy
        with _timed("bn_bwd_apply_half", (0, elems * (6 + (2 if mask == 4 else 0) + (2 if (dres is not None and ctx.relu) else 0)))):
            check(L.mcdseg_bn_bwd_apply_half(_p(dy), _p(y_cb) if mask == 4 else None, _p(z16), _p(ctx.z_bound), _p(mean), _p(rstd), _p(gamma),
                                             _p(beta), _p(dgamma), _p(dbeta), _p(dz_cb), _p(dz_bound),
                                             _p(dres) if (dres is not None and ctx.relu) else None, mask, 1, n, c, hw, _stream()),
                  "bn_bwd_apply_half")
        if ctx.res_box is not None and dres is not None:
            other, last = ctx.res_box.arrive()
            if last:
                dres = dres if other is None else dres + other
            else:
                ctx.res_box.leave(dres if other is None else dres + other)
                dres = None
        addend, dx_last = (None, True)
        if ctx.in_box is not None and ctx.needs_input_grad[0]:
            addend, dx_last = ctx.in_box.arrive()
        if ctx.needs_input_grad[1] and not (_wgrad_split_plan(desc, True) and desc.Cin % 8 == 0 and desc.Cout % 8 == 0):
            raise RuntimeError("mcdseg: a group of the 2-byte chain needs a weight-gradient plan that reads companions (%d -> %d channels)"
                               % (desc.Cin, desc.Cout))
        dx, dw = _conv_backward(desc, x, None, ctx.wd, ctx.needs_input_grad[0], ctx.needs_input_grad[1], dz_cb, ctx.x_cb, dz_bound,
                                ctx.x_bound, ctx.w_bound, defer=ctx.defer_ok, param=ctx.w_param, dx_addend=addend, dx16=ctx.x_half)
        if not dx_last:
            ctx.in_box.leave(dx)
            dx = None
        return (dx, dw, dgamma if ctx.needs_input_grad[2] else None, dbeta if ctx.needs_input_grad[3] else None, dres, None,
                None, None, None, None, None, None, None, None, None, None, None, None, None)

    @staticmethod
    def backward(ctx, dy, _dcb=None, _dbound=None):
        L = lib()
        if dy is None:
            return (None,) * 19
        if ctx.half:
            return _ConvBNAct._backward_half(ctx, dy)
        x, z, y, mean, rstd, gamma, y_cb, y_bound, beta = ctx.saved_tensors
        desc = ctx.desc
        if ctx.packed.key != ctx.pack_key:
            # the weight was updated in place (and its packed image re-packed) between this forward and its backward: the data gradient
            # would use the NEW weights -- what torch reports for a saved tensor as "modified by an inplace operation"
            raise RuntimeError("mcdseg: a convolution weight was modified between a forward pass and its backward pass "
                               "(an optimizer stepped the generator while its graph was still alive)")
        dy = _req(dy, "grad_output")
        n, c, hw = desc.N, desc.Cout, desc.Ho * desc.Wo
        split_d = _is_split(ctx.wd)
        x_cb, x_bound = ctx.x_cb, ctx.x_bound
        # the stem: no input gradient and an input without a companion, but its weight gradient runs on split operands too
        # (conv_wgrad_thin_tr.hip) -- from the zero-padded companion of the network input and the companion of dz
        stem_tr = (ctx.needs_input_grad[1] and not ctx.needs_input_grad[0] and x_cb is None and desc.Cin % 8 != 0 and PRESPLIT
                   and not ctx.x_virtual and len(_batch_pieces(desc, wgrad_cb=True)) == 1 and c % 8 == 0 and n * (c // 8) <= 65535
                   and _wgrad_thin_tr(desc))
        if stem_tr:
            x_cb, x_bound = split_companion_padded(x)
        want_cb = ctx.needs_input_grad[0] or (ctx.needs_input_grad[1] and x_cb is not None)
        use_cb = stem_tr or (want_cb and split_d and _cb_wanted(c) and n * (c // 8) <= 65535)
        # a ReLU group without residual: y > 0 <=> fma(z, gamma rstd, beta - mean gamma rstd) > 0 -- neither pass below reads y
        zmask = BN_ZMASK and ctx.relu and not ctx.has_res and (y_cb is None or use_cb)
        if ctx.compact and not use_cb:  # the plain backward kernels read the fp32 activation for the ReLU mask
            y, y_cb = materialize(y, y_cb, y_bound), None
        y_mask = (y if y_cb is None else None) if ctx.relu else None
        # a ReLU group with residual: the mask from the bit-plane the forward pass wrote, where the four-pixel backward kernel will run
        rmask = ctx.rmask if (ctx.relu and use_cb and not zmask and dy.data_ptr() % 16 == 0) else None
        dgamma, dbeta, dz_bound = _channel_reduce(dy, y_mask, z, mean, rstd, ctx.relu, gamma,
                                                  want_bound=_scaled() and (split_d or stem_tr or _wgrad_split_plan(desc)), train=ctx.training,
                                                  y_cb=y_cb if ctx.relu else None, zmask_beta=beta if zmask else None, rmask=rmask)
        dz = None
        dres = None
        if ctx.has_res and ctx.needs_input_grad[4]:
            dres = torch.empty_like(z) if ctx.relu else dy
        dz_cb = None
        # the fp32 dz is skipped when every consumer reads the split companion: dgrad (pre-split gather) and wgrad
        # (pre-split plans); a conv bias gradient or any fallback path still needs it
        # (cut batches keep their companions, see forward; the stem's window weight gradient takes the whole batch's companions whatever
        # the forward pass was cut into)
        single = single_piece = stem_tr or len(_batch_pieces(desc)) == 1 or desc.Cin > 16
        wgrad_cb = stem_tr or (x_cb is not None and use_cb and single_piece and _wgrad_split_plan(desc, True) and desc.Cin % 8 == 0
                               and desc.Cout % 8 == 0)
        skip_dz = (use_cb and single and not (ctx.has_bias and ctx.needs_input_grad[5])
                   and (not ctx.needs_input_grad[1] or wgrad_cb))
        if not skip_dz:
            dz = torch.empty_like(z)
        rd = 4 * n * c * hw * (2 + int(ctx.relu)) + 4 * n * c * hw * ((dz is not None) + (dres is not None and ctx.relu))
        if use_cb and zmask:
            dz_cb = _cb_alloc(n, c, hw, dy.device)
            with _timed("bn_bwd_apply_cb", (0, rd - 4 * n * c * hw + 2 * PIECES[CONV_MATH] * n * c * hw)):
                check(L.mcdseg_bn_bwd_apply_cb_zmask(_p(dy), _p(z), _p(mean), _p(rstd), _p(gamma), _p(beta), _p(dgamma), _p(dbeta), _p(dz),
                                                     _p(dz_cb), _p(dz_bound), MATH_ID[CONV_MATH], n, c, hw, int(ctx.training), _stream()),
                      "bn_bwd_apply_cb_zmask")
        elif use_cb and rmask is not None:
            dz_cb = _cb_alloc(n, c, hw, dy.device)
            with _timed("bn_bwd_apply_cb", (0, rd - 4 * n * c * hw + 2 * PIECES[CONV_MATH] * n * c * hw)):
                check(L.mcdseg_bn_bwd_apply_cb_mask(_p(dy), _p(rmask), _p(z), _p(mean), _p(rstd), _p(gamma), _p(dgamma), _p(dbeta), _p(dz),
                                                    _p(dres) if dres is not None else None, _p(dz_cb), _p(dz_bound), MATH_ID[CONV_MATH], n, c,
                                                    hw, int(ctx.training), _stream()), "bn_bwd_apply_cb_mask")
        elif use_cb:
            dz_cb = _cb_alloc(n, c, hw, dy.device)
            with _timed("bn_bwd_apply_cb", (0, rd + 2 * PIECES[CONV_MATH] * n * c * hw)):
                check(L.mcdseg_bn_bwd_apply_cb(_p(dy), _p(y_mask), _p(y_cb) if (ctx.relu and y_mask is None) else None, _p(z), _p(mean),
                                               _p(rstd), _p(gamma), _p(dgamma), _p(dbeta), _p(dz),
                                               _p(dres) if (dres is not None and ctx.relu) else None, _p(dz_cb), _p(dz_bound),
                                               MATH_ID[CONV_MATH], n, c, hw, int(ctx.relu), int(ctx.training), _stream()),
                      "bn_bwd_apply_cb")
        else:
            with _timed("bn_bwd_apply", (0, rd)):
                check(L.mcdseg_bn_bwd_apply(_p(dy), _p(y_mask), _p(z), _p(mean), _p(rstd), _p(gamma), _p(dgamma), _p(dbeta), _p(dz),
                                            _p(dres) if (dres is not None and ctx.relu) else None, n, c, hw, int(ctx.relu),
                                            int(ctx.training), _stream()), "bn_bwd_apply")
        if ctx.x_virtual and ctx.needs_input_grad[1] and not (wgrad_cb and dz_cb is not None and single):
            x = materialize(x, ctx.x_cb, ctx.x_bound)  # the weight gradient falls back to a kernel that reads fp32
        if DEBUG_TAPE is not None:  # kernel development (tools/op_contention.py): what the BN backward handed to the conv backward
            DEBUG_TAPE.append(dict(dy=dy, dz=dz, dz_cb=dz_cb, dz_bound=dz_bound, dgamma=dgamma, dbeta=dbeta, dres=dres, shape=(n, c, hw), z=z,
                                   y=y_mask, mean=mean, rstd=rstd, gamma=gamma, beta=beta, zmask=zmask))
        # the gradients this group shares with another producer (GradBox): the shortcut's goes into its box -- or comes back summed when
        # this group happens to be the last -- and the data gradient takes the box's tensor into its epilogue
        if dres is not None and ctx.res_half:
            dres = pack_bf16_units(dres)  # (the residual is an activation of the 2-byte chain: its gradient is owed as bf16 units)
        if ctx.res_box is not None and dres is not None:
            other, last = ctx.res_box.arrive()
            if last:
                dres = dres if other is None else dres + other
            else:
                ctx.res_box.leave(dres if other is None else dres + other)
                dres = None
        addend, dx_last = (None, True)
        if ctx.in_box is not None and ctx.needs_input_grad[0]:
            addend, dx_last = ctx.in_box.arrive()
        dx, dw = _conv_backward(desc, x, dz, ctx.wd, ctx.needs_input_grad[0], ctx.needs_input_grad[1], dz_cb, x_cb, dz_bound,
                                x_bound, ctx.w_bound, defer=ctx.defer_ok, param=ctx.w_param, dx_addend=addend, dx16=ctx.x_half)
        if not dx_last:
            ctx.in_box.leave(dx)
            dx = None
        dbias = None
        if ctx.has_bias and ctx.needs_input_grad[5]:
            # a bias in front of train-mode BN has zero gradient up to rounding (BN removes the channel mean);
            # it is still formed, as autograd does in the reference (CBR, models/dilated_fcn.py:632-644)
            _, dbias, _ = _channel_reduce(dz, None, None, None, None, False)
        return (dx, dw, dgamma if ctx.needs_input_grad[2] else None, dbeta if ctx.needs_input_grad[3] else None, dres, dbias,
                None, None, None, None, None, None, None, None, None, None, None, None, None)


def _conv_bn_act_inference(x, conv, bn, relu, residual):
    """eval-mode BN folded into the conv epilogue: one kernel, no z / bn_apply pass, nothing kept for backward"""
    L = lib()
    x = _req(materialize(x), "conv input")
    residual = _req(materialize(residual) if residual is not None else None, "residual")
    desc = conv_desc(x.shape, conv.weight.shape, conv.stride[0], conv.padding[0], conv.dilation[0])
    wf, _, _ = conv._packed.get(conv.weight, desc)
    c = desc.Cout
    scale = torch.empty(c, dtype=torch.float32, device=x.device)
    shift = torch.empty(c, dtype=torch.float32, device=x.device)
    check(L.mcdseg_bn_eval_affine(_p(bn.weight), _p(bn.bias), _p(bn.running_mean), _p(bn.running_var), _p(conv.bias), c,
                                  float(bn.eps), _p(scale), _p(shift), _stream()), "bn_eval_affine")
    y = torch.empty((desc.N, c, desc.Ho, desc.Wo), dtype=torch.float32, device=x.device)
    split = _is_split(wf)
    direct = split and bool(L.mcdseg_conv_split_direct_ok(ctypes.byref(desc)))
    x_bound = _bound_or_measure(x, None) if (split and not direct) else None
    for a, b in _batch_pieces(desc):
        d = desc if (a, b) == (0, desc.N) else _sub_desc(desc, b - a)
        with _timed(gemm_kernel_name(desc.Cout, desc.Cin, False, split, False, direct), conv_work(d)):
            args = (_p(scale), _p(shift), _p(residual[a:b]) if residual is not None else None, int(relu), _p(y[a:b]), _stream())
            if split:
                check(L.mcdseg_conv_split_fprop_affine(ctypes.byref(d), MATH_ID[CONV_MATH], _p(x[a:b]), None, _p(x_bound), _p(wf),
                                                       _p(conv._packed.w_bound), *args), "conv_split_fprop_affine")
            else:
                check(L.mcdseg_conv_fprop_affine(ctypes.byref(d), _p(x[a:b]), _p(wf), *args), "conv_fprop_affine")
    return y


def conv_bn_act(x, conv, bn, relu=True, residual=None, internal=False, in_box=None, res_box=None, thin_ok=False, shortcut_only=False):
    """y = act(bn(conv(x)) + residual) with the HIP kernels; ``conv``/``bn`` are the parameter-holding modules.
    ``internal``: the caller promises that only the next fused group reads the result (see INTERNAL_SKIP_Y).
    ``shortcut_only``: the caller promises that the result is only ever read as the ``residual`` of another group (a block's 1x1
    projection shortcut): no convolution gathers it, so its pre-split companion is not written (fp32 storage; round 5).
    ``in_box`` / ``res_box``: the ``GradBox`` through which the gradient of ``x`` / of ``residual`` is summed with its other producer."""
    geom = (conv.stride[0], conv.padding[0], conv.dilation[0])
    training = bn.training
    if not training and not torch.is_grad_enabled() and bn.track_running_stats and bn.running_mean is not None:
        return _conv_bn_act_inference(x, conv, bn, relu, residual)
    track = bn.track_running_stats and bn.running_mean is not None
    if not training and not track:
        raise NotImplementedError("mcdseg: eval-mode BatchNorm needs running statistics")
    momentum = 0.1 if bn.momentum is None else bn.momentum
    x_cb, x_bound = _cb_of(x)
    if x_cb is not None and conv.in_channels <= 16 and x_cb.numel() == x.numel() and PIECES.get(CONV_MATH, 0) > 1:
        # (the thin layers' window kernels multiply BOTH pieces of their operand whatever the arithmetic; no network of the reference
        # feeds a 16-channel convolution from inside the 2-byte chain, whose groups all have more than 16 input channels)
        raise RuntimeError("mcdseg: a convolution of %d input channels cannot consume a one-piece (2-byte chain) activation" % conv.in_channels)
    res_cb, res_bound = _cb_of(residual) if residual is not None else (None, None)
    skip_y = internal and INTERNAL_SKIP_Y and BN_ZMASK and relu and residual is None and _scaled()
    grads = torch.is_grad_enabled()
    aux = dict(x_virtual=is_virtual(x), res_virtual=is_virtual(residual), res_cb=res_cb, compact=_compact_now() or skip_y,
               single_piece_only=skip_y and not _compact_now(), thin_ok=thin_ok,
               no_cb=bool(shortcut_only and SHORTCUT_NO_CB and not relu and residual is None and not _compact_now()),
               half=_half_now(), x_half=is_half(x), res_half=is_half(residual),
               in_box=in_box.attach() if (in_box is not None and grads and x.requires_grad) else None,
               res_box=res_box.attach() if (res_box is not None and grads and residual is not None and residual.requires_grad) else None)
    y, y_cb, y_bound = _ConvBNAct.apply(x, _take_late(conv), bn.weight, bn.bias, residual, conv.bias, bn.running_mean if track else None,
                                        bn.running_var if track else None, bn.num_batches_tracked if track else None, conv._packed,
                                        geom, training, momentum, bn.eps, relu, x_cb, x_bound, res_bound, aux)
    if y_cb is not None or y_bound is not None:
        _attach_cb(y, y_cb, y_bound)  # the pre-split companion (and the bound) travel with the tensor object to the next convolution
    if y.stride(0) == 0 and y.numel() > 1:
        y._mcd_virtual = True  # compact storage: the companion IS the activation
    return y


def _attach_cb(y, y_cb, y_bound=None):
    # the companion (and the bound) is only valid for the values y holds NOW: remember the autograd version counter and the storage
    y._mcd_cb = (y_cb, y_bound, y._version, y.data_ptr())


def _cb_of(x):
    """(pre-split companion, bound scalar) attached by the producer of ``x``: (None, None) if ``x`` did not come straight from a
    fused BN group, or if anything wrote to ``x`` since (an in-place op of the caller -- ``x.add_()``, ``relu_()``, inplace
    dropout -- bumps ``x._version``; the consumer then splits the current values itself instead of reading a stale image)"""
    rec = getattr(x, "_mcd_cb", None)
    if rec is None:
        return None, None
    cb, bound, version, ptr = rec
    if x._version != version or x.data_ptr() != ptr or not (x.is_contiguous() or is_virtual(x)):
        return None, None
    if cb is not None and cb.numel() != PIECES.get(CONV_MATH, 0) * x.numel() and not (CONV_MATH == "f16x1" and cb.numel() == x.numel()):
        cb = None  # (f16x1 reads the leading piece only: a one-piece companion of the 2-byte chain serves every consumer)
    return cb, (bound if _scaled() else None)


# ------------------------------------------------------------------------------------------------ conv (+bias)
class _Conv2dBias(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, packed, geom, x_cb, x_bound):
        x = _req(x, "conv input")
        desc = conv_desc(x.shape, weight.shape, *geom)
        wf, wd, mpf = packed.get(getattr(weight, "_mcd_param", weight), desc)
        if _is_split(wf) and _scaled():
            x_bound = _bound_or_measure(x, x_bound)
        y, _, _ = _conv_fprop(desc, x, wf, _req(bias, "conv bias"), False, mpf, x_cb, x_bound, packed.w_bound)
        ctx.desc, ctx.wd, ctx.has_bias, ctx.w_bound = desc, wd, bias is not None, packed.w_bound
        ctx.x_bound = x_bound
        ctx.packed, ctx.pack_key = packed, packed.key
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        if ctx.packed.key != ctx.pack_key:
            raise RuntimeError("mcdseg: a convolution weight was modified between a forward pass and its backward pass")
        dy = _req(dy, "grad_output")
        dx, dw = _conv_backward(ctx.desc, x, dy, ctx.wd, ctx.needs_input_grad[0], ctx.needs_input_grad[1], None, None, None, ctx.x_bound,
                                ctx.w_bound)
        db = None
        if ctx.has_bias and ctx.needs_input_grad[2]:
            _, db, _ = _channel_reduce(dy, None, None, None, None, False)
        return dx, dw, db, None, None, None, None


def conv2d_bias(x, conv):
    geom = (conv.stride[0], conv.padding[0], conv.dilation[0])
    if is_virtual(x):
        raise RuntimeError("mcdseg: a compact activation left its trunk (the trunk's last layer writes fp32)")
    x_cb, x_bound = _cb_of(x)
    if _FWD_SYNC == "follow" and _FWD_LEAD_DONE is not None:
        # a convolution without BatchNorm has no per-layer event: the follower of a ForwardFork runs it behind the whole leading pass
        torch.cuda.current_stream().wait_event(_FWD_LEAD_DONE)
    return _Conv2dBias.apply(x, conv.weight, conv.bias, conv._packed, geom, x_cb, x_bound)


# ------------------------------------------------------------------------------------------------ x8 up-sampler
def _up8_bwd_input(dy, w, n, c, hi, wi):
    dx = torch.empty((n, c, hi, wi), dtype=torch.float32, device=dy.device)
    with _timed("up8_bwd_input", (0, 4 * n * c * hi * wi * 65)):
        check(lib().mcdseg_up8_bwd_input(_p(dy), _p(w), _p(dx), n, c, hi, wi, _stream()), "up8_bwd_input")
    return dx


def _up8_bwd_weight(dy, x, n, c, hi, wi):
    L = lib()
    ws = _ws(L.mcdseg_up8_bwd_weight_workspace_bytes(n, c, hi, wi), dy.device)
    dw = torch.empty((c, 1, 16, 16), dtype=torch.float32, device=dy.device)
    with _timed("up8_bwd_weight", (0, 4 * n * c * hi * wi * 65)):
        check(L.mcdseg_up8_bwd_weight(_p(dy), _p(x), _p(dw), n, c, hi, wi, _p(ws), ctypes.c_size_t(ws.numel() * 4), _stream()),
              "up8_bwd_weight")
    return dw


def _up8_bwd(dy, w, x, want_dx, want_dw):
    """(dx, dw) of the up-sampler from one staged read of ``dy`` (csrc/up8.hip: up8_bwd_band_kernel); either may be skipped"""
    if not (want_dx or want_dw):
        return None, None
    L = lib()
    n, c, hi, wi = x.shape
    dx = torch.empty((n, c, hi, wi), dtype=torch.float32, device=dy.device) if want_dx else None
    dw = torch.empty((c, 1, 16, 16), dtype=torch.float32, device=dy.device) if want_dw else None
    ws = _ws(L.mcdseg_up8_bwd_workspace_bytes(n, c, hi, wi), dy.device) if want_dw else None
    name = "up8_bwd_band_kernel<%s, %s>" % ("true" if want_dx else "false", "true" if want_dw else "false")
    with _timed(name, (0, 4 * n * c * hi * wi * (64 + int(want_dx) + int(want_dw)))):
        check(L.mcdseg_up8_bwd(_p(dy), _p(w), _p(x), _p(dx), _p(dw), n, c, hi, wi, _p(ws),
                               ctypes.c_size_t(ws.numel() * 4 if ws is not None else 0), _stream()), "up8_bwd")
    return dx, dw


def _check_up(x, w):
    if x.dim() != 4 or tuple(w.shape) != (x.shape[1], 1, 16, 16):
        raise ValueError("mcdseg: up8 expects x [N,C,H,W] and w [C,1,16,16], got %s / %s" % (tuple(x.shape), tuple(w.shape)))


class _Up8(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w):
        x, w = _req(x, "up8 input"), _req(w, "up8 weight")
        _check_up(x, w)
        n, c, hi, wi = x.shape
        y = torch.empty((n, c, 8 * hi, 8 * wi), dtype=torch.float32, device=x.device)
        with _timed("up8_fwd", (0, 4 * n * c * hi * wi * 65)):
            check(lib().mcdseg_up8_fwd(_p(x), _p(w), None, None, _p(y), n, c, hi, wi, _stream()), "up8_fwd")
        ctx.save_for_backward(x, w)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = _req(dy, "grad_output")
        n, c, hi, wi = x.shape
        return _up8_bwd(dy, w, x, ctx.needs_input_grad[0], ctx.needs_input_grad[1])


class _Up8Dual(torch.autograd.Function):
    """up(x1, w1) + up(x2, w2) in one pass over the full-resolution output."""

    @staticmethod
    def forward(ctx, x1, w1, x2, w2):
        x1, w1, x2, w2 = _req(x1, "up8 input"), _req(w1, "up8 weight"), _req(x2, "up8 input"), _req(w2, "up8 weight")
        _check_up(x1, w1), _check_up(x2, w2)
        if x1.shape != x2.shape:
            raise ValueError("mcdseg: up8_dual inputs differ in shape")
        n, c, hi, wi = x1.shape
        y = torch.empty((n, c, 8 * hi, 8 * wi), dtype=torch.float32, device=x1.device)
        with _timed("up8_fwd", (0, 4 * n * c * hi * wi * 66)):
            check(lib().mcdseg_up8_fwd(_p(x1), _p(w1), _p(x2), _p(w2), _p(y), n, c, hi, wi, _stream()), "up8_fwd")
        ctx.save_for_backward(x1, w1, x2, w2)
        return y

    @staticmethod
    def backward(ctx, dy):
        x1, w1, x2, w2 = ctx.saved_tensors
        dy = _req(dy, "grad_output")
        n, c, hi, wi = x1.shape
        need = ctx.needs_input_grad
        return _up8_bwd(dy, w1, x1, need[0], need[1]) + _up8_bwd(dy, w2, x2, need[2], need[3])


def up8(x, w):
    return _Up8.apply(x, w)


def up8_dual(x1, w1, x2, w2):
    return _Up8Dual.apply(x1, w1, x2, w2)


# ------------------------------------------------------------------------------------------------ multitask decoder
class _Bilinear8(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = _req(x, "bilinear8 input")
        n, c, hi, wi = x.shape
        y = torch.empty((n, c, 8 * hi, 8 * wi), dtype=torch.float32, device=x.device)
        check(lib().mcdseg_bilinear8_fwd(_p(x), _p(y), n, c, hi, wi, _stream()), "bilinear8_fwd")
        ctx.shape = (n, c, hi, wi)
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = _req(dy, "grad_output")
        n, c, hi, wi = ctx.shape
        dx = torch.empty((n, c, hi, wi), dtype=torch.float32, device=dy.device)
        check(lib().mcdseg_bilinear8_bwd(_p(dy), _p(dx), n, c, hi, wi, _stream()), "bilinear8_bwd")
        return dx


def bilinear8(x):
    """nn.Upsample(scale_factor=8, mode='bilinear') with align_corners=False"""
    return _Bilinear8.apply(x)


class _MSE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target):
        L = lib()
        pred, target = _req(pred, "mse input"), _req(target, "mse target")
        if pred.shape != target.shape:
            raise ValueError("mcdseg: mse_loss shapes differ: %s vs %s" % (tuple(pred.shape), tuple(target.shape)))
        n = pred.numel()
        grad = torch.empty_like(pred) if ctx.needs_input_grad[0] else None
        loss = torch.empty(1, dtype=torch.float32, device=pred.device)
        ws = _ws(L.mcdseg_mse_workspace_bytes(n), pred.device)
        check(L.mcdseg_mse(_p(pred), _p(target), _p(grad), _p(loss), n, _p(ws), ctypes.c_size_t(ws.numel() * 4), _stream()), "mse")
        ctx.g = grad
        return loss.reshape(())

    @staticmethod
    def backward(ctx, grad_out):
        g, ctx.g = ctx.g, None
        return _scale_(g, _req(grad_out.reshape(1), "grad_output")), None


def mse_loss(pred, target):
    return _MSE.apply(pred, target)


# ------------------------------------------------------------------------------------------------ losses
def label_weight_sum(labels, class_weight, n_class, ignore_index=-100):
    """device scalar sum_i w[labels_i]"""
    L = lib()
    labels = _req(labels, "labels", torch.int64)
    class_weight = _req(class_weight, "class weights")
    out = torch.empty(1, dtype=torch.float32, device=labels.device)
    ws = _ws(L.mcdseg_label_weight_sum_workspace_bytes(labels.numel()), labels.device)
    check(L.mcdseg_label_weight_sum(_p(labels), _p(class_weight), int(ignore_index), int(n_class), labels.numel(), _p(out), _p(ws),
                                    ctypes.c_size_t(ws.numel() * 4), _stream()), "label_weight_sum")
    return out


def ce_normaliser(labels, class_weight, n_class, ignore_index=-100):
    """CE normaliser for this rank: None (kernel computes the local sum) for single-process runs; under data
    parallelism the all-reduced sum divided by world size, so that the optimizer's 1/world gradient average
    reproduces the reference's one global weighted mean."""
    from . import dist as mdist
    if not mdist.is_distributed():
        return None
    w = label_weight_sum(labels, class_weight, n_class, ignore_index)
    mdist.all_reduce_sum_(w)
    return w / mdist.world_size()


def mcd_losses(z1, z2, labels, class_weight, ignore_index=-100, ce_coef=0.0, diff_coef=0.0, want_grad=True, wsum=None):
    """One fused pass.  Returns (losses[4] = CE1, CE2, Diff, sum w[y]; g1; g2) where
    g_k = ce_coef * dCE_k/dz_k + diff_coef * dDiff/dz_k (None when not requested)."""
    L = lib()
    z1 = _req(z1, "logits")
    z2 = _req(z2, "logits")
    if z1.dim() != 4 or (z2 is not None and z2.shape != z1.shape):
        raise ValueError("mcdseg: logits must be [N,C,H,W] and of equal shape")
    n, c, h, w = z1.shape
    if labels is not None:
        labels = _req(labels, "labels", torch.int64)
        if tuple(labels.shape) != (n, h, w):
            raise ValueError("mcdseg: labels must be [N,H,W] = %s, got %s" % ((n, h, w), tuple(labels.shape)))
    class_weight = _req(class_weight, "class weights")
    if class_weight is not None and class_weight.numel() != c:
        raise ValueError("mcdseg: class weight has %d entries for %d classes" % (class_weight.numel(), c))
    if labels is not None and ce_coef != 0.0 and wsum is None:
        wsum = ce_normaliser(labels, class_weight, c, ignore_index)
    wsum = _req(wsum, "CE normaliser")
    losses = torch.empty(4, dtype=torch.float32, device=z1.device)
    g1 = torch.empty_like(z1) if want_grad else None
    g2 = torch.empty_like(z2) if (want_grad and z2 is not None) else None
    ws = _ws(L.mcdseg_loss_workspace_bytes(n, h * w), z1.device)
    nz = (1 if z2 is None else 2) * n * c * h * w
    byts = 4 * nz * (2 if want_grad else 1) + (8 * n * h * w if labels is not None else 0)
    with _timed("softmax_ce_l1_kernel<48, %s>" % ("true" if z2 is not None else "false") if c > 24 else "softmax_ce_l1_kernel", (0, byts)):
        check(L.mcdseg_softmax_ce_l1(_p(z1), _p(z2), _p(labels), _p(class_weight), int(ignore_index), float(ce_coef),
                                     float(diff_coef), _p(wsum), _p(g1), _p(g2), _p(losses), n, c, h * w, _p(ws),
                                     ctypes.c_size_t(ws.numel() * 4), _stream()), "softmax_ce_l1")
    return losses, g1, g2


def up8_loss_kernel_name(n, c, hi, wi, two, labelled):
    """The kernel ``mcdseg_up8_softmax_ce_l1`` launches for this problem, as rocprofv3 prints it -- from the library's own dispatch
    (``mcdseg_up8_loss_variant``: the LDS-DMA kernel unless the option UP8_LOSS_DMA is 0 or a tensor outgrows a 32-bit buffer resource;
    the benchmark's 41 classes have an instantiation of their own)."""
    two = "true" if two else "false"
    v = lib().mcdseg_up8_loss_variant(int(n), int(c), int(hi), int(wi), int(bool(labelled)))
    if v > 0:
        return "up8_softmax_ce_l1_dma_kernel<%d, %s, %s>" % (v, two, "true" if c == v else "false")
    return "up8_softmax_ce_l1_kernel<%d, %s>" % (-v, two)


def up8_mcd_losses(s1, w1, s2, w2, labels, class_weight, ignore_index=-100, ce_coef=0.0, diff_coef=0.0, want_grad=True, wsum=None):
    """``mcd_losses(up8(s1, w1), up8(s2, w2), ...)`` without the full-resolution logits: the kernel forms each pixel's logits
    from the score maps [N,C,Hi,Wi] on the fly.  Returns (losses[4], g1, g2) with g_k [N,C,8Hi,8Wi] = the gradient w.r.t.
    the (never stored) logits of head k -- what ``_up8_bwd_input`` / ``_up8_bwd_weight`` consume."""
    L = lib()
    s1, w1, s2, w2 = _req(s1, "scores"), _req(w1, "up8 weight"), _req(s2, "scores"), _req(w2, "up8 weight")
    _check_up(s1, w1)
    if s2 is not None:
        _check_up(s2, w2)
        if s2.shape != s1.shape:
            raise ValueError("mcdseg: the two score maps differ in shape")
    n, c, hi, wi = s1.shape
    h, w = 8 * hi, 8 * wi
    if labels is not None:
        labels = _req(labels, "labels", torch.int64)
        if tuple(labels.shape) != (n, h, w):
            raise ValueError("mcdseg: labels must be [N,H,W] = %s, got %s" % ((n, h, w), tuple(labels.shape)))
    class_weight = _req(class_weight, "class weights")
    if class_weight is not None and class_weight.numel() != c:
        raise ValueError("mcdseg: class weight has %d entries for %d classes" % (class_weight.numel(), c))
    if labels is not None and ce_coef != 0.0 and wsum is None:
        wsum = ce_normaliser(labels, class_weight, c, ignore_index)
    wsum = _req(wsum, "CE normaliser")
    losses = torch.empty(4, dtype=torch.float32, device=s1.device)
    g1 = torch.empty((n, c, h, w), dtype=torch.float32, device=s1.device) if want_grad else None
    g2 = torch.empty((n, c, h, w), dtype=torch.float32, device=s1.device) if (want_grad and s2 is not None) else None
    ws = _ws(L.mcdseg_up8_loss_workspace_bytes(n, hi, wi), s1.device)
    heads = 1 if s2 is None else 2
    byts = 4 * heads * n * c * h * w * (1 if want_grad else 0) + 4 * heads * n * c * hi * wi + (8 * n * h * w if labels is not None else 0)
    with _timed(up8_loss_kernel_name(n, c, hi, wi, s2 is not None, labels is not None), (0, byts)):
        check(L.mcdseg_up8_softmax_ce_l1(_p(s1), _p(w1), _p(s2), _p(w2), _p(labels), _p(class_weight), int(ignore_index),
                                         float(ce_coef), float(diff_coef), _p(wsum), _p(g1), _p(g2), _p(losses), n, c, hi, wi,
                                         _p(ws), ctypes.c_size_t(ws.numel() * 4), _stream()), "up8_softmax_ce_l1")
    return losses, g1, g2


def up8_backward(g, s, w, want_input, want_weight):
    """(d/ds, d/dw) of the up-sampler for the logit gradient ``g`` (either may be skipped)."""
    return _up8_bwd(g, w, s, want_input, want_weight)


def predict_labels(z1, z2=None, n_used=None):
    """Inference tail of adapt_tester.py:101-124: (uint8 label map [N,H,W] = argmax over the first ``n_used`` classes
    of z1 or (z1+z2)/2, mean entropy scalar as util.calc_entropy defines it)."""
    L = lib()
    z1, z2 = _req(z1, "logits"), _req(z2, "logits")
    n, c, h, w = z1.shape
    n_used = c if n_used is None else int(n_used)
    labels = torch.empty((n, h, w), dtype=torch.uint8, device=z1.device)
    ent = torch.empty(1, dtype=torch.float32, device=z1.device)
    ws = _ws(L.mcdseg_predict_workspace_bytes(n, h * w), z1.device)
    check(L.mcdseg_predict_labels(_p(z1), _p(z2), _p(labels), _p(ent), n, c, n_used, h * w, _p(ws), ctypes.c_size_t(ws.numel() * 4),
                                  _stream()), "predict_labels")
    return labels, ent.reshape(())


# ------------------------------------------------------------------------------------------------ MFNet gate fusion
class _GateMix(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x1, x2, g):
        x1, x2, g = _req(x1, "x1"), _req(x2, "x2"), _req(g, "gate logits")
        if not (x1.shape == x2.shape == g.shape) or x1.numel() % 4:
            raise ValueError("mcdseg: gate_mix needs three tensors of one shape with a multiple of 4 elements")
        out = torch.empty_like(x1)
        check(lib().mcdseg_gate_mix_fwd(_p(x1), _p(x2), _p(g), _p(out), x1.numel(), _stream()), "gate_mix_fwd")
        ctx.save_for_backward(x1, x2, g)
        return out

    @staticmethod
    def backward(ctx, dy):
        x1, x2, g = ctx.saved_tensors
        dy = _req(dy, "grad_output")
        d1, d2, dg = torch.empty_like(x1), torch.empty_like(x1), torch.empty_like(x1)
        check(lib().mcdseg_gate_mix_bwd(_p(dy), _p(x1), _p(x2), _p(g), _p(d1), _p(d2), _p(dg), x1.numel(), _stream()), "gate_mix_bwd")
        return d1, d2, dg


def gate_mix(x1, x2, gate_logits):
    """x1*sigmoid(g) + x2*(1 - sigmoid(g)) (GateFusion.forward, models/fusion.py:19-22) in one pass, backward in one pass"""
    return _GateMix.apply(x1, x2, gate_logits)


class _SoftmaxCh(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = _req(x, "logits")
        n, c, h, w = x.shape
        y = torch.empty_like(x)
        check(lib().mcdseg_softmax_ch_fwd(_p(x), _p(y), n, c, h * w, _stream()), "softmax_ch_fwd")
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        dy = _req(dy, "grad_output")
        n, c, h, w = y.shape
        dx = torch.empty_like(y)
        check(lib().mcdseg_softmax_ch_bwd(_p(dy), _p(y), _p(dx), n, c, h * w, _stream()), "softmax_ch_bwd")
        return dx


def softmax_channels(x):
    """F.softmax over dim 1 of an NCHW tensor (the ScoreGateFusion pre-step, models/fusion.py:13-15)"""
    return _SoftmaxCh.apply(x)


class _ProbNLL(torch.autograd.Function):
    @staticmethod
    def forward(ctx, p, labels, weight, ignore_index, size_average):
        L = lib()
        p = _req(p, "probabilities")
        labels = _req(labels, "labels", torch.int64)
        weight = _req(weight, "class weights")
        n, c, h, w = p.shape
        if tuple(labels.shape) != (n, h, w):
            raise ValueError("mcdseg: labels %s do not match probabilities %s" % (tuple(labels.shape), tuple(p.shape)))
        loss = torch.empty(3, dtype=torch.float32, device=p.device)
        grad = torch.empty_like(p) if ctx.needs_input_grad[0] else None
        ws = torch.empty(L.mcdseg_prob_nll_workspace_bytes(n, h * w) // 8 + 1, dtype=torch.float64, device=p.device)
        check(L.mcdseg_prob_nll(_p(p), _p(labels), _p(weight), int(ignore_index), int(bool(size_average)), _p(grad), _p(loss), n, c,
                                h * w, _p(ws), ctypes.c_size_t(ws.numel() * 8), _stream()), "prob_nll")
        ctx.g = grad
        return loss[0].clone() if size_average else loss[1].clone()

    @staticmethod
    def backward(ctx, grad_out):
        g = ctx.g
        ctx.g = None
        return (g * grad_out if g is not None else None), None, None, None, None


def prob_cross_entropy2d(p, labels, weight=None, ignore_index=-100, size_average=True):
    """ProbCrossEntropyLoss2d (loss.py:16-30): NLLLoss2d(weight)(log(p), labels) with its gradient from the same call"""
    return _ProbNLL.apply(p, labels, weight, ignore_index, size_average)


def normalize_u8_(dst, src, mean, std, c_off=0):
    """ToTensor()+Normalize() of transform.py:302-315 on the device: ``src`` uint8 [N,H,W,Cs] (HWC, as PIL/numpy hand it
    over) is written as fp32 into channels [c_off, c_off+Cs) of ``dst`` [N,C,H,W].  mean/std: fp32 device tensors [Cs]."""
    src = _req(src, "uint8 image batch", torch.uint8)
    if src.dim() != 4:
        raise TypeError("mcdseg: normalize_u8_ takes a uint8 [N,H,W,C] tensor")
    if dst.dtype != torch.float32 or not dst.is_contiguous() or not dst.is_cuda:
        raise TypeError("mcdseg: normalize_u8_ writes into a contiguous fp32 GPU tensor")
    n, h, w, cs = src.shape
    if tuple(dst.shape[0:1] + dst.shape[2:]) != (n, h, w):
        raise ValueError("mcdseg: normalize_u8_ shape mismatch %s vs %s" % (tuple(src.shape), tuple(dst.shape)))
    mean, std = _req(mean.float(), "mean"), _req(std.float(), "std")
    check(lib().mcdseg_normalize_u8(_p(src), _p(dst), _p(mean), _p(std), n, h, w, cs, dst.shape[1], int(c_off), _stream()),
          "normalize_u8")
    return dst


def resize_u8(src, out_wh, nearest=False):
    """Scale(img_shape, Image.BILINEAR) (images, uint8 [N,H,W,C]) / Scale(img_shape, Image.NEAREST) (label maps, uint8 [N,H,W])
    of transform.py:303, 320 on the device -- Pillow's 8-bit resize arithmetic, bit for bit; ``out_wh`` = (W, H) as the reference
    passes it"""
    L = lib()
    src = _req(src, "uint8 batch", torch.uint8)
    ow, oh = int(out_wh[0]), int(out_wh[1])
    if nearest:
        if src.dim() != 3:
            raise TypeError("mcdseg: resize_u8(nearest) takes a uint8 [N,H,W] tensor")
        n, h, w = src.shape
        c = 1
    else:
        if src.dim() != 4:
            raise TypeError("mcdseg: resize_u8 takes a uint8 [N,H,W,C] tensor")
        n, h, w, c = src.shape
    if (ow, oh) == (w, h):
        return src
    ws = torch.empty(L.mcdseg_resize_workspace_bytes(n, h, w, c, oh, ow) // 4 + 2, dtype=torch.int32, device=src.device)
    if nearest:
        dst = torch.empty((n, oh, ow), dtype=torch.uint8, device=src.device)
        check(L.mcdseg_resize_nearest_u8(_p(src), _p(dst), n, h, w, oh, ow, _p(ws), ctypes.c_size_t(ws.numel() * 4), _stream()),
              "resize_nearest_u8")
    else:
        dst = torch.empty((n, oh, ow, c), dtype=torch.uint8, device=src.device)
        check(L.mcdseg_resize_bilinear_u8(_p(src), _p(dst), n, h, w, c, oh, ow, _p(ws), ctypes.c_size_t(ws.numel() * 4), _stream()),
              "resize_bilinear_u8")
    return dst


def relabel_u8(src, olabel, nlabel):
    """ToLabel()+ReLabel(olabel, nlabel) (transform.py:21-48): uint8 label maps -> int64 with the background id remapped"""
    src = _req(src, "uint8 label batch", torch.uint8)
    out = torch.empty(src.shape, dtype=torch.int64, device=src.device)
    check(lib().mcdseg_relabel_u8(_p(src), _p(out), src.numel(), int(olabel), int(nlabel), _stream()), "relabel_u8")
    return out


def confusion_hist_(hist, gt, pred):
    """hist[n*gt + pred] += 1 for 0 <= gt < n (eval.py:21-23 fast_hist); ``hist`` int64 [n,n] on the GPU, accumulated."""
    gt, pred = _req(gt.long(), "ground truth", torch.int64), _req(pred.long(), "prediction", torch.int64)
    if gt.numel() != pred.numel():
        raise ValueError("mcdseg: confusion_hist_ needs as many predictions as labels")
    if hist.dtype != torch.int64 or not hist.is_cuda or not hist.is_contiguous() or hist.dim() != 2 or hist.shape[0] != hist.shape[1]:
        raise TypeError("mcdseg: confusion_hist_ accumulates into a contiguous int64 [n,n] GPU tensor")
    if gt.numel():
        check(lib().mcdseg_confusion_hist(_p(gt), _p(pred), gt.numel(), hist.shape[0], _p(hist), _stream()), "confusion_hist")
    return hist


def _scale_(g, s):
    check(lib().mcdseg_scale_by_device_scalar(_p(g), _p(s), g.numel(), _stream()), "scale_by_device_scalar")
    return g


class _CrossEntropy2d(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, labels, class_weight, ignore_index, size_average):
        losses, g, _ = mcd_losses(z, None, labels, class_weight, ignore_index, ce_coef=1.0, want_grad=ctx.needs_input_grad[0])
        ctx.g = g
        ctx.size_average = size_average
        ctx.wsum = losses[3:4]
        return losses[0].clone() if size_average else losses[0] * losses[3]

    @staticmethod
    def backward(ctx, grad_out):
        g, ctx.g = ctx.g, None
        s = _req(grad_out.reshape(1), "grad_output")
        if not ctx.size_average:
            s = s * ctx.wsum
        return _scale_(g, s), None, None, None, None


class _Diff2d(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z1, z2):
        need = ctx.needs_input_grad[0] or ctx.needs_input_grad[1]
        losses, g1, g2 = mcd_losses(z1, z2, None, None, diff_coef=1.0, want_grad=need)
        ctx.g = (g1, g2)
        return losses[2].clone()

    @staticmethod
    def backward(ctx, grad_out):
        (g1, g2), ctx.g = ctx.g, None
        s = _req(grad_out.reshape(1), "grad_output")
        return _scale_(g1, s), _scale_(g2, s)


def cross_entropy2d(z, labels, class_weight=None, ignore_index=-100, size_average=True):
    return _CrossEntropy2d.apply(z, labels, class_weight, ignore_index, size_average)


def diff2d(z1, z2):
    return _Diff2d.apply(z1, z2)


# ------------------------------------------------------------------------------------------------ optimizer kernel
def sgd_momentum_flat_(p, g, v, lr, momentum, weight_decay, grad_scale=1.0, params=None):
    """``params``: the parameter objects living in ``p`` (their packed images go stale); None = every packed image does"""
    for t, name in ((p, "params"), (g, "grads"), (v, "momentum")):
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
            raise RuntimeError("mcdseg: flat SGD needs contiguous fp32 GPU buffers (%s)" % name)
    with _timed("sgd_momentum_kernel", (0, 20 * p.numel())):
        check(lib().mcdseg_sgd_momentum_flat(_p(p), _p(g), _p(v), p.numel(), float(lr), float(momentum), float(weight_decay),
                                             float(grad_scale), _stream()), "sgd_momentum_flat")
    bump_weight_epoch(params)
