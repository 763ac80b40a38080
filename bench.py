#!/usr/bin/env python3
"""Benchmark of the MCD hot path on MI355X (BASELINE.json metric).

One *step* = one full three-step MCD update (A: G+F on source, B: F on source CE - target discrepancy,
C: num_k = 4 generator updates on the target discrepancy; adapt_trainer.py:155-220 of the reference) over
one synthetic batch of N (source, target) RGB+HHA pairs per GPU.  The timed schedule runs 6 generator forward and 5 generator
backward passes per step: the reference's literal 7 + 7, minus the two step-B backward passes whose gradients it zeroes unused
(SURVEY.md section 3.1), minus step B's target forward, which IS step C's first one (solvers/solver.py; bit-identical results).
The literal 7-forward schedule is measured in the same run and printed next to it (``value_literal_schedule``).
``value`` = pairs/s over the whole job, inputs resident in HBM; every step gets a batch it has not seen (a pool of pre-staged
batches, per-tensor caches dropped), so the per-batch work -- the input's bound and its padded companion -- is inside the timed region.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python bench.py --gpus 8 --steps 5 --warmup 2          # starts the 8 ranks itself (before touching a GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  Every kernel family of the step is bracketed by HIP events on the launch stream inside
the timed region; ``roofline`` prices the kernel with the largest summed time (``dominant``): its algorithmic FLOPs or
bytes per launch / its average launch duration, against the MFMA peak of the arithmetic it executes or the HBM peak.
``roofline_forward`` / ``roofline_wgrad`` are the same for the forward / weight-gradient convolution carrying the most FLOPs.  The pass is matrix-rate bound,
not HBM bound: ``step_accounting`` says so next to the raw HBM fraction the metric asks for.
``cpu_baseline`` times the CPU oracle (plain PyTorch restatement of the reference) on a bounded sample of the same
workload on the host cores (rank 0, N = 1 only).  ``other_configs`` (N = 1 only, after the timed region): full A+B+C steps of BASELINE
configs 3-5 at their stated per-GPU sizes -- parity-tested configurations, timed here so that the driver sees them; they never enter
``value``.  ``collective_ms_per_step`` (N > 1): the time the compute stream stands still for the gradient exchange.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "multichannel-semseg-with-uda_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)
os.environ.setdefault("MCDSEG_PRETRAINED", "0")

PEAK_FP32_TFLOPS = 157.3   # MI355X fp32 vector = fp32-input MFMA (MI355X_MICROARCH.md, chip-level table)
PEAK_16BIT_TFLOPS = 2500.0  # dense bf16 / fp16 MFMA (MI355X_MICROARCH.md, chip-level table; the 5 PF figure is 2:1 sparse)
PEAK_HBM_GBS = 8000.0      # HBM3E spec
# algorithmic work per image-pass (one image through G+F1+F2 forward+backward incl. loss), drn_d_38 6x480x640
# (SURVEY.md section 8d / BASELINE.md section 3): 781.5 GFLOP, 1.94 GB
GF_FWD_PER_IMG = 260.5
GB_FWD_PER_IMG, GB_BWD_PER_IMG = 0.71, 1.23

# executed matrix FLOPs per algorithmic FLOP, the MFMA peak they are priced against, and what the arithmetic is
MATH = {
    "f32": (1.0, PEAK_FP32_TFLOPS, "v_mfma_f32_32x32x2_f32"),
    "bf16x6": (6.0, PEAK_16BIT_TFLOPS, "fp32 operands as 3-way bf16 split, 6 cross terms on v_mfma_f32_32x32x16_bf16"),
    "f16x3": (3.0, PEAK_16BIT_TFLOPS, "fp32 operands as scaled 2-way fp16 split, 3 cross terms on v_mfma_f32_32x32x16_f16"),
    "f16x1": (1.0, PEAK_16BIT_TFLOPS, "operands rounded to one scaled fp16 piece, 1 term on v_mfma_f32_32x32x16_f16 (reduced precision)"),
}
DTYPE_LABEL = {"f32": "f32",
               "bf16x6": "f32 (operands split 3-way into bf16, 6 cross terms on the bf16 MFMA pipe, fp32 accumulate)",
               "f16x3": "f32 (operands split 2-way into scaled fp16, 3 cross terms on the fp16 MFMA pipe, fp32 accumulate)",
               "f16x1": "f16 (operands rounded to one scaled fp16 piece, fp32 accumulate, fp32 BatchNorm / loss / optimizer arithmetic; with compact "
                        "storage every trunk tensor is one 16-bit value per element) -- REDUCED precision, not the judged configuration",
               "mixed": "f32 (bf16x6 split for forward/dgrad, f32 MFMA for wgrad)"}
TIMED_FAMILIES = {
    "all": ["conv_", "bn_", "up8_", "softmax_ce_l1", "sgd_", "loss_", "wgrad_", "split_", "pack_"],
    "conv": ["conv_gemm", "conv_stem", "conv_wgrad", "softmax_ce_l1"],
    "none": [],
}


class LaunchTimer:
    """Collects HIP-event pairs recorded around kernel launches (mcdseg.ops.LAUNCH_TIMER) on the launch stream."""

    def __init__(self, names):
        self.names = tuple(names)
        self.records = []
        self.enabled = False

    def wants(self, name):
        return self.enabled and name.startswith(self.names)

    def add(self, name, work, t0, t1):
        self.records.append((name, work, t0, t1))

    def summary(self):
        out = {}
        for name, (flops, byts), t0, t1 in self.records:
            ms = t0.elapsed_time(t1)
            s = out.setdefault(name, dict(launches=0, ms=0.0, flops=0.0, bytes=0.0))
            s["launches"] += 1
            s["ms"] += ms
            s["flops"] += flops
            s["bytes"] += byts
        return out


def synthetic_batch(n, ch, h, w, n_class, seed):
    """N(0,1) images (stand-in for ImageNet-normalised RGB+HHA), labels U{0..n_class-1}."""
    import torch
    g = torch.Generator().manual_seed(seed)
    src = torch.randn(n, ch, h, w, generator=g)
    lbl = torch.randint(0, n_class, (n, h, w), generator=g, dtype=torch.int64)
    tgt = torch.randn(n, ch, h, w, generator=g)
    return src, lbl, tgt


def build_hip(args, dev):
    import torch
    from loss import CrossEntropyLoss2d, get_prob_distance_criterion
    from models.model_util import get_models, get_optimizer
    from solvers.solver import MCDSolver
    torch.manual_seed(0)
    g, f1, f2 = get_models(args.net, args.input_ch, args.n_class, method="MCD")
    for m in (g, f1, f2):
        m.to(dev).train()
    og = get_optimizer(g.parameters(), "sgd", 1e-3, 0.9, 2e-5)
    of = get_optimizer(list(f1.parameters()) + list(f2.parameters()), "sgd", 1e-3, 0.9, 2e-5)
    w = torch.ones(args.n_class)
    w[args.n_class - 1] = 0
    solver = MCDSolver(g, f1, f2, og, of, CrossEntropyLoss2d(w.to(dev)), get_prob_distance_criterion("diff"), num_k=4)
    if args.no_forward_reuse:
        solver.reuse_tgt = False
    return solver, (g, f1, f2)


ARITHMETIC_NOTE = {
    "cfg5": "f16x3, the fp32-grade arithmetic of the judged line (22-bit operands, fp32 accumulation).  BASELINE config 5 says 'bf16': the "
            "reduced-precision counterpart is the cfg5_f16 entry",
    "cfg5_f16": "f16x1 with 2-byte activation storage = BASELINE config 5's 'bf16': every trunk tensor is ONE 16-bit value per element -- "
                "activations, pre-BatchNorm z and dz as per-tensor-scaled fp16 (11 significant bits where bf16 keeps 8; the scale comes from a "
                "bound known before the tensor is written: Samuelson's inequality for BatchNorm outputs, K max|x| max|w| for z), the gradients "
                "between the groups as bf16 (they span more binades than one scale covers and have no a-priori bound); the convolutions multiply "
                "the one fp16 piece (gfx950 runs fp16 and bf16 MFMA at the same rate) and accumulate in fp32, BatchNorm statistics come from the "
                "fp32 accumulators, BatchNorm / loss / optimizer arithmetic is fp32",
    "cfg2_bf16x6": "strict fp32: three bf16 pieces per operand (24 bits), six cross terms, no per-tensor scale",
    "cfg2_f32": "strict fp32: v_mfma_f32_32x32x2_f32, the exact k-ordered fp32 FMA chain",
}


def other_config(tag, dev, steps, n_class=41, want_roofline=False):
    """One of BASELINE's other configurations on synthetic device-resident batches: 1 warm-up + ``steps`` timed full A+B+C steps
    (adapt_mfnet_trainer.py:174-244, adapt_multitask_trainer.py:166-239, adapt_trainer.py:155-220 on drn_d_105); ``cfg2_bf16x6`` /
    ``cfg2_f32``: the judged configuration in the two strict-fp32 arithmetics.  ``want_roofline``: one more step whose launches carry
    HIP-event pairs (every kernel alone on the main stream), reduced to the roofline entry of the kernel with the largest summed time."""
    import torch
    from loss import CrossEntropyLoss2d, Diff2d, get_prob_distance_criterion
    from mcdseg import ops
    from models.model_util import get_models, get_multitask_models, get_optimizer
    from solvers.solver import MCDSolver, MFNetMCDSolver, MultiTaskMCDSolver
    n, h, w, net, math, storage = {"cfg3": (16, 480, 640, "drn_d_38", None, "fp32"), "cfg4": (8, 480, 640, "drn_d_38", None, "fp32"),
                                   "cfg5": (32, 720, 1280, "drn_d_105", None, "compact"),
                                   "cfg5_f16": (32, 720, 1280, "drn_d_105", "f16x1", "compact"),
                                   "cfg2_bf16x6": (16, 480, 640, "drn_d_38", "bf16x6", "fp32"),
                                   "cfg2_f32": (16, 480, 640, "drn_d_38", "f32", "fp32")}[tag]
    prev = (ops.CONV_MATH, ops.ACT_STORAGE)
    ops.CONV_MATH, ops.ACT_STORAGE = math or prev[0], storage
    try:
        def opt(params):
            return get_optimizer(params, "sgd", 1e-3, 0.9, 2e-5)
        cw = torch.ones(n_class)
        cw[n_class - 1] = 0
        crit, crit_d = CrossEntropyLoss2d(cw.to(dev)), get_prob_distance_criterion("diff")
        torch.manual_seed(0)
        if tag == "cfg3":
            g3, g1, f1, f2 = get_models(net, 6, n_class, method="MFNet-ScoreAddFusion")
            for m in (g3, g1, f1, f2):
                m.to(dev).train()
            solver = MFNetMCDSolver(g3, g1, f1, f2, opt(list(g3.parameters()) + list(g1.parameters())),
                                    opt(list(f1.parameters()) + list(f2.parameters())), crit, crit_d, num_k=4)
            what = "adapt_mfnet_trainer MFNet-ScoreAddFusion, two drn_d_38 encoders"
        elif tag == "cfg4":
            enc, dec = get_multitask_models(net, 6, n_class, CrossEntropyLoss2d(cw), Diff2d())
            enc.to(dev).train(), dec.to(dev).train()
            solver = MultiTaskMCDSolver(enc, dec, opt(enc.parameters()), opt(dec.parameters()), num_k=4)
            what = "adapt_multitask_trainer seg + HHA-regression decoders, drn_d_38 RGB encoder"
        else:
            g, f1, f2 = get_models(net, 6, n_class, method="MCD")
            for m in (g, f1, f2):
                m.to(dev).train()
            solver = MCDSolver(g, f1, f2, opt(g.parameters()), opt(list(f1.parameters()) + list(f2.parameters())), crit, crit_d, num_k=4)
            what = (("adapt_trainer MCD drn_d_105, 2-byte activation storage (one scaled fp16 piece per activation / z / dz, bf16 gradients)"
                     if (tag == "cfg5_f16" and ops.HALF_STORAGE) else
                     "adapt_trainer MCD drn_d_105, activations kept as 2 x fp16 companions (MCDSEG_ACT_STORAGE=compact)") if tag.startswith("cfg5")
                    else "adapt_trainer MCD early-fusion drn_d_38 6-ch (the judged configuration in another arithmetic)")
        gen = torch.Generator(device=dev).manual_seed(1234)
        src = torch.randn(n, 6, h, w, generator=gen, device=dev)
        lbl = torch.randint(0, n_class, (n, h, w), generator=gen, device=dev, dtype=torch.int64)
        tgt = torch.randn(n, 6, h, w, generator=gen, device=dev)
        torch.cuda.reset_peak_memory_stats(dev)
        ops.WGRAD_STREAM_STATS.update(deferred=0, no_room=0)
        out = solver.step(src, lbl, tgt)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            out = solver.step(src, lbl, tgt)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        res = {"workload": "%s, bs=%d/GPU synthetic 6x%dx%d, full A+B+C step (num_k=4)" % (what, n, h, w), "conv_math": ops.CONV_MATH,
               "steps": steps, "ms_per_step": round(1e3 * dt, 1), "pairs_per_s": round(n / dt, 3),
               "peak_gb": round(torch.cuda.max_memory_allocated(dev) / 2 ** 30, 1),
               "peak_reserved_gb": round(torch.cuda.max_memory_reserved(dev) / 2 ** 30, 1), "c_loss": float(out[0]), "d_loss": float(out[1]),
               "wgrad_stream": dict(ops.WGRAD_STREAM_STATS)}
        if tag in ARITHMETIC_NOTE:
            res["arithmetic"] = ARITHMETIC_NOTE[tag]
        if want_roofline:
            timer = LaunchTimer(TIMED_FAMILIES["all"])
            timer.enabled, ops.LAUNCH_TIMER = True, timer
            try:
                solver.step(src, lbl, tgt)
                torch.cuda.synchronize()
            finally:
                ops.LAUNCH_TIMER = None
            kern = timer.summary()
            if kern:
                total = sum(k["ms"] for k in kern.values()) or 1.0
                for k in kern.values():
                    k["share"] = round(k["ms"] / total, 4)
                dom = max(kern, key=lambda n: kern[n]["ms"])
                res["roofline"] = kernel_roofline(dom, kern[dom], ops.CONV_MATH, None)
                res["timed_kernel_ms_per_step"] = round(total, 1)
                res["kernel_ms"] = {n: round(v["ms"], 1) for n, v in sorted(kern.items(), key=lambda kv: -kv[1]["ms"])[:16]}
        del solver, src, lbl, tgt
        return res
    finally:
        ops.CONV_MATH, ops.ACT_STORAGE = prev
        torch.cuda.empty_cache()


def is_forward_conv(name):
    """conv_gemm*<..., DGRAD, PRESPLIT> / conv_gemm_split_pp_kernel<P, DGRAD>: the instantiations with DGRAD = false"""
    if not name.startswith("conv_gemm") or "<" not in name:
        return False
    args = [a.strip() for a in name[name.index("<") + 1:name.rindex(">")].split(",")]
    if name.startswith("conv_gemm_split_pp_kernel"):
        return args[1] == "false"
    return (args[-2] if name.startswith("conv_gemm_split_kernel") else args[-1]) == "false"


def host_cpu():
    """(model name, physical cores, logical cpus) of the host, from /proc/cpuinfo"""
    model, phys, logical = "unknown", set(), 0
    try:
        pid = None
        for line in open("/proc/cpuinfo"):
            k, _, v = line.partition(":")
            k, v = k.strip(), v.strip()
            if k == "model name":
                model = v
            elif k == "processor":
                logical += 1
            elif k == "physical id":
                pid = v
            elif k == "core id":
                phys.add((pid, v))
    except OSError:
        pass
    return model, (len(phys) or (os.cpu_count() or 1)), (logical or (os.cpu_count() or 1))


def cpu_baseline(args):
    """The CPU oracle on the host cores: full MCD steps on ``--cpu_pairs`` (src,tgt) pairs of the bench geometry --
    one warm-up step, then up to ``--cpu_steps`` timed steps (median), bounded by ``--cpu_budget_s`` seconds of wall
    clock (at least one timed step) so the default run stays within minutes."""
    import torch
    from oracle import ref_loss, ref_mcd, ref_models
    model, phys, logical = host_cpu()
    threads = max(1, min(phys, logical))
    prev = torch.get_num_threads()
    torch.set_num_threads(threads)
    try:
        torch.manual_seed(0)
        g, f1, f2 = ref_models.get_models(args.net, args.input_ch, args.n_class)
        for m in (g, f1, f2):
            m.train()
        og = ref_models.get_optimizer(g.parameters(), "sgd", 1e-3, 0.9, 2e-5)
        of = ref_models.get_optimizer(list(f1.parameters()) + list(f2.parameters()), "sgd", 1e-3, 0.9, 2e-5)
        w = ref_loss.class_weights(args.n_class)
        n = args.cpu_pairs
        src, lbl, tgt = synthetic_batch(n, args.input_ch, args.height, args.width, args.n_class, 1234)
        crit, critd = ref_loss.CrossEntropyLoss2d(w), ref_loss.Diff2d()
        t_begin = time.perf_counter()
        times = []
        for i in range(1 + args.cpu_steps):
            t0 = time.perf_counter()
            ref_mcd.mcd_step(g, f1, f2, og, of, crit, critd, src, lbl, tgt)
            dt = time.perf_counter() - t0
            if i > 0:
                times.append(dt)
            elapsed = time.perf_counter() - t_begin
            if i >= 1 and elapsed + dt > args.cpu_budget_s:
                break
    finally:
        torch.set_num_threads(prev)
    times.sort()
    med = times[len(times) // 2] if len(times) % 2 else 0.5 * (times[len(times) // 2 - 1] + times[len(times) // 2])
    return {"value": n / med, "unit": "img/s", "cores": threads, "kind": "port", "cpu_model": model, "physical_cores": phys,
            "logical_cpus": logical, "timed_steps": len(times), "step_seconds": [round(t, 2) for t in times],
            "sample": "CPU oracle, full MCD step (A+B+C, num_k=4; 7 fwd + 7 bwd passes) on %d pair(s) of %dx%dx%d: 1 warm-up + %d timed "
                      "step(s), median %.1f s/step, %d threads on %s (%d physical cores)"
                      % (n, args.input_ch, args.height, args.width, len(times), med, threads, model, phys)}


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def spawn_ranks(args, argv):
    """``python bench.py --gpus N`` without a launcher: start N ranks as CHILD processes through torch.distributed.run --
    this process has not touched a GPU (and never does), it relays the children's output and exit code."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    return subprocess.run(cmd, env=env).returncode


def dry_run(args):
    """Launcher / timing protocol without a GPU (MCDSEG_BENCH_DRY=1; tests/test_bench_launcher.py): gloo ranks, a stand-in step,
    the same barriers, MAX over ranks and the one JSON line from rank 0."""
    import torch
    from mcdseg import dist as mdist
    rank, world, _ = mdist.init_from_env(backend="gloo")
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    mdist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.01 * (1 + rank))
    mdist.barrier()
    el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    if world > 1:
        torch.distributed.all_reduce(el, op=torch.distributed.ReduceOp.MAX)
    if rank == 0:
        print(json.dumps({"metric": "dry run (no GPU work)", "value": args.batch * world * args.steps / float(el), "unit": "img/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * float(el) / args.steps,
                          "scaling": "weak", "data": "none"}), file=sys.__stdout__, flush=True)
    mdist.barrier()
    if world > 1:
        torch.distributed.destroy_process_group()


def kernel_roofline(name, k, math, traffic):
    """roofline entry of one timed kernel family: MFMA-bound for the convolutions, HBM-bound for the streaming kernels"""
    if name.startswith("conv_") and k["flops"] > 0:
        if "SplitF16x1" in name:
            mult, peak, what = MATH["f16x1"]
        elif "SplitF16x3" in name:
            mult, peak, what = MATH["f16x3"]
        elif "SplitBf16x6" in name or "x6" in name:
            mult, peak, what = MATH["bf16x6"]
        else:
            mult, peak, what = MATH["f32"]
        tfl = k["flops"] / (k["ms"] * 1e-3) / 1e12
        return {"bound": "mfma", "kernel": "%s (%s)" % (name, what), "achieved": round(mult * tfl, 2), "peak": peak, "unit": "TFLOP/s",
                "frac": round(mult * tfl / peak, 4), "frac_algorithmic": round(tfl / peak, 4),
                "frac_of_fp32_peak": round(tfl / PEAK_FP32_TFLOPS, 4), "traffic": traffic, "alg_tflops_fp32_equivalent": round(tfl, 2),
                "frac_note": "frac prices the EXECUTED 16-bit flops (executed_flops_per_alg_flop per algorithmic flop) against the dense MFMA "
                             "peak; frac_algorithmic prices the algorithmic (fp32-equivalent) flops against the same peak; frac_of_fp32_peak "
                             "against the %.1f TFLOP/s fp32 pipe" % PEAK_FP32_TFLOPS,
                "executed_flops_per_alg_flop": mult, "alg_bytes_per_launch": k["bytes"] / k["launches"], "launches": k["launches"],
                "avg_launch_ms": round(k["ms"] / k["launches"], 4), "alg_flops_per_launch": k["flops"] / k["launches"],
                "share_of_timed_kernel_ms": k.get("share")}
    gbs = k["bytes"] / (k["ms"] * 1e-3) / 1e9
    return {"bound": "hbm", "kernel": name, "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4),
            "traffic": traffic, "alg_bytes_per_launch": k["bytes"] / k["launches"], "launches": k["launches"],
            "avg_launch_ms": round(k["ms"] / k["launches"], 4), "share_of_timed_kernel_ms": k.get("share")}


def pmc_traffic(name, kern=None, timer_steps=1):
    """(HBM-side bytes per launch of kernel ``name``, where the figure comes from) out of the newest committed PMC table
    (profiles/<round>_pmc_traffic.json: FETCH_SIZE and WRITE_SIZE collected in separate rocprofv3 --pmc passes of this same bench,
    FETCH_SIZE doubled per the gfx950 note).  The table is a LOOKUP, so it is refused -- (None, why) -- unless it was made from a run
    of THIS set of kernels: every templated convolution kernel this run timed (``kern``: the launch timer's summary) must be in it,
    and ``name`` must have been launched as often per step there as here."""
    import glob
    from mcdseg import _lib
    tables = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")), reverse=True)
    if not tables:
        return None, "no profiles/*_pmc_traffic.json"
    # the table made from THIS build's kernel sources (tools/summarize_profiles.py records their fingerprint): a kernel re-tuned under
    # its old name must not be priced with the old build's counters
    here = _lib.source_fingerprint()
    fn, doc = None, None
    for cand in tables:
        try:
            d = json.load(open(cand))
        except (ValueError, OSError):
            continue
        if d.get("source_fingerprint") == here:
            fn, doc = cand, d
            break
    if fn is None:
        return None, "no table of this build: the kernel sources changed since %s was collected (fingerprint %s... here)" % (
            os.path.relpath(tables[0], ROOT), here[:12])
    rel = os.path.relpath(fn, ROOT)
    try:
        table = doc["kernels"]
    except KeyError as e:
        return None, "%s unreadable (%s)" % (rel, type(e).__name__)
    if kern is not None:
        missing = sorted(n for n in kern if n.startswith("conv_") and "<" in n and n not in table)
        if missing:
            return None, "%s is stale: it has no entry for %s" % (rel, ", ".join(missing))
        per_step = kern[name]["launches"] / float(max(1, timer_steps)) if name in kern else None
        if name in table and per_step is not None and table[name].get("launches_per_step") not in (None, per_step):
            return None, "%s is stale: %s ran %s times per step there, %s here" % (rel, name, table[name].get("launches_per_step"), per_step)
    if name not in table:
        return None, "%s has no entry for %s" % (rel, name)
    return table[name]["hbm_bytes_per_launch"], rel


def main():
    # the contract is ONE JSON line on stdout: whatever the model constructors print on the way (the reference's MFNet announces its
    # fusion type, models/fusion.py) goes to stderr
    sys.stdout = sys.stderr
    try:
        _main()
    finally:
        sys.stdout = sys.__stdout__


def _main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=16, help="(src,tgt) pairs per GPU")
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--net", default="drn_d_38")
    ap.add_argument("--input_ch", type=int, default=6)
    ap.add_argument("--n_class", type=int, default=41)
    ap.add_argument("--cpu_pairs", type=int, default=2)
    ap.add_argument("--cpu_steps", type=int, default=3, help="timed CPU-oracle steps after one warm-up (fewer if the budget runs out)")
    ap.add_argument("--cpu_budget_s", type=float, default=150.0)
    ap.add_argument("--no_cpu_baseline", action="store_true")
    ap.add_argument("--batch_pool", type=int, default=4, help="distinct pre-staged (src, lbl, tgt) batches visited round-robin")
    ap.add_argument("--literal_steps", type=int, default=4,
                    help="extra steps with the literal 7-forward schedule after the timed region, reported as 'literal_schedule' (0 = skip)")
    ap.add_argument("--no_forward_reuse", action="store_true",
                    help="literal schedule of adapt_trainer.py: step B's target forward and step C's first one run separately "
                         "(7 generator forwards per step instead of 6; same weights, statistics and losses bit for bit)")
    ap.add_argument("--dtype", choices=["f32", "f16"], default="f32",
                    help="f32 (default, the judged configuration): fp32-grade split arithmetic; f16: reduced precision, one fp16 term per product "
                         "(MCDSEG_CONV_MATH=f16x1; BASELINE config 5's intent) -- reported with its own dtype label")
    ap.add_argument("--other_configs", default="cfg3,cfg4,cfg5,cfg5_f16",
                    help="BASELINE configs 3-5 timed after the judged region (N = 1 only; '' = skip): cfg3 MFNet N=16, cfg4 multitask N=8, "
                         "cfg5 drn_d_105 N=32 at 720x1280 with compact activation storage, cfg5_f16 the same in the reduced-precision arithmetic")
    ap.add_argument("--other_steps", type=int, default=3)
    ap.add_argument("--strict_steps", type=int, default=3,
                    help="timed steps of the judged configuration in each strict-fp32 arithmetic (bf16x6, f32 MFMA) after the timed region, "
                         "reported as 'strict_fp32' (0 = skip)")
    ap.add_argument("--timer", choices=sorted(TIMED_FAMILIES), default="all", help="kernel families bracketed by HIP events")
    ap.add_argument("--timer_steps", type=int, default=1,
                    help="how many of the timed steps (the last ones) carry the per-launch HIP events: bracketing all ~1 500 launches "
                         "of a step costs 3.6 %% of its time (measured, profiles/README.md) and such a step runs its weight gradients on "
                         "the main stream (every pair brackets a kernel running alone), another 3.4 %%; so by default one step pays it")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher: become one.  Nothing above or below this line has initialised a GPU in this process.
        sys.exit(spawn_ranks(args, sys.argv[1:]))
    if os.environ.get("MCDSEG_BENCH_DRY") == "1":
        return dry_run(args)

    import torch
    from mcdseg import dist as mdist
    from mcdseg import ops
    if args.dtype == "f16":  # (as the trainers' --dtype f16: the one-term arithmetic with 2-byte activation storage inside the trunk)
        ops.CONV_MATH = "f16x1"
        if "MCDSEG_ACT_STORAGE" not in os.environ:
            ops.ACT_STORAGE = "compact"
    rank, world, local = mdist.init_from_env()
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP kernels are the only implementation (no CPU fallback)")
    dev = torch.device("cuda", local if world > 1 else 0)
    torch.cuda.set_device(dev)
    if os.environ.get("MCDSEG_BENCH_MAIN_STREAM") == "1":  # (experiments: the step on a stream of its own instead of the legacy default stream)
        torch.cuda.set_stream(torch.cuda.Stream(dev))

    solver, models = build_hip(args, dev)
    # a pool of pre-staged batches (device-resident, as the metric asks), visited round-robin; the caches ops.py hangs on a batch
    # tensor (its bound scalar, the stem's zero-padded companion) are dropped before every step, as for a batch never seen before
    pool = [tuple(t.to(dev) for t in synthetic_batch(args.batch, args.input_ch, args.height, args.width, args.n_class,
                                                     1234 + rank + 1000 * j)) for j in range(max(1, args.batch_pool))]
    turn = [0]

    def next_batch():
        b = pool[turn[0] % len(pool)]
        turn[0] += 1
        for t in b:
            for a in ("_mcd_cbp", "_mcd_bound", "_mcd_cb"):
                if hasattr(t, a):
                    delattr(t, a)
        return b

    timer = LaunchTimer(TIMED_FAMILIES[args.timer])
    ops.LAUNCH_TIMER = timer

    for _ in range(args.warmup):
        c_loss, d_loss = solver.step(*next_batch())
    torch.cuda.synchronize()
    mdist.barrier()
    torch.cuda.synchronize()
    ops.WGRAD_STREAM_STATS.update(deferred=0, no_room=0)
    mdist.COLLECTIVE_EVENTS = [] if mdist.is_distributed() else None
    t0 = time.perf_counter()
    for i in range(args.steps):
        timer.enabled = i >= args.steps - args.timer_steps
        c_loss, d_loss = solver.step(*next_batch())
    torch.cuda.synchronize()
    mdist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    timer.enabled = False
    wgrad_stream = dict(ops.WGRAD_STREAM_STATS)
    collectives, mdist.COLLECTIVE_EVENTS = mdist.COLLECTIVE_EVENTS, None

    def coll_summary(records, steps):
        if records is None:
            return None
        ms = [a.elapsed_time(b) for _, a, b in records]
        big = [(nb, m) for (nb, _, _), m in zip(records, ms) if nb >= (1 << 20)]
        return {"collective_ms_per_step": round(sum(ms) / steps, 3), "collectives_per_step": round(len(ms) / steps, 1),
                "bytes_per_step": int(sum(nb for nb, _, _ in records) / steps),
                "large_all_reduce_avg_ms": round(sum(m for _, m in big) / len(big), 3) if big else None,
                "large_all_reduce_avg_mb": round(sum(nb for nb, _ in big) / len(big) / 1e6, 1) if big else None,
                "note": "HIP events on the compute stream around every all-reduce (or around the wait for a bucketed one): the time that "
                        "stream stands still for the exchange, rank 0"}
    coll = coll_summary(collectives, args.steps)
    # (the name the JSON line gives the exchange: RCCL is torch.distributed's "nccl" backend on ROCm; tests on one GPU use gloo)
    backend_name = {"nccl": "rccl"}.get(torch.distributed.get_backend(), torch.distributed.get_backend()) if mdist.is_distributed() else "none"
    if mdist.is_distributed() and mdist.native_comm() is not None:
        backend_name = "rccl through the C ABI (mcdseg_allreduce)"  # MCDSEG_NATIVE_RCCL=1
    el = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if mdist.is_distributed():
        torch.distributed.all_reduce(el, op=torch.distributed.ReduceOp.MAX)
    elapsed = float(el)
    c_loss, d_loss = float(c_loss), float(d_loss)

    # for reference, outside the timed region: the same step with the reference's literal schedule (7 generator forwards)
    literal = None
    if solver.reuse_tgt and args.literal_steps > 0:
        solver.reuse_tgt = False
        solver.step(*next_batch())
        torch.cuda.synchronize()
        mdist.barrier()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.literal_steps):
            solver.step(*next_batch())
        torch.cuda.synchronize()
        mdist.barrier()
        torch.cuda.synchronize()
        lt = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device=dev)
        if mdist.is_distributed():
            torch.distributed.all_reduce(lt, op=torch.distributed.ReduceOp.MAX)
        solver.reuse_tgt = True
        literal = {"steps": args.literal_steps, "ms_per_step": round(1e3 * float(lt) / args.literal_steps, 2),
                   "value": round(args.batch * world * args.literal_steps / float(lt), 3),
                   "note": "same build with solver.reuse_tgt = False (7 generator forwards + 5 backwards per step), measured after the "
                           "timed region; identical weights, statistics and losses"}

    # N > 1, outside the timed region: the same step with the OTHER setting of MCDSEG_DP_OVERLAP (gradients exchanged in buckets during
    # the backward pass instead of one all-reduce of the flat buffer in step(), or the reverse) -- one invocation decides the default
    dp_other = None
    if mdist.is_distributed() and args.literal_steps > 0:
        from mcdseg import optim as moptim
        was = moptim.DP_OVERLAP
        opts = [o for o in (getattr(solver, "opt_g", None), getattr(solver, "opt_f", None)) if hasattr(o, "_setup_overlap")]
        moptim.DP_OVERLAP = not was
        for o in opts:
            o._setup_overlap()
        solver.step(*next_batch())
        torch.cuda.synchronize()
        mdist.barrier()
        torch.cuda.synchronize()
        mdist.COLLECTIVE_EVENTS = []
        t3 = time.perf_counter()
        for _ in range(args.literal_steps):
            solver.step(*next_batch())
        torch.cuda.synchronize()
        mdist.barrier()
        torch.cuda.synchronize()
        dt3 = torch.tensor([time.perf_counter() - t3], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(dt3, op=torch.distributed.ReduceOp.MAX)
        recs, mdist.COLLECTIVE_EVENTS = mdist.COLLECTIVE_EVENTS, None
        dp_other = {"MCDSEG_DP_OVERLAP": "1" if moptim.DP_OVERLAP else "0", "steps": args.literal_steps,
                    "ms_per_step": round(1e3 * float(dt3) / args.literal_steps, 2),
                    "value": round(args.batch * world * args.literal_steps / float(dt3), 3), "collectives": coll_summary(recs, args.literal_steps),
                    "note": "same build and ranks with the other setting of MCDSEG_DP_OVERLAP, measured after the timed region (MAX over ranks); "
                            "the timed line above ran with MCDSEG_DP_OVERLAP=%s" % ("1" if was else "0")}
        moptim.DP_OVERLAP = was
        for o in opts:
            o._setup_overlap()

    # likewise outside the timed region: the same step with the weight gradients on the main stream, the same-box
    # price of the second stream
    one_stream = None
    if ops.OVERLAP_WGRAD != "0" and args.literal_steps > 0:
        mode, ops.OVERLAP_WGRAD = ops.OVERLAP_WGRAD, "0"
        solver.step(*next_batch())
        torch.cuda.synchronize()
        mdist.barrier()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        for _ in range(args.literal_steps):
            solver.step(*next_batch())
        torch.cuda.synchronize()
        mdist.barrier()
        torch.cuda.synchronize()
        ot = torch.tensor([time.perf_counter() - t2], dtype=torch.float64, device=dev)
        if mdist.is_distributed():
            torch.distributed.all_reduce(ot, op=torch.distributed.ReduceOp.MAX)
        ops.OVERLAP_WGRAD = mode
        one_stream = {"steps": args.literal_steps, "ms_per_step": round(1e3 * float(ot) / args.literal_steps, 2),
                      "value": round(args.batch * world * args.literal_steps / float(ot), 3),
                      "note": "same build with MCDSEG_OVERLAP_WGRAD=0 (every kernel on one stream), measured after the timed region; "
                              "identical results bit for bit"}

    others = None
    judged_cfg = (args.net, args.batch, args.height, args.width, args.input_ch) == ("drn_d_38", 16, 480, 640, 6)
    if world == 1 and not mdist.is_distributed() and judged_cfg and args.other_configs.strip():
        # free the judged configuration first: cfg5 at N = 32 takes 232 of the 288 GB
        kern_summary = timer.summary()
        reuse_tgt = solver.reuse_tgt
        ops.LAUNCH_TIMER = None
        del solver, models, pool
        torch.cuda.empty_cache()
        others = {}
        for tag in [t.strip() for t in args.other_configs.split(",") if t.strip()]:
            try:
                others[tag] = other_config(tag, dev, max(1, args.other_steps), args.n_class, want_roofline=tag.startswith("cfg"))
            except Exception as e:  # (a configuration that does not fit or fails must not take the judged line down)
                others[tag] = {"error": "%s: %s" % (type(e).__name__, e)}
                torch.cuda.empty_cache()
    else:
        kern_summary, reuse_tgt = timer.summary(), solver.reuse_tgt
        ops.LAUNCH_TIMER = None
        del solver, models, pool
        torch.cuda.empty_cache()
    # the judged configuration in the two STRICT fp32 arithmetics (24-bit operands), after the timed region: what an IEEE-grade
    # product costs next to the default's 22-bit operands
    strict = None
    if world == 1 and not mdist.is_distributed() and judged_cfg and args.strict_steps > 0 and ops.CONV_MATH == "f16x3":
        strict = {}
        for tag, key in (("cfg2_bf16x6", "bf16x6"), ("cfg2_f32", "f32")):
            try:
                r = other_config(tag, dev, args.strict_steps, args.n_class)
                strict[key] = {"ms_per_step": r["ms_per_step"], "value": r["pairs_per_s"], "steps": r["steps"], "arithmetic": r["arithmetic"],
                               "c_loss": r["c_loss"], "d_loss": r["d_loss"]}
            except Exception as e:
                strict[key] = {"error": "%s: %s" % (type(e).__name__, e)}
                torch.cuda.empty_cache()

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        pairs = args.batch * world * args.steps
        value = pairs / elapsed
        kern = kern_summary
        total_ms = sum(k["ms"] for k in kern.values()) or 1.0
        for k in kern.values():
            k["avg_ms"] = k["ms"] / max(k["launches"], 1)
            k["tflops"] = k["flops"] / (k["ms"] * 1e-3) / 1e12 if k["ms"] > 0 else 0.0
            k["gbs"] = k["bytes"] / (k["ms"] * 1e-3) / 1e9 if k["ms"] > 0 else 0.0
            k["share"] = round(k["ms"] / total_ms, 4)
        default_cfg = (args.net, args.batch, args.height, args.width, args.input_ch) == ("drn_d_38", 16, 480, 640, 6)
        # dominant kernel = the one with the largest summed duration inside the timed region; in the steps that carry the event
        # pairs every launch runs alone (ops._conv_backward keeps the weight gradients on the main stream there)
        roofline = roofline_fwd = roofline_wg = None
        tsteps = max(1, min(args.steps, args.timer_steps))

        def priced(name, table):
            traffic, source = pmc_traffic(name, kern, tsteps) if default_cfg else (None, "not the judged configuration")
            r = kernel_roofline(name, table[name], ops.CONV_MATH, traffic)
            r["traffic_source"] = source
            return r

        if kern:
            dom = max(kern, key=lambda n: kern[n]["ms"])
            roofline = priced(dom, kern)
            fwd = {n: v for n, v in kern.items() if is_forward_conv(n)}
            if fwd:
                roofline_fwd = priced(max(fwd, key=lambda n: fwd[n]["flops"]), fwd)
            wgk = {n: v for n, v in kern.items() if n.startswith("conv_wgrad")}
            if wgk:
                roofline_wg = priced(max(wgk, key=lambda n: wgk[n]["flops"]), wgk)
        # whole-step accounting on the algorithmic work of SURVEY.md 8d (drn_d_38 @ 6x480x640 only)
        step_acc = None
        if args.net == "drn_d_38" and (args.height, args.width, args.input_ch) == (480, 640, 6):
            # passes actually executed per pair: step B's two generator backward passes elided; its target forward is step C's first
            fwd_p, bwd_p = (6 if reuse_tgt else 7), 5
            gf = GF_FWD_PER_IMG * (fwd_p + 2 * bwd_p)
            gb = GB_FWD_PER_IMG * fwd_p + GB_BWD_PER_IMG * bwd_p
            t_pair = elapsed / (args.batch * args.steps)
            mult, peak, _ = MATH.get(ops.CONV_MATH, MATH["bf16x6"])
            t_mfma = gf * mult / (peak * 1e3)  # seconds per pair if every FLOP ran at the matrix peak of the arithmetic used
            t_hbm = gb / PEAK_HBM_GBS
            step_acc = {"alg_gflop_per_pair": round(gf, 1), "alg_gb_per_pair": round(gb, 2),
                        "tflops": round(gf / t_pair / 1e3, 2), "frac_fp32_peak": round(gf / t_pair / 1e3 / PEAK_FP32_TFLOPS, 4),
                        "frac_mfma_roof_of_conv_math": round(t_mfma / t_pair, 4),
                        "hbm_gbs": round(gb / t_pair, 1), "frac_hbm_peak": round(gb / t_pair / PEAK_HBM_GBS, 4),
                        "frac_hbm_peak_ceiling": round(t_hbm / max(t_hbm, t_mfma), 4),
                        "note": "the step is matrix-rate bound (SURVEY.md F7): even at the MFMA peak of its arithmetic the algorithmic "
                                "bytes would use only frac_hbm_peak_ceiling of the 8 TB/s HBM peak, so the north-star's 40 %-of-HBM "
                                "target is out of reach for fp32-grade results; per-kernel HBM fractions are under 'kernels'",
                        "timed_kernel_ms_per_step": round(total_ms / max(1, min(args.steps, args.timer_steps)), 2),
                        "ref_faithful_gflop_per_pair": round(GF_FWD_PER_IMG * 21, 1)}
        line = {
            "metric": "RGB-D img/s (6x480x640) MCD train step, drn_d_38", "value": round(value, 3), "unit": "img/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 2),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": DTYPE_LABEL.get(ops.CONV_MATH, ops.CONV_MATH), "data": "synthetic",
            "config": {"workload": "adapt_trainer MCD early-fusion %s %d-ch, bs=%d/GPU synthetic %dx%d, full A+B+C step (num_k=4)"
                                   % (args.net, args.input_ch, args.batch, args.height, args.width),
                       "pairs_per_gpu": args.batch, "global_pairs": args.batch * world, "parallelism": "dp%d" % world,
                       "collectives": ("none (single process)" if not mdist.is_distributed() else
                                       ("%s (forced, 1 rank)" % backend_name if world == 1 else "%s all-reduce of the flat gradient buffer over %d ranks, one rank "
                                        "per GPU%s" % (backend_name, world, ", bucketed during backward (MCDSEG_DP_OVERLAP=1)" if os.environ.get("MCDSEG_DP_OVERLAP") == "1" else ""))),
                       "streams": {"0": "one stream", "1": "weight gradients on a side stream beside their data gradient",
                                   "2": "weight gradients of the trunk on a side stream, joined when each backward pass ends (MCDSEG_OVERLAP_WGRAD=2); "
                                        "the timer_steps run them on the main stream, so every HIP-event pair brackets a kernel running alone"}[ops.OVERLAP_WGRAD],
                       "n_class": args.n_class, "conv_math": ops.CONV_MATH, "c_loss": c_loss, "d_loss": d_loss, "timer": args.timer,
                       "timer_steps": min(args.steps, args.timer_steps),
                       "schedule": "the reference's A+B+C statements with the results-neutral elisions of solvers/solver.py: no generator "
                                   "backward in step B (its gradients are zeroed unused)" + (
                                       "; step B's target forward doubles as step C's first (generator unchanged in between; BatchNorm "
                                       "running update applied twice) -- %d generator forwards + 5 backwards per step, bit-identical "
                                       "weights/statistics/losses to the literal 7-forward schedule (--no_forward_reuse runs that)"
                                       % 6 if reuse_tgt else "; literal 7 generator forwards + 5 backwards per step")},
            "value_literal_schedule": literal["value"] if literal else (round(value, 3) if not reuse_tgt else None),
            "ms_per_step_literal_schedule": literal["ms_per_step"] if literal else (round(ms_per_step, 2) if not reuse_tgt else None),
            "literal_schedule": literal,
            "one_stream": one_stream,
            "wgrad_stream": wgrad_stream,
            "collectives": coll,
            "dp_overlap_other_setting": dp_other,
            "other_configs": others,
            "strict_fp32": strict,
            "roofline": roofline,
            "roofline_forward": roofline_fwd,
            "roofline_wgrad": roofline_wg,
            "step_accounting": step_acc,
            "kernels": {k: {"launches": v["launches"], "ms_total": round(v["ms"], 2), "avg_ms": round(v["avg_ms"], 4), "share": v["share"],
                            "tflops": round(v["tflops"], 2), "alg_gbs": round(v["gbs"], 1)}
                        for k, v in sorted(kern.items(), key=lambda kv: -kv[1]["ms"])},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args)
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line), file=sys.__stdout__, flush=True)
    mdist.barrier()
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
