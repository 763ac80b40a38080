#!/usr/bin/env python3
"""Benchmark of the MCD hot path on MI355X (BASELINE.json metric).

One *step* = one full three-step MCD update (A: G+F on source, B: F on source CE - target discrepancy,
C: num_k = 4 generator updates on the target discrepancy; adapt_trainer.py:155-220 of the reference) over
one synthetic batch of N (source, target) RGB+HHA pairs per GPU, i.e. 7 generator forward passes and
5 generator backward passes (the reference-faithful count is 7; the two unused step-B backward passes are
elided).  ``value`` = pairs/s over the whole job, inputs resident in HBM.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  ``roofline`` prices the dominant kernel (the 128x128-tile implicit-GEMM
convolution) from HIP events recorded around its launches inside the timed region: algorithmic FLOPs
of those launches / their summed duration, against the fp32 MFMA peak -- the pass is fp32-FLOP-bound,
not HBM-bound; the HBM fractions of the whole step and of the streaming loss kernel are
reported next to it.  ``cpu_baseline`` times the CPU oracle (plain PyTorch restatement of the reference)
on a bounded sample of the same workload on the host cores.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "multichannel-semseg-with-uda_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)
os.environ.setdefault("MCDSEG_PRETRAINED", "0")

import torch  # noqa: E402

PEAK_FP32_TFLOPS = 157.3   # MI355X fp32 vector = fp32-input MFMA (MI355X_MICROARCH.md, chip-level table)
PEAK_BF16_TFLOPS = 2500.0  # dense bf16 MFMA (MI355X_MICROARCH.md, chip-level table; the 5 PF figure is 2:1 sparse)
PEAK_HBM_GBS = 8000.0      # HBM3E spec
# algorithmic work per image-pass (one image through G+F1+F2 forward+backward incl. loss), drn_d_38 6x480x640
# (SURVEY.md section 8d / BASELINE.md section 3): 781.5 GFLOP, 1.94 GB
GF_FWD_PER_IMG = 260.5
GB_FWD_PER_IMG, GB_BWD_PER_IMG = 0.71, 1.23


DTYPE_LABEL = {"f32": "f32", "bf16x6": "f32 (operands split 3-way into bf16, 6 cross terms on the bf16 MFMA pipe, fp32 accumulate)",
               "mixed": "f32 (bf16x6 split for forward/dgrad, f32 MFMA for wgrad)"}


class LaunchTimer:
    """Collects HIP-event pairs recorded around selected kernel launches (mcdseg.ops.LAUNCH_TIMER)."""

    def __init__(self, names):
        self.names = set(names)
        self.records = []
        self.enabled = False

    def wants(self, name):
        return self.enabled and any(name.startswith(n) for n in self.names)

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
    g = torch.Generator().manual_seed(seed)
    src = torch.randn(n, ch, h, w, generator=g)
    lbl = torch.randint(0, n_class, (n, h, w), generator=g, dtype=torch.int64)
    tgt = torch.randn(n, ch, h, w, generator=g)
    return src, lbl, tgt


def build_hip(args, dev):
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
    return solver, (g, f1, f2)


def cpu_baseline(args):
    """The CPU oracle on the host cores: one full MCD step on ONE (src,tgt) pair of the bench geometry."""
    from oracle import ref_loss, ref_mcd, ref_models
    torch.manual_seed(0)
    threads = torch.get_num_threads()
    g, f1, f2 = ref_models.get_models(args.net, args.input_ch, args.n_class)
    for m in (g, f1, f2):
        m.train()
    og = ref_models.get_optimizer(g.parameters(), "sgd", 1e-3, 0.9, 2e-5)
    of = ref_models.get_optimizer(list(f1.parameters()) + list(f2.parameters()), "sgd", 1e-3, 0.9, 2e-5)
    w = ref_loss.class_weights(args.n_class)
    n = args.cpu_pairs
    src, lbl, tgt = synthetic_batch(n, args.input_ch, args.height, args.width, args.n_class, 1234)
    t0 = time.perf_counter()
    ref_mcd.mcd_step(g, f1, f2, og, of, ref_loss.CrossEntropyLoss2d(w), ref_loss.Diff2d(), src, lbl, tgt)
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "img/s", "cores": threads, "kind": "port",
            "sample": "1 full MCD step (A+B+C, num_k=4; 7 fwd + 7 bwd passes) of the CPU oracle on %d pair(s) of %dx%dx%d, "
                      "%.1f s on %d threads, no warm-up" % (n, args.input_ch, args.height, args.width, dt, threads)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=16, help="(src,tgt) pairs per GPU")
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--net", default="drn_d_38")
    ap.add_argument("--input_ch", type=int, default=6)
    ap.add_argument("--n_class", type=int, default=41)
    ap.add_argument("--cpu_pairs", type=int, default=1)
    ap.add_argument("--no_cpu_baseline", action="store_true")
    args = ap.parse_args()

    from mcdseg import dist as mdist
    from mcdseg import ops
    rank, world, local = mdist.init_from_env()
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP kernels are the only implementation (no CPU fallback)")
    dev = torch.device("cuda", local if world > 1 else 0)
    torch.cuda.set_device(dev)

    solver, models = build_hip(args, dev)
    src, lbl, tgt = (t.to(dev) for t in synthetic_batch(args.batch, args.input_ch, args.height, args.width, args.n_class,
                                                        1234 + rank))
    timer = LaunchTimer(["conv_gemm_kernel", "conv_gemm_x6_kernel", "softmax_ce_l1_kernel"])
    ops.LAUNCH_TIMER = timer

    for _ in range(args.warmup):
        c_loss, d_loss = solver.step(src, lbl, tgt)
    torch.cuda.synchronize()
    mdist.barrier()
    torch.cuda.synchronize()
    timer.enabled = True
    t0 = time.perf_counter()
    for _ in range(args.steps):
        c_loss, d_loss = solver.step(src, lbl, tgt)
    torch.cuda.synchronize()
    mdist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    timer.enabled = False
    el = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        torch.distributed.all_reduce(el, op=torch.distributed.ReduceOp.MAX)
    elapsed = float(el)
    c_loss, d_loss = float(c_loss), float(d_loss)

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        pairs = args.batch * world * args.steps
        value = pairs / elapsed
        kern = timer.summary()
        for k in kern.values():
            k["avg_ms"] = k["ms"] / max(k["launches"], 1)
            k["tflops"] = k["flops"] / (k["ms"] * 1e-3) / 1e12 if k["ms"] > 0 else 0.0
            k["gbs"] = k["bytes"] / (k["ms"] * 1e-3) / 1e9 if k["ms"] > 0 else 0.0
        convs = {k: v for k, v in kern.items() if k.startswith("conv_gemm")}
        # dominant kernel = the instantiation carrying the most algorithmic FLOPs of the step (the forward 128x128
        # tile: 7 forward passes vs 5 backward); its launches run alone on the stream, so the event pairs are clean
        dom_name = max(convs, key=lambda k: convs[k]["flops"]) if convs else None
        roofline = None
        traffic = None
        default_cfg = (args.net, args.batch, args.height, args.width, args.input_ch) == ("drn_d_38", 16, 480, 640, 6)
        if dom_name and default_cfg:  # the PMC passes were taken on exactly this workload
            # HBM-side bytes per launch from the committed PMC passes (profiles/<round>_pmc_traffic.json: FETCH_SIZE and
            # WRITE_SIZE collected in separate rocprofv3 --pmc runs of this same bench, FETCH_SIZE doubled per the gfx950 note)
            import glob
            for fn in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")), reverse=True):
                try:
                    traffic = json.load(open(fn))["kernels"][dom_name]["hbm_bytes_per_launch"]
                    break
                except (KeyError, ValueError, OSError):
                    continue
        if dom_name:
            dom = convs[dom_name]
            x6 = "x6" in dom_name
            # bf16x6: every algorithmic fp32 MAC is executed as six bf16 MACs on v_mfma_f32_32x32x16_bf16, so the kernel is
            # priced in EXECUTED bf16 FLOPs against the dense bf16 MFMA peak; the f32 kernel against the f32 MFMA peak
            mult, peak = (6.0, PEAK_BF16_TFLOPS) if x6 else (1.0, PEAK_FP32_TFLOPS)
            roofline = {"bound": "mfma",
                        "kernel": dom_name + (" (implicit-GEMM conv, fp32 operands as 3-way bf16 split, 6 cross terms on "
                                              "v_mfma_f32_32x32x16_bf16)" if x6 else " (implicit-GEMM conv on v_mfma_f32_32x32x2_f32)"),
                        "achieved": round(mult * dom["tflops"], 2), "peak": peak, "unit": "TFLOP/s",
                        "frac": round(mult * dom["tflops"] / peak, 4), "traffic": traffic,
                        "alg_tflops_fp32_equivalent": round(dom["tflops"], 2),
                        "alg_bytes_per_launch": dom["bytes"] / dom["launches"],
                        "launches": dom["launches"], "avg_launch_ms": round(dom["avg_ms"], 4),
                        "alg_flops_per_launch": dom["flops"] / dom["launches"]}
        # whole-step accounting on the algorithmic work of SURVEY.md 8d (drn_d_38 @ 6x480x640 only)
        step_acc = None
        if args.net == "drn_d_38" and (args.height, args.width, args.input_ch) == (480, 640, 6):
            fwd, bwd = 7, 5  # passes actually executed per pair (step-B generator backward elided)
            gf = GF_FWD_PER_IMG * (fwd + 2 * bwd)
            gb = GB_FWD_PER_IMG * fwd + GB_BWD_PER_IMG * bwd
            t_pair = elapsed / (args.batch * args.steps)
            step_acc = {"alg_gflop_per_pair": round(gf, 1), "alg_gb_per_pair": round(gb, 2),
                        "tflops": round(gf / t_pair / 1e3, 2), "frac_fp32_peak": round(gf / t_pair / 1e3 / PEAK_FP32_TFLOPS, 4),
                        "hbm_gbs": round(gb / t_pair, 1), "frac_hbm_peak": round(gb / t_pair / PEAK_HBM_GBS, 4),
                        "ref_faithful_gflop_per_pair": round(GF_FWD_PER_IMG * 21, 1)}
        line = {
            "metric": "RGB-D img/s (6x480x640) MCD train step, drn_d_38", "value": round(value, 3), "unit": "img/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 2),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": DTYPE_LABEL.get(ops.CONV_MATH, ops.CONV_MATH), "data": "synthetic",
            "config": {"workload": "adapt_trainer MCD early-fusion %s %d-ch, bs=%d/GPU synthetic %dx%d, full A+B+C step (num_k=4)"
                                   % (args.net, args.input_ch, args.batch, args.height, args.width),
                       "pairs_per_gpu": args.batch, "global_pairs": args.batch * world, "parallelism": "dp%d" % world,
                       "n_class": args.n_class, "conv_math": ops.CONV_MATH, "c_loss": c_loss, "d_loss": d_loss},
            "roofline": roofline,
            "step_accounting": step_acc,
            "kernels": {k: {"launches": v["launches"], "ms_total": round(v["ms"], 2), "avg_ms": round(v["avg_ms"], 4),
                            "tflops": round(v["tflops"], 2), "alg_gbs": round(v["gbs"], 1)} for k, v in kern.items()},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args)
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line))
    mdist.barrier()
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
