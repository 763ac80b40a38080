#!/usr/bin/env python3
"""Turn raw rocprofv3 output merged into gpurun_out/ into the small, committed summaries under profiles/.

    python tools/summarize_profiles.py r01c      # reads gpurun_out/r01c_{stats,fetch,write}, writes profiles/history/r01c_*

* <tag>_kernel_stats.csv : rocprofv3 --kernel-trace --stats summary of `bench.py --steps 2 --warmup 1` (verbatim)
* <tag>_pmc_sq.json      : per kernel MFMA-pipe busy fraction and wave-cycle breakdown from gpurun_out/<tag>_sq (SQ counters)
* <tag>_pmc_traffic.json : per kernel (all launches of one MCD step): launches, mean FETCH_SIZE / WRITE_SIZE, and
                           HBM-side bytes per launch = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024.  The two counters come from
                           separate --pmc passes (TCC slot budget); FETCH_SIZE is doubled per the gfx950 note in
                           MI355X_MICROARCH.md (HBM section) -- calibrated here on kernels with known byte counts
                           (softmax_ce_l1: 2 x 806 MB read -> FETCH_SIZE 806 MB; bn_apply, up8_fwd likewise).
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    return name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]


def counters(path, counter):
    agg = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        d = agg.setdefault(short(r["Kernel_Name"]), [])
        d.append(float(r["Counter_Value"]))
    return agg


def newest(pattern):
    """a tag collected twice leaves both runs' files in gpurun_out/ (gpurun merges, it does not replace): the latest run counts"""
    return sorted(glob.glob(pattern), key=os.path.getmtime, reverse=True)


def main(tag):
    out = os.path.join(ROOT, "profiles")
    src = os.path.join(ROOT, "gpurun_out")
    stats = newest(os.path.join(src, tag + "_stats", "*", "*kernel_stats.csv"))
    if stats:
        shutil.copy(stats[0], os.path.join(out, tag + "_bench_kernel_stats.csv"))
    f = newest(os.path.join(src, tag + "_fetch", "*", "*counter_collection.csv"))
    w = newest(os.path.join(src, tag + "_write", "*", "*counter_collection.csv"))
    if f and w:
        fetch, write = counters(f[0], "FETCH_SIZE"), counters(w[0], "WRITE_SIZE")
        table = {}
        for k, v in fetch.items():
            if k not in write or not k.startswith(("conv_", "bn_", "up8_", "softmax", "sgd", "wgrad", "pack", "bilinear", "mse")):
                continue
            fk, wk = sum(v) / len(v), sum(write[k]) / len(write[k])
            table[k] = {"launches_per_step": len(v), "fetch_size_kb_mean": round(fk, 1), "write_size_kb_mean": round(wk, 1),
                        "hbm_bytes_per_launch": int((2 * fk + wk) * 1024)}
        sys.path.insert(0, os.path.join(ROOT, "multichannel-semseg-with-uda_amd"))
        from mcdseg import _lib
        json.dump({"command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE (two passes) -- python3 bench.py --steps 1 --warmup 0",
                   # (of the kernel sources the counters were collected with, written on the box by tools/run_profiles.sh: bench.py checks it)
                   "source_fingerprint": (open(os.path.join(src, tag + "_fingerprint.txt")).read().strip()
                                          if os.path.exists(os.path.join(src, tag + "_fingerprint.txt")) else _lib.source_fingerprint()),
                   "units": "FETCH_SIZE/WRITE_SIZE in KiB as reported; hbm_bytes = (2*FETCH + WRITE) * 1024 (gfx950 FETCH_SIZE halving)",
                   "kernels": table}, open(os.path.join(out, tag + "_pmc_traffic.json"), "w"), indent=1, sort_keys=True)
    q = newest(os.path.join(src, tag + "_sq", "*", "*counter_collection.csv"))
    if q:
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(q[0])):
            agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
        table = {}
        for k, v in agg.items():
            if not k.startswith(("conv_", "bn_", "softmax", "up8")) or "GRBM_GUI_ACTIVE" not in v:
                continue
            tot = {c: sum(x) for c, x in v.items()}
            n = len(v["SQ_WAVE_CYCLES"])
            cyc = tot["GRBM_GUI_ACTIVE"] / 8.0  # the counter sums the 8 XCDs
            if cyc <= 0 or tot["SQ_WAVE_CYCLES"] <= 0:
                continue
            table[k] = {"launches_per_step": n, "kernel_cycles_per_launch": int(cyc / n),
                        "mfma_pipe_busy_frac": round(tot["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * cyc), 3),
                        "wave_cycles_parked_frac": round(tot["SQ_WAIT_ANY"] / tot["SQ_WAVE_CYCLES"], 3),
                        "wave_cycles_issue_stalled_frac": round(tot["SQ_WAIT_INST_ANY"] / tot["SQ_WAVE_CYCLES"], 3),
                        "wave_cycles_issuing_frac": round(tot["SQ_ACTIVE_INST_ANY"] / tot["SQ_WAVE_CYCLES"], 3)}
        json.dump({"command": "rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY "
                              "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -- python3 bench.py --steps 1 --warmup 0 --no_cpu_baseline",
                   "notes": "kernel cycles = GRBM_GUI_ACTIVE / 8 (the counter sums the 8 XCDs); mfma_pipe_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / "
                            "(1024 SIMDs * kernel cycles), i.e. at the clock the kernel actually ran at; the wave-cycle fractions are of "
                            "SQ_WAVE_CYCLES (parked = s_waitcnt / barrier, issue_stalled = waiting on a pipe, mostly the MFMA pipe)",
                   "kernels": table}, open(os.path.join(out, tag + "_pmc_sq.json"), "w"), indent=1, sort_keys=True)
    for suffix in ("", "_profiled", "_bf16x6", "_f32mfma", "_f16x1"):  # the benchmark lines (the last line that parses as JSON)
        b = os.path.join(src, tag + "_bench" + suffix + ".json")
        if not os.path.exists(b):
            continue
        lines = [ln for ln in open(b).read().splitlines() if ln.startswith("{")]
        if lines:
            json.loads(lines[-1])
            open(os.path.join(out, tag + "_bench_line" + suffix + ".json"), "w").write(lines[-1] + "\n")
    print("profiles/ updated for", tag)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r01c")
