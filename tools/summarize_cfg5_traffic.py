#!/usr/bin/env python3
"""gpurun_out/<tag>_cfg5_{half,full}_{fetch,write} (tools/run_cfg5_traffic.sh) -> profiles/<tag>_cfg5_bn_traffic.json: HBM bytes of every
kernel family of one MCD step of drn_d_105 at 8 x 6 x 720 x 1280 in the one-term arithmetic, with round 5's storage ("full": two fp16
pieces per activation, fp32 z and gradients) and with round 6's 2-byte storage ("half").  hbm bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024
(MI355X_MICROARCH.md, HBM section: the gfx950 FETCH_SIZE correction), per launch and summed over the step."""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    if name.startswith("_ZN"):  # (a name rocprofv3 could not demangle: _Float16 / __bf16 vector types in the signature)
        import re
        m = re.search(r"\d+([a-z_0-9]+_kernel)(?:ILi(\d+)E)?", name)
        if m:
            return m.group(1) + ("<%s>" % m.group(2) if m.group(2) else "")
    return name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]


def family(k):
    for f in ("bn_apply", "bn_bwd_apply", "bn_bwd_reduce", "bn_bwd_finalize", "bn_stats", "conv_gemm_split_pp", "conv_gemm_split", "conv_wgrad", "conv_thin",
              "conv_stem", "up8", "softmax", "sgd", "pack"):
        if k.startswith(f):
            return f
    return "other"


def counters(path, counter):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            agg[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return agg


def main(tag):
    src = os.path.join(ROOT, "gpurun_out")
    out = {}
    for mode in ("full", "half"):
        f = glob.glob(os.path.join(src, "%s_cfg5_%s_fetch" % (tag, mode), "*", "*counter_collection.csv"))
        w = glob.glob(os.path.join(src, "%s_cfg5_%s_write" % (tag, mode), "*", "*counter_collection.csv"))
        if not (f and w):
            print("missing counter files for", mode)
            continue
        fetch, write = counters(f[0], "FETCH_SIZE"), counters(w[0], "WRITE_SIZE")
        fam = collections.defaultdict(lambda: [0, 0.0])
        kern = {}
        for k, v in fetch.items():
            if k not in write:
                continue
            byts = (2 * sum(v) + sum(write[k])) * 1024
            fam[family(k)][0] += len(v)
            fam[family(k)][1] += byts
            if k.startswith("bn_"):
                kern[k] = {"launches": len(v), "hbm_mb_per_launch": round(byts / len(v) / 1e6, 1)}
        out[mode] = {"families": {k: {"launches": n, "hbm_gb_per_step": round(b / 1e9, 2), "hbm_mb_per_launch": round(b / n / 1e6, 1)}
                                  for k, (n, b) in sorted(fam.items())}, "batchnorm_kernels": kern}
    if "full" in out and "half" in out:
        bn = lambda m: sum(v["hbm_gb_per_step"] for k, v in out[m]["families"].items() if k.startswith("bn_"))  # noqa: E731
        out["batchnorm_hbm_gb_per_step"] = {"full": round(bn("full"), 1), "half": round(bn("half"), 1), "ratio": round(bn("half") / bn("full"), 3)}
        tot = lambda m: sum(v["hbm_gb_per_step"] for v in out[m]["families"].values())  # noqa: E731
        out["all_kernels_hbm_gb_per_step"] = {"full": round(tot("full"), 1), "half": round(tot("half"), 1), "ratio": round(tot("half") / tot("full"), 3)}
    q = glob.glob(os.path.join(src, "%s_cfg5_half_sq" % tag, "*", "*counter_collection.csv"))
    if q:
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(q[0])):
            agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
        sq = {}
        for k, v in agg.items():
            if not k.startswith(("conv_", "bn_")) or "GRBM_GUI_ACTIVE" not in v:
                continue
            tot = {c: sum(x) for c, x in v.items()}
            cyc = tot["GRBM_GUI_ACTIVE"] / 8.0  # the counter sums the 8 XCDs
            if cyc <= 0 or tot["SQ_WAVE_CYCLES"] <= 0:
                continue
            sq[k] = {"launches": len(v["SQ_WAVE_CYCLES"]), "kernel_cycles_per_launch": int(cyc / len(v["SQ_WAVE_CYCLES"])),
                     "mfma_pipe_busy_frac": round(tot["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * cyc), 3),
                     "wave_cycles_parked_frac": round(tot["SQ_WAIT_ANY"] / tot["SQ_WAVE_CYCLES"], 3),
                     "wave_cycles_issue_stalled_frac": round(tot["SQ_WAIT_INST_ANY"] / tot["SQ_WAVE_CYCLES"], 3)}
        out["half_sq"] = {k: sq[k] for k in sorted(sq, key=lambda n: -sq[n]["kernel_cycles_per_launch"] * sq[n]["launches"])[:24]}
    out["command"] = "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE (separate passes) -- python3 tools/bench_configs.py --cfg cfg5 --n5 8 --hw5 720 1280 --steps 0 " \
                     "with MCDSEG_CONV_MATH=f16x1 MCDSEG_ACT_STORAGE=compact and MCDSEG_HALF_STORAGE=0 (full) / 1 (half): one MCD step"
    path = os.path.join(ROOT, "profiles", tag + "_cfg5_bn_traffic.json")
    json.dump(out, open(path, "w"), indent=1, sort_keys=True)
    print(path)
    print(json.dumps({k: v for k, v in out.items() if k.endswith("per_step")}, indent=1))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r06")
