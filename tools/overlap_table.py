#!/usr/bin/env python3
"""What the two streams of a benchmark step actually overlap.

Input: a rocprofv3 ``--kernel-trace`` CSV of ``bench.py`` (start / end timestamps and the queue of every dispatch).  The timed steps are
cut out between optimizer launches (7 ``sgd_momentum`` launches per MCD step), every kernel is classed as MATRIX-bound (the
convolution GEMMs) or HBM-bound (BatchNorm passes, up-sampler, loss, slab reduces, optimizer, packing), and for every kernel
family the table says how much of its time it ran ALONE, beside a matrix-bound kernel of ANOTHER queue, and beside an HBM-bound one.

    python tools/overlap_table.py gpurun_out/<tag>_trace [--warmup 2 --steps 3] > profiles/<tag>_overlap.txt
"""
import argparse
import bisect
import collections
import csv
import glob
import json
import os
import sys

MATRIX = ("conv_gemm", "conv_wgrad", "conv_stem", "conv_thin_window")
KEEP = MATRIX + ("bn_", "up8_", "softmax", "sgd_", "wgrad_", "pack_", "split_", "absmax", "label_weight", "unsplit")


def short(name):
    return name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]


def family(name):
    """the template arguments that only name the arithmetic are dropped; tile shapes stay"""
    return name.replace("SplitF16x3, ", "").replace("<SplitF16x3>", "")


def overlap(iv_starts, iv_ends, s, e):
    """length of [s, e) covered by the sorted, disjoint intervals"""
    i = bisect.bisect_right(iv_ends, s)
    tot = 0
    while i < len(iv_starts) and iv_starts[i] < e:
        tot += min(e, iv_ends[i]) - max(s, iv_starts[i])
        i += 1
    return tot


def merged(ivs):
    ivs = sorted(ivs)
    out = []
    for s, e in ivs:
        if out and s <= out[-1][1]:
            out[-1][1] = max(out[-1][1], e)
        else:
            out.append([s, e])
    return [a for a, _ in out], [b for _, b in out]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace_dir")
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--json", default=None)
    args = ap.parse_args()
    files = glob.glob(os.path.join(args.trace_dir, "*", "*kernel_trace.csv")) or glob.glob(os.path.join(args.trace_dir, "*kernel_trace.csv"))
    if not files:
        sys.exit("no *kernel_trace.csv under %s" % args.trace_dir)
    rows = []
    for r in csv.DictReader(open(files[0])):
        n = short(r["Kernel_Name"])
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n, r.get("Queue_Id", "0")))
    rows.sort()
    sgd = [r for r in rows if r[2].startswith("sgd_momentum")]
    lo = sgd[7 * args.warmup - 1][1] if args.warmup > 0 else rows[0][0]
    hi = sgd[7 * (args.warmup + args.steps) - 1][1]
    sel = [r for r in rows if r[0] >= lo and r[1] <= hi]
    span_ms = (hi - lo) / 1e6
    queues = sorted({r[3] for r in sel})
    # per queue and class: merged intervals
    per = {}
    for q in queues:
        for cls in ("matrix", "hbm"):
            per[(q, cls)] = merged([(s, e) for s, e, n, qq in sel if qq == q and (n.startswith(MATRIX) == (cls == "matrix"))])
    fam = collections.OrderedDict()
    for s, e, n, q in sel:
        if not n.startswith(KEEP):
            n = "(other: ATen / copies)"
        f = fam.setdefault(family(n), dict(launches=0, ms=0.0, beside_matrix=0.0, beside_hbm=0.0, queue=collections.Counter()))
        f["launches"] += 1
        f["ms"] += (e - s) / 1e6
        f["queue"][q] += 1
        om = sum(overlap(*per[(qq, "matrix")], s, e) for qq in queues if qq != q)
        oh = sum(overlap(*per[(qq, "hbm")], s, e) for qq in queues if qq != q)
        f["beside_matrix"] += min(om, e - s) / 1e6
        f["beside_hbm"] += min(oh, e - s) / 1e6
    steps = float(args.steps)
    busy_s, busy_e = merged([(s, e) for s, e, _, _ in sel])
    busy = sum(b - a for a, b in zip(busy_s, busy_e)) / 1e6
    total = sum(f["ms"] for f in fam.values())
    print("%d timed steps, %.2f ms per step under the profiler; kernels %.2f ms per step summed over the queues, %.2f ms of the step "
          "have at least one kernel running (%.2f ms idle between kernels), %.2f ms run two deep"
          % (args.steps, span_ms / steps, total / steps, busy / steps, (span_ms - busy) / steps, (total - busy) / steps))
    print("queues: %s" % ", ".join("%s (%d launches)" % (q, sum(1 for r in sel if r[3] == q)) for q in queues))
    print("%-64s %6s %9s %9s %12s %10s  %s" % ("kernel family (per step)", "calls", "ms", "alone", "beside MFMA", "beside HBM", "class"))
    table = {}
    for n, f in sorted(fam.items(), key=lambda kv: -kv[1]["ms"]):
        alone = max(0.0, f["ms"] - f["beside_matrix"] - f["beside_hbm"])
        cls = "matrix" if n.startswith(MATRIX) else "hbm"
        table[n] = dict(launches_per_step=f["launches"] / steps, ms_per_step=round(f["ms"] / steps, 3), alone_ms=round(alone / steps, 3),
                        beside_matrix_ms=round(f["beside_matrix"] / steps, 3), beside_hbm_ms=round(f["beside_hbm"] / steps, 3), cls=cls)
        if f["ms"] / steps >= 0.05:
            print("%-64s %6.0f %9.2f %9.2f %12.2f %10.2f  %s" % (n[:64], f["launches"] / steps, f["ms"] / steps, alone / steps,
                                                               f["beside_matrix"] / steps, f["beside_hbm"] / steps, cls))
    for cls in ("matrix", "hbm"):
        rows_c = [v for v in table.values() if v["cls"] == cls]
        print("%-64s %6s %9.2f %9.2f %12.2f %10.2f" % ("all %s-bound kernels" % cls, "", sum(v["ms_per_step"] for v in rows_c),
                                                      sum(v["alone_ms"] for v in rows_c), sum(v["beside_matrix_ms"] for v in rows_c),
                                                      sum(v["beside_hbm_ms"] for v in rows_c)))
    if args.json:
        json.dump(dict(steps=args.steps, ms_per_step=span_ms / steps, busy_ms_per_step=busy / steps, families=table), open(args.json, "w"),
                  indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
