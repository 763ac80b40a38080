#!/bin/bash
# development: SQ counters and durations of the convolution kernels of one layer (tools/bench_layers.py) under rocprofv3
#   bash tools/probes/prof_layers.sh "L6 512" [tag]
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; OUT=$PWD/gpurun_out
L=${1:-L6 512}; TAG=${2:-prof}
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${TAG}_trace" -- python3 tools/bench_layers.py --reps 5 --only "$L" > /dev/null 2> "$OUT/${TAG}_trace.err"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/${TAG}_sq" -- python3 tools/bench_layers.py --reps 3 --only "$L" > /dev/null 2> "$OUT/${TAG}_sq.err"
python3 - "$OUT" "$TAG" <<'PY'
import csv, glob, collections, os, sys
out, tag = sys.argv[1], sys.argv[2]
f = glob.glob(out + '/' + tag + '_trace/*/*kernel_stats.csv')
if f:
    for r in csv.DictReader(open(f[0])):
        n = r['Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
        if n.startswith(('conv_', 'wgrad_')): print('%-70s calls %4s avg %9.1f us' % (n[:70], r['Calls'], float(r['AverageNs']) / 1e3))
f = glob.glob(out + '/' + tag + '_sq/*/*counter_collection.csv')
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    k = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
    agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in agg.items():
    if not k.startswith(('conv_gemm', 'conv_wgrad')) or 'GRBM_GUI_ACTIVE' not in v: continue
    tot = {c: sum(x) for c, x in v.items()}
    n = len(v['SQ_WAVE_CYCLES']); cyc = tot['GRBM_GUI_ACTIVE'] / 8.0
    print('%-70s n=%2d cyc/launch=%8d mfma_busy=%.3f parked=%.3f stalled=%.3f issuing=%.3f' % (k[:70], n, cyc / n, tot['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024.0 * cyc),
          tot['SQ_WAIT_ANY'] / tot['SQ_WAVE_CYCLES'], tot['SQ_WAIT_INST_ANY'] / tot['SQ_WAVE_CYCLES'], tot['SQ_ACTIVE_INST_ANY'] / tot['SQ_WAVE_CYCLES']))
PY
find "$OUT/${TAG}_trace" "$OUT/${TAG}_sq" -name "*.csv" -size +2M -delete
