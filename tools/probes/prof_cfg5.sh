#!/bin/bash
# development: kernel statistics of BASELINE config 5 at its stated size (drn_d_105, N = 32, 720 x 1280, compact storage)
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; OUT=$PWD/gpurun_out
export MCDSEG_ACT_STORAGE=compact
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/cfg5_stats" -- python3 tools/bench_configs.py --cfg cfg5 --n5 32 --hw5 720 1280 --steps 2 > "$OUT/cfg5_bench.txt" 2> "$OUT/cfg5_stats.err"
cat "$OUT/cfg5_bench.txt" | tail -3
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/cfg5_stats/*/*kernel_stats.csv')[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('total kernel ms', tot / 1e6)
for r in rows[:28]:
    n = r['Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
    print('%-78s %6s %9.1f ms %5.1f%% avg %8.1f us' % (n[:78], r['Calls'], float(r['TotalDurationNs']) / 1e6, float(r['Percentage']), float(r['AverageNs']) / 1e3))
PY
find "$OUT/cfg5_stats" -name "*kernel_trace.csv" -delete
