#!/bin/bash
# development: sizes and durations of the ATen element-wise kernels inside one benchmark step (rocprofv3 kernel trace, summarised on the box)
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; OUT=$PWD/gpurun_out
rocprofv3 --kernel-trace --output-format csv -d "$OUT/aten_trace" -- python3 bench.py --steps 1 --warmup 1 --literal_steps 0 --no_cpu_baseline --other_configs "" > /dev/null 2> "$OUT/aten_trace.err"
python3 - "$OUT" <<'PY'
import csv, glob, collections, sys
f = glob.glob(sys.argv[1] + '/aten_trace/*/*kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
agg = collections.defaultdict(list)
for r in rows:
    n = r['Kernel_Name']
    if 'at::native' not in n: continue
    short = n.split('<')[0].replace('void ', '') + ' ' + ('add' if 'CUDAFunctor_add' in n else ('mul' if 'MulFunctor' in n or 'mul' in n.lower() else ('fill' if 'Fill' in n else ('copy' if 'copy' in n.lower() else 'other'))))
    g = int(r['Grid_Size']) if 'Grid_Size' in r else int(r.get('Grid_Size_X', 0))
    agg[(short, g)].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
tot = 0
for (k, g), v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    tot += sum(v)
    print('%-60s grid %10d  n=%4d  avg %8.1f us  total %8.2f ms' % (k[:60], g, len(v), sum(v) / len(v), sum(v) / 1e3))
print('total ATen ms (2 steps):', tot / 1e3)
PY
rm -rf "$OUT/aten_trace"
