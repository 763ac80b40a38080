#!/bin/bash
# development: where the GPU idles inside a benchmark step -- union of the kernel intervals of a rocprofv3 kernel trace (both streams),
# gaps above 3 us attributed to the kernel that ended before them
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; OUT=$PWD/gpurun_out
rocprofv3 --kernel-trace --output-format csv -d "$OUT/gaps_trace" -- python3 bench.py --steps 3 --warmup 2 --literal_steps 0 --no_cpu_baseline --other_configs "" > "$OUT/gaps_bench.json" 2> "$OUT/gaps_trace.err"
python3 - "$OUT" <<'PY'
import csv, glob, collections, sys, json
out = sys.argv[1]
f = glob.glob(out + '/gaps_trace/*/*kernel_trace.csv')[0]
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]) for r in csv.DictReader(open(f))]
rows.sort()
line = [l for l in open(out + '/gaps_bench.json').read().splitlines() if l.startswith('{')][-1]
print('ms_per_step (under the profiler):', json.loads(line)['ms_per_step'])
# the timed region: 7 optimizer launches per step; 2 warm-up steps, then 3 timed ones
sgd = [r for r in rows if r[2].startswith('sgd_momentum')]
lo, hi = sgd[13][1], sgd[34][1]
sel = [r for r in rows if r[0] >= lo and r[1] <= hi]
busy = 0; cur_end = sel[0][0]; gaps = collections.defaultdict(lambda: [0, 0.0]); last = None; big = []
for s, e, n in sel:
    if s > cur_end:
        g = (s - cur_end) / 1e3
        if g > 3.0:
            gaps[last][0] += 1; gaps[last][1] += g
            if g > 200: big.append((g, last, n))
        cur_end = e; last = n
    elif e > cur_end:
        cur_end = e; last = n
    busy += 0
span = (sel[-1][1] - sel[0][0]) / 1e6
idle = sum(v[1] for v in gaps.values()) / 1e3
print('sample span %.1f ms, idle in gaps > 3 us: %.2f ms (%.1f %%)' % (span, idle, 100 * idle / span))
for k, v in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:14]:
    print('  after %-70s n=%4d  %8.2f ms' % ((k or '?')[:70], v[0], v[1] / 1e3))
for g, a, b in sorted(big, reverse=True)[:8]:
    print('  gap %.0f us between %s and %s' % (g, a[:50], b[:50]))
PY
rm -rf "$OUT/gaps_trace"
