cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; OUT=$PWD/gpurun_out
for pp in 0 2 3; do
  export MCDSEG_PINGPONG=$pp
  for L in "L5 256" "L6 512"; do
    tag=$(echo "r04e_sq_pp${pp}_${L}" | tr ' ' '_')
    rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/$tag" -- python3 tools/bench_layers.py --reps 3 --only "$L" > /dev/null 2> "$OUT/$tag.err"
  done
done
python3 - <<'PY'
import csv, glob, collections, os
for d in sorted(glob.glob('gpurun_out/r04e_sq_pp*')):
    if not os.path.isdir(d): continue
    f = glob.glob(d + '/*/*counter_collection.csv')
    if not f: print(d, 'no csv'); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        k = r['Kernel_Name'].replace('(anonymous namespace)::','').replace('void ','').split('(')[0]
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in agg.items():
        if not k.startswith('conv_gemm'): continue
        tot = {c: sum(x) for c, x in v.items()}
        n = len(v['SQ_WAVE_CYCLES']); cyc = tot['GRBM_GUI_ACTIVE'] / 8.0
        print('%s | %-62s n=%2d cyc/launch=%8d mfma_busy=%.3f parked=%.3f stalled=%.3f issuing=%.3f' % (os.path.basename(d)[8:], k[:62], n, cyc / n,
              tot['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024.0 * cyc), tot['SQ_WAIT_ANY'] / tot['SQ_WAVE_CYCLES'], tot['SQ_WAIT_INST_ANY'] / tot['SQ_WAVE_CYCLES'],
              tot['SQ_ACTIVE_INST_ANY'] / tot['SQ_WAVE_CYCLES']))
PY
find gpurun_out -name "*.csv" -size +2M -delete
