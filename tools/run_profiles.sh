#!/bin/bash
# On the GPU box (gpurun): the round's measurements -- rocprofv3 kernel stats, HBM-traffic and SQ counter passes (separate runs, counters
# never combined with trace domains), and the benchmark lines of the default arithmetic, of the 24-bit alternatives and of the
# reduced-precision mode.  Usage: bash tools/run_profiles.sh r05z [quick]    (raw output under gpurun_out/, summaries via
# tools/summarize_profiles.py <tag> afterwards)
set -u
TAG=${1:-r05ze}
cd "${GRAFT_REPO_ROOT:-.}"
OUT=$PWD/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
# the fingerprint of the kernel sources these counters are collected with (bench.py quotes a PMC table only for the build it was made from)
python3 -c "import sys; sys.path.insert(0, 'multichannel-semseg-with-uda_amd'); from mcdseg import _lib; print(_lib.source_fingerprint())" > "$OUT/${TAG}_fingerprint.txt"
python3 bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/${TAG}_bench.json" 2> "$OUT/${TAG}_bench.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${TAG}_stats" -- python3 bench.py --steps 3 --warmup 1 --no_cpu_baseline --other_configs "" --literal_steps 0 --strict_steps 0 > "$OUT/${TAG}_bench_profiled.json" 2> "$OUT/${TAG}_stats.err"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/${TAG}_fetch" -- python3 bench.py --steps 1 --warmup 0 --no_cpu_baseline --literal_steps 0 --strict_steps 0 --other_configs "" > /dev/null 2> "$OUT/${TAG}_fetch.err"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/${TAG}_write" -- python3 bench.py --steps 1 --warmup 0 --no_cpu_baseline --literal_steps 0 --strict_steps 0 --other_configs "" > /dev/null 2> "$OUT/${TAG}_write.err"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/${TAG}_sq" -- python3 bench.py --steps 1 --warmup 0 --no_cpu_baseline --literal_steps 0 --strict_steps 0 --other_configs "" > /dev/null 2> "$OUT/${TAG}_sq.err"
if [ "${2:-}" != "quick" ]; then
  MCDSEG_CONV_MATH=bf16x6 python3 bench.py --gpus 1 --steps 10 --warmup 3 --no_cpu_baseline --strict_steps 0 --other_configs "" > "$OUT/${TAG}_bench_bf16x6.json" 2> /dev/null
  MCDSEG_CONV_MATH=f32 python3 bench.py --gpus 1 --steps 4 --warmup 1 --no_cpu_baseline --strict_steps 0 --other_configs "" > "$OUT/${TAG}_bench_f32mfma.json" 2> /dev/null
  python3 bench.py --gpus 1 --steps 10 --warmup 3 --no_cpu_baseline --strict_steps 0 --other_configs "" --dtype f16 > "$OUT/${TAG}_bench_f16x1.json" 2> /dev/null
fi
# keep what travels back small: the per-dispatch traces are large
find "$OUT/${TAG}_stats" -name "*kernel_trace.csv" -size +8M -delete 2>/dev/null
ls -la "$OUT" | grep "${TAG}" | head -20
