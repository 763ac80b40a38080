#!/bin/bash
# On the GPU box (gpurun): what the streams of a default benchmark step overlap, and how long the chip is idle between kernels --
# rocprofv3 --kernel-trace of five untimed-by-events steps, reduced by tools/overlap_table.py.   bash tools/run_overlap.sh r06
set -u
TAG=${1:-r06}
cd "${GRAFT_REPO_ROOT:-.}"
OUT=$PWD/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp MCDSEG_PRETRAINED=0
ulimit -c 0
rocprofv3 --kernel-trace --output-format csv -d "$OUT/${TAG}_trace" -- python3 bench.py --steps 3 --warmup 2 --no_cpu_baseline --other_configs "" --literal_steps 0 --strict_steps 0 --timer none > "$OUT/${TAG}_trace_bench.json" 2> "$OUT/${TAG}_trace.err"
python3 tools/overlap_table.py "$OUT/${TAG}_trace" --warmup 2 --steps 3 > "$OUT/${TAG}_overlap.txt" 2>> "$OUT/${TAG}_trace.err"
find "$OUT/${TAG}_trace" -name "*kernel_trace.csv" -delete 2>/dev/null
head -3 "$OUT/${TAG}_overlap.txt"
