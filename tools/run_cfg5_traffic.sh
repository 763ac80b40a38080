#!/bin/bash
# On the GPU box (gpurun): HBM traffic of BASELINE config 5's network (drn_d_105, 6 x 720 x 1280, N = 8 -- the per-launch bytes scale with N)
# in the one-term arithmetic with round 5's storage (two fp16 pieces, fp32 z and gradients: MCDSEG_HALF_STORAGE=0) and with the 2-byte
# activation storage of round 6.  Separate --pmc passes (FETCH_SIZE, WRITE_SIZE), counters never combined with trace domains.
#   bash tools/run_cfg5_traffic.sh r06     ->  gpurun_out/r06_cfg5_{half,full}_{fetch,write}; summary: tools/summarize_cfg5_traffic.py r06
set -u
TAG=${1:-r06}
cd "${GRAFT_REPO_ROOT:-.}"
OUT=$PWD/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp MCDSEG_PRETRAINED=0 MCDSEG_CONV_MATH=f16x1 MCDSEG_ACT_STORAGE=compact
ulimit -c 0
for mode in half full; do
  if [ $mode = full ]; then export MCDSEG_HALF_STORAGE=0; else export MCDSEG_HALF_STORAGE=1; fi
  for ctr in FETCH_SIZE WRITE_SIZE; do
    d="$OUT/${TAG}_cfg5_${mode}_$(echo $ctr | tr A-Z a-z | cut -d_ -f1)"
    rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d "$d" -- python3 tools/bench_configs.py --cfg cfg5 --n5 8 --hw5 720 1280 --steps 0 > "$d.log" 2>&1
    find "$d" -name "*kernel_trace.csv" -delete 2>/dev/null
  done
done
# SQ counters of the 2-byte chain's step (matrix-pipe busy fraction of the one-term kernels, wave-cycle breakdown of the BatchNorm kernels)
export MCDSEG_HALF_STORAGE=1
d="$OUT/${TAG}_cfg5_half_sq"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$d" -- python3 tools/bench_configs.py --cfg cfg5 --n5 8 --hw5 720 1280 --steps 0 > "$d.log" 2>&1
find "$d" -name "*kernel_trace.csv" -delete 2>/dev/null
ls -la "$OUT" | grep "${TAG}_cfg5" | head
