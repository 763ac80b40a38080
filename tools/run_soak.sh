#!/bin/bash
# On the GPU box (gpurun): the determinism soak of the kernels whose waits are hand-counted around LDS-DMAs the compiler does not see
#  and of round 6's 2-byte chain -- each op in a loop in TWO processes on the one device, every output compared bit
# for bit with iteration 0.  bash tools/run_soak.sh r06  ->  gpurun_out/r06_determinism_soak.log (kept as profiles/r06_determinism_soak.log)
set -u
TAG=${1:-r06}
cd "${GRAFT_REPO_ROOT:-.}"
LOG=$PWD/gpurun_out/${TAG}_determinism_soak.log
mkdir -p gpurun_out
ulimit -c 0
{
  echo "# $(date -u) two processes per op on one MI355X; a line '[pid] ... done: 0 of N iterations differ' per process is a pass"
  for spec in "loss 2000 16,41,60,80 1" "up8_bwd 2000 16,41,60,80 1" "conv 2000 16,256,60,80 2" "conv 1000 16,64,120,160 1" "conv 1000 16,512,60,80 4" "half 1000 8,256,90,160 1"; do
    set -- $spec
    echo "== op $1, $2 iterations, shape $3, dilation $4"
    timeout 1500 python3 tools/op_contention.py --op $1 --procs 2 --iters $2 --shape $3 --dil $4 2>&1 | grep -v "amdgpu.ids"
    echo "== exit code $?"
  done
} > "$LOG" 2>&1
tail -30 "$LOG"
