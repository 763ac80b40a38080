#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_trainers_gpu.py -q -m gpu -x -k "two_ranks" 2>&1 | tail -3
Q="--steps 12 --warmup 4 --no_cpu_baseline --other_configs= --literal_steps 0 --strict_steps 0"
show='import sys,json; d=json.loads(sys.stdin.read()); k=d["kernels"]; print(sys.argv[1], d["ms_per_step"], k.get("pack_weights_multi_kernel"))'
for b in 96 192 384 768; do
  MCDSEG_PACK_BLOCKS=$b python bench.py $Q 2>/dev/null | python -c "$show" "pack blocks $b"
done
