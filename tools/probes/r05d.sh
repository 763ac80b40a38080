#!/bin/bash
# round 5, fourth GPU call: slab reduce with 16 loads in flight; split counts of the 128 / 64-channel weight gradients
cd "${GRAFT_REPO_ROOT:-.}"
OUT=$PWD/gpurun_out; mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "wgrad_presplit_operands or conv_fprop_dgrad_wgrad or wgrad_large_tile or two_taps" 2>&1 | tail -3
Q="--steps 12 --warmup 4 --no_cpu_baseline --other_configs= --literal_steps 0 --strict_steps 0"
show='import sys,json; d=json.loads(sys.stdin.read()); k=d["kernels"]; print(sys.argv[1], d["ms_per_step"], d["one_stream"], {n.replace("conv_wgrad_",""):(v["launches"],v["avg_ms"]) for n,v in k.items() if "wgrad" in n and "pp" not in n})'
for w in 1024 768 512 384; do
  MCDSEG_WGRAD_WGS=$w python bench.py $Q 2>/dev/null | python -c "$show" "WGS=$w"
done
python bench.py $Q 2>/dev/null | python -c "$show" "default again"
