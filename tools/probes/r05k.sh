#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
D=$PWD/multichannel-semseg-with-uda_amd/mcdseg
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "conv_fprop_dgrad_wgrad or presplit or conv_split_accuracy or large_tile or pingpong_tile or nonfinite or scaling or stale" 2>&1 | tail -3
for L in "L3 64->64" "L3 32->64" "L4 64->128"; do
  MCDSEG_LIB=$D/libmcdseg_old64.so python tools/bench_layers.py --only "$L" --reps 20 2>&1 | grep "pre-split" | sed 's/.*pre-split/register staging '"$L"'/'
  python tools/bench_layers.py --only "$L" --reps 20 2>&1 | grep "pre-split" | sed 's/.*pre-split/all-DMA          '"$L"'/'
done
Q="--steps 12 --warmup 4 --no_cpu_baseline --other_configs= --literal_steps 0 --strict_steps 0"
show='import sys,json; d=json.loads(sys.stdin.read()); k=d["kernels"]; print(sys.argv[1], d["ms_per_step"], {n[-26:]:(v["launches"],v["avg_ms"]) for n,v in k.items() if "2, 2, 1, 4" in n})'
for i in 1 2; do
MCDSEG_LIB=$D/libmcdseg_old64.so python bench.py $Q 2>/dev/null | python -c "$show" "register staging"
python bench.py $Q 2>/dev/null | python -c "$show" "all-DMA"
done
