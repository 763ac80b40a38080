#!/bin/bash
# round 5, third GPU call: the 64 x 640 ping-pong tile (tests + A/B), cfg5 geometry test
cd "${GRAFT_REPO_ROOT:-.}"
OUT=$PWD/gpurun_out; mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_kernels_gpu.py -q -m gpu --durations=5 \
  -k "cfg5_geometry or cfg2_full_batch or pingpong_wide_tile_by_default or conv_pingpong_tile or residual_gradient_fold" \
  > "$OUT/r05c_tests.txt" 2>&1
tail -12 "$OUT/r05c_tests.txt"; grep "cfg5 geometry N=" "$OUT/r05c_tests.txt"
python tools/bench_layers.py --only "L3 64->64" 2>&1 | grep -v amdgpu | tail -3
MCDSEG_PP_WIDE64=0 python tools/bench_layers.py --only "L3 64->64" 2>&1 | grep -v amdgpu | tail -3
python tools/bench_layers.py --only "L3 32->64" 2>&1 | grep -v amdgpu | tail -3
Q="--steps 12 --warmup 4 --no_cpu_baseline --other_configs= --literal_steps 0 --strict_steps 0"
for i in 1 2; do
  MCDSEG_PP_WIDE64=0 python bench.py $Q 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('4-wave 64-row tiles', d['ms_per_step'])"
  python bench.py $Q 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernels']; print('64 x 640 tile', d['ms_per_step']); print({n:(v['launches'],v['avg_ms'],v['tflops']) for n,v in k.items() if '1, 5, 1, 4' in n or '2, 2, 1, 4' in n})"
done
