#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "row_of_taps or wgrad_presplit_operands or wgrad_pingpong or two_taps or wgrad_large" 2>&1 | tail -8
python tools/bench_layers.py --only "L4 128->128" 2>&1 | grep -v amdgpu | tail -2
MCDSEG_WGRAD_PP3=0 python tools/bench_layers.py --only "L4 128->128" 2>&1 | grep -v amdgpu | tail -2
python tools/bench_layers.py --only "L5 128->256" 2>&1 | grep -v amdgpu | tail -2
MCDSEG_WGRAD_PP3=0 python tools/bench_layers.py --only "L5 128->256" 2>&1 | grep -v amdgpu | tail -2
Q="--steps 12 --warmup 4 --no_cpu_baseline --other_configs= --literal_steps 0 --strict_steps 0"
show='import sys,json; d=json.loads(sys.stdin.read()); k=d["kernels"]; print(sys.argv[1], d["ms_per_step"], {n.replace("conv_wgrad_",""):(v["launches"],v["avg_ms"],v["tflops"]) for n,v in k.items() if "wgrad" in n and ("pp3" in n or "tr_kernel" in n)})'
for i in 1 2; do
MCDSEG_WGRAD_PP3=0 python bench.py $Q 2>/dev/null | python -c "$show" "tr kernel"
python bench.py $Q 2>/dev/null | python -c "$show" "row of taps"
done
