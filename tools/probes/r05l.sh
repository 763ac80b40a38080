#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "relu_bit_plane or mask_from_z or internal_groups or conv_bn_act or residual or compact" 2>&1 | tail -5
timeout 600 python -m pytest tests/test_model_gpu.py -q -m gpu -x -k "three_step_small or backward_small or residual_gradient_fold or forward_fork" 2>&1 | tail -3
Q="--steps 12 --warmup 4 --no_cpu_baseline --other_configs= --literal_steps 0 --strict_steps 0"
show='import sys,json; d=json.loads(sys.stdin.read()); k=d["kernels"]; print(sys.argv[1], d["ms_per_step"], {n:(v["launches"],v["ms_total"],v["alg_gbs"]) for n,v in k.items() if n.startswith("bn_")})'
for i in 1 2; do
MCDSEG_RELU_MASK=0 python bench.py $Q 2>/dev/null | python -c "$show" "fp32 y mask"
python bench.py $Q 2>/dev/null | python -c "$show" "bit-plane"
done
