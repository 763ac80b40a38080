#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_model_gpu.py -q -m gpu -x -k "conv_chain_stages or forward_small or backward_small or three_step_small or drn_c or model_variants or residual_gradient_fold or fused_up_loss or forward_fork" 2>&1 | tail -4
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "internal_groups or compact or stale or presplit" 2>&1 | tail -3
Q="--steps 12 --warmup 4 --no_cpu_baseline --other_configs= --literal_steps 0 --strict_steps 0"
show='import sys,json; d=json.loads(sys.stdin.read()); k=d["kernels"]; print(sys.argv[1], d["ms_per_step"], {n:(v["launches"],v["ms_total"],v["alg_gbs"]) for n,v in k.items() if n.startswith("bn_apply")})'
for i in 1 2; do
MCDSEG_INTERNAL_STAGES=0 python bench.py $Q 2>/dev/null | python -c "$show" "chains write fp32"
python bench.py $Q 2>/dev/null | python -c "$show" "companions only"
done
