#!/bin/bash
# round 5, second GPU call: pending parity tests, Winograd probe, the two streams' overlap table, BN one-launch / tr64 A/B, cfg5 deferral
cd "${GRAFT_REPO_ROOT:-.}"
OUT=$PWD/gpurun_out; mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 1800 python -m pytest tests/test_model_gpu.py tests/test_kernels_gpu.py -q -m gpu --durations=8 \
  -k "cfg5_geometry or cfg3_full_batch or cfg2_full_batch or wgrad_pingpong_stream_k or statistics_in_one_launch or target_forward_reuse_is_bitwise_the_literal or wgrad_presplit_operands or conv_fprop_dgrad_wgrad" \
  > "$OUT/r05b_tests.txt" 2>&1
tail -25 "$OUT/r05b_tests.txt"
python tools/probes/winograd_probe.py 2>&1 | grep -v amdgpu.ids | tee "$OUT/r05b_winograd.txt"
Q="--steps 12 --warmup 4 --no_cpu_baseline --other_configs= --literal_steps 0 --strict_steps 0"
for i in 1 2; do
  MCDSEG_BN_STATS_ONE=0 python bench.py $Q 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('two-stage stats', d['ms_per_step'], d['kernels'].get('bn_stats_finalize'))"
  python bench.py $Q 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernels']; print('one-launch stats', d['ms_per_step'], k.get('bn_stats_finalize')); print({n:(v['launches'],v['avg_ms']) for n,v in k.items() if 'wgrad' in n})"
done
rocprofv3 --kernel-trace --output-format csv -d "$OUT/r05b_trace" -- python3 bench.py --steps 3 --warmup 2 --timer_steps 0 --literal_steps 0 --strict_steps 0 --no_cpu_baseline --other_configs "" > "$OUT/r05b_trace_bench.json" 2> "$OUT/r05b_trace.err"
python tools/overlap_table.py "$OUT/r05b_trace" --warmup 2 --steps 3 --json "$OUT/r05b_overlap.json" | tee "$OUT/r05b_overlap.txt"
rm -rf "$OUT/r05b_trace"
MCDSEG_ACT_STORAGE=compact python tools/bench_configs.py --cfg cfg5 --n5 32 --hw5 720 1280 --steps 2 2>&1 | grep -v amdgpu.ids | tee "$OUT/r05b_cfg5_default.txt"
MCDSEG_OVERLAP_WGRAD_MEM=1.0 MCDSEG_OVERLAP_WGRAD_RESERVED=1.0 MCDSEG_ACT_STORAGE=compact python tools/bench_configs.py --cfg cfg5 --n5 32 --hw5 720 1280 --steps 2 2>&1 | grep -v amdgpu.ids | tee "$OUT/r05b_cfg5_forced.txt"
