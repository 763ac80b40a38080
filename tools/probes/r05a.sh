#!/bin/bash
# round 5, first GPU call: the new parity tests, the cfg5 truth run, a bench line of this box and the BN one-launch A/B
cd "${GRAFT_REPO_ROOT:-.}"
OUT=$PWD/gpurun_out; mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_kernels_gpu.py -x -q -m gpu --durations=8 \
  -k "cfg5_geometry or cfg3_full_batch or wgrad_pingpong_stream_k or statistics_in_one_launch or target_forward_reuse_is_bitwise_the_literal or step_b_forward_fork" \
  > "$OUT/r05a_tests.txt" 2>&1
tail -15 "$OUT/r05a_tests.txt"
timeout 900 python tests/grad_truth_cfg2.py --cfg5 > "$OUT/r05_grad_truth_cfg5.txt" 2>&1
cat "$OUT/r05_grad_truth_cfg5.txt"
Q="--steps 12 --warmup 4 --no_cpu_baseline --other_configs= --literal_steps 0 --strict_steps 0"
for i in 1 2; do
  MCDSEG_BN_STATS_ONE=0 python bench.py $Q 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('two-stage stats', d['ms_per_step'])"
  python bench.py $Q 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('one-launch stats', d['ms_per_step'], d['kernels'].get('bn_stats_finalize'))"
done
python bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/r05a_bench.json" 2> "$OUT/r05a_bench.err"
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r05a_bench.json') if l.startswith('{')][-1])
print(d['ms_per_step'], d['value'], d.get('strict_fp32'), {k:(v.get('ms_per_step'), v.get('roofline',{}).get('frac') if isinstance(v.get('roofline'),dict) else None, v.get('wgrad_stream'), v.get('peak_reserved_gb')) for k,v in (d.get('other_configs') or {}).items()})
print(d['roofline'])
PY
