#!/bin/bash
# round 5, GPU call o: the LDS-DMA loss kernel, hidden-DMA band kernels, load-first BatchNorm kernels -- tests, then a same-box A/B against the
# previous commit's kernels (libmcdseg_prev.so)
cd "$(dirname "$0")/.."
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x 2>&1 | grep -E "passed|failed|Error" | head -5
B="python bench.py --steps 12 --warmup 4 --no_cpu_baseline --other_configs= --literal_steps 0 --strict_steps 0"
for i in 1 2; do
  for lib in prev new; do
    if [ $lib = prev ]; then export MCDSEG_LIB=$PWD/multichannel-semseg-with-uda_amd/mcdseg/libmcdseg_prev.so; else unset MCDSEG_LIB; fi
    $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels']; print('$lib', d['ms_per_step'], {n[:28]: round(v['ms_total']/max(1,d['steps']),2) for n,v in k.items() if n.startswith(('bn_','up8'))})"
  done
done
