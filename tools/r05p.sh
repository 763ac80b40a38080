#!/bin/bash
# round 5, GPU call p: hidden DMA in the 64-wide weight gradient, the stem's tile loads issued together -- tests, then a same-box A/B against
# libmcdseg_prev.so (the kernels of commit 0a1c5a8^: before the loss / up-sampler / BatchNorm / tr64 / stem changes)
cd "$(dirname "$0")/.."
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "wgrad or stem or conv" 2>&1 | grep -E "passed|failed|Error" | head -5
B="python bench.py --steps 12 --warmup 4 --no_cpu_baseline --other_configs= --literal_steps 0 --strict_steps 0"
for i in 1 2; do
  for lib in prev new; do
    if [ $lib = prev ]; then export MCDSEG_LIB=$PWD/multichannel-semseg-with-uda_amd/mcdseg/libmcdseg_prev.so; else unset MCDSEG_LIB; fi
    $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels']; print('$lib', d['ms_per_step'], {n[:34]: round(v['avg_ms'],4) for n,v in k.items() if n.startswith(('conv_stem','conv_wgrad_split_tr64','up8','bn_'))})"
  done
done
