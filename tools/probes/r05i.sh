#!/bin/bash
# timing-only: a barrier + DMA wait every second K-step in the 4-wave kernels (MCD_ABLATE=128), and with the pixel operand moved once per chunk on top (192)
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
D=$PWD/multichannel-semseg-with-uda_amd/mcdseg
for L in "L3 64->64" "L3 32->64" "L4 64->128"; do
  python tools/bench_layers.py --only "$L" --reps 20 2>&1 | grep "pre-split" | sed 's/.*pre-split/shipped        '"$L"'/'
  MCDSEG_LIB=$D/libmcdseg_b2.so python tools/bench_layers.py --only "$L" --reps 20 2>&1 | grep "pre-split" | sed 's/.*pre-split/half barriers '"$L"'/'
  MCDSEG_LIB=$D/libmcdseg_b2w.so python tools/bench_layers.py --only "$L" --reps 20 2>&1 | grep "pre-split" | sed 's/.*pre-split/both          '"$L"'/'
done
