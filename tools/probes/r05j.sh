#!/bin/bash
# timing-only: the 4-wave forward kernel of the 64-row layers without its matrix instructions (8) / with nothing but them (1+2+4+16+32 = 55)
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
D=$PWD/multichannel-semseg-with-uda_amd/mcdseg
for L in "L3 64->64"; do
  python tools/bench_layers.py --only "$L" --reps 20 2>&1 | grep "pre-split" | sed 's/.*pre-split/shipped   '"$L"'/'
  MCDSEG_LIB=$D/libmcdseg_nomfma.so python tools/bench_layers.py --only "$L" --reps 20 2>&1 | grep "pre-split" | sed 's/.*pre-split/no MFMA   '"$L"'/'
  MCDSEG_LIB=$D/libmcdseg_onlymfma.so python tools/bench_layers.py --only "$L" --reps 20 2>&1 | grep "pre-split" | sed 's/.*pre-split/MFMA only '"$L"'/'
done
