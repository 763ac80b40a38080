#!/bin/bash
# timing-only: what would a staged pixel window (B operand moved once per channel chunk instead of once per tap) be worth at most?
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
V=$PWD/multichannel-semseg-with-uda_amd/mcdseg/libmcdseg_win.so
for L in "L3 64->64" "L4 128->128" "L5 256->256" "L6 512->512"; do
  python tools/bench_layers.py --only "$L" --reps 20 2>&1 | grep "pre-split" | sed 's/.*pre-split/shipped  '"$L"'/'
  MCDSEG_LIB=$V python tools/bench_layers.py --only "$L" --reps 20 2>&1 | grep "pre-split" | sed 's/.*pre-split/ablated  '"$L"'/'
done
