#!/bin/bash
# Development tool: libmcdseg_prev.so = the kernels of another commit, for a same-box A/B in ONE gpurun call (select with MCDSEG_LIB, as
# tools/r05o.sh / r05p.sh do).  Only the sources that differ from the working tree are compiled from that commit (with the flags of
# mcdseg/_lib.py: the files of NO_PACKED_F32 without packed-fp32 instructions); every other object is the current build's.
#   bash tools/build_prev_lib.sh <commit> bn loss up8        (run `python -c "import __graft_entry__ as g; g.build()"` first)
set -eu
cd "$(dirname "$0")/.."
REV=$1; shift
P=multichannel-semseg-with-uda_amd/csrc
T=$(mktemp -d)
for f in $(git ls-tree -r --name-only "$REV" $P | grep -E "\.(h|hip)$"); do git show "$REV:$f" > "$T/$(basename "$f")"; done
SKIP=""
for f in "$@"; do
  NOPK=""
  case "$f" in bn|loss|multitask|fusion|io|sgd|up8) NOPK="-Xclang -target-feature -Xclang -packed-fp32-ops";; esac
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -fvisibility=hidden $NOPK -I include -I "$T" -c "$T/$f.hip" -o "$T/$f.o" 2> /dev/null &
  SKIP="$SKIP|/$f.o"
done
wait
OBJS=$(ls $P/build/*.o | grep -Ev "${SKIP#|}")
NEW=$(for f in "$@"; do echo "$T/$f.o"; done)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -fvisibility=hidden -o multichannel-semseg-with-uda_amd/mcdseg/libmcdseg_prev.so $OBJS $NEW
ls -la multichannel-semseg-with-uda_amd/mcdseg/libmcdseg_prev.so
