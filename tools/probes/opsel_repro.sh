#!/bin/bash
# DESIGN.md section 5a, reproducer attempt (tools/probes/opsel_repro.hip): the instruction alone, then two PROCESSES at once on the same
# GPU (the condition under which round 2's failure appeared), then the control encoding under the same contention.
#   bash tools/probes/opsel_repro.sh [seconds per leg]
cd "$(dirname "$0")"
S=${1:-20}
[ -x ./opsel_repro ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o opsel_repro opsel_repro.hip || exit 1
./opsel_repro "$S" 0 solo
./opsel_repro "$S" 0 duo-a & ./opsel_repro "$S" 0 duo-b & wait
./opsel_repro "$S" 1 control-a & ./opsel_repro "$S" 1 control-b & wait
# long launches (7 ms: a process switch now falls INSIDE a kernel, wave state is saved and restored)
./opsel_repro "$S" 0 long-a 200 & ./opsel_repro "$S" 0 long-b 200 & wait
