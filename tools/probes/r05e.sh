#!/bin/bash
# round 5: the side stream on a subset of the CUs (hipExtStreamCreateWithCUMask); the main stream a stream of its own (a masked stream
# is a blocking stream: it would synchronise with the legacy default stream launch by launch)
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
Q="--steps 12 --warmup 4 --no_cpu_baseline --other_configs= --literal_steps 0 --strict_steps 0"
show='import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], d["ms_per_step"], d["wgrad_stream"])'
python bench.py $Q 2>/dev/null | python -c "$show" "default (legacy default stream)"
MCDSEG_BENCH_MAIN_STREAM=1 python bench.py $Q 2>/dev/null | python -c "$show" "main stream of its own"
for n in 256 240 224 192 160; do
  MCDSEG_BENCH_MAIN_STREAM=1 MCDSEG_SIDE_CUS=$n timeout 300 python bench.py $Q 2>/tmp/err.txt | python -c "$show" "side stream on $n CUs" || tail -3 /tmp/err.txt
done
MCDSEG_BENCH_MAIN_STREAM=1 python bench.py $Q 2>/dev/null | python -c "$show" "main stream of its own"
