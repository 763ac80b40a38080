#!/bin/bash
# development: what each piece of round 4 buys, on ONE box -- the default build against the same build with one knob turned off
cd "${GRAFT_REPO_ROOT:-.}"
run() {  # label, env assignments...
  label=$1; shift
  ms=$(env "$@" python3 bench.py --steps 12 --warmup 4 --no_cpu_baseline --other_configs "" --literal_steps 0 2>/dev/null | python3 -c "import json,sys; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print(d['ms_per_step'])")
  printf "%-74s %s ms\n" "$label" "$ms"
}
run "default" MCDSEG_DUMMY=0
run "default (again)" MCDSEG_DUMMY=0
run "no ping-pong conv kernels (4-wave tiles: MCDSEG_PINGPONG=0)" MCDSEG_PINGPONG=0
run "no 320 / 160-pixel tiles (two-launch plan only: MCDSEG_PP_WIDE_FILL=101)" MCDSEG_PP_WIDE_FILL=101
run "no 128 x 320 tile (MCDSEG_PP_WIDE128=0)" MCDSEG_PP_WIDE128=0
run "weight gradient: stream-K instead of the slab plan (MCDSEG_WGRAD_PP=1)" MCDSEG_WGRAD_PP=1
run "weight gradient: 4-wave kernels (MCDSEG_WGRAD_PP=0)" MCDSEG_WGRAD_PP=0
run "residual adds by autograd (MCDSEG_FUSE_RES_ADD=0)" MCDSEG_FUSE_RES_ADD=0
run "addend loaded value by value (MCDSEG_DGRAD_ADD_LDS=0)" MCDSEG_DGRAD_ADD_LDS=0
run "step B's passes one after the other (MCDSEG_OVERLAP_STEPB=0)" MCDSEG_OVERLAP_STEPB=0
run "BatchNorm apply kernels front to back (MCDSEG_BN_REVERSE=0)" MCDSEG_BN_REVERSE=0
run "stride-2 data gradients class by class (MCDSEG_DGRAD_INTERLEAVE=0)" MCDSEG_DGRAD_INTERLEAVE=0
run "one stream (MCDSEG_OVERLAP_WGRAD=0 MCDSEG_OVERLAP_STEPB=0)" MCDSEG_OVERLAP_WGRAD=0 MCDSEG_OVERLAP_STEPB=0
run "all of the above off" MCDSEG_PINGPONG=0 MCDSEG_WGRAD_PP=0 MCDSEG_FUSE_RES_ADD=0 MCDSEG_OVERLAP_STEPB=0 MCDSEG_BN_REVERSE=0 MCDSEG_DGRAD_INTERLEAVE=0
run "default (last)" MCDSEG_DUMMY=0
