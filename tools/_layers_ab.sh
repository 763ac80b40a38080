mkdir -p gpurun_out/r03p
for m in f16x1 f16x3; do for l in "L5 256" "L6 512"; do python tools/bench_layers.py --math $m --only "$l" --reps 10 2>&1 | grep "^L" | sed 's/.*pre-split/pre-split/' | sed "s/^/$m $l  /"; done; done > gpurun_out/r03p/wg_pipe2.log 2>&1
cat gpurun_out/r03p/wg_pipe2.log
