mkdir -p gpurun_out/r03t
for cfg in "1 2048" "1 1800" "0 1024" "0 2048"; do set -- $cfg; for l in "L4 128" "L5 128->256" "L5 256"; do MCDSEG_WGRAD_TWOTAP=$1 MCDSEG_WGRAD_WGS=$2 python tools/bench_layers.py --only "$l" --reps 10 2>&1 | grep "^L" | sed 's/.*pre-split/pre-split/' | sed 's/fprop.*wgrad/wgrad/' | sed "s/^/twotap=$1 wgs=$2 $l  /"; done; done > gpurun_out/r03t/twotap2.log 2>&1
cat gpurun_out/r03t/twotap2.log
