# validation of the weight-gradient side stream after a change (tests, forced-lag contention run, same-box A/B, the other configs)
python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "deferred or batch_split or large_tile" 2>&1 | tail -n 4
python tools/op_contention.py --procs 2 --iters 40 --shape 16,512,60,80 --dil 4 --side_lag 20000000 2>&1 | grep done
bash tools/probes/ab_overlap.sh
for l in 2 8; do
  MCDSEG_OVERLAP_WGRAD_LAG=$l python bench.py --steps 8 --warmup 3 --no_cpu_baseline --timer_steps 0 --literal_steps 0 2>/dev/null | tail -n 1 > gpurun_out/r03_overlap/line.json
  python -c "import json; d=json.load(open('gpurun_out/r03_overlap/line.json')); print('MAX_LAG $l:', d['ms_per_step'], 'ms/step')"
done
(timeout 900 python tools/bench_configs.py --cfg cfg5 --n5 8 --hw5 480 640 --steps 3
 MCDSEG_ACT_STORAGE=compact timeout 900 python tools/bench_configs.py --cfg cfg5 --n5 32 --hw5 720 1280 --steps 2
 python tools/bench_configs.py --cfg cfg2,cfg3,cfg4 --steps 3) 2>&1 | grep "^cfg\|side stream"
