ulimit -c 0
export MCDSEG_PRETRAINED=0
python -m pytest tests -q -m gpu 2>&1 | tail -8 > gpurun_out/r06j_suite.log
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06z_bench.json 2> gpurun_out/r06z_bench.err
python bench.py --gpus 1 --steps 10 --warmup 3 --no_cpu_baseline --strict_steps 0 --other_configs "" --dtype f16 > gpurun_out/r06z_bench_f16x1.json 2>> gpurun_out/r06z_bench.err
