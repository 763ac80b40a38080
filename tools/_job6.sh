ulimit -c 0
export MCDSEG_PRETRAINED=0
python -m pytest tests/test_kernels_gpu.py -q -x -k "bn or relu or mask or group or conv_bn" 2>&1 | tail -6 > gpurun_out/r06h_bn_tests.log
python bench.py --gpus 1 --steps 10 --warmup 3 --no_cpu_baseline --other_configs "" --strict_steps 0 > gpurun_out/r06h_bench.json 2> gpurun_out/r06h_bench.err
MCDSEG_BN_REDUCE_V4=0 python bench.py --gpus 1 --steps 10 --warmup 3 --no_cpu_baseline --other_configs "" --strict_steps 0 > gpurun_out/r06h_bench_v4off.json 2>> gpurun_out/r06h_bench.err
python -m pytest tests -q -m gpu 2>&1 | tail -8 > gpurun_out/r06h_suite.log
