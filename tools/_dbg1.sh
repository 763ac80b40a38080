ulimit -c 0
free -g | head -2
python -m pytest tests/test_half_storage_gpu.py -x -q -s 2>&1 | tail -15
python -m pytest tests/test_kernels_gpu.py -x -q -k "pingpong or f16x1" 2>&1 | tail -5
export MCDSEG_PRETRAINED=0
MCDSEG_CONV_MATH=f16x1 MCDSEG_ACT_STORAGE=compact timeout 600 python tools/bench_configs.py --cfg cfg5 --n5 8 --hw5 720 1280 --steps 1 2>&1 | tail -3
timeout 900 python tools/bench_one_config.py cfg5_f16 2 > gpurun_out/r06c_cfg5_f16.json 2> gpurun_out/r06c_cfg5_f16.err
tail -3 gpurun_out/r06c_cfg5_f16.err
MCDSEG_PP_DEEP=0 timeout 900 python tools/bench_one_config.py cfg5_f16 2 --no-roofline > gpurun_out/r06c_cfg5_f16_nodeep.json 2>> gpurun_out/r06c_cfg5_f16.err
