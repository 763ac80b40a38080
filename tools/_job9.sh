ulimit -c 0
export MCDSEG_PRETRAINED=0
python -m pytest tests/test_model_gpu.py -q -k "cfg5 or d105 or full_resolution or batch_split" 2>&1 | tail -8 > gpurun_out/r06k_cfg5_tests.log
python tools/bench_one_config.py cfg5_f16 2 > gpurun_out/r06k_cfg5_f16.json 2> gpurun_out/r06k.err
python tools/bench_one_config.py cfg5 2 --no-roofline > gpurun_out/r06k_cfg5.json 2>> gpurun_out/r06k.err
