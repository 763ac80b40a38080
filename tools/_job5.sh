ulimit -c 0
export MCDSEG_PRETRAINED=0
python -m pytest tests/test_kernels_gpu.py -q -x -k "bn or relu or mask or loss" 2>&1 | tail -3 > gpurun_out/r06g_bn_tests.log
bash tools/run_profiles.sh r06 > gpurun_out/r06g_profiles.log 2>&1
