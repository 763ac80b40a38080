ulimit -c 0
export MCDSEG_PRETRAINED=0
python -m pytest tests -q -m gpu 2>&1 | tail -30 > gpurun_out/r06e_suite.log
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06e_bench.json 2> gpurun_out/r06e_bench.err
python tools/host_profile.py cfg2 > gpurun_out/r06e_host_cfg2.txt 2>&1
python tools/host_profile.py cfg4 > gpurun_out/r06e_host_cfg4.txt 2>&1
bash tools/run_soak.sh r06 > /dev/null 2>&1
