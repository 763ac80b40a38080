ulimit -c 0
export MCDSEG_PRETRAINED=0
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06i_smoke.log 2>&1
python -m pytest tests/test_trainers_gpu.py -q -k "rccl or two_ranks" 2>&1 | tail -6 > gpurun_out/r06i_rccl.log
