ulimit -c 0
export MCDSEG_PRETRAINED=0
python -m pytest tests -q -m gpu 2>&1 | tail -40 > gpurun_out/r06d_suite.log
bash tools/run_cfg5_traffic.sh r06 > gpurun_out/r06d_traffic.log 2>&1
mkdir -p gpurun_out/grad_truth
python tests/golden/make_grad_truth.py --out gpurun_out/grad_truth cfg2 cfg3 cfg4 cfg5n2 > gpurun_out/r06d_truth_gen.log 2>&1
cp gpurun_out/grad_truth/*.npz tests/golden/
python tests/truth.py cfg2 cfg3 cfg4 cfg5n2 > gpurun_out/r06d_truth_report.txt 2>&1
python tests/golden/make_grad_truth.py --out gpurun_out/grad_truth cfg5n8 >> gpurun_out/r06d_truth_gen.log 2>&1
cp gpurun_out/grad_truth/*.npz tests/golden/
python tests/truth.py cfg5n8 >> gpurun_out/r06d_truth_report.txt 2>&1
