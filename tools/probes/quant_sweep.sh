mkdir -p gpurun_out/r03_quant
for H in 40 48 51 52 56 60 64 70 76 77 80 90 102 103 128; do
  echo "H=$H" ; python tools/bench_layers.py --only "L6 512->512" --hw $H 80 --reps 20 2>/dev/null | grep "L6 512"
done > gpurun_out/r03_quant/sweep_h.txt 2>&1
cat gpurun_out/r03_quant/sweep_h.txt | cut -c1-60,88-
