# same-box A/B of the weight-gradient side stream: no event timers, no literal-schedule pass
mkdir -p gpurun_out/r03_overlap
run() {
  env "$@" python bench.py --steps 8 --warmup 3 --no_cpu_baseline --timer_steps 0 --literal_steps 0 2>gpurun_out/r03_overlap/err.txt | tail -n 1 > gpurun_out/r03_overlap/line.json
  python - "$*" <<'P'
import json,sys
try:
    d=json.load(open("gpurun_out/r03_overlap/line.json"))
    print("%-60s %.2f ms/step  %.2f img/s" % (sys.argv[1], d["ms_per_step"], d["value"]))
except Exception as e:
    print(sys.argv[1], "FAILED", e); print(open("gpurun_out/r03_overlap/err.txt").read()[-1500:])
P
}
for m in 0 2 1 0 2; do run MCDSEG_OVERLAP_WGRAD=$m; done
