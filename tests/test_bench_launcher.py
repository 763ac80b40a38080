"""CPU: ``python bench.py --gpus 2`` without a launcher starts its two ranks itself (child processes through
torch.distributed.run, rendezvous on 127.0.0.1) and rank 0 prints the one JSON line with n_gpus = 2.  Runs the dry mode
(MCDSEG_BENCH_DRY=1: gloo, a stand-in step, the real barriers / MAX-over-ranks protocol) -- no GPU work."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(argv, extra_env):
    env = dict(os.environ, MCDSEG_BENCH_DRY="1", OMP_NUM_THREADS="1", **extra_env)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, env=env, capture_output=True, text=True, timeout=600)


def test_bench_spawns_its_ranks():
    r = _run(["--gpus", "2", "--steps", "3", "--warmup", "0"], {})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 3 and rec["scaling"] == "weak"
    # MAX over ranks: rank 1's stand-in step is twice as long as rank 0's
    assert rec["ms_per_step"] >= 19.0


def test_bench_rejects_world_size_mismatch():
    env = dict(os.environ, MCDSEG_BENCH_DRY="1", WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)
