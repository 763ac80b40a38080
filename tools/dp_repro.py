"""Determinism matrix (kernel development): runs tools/dp_dump_worker.py in several process configurations on ONE GPU and
diffs the full tensors against the first solo run.  This process never touches the GPU (it only spawns children).

  python tools/dp_repro.py [--mode step|grads] [--reps 3] [--cases solo,world2,stress] [--env K=V,K=V] [--tag name]

cases: solo = world 1, alone on the device; world2 = two ranks (gloo) on the device; stress = world 1 while a child process streams
large device copies (tools/gpu_stress.py); stress2 = world 2 + the copy loop."""
import argparse
import os
import socket
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tools", "dp_dump_worker.py")
STRESS = os.path.join(ROOT, "tools", "gpu_stress.py")
SAVE_DIR = os.environ.get("DP_SAVE_DIR")


def run_world(world, out, mode, env):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    e = dict(os.environ, MCDSEG_SINGLE_DEVICE="1", MCDSEG_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4",
             MCDSEG_PRETRAINED="0")
    e.update(env)
    if world == 1:
        cmd = [sys.executable, WORKER, out, mode]
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr",
               "127.0.0.1", "--master-port", str(port), WORKER, out, mode]
    r = subprocess.run(cmd, env=e, capture_output=True, text=True, timeout=900)
    if r.returncode != 0:
        print("worker failed:", r.stdout[-1500:], r.stderr[-3000:])
        raise SystemExit(1)
    return ["%s.rank%d" % (out, k) for k in range(world)]


def diff(ref, other, limit=12):
    import torch
    a, b = torch.load(ref), torch.load(other)
    bad = []
    for k in a:
        x, y = a[k], b[k]
        if x.dtype.is_floating_point:
            ne = (x != y) & ~(x.isnan() & y.isnan())
        else:
            ne = x != y
        cnt = int(ne.sum())
        if cnt:
            d = (x.double() - y.double()).abs()
            sc = float(x.double().abs().max())
            bad.append((k, cnt, x.numel(), float(d.max()), sc))
    if not bad:
        return "identical (%d tensors)" % len(a)
    if SAVE_DIR:  # keep the differing tensor closest to the loss (last in model order) from both runs
        k = bad[-1][0]
        os.makedirs(SAVE_DIR, exist_ok=True)
        torch.save({"key": k, "ref": a[k], "other": b[k]}, os.path.join(SAVE_DIR, "%s_vs_%s.pt" % (os.path.basename(ref), os.path.basename(other))))
    lines = ["%d of %d tensors differ" % (len(bad), len(a))]
    shown = bad if len(bad) <= 2 * limit else bad[:limit] + [None] + bad[-limit:]
    for item in shown:
        if item is None:
            lines.append("    ...")
        else:
            lines.append("    %-44s %8d / %-9d differ  max|d| %.3e  (scale %.3e)" % item)
    return "\n".join(lines)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", default="step")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--cases", default="solo,world2,stress")
    ap.add_argument("--env", default="")
    ap.add_argument("--tag", default="")
    a = ap.parse_args()
    env = dict(kv.split("=", 1) for kv in a.env.split(",") if kv)
    tmp = tempfile.mkdtemp(prefix="dprepro")
    print("=== dp_repro mode=%s env=%s %s" % (a.mode, env, a.tag), flush=True)
    ref = run_world(1, os.path.join(tmp, "ref"), a.mode, env)[0]
    differing = 0
    for case in a.cases.split(","):
        world = 2 if case in ("world2", "stress2") else 1
        stress = None
        if case.startswith("stress"):
            stress = subprocess.Popen([sys.executable, STRESS], stdout=subprocess.PIPE, text=True)
            stress.stdout.readline()  # "ready"
        try:
            for rep in range(a.reps):
                t0 = time.time()
                files = run_world(world, os.path.join(tmp, "%s%d" % (case, rep)), a.mode, env)
                for f in files:
                    verdict = diff(ref, f)
                    differing += not verdict.startswith("identical")
                    print("%-8s rep %d %s (%.0f s): %s" % (case, rep, os.path.basename(f), time.time() - t0, verdict), flush=True)
                if world == 2:
                    print("%-8s rep %d rank0 vs rank1: %s" % (case, rep, diff(files[0], files[1])), flush=True)
        finally:
            if stress is not None:
                stress.terminate()
                stress.wait()
    print("=== %d run(s) differ from the solo reference" % differing, flush=True)
    sys.exit(1 if differing else 0)


if __name__ == "__main__":
    main()
