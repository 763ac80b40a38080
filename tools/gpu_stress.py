"""A neighbour that loads the memory system: streams device-to-device copies (and a few GEMM-free elementwise passes) on cuda:0
until terminated.  Started as a CHILD process by the determinism tools / tests before they make any GPU call of their own.
Prints "ready" once the loop is running.  GPU_STRESS_MB sets the buffer size (default 1024), GPU_STRESS_SECONDS a time limit."""
import os
import sys
import time

import torch


def main():
    mb = int(os.environ.get("GPU_STRESS_MB", "1024"))
    limit = float(os.environ.get("GPU_STRESS_SECONDS", "600"))
    dev = torch.device("cuda:0")
    a = torch.empty(mb * 1024 * 1024 // 4, dtype=torch.float32, device=dev).normal_()
    b = torch.empty_like(a)
    torch.cuda.synchronize()
    print("ready", flush=True)
    t0 = time.time()
    while time.time() - t0 < limit:
        for _ in range(8):
            b.copy_(a)
            a.add_(b, alpha=1e-9)
        torch.cuda.synchronize()
    return 0


if __name__ == "__main__":
    sys.exit(main())
