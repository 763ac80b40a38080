#!/usr/bin/env python3
"""Development tool (GPU): one of bench.py's ``other_configs`` entries alone, as JSON (the same code path as the benchmark line's entry).
    python tools/bench_one_config.py cfg5_f16 [steps] [--no-roofline]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (puts the package on sys.path)
import torch  # noqa: E402

if __name__ == "__main__":
    tag = sys.argv[1] if len(sys.argv) > 1 else "cfg5_f16"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 2
    res = bench.other_config(tag, torch.device("cuda:0"), steps, want_roofline="--no-roofline" not in sys.argv)
    print(json.dumps({tag: res}))
