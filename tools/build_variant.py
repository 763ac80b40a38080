#!/usr/bin/env python3
"""Development tool: build a variant of libmcdseg.so with extra -D flags next to the shipped one, for A/B timing on the GPU
box in ONE gpurun call:   python tools/build_variant.py prio -DMCD_SETPRIO=1
-> multichannel-semseg-with-uda_amd/mcdseg/libmcdseg_prio.so; select it with MCDSEG_LIB=<path>."""
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "multichannel-semseg-with-uda_amd")
tag, flags = sys.argv[1], sys.argv[2:]
out = os.path.join(PKG, "mcdseg", "libmcdseg_%s.so" % tag)
srcs = sorted(glob.glob(os.path.join(PKG, "csrc", "*.hip")))
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-fvisibility=hidden", "-shared",
       "-I", os.path.join(ROOT, "include"), "-I", os.path.join(PKG, "csrc"), "-o", out] + flags + srcs
subprocess.run(cmd, check=True)
print(out)
