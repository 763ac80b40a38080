#!/usr/bin/env python3
"""VGPR / AGPR / SGPR / spill / LDS / scratch figures of the kernels in libmcdseg.so (from the code objects' metadata notes; no GPU):
    python tools/kernel_resources.py [substring ...]"""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.environ.get("MCDSEG_LIB") or os.path.join(ROOT, "multichannel-semseg-with-uda_amd", "mcdseg", "libmcdseg.so")
LLVM = os.environ.get("ROCM_LLVM_BIN", "/opt/rocm/lib/llvm/bin")


def main():
    pats = sys.argv[1:]
    with tempfile.TemporaryDirectory() as tmp:
        copy = os.path.join(tmp, "lib.so")
        with open(LIB, "rb") as f, open(copy, "wb") as g:
            g.write(f.read())
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", copy], check=True, capture_output=True, cwd=tmp)
        rows = []
        for co in sorted(glob.glob(os.path.join(tmp, "lib.so.*gfx950*"))):
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True).stdout
            for blk in notes.split("- .agpr_count:")[1:]:
                def f(key):
                    m = re.search(r"\.%s:\s+(\S+)" % key, blk)
                    return m.group(1) if m else "?"
                name = f("name")
                dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
                dem = re.sub(r"\(anonymous namespace\)::", "", dem).split("(")[0].replace("void ", "")
                rows.append((dem, blk.split()[0], f("vgpr_count"), f("vgpr_spill_count"), f("sgpr_count"), f("sgpr_spill_count"),
                             f("group_segment_fixed_size"), f("private_segment_fixed_size")))
    print("%-86s %5s %5s %6s %5s %6s %7s %8s" % ("kernel", "agpr", "vgpr", "vspill", "sgpr", "sspill", "lds", "scratch"))
    for r in sorted(rows):
        if not pats or any(p in r[0] for p in pats):
            print("%-86s %5s %5s %6s %5s %6s %7s %8s" % ((r[0][:86],) + r[1:]))


if __name__ == "__main__":
    main()
