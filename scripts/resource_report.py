#!/usr/bin/env python3
"""Compile csrc/qtomo.hip for gfx950 with -Rpass-analysis=kernel-resource-usage and print, per kernel, VGPRs / AGPRs /
scratch bytes per lane / occupancy / spills (no GPU needed).  Usage: python scripts/resource_report.py [filter]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from quantpy_amd.build import CSRC, SCHED_FLAGS, _hipcc  # noqa: E402

out = subprocess.run([_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", *SCHED_FLAGS,
                      "-Rpass-analysis=kernel-resource-usage", "-o", "/tmp/qtomo_report.so", os.path.join(CSRC, "qtomo.hip")],
                     capture_output=True, text=True)
text = out.stderr
filt = sys.argv[1] if len(sys.argv) > 1 else ""
names = re.findall(r"Function Name: (\S+)", text)
demangled = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
blocks = re.split(r"remark: [^\n]*Function Name: ", text)[1:]
print(f"{'kernel':58s} VGPR AGPR scratch occ sgprSpill vgprSpill  LDS")
for blk, name in zip(blocks, demangled):
    def g(key):
        m = re.search(key + r": (\d+)", blk)
        return int(m.group(1)) if m else -1
    name = re.sub(r"\(.*", "", name).replace("void qt::", "")
    if filt and filt not in name:
        continue
    scratch, occ, lds = g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]")
    print(f"{name:58s} {g('VGPRs'):4d} {g('AGPRs'):4d} {scratch:7d} {occ:3d} {g('SGPRs Spill'):9d} {g('VGPRs Spill'):9d} {lds:5d}")
