#!/usr/bin/env python3
"""Static instruction mix of one kernel in the gfx950 assembly of csrc/qtomo.hip (no GPU needed).
Usage: python scripts/isa_mix.py <mangled-name-substring> [asm-file]      (asm: hipcc -S --cuda-device-only)"""
import collections
import re
import sys

pat = sys.argv[1]
asm = sys.argv[2] if len(sys.argv) > 2 else "/tmp/qtomo.s"
lines = open(asm).read().split("\n")
start = None
for i, ln in enumerate(lines):
    m = re.match(r"^(_Z\w+):", ln)
    if m and pat in m.group(1):
        start, name = i, m.group(1)
        break
assert start is not None, "kernel not found"
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
mix = collections.Counter()
detail = collections.Counter()
for ln in lines[start:end]:
    t = ln.strip()
    if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
        continue
    op = t.split()[0]
    detail[op] += 1
    if op.startswith("v_mfma"): k = "mfma"
    elif op.startswith(("ds_",)): k = "lds"
    elif op.startswith(("global_", "flat_", "buffer_")): k = "vmem"
    elif op.startswith("scratch_"): k = "scratch"
    elif op.startswith("v_accvgpr"): k = "accvgpr"
    elif op.startswith("v_") and ("f64" in op): k = "valu_f64"
    elif op.endswith("_dpp") or "dpp" in t: k = "valu_dpp"
    elif op.startswith("v_"): k = "valu_other"
    elif op.startswith("s_waitcnt"): k = "waitcnt"
    elif op.startswith("s_"): k = "salu"
    else: k = "other"
    mix[k] += 1
print(name)
tot = sum(mix.values())
for k, v in mix.most_common():
    print(f"  {k:12s} {v:6d}  {100*v/tot:5.1f}%")
print("  total", tot)
print("  top ops:", ", ".join(f"{o} {c}" for o, c in detail.most_common(24)))
