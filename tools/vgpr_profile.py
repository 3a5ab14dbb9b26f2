#!/usr/bin/env python3
"""Highest vector register named per stretch of a kernel's assembly (build/vq_few.s), to find the code that sets the
kernel's register count.  Usage: tools/vgpr_profile.py <substring of the mangled kernel name> [lines per stretch]"""
import re
import sys

name = sys.argv[1]
step = int(sys.argv[2]) if len(sys.argv) > 2 else 100
lines = open("build/vq_few.s").read().split("\n")
start = next(i for i, l in enumerate(lines) if name in l and not l.startswith("\t") and ":" in l)
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
body = lines[start:end]
for lo in range(0, len(body), step):
    chunk = body[lo:lo + step]
    hi = 0
    for l in chunk:
        for m in re.finditer(r"\bv(\d+)\b|v\[(\d+):(\d+)\]", l):
            hi = max(hi, int(m.group(1) or m.group(3)))
    kinds = {}
    for l in chunk:
        t = l.strip().split(" ")[0].split("\t")[0]
        for k in ("global_load", "global_store", "s_load", "v_sad", "ds_", "v_readlane", "v_writelane", "scratch_", "v_add_f64", "global_atomic"):
            if t.startswith(k):
                kinds[k] = kinds.get(k, 0) + 1
    print(f"{lo:6d} maxv {hi:3d}  " + " ".join(f"{k}:{v}" for k, v in sorted(kinds.items())))
