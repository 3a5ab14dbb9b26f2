#!/usr/bin/env python3
"""Per-kernel durations and the idle gaps between consecutive kernels from a rocprofv3 kernel trace CSV.
Usage: python3 tools/gaps.py <p_kernel_trace.csv> [first] [count]"""
import csv
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
first = int(sys.argv[2]) if len(sys.argv) > 2 else max(0, len(rows) - 40)
count = int(sys.argv[3]) if len(sys.argv) > 3 else 40
prev_end = None
for r in rows[first:first + count]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    print(f"{name:62s} dur {(e - s) / 1e3:8.1f} us  gap {gap:6.1f} us")
    prev_end = e
