#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs (gpurun_out/pmc/*/p_counter_collection.csv): per kernel and
counter, the mean over dispatches of the full-size launches."""
import csv
import glob
import collections
import sys

root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc"
agg = collections.defaultdict(list)
for f in sorted(glob.glob(root + "/*/p_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        short = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        if not short.startswith("k_"):
            continue
        key = (short[:40], r.get("Grid_Size", r.get("Grid_Size_X", "")), r.get("LDS_Block_Size", ""), r["Counter_Name"])
        agg[key].append(float(r["Counter_Value"]))
last = None
for key in sorted(agg):
    k = key[:3]
    if k != last:
        print(f"\n{key[0]}  grid={key[1]} lds={key[2]}  dispatches={len(agg[key])}")
        last = k
    v = agg[key]
    print(f"    {key[3]:28s} mean {sum(v)/len(v):16.1f}")
