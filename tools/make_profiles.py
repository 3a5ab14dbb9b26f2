#!/usr/bin/env python3
"""Copies the rocprofv3 outputs of tools/gpu_round.sh from gpurun_out/ into profiles/ (filtered to this
library's kernels) and derives the per-kernel PMC table and the summary bench.py reads.
Usage: python3 tools/make_profiles.py r03"""
import collections
import csv
import glob
import hashlib
import json
import os
import shutil
import statistics
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, "profiles")
src = os.path.join(root, "gpurun_out")
os.makedirs(out, exist_ok=True)


def short(name):
    return name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]


def is_ours(name):
    return short(name).startswith("k_")


lib_sha = hashlib.sha256(open(os.path.join(root, "vector_quantization_amd", "lib", "libvqhip.so"), "rb").read()).hexdigest()[:16]

# ---- bench line + kernel trace of the bench command --------------------------------------------------
shutil.copy(os.path.join(src, "bench.json"), os.path.join(out, f"{tag}_bench.json"))
bench = json.load(open(os.path.join(src, "bench.json")))
assert bench["library_sha256_16"] == lib_sha, "bench.json was made with another library build"
stats = glob.glob(os.path.join(src, "prof", "**", "bench_kernel_stats.csv"), recursive=True)[0]
rows = [r for r in csv.DictReader(open(stats)) if is_ours(r["Name"])]
with open(os.path.join(out, f"{tag}_bench_kernel_stats.csv"), "w", newline="") as f:
    w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
    w.writeheader()
    w.writerows(rows)
trace = glob.glob(os.path.join(src, "prof", "**", "bench_kernel_trace.csv"), recursive=True)[0]
dur = collections.defaultdict(list)
regs = {}
for r in csv.DictReader(open(trace)):
    if not is_ours(r["Kernel_Name"]):
        continue
    key = (short(r["Kernel_Name"]), int(r["Grid_Size_X"]))
    dur[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    regs[key] = dict(vgpr=int(r["VGPR_Count"]), sgpr=int(r["SGPR_Count"]), lds=int(r["LDS_Block_Size"]),
                     scratch=int(r["Scratch_Size"]), workgroup=int(r["Workgroup_Size_X"]))
durations = {f"{k}[grid={g}]": dict(n=len(v), median_us=statistics.median(v), mean_us=sum(v) / len(v), min_us=min(v), **regs[(k, g)])
             for (k, g), v in sorted(dur.items())}
json.dump(durations, open(os.path.join(out, f"{tag}_bench_kernel_durations.json"), "w"), indent=1)

# ---- PMC passes over tools/prof_encode.py ------------------------------------------------------------------
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for name in ("sq1", "sq2", "fetch", "write"):
    files = glob.glob(os.path.join(src, "pmc", name, "**", "p_counter_collection.csv"), recursive=True)
    if not files:
        continue
    rows = [r for r in csv.DictReader(open(files[0])) if is_ours(r["Kernel_Name"])]
    with open(os.path.join(out, f"{tag}_pmc_{name}.csv"), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
        w.writeheader()
        w.writerows(rows)
    for r in rows:
        key = f"{short(r['Kernel_Name'])}[grid={r.get('Grid_Size', r.get('Grid_Size_X', ''))}]"
        agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
per_kernel = []
for key in sorted(agg):
    c = {n: sum(v) / len(v) for n, v in agg[key].items()}
    e = dict(kernel=key, dispatches=max(len(v) for v in agg[key].values()), counters=c)
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        # FETCH_SIZE / WRITE_SIZE count KB; FETCH_SIZE is doubled (gfx950 correction for wide coalesced reads)
        e["hbm_read_bytes"] = c["FETCH_SIZE"] * 1024 * 2
        e["hbm_write_bytes"] = c["WRITE_SIZE"] * 1024
    if "GRBM_GUI_ACTIVE" in c and "SQ_INSTS_VALU" in c:
        e["shader_cycles"] = c["GRBM_GUI_ACTIVE"] / 8              # the counter adds the 8 XCDs
        e["valu_issue_busy_frac"] = c["SQ_INSTS_VALU"] * 4 / (e["shader_cycles"] * 1024)   # 4 cycles per wave64 op, 1024 SIMDs
    per_kernel.append(e)
json.dump(per_kernel, open(os.path.join(out, f"{tag}_pmc_per_kernel.json"), "w"), indent=1)

N = 2048 * 2048
big = [e for e in per_kernel if "hbm_read_bytes" in e]


def grid_of(key):
    return int(key.split("[grid=")[1].rstrip("]") or 0)


def full_size(key):                      # the launches over all 4 194 304 vectors of the workload
    return grid_of(key) >= 262144


def role_bytes(pred):
    c = [e for e in big if pred(e["kernel"]) and full_size(e["kernel"])]
    if not c:
        return None
    e = max(c, key=lambda e: grid_of(e["kernel"]))
    return dict(kernel=e["kernel"], hbm_bytes=e["hbm_read_bytes"] + e["hbm_write_bytes"])


roles = dict(
    search_filter=role_bytes(lambda k: k.startswith("k_assign_fast") and k.split("[")[0].endswith("false, true, false>")),
    search_full=role_bytes(lambda k: k.startswith("k_assign_fast") and k.split("[")[0].endswith("false, false, false>")),
    lbg_assign=role_bytes(lambda k: k.startswith("k_assign_fast") and k.split("[")[0].endswith("false, true, true>")),
    image_sort=role_bytes(lambda k: k.startswith("k_window_sort") and "SrcImage" in k),
    decode=role_bytes(lambda k: k.startswith("k_decode")),
    blocks=role_bytes(lambda k: k.startswith("k_blocks")))
summary = dict(
    roles=roles,
    library_sha256_16=lib_sha,
    workload="tools/prof_encode.py: 8192x8192 gray, 4x4 blocks, K=256 (N = 4194304 block vectors)",
    hbm_bytes_by_kernel={e["kernel"]: e["hbm_read_bytes"] + e["hbm_write_bytes"] for e in big
                         if full_size(e["kernel"])},
    valu={e["kernel"]: dict(valu_wave_insts=e["counters"]["SQ_INSTS_VALU"], valu_issue_busy_frac=e["valu_issue_busy_frac"])
          for e in per_kernel if "valu_issue_busy_frac" in e and e["kernel"].startswith("k_assign_fast") and grid_of(e["kernel"]) >= 262144},
    note="FETCH_SIZE (KB) doubled per the gfx950 correction for wide coalesced reads; WRITE_SIZE (KB) as is; separate "
         "rocprofv3 --pmc passes over tools/prof_encode.py on the library build named above (PMC passes run at lower "
         "clocks than un-profiled runs: durations come from the kernel trace, not from these passes).")
json.dump(summary, open(os.path.join(out, f"{tag}_pmc_summary.json"), "w"), indent=1)
print(json.dumps(summary, indent=1))

# ---- PMC passes over tools/prof_shapes.py (d = 192 / K = 4096, d = 48 / K = 256, K = 1024 / d = 16) ------------------
# kept: the launches over the measured images (large grids) of the search, fix-up, sort and accumulation kernels
KEEP = ("k_assign_fast", "k_fixup", "k_window_sort", "k_sort_", "k_accum_dist", "k_prep_")
agg = collections.defaultdict(lambda: collections.defaultdict(list))
shape_dir = os.path.join(src, "pmc_shapes")
for name in ("sq1", "sq2", "lds", "fetch", "write"):
    files = glob.glob(os.path.join(shape_dir, name, "**", "p_counter_collection.csv"), recursive=True)
    if not files:
        continue
    rows = [r for r in csv.DictReader(open(files[0]))
            if short(r["Kernel_Name"]).startswith(KEEP) and int(r.get("Grid_Size", r.get("Grid_Size_X", 0)) or 0) >= 65536]
    if not rows:
        continue
    with open(os.path.join(out, f"{tag}_pmc_shapes_{name}.csv"), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
        w.writeheader()
        w.writerows(rows)
    for r in rows:
        key = f"{short(r['Kernel_Name'])}[grid={r.get('Grid_Size', r.get('Grid_Size_X', ''))}]"
        agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
shapes = []
for key in sorted(agg):
    c = {n: sum(v) / len(v) for n, v in agg[key].items()}
    e = dict(kernel=key, dispatches=max(len(v) for v in agg[key].values()), counters=c)
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        e["hbm_read_bytes"] = c["FETCH_SIZE"] * 1024 * 2
        e["hbm_write_bytes"] = c["WRITE_SIZE"] * 1024
    if "GRBM_GUI_ACTIVE" in c and "SQ_INSTS_VALU" in c:
        e["shader_cycles"] = c["GRBM_GUI_ACTIVE"] / 8
        e["valu_issue_busy_frac"] = c["SQ_INSTS_VALU"] * 4 / (e["shader_cycles"] * 1024)
    if "SQ_WAVE_CYCLES" in c and "SQ_WAVES" in c and "GRBM_GUI_ACTIVE" in c:
        # SQ_WAVE_CYCLES counts quad-cycles: mean resident waves per SIMD over the launch
        e["mean_waves_per_simd"] = c["SQ_WAVE_CYCLES"] * 4 / (c["GRBM_GUI_ACTIVE"] / 8 * 1024)
    if "SQ_LDS_IDX_ACTIVE" in c and "GRBM_GUI_ACTIVE" in c:
        e["lds_active_frac_per_cu"] = c["SQ_LDS_IDX_ACTIVE"] / 256 / (c["GRBM_GUI_ACTIVE"] / 8)
    shapes.append(e)
if shapes:
    json.dump(dict(library_sha256_16=lib_sha, workload="tools/prof_shapes.py", kernels=shapes),
              open(os.path.join(out, f"{tag}_pmc_shapes_per_kernel.json"), "w"), indent=1)
    trace = glob.glob(os.path.join(shape_dir, "trace", "**", "p_kernel_stats.csv"), recursive=True)
    if trace:
        rows = [r for r in csv.DictReader(open(trace[0])) if is_ours(r["Name"])]
        with open(os.path.join(out, f"{tag}_shapes_kernel_stats.csv"), "w", newline="") as f:
            w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
            w.writeheader()
            w.writerows(rows)
