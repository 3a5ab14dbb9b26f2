#!/bin/bash
# Per-kernel durations (rocprofv3 kernel trace) of tools/tune_variants.py for every library build
# under vector_quantization_amd/lib/.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for so in vector_quantization_amd/lib/libvqhip.so vector_quantization_amd/lib/variant_*.so; do
  [ -f "$so" ] || continue
  name=$(basename $so .so)
  export VQ_HIP_LIB=$PWD/$so
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tp/$name -o t -- python3 tools/tune_variants.py > gpurun_out/tp_$name.log 2>&1 || { echo "$name failed"; tail -5 gpurun_out/tp_$name.log; exit 1; }
  python3 - "$name" <<'PY'
import csv, sys, glob, collections, statistics
name = sys.argv[1]
f = glob.glob(f"gpurun_out/tp/{name}/**/t_kernel_trace.csv", recursive=True)[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    if not k.startswith("k_"): continue
    d[(k, r["Grid_Size_X"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
out = []
for (k, g), v in sorted(d.items()):
    if int(g) >= 1024:
        out.append(f"{k}[{g}] n={len(v)} med={statistics.median(v):.1f}")
print(name, " | ".join(out))
PY
done
