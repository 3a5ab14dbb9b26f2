#!/bin/bash
set -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/fix -o f -- python3 tools/prof_fixup.py > gpurun_out/fix.log 2>&1 || { tail -5 gpurun_out/fix.log; exit 1; }
python3 - <<'PY'
import csv
rows=[r for r in csv.DictReader(open('gpurun_out/fix/f_kernel_trace.csv')) if 'k_fixup' in r['Kernel_Name'] or 'k_assign_fast' in r['Kernel_Name']]
for r in rows[-12:]:
    print(r['Kernel_Name'][26:50], r['Grid_Size_X'], (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
PY
