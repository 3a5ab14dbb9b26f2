#!/bin/bash
# One PMC pass (instruction mix) over tools/prof_encode.py.
set -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc
timeout -s KILL 150 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d gpurun_out/pmc/sq1 -o p -- python3 tools/prof_encode.py 3 > gpurun_out/pmc/sq1.log 2>&1 || { echo "pass failed"; tail -5 gpurun_out/pmc/sq1.log; exit 1; }
python3 tools/pmc_summary.py gpurun_out/pmc > gpurun_out/pmc/summary.txt
grep -A9 "k_assign_fast" gpurun_out/pmc/summary.txt
