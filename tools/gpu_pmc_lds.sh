#!/bin/bash
# One PMC pass (LDS activity) over tools/prof_encode.py.
set -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
rm -rf gpurun_out/pmc_lds; mkdir -p gpurun_out/pmc_lds
timeout -s KILL 150 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_WAVE_CYCLES --output-format csv -d gpurun_out/pmc_lds/a -o p -- python3 tools/prof_encode.py 3 > gpurun_out/pmc_lds/a.log 2>&1 || { echo "pass failed"; tail -5 gpurun_out/pmc_lds/a.log; exit 1; }
python3 tools/pmc_summary.py gpurun_out/pmc_lds > gpurun_out/pmc_lds/summary.txt
grep -A9 "k_assign_fast<8, 4, 0, false, true\|k_accum_dist" gpurun_out/pmc_lds/summary.txt
