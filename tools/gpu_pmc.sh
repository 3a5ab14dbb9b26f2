#!/bin/bash
# Kernel trace + PMC passes over tools/prof_encode.py, one rocprofv3 run per counter group.
set -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
rm -rf gpurun_out/pmc; mkdir -p gpurun_out/pmc
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pmc/trace -o p -- python3 tools/prof_encode.py 10 > gpurun_out/pmc/trace.log 2>&1 || { echo "trace failed"; tail -5 gpurun_out/pmc/trace.log; exit 1; }
run() { # name, counters...
  local name=$1; shift
  timeout -s KILL 150 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/pmc/$name -o p -- python3 tools/prof_encode.py 3 > gpurun_out/pmc/$name.log 2>&1 || { echo "pass $name failed"; tail -5 gpurun_out/pmc/$name.log; exit 1; }
}
run sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY
run sq2 SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU GRBM_GUI_ACTIVE
run fetch FETCH_SIZE
run write WRITE_SIZE
python3 tools/pmc_summary.py gpurun_out/pmc > gpurun_out/pmc/summary.txt
find gpurun_out/pmc -name "*stats*.csv" | head -3
