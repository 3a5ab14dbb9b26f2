#!/bin/bash
# Kernel trace + PMC passes over tools/prof_shapes.py (d = 192 / K = 4096, d = 48 / K = 256, K = 1024 / d = 16),
# one rocprofv3 run per counter group.  Outputs under gpurun_out/pmc_shapes/.
set -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
rm -rf gpurun_out/pmc_shapes; mkdir -p gpurun_out/pmc_shapes
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pmc_shapes/trace -o p -- python3 tools/prof_shapes.py 5 > gpurun_out/pmc_shapes/trace.log 2>&1 || { echo "trace failed"; tail -5 gpurun_out/pmc_shapes/trace.log; exit 1; }
run() { # name, counters...
  local name=$1; shift
  timeout -s KILL 300 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/pmc_shapes/$name -o p -- python3 tools/prof_shapes.py 2 > gpurun_out/pmc_shapes/$name.log 2>&1 || { echo "pass $name failed"; tail -5 gpurun_out/pmc_shapes/$name.log; exit 1; }
}
run sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY
run sq2 SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU GRBM_GUI_ACTIVE
run lds SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_INSTS_LDS
run fetch FETCH_SIZE
run write WRITE_SIZE
python3 tools/pmc_summary.py gpurun_out/pmc_shapes > gpurun_out/pmc_shapes/summary.txt
echo "pmc shapes done"
