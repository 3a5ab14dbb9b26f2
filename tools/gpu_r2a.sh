#!/bin/bash
# Round-2 first hardware session: operand-form VALU microbenchmark, default GPU tests + bench, then the
# opt-in kernels one switch at a time.  Stops at the first failure.  Outputs under gpurun_out/.
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 180 tools/ubench/valu_bank > gpurun_out/valu_bank.txt 2>&1 || { echo "valu_bank failed"; tail -5 gpurun_out/valu_bank.txt; exit 1; }
echo "ubench done"
timeout -k 10 400 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/r2a_tests.log 2>&1 || { echo "default tests failed"; tail -30 gpurun_out/r2a_tests.log; exit 1; }
tail -1 gpurun_out/r2a_tests.log
timeout -k 10 300 python -u bench.py --steps 20 --warmup 3 --cpu-sample 0 > gpurun_out/r2a_bench.json 2> gpurun_out/r2a_bench.err || { echo "bench failed"; tail -5 gpurun_out/r2a_bench.err; exit 1; }
echo "bench done"
bash tools/gpu_optins.sh
