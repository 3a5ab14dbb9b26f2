#!/bin/bash
# One GPU-box session for the round's evidence: smoke, GPU tests, the bench line, a kernel trace of the bench
# command, PMC passes.  Outputs under gpurun_out/ (tools/make_profiles.py copies what is kept into profiles/).
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 120 python -u -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1 || { echo smoke failed; tail -20 gpurun_out/smoke.log; exit 1; }
tail -1 gpurun_out/smoke.log
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/gpu_tests.log 2>&1 || { echo tests failed; tail -30 gpurun_out/gpu_tests.log; exit 1; }
tail -1 gpurun_out/gpu_tests.log
timeout -k 10 500 python -u bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err || { echo bench failed; tail -20 gpurun_out/bench.err; exit 1; }
echo "bench done"
rm -rf gpurun_out/prof
timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o bench -- python3 bench.py --no-shapes --cpu-sample 0 > gpurun_out/prof_bench.json 2> gpurun_out/prof.err || { echo rocprof failed; tail -20 gpurun_out/prof.err; exit 1; }
echo "trace done"
bash tools/gpu_pmc.sh || exit 1
echo "pmc done"
bash tools/gpu_pmc_shapes.sh || exit 1
