#!/bin/bash
# One GPU-box session: smoke, bench, kernel-trace profile of the bench.  Outputs under gpurun_out/.
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 120 python -u -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1 || { echo smoke failed; tail -20 gpurun_out/smoke.log; exit 1; }
timeout -k 10 300 python -u bench.py --steps 20 --warmup 3 > gpurun_out/bench.json 2> gpurun_out/bench.err || { echo bench failed; tail -20 gpurun_out/bench.err; exit 1; }
cat gpurun_out/bench.json
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o bench -- python3 bench.py --steps 20 --warmup 3 --cpu-sample 0 > gpurun_out/prof_bench.json 2> gpurun_out/prof.err || { echo rocprof failed; tail -20 gpurun_out/prof.err; exit 1; }
ls -R gpurun_out/prof | head -20
