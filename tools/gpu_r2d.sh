#!/bin/bash
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r2d_tests.log 2>&1 || { echo "tests failed"; tail -40 gpurun_out/r2d_tests.log; exit 1; }
tail -1 gpurun_out/r2d_tests.log
timeout -k 10 200 python3 tools/time_img.py 2>&1 | grep -v amdgpu.ids
timeout -k 10 500 python -u bench.py > gpurun_out/r2d_bench.json 2> gpurun_out/r2d_bench.err || { echo "bench failed"; tail -20 gpurun_out/r2d_bench.err; exit 1; }
python3 -c "
import json
r=json.load(open('gpurun_out/r2d_bench.json'))
print(json.dumps({k:v for k,v in r.items() if k not in ('roofline','hbm_kernels','config')}, indent=1))
ro=r['roofline']; print({k:ro[k] for k in ro if k not in ('note','kernel')})
for k,v in r['hbm_kernels'].items(): print(k,{a:b for a,b in v.items() if a not in ('note','kernel')})
"
