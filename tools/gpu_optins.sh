#!/bin/bash
# Hardware check of the kernels that are still opt-in (environment switches): for each switch the GPU
# parity tests run with it set, then the bench, so that one GPU session says which of them can become
# the default.  Stops at the first failure.  Outputs under gpurun_out/optin_<name>.*.
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for flag in VQ_SORT_LDS VQ_DECODE_BLOCKROWS VQ_FIXUP_PRUNE VQ_PRUNE_WIDE; do
  echo "== $flag"
  env $flag=1 timeout -k 10 400 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread \
      > gpurun_out/optin_$flag.tests.log 2>&1 || { echo "$flag: tests failed"; tail -20 gpurun_out/optin_$flag.tests.log; exit 1; }
  tail -1 gpurun_out/optin_$flag.tests.log
  env $flag=1 timeout -k 10 300 python -u bench.py --steps 20 --warmup 3 --cpu-sample 0 \
      > gpurun_out/optin_$flag.bench.json 2> gpurun_out/optin_$flag.bench.err || { echo "$flag: bench failed"; tail -5 gpurun_out/optin_$flag.bench.err; exit 1; }
  python3 - "$flag" <<'PY'
import json, sys
r = json.load(open(f"gpurun_out/optin_{sys.argv[1]}.bench.json"))
print(f"   value {r['value']:.3e}  ms/step {r['ms_per_step']:.4f}  encode_ms {r['encode_ms']:.4f}  sort_ms {r['sort']['kernel_ms']:.4f}  "
      f"search_ms {r['roofline']['kernel_ms']:.4f}  decode GB/s {r['decode_gbs']:.0f}  lbg ms/iter {r['lbg']['ms_per_iter']:.4f}")
PY
done
