#!/bin/bash
# GPU tests (incl. the not-yet-collected no-allocation test), bench, shape sweep per library variant.
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests tests/wip_no_alloc.py -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r2b_tests.log 2>&1 || { echo "tests failed"; tail -40 gpurun_out/r2b_tests.log; exit 1; }
tail -1 gpurun_out/r2b_tests.log
timeout -k 10 120 python -u -m pytest tests/wip_no_alloc.py -m gpu -x -q --timeout 100 --timeout-method thread > gpurun_out/r2b_noalloc_alone.log 2>&1 || { echo "no-alloc alone failed"; tail -40 gpurun_out/r2b_noalloc_alone.log; exit 1; }
tail -1 gpurun_out/r2b_noalloc_alone.log
timeout -k 10 300 python -u bench.py --steps 20 --warmup 3 --cpu-sample 0 > gpurun_out/r2b_bench.json 2> gpurun_out/r2b_bench.err || { echo "bench failed"; tail -5 gpurun_out/r2b_bench.err; exit 1; }
python3 - <<'PY'
import json
r = json.load(open("gpurun_out/r2b_bench.json"))
print(f"   value {r['value']:.3e}  ms/step {r['ms_per_step']:.4f}  encode_ms {r['encode_ms']:.4f}  sort_ms {r['sort']['kernel_ms']:.4f}  "
      f"search_ms {r['roofline']['kernel_ms']:.4f}  decode GB/s {r['decode_gbs']:.0f}  lbg ms/iter {r['lbg']['ms_per_iter']:.4f}")
PY
for so in vector_quantization_amd/lib/libvqhip.so vector_quantization_amd/lib/variant_*.so; do
  echo "== $so"
  VQ_HIP_LIB=$PWD/$so timeout -k 10 200 python3 -u tools/shape_sweep.py > gpurun_out/r2b_sweep_$(basename $so .so).txt 2>&1 || { echo "sweep failed"; tail -5 gpurun_out/r2b_sweep_$(basename $so .so).txt; exit 1; }
  cat gpurun_out/r2b_sweep_$(basename $so .so).txt
done
