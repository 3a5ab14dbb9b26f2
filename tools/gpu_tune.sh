#!/bin/bash
cd "$(dirname "$0")/.."
for rep in 1; do
for so in vector_quantization_amd/lib/libvqhip.so vector_quantization_amd/lib/variant_*.so; do
  echo -n "$(basename $so) "; VQ_HIP_LIB=$PWD/$so timeout -k 10 120 python3 tools/tune_variants.py 2>&1 | tail -1 | cut -c1-250 || exit 1
done
done
