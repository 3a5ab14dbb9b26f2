#!/bin/bash
# Device assembly of the library (few-shapes build) into build/vq_few.s, then the resource summary of one kernel.
# Usage: tools/asm_of.sh [extra hipcc flags]
cd "$(dirname "$0")/.."
mkdir -p build
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -Iinclude --cuda-device-only -S -DVQ_FEW_SHAPES "$@" \
    vector_quantization_amd/csrc/vq_hip.hip -o build/vq_few.s
