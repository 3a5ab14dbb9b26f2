#!/bin/bash
# Builds libvqhip.so for gfx950 (MI355X) in-tree.  hipcc cross-compiles without a GPU.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
mkdir -p "$HERE/../lib"
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -ffp-contract=off \
    -I"$HERE/../../include" "$HERE/vq_hip.hip" -o "$HERE/../lib/libvqhip.so" "$@"
