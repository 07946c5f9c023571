#!/bin/bash
# Device assembly of the library's kernels for gfx950:  tools/isa.sh out.s [-DFLAG ...]
out=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math --cuda-device-only -S -x hip "$@" pisces_amd/csrc/pisces_hip.hip -o "$out"
