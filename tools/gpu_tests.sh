#!/bin/bash
# The GPU suite with its log kept under gpurun_out/<tag>/ (the ROCm banner lines of the box filtered from what is shown):
#   bash tools/gpu_tests.sh <tag> [pytest arguments, default: tests -m gpu]
TAG=${1:-r06}; shift
mkdir -p gpurun_out/$TAG
if [ $# -eq 0 ]; then set -- tests -m gpu; fi
timeout 900 python -m pytest "$@" -x -q > gpurun_out/$TAG/gpu_tests.log 2>&1
grep -v "^ROCm\|^Hostname\|^Librccl\|^RCCL\|^HIP\|amdgpu.ids" gpurun_out/$TAG/gpu_tests.log | tail -12
