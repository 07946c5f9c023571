#!/bin/bash
# Development A / B: add_fused_kernel's stream role as k persistent workgroups a CU in front of the read role (0 = read role in front)
for k in 0 1 2 3 4 0 2 3; do echo "stream_wgs_per_cu $k: $(PISCES_HIP_STREAM_WGS_PER_CU=$k python tools/chain_bench.py | tail -1 | sed 's/.*device chain//')"; done
for k in 0 2 3; do echo "config5 stream_wgs_per_cu $k: $(PISCES_HIP_STREAM_WGS_PER_CU=$k python tools/chain_bench.py --loci 15000 --depth 5000 --minbq 30 | tail -1 | sed 's/.*device chain//')"; done
for k in 2 3; do PISCES_HIP_STREAM_WGS_PER_CU=$k PISCES_HIP_LIB=$PWD/gpurun_scratch/libstamps.so python tools/chain_bench.py --reps 3 2>&1 | grep -E "stamps" | tail -1; done
