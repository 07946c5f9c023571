#!/bin/bash
# Development A / B of add_fused_kernel's launch shape on config 2's batch (device chain by events, tools/chain_bench.py)
for lib in gpurun_scratch/libocc4.so gpurun_scratch/libocc5.so gpurun_scratch/libocc6.so; do
  [ -f $lib ] || continue
  for st in 1 2 3; do
    for r in 1 2; do echo "$lib stride $st: $(PISCES_HIP_LIB=$PWD/$lib PISCES_HIP_ROLE_STRIDE=$st python tools/chain_bench.py | tail -1 | sed 's/.*device chain//')"; done
  done
  echo "$lib config5: $(PISCES_HIP_LIB=$PWD/$lib python tools/chain_bench.py --loci 15000 --depth 5000 --minbq 30 | tail -1 | sed 's/.*device chain//')"
done
[ -f gpurun_scratch/libstamps.so ] && PISCES_HIP_LIB=$PWD/gpurun_scratch/libstamps.so python tools/chain_bench.py --reps 3 2>&1 | grep -E "stamps" | tail -2
