#!/bin/bash
# Development: add_fused_kernel with parts left out (PISCES_ADD_ABLATE, store_kernels.hip.h): its duration under rocprofv3 each.
#   on the build box:  bash tools/add_ablate.sh build "1 2 3"      on the GPU box:  bash tools/add_ablate.sh run "0 1 2 3"
LIST=${2:-"1 2 3 4 5 6 7 8 9 10"}
if [ "$1" = build ]; then
  for a in $LIST; do
    python -c "
from pisces_amd import build
build.build_native(out='gpurun_scratch/libab$a.so', extra_flags=['-DPISCES_ADD_ABLATE=$a'])" > /tmp/ab$a.log 2>&1 &
  done
  wait; ls gpurun_scratch/
else
  mkdir -p gpurun_out/ablate
  REPO=$(pwd); cd /tmp && export TMPDIR=/tmp; cd $REPO
  for a in $LIST; do
    lib=gpurun_scratch/libab$a.so; [ $a = 0 ] && lib=pisces_amd/libpisceship.so
    rm -rf /tmp/ab_$a
    PISCES_HIP_LIB=$PWD/$lib timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ab_$a -o s -- python tools/chain_bench.py > /tmp/ab_$a.log 2>&1
    f=$(find /tmp/ab_$a -name "*kernel_stats.csv" | head -1)
    echo "ablate $a: $(grep -E 'add_fused|gather_direct|call_store_tiles' $f | awk -F, '{printf "%s avg %.1f us min %.1f | ", substr($1,1,40), $4/1000, $6/1000}') $(grep 'device chain' /tmp/ab_$a.log | sed 's/.*device chain//')" | tee -a gpurun_out/ablate/add_ablate.txt
  done
fi
