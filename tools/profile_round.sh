#!/bin/bash
# Collects what profiles/ holds for a round, on the GPU box:  bash tools/profile_round.sh r01
# (1) un-profiled bench line, (2) rocprofv3 kernel-trace stats of the same command, (3) PMC HBM traffic in separate passes.
set -u
TAG=${1:-r01}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $REPO
timeout 600 python bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err
tail -1 $OUT/bench_n1.json
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o bench -- python bench.py --no-cpu-baseline --no-pipelined --no-shard-check > $OUT/bench_prof.log 2>&1
find /tmp/prof_$TAG -name "*kernel_stats.csv" -exec cp {} $OUT/bench_kernel_stats.csv \;
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $C --kernel-trace --kernel-include-regex call_tiles --output-format csv -d /tmp/pmc_${TAG}_$C -o pmc -- python bench.py --no-cpu-baseline --no-pipelined --no-shard-check --steps 12 --warmup 2 > $OUT/pmc_$C.log 2>&1
  f=$(find /tmp/pmc_${TAG}_$C -name "*counter_collection.csv" | head -1)
  python - "$f" $C <<'PY' | tee $OUT/pmc_$C.txt
import sys, csv, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if 'call_tiles' in r['Kernel_Name']:
        acc[(r['Kernel_Name'][:60], r['Counter_Name'])].append(float(r['Counter_Value']))
for k, v in acc.items():
    print(k[0], k[1], "mean per dispatch", sum(v) / len(v), "n", len(v))
PY
done
head -5 $OUT/bench_kernel_stats.csv
