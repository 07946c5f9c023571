REPO=$(pwd); cd /tmp && export TMPDIR=/tmp; cd $REPO
for lib in pisces_amd/libpisceship.so gpurun_scratch/libsa1.so; do
  rm -rf /tmp/pp; PISCES_HIP_LIB=$PWD/$lib timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -o s -- python tools/store_bench.py --reps 8 > /tmp/pp.log 2>&1
  python - $lib <<'PY'
import csv, glob, sys
f = glob.glob('/tmp/pp/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'call_store_tiles' in r['Name']:
        print(sys.argv[1], r['Calls'], 'avg %.1f min %.1f max %.1f us' % (float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3))
PY
done
