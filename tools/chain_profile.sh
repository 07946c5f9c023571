REPO=$(pwd); cd /tmp && export TMPDIR=/tmp; cd $REPO
for lib in pisces_amd/libpisceship.so; do
  rm -rf /tmp/pp; PISCES_HIP_LIB=$PWD/$lib timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -o s -- python tools/chain_bench.py > /tmp/pp.log 2>&1
  echo "== $lib: $(grep 'device chain' /tmp/pp.log | sed 's/.*device chain//')"
  python - <<'PY'
import csv, glob
f = glob.glob('/tmp/pp/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = r['Name']
    if 'pisces' in n and 'build_' not in n:
        print('  ', n.split('(')[0][-45:], r['Calls'], 'avg %.1f min %.1f max %.1f us' % (float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3))
PY
done
python tools/chain_bench.py | tail -1
python tools/chain_bench.py --loci 15000 --depth 5000 --minbq 30 | tail -1
