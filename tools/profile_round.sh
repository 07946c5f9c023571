#!/bin/bash
# Collects what profiles/ holds for a round, on the GPU box:  bash tools/profile_round.sh r03
# (1) un-profiled bench line, (2) rocprofv3 kernel-trace stats of the same command, (3) PMC HBM traffic in separate passes (counters
# restricted to the hot kernel: the synthetic generator's thousands of small torch kernels are not instrumented), (4) SQ / LDS counters
# of the hot kernel, (5) the streaming surface's and the BGZF / BAM kernels' statistics and counters.
set -u
TAG=${1:-r06}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $REPO
PROF_FLAGS="--no-cpu-baseline --no-end-to-end --no-pipelined --no-shard-check"
timeout 900 python bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err
tail -1 $OUT/bench_n1.json | cut -c1-400
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o bench -- python bench.py $PROF_FLAGS > $OUT/bench_prof.log 2>&1
find /tmp/prof_$TAG -name "*kernel_stats.csv" -exec cp {} $OUT/bench_kernel_stats.csv \;
grep call_tiles $OUT/bench_kernel_stats.csv | sed 's/.*)",//'
pmc() {   # pmc <name> <kernel regex> <counters...> -- <command...>
  local name=$1 regex=$2; shift 2
  local counters=()
  while [ "$1" != "--" ]; do counters+=("$1"); shift; done; shift
  rm -rf /tmp/pmc_$name
  timeout 300 rocprofv3 --pmc "${counters[@]}" --kernel-trace --kernel-include-regex "$regex" --output-format csv -d /tmp/pmc_$name -o pmc -- "$@" > $OUT/pmc_$name.log 2>&1
  f=$(find /tmp/pmc_$name -name "*counter_collection.csv" | head -1)
  python - "$f" "$regex" <<'PY' | tee $OUT/pmc_$name.txt
import sys, csv, collections, re
acc = collections.defaultdict(list)
try:
    for r in csv.DictReader(open(sys.argv[1])):
        if re.search(sys.argv[2], r['Kernel_Name']):
            acc[(r['Kernel_Name'].split('(')[0][-60:], r['Counter_Name'])].append(float(r['Counter_Value']))
except Exception as e:
    print("no counter file:", e)
for k, v in sorted(acc.items()):
    print(k[0], k[1], "mean per dispatch", sum(v) / len(v), "n", len(v))
PY
}
pmc FETCH_SIZE call_tiles FETCH_SIZE -- python bench.py $PROF_FLAGS --steps 12 --warmup 2
pmc WRITE_SIZE call_tiles WRITE_SIZE -- python bench.py $PROF_FLAGS --steps 12 --warmup 2
pmc sq_lds call_tiles SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS -- python bench.py $PROF_FLAGS --steps 12 --warmup 2
pmc sq_wave call_tiles SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -- python bench.py $PROF_FLAGS --steps 12 --warmup 2
# streaming surface + device finder + BGZF / BAM kernels
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${TAG}_stream -o s -- python tools/hostfed_bench.py > $OUT/hostfed.log 2>&1
find /tmp/prof_${TAG}_stream -name "*kernel_stats.csv" -exec cp {} $OUT/streaming_kernel_stats.csv \;
grep "host-fed" $OUT/hostfed.log | tail -1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${TAG}_bam -o s -- python tools/bam_bench.py 400000 > $OUT/bam_bench.log 2>&1
find /tmp/prof_${TAG}_bam -name "*kernel_stats.csv" -exec cp {} $OUT/bam_kernel_stats.csv \;
grep "bam decode" $OUT/bam_bench.log | tail -1
timeout 300 python tools/bgzf_bench.py 256 2>&1 | grep "bgzf inflate" | tee $OUT/bgzf_bench.txt
pmc bgzf_mem bgzf_inflate FETCH_SIZE -- python tools/bgzf_bench.py 256
pmc bgzf_sq bgzf_inflate SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -- python tools/bgzf_bench.py 256
pmc bgzf_lds bgzf_inflate SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -- python tools/bgzf_bench.py 256
# the streaming surface's device chain on the whole of config 2 in one add_reads + flush: the read store (fused reads -> LDS histogram ->
# calls kernel) and, for the A/B, the observation log chain it replaced
for path in store log; do
  PISCES_HIP_READ_PATH=$path timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${TAG}_$path -o s -- python tools/store_bench.py > $OUT/store_bench_$path.log 2>&1
  find /tmp/prof_${TAG}_$path -name "*kernel_stats.csv" -exec cp {} $OUT/store_${path}_kernel_stats.csv \;
  grep "store_bench" $OUT/store_bench_$path.log | tail -1
done
pmc store_fetch call_store_tiles FETCH_SIZE -- python tools/store_bench.py
pmc store_write call_store_tiles WRITE_SIZE -- python tools/store_bench.py
pmc store_calib call_store_tiles FETCH_SIZE -- python tools/store_traffic_calibration.py
grep store_traffic_calibration $OUT/pmc_store_calib.log | tee -a $OUT/pmc_store_calib.txt
pmc store_sq call_store_tiles SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -- python tools/store_bench.py
pmc store_lds call_store_tiles SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU -- python tools/store_bench.py
# BASELINE config 3's mix (2000x, SNV + insertion / deletion / MNV candidates), flushed block by block
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${TAG}_c3 -o s -- python tools/config3_host_profile.py 40 > $OUT/config3.log 2>&1
find /tmp/prof_${TAG}_c3 -name "*kernel_stats.csv" -exec cp {} $OUT/config3_kernel_stats.csv \;
grep "rep 2" $OUT/config3.log
# the device chain reads in HBM -> records in HBM (bench.py's roofline_chain): un-profiled spans by events, then the kernels under rocprofv3
for cfgname in "2 --loci 100000 --depth 500" "5 --loci 15000 --depth 5000 --minbq 30"; do
  set -- $cfgname; n=$1; shift
  timeout 300 python tools/chain_bench.py "$@" 2>&1 | grep chain_bench | tee -a $OUT/chain.txt
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${TAG}_chain$n -o s -- python tools/chain_bench.py "$@" > $OUT/chain${n}_prof.log 2>&1
  find /tmp/prof_${TAG}_chain$n -name "*kernel_stats.csv" -exec cp {} $OUT/chain${n}_kernel_stats.csv \;
done
timeout 300 python tools/pair_bench.py 2>&1 | grep pair_bench | tee $OUT/pair_bench.txt
pmc chain_fetch add_fused FETCH_SIZE -- python tools/chain_bench.py --reps 4
pmc chain_write add_fused WRITE_SIZE -- python tools/chain_bench.py --reps 4
# HBM traffic of the hot kernel per launch, for bench.py's roofline.traffic: FETCH_SIZE / WRITE_SIZE come in KiB; on gfx950 FETCH_SIZE
# reports half the bytes of a wide coalesced streaming read (16 B per lane: this kernel's loads), so it is doubled (MI355X_MICROARCH.md, HBM)
python - $OUT $TAG <<'PY'
import json, os, re, sys, time
sys.path.insert(0, os.getcwd())
out, tag = sys.argv[1], sys.argv[2]
def mean(name):
    for line in open(f"{out}/pmc_{name}.txt"):
        m = re.search(r"call_tiles\S* " + name + r" mean per dispatch ([0-9.e+]+)", line)
        if m:
            return float(m.group(1))
    return None
f, w = mean("FETCH_SIZE"), mean("WRITE_SIZE")
line = json.loads(open(f"{out}/bench_n1.json").read().strip().splitlines()[-1])
if f and w:
    prev = json.load(open("profiles/traffic.json"))
    corr = prev.get("fetch_correction", 2.0)
    t = {"loci": line["config"]["loci_per_gpu_per_step"], "depth": line["config"]["depth"], "tile_loci": line["config"]["tile_loci"],
         "hbm_bytes_per_launch": f * 1024 * corr + w * 1024, "fetch_size_kib_raw": f, "fetch_correction": corr, "write_size_kib_raw": w,
         "kernel": "pisces::call_tiles_wave_kernel", "run": f"{tag} {time.strftime('%Y-%m-%d %H:%M:%S')} tools/profile_round.sh",
         "source_hash": __import__("pisces_amd.build", fromlist=["x"]).source_hash(),
         "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE --kernel-include-regex call_tiles, separate passes over bench.py (tools/profile_round.sh)"}
    sf, sw = None, None
    for name, key in (("store_fetch", "FETCH_SIZE"), ("store_write", "WRITE_SIZE")):
        for ln in open(f"{out}/pmc_{name}.txt"):
            m = re.search(r"call_store_tiles\S* " + key + r" mean per dispatch ([0-9.e+]+)", ln)
            if m:
                sf, sw = (float(m.group(1)), sw) if key == "FETCH_SIZE" else (sf, float(m.group(1)))
    if sf and sw:   # tools/store_bench.py's one-batch flush of the same configuration (4 B / lane loads: the same x 2, tools/fetch_calibration.py)
        t["streaming"] = {"kernel": "pisces::call_store_tiles_kernel<2>", "loci": t["loci"], "depth": t["depth"], "fetch_size_kib_raw": sf, "fetch_correction": corr,
                          "write_size_kib_raw": sw, "hbm_bytes_per_launch": sf * 1024 * corr + sw * 1024, "run": t["run"] + " (pmc_store_fetch / pmc_store_write over tools/store_bench.py)"}
    cf, cw = None, None
    for name, key in (("chain_fetch", "FETCH_SIZE"), ("chain_write", "WRITE_SIZE")):
        try:
            for ln in open(f"{out}/pmc_{name}.txt"):
                m = re.search(r"add_fused\S* " + key + r" mean per dispatch ([0-9.e+]+)", ln)
                if m:
                    cf, cw = (float(m.group(1)), cw) if key == "FETCH_SIZE" else (cf, float(m.group(1)))
        except OSError:
            pass
    if cf and cw and "streaming" in t:   # the chain of roofline_chain: the add's launch + the flush's kernel + the compaction's 64 B a record in and out
        add_bytes = cf * 1024 * corr + cw * 1024
        t["chain"] = {"kernels": "pisces::add_fused_kernel + pisces::call_store_tiles_kernel<2> + pisces::gather_direct_kernel", "loci": t["loci"], "depth": t["depth"],
                      "add_fused_fetch_size_kib_raw": cf, "add_fused_write_size_kib_raw": cw, "fetch_correction": corr, "add_fused_hbm_bytes": add_bytes,
                      "hbm_bytes_per_batch": add_bytes + t["streaming"]["hbm_bytes_per_launch"] + 2 * 64.0 * t["loci"],
                      "run": t["run"] + " (pmc_chain_fetch / pmc_chain_write over tools/chain_bench.py; the flush kernel's as `streaming`; the compaction counted at 64 B a record in and out)"}
    json.dump(t, open(f"{out}/traffic.json", "w"), indent=1)
    print("traffic.json:", t["hbm_bytes_per_launch"], "bytes per launch; algorithmic", line["roofline"]["algorithmic_bytes_per_launch"])
PY
