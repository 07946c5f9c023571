#!/usr/bin/env python3
"""Development: one add + ONE flush of 30 blocks of BASELINE config 3's mix (30 000 loci x 2000x = 400 000 reads, MNV calling on) with the
host phases of the flush printed (PISCES_HIP_HOST_PROFILE=1): what bench.py --config 3 repeats 34 times.  `device` as the first argument:
the reads are handed over in device memory (pisces_hip_add_device_reads)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["PISCES_HIP_HOST_PROFILE"] = "1"
from pisces_amd import _abi, engine, synth
device_fed = len(sys.argv) > 1 and sys.argv[1] == "device"
seed, depth, amps = 33, 2000, 200
cfg = _abi.default_config(call_mnvs=1, max_mnv_length=3, max_gap_between_mnv=1)
n_loci = amps * synth.READ_LEN
ref = synth.reference_of(n_loci, seed, device="cuda")
p = synth.make_pileup(n_loci, depth, seed=seed, device="cuda", first_locus=0, total_loci=n_loci, with_tuples=False)
batch, planted = synth.mixed_reads(p, seed)
dbatch = engine.DeviceReadBatch.from_host(batch, "cuda:0") if device_fed else None
with engine.HipVariantCaller(cfg) as c:
    c.SetReference(ref)
    for rep in range(4):
        if rep == 1:
            c.HostTime(reset=True)
        t0 = time.perf_counter()
        c.AddDeviceReads(dbatch) if device_fed else c.AddAlleleCounts(batch)
        t1 = time.perf_counter()
        n = len(c.CallView(None))
        t2 = time.perf_counter()
        print(f"rep {rep}: add {1e3*(t1-t0):.2f} ms, ONE flush of {n_loci} loci {1e3*(t2-t1):.2f} ms, {n} records, {batch.n_reads} reads", flush=True)
    print(c.HostTime())
