#!/usr/bin/env python3
"""Development: where the host's time goes in one contig of BASELINE config 4 (intervals of 150 bp, 200x, SNV + indel): the phases of the
flushes (PISCES_HIP_HOST_PROFILE=1 prints them when the handle goes)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["PISCES_HIP_HOST_PROFILE"] = "1"
from pisces_amd import _abi, config4, engine
n_iv = int(sys.argv[1]) if len(sys.argv) > 1 else 6000
cfg = _abi.default_config(emit_zero_coverage_refs=1)
job = config4.make_contig(3, n_iv, depth=200, device="cuda:0")
for rep in range(2):
    t0 = time.perf_counter()
    recs, _, stats, owned = config4.run_piece(engine, cfg, job, device=0, with_alleles=False)
    dt = time.perf_counter() - t0
    print(f"rep {rep}: {n_iv} intervals = {n_iv * config4.INTERVAL} loci, {job['batch'].n_reads} reads: {dt*1e3:.1f} ms -> {n_iv * config4.INTERVAL / dt:.3g} loci/s; "
          f"library: {stats['host_time']}", flush=True)
