#!/usr/bin/env python3
"""Development: throughput of the STREAMING surface (host reads -> pisces_hip_add_reads -> pisces_hip_flush), i.e. the
PCIe-inclusive, host-fed rate of the drop-in boundary.  Not bench.py's metric."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--loci", type=int, default=30000)
    ap.add_argument("--depth", type=int, default=500)
    ap.add_argument("--amplicons-per-batch", type=int, default=7)   # ~1050 loci: one block of reads per add_reads call
    a = ap.parse_args()
    from pisces_amd import _abi, engine, synth
    p = synth.make_pileup(a.loci, a.depth, seed=7)
    ref = p.ref.cpu().numpy()
    A = p.base.shape[0]
    batches = [(a0, synth.reads_of(p, a.amplicons_per_batch, first_amplicon=a0)) for a0 in range(0, A, a.amplicons_per_batch)]
    n_reads = sum(b.n_reads for _, b in batches)
    cfg = _abi.default_config()
    for rep in range(2):
        with engine.HipVariantCaller(cfg) as c:
            c.SetReference(ref)
            t_add = t_call = 0.0
            n_rec = 0
            t0 = time.perf_counter()
            for a0, b in batches:
                t1 = time.perf_counter()
                c.AddAlleleCounts(b)
                t2 = time.perf_counter()
                recs = c.Call(p.region_start + a0 * synth.READ_LEN - 1)   # LastClearedPosition: everything before this batch
                t3 = time.perf_counter()
                t_add += t2 - t1; t_call += t3 - t2; n_rec += len(recs)
            t1 = time.perf_counter()
            recs = c.Call(None)
            t_call += time.perf_counter() - t1
            n_rec += len(recs)
            dt = time.perf_counter() - t0
        print(f"host-fed: {a.loci} loci x {a.depth}x, {n_reads} reads in {len(batches)} add_reads calls: {dt*1e3:.1f} ms "
              f"(add_reads {t_add*1e3:.1f} ms, flush {t_call*1e3:.1f} ms) -> {a.loci/dt:.3g} loci/s, {n_rec} records", flush=True)


if __name__ == "__main__":
    main()
