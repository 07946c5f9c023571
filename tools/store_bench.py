#!/usr/bin/env python3
"""Development: the streaming surface's device chain reads -> records on a whole configuration in ONE add_reads + flush (so that kernel
times, not launch latencies, are what shows), for the read store (default) or the observation log (PISCES_HIP_READ_PATH=log).
    python tools/store_bench.py [--loci 100000] [--depth 500] [--reps 5]
Run it under `rocprofv3 --kernel-trace --stats` for the per-kernel times (tools/profile_round.sh does)."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--loci", type=int, default=100_000)
    ap.add_argument("--depth", type=int, default=500)
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    from pisces_amd import _abi, engine, synth
    p = synth.make_pileup(a.loci, a.depth, seed=7)
    ref = p.ref.cpu().numpy()
    whole = synth.reads_of(p, p.base.shape[0], first_amplicon=0)
    cfg = _abi.default_config()
    path = os.environ.get("PISCES_HIP_READ_PATH", "store")
    with engine.HipVariantCaller(cfg) as c:
        c.SetReference(ref)
        best = None
        for rep in range(a.reps):
            t0 = time.perf_counter()
            c.AddAlleleCounts(whole)
            t1 = time.perf_counter()
            try:
                recs = c.Call(None, capacity=4 * a.loci)
            except Exception as e:   # (ablation builds make no usable records)
                recs = []
            t2 = time.perf_counter()
            if best is None or t2 - t0 < best[0]:
                best = (t2 - t0, t1 - t0, t2 - t1)
        n_obs = int(np.diff(whole.seq_offset).sum())
    print(f"store_bench[{path}]: {a.loci} loci x {a.depth}x, {whole.n_reads} reads, {n_obs} observations, {len(recs)} records: best of {a.reps}: "
          f"{best[0]*1e3:.2f} ms (add_reads {best[1]*1e3:.2f}, flush {best[2]*1e3:.2f}) -> {a.loci/best[0]:.3g} loci/s", flush=True)


if __name__ == "__main__":
    main()
