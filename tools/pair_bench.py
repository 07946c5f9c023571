#!/usr/bin/env python3
"""Development: the per-block protocol of the streaming surface on BASELINE config 2's batch (96 blocks of 3 500 reads): add + flush,
the flush pair (CallBegin / CallEndView) and the pair with the reads written into the pinned staging buffer — bench.py's
end_to_end_full.per_block / per_block_pair / per_block_pair_staged.     python tools/pair_bench.py [--loci 100000]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--loci", type=int, default=100_000)
    ap.add_argument("--depth", type=int, default=500)
    ap.add_argument("--reps", type=int, default=4)
    a = ap.parse_args()
    from pisces_amd import _abi, engine, synth
    p = synth.make_pileup(a.loci, a.depth, seed=7, with_tuples=False)
    ref = p.ref.cpu().numpy()
    n_amp = p.base.shape[0]
    per_block = [(a0, synth.reads_of(p, min(7, n_amp - a0), first_amplicon=a0)) for a0 in range(0, n_amp, 7)]
    cfg = _abi.default_config()
    out = {}
    with engine.HipVariantCaller(cfg) as c:
        c.SetReference(ref)
        best = None
        for rep in range(a.reps):
            n_rec = 0
            t0 = time.perf_counter()
            for a0, b in per_block:
                c.AddAlleleCounts(b)
                n_rec += len(c.CallView(p.region_start + a0 * synth.READ_LEN - 1))
            n_rec += len(c.CallView(None))
            dt = time.perf_counter() - t0
            if rep > 0:
                best = dt if best is None else min(best, dt)
        out["per_block"] = a.loci / best
        want = n_rec
        for label, stage in (("per_block_pair", False), ("per_block_pair_staged", True)):
            best = None
            for rep in range(a.reps):
                if rep == a.reps - 1:
                    c.HostTime(reset=True)
                n_rec, dt, pending = 0, 0.0, False
                for a0, b in per_block:
                    staged = c.StageReads(b) if stage else b
                    t0 = time.perf_counter()
                    c.AddAlleleCounts(staged)
                    if pending:
                        n_rec += len(c.CallEndView())
                    c.CallBegin(p.region_start + a0 * synth.READ_LEN - 1)
                    pending = True
                    dt += time.perf_counter() - t0
                t0 = time.perf_counter()
                n_rec += len(c.CallEndView())
                c.CallBegin(None)
                n_rec += len(c.CallEndView())
                dt += time.perf_counter() - t0
                assert n_rec == want, (n_rec, want)
                if rep > 0:
                    best = dt if best is None else min(best, dt)
            out[label] = a.loci / best
            ht = c.HostTime(reset=True)
            print(f"pair_bench: {label}: last pass {dt * 1e3:.2f} ms; inside the library: add_reads {ht['add_reads_s'] * 1e3:.2f} ms, flush_begin + flush_end {ht['flush_s'] * 1e3:.2f} ms "
                  f"(of which waiting for the device {ht['flush_wait_s'] * 1e3:.2f} ms); {len(per_block)} blocks", flush=True)
    print("pair_bench: " + ", ".join(f"{k} {v:.3g} loci/s ({a.loci / v / len(per_block) * 1e6:.1f} us a block)" for k, v in out.items()), flush=True)


if __name__ == "__main__":
    main()
