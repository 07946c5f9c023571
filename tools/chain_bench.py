#!/usr/bin/env python3
"""Development: reads in DEVICE memory -> records (pisces_hip_add_device_reads + pisces_hip_flush_view) on one batch, the chain
`roofline_chain` of bench.py times.
    python tools/chain_bench.py [--loci 100000] [--depth 500] [--reps 8] [--minbq 20]
Run it under `rocprofv3 --kernel-trace --stats` for the per-kernel times."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--loci", type=int, default=100_000)
    ap.add_argument("--depth", type=int, default=500)
    ap.add_argument("--reps", type=int, default=8)
    ap.add_argument("--minbq", type=int, default=20)
    a = ap.parse_args()
    import torch
    from pisces_amd import _abi, engine, synth
    p = synth.make_pileup(a.loci, a.depth, seed=7, device="cuda", with_tuples=False)
    ref = p.ref.cpu().numpy()
    whole = synth.reads_of(p, p.base.shape[0], first_amplicon=0)
    d = engine.DeviceReadBatch.from_host(whole, "cuda:0")
    cfg = _abi.default_config(min_base_call_quality=a.minbq)
    with engine.HipVariantCaller(cfg) as c:
        c.SetReference(ref)
        c.SetChainTiming(True)
        best, chains = None, []
        for rep in range(a.reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            try:
                c.AddDeviceReads(d)
                t1 = time.perf_counter()
                n = len(c.CallView(None))
                t2 = time.perf_counter()
            except Exception as e:   # (ablation builds refuse the batch or make no usable records: the kernels still ran)
                print("chain_bench:", str(e)[:120])
                t1 = t2 = time.perf_counter()
                n = 0
            if rep and (best is None or t2 - t0 < best[0]):
                best = (t2 - t0, t1 - t0, t2 - t1)
            if rep and n:
                chains.append(c.ChainTime())
    chain = min(chains, key=sum) if chains else (0.0, 1e-9)
    chain = f"add {chain[0]*1e3:.1f} us + flush {chain[1]*1e3:.1f} us = {sum(chain)*1e3:.1f} us -> {(2.0 * whole.n_bases + 64.0 * n) / (sum(chain) * 1e-3) / 8e12:.3f} of 8 TB/s"
    nbytes = 2.0 * whole.n_bases + 64.0 * n
    print(f"chain_bench: {a.loci} loci x {a.depth}x, {whole.n_reads} reads, {n} records: best of {a.reps - 1}: {best[0]*1e6:.1f} us "
          f"(add_device_reads {best[1]*1e6:.1f}, flush_view {best[2]*1e6:.1f}) -> {a.loci/best[0]:.3g} loci/s, {nbytes/best[0]/8e12:.3f} of 8 TB/s by the wall clock"
          + (f"; device chain {chain}" if chain else ""), flush=True)


if __name__ == "__main__":
    main()
