#!/usr/bin/env python3
"""Development: wall time per step of back-to-back call_tiles launches (config 2 by default) on one stream with / without
dispatch-bound timing events, and round-robin over S streams with their own output buffers (tails of one launch overlap the
streaming phase of the next)."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--loci", type=int, default=100_000)
    ap.add_argument("--depth", type=int, default=500)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--ring", type=int, default=6)
    ap.add_argument("--seed", type=int, default=100)
    ap.add_argument("--keep-base", action="store_true")
    ap.add_argument("--null-first", action="store_true", help="stream 0 = torch's current (null) stream")
    ap.add_argument("--dist", action="store_true", help="import torch.distributed first")
    ap.add_argument("--early-handle", action="store_true", help="create the library handle before the inputs, as bench.py does")
    ap.add_argument("--pre", default="", help="comma list of bench.py actions to perform first: totals,probe,cpu")
    a = ap.parse_args()
    import torch
    from pisces_amd import _abi, engine, synth
    dev = torch.device("cuda", 0)
    if a.dist:
        import torch.distributed as dist  # noqa: F401
    early = engine.HipVariantCaller(_abi.default_config()) if a.early_handle else None
    ring = [synth.make_pileup(a.loci, a.depth, seed=a.seed + b, device=dev) for b in range(a.ring)]
    for p in ring:
        if not (a.keep_base and p is ring[0]):
            p.base = p.qual = None
    torch.cuda.empty_cache()
    nt = ring[0].n_tiles
    cap = nt * 256
    with (early if early is not None else engine.HipVariantCaller(_abi.default_config())) as c:
        def run(n_streams, timing):
            streams = [torch.cuda.Stream(dev) for _ in range(n_streams)]
            if a.null_first:
                streams[0] = None   # the handle's own stream
            recs = [torch.zeros(cap * 64, dtype=torch.uint8, device=dev) for _ in range(n_streams)]
            trs = [torch.zeros(nt * 48, dtype=torch.uint8, device=dev) for _ in range(n_streams)]
            torch.cuda.synchronize()   # the fills sit on torch's null stream, the launches do not

            def step(i):
                p = ring[i % a.ring]
                s = i % n_streams
                c.call_tiles(p.tuples.data_ptr(), p.tiles.data_ptr(), p.n_tiles, p.ref.data_ptr(), 1, p.ref_len,
                             recs[s].data_ptr(), cap, trs[s].data_ptr(), streams[s])
            c.set_timing(timing)
            for i in range(10):
                step(i)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(a.steps):
                step(i)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / a.steps
            c.set_timing(0)
            print(f"streams={n_streams} timing_every={timing}: {dt*1e6:.1f} us/step  {a.loci/dt/1e9:.3f} G loci/s", flush=True)
        pre = a.pre.split(",") if a.pre else []
        if "totals" in pre:
            c.device_totals(reset=True)
        if "probe" in pre:
            c.probe_read_bandwidth(1 << 30, 6)
        if "cpu" in pre:
            import numpy as np
            x = torch.zeros(nt * 48, dtype=torch.uint8, device=dev).cpu().numpy()
            np.sum(x)
        for n_streams, timing in [(1, 0), (1, 4), (3, 0)]:
            run(n_streams, timing)


if __name__ == "__main__":
    main()
