#!/usr/bin/env python3
"""Development: BASELINE config 2 steps as plain launches vs one HIP graph (pisces_hip_call_tiles_graph_build / _launch)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pisces_amd import _abi, engine, synth
dev = torch.device("cuda", 0)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
with engine.HipVariantCaller(_abi.default_config()) as c:
    tile = c.balanced_tile_loci(100_000)
    ring = [synth.make_pileup(100_000, 500, seed=100 + b, device=dev, tile=tile) for b in range(4)]
    nt = ring[0].n_tiles; cap = nt * 256
    rec = torch.zeros(cap * 64, dtype=torch.uint8, device=dev); tr = torch.zeros(nt * 48, dtype=torch.uint8, device=dev)
    st = None   # the handle's own stream (fills are synchronised before the first launch)
    torch.cuda.synchronize()
    def batch(i):
        p = ring[i % 4]
        return (p.tuples.data_ptr(), p.tiles.data_ptr(), p.n_tiles, p.ref.data_ptr(), p.ref_start, p.ref_len, rec.data_ptr(), cap, tr.data_ptr())
    for i in range(8):
        c.call_tiles(*batch(i)[:6], rec.data_ptr(), cap, tr.data_ptr(), st)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        c.call_tiles(*batch(i)[:6], rec.data_ptr(), cap, tr.data_ptr(), st)
    torch.cuda.synchronize()
    plain = (time.perf_counter() - t0) / steps
    gid = c.call_tiles_graph_build([batch(i) for i in range(steps)])
    c.call_tiles_graph_launch(gid); c.synchronize()
    best = 1e9
    for rep in range(5):
        t0 = time.perf_counter()
        c.call_tiles_graph_launch(gid)
        c.synchronize()
        best = min(best, (time.perf_counter() - t0) / steps)
    print(f"plain launches {plain*1e6:.1f} us/step; graph {best*1e6:.1f} us/step")
