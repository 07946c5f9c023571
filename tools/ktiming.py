#!/usr/bin/env python3
"""Development: per-tile phase timing of call_tiles_kernel (needs the -DPISCES_TIMING build)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pisces_amd import _abi, engine, synth
loci = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
dev = torch.device("cuda", 0)
p = synth.make_pileup(loci, 500, seed=5, device=dev)
nt = p.n_tiles; cap = nt * 256
rec = torch.zeros(cap * 64, dtype=torch.uint8, device=dev)
tr = torch.zeros(nt * 48, dtype=torch.uint8, device=dev)
with engine.HipVariantCaller(_abi.default_config()) as c:
    for _ in range(3):
        c.call_tiles(p.tuples.data_ptr(), p.tiles.data_ptr(), nt, p.ref.data_ptr(), 1, p.ref_len, rec.data_ptr(), cap, tr.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    ms = c.last_kernel_ms()
t = tr.cpu().numpy().view(_abi.TILE_RESULT_DTYPE)
t0, t1, t2 = (t[k].astype(np.int64) for k in ("record_begin", "n_records", "n_candidate_loci"))
base = t0.min()
t0, t1, t2 = (t0 - base) / 100.0, (t1 - base) / 100.0, (t2 - base) / 100.0   # microseconds
print(f"kernel {ms*1e3:.1f} us; tiles {nt}; span of stamps {t2.max():.1f} us")
for name, v in (("start", t0), ("stream end", t1), ("call end", t2), ("stream dur", t1 - t0), ("call dur", t2 - t1)):
    print(f"{name:12s} us  min/p10/p50/p90/p99/max:", np.round(np.percentile(v, [0, 10, 50, 90, 99, 100]), 1))
print("t(us)   streaming  calling")
for g in np.arange(0, t2.max() + 1, 4.0):
    print(f"{g:6.0f} {int(((t0 <= g) & (t1 > g)).sum()):9d} {int(((t1 <= g) & (t2 > g)).sum()):8d}")
r1, r2 = t0 < 5, t0 >= 5
for name, m in (("round 1", r1), ("round 2", r2)):
    if m.any():
        print(f"{name}: n={int(m.sum())} stream dur p10/p50/p90: {np.round(np.percentile((t1 - t0)[m], [10, 50, 90]), 1)}  call dur p10/p50/p90: {np.round(np.percentile((t2 - t1)[m], [10, 50, 90]), 1)}")
v = t["valid"].astype(np.int64) / 100.0
for i, name in enumerate(("wave0 reference lanes", "wait waves 1,2", "wait wave 1 assembly", "tail (masks, directory)")):
    print(f"{name:26s} us p10/p50/p90/p99: {np.round(np.percentile(v[:, i], [10, 50, 90, 99]), 2)}")
