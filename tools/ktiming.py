#!/usr/bin/env python3
"""Development: per-tile phase timing of the hot kernel (needs the -DPISCES_TIMING build)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pisces_amd import _abi, engine, synth
loci = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
dev = torch.device("cuda", 0)
tile = int(sys.argv[2]) if len(sys.argv) > 2 else 64
p = synth.make_pileup(loci, 500, seed=5, device=dev, tile=tile)
nt = p.n_tiles; cap = nt * 256
rec = torch.zeros(cap * 64, dtype=torch.uint8, device=dev)
tr = torch.zeros(nt * 48, dtype=torch.uint8, device=dev)
with engine.HipVariantCaller(_abi.default_config()) as c:
    c.set_timing(True)
    torch.cuda.synchronize()
    for _ in range(3):
        c.call_tiles(p.tuples.data_ptr(), p.tiles.data_ptr(), nt, p.ref.data_ptr(), 1, p.ref_len, rec.data_ptr(), cap, tr.data_ptr(), None)
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
hw = t["n_called"].astype(np.int64) & 0xFFFFFFFF
xcc = t["valid"][:, 0].astype(np.int64) & 0xF
cu = (hw >> 8) & 0xF; sh = (hw >> 12) & 1; se = (hw >> 13) & 7; simd = (hw >> 4) & 3
cuid = ((xcc * 8 + se) * 2 + sh) * 16 + cu
u, inv, cnt = np.unique(cuid, return_inverse=True, return_counts=True)
print(f"distinct CUs used: {len(u)}; tiles per CU min/median/max: {cnt.min()} {int(np.median(cnt))} {cnt.max()}; XCC counts {np.bincount(xcc)}")
print("tiles/CU histogram:", dict(zip(*np.unique(cnt, return_counts=True))))
per = cnt[inv]
for n in np.unique(per):
    m = per == n
    print(f"  tiles on CUs with {n} tiles: n={int(m.sum())} stream dur p50 {np.percentile((t1-t0)[m],50):.1f} p90 {np.percentile((t1-t0)[m],90):.1f}; call dur p50 {np.percentile((t2-t1)[m],50):.1f}")
sid = cuid * 4 + simd
us, invs, cnts = np.unique(sid, return_inverse=True, return_counts=True)
pers = cnts[invs]
for n in np.unique(pers):
    m = pers == n
    print(f"  tiles on SIMDs with {n} tile-waves(0): n={int(m.sum())} stream dur p50 {np.percentile((t1-t0)[m],50):.1f} p90 {np.percentile((t1-t0)[m],90):.1f}")
v = t["valid"].astype(np.int64) / 100.0
for i, name in ((2, "counts + Reference"), (3, "variants + meeting"), (5, "directory")):
    print(f"{name:22s} us p10/p50/p90/max: {np.round(np.percentile(v[:, i], [10, 50, 90, 100]), 2)}")
