#!/usr/bin/env python3
"""Development: per-tile phase timing of call_store_tiles_kernel (needs a -DPISCES_STORE_TIMING build in PISCES_HIP_LIB)."""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
path = os.path.join(tempfile.gettempdir(), "store_tr.bin")
os.environ["PISCES_HIP_DUMP_TILE_RESULTS"] = path
from pisces_amd import _abi, engine, synth
loci = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
p = synth.make_pileup(loci, 500, seed=7)
whole = synth.reads_of(p, p.base.shape[0], first_amplicon=0)
with engine.HipVariantCaller(_abi.default_config()) as c:
    c.SetReference(p.ref.cpu().numpy())
    for rep in range(3):
        c.AddAlleleCounts(whole)
        try:
            c.Call(None, capacity=4 * loci)
        except Exception as e:   # (the records are not usable in a timing build)
            print("flush:", e)
nt = (loci + 999) // 1000 * 16
t = np.fromfile(path, dtype=np.int32).reshape(-1, 16)[:nt]
t0, ts, tw, te = (t[:, k].astype(np.int64) for k in range(4))
base = t0.min()
t0, ts, tw, te = ((x - base) / 100.0 for x in (t0, ts, tw, te))
nreads = t[:, 4].astype(np.int64)
print(f"tiles {nt}; span {te.max():.1f} us; reads in range p10/p50/p90/max {np.percentile(nreads, [10, 50, 90, 100])}")
for name, v in (("start", t0), ("search done", ts - t0), ("walk dur", tw - ts), ("call dur", te - tw), ("end", te)):
    print(f"{name:12s} us  min/p10/p50/p90/p99/max:", np.round(np.percentile(v, [0, 10, 50, 90, 99, 100]), 1))
for lo, hi in ((0, 600), (600, 2000)):
    m = (nreads >= lo) & (nreads < hi)
    if m.any():
        print(f"tiles with {lo}-{hi} reads: n={int(m.sum())} walk dur p50 {np.percentile((tw - ts)[m], 50):.1f} p90 {np.percentile((tw - ts)[m], 90):.1f} us; per 16 reads p50 {np.percentile(((tw - ts) / np.maximum(nreads, 1) * 16)[m], 50) * 1000:.0f} ns")
print("t(us)  searching  walking  calling")
for g in np.arange(0, te.max() + 1, 4.0):
    print(f"{g:6.0f} {int(((t0 <= g) & (ts > g)).sum()):9d} {int(((ts <= g) & (tw > g)).sum()):8d} {int(((tw <= g) & (te > g)).sum()):8d}")
# per CU (XCC id, SE / SH / CU bits of HW_ID): how much work it was dealt and when its last tile ended
cu = (t[:, 6].astype(np.int64) & 0xF) * 256 + ((t[:, 5].astype(np.int64) >> 8) & 0xFF)   # XCC_ID, HW_ID.cu_id[11:8] | sh_id[12] | se_id[15:13]
ids = np.unique(cu)
work = np.array([nreads[cu == i].sum() for i in ids])
last = np.array([te[cu == i].max() for i in ids])
ntile = np.array([(cu == i).sum() for i in ids])
print(f"CUs seen {len(ids)}; tiles per CU min/p50/max {ntile.min()} {int(np.median(ntile))} {ntile.max()}; fragments per CU min/p10/p50/p90/max "
      f"{np.round(np.percentile(work, [0, 10, 50, 90, 100]))}; max / mean {work.max() / work.mean():.2f}")
for k in sorted(set(ntile.tolist())):
    m = ntile == k
    print(f"  CUs with {k} tiles: {int(m.sum())}; their last end p50 {np.percentile(last[m], 50):.1f} max {last[m].max():.1f} us; fragments p50 {np.percentile(work[m], 50):.0f}")
print(f"last tile's end per CU us min/p10/p50/p90/max {np.round(np.percentile(last, [0, 10, 50, 90, 100]), 1)}; correlation(work, end) {np.corrcoef(work, last)[0, 1]:.2f}")
# wave 0 of every tile: shader-clock cycles of its walk loop by part
it = np.maximum(t[:, 10].astype(np.int64), 1)
for name, k in (("list the next block's pairs (waits for its descriptors)", 7), ("consume 8 units / issue 8 units", 8), ("read the 8 pair entries (LDS round trip)", 9)):
    v = t[:, k].astype(np.int64)
    print(f"{name:48s} cycles per iteration p10/p50/p90 {np.round(np.percentile(v / it, [10, 50, 90]))}; per tile p50 {np.percentile(v, 50):.0f}")
print("iterations per wave p50", np.percentile(it, 50))
# which CU a workgroup went to, in dispatch order (tr[11] = blockIdx.x): XCD = b & 7 by the round-robin of the dispatcher, then?
if t.shape[1] > 11 and t[:, 11].max() > 0:
    b = t[:, 11].astype(np.int64)
    o = np.argsort(b)
    xcc = (t[:, 6].astype(np.int64) & 0xF)[o]
    print("XCC of blocks 0..15:", xcc[:16].tolist(), "; blocks whose XCC != b & 7:", int((xcc != (np.arange(len(o)) & 7)).sum()))
    cu_o = cu[o]
    for x in (0, 3):
        seq = cu_o[x::8] & 0xFF
        print(f"XCD {x}: CU field of its workgroups in order (first 72):", seq[:72].tolist())
        per = 32
        same = int((seq[per:2 * per] == seq[:per]).sum())
        print(f"   second round of 32 on the same CUs as the first: {same} of 32")
