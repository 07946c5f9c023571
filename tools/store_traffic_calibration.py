#!/usr/bin/env python3
"""Development: what FETCH_SIZE reports for call_store_tiles_kernel when no byte can be read twice — reads of 32 bases that each lie inside
ONE 64-locus tile (so no two tiles share a read), a known byte count: 1 B per base (its row code: what the round-5 kernel loads) + 16 B per
fragment + the reference window.
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --kernel-include-regex call_store_tiles ... -- python tools/store_traffic_calibration.py
The counter / the known bytes is the factor for this kernel's access pattern (4 lanes x 8 B per pair, any alignment); the same run with
reads of 150 bases (tools/store_bench.py) then says how many times a byte crosses HBM."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pisces_amd import _abi, engine

n_loci, depth, L = 96_000, 512, 32
rng = np.random.default_rng(1)
ref = rng.choice(np.frombuffer(b"ACGT", np.uint8), n_loci + 2000)
# tiles of a block start at block_start + 64 k (1000-locus blocks: 15 tiles of 64 and one of 40): two reads of 32 bases per tile and layer
starts = []
for b in range(n_loci // 1000):
    for t in range(15):
        starts += [1 + 1000 * b + 64 * t, 1 + 1000 * b + 64 * t + 32]
starts = np.repeat(np.array(starts, dtype=np.int32), depth)
n = len(starts)
bases = ref[(starts[:, None] - 1 + np.arange(L)[None, :])].reshape(-1).copy()
batch = _abi.ReadBatch.from_arrays(position=starts, flags=(rng.integers(0, 2, n)).astype(np.uint8), cigar_offset=np.arange(n + 1, dtype=np.int32),
                                   cigar_op=np.full(n, ord("M"), np.uint8), cigar_len=np.full(n, L, np.uint32),
                                   seq_offset=(np.arange(n + 1, dtype=np.int64) * L).astype(np.int32), bases=bases, quals=np.full(n * L, 37, np.uint8))
with engine.HipVariantCaller(_abi.default_config()) as c:
    c.SetReference(ref)
    for rep in range(4):
        c.AddAlleleCounts(batch)
        recs = c.Call(None, capacity=1 << 18, reuse_buffer=True)
print(f"store_traffic_calibration: {n} reads of {L} bases, each inside one tile: {n * L / 1e6:.1f} MB of row codes + {16 * n / 1e6:.1f} MB of fragments "
      f"+ {64 * len(recs) / 1e6:.1f} MB of records written; {len(recs)} records")
