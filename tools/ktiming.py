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
tr = torch.zeros(nt * 16, dtype=torch.uint8, device=dev)
with engine.HipVariantCaller(_abi.default_config()) as c:
    for _ in range(3):
        c.call_tiles(p.tuples.data_ptr(), p.tiles.data_ptr(), nt, p.ref.data_ptr(), 1, p.ref_len, rec.data_ptr(), cap, None, tr.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    ms = c.last_kernel_ms()
t = tr.cpu().numpy().view(_abi.TILE_RESULT_DTYPE)
st, w0, w1, ce = (t[k].astype(np.int64) / 100.0 for k in ("record_begin", "n_records", "n_candidate_loci", "reserved"))
print(f"kernel {ms*1e3:.1f} us; tiles {nt}")
for name, v in (("stream", st), ("wave0 ref done", w0), ("wave1 var done", w1), ("call end", ce)):
    print(f"{name:16s} us  min/p10/p50/p90/p99/max:", np.round(np.percentile(v, [0, 10, 50, 90, 99, 100]), 1))
