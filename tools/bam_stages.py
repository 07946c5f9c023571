#!/usr/bin/env python3
"""Development: where the time of the BAM-bytes surface goes, call by call (bgzf_scan, pisces_hip_bam_decode,
pisces_hip_add_decoded_reads, pisces_hip_flush), on a synthetic BAM of random 150-base reads (compresses ~2.6x like real data).
    python tools/bam_stages.py [reads]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from tools.bam_bench import make_bam
from tools.bgzf_bench import make_bgzf


def main():
    from pisces_amd import _abi, engine
    n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
    stream = make_bam(n_reads)
    data = make_bgzf(stream, 6)
    arr = np.frombuffer(data, dtype=np.uint8)
    with engine.HipVariantCaller(_abi.default_config()) as c:
        c.SetReference(np.full(1000 + n_reads // 3 + 3000, ord("A"), np.uint8))
        for rep in range(4):
            t0 = time.perf_counter()
            engine.bgzf_scan(arr)
            t1 = time.perf_counter()
            counts = c.bam_decode(data, 0)
            t2 = time.perf_counter()
            c.AddDecodedReads()
            t3 = time.perf_counter()
            recs = c.Call(None, capacity=1 << 20, reuse_buffer=True)
            t4 = time.perf_counter()
            print(f"rep {rep}: {n_reads} reads, {len(stream)/1e6:.0f} MB from {len(data)/1e6:.0f} MB: scan {1e3*(t1-t0):.2f} ms, bam_decode (incl. its own scan) "
                  f"{1e3*(t2-t1):.2f} ms, add_decoded_reads {1e3*(t3-t2):.2f} ms, flush {1e3*(t4-t3):.2f} ms ({len(recs)} records; record chain {counts['chain']})", flush=True)
        assert counts["reads"] == n_reads


if __name__ == "__main__":
    main()
