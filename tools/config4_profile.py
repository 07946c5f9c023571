"""Where a piece of BASELINE config 4 spends its time (development): python tools/config4_profile.py [contig] [device|host].
The steps of config4.run_piece timed one by one on the whole contig as one piece."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pisces_amd import _abi, config4, engine, shard   # noqa: E402


def main():
    contig = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    mode = sys.argv[2] if len(sys.argv) > 2 else "host"
    sizes = config4.contig_intervals(200_000)
    cfg = _abi.default_config(emit_zero_coverage_refs=1)
    job = config4.make_contig(contig, sizes[contig], depth=200, device="cuda:0")
    t = {}

    def lap(name, t0):
        torch.cuda.synchronize()
        t[name] = t.get(name, 0.0) + time.perf_counter() - t0

    for rep in range(2):
        t.clear()
        t_all = time.perf_counter()
        t0 = time.perf_counter()
        starts, ends = job["starts"], job["ends"]
        pos = job["arrays"][0].astype(np.int64)
        lo, hi = 1, len(job["ref"])
        keep = (ends >= lo) & (starts <= hi)
        ivs = list(zip(np.maximum(starts[keep], lo).tolist(), np.minimum(ends[keep], hi).tolist()))
        idx, owner = shard.reads_for_shard(pos, job["read_end"], lo, hi, 166)
        lap("python: intervals + reads_for_shard", t0)
        t0 = time.perf_counter()
        c = engine.HipVariantCaller(cfg, device=0)
        lap("create", t0)
        t0 = time.perf_counter(); c.SetReference(job["ref"]); lap("SetReference", t0)
        t0 = time.perf_counter(); c.SetIntervals(ivs); c.SetOwnedRange(lo, hi); lap("SetIntervals", t0)
        i0, i1 = int(idx[0]), int(idx[-1]) + 1
        rows = 0
        for a in range(i0, i1, 400_000):
            b = min(a + 400_000, i1)
            t0 = time.perf_counter(); rb = config4.read_batch_range(job["arrays"], a, b); lap("python: read_batch_range", t0)
            if mode == "device":
                t0 = time.perf_counter(); db = engine.DeviceReadBatch.from_host(rb); db.synchronize(); lap("(untimed in device mode) upload", t0)
                t0 = time.perf_counter(); c.AddDeviceReads(db); lap("AddDeviceReads", t0)
            else:
                t0 = time.perf_counter(); c.AddAlleleCounts(rb); lap("AddAlleleCounts", t0)
            if b < i1:
                t0 = time.perf_counter(); v = c.CallView(int(pos[b]) - 1); lap("CallView", t0)
                t0 = time.perf_counter(); rows += len(v); p = v["position"]; _ = int((np.diff(p) != 0).sum()); lap("python: count rows", t0)
        t0 = time.perf_counter(); v = c.CallView(None); rows += len(v); lap("CallView", t0)
        ht = c.HostTime()
        t0 = time.perf_counter(); c.close(); lap("close", t0)
        total = time.perf_counter() - t_all
        print(f"rep {rep}: contig {contig}, {sizes[contig] * 150} loci, {i1 - i0} reads, {rows} rows, {total * 1e3:.1f} ms "
              f"({sizes[contig] * 150 / total / 1e6:.2f} M loci/s); library: {ht}")
        for k, v in sorted(t.items(), key=lambda kv: -kv[1]):
            print(f"    {v * 1e3:9.2f} ms  {k}")


if __name__ == "__main__":
    main()
