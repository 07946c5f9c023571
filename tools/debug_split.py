"""Development: the records on which the split form and the legacy form of MNV calling differ, for one seed of
tests/test_gpu_parity.py::test_split_form_equals_the_legacy_form_on_random_schedules, with the oracle's answer beside them."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pisces_amd import _abi, engine
from tests import orc
from tests.test_gpu_parity import _mnv_reads
from tests.test_read_store import env

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 3
rng = np.random.default_rng(9000 + seed)
ref = bytes(rng.choice(list(b"ACGT"), 4200).astype(np.uint8))
reads = _mnv_reads(rng, bytearray(ref), int(rng.integers(1500, 4000)), region=(50, 4000), snv_rate=float(rng.choice([0.002, 0.006])))
for i, r in enumerate(reads):
    if i % 37 == 0 and r["cigar"] == [("M", 100)]:
        k = int(rng.integers(10, 80))
        r["cigar"] = [("=", k), ("X", 2), ("M", 100 - k - 2)]
reads.sort(key=lambda r: r["pos"])
kw = dict(call_mnvs=1, max_mnv_length=int(rng.choice([2, 3, 5])), max_gap_between_mnv=int(rng.choice([0, 1, 2])), collapse=int(rng.integers(0, 2)),
          include_reference_calls=int(rng.integers(0, 2)), min_frequency=float(rng.choice([0.01, 0.05])))
cfg = _abi.default_config(**kw)
intervals = [(200, 1700), (1900, 3100), (3300, 3900)] if seed % 4 == 1 else None
forced = [(1500, chr(ref[1499]), "A" if chr(ref[1499]) != "A" else "C"),
          (2600, ref[2599:2601].decode(), "TT" if ref[2599] != ord("T") and ref[2600] != ord("T") else "GG" if ref[2599] != ord("G") and ref[2600] != ord("G") else "CC")] if seed % 4 == 2 else None
cuts = sorted(set(int(x) for x in rng.integers(0, len(reads), 4)) | {len(reads)})
ups = [int(x) for x in sorted(rng.integers(600, 3900, len(cuts) - 1))] + [None]
print("seed", seed, kw, "cuts", cuts, "ups", ups, "intervals", intervals, "forced", forced)
out = []
real_ups = []
for split in (None, 0):
    with env(PISCES_HIP_MNV_SPLIT=split):
        with engine.HipVariantCaller(cfg) as c:
            c.SetReference(ref)
            if intervals: c.SetIntervals(intervals)
            if forced: c.SetForcedAlleles(forced)
            rows, alleles, a0 = [], [], 0
            for cut, up in zip(cuts, ups):
                c.AddAlleleCounts(_abi.ReadBatch(reads[a0:cut]))
                a0 = cut
                if up is not None:
                    up = min(up, reads[cut - 1]["pos"] - 1) if cut else up
                if split is None: real_ups.append(up)
                r, a = c.CallWithAlleles(up, capacity=1 << 15)
                rows.append(r); alleles += a
            out.append((np.concatenate(rows), alleles, c.Stats()))
print("real ups", real_ups)
def key(rows, alleles):
    d = {}
    for r, a in zip(rows, alleles):
        d[(int(r["position"]), a[0], a[1])] = (int(r["allele_support"]), int(r["total_coverage"]), int(r["reference_support"]), int(r["variant_qscore"]), int(r["filter_bits"]), int(r["info"]) & 15,
                                                tuple(int(x) for x in r["support_by_dir"]))
    return d
# CallWithAlleles gives allele strings only for candidate rows? build (ref, alt) for every row
def all_alleles(rows, alleles):
    base = "AGCTND"
    it = iter(alleles)
    res = []
    for r in rows:
        cat = (int(r["info"]) >> 4) & 7
        res.append(None)
    return res
g, w = out[0], out[1]
print("rows", len(g[0]), len(w[0]), "alleles", len(g[1]), len(w[1]), "stats", g[2], w[2])
def rowkey(r):
    return (int(r["position"]), (int(r["info"]) >> 4) & 7, (int(r["info"]) >> 7) & 7, (int(r["info"]) >> 10) & 7)
gd, wd = {}, {}
for name, (rows, _, _), d in (("split", g, gd), ("legacy", w, wd)):
    for r in rows:
        d.setdefault(rowkey(r), []).append((int(r["allele_support"]), int(r["total_coverage"]), int(r["reference_support"]), int(r["variant_qscore"]), int(r["filter_bits"]), int(r["info"]) & 15, tuple(int(x) for x in r["support_by_dir"])))
keys = sorted(set(gd) | set(wd))
n = 0
for k in keys:
    if gd.get(k) != wd.get(k):
        print(k, "split", gd.get(k), "legacy", wd.get(k))
        n += 1
        if n > 25: break
sched = [u for u in real_ups if u is not None]
if not intervals:
    exp, exp_alleles, exp_called = orc.run_reads_schedule(_abi.ReadBatch(reads), np.frombuffer(ref, np.uint8), 1, len(ref), cfg, sched, forced=forced or ())
    od = {}
    for r in exp:
        od.setdefault(rowkey(r), []).append((int(r["allele_support"]), int(r["total_coverage"]), int(r["reference_support"]), int(r["variant_qscore"]), int(r["filter_bits"]), int(r["info"]) & 15, tuple(int(x) for x in r["support_by_dir"])))
    print("oracle rows", len(exp), "== split", od == gd, "== legacy", od == wd)
    n = 0
    for k in sorted(set(od) | set(gd) | set(wd)):
        if not (od.get(k) == gd.get(k) == wd.get(k)):
            print(k, "oracle", od.get(k), "split", gd.get(k), "legacy", wd.get(k)); n += 1
            if n > 25: break
