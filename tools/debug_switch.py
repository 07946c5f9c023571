"""Development: default product vs oracle for one seed of test_every_switch_of_the_library_leaves_the_records_alone."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pisces_amd import _abi, engine
from tests import orc
from tests.test_gpu_parity import _mnv_reads
from tests.test_read_store import random_reads, env

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
only_exotic = len(sys.argv) > 2
rng = np.random.default_rng(7100 + seed)
ref = bytes(rng.choice(list(b"ACGT"), 3300).astype(np.uint8))
reads = _mnv_reads(rng, bytearray(ref), int(rng.integers(1200, 3000)), region=(50, 3100), snv_rate=float(rng.choice([0.002, 0.006])))
if seed % 3 == 0:
    reads += random_reads(rng, 300, 60, 2900, exotic=False, sort=False)
reads.sort(key=lambda r: r["pos"])
ploidy = int(rng.choice([0, 0, 0, 1, 2]))
kw = dict(call_mnvs=int(rng.integers(0, 2)), max_mnv_length=int(rng.choice([2, 3])), max_gap_between_mnv=int(rng.choice([0, 1])),
          collapse=int(rng.integers(0, 2)), include_reference_calls=int(rng.integers(0, 2)), ploidy=ploidy,
          noise_model=int(rng.choice([0, 0, 1])), strand_bias_model=int(rng.choice([1, 1, 2])),
          min_frequency=0.2 if ploidy else float(rng.choice([0.01, 0.05])))
if ploidy:
    kw.update(variant_freq_filter=0.2, low_gq_filter=30, max_genotype_qscore=1000)
print(seed, kw)
cfg = _abi.default_config(**kw)
cuts = sorted(set(int(x) for x in rng.integers(1, len(reads), 3)) | {len(reads)})
ups = [int(x) for x in sorted(rng.integers(600, 3000, len(cuts) - 1))] + [None]
with engine.HipVariantCaller(cfg) as c:
    c.SetReference(ref)
    rows, alleles, a0 = [], [], 0
    for cut, up in zip(cuts, ups):
        c.AddAlleleCounts(_abi.ReadBatch(reads[a0:cut])); a0 = cut
        if up is not None: up = min(up, reads[cut - 1]["pos"] - 1)
        r, a = c.CallWithAlleles(up, capacity=1 << 15)
        rows.append(r); alleles += a
    got = np.concatenate(rows)
schedule = [min(up, reads[cut - 1]["pos"] - 1) for cut, up in zip(cuts, ups) if up is not None]
print("cuts", cuts, "schedule", schedule, "last read pos at cuts", [reads[c - 1]["pos"] for c in cuts])
exp, exp_alleles, exp_called = orc.run_reads_schedule(_abi.ReadBatch(reads), np.frombuffer(ref, np.uint8), 1, len(ref), cfg, schedule)
print(len(got), len(exp))
gd = {(int(r["position"]), a): (int(r["allele_support"]), int(r["total_coverage"])) for r, a in zip(got, alleles)}
od = {(int(r["position"]), a): (int(r["allele_support"]), int(r["total_coverage"])) for r, a in zip(exp, exp_alleles)}
n = 0
for k in sorted(set(gd) | set(od)):
    if gd.get(k) != od.get(k):
        print(k, "product", gd.get(k), "oracle", od.get(k)); n += 1
        if n > 20: break
