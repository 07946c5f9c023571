import time, numpy as np, sys, os
sys.path.insert(0, os.getcwd())
from pisces_amd import _abi, engine
from tests.test_read_store import _eqx_reads
from tests import orc
rng = np.random.default_rng(1)
ref = bytes(rng.choice(list(b"ACGT"), 3300).astype(np.uint8))
t0 = time.time(); reads = _eqx_reads(rng, ref, n=30000); print("made", len(reads), "reads in %.1f s" % (time.time() - t0))
batch = _abi.ReadBatch(reads)
cfg = _abi.default_config(call_mnvs=0)
for rep in range(3):
    with engine.HipVariantCaller(cfg) as c:
        c.SetReference(ref)
        t0 = time.time(); c.AddAlleleCounts(batch); t1 = time.time()
        rows, alleles = c.CallWithAlleles(None, capacity=1 << 16); t2 = time.time()
    print("add %.1f ms, flush %.1f ms, %d rows" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, len(rows)))
t0 = time.time()
exp, ea, called = orc.run_reads_schedule(batch, np.frombuffer(ref, np.uint8), 1, len(ref), cfg, [])
print("oracle %.1f s" % (time.time() - t0), len(exp), ea == alleles, (exp["allele_support"] == rows["allele_support"]).all())
