#!/usr/bin/env python3
"""Regenerates tests/golden/synthetic_small.npz: a seeded synthetic pileup (observation tuples by tile + reference bases) and the
called alleles the ORACLE produces for it with the reference's default configuration.  Data only; the fixture lets the GPU
parity test run against committed expected outputs.  Run from the repo root:  python tests/golden/make_synthetic_golden.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from pisces_amd import _abi, synth   # noqa: E402
from tests import orc                # noqa: E402

p = synth.make_pileup(n_loci=960, depth=120, seed=424242, snv_every=23, snv_offset=5, vaf_range=(0.01, 0.6), p_lowq=0.04)
cfg = _abi.default_config()
pos, tup = synth.observations_of(p)
ref = p.ref.cpu().numpy()
exp, nloci = orc.run_observations(pos, tup, ref, p.region_start, p.n_loci, cfg)
out = os.path.join(ROOT, "tests", "golden", "synthetic_small.npz")
np.savez_compressed(out, positions=pos.astype(np.int32), tuples=tup.astype(np.uint32), ref=ref, region_start=np.int32(p.region_start),
                    n_loci=np.int32(p.n_loci), expected=exp.view(np.uint8), n_candidate_loci=np.int64(nloci))
print(out, os.path.getsize(out), "bytes;", len(exp), "records;", int((_abi.info_category(exp["info"]) == _abi.CAT_SNV).sum()), "SNVs")
