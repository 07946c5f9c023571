"""The reference's own unit-test known answers through the PRODUCT (libpisceship.so), not through the oracle.

tests/test_oracle_golden.py pins the oracle (a CPU restatement, test infrastructure) with these tables; the product's host collapser,
MNV reallocator, per-locus genotypers and candidate kernel were until now only compared WITH the oracle — and product and oracle are
restatements of the same C# by one author, so a shared misreading would pass.  Here the expectations come from the JSON files under
tests/golden/ (transcribed from the reference's tests, file:line in each), and the code that runs is the library's:

  * CPU (no GPU needed: host code of the product behind its C ABI)
      mnv_reallocator_cases.json   MNVReallocatorTests.cs:18-662            -> pisces_hip_reallocate_failed_mnvs
      diploid_cases.json           GenotypeCalculatorTest.cs, DiploidGenotypeQualityCalculatorTests.cs, HaploidGenotypeCalculatorTests.cs
                                                                             -> pisces_hip_set_genotypes, pisces_hip_diploid_genotype_qscore
  * GPU (through pisces_hip_add_observations + pisces_hip_add_candidates + pisces_hip_flush_ex: counts injected cell by cell as the
    tests' mock IAlleleSource returns them, candidates as the tests build them)
      collapser_cases.json         VariantCollapserTests.cs:18-204, 869-1033  -> the flush's collapser, frequencies from the device's counts
      coverage_spanning.json       CoverageCalculatorTests.cs:69-702          -> the candidate kernel's coverage / the tile kernels' point coverage
      caller_matrix.json           VariantCallerTests.cs:27-277               -> IsCallable by coverage / frequency / q-score, reference pruning
"""
import ctypes as C
import json
import os

import numpy as np
import pytest

from pisces_amd import _abi

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
load = lambda name: json.load(open(os.path.join(G, name)))
ALLELE = {"A": 0, "G": 1, "C": 2, "T": 3, "N": 4, "Deletion": 5, "Del": 5}
CAT = {"Snv": _abi.CAT_SNV, "Insertion": _abi.CAT_INSERTION, "Deletion": _abi.CAT_DELETION, "Mnv": _abi.CAT_MNV, "Reference": _abi.CAT_REFERENCE}
GT_CODE = {"HeterozygousAlt1Alt2": 0, "Alt12LikeNoCall": 1, "HeterozygousAltRef": 2, "HomozygousAlt": 3, "HomozygousRef": 4,
           "RefLikeNoCall": 5, "AltLikeNoCall": 6, "RefAndNoCall": 7, "AltAndNoCall": 8}


# ============================================================================================ CPU: the host half as functions
def _candidates(items):
    arr = (_abi.PiscesCandidate * max(1, len(items)))()
    pool = bytearray()
    for i, d in enumerate(items):
        c = arr[i]
        c.position, c.category = d["position"], d["category"]
        c.ref_len, c.alt_len = len(d["ref"]), len(d["alt"])
        for k in range(3):
            c.support_by_dir[k] = d["dirs"][k]
        c.allele_offset = len(pool)
        pool += d["ref"].encode() + d["alt"].encode()
    return arr, np.frombuffer(bytes(pool) or b"\0", dtype=np.uint8).copy(), len(pool)


def _product_reallocate(failed, callable_, block_max):
    from pisces_amd import _native
    both = list(failed) + list(callable_)
    arr, pool, nb = _candidates(both)
    cap = len(callable_) + 64 * max(len(failed), 1)
    out_c, out_o = (_abi.PiscesCandidate * cap)(), (_abi.PiscesCandidate * cap)()
    out_pool = np.zeros(1 << 16, dtype=np.uint8)
    nc, no, nbytes = C.c_int64(0), C.c_int64(0), C.c_int64(0)
    f_ptr = C.cast(arr, C.c_void_p)
    c_ptr = C.c_void_p(C.addressof(arr) + len(failed) * C.sizeof(_abi.PiscesCandidate))
    rc = _native.lib.pisces_hip_reallocate_failed_mnvs(f_ptr, len(failed), c_ptr, len(callable_), pool.ctypes.data, nb,
                                                       -1 if block_max is None else int(block_max), C.cast(out_c, C.c_void_p), cap, C.byref(nc),
                                                       C.cast(out_o, C.c_void_p), cap, C.byref(no), out_pool.ctypes.data, len(out_pool), C.byref(nbytes))
    assert rc == 0, rc

    def rows(a, n):
        out = []
        for i in range(n):
            c = a[i]
            o = c.allele_offset
            out.append({"position": c.position, "ref": bytes(out_pool[o:o + c.ref_len]).decode(), "alt": bytes(out_pool[o + c.ref_len:o + c.ref_len + c.alt_len]).decode(),
                        "dirs": [c.support_by_dir[k] for k in range(3)], "category": c.category})
        return out
    return rows(out_c, nc.value), rows(out_o, no.value)


def _as_the_product_holds_it(d):
    """The reference's tests build CalledAlleles whose AlleleSupport and SupportByDirection are set independently (a callable allele with
    AlleleSupport 5 and no directional support).  The product keeps ONE number: AlleleSupport is the sum of support_by_dir.  An allele
    without directional support is therefore handed over with its AlleleSupport in the first direction — the order the reallocator ranks
    overlapping alleles in (longer first, then more AlleleSupport) is the test's, and what it must ADD (the failed MNV's directional
    support) is checked direction by direction."""
    dirs = list(d["dirs"])
    if sum(dirs) == 0 and d["support"] != 0:
        dirs = [d["support"], 0, 0]
    return {"position": d["position"], "ref": d["ref"], "alt": d["alt"], "category": d["category"], "dirs": dirs}


def _norm(rows):
    return sorted((r["position"], r["ref"], r["alt"], tuple(r["dirs"]), r["category"]) for r in rows)


@pytest.mark.parametrize("case", load("mnv_reallocator_cases.json")["cases"], ids=lambda c: c["name"][:40])
def test_product_mnv_reallocator_reference_cases(case):
    """MNVReallocatorTests.cs:18-662 through the library's own MnvReallocator (mnv_reallocate_failed, the function pisces_hip_flush runs
    between its two device passes): who takes the failed MNV's support, what is broken down to smaller MNVs / SNVs, what goes to the
    next block.  Expected directional support = the callable allele's own + what the test expects it to gain."""
    failed = [_as_the_product_holds_it(d) for d in case["failed"]]
    callable_ = [_as_the_product_holds_it(d) for d in case["callable"]]
    got_callable, got_outside = _product_reallocate(failed, callable_, case["max"])
    own = {}
    for d_in, d_held in zip(case["callable"], callable_):
        own[(d_in["position"], d_in["ref"], d_in["alt"])] = (d_in, d_held["dirs"])

    def expected(rows):
        out = []
        for r in rows:
            k = (r["position"], r["ref"], r["alt"])
            if k in own:   # what the allele gains in the test (directional support; AlleleSupport alone where the failed MNV has none), on top of what the product was given
                given, held = own[k]
                gain = [r["dirs"][i] - given["dirs"][i] for i in range(3)]
                if not any(gain):
                    gain = [r["support"] - given["support"], 0, 0]
                dirs = [held[i] + gain[i] for i in range(3)]
            else:          # an allele the reallocator makes: it carries the failed MNV's support
                dirs = _as_the_product_holds_it(r)["dirs"]
            out.append({"position": r["position"], "ref": r["ref"], "alt": r["alt"], "dirs": dirs, "category": r["category"]})
        return out
    if "expect_callable" in case:
        assert _norm(got_callable) == _norm(expected(case["expect_callable"]))
    for want in case.get("expect_callable_contains", []):
        assert _norm(got_callable).count(_norm(expected([want]))[0]) == 1
    assert _norm(got_outside) == _norm(expected(case["expect_outside"]))


def _product_genotypes(alleles, ploidy, min_depth, min_gq=0, max_gq=0):
    from pisces_amd import _native
    cfg = _abi.default_config(ploidy=ploidy, min_coverage=min_depth, min_genotype_qscore=min_gq, max_genotype_qscore=max_gq)
    arr = (_abi.PiscesGenotypeAllele * max(1, len(alleles)))()
    pool = bytearray()
    for i, d in enumerate(alleles):
        a = arr[i]
        a.category, a.ref_len, a.alt_len = d["category"], len(d["ref"]), len(d["alt"])
        a.support, a.coverage, a.reference_support = d["support"], d["coverage"], d["ref_support"]
        a.allele_offset = len(pool)
        pool += d["ref"].encode() + d["alt"].encode()
    poola = np.frombuffer(bytes(pool) or b"\0", dtype=np.uint8).copy()
    gt = _native.lib.pisces_hip_set_genotypes(C.byref(cfg), arr, len(alleles), poola.ctypes.data, len(pool))
    return gt, [int(arr[i].prune) for i in range(len(alleles))], [int(arr[i].genotype) for i in range(len(alleles))]


_DIPLOID = load("diploid_cases.json")


@pytest.mark.parametrize("case", _DIPLOID["genotype_scenarios"], ids=lambda c: "%s-%s-%s" % (c["genotype"], c["ref_freqs"], c["alt_freqs"]))
def test_product_diploid_genotype_scenarios(case):
    """GenotypeCalculatorTest.DiploidGenotypeScenarios through its harness (:107-147) and the library's DiploidThresholdingGenotyper
    (diploid.cpp, the host pass of pisces_hip_flush in PloidyModel.DiploidByThresholding)."""
    cov = case["coverage"]
    alleles, ref_freq = [], 0.0
    for rf in case["ref_freqs"]:
        sup = int(np.float32(rf) * np.float32(cov))
        alleles.append({"category": _abi.CAT_REFERENCE, "ref": "A", "alt": "A", "support": sup, "coverage": cov, "ref_support": sup})
        ref_freq = float(np.float32(rf))
    if ref_freq == 0:
        ref_freq = 1.0 - float(np.sum(np.array(case["alt_freqs"], dtype=np.float32), dtype=np.float32))
    for vf in case["alt_freqs"]:
        alleles.append({"category": _abi.CAT_SNV, "ref": "A", "alt": "T", "support": int(np.float32(vf) * np.float32(cov)), "coverage": cov,
                        "ref_support": int(ref_freq * cov)})
    gt, prune, per = _product_genotypes(alleles, _abi.PLOIDY_DIPLOID if hasattr(_abi, "PLOIDY_DIPLOID") else 1, _DIPLOID["min_depth_to_genotype"])
    assert gt == GT_CODE[case["genotype"]] and sum(prune) == case["prune"]
    assert all(g == gt for g in per)


@pytest.mark.parametrize("table", _DIPLOID["genotype_qscores"], ids=lambda t: "%s-%d" % (t["genotype"], t["depth"]))
def test_product_diploid_genotype_qscores(table):
    """DiploidGenotypeQualityCalculatorTests (:16-96, :103-117) through TestCalculation (:124-134) and the library's calculator."""
    from pisces_amd import _native
    gt = GT_CODE[table["genotype"]]
    for f, want in zip(table["frequencies"], table["expected"]):
        depth = float(table["depth"])
        support = int(depth * f)
        if table["genotype"] == "HomozygousRef":
            support = int(depth * (1.0 - f))
        assert _native.lib.pisces_hip_diploid_genotype_qscore(gt, int(depth), support, 0, 2147483647) == want, (table["genotype"], depth, f)


@pytest.mark.parametrize("want,prune,ref_freq,alt_freqs,coverage", [
    (9, 2, 0.80, [0.01, 0.01], 1000),     # HemizygousRefTest
    (11, 2, 0.70, [0.01, 0.01], 1000),    # NoCallDueToRefMajorVf
    (11, 2, 0.22, [0.75, 0.01], 1000),    # NoCallDueToRefMinorVf
    (11, 2, 0.80, [0.01, 0.01], 10),      # NoCallDueToCoverge
    (10, 1, 0.10, [0.75, 0.01], 1000),    # HemizygousAlt
])
def test_product_haploid_genotype_scenarios(want, prune, ref_freq, alt_freqs, coverage):
    """HaploidGenotypeCalculatorTests.cs:59-96 through its harness (:20-57) and the library's HaploidGenotyper."""
    sup = int(np.float32(ref_freq) * np.float32(coverage))
    alleles = [{"category": _abi.CAT_REFERENCE, "ref": "A", "alt": "A", "support": sup, "coverage": coverage, "ref_support": sup}]
    for vf in alt_freqs:
        alleles.append({"category": _abi.CAT_SNV, "ref": "A", "alt": "T", "support": int(np.float32(vf) * np.float32(coverage)),
                        "coverage": coverage, "ref_support": int(float(np.float32(ref_freq)) * coverage)})
    gt, pr, per = _product_genotypes(alleles, 2, 100, min_gq=0, max_gq=100)
    assert gt == want and sum(pr) == prune and all(g == gt for g in per)


# ============================================================================================ GPU: through the flush
@pytest.fixture(scope="module")
def torch_cuda():
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch


def _stage(c, cells):
    """cells: {(position, allele type, direction, anchor bin): count} -> that many observations each (IAlleleSource.GetAlleleCount then
    returns exactly these cells)."""
    pos, tup = [], []
    for (p, a, d, anchor), n in cells.items():
        if n <= 0:
            continue
        q = 255 if a == 5 else 30
        pos += [p] * n
        tup += [_abi.tuple_pack(0, anchor, d, a, q)] * n
    if pos:
        c.AddObservations(np.array(pos, dtype=np.int32), np.array(tup, dtype=np.uint32))


def _permissive(**kw):
    """Every candidate comes back as a row: no coverage / frequency / q-score threshold, no filter."""
    base = dict(min_coverage=0, min_variant_qscore=0, min_frequency=0.0, rmxn_max_repeat_length=-1, variant_qscore_filter=-1, low_depth_filter=-1,
                variant_freq_filter=-1.0, no_call_filter_threshold=-1.0)
    base.update(kw)
    return _abi.default_config(**base)


@pytest.mark.gpu
@pytest.mark.parametrize("case", load("collapser_cases.json")["cases"], ids=lambda c: c["name"])
def test_product_collapser_reference_cases(torch_cuda, case):
    """VariantCollapserTests (HappyPath :18-65, NegativeCases :112-204, suites :949-1033) through ExecuteTest's set-up (:869-916): every
    candidate with support [1, 0, 0], a state in which every allele count is 1, thresholds 0 — and the library's flush: candidates in
    through pisces_hip_add_candidates, counts through pisces_hip_add_observations, the collapser between them and the candidate kernel.
    What comes back: as many insertion / deletion rows as the test expects candidates to be left, TotalNumCollapsed, the survivor's
    support; in the given and in the reversed order."""
    from pisces_amd import engine
    ref = np.frombuffer(b"ACGT" * 16, dtype=np.uint8)
    cfg = _permissive(collapse=1, collapse_freq_threshold=0.0, collapse_freq_ratio_threshold=0.0, include_reference_calls=0)
    cells = {(p, a, d, 5): 1 for p in range(1, 40) for a in range(6) for d in range(3)}
    for order in (case["candidates"], list(reversed(case["candidates"]))):
        with engine.HipVariantCaller(cfg) as c:
            c.SetReference(ref)
            _stage(c, cells)
            c.AddCandidates([{"position": x["pos"], "category": CAT[x["category"]], "ref": x["ref"], "alt": x["alt"], "support_by_dir": (1, 0, 0),
                              "open_left": x["open_left"], "open_right": x["open_right"]} for x in order])
            recs, alleles = c.CallWithAlleles()
            stats = c.Stats()
        rows = [(r, a) for r, a in zip(recs, alleles) if _abi.info_category(r["info"]) in (_abi.CAT_INSERTION, _abi.CAT_DELETION)]
        assert len(rows) == case["expected_count"], [(int(r["position"]), a) for r, a in rows]
        assert stats["TotalNumCollapsed"] == len(order) - case["expected_count"]
        if case["expected_support_first"] is not None:
            assert int(rows[0][0]["allele_support"]) == case["expected_support_first"]


def _coverage_cells(case):
    cells = {}
    for row in case["counts"]:
        a = ALLELE[row["allele"]]
        if "dirs" in row:
            for d, v in enumerate(row["dirs"]):
                cells[(row["coord"], a, d, 5)] = v
        else:
            for d, per in enumerate(row["anchors"]):
                for anchor, v in per.items():
                    cells[(row["coord"], a, d, int(anchor))] = v
    return cells


def _coverage_row(case, support, well_anchored):
    """The row the library reports for the case's allele with the given stitched support / well-anchored support."""
    from pisces_amd import engine
    v = case["allele"]
    ref = bytearray(b"A" * 32)
    ref[v["pos"] - 1:v["pos"] - 1 + len(v["ref"])] = v["ref"].encode()
    cat = v["category"]
    # SNV candidates are the caller's only with MNV calling on (off: they are the allele counts); a Reference allele is the gVCF's
    cfg = _permissive(call_mnvs=1 if cat in ("Snv", "Mnv") else 0, include_reference_calls=1 if cat == "Reference" else 0, collapse=0, max_mnv_length=8)
    # (a Reference allele on a locus whose counts show other bases: the library makes SNV candidates of those bases unless the candidates
    # are the read walk's alone — MNV calling on in its earlier form, where the tile kernels emit Reference records only)
    if cat == "Reference":
        cfg = _permissive(call_mnvs=1, include_reference_calls=1, collapse=0)
    old = os.environ.get("PISCES_HIP_MNV_SPLIT")
    if cat == "Reference":
        os.environ["PISCES_HIP_MNV_SPLIT"] = "0"
    try:
        with engine.HipVariantCaller(cfg) as c:
            c.SetReference(np.frombuffer(bytes(ref), dtype=np.uint8))
            _stage(c, _coverage_cells(case))
            if case.get("taken_ref"):
                c.AddGappedMnvRefCount({v["pos"]: case["taken_ref"]})
            if cat != "Reference":
                c.AddCandidates([{"position": v["pos"], "category": CAT[cat], "ref": v["ref"], "alt": v["alt"], "support_by_dir": (0, 0, support),
                                  "well_anchored_by_dir": (0, 0, well_anchored)}])
            recs, alleles = c.CallWithAlleles()
    finally:
        if cat == "Reference":
            if old is None:
                os.environ.pop("PISCES_HIP_MNV_SPLIT", None)
            else:
                os.environ["PISCES_HIP_MNV_SPLIT"] = old
    rows = [r for r, a in zip(recs, alleles) if int(r["position"]) == v["pos"] and a == (v["ref"], v["alt"]) and _abi.info_category(r["info"]) == CAT[cat]]
    assert len(rows) == 1, [(int(r["position"]), a) for r, a in zip(recs, alleles)]
    return rows[0]


@pytest.mark.gpu
@pytest.mark.parametrize("case", load("coverage_spanning.json")["cases"], ids=lambda c: c["name"])
def test_product_coverage_calculator_reference_cases(torch_cuda, case):
    """CoverageCalculatorTests.ComputeCoverage_* (:69-702) through the library: the counts of the test's mock allele source cell by cell
    (allele, direction, anchor bin), the allele as a candidate (or, a Reference allele, as the gVCF's), and the coverage the record comes
    back with.  The product always tracks anchors (TrackedAnchorSize 5), so an insertion is run the three anchor-aware ways of the test's
    harness (:704-760: fully anchored, fully unanchored, half and half); alleles of other kinds do not look at the bins."""
    exp_dir, exp_total = case["by_dir"], case["total"]
    cat = case["allele"]["category"]

    def check(row, by_dir, total, support):
        assert int(row["total_coverage"]) == total
        got = [int(x) for x in row["coverage_by_dir"]][: len(by_dir)]
        assert got == list(by_dir)
        if case.get("check_aux", True):
            if cat == "Reference":
                assert int(row["allele_support"]) == case.get("snv_ref", 0)
            elif cat == "Snv":
                assert int(row["reference_support"]) == case.get("snv_ref", 0)
            else:
                assert int(row["reference_support"]) == total - support
    if cat != "Insertion":
        row = _coverage_row(case, 5, 5)
        check(row, exp_dir, exp_total, 5)
        if "expect_ref_support" in case:
            assert int(row["reference_support"]) == case["expect_ref_support"]
        return
    suspicious = case.get("suspicious", 0)
    aware = case.get("by_dir_anchor_aware")
    check(_coverage_row(case, 5, 5), aware if aware is not None else exp_dir, sum(aware) if aware is not None else exp_total - suspicious, 5)
    check(_coverage_row(case, 5, 0), exp_dir, exp_total, 5)
    from_unanchored = np.float32(suspicious) * np.float32(0.5)
    total_support = int(from_unanchored + np.float32(0.5) * np.float32(exp_total - suspicious))
    check(_coverage_row(case, total_support, int(total_support - from_unanchored)), exp_dir, exp_total, total_support)


_MATRIX = load("caller_matrix.json")


@pytest.mark.gpu
@pytest.mark.parametrize("sc", _MATRIX["scenarios"], ids=lambda s: s["name"])
def test_product_allele_caller_matrix(torch_cuda, sc):
    """VariantCallerTests.EvaluateVariants (:27-277) through the library: the mock state's counts (100 / 1 for every allele and direction
    at the high- / low-coverage coordinate), the test's candidates (SNV candidates are the caller's with MNV calling on), its thresholds —
    and the alleles that come back.  Candidates on chr1..chr4 are separate loci with one coordinate: one handle each.  A Reference
    candidate cannot be handed over (Reference alleles are the gVCF's: RegionState.GetAllCandidates): a scenario that lists one runs with
    reference calls on for its chromosome, which changes IsCallable only through the minimum coverage — the one scenario where that
    matters (a Reference allele kept because the variant fails the coverage test without gVCF) is not expressible and is skipped."""
    from pisces_amd import engine
    from tests import orc
    if sc["name"] == "reference_kept_when_the_variant_fails_coverage":
        pytest.skip("a Reference candidate without reference calls cannot be brought through the C ABI")
    ov = {k: v for k, v in sc["config"].items() if k in ("max_variant_qscore", "noise_level", "min_coverage", "min_variant_qscore", "min_frequency",
                                                          "low_gq_filter", "max_genotype_qscore")}
    if "min_frequency_num" in sc["config"]:
        ov["min_frequency"] = float(np.float32(sc["config"]["min_frequency_num"]) / np.float32(sc["config"]["min_frequency_den"]))
    if "min_variant_qscore_from" in sc["config"]:   # "one above the q-score of support 40 at coverage 1500": the q-score golden of QualityCalculatorTests pins the oracle's
        f = sc["config"]["min_variant_qscore_from"]
        ov["min_variant_qscore"] = orc.lib.orc_poisson_qscore(f["support"], f["coverage"], 20, 100) + f["plus"]
    by_chr = {}
    for name in sc["candidates"]:
        by_chr.setdefault(_MATRIX["candidates"][name]["chr"], []).append(_MATRIX["candidates"][name])
    got = []
    for chrom, cands in sorted(by_chr.items()):
        has_ref = any(x["category"] == "Reference" for x in cands)
        cfg = _abi.default_config(rmxn_max_repeat_length=-1, variant_qscore_filter=-1, low_depth_filter=-1, variant_freq_filter=-1.0,
                                  no_call_filter_threshold=-1.0, call_mnvs=1, collapse=0,
                                  **dict(ov, include_reference_calls=1 if has_ref else sc["config"]["include_reference_calls"]))
        cells = {}
        for p in sorted({x["pos"] for x in cands}):
            mult = _MATRIX["counts_per_allele_direction"][str(p)]
            cells.update({(p, a, d, 5): mult for a in range(6) for d in range(3)})
        # The mock state shows every base on every locus; the test's candidates are all the candidates there are.  Where the chromosome has
        # a variant candidate its locus is the candidate path's (a caller's candidate marks it); a chromosome with a Reference candidate
        # only runs MNV calling in its earlier form (PISCES_HIP_MNV_SPLIT=0: SNV candidates are never made from the counts).
        only_reference = all(x["category"] == "Reference" for x in cands)
        old = os.environ.get("PISCES_HIP_MNV_SPLIT")
        if only_reference:
            os.environ["PISCES_HIP_MNV_SPLIT"] = "0"
        try:
            with engine.HipVariantCaller(cfg) as c:
                c.SetReference(np.frombuffer(b"A" * 600, dtype=np.uint8))
                _stage(c, cells)
                c.AddCandidates([{"position": x["pos"], "category": CAT[x["category"]], "ref": x["ref"], "alt": x["alt"], "support_by_dir": tuple(x["support"])}
                                 for x in cands if x["category"] != "Reference"])
                recs, alleles = c.CallWithAlleles()
        finally:
            if only_reference:
                if old is None:
                    os.environ.pop("PISCES_HIP_MNV_SPLIT", None)
                else:
                    os.environ["PISCES_HIP_MNV_SPLIT"] = old
        for r, a in zip(recs, alleles):
            cat = _abi.info_category(r["info"])
            got.append((chrom, int(r["position"]), a[0], a[1], cat, None if cat == _abi.CAT_REFERENCE else int(r["allele_support"])))
    want = []
    for name in sc["called"]:
        x = _MATRIX["candidates"][name]
        want.append((x["chr"], x["pos"], x["ref"], x["alt"], CAT[x["category"]], None if x["category"] == "Reference" else sum(x["support"])))
    assert sorted(got, key=str) == sorted(want, key=str)
