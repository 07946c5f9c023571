"""VCF body lines: pisces_hip_format_vcf against the reference's known answers
(src/test/Pisces.IO.Tests/UnitTests/VcfFormatterTests.cs, VcfFileWriterTests.cs) and against body lines Pisces itself wrote
(tests/golden/vcf_lines.json, from the reference's test-data VCFs).  Pure CPU: the formatter has no device code."""
import json
import os

import ctypes as C

import numpy as np
import pytest

from pisces_amd import _abi, engine

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
BASE = {"A": 0, "G": 1, "C": 2, "T": 3, "N": 4}
GT = {"1/2": 0, "0/1": 2, "1/1": 3, "0/0": 4, "0/.": 7, "1/.": 8}
FILTER_BIT = {"SB": 0, "LowDP": 4, "LowVariantFreq": 5, "LowGQ": 6, "NC": 12}
CAT_SNV, CAT_INS, CAT_DEL, CAT_MNV, CAT_REF = 0, 1, 2, 3, 4


def record(pos, ref, alt, gt, cat, cov, support, ref_support, q, gq, sb_score=1.0, no_calls=0, filters=()):
    r = np.zeros(1, dtype=_abi.CALLED_ALLELE_DTYPE)
    r["position"], r["total_coverage"], r["allele_support"], r["reference_support"] = pos, cov, support, ref_support
    r["num_no_calls"], r["variant_qscore"], r["genotype_qscore"], r["strand_bias_score"] = no_calls, q, gq, sb_score
    bits = 0
    for f in filters:
        bits |= 1 << f
    r["filter_bits"] = bits
    r["info"] = gt | (cat << 4) | (BASE.get(ref[:1], 4) << 7) | (BASE.get(alt[:1], 4) << 10)
    return r


def test_adhoc_line_of_the_writer_test():
    # VcfFileWriterTests.TestSomaticStyleWithVariants-style line: q filter 20, frequency thresholds 0.007 (=> 4 VF decimals... the
    # format takes max(sig digits of 0.007) = 4), NL 23, NC on
    r = record(55141055, "A", "G", GT["1/1"], CAT_SNV, cov=5394, support=5387, ref_support=7, q=0, gq=0, sb_score=1.0)
    text = engine.format_vcf("chr4", r, variant_quality_filter=20, min_frequency_threshold=0.007, frequency_filter_threshold=0.007,
                             noise_level=23, output_no_call_fraction=1)
    assert text == "chr4\t55141055\t.\tA\tG\t0\tPASS\tDP=5394\tGT:GQ:AD:DP:VF:NL:SB:NC\t1/1:0:7,5387:5394:0.9987:23:0.0000:0.0000\n"


def test_info_and_format_merge_samples():
    # VcfFormatterTests.InfoAndFormatMerge: the one-variant cases (4 VF decimals there: a 0.001 threshold, "0.001".Length - 1).  The reference test hands
    # ConstructFormatAndSampleString an arbitrary DP (63); through the writer DP is GetDepthCountInt = max(coverage, ...) = 100
    kw = dict(min_frequency_threshold=0.001, frequency_filter_threshold=-1.0, noise_level=23, output_no_call_fraction=1)
    ref = record(10, "A", "A", GT["0/0"], CAT_REF, cov=490, support=490, ref_support=490, q=100, gq=42)
    line = engine.format_vcf("chr1", ref, **kw).rstrip("\n").split("\t")
    assert line[8] == "GT:GQ:AD:DP:VF:NL:SB:NC" and line[9] == "0/0:42:490:490:0.0000:23:0.0000:0.0000" and line[4] == "."
    het = record(10, "A", "T", GT["0/1"], CAT_SNV, cov=100, support=10, ref_support=0, q=100, gq=200)
    line = engine.format_vcf("chr1", het, **kw).rstrip("\n").split("\t")
    assert line[9] == "0/1:200:0,10:100:0.1000:23:0.0000:0.0000" and line[7] == "DP=100" and line[4] == "T"


def test_filters_in_processor_order_and_names():
    # MapFilter names (VcfFormatter.cs:143-182) joined in AlleleProcessor.ApplyFilters order; VcfFileWriterTests_Test1_expected.vcf
    # holds "LowDP;q20;SB"
    r = record(567, "A", "T", GT["0/1"], CAT_SNV, 0, 0, 0, 20, 20, filters=(0, 3, 4))
    f = engine.format_vcf("chr1", r, variant_quality_filter=20).split("\t")[6]
    assert f == "LowDP;q20;SB"
    r = record(567, "A", "T", GT["0/1"], CAT_SNV, 0, 0, 0, 20, 20, filters=(0, 3, 4, 5, 9, 12))
    f = engine.format_vcf("chr1", r).split("\t")[6]
    assert f == "LowDP;q30;NC;SB;R5x9;LowVariantFreq"
    with pytest.raises(engine.PiscesHipError):   # InvalidDataException in the reference: filter set but threshold null
        engine.format_vcf("chr1", r, variant_quality_filter=-1)


def test_genotype_strings():
    # VcfFormatterTests genotype map
    want = {0: "1/2", 1: "./.", 2: "0/1", 3: "1/1", 4: "0/0", 5: "./.", 6: "./.", 7: "0/.", 8: "1/."}
    for g, s in want.items():
        r = record(5, "A", "T", g, CAT_SNV, 100, 50, 50, 100, 100)
        assert engine.format_vcf("chr1", r).split("\t")[9].split(":")[0] == s


def test_frequency_decimals_follow_the_thresholds():
    # UpdateFrequencyFormat: digits of Single.ToString() of the thresholds; 1E-05 -> 5
    r = record(5, "A", "T", GT["0/1"], CAT_SNV, 3, 1, 2, 100, 100)
    vf = lambda **kw: engine.format_vcf("chr1", r, **kw).split("\t")[9].split(":")[4]
    assert vf(min_frequency_threshold=0.01, frequency_filter_threshold=-1.0) == "0.333"
    assert vf(min_frequency_threshold=0.01, frequency_filter_threshold=0.0001) == "0.33333"   # "0.0001".Length - 1 = 5
    assert vf(min_frequency_threshold=1e-5, frequency_filter_threshold=-1.0) == "0.33333"
    assert vf(min_frequency_threshold=0.5, frequency_filter_threshold=-1.0) == "0.33"
    # half-up on the 7 significant digits of the Single, not on the binary value: 0.0625 -> "0.063" (banker's would give 0.062)
    r = record(5, "A", "T", GT["0/1"], CAT_SNV, 16, 1, 15, 100, 100)
    assert engine.format_vcf("chr1", r, min_frequency_threshold=0.01, frequency_filter_threshold=-1.0).split("\t")[9].split(":")[4] == "0.063"


def test_indel_rows_take_their_alleles_from_the_candidates():
    d = record(100, "A", "A", GT["0/1"], CAT_DEL, 200, 50, 150, 100, 100)
    i = record(100, "A", "A", GT["0/1"], CAT_INS, 200, 20, 150, 100, 100)
    s = record(101, "C", "T", GT["0/1"], CAT_SNV, 200, 20, 180, 100, 100)
    recs = np.concatenate([d, i, s])
    text = engine.format_vcf("chr7", recs, alleles=[("ACG", "A"), ("A", "ATT"), ("C", "T")])
    rows = [l.split("\t") for l in text.rstrip("\n").split("\n")]
    assert [(r[3], r[4]) for r in rows] == [("ACG", "A"), ("A", "ATT"), ("C", "T")]
    assert rows[0][9].startswith("0/1:100:150,50:200:0.250")


def test_empty_and_capacity():
    assert engine.format_vcf("chr1", np.zeros(0, dtype=_abi.CALLED_ALLELE_DTYPE)) == ""
    many = np.concatenate([record(p, "A", "T", GT["0/1"], CAT_SNV, 100000, 5000, 95000, 100, 100) for p in range(1, 40)])
    text = engine.format_vcf("chr_with_a_rather_long_name_" * 12, many)   # forces the second, larger buffer
    assert text.count("\n") == 39


def _parse(line, spec):
    c = line.split("\t")
    fmt, smp = c[8].split(":"), c[9].split(":")
    f = dict(zip(fmt, smp))
    ref, alt = c[3], c[4]
    if "," in alt or "," in f["AD"] and len(f["AD"].split(",")) > 2:
        return None   # crushed / multi-allelic rows are not produced by the uncrushed writer
    gts = f["GT"]
    if alt == ".":
        cat = CAT_REF
    elif len(ref) == 1 and len(alt) == 1:
        cat = CAT_SNV
    elif len(ref) > len(alt) and len(alt) == 1:
        cat = CAT_DEL
    elif len(alt) > len(ref) and len(ref) == 1:
        cat = CAT_INS
    else:
        cat = CAT_MNV
    if gts == "./.":
        gt = 5 if alt == "." else 6
    elif gts in GT:
        gt = GT[gts]
    else:
        return None
    if (alt == ".") != (gt in (4, 5, 7)):
        return None   # hand-built rows of the writer tests (reference-type allele with an alt genotype)
    ad = [int(x) for x in f["AD"].split(",")]
    dp = int(f["DP"])
    if cat == CAT_REF:
        if len(ad) != 1:
            return None
        support, ref_support = ad[0], ad[0]
    else:
        if len(ad) != 2:
            return None
        ref_support, support = ad
    filters = []
    if c[6] != "PASS":
        for name in c[6].split(";"):
            if name in FILTER_BIT:
                filters.append(FILTER_BIT[name])
            elif spec["q"] is not None and name == "q%d" % spec["q"]:
                filters.append(3)
            elif spec["rmxn"] and name == "R%dx%d" % tuple(spec["rmxn"]):
                filters.append(9)
            else:
                return None   # filters of other tools (phasing, Psara, multi-allelic tags)
    sb = float(f["SB"]) if "SB" in f else 0.0
    no_calls = 0
    if "NC" in f:
        nc = float(f["NC"])
        no_calls = int(round(nc * dp / (1.0 - nc))) if nc < 1.0 else 0
    r = record(int(c[1]), ref, alt if alt != "." else ref, gt, cat, dp, support, ref_support, int(c[5]), int(f["GQ"]),
               sb_score=10.0 ** (sb / 10.0), no_calls=no_calls, filters=filters)
    decimals = len(f["VF"].split(".")[1])
    return r, (ref, alt if alt != "." else ref), f, decimals


@pytest.mark.parametrize("spec", json.load(open(os.path.join(GOLDEN, "vcf_lines.json"))), ids=lambda s: os.path.basename(s["file"]))
def test_lines_pisces_wrote_are_reproduced(spec):
    """Every single-allele body line of a reference test-data VCF, parsed back into a record and re-formatted, must come out
    byte-identical.  Rows are skipped only when the record cannot express them (see _parse) or when the line states a depth the
    writer of this version cannot produce (DP below AD or below the coverage the VF was computed on, older writers)."""
    checked = 0
    for line in spec["lines"]:
        p = _parse(line, spec)
        if p is None:
            continue
        r, alleles, f, decimals = p
        support, cov = int(r["allele_support"][0]), int(r["total_coverage"][0])
        if "NL" in f and support == 0:
            continue   # NL of a zero-support row: this version leaves NoiseLevelApplied 0, older writers printed the configured level
        if "SB" in f and support == 0 and f["SB"] != "0.0000":
            continue   # same: strand bias is only computed for support > 0 (AlleleCaller.cs:211-228)
        if max(support + (0 if alleles[0] == alleles[1] else int(r["reference_support"][0])), support) > cov:
            continue   # DP would be raised by GetDepthCountInt; the VF of the line was computed on a coverage we cannot recover
        kw = dict(variant_quality_filter=spec["q"] if spec["q"] is not None else -1,
                  rmxn_max_repeat_length=spec["rmxn"][0] if spec["rmxn"] else -1, rmxn_min_repetitions=spec["rmxn"][1] if spec["rmxn"] else -1,
                  noise_level=int(f["NL"]) if "NL" in f else 20, output_strand_bias_and_noise_level=int("NL" in f),
                  output_no_call_fraction=int("NC" in f), min_frequency_threshold=10.0 ** -(decimals - 1), frequency_filter_threshold=-1.0)
        got = engine.format_vcf(line.split("\t")[0], r, alleles=[alleles], **kw)
        assert got == line + "\n", (got, line)
        checked += 1
    assert checked >= (7 * len(spec["lines"])) // 10, (checked, len(spec["lines"]))   # 85/116, 177/177, 39/39, 36/36, 89/90


def test_crushed_and_padded_lines_of_the_writer_test():
    """VcfFileWriterTests.TestDiploidStyleWithVariantsAndPadding (src/test/Pisces.IO.Tests/UnitTests/VcfFileWriterTests.cs:160-262): three
    alleles (two of them co-located, 1/2), intervals 2-3, 6-8, 10-11 over a reference of C's, written in two calls and finished; the
    body of VcfFileWriterTests_Crushed_Padded_expected.vcf, byte for byte."""
    a = record(7, "C", "A", GT["1/1"], CAT_SNV, cov=5394, support=2387, ref_support=7, q=0, gq=0)
    b = record(10, "A", "G", GT["1/2"], CAT_SNV, cov=5394, support=2387, ref_support=7, q=0, gq=0)
    c = record(10, "A", "A", GT["1/2"], CAT_DEL, cov=5394, support=2000, ref_support=7, q=0, gq=0)
    kw = dict(variant_quality_filter=20, min_frequency_threshold=0.007, frequency_filter_threshold=0.007, noise_level=23,
              output_no_call_fraction=1, crush=1)
    pad = dict(state=engine.new_pad_state(), reference=b"C" * 15, intervals=[(2, 3), (6, 8), (10, 11)])
    text = engine.format_vcf("chr4", a, alleles=[("C", "A")], pad=pad, **kw)
    text += engine.format_vcf("chr4", np.concatenate([b, c]), alleles=[("A", "G"), ("AA", "G")], pad=pad, **kw)
    text += engine.format_vcf("chr4", np.zeros(0, dtype=_abi.CALLED_ALLELE_DTYPE), pad=dict(pad, finish=True), **kw)
    want = [
        "chr4\t2\t.\tC\t.\t0\tLowDP\tDP=0\tGT:GQ:AD:DP:VF:NL:SB:NC\t./.:0:0:0:0.0000:23:0.0000:0.0000",
        "chr4\t3\t.\tC\t.\t0\tLowDP\tDP=0\tGT:GQ:AD:DP:VF:NL:SB:NC\t./.:0:0:0:0.0000:23:0.0000:0.0000",
        "chr4\t6\t.\tC\t.\t0\tLowDP\tDP=0\tGT:GQ:AD:DP:VF:NL:SB:NC\t./.:0:0:0:0.0000:23:0.0000:0.0000",
        "chr4\t7\t.\tC\tA\t0\tPASS\tDP=5394\tGT:GQ:AD:DP:VF:NL:SB:NC\t1/1:0:7,2387:5394:0.4425:23:0.0000:0.0000",
        "chr4\t8\t.\tC\t.\t0\tLowDP\tDP=0\tGT:GQ:AD:DP:VF:NL:SB:NC\t./.:0:0:0:0.0000:23:0.0000:0.0000",
        "chr4\t10\t.\tAA\tGA,G\t0\tPASS\tDP=5394\tGT:GQ:AD:DP:VF:NL:SB:NC\t1/2:0:2387,2000:5394:0.8133:23:0.0000:0.0000",
        "chr4\t11\t.\tC\t.\t0\tLowDP\tDP=0\tGT:GQ:AD:DP:VF:NL:SB:NC\t./.:0:0:0:0.0000:23:0.0000:0.0000",
    ]
    assert text.rstrip("\n").split("\n") == want
    assert (pad["state"].last_variant_position_written, pad["state"].last_padded_position) == (0, 11)


def test_padding_needs_its_inputs_and_keeps_state_on_a_short_buffer():
    r = record(7, "C", "A", GT["1/1"], CAT_SNV, cov=100, support=60, ref_support=40, q=100, gq=100)
    st = engine.new_pad_state()
    with pytest.raises(engine.PiscesHipError):   # state without a reference
        cfg = _abi.PiscesVcfConfig()
        engine.lib.pisces_hip_vcf_default_config(C.byref(cfg))
        rc = engine.lib.pisces_hip_format_vcf_padded(C.byref(cfg), b"chr1", r.ctypes.data, 1, None, None, None, None, 0, None, None, 0,
                                                     C.byref(st), 0, None, 0)
        if rc < 0:
            raise engine.PiscesHipError(int(rc), "invalid")
    # a buffer that is too small reports the size and leaves the cursors alone
    cfg = _abi.PiscesVcfConfig()
    engine.lib.pisces_hip_vcf_default_config(C.byref(cfg))
    ref = np.frombuffer(b"C" * 15, dtype=np.uint8)
    starts, ends = np.array([2], dtype=np.int32), np.array([9], dtype=np.int32)
    buf = C.create_string_buffer(8)
    need = engine.lib.pisces_hip_format_vcf_padded(C.byref(cfg), b"chr1", r.ctypes.data, 1, None, None, None, ref.ctypes.data, 15,
                                                   starts.ctypes.data, ends.ctypes.data, 1, C.byref(st), 1, buf, 8)
    assert need > 8 and (st.last_variant_position_written, st.last_padded_position, st.last_cleared_interval_index) == (0, 0, -1)


def test_phase_set_index_orders_the_unspecified_allele_of_a_1_2_line():
    """SetUncrushedReferenceAndAlt (VcfFormatter.cs:434-448) and GetAlleleCountString (:396-421): the first variant allele of a diploid 1/2
    call (PhaseSetIndex 1, carried in filter_bits 14..15) prints ALT,<M> and ref,support,other; the second <M>,ALT and ref,other,support;
    MultiAllelicSite keeps its place before LowGQ."""
    a = record(10, "A", "G", GT["1/2"], CAT_SNV, cov=200, support=90, ref_support=5, q=100, gq=50)
    b = record(10, "A", "T", GT["1/2"], CAT_SNV, cov=200, support=100, ref_support=5, q=100, gq=50)
    a["filter_bits"] |= 1 << 14
    b["filter_bits"] |= (2 << 14) | (1 << 8) | (1 << 6) | (1 << 0)
    la = engine.format_vcf("chr1", a).rstrip("\n").split("\t")
    lb = engine.format_vcf("chr1", b).rstrip("\n").split("\t")
    assert la[4] == "G,<M>" and la[9].split(":")[2] == "5,90,105" and la[6] == "PASS"
    assert lb[4] == "<M>,T" and lb[9].split(":")[2] == "5,95,100" and lb[6] == "SB;MultiAllelicSite;LowGQ"
