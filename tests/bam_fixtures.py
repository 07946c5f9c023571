"""The end-to-end fixtures made from the reference's own BAMs (tests/golden/extract_bam_fixture.py): reads, reference window, the VCF body
lines Pisces wrote / its functional tests expect, and the options of the run that produced them."""
import os

import numpy as np

from pisces_amd import _abi

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")

# name -> (chromosome, PiscesHipConfig overrides, VCF writer overrides, intervals (absolute, inclusive) or None, which expected rows)
# low_depth_filter = -1: the functional-test harness builds its options without VariantCallingParameters.Validate(), which is what
# turns the null LowDepthFilter into MinimumCoverage (VariantCallingParameters.cs:134-141), so zero-coverage rows stay PASS there.
CASES = {
    # SimpleSnv, last Execute: gVCF over Sample_S1_negative.picard (SomaticVariantCallerFunctionalTests.cs:31-65)
    "bam_chr19": dict(chrom="chr19", cfg=dict(low_depth_filter=-1, emit_zero_coverage_refs=1), vcf=dict(),
                      intervals=[(3118880, 3118890), (3118942, 3118942)], mode="all"),
    # chr17 of the same BAM.  IntervalTestingWithMultipleSamples (:168-300), second sample: gVCF over poorlyOrdered.picard (29 rows of
    # Chr17again.expected.genome.vcf); first sample: gVCF over chr17int.picard (the 11 rows of Chr17Chr19.expected.genome.vcf);
    # IntervalTestingWithVcf (:101-166): the same interval, variants only (the one row of Chr17Chr19.expected.vcf).  The two gVCFs were
    # written by "Pisces 1.0.0.0", whose reference rows carry GQ 100 where 5.2.11 writes the somatic genotype model's 43 (as in
    # Sample_S1.genome.vcf, same reads' chr19 twin) — the test that owns them compares parsed rows with themselves (:304-319) — so the GQ
    # of reference rows is left out of the comparison; every other field of every row, and the variant row whole, must match.
    "bam_chr17_again": dict(file="bam_chr17", expected="expected_vcf", chrom="chr17", cfg=dict(low_depth_filter=-1, emit_zero_coverage_refs=1), vcf=dict(),
                            intervals=[(7572952, 7572980)], mode="all_but_reference_gq"),
    "bam_chr17_int": dict(file="bam_chr17", expected="expected_vcf_int", chrom="chr17", cfg=dict(low_depth_filter=-1, emit_zero_coverage_refs=1), vcf=dict(),
                          intervals=[(7572980, 7572990)], mode="all_but_reference_gq"),
    "bam_chr17_vcf": dict(file="bam_chr17", expected="expected_vcf_variants", chrom="chr17", cfg=dict(low_depth_filter=-1, include_reference_calls=0),
                          vcf=dict(), intervals=[(7572980, 7572990)], mode="all"),
    # Pisces_PhiX (BugGenomeTests.cs:87-178): NL 1000, minimum frequency 0.0001, minimum variant q-score 3; the seven SNVs are the whole
    # expected variant set
    "bam_phix": dict(chrom="phix", cfg=dict(low_depth_filter=-1, noise_level=1000, min_frequency=0.0001, min_variant_qscore=3),
                     vcf=dict(noise_level=1000, min_frequency_threshold=0.0001), intervals=None, mode="variants"),
    # ExecuteEdgeInsertion (:540-612): the whole gVCF of the run, 127 rows
    "bam_edge_ins": dict(chrom="chr7", cfg=dict(low_depth_filter=-1), vcf=dict(), intervals=None, mode="all"),
    # the deletion twin (:462-538): exactly one variant
    "bam_edge_del": dict(chrom="chr7", cfg=dict(low_depth_filter=-1), vcf=dict(), intervals=None, mode="variant_alleles"),
    # BasicMnvTesting (:381-424): MNV calling on (MaxSizeMNV 15, MaxGapBetweenMNV 10), collapser off; exactly three variants, two of them
    # MNVs that start at the same position
    "bam_small_s1": dict(chrom="chr1", cfg=dict(low_depth_filter=-1, call_mnvs=1, max_mnv_length=15, max_gap_between_mnv=10, collapse=0),
                         vcf=dict(), intervals=None, mode="variant_alleles"),
}


def load(name):
    z = np.load(os.path.join(GOLDEN, CASES.get(name, {}).get("file", name) + ".npz"))
    batch = _abi.ReadBatch.from_arrays(position=z["position"], flags=z["flags"], cigar_offset=z["cigar_offset"], cigar_op=z["cigar_op"],
                                       cigar_len=z["cigar_len"], seq_offset=z["seq_offset"], bases=z["bases"], quals=z["quals"])
    return z, batch


def expected_lines(name, z):
    return [str(x) for x in z[CASES[name].get("expected", "expected_vcf")]]


def check_lines(case, lines, expected):
    """lines: VCF body lines we produced for the run; expected: the fixture's rows."""
    if case["mode"] == "all":
        assert lines == expected
    elif case["mode"] == "all_but_reference_gq":
        def without_gq(l):
            c = l.split("\t")
            if c[4] == ".":
                f = c[9].split(":")
                f[1] = "*"
                c[9] = ":".join(f)
            return "\t".join(c)
        assert [without_gq(l) for l in lines] == [without_gq(l) for l in expected] and any(l.split("\t")[4] != "." for l in lines) == any(l.split("\t")[4] != "." for l in expected)
    elif case["mode"] == "variants":
        assert [l for l in lines if l.split("\t")[4] != "."] == expected
    else:
        assert ["\t".join(l.split("\t")[:5]) for l in lines if l.split("\t")[4] != "."] == expected
