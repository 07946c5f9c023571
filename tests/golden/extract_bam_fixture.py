"""Turns the reference's end-to-end test inputs into a small data fixture (no BAM / FASTA travels to the GPU box):
    python tests/golden/extract_bam_fixture.py /root/reference   ->  tests/golden/bam_chr19.npz
Inputs (data files of the reference's own functional tests, src/test/Pisces.Tests/FunctionalTests/
SomaticVariantCallerFunctionalTests.cs:31-65 SimpleSnv, :168-300 IntervalTestingWithMultipleSamples):
  * src/test/Pisces.Tests/TestData/Chr17Chr19.bam (== Chr17again.bam, byte for byte): the chr19 alignments, filtered as
    AlignmentSource.ShouldSkipRead does (src/exe/Pisces/Logic/Alignment/AlignmentsSource.cs:84-92) with the default
    BamFilterParameters (MinimumMapQuality 1, RemoveDuplicates, proper pairs not required);
  * src/test/SharedData/Genomes/chr19/chr19.fa: the 1 200 bases around the reads (positions are kept absolute minus OFFSET,
    a multiple of the 1000-locus block size, so block keys are those of the real run);
  * src/test/Pisces.Tests/TestData/Sample_S1.genome.vcf (chr19:3118880-3118890, gVCF) and the chr19:3118942 row of
    Chr17again.expected.genome.vcf: the body lines Pisces wrote, as text;
  * src/test/SharedData/Bams/PhiX_S3.bam + Genomes/PhiX/WholeGenomeFasta/genome.fa with the seven SNVs Pisces_PhiX expects
    (BugGenomeTests.cs:87-178) -> tests/golden/bam_phix.npz.
The BGZF/BAM decoding below is a minimal reader written for this script (SAM/BAM specification section 4)."""
import os
import struct
import sys
import zlib

import numpy as np

OFFSET = 3118000
WINDOW = (3118001, 3119000)
CIGAR_OPS = "MIDNSHP=X"
SEQ_CODE = "=ACMGRSVTWYHKDBN"


def bgzf_decompress(path):
    data = open(path, "rb").read()
    out = []
    i = 0
    while i < len(data):
        assert data[i:i + 4] == b"\x1f\x8b\x08\x04", "not a BGZF block"
        xlen = struct.unpack_from("<H", data, i + 10)[0]
        j, bsize = i + 12, None
        while j < i + 12 + xlen:
            si1, si2, slen = data[j], data[j + 1], struct.unpack_from("<H", data, j + 2)[0]
            if si1 == 66 and si2 == 67:
                bsize = struct.unpack_from("<H", data, j + 4)[0]
            j += 4 + slen
        cdata = data[i + 12 + xlen:i + bsize + 1 - 8]
        out.append(zlib.decompress(cdata, -15))
        i += bsize + 1
    return b"".join(out)


def read_bam(path):
    b = bgzf_decompress(path)
    assert b[:4] == b"BAM\x01"
    l_text = struct.unpack_from("<i", b, 4)[0]
    p = 8 + l_text
    n_ref = struct.unpack_from("<i", b, p)[0]
    p += 4
    refs = []
    for _ in range(n_ref):
        l_name = struct.unpack_from("<i", b, p)[0]
        name = b[p + 4:p + 4 + l_name - 1].decode()
        p += 4 + l_name + 4
        refs.append(name)
    reads = []
    while p < len(b):
        block_size = struct.unpack_from("<i", b, p)[0]
        ref_id, pos, l_read_name, mapq, _bin, n_cigar, flag, l_seq, _nref, _npos, _tlen = struct.unpack_from("<iiBBHHHiiii", b, p + 4)
        q = p + 36
        name = b[q:q + l_read_name - 1].decode()
        q += l_read_name
        cigar = []
        for k in range(n_cigar):
            v = struct.unpack_from("<I", b, q + 4 * k)[0]
            cigar.append((CIGAR_OPS[v & 0xF], v >> 4))
        q += 4 * n_cigar
        seq = "".join(SEQ_CODE[(b[q + (k >> 1)] >> (4 if (k & 1) == 0 else 0)) & 0xF] for k in range(l_seq))
        q += (l_seq + 1) // 2
        qual = np.frombuffer(b[q:q + l_seq], dtype=np.uint8).copy()
        q += l_seq
        tags = b[q:p + 4 + block_size]
        reads.append(dict(ref=refs[ref_id] if ref_id >= 0 else None, pos=pos + 1, mapq=mapq, flag=flag, cigar=cigar, seq=seq, qual=qual,
                          name=name, has_xd=b"XDZ" in tags, tags=bytes(tags)))
        p += 4 + block_size
    return refs, reads


def fasta_slice(path, start, end):
    """1-based inclusive [start, end] of a single-sequence FASTA, upper-cased (Genome.cs:84-96 upper-cases the chromosome)."""
    seq = []
    n = 0
    with open(path) as f:
        for line in f:
            if line.startswith(">"):
                continue
            line = line.strip()
            lo, hi = n + 1, n + len(line)
            if hi >= start and lo <= end:
                seq.append(line[max(start, lo) - lo:min(end, hi) - lo + 1])
            n = hi
            if n >= end:
                break
    return "".join(seq).upper()


def extract(root, bam, chrom, fasta, window, offset, expected, dst_name, ref_literal=None, more=None):
    """Filters `chrom` alignments as AlignmentSource.ShouldSkipRead does and stores them with the reference window."""
    t = os.path.join(root, "src/test")
    refs, reads = read_bam(os.path.join(t, bam))
    keep = []
    skipped = 0
    for r in reads:
        if r["ref"] != chrom:
            if r["ref"] is None:
                skipped += 1   # unplaced reads follow the last chromosome in a sorted BAM; the extractor never reaches them per chromosome
            continue
        unmapped, secondary, dup = r["flag"] & 0x4, r["flag"] & 0x100, r["flag"] & 0x400
        # Read.IsPrimaryAlignment = !secondary (BamAlignment.IsPrimaryAlignment); supplementary reads are kept by the reference
        if unmapped or secondary or dup or r["mapq"] < 1 or not r["cigar"]:
            skipped += 1
            continue
        assert not r["has_xd"], "stitched reads (XD tag) are not expected in this BAM"
        span = sum(ln for op, ln in r["cigar"] if op in "MDN=X")
        # (the mock chromosomes end inside the reads: alignments run past the chromosome end there, as in the reference's run)
        assert window[0] <= r["pos"] and (ref_literal is not None or r["pos"] + span - 1 <= window[1]), (r["pos"], span)
        keep.append(r)
    ref = ref_literal if ref_literal is not None else fasta_slice(os.path.join(t, fasta), *window)
    assert len(ref) == window[1] - window[0] + 1
    position = np.array([r["pos"] - offset for r in keep], dtype=np.int32)
    flags = np.array([1 if r["flag"] & 0x10 else 0 for r in keep], dtype=np.uint8)
    cig_off, cig_op, cig_len, seq_off, bases, quals = [0], [], [], [0], [], []
    for r in keep:
        for op, ln in r["cigar"]:
            cig_op.append(ord(op))
            cig_len.append(ln)
        cig_off.append(len(cig_op))
        bases.append(np.frombuffer(r["seq"].encode(), dtype=np.uint8))
        quals.append(r["qual"])
        seq_off.append(seq_off[-1] + len(r["seq"]))
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), dst_name)
    np.savez_compressed(
        dst, offset=np.int32(offset), ref_start=np.int32(window[0] - offset), ref=np.frombuffer(ref.encode(), dtype=np.uint8),
        position=position, flags=flags, cigar_offset=np.array(cig_off, dtype=np.int32), cigar_op=np.array(cig_op, dtype=np.uint8),
        cigar_len=np.array(cig_len, dtype=np.uint32), seq_offset=np.array(seq_off, dtype=np.int32), bases=np.concatenate(bases),
        quals=np.concatenate(quals), expected_vcf=np.array(expected), n_skipped=np.int32(skipped), **{k: np.array(v) for k, v in (more or {}).items()})
    cig = {}
    for r in keep:
        k = "".join(op for op, ln in r["cigar"])
        cig[k] = cig.get(k, 0) + 1
    print(dst_name, "reads kept", len(keep), "skipped", skipped, "expected lines", len(expected), os.path.getsize(dst), "bytes; cigar shapes", cig,
          "reverse", int(flags.sum()), "positions", position.min() + offset, "..", position.max() + offset)


# Pisces_PhiX (src/test/Pisces.Tests/FunctionalTests/BugGenomeTests.cs:87-178): the seven expected SNVs, with the VCF rows the test
# quotes for them (:152-158), tabs restored
PHIX_ROWS = [
    ("14", "T", "C", "3", "q30;LowVariantFreq", "236", "0/1:3:234,1:236:0.00424:1000:-100.0000"),
    ("14", "T", "G", "3", "q30;LowVariantFreq", "236", "0/1:3:234,1:236:0.00424:1000:-100.0000"),
    ("19", "G", "T", "3", "q30;LowVariantFreq", "243", "0/1:3:242,1:243:0.00412:1000:-100.0000"),
    ("22", "G", "A", "3", "q30;LowVariantFreq", "225", "0/1:3:224,1:225:0.00444:1000:-100.0000"),
    ("25", "G", "T", "3", "q30;LowVariantFreq", "244", "0/1:3:243,1:244:0.00410:1000:-100.0000"),
    ("26", "A", "C", "3", "q30;LowVariantFreq", "242", "0/1:3:241,1:242:0.00413:1000:-100.0000"),
    ("42", "A", "T", "3", "q30;LowVariantFreq", "199", "0/1:3:198,1:199:0.00503:1000:-100.0000"),
]


# mock chr7 of the two edge-indel tests (SomaticVariantCallerFunctionalTests.cs:509-516 and :586-594): N padding + the amplicon
EDGE_AMPLICON = ("GTTGGTCTTCTATTTTATGCGAATTCTTCTAAGATTCCCAGGTTATTTATCATAAGAATTACATTTACATGGCAAATTTAGTTCTGTTCCTAGAAATATCTCCATGACAACCAAAAGGAACTCC"
                 "TAATTTCTGGCACACATTACTTCAGGGGT")


# mock chr1 of BasicMnvTesting (SomaticVariantCallerFunctionalTests.cs:391-398)
SMALL_S1_CHR1 = "TTGTCAGTGCGCTTTTCCCAACACCACCTGCTCCGACCACCACCAGTTTGTACTCAGTCATTTCACACCAGCAAGAACCTGTTGGAAACCAGTAATCAGGGTTAATTGGCGGCG"


def main(root):
    t = os.path.join(root, "src/test")
    # ExecuteEdgeInsertion (:540-612): edgeIns_S2.genome.vcf is the gVCF that run writes next to the BAM
    with open(os.path.join(t, "Pisces.Tests/TestData/edgeIns_S2.genome.vcf")) as f:
        rows = [l.rstrip("\r\n") for l in f if l.startswith("chr7\t")]
    # (its REF column shows the run had the amplicon at position 63, i.e. the 62-N padding of the deletion twin, not the 60 N the
    # insertion test's source carries today)
    ref = "N" * 62 + EDGE_AMPLICON
    extract(root, "Pisces.Tests/TestData/edgeIns_S2.bam", "chr7", None, (1, len(ref)), 0, rows, "bam_edge_ins.npz", ref_literal=ref)
    # the deletion twin (:462-538): expects exactly one variant, chr7:107 ATTT>A
    ref = "N" * 62 + EDGE_AMPLICON
    extract(root, "Pisces.Tests/TestData/edgeIndel_S2.bam", "chr7", None, (1, len(ref)), 0, ["chr7\t107\t.\tATTT\tA"], "bam_edge_del.npz",
            ref_literal=ref)
    # BasicMnvTesting (:381-424): small_S1.bam on the test's mock chr1, MNV calling on (MaxSizeMNV 15, MaxGapBetweenMNV 10), collapser
    # off; exactly these three variants
    extract(root, "Pisces.Tests/TestData/small_S1.bam", "chr1", None, (1, len(SMALL_S1_CHR1)), 0,
            ["chr1\t27\t.\tCC\tTT", "chr1\t27\t.\tCCTGCTCCG\tTTTGCTCCA", "chr1\t35\t.\tG\tA"], "bam_small_s1.npz", ref_literal=SMALL_S1_CHR1)
    # Sample_S1.genome.vcf is what the last run of SimpleSnv leaves behind (gVCF, Sample_S1_negative.picard: chr19:3118880-3118890);
    # the variant row comes from Chr17again.expected.genome.vcf (IntervalTestingWithMultipleSamples, same reads: Chr17again.bam ==
    # Chr17Chr19.bam, whose chr19 alignments are those of Sample_S1.bam)
    with open(os.path.join(t, "Pisces.Tests/TestData/Sample_S1.genome.vcf")) as f:
        expected = [l.rstrip("\r\n") for l in f if l.startswith("chr19\t")]
    with open(os.path.join(t, "Pisces.Tests/TestData/Chr17again.expected.genome.vcf")) as f:
        expected += [l.rstrip("\r\n") for l in f if l.startswith("chr19\t3118942\t")]
    extract(root, "Pisces.Tests/TestData/Chr17Chr19.bam", "chr19", "SharedData/Genomes/chr19/chr19.fa", WINDOW, OFFSET, expected, "bam_chr19.npz")
    # chr17 of the same BAM (IntervalTestingWithVcf :101-166, IntervalTestingWithMultipleSamples :168-300).  The runs' genome "fourChrs" is an
    # index without its fasta in the reference's tree; the reference bases come from the REF column of the rows those runs wrote (every
    # position the three VCFs cover: 7572952-7572990), N elsewhere — outside the intervals nothing is called, and with MNV calling off
    # the only thing that looks at reference bases beside a called position is the RMxN filter, which does not apply to the one variant
    # (its frequency 0.504 is above the filter's 0.35).
    known = {}
    sets = {}
    for key, name, keep in (("expected_vcf", "Chr17again.expected.genome.vcf", lambda l: l.startswith("chr17\t")),
                            ("expected_vcf_int", "Chr17Chr19.expected.genome.vcf", lambda l: l.startswith("chr17\t")),
                            ("expected_vcf_variants", "Chr17Chr19.expected.vcf", lambda l: l.startswith("chr17\t"))):
        with open(os.path.join(t, "Pisces.Tests/TestData", name)) as f:
            sets[key] = [l.rstrip("\r\n") for l in f if keep(l)]
        for l in sets[key]:
            c = l.split("\t")
            assert len(c[3]) == 1 and known.get(int(c[1]), c[3]) == c[3]
            known[int(c[1])] = c[3]
    assert sorted(known) == list(range(7572952, 7572991)) and [len(sets[k]) for k in ("expected_vcf", "expected_vcf_int", "expected_vcf_variants")] == [29, 11, 1]
    off17, win17 = 7572000, (7572001, 7573200)
    ref = "".join(known.get(p, "N") for p in range(win17[0], win17[1] + 1))
    extract(root, "Pisces.Tests/TestData/Chr17Chr19.bam", "chr17", None, win17, off17, sets["expected_vcf"], "bam_chr17.npz", ref_literal=ref,
            more={"expected_vcf_int": sets["expected_vcf_int"], "expected_vcf_variants": sets["expected_vcf_variants"]})
    rows = ["phix\t%s\t.\t%s\t%s\t%s\t%s\tDP=%s\tGT:GQ:AD:DP:VF:NL:SB\t%s" % r for r in PHIX_ROWS]
    extract(root, "SharedData/Bams/PhiX_S3.bam", "phix", "SharedData/Genomes/PhiX/WholeGenomeFasta/genome.fa", (1, 5386), 0, rows, "bam_phix.npz")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "/root/reference")
