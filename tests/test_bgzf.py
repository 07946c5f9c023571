"""SURVEY row f4 (upstream stage): BGZF inflate on the device.  The checker is zlib (RFC 1951's reference implementation, Python's
stdlib binding - test infrastructure like oracle/): every block the device inflates must equal zlib's bytes for the same payload.
Fixtures: the bytes of four BAM files the reference's own tests hold (tests/golden/bgzf_fixtures.npz, data only)."""
import gzip
import os
import struct
import zlib

import numpy as np
import pytest

from pisces_amd import _abi, engine

_FIXTURES = np.load(os.path.join(os.path.dirname(__file__), "golden", "bgzf_fixtures.npz"))
NAMES = sorted(_FIXTURES.files)


def bam_bytes(name):
    return _FIXTURES[name].tobytes()


def python_block_table(data):
    """BGZF block walk restated in Python (SAM spec 4.1; BamReader.ReadBlock reads BSIZE at byte 16 of the 18-byte header)."""
    out, pos, total = [], 0, 0
    while pos < len(data):
        assert data[pos:pos + 4] == b"\x1f\x8b\x08\x04"
        xlen = struct.unpack_from("<H", data, pos + 10)[0]
        extra = data[pos + 12: pos + 12 + xlen]
        bsize, x = None, 0
        while x + 4 <= len(extra):
            slen = struct.unpack_from("<H", extra, x + 2)[0]
            if extra[x:x + 2] == b"BC":
                bsize = struct.unpack_from("<H", extra, x + 4)[0] + 1
            x += 4 + slen
        crc, isize = struct.unpack_from("<II", data, pos + bsize - 8)
        out.append((pos + 12 + xlen, bsize - 12 - xlen - 8, total, isize, crc))
        total += isize
        pos += bsize
    return out, total


def make_bgzf(chunks, level=6, strategy=zlib.Z_DEFAULT_STRATEGY):
    """chunks of <= 65280 bytes -> BGZF members + the empty end-of-file block."""
    out = bytearray()
    for chunk in list(chunks) + [b""]:
        co = zlib.compressobj(level, zlib.DEFLATED, -15, 9, strategy)
        payload = co.compress(chunk) + co.flush()
        bsize = 18 + len(payload) + 8
        assert bsize <= 65536 + 26
        out += b"\x1f\x8b\x08\x04" + b"\0" * 4 + b"\0\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, bsize - 1)
        out += payload + struct.pack("<II", zlib.crc32(chunk) & 0xFFFFFFFF, len(chunk))
    return bytes(out)


@pytest.mark.parametrize("name", NAMES)
def test_block_table_of_the_reference_bams(name):
    data = bam_bytes(name)
    blocks, total = engine.bgzf_scan(data)
    exp, exp_total = python_block_table(data)
    assert total == exp_total and len(blocks) == len(exp) and len(exp) >= 2
    for b, e in zip(blocks, exp):
        assert (b.in_offset, b.in_length, b.out_offset, b.out_length, b.crc32) == e
    assert blocks[len(blocks) - 1].out_length == 0   # the end-of-file marker block
    # the checker itself agrees with the gzip reader on these files
    assert sum(len(zlib.decompress(data[o:o + n], -15)) for o, n, *_ in exp) == len(gzip.decompress(data)) == total


def test_scan_rejects_what_is_not_bgzf():
    data = bam_bytes(NAMES[0])
    for bad in (data[:-1], b"BAM\x01" + data, data[:10], gzip.compress(b"plain gzip has no BC field")):
        with pytest.raises(engine.PiscesHipError):
            engine.bgzf_scan(bad)
    blocks, total = engine.bgzf_scan(b"")
    assert len(blocks) == 0 and total == 0


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_device_inflate_of_the_reference_bams_equals_zlib(name):
    data = bam_bytes(name)
    with engine.HipVariantCaller(_abi.default_config()) as c:
        got, blocks, ms = c.bgzf_inflate(data)
    exp = b"".join(zlib.decompress(data[b.in_offset:b.in_offset + b.in_length], -15) for b in blocks)
    assert got == exp and got == gzip.decompress(data)
    assert got[:4] == b"BAM\x01" and ms > 0


@pytest.mark.gpu
def test_device_inflate_on_synthetic_streams(tmp_path):
    """Stored / fixed / dynamic blocks, every compression level, runs (distance 1), maximum-size blocks, empty blocks, several DEFLATE
    blocks in one member; a corrupt payload and a wrong CRC are reported, not returned."""
    rng = np.random.default_rng(3)
    text = bytes(rng.choice(list(b"ACGTN\n\t0123456789"), 400_000, p=[.2, .2, .2, .2, .02, .02, .02] + [.014] * 10).astype(np.uint8))
    noise = rng.integers(0, 256, 200_000, dtype=np.uint8).tobytes()
    runs = b"".join(bytes([int(v)]) * int(n) for v, n in zip(rng.integers(0, 256, 400), rng.integers(1, 600, 400)))
    cases = []
    for level in (0, 1, 6, 9):
        for src in (text, noise, runs):
            cases.append(make_bgzf([src[i:i + 65280] for i in range(0, len(src), 65280)], level))
    cases.append(make_bgzf([text[:65280]], 6, zlib.Z_FIXED))
    cases.append(make_bgzf([text[:3000]], 6, zlib.Z_HUFFMAN_ONLY))
    cases.append(make_bgzf([b"", b"A", b"", text[:17], b"\0" * 65280]))
    # a member whose stream holds several DEFLATE blocks (sync flushes add empty stored blocks in between)
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    parts = co.compress(text[:20000]) + co.flush(zlib.Z_FULL_FLUSH) + co.compress(noise[:3000]) + co.flush(zlib.Z_SYNC_FLUSH) + co.compress(runs[:9000]) + co.flush()
    chunk = text[:20000] + noise[:3000] + runs[:9000]
    multi = (b"\x1f\x8b\x08\x04" + b"\0" * 4 + b"\0\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, 18 + len(parts) + 8 - 1) + parts +
             struct.pack("<II", zlib.crc32(chunk), len(chunk)))
    cases.append(multi + make_bgzf([]))
    with engine.HipVariantCaller(_abi.default_config()) as c:
        for data in cases:
            got, blocks, _ = c.bgzf_inflate(data)
            assert got == gzip.decompress(data)
        good = bytearray(cases[4])
        blocks, _ = engine.bgzf_scan(bytes(good))
        bad = bytearray(good)
        o = blocks[1].in_offset
        bad[o: o + 40] = bytes(40)   # block 1's stream: garbage
        with pytest.raises(engine.PiscesHipError, match="block 1"):
            c.bgzf_inflate(bytes(bad))
        bad = bytearray(good)
        bad[blocks[0].in_offset + blocks[0].in_length] ^= 0xFF   # first CRC byte of block 0
        with pytest.raises(engine.PiscesHipError, match="CRC-32 mismatch in block 0"):
            c.bgzf_inflate(bytes(bad))
        got, _, _ = c.bgzf_inflate(bytes(bad), check_crc=False)
        assert got == gzip.decompress(bytes(good))


@pytest.mark.gpu
def test_device_inflate_survives_corrupted_streams():
    """300 mutated copies of valid members (bit flips, zeroed and randomised stretches, truncated payloads with the table left as it
    was): every call returns - with an error naming a block, or with exactly zlib's bytes when the damage missed the stream - and the
    next call on the same handle still works.  (The decoder's loops are bounded by ISIZE and by the payload length.)"""
    rng = np.random.default_rng(11)
    text = bytes(rng.choice(list(b"ACGT\n0123456789"), 150_000).astype(np.uint8))
    base = [make_bgzf([text[i:i + 50_000] for i in range(0, len(text), 50_000)], lvl, st)
            for lvl, st in ((6, zlib.Z_DEFAULT_STRATEGY), (1, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_FIXED), (0, zlib.Z_DEFAULT_STRATEGY))]
    errors = same = 0
    with engine.HipVariantCaller(_abi.default_config()) as c:
        for trial in range(300):
            good = base[trial % len(base)]
            blocks, _ = engine.bgzf_scan(good)
            b = blocks[int(rng.integers(0, len(blocks) - 1))]
            bad = bytearray(good)
            kind = trial % 3
            at = b.in_offset + int(rng.integers(0, max(b.in_length - 8, 1)))
            if kind == 0:
                bad[at] ^= 1 << int(rng.integers(0, 8))
            elif kind == 1:
                n = int(rng.integers(1, 64))
                bad[at:at + n] = rng.integers(0, 256, n, dtype=np.uint8).tobytes()[: len(bad[at:at + n])]
            else:
                bad[at:b.in_offset + b.in_length] = bytes(b.in_offset + b.in_length - at)
            try:
                got, _, _ = c.bgzf_inflate(bytes(bad))
                assert got == gzip.decompress(good)
                same += 1
            except engine.PiscesHipError as e:
                assert "block" in str(e)
                errors += 1
        got, _, _ = c.bgzf_inflate(base[0])
        assert got == text
    assert errors > 250 and errors + same == 300


# ---------------------------------------------------------------- BAM records cut on the device (rest of row f4)
def _bam_reads_reference(file_bytes):
    """The reads of a BAM file by a plain host reader (the minimal BGZF / BAM reader of tests/golden/extract_bam_fixture.py)."""
    import importlib.util
    import tempfile
    spec = importlib.util.spec_from_file_location("extract_bam_fixture", os.path.join(os.path.dirname(__file__), "golden", "extract_bam_fixture.py"))
    xb = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(xb)
    with tempfile.NamedTemporaryFile(suffix=".bam") as f:
        f.write(bytes(file_bytes))
        f.flush()
        return xb.read_bam(f.name)


def _kept(reads, chrom, min_mapq=1, skip_dups=True):
    """AlignmentSource.ShouldSkipRead (AlignmentsSource.cs:84-92) as extract_bam_fixture.extract applies it"""
    keep = []
    for r in reads:
        if r["ref"] != chrom:
            continue
        if (r["flag"] & 0x4) or (r["flag"] & 0x100) or (skip_dups and (r["flag"] & 0x400)) or r["mapq"] < min_mapq or not r["cigar"]:
            continue
        keep.append(r)
    return keep


@pytest.mark.gpu
@pytest.mark.parametrize("name,chrom", [("PhiX_S3", "phix"), ("Sample_S1", "chr19"), ("small_S1", "chr1"), ("Ins_L3_var12_S12", None)])
def test_bam_records_are_cut_on_the_device(name, chrom):
    """pisces_hip_bam_decode on four BAM files of the reference's tests: the device-built read batch (record chain cut by pointer
    jumping, ShouldSkipRead, 4-bit bases / CIGAR words / qualities decoded by bam_decode_kernel) equals, array by array, what a
    plain host reader makes of the same bytes, for every reference sequence of the file and for two filter settings."""
    import torch
    assert torch.cuda.is_available()
    from pisces_amd import engine
    data = _FIXTURES[name]
    refs, reads = _bam_reads_reference(data)
    chroms = [chrom] if chrom else sorted({r["ref"] for r in reads if r["ref"] is not None})[:3]
    n_checked = 0
    with engine.HipVariantCaller(_abi.default_config()) as c:
        for ch in chroms:
            for min_mapq, skip_dups in ((1, True), (0, False)):
                keep = _kept(reads, ch, min_mapq, skip_dups)
                counts = c.bam_decode(data, refs.index(ch), min_mapq, skip_dups)
                assert counts["reads"] == len(keep)
                assert counts["reads"] + counts["skipped"] == sum(1 for r in reads if r["ref"] == ch)
                got = c.bam_fetch()
                np.testing.assert_array_equal(got["position"], np.array([r["pos"] for r in keep], np.int32))
                np.testing.assert_array_equal(got["flags"], np.array([1 if r["flag"] & 0x10 else 0 for r in keep], np.uint8))
                ops = [(ord(o), l) for r in keep for o, l in r["cigar"]]
                np.testing.assert_array_equal(got["cigar_op"], np.array([o for o, _ in ops], np.uint8))
                np.testing.assert_array_equal(got["cigar_len"], np.array([l for _, l in ops], np.uint32))
                np.testing.assert_array_equal(got["cigar_offset"], np.cumsum([0] + [len(r["cigar"]) for r in keep]).astype(np.int32))
                np.testing.assert_array_equal(got["seq_offset"], np.cumsum([0] + [len(r["seq"]) for r in keep]).astype(np.int32))
                assert got["bases"].tobytes() == "".join(r["seq"] for r in keep).encode()
                assert got["quals"].tobytes() == b"".join(r["qual"].tobytes() for r in keep)
                n_checked += len(keep)
    assert n_checked > 0


@pytest.mark.gpu
@pytest.mark.parametrize("name,bam", [("bam_phix", "PhiX_S3"), ("bam_small_s1", "small_S1"), ("bam_chr19", "Sample_S1"), ("bam_chr19", "Chr17Chr19"),
                                      ("bam_chr17_again", "Chr17Chr19"), ("bam_chr17_int", "Chr17Chr19"), ("bam_chr17_vcf", "Chr17Chr19"),
                                      ("bam_edge_ins", "edgeIns_S2"), ("bam_edge_del", "edgeIndel_S2")])
def test_bam_bytes_to_vcf_rows_without_the_reads_leaving_the_device(name, bam):
    """The whole upstream path on the device: compressed BAM bytes in (the only bulk PCIe traffic), BGZF inflate, records cut and
    decoded, read walk + candidate discovery straight from the decoded batch (pisces_hip_add_decoded_reads), calls, VCF text — the
    body lines Pisces wrote for these BAMs (every BAM of the reference's tests that has such rows beside it: tests/bam_fixtures.py), and the
    records of the host-fed path on the same reads.  A BAM's positions are the chromosome's: the reference is the fixture's window behind
    as many N as lie in front of it."""
    import torch
    assert torch.cuda.is_available()
    from pisces_amd import engine
    from tests import bam_fixtures
    case = bam_fixtures.CASES[name]
    z, batch = bam_fixtures.load(name)
    off = int(z["offset"])
    data = _FIXTURES[bam]
    refs, _ = _bam_reads_reference(data)
    cfg = _abi.default_config(**case["cfg"])
    ref = np.concatenate([np.full(off + int(z["ref_start"]) - 1, ord("N"), dtype=np.uint8), z["ref"]])
    with engine.HipVariantCaller(cfg) as c:
        c.SetReference(ref)
        if case["intervals"]:
            c.SetIntervals(case["intervals"])
        counts = c.bam_decode(data, refs.index(case["chrom"]))
        assert counts["reads"] == batch.n_reads
        c.AddDecodedReads()
        got, got_alleles = c.CallWithAlleles()
        stats = c.Stats()
    with engine.HipVariantCaller(cfg) as c:   # the host-fed path on the fixture's reads (positions relative to the fixture's offset)
        c.SetReference(z["ref"] if int(z["ref_start"]) == 1 else np.concatenate([np.full(int(z["ref_start"]) - 1, ord("N"), dtype=np.uint8), z["ref"]]))
        if case["intervals"]:
            c.SetIntervals([(a - off, b - off) for a, b in case["intervals"]])
        c.AddAlleleCounts(batch)
        want, want_alleles = c.CallWithAlleles()
    want = want.copy()
    want["position"] += off
    assert got.tobytes() == want.tobytes() and got_alleles == want_alleles and stats["reads"] == batch.n_reads
    text = engine.format_vcf(case["chrom"], got, alleles=got_alleles, noise_level_from_records=1, **case["vcf"])
    bam_fixtures.check_lines(case, text.rstrip("\n").split("\n") if text else [], bam_fixtures.expected_lines(name, z))


def _synthetic_bam(read_lens, header_text=b"@HD\tVN:1.6\n", seed=5):
    """An uncompressed BAM stream: header + one record per entry of read_lens (CIGAR <len>M, random bases / qualities)."""
    rng = np.random.default_rng(seed)
    out = bytearray(b"BAM\x01" + struct.pack("<i", len(header_text)) + header_text + struct.pack("<i", 1) + struct.pack("<i", 5) + b"chr1\0" +
                    struct.pack("<i", 250_000_000))
    pos = 1000
    for i, n in enumerate(read_lens):
        name = b"r%07d\0" % i
        seq = rng.integers(0, 4, n)
        packed = np.zeros((n + 1) // 2, np.uint8)
        codes = np.array([1, 2, 4, 8], np.uint8)[seq]
        packed[: n // 2] = (codes[0:n - n % 2:2] << 4) | codes[1::2]
        if n % 2:
            packed[-1] = codes[-1] << 4
        qual = rng.integers(2, 42, n).astype(np.uint8)
        body = struct.pack("<iiBBHHHiiii", 0, pos, len(name), 60, 4681, 1, 16 if i % 2 else 0, n, -1, -1, 0) + name + struct.pack("<I", (n << 4) | 0) + \
            packed.tobytes() + qual.tobytes()
        out += struct.pack("<i", len(body)) + body
        pos += int(rng.integers(0, 3))
    return bytes(out)


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["short_records", "long_header", "long_records", "last_record_straddles", "mixed"])
def test_record_chain_is_cut_the_same_with_guessed_and_with_hopped_entries(case):
    """bam_entry_guess_kernel / bam_entry_check_kernel take every chunk's entry from the shared exit of the chunk before it and fall back
    to the serial hop when a chunk has none (records longer than the 4 KiB the guess looks at); a header that ends deep inside a chunk
    and a last record that ends in a chunk of its own are taken without it: either way the read batch is what a plain host reader makes
    of the bytes."""
    import torch
    assert torch.cuda.is_available()
    rng = np.random.default_rng(11)
    if case == "short_records":
        lens, header = [150] * 3000, b"@HD\tVN:1.6\n"
    elif case == "long_header":
        lens, header = [150] * 2000, b"@HD\tVN:1.6\n" + b"".join(b"@CO\tline %06d of a long header\n" % i for i in range(2500))
    elif case == "long_records":
        lens, header = [int(x) for x in rng.integers(3000, 12000, 120)], b"@HD\tVN:1.6\n"
    elif case == "last_record_straddles":
        lens, header = [150] * 1000, b"@HD\tVN:1.6\n"
    else:
        lens, header = [int(x) for x in rng.choice([36, 150, 151, 250, 5000, 9000], 1500, p=[.2, .4, .2, .15, .03, .02])], b"@HD\tVN:1.6\n"
    stream = _synthetic_bam(lens, header)
    if case == "last_record_straddles":
        # pad the header so that the last record begins a few bytes before a 32 KiB boundary and ends behind it
        rec = 4 + 32 + 9 + 4 + 75 + 150
        short = (32768 - (len(stream) - rec) % 32768 - 10) % 32768
        stream = _synthetic_bam(lens, header + b"@CO\t" + b"x" * (short - 5) + b"\n")
        assert (len(stream) - rec) % 32768 == 32768 - 10
    data = make_bgzf([stream[i:i + 60000] for i in range(0, len(stream), 60000)])
    refs, reads = _bam_reads_reference(data)
    keep = _kept(reads, "chr1")
    assert len(keep) == len(lens)
    modes = []
    for force in ("0", "1"):
        os.environ["PISCES_HIP_BAM_SERIAL_CHAIN"] = force
        try:
            with engine.HipVariantCaller(_abi.default_config()) as c:
                counts = c.bam_decode(data, 0)
                assert counts["reads"] == len(keep)
                modes.append(counts["chain"])
                got = c.bam_fetch()
        finally:
            os.environ.pop("PISCES_HIP_BAM_SERIAL_CHAIN", None)
        np.testing.assert_array_equal(got["position"], np.array([r["pos"] for r in keep], np.int32))
        np.testing.assert_array_equal(got["seq_offset"], np.cumsum([0] + [len(r["seq"]) for r in keep]).astype(np.int32))
        assert got["bases"].tobytes() == "".join(r["seq"] for r in keep).encode()
        assert got["quals"].tobytes() == b"".join(r["qual"].tobytes() for r in keep)
    assert modes[1] == "hopped" and (modes[0] == "guessed" or case in ("long_records", "mixed")), modes


@pytest.mark.gpu
def test_bam_decode_survives_corrupted_records():
    """Bytes of the inflated BAM stream overwritten at random (block sizes, field lengths, CIGAR counts): pisces_hip_bam_decode either
    refuses the stream with a message, or gives the reads a plain host reader makes of the same bytes — it never reads outside the
    stream (a record must hold what its fields announce before any kernel takes its arrays by those lengths)."""
    import torch
    assert torch.cuda.is_available()
    rng = np.random.default_rng(23)
    good = _synthetic_bam([150] * 400 + [36, 5000, 151] * 20)
    refused = same = 0
    with engine.HipVariantCaller(_abi.default_config()) as c:
        for trial in range(80):
            bad = bytearray(good)
            for _ in range(int(rng.integers(1, 4))):
                at = int(rng.integers(60, len(bad) - 8))
                kind = trial % 4
                if kind == 0:
                    bad[at] = int(rng.integers(0, 256))
                elif kind == 1:
                    bad[at:at + 4] = struct.pack("<i", int(rng.integers(-5, 70000)))
                elif kind == 2:
                    bad[at:at + 4] = struct.pack("<I", int(rng.integers(0, 2**32)))
                else:
                    bad[at:at + 2] = struct.pack("<H", int(rng.integers(0, 65536)))
            stream = bytes(bad)
            data = make_bgzf([stream[i:i + 60000] for i in range(0, len(stream), 60000)])
            try:
                counts = c.bam_decode(data, 0)
            except engine.PiscesHipError as e:
                assert "bam_decode" in str(e)
                refused += 1
                continue
            got = c.bam_fetch()
            try:
                refs, reads = _bam_reads_reference(data)
                keep = _kept(reads, "chr1")
            except Exception:
                continue   # the plain reader gave up on a stream the device took: nothing to compare with
            if counts["reads"] == len(keep):
                np.testing.assert_array_equal(got["position"], np.array([r["pos"] for r in keep], np.int32))
                same += 1
    assert refused + same > 40 and same > 5, (refused, same)


@pytest.mark.gpu
def test_device_inflate_fuzz_against_zlib():
    """A few thousand BGZF members of random make -- alphabet size, run structure, repeats at short and long distances, every zlib
    level and strategy (default, filtered, Huffman only, RLE, fixed codes), lengths from 0 to the 65 280 bytes a member may hold --
    inflated on the device in a handful of launches and compared with zlib byte for byte.  (The decoder takes the payload in groups
    of 64 bit offsets with the next two groups looked up ahead: members of every length put block ends, long codes and length /
    distance pairs at every offset of a group.)"""
    import torch
    assert torch.cuda.is_available()
    rng = np.random.default_rng(77)
    strategies = [zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED]

    def payload():
        kind = int(rng.integers(0, 6))
        n = int(rng.choice([0, 1, 2, 3, 7, 64, 257, 1000, 5000, 20000, 65280])) if rng.random() < 0.5 else int(rng.integers(0, 65281))
        if kind == 0:      # small alphabet (quality strings)
            return bytes(rng.integers(33, 33 + int(rng.integers(1, 8)), n, dtype=np.uint8))
        if kind == 1:      # all byte values, skewed: codes longer than the look-up tables' indexes
            p = 1.0 / (np.arange(256) + 1.0) ** float(rng.uniform(0.5, 2.5))
            return bytes(rng.choice(256, n, p=p / p.sum()).astype(np.uint8))
        if kind == 2:      # runs (distance 1 and short periods)
            out = bytearray()
            while len(out) < n:
                out += bytes(rng.integers(0, 256, int(rng.integers(1, 5)), dtype=np.uint8)) * int(rng.integers(1, 400))
            return bytes(out[:n])
        if kind == 3:      # records with a shared prefix (repeats at a fixed distance, as BAM records have)
            rec = bytes(rng.integers(0, 256, int(rng.integers(8, 60)), dtype=np.uint8))
            out = bytearray()
            while len(out) < n:
                out += rec + bytes(rng.integers(0, 256, int(rng.integers(4, 300)), dtype=np.uint8))
            return bytes(out[:n])
        if kind == 4:      # long-distance repeats
            base = bytes(rng.integers(0, 256, int(rng.integers(100, 30000)), dtype=np.uint8))
            return (base * (n // max(len(base), 1) + 1))[:n]
        return bytes(rng.integers(0, 256, n, dtype=np.uint8))   # noise

    with engine.HipVariantCaller(_abi.default_config()) as c:
        for launch in range(6):
            members, want = [], []
            for _ in range(400):
                src = payload()
                level = int(rng.integers(1, 10))
                strategy = strategies[int(rng.integers(0, len(strategies)))]
                members.append(make_bgzf([src], level, strategy)[:-28])   # (without the empty end-of-file member make_bgzf appends)
                want.append(src)
            data = b"".join(members) + make_bgzf([])
            got, blocks, _ = c.bgzf_inflate(data)
            assert len(blocks) == len(members) + 1
            assert got == b"".join(want)


# ---------------------------------------------------------------- stitched reads (XD tag) through the BAM surface
def _string_tag(tags, name):
    """BamAlignment.GetStringTag on a record's auxiliary bytes (SAM specification 4.2.4)."""
    p, size = 0, {ord("A"): 1, ord("c"): 1, ord("C"): 1, ord("s"): 2, ord("S"): 2, ord("i"): 4, ord("I"): 4, ord("f"): 4}
    while p + 3 <= len(tags):
        tag, ty = tags[p:p + 2], tags[p + 2]
        p += 3
        if ty in (ord("Z"), ord("H")):
            q = tags.index(b"\0", p)
            if tag == name and ty == ord("Z"):
                return tags[p:q].decode()
            p = q + 1
        elif ty == ord("B"):
            sub, count = tags[p], int.from_bytes(tags[p + 1:p + 5], "little")
            p += 5 + count * size[sub]
        else:
            p += size[ty]
    return None


def _expected_directions(keep):
    """PiscesReadBatch.directions / deletion_directions of reads parsed on the host: Read.SequencedBaseDirectionMap from the XD tag
    (Read.cs:340-400, 664-682; _abi.directions_from_xd), the strand's direction for a read without the tag."""
    dirs, dd = [], []
    for r in keep:
        xd = _string_tag(r["tags"], b"XD")
        if xd is None:
            dirs += [_abi.DIR_REVERSE if r["flag"] & 0x10 else _abi.DIR_FORWARD] * len(r["seq"])
            dd += [_abi.DIR_UNTRACKED] * (2 * len(r["cigar"]))
        else:
            d, e = _abi.directions_from_xd(xd, r["cigar"])
            dirs += list(d) + [_abi.DIR_FORWARD] * (len(r["seq"]) - len(d))
            dd += [x for pair in e for x in pair]
    return np.array(dirs, np.uint8), np.array(dd, np.uint8)


def _stitched_synthetic_bam(rng, n_reads=400, bad_read=None):
    """A BAM whose reads carry XD tags (stitched pairs: F / S / R runs over the expanded CIGAR, deletions included) between other
    auxiliary fields of every value type; a third of the reads carry no XD tag.  bad_read: that read's tag is malformed."""
    import struct
    hdr = b"BAM\x01" + struct.pack("<i", 0) + struct.pack("<i", 1) + struct.pack("<i", 5) + b"chr1\0" + struct.pack("<i", 1_000_000)
    code = {c: i for i, c in enumerate("=ACMGRSVTWYHKDBN")}
    out, pos = [hdr], 500
    for i in range(n_reads):
        pos += int(rng.integers(0, 8))
        ops = [("M", int(rng.integers(20, 60)))]
        if i % 3 == 1:
            ops += [("D", int(rng.integers(1, 7))), ("M", int(rng.integers(10, 40)))]
        if i % 3 == 2:
            ops = [("S", 3)] + ops + [("I", int(rng.integers(1, 5))), ("M", int(rng.integers(10, 40)))]
        l_seq = sum(l for o, l in ops if o in "MIS")
        span = sum(l for o, l in ops)
        seq = "".join(rng.choice(list("ACGT"), l_seq))
        qual = rng.choice([12, 30, 38], l_seq, p=[.05, .3, .65]).astype(np.uint8).tobytes()
        aux = b"NMC" + bytes([int(rng.integers(0, 5))]) + b"ZBBs" + struct.pack("<i", 3) + struct.pack("<3h", 1, -2, 3) + b"MDZ" + b"12A3\0" + b"ASi" + struct.pack("<i", 77)
        if i % 3 != 0:
            a = int(rng.integers(1, span // 2))
            b = int(rng.integers(1, span - a)) if span - a > 1 else 0
            xd = f"{a}F" + (f"{span - a - b}S" if span - a - b > 0 else "") + (f"{b}R" if b > 0 else "")
            if bad_read == i:
                xd = f"{a}F7"                                  # digits without a direction
            aux += b"XDZ" + xd.encode() + b"\0" + b"XRZFR\0"
        aux += b"XZf" + struct.pack("<f", 1.5)
        name = b"r%05d\0" % i
        cig = b"".join(struct.pack("<I", (l << 4) | "MIDNSHP=X".index(o)) for o, l in ops)
        packed = bytearray((l_seq + 1) // 2)
        for k, ch in enumerate(seq):
            packed[k >> 1] |= code[ch] << (4 if k % 2 == 0 else 0)
        flag = 16 if i % 2 else 0
        body = struct.pack("<iiBBHHHiiii", 0, pos - 1, len(name), 60, 0, len(ops), flag, l_seq, -1, -1, 0) + name + cig + bytes(packed) + qual + aux
        out.append(struct.pack("<i", len(body)) + body)
    return _bgzf_of(b"".join(out))


def _bgzf_of(stream, chunk=60000):
    return make_bgzf([stream[i:i + chunk] for i in range(0, len(stream), chunk)])


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["reference fixture", "synthetic"])
def test_stitched_reads_through_the_bam_surface(which):
    """A stitched BAM (the Stitcher's XD tag: one DirectionType per base of the expanded CIGAR) through pisces_hip_bam_decode +
    pisces_hip_add_decoded_reads: the per-base directions and the directions inside deletions the device makes of the tags equal
    Read.SequencedBaseDirectionMap / CigarDirections restated on the host, and counts, candidates and records equal the host-fed
    path's with the same direction maps (DirectionType.Stitched columns included)."""
    import torch
    assert torch.cuda.is_available()
    from pisces_amd import engine
    if which == "reference fixture":
        data = np.load(os.path.join(os.path.dirname(__file__), "golden", "bam_stitched.npz"))["collapsed_test_stitched"].tobytes()
        chrom = "chr1"
    else:
        data, chrom = _stitched_synthetic_bam(np.random.default_rng(8)), "chr1"
    refs, reads = _bam_reads_reference(data)
    keep = _kept(reads, chrom)
    assert keep and any(_string_tag(r["tags"], b"XD") for r in keep)
    want_dirs, want_dd = _expected_directions(keep)
    lo, hi = min(r["pos"] for r in keep), max(r["pos"] + 400 for r in keep)
    ref = np.frombuffer(bytes(np.random.default_rng(1).choice(list(b"ACGT"), hi + 100).astype(np.uint8)), dtype=np.uint8)
    cfg = _abi.default_config(expect_stitched_reads=1, min_coverage=1, low_depth_filter=1)
    batch = _abi.ReadBatch([{"pos": r["pos"], "cigar": r["cigar"], "seq": r["seq"], "quals": r["qual"].tolist(), "reverse": bool(r["flag"] & 0x10),
                             "xd": _string_tag(r["tags"], b"XD")} for r in keep])
    with engine.HipVariantCaller(cfg) as c:
        c.SetReference(ref)
        counts = c.bam_decode(data, refs.index(chrom))
        assert counts["reads"] == len(keep)
        got = c.bam_fetch_directions()
        assert got is not None
        np.testing.assert_array_equal(got[0], want_dirs)
        np.testing.assert_array_equal(got[1], want_dd)
        c.AddDecodedReads()
        got_counts = c.GetCounts(lo, hi - lo)
        got_recs, got_alleles = c.CallWithAlleles()
    with engine.HipVariantCaller(cfg) as c:
        c.SetReference(ref)
        c.AddAlleleCounts(batch)
        want_counts = c.GetCounts(lo, hi - lo)
        want_recs, want_alleles = c.CallWithAlleles()
    np.testing.assert_array_equal(got_counts, want_counts)
    assert got_counts.reshape(-1, 6, 3, 11)[:, :, _abi.DIR_STITCHED, :].sum() > 0
    assert got_recs.tobytes() == want_recs.tobytes() and got_alleles == want_alleles and len(got_recs) > 100
    if which == "synthetic":   # a malformed tag is refused as CigarDirection's constructor refuses it (CigarDirection.cs:37-40)
        bad = _stitched_synthetic_bam(np.random.default_rng(8), bad_read=7)
        with engine.HipVariantCaller(cfg) as c:
            c.SetReference(ref)
            c.bam_decode(bad, 0)
            with pytest.raises(engine.PiscesHipError) as e:
                c.AddDecodedReads()
            assert e.value.code == _abi.E_INVALID_ARG and "direction string" in e.value.message
        plain = _synthetic_bam([100] * 50)   # no XD tag anywhere: no direction arrays are made
        with engine.HipVariantCaller(cfg) as c:
            c.bam_decode(_bgzf_of(bytes(plain)), 0)
            assert c.bam_fetch_directions() is None


@pytest.mark.gpu
def test_a_second_decode_right_behind_an_add_does_not_disturb_the_first_batch_s_candidates():
    """MNV calling on: the candidate walk of a batch is counted at the add and its records are written by the next entry
    (finish_candidate_discovery) — from the arrays of the batch, which for a small decoded batch are the decode's own buffers.  A second
    pisces_hip_bam_decode overwrites those: it must first let the pending half run.  Two small BAMs decoded and added back to back against
    (a) the same with the candidates taken between them and (b) the reads of both files added from host arrays."""
    import torch
    assert torch.cuda.is_available()
    rng = np.random.default_rng(3)
    streams = [_synthetic_bam([120] * 400, seed=s) for s in (5, 6)]
    files = [make_bgzf([st[i:i + 60000] for i in range(0, len(st), 60000)]) for st in streams]
    ref = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, 3000)]
    cfg = _abi.default_config(call_mnvs=1, max_mnv_length=3, max_gap_between_mnv=1, min_base_call_quality=20)
    outs = []
    for how in ("back to back", "candidates taken in between", "host arrays"):
        os.environ["PISCES_HIP_STORE_DIRECT_BYTES"] = str(1 << 40)   # every batch joins the open segment: its arrays stay the decode's
        try:
            c = engine.HipVariantCaller(cfg)
        finally:
            os.environ.pop("PISCES_HIP_STORE_DIRECT_BYTES", None)
        if True:
            with c:
                c.SetReference(ref)
                for data in files:
                    c.bam_decode(data, 0)
                    if how == "host arrays":
                        c.AddAlleleCounts(_abi.ReadBatch.from_arrays(**c.bam_fetch()))
                    else:
                        c.AddDecodedReads()
                        if how != "back to back":
                            assert len(c.GetCandidates(None)) > 100
                rows, alleles = c.CallWithAlleles(None, capacity=1 << 16)
                outs.append((rows, alleles, c.Stats()))
    assert len(outs[0][0]) > 1000 and len(outs[0][1]) > 0
    for o in outs[1:]:
        assert o[0].tobytes() == outs[0][0].tobytes() and o[1] == outs[0][1] and o[2] == outs[0][2]
