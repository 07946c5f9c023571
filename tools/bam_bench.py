#!/usr/bin/env python3
"""Development: throughput of pisces_hip_bam_decode (compressed BAM bytes -> device-resident read batch) and of the whole
bytes -> reads -> add_decoded_reads step, on a synthetic single-chromosome BAM of 150-base reads.
    python tools/bam_bench.py [reads]"""
import os, struct, sys, time, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from tools.bgzf_bench import make_bgzf


def make_bam(n_reads, read_len=150, seed=3, copies=1, from_reference=False):
    """header + n_reads * copies records; copies > 1: the same reads again, each copy moved behind the one before it on the chromosome
    (only the position bytes differ: random bases and qualities are what takes the time to make, and DEFLATE's 32 KiB window does not
    see from one copy into the next).  from_reference: the reads are windows of a random reference with 0.5 % substituted bases (a BAM
    of a sample that looks like its genome: about one called allele a locus) instead of random bases (every base a mismatch), and the
    reference comes back with the bytes: (stream, reference letters, 1-based)."""
    rng = np.random.default_rng(seed)
    hdr = b"BAM\x01" + struct.pack("<i", 0) + struct.pack("<i", 1) + struct.pack("<i", 5) + b"chr1\0" + struct.pack("<i", 250_000_000)
    name = b"read/0000000\0"
    fixed = 32 + len(name) + 4 + (read_len + 1) // 2 + read_len
    rec = np.zeros((n_reads, 4 + fixed), dtype=np.uint8)
    pos = np.sort(rng.integers(1000, 1000 + n_reads // 3 + 1000, n_reads)).astype(np.int32)

    def put32(col, v):
        rec[:, col:col + 4] = np.asarray(v, dtype="<i4").reshape(-1, 1).view(np.uint8).reshape(-1, 4) if np.ndim(v) else np.frombuffer(struct.pack("<i", v), np.uint8)

    put32(0, fixed)
    put32(4, 0)                          # refID
    put32(8, pos)
    rec[:, 12] = len(name)
    rec[:, 13] = 60                      # mapq
    rec[:, 16:18] = np.frombuffer(struct.pack("<H", 1), np.uint8)     # n_cigar_op
    flags = np.where(rng.random(n_reads) < 0.5, 16, 0).astype("<u2")
    rec[:, 18:20] = flags.reshape(-1, 1).view(np.uint8).reshape(-1, 2)
    put32(20, read_len)
    put32(24, -1); put32(28, -1); put32(32, 0)
    rec[:, 36:36 + len(name)] = np.frombuffer(name, np.uint8)
    c0 = 36 + len(name)
    rec[:, c0:c0 + 4] = np.frombuffer(struct.pack("<I", (read_len << 4) | 0), np.uint8)
    fixed_span = n_reads // 3 + 2000 + read_len + 1     # positions lie in [1000, 1000 + n_reads // 3 + 1000)
    reference = None
    if from_reference:
        unit = rng.integers(0, 4, fixed_span).astype(np.uint8)
        letters = unit[(pos[:, None] - 1000) + np.arange(read_len)[None, :]]
        wrong = rng.random((n_reads, read_len)) < 0.005
        letters = np.where(wrong, (letters + rng.integers(1, 4, (n_reads, read_len))) & 3, letters).astype(np.uint8)
        codes = np.array([1, 2, 4, 8], dtype=np.uint8)[letters]
        reference = np.frombuffer(b"ACGT", np.uint8)[np.concatenate([rng.integers(0, 4, 1000).astype(np.uint8), np.tile(unit, copies), rng.integers(0, 4, 1000).astype(np.uint8)])]
    else:
        codes = np.array([1, 2, 4, 8], dtype=np.uint8)[rng.integers(0, 4, (n_reads, read_len))]
    rec[:, c0 + 4:c0 + 4 + read_len // 2] = (codes[:, 0::2] << 4) | codes[:, 1::2]
    q0 = c0 + 4 + (read_len + 1) // 2
    rec[:, q0:q0 + read_len] = rng.choice(np.array([12, 23, 30, 37, 41], dtype=np.uint8), (n_reads, read_len), p=[.03, .12, .2, .45, .2])
    if copies == 1 and not from_reference:
        return hdr + rec.tobytes()
    parts, span = [hdr], (fixed_span if from_reference else int(pos[-1]) - 1000 + read_len + 1)
    for k in range(copies):
        put32(8, pos + k * span)
        parts.append(rec.tobytes())
    return (b"".join(parts), reference) if from_reference else b"".join(parts)


def bam_of_read_batch(rb, chrom=b"chr1", chrom_len=250_000_000):
    """An uncompressed BAM stream (header + records) of a pisces_amd._abi.ReadBatch whose reads all have the CIGAR <len>M (the
    synthetic amplicon reads): what samtools would write for them, mapping quality 60, no tags."""
    n = rb.n_reads
    seq_off = np.asarray(rb.seq_offset)
    lens = np.diff(seq_off)
    read_len = int(lens[0])
    assert (lens == read_len).all() and (np.diff(np.asarray(rb.cigar_offset)) == 1).all() and (np.asarray(rb.cigar_op) == ord("M")).all()
    hdr = b"BAM\x01" + struct.pack("<i", 0) + struct.pack("<i", 1) + struct.pack("<i", len(chrom) + 1) + chrom + b"\0" + struct.pack("<i", chrom_len)
    name = b"read/0000000\0"
    fixed = 32 + len(name) + 4 + (read_len + 1) // 2 + read_len
    rec = np.zeros((n, 4 + fixed), dtype=np.uint8)

    def put32(col, v):
        rec[:, col:col + 4] = np.asarray(v, dtype="<i4").reshape(-1, 1).view(np.uint8).reshape(-1, 4) if np.ndim(v) else np.frombuffer(struct.pack("<i", v), np.uint8)

    put32(0, fixed)
    put32(4, 0)
    put32(8, np.asarray(rb.position, dtype=np.int32) - 1)   # BAM positions are 0-based
    rec[:, 12] = len(name)
    rec[:, 13] = 60
    rec[:, 16:18] = np.frombuffer(struct.pack("<H", 1), np.uint8)
    flags = np.where(np.asarray(rb.flags) & 1, 16, 0).astype("<u2")
    rec[:, 18:20] = flags.reshape(-1, 1).view(np.uint8).reshape(-1, 2)
    put32(20, read_len)
    put32(24, -1); put32(28, -1); put32(32, 0)
    rec[:, 36:36 + len(name)] = np.frombuffer(name, np.uint8)
    c0 = 36 + len(name)
    rec[:, c0:c0 + 4] = np.frombuffer(struct.pack("<I", (read_len << 4) | 0), np.uint8)
    code = np.full(256, 15, dtype=np.uint8)
    for ch, v in ((b"A", 1), (b"C", 2), (b"G", 4), (b"T", 8)):
        code[ch[0]] = v
    codes = code[np.asarray(rb.bases).reshape(n, read_len)]
    if read_len % 2:
        codes = np.concatenate([codes, np.zeros((n, 1), np.uint8)], axis=1)
    rec[:, c0 + 4:c0 + 4 + (read_len + 1) // 2] = (codes[:, 0::2] << 4) | codes[:, 1::2]
    q0 = c0 + 4 + (read_len + 1) // 2
    rec[:, q0:q0 + read_len] = np.asarray(rb.quals).reshape(n, read_len)
    return hdr + rec.tobytes()


def main():
    from pisces_amd import _abi, engine
    n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 400_000
    stream = make_bam(n_reads)
    data = make_bgzf(stream, 6)
    with engine.HipVariantCaller(_abi.default_config()) as c:
        c.SetReference(np.full(1000 + n_reads // 3 + 3000, ord("A"), np.uint8))
        staged = data
        for rep in range(3):
            t0 = time.perf_counter()
            counts = c.bam_decode(staged, 0)
            t1 = time.perf_counter()
            c.AddDecodedReads()
            c.synchronize()
            t2 = time.perf_counter()
            c.Call(None, capacity=1 << 20)
        assert counts["reads"] == n_reads
        t0z = time.perf_counter()
        nb = len(zlib.decompress(data[18:18 + 60000], -15)) if False else 0
    print(f"bam decode: {n_reads} reads, {len(stream)/1e6:.0f} MB inflated from {len(data)/1e6:.0f} MB: bytes -> device read batch {1e3*(t1-t0):.1f} ms "
          f"({n_reads/(t1-t0)/1e6:.1f} M reads/s, {len(stream)/(t1-t0)/1e9:.2f} GB/s inflated), + read walk on the decoded batch {1e3*(t2-t1):.1f} ms")


if __name__ == "__main__":
    main()
