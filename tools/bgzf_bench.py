#!/usr/bin/env python3
"""Development: throughput of pisces_hip_bgzf_inflate (kernel time from HIP events) against zlib on one host core."""
import os, struct, sys, time, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def make_bgzf(src, level):
    out = bytearray()
    for i in list(range(0, len(src), 65280)) + [None]:
        chunk = b"" if i is None else src[i:i + 65280]
        co = zlib.compressobj(level, zlib.DEFLATED, -15)
        payload = co.compress(chunk) + co.flush()
        out += b"\x1f\x8b\x08\x04" + b"\0" * 4 + b"\0\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, 18 + len(payload) + 8 - 1)
        out += payload + struct.pack("<II", zlib.crc32(chunk), len(chunk))
    return bytes(out)


def main():
    from pisces_amd import _abi, engine
    mb = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    rng = np.random.default_rng(1)
    # BAM-like bytes: 4-bit packed bases (high entropy), qualities from a small alphabet, small integers
    n = mb << 20
    seq = rng.integers(0, 256, n // 3, dtype=np.uint8)
    qual = rng.choice(np.array([12, 23, 30, 37, 41], dtype=np.uint8), n // 2, p=[.03, .12, .2, .45, .2])
    core = np.tile(rng.integers(0, 40, 4096, dtype=np.uint8), n // 6 // 4096 + 1)[: n - len(seq) - len(qual)]
    parts = [seq, qual, core]
    src = b"".join(np.concatenate([p[i:i + 21760] for p in parts]).tobytes() for i in range(0, max(len(p) for p in parts), 21760))
    data = make_bgzf(src, 6)
    with engine.HipVariantCaller(_abi.default_config()) as c:
        got, blocks, ms = c.bgzf_inflate(data, check_crc=False)
        got, blocks, ms = c.bgzf_inflate(data, check_crc=False)
        t0 = time.perf_counter()
        c.bgzf_inflate(data, check_crc=False)
        wall_nocrc = time.perf_counter() - t0
        t0 = time.perf_counter()
        got2, _, _ = c.bgzf_inflate(data, check_crc=os.environ.get("BGZF_BENCH_NOVERIFY") != "1")
        wall = time.perf_counter() - t0
    assert got == src or os.environ.get("BGZF_BENCH_NOVERIFY") == "1"   # (ablation builds produce no bytes)
    t0 = time.perf_counter()
    sample = blocks[: max(1, len(blocks) // 8)]
    nb = sum(len(zlib.decompress(data[b.in_offset:b.in_offset + b.in_length], -15)) for b in sample)
    cpu = nb / (time.perf_counter() - t0)
    print(f"bgzf inflate: {len(src)/1e6:.0f} MB inflated from {len(data)/1e6:.0f} MB in {len(blocks)} blocks: kernel {ms:.2f} ms = "
          f"{len(src)/ms/1e6:.2f} GB/s inflated; call incl. PCIe both ways {wall_nocrc*1e3:.0f} ms, with the CRC check {wall*1e3:.0f} ms; zlib on one core {cpu/1e6:.0f} MB/s")


if __name__ == "__main__":
    main()
