#!/usr/bin/env python3
"""Makes tests/golden/bgzf_fixtures.npz: the bytes of BAM files the reference's own tests hold, as data (inputs of the BGZF inflate /
BAM decode tests, tests/test_bgzf.py).  Run in the build container, where /root/reference is mounted:
    python tests/golden/extract_bam_bytes.py"""
import os

import numpy as np

ROOT = os.environ.get("PISCES_REFERENCE", "/root/reference")
FILES = {
    "Sample_S1": "src/test/Pisces.Tests/TestData/Sample_S1.bam",
    "PhiX_S3": "src/test/SharedData/Bams/PhiX_S3.bam",
    "small_S1": "src/test/SharedData/Bams/small_S1.bam",
    "Ins_L3_var12_S12": "src/test/Pisces.IO.Tests/TestData/Ins-L3-var12_S12.bam",
    "Chr17Chr19": "src/test/Pisces.Tests/TestData/Chr17Chr19.bam",
    "edgeIns_S2": "src/test/Pisces.Tests/TestData/edgeIns_S2.bam",
    "edgeIndel_S2": "src/test/Pisces.Tests/TestData/edgeIndel_S2.bam",
}
arrays = {k: np.frombuffer(open(os.path.join(ROOT, v), "rb").read(), dtype=np.uint8) for k, v in FILES.items()}
dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bgzf_fixtures.npz")
np.savez_compressed(dst, **arrays)
print({k: len(v) for k, v in arrays.items()}, os.path.getsize(dst), "bytes")
