#!/usr/bin/env python3
"""Makes tests/golden/bam_stitched.npz: the bytes of a stitched BAM the reference's tests hold (reads with the Stitcher's XD / XR tags),
as data.  Run in the build container, where /root/reference is mounted:  python tests/golden/extract_stitched_bam.py"""
import os
import numpy as np

ROOT = os.environ.get("PISCES_REFERENCE", "/root/reference")
SRC = "src/test/Pisces.Tests/TestData/collapsed.test.stitched.bam"
data = np.frombuffer(open(os.path.join(ROOT, SRC), "rb").read(), dtype=np.uint8)
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "bam_stitched.npz"), collapsed_test_stitched=data, source=np.array(SRC))
print(len(data), "bytes")
