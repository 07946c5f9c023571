#!/usr/bin/env python3
"""Development: what rocprofv3's FETCH_SIZE reports for a known byte count at the two load widths the kernels here use — 16 bytes a lane
(call_tiles_wave_kernel: MI355X_MICROARCH.md says the counter shows half) and 4 bytes a lane (call_store_tiles_kernel: not calibrated there).
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --kernel-include-regex read_probe ... -- python tools/fetch_calibration.py
reads 1 GiB (past the 256 MiB Infinity Cache) with each; the counter of each kernel / 1 GiB is that width's factor."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pisces_amd import _abi, engine

with engine.HipVariantCaller(_abi.default_config()) as c:
    wide = c.probe_read_bandwidth(1 << 30, 3)
    os.environ["PISCES_HIP_PROBE_DWORD"] = "1"
    narrow = c.probe_read_bandwidth(1 << 30, 3)
print(f"fetch_calibration: 1 GiB read with 16 B / lane {wide:.0f} GB/s, with 4 B / lane {narrow:.0f} GB/s")
