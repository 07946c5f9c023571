#!/usr/bin/env python3
"""Development: one of bench.py's config samples alone (for rocprofv3 --hip-trace --kernel-trace): python tools/sample_trace.py 3|4|5"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from pisces_amd import engine
which = sys.argv[1] if len(sys.argv) > 1 else "5"
fn = {"3": bench.config3_sample, "4": bench.config4_sample, "5": bench.config5_sample}[which]
r = fn(engine, torch)
print({k: r[k] for k in ("frac", "value", "seconds", "loci", "reads") if k in r}, {k: r.get(k) for k in ("seconds_in_add", "host_seconds")})
