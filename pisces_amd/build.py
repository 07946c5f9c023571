"""Builds libpisceship.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

hipcc cross-compiles without a GPU.  -ffp-contract=off: the call phase restates C#
double/float arithmetic that never fuses multiply-add (see csrc/device_math.hip.h).
"""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libpisceship.so")
SOURCES = ["pisces_hip.hip", "expander.cpp", "finder.cpp", "vcf_format.cpp", "diploid.cpp"]
HEADERS = ["kernels.hip.h", "stream_kernels.hip.h", "store_kernels.hip.h", "surface_store.inc.h", "finder_kernels.hip.h", "bgzf_kernels.hip.h", "bam_kernels.hip.h", "device_math.hip.h", "surface_reads.inc.h", "surface_flush.inc.h", "surface_device.inc.h", "surface_bam.inc.h", "surface_comm.inc.h", "expander.h", "read_walk.h", "finder.h", "finder_walk.h", "diploid.h", "genotype_core.h", os.path.join("..", "..", "include", "pisces_hip.h")]


def hipcc_path():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm: /opt/rocm/bin/hipcc)")


def is_stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def source_hash():
    """sha1 over every source and header the library is built from: what a committed measurement (profiles/traffic.json) is stamped with,
    so that a figure collected on other kernels is recognised as stale whatever the files' mtimes say."""
    import hashlib
    hsh = hashlib.sha1()
    for f in sorted(SOURCES + HEADERS):
        with open(os.path.join(CSRC, f), "rb") as fp:
            hsh.update(f.encode() + b"\0" + fp.read())
    return hsh.hexdigest()


LAST_BUILD = {"mode": None, "seconds": 0.0}   # what the last build_native of the product library did: "rebuilt" or "reused"


def build_native(force=False, verbose=False, extra_flags=(), out=None):
    """out != None builds a development variant (e.g. an ablation) next to the product library."""
    import time
    lib = LIB if out is None else out
    if out is None and not force and not is_stale():
        LAST_BUILD.update(mode="reused", seconds=0.0)
        return lib
    t0 = time.time()
    tmp = f"{lib}.{os.getpid()}.tmp"
    cmd = [hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
           "-fno-fast-math", "-Wall", "-Wno-unused-function", "-x", "hip",
           *extra_flags, *[os.path.join(CSRC, s) for s in SOURCES], "-lpthread", "-ldl", "-o", tmp]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    os.replace(tmp, lib)
    if out is None:
        LAST_BUILD.update(mode="rebuilt", seconds=time.time() - t0)
    return lib


if __name__ == "__main__":
    import sys
    build_native(force="--force" in sys.argv, verbose=True)
    print(LIB)
