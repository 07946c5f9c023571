#!/usr/bin/env python3
"""Copies what tools/profile_round.sh (and the bench runs beside it) left under gpurun_out/<tag>/ into profiles/ under the round's names:
    python tools/collect_profiles.py r04"""
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
src = os.path.join("gpurun_out", tag)
dst = "profiles"
plain = {"bench_n1.json": "bench_n1.json", "bench_kernel_stats.csv": "bench_kernel_stats.csv", "streaming_kernel_stats.csv": "streaming_kernel_stats.csv",
         "bam_kernel_stats.csv": "bam_kernel_stats.csv", "store_store_kernel_stats.csv": "store_store_kernel_stats.csv",
         "store_log_kernel_stats.csv": "store_log_kernel_stats.csv", "config3_kernel_stats.csv": "config3_blocks_kernel_stats.csv",
         "config3_batch_kernel_stats.csv": "config3_kernel_stats.csv", "config3_full.json": "config3_full.json", "config5_full.json": "config5_full.json", "config5_kernel_stats.csv": "config5_kernel_stats.csv",
         "config4.json": "config4.json", "bench_n1_k20.json": "bench_n1_k20.json", "bench_n2_one_device_b.json": "bench_n2_one_device.json",
         "store_timing.txt": "store_timing.txt", "bgzf_bench.txt": "bgzf_bench.txt", "chain2_kernel_stats.csv": "chain_config2_kernel_stats.csv",
         "chain5_kernel_stats.csv": "chain_config5_kernel_stats.csv", "chain.txt": "chain_bench.txt", "pair_bench.txt": "pair_bench.txt"}
for a, b in plain.items():
    if os.path.exists(os.path.join(src, a)):
        shutil.copyfile(os.path.join(src, a), os.path.join(dst, f"{tag}_{b}"))


def cat(names, out):
    with open(os.path.join(dst, f"{tag}_{out}"), "w") as f:
        for n in names:
            p = os.path.join(src, n)
            if os.path.exists(p):
                f.write(f"==== {n}\n" + open(p).read() + "\n")


cat(["pmc_FETCH_SIZE.txt", "pmc_WRITE_SIZE.txt", "pmc_sq_lds.txt", "pmc_sq_wave.txt"], "sq_counters.txt")
cat(["pmc_store_fetch.txt", "pmc_store_write.txt", "pmc_store_calib.txt", "pmc_store_sq.txt", "pmc_store_lds.txt"], "store_counters.txt")
cat(["pmc_bgzf_mem.txt", "pmc_bgzf_sq.txt", "pmc_bgzf_lds.txt"], "bgzf_counters.txt")
cat(["pmc_chain_fetch.txt", "pmc_chain_write.txt"], "chain_counters.txt")
for name, out in (("hostfed.log", "streaming_bench.txt"), ("bam_bench.log", "bam_bench.txt")):
    p = os.path.join(src, name)
    if os.path.exists(p):
        keep = [l for l in open(p) if not l.startswith(("E2", "W2", "I2")) and "rocprofv3" not in l]
        open(os.path.join(dst, f"{tag}_{out}"), "w").writelines(keep[-40:])
p = os.path.join(src, "gpu_suite_full.log")
if os.path.exists(p):
    lines = [l for l in open(p) if ("passed" in l or "failed" in l or "FAILED" in l or "skipped" in l)]
    target = os.path.join(dst, f"{tag}_gpu_suites.txt")
    if os.path.exists(target):   # the round's record of its suite runs is kept by hand: show the new run, leave the file
        print("gpu suite of this run (add it to %s): %s" % (target, "".join(lines[-2:]).strip()))
    else:
        open(target, "w").writelines(["python -m pytest tests -m gpu -q  (one MI355X)\n"] + lines[-6:])
if os.path.exists(os.path.join(src, "traffic.json")):
    shutil.copyfile(os.path.join(src, "traffic.json"), os.path.join(dst, "traffic.json"))
print(sorted(f for f in os.listdir(dst) if f.startswith(tag)))
