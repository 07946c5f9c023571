#!/usr/bin/env python3
"""bench.py — candidate loci/s of the pileup-and-likelihood hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A step is ONE pass of the hot path (pisces_hip_call_tiles: observation tuples -> LDS allele-count
histograms -> coverage / Poisson q-score / strand bias / somatic genotype / filters -> 64-byte called-allele
records) over one synthetic batch of BASELINE.json config 2: 100 000 loci x 500x amplicon pileup, SNV-only,
gVCF on, reference defaults.  Inputs are resident in HBM when the timed region starts.  Batches rotate through
a ring of distinct pileups whose total size exceeds the 256 MiB Infinity Cache, so every step streams from HBM.
Loci shard by interval across ranks (weak scaling: every rank owns its own 100k-locus shard per step); the only
collective is one RCCL all-reduce of the int64[4] per-chromosome summary at the end of the timed region.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` and `cpu_baseline` objects.
"""
import argparse
import json
import os
import sys
import time

# Streams of independent batches only overlap when each has its own hardware queue; the ROCm runtime shares 4 per process by default.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
# Kernel arguments in device memory instead of host memory: every wave's first instructions read the argument block, and from host
# memory that is a PCIe round trip per wave at the head of a 38 us launch of 3572 waves (measured: 40.1 -> 38.5 us per launch).
# Must be in the environment before the HIP runtime starts; INTEGRATION.md asks the host process for the same.
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md); ~6.3 TB/s achievable
N_LOCI = 100_000
DEPTH = 500
RING_BATCHES = 6        # 6 x ~200 MB of tuples > 256 MiB Infinity Cache
BASE_SEED = 20260928
TIME_EVERY = 1          # the second pass (below) puts dispatch-bound HIP events on every launch: the durations then are those rocprofv3
                        # reports for the same command.  The timed region itself carries no per-launch events (they cost 5-10 us of every
                        # step: rounds 1-2 quoted `value` with them in) — one HIP event in front of it and one behind it on the launch stream
PIPELINE_STREAMS = 3    # the extra 'pipelined' figure: pisces_hip_call_tiles_batched spreads the steps over its 3 lanes


def algorithmic_bytes(n_obs, n_loci, n_records):
    """4 B per observation tuple + 1 reference byte per locus + 64 B per called-allele record."""
    return 4 * n_obs + n_loci + 64 * n_records


def host_cpu():
    """(model name, physical cores, hardware threads) of the box from /proc/cpuinfo."""
    model, cores = "unknown", set()
    try:
        phys = core = None
        for line in open("/proc/cpuinfo"):
            k, _, v = line.partition(":")
            k, v = k.strip(), v.strip()
            if k == "model name":
                model = v
            elif k == "physical id":
                phys = v
            elif k == "core id":
                core = v
            elif not k and phys is not None:
                cores.add((phys, core))
                phys = core = None
        if phys is not None:
            cores.add((phys, core))
    except OSError:
        pass
    threads = os.cpu_count() or 1
    return model, (len(cores) or threads), threads


def cpu_baseline(torch, pileup, cfg, budget_s=14.0):
    """The oracle (CPU restatement of the reference C# path: per read FindCandidates -> AddCandidates ->
    AddAlleleCounts, then AlleleCaller over every locus) on a bounded sample of the same workload.
    `cpu_baseline`: single-threaded — the reference runs one thread per (BAM, chromosome).
    `cpu_baseline_threads`: the -threadbychr analogue, one thread per interval shard on every host core."""
    import numpy as np
    from pisces_amd import synth
    from tests import orc   # the oracle is test infrastructure; here it is only the thing being timed
    ref = pileup.ref.cpu().numpy()
    # probe with 4 amplicons (600 loci) to size a ~budget_s sample
    probe = synth.reads_of(pileup, 4)
    t0 = time.perf_counter()
    orc.run_reads(probe, ref, pileup.region_start, 4 * synth.READ_LEN, cfg)
    dt = max(time.perf_counter() - t0, 1e-4)
    n_amp_total = pileup.base.shape[0]
    n_amp = int(min(n_amp_total, max(4, budget_s / (dt / 4))))
    batch = synth.reads_of(pileup, n_amp)
    n_loci = min(pileup.n_loci, n_amp * synth.READ_LEN)
    est = dt / 4 * n_amp
    repeats = int(max(1, min(20, round(0.6 * budget_s / max(est, 1e-3)))))   # whole batch is only seconds: repeat it
    t0 = time.perf_counter()
    loci = 0
    for _ in range(repeats):
        _, n = orc.run_reads(batch, ref, pileup.region_start, n_loci, cfg)
        loci += n
    dt1 = time.perf_counter() - t0
    model, phys_cores, hw_threads = host_cpu()
    single = {"value": loci / dt1, "unit": "candidate loci/s", "cores": 1, "kind": "port", "cpu": model,
              "sample": f"first {n_loci} loci x {pileup.depth}x of batch 0 ({batch.n_reads} reads) x {repeats} passes, {dt1:.1f} s"}

    # interval shards, one thread each (ctypes releases the GIL inside the oracle)
    cores = phys_cores   # one thread per physical core, not per hardware thread
    shards = []
    for idx in np.array_split(np.arange(n_amp_total), min(cores, n_amp_total)):
        a0, na = int(idx[0]), len(idx)
        b = synth.reads_of(pileup, na, first_amplicon=pileup.first_amplicon + a0)
        start = pileup.region_start + a0 * synth.READ_LEN
        nl = min(pileup.n_loci - a0 * synth.READ_LEN, na * synth.READ_LEN)
        shards.append((b, start, nl))
    passes = int(max(1, min(40, round(0.5 * budget_s * cores / max(dt / 4 * n_amp_total, 1e-3)))))
    t0 = time.perf_counter()
    _, loci_n = orc.run_reads_sharded(shards, ref, cfg, passes=passes)   # pthreads inside the oracle library
    dtn = time.perf_counter() - t0
    multi = {"value": loci_n / dtn, "unit": "candidate loci/s", "cores": len(shards), "kind": "port", "cpu": model,
             "physical_cores": phys_cores, "hardware_threads": hw_threads,
             "sample": f"batch 0 ({pileup.n_loci} loci x {pileup.depth}x) in {len(shards)} interval shards, one thread each, x {passes} passes, {dtn:.1f} s"}
    return single, multi


def _bgzf_with_offsets(stream, block=0xFF00):
    """BGZF of `stream` (DEFLATE level 1) + the start of every block in the file and in the stream: virtual offsets for a .bai."""
    import struct
    import zlib
    out, file_at, stream_at = [], [], []
    at = 0
    for o in range(0, len(stream), block):
        raw = stream[o:o + block]
        co = zlib.compressobj(1, zlib.DEFLATED, -15)
        body = co.compress(raw) + co.flush()
        blk = (b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", len(body) + 25) + body + struct.pack("<II", zlib.crc32(raw), len(raw)))
        file_at.append(at)
        stream_at.append(o)
        out.append(blk)
        at += len(blk)
    out.append(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"))   # the EOF block
    return b"".join(out), file_at, stream_at


def reference_csharp_baseline(pileup, cfg, n_amplicons=40, timeout_s=120):
    """SURVEY 8d / BASELINE.md 3.2: when the box has a `dotnet` host AND PISCES_REF_DLL names a Pisces build, the reference C# itself is timed
    on a bounded sample of the same reads (a synthetic BAM + .bai + genome folder written here; `dotnet Pisces.dll -bam .. -g .. -t 1`,
    BaseApplication.cs:141-151 prints what it did; the wall clock of the process is what is reported).  Neither exists in the image this
    was written in: the probe's outcome is part of the bench line either way, and the port stays the baseline then."""
    import shutil
    import struct
    import subprocess
    import tempfile
    import numpy as np
    from pisces_amd import synth
    dotnet, dll = shutil.which("dotnet"), os.environ.get("PISCES_REF_DLL")
    probe = {"dotnet": dotnet, "PISCES_REF_DLL": dll, "ran": False}
    if not dotnet or not dll or not os.path.exists(dll):
        probe["why"] = "no `dotnet` on PATH" if not dotnet else "PISCES_REF_DLL is not set" if not dll else "PISCES_REF_DLL does not exist"
        return probe
    try:
        from tools.bam_bench import bam_of_read_batch
        n_amp = min(n_amplicons, pileup.base.shape[0])
        rb = synth.reads_of(pileup, n_amp)
        ref = pileup.ref.cpu().numpy()
        chrom_len = int(pileup.ref_start - 1 + len(ref))
        stream = bam_of_read_batch(rb, chrom=b"chr1", chrom_len=chrom_len)
        data, file_at, stream_at = _bgzf_with_offsets(stream)
        # record starts in the stream -> virtual offsets; one chunk per bin, the linear index per 16 KiB window (SAM spec 5.2)
        n = rb.n_reads
        rec_len = (len(stream) - (12 + 4 + 5 + 4)) // max(n, 1)
        starts = len(stream) - n * rec_len + np.arange(n + 1, dtype=np.int64) * rec_len
        blk = np.searchsorted(np.asarray(stream_at), starts, side="right") - 1
        voff = (np.asarray(file_at, dtype=np.int64)[blk] << 16) | (starts - np.asarray(stream_at, dtype=np.int64)[blk])
        beg = np.asarray(rb.position, dtype=np.int64) - 1
        end = beg + np.diff(np.asarray(rb.seq_offset))

        def reg2bin(b, e):
            e = e - 1
            for shift, base in ((14, 4681), (17, 585), (20, 73), (23, 9), (26, 1)):
                if b >> shift == e >> shift:
                    return base + (b >> shift)
            return 0
        bins = {}
        for i in range(n):
            k = reg2bin(int(beg[i]), int(end[i]))
            lo, hi = bins.get(k, (int(voff[i]), int(voff[i + 1])))
            bins[k] = (min(lo, int(voff[i])), max(hi, int(voff[i + 1])))
        n_intv = int(end.max() >> 14) + 1
        lin = np.zeros(n_intv, dtype=np.uint64)
        for i in range(n):
            for w in range(int(beg[i]) >> 14, (int(end[i]) - 1 >> 14) + 1):
                if lin[w] == 0 or voff[i] < lin[w]:
                    lin[w] = voff[i]
        bai = b"BAI\x01" + struct.pack("<ii", 1, len(bins))
        for k in sorted(bins):
            bai += struct.pack("<IiQQ", k, 1, bins[k][0], bins[k][1])
        bai += struct.pack("<i", n_intv) + lin.tobytes() + struct.pack("<Q", 0)
        with tempfile.TemporaryDirectory() as tmp:
            gdir = os.path.join(tmp, "genome")
            os.mkdir(gdir)
            seq = (b"N" * (pileup.ref_start - 1) + ref.tobytes()).decode()
            with open(os.path.join(gdir, "genome.fa"), "w") as f:
                f.write(">chr1\n" + "\n".join(seq[i:i + 60] for i in range(0, len(seq), 60)) + "\n")
            with open(os.path.join(gdir, "genome.fa.fai"), "w") as f:
                f.write(f"chr1\t{len(seq)}\t6\t60\t61\n")
            with open(os.path.join(gdir, "GenomeSize.xml"), "w") as f:
                f.write(f'<sequences genomeName="synthetic">\n  <chromosome fileName="genome.fa" contigName="chr1" totalBases="{len(seq)}" isCircular="false" '
                        f'md5="00000000000000000000000000000000" ploidy="2" knownBases="{len(seq)}" />\n</sequences>')
            bam = os.path.join(tmp, "sample.bam")
            open(bam, "wb").write(data)
            open(bam + ".bai", "wb").write(bai)
            cmd = [dotnet, dll, "-bam", bam, "-g", gdir, "-t", "1", "-gvcf", "true", "-minbq", str(cfg.min_base_call_quality), "-OutFolder", os.path.join(tmp, "out")]
            t0 = time.perf_counter()
            run = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s)
            dt = time.perf_counter() - t0
            n_loci = min(pileup.n_loci, n_amp * synth.READ_LEN)
            probe.update(ran=run.returncode == 0, returncode=run.returncode, seconds=dt, command=" ".join(cmd[:2]) + " -bam <synthetic> -g <synthetic> -t 1 -gvcf true",
                         stdout_tail=run.stdout[-300:], stderr_tail=run.stderr[-300:])
            if run.returncode == 0:
                probe["cpu_baseline"] = {"value": n_loci / dt, "unit": "candidate loci/s", "cores": 1, "kind": "reference",
                                         "sample": f"first {n_loci} loci x {pileup.depth}x of batch 0 ({n} reads), dotnet Pisces.dll -t 1, process wall clock {dt:.1f} s (start-up included)"}
    except Exception as e:   # noqa: BLE001  (a probe: it must not cost the bench line)
        probe["why"] = "the reference run failed: " + str(e)[:300]
    return probe


def end_to_end(pileup, cfg, engine, loci=30_000):
    """the end-to-end rate of the drop-in boundary (the streaming surface the C# shim drives), host buffers in,
    called alleles out, H2D / D2H and every host pass included: reads of the first `loci` loci of batch 0 through
    pisces_hip_add_reads + pisces_hip_flush, (a) one pair per 1000-locus block as SmallVariantCaller.Execute does, (b) 30 blocks per
    pair (HipEngine.BlocksPerFlush).  This is the figure comparable to `cpu_baseline` (same scope: reads -> called alleles)."""
    from pisces_amd import synth
    ref = pileup.ref.cpu().numpy()
    n_amp_total = pileup.base.shape[0]
    n_amp = min(n_amp_total, max(7, loci // synth.READ_LEN))
    out = {}
    for label, per_call in (("per_block", 7), ("batched_30_blocks", 200)):   # 7 amplicons = 1050 loci: one block of reads per call
        batches = [(a0, synth.reads_of(pileup, min(per_call, n_amp - a0), first_amplicon=pileup.first_amplicon + a0)) for a0 in range(0, n_amp, per_call)]
        n_loci = min(pileup.n_loci, n_amp * synth.READ_LEN)
        best = None
        with engine.HipVariantCaller(cfg) as c:   # one handle, as one (BAM, chromosome) job has: the first pass pays for its pinned staging
            c.SetReference(ref)                    # and device buffers (grow-only), the later ones are the steady state
            for rep in range(4):
                n_rec = 0
                t0 = time.perf_counter()
                for a0, b in batches:
                    c.AddAlleleCounts(b)
                    n_rec += len(c.CallView(pileup.region_start + a0 * synth.READ_LEN - 1))   # LastClearedPosition
                n_rec += len(c.CallView(None))   # (the rows are read in place, as HipEngine.Flush reads them)
                dt = time.perf_counter() - t0
                if rep > 0:
                    best = dt if best is None else min(best, dt)
        out[label] = {"value": n_loci / best, "unit": "candidate loci/s", "seconds": best, "loci": n_loci, "records": n_rec,
                      "add_reads_flush_pairs": len(batches), "reads": int(sum(b.n_reads for _, b in batches))}
    # (a') block by block through the flush PAIR (pisces_hip_flush_begin / pisces_hip_flush_end): the device works on block k while the
    # host adds the reads of block k + 1; the alleles of block k are taken before block k + 1 is flushed.  Same calls otherwise.
    try:
        per_block = [(a0, synth.reads_of(pileup, min(7, n_amp - a0), first_amplicon=pileup.first_amplicon + a0)) for a0 in range(0, n_amp, 7)]
        for label, stage in (("per_block_pair", False), ("per_block_pair_staged", True)):
            best = None
            with engine.HipVariantCaller(cfg) as c:
                c.SetReference(ref)
                for rep in range(4):
                    n_rec = 0
                    dt = 0.0
                    pending = False
                    for a0, b in per_block:
                        staged = c.StageReads(b) if stage else b       # (filling the staging buffer: the caller's marshalling, not timed)
                        t0 = time.perf_counter()
                        c.AddAlleleCounts(staged)
                        if pending:
                            n_rec += len(c.CallEndView())
                        c.CallBegin(pileup.region_start + a0 * synth.READ_LEN - 1)
                        pending = True
                        dt += time.perf_counter() - t0
                    t0 = time.perf_counter()
                    n_rec += len(c.CallEndView())
                    c.CallBegin(None)
                    n_rec += len(c.CallEndView())
                    dt += time.perf_counter() - t0
                    if rep > 0:
                        best = dt if best is None else min(best, dt)
            assert n_rec == out["per_block"]["records"], (n_rec, out["per_block"]["records"])
            out[label] = {"value": n_loci / best, "unit": "candidate loci/s", "seconds": best, "loci": n_loci, "records": n_rec,
                          "add_reads_flush_pairs": len(per_block)}
    except Exception as e:   # noqa: BLE001
        out["per_block_pair"] = {"error": str(e)[:200]}
    # (b') the 30-block pairs with the reads written by the caller straight into the library's pinned staging buffer
    # (pisces_hip_stage_reads: what a host that marshals its reads anyway does; filling the buffer is that marshalling and is not
    # timed, as making the read arrays is not timed in (a) and (b)): pisces_hip_add_reads then sends the batch without a copy of its own
    try:
        best = None
        with engine.HipVariantCaller(cfg) as c:
            c.SetReference(ref)
            for rep in range(4):
                n_rec = 0
                dt = 0.0
                for a0, b in batches:
                    staged = c.StageReads(b)
                    t0 = time.perf_counter()
                    c.AddAlleleCounts(staged)
                    n_rec += len(c.CallView(pileup.region_start + a0 * synth.READ_LEN - 1))
                    dt += time.perf_counter() - t0
                t0 = time.perf_counter()
                n_rec += len(c.CallView(None))
                dt += time.perf_counter() - t0
                if rep > 0:
                    best = dt if best is None else min(best, dt)
        assert n_rec == out["batched_30_blocks"]["records"]
        out["staged_30_blocks"] = {"value": n_loci / best, "unit": "candidate loci/s", "seconds": best, "loci": n_loci, "records": n_rec,
                                   "scope": "as batched_30_blocks, the read arrays written by the caller into the library's pinned staging buffer "
                                            "(pisces_hip_stage_reads; filling it is the caller's marshalling, not timed)"}
    except Exception as e:   # noqa: BLE001
        out["staged_30_blocks"] = {"error": str(e)[:200]}
    out["scope"] = ("host read buffers -> pisces_hip_add_reads -> pisces_hip_flush_view / _flush_end_view -> records in host memory (the library's pinned buffer, "
                    "read in place as dotnet/HipEngine.cs reads them; PCIe both ways; one handle, best of 3 passes after a warm-up pass)")
    # (c) the same reads as the bytes of a BAM file (BGZF, zlib level 6): inflated, cut into records, filtered, walked and called on the
    # device (pisces_hip_bam_decode -> pisces_hip_add_decoded_reads -> pisces_hip_flush); only the compressed bytes cross PCIe
    try:
        from tools.bam_bench import bam_of_read_batch
        from tools.bgzf_bench import make_bgzf
        rb = synth.reads_of(pileup, n_amp, first_amplicon=pileup.first_amplicon)
        data = make_bgzf(bam_of_read_batch(rb), 6)
        best = None
        with engine.HipVariantCaller(cfg) as c:
            c.SetReference(ref)
            for rep in range(4):
                t0 = time.perf_counter()
                counts = c.bam_decode(data, 0)
                c.AddDecodedReads()
                n_rec = len(c.CallView(None))
                dt = time.perf_counter() - t0
                if rep > 0:
                    best = dt if best is None else min(best, dt)
        assert counts["reads"] == rb.n_reads and n_rec == out["batched_30_blocks"]["records"], (counts, n_rec)
        out["from_bam_bytes"] = {"value": n_loci / best, "unit": "candidate loci/s", "seconds": best, "loci": n_loci, "records": n_rec,
                                 "reads": int(rb.n_reads), "compressed_bytes": len(data),
                                 "scope": "BGZF-compressed BAM bytes on the host -> inflate, record cut, ShouldSkipRead, read walk, calls on the device -> host records"}
    except Exception as e:   # noqa: BLE001  (an extra figure: it must not cost the bench line)
        out["from_bam_bytes"] = {"error": str(e)[:200]}
    return out


def chain_roofline(pileup, cfg, engine, torch, reps=10):
    """`roofline_chain`: the DEVICE time of reads in HBM -> records in HBM through the streaming surface, on all of the configuration the
    metric is quoted on (BASELINE config 2, batch 0: 333 500 reads): pisces_hip_add_device_reads (the caller's arrays into the store, the
    checks, descriptors / fragments / row codes: one launch) + pisces_hip_flush_view (position grid, the flush's kernel, the ordered
    compaction), each span by two HIP events on the handle's stream (pisces_hip_set_chain_timing): from the first kernel an entry point
    enqueues to the last — the host's turns inside a span count, the records' transfer to the host (behind the last event) does not.
    Algorithmic bytes as SURVEY 8d counts the path (2 B per aligned base + 64 B per record).  Reference: what one AddAlleleCounts +
    FindCandidates per read and one GetCandidatesToProcess + Call do (RegionStateManager.cs:118-220, CandidateVariantFinder.cs:36-83)."""
    from pisces_amd import synth
    ref = pileup.ref.cpu().numpy()
    n_amp = pileup.base.shape[0]
    whole = synth.reads_of(pileup, n_amp, first_amplicon=pileup.first_amplicon)
    d = engine.DeviceReadBatch.from_host(whole, "cuda:0")
    spans, walls, n_rec = [], [], 0
    with engine.HipVariantCaller(cfg) as c:
        c.SetReference(ref)
        c.SetChainTiming(True)
        for rep in range(reps + 2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            c.AddDeviceReads(d)
            n_rec = len(c.CallView(None))
            dt = time.perf_counter() - t0
            if rep >= 2:
                spans.append(c.ChainTime())
                walls.append(dt)
        c.SetChainTiming(False)
    nbytes = 2.0 * int(whole.n_bases) + 64.0 * n_rec
    add_ms = sum(a for a, _ in spans) / len(spans)
    flush_ms = sum(f for _, f in spans) / len(spans)
    chain_ms = add_ms + flush_ms
    best = min(a + f for a, f in spans)
    achieved = nbytes / (chain_ms * 1e-3) / 1e9
    traffic, traffic_run = None, None
    try:   # separate rocprofv3 --pmc passes over tools/chain_bench.py (the same pair), tools/profile_round.sh
        from pisces_amd import build as native_build
        whole_file = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
        tj = whole_file.get("chain", {})
        if tj.get("loci") == pileup.n_loci and tj.get("depth") == pileup.depth:
            if whole_file.get("source_hash") == native_build.source_hash():   # (collected on these kernels)
                traffic, traffic_run = tj.get("hbm_bytes_per_batch"), tj.get("run")
            else:
                traffic_run = "stale: profiles/traffic.json was collected on other sources (%s)" % tj.get("run")
    except Exception:   # noqa: BLE001
        pass
    return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_from": traffic_run,
            "chain_ms": chain_ms, "add_device_reads_ms": add_ms, "flush_view_ms": flush_ms, "best_chain_ms": best, "pairs_timed": len(spans),
            "algorithmic_bytes_per_batch": nbytes, "reads": int(whole.n_reads), "records": n_rec, "loci": pileup.n_loci,
            "wall_clock_ms_per_pair": sum(walls) / len(walls) * 1e3,
            "what": "device time (HIP events on the handle's stream, mean of the timed pairs) of pisces_hip_add_device_reads + pisces_hip_flush_view on "
                    "BASELINE config 2's batch: reads in HBM -> compacted records in HBM; the D2H transfer of the records and the host's work outside "
                    "the two spans are in wall_clock_ms_per_pair only"}


def end_to_end_full(pileup, cfg, engine, torch):
    """The streaming surface on ALL of the configuration the metric is quoted on (BASELINE config 2: 100 000 loci x 500x = 333 500 reads
    of batch 0), not a 30 000-locus sample: in one add_reads + flush, and block by block as SmallVariantCaller drives it
    (SmallVariantCaller.cs:88-112,157-189).  The flush's kernel is timed with HIP events bound to its dispatch: `roofline_streaming` is
    the read store's kernel against the bytes the reads -> records path has to move (2 B per aligned base + 64 B per record)."""
    import numpy as np
    from pisces_amd import synth
    ref = pileup.ref.cpu().numpy()
    n_amp = pileup.base.shape[0]
    whole = synth.reads_of(pileup, n_amp, first_amplicon=pileup.first_amplicon)
    n_bases = int(whole.n_bases)
    out, roof = {}, None
    with engine.HipVariantCaller(cfg) as c:
        c.SetReference(ref)
        best, n_rec = None, 0
        for rep in range(13):   # (one warm-up pass, twelve timed: the flush kernel's duration is the mean over their dispatch events)
            if rep == 1:
                c.set_timing(1)
                c.HostTime(reset=True)
            t0 = time.perf_counter()
            c.AddAlleleCounts(whole)
            n_rec = len(c.CallView(None))
            dt = time.perf_counter() - t0
            if rep > 0:
                best = dt if best is None else min(best, dt)
        kernel_ms_total, launches = c.kernel_time()
        c.set_timing(False)
        ht = c.HostTime(reset=True)
        out["one_batch"] = {"value": pileup.n_loci / best, "unit": "candidate loci/s", "seconds": best, "loci": pileup.n_loci, "reads": int(whole.n_reads),
                            "records": n_rec, "host_ms_per_flush": ht["host_ms_per_flush"]}
        if launches:
            kernel_ms = kernel_ms_total / launches
            nbytes = 2.0 * n_bases + 64.0 * n_rec
            achieved = nbytes / (kernel_ms * 1e-3) / 1e9
            traffic, traffic_run = None, None
            try:   # separate rocprofv3 --pmc passes over tools/store_bench.py (the same one-batch flush), tools/profile_round.sh
                from pisces_amd import build as native_build
                whole_file = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
                tj = whole_file.get("streaming", {})
                if tj.get("loci") == pileup.n_loci and tj.get("depth") == pileup.depth:
                    if whole_file.get("source_hash") == native_build.source_hash():   # (collected on these kernels)
                        traffic, traffic_run = tj.get("hbm_bytes_per_launch"), tj.get("run")
                    else:
                        traffic_run = "stale: profiles/traffic.json was collected on other sources (%s)" % tj.get("run")
            except Exception:   # noqa: BLE001
                pass
            roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                    "traffic_from": traffic_run,
                    "kernel": "pisces::call_store_tiles_kernel", "kernel_ms": kernel_ms, "launches_timed": int(launches),
                    "algorithmic_bytes_per_launch": nbytes,
                    "what": "reads in HBM -> LDS histogram -> 64-byte records, one launch per flush.  Algorithmic bytes as SURVEY 8d counts them "
                            "(2 B per aligned base: base + quality, + 64 B per record); since round 5 the kernel itself loads 1 B per base (the row code "
                            "encode_rows made of base and quality when the batch was added) and is instruction-issue-bound (DESIGN.md section 3.10)"}
        per_block = [(a0, synth.reads_of(pileup, min(7, n_amp - a0), first_amplicon=pileup.first_amplicon + a0)) for a0 in range(0, n_amp, 7)]
        best = None
        for rep in range(3):
            if rep == 1:
                c.HostTime(reset=True)
            n_rec = 0
            t0 = time.perf_counter()
            for a0, b in per_block:
                c.AddAlleleCounts(b)
                n_rec += len(c.CallView(pileup.region_start + a0 * synth.READ_LEN - 1))
            n_rec += len(c.CallView(None))
            dt = time.perf_counter() - t0
            if rep > 0:
                best = dt if best is None else min(best, dt)
        ht = c.HostTime(reset=True)
        out["per_block"] = {"value": pileup.n_loci / best, "unit": "candidate loci/s", "seconds": best, "loci": pileup.n_loci, "records": n_rec,
                            "add_reads_flush_pairs": len(per_block), "host_ms_per_flush": ht["host_ms_per_flush"],
                            "host_ms_per_add_reads": ht["add_reads_s"] / max(2 * len(per_block), 1) * 1e3}
        # the same 96 blocks through the flush pair, as dotnet/HipFactory.cs drives the library (HipAlleleCaller.Call = FlushEnd of block k - 1,
        # FlushBegin of block k), and with the caller's reads written into the pinned staging buffer (HipEngine.FlushStagedReads)
        want_records = n_rec
        for label, stage in (("per_block_pair", False), ("per_block_pair_staged", True)):
            best = None
            for rep in range(3):
                n_rec, dt, pending = 0, 0.0, False
                for a0, b in per_block:
                    staged = c.StageReads(b) if stage else b       # (filling the staging buffer: the caller's marshalling, not timed)
                    t0 = time.perf_counter()
                    c.AddAlleleCounts(staged)
                    if pending:
                        n_rec += len(c.CallEndView())
                    c.CallBegin(pileup.region_start + a0 * synth.READ_LEN - 1)
                    pending = True
                    dt += time.perf_counter() - t0
                t0 = time.perf_counter()
                n_rec += len(c.CallEndView())
                c.CallBegin(None)
                n_rec += len(c.CallEndView())
                dt += time.perf_counter() - t0
                if rep > 0:
                    best = dt if best is None else min(best, dt)
            assert n_rec == want_records, (n_rec, want_records)
            out[label] = {"value": pileup.n_loci / best, "unit": "candidate loci/s", "seconds": best, "loci": pileup.n_loci, "records": n_rec,
                          "add_reads_flush_pairs": len(per_block)}
    # BASELINE config 3's mix (SNV + MNV + deletions + insertions at 2000x, MNV calling on) block by block: the flushes whose host half is
    # not empty (candidate merge, VariantCollapser, MnvReallocator between the device passes)
    try:
        seed, depth, amps = 33, 2000, 40
        cfg3 = _abi_config(call_mnvs=1, max_mnv_length=3, max_gap_between_mnv=1)
        n_loci = amps * synth.READ_LEN
        ref3 = synth.reference_of(n_loci, seed, device="cuda")
        p3 = synth.make_pileup(n_loci, depth, seed=seed, device="cuda", first_locus=0, total_loci=n_loci, with_tuples=False)
        batch, planted = synth.mixed_reads(p3, seed)
        with engine.HipVariantCaller(cfg3) as c:
            c.SetReference(ref3)
            best = None
            for rep in range(3):
                if rep == 1:
                    c.HostTime(reset=True)
                    c.TransferBytes(reset=True)
                t0 = time.perf_counter()
                c.AddAlleleCounts(batch)
                n_rec = 0
                for up_to in range(1000, n_loci, 1000):
                    n_rec += len(c.CallView(up_to))
                n_rec += len(c.CallView(None))
                dt = time.perf_counter() - t0
                if rep > 0:
                    best = dt if best is None else min(best, dt)
            ht = c.HostTime(reset=True)
            tb = c.TransferBytes(reset=True)
        per_flush = max(ht["flushes"], 1)
        out["config3_mix_sample"] = {"pcie_bytes_per_flush": {k: v / per_flush for k, v in tb.items()},"value": n_loci / best, "unit": "candidate loci/s", "seconds": best, "loci": n_loci, "depth": depth, "reads": int(batch.n_reads),
                                     "records": n_rec, "planted_events": len(planted), "flushes": ht["flushes"] // 2,
                                     "host_ms_per_flush": ht["host_ms_per_flush"], "device_wait_ms_per_flush": ht["flush_wait_s"] / max(ht["flushes"], 1) * 1e3}
    except Exception as e:   # noqa: BLE001
        out["config3_mix_sample"] = {"error": str(e)[:200]}
    return out, roof


def config3_sample(engine, torch, amps=200):
    """BASELINE config 3's mix on a sample that finishes in seconds (200 amplicons = 30 000 loci x 2000x = 400 000 reads; SNVs + MNVs +
    deletions + insertions, MNV calling on): the reads in DEVICE memory (pisces_hip_add_device_reads) -> records, one add + one flush, best of
    three — the form `python bench.py --config 3` runs over all of the 1 M loci.  roofline: the path's algorithmic bytes (2 B per aligned
    base + 64 B per record) over the wall clock of the pair."""
    from pisces_amd import synth
    seed, depth = 33, 2000
    cfg3 = _abi_config(call_mnvs=1, max_mnv_length=3, max_gap_between_mnv=1)
    n_loci = amps * synth.READ_LEN
    ref3 = synth.reference_of(n_loci, seed, device="cuda")
    p3 = synth.make_pileup(n_loci, depth, seed=seed, device="cuda", first_locus=0, total_loci=n_loci, with_tuples=False)
    batch, planted = synth.mixed_reads(p3, seed)
    dbatch = engine.DeviceReadBatch.from_host(batch, "cuda:0")
    with engine.HipVariantCaller(cfg3) as c:
        c.SetReference(ref3)
        best, best_add, n_rec = None, None, 0
        for rep in range(4):
            if rep == 1:
                c.HostTime(reset=True)
                c.TransferBytes(reset=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            c.AddDeviceReads(dbatch)
            t1 = time.perf_counter()
            n_rec = len(c.CallView(None))
            dt = time.perf_counter() - t0
            if rep > 0 and (best is None or dt < best):
                best, best_add = dt, t1 - t0
        ht = c.HostTime(reset=True)
        tb = c.TransferBytes(reset=True)
    nbytes = 2.0 * batch.n_bases + 64.0 * n_rec
    return {"bound": "hbm", "achieved": nbytes / best / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": nbytes / best / 1e9 / HBM_PEAK_GBS,
            "algorithmic_bytes": nbytes, "value": n_loci / best, "value_unit": "candidate loci/s", "seconds": best, "seconds_in_add": best_add,
            "loci": n_loci, "depth": depth, "reads": int(batch.n_reads), "records": n_rec, "planted_events": len(planted),
            "host_seconds_in_flush_per_flush": ht["host_ms_per_flush"] / 1e3, "pcie_bytes_per_flush": {k: v / max(ht["flushes"], 1) for k, v in tb.items()},
            "what": "BASELINE config 3's mix, a 30 000-locus sample, reads in device memory -> records on the host: pisces_hip_add_device_reads + one "
                    "pisces_hip_flush_view (device checks, read store, candidate discovery + merge, dirty loci, collapser / reallocator, call kernels); wall clock, "
                    "not a per-kernel figure: profiles/r05_config3_kernel_stats.csv has those; all of config 3: python bench.py --config 3"}


def config5_sample(engine, torch, amps=100):
    """BASELINE config 5's settings (0.5 % VAF SNVs at 5000x, -minbq 30 -minvf 0.005 -sbfilter 0.5 -vqfilter 30, gVCF) on a sample that
    finishes in seconds (100 amplicons = 15 000 loci x 5000x = 500 000 reads): reads in DEVICE memory -> records, one add + one flush, best
    of three — the form `python bench.py --config 5` runs over all 100 000 loci.  roofline: the path's algorithmic bytes (2 B per aligned
    base + 64 B per record) over the wall clock of the pair."""
    from pisces_amd import synth
    seed, depth = 23, 5000
    cfg5 = _abi_config(min_base_call_quality=30, noise_level=30, min_frequency=0.005, variant_freq_filter=0.005, genotype_min_freq_filter=0.005,
                       target_lod_frequency=0.005, strand_bias_threshold=0.5, variant_qscore_filter=30)
    n_loci = amps * synth.READ_LEN
    ref5 = synth.reference_of(n_loci, seed, device="cuda")
    p5 = synth.make_pileup(n_loci, depth, seed=seed, device="cuda", first_locus=0, total_loci=n_loci, with_tuples=False,
                           vaf_range=(0.005, 0.005), snv_every=50, snv_offset=17, q_lo=12)
    batch = synth.reads_of(p5, amps, first_amplicon=0)
    dbatch = engine.DeviceReadBatch.from_host(batch, "cuda:0")
    del p5
    with engine.HipVariantCaller(cfg5) as c:
        c.SetReference(ref5)
        best, best_add, n_rec = None, None, 0
        for rep in range(4):
            if rep == 1:
                c.HostTime(reset=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            c.AddDeviceReads(dbatch)
            t1 = time.perf_counter()
            n_rec = len(c.CallView(None))
            dt = time.perf_counter() - t0
            if rep > 0 and (best is None or dt < best):
                best, best_add = dt, t1 - t0
        ht = c.HostTime(reset=True)
    nbytes = 2.0 * batch.n_bases + 64.0 * n_rec
    return {"bound": "hbm", "achieved": nbytes / best / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": nbytes / best / 1e9 / HBM_PEAK_GBS,
            "algorithmic_bytes": nbytes, "value": n_loci / best, "value_unit": "candidate loci/s", "seconds": best, "seconds_in_add": best_add,
            "loci": n_loci, "depth": depth, "reads": int(batch.n_reads), "records": n_rec, "host_seconds_in_flush_per_flush": ht["host_ms_per_flush"] / 1e3,
            "what": "BASELINE config 5's settings, a 15 000-locus sample, reads in device memory -> records on the host: pisces_hip_add_device_reads + one "
                    "pisces_hip_flush_view; wall clock of the pair, not a per-kernel figure; all of config 5: python bench.py --config 5"}


def config4_sample(engine, torch, n_intervals=2000):
    """BASELINE config 4's data (150 bp intervals 300 bp apart at 200x, SNVs + small insertions / deletions, interval set applied,
    zero-coverage rows on) on ONE synthetic contig of 2 000 intervals = 300 000 loci (the whole job is 200 000 intervals on 24 contigs cut
    8 ways: `python bench.py --config 4`, and `config4_strong_scaling` of a multi-GPU line): the contig's reads in device memory, the
    streaming surface in stretches of 400 000 reads as pisces_amd.config4.run_piece drives it, records looked at in place; best of three."""
    from pisces_amd import config4
    cfg4 = _abi_config(emit_zero_coverage_refs=1)
    job = config4.make_contig(0, n_intervals, depth=200, device="cuda:0")
    dev_arrays = config4.device_arrays(job, "cuda:0")
    plan = config4.piece_plan(job, None, None)
    chunks = config4.device_chunks(engine, job, dev_arrays, plan)
    best, recs, stats = None, None, None
    # the contig's handle, made and given the reference once — what a chromosome job does at its start (one SmallVariantCaller per
    # chromosome, Factory.cs:253-269) — and the piece run on it pass after pass: set_intervals + owned range, the reads in stretches, flushes
    with engine.HipVariantCaller(cfg4, device=0) as contig_caller:
        contig_caller.SetReference(job["ref"])
        for rep in range(4):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            recs, _, stats, owned = config4.run_piece(engine, cfg4, job, None, None, device=0, with_alleles=False, keep_records=False, plan=plan, chunks=chunks,
                                                      count_loci=rep == 0, caller=contig_caller)   # (the untimed pass counts the loci from the rows)
            dt = time.perf_counter() - t0
            if rep == 0:
                assert recs["loci"] == config4.plan_loci(plan), (recs, config4.plan_loci(plan))
            recs["loci"] = config4.plan_loci(plan)
            if rep > 0 and (best is None or dt < best):
                best = dt
    n_bases = int(job["batch"].n_bases)
    nbytes = 2.0 * n_bases + 64.0 * recs["n"]
    return {"bound": "hbm", "achieved": nbytes / best / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": nbytes / best / 1e9 / HBM_PEAK_GBS,
            "algorithmic_bytes": nbytes, "value": recs["loci"] / best, "value_unit": "candidate loci/s", "seconds": best, "loci": recs["loci"], "depth": 200,
            "intervals": n_intervals, "reads": int(job["batch"].n_reads), "records": recs["n"],
            "host_seconds": {k: stats["host_time"][k] for k in ("add_reads_s", "flush_s", "flush_wait_s")},
            "what": "BASELINE config 4's data on one contig of 2 000 intervals (300 000 loci x 200x), reads in device memory -> records read in place: "
                    "set_intervals + owned range + add_device_reads / flush_view in stretches on the contig's handle (made and given the reference once, before the "
                    "timed passes: one handle per contig, its pieces in turn); wall clock per GPU"}


def from_large_bam(cfg, engine, reads=400_000, copies=9):
    """the BAM surface on a file of >= 256 MB: 3.6 M reads of 150 bases drawn from a random reference with 0.5 % wrong
    bases at ~450x (tools/bam_bench.make_bam), BGZF at zlib level 1 (1.0 GB inflated, ~265 MB compressed: position-sorted reads of one
    locus repeat each other inside DEFLATE's window, as in a real high-depth BAM), through pisces_hip_bam_decode ->
    pisces_hip_add_decoded_reads -> flush; only the compressed bytes cross PCIe on the way in.  Generation (~25 s of numpy and zlib) is
    not timed."""
    import numpy as np
    from tools.bam_bench import make_bam
    from tools.bgzf_bench import make_bgzf
    stream, ref = make_bam(reads, copies=copies, from_reference=True)
    data = make_bgzf(stream, 1)
    out = {"reads": reads * copies, "compressed_bytes": len(data), "inflated_bytes": len(stream)}
    del stream
    with engine.HipVariantCaller(cfg) as c:
        c.SetReference(ref)
        for label in ("view", "copy"):
            best, n_rec, n_loci = None, 0, 0
            for rep in range(3):
                t0 = time.perf_counter()
                counts = c.bam_decode(data, 0)
                c.AddDecodedReads()
                recs = c.CallView(None) if label == "view" else c.Call(None, capacity=1 << 22, reuse_buffer=True)
                dt = time.perf_counter() - t0
                n_rec = len(recs)
                if rep == 0:
                    n_loci = int(len(np.unique(recs["position"])))
                else:
                    best = dt if best is None else min(best, dt)
            assert counts["reads"] == reads * copies, counts
            out[label] = {"value": n_loci / best, "unit": "candidate loci/s", "seconds": best, "loci": n_loci, "records": n_rec,
                          "compressed_GB_per_s": len(data) / best / 1e9}
    out["scope"] = ("BGZF-compressed BAM bytes on the host -> inflate, record cut, ShouldSkipRead, read store, calls on the device -> records on the host; "
                    "view: read in place in the library's pinned buffer (pisces_hip_flush_view), copy: copied into the caller's array (pisces_hip_flush)")
    return out


def _abi_config(**kw):
    from pisces_amd import _abi
    return _abi.default_config(**kw)


def config4_job(torch, dist, world, rank, local_rank, use_dist):
    """BASELINE config 4 as SURVEY 8d states it (30 M loci x 200x, 200 000 intervals on 24 contigs, SNV + indel, cut 8 ways by interval).
    Rank r of `world` takes the shards r, r + world, ... of the 8-way cut in turn on its own GPU (one process: all eight in turn on
    cuda:0), the per-chromosome totals are all-reduced (RCCL) and the job's rate is loci / the slowest rank's time: STRONG scaling — the
    job is the same whatever the number of GPUs.  Timed: the streaming surface per (contig, range) piece — set_reference, set_intervals,
    the reads in stretches, flushes, host records out.  `value`: a contig's reads lie in device memory when its pieces start
    (pisces_hip_add_device_reads), as section 4 of the task asks of every `value`; `host_fed`: the same pieces from host arrays
    (pisces_hip_add_reads: 12 GB over PCIe).  Making the synthetic contigs is not timed.  Returns the line (rank 0)."""
    import numpy as np
    from pisces_amd import _abi, config4, engine
    dev = torch.device("cuda", local_rank)
    depth, n_shards = 200, 8
    sizes = config4.contig_intervals(200_000)
    cfg = _abi.default_config(emit_zero_coverage_refs=1)
    shards = config4.partition(sizes, n_shards, block_size=cfg.block_size, depth=depth)
    mine = [r for r in range(n_shards) if r % world == rank]
    need = sorted({c for r in mine for c, _, _ in shards[r]})
    t_shard = {r: 0.0 for r in mine}
    t_fed = {r: 0.0 for r in mine}
    loci_shard = {r: 0 for r in mine}
    totals = np.zeros(4, dtype=np.int64)
    lib_time, lib_time_fed = {}, {}
    for c in need:
        job = config4.make_contig(c, sizes[c], depth=depth, device=f"cuda:{local_rank}")
        dev_arrays = config4.device_arrays(job, f"cuda:{local_rank}")      # the contig's reads in HBM before anything is timed (<= 1 GB)
        # one handle per contig (one SmallVariantCaller per chromosome job, Factory.cs:253-269), made and given the reference inside the
        # timed region of the contig's first piece; its pieces run on it in turn
        contig_caller, last_r = None, mine[0]
        for r in mine:
            for cc, lo, hi in shards[r]:
                if cc != c:
                    continue
                plan = config4.piece_plan(job, lo, hi)
                chunks = config4.device_chunks(engine, job, dev_arrays, plan)
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                last_r = r
                if contig_caller is None:
                    contig_caller = engine.HipVariantCaller(cfg, device=local_rank)
                    contig_caller.SetReference(job["ref"])
                recs, _, stats, owned = config4.run_piece(engine, cfg, job, lo, hi, device=local_rank, with_alleles=False, keep_records=False,
                                                          plan=plan, chunks=chunks, count_loci=False, caller=contig_caller)
                t_shard[r] += time.perf_counter() - t0
                # (the loci the piece reports: every position of its clipped intervals has a row — zero-coverage rows are on; counted from
                # the rows themselves in the host-fed pass below, which asserts the two agree: a numpy pass over 30 M rows is the harness, not the path)
                recs["loci"] = config4.plan_loci(plan)
                loci_shard[r] += recs["loci"]
                totals += np.array([stats["TotalNumCalled"], stats["TotalNumCollapsed"], owned, stats["reads_skipped"]])
                for k in ("add_reads_s", "flush_s", "flush_wait_s"):
                    lib_time[k] = lib_time.get(k, 0.0) + stats["host_time"][k]
                del chunks
                # the same piece from host arrays (pisces_hip_add_reads: the reads cross PCIe), the piece's plan made inside the timed region
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                recs_f, _, stats_f, _ = config4.run_piece(engine, cfg, job, lo, hi, device=local_rank, with_alleles=False, keep_records=False)
                t_fed[r] += time.perf_counter() - t0
                assert recs_f == recs and stats_f["TotalNumCalled"] == stats["TotalNumCalled"]
                for k in ("add_reads_s", "flush_s", "flush_wait_s"):
                    lib_time_fed[k] = lib_time_fed.get(k, 0.0) + stats_f["host_time"][k]
        if contig_caller is not None:
            t0 = time.perf_counter()
            contig_caller.close()
            t_shard[last_r] += time.perf_counter() - t0   # (the handle's end belongs to the contig's last piece)
        del job, dev_arrays
    elapsed = sum(t_shard.values())
    summary = torch.tensor(totals.tolist() + [sum(loci_shard.values())], dtype=torch.int64, device=dev)
    t = torch.tensor([elapsed, sum(t_fed.values())], dtype=torch.float64, device=dev)
    per_rank = torch.zeros(world, dtype=torch.float64, device=dev)
    per_rank[rank] = sum(loci_shard.values()) / max(elapsed, 1e-9)
    if use_dist:
        dist.all_reduce(summary, op=dist.ReduceOp.SUM)   # the per-chromosome totals over the shards (RCCL over xGMI)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(per_rank, op=dist.ReduceOp.SUM)
    # the same totals through the C ABI (pisces_hip_comm_* / pisces_hip_reduce_summary: RCCL bound by the library, what a C# host with one
    # process per GPU calls); on a side thread with a time limit: a communicator that cannot be set up costs this field, not the line
    c_abi = None
    if use_dist and world > 1:
        import threading
        box = {}

        def through_the_c_abi():
            try:
                with engine.HipVariantCaller(cfg, device=local_rank) as hc:
                    ids = [engine.HipVariantCaller.comm_unique_id() if rank == 0 else None]
                    dist.broadcast_object_list(ids, src=0)
                    hc.comm_init(ids[0], rank, world)
                    red = hc.reduce_summary([int(x) for x in totals.tolist()])
                    box["r"] = {"ok": red == [int(x) for x in summary[:4].tolist()], "summary": red}
            except Exception as e:   # noqa: BLE001
                box["r"] = {"ok": False, "error": str(e)[:200]}
        th = threading.Thread(target=through_the_c_abi, daemon=True)
        th.start()
        th.join(60.0)
        c_abi = box.get("r", {"ok": False, "error": "no answer within 60 s"})
    if rank != 0:
        return None
    loci = int(summary[4].item())
    out = {"metric": "candidate loci/s at 200x depth, interval-sharded (BASELINE config 4)", "value": loci / float(t[0].item()), "unit": "candidate loci/s",
           "n_gpus": world, "steps": 1, "warmup": 0, "ms_per_step": float(t[0].item()) * 1e3, "higher_is_better": True, "scaling": "strong",
           "vs_baseline": None, "dtype": "int32 counts + f64 likelihoods", "data": "synthetic",
           "config": {"workload": "BASELINE config 4: 30 M loci x 200x over 200 000 intervals of 150 bp on 24 contigs, SNV + indel at 1/10 of config 3's "
                                  "density, cut 8 ways by interval; streaming surface (a contig's reads in device memory when its pieces start, "
                                  "host records out), shards "
                                  + ("in turn on one GPU" if world == 1 else f"over {world} GPUs"),
                      "loci": loci, "reads": int(summary[2].item()), "intervals": 200_000, "contigs": 24, "shards": n_shards},
           "totals": {"allelesCalled": int(summary[0].item()), "variantsCollapsed": int(summary[1].item()), "readsProcessed": int(summary[2].item()),
                      "readsSkipped": int(summary[3].item())},
           "loci_per_s_by_rank": [float(x) for x in per_rank.tolist()],
           "rank0_seconds_inside_the_library": lib_time,
           "host_fed": {"value": loci / float(t[1].item()), "unit": "candidate loci/s", "seconds": float(t[1].item()),
                        "rank0_seconds_inside_the_library": lib_time_fed,
                        "what": "the same pieces from host arrays (pisces_hip_add_reads): 12 GB of reads cross PCIe, and every piece's intervals and "
                                "read range are worked out inside the timed region"},
           "shards_rank0": [{"shard": r, "pieces": len(shards[r]), "loci": loci_shard[r], "seconds": t_shard[r], "loci_per_s": loci_shard[r] / t_shard[r]}
                            for r in mine]}
    if c_abi is not None:
        out["c_abi_reduce"] = c_abi
    assert loci == 30_000_000 and out["totals"]["readsProcessed"] == 200_000 * depth
    return out


def run_config4(args):
    """`python bench.py --config 4` (one process: the eight shards in turn on cuda:0; under torch.distributed.run with N ranks: spread over
    the ranks' GPUs): see config4_job."""
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=dev)
    out = config4_job(torch, dist, world, rank, local_rank, use_dist)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


def run_stream_config(args):
    """BASELINE configs 3 and 5 as SURVEY 8d states them, from READS through the streaming surface (`python bench.py --config 3 | 5`):
      3: 1 M loci x 2000x = 13.3 M reads, SNVs + MNVs (2-3 bases) + deletions (1-10) + insertions (1-6), -callmnvs true -maxmnvlength 3
         -maxgapbetweenmnv 1, gVCF;
      5: 100 000 loci x 5000x = 3.3 M reads, planted 0.5 % VAF SNVs, -minbq 30 (=> NL 30) -minvf 0.005 -sbfilter 0.5 -vqfilter 30, gVCF.
    The amplicons are handed over in stretches (pisces_hip_add_device_reads — the reads lie in device memory when the timed region starts,
    SURVEY 8d — then pisces_hip_flush_view up to the stretch's last cleared position), records out on the host; the same stretches from host
    arrays (pisces_hip_add_reads, PCIe-inclusive) are timed beside it as `host_fed`; making the synthetic reads is not timed.  N ranks: rank r takes the r-th
    contiguous range of amplicons (amplicons do not overlap: no halo), totals all-reduced, `value` = loci / the slowest rank's time."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from pisces_amd import _abi, engine, synth
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=dev)
    if args.config == 3:
        n_loci_all, depth, seed, stretch = 1_000_000, 2000, 33, 200
        cfg = _abi.default_config(call_mnvs=1, max_mnv_length=3, max_gap_between_mnv=1)
        synth_kw = {}
        what = ("BASELINE config 3: 1 M loci x 2000x, SNV + MNV (2-3) + deletions (1-10) + insertions (1-6), MNV calling on, gVCF; "
                "streaming surface from reads (device read store, device candidate discovery and merge, collapser, reallocator, candidate kernel)")
    else:
        n_loci_all, depth, seed, stretch = 100_000, 5000, 23, 200   # (stretches of 1 M reads = 300 MB of bases + qualities: a flush of 30 000 loci fills the chip; 40 amplicons were 94 tiles)
        cfg = _abi.default_config(min_base_call_quality=30, noise_level=30, min_frequency=0.005, variant_freq_filter=0.005, genotype_min_freq_filter=0.005,
                                  target_lod_frequency=0.005, strand_bias_threshold=0.5, variant_qscore_filter=30)
        synth_kw = dict(vaf_range=(0.005, 0.005), snv_every=50, snv_offset=17, q_lo=12)
        what = "BASELINE config 5: 100 000 loci x 5000x, 0.5 % VAF SNVs, -minbq 30 (NL 30) -minvf 0.005 -sbfilter 0.5 -vqfilter 30, gVCF; streaming surface from reads"
    n_amp_all = n_loci_all // synth.READ_LEN
    n_loci_all = n_amp_all * synth.READ_LEN
    a_lo, a_hi = rank * n_amp_all // world, (rank + 1) * n_amp_all // world
    ref = synth.reference_of(n_loci_all, seed, device=f"cuda:{local_rank}")
    origin = synth.READ_LEN + 1
    # The stretches are made first (the reads of all of them stay in device memory: 4 GB for config 3), then timed pass by pass, each pass
    # on a handle of its own: three passes from device memory (pisces_hip_add_device_reads: the metric's form, SURVEY 8d — inputs resident
    # in HBM when the timed region starts; `value` is their median: the path is bound by one host core, and a neighbour on the box shows),
    # then one from host arrays over PCIe (pisces_hip_add_reads: what a host that holds the reads pays; 2 bytes per base at the host
    # link's rate bound it whatever the device does).
    stretches = []
    n_reads = n_bases = 0
    for a0 in range(a_lo, a_hi, stretch):
        na = min(stretch, a_hi - a0)
        p = synth.make_pileup(na * synth.READ_LEN, depth, seed=seed, device=f"cuda:{local_rank}", first_locus=a0 * synth.READ_LEN, total_loci=n_loci_all,
                              with_tuples=False, **synth_kw)
        batch = synth.mixed_reads(p, seed)[0] if args.config == 3 else synth.reads_of(p, na, first_amplicon=a0)
        n_reads += int(batch.n_reads)
        n_bases += int(batch.n_bases)
        stretches.append((batch, engine.DeviceReadBatch.from_host(batch, f"cuda:{local_rank}"), origin + (a0 + na) * synth.READ_LEN - 1))   # (.., the next stretch's first position - 1)
        del p
    torch.cuda.synchronize(dev)

    def one_pass(from_device):
        # SmallVariantCaller's order (SmallVariantCaller.cs:88-105): a read is added, THEN Call(its position - 1) clears what lies behind it.
        # Stretch by stretch: the reads of stretch k + 1 are added, then the flush up to their first position - 1 takes stretch k — the device
        # discovers the candidates of k + 1 while the host works on the flush of k.
        with engine.HipVariantCaller(cfg, device=local_rank) as c:
            c.SetReference(ref)
            torch.cuda.synchronize(dev)
            n_rec, prev_up_to = 0, None
            t0 = time.perf_counter()
            for batch, dbatch, up_to in stretches:
                if from_device:
                    c.AddDeviceReads(dbatch)
                else:
                    c.AddAlleleCounts(batch)
                if prev_up_to is not None:
                    n_rec += len(c.CallView(prev_up_to))
                prev_up_to = up_to
            n_rec += len(c.CallView(None))
            seconds = time.perf_counter() - t0
            return seconds, n_rec, c.Stats(), c.HostTime(), c.TransferBytes()

    passes = [one_pass(True) for _ in range(3)]
    assert all(q[1:3] == passes[0][1:3] for q in passes)
    passes.sort(key=lambda q: q[0])
    elapsed, n_rec, stats, host, pcie = passes[1]
    pass_seconds = [q[0] for q in passes]
    elapsed_h, n_rec_h, stats_h, host_h, pcie_h = one_pass(False)
    assert stats_h == stats and n_rec_h == n_rec

    def handles_in_parallel(n_threads):
        # One handle is bound by ONE host core.  The reference runs a job per chromosome on a thread pool
        # (BaseGenomeProcessor.cs:40-90, JobManager.cs:70-73): here n_threads handles, each over its own contiguous share of the stretches
        # (amplicons do not overlap: the shares' records are the whole's), on n_threads host threads against the one GPU.
        import threading
        bounds = [len(stretches) * k // n_threads for k in range(n_threads + 1)]
        handles = [engine.HipVariantCaller(cfg, device=local_rank) for _ in range(n_threads)]
        for c in handles:
            c.SetReference(ref)
        torch.cuda.synchronize(dev)
        rows, errors = [0] * n_threads, []

        def work(k):
            try:
                c, prev = handles[k], None
                for batch, dbatch, up_to in stretches[bounds[k]:bounds[k + 1]]:
                    c.AddDeviceReads(dbatch)
                    if prev is not None:
                        rows[k] += len(c.CallView(prev))
                    prev = up_to
                rows[k] += len(c.CallView(None))
            except Exception as e:   # noqa: BLE001
                errors.append(repr(e))
        threads = [threading.Thread(target=work, args=(k,)) for k in range(n_threads)]
        t0 = time.perf_counter()
        for th in threads:
            th.start()
        for th in threads:
            th.join()
        seconds = time.perf_counter() - t0
        for c in handles:
            c.close()
        assert not errors and sum(rows) == n_rec, (errors, sum(rows), n_rec)
        return seconds
    par = None
    if world == 1 and args.config == 3 and len(stretches) >= 8:   # (config 5's 17 stretches are 5 ms of device work: nothing for threads to share)
        par = {"threads": 4, "seconds": sorted(handles_in_parallel(4) for _ in range(3))[1]}
    del stretches
    loci_mine = (a_hi - a_lo) * synth.READ_LEN
    summary = torch.tensor([stats["TotalNumCalled"], stats["TotalNumCollapsed"], stats["reads"], stats["reads_skipped"], loci_mine, n_rec], dtype=torch.int64, device=dev)
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(summary, op=dist.ReduceOp.SUM)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        loci = int(summary[4].item())
        # the path's algorithmic bytes: 2 B per aligned base in, 64 B per record out
        algo_bytes = 2.0 * n_bases + 64.0 * n_rec
        out = {"metric": f"candidate loci/s at {depth}x depth from reads (BASELINE config {args.config})", "value": loci / float(t.item()), "unit": "candidate loci/s",
               "n_gpus": world, "steps": 1, "warmup": 0, "ms_per_step": float(t.item()) * 1e3, "higher_is_better": True, "scaling": "strong",
               "vs_baseline": None, "dtype": "int32 counts + f64 likelihoods", "data": "synthetic",
               "config": {"workload": what, "loci": loci, "depth": depth, "reads": int(summary[2].item()), "amplicons_per_add_reads": stretch,
                          "parallelism": f"amplicon ranges x{world}"},
               "totals": {"allelesCalled": int(summary[0].item()), "variantsCollapsed": int(summary[1].item()), "readsProcessed": int(summary[2].item()),
                          "readsSkipped": int(summary[3].item()), "records": int(summary[5].item())},
               "rank0": {"host_seconds_in_add_reads": host["add_reads_s"], "host_seconds_in_flushes": host["flush_s"], "of_those_waiting_for_the_device": host["flush_wait_s"],
                         "flushes": host["flushes"], "pcie_bytes": pcie, "seconds_of_the_three_passes": pass_seconds, "value_is": "their median"},
               f"roofline_config{args.config}": {"bound": "hbm", "achieved": algo_bytes / elapsed / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                                 "frac": algo_bytes / elapsed / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes": algo_bytes,
                                                 "seconds": elapsed, "what": "reads in device memory (2 B per aligned base) -> records (64 B each), wall clock of pisces_hip_add_device_reads + "
                                                 "pisces_hip_flush_view over all stretches on rank 0: candidate discovery, merge, collapser / reallocator on the host and the call kernels "
                                                 "included; per-kernel times: profiles/r05_config3_kernel_stats.csv (config 3), profiles/r05_config5_kernel_stats.csv (config 5)"},
               "host_fed": {"value": loci_mine / elapsed_h, "unit": "candidate loci/s (rank 0)", "seconds": elapsed_h, "host_seconds_in_add_reads": host_h["add_reads_s"],
                            "host_seconds_in_flushes": host_h["flush_s"], "pcie_bytes": pcie_h,
                            "what": "the same stretches from host arrays (pisces_hip_add_reads): the reads cross PCIe, 2 B per base"}}
        if par:
            out["handles_in_parallel"] = {"threads": par["threads"], "value": loci / par["seconds"], "unit": "candidate loci/s", "seconds": par["seconds"],
                                          "what": "the same stretches over four handles on four host threads, one GPU (median of three passes): a job per "
                                                  "chromosome on a thread pool is the reference's own model; one handle is bound by one host core"}
        assert out["totals"]["readsProcessed"] == n_reads or use_dist
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


def launch_ranks(args, argv, device_count=None, run=None):
    """`python bench.py --gpus N` (N > 1) with no launcher around it starts its N ranks ITSELF: the same command line under
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1`, one rank a GPU (the driver's own form;
    under that launcher WORLD_SIZE is set and this returns None: the rank runs).  Fewer devices than ranks is an error, not a silent
    one-rank run — unless PISCES_BENCH_ONE_DEVICE=1 (development: every rank on cuda:0, process group on gloo).  Returns None when this
    process is to run the bench itself, else the exit status of the launcher.  `device_count` / `run` are the test's seams."""
    if args.gpus <= 1 or "WORLD_SIZE" in os.environ:
        if "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) != args.gpus:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={os.environ['WORLD_SIZE']}: the launcher's --nproc-per-node and --gpus must agree",
                  file=sys.stderr)
            return 2
        return None
    if device_count is None:
        from pisces_amd import engine
        device_count = engine.device_count
    have = device_count()
    if have < args.gpus and os.environ.get("PISCES_BENCH_ONE_DEVICE") != "1":
        print(f"bench.py: --gpus {args.gpus} asked for, pisces_hip_device_count() = {have}: not enough MI355X devices on this node "
              f"(PISCES_BENCH_ONE_DEVICE=1 runs the {args.gpus}-rank code path on one device for development; its numbers mean nothing)",
              file=sys.stderr)
        return 3
    import socket
    import subprocess
    with socket.socket() as so:                                     # a free rendezvous port on the loopback
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    print("bench.py: starting " + str(args.gpus) + " ranks: " + " ".join(cmd), file=sys.stderr)
    return (run or subprocess.call)(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=12)
    ap.add_argument("--loci", type=int, default=N_LOCI)
    ap.add_argument("--depth", type=int, default=DEPTH)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-end-to-end", action="store_true", help="skip the streaming-surface figures (host reads -> records)")
    ap.add_argument("--no-large-bam", action="store_true", help="skip the 265 MB BAM figure of end_to_end_full (~25 s to make the file)")
    ap.add_argument("--no-shard-check", action="store_true", help="skip the on-device check of one cut of the interval partition")
    ap.add_argument("--no-pipelined", action="store_true", help="skip the extra multi-stream figure (profiling runs: its overlapped "
                    "launches would mix into the per-kernel statistics of the timed region)")
    ap.add_argument("--config", type=int, default=2, choices=[2, 3, 4, 5],
                    help="2 (default): the configuration the metric is quoted on; 3 / 5: BASELINE configs 3 / 5 as stated, from reads through the "
                         "streaming surface; 4: BASELINE config 4 as stated")
    args = ap.parse_args()
    launched = launch_ranks(args, sys.argv[1:])
    if launched is not None:
        return launched
    if args.config == 4:
        return run_config4(args)
    if args.config in (3, 5):
        return run_stream_config(args)

    import numpy as np
    import torch
    import torch.distributed as dist

    from pisces_amd import _abi, engine, shard, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback exists)"
    # development only: PISCES_BENCH_ONE_DEVICE=1 puts every rank on cuda:0 and the process group on gloo, so that the N > 1 code path
    # (partition, shard check across a real cut, summary reduce) can be exercised on a one-GPU box; the numbers of such a run mean nothing
    one_device = os.environ.get("PISCES_BENCH_ONE_DEVICE") == "1"
    if one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # the collectives also run at world size 1 when launched by torch.distributed.run (exercises the RCCL path on a 1-GPU box)
    use_dist = world > 1 or "TORCHELASTIC_RUN_ID" in os.environ
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_device:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=dev)

    cfg = _abi.default_config()
    caller = engine.HipVariantCaller(cfg, device=local_rank)

    # ---- the job: ONE interval set of world x args.loci loci (amplicon intervals of 150 loci), partitioned by interval across the
    # ranks (contiguous shards balanced by length x depth, cuts on the 1000-locus block grid).  Weak scaling: the set grows
    # with the number of GPUs, every rank owns ~args.loci loci.  At world size 1 the single shard is BASELINE config 2 itself. ----
    total_loci = args.loci * world
    origin = synth.READ_LEN + 1                                      # position of locus 0 (the generator's flank comes first)
    n_amp = -(-total_loci // synth.READ_LEN)
    intervals = [(origin + a * synth.READ_LEN, origin + min((a + 1) * synth.READ_LEN, total_loci) - 1) for a in range(n_amp)]
    parts = shard.partition_intervals(intervals, world, block_size=cfg.block_size, weights=[args.depth] * len(intervals))
    own_lo, own_hi, own_intervals = parts[rank]
    assert own_hi >= own_lo, "more ranks than blocks"
    own_lo, own_hi = max(own_lo, intervals[0][0]), min(own_hi, intervals[-1][1])
    first_locus, my_loci = own_lo - origin, own_hi - own_lo + 1

    # ---- inputs, resident in HBM: RING_BATCHES distinct pileups of the whole set, this rank making only its own shard of each ----
    tile_loci = caller.balanced_tile_loci(my_loci)   # every CU gets the same number of tiles (100 000 loci: 56 -> 1786 tiles, 7 per CU)
    ring = [synth.make_pileup(my_loci, args.depth, seed=BASE_SEED + b, device=dev, first_locus=first_locus, total_loci=total_loci, tile=tile_loci)
            for b in range(RING_BATCHES)]
    for p in ring:
        p.base = p.base if p is ring[0] else None   # keep the read matrices of batch 0 only (CPU baseline / end-to-end sample)
        p.qual = p.qual if p is ring[0] else None
    torch.cuda.empty_cache()
    n_tiles = ring[0].n_tiles
    cap = n_tiles * _abi.SLOTS_PER_TILE   # slot layout (include/pisces_hip.h PiscesTileResult): no allocation atomics
    # Stream discipline: the launches go to the handle's own (non-blocking) stream, which does not order against torch's default (null)
    # stream.  Every output buffer of this run — the pipelined figure's per-lane ones too — is therefore allocated and zero-filled HERE, on
    # the handle's stream (engine.torch_stream(): an ExternalStream over pisces_hip_get_stream), and the device is synchronised once before
    # the first launch; no torch allocation sits between two library calls below.
    stream = caller.torch_stream()
    with torch.cuda.stream(stream):
        records = torch.zeros(cap * 64, dtype=torch.uint8, device=dev)
        tile_results = torch.zeros(n_tiles * _abi.TILE_RESULT_DTYPE.itemsize, dtype=torch.uint8, device=dev)
        n_extra = 0 if args.no_pipelined else PIPELINE_STREAMS - 1
        p_records = [records] + [torch.zeros_like(records) for _ in range(n_extra)]
        p_results = [tile_results] + [torch.zeros_like(tile_results) for _ in range(n_extra)]
    torch.cuda.synchronize(dev)

    # (the arguments of a step are made once: nine data_ptr() calls a step are ~4 us of Python in front of the first launch)
    step_args = [(p.tuples.data_ptr(), p.tiles.data_ptr(), p.n_tiles, p.ref.data_ptr(), p.ref_start, p.ref_len,
                  records.data_ptr(), cap, tile_results.data_ptr(), stream) for p in ring]

    def step(i):
        caller.call_tiles(*step_args[i % RING_BATCHES])

    def barrier():
        if use_dist:
            dist.barrier()

    # The K timed steps as ONE HIP graph (pisces_hip_call_tiles_graph_build: the K launches captured in order, replayed with one
    # submission): the launches then follow each other at the device's pace — a launch call per step through Python leaves ~1.3 us between
    # 38 us kernels.  Measured (round 4, K = 20): the device-side span per launch is the same 40.3 us either way (kernel 38.8 + the
    # dispatch gap), and hipGraphLaunch's own host cost makes the wall clock 1 us per step WORSE than the plain loop — what K = 20 shows
    # in ms_per_step is ~75 us of fixed host <-> device latency around the K launches (event records, the wake-up of the final wait, the
    # totals' read-back), not launch gaps.  So the plain loop stays the default; BENCH_GRAPH_LAUNCHES=1 replays the graph instead.
    def tile_batch(i):
        p = ring[i % RING_BATCHES]
        return (p.tuples.data_ptr(), p.tiles.data_ptr(), p.n_tiles, p.ref.data_ptr(), p.ref_start, p.ref_len, records.data_ptr(), cap, tile_results.data_ptr())
    use_graph = os.environ.get("BENCH_GRAPH_LAUNCHES") == "1"
    graph_id = caller.call_tiles_graph_build([tile_batch(args.warmup + i) for i in range(args.steps)]) if use_graph else None

    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize(dev)
    caller.device_totals(reset=True)

    # ---- timed region: exactly K steps, bracketed by barrier + synchronize ----
    barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    if use_graph:
        caller.call_tiles_graph_launch(graph_id, stream)
    else:
        for i in range(args.steps):
            step(args.warmup + i)
    totals = caller.device_totals()        # (waits for the K launches: their totals arrive in pinned memory behind them, one stream wait)
    if use_dist:
        summary = torch.tensor([totals["records"], totals["candidate_loci"], totals["called"], totals["tiles"]], dtype=torch.int64).to(dev)
        shard.reduce_summary(summary)   # the per-chromosome summary reduce: one all-reduce(sum) of int64[4] (RCCL over xGMI)
        torch.cuda.synchronize(dev)
    barrier()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    # ---- the same K launches once more between two HIP events on the launch stream (torch.cuda.Event would see torch's only): the
    # device-side span of the K launches back to back.  Outside the timed region: the two event records are ~8 us of its ~830. ----
    caller.mark(0, stream)
    for i in range(args.steps):
        step(args.warmup + i)
    caller.mark(1, stream)
    torch.cuda.synchronize(dev)
    span_ms = caller.marked_ms() / args.steps

    # ---- the same K launches once more, every one with the HIP events of its own dispatch (hipExtLaunchKernel start / stop): the
    # kernel's duration as rocprofv3 reports it.  Outside the timed region: the events put 5-10 us between launches. ----
    caller.set_timing(TIME_EVERY)
    for i in range(args.steps):
        step(args.warmup + i)
    torch.cuda.synchronize(dev)
    kernel_ms_total, launches = caller.kernel_time()
    caller.set_timing(False)
    caller.device_totals(reset=True)

    # ---- sanity on the last step's output (outside the timed region) ----
    tr = tile_results.cpu().numpy().view(_abi.TILE_RESULT_DTYPE)
    assert int(tr["n_records"].sum()) * args.steps == totals["records"]
    assert int(tr["n_candidate_loci"].sum()) == my_loci, "every covered locus must be a candidate locus in gVCF mode"

    if not use_dist:   # (one rank: the summary is this rank's totals; the tensor is made outside the timed region)
        summary = torch.tensor([totals["records"], totals["candidate_loci"], totals["called"], totals["tiles"]], dtype=torch.int64)
    total_records, sum_loci = int(summary[0].item()), int(summary[1].item())
    assert sum_loci == total_loci * args.steps, (sum_loci, total_loci, args.steps)   # the shards cover the interval set exactly once
    value = sum_loci / elapsed

    # ---- one cut of the partition checked on the device (outside the timed region): the two sides of this rank's cut, each called by
    # its own handle from the reads shard.reads_for_shard hands it (halo reads on both sides), concatenate to the unsharded window ----
    shard_check = None
    if not args.no_shard_check:
        bs = cfg.block_size
        if world > 1:
            cut = own_lo if rank > 0 else own_hi + 1
        else:
            cut = shard.partition_intervals(intervals, 2, block_size=bs)[1][0]   # where a second rank would start
        w_lo, w_hi = max(cut - bs, intervals[0][0]), min(cut + bs - 1, intervals[-1][1])
        g_lo = max(w_lo - origin - synth.READ_LEN, 0)
        g_hi = min(w_hi - origin + synth.READ_LEN, total_loci - 1)
        wp = synth.make_pileup(g_hi - g_lo + 1, args.depth, seed=BASE_SEED, device=dev, first_locus=g_lo, total_loci=total_loci)
        rb = synth.reads_of(wp)
        arrays = (rb.position, rb.flags, rb.cigar_offset, rb.cigar_op, rb.cigar_len, rb.seq_offset, rb.bases, rb.quals)
        ref_slice = wp.ref.cpu().numpy()
        ref_full = np.full(wp.ref_start - 1 + len(ref_slice), ord("N"), dtype=np.uint8)
        ref_full[wp.ref_start - 1:] = ref_slice
        n_rec, n_counted = shard.verify_cut(lambda: engine.HipVariantCaller(cfg, device=local_rank), ref_full, arrays, w_lo, cut, w_hi,
                                            halo=synth.READ_LEN + 10)
        ok = torch.tensor([1], dtype=torch.int64, device=dev)
        if use_dist:
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        shard_check = {"cut": int(cut), "window": [int(w_lo), int(w_hi)], "records": int(n_rec), "reads": int(n_counted),
                       "ranks_ok": int(ok.item()) == 1, "what": "two handles fed by shard.reads_for_shard == one handle, byte for byte"}
        del wp

    # ---- extra figure, outside the timed region: the same K steps handed to pisces_hip_call_tiles_batched in one call (own output
    # buffers per lane).  Batches are independent, so the call phase at the end of one launch overlaps the streaming phase of the next, which a
    # single in-order stream forbids; this is how a host with several blocks in flight drives the library.
    pipelined_elapsed = None
    if not args.no_pipelined:
        # pisces_hip_call_tiles_batched spreads the launches over the handle's own HIP streams ("lanes"); every batch in flight needs its
        # own output buffers
        # (allocated with the other buffers in front of the first launch: see "Stream discipline" above)
        def pbatches(first, n):
            out = []
            for i in range(first, first + n):
                p, k = ring[i % RING_BATCHES], i % PIPELINE_STREAMS
                out.append((p.tuples.data_ptr(), p.tiles.data_ptr(), p.n_tiles, p.ref.data_ptr(), p.ref_start, p.ref_len, p_records[k].data_ptr(), cap,
                            p_results[k].data_ptr()))
            return out

        caller.call_tiles_batched(pbatches(0, 2 * PIPELINE_STREAMS))
        caller.synchronize()
        todo = pbatches(0, args.steps)
        barrier()
        tp0 = time.perf_counter()
        caller.call_tiles_batched(todo)     # the same K steps, one call; complete after synchronize()
        caller.synchronize()
        barrier()
        tp = torch.tensor([time.perf_counter() - tp0], dtype=torch.float64, device=dev)
        if use_dist:
            dist.all_reduce(tp, op=dist.ReduceOp.MAX)
        pipelined_elapsed = float(tp.item())
        for k in range(1, PIPELINE_STREAMS):   # every lane's last output equals a serial launch's (same batch -> same records)
            trk = p_results[k].cpu().numpy().view(_abi.TILE_RESULT_DTYPE)
            assert int(trk["n_candidate_loci"].sum()) == my_loci
    # ---- the same summary through the C ABI (pisces_hip_reduce_summary: RCCL bound by the library, what a host without torch
    # calls), outside the timed region, after every torch collective of this run, and not fatal: it runs on a side thread with a time
    # limit, so that a communicator that cannot be set up on some node costs this field and not the bench line ----
    c_abi_reduce, c_abi_hung = None, False
    if world > 1:
        import threading
        box = {}

        def through_the_c_abi():
            try:
                ids = [engine.HipVariantCaller.comm_unique_id() if rank == 0 else None]
                dist.broadcast_object_list(ids, src=0)
                caller.comm_init(ids[0], rank, world)
                red = caller.reduce_summary([totals["records"], totals["candidate_loci"], totals["called"], totals["tiles"]])
                box["r"] = {"ok": red == [int(x) for x in summary.tolist()], "summary": red, "ranks": caller.comm_ranks(),
                            "library": engine.HipVariantCaller.comm_library()}
            except Exception as e:   # noqa: BLE001
                box["r"] = {"ok": False, "error": str(e)[:200]}

        th = threading.Thread(target=through_the_c_abi, daemon=True)
        th.start()
        th.join(90.0)
        c_abi_hung = th.is_alive()
        c_abi_reduce = box.get("r", {"ok": False, "error": "no answer within 90 s"})

    # ---- N > 1: BASELINE's multi-GPU configuration (config 4: 30 M loci x 200x cut 8 ways by interval, STRONG scaling) beside the
    # weak-scaling `value` (which stays config 2 per GPU, so that the line's value means the same thing at every N): every rank takes its
    # shards of the 8-way cut in turn, totals reduced.  Outside the timed region; BENCH_CONFIG4=0 skips it (it takes ~2 minutes / N). ----
    config4_line = None
    if world > 1 and os.environ.get("BENCH_CONFIG4", "1") != "0":
        try:
            config4_line = config4_job(torch, dist, world, rank, local_rank, use_dist)
        except Exception as e:   # noqa: BLE001  (an extra figure: it must not cost the bench line)
            config4_line = {"error": str(e)[:300]}

    if rank == 0:
        # roofline of the dominant (only) kernel: algorithmic bytes per launch / mean kernel duration from HIP
        # events recorded on the launch stream around every TIME_EVERY-th launch of the timed region
        n_obs = float(np.mean([p.n_obs for p in ring]))
        rec_per_launch = totals["records"] / max(args.steps, 1)
        bytes_per_launch = algorithmic_bytes(n_obs, my_loci, rec_per_launch)
        kernel_ms = kernel_ms_total / max(launches, 1)
        # `achieved`: against the kernel's own duration (the events of its dispatch, second pass: what rocprofv3's kernel statistics of this
        # command say); timed_region_ms_per_launch is the timed region's two events / K, gaps between launches included
        achieved = bytes_per_launch / (kernel_ms * 1e-3) / 1e9
        traffic, traffic_run = None, None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):   # written from separate rocprofv3 --pmc passes of this same command
            try:
                tj = json.load(open(tpath))
                from pisces_amd import build as native_build
                if tj.get("loci") == args.loci and tj.get("depth") == args.depth:
                    # the figure is only printed for the kernels it was collected on: the file carries the hash of the library's sources
                    if tj.get("source_hash") == native_build.source_hash():
                        traffic = tj.get("hbm_bytes_per_launch")
                        traffic_run = tj.get("run")
                    else:
                        traffic_run = "stale: profiles/traffic.json was collected on other sources (%s); re-run tools/profile_round.sh" % tj.get("run")
            except Exception:
                traffic = None
        out = {
            "metric": "candidate loci/s at 500x depth",
            "value": value,
            "unit": "candidate loci/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "int32 counts + f64 likelihoods",
            "data": "synthetic",
            "config": {"workload": f"BASELINE config 2: synthetic {args.loci} loci x {args.depth}x amplicon pileup, SNV-only, "
                                   "gVCF, per GPU per step; device-resident packed tuples",
                       "loci_per_gpu_per_step": args.loci, "depth": args.depth, "observations_per_step": int(n_obs),
                       "records_per_step": rec_per_launch, "ring_batches": RING_BATCHES, "tile_loci": tile_loci, "tiles_per_step": n_tiles,
                       "interval_set": {"loci": total_loci, "intervals": len(intervals), "rank0_shard": [int(own_lo), int(own_hi)]},
                       "parallelism": f"interval-shard x{world}"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "pisces::call_tiles_wave_kernel", "kernel_ms": kernel_ms, "launches_timed": launches,
                         "timed_region_ms_per_launch": span_ms, "frac_timed_region": bytes_per_launch / (span_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "algorithmic_bytes_per_launch": bytes_per_launch,
                         "how": "kernel_ms: the K launches of the timed region launched once more, each with the HIP events of its own dispatch "
                                "(hipExtLaunchKernel start / stop: the duration rocprofv3 reports; those events put 5-10 us between launches, so "
                                "the timed region carries none); timed_region_ms_per_launch: the K launches once more with one HIP event in front "
                                "of and one behind them on the launch stream, / K (gaps between launches included)"},
        }
        # context only (SURVEY 8d asks for the measured peak beside the spec one; frac stays against the spec peak):
        # a plain streaming read of 1 GiB with the kernel's own load pattern
        out["roofline"]["peak_measured_read"] = caller.probe_read_bandwidth(1 << 30, 6)
        out["roofline"]["traffic_from"] = traffic_run   # the rocprofv3 --pmc passes of tools/profile_round.sh the figure was collected in (not this run)
        if pipelined_elapsed is not None:
            p_ms = pipelined_elapsed / args.steps * 1e3
            out["pipelined"] = {"streams": PIPELINE_STREAMS, "value": total_loci * args.steps / pipelined_elapsed,
                                "unit": "candidate loci/s", "ms_per_step": p_ms,
                                "hbm_frac": bytes_per_launch / (p_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                "note": "same steps over several HIP streams; not the contract's value, not a per-kernel roofline"}
        if shard_check is not None:
            out["shard_check"] = shard_check
        # ranks that joined the collectives of this run: torch.distributed's process group (the timed region's reduce) and, N > 1, ncclCommCount
        # of the communicator the library itself made through the C ABI (1 at N = 1: no communicator, the value is the 1-GPU path's)
        out["rccl_ranks"] = {"torch_process_group": dist.get_world_size() if use_dist else 1,
                             "c_abi_comm_count": (c_abi_reduce or {}).get("ranks", caller.comm_ranks() if world == 1 else None)}
        if c_abi_reduce is not None:
            out["c_abi_reduce"] = c_abi_reduce
        if config4_line is not None:
            out["config4_strong_scaling"] = config4_line
        if not args.no_end_to_end and world == 1:     # the drop-in boundary's own rates, rank 0 at N=1 only
            out["end_to_end"] = end_to_end(ring[0], cfg, engine)
            try:
                out["end_to_end_full"], out["roofline_streaming"] = end_to_end_full(ring[0], cfg, engine, torch)
            except Exception as e:   # noqa: BLE001  (extra figures: they must not cost the bench line)
                out["end_to_end_full"] = {"error": str(e)[:200]}
            try:
                out["roofline_chain"] = chain_roofline(ring[0], cfg, engine, torch)
            except Exception as e:   # noqa: BLE001  (extra figures: they must not cost the bench line)
                out["roofline_chain"] = {"error": str(e)[:200]}
            for key, sample in (("roofline_config3", config3_sample), ("roofline_config5", config5_sample), ("roofline_config4", config4_sample)):
                try:
                    out[key] = sample(engine, torch)
                except Exception as e:   # noqa: BLE001  (extra figures: they must not cost the bench line)
                    out[key] = {"error": str(e)[:200]}
            if not args.no_large_bam:
                try:
                    out["end_to_end_full"]["from_bam_bytes_large"] = from_large_bam(cfg, engine)
                except Exception as e:   # noqa: BLE001
                    out["end_to_end_full"]["from_bam_bytes_large"] = {"error": str(e)[:200]}
        if not args.no_cpu_baseline and world == 1:   # timed on rank 0 at N=1 only
            out["cpu_baseline"], out["cpu_baseline_threads"] = cpu_baseline(torch, ring[0], cfg)
            # the reference C# itself beside the port, when the box can run it: the probe's outcome is reported either way
            out["cpu_baseline_reference"] = reference_csharp_baseline(ring[0], cfg)
            # (the C# run is a process wall clock over ~6 000 loci: dotnet start-up and the genome load are inside it, so it stays under its
            # own key and never replaces the in-process port as `cpu_baseline`)
        print(json.dumps(out), flush=True)
    if c_abi_hung:   # a side thread sits in a communicator that never came up: nothing more to do in this process
        sys.stdout.flush()
        os._exit(0)
    caller.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    sys.exit(main())
