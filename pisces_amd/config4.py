"""BASELINE config 4 as SURVEY.md section 8d states it: 30 M loci x 200x over 200 000 intervals of 150 bp on 24 synthetic contigs,
SNVs + small insertions / deletions at a tenth of config 3's density, cut 8 ways by interval with the shards' outputs concatenated in
order (the reference runs one job per chromosome and concatenates: src/lib/Pisces.Processing/Logic/BaseGenomeProcessor.cs:40-90,
src/exe/Pisces/Logic/Processing/GenomeProcessor.cs:156-186; interval sets: src/lib/Pisces.Domain/Models/IntervalSet.cs:38-74).

A contig is made as a contiguous amplicon pileup (pisces_amd.synth) and then spread out: interval i sits at 151 + 300 i .. + 149, the
150 positions behind it are uncovered.  Contig sizes follow the human chromosomes, so that an 8-way cut by weight falls INSIDE contigs.
The whole set is 12 GB of reads: one MI355X holds any contig (<= 1 GB) with room to spare, and the eight shards can run in turn on one
GPU — which is how the tests and `bench.py --config 4` run it when there is one GPU.
"""
import numpy as np

from . import _abi, shard, synth

INTERVAL, PITCH, FIRST = synth.READ_LEN, 2 * synth.READ_LEN, synth.READ_LEN + 1
# relative sizes of chr1 .. chr22, X, Y (Mb, GRCh38)
CONTIG_MB = [248, 242, 198, 190, 182, 171, 159, 145, 138, 134, 135, 133, 114, 107, 102, 90, 83, 80, 59, 64, 47, 51, 156, 57]


def contig_intervals(total_intervals=200_000):
    """Intervals per contig (24 numbers adding up to total_intervals)."""
    w = np.array(CONTIG_MB, dtype=np.float64)
    n = np.floor(total_intervals * w / w.sum()).astype(np.int64)
    n[0] += total_intervals - int(n.sum())
    return [int(x) for x in n]


def interval_bounds(n_intervals):
    starts = FIRST + PITCH * np.arange(n_intervals, dtype=np.int64)
    return starts.astype(np.int32), (starts + INTERVAL - 1).astype(np.int32)


def contig_length(n_intervals):
    return FIRST + PITCH * n_intervals + INTERVAL


def make_contig(c, n_intervals, depth=200, seed=20260930, device="cuda"):
    """Reference, intervals and position-sorted reads of contig `c`.  Returns a dict: ref (uint8 ASCII, position p = ref[p - 1]),
    starts / ends (int32), batch (ReadBatch), arrays (the batch's numpy arrays, shard.read_batch_subset order), read_end (int64),
    planted [(kind, position, ref, alt)]."""
    n_loci = n_intervals * INTERVAL
    p = synth.make_pileup(n_loci, depth, seed=seed + 1000 * c, device=device, first_locus=0, total_loci=n_loci, with_tuples=False)
    batch, planted = synth.mixed_reads(p, seed + c, sparse=10, kinds="DI")
    origin = p.flank + 1

    def spread(pos):   # compact coordinate -> the contig's
        rel = np.asarray(pos, dtype=np.int64) - origin
        return (FIRST + rel // INTERVAL * PITCH + rel % INTERVAL).astype(np.int32)

    compact = p.ref.cpu().numpy()
    rng = np.random.default_rng(seed * 7 + c)
    ref = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, contig_length(n_intervals))]
    body = compact[origin - p.ref_start: origin - p.ref_start + n_loci].reshape(n_intervals, INTERVAL)
    idx = (FIRST - 1 + PITCH * np.arange(n_intervals))[:, None] + np.arange(INTERVAL)[None, :]
    ref[idx] = body
    position = spread(batch.position)
    arrays = (position, batch.flags, batch.cigar_offset, batch.cigar_op, batch.cigar_len, batch.seq_offset, batch.bases, batch.quals)
    new_batch = _abi.ReadBatch.from_arrays(*arrays)
    span = np.zeros(len(position), dtype=np.int64)
    refspan = np.isin(batch.cigar_op, np.frombuffer(b"MDN=X", dtype=np.uint8))
    np.add.at(span, np.repeat(np.arange(len(position)), np.diff(batch.cigar_offset)), np.where(refspan, batch.cigar_len, 0).astype(np.int64))
    starts, ends = interval_bounds(n_intervals)
    return {"contig": c, "ref": ref, "starts": starts, "ends": ends, "batch": new_batch, "arrays": arrays,
            "read_end": position.astype(np.int64) + np.maximum(span, 1) - 1,
            "planted": [(k, int(spread([pos])[0]), r, a) for k, pos, r, a in planted], "n_loci": n_loci, "depth": depth}


def partition(sizes, world, block_size=1000, depth=200):
    """The 8-way (world-way) cut of the whole interval set: the contigs laid end to end on one axis (each starting on a block edge),
    shard.partition_intervals over all the intervals, and every shard mapped back to pieces (contig, owned_lo, owned_hi) in contig
    coordinates.  Cuts fall on the 1000-locus block grid of the contig they fall in."""
    offs, acc = [], 0
    for n in sizes:
        offs.append(acc)
        acc += -(-contig_length(n) // block_size) * block_size
    ivs = []
    for c, n in enumerate(sizes):
        s, e = interval_bounds(n)
        ivs += list(zip((s.astype(np.int64) + offs[c]).tolist(), (e.astype(np.int64) + offs[c]).tolist()))
    parts = shard.partition_intervals(ivs, world, block_size=block_size, weights=[depth] * len(ivs))
    out = []
    for lo, hi, _ in parts:
        pieces = []
        for c, n in enumerate(sizes):
            c_lo, c_hi = offs[c] + 1, offs[c] + contig_length(n)
            a, b = max(lo, c_lo), min(hi, c_hi)
            if a <= b:
                pieces.append((c, int(a - offs[c]), int(b - offs[c])))
        out.append(pieces)
    return out


def read_batch_range(arrays, i0, i1):
    """ReadBatch of the reads [i0, i1) of position-sorted arrays (a shard's reads are a contiguous range of them)."""
    pos, flags, cig_off, cig_op, cig_len, seq_off, bases, quals = arrays
    c0, c1, s0, s1 = int(cig_off[i0]), int(cig_off[i1]), int(seq_off[i0]), int(seq_off[i1])
    return _abi.ReadBatch.from_arrays(position=pos[i0:i1], flags=flags[i0:i1], cigar_offset=cig_off[i0:i1 + 1] - c0, cigar_op=cig_op[c0:c1],
                                      cigar_len=cig_len[c0:c1], seq_offset=seq_off[i0:i1 + 1] - s0, bases=bases[s0:s1], quals=quals[s0:s1])


def piece_plan(job, lo=None, hi=None, halo=synth.READ_LEN + 16):
    """What a (contig, owned range) piece is made of: the clipped intervals, the reads shard.reads_for_shard gives it ([i0, i1) of the
    position-sorted reads) and how many of them it owns (a read is counted by the piece that owns its start)."""
    starts, ends = job["starts"], job["ends"]
    pos = job["arrays"][0].astype(np.int64)
    if lo is None:
        lo, hi = 1, len(job["ref"])
    keep = (ends >= lo) & (starts <= hi)
    ivs = np.stack([np.maximum(starts[keep], lo), np.minimum(ends[keep], hi)], axis=1).astype(np.int32)   # (n, 2): first, last position
    idx, owner = shard.reads_for_shard(pos, job["read_end"], lo, hi, halo)
    i0 = i1 = 0
    if len(idx):
        assert idx[-1] - idx[0] + 1 == len(idx)      # position-sorted reads: a contiguous range
        i0, i1 = int(idx[0]), int(idx[-1]) + 1
    return {"lo": lo, "hi": hi, "intervals": ivs, "i0": i0, "i1": i1, "pos": pos, "owned": int(owner.sum())}


def plan_loci(plan):
    """Positions of the piece's clipped intervals: with zero-coverage reference rows on (config 4's setting) every one of them has a row."""
    iv = np.asarray(plan["intervals"], dtype=np.int64).reshape(-1, 2)
    return int((iv[:, 1] - iv[:, 0] + 1).sum())


def device_arrays(job, device="cuda:0"):
    """The contig's read arrays in device memory (torch tensors, the order of job["arrays"])."""
    import torch
    return [torch.from_numpy(np.ascontiguousarray(a if a.dtype != np.uint32 else a.view(np.int32))).to(device) for a in job["arrays"]]


def device_chunks(engine, job, dev_arrays, plan, chunk_reads=400_000):
    """The stretches of reads run_piece adds, as engine.DeviceReadBatch views of the contig's arrays in device memory (offsets rebased to
    the stretch: two small tensors of their own)."""
    pos, flags, cig_off, cig_op, cig_len, seq_off, bases, quals = dev_arrays
    h_cig_off, h_seq_off = job["arrays"][2], job["arrays"][5]
    out = []
    for a in range(plan["i0"], plan["i1"], chunk_reads):
        b = min(a + chunk_reads, plan["i1"])
        c0, c1, s0, s1 = int(h_cig_off[a]), int(h_cig_off[b]), int(h_seq_off[a]), int(h_seq_off[b])
        out.append(engine.DeviceReadBatch(pos[a:b], flags[a:b], cig_off[a:b + 1] - c0, cig_op[c0:c1], cig_len[c0:c1], seq_off[a:b + 1] - s0,
                                          bases[s0:s1], quals[s0:s1], n_ops=c1 - c0, n_bases=s1 - s0))
    if out:
        out[0].synchronize()
    return out


def run_piece(engine, cfg, job, lo=None, hi=None, halo=synth.READ_LEN + 16, device=0, chunk_reads=400_000, with_alleles=True, keep_records=True,
              plan=None, chunks=None, count_loci=True, caller=None):
    """One (contig, owned range) job on one handle: set_reference, set_intervals (clipped to the range), the reads
    shard.reads_for_shard gives it, one final flush.  lo / hi None: the whole contig.  Returns (records, alleles, stats, reads counted:
    a read is counted by the piece that owns its start).  keep_records=False (bench.py): the rows are looked at where they lie
    (pisces_hip_flush_view) and only counted — `records` is then {"n": rows, "loci": distinct positions}.  chunks (device_chunks of
    `plan`): the reads are handed over in device memory (pisces_hip_add_device_reads) instead of from the host arrays.  caller (an
    engine.HipVariantCaller that has the contig's reference already): the piece runs on that handle — one handle per contig, as the
    reference has one SmallVariantCaller per chromosome job (Factory.cs:253-269), its pieces in turn: intervals and owned range are set per
    piece, a final flush leaves the handle empty for the next; the stats returned are the piece's own (differences)."""
    if plan is None:
        plan = piece_plan(job, lo, hi, halo)
    lo, hi, pos, i0, i1 = plan["lo"], plan["hi"], plan["pos"], plan["i0"], plan["i1"]
    recs, alleles = [], []
    n_rows = n_loci = 0
    last_position = [0]

    def count(view):   # rows arrive in position order, flush after flush
        nonlocal n_rows, n_loci
        if len(view) and not count_loci:   # (the harness' own pass over the rows stays out of a timed region: plan_loci() says how many loci the piece reports)
            n_rows += len(view)
        elif len(view):
            p = view["position"]
            n_rows += len(view)
            n_loci += int((np.diff(p) != 0).sum()) + (1 if int(p[0]) != last_position[0] else 0)
            last_position[0] = int(p[-1])
        return view[:0]

    take = (lambda view: view.copy()) if keep_records else count
    import contextlib
    own = caller is None
    with (engine.HipVariantCaller(cfg, device=device) if own else contextlib.nullcontext(caller)) as c:
        before = None if own else c.Stats()
        if own:
            c.SetReference(job["ref"])
        else:
            c.HostTime(reset=True)
        c.SetIntervals(plan["intervals"])
        c.SetOwnedRange(lo, hi)
        for k, a in enumerate(range(i0, i1, chunk_reads)):   # the streaming protocol: add a stretch of reads, call what lies behind them
            b = min(a + chunk_reads, i1)
            if chunks is not None:
                c.AddDeviceReads(chunks[k])
            else:
                c.AddAlleleCounts(read_batch_range(job["arrays"], a, b))
            if b < i1:
                r, al = c.CallWithAlleles(int(pos[b]) - 1, capacity=1 << 20) if with_alleles else (take(c.CallView(int(pos[b]) - 1)), [])
                recs.append(r)
                alleles += al
        r, al = c.CallWithAlleles(None, capacity=1 << 20) if with_alleles else (take(c.CallView(None)), [])   # (rows read in place; kept by copying them once)
        recs.append(r)
        alleles += al
        stats = c.Stats()
        if before is not None:
            stats = {k: (v - before[k] if isinstance(v, (int, np.integer)) and k in before else v) for k, v in stats.items()}
        stats["host_time"] = c.HostTime()
    if not keep_records:
        return {"n": n_rows, "loci": n_loci}, alleles, stats, plan["owned"]
    return np.concatenate(recs), alleles, stats, plan["owned"]
