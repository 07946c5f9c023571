"""Interval sharding across the GPUs of one node.

Loci shard by genomic interval: tiles are already in genomic order, each rank owns a contiguous, balanced range
of tiles and calls only loci it owns; concatenating rank outputs in rank order is genomic order. There is no
data-path collective. The single exchange is one all-reduce(sum) of the int64[4] per-chromosome summary —
RCCL over xGMI on GPUs (backend "nccl"), gloo in the CPU tests — reproducing the reference's totals line
(src/exe/Pisces/Logic/SmallVariantCaller.cs:114-115).
"""
import torch
import torch.distributed as dist


def tile_range(n_tiles, rank, world):
    """[lo, hi) of the tiles owned by `rank`: contiguous, sizes differ by at most one."""
    base, rem = divmod(n_tiles, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def reduce_summary(summary):
    """In-place sum over ranks of an int64 summary tensor (no-op without an initialised process group)."""
    assert summary.dtype == torch.int64
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(summary, op=dist.ReduceOp.SUM)
    return summary


def partition_intervals(intervals, world, block_size=1000, weights=None):
    """contiguous shards of a sorted, disjoint interval set [(start, end)], inclusive, balanced by
    sum(interval length x weight) (weight = expected depth; 1 when None).  Cuts fall on the block grid (multiples of block_size), so a
    1000-locus block is never split between ranks and the block schedule of every shard is the single-GPU one; an interval that
    straddles a cut is clipped on both sides.  Returns, per rank, (owned_lo, owned_hi, [clipped intervals]); ranks past the work get
    (0, -1, []).  Concatenating the ranks' calls in rank order is genomic order."""
    ivs = [(int(a), int(b)) for a, b in intervals]
    assert all(a <= b for a, b in ivs) and all(x[1] < y[0] for x, y in zip(ivs, ivs[1:])), "intervals must be sorted and disjoint"
    w = [1.0] * len(ivs) if weights is None else [float(x) for x in weights]
    # weight per block of the grid
    blocks = {}
    for (a, b), wt in zip(ivs, w):
        k = (a - 1) // block_size
        while k * block_size + 1 <= b:
            lo, hi = max(a, k * block_size + 1), min(b, (k + 1) * block_size)
            blocks[k] = blocks.get(k, 0.0) + (hi - lo + 1) * wt
            k += 1
    keys = sorted(blocks)
    total = sum(blocks.values())
    out, i, acc = [], 0, 0.0
    for r in range(world):
        target = total * (r + 1) / world
        first = i
        while i < len(keys) and (acc + blocks[keys[i]] <= target + 1e-9 or i == first) and (len(keys) - i > world - 1 - r):
            acc += blocks[keys[i]]
            i += 1
        if r == world - 1:
            while i < len(keys):
                acc += blocks[keys[i]]
                i += 1
        if first == i:
            out.append((0, -1, []))
            continue
        lo, hi = keys[first] * block_size + 1, (keys[i - 1] + 1) * block_size
        out.append((lo, hi, [(max(a, lo), min(b, hi)) for a, b in ivs if b >= lo and a <= hi]))
    return out


def reads_for_shard(read_starts, read_ends, owned_lo, owned_hi, halo):
    """Indices of the reads a shard needs: every read overlapping [owned_lo - halo, owned_hi + halo] (halo = longest read span +
    longest variant span; reads near a cut go to both sides, no inter-GPU exchange).  A read is COUNTED (Totals line) by the shard that
    owns its start."""
    import numpy as np
    s, e = np.asarray(read_starts), np.asarray(read_ends)
    need = (e >= owned_lo - halo) & (s <= owned_hi + halo)
    owner = (s >= owned_lo) & (s <= owned_hi)
    return np.nonzero(need)[0], owner[need]


def read_batch_subset(batch_arrays, idx):
    """ReadBatch of the reads `idx` of arrays made by reads_as_arrays (fixed-length single-M reads are not assumed)."""
    from . import _abi
    import numpy as np
    pos, flags, cig_off, cig_op, cig_len, seq_off, bases, quals = batch_arrays
    idx = np.asarray(idx, dtype=np.int64)
    n = len(idx)
    ncig = (cig_off[idx + 1] - cig_off[idx]).astype(np.int64)
    nseq = (seq_off[idx + 1] - seq_off[idx]).astype(np.int64)
    new_cig_off = np.concatenate([[0], np.cumsum(ncig)]).astype(np.int32)
    new_seq_off = np.concatenate([[0], np.cumsum(nseq)]).astype(np.int32)
    cig_idx = np.concatenate([np.arange(cig_off[i], cig_off[i + 1]) for i in idx]) if n else np.zeros(0, np.int64)
    seq_idx = np.concatenate([np.arange(seq_off[i], seq_off[i + 1]) for i in idx]) if n else np.zeros(0, np.int64)
    return _abi.ReadBatch.from_arrays(position=pos[idx], flags=flags[idx], cigar_offset=new_cig_off, cigar_op=cig_op[cig_idx],
                                      cigar_len=cig_len[cig_idx], seq_offset=new_seq_off, bases=bases[seq_idx], quals=quals[seq_idx])


def verify_cut(make_caller, ref_full, batch_arrays, window_lo, cut, window_hi, halo):
    """One cut of an interval partition on the device, against the unsharded run of the same window: the loci
    [window_lo, cut - 1] and [cut, window_hi] called by two handles, each fed the reads reads_for_shard gives it (halo reads go to both)
    and reporting only the loci it owns, must concatenate to what one handle calls for [window_lo, window_hi] from all the reads.
    make_caller() -> a fresh HipVariantCaller; batch_arrays: the position-sorted reads as numpy arrays (ReadBatch.from_arrays order).
    Returns (records, reads counted by their owners) and raises AssertionError on any difference."""
    import numpy as np
    pos, flags, cig_off, cig_op, cig_len, seq_off, bases, quals = batch_arrays
    starts = pos.astype(np.int64)
    span = np.zeros(len(pos), dtype=np.int64)
    refspan = np.isin(cig_op, np.frombuffer(b"MDN=X", dtype=np.uint8))
    np.add.at(span, np.repeat(np.arange(len(pos)), np.diff(cig_off)), np.where(refspan, cig_len, 0).astype(np.int64))
    ends = starts + np.maximum(span, 1) - 1

    def run(lo, hi, idx):
        with make_caller() as c:
            c.SetReference(ref_full)
            c.SetIntervals([(int(lo), int(hi))])
            if len(idx):
                c.AddAlleleCounts(read_batch_subset(batch_arrays, idx))
            return c.Call(None, capacity=1 << 17)

    whole = run(window_lo, window_hi, np.arange(len(pos)))
    parts, counted = [], 0
    for lo, hi in ((window_lo, cut - 1), (cut, window_hi)):
        idx, owner = reads_for_shard(starts, ends, lo, hi, halo)
        counted += int(owner.sum())
        parts.append(run(lo, hi, idx))
    got = np.concatenate(parts)
    assert got.tobytes() == whole.tobytes(), "sharded halves differ from the unsharded window"
    in_window = int(((starts >= window_lo) & (starts <= window_hi)).sum())
    assert counted == in_window, (counted, in_window)
    return len(whole), counted
