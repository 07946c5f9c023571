"""BASELINE config 4 as SURVEY.md section 8d states it, on ONE GPU: 30 M loci x 200x over 200 000 intervals of 150 bp on 24 contigs,
SNVs + small insertions / deletions at a tenth of config 3's density, the interval set cut 8 ways (shard.partition_intervals over the
contigs laid end to end) and the eight shards run in turn, every (contig, range) piece on a handle of its own with halo reads
(shard.reads_for_shard), outputs concatenated in order.  Three contigs that a cut falls in are ALSO run unsharded, and the sharded
output must equal that byte for byte (records and allele strings); every contig is held to size-independent properties; the
per-piece totals add up to the unsharded ones (the vector pisces_hip_reduce_summary adds over the GPUs of a node).
Reference: one job per chromosome, outputs concatenated (BaseGenomeProcessor.cs:40-90, GenomeProcessor.cs:156-186)."""
import numpy as np
import pytest

from pisces_amd import _abi

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch


def _oracle_windows(config4, cfg, job, got, got_alleles, cuts, chunk_reads=400_000, n_random=4):
    """1000-locus blocks of a contig's (sharded, concatenated) output against the oracle, records and allele strings: the block every
    shard cut falls in front of, the block a stretch of reads ends in (where run_piece's flush cuts the run), the contig's last block
    and a few seeded anywhere.  The oracle runs the reads of the block and its two neighbours (interval set and the whole contig's
    flush schedule as far as it falls there); the middle block is compared.  Returns the number of blocks checked."""
    from tests import orc
    from tests.test_gpu_parity import assert_records_match
    pos = job["arrays"][0].astype(np.int64)
    ups_all = [int(pos[b]) - 1 for b in range(chunk_reads, len(pos), chunk_reads)]
    last_position = int(job["ends"][-1])
    last_block = (last_position - 1) // 1000
    blocks = {last_block, 1} | {(lo - 1) // 1000 for lo in cuts} | {(lo - 2) // 1000 for lo in cuts} | {(u - 1) // 1000 for u in ups_all[:2]}
    rng = np.random.default_rng(100 + job["contig"])
    while len(blocks) < len(cuts) * 2 + 2 + min(len(ups_all), 2) + n_random:
        blocks.add(int(rng.integers(1, last_block)))
    blocks = sorted(b for b in blocks if 1 <= b <= last_block)
    for k in blocks:
        lo, hi = (k - 1) * 1000 + 1, min((k + 2) * 1000, len(job["ref"]))
        i0, i1 = int(np.searchsorted(pos, lo - 2 * config4.INTERVAL)), int(np.searchsorted(pos, hi + 1))   # (no read is two intervals long)
        batch = config4.read_batch_range(job["arrays"], i0, i1)
        keep = (job["ends"] >= lo) & (job["starts"] <= hi)
        ivs = list(zip(np.maximum(job["starts"][keep], lo).tolist(), np.minimum(job["ends"][keep], hi).tolist()))
        exp, exp_alleles, _ = orc.run_reads_schedule(batch, job["ref"], lo, hi - lo + 1, cfg, [u for u in ups_all if lo <= u <= hi], intervals=ivs)
        sel = (exp["position"] > k * 1000) & (exp["position"] <= (k + 1) * 1000)
        g0, g1 = np.searchsorted(got["position"], [k * 1000 + 1, (k + 1) * 1000 + 1])
        assert g1 - g0 == int(sel.sum()) > 0, (job["contig"], k, int(g0), int(g1), int(sel.sum()))
        assert_records_match(got[g0:g1], exp[sel])
        assert got_alleles[g0:g1] == [x for x, s_ in zip(exp_alleles, sel) if s_], (job["contig"], k)
    return len(blocks)


def test_config4_as_stated_eight_shards_in_turn_on_one_gpu(torch_cuda):
    from pisces_amd import config4, engine
    depth, world = 200, 8
    sizes = config4.contig_intervals(200_000)
    assert sum(sizes) == 200_000 and len(sizes) == 24 and sum(sizes) * config4.INTERVAL == 30_000_000
    cfg = _abi.default_config(emit_zero_coverage_refs=1)
    shards = config4.partition(sizes, world, block_size=cfg.block_size, depth=depth)
    assert len(shards) == world and all(shards)
    pieces_of = {}
    for r, pieces in enumerate(shards):
        for c, lo, hi in pieces:
            pieces_of.setdefault(c, []).append((r, lo, hi))
    cut_contigs = [c for c, pc in pieces_of.items() if len(pc) > 1]
    assert len(cut_contigs) >= 3, cut_contigs             # an 8-way cut of 24 unequal contigs falls inside contigs
    verify = set(cut_contigs[:2] + cut_contigs[-1:])
    shard_loci = np.zeros(world, dtype=np.int64)
    totals = np.zeros(4, dtype=np.int64)
    n_reads_total = n_loci_total = checked_blocks = 0
    for c, n_iv in enumerate(sizes):
        job = config4.make_contig(c, n_iv, depth=depth)
        n_reads_total += job["batch"].n_reads
        got, got_alleles, counted, called = [], [], 0, 0
        for r, lo, hi in pieces_of[c]:
            recs, alleles, stats, owned = config4.run_piece(engine, cfg, job, lo, hi)
            got.append(recs)
            got_alleles += alleles
            counted += owned
            called += stats["TotalNumCalled"]
            shard_loci[r] += len(np.unique(recs["position"]))
            totals += np.array([stats["TotalNumCalled"], stats["TotalNumCollapsed"], owned, stats["reads_skipped"]])
        got = np.concatenate(got)
        # ---- properties, every contig
        assert counted == job["batch"].n_reads                                   # every read counted by exactly one piece
        assert (np.diff(got["position"]) >= 0).all()
        covered = np.unique(got["position"])
        expect = (job["starts"].astype(np.int64)[:, None] + np.arange(config4.INTERVAL)[None, :]).reshape(-1)
        assert np.array_equal(covered, expect)                                   # every locus of every interval is a candidate locus, nothing outside
        n_loci_total += len(covered)
        cats = (got["info"] >> 4) & 7
        point = (cats == _abi.CAT_SNV) | (cats == _abi.CAT_REFERENCE)
        near_indel = np.zeros(len(got), dtype=bool)
        for kind, pos, ref, alt in job["planted"]:
            near_indel |= (got["position"] >= pos) & (got["position"] <= pos + max(len(ref), len(alt)) + 1)
        assert ((got["total_coverage"] + got["num_no_calls"])[point & ~near_indel] == depth).all()
        for kind, cat in (("D", _abi.CAT_DELETION), ("I", _abi.CAT_INSERTION)):
            want = np.array(sorted(pos for k, pos, _, _ in job["planted"] if k == kind))
            have = np.unique(got["position"][cats == cat])
            assert len(want) == 0 or np.isin(want, have).mean() >= 0.95, (c, kind, len(want), int(np.isin(want, have).sum()))
        # ---- the unsharded contig, byte for byte
        if c in verify:
            whole, whole_alleles, wstats, _ = config4.run_piece(engine, cfg, job)
            assert got.tobytes() == whole.tobytes() and got_alleles == whole_alleles, c
            assert called == wstats["TotalNumCalled"] and wstats["reads"] == job["batch"].n_reads
            checked_blocks += _oracle_windows(config4, cfg, job, got, got_alleles, [lo for _, lo, _ in pieces_of[c][1:]])
        del job, got
    assert n_loci_total == 30_000_000 and n_reads_total == 200_000 * depth
    assert checked_blocks >= 16                                                   # blocks held to the oracle, record for record
    assert totals[2] == n_reads_total and totals[3] == 0
    assert shard_loci.sum() == n_loci_total and shard_loci.min() > 0.8 * shard_loci.mean()   # the cut is balanced


def test_run_piece_counts_rows_in_place_as_it_would_keep_them(torch_cuda):
    """config4.run_piece(keep_records=False) — what bench.py --config 4 times: the rows are looked at where they lie (pisces_hip_flush_view)
    and only counted — reports the rows and distinct positions of the records it would have returned."""
    from pisces_amd import config4, engine
    cfg = _abi.default_config(emit_zero_coverage_refs=1)
    job = config4.make_contig(5, 400, depth=60)
    recs, _, stats, owned = config4.run_piece(engine, cfg, job, with_alleles=False, chunk_reads=7000)
    counted, _, stats2, owned2 = config4.run_piece(engine, cfg, job, with_alleles=False, keep_records=False, chunk_reads=7000)
    assert counted == {"n": len(recs), "loci": len(np.unique(recs["position"]))} and len(recs) >= 400 * config4.INTERVAL
    assert owned == owned2 == job["batch"].n_reads
    assert {k: v for k, v in stats.items() if k != "host_time"} == {k: v for k, v in stats2.items() if k != "host_time"}


def test_the_pieces_of_a_contig_on_one_handle_give_what_handles_of_their_own_give(torch_cuda):
    """One handle per contig, its (contig, range) pieces run on it in turn (config4.run_piece(caller=...): intervals and owned range set per
    piece, every piece ended by a final flush that leaves the handle empty) — what bench.py --config 4 times — against a handle per piece:
    the same records, allele strings and per-piece stats; and the contig's handle takes the whole contig afterwards as a fresh one does."""
    from pisces_amd import config4, engine
    cfg = _abi.default_config(emit_zero_coverage_refs=1)
    job = config4.make_contig(7, 500, depth=50)
    cuts = [(None, None), (1, 40_000), (40_001, 93_000), (93_001, len(job["ref"]))]
    fresh = [config4.run_piece(engine, cfg, job, lo, hi, chunk_reads=6000) for lo, hi in cuts]
    with engine.HipVariantCaller(cfg) as contig_caller:
        contig_caller.SetReference(job["ref"])
        for (lo, hi), want in list(zip(cuts, fresh))[1:] + [(cuts[0], fresh[0])]:
            recs, alleles, stats, owned = config4.run_piece(engine, cfg, job, lo, hi, chunk_reads=6000, caller=contig_caller)
            assert recs.tobytes() == want[0].tobytes() and alleles == want[1] and owned == want[3], (lo, hi)
            assert {k: v for k, v in stats.items() if k != "host_time"} == {k: v for k, v in want[2].items() if k != "host_time"}, (lo, hi)
    assert np.concatenate([f[0] for f in fresh[1:]]).tobytes() == fresh[0][0].tobytes()
