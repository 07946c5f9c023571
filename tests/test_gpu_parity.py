"""Parity of the HIP path (through the C ABI) against the oracle on a real MI355X.
Bit-exact for every integer field of the 64-byte record (position, coverage, support, counts,
filters, genotype) and for the q-scores (north_star allows +-1 Phred; none is used), strand-bias score to 1e-9 relative."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from pisces_amd import _abi
from tests import orc
from tests.test_read_store import env

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
DIR = {"F": 0, "R": 1, "S": 2}
ALLELE = {"A": 0, "G": 1, "C": 2, "T": 3, "N": 4, "D": 5}

INT_FIELDS = ["position", "total_coverage", "allele_support", "reference_support", "num_no_calls", "coverage_by_dir",
              "support_by_dir"]


def assert_records_match(got, exp, q_tol=0):
    """Every field exact by default (what is observed: the device restates the C# arithmetic operation by operation).  With q_tol = 1
    (north_star's +-1 Phred) a q-score may move by one on at most 0.1 % of the rows, and on those rows only what a threshold crossing
    explains may differ: the LowVariantQscore / LowGQ filter bits."""
    assert len(got) == len(exp), (len(got), len(exp))
    for f in INT_FIELDS:
        np.testing.assert_array_equal(got[f], exp[f], err_msg=f)
    dq = np.abs(got["variant_qscore"].astype(np.int64) - exp["variant_qscore"])
    dg = np.abs(got["genotype_qscore"].astype(np.int64) - exp["genotype_qscore"])
    assert dq.max(initial=0) <= q_tol and dg.max(initial=0) <= q_tol, (dq.max(), dg.max())
    moved = (dq != 0) | (dg != 0)
    assert int(moved.sum()) <= len(got) // 1000, (int(moved.sum()), len(got))
    np.testing.assert_array_equal(got["info"], exp["info"])
    np.testing.assert_array_equal(got["noise_level"], exp["noise_level"])   # CalledAllele.NoiseLevelApplied
    q_bits = np.uint16((1 << 3) | (1 << 6))   # FilterType.LowVariantQscore, LowGenotypeQuality
    np.testing.assert_array_equal(got["filter_bits"][~moved], exp["filter_bits"][~moved])
    np.testing.assert_array_equal(got["filter_bits"][moved] & ~q_bits, exp["filter_bits"][moved] & ~q_bits)
    np.testing.assert_allclose(got["strand_bias_score"], exp["strand_bias_score"], rtol=1e-9, atol=1e-300)
    return int(moved.sum())


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    return torch


def run_fused(torch, caller, p, compact=False, ref_start=1):
    """One pisces_hip_call_tiles launch on device-resident buffers; returns the called alleles in order.
    compact=False: read the slot layout through the validity masks; compact=True: pisces_hip_compact_records.
    Stream discipline (the round-4 flake: three torch.zeros fills on torch's null stream landed AFTER gather_records_kernel had written the
    count, because the handle's stream is non-blocking): EVERY buffer is allocated and filled on the handle's own stream
    (engine.torch_stream(), an ExternalStream over pisces_hip_get_stream) before the first library call, the launches go to that same
    stream, and the inputs torch made earlier on its default stream are waited for once.  One queue, no cross-stream ordering left."""
    dev = p.tuples.device
    cap = p.n_tiles * _abi.SLOTS_PER_TILE
    ts = caller.torch_stream()
    with torch.cuda.stream(ts):
        recs = torch.zeros(cap * 64, dtype=torch.uint8, device=dev)
        tres = torch.zeros(p.n_tiles * _abi.TILE_RESULT_DTYPE.itemsize, dtype=torch.uint8, device=dev)
        out_d = torch.zeros(cap * 64 if compact else 1, dtype=torch.uint8, device=dev)
        offs = torch.zeros(max(p.n_tiles, 1), dtype=torch.int32, device=dev)
        count = torch.zeros(1, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()   # p.tuples / p.tiles / p.ref were made on torch's default stream
    caller.call_tiles(p.tuples.data_ptr(), p.tiles.data_ptr(), p.n_tiles, p.ref.data_ptr(), ref_start, p.ref_len,
                      recs.data_ptr(), cap, tres.data_ptr(), ts)
    if compact:
        caller.compact_records(recs.data_ptr(), tres.data_ptr(), p.n_tiles, offs.data_ptr(), out_d.data_ptr(), cap,
                               count.data_ptr(), ts)
    caller.synchronize()
    torch.cuda.synchronize()
    tr = tres.cpu().numpy().view(_abi.TILE_RESULT_DTYPE)
    bad = np.nonzero(tr["record_begin"] != np.arange(len(tr)) * _abi.SLOTS_PER_TILE)[0]
    assert len(bad) == 0, (len(bad), len(tr), bad[:8].tolist(), bad[-8:].tolist(), tr["record_begin"][bad[:8]].tolist())
    if compact:
        n = int(count.item())
        assert n == int(tr["n_records"].sum())
        assert (offs.cpu().numpy()[: p.n_tiles] == np.cumsum(np.r_[0, tr["n_records"]])[:-1]).all()
        return out_d.cpu().numpy().view(_abi.CALLED_ALLELE_DTYPE)[:n].copy(), tr
    out = _abi.records_in_order(recs.cpu().numpy().view(_abi.CALLED_ALLELE_DTYPE), tr)
    assert len(out) == int(tr["n_records"].sum())
    return out, tr


def test_compacted_launch_is_deterministic_over_three_hundred_runs(torch_cuda):
    """Regression for the round-4 flake (GPUTEST_r04: `assert 0 == 313` in draw 9 of the device-resident fuzz, an 8-tile input): a
    compacted launch on a tiny input, 300 times over with fresh buffers every time, each run's count, offsets and rows equal to the
    first's.  With buffers filled on torch's null stream and launches on the handle's non-blocking stream, some of these lose the race."""
    from pisces_amd import engine, synth
    torch = torch_cuda
    p = synth.make_pileup(n_loci=8 * 64, depth=40, seed=9, device="cuda")
    assert p.n_tiles == 8
    with engine.HipVariantCaller(_abi.default_config()) as caller:
        want, tr0 = run_fused(torch, caller, p, compact=True)
        assert len(want) > 8 * 64 - 1
        for i in range(300):
            got, tr = run_fused(torch, caller, p, compact=True)
            assert got.tobytes() == want.tobytes() and tr.tobytes() == tr0.tobytes(), i


@pytest.mark.parametrize("n_loci,depth", [(3 * 64, 30), (1786 * 56, 12), (4700 * 64, 6)], ids=["3_tiles", "1786_tiles", "4700_tiles"])
def test_the_three_forms_of_the_ordered_compaction_give_the_same_rows(torch_cuda, monkeypatch, n_loci, depth):
    """pisces_hip_compact_records is ONE launch since round 6 — up to 4 096 tiles a tile's wave adds up the record counts of the tiles
    before it (gather_direct_kernel), beyond them a decoupled look-back over workgroups of sixteen tiles (compact_records_kernel) — where it
    was a scan launch and a gather launch (PISCES_HIP_COMPACT=two, kept for the A / B).  The three forms (the look-back forced on the small
    launches too) must write the same count, the same offsets and the same rows, launch after launch on one handle (the look-back's words
    carry the launch's number: nothing is cleared in between), and equal the rows read through the validity masks."""
    from pisces_amd import engine, synth
    torch = torch_cuda
    p = synth.make_pileup(n_loci=n_loci, depth=depth, seed=21, device="cuda")
    cfg = _abi.default_config()
    rows = {}
    for mode in ("", "two", "lookback"):
        if mode:
            monkeypatch.setenv("PISCES_HIP_COMPACT", mode)
        else:
            monkeypatch.delenv("PISCES_HIP_COMPACT", raising=False)
        with engine.HipVariantCaller(cfg) as caller:
            plain, tr0 = run_fused(torch, caller, p)
            for launch in range(3):
                got, tr = run_fused(torch, caller, p, compact=True)
                assert got.tobytes() == plain.tobytes() and tr.tobytes() == tr0.tobytes(), (mode, launch)
            rows[mode] = plain.tobytes()
    assert rows[""] == rows["two"] == rows["lookback"] and len(rows[""]) >= 64 * n_loci


def test_chain_timing_brackets_an_add_of_device_reads_and_a_flush(torch_cuda):
    """pisces_hip_set_chain_timing / pisces_hip_chain_time (bench.py's roofline_chain): refused before an add + flush pair has run with timing
    on; afterwards two positive spans; the rows are the rows of an untimed handle."""
    from pisces_amd import engine, synth
    p = synth.make_pileup(n_loci=3000, depth=60, seed=5, device="cuda", with_tuples=False)
    batch = synth.reads_of(p, p.base.shape[0], first_amplicon=0)
    d = engine.DeviceReadBatch.from_host(batch, "cuda:0")
    ref = p.ref.cpu().numpy()
    cfg = _abi.default_config()
    with engine.HipVariantCaller(cfg) as c:
        c.SetReference(ref)
        c.AddDeviceReads(d)
        want = c.Call(None)
    with engine.HipVariantCaller(cfg) as c:
        c.SetReference(ref)
        with pytest.raises(engine.PiscesHipError) as e:
            c.ChainTime()
        assert e.value.code == _abi.E_STATE
        c.SetChainTiming(True)
        with pytest.raises(engine.PiscesHipError):
            c.ChainTime()
        for _ in range(2):
            c.AddDeviceReads(d)
            got = c.Call(None)
            add_ms, flush_ms = c.ChainTime()
            assert 0.0 < add_ms < 100.0 and 0.0 < flush_ms < 100.0
            assert got.tobytes() == want.tobytes()
        c.SetChainTiming(False)
        with pytest.raises(engine.PiscesHipError):
            c.ChainTime()


def test_the_null_stream_is_refused_by_the_engine(torch_cuda):
    """HIP's null stream (what torch.cuda.current_stream().cuda_stream is in a default torch context) is not a stream the library can
    launch on -- the C ABI reads NULL as the handle's own stream -- so the engine makes the caller choose: None, or a real stream."""
    from pisces_amd import engine
    torch = torch_cuda
    with engine.HipVariantCaller(_abi.default_config()) as caller:
        assert caller.stream_handle() != 0 and caller.torch_stream().cuda_stream == caller.stream_handle()
        if torch.cuda.current_stream().cuda_stream == 0:
            with pytest.raises(ValueError, match="null stream"):
                caller.mark(0, torch.cuda.current_stream())
        with pytest.raises(ValueError, match="null stream"):
            caller.call_tiles(0, 0, 0, 0, 1, 0, 0, 0, 0, 0)
        caller.mark(0, None)
        caller.mark(1, caller.torch_stream())
        assert caller.marked_ms() >= 0.0


@pytest.mark.parametrize("n_loci,depth,seed", [(1000, 500, 1), (777, 37, 2), (64, 5000, 3), (130, 1, 4), (2000, 200, 5)])
def test_fused_kernel_matches_oracle_on_synthetic_pileups(torch_cuda, n_loci, depth, seed):
    from pisces_amd import engine, synth
    torch = torch_cuda
    p = synth.make_pileup(n_loci=n_loci, depth=depth, seed=seed, device="cuda")
    cfg = _abi.default_config()
    with engine.HipVariantCaller(cfg) as caller:
        got, tr = run_fused(torch, caller, p)
        got_c, tr_c = run_fused(torch, caller, p, compact=True)
        totals = caller.device_totals()
    pos, tup = synth.observations_of(p)
    exp, nloci = orc.run_observations(pos, tup, p.ref.cpu().numpy(), p.region_start, p.n_loci, cfg)
    assert_records_match(got, exp)
    assert got.tobytes() == got_c.tobytes()
    assert int(tr["n_candidate_loci"].sum()) == nloci
    assert totals["records"] == 2 * len(exp) and totals["candidate_loci"] == 2 * nloci and totals["tiles"] == 2 * p.n_tiles


@pytest.mark.parametrize("form", ["block", "wave", "wave2", "auto"])
@pytest.mark.parametrize("kw", [
    dict(n_loci=1500, depth=400, seed=11),                                     # BASELINE-like: one planted SNV per 100 loci
    dict(n_loci=700, depth=250, seed=12, snv_every=3, snv_offset=1, vaf_range=(0.01, 0.9)),  # ~21 variant candidates per tile
    dict(n_loci=500, depth=150, seed=13, base_error=0.09, snv_every=5, snv_offset=2),        # every error allele is a candidate: ~190 per tile
])
def test_every_form_of_the_hot_kernel_matches_oracle(torch_cuda, monkeypatch, form, kw):
    """The three forms of the hot kernel (one 4-wave workgroup / one wave / two waves per tile; `auto` is the product
    default) on pileups that exercise one, several and more-than-one-LDS-round of variant candidates per tile."""
    from pisces_amd import engine, synth
    torch = torch_cuda
    monkeypatch.setenv("PISCES_HIP_KERNEL", form)
    p = synth.make_pileup(device="cuda", **kw)
    cfg = _abi.default_config()
    with engine.HipVariantCaller(cfg) as caller:
        got, tr = run_fused(torch, caller, p)
    pos, tup = synth.observations_of(p)
    exp, nloci = orc.run_observations(pos, tup, p.ref.cpu().numpy(), p.region_start, p.n_loci, cfg)
    assert_records_match(got, exp)
    assert int(tr["n_candidate_loci"].sum()) == nloci
    assert int((_abi.info_category(got["info"]) == _abi.CAT_SNV).sum()) > 0


@pytest.mark.parametrize("overrides", [
    dict(min_base_call_quality=30, noise_level=30, min_frequency=0.005, variant_freq_filter=0.005,
         genotype_min_freq_filter=0.005, target_lod_frequency=0.005),                       # BASELINE config 5 settings
    dict(include_reference_calls=0),                                                          # plain VCF
    dict(strand_bias_model=_abi.SB_POISSON, filter_single_strand=1, max_variant_qscore=2000, max_genotype_qscore=2000),
    dict(emit_zero_coverage_refs=1, low_gq_filter=30, low_depth_filter=100, min_coverage=50),
    dict(noise_level=37, variant_qscore_filter=-1, no_call_filter_threshold=0.01, rmxn_max_repeat_length=-1),
])
def test_fused_kernel_config_variants(torch_cuda, overrides):
    from pisces_amd import engine, synth
    torch = torch_cuda
    p = synth.make_pileup(n_loci=900, depth=300, seed=21, device="cuda", vaf_range=(0.004, 0.9), p_lowq=0.05)
    cfg = _abi.default_config(**overrides)
    with engine.HipVariantCaller(cfg) as caller:
        got, _ = run_fused(torch, caller, p)
    pos, tup = synth.observations_of(p)
    exp, _ = orc.run_observations(pos, tup, p.ref.cpu().numpy(), p.region_start, p.n_loci, cfg)
    assert_records_match(got, exp)


def test_fused_kernel_edge_inputs(torch_cuda):
    """Empty tiles, ragged last tile, N reference bases, stitched direction, deletions, every-base-low-quality,
    homopolymer reference (RMxN), unaligned tuple segments."""
    from pisces_amd import engine
    torch = torch_cuda
    rng = np.random.default_rng(5)
    n_loci, start = 200, 11
    ref = np.frombuffer(b"N" * 10 + b"A" * 12 + b"C" * 12 + b"ACGT" * 60, dtype=np.uint8)[: start - 1 + n_loci + 20].copy()
    ref[60] = ord("N")
    n_obs = 30000
    pos = rng.integers(start, start + n_loci, n_obs).astype(np.int32)
    pos[pos == start + 150] = start + 151                      # one zero-coverage locus
    pos = pos[(pos < start + 64) | (pos >= start + 128)]       # one wholly empty tile
    n_obs = len(pos)
    allele = rng.choice(6, n_obs, p=[.3, .2, .2, .2, .02, .08])
    qual = np.where(allele == 5, 255, rng.choice([5, 19, 20, 37], n_obs, p=[.05, .05, .1, .8]))
    tup = _abi.tuple_pack(np.zeros(n_obs, np.uint32), rng.integers(0, 11, n_obs), rng.integers(0, 3, n_obs), allele, qual)
    cfg = _abi.default_config(emit_zero_coverage_refs=1)
    exp, nloci = orc.run_observations(pos, tup, ref, start, n_loci, cfg)
    # tile-bucket WITHOUT padding so segments start unaligned (exercises the scalar head/tail path)
    tiles = np.zeros(4, dtype=_abi.TILE_DTYPE)
    segs, cursor = [np.full(1, _abi.TUPLE_PAD, np.uint32)], 1
    for t in range(4):
        l0, l1 = t * 64, min(n_loci, t * 64 + 64)
        m = (pos >= start + l0) & (pos < start + l1)
        seg = _abi.tuple_with_locus(tup[m], pos[m] - (start + l0))
        tiles[t] = (start + l0, l1 - l0, cursor, cursor + len(seg))
        segs.append(seg)
        cursor += len(seg)
    d_tup = torch.from_numpy(np.concatenate(segs).view(np.int32)).cuda()
    d_tiles = torch.from_numpy(tiles.view(np.uint8)).cuda()
    d_ref = torch.from_numpy(ref).cuda()

    from types import SimpleNamespace
    view = SimpleNamespace(tuples=d_tup, tiles=d_tiles, n_tiles=4, ref=d_ref, ref_len=len(ref))
    with engine.HipVariantCaller(cfg) as caller:
        got, tr = run_fused(torch, caller, view)
    assert_records_match(got, exp)
    assert int(tr["n_candidate_loci"].sum()) == nloci == n_loci
    assert tr[1]["n_records"] == 64   # the empty tile still reports its zero-coverage reference rows
    assert int(tr["n_called"].sum()) >= len(got)


def test_rmxn_filter_on_device(torch_cuda):
    """An SNV between two >= 9-long homopolymers gets the RMxN filter (RMxNCalculator.cs:19-38)."""
    from pisces_amd import engine
    ref = np.frombuffer(b"GATTACAGAT" + b"A" * 10 + b"C" * 10 + b"GATTACAGAT", dtype=np.uint8).copy()
    pos_snv = 20   # last A, alt C
    pos, tup = [], []
    for p in range(5, 36):
        rb = _abi.ALLELE_OF_BASE[chr(ref[p - 1])]
        for i in range(200):
            a = _abi.ALLELE_C if (p == pos_snv and i < 40) else rb
            pos.append(p)
            tup.append(_abi.tuple_pack(0, 5, i % 2, a, 37))
    pos, tup = np.array(pos, np.int32), np.array(tup, np.uint32)
    cfg = _abi.default_config()
    exp, _ = orc.run_observations(pos, tup, ref, 1, len(ref), cfg)
    with engine.HipVariantCaller(cfg) as c:
        c.SetReference(ref)
        c.AddObservations(pos, tup)
        got = c.Call()
    assert_records_match(got, exp)
    row = got[got["position"] == pos_snv][0]
    assert _abi.info_category(row["info"]) == _abi.CAT_SNV and row["filter_bits"] & (1 << _abi.FILTER_RMXN)


# ---------------------------------------------------------------- streaming surface (IStateManager protocol)
def _read_dict(rd, default_q):
    d = {"pos": rd["pos"], "seq": rd["seq"],
         "cigar": orc.parse_cigar(rd["cigar"]) if "cigar" in rd else [("M", len(rd["seq"]))],
         "quals": rd.get("quals", [rd.get("qual", default_q)] * len(rd["seq"]))}
    if "dirs" in rd:
        d["dirs"] = [DIR[rd["dirs"]]] * len(rd["seq"])
    return d


def test_get_allele_count_reference_scenarios(torch_cuda):
    """RegionStateManagerTests.AddAndGetAlleleCounts / _PoorQualDeletions through AddAlleleCounts + GetAlleleCount."""
    from pisces_amd import engine
    g = json.load(open(os.path.join(G, "region_state.json")))
    for sc in g["poor_qual_deletions"]["scenarios"]:
        with engine.HipVariantCaller(_abi.default_config(min_base_call_quality=g["poor_qual_deletions"]["min_quality"],
                                                         noise_level=25)) as c:
            c.AddAlleleCounts([_read_dict(r, 30) for r in sc["reads"]])
            for e in sc["expect_ranges"]:
                for pos in range(e["from"], e["to"] + 1):
                    assert c.GetAlleleCount(pos, ALLELE[e["allele"]], DIR[e["dir"]]) == e["count"], (sc["name"], pos)
    a = g["add_and_get"]
    reads = [r for r in a["reads"] if "posmap_unmapped_index" not in r]
    with engine.HipVariantCaller(_abi.default_config(min_base_call_quality=a["min_quality"], noise_level=25)) as c:
        c.AddAlleleCounts([_read_dict(r, a["min_quality"]) for r in reads])
        st = orc.State(900, 300, min_bq=a["min_quality"])
        for r in reads:
            d = _read_dict(r, a["min_quality"])
            st.add_allele_counts(orc.make_read(d["pos"], d["seq"], cigar=d["cigar"], quals=d["quals"], dirs=d.get("dirs")))
        np.testing.assert_array_equal(c.GetCounts(900, 300), st.counts())
        with pytest.raises(engine.PiscesHipError):   # Assert.Throws<ArgumentException>(GetAlleleCount(0, ..))
            c.GetAlleleCount(0, _abi.ALLELE_A, _abi.DIR_FORWARD)
        # anchor windows served from device counts (AlleleCountHelperTests semantics)
        assert c.GetAlleleCount(1001, _abi.ALLELE_A, _abi.DIR_FORWARD) == 2
        assert c.GetAlleleCount(1001, _abi.ALLELE_A, _abi.DIR_FORWARD, minAnchor=1) == \
            st.get_allele_count(1001, _abi.ALLELE_A, _abi.DIR_FORWARD, 1)


def test_streaming_protocol_matches_oracle_and_block_schedule(torch_cuda):
    """SmallVariantCaller loop: AddAlleleCounts(read); Call(LastClearedPosition) ... Call(null).
    Blocks are emitted once upTo passes them (RegionStateManager.cs:283-334); the union equals the oracle."""
    from pisces_amd import engine, synth
    p = synth.make_pileup(n_loci=3300, depth=40, seed=9)          # CPU tensors: spans 4+ blocks of 1000
    cfg = _abi.default_config()
    batch = synth.reads_of(p)
    exp, _ = orc.run_reads(batch, p.ref.numpy(), p.region_start, p.n_loci, cfg)
    with engine.HipVariantCaller(cfg) as c:
        c.SetReference(p.ref.numpy())
        got = []
        n_amp = batch.n_reads // p.depth
        for a in range(n_amp):   # feed one amplicon (40 reads sharing a start) at a time, like position-sorted reads
            lo, hi = a * p.depth, (a + 1) * p.depth
            sub = _abi.ReadBatch.from_arrays(
                batch.position[lo:hi], batch.flags[lo:hi], batch.cigar_offset[lo:hi + 1] - batch.cigar_offset[lo],
                batch.cigar_op[batch.cigar_offset[lo]:batch.cigar_offset[hi]],
                batch.cigar_len[batch.cigar_offset[lo]:batch.cigar_offset[hi]],
                batch.seq_offset[lo:hi + 1] - batch.seq_offset[lo],
                batch.bases[batch.seq_offset[lo]:batch.seq_offset[hi]], batch.quals[batch.seq_offset[lo]:batch.seq_offset[hi]])
            c.AddAlleleCounts(sub)
            out = c.Call(int(batch.position[lo]) - 1)              # LastClearedPosition = lastReadPos - 1
            if len(out):
                # nothing beyond the cleared position, whole blocks only
                assert out["position"].max() <= int(batch.position[lo]) - 1
                assert out["position"].max() % 1000 == 0 or out["position"].max() == exp["position"].max()
            got.append(out)
        got.append(c.Call(None))
        got = np.concatenate(got)
        assert_records_match(got, exp)
        assert c.Stats()["reads"] == batch.n_reads
        assert len(c.Call(None)) == 0   # nothing left


def test_intervals_and_gapped_mnv_ref(torch_cuda):
    from pisces_amd import engine, synth
    p = synth.make_pileup(n_loci=1500, depth=80, seed=13)
    cfg = _abi.default_config(emit_zero_coverage_refs=1)
    pos, tup = synth.observations_of(p)
    intervals = [(p.region_start + 100, p.region_start + 180), (p.region_start + 900, p.region_start + 1010)]
    gapped = {p.region_start + 120: 7, p.region_start + 950: 10 ** 6}
    exp_all, _ = orc.run_observations(pos, tup, p.ref.numpy(), p.region_start, p.n_loci, cfg)
    keep = np.zeros(len(exp_all), bool)
    for s, e in intervals:
        keep |= (exp_all["position"] >= s) & (exp_all["position"] <= e)
    with engine.HipVariantCaller(cfg) as c:
        c.SetReference(p.ref.numpy())
        c.SetIntervals(intervals)
        c.AddObservations(pos, tup)
        got = c.Call()
    assert_records_match(got, exp_all[keep])
    # gapped-MNV reference counts lower Reference AlleleSupport / SNV ReferenceSupport (CoverageCalculator.cs:82-97)
    st = orc.State(p.region_start, p.n_loci, min_bq=20)
    # (oracle side: same observations, then AddGappedMnvRefCount, then call)
    import ctypes as C
    with engine.HipVariantCaller(_abi.default_config()) as c:
        c.SetReference(p.ref.numpy())
        c.AddObservations(pos, tup)
        c.AddGappedMnvRefCount(gapped)
        got = c.Call()
    base, _ = orc.run_observations(pos, tup, p.ref.numpy(), p.region_start, p.n_loci, _abi.default_config())
    for gp, gc in gapped.items():
        b = base[base["position"] == gp]
        o = got[got["position"] == gp]
        assert len(b) == len(o)
        for rb, ro in zip(b, o):
            if _abi.info_category(rb["info"]) == _abi.CAT_REFERENCE:
                assert ro["allele_support"] == max(0, rb["allele_support"] - gc)
            else:
                assert ro["reference_support"] == max(0, rb["reference_support"] - gc)
    untouched = ~np.isin(base["position"], list(gapped))
    assert_records_match(got[~np.isin(got["position"], list(gapped))], base[untouched])


def test_error_paths(torch_cuda):
    from pisces_amd import engine
    with engine.HipVariantCaller() as c:
        with pytest.raises(engine.PiscesHipError) as e:
            c.AddObservations(np.array([0], np.int32), np.array([0], np.uint32))   # position <= 0
        assert e.value.code == _abi.E_INVALID_ARG and "greater than 0" in e.value.message
        c.AddObservations(np.array([5], np.int32), np.array([_abi.tuple_pack(0, 5, 0, 0, 30)], np.uint32))
        with pytest.raises(engine.PiscesHipError) as e:
            c.Call()                                                                # no reference set
        assert e.value.code == _abi.E_STATE
        # a malformed batch is rejected as a whole, before any of it is committed (argument errors of the reference's walk:
        # RegionStateManager.cs:363-364 position, Read.ValidateCigar, DirectionType range)
        good = {"pos": 20, "seq": "ACGTACGT", "cigar": [("M", 8)], "quals": [30] * 8, "reverse": False}
        for bad, needle in (({"pos": 0, "seq": "ACGT", "cigar": [("M", 4)], "quals": [30] * 4, "reverse": False}, "greater than 0"),
                            ({"pos": 9, "seq": "ACGT", "cigar": [("M", 3), ("I", 4)], "quals": [30] * 4, "reverse": False}, "CIGAR"),
                            ({"pos": 9, "seq": "ACGT", "cigar": [("M", 4)], "quals": [30] * 4, "reverse": False, "dirs": [0, 1, 3, 0]}, "CIGAR")):
            before = c.Stats()
            with pytest.raises(engine.PiscesHipError) as e:
                c.AddAlleleCounts(_abi.ReadBatch([good, bad]))
            assert e.value.code == _abi.E_INVALID_ARG and needle in e.value.message
            assert c.Stats() == before and c.GetCounts(20, 8).sum() == 0
        c.AddAlleleCounts(_abi.ReadBatch([good]))
        assert c.GetCounts(20, 8).sum() == 8 and c.Stats()["reads"] == 1 and c.Stats()["reads_skipped"] == 0
    with pytest.raises(engine.PiscesHipError):
        engine.HipVariantCaller(_abi.default_config(strand_bias_model=7))
    with pytest.raises(engine.PiscesHipError):
        engine.HipVariantCaller(_abi.default_config(abi_version=99))
    # candidates and forced alleles the caller brings: malformed entries are refused, invalid forced alleles dropped as Factory drops them
    with engine.HipVariantCaller() as c:
        c.SetReference(np.frombuffer(b"ACGT" * 600, dtype=np.uint8))
        for bad in ({"position": 0, "category": 0, "ref": "A", "alt": "C"}, {"position": 5, "category": 9, "ref": "A", "alt": "C"},
                    {"position": 5, "category": 0, "ref": "", "alt": "C"}):
            with pytest.raises(engine.PiscesHipError) as e:
                c.AddCandidates([bad])
            assert e.value.code == _abi.E_INVALID_ARG
        assert c.GetCandidates() == []
        c.AddCandidates([{"position": 9, "category": _abi.CAT_DELETION, "ref": "ACG", "alt": "A", "support_by_dir": (3, 2, 0)}] * 2)
        got = c.GetCandidates()
        assert len(got) == 1 and got[0]["support_by_dir"] == [6, 4, 0] and (got[0]["ref"], got[0]["alt"]) == ("ACG", "A")   # merged (RegionState.AddCandidate)
        # ref == alt and an ALT outside A/C/G/T are no forced alleles (Factory.IsValidAlt); duplicates count once (a HashSet)
        c.SetForcedAlleles([(30, "G", "G"), (31, "T", "N"), (32, "A", "C"), (32, "A", "C"), (33, "AC", "A")])
        c.AddAlleleCounts(_abi.ReadBatch([{"pos": 20, "seq": "TACGTACGTACGTACGTACG", "cigar": [("M", 20)], "quals": [30] * 20, "reverse": False}]))
        recs, alleles = c.CallWithAlleles()
        forced_rows = [(int(r["position"]), a) for r, a in zip(recs, alleles) if (int(r["filter_bits"]) >> _abi.FILTER_FORCED_REPORT) & 1]
        assert forced_rows == [(32, ("A", "C")), (33, ("AC", "A"))]
        with pytest.raises(engine.PiscesHipError) as e:
            c.SetForcedAlleles([(40, "A", "C")])        # after forced alleles became candidates: refused
        assert e.value.code == _abi.E_INVALID_ARG


# ---------------------------------------------------------------- BASELINE sizes: size-independent properties

def assert_random_windows_match_oracle(p, got, cfg, n_windows=16, window_loci=1000, seed=7):
    """Seeded random windows of one 1000-locus block each over the WHOLE launch — always the first and the last block, the rest anywhere
    (tiles late in a multi-round launch, the far end of the position range) — compared record for record with the oracle run on those
    tiles' observations (a tile's records depend on its own observations only).  Returns the loci checked."""
    from pisces_amd import synth
    tiles = p.tiles.cpu().numpy().view(_abi.TILE_DTYPE)
    per = max(1, min(p.n_tiles, -(-window_loci // max(int(tiles[0]["n_loci"]), 1))))
    rng = np.random.default_rng(seed)
    firsts = {0, p.n_tiles - per}
    while len(firsts) < min(n_windows, p.n_tiles - per + 1):
        firsts.add(int(rng.integers(0, p.n_tiles - per + 1)))
    ref = p.ref.cpu().numpy()
    checked = 0
    for t0 in sorted(firsts):
        start = int(tiles[t0]["start_position"])
        n = int(tiles[t0 + per - 1]["start_position"] + tiles[t0 + per - 1]["n_loci"]) - start
        pos, tup = synth.observations_of(p, per, first_tile=t0)
        exp, _ = orc.run_observations(pos, tup, ref, start, n, cfg)
        lo, hi = np.searchsorted(got["position"], [start, start + n])
        assert_records_match(got[lo:hi], exp)
        checked += n
    return checked

def test_full_size_properties_config2(torch_cuda):
    """100k loci x 500x (BASELINE config 2): exact depth everywhere, one candidate locus per locus, linearity
    (two launches over halves == one launch), idempotence (same launch twice), a sampled slice against the oracle."""
    from pisces_amd import engine, synth
    torch = torch_cuda
    p = synth.make_pileup(n_loci=100_000, depth=500, device="cuda")
    cfg = _abi.default_config()
    with engine.HipVariantCaller(cfg) as caller:
        got, tr = run_fused(torch, caller, p)
        again, _ = run_fused(torch, caller, p)
    assert got.tobytes() == again.tobytes()
    assert int(tr["n_candidate_loci"].sum()) == 100_000
    assert ((got["total_coverage"] + got["num_no_calls"]) == 500).all()
    assert (np.diff(got["position"]) >= 0).all()
    ref_rows = np.array([_abi.info_category(i) == _abi.CAT_REFERENCE for i in got["info"][:5000]])
    assert (got["allele_support"][:5000][ref_rows] == got["reference_support"][:5000][ref_rows]).all()
    # a checksum of checksums: total support of all alleles equals the number of quality-passing ACGT observations
    # that match a called allele; cheaper: sum(coverage) + sum(no calls) per locus == depth (done above), and the
    # sample below pins the rest against the oracle
    n_t = 40
    pos, tup = synth.observations_of(p, n_t)
    exp, _ = orc.run_observations(pos, tup, p.ref.cpu().numpy(), p.region_start, n_t * 64, cfg)
    assert_records_match(got[got["position"] < p.region_start + n_t * 64], exp)
    # and sixteen random 1000-locus windows over the whole launch (the last block among them) against the oracle
    assert assert_random_windows_match_oracle(p, got, cfg) >= 16_000


@pytest.mark.gpu
@pytest.mark.parametrize("n_loci,depth,synth_kw,cfg_kw", [
    (1_000_000, 2000, {}, {}),                                                             # BASELINE config 3 size (SNV-only pileup)
    (3_750_000, 200, {}, {}),                                                              # config 4: one GPU's eighth of 30 M loci x 200x
    (100_000, 5000, dict(vaf_range=(0.005, 0.005)),                                        # 0.5 % VAF at 5000x, gVCF, filters on, with a noise level
     dict(min_frequency=0.002, variant_freq_filter=0.002, noise_level=35)),                 # (-nl 35) at which every planted variant stands out
    (100_000, 5000, dict(vaf_range=(0.005, 0.005), snv_every=50, snv_offset=17, q_lo=12),  # BASELINE config 5 as SURVEY 8d states it: -minbq 30 => NL 30,
     dict(min_base_call_quality=30, noise_level=30, min_frequency=0.005, variant_freq_filter=0.005,   # -minvf 0.005, -sbfilter 0.5, -vqfilter 30, gVCF;
          strand_bias_threshold=0.5, variant_qscore_filter=30)),                                     # 25 expected reads per site: about half clear 0.5 %
], ids=["config3_1Mx2000", "config4_shard_3.75Mx200", "config5_100kx5000_lowvaf_nl35", "config5_100kx5000_as_stated"])
def test_full_size_properties_other_baseline_configs(torch_cuda, n_loci, depth, synth_kw, cfg_kw):
    """The other BASELINE sizes through size-independent properties: idempotence (the same launch twice, byte for byte), exact depth at
    every locus, one candidate locus per locus, sortedness, planted low-frequency variants found, and the first 640 loci against the oracle."""
    from pisces_amd import engine, synth
    torch = torch_cuda
    p = synth.make_pileup(n_loci=n_loci, depth=depth, seed=5, device="cuda", **synth_kw)
    p.base = p.qual = None
    torch.cuda.empty_cache()
    cfg = _abi.default_config(**cfg_kw)
    with engine.HipVariantCaller(cfg) as caller:
        got, tr = run_fused(torch, caller, p)
        again, _ = run_fused(torch, caller, p)
    assert got.tobytes() == again.tobytes()
    assert int(tr["n_candidate_loci"].sum()) == n_loci
    assert ((got["total_coverage"] + got["num_no_calls"]) == depth).all()
    assert (np.diff(got["position"]) >= 0).all()
    cats = (got["info"] >> 4) & 7
    n_planted = len(p.planted)
    found = np.isin(got["position"][cats == _abi.CAT_SNV] - p.region_start, p.planted.cpu().numpy() if hasattr(p.planted, "cpu") else p.planted)
    # at exactly 0.5 % VAF against a 0.5 % emit threshold a planted site is called when its sampled support reaches the threshold: about half
    min_found = 0.3 if cfg_kw.get("min_frequency", 0) >= 0.005 else 0.9
    assert found.sum() >= min_found * n_planted, (int(found.sum()), n_planted)
    n_t = 10
    pos, tup = synth.observations_of(p, n_t)
    exp, _ = orc.run_observations(pos, tup, p.ref.cpu().numpy(), p.region_start, n_t * 64, cfg)
    assert_records_match(got[got["position"] < p.region_start + n_t * 64], exp)
    # sixteen random 1000-locus windows over the whole launch — the last block, tiles of late rounds, the far end of the position range
    assert assert_random_windows_match_oracle(p, got, cfg) >= 16_000
    del p
    torch.cuda.empty_cache()


# ---------------------------------------------------------------- insertions / deletions (host finder + spanning coverage on the device)
def _indel_reads(rng, ref, start, depth, plan, read_len=150, p_lowq=0.02):
    """`depth` reads of one amplicon at `start`; plan = list of (offset_in_read, kind 'D'|'I', length, fraction)."""
    reads = []
    for i in range(depth):
        ops, seq, ref_off = [], [], 0
        cursor = 0   # offset in amplicon (reference coordinates, 0-based from start)
        events = sorted([e for e in plan if rng.random() < e[3]], key=lambda e: e[0])
        for off, kind, ln, _ in events:
            if off <= cursor:
                continue
            ops.append(("M", off - cursor))
            seq.append(ref[start - 1 + cursor: start - 1 + off])
            cursor = off
            if kind == "D":
                ops.append(("D", ln))
                cursor += ln
            else:
                ops.append(("I", ln))
                seq.append(bytes(rng.choice(list(b"ACGT"), ln).astype(np.uint8)))
        tail = read_len - sum(l for o, l in ops if o in "MI")
        if tail > 0:
            ops.append(("M", tail))
            seq.append(ref[start - 1 + cursor: start - 1 + cursor + tail])
        s = b"".join(seq)
        q = np.where(rng.random(len(s)) < p_lowq, 12, 37).astype(np.uint8)
        reads.append({"pos": start, "cigar": ops, "seq": s.decode(), "quals": q.tolist(), "reverse": bool(i % 2)})
    return reads


def test_indels_match_oracle_with_allele_strings(torch_cuda):
    from pisces_amd import engine
    rng = np.random.default_rng(42)
    ref = bytes(rng.choice(list(b"ACGT"), 1400).astype(np.uint8))
    ref = ref[:300] + b"A" * 12 + ref[312:]          # a homopolymer: deletions inside it get the RMxN filter
    reads = []
    reads += _indel_reads(rng, ref, 101, 80, [(40, "D", 3, 0.3), (90, "I", 2, 0.25), (3, "I", 1, 0.2), (140, "D", 5, 0.2)])
    reads += _indel_reads(rng, ref, 251, 80, [(52, "D", 2, 0.4), (100, "I", 6, 0.1), (100, "D", 1, 0.1)])
    reads += _indel_reads(rng, ref, 401, 60, [(75, "D", 10, 0.5)])
    reads += _indel_reads(rng, ref, 930, 70, [(68, "D", 7, 0.35), (20, "I", 3, 0.3)])   # deletion across the 1000 block edge
    batch = _abi.ReadBatch(reads)
    refa = np.frombuffer(ref, dtype=np.uint8)
    for overrides in (dict(), dict(include_reference_calls=0), dict(min_frequency=0.3, variant_freq_filter=0.3)):
        cfg = _abi.default_config(**overrides)
        exp, exp_alleles, _, exp_called = orc.run_reads_full(batch, refa, 1, len(ref), cfg)
        with engine.HipVariantCaller(cfg) as c:
            c.SetReference(refa)
            c.AddAlleleCounts(batch)
            cands = c.GetCandidates()
            got, got_alleles = c.CallWithAlleles()
            assert c.Stats()["TotalNumCalled"] == exp_called
        assert any(x["category"] == _abi.CAT_DELETION for x in cands) and any(x["category"] == _abi.CAT_INSERTION for x in cands)
        assert_records_match(got, exp)
        assert got_alleles == exp_alleles
        cats = {int(_abi.info_category(i)) for i in got["info"]}
        if not overrides:
            assert {_abi.CAT_DELETION, _abi.CAT_INSERTION, _abi.CAT_SNV, _abi.CAT_REFERENCE} >= cats >= {_abi.CAT_DELETION, _abi.CAT_INSERTION}
            rmxn_rows = [(r, a) for r, a in zip(got, got_alleles)
                         if _abi.info_category(r["info"]) == _abi.CAT_DELETION and 295 <= r["position"] <= 312]
    # the deletion inside the A homopolymer carries the RMxN filter (and the oracle agrees: filter_bits matched above)
    assert rmxn_rows and all(r["filter_bits"] & (1 << _abi.FILTER_RMXN) for r, _ in rmxn_rows)


def test_spanning_alleles_hold_their_block(torch_cuda):
    """A block whose deletion reaches past upTo is not emitted (RegionStateManager.cs:304-308)."""
    from pisces_amd import engine
    rng = np.random.default_rng(3)
    ref = bytes(rng.choice(list(b"ACGT"), 2400).astype(np.uint8))
    reads = _indel_reads(rng, ref, 930, 50, [(68, "D", 7, 0.5)])     # deletion 998..1004, anchor 997, endpoint 1005
    batch = _abi.ReadBatch(reads)
    refa = np.frombuffer(ref, dtype=np.uint8)
    cfg = _abi.default_config()
    exp, exp_alleles, _, _ = orc.run_reads_full(batch, refa, 1, len(ref), cfg)
    with engine.HipVariantCaller(cfg) as c:
        c.SetReference(refa)
        c.AddAlleleCounts(batch)
        assert len(c.Call(1001)) == 0          # block 1 is held: MaxAlleleEndpoint 1005 > 1001
        out2, al2 = c.CallWithAlleles(2001)    # entering block 3: blocks 1 and 2 are released together
        rest, al3 = c.CallWithAlleles(None)
    got = np.concatenate([out2, rest])
    assert_records_match(got, exp)
    assert al2 + al3 == exp_alleles


def test_device_read_walk_matches_oracle_on_random_cigars(torch_cuda):
    """pisces_hip_add_reads walks the reads ON THE DEVICE (expand_reads_kernel) into the observation log;
    the anchor-resolved counts it serves (IAlleleSource.GetAlleleCount over pisces_hip_get_counts) must equal the oracle's
    AddAlleleCounts (RegionStateManager.cs:118-220) for arbitrary CIGARs: insertions, deletions, soft clips at either end,
    terminal deletions, N bases, low qualities, stitched per-base directions, reads longer than one wave."""
    from pisces_amd import engine
    rng = np.random.default_rng(23)
    reads = []
    for i in range(600):
        ops = []
        for _ in range(int(rng.integers(1, 7))):
            ops.append((str(rng.choice(list("MMMIDSN"))), int(rng.integers(1, 40 if i % 7 == 0 else 12))))
        ops = [(o, l) for k, (o, l) in enumerate(ops) if o != "S" or k in (0, len(ops) - 1)]
        if not any(o == "M" for o, _ in ops):
            ops.append(("M", 5))
        if i % 11 == 0:
            ops.append(("D", int(rng.integers(1, 6))))          # read ends in a deletion
        if i % 13 == 0:
            ops += [("D", int(rng.integers(1, 6))), ("S", int(rng.integers(1, 5)))]   # ... before a soft clip
        rl = sum(l for o, l in ops if o in "MIS")
        rd = {"pos": int(rng.integers(940, 1150)), "cigar": ops,
              "seq": "".join(rng.choice(list("ACGTN"), rl, p=[.24, .24, .24, .24, .04])),
              "quals": rng.choice([10, 25, 37], rl, p=[.15, .2, .65]).astype(np.uint8).tolist(),
              "reverse": bool(rng.integers(0, 2))}
        if i % 5 == 0:
            rd["dirs"] = rng.integers(0, 3, rl).astype(np.uint8).tolist()
        reads.append(rd)
    st = orc.State(900, 2200, min_bq=20)
    for d in reads:
        assert st.add_allele_counts(orc.make_read(d["pos"], d["seq"], cigar=d["cigar"], quals=d["quals"], reverse=d["reverse"],
                                                  dirs=d.get("dirs"))) == 0
    with engine.HipVariantCaller(_abi.default_config()) as c:
        for k in range(0, len(reads), 97):                        # several add_reads calls: the log grows and keeps its content
            c.AddAlleleCounts(_abi.ReadBatch(reads[k:k + 97]))
        got = c.GetCounts(900, 2200)
        n_reads = c.Stats()["reads"]
    exp = st.counts()
    np.testing.assert_array_equal(got.reshape(exp.shape), exp)
    assert n_reads == len(reads)


def test_committed_fixture_without_the_oracle(torch_cuda):
    """The streaming surface against tests/golden/synthetic_small.npz: committed inputs, committed expected records."""
    from pisces_amd import engine
    z = np.load(os.path.join(G, "synthetic_small.npz"))
    exp = z["expected"].view(_abi.CALLED_ALLELE_DTYPE)
    with engine.HipVariantCaller(_abi.default_config()) as c:
        c.SetReference(z["ref"])
        c.AddObservations(z["positions"], z["tuples"])
        got = c.Call(None)
    assert_records_match(got, exp)


def test_two_interval_shards_on_one_gpu_equal_the_unsharded_run(torch_cuda):
    """SURVEY 8e on the device: ONE interval set partitioned by shard.partition_intervals; each shard is made on its own (only its
    locus range of the global pileup), called by its own handle, and the rank-order concatenation is the unsharded launch byte for
    byte, with summaries that add up (what bench.py does across GPUs).  Then the cut itself through the streaming surface: two handles
    fed by shard.reads_for_shard (halo reads on both sides) against one handle (shard.verify_cut)."""
    from pisces_amd import engine, shard, synth
    torch = torch_cuda
    total, depth, seed = 12_300, 120, 31
    origin = synth.READ_LEN + 1
    n_amp = -(-total // synth.READ_LEN)
    intervals = [(origin + a * synth.READ_LEN, origin + min((a + 1) * synth.READ_LEN, total) - 1) for a in range(n_amp)]
    cfg = _abi.default_config()
    whole_p = synth.make_pileup(total, depth, seed=seed, device="cuda")
    with engine.HipVariantCaller(cfg) as c:
        whole, tr_w = run_fused(torch, c, whole_p)
    for world in (2, 3, 8):   # (8: BASELINE config 4's rank count)
        parts = shard.partition_intervals(intervals, world, block_size=cfg.block_size)
        assert len(parts) == world and all(hi >= lo for lo, hi, _ in parts)
        got, n_rec, n_loci = [], 0, 0
        for lo, hi, clipped in parts:
            lo, hi = max(lo, intervals[0][0]), min(hi, intervals[-1][1])
            p = synth.make_pileup(hi - lo + 1, depth, seed=seed, device="cuda", first_locus=lo - origin, total_loci=total)
            with engine.HipVariantCaller(cfg) as c:
                recs, tr = run_fused(torch, c, p, ref_start=p.ref_start)
            assert recs["position"].min() >= lo and recs["position"].max() <= hi
            got.append(recs)
            n_rec += int(tr["n_records"].sum())
            n_loci += int(tr["n_candidate_loci"].sum())
        assert np.concatenate(got).tobytes() == whole.tobytes(), world
        assert n_rec == int(tr_w["n_records"].sum()) and n_loci == total
    # the cut of the 2-way partition through the streaming surface
    cut = shard.partition_intervals(intervals, 2, block_size=cfg.block_size)[1][0]
    g_lo, g_hi = cut - 1000 - origin - synth.READ_LEN, cut + 999 - origin + synth.READ_LEN
    wp = synth.make_pileup(g_hi - g_lo + 1, depth, seed=seed, device="cuda", first_locus=g_lo, total_loci=total)
    rb = synth.reads_of(wp)
    arrays = (rb.position, rb.flags, rb.cigar_offset, rb.cigar_op, rb.cigar_len, rb.seq_offset, rb.bases, rb.quals)
    ref_slice = wp.ref.cpu().numpy()
    ref_full = np.full(wp.ref_start - 1 + len(ref_slice), ord("N"), dtype=np.uint8)
    ref_full[wp.ref_start - 1:] = ref_slice
    n, counted = shard.verify_cut(lambda: engine.HipVariantCaller(cfg), ref_full, arrays, cut - 1000, cut, cut + 999, halo=synth.READ_LEN + 10)
    assert n == 2000 and counted > 0
    # ... and those window records are the unsharded launch's
    m = (whole["position"] >= cut - 1000) & (whole["position"] <= cut + 999)
    assert int(m.sum()) == n


def test_summary_reduce_through_the_c_abi(torch_cuda):
    """pisces_hip_comm_unique_id / comm_init / reduce_summary / comm_destroy (RCCL bound at run time): on a one-GPU box the communicator
    has one rank and the all-reduce returns the totals as they are; without a communicator the call is the identity."""
    from pisces_amd import engine
    with engine.HipVariantCaller(_abi.default_config()) as c:
        assert c.reduce_summary([1, 2, 3, 4]) == [1, 2, 3, 4]
        uid = engine.HipVariantCaller.comm_unique_id()
        assert len(uid) == 128 and any(uid)
        c.comm_init(uid, 0, 1)
        assert c.reduce_summary([5, 1 << 40, 7, 8]) == [5, 1 << 40, 7, 8]
        with pytest.raises(Exception):
            c.comm_init(uid, 0, 1)   # one communicator per handle


def _cand_dicts_oracle(reads, ref, call_mnvs, max_len, max_gap, snvs=True):
    exp = []
    for d in reads:
        rd = orc.make_read(d["pos"], d["seq"], cigar=d["cigar"], quals=d["quals"], reverse=d.get("reverse", False), dirs=d.get("dirs"))
        for c in orc.find_candidates(rd, ref.decode() if isinstance(ref, bytes) else ref, call_mnvs=call_mnvs, max_mnv=max_len, max_gap=max_gap):
            if not snvs and c.category in (_abi.CAT_SNV, _abi.CAT_MNV):
                continue
            exp.append({"position": c.position, "category": c.category, "ref": c.ref.decode(), "alt": c.alt.decode(),
                        "support_by_dir": list(c.support_by_dir), "well_anchored_by_dir": list(c.well_anchored_by_dir),
                        "open_left": bool(c.open_left), "open_right": bool(c.open_right)})
    return exp


def test_device_finder_on_the_reference_finder_cases(torch_cuda):
    """Row f1: candidate discovery on the DEVICE (find_count_kernel / find_emit_kernel through pisces_hip_find_candidates_device, the
    kernels pisces_hip_add_reads enqueues) over all 106 reads of the reference's VariantFinderTests (tests/golden/finder_cases.json:
    SNV, MNV, deletion and insertion suites): every expected candidate with its open ends, nothing else, and equal to the oracle's
    finder field by field."""
    from pisces_amd import engine
    g = json.load(open(os.path.join(G, "finder_cases.json")))
    cat = {"Snv": _abi.CAT_SNV, "Mnv": _abi.CAT_MNV, "Insertion": _abi.CAT_INSERTION, "Deletion": _abi.CAT_DELETION}
    n_checked = n_reads = 0
    for case in g["cases"]:
        if not case["read"]:
            continue
        start = 101
        ops = orc.parse_cigar(case["cigar"])
        clip = ops[0][1] if ops and ops[0][0] == "S" else 0
        ref = ("N" * (start - 1 - clip) + case["ref_under_read"] + "NNNNN").encode()
        rd = {"pos": start, "cigar": ops, "seq": case["read"], "quals": case["quals"], "reverse": False}
        cfg = _abi.default_config(min_base_call_quality=g["min_base_call_quality"])
        with engine.HipVariantCaller(cfg) as c:
            c.SetReference(ref)
            got = c.FindCandidates(_abi.ReadBatch([rd]), True, g["call_mnvs"], case["max_mnv_length"], case["max_gap"])
        n_reads += 1
        assert got == _cand_dicts_oracle([rd], ref, g["call_mnvs"], case["max_mnv_length"], case["max_gap"]), case
        assert len(got) == case["expected_count"], (case["cigar"], case["read"], got)
        key = lambda x: (x["position"], x["category"], x["ref"], x["alt"])
        for e in (case["expected"] if case["expected_count"] else []):
            m = [x for x in got if key(x) == (e["coord"] + start, cat[e["type"]], e["ref"], e["alt"])]
            assert len(m) == 1, (case, got)
            if e["open_left"] is not None:
                assert m[0]["open_left"] == e["open_left"]
            if e["open_right"] is not None:
                assert m[0]["open_right"] == e["open_right"]
            n_checked += 1
    assert n_reads >= 100 and n_checked >= 90


def test_device_finder_deletion_directions_and_random_reads(torch_cuda):
    """The 14 RunDeletionScenarios reads (support direction of a deletion inside a stitched read, deletion_directions from the XD tag)
    on the device finder, then 500 random reads (insertions, deletions, soft clips, mismatches, N bases, low qualities, stitched
    directions, long insertions that go through the byte pool) against the oracle's finder for four MNV settings, in one batch each:
    records in read order, field by field."""
    from pisces_amd import engine
    doc = json.load(open(os.path.join(G, "support_direction_deletion_cases.json")))
    code = {"Forward": _abi.DIR_FORWARD, "Reverse": _abi.DIR_REVERSE, "Stitched": _abi.DIR_STITCHED}
    rd = doc["read"]
    reads = [{"pos": rd["position"], "cigar": orc.parse_cigar(rd["cigar"]), "seq": rd["sequence"], "quals": [30] * len(rd["sequence"]),
              "xd": f'{c["num_forward"]}F{c["num_stitched"]}S{c["num_reverse"]}R'} for c in doc["cases"]]
    with engine.HipVariantCaller(_abi.default_config()) as c:
        c.SetReference(b"ATCG" * 5)
        got = c.FindCandidates(_abi.ReadBatch(reads), False)
    assert len(got) == len(doc["cases"]) == 14
    for x, case in zip(got, doc["cases"]):
        want = [0, 0, 0]
        want[code[case["expected"]]] = 1
        assert x["category"] == _abi.CAT_DELETION and x["support_by_dir"] == want, case["name"]
    rng = np.random.default_rng(2026)
    ref = bytes(rng.choice(list(b"ACGT"), 700).astype(np.uint8))
    reads = []
    for _ in range(500):
        ops = []
        for k in range(int(rng.integers(1, 5))):
            o = str(rng.choice(list("MMMMIDS")))
            ops.append((o, int(rng.integers(33, 60)) if (o == "I" and rng.random() < 0.1) else int(rng.integers(1, 30))))
        ops = [(o, l) for i, (o, l) in enumerate(ops) if o != "S" or i in (0, len(ops) - 1)]
        if not any(o == "M" for o, _ in ops):
            ops.insert(len(ops) // 2, ("M", 12))
        pos = int(rng.integers(20, 500))
        seq, rp = [], pos
        for o, l in ops:
            if o == "M":
                seg = bytearray(ref[rp - 1: rp - 1 + l])
                seg += bytes(rng.choice(list(b"ACGT"), l - len(seg)).astype(np.uint8)) if len(seg) < l else b""
                for i in range(len(seg)):
                    if rng.random() < 0.25:
                        seg[i] = int(rng.choice(list(b"ACGTN"), p=[.24, .24, .24, .24, .04]))
                seq.append(bytes(seg).decode())
                rp += l
            elif o == "D":
                rp += l
            else:
                seq.append("".join(rng.choice(list("ACGT"), l)))
        seq = "".join(seq)
        rl = len(seq)
        stitched = rng.random() < 0.3
        reads.append({"pos": pos, "cigar": ops, "seq": seq, "quals": rng.choice([10, 25, 37], rl, p=[.1, .15, .75]).astype(np.uint8).tolist(),
                      "reverse": bool(rng.integers(0, 2)), "dirs": rng.choice([0, 1, 2], rl).tolist() if stitched else None})
    with engine.HipVariantCaller(_abi.default_config()) as c:
        c.SetReference(ref)
        for snvs, call_mnvs, max_len, max_gap in ((False, False, 3, 1), (True, False, 3, 1), (True, True, 3, 1), (True, True, 40, 10), (True, True, 2, 0)):
            got = c.FindCandidates(_abi.ReadBatch(reads), snvs, call_mnvs, max_len, max_gap)
            exp = _cand_dicts_oracle(reads, ref, call_mnvs, max_len, max_gap, snvs=snvs)
            assert len(exp) > 100
            assert got == exp, (snvs, call_mnvs, max_len, max_gap)
            assert any(len(x["alt"]) > 33 for x in got)   # the byte pool was exercised


def test_flush_into_a_buffer_that_is_too_small_is_repeatable(torch_cuda):
    """SURVEY 8b ownership rule: output buffers are caller-allocated; pisces_hip_flush returns PISCES_E_BUFFER_TOO_SMALL (-2) with the
    needed count when the batch does not fit, changes nothing, and the repeated call with a larger buffer returns the identical batch
    (the blocks are retired only then).  Checked mid-stream and at the final flush, with insertions / deletions in the batch."""
    import ctypes as C
    from pisces_amd import engine, synth
    from pisces_amd._native import lib
    p = synth.make_pileup(2300, 60, seed=17)
    reads = synth.reads_of(p)
    ref = p.ref.cpu().numpy()
    cfg = _abi.default_config()
    with engine.HipVariantCaller(cfg) as a, engine.HipVariantCaller(cfg) as b:
        for c in (a, b):
            c.SetReference(ref)
            c.AddAlleleCounts(reads)
        for up_to in (p.region_start + 1500, None):
            want = a.Call(up_to, capacity=1 << 16)
            assert len(want) > 500
            n = C.c_int64(0)
            small = np.zeros(10, dtype=_abi.CALLED_ALLELE_DTYPE)
            for cap in (0, 10, len(want) - 1):
                rc = lib.pisces_hip_flush(b.handle, -1 if up_to is None else up_to, small.ctypes.data, min(cap, 10), C.byref(n))
                assert rc == _abi.E_BUFFER_TOO_SMALL and n.value == len(want), (rc, n.value, len(want))
            # the batch is made and waits for its buffers: until it is taken nothing may move the state
            with pytest.raises(Exception) as e:
                b.AddAlleleCounts(reads)
            assert e.value.code == _abi.E_STATE
            rc = lib.pisces_hip_flush(b.handle, 7 if up_to is None else -1, small.ctypes.data, 10, C.byref(n))
            assert rc == _abi.E_STATE
            out = np.zeros(len(want), dtype=_abi.CALLED_ALLELE_DTYPE)
            rc = lib.pisces_hip_flush(b.handle, -1 if up_to is None else up_to, out.ctypes.data, len(out), C.byref(n))
            assert rc == 0 and n.value == len(want)
            assert out.tobytes() == want.tobytes()
        assert a.Stats() == b.Stats()


@pytest.mark.parametrize("n_loci,depth", [(1_000_000, 2000)], ids=["config3_1Mx2000_snv_mnv_indel"])
def test_full_size_config3_mix_through_the_streaming_surface(torch_cuda, n_loci, depth):
    """BASELINE config 3 with its real mix at full size: 1 M loci x 2000x, SNVs + MNVs (2-3 bases) + deletions (1-10) + insertions
    (1-6), MNV calling on (-callmnvs true -maxmnvlength 3 -maxgapbetweenmnv 1), reads walked, candidates found (find_emit_kernel),
    collapsed, reallocated and called through pisces_hip_add_reads / pisces_hip_flush.  Properties over the whole run: sorted output,
    exact depth at every point record, every planted MNV / deletion / insertion called at its position, reads counted once; and the
    first two blocks, flushed block by block as SmallVariantCaller does, against the oracle running the same schedule (records and
    allele strings)."""
    from pisces_amd import engine, synth
    seed, chunk = 33, 200
    cfg = _abi.default_config(call_mnvs=1, max_mnv_length=3, max_gap_between_mnv=1)
    n_amp = n_loci // synth.READ_LEN
    n_loci = n_amp * synth.READ_LEN
    ref = synth.reference_of(n_loci, seed, device="cuda")
    origin = synth.READ_LEN + 1
    recs, planted, n_reads = [], [], 0
    with engine.HipVariantCaller(cfg) as c:
        c.SetReference(ref)
        # the first 14 amplicons (2100 loci) block by block, against the oracle
        p = synth.make_pileup(14 * synth.READ_LEN, depth, seed=seed, device="cuda", first_locus=0, total_loci=n_loci, with_tuples=False)
        batch, pl = synth.mixed_reads(p, seed)
        planted += pl
        n_reads += batch.n_reads
        c.AddAlleleCounts(batch)
        head, head_alleles = [], []
        for up_to in (1000, 2000):
            r, a = c.CallWithAlleles(up_to, capacity=1 << 14)
            head.append(r)
            head_alleles += a
        head = np.concatenate(head)
        exp, exp_alleles, _ = orc.run_reads_blocks(batch, ref, 1, 2100 + origin, cfg)
        keep = exp["position"] <= 2000
        assert int(keep.sum()) == len(head) > 1800
        assert head_alleles == [x for x, k in zip(exp_alleles, keep) if k]
        assert_records_match(head, exp[keep])
        recs.append(head)
        got_alleles = list(head_alleles)
        for a0 in range(14, n_amp, chunk):
            na = min(chunk, n_amp - a0)
            p = synth.make_pileup(na * synth.READ_LEN, depth, seed=seed, device="cuda", first_locus=a0 * synth.READ_LEN, total_loci=n_loci,
                                  with_tuples=False)
            batch, pl = synth.mixed_reads(p, seed)
            planted += pl
            n_reads += batch.n_reads
            c.AddAlleleCounts(batch)
            r, a = c.CallWithAlleles(origin + a0 * synth.READ_LEN - 1, capacity=1 << 19)
            recs.append(r)
            got_alleles += a
            del p, batch
        r, a = c.CallWithAlleles(None, capacity=1 << 19)
        recs.append(r)
        got_alleles += a
        stats = c.Stats()
    got = np.concatenate(recs)
    assert stats["reads"] == n_reads and len(got_alleles) == len(got)
    assert (np.diff(got["position"]) >= 0).all()
    # ---- sixteen 1000-locus blocks anywhere in the run against the oracle: the last block, the blocks that hold a stretch boundary (where
    # one add_reads ended and the next began, and a flush cut the run), blocks late in the position range.  The oracle runs the reads of
    # the block and of its two neighbours through the flushes of the product's own schedule that fall there; the middle block is compared,
    # records and allele strings.
    ref_np = ref.cpu().numpy() if hasattr(ref, "cpu") else np.asarray(ref)
    ups_all = [origin + a0 * synth.READ_LEN - 1 for a0 in range(14, n_amp, chunk)]
    last_position = origin + n_loci - 1
    last_block = (last_position - 1) // 1000
    blocks = {last_block, 3}
    if len(ups_all) > 1:
        blocks |= {(ups_all[1] - 1) // 1000, (ups_all[len(ups_all) // 2] - 1) // 1000, (ups_all[-1] - 1) // 1000}
    rng = np.random.default_rng(11)
    while len(blocks) < min(16, last_block - 2):
        blocks.add(int(rng.integers(3, last_block)))
    for k in sorted(blocks):
        lo, hi = (k - 1) * 1000 + 1, min((k + 2) * 1000, last_position)
        a_lo, a_hi = max(0, (lo - origin) // synth.READ_LEN), min(n_amp, -(-(hi + 1 - origin) // synth.READ_LEN))
        p = synth.make_pileup((a_hi - a_lo) * synth.READ_LEN, depth, seed=seed, device="cuda", first_locus=a_lo * synth.READ_LEN, total_loci=n_loci,
                              with_tuples=False)
        batch, _ = synth.mixed_reads(p, seed)
        exp, exp_alleles, _ = orc.run_reads_schedule(batch, ref_np, lo, hi - lo + 1, cfg, [u for u in ups_all if lo <= u <= hi])
        sel = (exp["position"] > k * 1000) & (exp["position"] <= (k + 1) * 1000)
        g0, g1 = np.searchsorted(got["position"], [k * 1000 + 1, (k + 1) * 1000 + 1])
        assert g1 - g0 == int(sel.sum()) >= min(1000, last_position - k * 1000), (k, g0, g1, int(sel.sum()))
        assert_records_match(got[g0:g1], exp[sel])
        assert got_alleles[g0:g1] == [x for x, s_ in zip(exp_alleles, sel) if s_], k
        del p, batch
    cats = (got["info"] >> 4) & 7
    point = (cats == _abi.CAT_SNV) | (cats == _abi.CAT_REFERENCE)
    assert ((got["total_coverage"] + got["num_no_calls"])[point] == depth).all()
    assert len(np.unique(got["position"])) == n_loci   # every locus of the region is a candidate locus
    for kind, cat in (("M", _abi.CAT_MNV), ("D", _abi.CAT_DELETION), ("I", _abi.CAT_INSERTION)):
        want = np.array(sorted(pos for k, pos, _, _ in planted if k == kind))
        have = np.unique(got["position"][cats == cat])
        assert len(want) > 400 and np.isin(want, have).mean() >= 0.99, (kind, len(want), int(np.isin(want, have).sum()))


def test_streaming_surface_equals_device_resident_surface_at_size(torch_cuda):
    """The two surfaces of the boundary on the same pileup (20 000 loci x 500x, 66 700 reads): reads walked on the device, block by
    block through add_reads / flush as SmallVariantCaller drives them, must give exactly the records of one call_tiles launch over
    the pre-bucketed tuples (tile order inside the observation log is not defined, the counts are)."""
    from pisces_amd import engine, synth
    torch = torch_cuda
    p = synth.make_pileup(n_loci=20_000, depth=500, seed=77, device="cuda")
    cfg = _abi.default_config()
    with engine.HipVariantCaller(cfg) as caller:
        resident, _ = run_fused(torch, caller, p)
    p.base, p.qual = p.base.cpu(), p.qual.cpu()
    A = p.base.shape[0]
    streamed = []
    with engine.HipVariantCaller(cfg) as c:
        c.SetReference(p.ref.cpu().numpy())
        for a0 in range(0, A, 5):
            c.AddAlleleCounts(synth.reads_of(p, 5, first_amplicon=a0))
            streamed.append(c.Call(p.region_start + a0 * synth.READ_LEN - 1))
        streamed.append(c.Call(None))
        stats = c.Stats()
    streamed = np.concatenate(streamed)
    assert streamed.tobytes() == resident.tobytes()
    assert stats["reads"] == A * 500 and stats["reads_skipped"] == 0
    # the whole pileup in ONE add_reads call: 20 MB of bases + qualities, staged in slices by worker threads under the PCIe transfer
    # (stitched per-base directions ride along as a third bulk array)
    whole = synth.reads_of(p, A, first_amplicon=0)
    assert 2 * whole.n_bases >= (16 << 20)
    dirs = np.where(np.repeat(whole.flags & 1, np.diff(whole.seq_offset)) == 1, _abi.DIR_REVERSE, _abi.DIR_FORWARD).astype(np.uint8)
    with_dirs = _abi.ReadBatch.from_arrays(whole.position, whole.flags, whole.cigar_offset, whole.cigar_op, whole.cigar_len, whole.seq_offset,
                                           whole.bases, whole.quals, directions=dirs)
    for batch in (whole, with_dirs):
        with engine.HipVariantCaller(cfg) as c:
            c.SetReference(p.ref.cpu().numpy())
            c.AddAlleleCounts(batch)
            once = c.Call(None)
        assert once.tobytes() == resident.tobytes()
        # the same batch written by the caller into the library's pinned staging buffer (pisces_hip_stage_reads): sent as it lies
        with engine.HipVariantCaller(cfg) as c:
            c.SetReference(p.ref.cpu().numpy())
            c.AddAlleleCounts(c.StageReads(batch))
            once = c.Call(None)
        assert once.tobytes() == resident.tobytes()
    # staged and plain batches in turn, block by block
    streamed = []
    with engine.HipVariantCaller(cfg) as c:
        c.SetReference(p.ref.cpu().numpy())
        for k, a0 in enumerate(range(0, A, 5)):
            b = synth.reads_of(p, 5, first_amplicon=a0)
            c.AddAlleleCounts(c.StageReads(b) if k % 3 != 2 else b)
            streamed.append(c.Call(p.region_start + a0 * synth.READ_LEN - 1))
        streamed.append(c.Call(None))
    assert np.concatenate(streamed).tobytes() == resident.tobytes()


def test_collapser_on_open_ended_indels_matches_oracle(torch_cuda):
    """with the collapser on (the reference default) reads that end inside an insertion, start inside it, end in
    a deletion or start with one produce open-ended candidates, which VariantCollapser folds into the anchored candidate
    (VariantCollapser.cs:31-174).  The library collapses its insertion / deletion candidates on the host with frequencies from the
    device counts; SNV twins are the counts already.  Records, allele strings and TotalNumCalled against the oracle."""
    from pisces_amd import engine
    rng = np.random.default_rng(5)
    ref = bytes(rng.choice(list(b"ACGT"), 900).astype(np.uint8))
    P, INS = 300, b"ACGTTGCA"      # insertion after position P
    Q, DEL = 520, 6                # deletion of positions Q+1 .. Q+DEL
    reads = []

    def add(pos, ops, seq, i):
        q = np.where(rng.random(len(seq)) < 0.01, 12, 37).astype(np.uint8)
        reads.append({"pos": pos, "cigar": ops, "seq": seq.decode(), "quals": q.tolist(), "reverse": bool(i % 2)})

    for i in range(60):            # plain coverage over both sites
        add(P - 70, [("M", 150)], ref[P - 71: P + 79], i)
        add(Q - 70, [("M", 150)], ref[Q - 71: Q + 79], i)
    for i in range(40):            # the full insertion, anchored on both sides
        add(P - 50, [("M", 51), ("I", len(INS)), ("M", 60)], ref[P - 51: P] + INS + ref[P: P + 60], i)
    for i in range(16):            # reads that end inside the insertion (open on the right)
        k = 3 + i % 4
        add(P - 60, [("M", 61), ("I", k)], ref[P - 61: P] + INS[:k], i)
    for i in range(16):            # reads that start inside the insertion (open on the left)
        k = 3 + i % 4
        add(P + 1, [("I", k), ("M", 80)], INS[len(INS) - k:] + ref[P: P + 80], i)
    for i in range(40):            # the full deletion
        add(Q - 50, [("M", 51), ("D", DEL), ("M", 60)], ref[Q - 51: Q] + ref[Q + DEL: Q + DEL + 60], i)
    for i in range(12):            # reads that end in the deletion / start with it
        add(Q - 60, [("M", 61), ("D", DEL)], ref[Q - 61: Q], i)
        add(Q + DEL + 1, [("D", DEL), ("M", 70)], ref[Q + DEL: Q + DEL + 70], i)
    batch = _abi.ReadBatch(reads)
    refa = np.frombuffer(ref, dtype=np.uint8)
    out = {}
    for collapse in (0, 1):
        cfg = _abi.default_config(collapse=collapse)
        exp, exp_alleles, _, exp_called = orc.run_reads_full(batch, refa, 1, len(ref), cfg)
        with engine.HipVariantCaller(cfg) as c:
            c.SetReference(refa)
            c.AddAlleleCounts(batch)
            got, got_alleles = c.CallWithAlleles()
            stats = c.Stats()
        assert_records_match(got, exp)
        assert got_alleles == exp_alleles
        assert stats["TotalNumCalled"] == exp_called
        out[collapse] = (got, got_alleles, stats)
    # collapsing happened and changed the call set: fewer rows, more support on the anchored insertion
    assert out[1][2]["TotalNumCollapsed"] > 0 and out[0][2]["TotalNumCollapsed"] == 0
    full = ("%c" % ref[P - 1], ("%c" % ref[P - 1]) + INS.decode())
    sup = {k: [int(r["allele_support"]) for r, a in zip(out[k][0], out[k][1]) if a == full and r["position"] == P] for k in (0, 1)}
    assert sup[0] and sup[1] and sup[1][0] > sup[0][0]
    assert len(out[1][0]) <= len(out[0][0])


def test_known_variants_steer_the_collapser_as_in_the_reference(torch_cuda):
    """VariantCollapser.cs:16-24, 178-190, 216-218: a candidate that equals a known (prior) variant of the chromosome is anchored on both
    sides and comes first among the potential matches of an open-ended candidate.  Two insertions behind one position, a long one and a
    shorter one with the same first bases; reads that end inside the insertion match both and join the LONGER one — unless the shorter one
    is known (pisces_hip_set_known_variants; the oracle: orc_set_known_variants).  Records, allele strings and totals against the oracle,
    with and without."""
    from pisces_amd import engine
    rng = np.random.default_rng(15)
    ref = bytes(rng.choice(list(b"ACGT"), 700).astype(np.uint8))
    P, LONG, SHORT = 300, b"ACGTTGCA", b"ACGTA"
    reads = []

    def add(pos, ops, seq, i):
        reads.append({"pos": pos, "cigar": ops, "seq": seq.decode(), "quals": [37] * len(seq), "reverse": bool(i % 2)})

    for i in range(80):
        add(P - 70, [("M", 150)], ref[P - 71: P + 79], i)
    for i in range(30):
        add(P - 50, [("M", 51), ("I", len(LONG)), ("M", 60)], ref[P - 51: P] + LONG + ref[P: P + 60], i)
    for i in range(20):
        add(P - 50, [("M", 51), ("I", len(SHORT)), ("M", 60)], ref[P - 51: P] + SHORT + ref[P: P + 60], i)
    for i in range(24):            # reads that end inside the insertion after ACG / ACGT: both insertions start that way
        k = 3 + i % 2
        add(P - 60, [("M", 61), ("I", k)], ref[P - 61: P] + LONG[:k], i)
    batch = _abi.ReadBatch(reads)
    refa = np.frombuffer(ref, dtype=np.uint8)
    anchor = "%c" % ref[P - 1]
    short_allele, long_allele = (anchor, anchor + SHORT.decode()), (anchor, anchor + LONG.decode())
    cfg = _abi.default_config(collapse=1)
    sup = {}
    for known in ([], [(P, short_allele[0], short_allele[1])]):
        keep = orc.set_known_variants([(p, _abi.CAT_INSERTION, r, a) for p, r, a in known])
        try:
            exp, exp_alleles, _, exp_called = orc.run_reads_full(batch, refa, 1, len(ref), cfg)
        finally:
            orc.set_known_variants([])
        del keep
        with engine.HipVariantCaller(cfg) as c:
            c.SetReference(refa)
            c.SetKnownVariants(known)
            c.AddAlleleCounts(batch)
            got, got_alleles = c.CallWithAlleles()
            stats = c.Stats()
        assert_records_match(got, exp)
        assert got_alleles == exp_alleles
        assert stats["TotalNumCalled"] == exp_called and stats["TotalNumCollapsed"] == 2   # (the two distinct open-ended candidates: ACG, ACGT)
        sup[bool(known)] = {a: int(r["allele_support"]) for r, a in zip(got, got_alleles) if r["position"] == P and a in (short_allele, long_allele)}
    assert sup[False] == {long_allele: 30 + 24, short_allele: 20}     # the open-ended reads join the longer insertion ...
    assert sup[True] == {long_allele: 30, short_allele: 20 + 24}      # ... or the known one


def test_mnvs_can_be_left_out_of_the_collapser_as_in_the_reference(torch_cuda):
    """PiscesApplicationOptions.ExcludeMNVsFromCollapsing (VariantCollapser.cs:33; VariantCollapserTests.cs:383-425 Collapse_IgnoreMNVs):
    reads that carry a two-base MNV at P, P + 1, and reads that END on P with the MNV's first base — an SNV candidate that is open on the
    right, which the collapser merges into the MNV (same position, the MNV's bases start with it) unless MNVs are left out of its
    targets (pisces_hip_set_exclude_mnvs_from_collapsing; the oracle: orc_set_exclude_mnvs_from_collapsing).  Records, allele strings
    and totals against the oracle, both ways."""
    from pisces_amd import engine
    rng = np.random.default_rng(23)
    ref = bytearray(rng.choice(list(b"ACGT"), 700).astype(np.uint8))
    P = 300
    ref[P - 1:P + 1] = b"AC"
    ref = bytes(ref)
    reads = []

    def add(pos, seq, i):
        reads.append({"pos": pos, "cigar": [("M", len(seq))], "seq": seq.decode(), "quals": [37] * len(seq), "reverse": bool(i % 2)})

    for i in range(60):
        add(P - 70, ref[P - 71: P + 79], i)
    for i in range(30):            # the MNV AC > GT inside the read
        add(P - 50, ref[P - 51: P - 1] + b"GT" + ref[P + 1: P + 70], i)
    for i in range(20):            # reads that end on P with G: an SNV A > G that is open on the right
        add(P - 80, ref[P - 81: P - 1] + b"G", i)
    batch = _abi.ReadBatch(reads)
    refa = np.frombuffer(ref, dtype=np.uint8)
    cfg = _abi.default_config(call_mnvs=1, max_mnv_length=3, max_gap_between_mnv=1, collapse=1)
    seen = {}
    for exclude in (False, True):
        orc.set_exclude_mnvs_from_collapsing(exclude)
        try:
            exp, exp_alleles, _, exp_called = orc.run_reads_full(batch, refa, 1, len(ref), cfg)
        finally:
            orc.set_exclude_mnvs_from_collapsing(False)
        with engine.HipVariantCaller(cfg) as c:
            c.SetReference(refa)
            c.SetExcludeMNVsFromCollapsing(exclude)
            c.AddAlleleCounts(batch)
            got, got_alleles = c.CallWithAlleles()
            stats = c.Stats()
        assert_records_match(got, exp)
        assert got_alleles == exp_alleles
        assert stats["TotalNumCalled"] == exp_called
        seen[exclude] = ({a: int(r["allele_support"]) for r, a in zip(got, got_alleles) if r["position"] == P and a != ("A", "A")}, stats["TotalNumCollapsed"])
    assert seen[False] == ({("AC", "GT"): 30 + 20}, 1)                       # the open-ended SNV joins the MNV ...
    assert seen[True] == ({("AC", "GT"): 30, ("A", "G"): 20}, 0)             # ... or stays a call of its own


def test_streaming_mix_of_snvs_and_indels_across_blocks_matches_oracle(torch_cuda):
    """BASELINE config 3 / 4 in the small (without MNV calling): 12 000 loci x 120x in 80 amplicons over 13 blocks, sequencing errors,
    planted SNVs, and at ~every 1000th locus a deletion (1-10 bp) or an insertion (1-6 bp) in a third of the reads — some of them
    next to block edges.  Fed block by block through add_reads / flush (device read walk, host finder + collapser, device
    spanning calls, block hold rule) and compared with the oracle over the whole region: records, allele strings, totals."""
    from pisces_amd import engine
    rng = np.random.default_rng(2026)
    n_amp, depth, L = 80, 120, 150
    ref = bytes(rng.choice(list(b"ACGT"), n_amp * L + 400).astype(np.uint8))
    start0 = 101
    reads, batches = [], []
    for a in range(n_amp):
        start = start0 + a * L
        plan = []
        if a % 7 == 3:
            plan.append((int(rng.integers(20, 120)), "D", int(rng.integers(1, 11)), 0.35))
        if a % 7 == 5:
            plan.append((int(rng.integers(20, 120)), "I", int(rng.integers(1, 7)), 0.35))
        if a % 13 == 6:
            plan.append((146, "D", 6, 0.4))          # reaches into the next amplicon (and, for some, the next block)
        amp = _indel_reads(rng, ref, start, depth, plan, read_len=L, p_lowq=0.03)
        # planted SNV and sequencing errors on the M-only part of the reads
        snv_off, snv_alt = int(rng.integers(5, 15)), "ACGT"[int(rng.integers(0, 4))]
        for r in amp:
            s = list(r["seq"])
            if r["cigar"][0][0] == "M" and r["cigar"][0][1] > snv_off + 1 and rng.random() < 0.2:
                s[snv_off] = snv_alt
            for j in np.nonzero(rng.random(len(s)) < 0.002)[0]:
                s[j] = "ACGT"[int(rng.integers(0, 4))]
            r["seq"] = "".join(s)
        batches.append((start, amp))
        reads += amp
    refa = np.frombuffer(ref, dtype=np.uint8)
    cfg = _abi.default_config()
    exp, exp_alleles, _, exp_called = orc.run_reads_full(_abi.ReadBatch(reads), refa, 1, len(ref), cfg)
    got, got_alleles = [], []
    with engine.HipVariantCaller(cfg) as c:
        c.SetReference(refa)
        for start, amp in batches:
            c.AddAlleleCounts(_abi.ReadBatch(amp))
            g, ga = c.CallWithAlleles(start - 1)
            got.append(g); got_alleles += ga
        g, ga = c.CallWithAlleles(None)
        got.append(g); got_alleles += ga
        stats = c.Stats()
    got = np.concatenate(got)
    assert_records_match(got, exp)
    assert got_alleles == exp_alleles
    assert stats["TotalNumCalled"] == exp_called and stats["reads"] == len(reads)
    cats = [int(_abi.info_category(i)) for i in got["info"]]
    assert cats.count(_abi.CAT_DELETION) >= 10 and cats.count(_abi.CAT_INSERTION) >= 5 and cats.count(_abi.CAT_SNV) >= 40
    assert (np.diff(got["position"]) >= 0).all()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["bam_chr19", "bam_chr17_again", "bam_chr17_int", "bam_chr17_vcf", "bam_phix", "bam_edge_ins", "bam_edge_del"])
def test_reference_bams_through_the_library_give_the_vcf_rows_pisces_wrote(torch_cuda, name):
    """End to end on the device: the reads of the reference's own test BAMs through the streaming surface (device read walk, finder,
    collapser, call kernels) and pisces_hip_format_vcf must reproduce the VCF body lines Pisces wrote for them byte for byte, and
    the records must equal the oracle's (tests/bam_fixtures.py, tests/golden/extract_bam_fixture.py)."""
    from pisces_amd import engine
    from tests import bam_fixtures
    case = bam_fixtures.CASES[name]
    z, batch = bam_fixtures.load(name)
    off = int(z["offset"])
    cfg = _abi.default_config(**case["cfg"])
    intervals = [(a - off, b - off) for a, b in case["intervals"]] if case["intervals"] else None
    with engine.HipVariantCaller(cfg) as c:
        c.SetReference(z["ref"])
        if intervals:
            c.SetIntervals(intervals)
        c.AddAlleleCounts(batch)
        got, got_alleles = c.CallWithAlleles()
    exp, exp_alleles = [], []
    for start, loci in ([(a, b - a + 1) for a, b in intervals] if intervals else [(1, len(z["ref"]))]):
        r, a, _, _ = orc.run_reads_full(batch, z["ref"], start, loci, cfg)
        exp.append(r)
        exp_alleles += a
    assert_records_match(got, np.concatenate(exp))
    assert got_alleles == exp_alleles
    got = got.copy()
    got["position"] += off
    text = engine.format_vcf(case["chrom"], got, alleles=got_alleles, noise_level_from_records=1, **case["vcf"])
    bam_fixtures.check_lines(case, text.rstrip("\n").split("\n") if text else [], bam_fixtures.expected_lines(name, z))


def _mnv_reads(rng, ref, n_reads, read_len=100, region=(50, 1900), snv_rate=0.004):
    """Reads with planted MNVs (some gapped), SNVs next to them, random errors that make failing MNV candidates, low-quality bases that
    open candidates up, and a few indels."""
    L = len(ref)
    planted = []   # (position, alt string over consecutive reference positions, fraction)
    p = region[0] + 20
    while p < region[1] - 40:
        kind = int(rng.integers(0, 5))
        if kind == 0:
            alt = "".join(rng.choice([b for b in "ACGT" if b != chr(ref[p - 1 + i])]) for i in range(2))
        elif kind == 1:
            alt = "".join(rng.choice([b for b in "ACGT" if b != chr(ref[p - 1 + i])]) for i in range(3))
        elif kind == 2:   # gapped: mismatch, reference, mismatch
            alt = rng.choice([b for b in "ACGT" if b != chr(ref[p - 1])]) + chr(ref[p]) + rng.choice([b for b in "ACGT" if b != chr(ref[p + 1])])
        else:
            alt = str(rng.choice([b for b in "ACGT" if b != chr(ref[p - 1])]))
        planted.append((p, alt, float(rng.choice([0.004, 0.03, 0.2, 0.6]))))
        p += int(rng.integers(7, 40))
    reads = []
    for i in range(n_reads):
        start = int(rng.integers(region[0], region[1] - read_len))
        seq = bytearray(ref[start - 1: start - 1 + read_len])
        for (pp, alt, frac) in planted:
            if start <= pp and pp + len(alt) <= start + read_len and rng.random() < frac:
                seq[pp - start: pp - start + len(alt)] = alt.encode()
            elif start <= pp < start + read_len and rng.random() < frac * 0.3:   # partial: only the first base
                seq[pp - start] = ord(alt[0])
        for k in range(read_len):
            if rng.random() < snv_rate:
                seq[k] = int(rng.choice(list(b"ACGT")))
        quals = np.where(rng.random(read_len) < 0.03, 12, 37).astype(np.uint8)
        ops = [("M", read_len)]
        s = bytes(seq).decode()
        if rng.random() < 0.03:   # a deletion or an insertion in the middle
            k = int(rng.integers(20, read_len - 20))
            if rng.random() < 0.5:
                ops = [("M", k), ("D", 3), ("M", read_len - k)]
                s = s[:k] + bytes(ref[start - 1 + k + 3: start - 1 + read_len + 3]).decode()
            else:
                ops = [("M", k), ("I", 2), ("M", read_len - k - 2)]
                s = s[:k] + "GA" + s[k: read_len - 2]
        reads.append({"pos": start, "cigar": ops, "seq": s, "quals": quals.tolist(), "reverse": bool(i % 2)})
    return reads


@pytest.mark.gpu
@pytest.mark.parametrize("collapse", [0, 1])
@pytest.mark.parametrize("mnv", [(3, 1), (6, 2)])
def test_mnv_calling_matches_oracle(torch_cuda, collapse, mnv):
    """SURVEY section 8 rows a4 / a10 / f2 with -callmnvs: SNV / MNV candidates from the read walk, MNV candidates processed first,
    failed MNVs reallocated (MnvReallocator) to sub-MNVs / SNVs / Reference alleles, gapped-MNV reference counts, every callable allele
    processed again.  Records, allele strings and TotalNumCalled against the oracle, whole region in one block grid flush."""
    from pisces_amd import engine
    rng = np.random.default_rng(100 + 10 * collapse + mnv[0])
    ref = bytes(rng.choice(list(b"ACGT"), 2000).astype(np.uint8))
    reads = _mnv_reads(rng, ref, 3000)
    batch = _abi.ReadBatch(reads)
    refa = np.frombuffer(ref, dtype=np.uint8)
    cfg = _abi.default_config(call_mnvs=1, max_mnv_length=mnv[0], max_gap_between_mnv=mnv[1], collapse=collapse, block_size=2000)
    exp, exp_alleles, _, exp_called = orc.run_reads_full(batch, refa, 1, len(ref), cfg)
    with engine.HipVariantCaller(cfg) as c:
        c.SetReference(refa)
        c.AddAlleleCounts(batch)
        got, got_alleles = c.CallWithAlleles()
        stats = c.Stats()
    cats = (exp["info"] >> 4) & 7
    assert (cats == _abi.CAT_MNV).sum() >= 5 and (cats == _abi.CAT_SNV).sum() >= 10
    assert got_alleles == exp_alleles
    assert_records_match(got, exp)
    assert stats["TotalNumCalled"] == exp_called


@pytest.mark.gpu
def test_small_s1_bam_mnvs_through_the_library(torch_cuda):
    """BasicMnvTesting (SomaticVariantCallerFunctionalTests.cs:381-424) on the device: small_S1.bam on the test's mock chr1 with MNV calling
    on gives exactly the three expected variants, and the records equal the oracle's."""
    from pisces_amd import engine
    from tests import bam_fixtures
    case = bam_fixtures.CASES["bam_small_s1"]
    z, batch = bam_fixtures.load("bam_small_s1")
    cfg = _abi.default_config(**case["cfg"])
    exp, exp_alleles, _, exp_called = orc.run_reads_full(batch, z["ref"], 1, len(z["ref"]), cfg)
    with engine.HipVariantCaller(cfg) as c:
        c.SetReference(z["ref"])
        c.AddAlleleCounts(batch)
        got, got_alleles = c.CallWithAlleles()
        stats = c.Stats()
    assert got_alleles == exp_alleles
    assert_records_match(got, exp)
    assert stats["TotalNumCalled"] == exp_called
    text = engine.format_vcf(case["chrom"], got, alleles=got_alleles)
    bam_fixtures.check_lines(case, text.rstrip("\n").split("\n"), bam_fixtures.expected_lines("bam_small_s1", z))


@pytest.mark.gpu
def test_mnv_calling_over_the_block_schedule(torch_cuda):
    """MNV calling through the streaming block schedule (1000-locus blocks, flush per block as SmallVariantCaller does): blocks are
    called one by one, each with its own reallocation, and the concatenation equals the oracle's single-window run as long as no MNV
    candidate straddles a block edge (reads are kept 6 loci away from the edges here; straddling is MnvReallocator's block logic,
    pinned on the CPU by the reference's BlockStraddling cases)."""
    from pisces_amd import engine
    rng = np.random.default_rng(77)
    ref = bytes(rng.choice(list(b"ACGT"), 3000).astype(np.uint8))
    reads = []
    for lo in (1, 1001, 2001):   # reads stay inside their block
        reads += _mnv_reads(rng, ref, 1200, read_len=90, region=(lo + 6, lo + 993))
    reads.sort(key=lambda r: r["pos"])
    refa = np.frombuffer(ref, dtype=np.uint8)
    cfg = _abi.default_config(call_mnvs=1)
    exp, exp_alleles, _, exp_called = orc.run_reads_full(_abi.ReadBatch(reads), refa, 1, len(ref), cfg)
    got, got_alleles = [], []
    with engine.HipVariantCaller(cfg) as c:
        c.SetReference(refa)
        for i in range(0, len(reads), 400):
            chunk = reads[i: i + 400]
            c.AddAlleleCounts(_abi.ReadBatch(chunk))
            r, a = c.CallWithAlleles(upToPosition=chunk[-1]["pos"] - 1)
            got.append(r)
            got_alleles += a
        r, a = c.CallWithAlleles()
        got.append(r)
        got_alleles += a
        stats = c.Stats()
    got = np.concatenate(got)
    assert sum(len(x) > 0 for x in [got]) and len(got) == len(exp)
    assert got_alleles == exp_alleles
    assert_records_match(got, exp)
    assert stats["TotalNumCalled"] == exp_called


@pytest.mark.gpu
@pytest.mark.parametrize("call_mnvs", [0, 1])
def test_window_noise_model_matches_oracle(torch_cuda, call_mnvs):
    """SURVEY section 8 row a3, NoiseModel.Window: the base-quality sums are accumulated on the device next to the counts (cell by cell,
    fixed-point integer atomics) and every allele's q-score uses (int)PtoQ(SumOfBaseQuality / TotalCoverage): SNVs, Reference alleles,
    insertions / deletions (start + end point sums) and, with MNV calling on, MNVs.  Mixed base qualities, so no locus sits on an integer
    edge of PtoQ (there the order of the reference's FP64 additions, which no parallel sum reproduces, could decide); records against the
    oracle."""
    from pisces_amd import engine
    rng = np.random.default_rng(300 + call_mnvs)
    ref = bytes(rng.choice(list(b"ACGT"), 1500).astype(np.uint8))
    reads = _mnv_reads(rng, ref, 2500, region=(40, 1450))
    for i in range(160):   # a deletion of 701..703 and an insertion after 900, each in 80 reads
        if i % 2:
            reads.append({"pos": 660, "cigar": [("M", 41), ("D", 3), ("M", 50)], "seq": (ref[659:700] + ref[703:753]).decode(), "reverse": bool(i & 2)})
        else:
            reads.append({"pos": 860, "cigar": [("M", 41), ("I", 4), ("M", 50)], "seq": (ref[859:900] + b"TTGA" + ref[900:950]).decode(), "reverse": bool(i & 2)})
    for r in reads:   # mixed qualities: 12 (below the threshold), 23, 30, 37, 41
        r["quals"] = rng.choice([12, 23, 30, 37, 41], len(r["seq"]), p=[.04, .2, .2, .4, .16]).astype(np.uint8).tolist()
    batch = _abi.ReadBatch(reads)
    refa = np.frombuffer(ref, dtype=np.uint8)
    cfg = _abi.default_config(noise_model=1, call_mnvs=call_mnvs, block_size=2000, max_variant_qscore=200)
    exp, exp_alleles, _, exp_called = orc.run_reads_full(batch, refa, 1, len(ref), cfg)
    flat, _, _, _ = orc.run_reads_full(batch, refa, 1, len(ref), _abi.default_config(call_mnvs=call_mnvs, block_size=2000, max_variant_qscore=200))
    assert len(flat) != len(exp) or (flat["variant_qscore"] != exp["variant_qscore"]).any()   # the model matters on this input
    with engine.HipVariantCaller(cfg) as c:
        c.SetReference(refa)
        c.AddAlleleCounts(batch)
        got, got_alleles = c.CallWithAlleles()
        stats = c.Stats()
    cats = (exp["info"] >> 4) & 7
    assert (cats == _abi.CAT_SNV).sum() >= 10 and ((cats == _abi.CAT_DELETION) | (cats == _abi.CAT_INSERTION)).sum() >= 1
    assert got_alleles == exp_alleles
    assert_records_match(got, exp)
    assert stats["TotalNumCalled"] == exp_called


@pytest.mark.gpu
def _germline_reads(seed, with_deletion=True):
    """Germline-like reads: het and hom SNVs, a het deletion (optional), a tri-allelic site, a 1/2 site, sub-threshold alleles, a shallow
    stretch.  Returns (reference bytes, reads in position order)."""
    rng = np.random.default_rng(seed)
    ref = bytearray(rng.choice(list(b"ACGT"), 1600).astype(np.uint8))
    def other(p, k=0):
        return [b for b in b"ACGT" if b != ref[p - 1]][k]
    planted = {}   # position -> list of (alt byte, cumulative fraction)
    p = 60
    kinds = ["het", "hom", "quarter", "tri", "alt12", "low"]
    i = 0
    while p < 1500:
        k = kinds[i % len(kinds)]
        planted[p] = {"het": [(other(p), 0.5)], "hom": [(other(p), 0.97)], "quarter": [(other(p), 0.27)],
                      "tri": [(other(p), 0.3), (other(p, 1), 0.55), (other(p, 2), 0.8)], "alt12": [(other(p), 0.48), (other(p, 1), 0.97)],
                      "low": [(other(p), 0.1)]}[k]
        p += int(rng.integers(15, 40))
        i += 1
    reads = []
    for n in range(2400):
        shallow = n >= 2340                      # a stretch covered by only 60 reads: below MinimumCoverage on parts of it
        start = int(rng.integers(1390, 1480)) if shallow else int(rng.integers(30, 1300))
        L = 100
        seq = bytearray(ref[start - 1: start - 1 + L])
        for pos, alts in planted.items():
            if start <= pos < start + L:
                u = rng.random()
                for alt, cum in alts:
                    if u < cum:
                        seq[pos - start] = alt
                        break
        for k in range(L):
            if rng.random() < 0.002:
                seq[k] = int(rng.choice(list(b"ACGT")))
        ops, s = [("M", L)], bytes(seq).decode()
        if with_deletion and start <= 720 and start + L > 745 and rng.random() < 0.45:   # a het deletion of 731..733
            k = 730 - start + 1
            ops = [("M", k), ("D", 3), ("M", L - k)]
            s = s[:k] + bytes(ref[start - 1 + k + 3: start - 1 + L + 3]).decode()
        reads.append({"pos": start, "cigar": ops, "seq": s, "quals": np.where(rng.random(L) < 0.02, 12, 37).astype(np.uint8).tolist(),
                      "reverse": bool(n % 2)})
    reads.sort(key=lambda r: r["pos"])
    return ref, reads


@pytest.mark.parametrize("sb_model,ploidy", [(1, 1), (2, 1), (1, 2)])   # Extended / Diploid strand bias; diploid, diploid, haploid genotyper
def test_diploid_genotyping_matches_oracle(torch_cuda, sb_model, ploidy):
    """PloidyModel.DiploidByThresholding (one genotype per locus from the variant frequencies, alleles beyond
    the ploidy pruned, diploid genotype q-scores, MultiAllelicSite / LowGQ filters, phase set index), PloidyModel.Haploid (hemizygous
    calls, its own q-score) and the Diploid strand-bias model,
    on germline-like reads: het and hom SNVs, a het deletion, a tri-allelic site, a 1/2 site, sub-threshold alleles, a shallow stretch.
    Records, allele strings and TotalNumCalled against the oracle."""
    from pisces_amd import engine
    ref, reads = _germline_reads(500 + sb_model)
    batch = _abi.ReadBatch(reads)
    refa = np.frombuffer(bytes(ref), dtype=np.uint8)
    cfg = _abi.default_config(ploidy=ploidy, strand_bias_model=sb_model, min_frequency=0.2, variant_freq_filter=0.2, low_gq_filter=30,
                              max_genotype_qscore=1000, block_size=2000)
    exp, exp_alleles, _, exp_called = orc.run_reads_full(batch, refa, 1, len(ref), cfg)
    gts = set((exp["info"] & 15).tolist())
    if ploidy == 1:
        assert {0, 2, 3, 4}.issubset(gts) and (6 in gts or 1 in gts), gts       # 1/2, 0/1, 1/1, 0/0 and a multi-allelic no-call
        assert ((exp["filter_bits"] >> 8) & 1).any() and ((exp["filter_bits"] >> 14) == 2).any()
    else:
        assert gts == {9, 10, 11}, gts                                          # HemizygousRef / Alt / NoCall (HaploidGenotyper)
    assert ploidy == 2 or ((((exp["info"] >> 4) & 7) == _abi.CAT_DELETION)).any()   # (a het deletion is no hemizygous call)
    with engine.HipVariantCaller(cfg) as c:
        c.SetReference(refa)
        c.AddAlleleCounts(batch)
        got, got_alleles = c.CallWithAlleles()
        stats = c.Stats()
    assert got_alleles == exp_alleles
    assert_records_match(got, exp)
    assert stats["TotalNumCalled"] == exp_called


@pytest.mark.gpu
@pytest.mark.parametrize("ploidy", [1, 2], ids=["diploid", "haploid"])
@pytest.mark.parametrize("gvcf", [1, 0], ids=["gvcf", "variants only"])
def test_genotypes_made_on_the_device_equal_the_host_pass_and_the_oracle(torch_cuda, ploidy, gvcf):
    """genotype_loci_kernel (lane = locus over the tile kernels' record slots: DiploidThresholdingGenotyper.cs:54-141, HaploidGenotyper.cs:36-83,
    the genotype q-scores, LowGQ / MultiAllelicSite, alleles beyond the ploidy gone before the compaction) against the host pass of the
    flush (PISCES_HIP_DEVICE_GENOTYPER=0: diploid.cpp over the downloaded rows — the same genotype_core.h) and the oracle: reads without
    insertions / deletions, so that no candidate row joins the tile kernels' rows; block by block and in one flush; and through the
    device-resident surface (pisces_hip_call_tiles + pisces_hip_compact_records on a diploid / haploid handle, which used to refuse)."""
    from pisces_amd import engine, synth
    torch = torch_cuda
    ref, reads = _germline_reads(640 + ploidy, with_deletion=False)
    batch = _abi.ReadBatch(reads)
    refa = np.frombuffer(bytes(ref), dtype=np.uint8)
    cfg = _abi.default_config(ploidy=ploidy, min_frequency=0.2, variant_freq_filter=0.2, low_gq_filter=30, max_genotype_qscore=1000,
                              include_reference_calls=gvcf)
    exp, exp_alleles, _, exp_called = orc.run_reads_full(batch, refa, 1, len(ref), cfg)
    assert set(((exp["info"] >> 4) & 7).tolist()) <= {_abi.CAT_SNV, _abi.CAT_REFERENCE} and len(exp) > (1000 if gvcf else 5)
    if ploidy == 1:
        assert ((exp["filter_bits"] >> 8) & 1).any() and ((exp["filter_bits"] >> 14) == 2).any()   # a multi-allelic site, a second phase-set index
    out = {}
    for on_device in (1, 0):
        with env(PISCES_HIP_DEVICE_GENOTYPER=on_device):
            with engine.HipVariantCaller(cfg) as c:
                c.SetReference(refa)
                c.AddAlleleCounts(batch)
                rows = [c.Call(up, capacity=1 << 14) for up in (1000, None)]
                out[on_device] = (np.concatenate(rows), c.Stats())
    assert out[1][0].tobytes() == out[0][0].tobytes() and out[1][1] == out[0][1]
    assert_records_match(out[1][0], exp)
    assert out[1][1]["TotalNumCalled"] == exp_called
    # the device-resident surface on a pileup of its own: slots genotyped in place, pruned rows never compacted, totals taken back
    p = synth.make_pileup(n_loci=1500, depth=120, seed=77 + ploidy, device="cuda", snv_every=4, snv_offset=1, vaf_range=(0.05, 0.99))
    with engine.HipVariantCaller(cfg) as c:
        got, tr = run_fused(torch, c, p, compact=True)
        totals = c.device_totals()
    pos, tup = synth.observations_of(p)
    want, nloci = orc.run_observations(pos, tup, p.ref.cpu().numpy(), p.region_start, p.n_loci, cfg)
    assert_records_match(got, want)
    assert totals["records"] == len(want) and totals["candidate_loci"] == nloci == int(tr["n_candidate_loci"].sum())
    assert len(want) > 20


@pytest.mark.gpu
def test_mnvs_that_straddle_a_block_edge(torch_cuda):
    """MnvReallocator's block logic on the device path: failed MNV candidates that reach past the last cleared block are peeled (the part
    in the next block returns to the state as a candidate of that block, AlleleCaller.cs:91-93), callable ones are called from the
    block they start in once the schedule has moved past their end (MaxAlleleEndpoint hold).  Low-frequency MNVs of length 3 planted across the
    1000|1001 and 2000|2001 edges next to SNVs that only become callable with the peeled part; the oracle runs the same block schedule (orc_run_reads_blocks)."""
    from pisces_amd import engine
    rng = np.random.default_rng(909)
    ref = bytearray(rng.choice(list(b"ACGT"), 3000).astype(np.uint8))
    def mut(p, n):
        return "".join(chr([b for b in b"ACGT" if b != ref[p - 1 + i]][(i + p) % 3]) for i in range(n))
    m999, m1998 = mut(999, 3), mut(1999, 3)
    # fractions of the covering reads: the 1.2 % MNVs fail on their q-score, and so would the 1.2 % SNVs under their last base alone;
    # together (the reallocated part crosses the block edge as a leftover candidate) they are called
    planted = [(999, m999, 0.012), (1001, m999[2], 0.012), (1999, m1998, 0.012), (2001, m1998[2], 0.012), (500, mut(500, 3), 0.25),
               (1040, mut(1040, 2), 0.2)]
    reads = []
    for n in range(4000):
        start = int(rng.integers(900, 1080)) if n % 3 == 0 else int(rng.integers(1900, 2080)) if n % 3 == 1 else int(rng.integers(430, 560))
        L = 100
        seq = bytearray(ref[start - 1: start - 1 + L])
        for (p, alt, frac) in planted:
            if start <= p and p + len(alt) <= start + L and rng.random() < frac:
                seq[p - start: p - start + len(alt)] = alt.encode()
                break
        reads.append({"pos": start, "cigar": [("M", L)], "seq": bytes(seq).decode(), "quals": [37] * L, "reverse": bool(n % 2)})
    reads.sort(key=lambda r: r["pos"])
    batch = _abi.ReadBatch(reads)
    refa = np.frombuffer(bytes(ref), dtype=np.uint8)
    cfg = _abi.default_config(call_mnvs=1, collapse=0)
    exp, exp_alleles, exp_called = orc.run_reads_blocks(batch, refa, 1, len(ref), cfg)
    called = {(int(r["position"]), a) for r, a in zip(exp, exp_alleles) if a[0] != a[1]}
    assert (1001, (chr(ref[1000]), m999[2])) in called and (2001, (chr(ref[2000]), m1998[2])) in called   # SNVs that need the leftovers
    with engine.HipVariantCaller(cfg) as c:
        c.SetReference(refa)
        c.AddAlleleCounts(batch)
        got, got_alleles = [], []
        for up_to in (1500, 2500, None):
            r, a = c.CallWithAlleles(upToPosition=up_to)
            got.append(r)
            got_alleles += a
        stats = c.Stats()
    got = np.concatenate(got)
    assert got_alleles == exp_alleles
    assert_records_match(got, exp)
    assert stats["TotalNumCalled"] == exp_called


@pytest.mark.gpu
def test_mnv_mode_on_an_snv_only_pileup_equals_the_count_derived_calls(torch_cuda):
    """With -callmnvs on an SNV-only amplicon pileup (BASELINE config 2 style, 12 000 loci x 300x) the candidates come from the read walk
    instead of the allele counts, go through MNV-first processing / reallocation of the (failing) error MNVs and the candidate kernel:
    the result must equal the oracle's, and the called variants those of the default mode."""
    from pisces_amd import engine, synth
    p = synth.make_pileup(12000, 300, seed=11)
    ref = p.ref.cpu().numpy()
    batch = synth.reads_of(p)
    out = {}
    for mnv in (0, 1):
        cfg = _abi.default_config(call_mnvs=mnv)
        exp, exp_alleles, _, exp_called = orc.run_reads_full(batch, ref, p.region_start, p.n_loci, cfg)
        with engine.HipVariantCaller(cfg) as c:
            c.SetReference(ref)
            c.AddAlleleCounts(batch)
            got, got_alleles = c.CallWithAlleles()
            stats = c.Stats()
        assert got_alleles == exp_alleles
        assert_records_match(got, exp)
        assert stats["TotalNumCalled"] == exp_called
        out[mnv] = [(int(r["position"]), a) for r, a in zip(got, got_alleles) if a[0] != a[1]]
    assert len(out[0]) >= 100 and set(out[0]) <= set(out[1]) | set(out[0])


@pytest.mark.gpu
def test_batched_launches_equal_single_launches(torch_cuda):
    """pisces_hip_call_tiles_batched (independent batches spread over the handle's HIP streams) gives every batch exactly the records of its
    own pisces_hip_call_tiles launch, whatever runs beside it; argument errors come back before anything is launched."""
    import torch
    from pisces_amd import engine, synth
    dev = torch.device("cuda", 0)
    ps = [synth.make_pileup(3000 + 640 * i, 60 + 25 * i, seed=40 + i, device=dev) for i in range(5)]
    with engine.HipVariantCaller(_abi.default_config()) as c:
        outs, single = [], []
        for p in ps:
            cap = p.n_tiles * _abi.SLOTS_PER_TILE
            rec = torch.zeros(cap * 64, dtype=torch.uint8, device=dev)
            tr = torch.zeros(p.n_tiles * _abi.TILE_RESULT_DTYPE.itemsize, dtype=torch.uint8, device=dev)
            torch.cuda.synchronize()   # (torch fills on its own stream, the library launches on the handle's: the fills must have landed)
            c.call_tiles(p.tuples.data_ptr(), p.tiles.data_ptr(), p.n_tiles, p.ref.data_ptr(), 1, p.ref_len, rec.data_ptr(), cap, tr.data_ptr())
            c.synchronize()
            torch.cuda.synchronize()
            trn = tr.cpu().numpy().view(_abi.TILE_RESULT_DTYPE)
            single.append(_abi.records_in_order(rec.cpu().numpy().view(_abi.CALLED_ALLELE_DTYPE), trn).copy())
            outs.append((torch.zeros_like(rec), torch.zeros_like(tr), cap))
        torch.cuda.synchronize()
        batches = [(p.tuples.data_ptr(), p.tiles.data_ptr(), p.n_tiles, p.ref.data_ptr(), 1, p.ref_len, o[0].data_ptr(), o[2], o[1].data_ptr())
                   for p, o in zip(ps, outs)]
        for _ in range(3):
            c.call_tiles_batched(batches)
        c.synchronize()
        torch.cuda.synchronize()
        for o, want in zip(outs, single):
            trn = o[1].cpu().numpy().view(_abi.TILE_RESULT_DTYPE)
            got = _abi.records_in_order(o[0].cpu().numpy().view(_abi.CALLED_ALLELE_DTYPE), trn)
            assert got.tobytes() == want.tobytes()
        bad = list(batches)
        bad[2] = bad[2][:7] + (10,) + bad[2][8:]          # record capacity below 256 slots per tile
        with pytest.raises(engine.PiscesHipError) as e:
            c.call_tiles_batched(bad)
        assert e.value.code == _abi.E_BUFFER_TOO_SMALL
        c.call_tiles_batched([])


@pytest.mark.gpu
def test_random_configuration_matrix_matches_oracle(torch_cuda):
    """Forty random combinations of the modes (MNV calling, collapser, noise model, ploidy, strand-bias model, thresholds, gVCF on / off,
    intervals) on one read set with SNVs, MNVs, an insertion and a deletion: the device path against the oracle, every field."""
    from pisces_amd import engine
    rng = np.random.default_rng(2024)
    ref = bytes(rng.choice(list(b"ACGT"), 1300).astype(np.uint8))
    reads = _mnv_reads(rng, ref, 2200, region=(30, 1250))
    for i in range(120):
        if i % 2:
            reads.append({"pos": 560, "cigar": [("M", 41), ("D", 2), ("M", 50)], "seq": (ref[559:600] + ref[602:652]).decode(), "reverse": bool(i & 2)})
        else:
            reads.append({"pos": 760, "cigar": [("M", 41), ("I", 3), ("M", 50)], "seq": (ref[759:800] + b"GAT" + ref[800:850]).decode(), "reverse": bool(i & 2)})
    for r in reads:
        r["quals"] = rng.choice([12, 23, 30, 37, 41], len(r["seq"]), p=[.03, .15, .2, .45, .17]).astype(np.uint8).tolist()
    reads.sort(key=lambda r: r["pos"])
    batch = _abi.ReadBatch(reads)
    refa = np.frombuffer(ref, dtype=np.uint8)
    checked = 0
    for trial in range(40):
        ploidy = int(rng.choice([0, 0, 1, 2]))
        kw = dict(call_mnvs=int(rng.integers(0, 2)), collapse=int(rng.integers(0, 2)), noise_model=int(rng.integers(0, 2)), ploidy=ploidy,
                  strand_bias_model=int(rng.choice([0, 1, 2])), include_reference_calls=int(rng.integers(0, 2)),
                  min_frequency=float(rng.choice([0.005, 0.01, 0.05])) if ploidy == 0 else 0.2,
                  min_variant_qscore=int(rng.choice([10, 20, 30])), max_variant_qscore=int(rng.choice([100, 500])),
                  filter_single_strand=int(rng.integers(0, 2)), low_gq_filter=int(rng.choice([-1, 30])), block_size=2000,
                  max_mnv_length=int(rng.choice([2, 3, 5])), max_gap_between_mnv=int(rng.choice([0, 1, 2])),
                  variant_qscore_filter=int(rng.choice([20, 30, 60])))
        kw["variant_freq_filter"] = kw["min_frequency"]
        intervals = [(100, 400), (520, 900), (1000, 1100)] if rng.random() < 0.3 else None
        if intervals:
            kw["emit_zero_coverage_refs"] = 1
        cfg = _abi.default_config(**kw)
        # the reference calls every candidate of the block (MNV reallocation included) and applies the intervals only when it
        # reports (AlleleCaller.ShouldReport :260-263), so the expectation is the whole region filtered by interval
        exp, exp_alleles, _, _ = orc.run_reads_full(batch, refa, 1, len(ref), cfg)
        if intervals:
            keep = np.zeros(len(exp), dtype=bool)
            for (a, b) in intervals:
                keep |= (exp["position"] >= a) & (exp["position"] <= b)
            exp_alleles = [al for al, k in zip(exp_alleles, keep) if k]
            exp = exp[keep]
        with engine.HipVariantCaller(cfg) as c:
            c.SetReference(refa)
            if intervals:
                c.SetIntervals(intervals)
            c.AddAlleleCounts(batch)
            got, got_alleles = c.CallWithAlleles()
        assert got_alleles == exp_alleles, (trial, kw, intervals)
        assert_records_match(got, exp)
        checked += 1
    assert checked == 40


@pytest.mark.gpu
def test_stitched_reads_with_deletions_use_the_expanded_direction_map(torch_cuda):
    """Stitched reads (XD tag) whose deletion sits on the forward / stitched / reverse boundaries: the candidate's support direction
    comes from the directions INSIDE the deletion (PiscesReadBatch.deletion_directions; GetDeletionDirectionForStitchedRead
    CandidateVariantFinder.cs:468-487), so the deletion's support by direction and its strand bias differ from the anchor rule's."""
    from pisces_amd import engine
    rng = np.random.default_rng(5)
    ref = bytes(rng.choice(list(b"ACGT"), 500).astype(np.uint8))
    reads = []
    for k in range(300):
        start = 100 + int(rng.integers(0, 40))
        left = 200 - start + 1        # start..200 aligned, 201..203 deleted in every carrier
        carrier = k % 3 != 0
        cigar = [("M", left), ("D", 3), ("M", 60)] if carrier else [("M", left + 63)]
        seq = (ref[start - 1:200] + ref[203:263]) if carrier else ref[start - 1:start - 1 + left + 63]
        n_exp = sum(l for _, l in cigar)
        a = int(rng.integers(left - 4, left + 6))          # the stitched region starts around the deletion ...
        b = a + int(rng.integers(0, 8))                    # ... and may end inside or after it
        reads.append({"pos": start, "cigar": cigar, "seq": seq.decode(), "quals": [35] * len(seq),
                      "xd": f"{a}F{b - a}S{n_exp - b}R" if b > a else f"{a}F{n_exp - a}R"})
    reads.sort(key=lambda r: r["pos"])
    batch = _abi.ReadBatch(reads)
    refa = np.frombuffer(ref, dtype=np.uint8)
    kw = dict(include_reference_calls=0, min_frequency=0.01, variant_freq_filter=0.01)
    cfg = _abi.default_config(**kw)
    exp, exp_alleles, _, _ = orc.run_reads_full(batch, refa, 1, len(ref), cfg)
    with engine.HipVariantCaller(cfg) as c:
        c.SetReference(refa)
        c.AddAlleleCounts(batch)
        got, got_alleles = c.CallWithAlleles()
    assert got_alleles == exp_alleles
    assert_records_match(got, exp)
    dele = got[[_abi.info_category(i) == _abi.CAT_DELETION for i in got["info"]]]
    assert len(dele) == 1 and dele[0]["position"] == 200 and dele[0]["allele_support"] == 200
    # the same reads without the map: the anchor rule gives other per-direction supports
    plain = _abi.ReadBatch.from_arrays(batch.position, batch.flags, batch.cigar_offset, batch.cigar_op, batch.cigar_len, batch.seq_offset,
                                       batch.bases, batch.quals, directions=batch.directions)
    old, _, _, _ = orc.run_reads_full(plain, refa, 1, len(ref), cfg)
    old_del = old[[_abi.info_category(i) == _abi.CAT_DELETION for i in old["info"]]]
    assert list(old_del[0]["support_by_dir"]) != list(dele[0]["support_by_dir"])


@pytest.mark.gpu
def test_concurrent_handles_on_threads_share_nothing(torch_cuda):
    """-threadbychr: several SmallVariantCallers (one handle each) run at once on raw threads of one process (JobManager.cs:70-73).
    Six threads, each with its own handle, configuration (somatic / MNV + collapser / Window / diploid) and reads, three rounds each:
    every thread's records equal the oracle's for ITS job (no global mutable state in the library; ctypes releases the GIL)."""
    import threading
    from pisces_amd import engine
    modes = [dict(), dict(call_mnvs=1, collapse=1), dict(noise_model=1), dict(ploidy=1, min_frequency=0.2, variant_freq_filter=0.2),
             dict(call_mnvs=1, max_mnv_length=5, max_gap_between_mnv=2), dict(include_reference_calls=0, strand_bias_model=1)]
    jobs = []
    for j, kw in enumerate(modes):
        rng = np.random.default_rng(900 + j)
        ref = bytes(rng.choice(list(b"ACGT"), 1500).astype(np.uint8))
        reads = _mnv_reads(rng, ref, 1500 + 200 * j, region=(30, 1400))
        reads.sort(key=lambda r: r["pos"])
        batch = _abi.ReadBatch(reads)
        refa = np.frombuffer(ref, dtype=np.uint8)
        cfg = _abi.default_config(block_size=2000, **kw)
        exp, exp_alleles, _, _ = orc.run_reads_full(batch, refa, 1, len(ref), cfg)
        jobs.append((cfg, refa, batch, exp, exp_alleles))
    errors = []
    start = threading.Barrier(len(jobs))

    def run(j):
        try:
            cfg, refa, batch, exp, exp_alleles = jobs[j]
            start.wait()
            for _ in range(3):
                with engine.HipVariantCaller(cfg) as c:
                    c.SetReference(refa)
                    c.AddAlleleCounts(batch)
                    got, got_alleles = c.CallWithAlleles()
                assert got_alleles == exp_alleles
                assert_records_match(got, exp)
        except BaseException as e:   # noqa: BLE001 - reported by the main thread
            errors.append((j, repr(e)[:400]))

    threads = [threading.Thread(target=run, args=(j,)) for j in range(len(jobs))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


def _cross_block_collapse_case(seed=4242):
    """Reads over the 1000|1001 and 2000|2001 block edges: three-base MNVs at 999 and 1999 at 30 %, reads that begin on the second or third
    base of them (their part of the MNV is an open-left candidate: a two-base MNV of the same block, an SNV of the NEXT block), SNVs at
    1020 / 1050 / 1600 and an MNV at 2040 some of whose reads begin on them (open-left candidates with nothing to collapse into)."""
    rng = np.random.default_rng(seed)
    ref = bytearray(rng.choice(list(b"ACGT"), 3200).astype(np.uint8))
    def mut(p, n):
        return "".join(chr([b for b in b"ACGT" if b != ref[p - 1 + i]][(i + p) % 3]) for i in range(n))
    planted = [(999, mut(999, 3), 0.30), (1999, mut(1999, 3), 0.30), (1020, mut(1020, 1), 0.25), (1600, mut(1600, 1), 0.25),
               (1050, mut(1050, 1), 0.3), (2040, mut(2040, 2), 0.25)]
    reads, L = [], 100
    for n in range(5000):
        k = n % 5
        if k == 0: start = int(rng.integers(905, 999))
        elif k == 1: start = int(rng.choice([1000, 1001, 1050, 1020]))
        elif k == 2: start = int(rng.integers(1905, 1999))
        elif k == 3: start = int(rng.choice([2000, 2001, 2040, 2041]))
        else: start = int(rng.integers(1400, 1700))
        seq = bytearray(ref[start - 1: start - 1 + L])
        for (p, alt, frac) in planted:
            lo, hi = max(p, start), min(p + len(alt), start + L)
            if lo < hi and rng.random() < frac:   # the part of the allele the read covers
                seq[lo - start: hi - start] = alt[lo - p: hi - p].encode()
        reads.append({"pos": start, "cigar": [("M", L)], "seq": bytes(seq).decode(), "quals": [37] * L, "reverse": bool(n % 2)})
    reads.sort(key=lambda r: r["pos"])
    return ref, reads, planted


@pytest.mark.gpu
@pytest.mark.parametrize("schedule", [(1500, 2600), (1003, 2002), (1001, 1500, 2001, 2999, 3100)])
def test_collapsable_candidates_of_the_next_block_join_the_batch(torch_cuda, schedule):
    """AddCollapsableFromOtherBlocks (RegionStateManager.cs:321-324, 441-457) in pisces_hip_flush_ex: with the collapser and MNV calling on,
    a batch whose MNV reaches past its last cleared position takes the collapsable SNV / MNV candidates of the held blocks up to upTo;
    the open-left SNV that is the last base of the MNV (a candidate of the next block) collapses into it, candidates that do not collapse
    return to their block and are called with it.  The oracle runs the same upTo schedule (orc_run_reads_schedule); without the step the
    MNVs would keep 400 instead of 482 supporting reads and their last bases would be called as SNVs of their own."""
    from pisces_amd import engine
    ref, reads, planted = _cross_block_collapse_case()
    batch = _abi.ReadBatch(reads)
    refa = np.frombuffer(bytes(ref), dtype=np.uint8)
    cfg = _abi.default_config(call_mnvs=1, collapse=1)
    exp, exp_alleles, exp_called = orc.run_reads_schedule(batch, refa, 1, len(ref), cfg, list(schedule))
    plain, plain_alleles, _ = orc.run_reads_blocks(batch, refa, 1, len(ref), cfg)
    support = lambda recs, alleles, p, n: [int(r["allele_support"]) for r, a in zip(recs, alleles) if int(r["position"]) == p and len(a[1]) == n and a[0] != a[1]]
    if schedule[0] > 1001:   # (upTo = 1001 first: block 1 clears with nothing of block 2 at or below upTo but the SNV at 1001 itself)
        assert support(exp, exp_alleles, 999, 3)[0] > support(plain, plain_alleles, 999, 3)[0]   # the step matters on this input
    with engine.HipVariantCaller(cfg) as c:
        c.SetReference(refa)
        c.AddAlleleCounts(batch)
        got, got_alleles = [], []
        for up_to in tuple(schedule) + (None,):
            r, a = c.CallWithAlleles(upToPosition=up_to)
            got.append(r)
            got_alleles += a
        stats = c.Stats()
    got = np.concatenate(got)
    assert got_alleles == exp_alleles
    assert_records_match(got, exp)
    assert stats["TotalNumCalled"] == exp_called


# ---- forced genotyping alleles (-forcedalleles) ------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("run", ["noisy", "forced1", "forced2"])
def test_forced_gt_functional_test_vcfs_on_the_device_path(torch_cuda, run):
    """ForcedGTFxnlTest.RunForcedGT through libpisceship: the reads of PhiX_S3.bam, the options of the three runs, the nine forced
    alleles of the test's input VCF (pisces_hip_set_forced_alleles), flushed block by block as the reads pass; the records equal the
    oracle's and pisces_hip_format_vcf writes the body lines of PhiX_S3.noisy.vcf / Forced1.vcf / Forced2.vcf byte for byte."""
    from pisces_amd import engine
    from tests import bam_fixtures
    from tests.test_oracle_golden import FORCED_GT, forced_gt_config
    r = FORCED_GT["runs"][run]
    z, batch = bam_fixtures.load("bam_phix")
    cfg = _abi.default_config(**forced_gt_config(r["min_variant_qscore"]))
    forced = [tuple(f) for f in FORCED_GT["forced"]] if r["forced"] else []
    schedule = [1500, 2500, 3500, 4500]
    exp, exp_alleles, exp_called = orc.run_reads_schedule(batch, z["ref"], 1, len(z["ref"]), cfg, schedule, forced=forced)
    with engine.HipVariantCaller(cfg) as c:
        c.SetReference(z["ref"])
        if forced:
            c.SetForcedAlleles(forced)
        c.AddAlleleCounts(batch)
        got, got_alleles = [], []
        for up_to in schedule + [None]:
            rr, a = c.CallWithAlleles(upToPosition=up_to)
            got.append(rr)
            got_alleles += a
        stats = c.Stats()
    got = np.concatenate(got)
    assert got_alleles == exp_alleles
    assert_records_match(got, exp)
    assert stats["TotalNumCalled"] == exp_called
    text = engine.format_vcf("phix", got, alleles=got_alleles, noise_level_from_records=1, noise_level=40, min_frequency_threshold=0.00001)
    last = int(r["lines"][-1].split("\t")[1])
    assert [l for l in text.rstrip("\n").split("\n") if int(l.split("\t")[1]) <= last] == r["lines"]


def _forced_case(seed=77):
    """300x reads over 600 loci with SNVs at 100 (30 %), 200 (2 %: below the frequency cut) and a 3-base deletion after 300 (20 %); forced
    alleles: the three of them, an SNV nobody has (150), another base at 100, an insertion nobody has (250), an MNV nobody has (400), a
    position without reads (590), and one outside the reference window's intervals when intervals are given."""
    rng = np.random.default_rng(seed)
    ref = bytearray(rng.choice(list(b"ACGT"), 640).astype(np.uint8))
    other = lambda p, k=0: chr([b for b in b"ACGT" if b != ref[p - 1]][k])
    reads, L = [], 80
    for n in range(1800):
        start = int(rng.integers(1, 480))
        seq = bytearray(ref[start - 1: start - 1 + L])
        cigar = [("M", L)]
        u = rng.random()
        if start <= 100 < start + L and u < 0.30:
            seq[100 - start] = ord(other(100))
        if start <= 200 < start + L and u > 0.98:
            seq[200 - start] = ord(other(200))
        if start + 5 <= 300 and 304 + 5 <= start + L and 0.4 < u < 0.6:   # deletion of 301..303, anchor base 300
            k = 300 - start + 1
            seq = seq[:k] + bytearray(ref[303: 303 + L - k])
            cigar = [("M", k), ("D", 3), ("M", L - k)]
        reads.append({"pos": start, "cigar": cigar, "seq": bytes(seq).decode(), "quals": [35] * L, "reverse": bool(n % 2)})
    reads.sort(key=lambda r: r["pos"])
    refs = lambda p, n: bytes(ref[p - 1: p - 1 + n]).decode()
    forced = [(100, refs(100, 1), other(100)), (100, refs(100, 1), other(100, 1)), (150, refs(150, 1), other(150)), (200, refs(200, 1), other(200)),
              (300, refs(300, 4), refs(300, 1)), (250, refs(250, 1), refs(250, 1) + "ACG"), (400, refs(400, 3), other(400) + other(401) + other(402)),
              (590, refs(590, 1), other(590))]
    return ref, reads, forced


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["somatic_gvcf", "somatic_vcf", "mnv_vcf", "diploid_gvcf", "diploid_vcf", "low_gq"])
def test_forced_alleles_against_the_oracle(torch_cuda, mode):
    """Forced alleles on the default path (MNV calling off: SNVs are the device counts, a forced SNV takes its support from them), with
    and without Reference rows (gVCF on: the Reference row of the tile kernels stays beside a ForcedReport row; off: Reference rows at
    the forced positions only, RegionState.cs:393-450), with MNV calling on, with the diploid genotyper (DiploidLocusProcessor: forced
    alleles take the genotype the others imply, PISCES_GT_OTHERS beside a heterozygous call) and with the LowGQ filter (a forced row
    has genotype q-score 0).  Records, allele strings, TotalNumCalled and the VCF text equal the oracle's."""
    from pisces_amd import engine
    ref, reads, forced = _forced_case()
    batch = _abi.ReadBatch(reads)
    refa = np.frombuffer(bytes(ref), dtype=np.uint8)
    over = dict(somatic_gvcf=dict(), somatic_vcf=dict(include_reference_calls=0), mnv_vcf=dict(include_reference_calls=0, call_mnvs=1, collapse=0),
                diploid_gvcf=dict(ploidy=1), diploid_vcf=dict(ploidy=1, include_reference_calls=0),
                low_gq=dict(low_gq_filter=20))[mode]
    cfg = _abi.default_config(block_size=250, **over)
    schedule = [260, 520]
    exp, exp_alleles, exp_called = orc.run_reads_schedule(batch, refa, 1, len(ref), cfg, schedule, forced=forced)
    forced_rows = [(int(r["position"]), a) for r, a in zip(exp, exp_alleles) if (int(r["filter_bits"]) >> _abi.FILTER_FORCED_REPORT) & 1]
    assert len(forced_rows) >= 5 and (100, forced[0][1:]) not in forced_rows   # the real SNV at 100 is called, not forced
    with engine.HipVariantCaller(cfg) as c:
        c.SetReference(refa)
        c.SetForcedAlleles(forced)
        c.AddAlleleCounts(batch)
        got, got_alleles = [], []
        for up_to in schedule + [None]:
            rr, a = c.CallWithAlleles(upToPosition=up_to)
            got.append(rr)
            got_alleles += a
        stats = c.Stats()
    got = np.concatenate(got)
    assert got_alleles == exp_alleles
    assert_records_match(got, exp)
    assert stats["TotalNumCalled"] == exp_called
    fmt = lambda recs, alleles: engine.format_vcf("chrF", recs, alleles=alleles, noise_level_from_records=1)
    assert fmt(got, got_alleles) == fmt(exp, exp_alleles)


@pytest.mark.gpu
def test_random_configuration_matrix_with_schedules_and_forced_alleles(torch_cuda):
    """Forty more random mode combinations, this time over a block schedule (400-locus blocks, random upTo positions: held blocks, MNV
    leftovers crossing block edges, collapsable candidates of later blocks) and with random forced alleles (present in the reads or
    not; SNVs, MNVs, insertions, deletions; some at positions without coverage): records, allele strings and TotalNumCalled against
    the oracle running the same schedule."""
    from pisces_amd import engine
    rng = np.random.default_rng(31337)
    ref = bytes(rng.choice(list(b"ACGT"), 1700).astype(np.uint8))
    reads = _mnv_reads(rng, ref, 2600, region=(30, 1500))
    for i in range(160):
        if i % 2:
            reads.append({"pos": 760, "cigar": [("M", 41), ("D", 2), ("M", 50)], "seq": (ref[759:800] + ref[802:852]).decode(), "reverse": bool(i & 2)})
        else:
            reads.append({"pos": 1160, "cigar": [("M", 41), ("I", 3), ("M", 50)], "seq": (ref[1159:1200] + b"GAT" + ref[1200:1250]).decode(), "reverse": bool(i & 2)})
    for r in reads:
        r["quals"] = rng.choice([12, 23, 30, 37, 41], len(r["seq"]), p=[.03, .15, .2, .45, .17]).astype(np.uint8).tolist()
    reads.sort(key=lambda r: r["pos"])
    batch = _abi.ReadBatch(reads)
    refa = np.frombuffer(ref, dtype=np.uint8)
    other = lambda p, k=0: chr([b for b in b"ACGT" if b != ref[p - 1]][k % 3])
    refs = lambda p, n: ref[p - 1: p - 1 + n].decode()
    for trial in range(40):
        ploidy = int(rng.choice([0, 0, 1, 2]))
        kw = dict(call_mnvs=int(rng.integers(0, 2)), collapse=int(rng.integers(0, 2)), noise_model=int(rng.integers(0, 2)), ploidy=ploidy,
                  strand_bias_model=int(rng.choice([0, 1, 2])), include_reference_calls=int(rng.integers(0, 2)),
                  min_frequency=float(rng.choice([0.005, 0.01, 0.05])) if ploidy == 0 else 0.2,
                  min_variant_qscore=int(rng.choice([10, 20, 30])), low_gq_filter=int(rng.choice([-1, 30])), block_size=400,
                  max_mnv_length=int(rng.choice([2, 3, 5])), max_gap_between_mnv=int(rng.choice([0, 1, 2])))
        kw["variant_freq_filter"] = kw["min_frequency"]
        cfg = _abi.default_config(**kw)
        schedule = sorted(int(x) for x in rng.integers(200, 1690, int(rng.integers(1, 6))))
        forced = []
        for _ in range(int(rng.integers(0, 9))):
            p = int(rng.integers(5, 1650))
            kind = int(rng.integers(0, 4))
            if kind == 0: forced.append((p, refs(p, 1), other(p, int(rng.integers(0, 3)))))
            elif kind == 1:
                n = int(rng.integers(2, 4))
                forced.append((p, refs(p, n), "".join(other(p + i, i) for i in range(n))))
            elif kind == 2: forced.append((p, refs(p, 1), refs(p, 1) + "ACGT"[: int(rng.integers(1, 4))]))
            else: forced.append((p, refs(p, 1 + int(rng.integers(1, 4))), refs(p, 1)))
        if rng.random() < 0.5:
            forced += [(760, refs(760, 3), refs(760, 1)), (1160, refs(1160, 1), refs(1160, 1) + "GAT")]   # the planted deletion / insertion
        forced = list(dict.fromkeys(forced))
        exp, exp_alleles, exp_called = orc.run_reads_schedule(batch, refa, 1, len(ref), cfg, schedule, forced=forced)
        with engine.HipVariantCaller(cfg) as c:
            c.SetReference(refa)
            if forced:
                c.SetForcedAlleles(forced)
            c.AddAlleleCounts(batch)
            got, got_alleles = [], []
            for up_to in schedule + [None]:
                rr, a = c.CallWithAlleles(upToPosition=up_to)
                got.append(rr)
                got_alleles += a
            stats = c.Stats()
        got = np.concatenate(got)
        assert got_alleles == exp_alleles, (trial, kw, schedule, forced)
        assert_records_match(got, exp)
        assert stats["TotalNumCalled"] == exp_called, (trial, kw, schedule, forced)


@pytest.mark.gpu
def test_base_quality_sums_are_exact_and_the_same_from_run_to_run(torch_cuda):
    """IAlleleSource.GetSumOfAlleleBaseQualities (pisces_hip_get_base_quality_sums; RegionState._sumOfAlleleBaseQualities): the sums are
    accumulated in fixed point (integer atomics, two 38-bit halves per cell), so they are the same bits whatever order the device adds
    in — two handles, several rounds — and they are the true sum of Math.Pow(10, -(int)q / 10f) over the cell's bases to the last place
    (checked against exact rational arithmetic); the oracle's read-order double sums agree to rounding.  Low-quality bases count under N,
    whose sums nobody reads; cells of A/C/G/T are compared."""
    from fractions import Fraction
    from pisces_amd import engine
    rng = np.random.default_rng(77)
    ref = bytes(rng.choice(list(b"ACGT"), 400).astype(np.uint8))
    reads = []
    for n in range(3000):
        start = int(rng.integers(1, 290))
        seq = bytearray(ref[start - 1: start + 99])
        for k in range(100):
            if rng.random() < 0.02:
                seq[k] = rng.choice(list(b"ACGT"))
        reads.append({"pos": start, "cigar": [("M", 100)], "seq": bytes(seq).decode(), "reverse": bool(n % 2),
                      "quals": rng.choice([2, 12, 23, 30, 37, 41], 100, p=[.02, .04, .2, .2, .38, .16]).astype(np.uint8).tolist()})
    batch = _abi.ReadBatch(reads)
    refa = np.frombuffer(ref, dtype=np.uint8)
    cfg = _abi.default_config(noise_model=1)
    runs = []
    for rep in range(3):
        with engine.HipVariantCaller(cfg) as c:
            c.SetReference(refa)
            c.AddAlleleCounts(batch)
            runs.append((c.GetBaseQualitySums(1, 390).copy(), None))
    assert runs[0][0].tobytes() == runs[1][0].tobytes() == runs[2][0].tobytes()
    sums = runs[0][0]                                                    # [390][6][3][11]
    # exact: per (position, allele, direction) over all anchors, from the reads themselves
    term = {q: Fraction(float(np.float64(10.0) ** np.float64(np.float32(-q) / np.float32(10.0)))) for q in (2, 12, 23, 30, 37, 41)}
    want = {}
    code = {ord("A"): 0, ord("G"): 1, ord("C"): 2, ord("T"): 3}
    for r in reads:
        d = 1 if r["reverse"] else 0
        for k, (b, q) in enumerate(zip(r["seq"].encode(), r["quals"])):
            if q < 20:
                continue                                              # counted under N
            key = (r["pos"] + k, code[b], d)
            want[key] = want.get(key, Fraction(0)) + term[q]
    checked = 0
    for (p, a, d), exact in want.items():
        if p > 390:
            continue
        got = float(sums[p - 1, a, d].sum())                            # the anchors of one cell group add up to the per-direction sum
        assert abs(Fraction(got) - exact) <= Fraction(np.spacing(float(exact))) * 8, (p, a, d, got, float(exact))
        checked += 1
    assert checked > 1500
    # the oracle's state (doubles added in read order) agrees to rounding
    st = orc.State(1, 400, min_bq=20, track_open_ended=False)
    for r in reads:
        assert st.add_allele_counts(orc.make_read(r["pos"], r["seq"], quals=r["quals"], reverse=r["reverse"])) == 0
    for p in range(1, 391, 7):
        for a in range(4):
            for d in range(2):
                o = orc.lib.orc_get_sum_base_quality(st.h, p, a, d, 0, -1, 0, 0)
                assert float(sums[p - 1, a, d].sum()) == pytest.approx(o, rel=1e-12, abs=1e-18)


def test_flush_pair_gives_what_flush_gives_while_the_next_reads_are_added(torch_cuda):
    """pisces_hip_flush_begin / pisces_hip_flush_end: the same alleles, call by call, as pisces_hip_flush in SmallVariantCaller's loop,
    with the next block's reads staged and added between begin and end (the compacted log is a bound long, holes behind what it kept);
    the state entries that must wait say so; a batch with host-side candidates (a deletion) is flushed synchronously inside begin."""
    from pisces_amd import engine, synth
    p = synth.make_pileup(n_loci=6300, depth=60, seed=19)
    cfg = _abi.default_config()
    A = p.base.shape[0]
    ref = p.ref.numpy()
    per_call = 3   # amplicons per add_reads: blocks end in the middle of some calls, so entries are kept across flushes
    want = []
    with engine.HipVariantCaller(cfg) as c:
        c.SetReference(ref)
        for a0 in range(0, A, per_call):
            c.AddAlleleCounts(synth.reads_of(p, min(per_call, A - a0), first_amplicon=a0))
            want.append(c.Call(p.region_start + a0 * synth.READ_LEN - 1).copy())
        want.append(c.Call(None).copy())
        want_stats = c.Stats()
    for staged in (False, True):
        got = []
        with engine.HipVariantCaller(cfg) as c:
            c.SetReference(ref)
            pending = False
            for k, a0 in enumerate(range(0, A, per_call)):
                b = synth.reads_of(p, min(per_call, A - a0), first_amplicon=a0)
                c.AddAlleleCounts(c.StageReads(b) if staged else b)      # (between begin and end of the flush before)
                if pending:
                    if k == 2:   # what has to wait says so, and nothing is lost by asking
                        with pytest.raises(engine.PiscesHipError) as e:
                            c.Call(None)
                        assert e.value.code == _abi.E_STATE
                        with pytest.raises(engine.PiscesHipError) as e:
                            c.CallBegin(None)
                        assert e.value.code == _abi.E_STATE
                        n = C.c_int64(0)
                        small = np.zeros(1, dtype=_abi.CALLED_ALLELE_DTYPE)
                        from pisces_amd._native import lib
                        rc = lib.pisces_hip_flush_end(c.handle, small.ctypes.data, 0, C.byref(n))
                        assert (rc == _abi.E_BUFFER_TOO_SMALL and n.value == len(want[k - 1])) or (rc == 0 and len(want[k - 1]) == 0)
                        if rc == 0:
                            got.append(small[:0])
                            pending = False
                    if pending:
                        got.append(c.CallEnd().copy())
                c.CallBegin(p.region_start + a0 * synth.READ_LEN - 1)
                pending = True
            got.append(c.CallEnd().copy())
            c.CallBegin(None)
            got.append(c.CallEnd().copy())
            with pytest.raises(engine.PiscesHipError) as e:
                c.CallEnd()
            assert e.value.code == _abi.E_STATE
            stats = c.Stats()
            assert len(c.Call(None)) == 0
        assert len(got) == len(want)
        for g, w in zip(got, want):
            assert g.tobytes() == w.tobytes()
        assert stats == want_stats
    # the pair and the plain flush in turn on one handle (the log keeps holes behind a pair's compaction; a plain flush squeezes them out)
    got = []
    with engine.HipVariantCaller(cfg) as c:
        c.SetReference(ref)
        c.SetIntervals([(p.region_start, p.region_start + p.n_loci - 1)])
        for k, a0 in enumerate(range(0, A, per_call)):
            c.AddAlleleCounts(synth.reads_of(p, min(per_call, A - a0), first_amplicon=a0))
            up_to = p.region_start + a0 * synth.READ_LEN - 1
            if k % 3 == 0:
                got.append(c.Call(up_to).copy())
            else:
                c.CallBegin(up_to)
                if k % 3 == 1:
                    _ = c.GetCounts(p.region_start + a0 * synth.READ_LEN, 4)   # (reading counts between begin and end is allowed)
                got.append(c.CallEnd().copy())
        got.append(c.Call(None).copy())
    assert np.concatenate(got).tobytes() == np.concatenate(want).tobytes()
    # host-side candidates: a deletion in some reads -> the pair must give what the plain flush gives
    rng = np.random.default_rng(2)
    refb = bytes(rng.choice(list(b"ACGT"), 2600).astype(np.uint8))
    reads = []
    for i in range(120):
        s = 900 + (i % 7) * 3
        if i % 3 == 0:
            reads.append({"pos": s, "cigar": [("M", 60), ("D", 4), ("M", 60)], "seq": (refb[s - 1:s + 59] + refb[s + 63:s + 123]).decode(), "quals": [37] * 120, "reverse": bool(i % 2)})
        else:
            reads.append({"pos": s, "cigar": [("M", 124)], "seq": refb[s - 1:s + 123].decode(), "quals": [37] * 124, "reverse": bool(i % 2)})
    with engine.HipVariantCaller(cfg) as c:
        c.SetReference(refb)
        c.AddAlleleCounts(reads)
        plain_way = c.Call(None).copy()
    with engine.HipVariantCaller(cfg) as c:
        c.SetReference(refb)
        c.AddAlleleCounts(reads)
        c.CallBegin(None)
        pair_way = c.CallEnd().copy()
    assert len(plain_way) > 0 and pair_way.tobytes() == plain_way.tobytes()
    assert any(_abi.info_category(int(r["info"])) == _abi.CAT_DELETION for r in plain_way)
    # ... and with the allele strings (pisces_hip_flush_end_ex = pisces_hip_flush_ex): what a host that writes VCF rows needs
    with engine.HipVariantCaller(cfg) as c:
        c.SetReference(refb)
        c.AddAlleleCounts(reads)
        plain_recs, plain_alleles = c.CallWithAlleles(None)
    with engine.HipVariantCaller(cfg) as c:
        c.SetReference(refb)
        c.AddAlleleCounts(reads)
        c.CallBegin(None)
        pair_recs, pair_alleles = c.CallEndWithAlleles(capacity=8)   # (too small on purpose: the counts come back and the call repeats)
    assert pair_recs.tobytes() == plain_recs.tobytes() and pair_alleles == plain_alleles
    assert any(len(r) == 5 and len(a) == 1 for r, a in pair_alleles)   # the 4-base deletion's strings
    # a flush that ran on the device alone: every row's alleles come from its base codes
    with engine.HipVariantCaller(cfg) as c:
        c.SetReference(refb)
        c.AddAlleleCounts([r for r in reads if len(r["cigar"]) == 1])
        c.CallBegin(None)
        pair_recs, pair_alleles = c.CallEndWithAlleles()
    with engine.HipVariantCaller(cfg) as c:
        c.SetReference(refb)
        c.AddAlleleCounts([r for r in reads if len(r["cigar"]) == 1])
        plain_recs, plain_alleles = c.CallWithAlleles(None)
    assert len(pair_recs) > 0 and pair_recs.tobytes() == plain_recs.tobytes() and pair_alleles == plain_alleles


@pytest.mark.gpu
def test_device_count(torch_cuda):
    from pisces_amd import engine
    assert engine.device_count() >= 1


@pytest.mark.gpu
def test_flush_views_hand_out_the_rows_the_copying_flushes_return(torch_cuda):
    """pisces_hip_flush_view / pisces_hip_flush_end_view: the same rows as pisces_hip_flush / pisces_hip_flush_end, read where they lie."""
    import ctypes as C
    from pisces_amd import _native, engine, synth
    cfg = _abi.default_config()
    p = synth.make_pileup(6000, 300, seed=5)
    ref = p.ref.cpu().numpy()
    A = p.base.shape[0]
    whole = synth.reads_of(p, A, first_amplicon=0)
    half = A // 2
    with engine.HipVariantCaller(cfg) as c:
        c.SetReference(ref)
        c.AddAlleleCounts(whole)
        want = [c.Call(p.region_start + half * synth.READ_LEN - 1).copy(), c.Call(None).copy()]
    assert len(want[0]) > 0 and len(want[1]) > 0
    with engine.HipVariantCaller(cfg) as c:
        c.SetReference(ref)
        c.AddAlleleCounts(whole)
        got0 = c.CallView(p.region_start + half * synth.READ_LEN - 1).copy()      # (the view dies with the next flush: copy before it)
        c.CallBegin(None)
        got1 = c.CallEndView().copy()
        assert len(c.CallView(None)) == 0                                           # nothing left
    assert got0.tobytes() == want[0].tobytes() and got1.tobytes() == want[1].tobytes()
    # a batch with host-side candidates (a deletion): the merged rows, the candidate index and the allele strings through the view
    rng = np.random.default_rng(2)
    refb = bytes(rng.choice(list(b"ACGT"), 2600).astype(np.uint8))
    reads = []
    for i in range(90):
        s = 900 + (i % 7) * 3
        if i % 3 == 0:
            reads.append({"pos": s, "cigar": [("M", 60), ("D", 4), ("M", 60)], "seq": (refb[s - 1:s + 59] + refb[s + 63:s + 123]).decode(), "quals": [37] * 120, "reverse": bool(i % 2)})
        else:
            reads.append({"pos": s, "cigar": [("M", 124)], "seq": refb[s - 1:s + 123].decode(), "quals": [37] * 124, "reverse": bool(i % 2)})
    with engine.HipVariantCaller(cfg) as c:
        c.SetReference(refb)
        c.AddAlleleCounts(reads)
        want_recs, want_alleles = c.CallWithAlleles(None)
    with engine.HipVariantCaller(cfg) as c:
        c.SetReference(refb)
        c.AddAlleleCounts(reads)
        rows, n, idx, cands, nc, pool, nb = C.c_void_p(), C.c_int64(0), C.c_void_p(), C.c_void_p(), C.c_int64(0), C.c_void_p(), C.c_int64(0)
        rc = _native.lib.pisces_hip_flush_view(c._h, -1, C.byref(rows), C.byref(n), C.byref(idx), C.byref(cands), C.byref(nc), C.byref(pool), C.byref(nb))
        assert rc == 0 and n.value == len(want_recs) and nc.value >= 1 and idx.value and cands.value and pool.value
        got = np.frombuffer((C.c_uint8 * (64 * n.value)).from_address(rows.value), dtype=_abi.CALLED_ALLELE_DTYPE).copy()
        index = np.frombuffer((C.c_int32 * n.value).from_address(idx.value), dtype=np.int32).copy()
        cand = (_abi.PiscesCandidate * nc.value).from_address(cands.value)
        text = bytes((C.c_uint8 * nb.value).from_address(pool.value))
        got_alleles = []
        for r, ci in zip(got, index):
            if ci < 0:
                got_alleles.append((_abi.BASE_OF_ALLELE[_abi.info_ref(r["info"])], _abi.BASE_OF_ALLELE[_abi.info_alt(r["info"])]))
            else:
                o = cand[ci].allele_offset
                got_alleles.append((text[o:o + cand[ci].ref_len].decode(), text[o + cand[ci].ref_len:o + cand[ci].ref_len + cand[ci].alt_len].decode()))
    assert got.tobytes() == want_recs.tobytes() and got_alleles == want_alleles
    # the pair the same way: pisces_hip_flush_end_view hands out the index, candidates and allele strings of a batch that was flushed inside flush_begin
    with engine.HipVariantCaller(cfg) as c:
        c.SetReference(refb)
        c.AddAlleleCounts(reads)
        c.CallBegin(None)
        rows, n, idx, cands, nc, pool, nb = C.c_void_p(), C.c_int64(0), C.c_void_p(), C.c_void_p(), C.c_int64(0), C.c_void_p(), C.c_int64(0)
        rc = _native.lib.pisces_hip_flush_end_view(c._h, C.byref(rows), C.byref(n), C.byref(idx), C.byref(cands), C.byref(nc), C.byref(pool), C.byref(nb))
        assert rc == 0 and n.value == len(want_recs) and nc.value >= 1 and idx.value and cands.value and pool.value
        got = np.frombuffer((C.c_uint8 * (64 * n.value)).from_address(rows.value), dtype=_abi.CALLED_ALLELE_DTYPE).copy()
        index = np.frombuffer((C.c_int32 * n.value).from_address(idx.value), dtype=np.int32).copy()
        cand = (_abi.PiscesCandidate * nc.value).from_address(cands.value)
        text = bytes((C.c_uint8 * nb.value).from_address(pool.value))
        pair_alleles = []
        for r, ci in zip(got, index):
            if ci < 0:
                pair_alleles.append((_abi.BASE_OF_ALLELE[_abi.info_ref(r["info"])], _abi.BASE_OF_ALLELE[_abi.info_alt(r["info"])]))
            else:
                o = cand[ci].allele_offset
                pair_alleles.append((text[o:o + cand[ci].ref_len].decode(), text[o + cand[ci].ref_len:o + cand[ci].ref_len + cand[ci].alt_len].decode()))
        assert got.tobytes() == want_recs.tobytes() and pair_alleles == want_alleles
        rc = _native.lib.pisces_hip_flush_end_view(c._h, C.byref(rows), C.byref(n), None, None, None, None, None)
        assert rc == _abi.E_STATE        # no flush_begin before it


def test_summary_reduce_through_the_c_abi_on_two_devices(torch_cuda):
    """pisces_hip_comm_init / pisces_hip_reduce_summary with one process per GPU (RCCL over xGMI, bound by the library): two ranks on two
    devices must both get the sums of what they handed in.  Skipped on a one-GPU box — RCCL refuses two ranks on one device — so that the
    first node with several GPUs that runs this suite is also the first to run that code (tools/reduce_check.py)."""
    import os
    import subprocess
    import sys
    if torch_cuda.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29517", os.path.join(root, "tools", "reduce_check.py")], cwd=root, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert r.stdout.count("reduce_check rank") == 2


@pytest.mark.gpu
@pytest.mark.parametrize("seed", list(range(20)))
def test_split_form_equals_the_legacy_form_on_random_schedules(torch_cuda, seed):
    """MNV calling on: the split form (plain SNV groups stay in the device's SNV store, the tile kernel calls SNVs from the counts off the
    dirty loci: DESIGN section 8a) against the form in which every candidate group visits the host (PISCES_HIP_MNV_SPLIT=0) — AlleleCaller's
    MNV pass, MnvReallocator and the all-alleles pass (AlleleCaller.cs:60-141) either way.  Random reads (planted MNVs, some gapped, SNVs
    beside them, errors that make failing MNVs, low-quality bases that open candidates up, insertions / deletions, X and = operations),
    reads added in batches that straddle the block edges, random flush schedules, collapser on / off, gVCF on / off, MNV limits, an
    interval set or forced alleles now and then: records, allele strings and totals must be the same — and the oracle's, which runs the same
    upTo schedule over all the reads (orc_run_reads_schedule; no interval sets there).  (An = operation over a base that differs from the
    reference is such a case: an allele count no SNV candidate stands for, a span mark of the walk like the bases of an X operation.)"""
    from pisces_amd import engine
    rng = np.random.default_rng(9000 + seed)
    ref = bytes(rng.choice(list(b"ACGT"), 4200).astype(np.uint8))
    reads = _mnv_reads(rng, bytearray(ref), int(rng.integers(1500, 4000)), region=(50, 4000), snv_rate=float(rng.choice([0.002, 0.006])))
    for i, r in enumerate(reads):   # a few reads whose M run is split in = / X operations (ProcessCigarOps walks M only)
        if i % 37 == 0 and r["cigar"] == [("M", 100)]:
            k = int(rng.integers(10, 80))
            r["cigar"] = [("=", k), ("X", 2), ("M", 100 - k - 2)]
    reads.sort(key=lambda r: r["pos"])
    kw = dict(call_mnvs=1, max_mnv_length=int(rng.choice([2, 3, 5])), max_gap_between_mnv=int(rng.choice([0, 1, 2])), collapse=int(rng.integers(0, 2)),
              include_reference_calls=int(rng.integers(0, 2)), min_frequency=float(rng.choice([0.01, 0.05])))
    cfg = _abi.default_config(**kw)
    intervals = [(200, 1700), (1900, 3100), (3300, 3900)] if seed % 4 == 1 else None
    forced = [(1500, chr(ref[1499]), "A" if chr(ref[1499]) != "A" else "C"),
              (2600, ref[2599:2601].decode(), "TT" if ref[2599] != ord("T") and ref[2600] != ord("T") else "GG" if ref[2599] != ord("G") and ref[2600] != ord("G") else "CC")] if seed % 4 == 2 else None
    cuts = sorted(set(int(x) for x in rng.integers(0, len(reads), 4)) | {len(reads)})
    ups = [int(x) for x in sorted(rng.integers(600, 3900, len(cuts) - 1))] + [None]
    host_cands = None
    if seed % 4 == 3:   # an SNV, an MNV and a deletion the host adds with support of their own
        def other(p):
            return "A" if ref[p - 1] != ord("A") else "C"
        host_cands = [dict(position=3950, category=_abi.CAT_SNV, ref=chr(ref[3949]), alt=other(3950), support_by_dir=(3, 2, 0), well_anchored_by_dir=(3, 2, 0)),
                      dict(position=3960, category=_abi.CAT_MNV, ref=ref[3959:3961].decode(), alt=other(3960) + other(3961), support_by_dir=(4, 4, 0),
                           well_anchored_by_dir=(4, 4, 0)),
                      dict(position=3970, category=_abi.CAT_DELETION, ref=ref[3969:3972].decode(), alt=chr(ref[3969]), support_by_dir=(5, 5, 0),
                           well_anchored_by_dir=(5, 5, 0))]
    out, schedule = [], []
    for split in (None, 0):
        with env(PISCES_HIP_MNV_SPLIT=split):
            with engine.HipVariantCaller(cfg) as c:
                c.SetReference(ref)
                if intervals:
                    c.SetIntervals(intervals)
                if forced:
                    c.SetForcedAlleles(forced)
                rows, alleles, a0 = [], [], 0
                for cut, up in zip(cuts, ups):
                    c.AddAlleleCounts(_abi.ReadBatch(reads[a0:cut]))
                    a0 = cut
                    if host_cands:   # IStateManager.AddCandidates from the host between the reads: their loci are the candidate kernel's
                        c.AddCandidates(host_cands)
                    if up is not None:
                        up = min(up, reads[cut - 1]["pos"] - 1) if cut else up   # (Call(upTo) behind the last read added, as SmallVariantCaller)
                        if split is None:
                            schedule.append(up)
                    r, a = c.CallWithAlleles(up, capacity=1 << 15)
                    rows.append(r)
                    alleles += a
                out.append((np.concatenate(rows), alleles, c.Stats()))
    (got, ga, gs), (want, wa, ws) = out
    cats = set(((want["info"] >> 4) & 7).tolist())
    assert _abi.CAT_MNV in cats and _abi.CAT_SNV in cats and len(want) > 50
    assert got.tobytes() == want.tobytes() and ga == wa and gs == ws, (seed, kw)
    if not intervals and not host_cands:
        exp, exp_alleles, exp_called = orc.run_reads_schedule(_abi.ReadBatch(reads), np.frombuffer(ref, np.uint8), 1, len(ref), cfg, schedule, forced=forced or ())
        assert ga == exp_alleles and gs["TotalNumCalled"] == exp_called
        assert_records_match(got, exp)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", list(range(12)))
def test_every_switch_of_the_library_leaves_the_records_alone(torch_cuda, seed):
    """The forms the library can take for the same work — rows merged in place or into a vector, a batch checked by the host's pass over
    the CIGARs or on the device, candidates merged on the host or on the device, the candidate walk base by base or events first, the
    germline genotypes by the kernel or by the host pass, every batch a segment of its own or appended to the open one, reads handed over
    from host arrays or in device memory, the candidates looked at between an add and its flush — on random reads, random modes (MNV
    calling, collapser, ploidy, gVCF, thresholds) and a random flush schedule: every form gives the records, allele strings and totals
    of the default (and the default the oracle's)."""
    from pisces_amd import engine
    rng = np.random.default_rng(7100 + seed)
    ref = bytes(rng.choice(list(b"ACGT"), 3300).astype(np.uint8))
    from tests.test_read_store import random_reads
    reads = _mnv_reads(rng, bytearray(ref), int(rng.integers(1200, 3000)), region=(50, 3100), snv_rate=float(rng.choice([0.002, 0.006])))
    if seed % 3 == 0:   # reads with any CIGAR (clips, skips, pads, terminal deletions), N bases, stitched per-base directions
        reads += random_reads(rng, 300, 60, 2900, exotic=False, sort=False)
    reads.sort(key=lambda r: r["pos"])
    ploidy = int(rng.choice([0, 0, 0, 1, 2]))
    kw = dict(call_mnvs=int(rng.integers(0, 2)), max_mnv_length=int(rng.choice([2, 3])), max_gap_between_mnv=int(rng.choice([0, 1])),
              collapse=int(rng.integers(0, 2)), include_reference_calls=int(rng.integers(0, 2)), ploidy=ploidy,
              noise_model=int(rng.choice([0, 0, 1])), strand_bias_model=int(rng.choice([1, 1, 2])),
              min_frequency=0.2 if ploidy else float(rng.choice([0.01, 0.05])))
    if ploidy:
        kw.update(variant_freq_filter=0.2, low_gq_filter=30, max_genotype_qscore=1000)
    cfg = _abi.default_config(**kw)
    cuts = sorted(set(int(x) for x in rng.integers(1, len(reads), 3)) | {len(reads)})
    ups = [int(x) for x in sorted(rng.integers(600, 3000, len(cuts) - 1))] + [None]

    def run(environ, device_reads=False, peek=False):
        with env(**environ):
            with engine.HipVariantCaller(cfg) as c:
                c.SetReference(ref)
                rows, alleles, a0 = [], [], 0
                for cut, up in zip(cuts, ups):
                    batch = _abi.ReadBatch(reads[a0:cut])
                    if device_reads:
                        c.AddDeviceReads(engine.DeviceReadBatch.from_host(batch))
                    else:
                        c.AddAlleleCounts(batch)
                    a0 = cut
                    if peek:
                        c.GetCandidates(None)
                    if up is not None:
                        up = min(up, reads[cut - 1]["pos"] - 1)
                    r, a = c.CallWithAlleles(up, capacity=1 << 15)
                    rows.append(r)
                    alleles += a
                return np.concatenate(rows), alleles, c.Stats()
    want = run({})
    assert len(want[0]) >= 1
    schedule = [min(up, reads[cut - 1]["pos"] - 1) for cut, up in zip(cuts, ups) if up is not None]
    exp, exp_alleles, exp_called = orc.run_reads_schedule(_abi.ReadBatch(reads), np.frombuffer(ref, np.uint8), 1, len(ref), cfg, schedule)
    assert want[1] == exp_alleles and want[2]["TotalNumCalled"] == exp_called
    assert_records_match(want[0], exp)
    forms = {"rows merged by copy": dict(environ=dict(PISCES_HIP_MERGE_IN_PLACE=0)),
             "checks on the device": dict(environ=dict(PISCES_HIP_DEVICE_CHECKS=1)),
             "checks on the host": dict(environ=dict(PISCES_HIP_DEVICE_CHECKS=0)),
             "candidates merged on the host": dict(environ=dict(PISCES_HIP_DEVICE_MERGE=0)),
             "walk base by base": dict(environ=dict(PISCES_HIP_FINDER="bases")),
             "genotypes by the host pass": dict(environ=dict(PISCES_HIP_DEVICE_GENOTYPER=0)),
             "every batch its own segment": dict(environ=dict(PISCES_HIP_STORE_DIRECT_BYTES=0)),
             "every batch appended": dict(environ=dict(PISCES_HIP_STORE_DIRECT_BYTES=1 << 40, PISCES_HIP_STORE_SEAL_BYTES=1 << 40)),
             "reads in device memory": dict(environ={}, device_reads=True),
             "candidates looked at after every add": dict(environ={}, peek=True),
             "every candidate group to the host": dict(environ=dict(PISCES_HIP_MNV_SPLIT=0))}
    for name, how in forms.items():
        got = run(**how)
        assert got[0].tobytes() == want[0].tobytes() and got[1] == want[1] and got[2] == want[2], (seed, name, kw)


# seeds of tests/fuzz_cases.py whose records once differed from the oracle's (what each one found is in DESIGN.md section 5), then a band
# of fresh ones
FUZZ_SEEDS_THAT_ONCE_DIFFERED = [101, 213, 353, 369, 466, 1874, 1878, 6065, 500132, 500714, 501344]


@pytest.mark.gpu
@pytest.mark.parametrize("seed", FUZZ_SEEDS_THAT_ONCE_DIFFERED + list(range(20000, 20030)) + list(range(100000, 100020)) + list(range(200000, 200060)) + list(range(300000, 300030)) + list(range(500000, 500030)) + list(range(600000, 600020)))
def test_streaming_surface_fuzz_against_the_oracle(torch_cuda, seed):
    """One draw of tests/fuzz_cases.py: random reads (planted MNVs and SNVs, any CIGAR, = and X operations, bases that are no A C G T N,
    stitched directions), random modes (MNV calling, collapser and its thresholds, ploidy, gVCF, zero-coverage rows, noise model,
    strand-bias model, quality / depth / frequency thresholds, block size, RMxN, forced alleles) and a random flush schedule through the
    streaming surface: records, allele strings and TotalNumCalled equal the oracle's run of the same schedule.  Seeds from 100 000 add a
    deep pile (counts beyond the memo tables) and more thresholds; seeds from 200 000 run one of the twenty forms the library can take
    (fuzz_cases.FORMS: every environment switch, reads in device memory, the observation-log chain, BAM bytes), three draws each; seeds
    from 300 000 an interval set (the oracle's schedule takes it); seeds from 500 000 candidates the host hands in itself; seeds from 600 000 collapser thresholds
    that keep an open-ended SNV and its twin apart."""
    from tests.fuzz_cases import one
    why, kw, rows, forced = one(seed)
    assert why is None, (seed, why, kw, forced)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", list(range(60)))
def test_device_resident_surface_fuzz_against_the_oracle(torch_cuda, seed):
    """One draw of fuzz_cases.one_tuples: bucketed observation tuples (1 to 700 loci at 3x to 6000x, any allele mix, qualities, directions,
    tile sizes 7 to 64, padded or unaligned segments, N and homopolymers in the reference) and every threshold of the configuration
    through pisces_hip_call_tiles (slot layout or compacted): every field of every record equals the oracle's over the same observations
    (52 600 draws were run beside the suite: tools/fuzz_oracle.py tuples first n)."""
    from tests.fuzz_cases import one_tuples
    why, kw, rows = one_tuples(seed)
    assert why is None, (seed, why, kw)
