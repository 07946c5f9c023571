"""The read store of the streaming surface (pisces_amd/csrc/store_kernels.hip.h, surface_store.inc.h):
pisces_hip_add_reads keeps the reads in HBM and pisces_hip_flush calls straight from them.  Checked here against the oracle
(RegionStateManager.AddAlleleCounts, RegionStateManager.cs:118-220; SmallVariantCaller's block schedule, SmallVariantCaller.cs:88-189)
and against the earlier chain (PISCES_HIP_READ_PATH=log: observation log + bucketing), record for record, byte for byte — through
every way a batch can join the store (a segment of its own, appended to the open segment, more batches than segments), with the
floor that DoneProcessing (RegionStateManager.cs:336-353) leaves behind, with reads out of position order, with bases that are no
A C G T N, and with a quality threshold the fused kernel is not compiled for."""
import contextlib
import os

import numpy as np
import pytest

from pisces_amd import _abi
from tests import orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch


@contextlib.contextmanager
def env(**kv):
    """Environment switches the library reads when a handle is created."""
    old = {k: os.environ.get(k) for k in kv}
    for k, v in kv.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = str(v)
    try:
        yield
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


STORE_MODES = {
    "default": {},
    "every batch its own segment": {"PISCES_HIP_STORE_DIRECT_BYTES": 0},
    "every batch appended to the open segment": {"PISCES_HIP_STORE_DIRECT_BYTES": 1 << 40, "PISCES_HIP_STORE_SEAL_BYTES": 1 << 40},
    "open segment closed after every batch": {"PISCES_HIP_STORE_DIRECT_BYTES": 1 << 40, "PISCES_HIP_STORE_SEAL_BYTES": 1},
}


def random_reads(rng, n, lo, hi, exotic=False, with_dirs=True, sort=True):
    """Reads with arbitrary CIGARs (insertions, deletions, skips, soft / hard clips, pads, terminal deletions), N bases, low qualities,
    stitched per-base directions; exotic: bases that are no A C G T N (IUPAC codes, lower case, '=', NUL, 0xFF)."""
    letters = list(b"ACGTN") + (list(b"RYKMacgtn=.*") + [0, 255, 0x44, 0x4B, 0x51] if exotic else [])
    p = np.array([.235] * 4 + [.03] + ([.03 / 17] * 17 if exotic else []))
    p = p / p.sum()
    reads = []
    for i in range(n):
        kind = i % 4
        if kind == 0:     # one aligned run
            ops = [("M", int(rng.integers(20, 160)))]
        elif kind == 1:   # clips around one aligned run, split in = / X, pads and hard clips
            ops = [("H", 3)] * int(rng.integers(0, 2)) + [("S", int(rng.integers(1, 9)))] * int(rng.integers(0, 2))
            ops += [("=", int(rng.integers(5, 60))), ("X", 1), ("P", 2), ("M", int(rng.integers(5, 60)))]
            ops += [("S", int(rng.integers(1, 9)))] * int(rng.integers(0, 2)) + [("H", 2)] * int(rng.integers(0, 2))
        else:             # anything
            ops = [(str(rng.choice(list("MMMIDSN"))), int(rng.integers(1, 30))) for _ in range(int(rng.integers(1, 7)))]
            ops = [(o, l) for k, (o, l) in enumerate(ops) if o != "S" or k in (0, len(ops) - 1)]
            if not any(o == "M" for o, _ in ops):
                ops.append(("M", 5))
            if i % 11 == 0:
                ops.append(("D", int(rng.integers(1, 6))))
            if i % 13 == 0:
                ops += [("D", int(rng.integers(1, 6))), ("S", int(rng.integers(1, 5)))]
        rl = sum(l for o, l in ops if o in "MIS=X")
        rd = {"pos": int(rng.integers(lo, hi)), "cigar": ops, "seq": bytes(rng.choice(letters, rl, p=p).astype(np.uint8)),
              "quals": rng.choice([10, 25, 37, 200], rl, p=[.15, .2, .63, .02]).astype(np.uint8).tolist(), "reverse": bool(rng.integers(0, 2))}
        if with_dirs and i % 5 == 0:
            rd["dirs"] = rng.integers(0, 3, rl).astype(np.uint8).tolist()
        reads.append(rd)
    if sort:
        reads.sort(key=lambda r: r["pos"])
    return reads


def oracle_counts(reads, start, n_loci, min_bq=20):
    st = orc.State(start, n_loci, min_bq=min_bq)
    for d in reads:
        assert st.add_allele_counts(orc.make_read(d["pos"], d["seq"], cigar=d["cigar"],
                                                  quals=d["quals"], reverse=d["reverse"], dirs=d.get("dirs"))) == 0
    return st.counts()


@pytest.mark.parametrize("mode", list(STORE_MODES))
@pytest.mark.parametrize("sort", [True, False], ids=["sorted", "unsorted"])
def test_counts_from_the_store_equal_the_oracle(torch_cuda, mode, sort):
    """IAlleleSource.GetAlleleCount served from the read store (accumulate_store_tiles_kernel) == the oracle's AddAlleleCounts, for
    arbitrary CIGARs and bases, however the batches joined the store and whether or not they came in position order (a segment that is
    not sorted is scanned in whole)."""
    from pisces_amd import engine
    rng = np.random.default_rng(5)
    reads = random_reads(rng, 700, 940, 1900, exotic=True, sort=sort)
    exp = oracle_counts(reads, 900, 2300)
    with env(PISCES_HIP_READ_PATH=None, **STORE_MODES[mode]):
        with engine.HipVariantCaller(_abi.default_config()) as c:
            for k in range(0, len(reads), 53):     # 14 batches: more than the store has segments
                c.AddAlleleCounts(_abi.ReadBatch(reads[k:k + 53]))
            got = c.GetCounts(900, 2300)
            assert c.Stats()["reads"] == len(reads)
    np.testing.assert_array_equal(got.reshape(exp.shape), exp)


def _schedule_run(reads_batches, ref, cfg, ups, environ):
    """add_reads batch k, then flush(ups[k]); a final flush at the end.  Returns records, allele strings, stats."""
    from pisces_amd import engine
    recs, alleles = [], []
    with env(**environ):
        with engine.HipVariantCaller(cfg) as c:
            c.SetReference(ref)
            for batch, up in zip(reads_batches, ups):
                c.AddAlleleCounts(batch)
                if up is not None:
                    r, a = c.CallWithAlleles(up, capacity=1 << 14)
                    recs.append(r)
                    alleles += a
            r, a = c.CallWithAlleles(None, capacity=1 << 14)
            recs.append(r)
            alleles += a
            stats = c.Stats()
    return np.concatenate(recs), alleles, stats


@pytest.mark.parametrize("mode", list(STORE_MODES))
@pytest.mark.parametrize("call_mnvs", [0, 1], ids=["snv_indel", "mnv"])
def test_block_schedule_from_the_store_equals_the_log_chain_and_the_oracle(torch_cuda, mode, call_mnvs):
    """SmallVariantCaller's protocol over four blocks, reads arriving in position order in batches that straddle block edges: after a
    flush the reads that reach into blocks still held stay in the store with the flushed positions floored (nothing is counted twice,
    nothing is lost).  Records and allele strings must equal what the observation-log chain gives, byte for byte, and the oracle
    running the same schedule."""
    rng = np.random.default_rng(17 + call_mnvs)
    ref = np.frombuffer(bytes(rng.choice(list(b"ACGT"), 4400).astype(np.uint8)), dtype=np.uint8)
    reads = []
    for i in range(2600):
        pos = int(rng.integers(60, 3900))
        ln = int(rng.integers(60, 151))
        seq = bytearray(ref[pos - 1:pos - 1 + ln].tobytes())
        ops = [("M", ln)]
        for k in range(len(seq)):
            if rng.random() < 0.004:
                seq[k] = rng.choice(list(b"ACGT"))
        site = pos // 97 * 97 + 50     # shared variant sites: SNVs, 2-base MNVs, deletions, insertions
        off = site - pos
        if 8 <= off < ln - 12:
            kind = (site // 97) % 4
            take = rng.random() < 0.35
            if take and kind == 0:
                seq[off] = ord("ACGT"[("ACGT".index(chr(ref[site - 1])) + 1) % 4])
            elif take and kind == 1:
                for d in (0, 1):
                    seq[off + d] = ord("ACGT"[("ACGT".index(chr(ref[site - 1 + d])) + 2) % 4])
            elif take and kind == 2:
                dl = 1 + (site // 97) % 5
                ops = [("M", off), ("D", dl), ("M", ln - off)]
            elif take and kind == 3:
                il = 1 + (site // 97) % 4
                seq[off:off] = bytes(rng.choice(list(b"ACGT"), il).astype(np.uint8))
                ops = [("M", off), ("I", il), ("M", ln - off)]
        if rng.random() < 0.2:
            clip = int(rng.integers(1, 6))
            seq = bytearray(rng.choice(list(b"ACGT"), clip).astype(np.uint8).tobytes()) + seq
            ops = [("S", clip)] + ops
        rl = sum(l for o, l in ops if o in "MIS")
        reads.append({"pos": pos, "cigar": ops, "seq": bytes(seq[:rl]), "quals": rng.choice([12, 30, 38], rl, p=[.04, .3, .66]).astype(np.uint8).tolist(),
                      "reverse": bool(rng.integers(0, 2))})
    reads.sort(key=lambda r: r["pos"])
    cfg = _abi.default_config(call_mnvs=call_mnvs, max_mnv_length=3, max_gap_between_mnv=1)
    # batches of ~330 reads; after each one the caller has seen reads up to the last one's position
    batches, ups = [], []
    for k in range(0, len(reads), 330):
        part = reads[k:k + 330]
        batches.append(_abi.ReadBatch(part))
        ups.append(part[-1]["pos"] - 1)
    got, got_alleles, stats = _schedule_run(batches, ref, cfg, ups, dict(PISCES_HIP_READ_PATH=None, **STORE_MODES[mode]))
    want, want_alleles, want_stats = _schedule_run(batches, ref, cfg, ups, dict(PISCES_HIP_READ_PATH="log"))
    assert got.tobytes() == want.tobytes() and got_alleles == want_alleles
    assert stats == want_stats and stats["reads"] == len(reads)
    exp, exp_alleles, total = orc.run_reads_schedule(_abi.ReadBatch(reads), ref, 1, len(ref), cfg, ups)
    assert len(exp) == len(got) > 3000 and exp_alleles == got_alleles
    for f in ("position", "total_coverage", "allele_support", "reference_support", "num_no_calls", "coverage_by_dir", "support_by_dir", "filter_bits",
              "info", "variant_qscore", "genotype_qscore"):
        assert (got[f] == exp[f]).all(), f
    assert stats["TotalNumCalled"] == total


@pytest.mark.parametrize("mode", ["default", "every batch appended to the open segment"])
def test_any_cigar_through_the_fused_kernel_equals_the_log_chain(torch_cuda, mode):
    """The flush kernel takes a read as the FRAGMENTS the shape kernel cuts it into (one per CIGAR operation: aligned runs, and gaps as
    deletion fragments gated by CheckDeletionQuality at the base that closes them, RegionStateManager.cs:131-213); a read whose
    fragments do not fit their fields (a skip of 70 000 positions) goes base by base.  Records over four blocks, flushed in two steps,
    must equal the observation-log chain's (whose read walk is read_walk.h's per-base function) byte for byte."""
    rng = np.random.default_rng(41)
    ref = np.frombuffer(bytes(rng.choice(list(b"ACGT"), 76000).astype(np.uint8)), dtype=np.uint8)
    reads = random_reads(rng, 1500, 20, 3800, exotic=True)
    for k in range(6):   # reads that skip far ahead: one fragment's offset does not fit, the read is walked base by base
        ln = 40 + k
        reads.append({"pos": 300 + 500 * k, "cigar": [("M", 20), ("N", 70000 + k), ("M", ln - 20)],
                      "seq": bytes(rng.choice(list(b"ACGT"), ln).astype(np.uint8)), "quals": [37] * ln, "reverse": bool(k & 1)})
    reads.sort(key=lambda r: r["pos"])
    cfg = _abi.default_config(min_coverage=1, low_depth_filter=1)
    batches = [_abi.ReadBatch(reads[:800]), _abi.ReadBatch(reads[800:])]
    ups = [reads[799]["pos"] - 1, 3000]
    got, got_alleles, stats = _schedule_run(batches, ref, cfg, ups, dict(PISCES_HIP_READ_PATH=None, **STORE_MODES[mode]))
    want, want_alleles, want_stats = _schedule_run(batches, ref, cfg, ups, dict(PISCES_HIP_READ_PATH="log"))
    assert len(got) == len(want) > 3000
    assert got.tobytes() == want.tobytes() and got_alleles == want_alleles and stats == want_stats
    assert (got["position"] > 70000).any()        # the far ends of the skipping reads were called


@pytest.mark.parametrize("shape", ["clusters 8 000 apart", "clusters 20 000 apart", "clusters 70 000 apart", "second batch before the first", "a pile on one position",
                                   "a pile, then a tail"])
@pytest.mark.parametrize("mode", ["default", "every batch appended to the open segment"])
def test_the_position_grid_finds_what_the_search_over_the_whole_segment_finds(torch_cuda, mode, shape):
    """A tile's fragment range starts from the segment's position grid (grid_cells in read_shape_kernel's launch, wave_lower_bound2_hinted) where the segment
    has one: records equal the log chain's when the reads lie in clusters with empty positions between them, when the gap between two
    reads is wider than one lane fills (the segment then goes without a grid), when a later batch starts before the grid's first cell,
    and when thousands of fragments share one cell (more than one narrowing round)."""
    rng = np.random.default_rng(31)
    ref = np.frombuffer(bytes(rng.choice(list(b"ACGT"), 160_000).astype(np.uint8)), dtype=np.uint8)
    if shape.startswith("clusters"):
        step = int(shape.split()[1] + shape.split()[2])   # (8 000 and 20 000: a wave fills the gap; 70 000: wider than kGridGapCells, the segment goes without a grid)
        reads = []
        for k in range(3):
            reads += random_reads(rng, 500, 1000 + k * step, 1400 + k * step, with_dirs=False)
        batches = [reads[:700], reads[700:]]
    elif shape == "second batch before the first":
        batches = [random_reads(rng, 600, 5000, 5600, with_dirs=False), random_reads(rng, 600, 3000, 5300, with_dirs=False)]
    else:
        pile = random_reads(rng, 3000, 2000, 2001, with_dirs=False)
        tail = random_reads(rng, 400, 2001, 2300, with_dirs=False) if "tail" in shape else []
        batches = [pile + tail]
    cfg = _abi.default_config(min_coverage=1, low_depth_filter=1)
    environ = dict(PISCES_HIP_READ_PATH=None, **STORE_MODES[mode])
    got_r, got_a, stats = _schedule_run([_abi.ReadBatch(b) for b in batches], ref, cfg, [None] * len(batches), environ)
    want_r, want_a, want_stats = _schedule_run([_abi.ReadBatch(b) for b in batches], ref, cfg, [None] * len(batches), dict(PISCES_HIP_READ_PATH="log"))
    assert len(got_r) > 300
    assert got_r.tobytes() == want_r.tobytes() and got_a == want_a and stats == want_stats


@pytest.mark.parametrize("gap_op", ["N", "D"])
def test_fragment_behind_a_gap_of_forty_thousand_positions(torch_cuda, gap_op):
    """A later fragment's offset from the read's position sits in the top 16 bits of the descriptor's signed 64-bit field: offsets of
    0x8000 .. 0xFFFF (a skip or a deletion of 33-65 thousand positions inside a read whose span still fits 16 bits) must come out
    unsigned — 20M40000N30M is two aligned runs 40 020 positions apart, not one run 25 516 positions BEFORE the read.  Counts
    (accumulate_store_tiles_kernel) against the oracle, records (call_store_tiles_kernel) against the log chain."""
    from pisces_amd import engine
    rng = np.random.default_rng(77)
    ref = np.frombuffer(bytes(rng.choice(list(b"ACGT"), 66000).astype(np.uint8)), dtype=np.uint8)
    reads = []
    for k in range(40):   # gaps from 32 700 to 65 400 positions: both sides of 0x8000, all inside a 16-bit span
        gap, ln = 32700 + 838 * k, 50
        reads.append({"pos": 100 + 3 * k, "cigar": [("M", 20), (gap_op, gap), ("M", ln - 20)],
                      "seq": bytes(rng.choice(list(b"ACGT"), ln).astype(np.uint8)), "quals": [37] * ln, "reverse": bool(k & 1)})
    reads.sort(key=lambda r: r["pos"])
    assert all(sum(l for o, l in r["cigar"] if o in "MDN") <= 0xFFFF for r in reads)
    exp = oracle_counts(reads, 1, 66000)
    cfg = _abi.default_config(min_coverage=1, low_depth_filter=1)
    with env(PISCES_HIP_READ_PATH=None):
        with engine.HipVariantCaller(cfg) as c:
            c.AddAlleleCounts(_abi.ReadBatch(reads))
            lo = 100 + 20 + 32700 - 5
            got = c.GetCounts(lo, 33000)
    np.testing.assert_array_equal(got.reshape(33000, -1), exp.reshape(66000, -1)[lo - 1:lo - 1 + 33000])
    assert got.sum() > 0
    got_r, _, stats = _schedule_run([_abi.ReadBatch(reads)], ref, cfg, [None], dict(PISCES_HIP_READ_PATH=None))
    want_r, _, want_stats = _schedule_run([_abi.ReadBatch(reads)], ref, cfg, [None], dict(PISCES_HIP_READ_PATH="log"))
    assert got_r.tobytes() == want_r.tobytes() and stats == want_stats
    assert (got_r["position"] > 40000).any()


def test_thresholds_the_fused_kernel_is_not_compiled_for(torch_cuda):
    """minBQ above 127 (the fused kernel compares seven bits) goes through the counts in HBM: same records as the log chain."""
    rng = np.random.default_rng(2)
    ref = np.frombuffer(bytes(rng.choice(list(b"ACGT"), 2300).astype(np.uint8)), dtype=np.uint8)
    reads = []
    for i in range(500):
        pos = int(rng.integers(10, 2000))
        ln = 100
        reads.append({"pos": pos, "cigar": [("M", ln)], "seq": ref[pos - 1:pos - 1 + ln].tobytes(),
                      "quals": rng.choice([100, 160, 250], ln).astype(np.uint8).tolist(), "reverse": bool(i & 1)})
    reads.sort(key=lambda r: r["pos"])
    cfg = _abi.default_config(min_base_call_quality=150, noise_level=20)
    got, _, _ = _schedule_run([_abi.ReadBatch(reads)], ref, cfg, [None], dict(PISCES_HIP_READ_PATH=None))
    want, _, _ = _schedule_run([_abi.ReadBatch(reads)], ref, cfg, [None], dict(PISCES_HIP_READ_PATH="log"))
    assert got.tobytes() == want.tobytes() and len(got) > 2000
    assert (got["num_no_calls"] > 0).any() and (got["total_coverage"] > 0).any()


@pytest.mark.gpu
@pytest.mark.parametrize("n_loci,depth", [(100_000, 500), (128_000, 60), (70_000, 100), (65_000, 40), (120_000, 40), (85_000, 40), (112_000, 40)],
                         ids=["config2_100kx500", "2048_tiles_x60", "1120_tiles_x100", "1040_tiles_groups_of_8", "1920_tiles_groups_of_2",
                              "1360_tiles_groups_of_3_and_cus_left_over", "1792_tiles_no_cu_with_a_tile_more"])
def test_whichever_tile_a_workgroup_takes_the_records_are_the_same(torch_cuda, n_loci, depth):
    """A launch of several tiles a CU deals its tiles by price (store_kernels.hip.h: exchanged_tile inside call_store_tiles_kernel, the
    default; tile_order_kernel in front of it, PISCES_HIP_TILE_ORDER=1): every tile must be taken exactly once, so the records of one
    flush are the bytes of the launch in position order (PISCES_HIP_TILE_ORDER=0).  Sizes: some CUs with one tile more than the
    others (1 600 and 1 120 tiles on 256 CUs: groups of four) and none (2 048, 1 792: groups of four, rotated); 2 / 16 / 10 of an XCD's 32
    CUs with a tile more (1 040 / 1 920 / 1 360 tiles: groups of eight / two / three, the last with CUs that belong to no group)."""
    from pisces_amd import engine, synth
    p = synth.make_pileup(n_loci=n_loci, depth=depth, seed=5)
    ref = p.ref.cpu().numpy()
    cfg = _abi.default_config()
    whole = synth.reads_of(p, p.base.shape[0], first_amplicon=0)
    out = {}
    for order in ("0", "1", "2"):
        with env(PISCES_HIP_TILE_ORDER=order):
            with engine.HipVariantCaller(cfg) as c:
                c.SetReference(ref)
                c.AddAlleleCounts(whole)
                out[order] = c.Call(None, capacity=2 * n_loci)
    assert len(out["0"]) >= n_loci
    assert out["1"].tobytes() == out["0"].tobytes()
    assert out["2"].tobytes() == out["0"].tobytes()


def test_tiles_of_any_size_and_switches_of_the_add_give_the_same_records(torch_cuda):
    """The tiles of a flush over whole blocks are described by value (RegularTiles) and may be of any size up to 64
    (PISCES_HIP_TILE_LOCI: 59 -> 17 tiles a block, 48 -> 21, 33 -> 31; the default 64 -> 16): a tile's records depend on its loci only, so
    the compacted rows of the flush are the same bytes.  The same for the switches round 6 added to the add of a batch in device memory:
    the position grid enqueued by the add itself (PISCES_HIP_DEFER_GRID=0) and the read role spread between the streaming workgroups
    (PISCES_HIP_ROLE_STRIDE=3)."""
    from pisces_amd import engine, synth
    p = synth.make_pileup(n_loci=40_000, depth=80, seed=17, device="cuda", with_tuples=False)
    ref = p.ref.cpu().numpy()
    cfg = _abi.default_config()
    whole = synth.reads_of(p, p.base.shape[0], first_amplicon=0)
    d = engine.DeviceReadBatch.from_host(whole, "cuda:0")
    out = {}
    for label, kw in (("default", {}), ("59", dict(PISCES_HIP_TILE_LOCI="59")), ("48", dict(PISCES_HIP_TILE_LOCI="48")), ("33", dict(PISCES_HIP_TILE_LOCI="33")),
                      ("grid_in_the_add", dict(PISCES_HIP_DEFER_GRID="0")), ("stride_3", dict(PISCES_HIP_ROLE_STRIDE="3"))):
        with env(**kw):
            with engine.HipVariantCaller(cfg) as c:
                c.SetReference(ref)
                c.AddDeviceReads(d)
                first = c.Call(20_000, capacity=80_000)      # (a flush, an add behind it, the final flush: the grid of the second batch extends the first's)
                out[label] = np.concatenate([first, c.Call(None, capacity=80_000)])
    assert len(out["default"]) >= 40_000
    for label, rows in out.items():
        assert rows.tobytes() == out["default"].tobytes(), label
    with engine.HipVariantCaller(cfg) as c:   # and the host-fed add gives them too
        c.SetReference(ref)
        c.AddAlleleCounts(whole)
        assert c.Call(None, capacity=80_000).tobytes() == out["default"].tobytes()


@pytest.mark.parametrize("n_loci,depth", [(100_000, 500)], ids=["config2_100kx500"])
def test_config2_at_full_size_store_equals_log_chain_and_oracle_slice(torch_cuda, n_loci, depth):
    """BASELINE config 2 (100 000 loci x 500x, 333 500 reads) through pisces_hip_add_reads / pisces_hip_flush: the read store and the
    observation-log chain give the same bytes, in one batch and block by block; the first 2 000 loci equal the oracle."""
    from pisces_amd import engine, synth
    p = synth.make_pileup(n_loci=n_loci, depth=depth, seed=11)
    ref = p.ref.cpu().numpy()
    cfg = _abi.default_config()
    A = p.base.shape[0]
    whole = synth.reads_of(p, A, first_amplicon=0)
    out = {}
    for path in ("store", "log"):
        with env(PISCES_HIP_READ_PATH=None if path == "store" else "log"):
            with engine.HipVariantCaller(cfg) as c:
                c.SetReference(ref)
                c.AddAlleleCounts(whole)
                one = c.Call(None, capacity=2 * n_loci)
                parts = []
                for a0 in range(0, A, 7):     # ~ one 1000-locus block of reads per call, as SmallVariantCaller feeds them
                    c.AddAlleleCounts(synth.reads_of(p, 7, first_amplicon=a0))
                    parts.append(c.Call(p.region_start + a0 * synth.READ_LEN - 1, capacity=1 << 13))
                parts.append(c.Call(None, capacity=1 << 13))
                out[path] = (one, np.concatenate(parts))
    assert out["store"][0].tobytes() == out["log"][0].tobytes() and len(out["store"][0]) >= n_loci
    assert out["store"][1].tobytes() == out["log"][1].tobytes()
    assert out["store"][0].tobytes() == out["store"][1].tobytes()
    head = synth.reads_of(p, 14, first_amplicon=0)
    exp, _ = orc.run_reads(head, ref, p.region_start, 2000, cfg)
    got = out["store"][0]
    got = got[got["position"] < p.region_start + 2000]
    assert got.tobytes() == exp.tobytes()


def test_config3_mix_store_equals_log_chain(torch_cuda):
    """BASELINE config 3's mix (SNVs, MNVs, deletions, insertions at 2000x, MNV calling on) over 30 amplicons, block by block: the read
    store and the log chain give the same records and allele strings (the full size runs in test_gpu_parity.py; this is the A / B)."""
    from pisces_amd import synth
    seed, depth, n_amp = 33, 2000, 30
    cfg = _abi.default_config(call_mnvs=1, max_mnv_length=3, max_gap_between_mnv=1)
    n_loci = n_amp * synth.READ_LEN
    ref = synth.reference_of(n_loci, seed, device="cuda")
    p = synth.make_pileup(n_loci, depth, seed=seed, device="cuda", first_locus=0, total_loci=n_loci, with_tuples=False)
    batch, planted = synth.mixed_reads(p, seed)
    ups = [1000, 2000, 3000, 4000]
    got, got_alleles, stats = _schedule_run([batch] + [_abi.ReadBatch([])] * 3, ref, cfg, ups, dict(PISCES_HIP_READ_PATH=None))
    want, want_alleles, want_stats = _schedule_run([batch] + [_abi.ReadBatch([])] * 3, ref, cfg, ups, dict(PISCES_HIP_READ_PATH="log"))
    assert got.tobytes() == want.tobytes() and got_alleles == want_alleles and stats == want_stats
    cats = (got["info"] >> 4) & 7
    assert len(planted) >= 5 and (cats == _abi.CAT_MNV).any() and (cats == _abi.CAT_DELETION).any() and (cats == _abi.CAT_INSERTION).any()


@pytest.mark.parametrize("collapse", [1, 0], ids=["open ends tracked", "open ends not part of the identity"])
def test_candidates_merged_on_the_device_equal_the_host_merge(torch_cuda, collapse):
    """RegionState.AddCandidate for the records of a batch on the device (found_merge_kernel / found_gather_kernel) against the host merge
    of one record per read event (PISCES_HIP_DEVICE_MERGE=1 / =0): the candidates the state holds (support and well-anchored support by
    direction, open ends, order of first arrival) and, after the flushes, records, allele strings and totals — on BASELINE config 3's
    mix (hundreds of reads per planted variant, thousands of single-read error candidates) and on random reads with long insertions
    (ALT alleles beyond the inline 32 bases live in the byte pool), with the collapser on (open ends are part of a candidate's identity)
    and off (a merged candidate keeps the open ends of its first arrival)."""
    from pisces_amd import engine, synth
    seed, depth, n_amp = 35, 1500, 14
    cfg = _abi.default_config(call_mnvs=1, max_mnv_length=3, max_gap_between_mnv=1, collapse=collapse)
    n_loci = n_amp * synth.READ_LEN
    ref = synth.reference_of(n_loci, seed, device="cuda")
    p = synth.make_pileup(n_loci, depth, seed=seed, device="cuda", first_locus=0, total_loci=n_loci, with_tuples=False)
    batch, planted = synth.mixed_reads(p, seed)
    rng = np.random.default_rng(5)
    refb = ref.cpu().numpy() if hasattr(ref, "cpu") else np.asarray(ref)
    extra = []
    for i in range(400):   # reads with an insertion of 20-50 bases (a few distinct ones, so that they merge) and soft clips
        pos = 50 + 61 * int(rng.integers(0, 20))
        ins = bytes(np.random.default_rng(int(rng.integers(0, 3))).choice(list(b"ACGT"), int(rng.integers(0, 2)) * 30 + 20).astype(np.uint8))
        left, right = 40, int(rng.integers(20, 70))
        seq = bytes(refb[pos - 1:pos - 1 + left]) + ins + bytes(refb[pos - 1 + left:pos - 1 + left + right])
        extra.append({"pos": pos, "cigar": [("M", left), ("I", len(ins)), ("M", right)], "seq": seq, "quals": [37] * len(seq), "reverse": bool(i % 2)})
    extra.sort(key=lambda r: r["pos"])
    extra = _abi.ReadBatch(extra)
    ups = [1000, None]
    results = []
    for merge in (1, 0):
        with env(PISCES_HIP_DEVICE_MERGE=merge):
            with engine.HipVariantCaller(cfg) as c:
                c.SetReference(ref)
                c.AddAlleleCounts(batch)
                c.AddAlleleCounts(extra)                      # (a second batch: its groups merge into the candidates of the first)
                cands = c.GetCandidates(None)
                recs, alleles = [], []
                for up in ups:
                    r, a = c.CallWithAlleles(up, capacity=1 << 14)
                    recs.append(r)
                    alleles += a
                pcie = c.TransferBytes()
                results.append((cands, np.concatenate(recs), alleles, c.Stats(), pcie))
    (cm, rm, am, sm, pm), (ch, rh, ah, sh, ph) = results
    assert len(cm) > 1000 and cm == ch                       # same candidates, same order, same sums
    assert any(len(c["alt"]) > 33 for c in cm) and any(sum(c["support_by_dir"]) > 100 for c in cm)
    assert rm.tobytes() == rh.tobytes() and am == ah and sm == sh
    assert pm["d2h_candidates"] < ph["d2h_candidates"]      # fewer bytes back: one record a candidate, not one a read event


def test_config5_depth_through_the_read_store_equals_log_chain_and_oracle(torch_cuda):
    """BASELINE config 5's settings (-minbq 30 => NL 30, -minvf 0.005, -sbfilter 0.5, -vqfilter 30, gVCF) at its depth (5000x, planted
    0.5 % VAF SNVs every 50 loci) from READS: 6 000 loci = 200 000 reads through pisces_hip_add_reads / pisces_hip_flush block by block —
    ~7 000 reads a tile in the fused kernel, counts of 5 000 a cell, a quality threshold above the default — against the observation-log
    chain (same bytes) and the oracle (first 1 000 loci)."""
    from pisces_amd import engine, synth
    n_loci, depth = 6000, 5000
    p = synth.make_pileup(n_loci=n_loci, depth=depth, seed=23, vaf_range=(0.005, 0.005), snv_every=50, snv_offset=17, q_lo=12)
    ref = p.ref.cpu().numpy()
    cfg = _abi.default_config(min_base_call_quality=30, noise_level=30, min_frequency=0.005, variant_freq_filter=0.005,
                              genotype_min_freq_filter=0.005, target_lod_frequency=0.005, strand_bias_threshold=0.5, variant_qscore_filter=30)
    A = p.base.shape[0]
    out = {}
    for path in ("store", "log"):
        with env(PISCES_HIP_READ_PATH=None if path == "store" else "log"):
            with engine.HipVariantCaller(cfg) as c:
                c.SetReference(ref)
                parts = []
                for a0 in range(0, A, 7):
                    c.AddAlleleCounts(synth.reads_of(p, min(7, A - a0), first_amplicon=a0))
                    parts.append(c.Call(p.region_start + a0 * synth.READ_LEN - 1, capacity=1 << 13))
                parts.append(c.Call(None, capacity=1 << 13))
                out[path] = np.concatenate(parts)
    got = out["store"]
    assert got.tobytes() == out["log"].tobytes() and len(got) >= n_loci
    assert (got["total_coverage"] + got["num_no_calls"]).max() == depth
    cats = (got["info"] >> 4) & 7
    assert (cats == _abi.CAT_SNV).sum() >= 20                     # about half of the 120 planted sites clear 0.5 %
    head = synth.reads_of(p, 7, first_amplicon=0)
    exp, _ = orc.run_reads(head, ref, p.region_start, 1000, cfg)
    assert got[got["position"] < p.region_start + 1000].tobytes() == exp.tobytes()


# ---- reads handed over in device memory (pisces_hip_add_device_reads) and the checks of a batch made on the device ----------------------
@pytest.mark.parametrize("call_mnvs", [0, 1], ids=["snv_indel", "mnv"])
@pytest.mark.parametrize("mode", ["default", "every batch appended to the open segment"])
def test_reads_handed_over_in_device_memory(torch_cuda, mode, call_mnvs):
    """pisces_hip_add_device_reads == pisces_hip_add_reads: the same reads (arbitrary CIGARs, per-base directions, N bases, low qualities),
    in three batches with a flush between them, once from host arrays and once from tensors in device memory — the records, the allele
    strings and the totals are the same bytes; so are they when a host batch is checked on the device (PISCES_HIP_DEVICE_CHECKS=1:
    read_prepare_kernel instead of the host's pass over the CIGARs)."""
    from pisces_amd import engine
    rng = np.random.default_rng(91 + call_mnvs)
    ref = np.frombuffer(bytes(rng.choice(list(b"ACGT"), 4600).astype(np.uint8)), dtype=np.uint8)
    reads = random_reads(rng, 1800, 30, 4100, exotic=False)
    for r in reads[::3]:   # reads that look like their reference (with a few mismatches), so that calls are made
        ln = len(r["seq"])
        if r["cigar"] == [("M", ln)]:
            seq = bytearray(ref[r["pos"] - 1:r["pos"] - 1 + ln].tobytes())
            for k in range(ln):
                if rng.random() < 0.02:
                    seq[k] = int(rng.choice(list(b"ACGT")))
            r["seq"] = bytes(seq)
    cfg = _abi.default_config(min_coverage=1, low_depth_filter=1, call_mnvs=call_mnvs)
    cuts = [0, 700, 1300, len(reads)]
    ups = [reads[699]["pos"] - 1, reads[1299]["pos"] - 1, None]

    def run(how, environ):
        recs, alleles = [], []
        with env(PISCES_HIP_READ_PATH=None, **environ):
            with engine.HipVariantCaller(cfg) as c:
                c.SetReference(ref)
                for k, up in enumerate(ups):
                    batch = _abi.ReadBatch(reads[cuts[k]:cuts[k + 1]])
                    (c.AddDeviceReads if how == "device" else c.AddAlleleCounts)(batch)
                    r, a = c.CallWithAlleles(up, capacity=1 << 15)
                    recs.append(r)
                    alleles += a
                stats = c.Stats()
        return np.concatenate(recs), alleles, stats
    want = run("host", dict(PISCES_HIP_DEVICE_CHECKS=0, **STORE_MODES[mode]))
    assert len(want[0]) > 3000
    for how, environ in (("device", STORE_MODES[mode]), ("host", dict(PISCES_HIP_DEVICE_CHECKS=1, **STORE_MODES[mode]))):
        got = run(how, environ)
        assert got[0].tobytes() == want[0].tobytes() and got[1] == want[1] and got[2] == want[2], (how, environ)


def test_checks_on_the_device_refuse_what_the_host_pass_refuses(torch_cuda):
    """A batch in device memory (and a host batch with PISCES_HIP_DEVICE_CHECKS=1) is refused as a whole for the reasons the host's pass over
    the CIGARs refuses one — position <= 0 (RegionStateManager.cs:363-364), a CIGAR that does not span the read (Read.ValidateCigar,
    Read.cs:603-605), a read past 2^31 - 1, a per-base direction or a deletion direction that is no DirectionType — with the same code and
    message, the state untouched; the reads in front of the bad one included."""
    from pisces_amd import engine
    good = {"pos": 20, "seq": "ACGTACGT", "cigar": [("M", 8)], "quals": [30] * 8, "reverse": False}
    bads = (({"pos": 0, "seq": "ACGT", "cigar": [("M", 4)], "quals": [30] * 4, "reverse": False}, "greater than 0"),
            ({"pos": 9, "seq": "ACGT", "cigar": [("M", 3), ("I", 4)], "quals": [30] * 4, "reverse": False}, "CIGAR does not match"),
            ({"pos": 9, "seq": "ACGT", "cigar": [("M", 2)], "quals": [30] * 4, "reverse": False}, "CIGAR does not match"),
            ({"pos": 2 ** 31 - 3, "seq": "ACGT", "cigar": [("M", 4)], "quals": [30] * 4, "reverse": False}, "2^31"),
            ({"pos": 9, "seq": "ACGT", "cigar": [("M", 4)], "quals": [30] * 4, "reverse": False, "dirs": [0, 1, 3, 0]}, "CIGAR does not match"),
            ({"pos": 9, "seq": "ACGT", "cigar": [("M", 2), ("D", 2), ("M", 2)], "quals": [30] * 4, "reverse": False, "del_dirs": [(255, 255), (7, 0), (255, 255)]},
             "deletion_directions"))
    for how in ("device", "host checked on the device"):
        with env(PISCES_HIP_READ_PATH=None, PISCES_HIP_DEVICE_CHECKS=1):
            with engine.HipVariantCaller() as c:
                add = c.AddDeviceReads if how == "device" else c.AddAlleleCounts
                for bad, needle in bads:
                    before = c.Stats()
                    with pytest.raises(engine.PiscesHipError) as e:
                        add(_abi.ReadBatch([good, bad]))
                    assert e.value.code == _abi.E_INVALID_ARG and needle in e.value.message, (how, needle, e.value.message)
                    assert c.Stats() == before and c.GetCounts(20, 8).sum() == 0
                add(_abi.ReadBatch([good]))
                assert c.GetCounts(20, 8).sum() == 8 and c.Stats()["reads"] == 1
                # two terminal cases of the block bookkeeping: a read across a block edge, and one far away
                add(_abi.ReadBatch([{"pos": 996, "seq": "ACGTACGT", "cigar": [("M", 8)], "quals": [30] * 8, "reverse": True},
                                    {"pos": 5_000_001, "seq": "ACGT", "cigar": [("M", 4)], "quals": [30] * 4, "reverse": False}]))
                assert c.GetCounts(996, 8).sum() == 8 and c.GetCounts(5_000_001, 4).sum() == 4 and c.Stats()["reads"] == 3


@pytest.mark.parametrize("how", ["device", "host_pass", "host_checked_on_the_device"])
def test_one_bad_read_in_a_large_plain_batch_is_found_on_every_path(torch_cuda, how):
    """70 000 plain <8>M reads — a batch that becomes a segment of its own (one launch makes its checks, descriptors, fragments and row
    codes: add_fused_kernel) and, from the host, a batch the bulk loops of add_reads establish as plain — with ONE bad read somewhere:
    a position of 0 in the middle, a CIGAR one base short near the end, a per-base direction of 3 in the batch's last bytes (checked by
    the stream role, whose verdict the host waits for apart).  Every path refuses the batch with the reference's message and leaves the
    handle as it was; the good batch then goes in and is counted."""
    from pisces_amd import engine
    n, L = 70_000, 8
    pos = (20 + 2 * np.arange(n)).astype(np.int32)

    def batch(bad=None):
        position, clen = pos.copy(), np.full(n, L, np.uint32)
        dirs = None
        if bad == "position":
            position[n // 2] = 0
        elif bad == "cigar":
            clen[n - 7] = L - 1
        elif bad == "direction":
            dirs = np.zeros(n * L, np.uint8)
            dirs[n * L - 3] = 3
        return _abi.ReadBatch.from_arrays(position, np.zeros(n, np.uint8), np.arange(n + 1, dtype=np.int32), np.full(n, ord("M"), np.uint8), clen,
                                          L * np.arange(n + 1, dtype=np.int32), np.tile(np.frombuffer(b"ACGTACGT", np.uint8), n), np.full(n * L, 30, np.uint8),
                                          directions=dirs)
    kw = {"device": {}, "host_pass": dict(PISCES_HIP_DEVICE_CHECKS="0"), "host_checked_on_the_device": dict(PISCES_HIP_DEVICE_CHECKS="1")}[how]
    with env(PISCES_HIP_READ_PATH=None, **kw):
        with engine.HipVariantCaller() as c:
            add = (lambda b: c.AddDeviceReads(engine.DeviceReadBatch.from_host(b, "cuda:0"))) if how == "device" else c.AddAlleleCounts
            for bad, needle in (("position", "greater than 0"), ("cigar", "CIGAR does not match"), ("direction", "CIGAR does not match")):
                before = c.Stats()
                with pytest.raises(engine.PiscesHipError) as e:
                    add(batch(bad))
                assert e.value.code == _abi.E_INVALID_ARG and needle in e.value.message, (how, bad, e.value.message)
                assert c.Stats() == before and c.GetCounts(20, 8).sum() == 0
            add(batch())
            assert c.Stats()["reads"] == n
            counts = c.GetCounts(int(pos[n // 2]), 2)
            assert counts.sum() == 2 * (L // 2)      # (reads two positions apart, eight bases long: four cover a position)


@pytest.mark.parametrize("call_mnvs", [0, 1], ids=["snv_indel", "mnv"])
def test_reads_added_ahead_of_the_flush_that_clears_what_lies_behind_them(torch_cuda, call_mnvs):
    """SmallVariantCaller adds a read and THEN calls up to its position - 1 (SmallVariantCaller.cs:88-105).  Stretch by stretch that is:
    add the reads of stretch k + 1, flush up to their first position - 1.  Such a flush does not wait for the candidates of the stretch
    just added (every read of it starts behind upTo), so the device discovers them under the flush's host work — and the records must be
    the ones the add-then-flush order gives: BASELINE config 3's mix (SNVs, MNVs, insertions, deletions at 2000x), six stretches."""
    from pisces_amd import engine, synth
    seed, depth, n_amp, per = 37, 1200, 24, 4
    cfg = _abi.default_config(call_mnvs=call_mnvs, max_mnv_length=3, max_gap_between_mnv=1)
    n_loci = n_amp * synth.READ_LEN
    ref = synth.reference_of(n_loci, seed, device="cuda")
    origin = synth.READ_LEN + 1
    stretches = []
    for a0 in range(0, n_amp, per):
        p = synth.make_pileup(per * synth.READ_LEN, depth, seed=seed, device="cuda", first_locus=a0 * synth.READ_LEN, total_loci=n_loci, with_tuples=False)
        stretches.append((synth.mixed_reads(p, seed)[0], origin + (a0 + per) * synth.READ_LEN - 1))

    def run(ahead, device_fed):
        recs, alleles = [], []
        with engine.HipVariantCaller(cfg) as c:
            c.SetReference(ref)
            add = c.AddDeviceReads if device_fed else c.AddAlleleCounts
            prev = None
            for batch, up_to in stretches:
                if not ahead and prev is not None:
                    pass
                add(batch)
                if ahead:
                    if prev is not None:
                        r, a = c.CallWithAlleles(prev, capacity=1 << 15)
                        recs.append(r); alleles += a
                    prev = up_to
                else:
                    r, a = c.CallWithAlleles(up_to, capacity=1 << 15)
                    recs.append(r); alleles += a
            r, a = c.CallWithAlleles(None, capacity=1 << 15)
            recs.append(r); alleles += a
            return np.concatenate(recs), alleles, c.Stats()
    want = run(False, False)
    assert len(want[0]) >= n_loci
    for ahead, device_fed in ((True, False), (True, True)):
        got = run(ahead, device_fed)
        assert got[0].tobytes() == want[0].tobytes() and got[1] == want[1] and got[2] == want[2], (ahead, device_fed)


@pytest.mark.parametrize("lead", ["I", "D"])
@pytest.mark.parametrize("device_fed", [False, True], ids=["host reads", "device reads"])
def test_a_leading_indel_one_past_a_block_edge_is_waited_for(torch_cuda, lead, device_fed):
    """A read whose first operation is I or D puts its candidate at position - 1 (CandidateVariantFinder.cs:52-76 through the walk's
    `coordinate = in_ref`): with reads at 1001 that is position 1000, the last of block 1.  The flush up to 1000 that follows the add
    (SmallVariantCaller.cs:88-105: add, then Call(position - 1)) must wait for that batch's candidates although every read of it starts
    behind upTo — it skipped them while it compared the batch's lowest READ position with upTo, cleared block 1 without
    the indel and re-created the block afterwards."""
    from pisces_amd import engine
    rng = np.random.default_rng(77)
    ref = rng.choice(np.frombuffer(b"ACGT", np.uint8), 2300)
    seq_at = lambda p, n: bytes(ref[p - 1: p - 1 + n])
    first = [dict(pos=int(p), cigar=[("M", 70)], seq=seq_at(int(p), 70), quals=bytes([35] * 70), reverse=bool(i & 1))
             for i, p in enumerate(np.sort(rng.integers(905, 960, 60)))]
    second = []
    for i in range(60):
        if i % 3 == 2:   # plain reads of the same stretch
            second.append(dict(pos=1001 + i % 5, cigar=[("M", 70)], seq=seq_at(1001 + i % 5, 70), quals=bytes([35] * 70), reverse=bool(i & 1)))
        elif lead == "I":
            second.append(dict(pos=1001, cigar=[("I", 3), ("M", 67)], seq=b"TTG" + seq_at(1001, 67), quals=bytes([35] * 70), reverse=bool(i & 1)))
        else:
            second.append(dict(pos=1001, cigar=[("D", 2), ("M", 70)], seq=seq_at(1003, 70), quals=bytes([35] * 70), reverse=bool(i & 1)))
    up_to = min(r["pos"] for r in second) - 1
    cfg = _abi.default_config()
    want, want_alleles, want_called = orc.run_reads_schedule(_abi.ReadBatch(first + second), ref, 1, len(ref), cfg, [up_to])
    assert [int(r["position"]) for r, a in zip(want, want_alleles) if len(a[0]) != len(a[1])] == [1000], "the oracle calls the leading indel"
    with engine.HipVariantCaller(cfg) as c:
        c.SetReference(ref)
        add = c.AddDeviceReads if device_fed else c.AddAlleleCounts
        add(_abi.ReadBatch(first))
        add(_abi.ReadBatch(second))
        r1, a1 = c.CallWithAlleles(up_to)
        r2, a2 = c.CallWithAlleles(None)
        got, got_alleles = np.concatenate([r1, r2]), a1 + a2
        assert got_alleles == want_alleles
        assert got.tobytes() == want.tobytes()
        assert c.Stats()["TotalNumCalled"] == want_called


@pytest.mark.parametrize("gvcf", [1, 0], ids=["gvcf", "variants only"])
def test_total_num_called_counts_the_callable_snvs_outside_the_intervals_when_asked(torch_cuda, gvcf):
    """AlleleCaller.Call counts an allele in IsCallable, before ShouldReport (AlleleCaller.cs:109-131): with an interval set the reference's
    TotalNumCalled includes the callable SNVs of loci the reads overhang the intervals by.  The library evaluates the intervals' loci only
    unless pisces_hip_set_exact_total_called asks for the reference's number (a counting launch over the off-interval loci of the
    flushed blocks).  Rows are the oracle's either way; the total is the oracle's with the switch on and smaller without."""
    from pisces_amd import engine
    rng = np.random.default_rng(99)
    ref = rng.choice(np.frombuffer(b"ACGT", np.uint8), 3300)
    intervals = [(200, 420), (1100, 1180), (2050, 2300)]
    reads = []
    alt_of = {ord("A"): "C", ord("C"): "G", ord("G"): "T", ord("T"): "A"}
    snv_at = set(range(150, 2400, 37))        # planted SNVs inside and outside the intervals
    for i, start in enumerate(np.sort(rng.integers(100, 2350, 900))):
        seq = bytearray(ref[start - 1: start - 1 + 100].tobytes())
        if i % 2:
            for p in range(start, start + 100):
                if p in snv_at:
                    seq[p - start] = ord(alt_of[seq[p - start]])
        reads.append(dict(pos=int(start), cigar=[("M", 100)], seq=bytes(seq), quals=bytes([35] * 100), reverse=bool(i & 2)))
    batch = _abi.ReadBatch(reads)
    cfg = _abi.default_config(include_reference_calls=gvcf)
    for schedule in ([], [1000, 2000]):
        want, want_alleles, want_called = orc.run_reads_schedule(batch, ref, 1, len(ref), cfg, schedule, intervals=intervals)
        totals = {}
        for exact in (False, True):
            with engine.HipVariantCaller(cfg) as c:
                c.SetReference(ref)
                c.SetIntervals(intervals)
                c.SetExactTotalNumCalled(exact)
                c.AddAlleleCounts(batch)
                rows = [c.CallWithAlleles(up) for up in schedule] + [c.CallWithAlleles(None)]
                got = np.concatenate([r for r, _ in rows])
                got_alleles = [a for _, al in rows for a in al]
                assert got_alleles == want_alleles and got.tobytes() == want.tobytes()
                totals[exact] = c.Stats()["TotalNumCalled"]
        assert totals[True] == want_called, (schedule, totals, want_called)
        assert totals[False] < want_called


@pytest.mark.gpu
@pytest.mark.parametrize("gvcf", [1, 0], ids=["gvcf", "variants only"])
def test_candidate_rows_merged_in_place_equal_the_rows_merged_by_copy(torch_cuda, gvcf):
    """The rows of the candidate kernel (insertions, deletions) join the tile kernels' rows inside the pinned download buffer
    (AlleleCaller.cs:146-147, :172-176: the Reference row of the position goes, the rows stay in position / ref / alt order); the A / B is
    the merge into a vector of its own (PISCES_HIP_MERGE_IN_PLACE=0).  A gVCF (a Reference row goes for nearly every row that comes: the rows
    shift by one for a position or two) and variants only (nothing goes: every row behind the first insertion moves), rows read as a view
    and through the caller's arrays, a handle made from what the last one left behind (pisces_hip_trim_memory gives that back)."""
    from pisces_amd import _native, engine, synth
    seed, depth, n_amp = 41, 300, 40
    cfg = _abi.default_config(emit_zero_coverage_refs=1) if gvcf else _abi.default_config(include_reference_calls=0)
    n_loci = n_amp * synth.READ_LEN
    ref = synth.reference_of(n_loci, seed, device="cuda")
    p = synth.make_pileup(n_loci, depth, seed=seed, device="cuda", first_locus=0, total_loci=n_loci, with_tuples=False)
    batch, planted = synth.mixed_reads(p, seed, kinds="DI")
    out = {}
    for in_place in (1, 0):
        with env(PISCES_HIP_MERGE_IN_PLACE=in_place):
            for how in ("view", "arrays"):
                with engine.HipVariantCaller(cfg) as c:
                    c.SetReference(ref)
                    c.AddAlleleCounts(batch)
                    rows = []
                    alleles = []
                    for up in (3000, None):
                        if how == "view":
                            rows.append(c.CallView(up).copy())
                        else:
                            r, a = c.CallWithAlleles(up, capacity=1 << 15)
                            rows.append(r)
                            alleles += a
                    out[(in_place, how)] = (np.concatenate(rows), alleles, c.Stats())
    want = out[(0, "arrays")]
    cats = (want[0]["info"] >> 4) & 7
    assert (cats == _abi.CAT_DELETION).sum() >= 3 and (cats == _abi.CAT_INSERTION).sum() >= 3 and len(want[1]) >= 6
    assert gvcf == int((cats == _abi.CAT_REFERENCE).any())
    for key, got in out.items():
        assert got[0].tobytes() == want[0].tobytes() and got[2] == want[2], key
        if key[1] == "arrays":
            assert got[1] == want[1], key
    assert _native.lib.pisces_hip_trim_memory() > 0 and _native.lib.pisces_hip_trim_memory() == 0


@pytest.mark.gpu
@pytest.mark.parametrize("form", ["events", "bases", "wave", "batch"])
def test_every_form_of_the_candidate_walk_equals_the_host_walk(torch_cuda, form):
    """ICandidateVariantFinder.FindCandidates (CandidateVariantFinder.cs:31-203) in the forms finder_kernels.hip.h holds — a lane a read
    with the events first (the default), a lane a read base by base, a wave a read, a wave for 64 reads (PISCES_HIP_FINDER) — against the
    host's base-by-base walk (pisces_hip_find_candidates, finder_walk.h): reads with any CIGAR, N bases, bytes that are no bases, low
    qualities, qualities above 127, M operations shorter than a word and longer than a chunk of 64 bases, MNV limits from 1 to 6 bases
    and gaps from 0 to 3."""
    from pisces_amd import engine
    rng = np.random.default_rng(77)
    ref = bytes(rng.choice(list(b"ACGTN"), 3000, p=[.245, .245, .245, .245, .02]).astype(np.uint8))
    reads = random_reads(rng, 600, 1, 2600, exotic=True)
    for i in range(60):   # long aligned runs that differ from the reference here and there (MNVs with gaps, runs that hit the length limit)
        pos = int(rng.integers(1, 2500))
        n = int(rng.integers(130, 400))
        seq = bytearray(ref[pos - 1:pos - 1 + n])
        for k in rng.integers(0, len(seq), 12):
            for j in range(int(rng.integers(1, 5))):
                if k + 2 * j < len(seq):
                    seq[k + 2 * j] = int(rng.choice(list(b"ACGT")))
        reads.append({"pos": pos, "cigar": [("M", len(seq))], "seq": bytes(seq), "quals": rng.choice([12, 30, 37], len(seq), p=[.05, .15, .8]).astype(np.uint8).tolist(),
                      "reverse": bool(i % 2)})
    reads.sort(key=lambda r: r["pos"])
    batch = _abi.ReadBatch(reads)
    with env(PISCES_HIP_FINDER=None if form == "events" else form):
        with engine.HipVariantCaller(_abi.default_config()) as c:
            c.SetReference(ref)
            for call_mnvs, max_len, max_gap in [(0, 3, 1), (1, 3, 1), (1, 1, 0), (1, 6, 3), (1, 2, 2)]:
                want = engine.find_candidates(batch, ref, 20, True, bool(call_mnvs), max_len, max_gap)
                got = c.FindCandidates(batch, True, bool(call_mnvs), max_len, max_gap)
                assert len(want) > 2000 and got == want, (form, call_mnvs, max_len, max_gap)


@pytest.mark.gpu
def test_a_batch_that_touches_more_blocks_than_the_device_lists(torch_cuda):
    """The touched blocks of a batch that was checked on the device come back as a list of keys the kernel writes (8 192 of them at most);
    a batch that touches more makes the host read the block map's words itself.  9 000 reads, each in a block of its own, one of them
    across a block edge: the same records and totals as with the host's pass over the CIGARs (RegionStateManager.cs:118-220, :385-391)."""
    from pisces_amd import engine
    rng = np.random.default_rng(9)
    n = 9000
    ref = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, n * 1000 + 2000)]
    reads = []
    for i in range(n):
        pos = 1000 * i + (996 if i == 4000 else int(rng.integers(5, 900)))
        seq = bytearray(ref[pos - 1:pos - 1 + 12].tobytes())
        if i % 3 == 0:
            seq[5] = ord("A") if seq[5] != ord("A") else ord("C")
        reads.append({"pos": pos, "seq": bytes(seq), "cigar": [("M", 12)], "quals": [35] * 12, "reverse": bool(i & 1)})
    batch = _abi.ReadBatch(reads)
    cfg = _abi.default_config(min_coverage=1, include_reference_calls=0)
    out = []
    for checks in (1, 0):
        with env(PISCES_HIP_DEVICE_CHECKS=checks):
            with engine.HipVariantCaller(cfg) as c:
                c.SetReference(ref)
                c.AddAlleleCounts(batch)
                out.append((c.Call(None, capacity=1 << 16), c.Stats()))
    (got, gs), (want, ws) = out
    assert len(want) == 3000 and got.tobytes() == want.tobytes() and gs == ws


def _eqx_reads(rng, ref, n=2400, lo=60, hi=2900):
    """Reads an aligner wrote with = and X operations (minimap2 --eqx) among reads with M operations, over shared mismatch sites: sites
    only X operations show, sites both kinds of reads show, = operations whose bases differ from the reference after all, low qualities
    and N bases under X, an insertion or a deletion between = runs."""
    reads = []
    for i in range(n):
        pos = int(rng.integers(lo, hi))
        ln = int(rng.integers(50, 140))
        seq = bytearray(ref[pos - 1:pos - 1 + ln])
        site = pos + 6 + (40 - (pos + 6)) % 61   # one shared site every 61 positions: the first one the read reaches
        off = site - pos
        kind = (site // 61) % 3             # 0: X reads only, 1: M reads only, 2: both
        eqx = i % 2 == 0
        mism = []
        if 6 <= off < ln - 6 and rng.random() < 0.45 and (kind == 2 or (kind == 0) == eqx):
            seq[off] = ord("ACGT"[("ACGT".index(chr(ref[site - 1])) + 1 + (i % 7 == 0)) % 4])
            mism.append(off)
        for k in range(ln):                 # sequencing errors
            if rng.random() < 0.004 and k not in mism:
                seq[k] = int(rng.choice(list(b"ACGTN")))
                if seq[k] != ref[pos - 1 + k]:
                    mism.append(k)
        quals = rng.choice([12, 30, 38], ln, p=[.05, .3, .65]).astype(np.uint8)
        if not eqx:
            ops = [("M", ln)]
        else:
            hide = i % 10 == 0              # this read's = runs are not checked against the reference: a mismatch stays inside one
            ops, run = [], 0
            for k in range(ln):
                if k in mism and not hide:
                    if run:
                        ops.append(("=", run))
                    if ops and ops[-1][0] == "X":
                        ops[-1] = ("X", ops[-1][1] + 1)
                    else:
                        ops.append(("X", 1))
                    run = 0
                else:
                    run += 1
            if run:
                ops.append(("=", run))
        seq = bytes(seq)
        if i % 9 == 4 and ln > 40:          # an insertion / a deletion in the middle (of either kind of read)
            cut, k = ln // 2, 1 + i % 3
            head, tail, at = [], [], 0
            for o, l in ops:
                if at + l <= cut:
                    head.append((o, l))
                elif at >= cut:
                    tail.append((o, l))
                else:
                    head.append((o, cut - at))
                    tail.append((o, l - (cut - at)))
                at += l
            if i % 2:
                seq = seq[:cut] + bytes(ref[pos - 1 + cut + k:pos - 1 + cut + k + (ln - cut)])
                ops = head + [("D", k), ("M" if not eqx else "=", len(seq) - cut)]
            else:
                ins = bytes(rng.choice(list(b"ACGT"), k).astype(np.uint8))
                seq = seq[:cut] + ins + seq[cut:]
                quals = np.concatenate([quals[:cut], np.full(k, 35, np.uint8), quals[cut:]])
                ops = head + [("I", k)] + tail
        rl = sum(l for o, l in ops if o in "MIS=X")
        reads.append({"pos": pos, "cigar": ops, "seq": seq[:rl], "quals": quals[:rl].tolist() + [30] * (rl - len(quals)), "reverse": bool(i % 3 == 0)})
    reads.sort(key=lambda r: r["pos"])
    return reads


def _bam_of_reads(reads, chrom_len=1_000_000):
    import struct
    from tests.test_bgzf import _bgzf_of
    hdr = b"BAM\x01" + struct.pack("<i", 0) + struct.pack("<i", 1) + struct.pack("<i", 5) + b"chr1\0" + struct.pack("<i", chrom_len)
    code = {c: i for i, c in enumerate("=ACMGRSVTWYHKDBN")}
    out = [hdr]
    for i, r in enumerate(reads):
        name = b"q%06d\0" % i
        cig = b"".join(struct.pack("<I", (l << 4) | "MIDNSHP=X".index(o)) for o, l in r["cigar"])
        seq = r["seq"].decode()
        packed = bytearray((len(seq) + 1) // 2)
        for k, ch in enumerate(seq):
            packed[k >> 1] |= code[ch] << (4 if k % 2 == 0 else 0)
        body = struct.pack("<iiBBHHHiiii", 0, r["pos"] - 1, len(name), 60, 0, len(r["cigar"]), 16 if r["reverse"] else 0, len(seq), -1, -1, 0)
        body += name + cig + bytes(packed) + bytes(r["quals"])
        out.append(struct.pack("<i", len(body)) + body)
    return _bgzf_of(b"".join(out))


@pytest.mark.parametrize("mode", ["somatic", "gvcf", "collapse", "diploid", "forced", "intervals", "window noise"])
def test_bases_of_x_and_eq_operations_are_counts_without_candidates(torch_cuda, mode):
    """ProcessCigarOps walks M, I and D operations only (CandidateVariantFinder.cs:36-83) while AddAlleleCounts counts every operation
    that spans read and reference, = and X included (CigarExtensions.IsReadSpan / IsReferenceSpan): a mismatch under an X operation is
    coverage and allele count, and no SNV candidate — the allele is called only where reads with M operations show it too, and then with
    THEIR support.  MNV calling off, where the tile kernels call SNVs from the allele counts: the walk leaves a record for every such
    base (finder_walk.h kFoundUnwalked), their loci are called from the counts less those bases (surface_flush.inc.h).  Through every
    way reads come in — host arrays, device memory, the observation-log chain, BAM bytes — and against the oracle's schedule."""
    from pisces_amd import engine
    rng = np.random.default_rng(4100)
    ref = bytes(rng.choice(list(b"ACGT"), 3300).astype(np.uint8))
    reads = _eqx_reads(rng, ref)
    assert sum(any(o in "=X" for o, _ in r["cigar"]) for r in reads) > 500
    kw = dict(call_mnvs=0, min_frequency=0.01, include_reference_calls=1 if mode in ("gvcf", "diploid", "forced") else 0, collapse=1 if mode == "collapse" else 0)
    if mode == "diploid":
        kw.update(ploidy=2, min_frequency=0.2, variant_freq_filter=0.2, low_gq_filter=30, max_genotype_qscore=1000)
    if mode == "window noise":
        kw.update(noise_model=1)
    cfg = _abi.default_config(**kw)
    sites = [p for p in range(100, 2900) if p % 61 == 40]
    other = lambda p: "ACGT"[("ACGT".index(chr(ref[p - 1])) + 1) % 4]
    forced = None
    if mode == "forced":   # an allele only X operations show, one M and X reads show, one nobody shows
        x_only, both = [p for p in sites if (p // 61) % 3 == 0], [p for p in sites if (p // 61) % 3 == 2]
        forced = [(x_only[3], chr(ref[x_only[3] - 1]), other(x_only[3])), (both[5], chr(ref[both[5] - 1]), other(both[5])),
                  (both[7] + 3, chr(ref[both[7] + 2]), other(both[7] + 3))]
    intervals = [(300, 1000), (1500, 1620), (2000, 2700)] if mode == "intervals" else None
    cuts = [600, 1300, 1800, len(reads)]
    ups = [reads[c - 1]["pos"] - 1 for c in cuts[:-1]] + [None]

    def run(environ, how="host"):
        with env(**environ):
            with engine.HipVariantCaller(cfg) as c:
                c.SetReference(ref)
                if intervals:
                    c.SetIntervals(intervals)
                if forced:
                    c.SetForcedAlleles(forced)
                rows, alleles, a0 = [], [], 0
                for cut, up in zip(cuts, ups):
                    part = reads[a0:cut]
                    if how == "device":
                        c.AddDeviceReads(engine.DeviceReadBatch.from_host(_abi.ReadBatch(part)))
                    elif how == "bam":
                        assert c.bam_decode(_bam_of_reads(part), 0)["reads"] == len(part)
                        c.AddDecodedReads()
                    else:
                        c.AddAlleleCounts(_abi.ReadBatch(part))
                    a0 = cut
                    r, a = c.CallWithAlleles(up, capacity=1 << 15)
                    rows.append(r)
                    alleles += a
                return np.concatenate(rows), alleles, c.Stats()["TotalNumCalled"]
    want = run({})
    if not intervals:
        exp, exp_alleles, exp_called = orc.run_reads_schedule(_abi.ReadBatch(reads), np.frombuffer(ref, np.uint8), 1, len(ref), cfg, ups[:-1], forced=forced or ())
        assert want[1] == exp_alleles and want[2] == exp_called
        for f in ("position", "total_coverage", "allele_support", "reference_support", "num_no_calls", "coverage_by_dir", "support_by_dir", "filter_bits",
                  "info", "variant_qscore", "genotype_qscore"):
            assert (want[0][f] == exp[f]).all(), f
    # what the test is about: no variant where only X operations show one, variants with the M reads' support where both do
    called = {(int(r["position"]), a[1]): int(r["allele_support"]) for r, a in zip(want[0], want[1]) if a[1] != "." and len(a[0]) == 1 and len(a[1]) == 1
              and not (r["filter_bits"] >> _abi.FILTER_FORCED_REPORT) & 1}
    def shown_under(r, p):   # the operation under which read r shows other(p) at p with a quality that counts, or None
        at, k = r["pos"], 0
        for o, l in r["cigar"]:
            if o in "M=X":
                if at <= p < at + l:
                    return o if r["seq"][k + p - at] == ord(other(p)) and r["quals"][k + p - at] >= 20 else None
                at, k = at + l, k + l
            elif o in "DN":
                at += l
            elif o in "IS":
                k += l
        return None
    in_reads = lambda p, kinds: sum(1 for r in reads if (shown_under(r, p) or "-") in kinds)
    inside = lambda p: not intervals or any(a <= p <= b for a, b in intervals)
    x_sites = [p for p in sites if (p // 61) % 3 == 0 and in_reads(p, "=X") >= 6 and in_reads(p, "M") == 0]
    both_sites = [p for p in sites if (p // 61) % 3 == 2 and in_reads(p, "M") >= 6 and in_reads(p, "=X") >= 6 and inside(p)]
    assert len(x_sites) > 5 and len(both_sites) > 5
    assert not any((p, other(p)) in called for p in x_sites)
    if mode != "diploid":
        assert all(called.get((p, other(p))) == in_reads(p, "M") for p in both_sites), [(p, called.get((p, other(p))), in_reads(p, "M")) for p in both_sites]
    for name, environ, how in [("reads in device memory", {}, "device"), ("the observation-log chain", dict(PISCES_HIP_READ_PATH="log"), "host"),
                               ("BAM bytes", {}, "bam"), ("BAM bytes, the log chain", dict(PISCES_HIP_READ_PATH="log"), "bam"),
                               ("checks on the device", dict(PISCES_HIP_DEVICE_CHECKS=1), "host"), ("candidates merged on the host", dict(PISCES_HIP_DEVICE_MERGE=0), "host"),
                               ("candidates merged on the device", dict(PISCES_HIP_DEVICE_MERGE=1), "host"), ("rows merged by copy", dict(PISCES_HIP_MERGE_IN_PLACE=0), "host")]:
        got = run(environ, how)
        assert got[0].tobytes() == want[0].tobytes() and got[1] == want[1] and got[2] == want[2], (mode, name)


@pytest.mark.gpu
@pytest.mark.parametrize("how", ["host reads", "device reads"])
def test_a_first_batch_that_touches_no_block_leaves_the_segment_without_a_grid(torch_cuda, how):
    """a segment's FIRST batch holds only reads that touch no block (soft clips from end to end), so it leaves no cell of the
    position grid; the batch that joins the same open segment next must not find a grid whose first cells keep the fill value (a tile
    there would see an empty fragment range and drop the reads over it).  Records == the log chain's, counts == the oracle's."""
    from pisces_amd import engine
    rng = np.random.default_rng(5)
    ref = np.frombuffer(bytes(rng.choice(list(b"ACGT"), 9000).astype(np.uint8)), dtype=np.uint8)
    clipped = [{"pos": 5000 + i, "cigar": [("S", 30)], "seq": bytes(rng.choice(list(b"ACGT"), 30).astype(np.uint8)), "quals": [37] * 30,
                "reverse": bool(i & 1)} for i in range(40)]
    real = random_reads(rng, 1500, 5100, 6500, with_dirs=False)
    cfg = _abi.default_config(min_coverage=1, low_depth_filter=1)
    mode = STORE_MODES["every batch appended to the open segment"]

    def run(environ):
        with env(**environ):
            with engine.HipVariantCaller(cfg) as c:
                c.SetReference(ref)
                add = c.AddDeviceReads if how == "device reads" else c.AddAlleleCounts
                add(_abi.ReadBatch(clipped))
                add(_abi.ReadBatch(real))
                counts = c.GetCounts(5000, 2000)
                r, a = c.CallWithAlleles(None, capacity=1 << 14)
                return r, a, counts, c.Stats()
    got = run(dict(PISCES_HIP_READ_PATH=None, **mode))
    with env(PISCES_HIP_READ_PATH="log"):
        with engine.HipVariantCaller(cfg) as c:
            c.SetReference(ref)
            c.AddAlleleCounts(_abi.ReadBatch(clipped))
            c.AddAlleleCounts(_abi.ReadBatch(real))
            want_r, want_a = c.CallWithAlleles(None, capacity=1 << 14)
    exp = oracle_counts(clipped + real, 1, 9000)
    np.testing.assert_array_equal(got[2].reshape(2000, -1), exp.reshape(9000, -1)[4999:6999])
    assert len(want_r) > 1000 and got[0].tobytes() == want_r.tobytes() and got[1] == want_a


@pytest.mark.gpu
def test_exact_total_called_is_refused_where_it_cannot_be_counted(torch_cuda):
    """pisces_hip_set_exact_total_called promises the reference's TotalNumCalled.  A configuration whose flush does not go
    through the read store's flush kernel (here NoiseModel.Window and the Diploid strand-bias model) cannot run the counting launch over the
    off-interval loci: the flush fails with PISCES_E_UNSUPPORTED instead of reporting the smaller total; with the switch off it runs."""
    from pisces_amd import engine
    rng = np.random.default_rng(3)
    ref = rng.choice(np.frombuffer(b"ACGT", np.uint8), 2300)
    reads = [dict(pos=int(s), cigar=[("M", 60)], seq=ref[s - 1:s + 59].tobytes(), quals=bytes([35] * 60), reverse=bool(i & 1))
             for i, s in enumerate(np.sort(rng.integers(50, 2100, 300)))]
    for kw in (dict(noise_model=1), dict(strand_bias_model=_abi.SB_DIPLOID)):
        cfg = _abi.default_config(**kw)
        for exact in (True, False):
            with engine.HipVariantCaller(cfg) as c:
                c.SetReference(ref)
                c.SetIntervals([(200, 700), (1200, 1500)])
                c.SetExactTotalNumCalled(exact)
                c.AddAlleleCounts(_abi.ReadBatch(reads))
                if exact:
                    with pytest.raises(engine.PiscesHipError) as e:
                        c.CallWithAlleles(None)
                    assert e.value.code == _abi.E_UNSUPPORTED and "exact_total_called" in e.value.message
                else:
                    rows, _ = c.CallWithAlleles(None)
                    assert len(rows) > 0


@pytest.mark.gpu
def test_device_checks_with_a_small_block_size_share_one_block_map(torch_cuda):
    """read_prepare_kernel's block map is kept in 32 copies while that is small; at a block size of 100 (86 MB of copies) the copies
    alias one map.  Device-fed reads then give the records of host-fed reads, whose blocks the host's own pass over the CIGARs lists."""
    from pisces_amd import engine
    rng = np.random.default_rng(12)
    ref = np.frombuffer(bytes(rng.choice(list(b"ACGT"), 5000).astype(np.uint8)), dtype=np.uint8)
    reads = random_reads(rng, 1500, 30, 4500, with_dirs=False)
    cfg = _abi.default_config(min_coverage=1, low_depth_filter=1, block_size=100)
    out = {}
    for how in ("host", "device"):
        with env(PISCES_HIP_READ_PATH=None, PISCES_HIP_DEVICE_CHECKS=0):
            with engine.HipVariantCaller(cfg) as c:
                c.SetReference(ref)
                recs = []
                for part, up in ((reads[:800], reads[799]["pos"] - 1), (reads[800:], None)):
                    (c.AddDeviceReads if how == "device" else c.AddAlleleCounts)(_abi.ReadBatch(part))
                    recs.append(c.CallWithAlleles(up, capacity=1 << 15))
                out[how] = (np.concatenate([r for r, _ in recs]).tobytes(), [a for _, al in recs for a in al], c.Stats())
    assert out["host"] == out["device"] and len(out["host"][0]) > 64 * 3000
