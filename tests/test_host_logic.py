"""Host logic that needs no GPU: the read -> observation-tuple expander (pisces_hip_expand_reads,
pure C++) against the oracle's AddAlleleCounts, and the synthetic generator's self-consistency."""
import json
import os

import numpy as np
import pytest

from pisces_amd import _abi, engine
from tests import orc

G = os.path.join(os.path.dirname(__file__), "golden")
DIR = {"F": 0, "R": 1, "S": 2}


def counts_from_observations(pos, tup, start, n_loci, min_bq):
    """numpy histogram of (position, tuple) observations with the kernel's 'qual < minBQ -> N' rule."""
    c = np.zeros((n_loci, 6, 3, _abi.NUM_ANCHORS), dtype=np.int32)
    _, anchor, direction, allele, qual = _abi.tuple_fields(tup)
    allele = np.where((allele < 4) & (qual < min_bq), 4, allele)
    keep = (pos >= start) & (pos < start + n_loci)
    np.add.at(c, (pos[keep] - start, allele[keep], direction[keep], anchor[keep]), 1)
    return c


def _read_dict(rd, default_q):
    d = {"pos": rd["pos"], "seq": rd["seq"], "cigar": orc.parse_cigar(rd["cigar"]) if "cigar" in rd else [("M", len(rd["seq"]))],
         "quals": rd.get("quals", [rd.get("qual", default_q)] * len(rd["seq"]))}
    if "dirs" in rd:
        d["dirs"] = [DIR[rd["dirs"]]] * len(rd["seq"])
    return d


def _scenarios():
    g = json.load(open(os.path.join(G, "region_state.json")))
    out = [("add_and_get", g["add_and_get"]["min_quality"],
            [r for r in g["add_and_get"]["reads"] if "posmap_unmapped_index" not in r] + [g["add_and_get"]["then_read"]])]
    for sc in g["poor_qual_deletions"]["scenarios"]:
        out.append((sc["name"], g["poor_qual_deletions"]["min_quality"], sc["reads"]))
    # extra CIGAR shapes: insertion, soft clips both ends, leading/trailing deletions, stitched directions
    out.append(("mixed", 20, [
        {"seq": "ACGTACGTACGT", "pos": 1010, "cigar": "3S4M2I3M", "qual": 30},
        {"seq": "ACGTACGTAC", "pos": 1020, "cigar": "5M3D5M", "quals": [30, 30, 30, 30, 10, 30, 30, 30, 30, 30]},
        {"seq": "ACGTACGTAC", "pos": 1020, "cigar": "5M3D5M", "quals": [30] * 10, "dirs": "S"},
        {"seq": "ACGTAC", "pos": 1030, "cigar": "2D6M", "qual": 30},
        {"seq": "ACGTAC", "pos": 1040, "cigar": "6M3D", "qual": 30, "dirs": "R"},
        {"seq": "ACGTACNN", "pos": 1050, "cigar": "6M2D2S", "qual": 30},
        {"seq": "A", "pos": 1060, "qual": 30},
    ]))
    return out


@pytest.mark.parametrize("name,min_bq,reads", _scenarios(), ids=[s[0] for s in _scenarios()])
def test_expander_matches_oracle_add_allele_counts(name, min_bq, reads):
    st = orc.State(900, 300, min_bq=min_bq)
    for rd in reads:
        d = _read_dict(rd, 30)
        assert st.add_allele_counts(orc.make_read(d["pos"], d["seq"], cigar=d["cigar"], quals=d["quals"],
                                                  dirs=d.get("dirs"))) == 0
    batch = _abi.ReadBatch([_read_dict(rd, 30) for rd in reads])
    pos, tup = engine.expand_reads(batch, min_bq)
    got = counts_from_observations(pos, tup, 900, 300, min_bq)
    np.testing.assert_array_equal(got, st.counts())


def test_expander_random_reads_match_oracle():
    rng = np.random.default_rng(7)
    reads = []
    for _ in range(300):
        ops = []
        n_ops = rng.integers(1, 6)
        for k in range(n_ops):
            ops.append((str(rng.choice(list("MMMIDS"))), int(rng.integers(1, 12))))
        # CIGAR hygiene the BAM spec guarantees: S only at the ends, at least one M
        ops = [(o, l) for i, (o, l) in enumerate(ops) if o != "S" or i in (0, len(ops) - 1)]
        if not any(o == "M" for o, _ in ops):
            ops.append(("M", 5))
        rl = sum(l for o, l in ops if o in "MIS")
        reads.append({"pos": int(rng.integers(950, 1100)), "cigar": ops,
                      "seq": "".join(rng.choice(list("ACGTN"), rl, p=[.24, .24, .24, .24, .04])),
                      "quals": rng.choice([10, 25, 37], rl, p=[.1, .2, .7]).astype(np.uint8).tolist(),
                      "reverse": bool(rng.integers(0, 2))})
    st = orc.State(900, 400, min_bq=20)
    for d in reads:
        assert st.add_allele_counts(orc.make_read(d["pos"], d["seq"], cigar=d["cigar"], quals=d["quals"],
                                                  reverse=d["reverse"])) == 0
    pos, tup = engine.expand_reads(_abi.ReadBatch(reads), 20)
    np.testing.assert_array_equal(counts_from_observations(pos, tup, 900, 400, 20), st.counts())


def test_synthetic_pileup_reads_and_tuples_agree():
    """The generator's two views (reads for the oracle, tile-bucketed tuples for the device) describe the
    same pileup: oracle(reads) == oracle(observations), and depth is exact at every locus."""
    from pisces_amd import synth
    p = synth.make_pileup(n_loci=500, depth=60, seed=11)
    cfg = _abi.default_config()
    ref = p.ref.numpy()
    a, na = orc.run_reads(synth.reads_of(p), ref, p.region_start, p.n_loci, cfg)
    pos, tup = synth.observations_of(p)
    b, nb = orc.run_observations(pos, tup, ref, p.region_start, p.n_loci, cfg)
    assert na == nb == 500
    assert a.tobytes() == b.tobytes()
    assert ((a["total_coverage"] + a["num_no_calls"]) == 60).all()
    # SNVs are called only at planted sites (0.1 % sequencing errors stay below the emit thresholds at depth 60)
    called_snv = a[[(_abi.info_category(i) == _abi.CAT_SNV) for i in a["info"]]]
    assert 0 < len(called_snv) and set((called_snv["position"] - p.region_start).tolist()) <= set(p.planted.tolist())
    # expander(reads) gives the same observations as the generator's tuple view (up to order)
    epos, etup = engine.expand_reads(synth.reads_of(p), 20)
    key = lambda P, T: np.sort((P.astype(np.int64) << 32) | (T & ~np.uint32(0xFC)).astype(np.int64))
    np.testing.assert_array_equal(key(epos, etup), key(pos, tup))


def test_indel_finder_matches_oracle_finder():
    """pisces_hip_find_indel_candidates (host C++) vs the oracle's CandidateVariantFinder restatement, per read:
    coordinates, alleles, support direction, well-anchored support and open-end flags of every insertion / deletion."""
    rng = np.random.default_rng(11)
    ref = bytes(rng.choice(list(b"ACGT"), 600).astype(np.uint8))
    reads = []
    for _ in range(400):
        ops = []
        for k in range(int(rng.integers(1, 6))):
            ops.append((str(rng.choice(list("MMMIDS"))), int(rng.integers(1, 9))))
        ops = [(o, l) for i, (o, l) in enumerate(ops) if o != "S" or i in (0, len(ops) - 1)]
        if not any(o == "M" for o, _ in ops):
            ops.insert(len(ops) // 2, ("M", 4))
        rl = sum(l for o, l in ops if o in "MIS")
        stitched = rng.random() < 0.3
        reads.append({"pos": int(rng.integers(20, 500)), "cigar": ops,
                      "seq": "".join(rng.choice(list("ACGT"), rl)),
                      "quals": rng.choice([10, 25, 37], rl, p=[.15, .15, .7]).astype(np.uint8).tolist(),
                      "reverse": bool(rng.integers(0, 2)),
                      "dirs": rng.choice([0, 1, 2], rl).tolist() if stitched else None})
    got = engine.find_indel_candidates(_abi.ReadBatch(reads), ref, 20)
    exp = []
    for d in reads:
        rd = orc.make_read(d["pos"], d["seq"], cigar=d["cigar"], quals=d["quals"], reverse=d["reverse"], dirs=d["dirs"])
        for c in orc.find_candidates(rd, ref.decode()):
            if c.category in (_abi.CAT_INSERTION, _abi.CAT_DELETION):
                exp.append({"position": c.position, "category": c.category, "ref": c.ref.decode(), "alt": c.alt.decode(),
                            "support_by_dir": list(c.support_by_dir), "well_anchored_by_dir": list(c.well_anchored_by_dir),
                            "open_left": bool(c.open_left), "open_right": bool(c.open_right)})
    assert len(exp) > 100
    assert got == exp


def test_interval_shards_concatenate_to_the_whole():
    """loci shard by interval with no exchange — the oracle over amplicon shards, concatenated in shard order,
    is the oracle over the whole region (what bench.py's threaded CPU baseline and the multi-GPU host rely on)."""
    from pisces_amd import synth
    p = synth.make_pileup(900, 30, seed=3)
    cfg = _abi.default_config()
    ref = p.ref.cpu().numpy()
    whole, n = orc.run_reads(synth.reads_of(p), ref, p.region_start, p.n_loci, cfg)
    parts = []
    for a0 in range(0, 6, 2):
        b = synth.reads_of(p, 2, first_amplicon=a0)
        r, _ = orc.run_reads(b, ref, p.region_start + a0 * synth.READ_LEN, 2 * synth.READ_LEN, cfg)
        parts.append(r)
    assert n == 900 and np.concatenate(parts).tobytes() == whole.tobytes()


def test_library_indel_finder_on_the_reference_cases():
    """pisces_hip_find_indel_candidates (finder.cpp, what pisces_hip_add_reads runs per read) on the reference's DeletionTests /
    InsertionTests known answers (tests/golden/finder_cases.json); SNV expectations of those cases belong to the device counts."""
    import json, os
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "finder_cases.json")))
    cat = {"Insertion": _abi.CAT_INSERTION, "Deletion": _abi.CAT_DELETION}
    n_checked = 0
    for case in g["cases"]:
        if not case["read"]:
            continue
        start = 101
        ops = orc.parse_cigar(case["cigar"])
        clip = ops[0][1] if ops and ops[0][0] == "S" else 0
        ref = ("N" * (start - 1 - clip) + case["ref_under_read"] + "NNNNN").encode()
        batch = _abi.ReadBatch([{"pos": start, "cigar": ops, "seq": case["read"], "quals": case["quals"], "reverse": False}])
        got = sorted(engine.find_indel_candidates(batch, ref, g["min_base_call_quality"]), key=lambda c: c["position"])
        exp = [e for e in case["expected"] if e["type"] in cat] if case["expected_count"] else []
        assert len(got) == len(exp), (case["cigar"], case["quals"], got)
        for c, e in zip(got, exp):
            assert (c["position"] - start, c["ref"], c["alt"], c["category"]) == (e["coord"], e["ref"], e["alt"], cat[e["type"]])
            if e["open_left"] is not None:
                assert c["open_left"] == e["open_left"]
            if e["open_right"] is not None:
                assert c["open_right"] == e["open_right"]
            n_checked += 1
    assert n_checked >= 30


def test_full_finder_matches_oracle_finder_with_mnvs():
    """pisces_hip_find_candidates with the M walk on (what add_reads runs when call_mnvs is set) vs the oracle's finder, per read and
    in read order: SNV / MNV / insertion / deletion candidates with alleles, direction, well-anchored support and open ends, for
    several MaxSizeMNV / MaxGapBetweenMNV settings."""
    rng = np.random.default_rng(23)
    ref = bytes(rng.choice(list(b"ACGT"), 600).astype(np.uint8))
    reads = []
    for _ in range(300):
        ops = []
        for k in range(int(rng.integers(1, 5))):
            ops.append((str(rng.choice(list("MMMMIDS"))), int(rng.integers(1, 30))))
        ops = [(o, l) for i, (o, l) in enumerate(ops) if o != "S" or i in (0, len(ops) - 1)]
        if not any(o == "M" for o, _ in ops):
            ops.insert(len(ops) // 2, ("M", 12))
        pos = int(rng.integers(20, 400))
        seq, rp = [], pos
        for o, l in ops:   # mostly the reference, with mismatches so that MNVs and gapped MNVs form
            if o == "M":
                seg = bytearray(ref[rp - 1: rp - 1 + l])
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
    for call_mnvs, max_len, max_gap in ((False, 3, 1), (True, 3, 1), (True, 15, 10), (True, 2, 0)):
        got = engine.find_candidates(_abi.ReadBatch(reads), ref, 20, True, call_mnvs, max_len, max_gap)
        exp = []
        for d in reads:
            rd = orc.make_read(d["pos"], d["seq"], cigar=d["cigar"], quals=d["quals"], reverse=d["reverse"], dirs=d["dirs"])
            for c in orc.find_candidates(rd, ref.decode(), call_mnvs=call_mnvs, max_mnv=max_len, max_gap=max_gap):
                exp.append({"position": c.position, "category": c.category, "ref": c.ref.decode(), "alt": c.alt.decode(),
                            "support_by_dir": list(c.support_by_dir), "well_anchored_by_dir": list(c.well_anchored_by_dir),
                            "open_left": bool(c.open_left), "open_right": bool(c.open_right)})
        assert sum(e["category"] == _abi.CAT_MNV for e in exp) > (50 if call_mnvs else -1)
        assert got == exp, (call_mnvs, max_len, max_gap)


def test_finder_read_overhanging_the_contig_end():
    """An M operation that runs past the end of the reference (circular contigs: PhiX, chrM): the walk stops at the last reference
    base and a pending mismatch comes out at its own coordinates — the reference's flush reads from `operationLength` there and
    Substring throws (CandidateVariantFinder.cs:103-104,162-165).  Library == oracle, nothing is read beyond the reference."""
    ref = b"ACGTACGTAC"   # 10 bases
    for call_mnvs in (False, True):
        for seq in ("GTAC" + "TGCATG", "GTAA" + "TGCATG", "GTGG" + "TGCATG"):   # read at 7: M walk over 7..10, then 6 bases of overhang
            reads = [{"pos": 7, "cigar": [("M", 10)], "seq": seq, "quals": [37] * 10, "reverse": False, "dirs": None}]
            got = engine.find_candidates(_abi.ReadBatch(reads), ref, 20, True, call_mnvs, 3, 1)
            rd = orc.make_read(7, seq, cigar=[("M", 10)], quals=[37] * 10)
            exp = [{"position": c.position, "category": c.category, "ref": c.ref.decode(), "alt": c.alt.decode(),
                    "support_by_dir": list(c.support_by_dir), "well_anchored_by_dir": list(c.well_anchored_by_dir),
                    "open_left": bool(c.open_left), "open_right": bool(c.open_right)}
                   for c in orc.find_candidates(rd, ref.decode(), call_mnvs=call_mnvs, max_mnv=3, max_gap=1)]
            assert got == exp, (call_mnvs, seq)
            assert all(1 <= e["position"] and e["position"] + len(e["ref"]) - 1 <= len(ref) for e in got), got
            if seq.startswith("GTAA"):
                assert [(e["position"], e["ref"], e["alt"]) for e in got] == [(10, "C", "A")]


def test_library_full_finder_on_all_reference_cases():
    """pisces_hip_find_candidates with callMNVs on, over all 106 reads of the reference's VariantFinderTests (SNV, MNV, deletion and
    insertion suites, tests/golden/finder_cases.json): every expected candidate, nothing else."""
    import json, os
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "finder_cases.json")))
    cat = {"Snv": _abi.CAT_SNV, "Mnv": _abi.CAT_MNV, "Insertion": _abi.CAT_INSERTION, "Deletion": _abi.CAT_DELETION}
    n_checked = 0
    for case in g["cases"]:
        if not case["read"]:
            continue
        start = 101
        ops = orc.parse_cigar(case["cigar"])
        clip = ops[0][1] if ops and ops[0][0] == "S" else 0
        ref = ("N" * (start - 1 - clip) + case["ref_under_read"] + "NNNNN").encode()
        batch = _abi.ReadBatch([{"pos": start, "cigar": ops, "seq": case["read"], "quals": case["quals"], "reverse": False}])
        got = engine.find_candidates(batch, ref, g["min_base_call_quality"], True, g["call_mnvs"], case["max_mnv_length"], case["max_gap"])
        exp = case["expected"] if case["expected_count"] else []
        assert len(got) == case["expected_count"], (case["cigar"], case["read"], got)
        key = lambda c: (c["position"], c["category"], c["ref"], c["alt"])
        got = sorted(got, key=key)
        for e in exp:
            m = [c for c in got if key(c) == (e["coord"] + start, cat[e["type"]], e["ref"], e["alt"])]
            assert len(m) == 1, (case, got)
            if e["open_left"] is not None:
                assert m[0]["open_left"] == e["open_left"]
            if e["open_right"] is not None:
                assert m[0]["open_right"] == e["open_right"]
            n_checked += 1
    assert n_checked >= 90


def _xd_of(expanded):
    out, i = "", 0
    while i < len(expanded):
        j = i
        while j < len(expanded) and expanded[j] == expanded[i]:
            j += 1
        out += f"{j - i}{'FRS'[expanded[i]]}"
        i = j
    return out


def test_library_finder_stitched_deletion_directions():
    """The deletion direction of stitched reads through the C-ABI (PiscesReadBatch.deletion_directions, filled from the XD tag by
    _abi.directions_from_xd as the C# shim fills it from CigarDirections.Expand()): the reference's 14 scenarios, then random stitched
    reads with deletions against the oracle's literal GetDeletionDirectionForStitchedRead on the expanded map."""
    import json
    doc = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "support_direction_deletion_cases.json")))
    code = {"Forward": _abi.DIR_FORWARD, "Reverse": _abi.DIR_REVERSE, "Stitched": _abi.DIR_STITCHED}
    rd = doc["read"]
    ref = b"ATCG" * 5
    reads = [{"pos": rd["position"], "cigar": orc.parse_cigar(rd["cigar"]), "seq": rd["sequence"], "quals": [30] * len(rd["sequence"]),
              "xd": f'{c["num_forward"]}F{c["num_stitched"]}S{c["num_reverse"]}R'} for c in doc["cases"]]
    got = engine.find_indel_candidates(_abi.ReadBatch(reads), ref, 20)
    assert len(got) == len(doc["cases"])
    for g, c in zip(got, doc["cases"]):
        want = [0, 0, 0]
        want[code[c["expected"]]] = 1
        assert g["category"] == _abi.CAT_DELETION and g["support_by_dir"] == want, c["name"]

    rng = np.random.default_rng(77)
    ref = bytes(rng.choice(list(b"ACGT"), 700).astype(np.uint8))
    reads, exp = [], []
    for k in range(500):
        ops = [("M", int(rng.integers(1, 12)))]
        for _ in range(int(rng.integers(1, 4))):
            ops.append((str(rng.choice(list("DDI"))), int(rng.integers(1, 7))))
            ops.append(("M", int(rng.integers(1, 12))))
        if rng.random() < 0.2:
            ops.insert(0, ("S", 3))
        n_exp = sum(l for _, l in ops)
        # F.. S.. R.. with random cut points, as a stitcher writes the XD tag
        a, b = sorted(rng.integers(0, n_exp + 1, 2).tolist())
        expanded = [0] * a + [2] * (b - a) + [1] * (n_exp - b)
        rl = sum(l for o, l in ops if o in "MIS")
        d = {"pos": int(rng.integers(20, 600)), "cigar": ops, "seq": "".join(rng.choice(list("ACGT"), rl)),
             "quals": rng.choice([10, 25, 37], rl, p=[.1, .2, .7]).astype(np.uint8).tolist()}
        tracked = rng.random() < 0.8   # the rest: reads without CigarDirections in the same batch
        sequenced, e = [], 0
        for o, l in ops:
            if o in "MIS":
                sequenced += expanded[e:e + l]
            e += l
        if tracked:
            d["xd"] = _xd_of(expanded)
        else:
            d["dirs"] = sequenced
        reads.append(d)
        r = orc.make_read(d["pos"], d["seq"], cigar=ops, quals=d["quals"], dirs=sequenced, expanded_dirs=expanded if tracked else None)
        for c in orc.find_candidates(r, ref.decode()):
            if c.category in (_abi.CAT_INSERTION, _abi.CAT_DELETION):
                exp.append({"position": c.position, "category": c.category, "ref": c.ref.decode(), "alt": c.alt.decode(),
                            "support_by_dir": list(c.support_by_dir), "well_anchored_by_dir": list(c.well_anchored_by_dir),
                            "open_left": bool(c.open_left), "open_right": bool(c.open_right)})
    batch = _abi.ReadBatch(reads)
    assert batch.deletion_directions is not None and (batch.deletion_directions != 255).any() and (batch.deletion_directions == 255).any()
    got = engine.find_indel_candidates(batch, ref, 20)
    assert sum(g["category"] == _abi.CAT_DELETION for g in got) > 200
    assert got == exp
