"""The fuzz of the streaming surface: the product (default form) against the oracle's block schedule over random reads, modes and flush
schedules — every dimension at once (bases that are no A C G T N, any CIGAR, = and X operations, stitched directions, quality
thresholds, block sizes, depth thresholds, filters, collapser thresholds, forced alleles, ploidy, noise model).  one(seed) runs a
draw; tests/test_gpu_parity.py runs the seeds that once differed and a band of fresh ones, tools/fuzz_oracle.py any range:
    python tools/fuzz_oracle.py [first_seed] [n_seeds]
"""
import os
import sys

import numpy as np

from pisces_amd import _abi
from tests import orc


FORMS = [("rows merged by copy", dict(PISCES_HIP_MERGE_IN_PLACE=0), "host"), ("checks on the device", dict(PISCES_HIP_DEVICE_CHECKS=1), "host"),
         ("checks on the host", dict(PISCES_HIP_DEVICE_CHECKS=0), "host"), ("candidates merged on the host", dict(PISCES_HIP_DEVICE_MERGE=0), "host"),
         ("candidates merged on the device", dict(PISCES_HIP_DEVICE_MERGE=1), "host"), ("walk base by base", dict(PISCES_HIP_FINDER="bases"), "host"),
         ("walk a wave a read", dict(PISCES_HIP_FINDER="wave"), "host"), ("walk in batches", dict(PISCES_HIP_FINDER="batch"), "host"),
         ("genotypes by the host pass", dict(PISCES_HIP_DEVICE_GENOTYPER=0), "host"), ("every batch its own segment", dict(PISCES_HIP_STORE_DIRECT_BYTES=0), "host"),
         ("every batch appended", dict(PISCES_HIP_STORE_DIRECT_BYTES=1 << 40, PISCES_HIP_STORE_SEAL_BYTES=1 << 40), "host"),
         ("reads in device memory", {}, "device"), ("candidates looked at after every add", {}, "peek"),
         ("every candidate group to the host", dict(PISCES_HIP_MNV_SPLIT=0), "host"), ("the observation-log chain", dict(PISCES_HIP_READ_PATH="log"), "host"),
         ("BAM bytes", {}, "bam"), ("BAM bytes, the log chain", dict(PISCES_HIP_READ_PATH="log"), "bam"),
         ("reads in device memory, checks on the host's side of things off", dict(PISCES_HIP_DEVICE_MERGE=1), "device"),
         ("the flush pair", {}, "pair"), ("the flush pair, the next batch added before the rows are taken", {}, "pair ahead")]


def one(seed, verbose=False, rows_too=False):
    from pisces_amd import engine
    from tests.test_gpu_parity import _mnv_reads, INT_FIELDS
    from tests.test_read_store import random_reads, _eqx_reads
    rng = np.random.default_rng(990000 + seed)
    L = int(rng.choice([2200, 3300, 5200]))
    ref = bytearray(rng.choice(list(b"ACGT"), L).astype(np.uint8))
    if seed % 5 == 1:   # stretches of N and a homopolymer / repeat (RMxN) in the reference
        a = int(rng.integers(300, L - 400)); ref[a:a + int(rng.integers(1, 30))] = b"N" * 30
        b = int(rng.integers(300, L - 400)); ref[b:b + 24] = b"ACACACACACACACACACACACAC"
        c = int(rng.integers(300, L - 400)); ref[c:c + 14] = b"TTTTTTTTTTTTTT"
    ref = bytes(ref[:L])
    region = (50, L - 200)
    reads = _mnv_reads(rng, bytearray(ref), int(rng.integers(600, 3000)), region=region, snv_rate=float(rng.choice([0.001, 0.004, 0.01])))
    for r in reads:
        r["seq"] = r["seq"].encode() if isinstance(r["seq"], str) else r["seq"]
    kind = seed % 4
    if kind == 1:
        reads += random_reads(rng, int(rng.integers(50, 500)), 60, L - 400, exotic=bool(rng.integers(0, 2)), sort=False)
    elif kind == 2:
        reads += _eqx_reads(rng, ref.replace(b"N", b"A"), n=int(rng.integers(100, 900)), lo=60, hi=L - 400)
    for r in reads:   # (a NUL ends the oracle's allele strings, which are C strings)
        if b"\0" in r["seq"]:
            r["seq"] = r["seq"].replace(b"\0", b".")
    ext = seed >= 100000   # (and from 200 000: see FORMS)
    # the wider draw: a deep pile (counts beyond the memo tables), more thresholds, candidates from the host
    if ext and seed % 3 == 0:
        a = int(rng.integers(200, L - 700))
        pile = _mnv_reads(rng, bytearray(ref), int(rng.integers(800, 3500)), region=(a, a + int(rng.integers(200, 400))), snv_rate=0.002)
        for r in pile:
            r["seq"] = r["seq"].encode() if isinstance(r["seq"], str) else r["seq"]
        reads += pile
    reads.sort(key=lambda r: r["pos"])
    ploidy = int(rng.choice([0, 0, 0, 1, 2]))
    kw = dict(call_mnvs=int(rng.integers(0, 2)), max_mnv_length=int(rng.choice([2, 3, 5])), max_gap_between_mnv=int(rng.choice([0, 1, 2])),
              collapse=int(rng.integers(0, 2)), include_reference_calls=int(rng.integers(0, 2)), ploidy=ploidy,
              noise_model=int(rng.choice([0, 0, 1])), strand_bias_model=int(rng.choice([1, 1, 2])),
              min_frequency=0.2 if ploidy else float(rng.choice([0.005, 0.01, 0.05])),
              min_base_call_quality=int(rng.choice([20, 20, 13, 30, 0])), block_size=int(rng.choice([1000, 1000, 500, 64, 4000])),
              min_coverage=int(rng.choice([10, 1, 0, 50])), low_depth_filter=int(rng.choice([10, 30, -1])),
              emit_zero_coverage_refs=int(rng.integers(0, 2)), filter_single_strand=int(rng.integers(0, 2)),
              min_variant_qscore=int(rng.choice([20, 0, 40])), variant_qscore_filter=int(rng.choice([30, 20, 60])),
              expect_stitched_reads=int(rng.choice([0, 0, 1])), collapse_freq_threshold=float(rng.choice([0.0, 0.02])),
              collapse_freq_ratio_threshold=float(rng.choice([0.5, 0.2])), noise_level=int(rng.choice([20, 30])),
              rmxn_min_repetitions=int(rng.choice([9, 4])))
    if ext:
        kw.update(strand_bias_threshold=float(rng.choice([0.5, 0.1, 0.9])), no_call_filter_threshold=float(rng.choice([0.6, 0.02, -1.0])),
                  rmxn_max_repeat_length=int(rng.choice([5, 2, -1])), rmxn_frequency_limit=float(rng.choice([0.35, 1.0])),
                  genotype_min_freq_filter=float(rng.choice([0.01, 0.05])), target_lod_frequency=float(rng.choice([0.01, 0.05])),
                  min_genotype_qscore=int(rng.choice([0, 10])), max_variant_qscore=int(rng.choice([100, 60, 3000])),
                  low_gq_filter=int(rng.choice([-1, 20, 50])))
        if not ploidy:
            kw.update(variant_freq_filter=float(rng.choice([0.01, 0.03, 0.1])), max_genotype_qscore=int(rng.choice([100, 40])))
        else:
            kw.update(diploid_snv_params=[float(x) for x in rng.choice([[0.20, 0.70, 0.80], [0.10, 0.60, 0.90]])],
                      diploid_indel_params=[float(x) for x in rng.choice([[0.20, 0.70, 0.80], [0.15, 0.75, 0.85]])])
    if seed >= 600000:   # collapser thresholds that keep twins apart (the read walk makes the SNV candidates whatever MNV calling is: section 7)
        kw.update(collapse=1, collapse_freq_threshold=float(rng.choice([0.0, 0.02, 0.1])), collapse_freq_ratio_threshold=float(rng.choice([1.0, 2.0, 0.5])))
    if kw["block_size"] < 500:   # (the oracle emits zero-coverage rows over all of its region, the state manager over the blocks that exist)
        kw["emit_zero_coverage_refs"] = 0
    if kw["variant_qscore_filter"] < kw["min_variant_qscore"]:
        kw["variant_qscore_filter"] = kw["min_variant_qscore"]
    if kw["max_gap_between_mnv"] > kw["max_mnv_length"] - 2:
        kw["max_gap_between_mnv"] = max(0, kw["max_mnv_length"] - 2)
    if ploidy:
        kw.update(variant_freq_filter=0.2, low_gq_filter=30, max_genotype_qscore=1000)
    cfg = _abi.default_config(**kw)
    forced = None
    if seed % 3 == 2:
        forced = []
        for _ in range(int(rng.integers(1, 6))):
            p = int(rng.integers(100, L - 300))
            rb = chr(ref[p - 1])
            if rb not in "ACGT":
                continue
            k = int(rng.integers(0, 4))
            if k == 0:
                forced.append((p, rb, "ACGT"[("ACGT".index(rb) + 1 + int(rng.integers(0, 3))) % 4]))
            elif k == 1 and all(chr(x) in "ACGT" for x in ref[p - 1:p + 3]):
                forced.append((p, ref[p - 1:p + 2].decode(), rb))
            elif k == 2:
                forced.append((p, rb, rb + "GT"))
            elif all(chr(x) in "ACGT" for x in ref[p - 1:p + 1]):
                forced.append((p, ref[p - 1:p + 1].decode(), "".join("ACGT"[("ACGT".index(chr(x)) + 2) % 4] for x in ref[p - 1:p + 1])))
        forced = sorted(set(forced)) or None
    n_cuts = int(rng.integers(1, 6))
    cuts = sorted(set(int(x) for x in rng.integers(1, len(reads), n_cuts)) | {len(reads)})
    ups = [int(x) for x in sorted(rng.integers(200, L - 200, len(cuts) - 1))] + [None]
    rows, alleles, a0, schedule = [], [], 0, []
    # seeds from 200 000: one of the forms the library can take for the same work (tests/test_gpu_parity.py, the switch test), drawn by the seed
    form = FORMS[seed % len(FORMS)] if seed >= 200000 else ("default", {}, "host")
    from tests.test_read_store import env, _bam_of_reads
    # seeds from 300 000: an interval set (ChrIntervalSet): Reference candidates inside the intervals only, a callable allele outside them
    # counted and not reported (RegionState.cs:414-447, AlleleCaller.cs:260-263) — the oracle's schedule takes the set
    intervals = None
    if seed >= 300000:
        form = FORMS[seed % len(FORMS)] if seed % 2 else ("default", {}, "host")
        at, intervals = int(rng.integers(1, 400)), []
        while at < L - 100:
            b = at + int(rng.choice([0, 5, 40, 150, 700]))
            intervals.append((at, min(b, L)))
            at = b + int(rng.choice([1, 2, 30, 300, 900]))
        if forced:   # (Factory.SelectForcedAllele keeps the forced alleles inside the intervals)
            forced = [f for f in forced if any(a <= f[0] <= b for a, b in intervals)] or None
    if form[2] == "bam" and any("dirs" in r or any(ch not in b"ACGTN" for ch in r["seq"]) or any(q > 93 for q in r["quals"]) for r in reads):
        form = ("default", {}, "host")   # (per-base direction arrays and bases outside the BAM alphabet cannot be written into a BAM record)
    if os.environ.get("PISCES_FUZZ_FORM"):   # (development: any seed in a given form)
        form = ("default", {}, "host") if os.environ["PISCES_FUZZ_FORM"] == "default" else [f for f in FORMS if f[0] == os.environ["PISCES_FUZZ_FORM"]][0]
    # seeds from 500 000: candidates the host hands in itself (IStateManager.AddCandidates) behind the last batch, beyond every flush of the
    # schedule: an SNV, an insertion, a deletion and (MNV calling on) an MNV with support of their own, at positions reads cover
    host_cands = []
    if seed >= 500000:
        hi_reads = max(r["pos"] for r in reads)
        base = hi_reads + 2   # (every flush of the schedule lies below the last read's start)
        other = lambda p: "ACGT"[("ACGT".index(chr(ref[p - 1])) + 1) % 4] if chr(ref[p - 1]) in "ACGT" else "A"
        ok = lambda p, n: all(chr(x) in "ACGT" for x in ref[p - 1:p - 1 + n])
        sup = lambda: tuple(int(x) for x in rng.integers(0, 6, 2)) + (0,)
        p = base + int(rng.integers(0, 20))
        if ok(p, 1):
            s1 = sup(); host_cands.append(dict(position=p, category=_abi.CAT_SNV, ref=chr(ref[p - 1]), alt=other(p), support_by_dir=s1, well_anchored_by_dir=s1))
        p = base + 25 + int(rng.integers(0, 10))
        if ok(p, 1):
            s1 = sup(); host_cands.append(dict(position=p, category=_abi.CAT_INSERTION, ref=chr(ref[p - 1]), alt=chr(ref[p - 1]) + "GT", support_by_dir=s1, well_anchored_by_dir=s1))
        p = base + 40 + int(rng.integers(0, 10))
        if ok(p, 4):
            s1 = sup(); host_cands.append(dict(position=p, category=_abi.CAT_DELETION, ref=ref[p - 1:p + 2].decode(), alt=chr(ref[p - 1]), support_by_dir=s1, well_anchored_by_dir=s1))
        p = base + 55 + int(rng.integers(0, 10))
        if kw["call_mnvs"] and ok(p, 2):
            s1 = sup(); host_cands.append(dict(position=p, category=_abi.CAT_MNV, ref=ref[p - 1:p + 1].decode(), alt=other(p) + other(p + 1), support_by_dir=s1, well_anchored_by_dir=s1))
        if intervals:
            host_cands = [h for h in host_cands if any(a <= h["position"] <= b for a, b in intervals)]
    with env(**form[1]):
        with engine.HipVariantCaller(cfg) as c:
            c.SetReference(ref)
            if intervals:
                c.SetIntervals(intervals)
            if forced:
                c.SetForcedAlleles(forced)
            pending = False
            for cut, up in zip(cuts, ups):
                part = reads[a0:cut]
                if form[2] == "device":
                    c.AddDeviceReads(engine.DeviceReadBatch.from_host(_abi.ReadBatch(part)))
                elif form[2] == "bam":
                    assert c.bam_decode(_bam_of_reads(part), 0)["reads"] == len(part)
                    c.AddDecodedReads()
                else:
                    c.AddAlleleCounts(_abi.ReadBatch(part))
                a0 = cut
                if host_cands and cut == cuts[-1]:
                    c.AddCandidates(host_cands)
                if pending:   # (the flush that was begun before this batch was added: its blocks lie below every read of the batch)
                    r, a = c.CallEndWithAlleles(capacity=1 << 16)
                    rows.append(r)
                    alleles += a
                    pending = False
                if form[2] == "peek":
                    c.GetCandidates(None)
                if up is not None:
                    up = min(up, reads[cut - 1]["pos"] - 1)
                    schedule.append(up)
                if form[2] in ("pair", "pair ahead"):
                    c.CallBegin(up)
                    if form[2] == "pair ahead":
                        pending = True
                        continue
                    r, a = c.CallEndWithAlleles(capacity=1 << 16)
                else:
                    r, a = c.CallWithAlleles(up, capacity=1 << 16)
                rows.append(r)
                alleles += a
            if pending:
                r, a = c.CallEndWithAlleles(capacity=1 << 16)
                rows.append(r)
                alleles += a
            called = c.Stats()["TotalNumCalled"]
    got = np.concatenate(rows)
    # (the oracle's region: the blocks the reads touch — it emits zero-coverage reference rows over all of its region, the state manager
    # only over blocks that exist)
    bs = kw["block_size"]
    reach = max(r["pos"] + sum(l for o, l in r["cigar"] if o in "MDN=X") - 1 for r in reads)
    region = min(len(ref), (reach + bs - 1) // bs * bs)
    exp, exp_alleles, exp_called = orc.run_reads_schedule(_abi.ReadBatch(reads), np.frombuffer(ref, np.uint8), 1, region, cfg, schedule, forced=forced or (),
                                                          intervals=intervals, host_candidates=host_cands)
    exp_all, exp_alleles_all = exp, exp_alleles
    if intervals and not (kw["call_mnvs"] or (kw["collapse"] and (kw["collapse_freq_threshold"] > 0 or kw["collapse_freq_ratio_threshold"] >= 1))):
        # MNV calling off, SNVs from the allele counts: the tile kernels run over the intervals only, so TotalNumCalled lacks the callable SNVs
        # OUTSIDE them, which the reference counts and does not report; the rows are all there
        exp_called = called
    why = None
    if alleles != exp_alleles:
        why = "alleles"
        if verbose:
            ga = set(zip(got["position"].tolist(), alleles)); ea = set(zip(exp["position"].tolist(), exp_alleles))
            print("  only product:", sorted(ga - ea)[:10]); print("  only oracle:", sorted(ea - ga)[:10])
            pa, ea_ = list(zip(got["position"].tolist(), alleles)), list(zip(exp["position"].tolist(), exp_alleles))
            k = next((i for i in range(min(len(pa), len(ea_))) if pa[i] != ea_[i]), min(len(pa), len(ea_)))
            print("  first difference at row", k, "product", pa[max(0, k - 2):k + 3], "oracle", ea_[max(0, k - 2):k + 3], "host candidates", host_cands, "forced", forced)
            print("  cuts", cuts, "schedule", schedule, "read positions at cuts", [reads[c - 1]["pos"] for c in cuts])
            for (p, (ra, aa)) in sorted((ga - ea) | (ea - ga))[:6]:
                if len(ra) != 1 or len(aa) != 1:
                    continue
                shown = {}
                for ri, r in enumerate(reads):
                    at, k = r["pos"], 0
                    for o, l in r["cigar"]:
                        if o in "M=X":
                            if at <= p < at + l and r["seq"][k + p - at] == ord(aa):
                                key = (o, "q>=" if r["quals"][k + p - at] >= kw["min_base_call_quality"] else "q<", sum(1 for c in cuts if c <= ri))
                                shown[key] = shown.get(key, 0) + 1
                            at, k = at + l, k + l
                        elif o in "DN":
                            at += l
                        elif o in "IS":
                            k += l
                print("   ", p, ra, aa, "shown as (op, quality, batch):", shown)
    elif len(got) != len(exp):
        why = "rows"
    else:
        for f in list(INT_FIELDS) + ["info", "variant_qscore", "genotype_qscore"]:
            if not (got[f] == exp[f]).all():
                why = f
                if verbose:
                    bad = np.nonzero((got[f] != exp[f]).reshape(len(got), -1).any(axis=1))[0][:5]
                    for i in bad:
                        print("  ", f, int(got["position"][i]), alleles[i], got[f][i], exp[f][i])
                        p, aa = int(got["position"][i]), alleles[i][1]
                        for ri, r in enumerate(reads):
                            at, k = r["pos"], 0
                            for ci, (o, l) in enumerate(r["cigar"]):
                                if o in "M=X":
                                    if at <= p < at + l and len(aa) == 1 and r["seq"][k + p - at] == ord(aa):
                                        print("     read", ri, "pos", r["pos"], r["cigar"], "op", ci, "offset", p - at, "q", r["quals"][k + p - at],
                                              "neighbours", r["seq"][max(0, k + p - at - 1):k + p - at + 2], r["quals"][max(0, k + p - at - 1):k + p - at + 2], ref[p - 2:p + 1])
                                    at, k = at + l, k + l
                                elif o in "DN":
                                    at += l
                                elif o in "IS":
                                    k += l
                break
        if why is None and called != exp_called:
            why = "TotalNumCalled %d != %d" % (called, exp_called)
    if why and form[0] != "default":
        why = form[0] + ": " + why
    if rows_too:
        return why, kw, (got, alleles, exp_all, exp_alleles_all, intervals, form), forced
    return why, kw, len(got), forced


def one_tuples(seed, verbose=False):
    """The device-resident surface (pisces_hip_call_tiles: bucketed observation tuples -> records, BASELINE config 2's path): random
    loci x depth, allele mix, qualities, directions and anchors, tile sizes and unaligned segments, a reference with N and homopolymers,
    and every threshold of the configuration, against the oracle over the same observations.  Returns (why or None, overrides, rows)."""
    import torch
    from types import SimpleNamespace
    from pisces_amd import engine
    from tests.test_gpu_parity import run_fused, INT_FIELDS
    rng = np.random.default_rng(770000 + seed)
    n_loci, start = int(rng.integers(1, 700)), int(rng.choice([1, 11, 1000]))
    ref = rng.choice(list(b"ACGT"), start - 1 + n_loci + 40).astype(np.uint8)
    if seed % 3 == 0:
        a = int(rng.integers(0, len(ref) - 30)); ref[a:a + int(rng.integers(1, 20))] = ord("N")
        b = int(rng.integers(0, len(ref) - 30)); ref[b:b + 12] = ord("A"); ref[b + 12:b + 24] = ord("C")
    depth = int(rng.choice([3, 30, 300, 1500, 6000 if n_loci < 120 else 300]))
    n_obs = n_loci * depth
    pos = rng.integers(start, start + n_loci, n_obs).astype(np.int32)
    if seed % 4 == 1:   # zero-coverage loci and a wholly empty stretch
        pos = pos[(pos % 7 != 3) & ((pos < start + n_loci // 3) | (pos >= start + n_loci // 3 + min(70, n_loci // 4)))]
    refa = np.array([_abi.ALLELE_OF_BASE.get(chr(c), 4) for c in ref], np.int64)[pos - 1]
    vaf = rng.choice([0.0, 0.002, 0.01, 0.05, 0.3, 0.6, 1.0], start + n_loci + 1)[pos]
    alt = ((refa + 1 + rng.integers(0, 3, len(pos))) % 4)
    allele = np.where(rng.random(len(pos)) < vaf, alt, refa)
    noise = rng.random(len(pos))
    allele = np.where(noise < 0.003, rng.integers(0, 4, len(pos)), allele)          # errors
    allele = np.where((noise > 0.99) & (noise < 0.995), 4, allele)                   # N
    allele = np.where(noise > 0.995, 5, allele)                                      # deletion
    qual = np.where(allele == 5, 255, rng.choice([2, 19, 20, 30, 37, 41], len(pos), p=[.02, .03, .05, .1, .7, .1]))
    dirs = rng.choice(3, len(pos), p=[.48, .48, .04]) if seed % 5 else rng.choice(3, len(pos), p=[.95, .03, .02])
    tup = _abi.tuple_pack(np.zeros(len(pos), np.uint32), rng.integers(0, 11, len(pos)), dirs, allele, qual)
    ploidy = int(rng.choice([0, 0, 0, 1, 2]))
    kw = dict(include_reference_calls=int(rng.integers(0, 2)), ploidy=ploidy, strand_bias_model=int(rng.choice([1, 1, 2])),
              min_frequency=0.2 if ploidy else float(rng.choice([0.005, 0.01, 0.05])), min_base_call_quality=int(rng.choice([20, 20, 13, 30, 0])),
              min_coverage=int(rng.choice([10, 1, 0, 50])), low_depth_filter=int(rng.choice([10, 30, -1])), emit_zero_coverage_refs=int(rng.integers(0, 2)),
              filter_single_strand=int(rng.integers(0, 2)), min_variant_qscore=int(rng.choice([20, 0, 40])), variant_qscore_filter=int(rng.choice([30, 20, 60])),
              expect_stitched_reads=int(rng.choice([0, 0, 1])), noise_level=int(rng.choice([20, 30, 15])),
              rmxn_min_repetitions=int(rng.choice([9, 4])), strand_bias_threshold=float(rng.choice([0.5, 0.1, 0.9])),
              no_call_filter_threshold=float(rng.choice([0.6, 0.02, -1.0])), rmxn_max_repeat_length=int(rng.choice([5, 2, -1])),
              rmxn_frequency_limit=float(rng.choice([0.35, 1.0])), genotype_min_freq_filter=float(rng.choice([0.01, 0.05])),
              target_lod_frequency=float(rng.choice([0.01, 0.05])), min_genotype_qscore=int(rng.choice([0, 10])),
              max_variant_qscore=int(rng.choice([100, 60, 3000])), low_gq_filter=int(rng.choice([-1, 20, 50])))
    if kw["variant_qscore_filter"] < kw["min_variant_qscore"]:
        kw["variant_qscore_filter"] = kw["min_variant_qscore"]
    if ploidy:
        kw.update(variant_freq_filter=0.2, low_gq_filter=30, max_genotype_qscore=1000,
                  diploid_snv_params=[float(x) for x in rng.choice([[0.20, 0.70, 0.80], [0.10, 0.60, 0.90]])])
    else:
        kw.update(variant_freq_filter=float(rng.choice([0.01, 0.03, 0.1])), max_genotype_qscore=int(rng.choice([100, 40])))
    cfg = _abi.default_config(**kw)
    exp, _ = orc.run_observations(pos, tup, ref, start, n_loci, cfg)
    tile = int(rng.choice([64, 64, 56, 33, 7]))
    n_tiles = (n_loci + tile - 1) // tile
    tiles = np.zeros(n_tiles, dtype=_abi.TILE_DTYPE)
    pad = bool(rng.integers(0, 2))
    segs, cursor = [np.full(1 if not pad else 4, _abi.TUPLE_PAD, np.uint32)], 1 if not pad else 4
    for t in range(n_tiles):
        l0, l1 = t * tile, min(n_loci, t * tile + tile)
        m = (pos >= start + l0) & (pos < start + l1)
        seg = _abi.tuple_with_locus(tup[m], pos[m] - (start + l0))
        if pad and len(seg) % 4:
            seg = np.concatenate([seg, np.full(4 - len(seg) % 4, _abi.TUPLE_PAD, np.uint32)])
        tiles[t] = (start + l0, l1 - l0, cursor, cursor + len(seg))
        segs.append(seg)
        cursor += len(seg)
    view = SimpleNamespace(tuples=torch.from_numpy(np.concatenate(segs).view(np.int32)).cuda(), tiles=torch.from_numpy(tiles.view(np.uint8)).cuda(),
                           n_tiles=n_tiles, ref=torch.from_numpy(ref).cuda(), ref_len=len(ref))
    with engine.HipVariantCaller(cfg) as caller:
        got, _ = run_fused(torch, caller, view, compact=bool(seed % 2))
    why = None
    if len(got) != len(exp):
        why = "rows %d != %d" % (len(got), len(exp))
    else:
        for f in list(INT_FIELDS) + ["info", "variant_qscore", "genotype_qscore"]:
            if not (got[f] == exp[f]).all():
                why = f
                if verbose:
                    for i in np.nonzero((got[f] != exp[f]).reshape(len(got), -1).any(axis=1))[0][:5]:
                        print("  ", f, int(got["position"][i]), got[f][i], exp[f][i], "cov", int(got["total_coverage"][i]), "sup", int(got["allele_support"][i]))
                break
        if why is None and not np.allclose(got["strand_bias_score"], exp["strand_bias_score"], rtol=1e-9, atol=1e-12, equal_nan=True):
            why = "strand_bias_score"
    return why, kw, len(got)
