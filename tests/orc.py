"""ctypes view of oracle/libpiscesoracle.so — TEST INFRASTRUCTURE (the CPU restatement of the
reference path).  Product code never imports this module."""
import ctypes as C
import os
import subprocess

import numpy as np

from pisces_amd import _abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORC_MAX_ALLELE = 320


class OrcRead(C.Structure):
    _fields_ = [
        ("position", C.c_int32),
        ("n_cigar", C.c_int32),
        ("cigar_op", C.POINTER(C.c_uint8)),
        ("cigar_len", C.POINTER(C.c_uint32)),
        ("read_len", C.c_int32),
        ("bases", C.POINTER(C.c_uint8)),
        ("quals", C.POINTER(C.c_uint8)),
        ("dirs", C.POINTER(C.c_uint8)),
        ("is_reverse", C.c_int32),
        ("posmap_override", C.POINTER(C.c_int32)),
        ("expanded_dirs", C.POINTER(C.c_uint8)),
        ("n_expanded", C.c_int32),
    ]


class OrcCandidate(C.Structure):
    _fields_ = [
        ("position", C.c_int32),
        ("category", C.c_int32),
        ("ref", C.c_char * ORC_MAX_ALLELE),
        ("alt", C.c_char * ORC_MAX_ALLELE),
        ("support_by_dir", C.c_int32 * 3),
        ("well_anchored_by_dir", C.c_int32 * 3),
        ("open_left", C.c_int32),
        ("open_right", C.c_int32),
        ("next", C.c_int32),
    ]


class OrcSbStats(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("chance_false_neg", "chance_false_pos", "chance_var_freq_gt_zero",
                                          "coverage", "frequency", "support")]


class OrcBiasResults(C.Structure):
    _fields_ = [
        ("bias_score", C.c_double),
        ("gatk_bias_score", C.c_double),
        ("bias_acceptable", C.c_int32),
        ("var_present_on_both", C.c_int32),
        ("cov_present_on_both", C.c_int32),
        ("forward", OrcSbStats),
        ("reverse", OrcSbStats),
        ("overall", OrcSbStats),
        ("stitched", OrcSbStats),
    ]


class OrcCalled(C.Structure):
    _fields_ = [
        ("position", C.c_int32), ("category", C.c_int32),
        ("ref", C.c_char * ORC_MAX_ALLELE), ("alt", C.c_char * ORC_MAX_ALLELE),
        ("support_by_dir", C.c_int32 * 3), ("well_anchored_by_dir", C.c_int32 * 3),
        ("allele_support", C.c_int32), ("well_anchored_support", C.c_int32),
        ("total_coverage", C.c_int32), ("reference_support", C.c_int32), ("num_no_calls", C.c_int32),
        ("coverage_by_dir", C.c_int32 * 3),
        ("confident_start", C.c_int32), ("confident_end", C.c_int32),
        ("suspicious_start", C.c_int32), ("suspicious_end", C.c_int32),
        ("unanchored_weight", C.c_double),
        ("sum_of_base_quality", C.c_double),
        ("variant_qscore", C.c_int32), ("noise_level_applied", C.c_int32),
        ("fraction_no_calls", C.c_float),
        ("sb", OrcBiasResults),
        ("has_sb", C.c_int32),
        ("filters", C.c_uint32),
        ("genotype", C.c_int32), ("genotype_qscore", C.c_int32),
    ]


class OrcShardJob(C.Structure):
    _fields_ = [("batch", C.POINTER(_abi.PiscesReadBatch)), ("ref_bases", C.POINTER(C.c_uint8)), ("ref_len", C.c_int64),
                ("region_start", C.c_int32), ("region_loci", C.c_int32), ("cfg", C.POINTER(_abi.PiscesHipConfig)),
                ("out", C.c_void_p), ("capacity", C.c_int64), ("passes", C.c_int32), ("pad", C.c_int32),
                ("n_out", C.c_int64), ("n_loci", C.c_int64)]


def _load():
    so = os.path.join(ROOT, "oracle", "libpiscesoracle.so")
    src = os.path.join(ROOT, "oracle", "pisces_oracle.c")
    if not os.path.exists(so) or (os.path.exists(src) and os.path.getmtime(so) < os.path.getmtime(src)):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
    L = C.CDLL(so)
    d, i32, i64, f32 = C.c_double, C.c_int32, C.c_int64, C.c_float
    P = C.POINTER
    sig = {
        "orc_poisson_cdf": (d, [d, d]),
        "orc_q_to_p": (d, [d]),
        "orc_p_to_q": (d, [d]),
        "orc_mathnet_gamma_ln": (d, [d]),
        "orc_mathnet_gamma_lower_regularized": (d, [d, d]),
        "orc_mathnet_poisson_cdf": (d, [d, d]),
        "orc_mathnet_poisson_ln_pmf": (d, [d, i32]),
        "orc_assign_pvalue": (d, [i32, i32, i32]),
        "orc_raw_poisson_qscore": (d, [i32, i32, i32]),
        "orc_poisson_qscore": (i32, [i32, i32, i32, i32]),
        "orc_strand_bias": (None, [P(i32), P(i32), i32, d, d, i32, P(OrcBiasResults)]),
        "orc_somatic_genotype": (i32, [i32, i32, i32, i32, f32, i32]),
        "orc_somatic_gq": (i32, [i32, i32, i32, i32, f32, i32, i32]),
        "orc_state_create": (C.c_void_p, [i32, i32, i32, i32, i32]),
        "orc_state_destroy": (None, [C.c_void_p]),
        "orc_add_allele_counts": (i32, [C.c_void_p, P(OrcRead)]),
        "orc_get_allele_count": (i32, [C.c_void_p, i32, i32, i32, i32, i32, i32, i32]),
        "orc_get_sum_base_quality": (d, [C.c_void_p, i32, i32, i32, i32, i32, i32, i32]),
        "orc_counts_ptr": (P(i32), [C.c_void_p]),
        "orc_num_anchor_indexes": (i32, [C.c_void_p]),
        "orc_add_gapped_mnv_ref": (None, [C.c_void_p, i32, i32]),
        "orc_add_candidate": (i32, [C.c_void_p, P(OrcCandidate)]),
        "orc_num_candidates": (i32, [C.c_void_p]),
        "orc_get_candidates": (i32, [C.c_void_p, P(OrcCandidate), i32]),
        "orc_find_candidates": (i32, [P(OrcRead), P(C.c_uint8), i64, i32, i32, i32, i32, i32, P(OrcCandidate), i32]),
        "orc_check_deletion_quality": (i32, [P(OrcRead), i32, i32]),
        "orc_deletion_direction_for_stitched_read": (i32, [P(OrcRead), i32, i32]),
        "orc_coverage_compute": (None, [P(OrcCalled), C.c_void_p, i32, i32]),
        "orc_called_from_candidate": (None, [P(OrcCalled), P(OrcCandidate)]),
        "orc_process_variant": (None, [P(OrcCalled), C.c_void_p, P(_abi.PiscesHipConfig)]),
        "orc_call_all": (i64, [C.c_void_p, P(C.c_uint8), i64, P(_abi.PiscesHipConfig), C.c_void_p, i64, P(OrcCalled),
                               P(i64)]),
        "orc_call_candidates": (i64, [C.c_void_p, P(OrcCandidate), i64, P(C.c_uint8), i64, P(_abi.PiscesHipConfig), C.c_void_p, i64,
                                      P(OrcCalled), P(i64)]),
        "orc_collapse": (i32, [P(OrcCandidate), i32, C.c_void_p, f32, f32, i32, i32, i32, i32, P(i32), P(OrcCandidate), P(i32)]),
        "orc_run_reads": (i64, [P(_abi.PiscesReadBatch), P(C.c_uint8), i64, i32, i32, P(_abi.PiscesHipConfig),
                                C.c_void_p, i64, P(i64)]),
        "orc_run_reads_full": (i64, [P(_abi.PiscesReadBatch), P(C.c_uint8), i64, i32, i32, P(_abi.PiscesHipConfig),
                                     C.c_void_p, i64, P(i64), P(OrcCalled), P(i64)]),
        "orc_run_observations": (i64, [P(i32), P(C.c_uint32), i64, P(C.c_uint8), i64, i32, i32,
                                       P(_abi.PiscesHipConfig), C.c_void_p, i64, P(i64)]),
        "orc_default_config": (None, [P(_abi.PiscesHipConfig)]),
        "orc_run_reads_sharded": (i32, [P(OrcShardJob), i32]),
        "orc_track_blocks": (None, [C.c_void_p, i32]),
        "orc_next_batch": (i32, [C.c_void_p, i32, P(i32), P(i32)]),
        "orc_batch_candidates": (i32, [C.c_void_p, i32, i32, i32, P(OrcCandidate), i32, P(i32)]),
        "orc_done_processing": (None, [C.c_void_p, i32]),
    }
    for name, (res, args) in sig.items():
        f = getattr(L, name)
        f.restype = res
        f.argtypes = args
    return L


lib = _load()


def make_read(pos, seq, cigar=None, quals=None, qual_all=30, reverse=False, dirs=None, posmap=None, expanded_dirs=None):
    """ReadTestHelper.CreateRead (src/test/TestUtilities/ReadTestHelper.cs:156-173): default Q30, <len>M."""
    seq_b = np.frombuffer(seq.encode() if isinstance(seq, str) else bytes(seq), dtype=np.uint8).copy()   # (bytes: any base value)
    n = len(seq_b)
    if cigar is None:
        cigar = [("M", n)]
    elif isinstance(cigar, str):
        cigar = parse_cigar(cigar)
    ops = np.array([ord(o) for o, _ in cigar], dtype=np.uint8)
    lens = np.array([l for _, l in cigar], dtype=np.uint32)
    q = np.array(quals if quals is not None else [qual_all] * n, dtype=np.uint8)
    keep = [seq_b, ops, lens, q]
    r = OrcRead()
    r.position = pos
    r.n_cigar = len(cigar)
    r.cigar_op = ops.ctypes.data_as(C.POINTER(C.c_uint8))
    r.cigar_len = lens.ctypes.data_as(C.POINTER(C.c_uint32))
    r.read_len = n
    r.bases = seq_b.ctypes.data_as(C.POINTER(C.c_uint8))
    r.quals = q.ctypes.data_as(C.POINTER(C.c_uint8))
    if dirs is not None:
        da = np.array(dirs, dtype=np.uint8)
        keep.append(da)
        r.dirs = da.ctypes.data_as(C.POINTER(C.c_uint8))
    r.is_reverse = 1 if reverse else 0
    if expanded_dirs is not None:   # Read.CigarDirections.Expand()
        ea = np.array(expanded_dirs, dtype=np.uint8)
        keep.append(ea)
        r.expanded_dirs = ea.ctypes.data_as(C.POINTER(C.c_uint8))
        r.n_expanded = len(ea)
    if posmap is not None:
        pm = np.array(posmap, dtype=np.int32)
        keep.append(pm)
        r.posmap_override = pm.ctypes.data_as(C.POINTER(C.c_int32))
    r._keep = keep
    return r


def parse_cigar(s):
    out, num = [], ""
    for ch in s:
        if ch.isdigit():
            num += ch
        else:
            out.append((ch, int(num)))
            num = ""
    return out


class State:
    """Dense-window RegionStateManager of the oracle."""

    def __init__(self, start, n_loci, min_bq=20, anchor_size=5, track_open_ended=False):
        self.h = lib.orc_state_create(start, n_loci, min_bq, anchor_size, 1 if track_open_ended else 0)
        self.start, self.n_loci = start, n_loci

    def __del__(self):
        if getattr(self, "h", None):
            lib.orc_state_destroy(self.h)
            self.h = None

    def add_allele_counts(self, read):
        return lib.orc_add_allele_counts(self.h, C.byref(read))

    def get_allele_count(self, pos, allele, direction, min_anchor=0, max_anchor=-1, from_end=False, symmetric=False):
        return lib.orc_get_allele_count(self.h, pos, allele, direction, min_anchor, max_anchor, int(from_end),
                                        int(symmetric))

    def counts(self):
        na = lib.orc_num_anchor_indexes(self.h)
        p = lib.orc_counts_ptr(self.h)
        return np.ctypeslib.as_array(p, shape=(self.n_loci, 6, 3, na)).copy()

    def set_count(self, pos, allele, direction, anchor, value):
        na = lib.orc_num_anchor_indexes(self.h)
        p = lib.orc_counts_ptr(self.h)
        a = np.ctypeslib.as_array(p, shape=(self.n_loci, 6, 3, na))
        a[pos - self.start, allele, direction, anchor] = value

    def add_candidate(self, cand):
        return lib.orc_add_candidate(self.h, C.byref(cand))

    def candidates(self):
        n = lib.orc_num_candidates(self.h)
        arr = (OrcCandidate * max(n, 1))()
        lib.orc_get_candidates(self.h, arr, n)
        return [arr[i] for i in range(n)]

    def track_blocks(self, block_size=1000):
        lib.orc_track_blocks(self.h, C.c_int32(block_size))

    def next_batch(self, up_to):
        """GetCandidatesToProcess's block choice: None (no batch), () (a batch without cleared blocks) or (first, last) positions."""
        first, last = C.c_int32(0), C.c_int32(0)
        rc = lib.orc_next_batch(self.h, C.c_int32(-1 if up_to is None else up_to), C.byref(first), C.byref(last))
        return None if rc < 0 else () if rc == 0 else (first.value, last.value)

    def batch_candidates(self, first, last, up_to):
        cap = lib.orc_num_candidates(self.h) + 1
        arr = (OrcCandidate * cap)()
        other = C.c_int32(0)
        n = lib.orc_batch_candidates(self.h, C.c_int32(first), C.c_int32(last), C.c_int32(-1 if up_to is None else up_to), arr, C.c_int32(cap),
                                     C.byref(other))
        return [arr[i] for i in range(n)], bool(other.value)

    def done_processing(self, last):
        lib.orc_done_processing(self.h, C.c_int32(last))

    def call_all(self, ref_bases, cfg, want_full=False):
        ref = np.frombuffer(ref_bases if isinstance(ref_bases, bytes) else ref_bases.encode(), dtype=np.uint8)
        cap = self.n_loci * 5 + 16
        out = np.zeros(cap, dtype=_abi.CALLED_ALLELE_DTYPE)
        full = (OrcCalled * cap)() if want_full else None
        total = C.c_int64(0)
        n = lib.orc_call_all(self.h, ref.ctypes.data_as(C.POINTER(C.c_uint8)), len(ref), C.byref(cfg),
                             out.ctypes.data, cap, full, C.byref(total))
        assert n >= 0, n
        if want_full:
            return out[:n], [full[i] for i in range(n)], total.value
        return out[:n]


def call_candidates(state, cands, cfg, ref_bases=b""):
    """AlleleCaller.Call over an explicit candidate batch; returns (records, full OrcCalled list, TotalNumCalled)."""
    ref = np.frombuffer(ref_bases, dtype=np.uint8) if len(ref_bases) else np.zeros(1, np.uint8)
    arr = (OrcCandidate * max(len(cands), 1))(*cands)
    cap = len(cands) + 16
    out = np.zeros(cap, dtype=_abi.CALLED_ALLELE_DTYPE)
    full = (OrcCalled * cap)()
    total = C.c_int64(0)
    n = lib.orc_call_candidates(state.h, arr, len(cands), ref.ctypes.data_as(C.POINTER(C.c_uint8)), len(ref_bases), C.byref(cfg),
                                out.ctypes.data, cap, full, C.byref(total))
    assert n >= 0, n
    return out[:n], [full[i] for i in range(n)], total.value


def set_known_variants(variants):
    """orc_set_known_variants: [(position, category, ref, alt)] the next collapse / schedule run annotates with; [] clears.  Returns the
    ctypes array (keep it alive while it is set)."""
    arr = (OrcCandidate * max(len(variants), 1))(*[make_candidate(p, cat, r, a) for (p, cat, r, a) in variants])
    lib.orc_set_known_variants.restype = None
    lib.orc_set_known_variants(arr if variants else None, C.c_int32(len(variants)))
    return arr


def set_exclude_mnvs_from_collapsing(on):
    """orc_set_exclude_mnvs_from_collapsing: PiscesApplicationOptions.ExcludeMNVsFromCollapsing for the schedule runs."""
    lib.orc_set_exclude_mnvs_from_collapsing.restype = None
    lib.orc_set_exclude_mnvs_from_collapsing(C.c_int32(1 if on else 0))


def collapse(state, cands, freq_threshold=0.0, freq_ratio_threshold=0.0, exclude_mnvs=False, consider_anchors=True, expect_stitched=False,
             max_cleared_position=None):
    """VariantCollapser.Collapse on a list of OrcCandidate; returns (collapsed list, TotalNumCollapsed, added back)."""
    n = len(cands)
    arr = (OrcCandidate * max(n, 1))(*cands)
    back = (OrcCandidate * max(n, 1))()
    nc, nb = C.c_int32(0), C.c_int32(0)
    m = lib.orc_collapse(arr, n, state.h, freq_threshold, freq_ratio_threshold, int(exclude_mnvs), int(consider_anchors), int(expect_stitched),
                         -1 if max_cleared_position is None else max_cleared_position, C.byref(nc), back, C.byref(nb))
    return [arr[i] for i in range(m)], nc.value, [back[i] for i in range(nb.value)]


def make_candidate(pos, category, ref, alt, support=(0, 0, 0), well_anchored=(0, 0, 0), open_left=False,
                   open_right=False):
    c = OrcCandidate()
    c.position = pos
    c.category = category
    c.ref = ref.encode()
    c.alt = alt.encode()
    for i in range(3):
        c.support_by_dir[i] = support[i]
        c.well_anchored_by_dir[i] = well_anchored[i]
    c.open_left = int(open_left)
    c.open_right = int(open_right)
    c.next = -1
    return c


def strand_bias(cov, sup, q_noise=20, min_vf=0.01, acceptance=0.5, model=_abi.SB_EXTENDED):
    r = OrcBiasResults()
    lib.orc_strand_bias((C.c_int32 * 3)(*cov), (C.c_int32 * 3)(*sup), q_noise, min_vf, acceptance, model, C.byref(r))
    return r


def find_candidates(read, ref, min_bq=20, max_mnv=3, max_gap=1, call_mnvs=False, anchor_size=5):
    refa = np.frombuffer(ref.encode() if isinstance(ref, str) else ref, dtype=np.uint8)
    out = (OrcCandidate * 256)()
    n = lib.orc_find_candidates(C.byref(read), refa.ctypes.data_as(C.POINTER(C.c_uint8)), len(refa), min_bq, max_mnv,
                                max_gap, int(call_mnvs), anchor_size, out, 256)
    assert n >= 0, n
    return [out[i] for i in range(n)]


def run_observations(positions, tuples, ref, region_start, region_loci, cfg):
    positions = np.ascontiguousarray(positions, np.int32)
    tuples = np.ascontiguousarray(tuples, np.uint32)
    refa = np.ascontiguousarray(ref, np.uint8)
    cap = region_loci * 5 + 16
    out = np.zeros(cap, dtype=_abi.CALLED_ALLELE_DTYPE)
    nloci = C.c_int64(0)
    n = lib.orc_run_observations(positions.ctypes.data_as(C.POINTER(C.c_int32)),
                                 tuples.ctypes.data_as(C.POINTER(C.c_uint32)), len(tuples),
                                 refa.ctypes.data_as(C.POINTER(C.c_uint8)), len(refa), region_start, region_loci,
                                 C.byref(cfg), out.ctypes.data, cap, C.byref(nloci))
    assert n >= 0, n
    return out[:n], nloci.value


def run_reads(batch, ref, region_start, region_loci, cfg):
    refa = np.ascontiguousarray(ref, np.uint8)
    cap = region_loci * 5 + 16
    out = np.zeros(cap, dtype=_abi.CALLED_ALLELE_DTYPE)
    nloci = C.c_int64(0)
    n = lib.orc_run_reads(C.byref(batch.c), refa.ctypes.data_as(C.POINTER(C.c_uint8)), len(refa), region_start,
                          region_loci, C.byref(cfg), out.ctypes.data, cap, C.byref(nloci))
    assert n >= 0, n
    return out[:n], nloci.value


def run_reads_full(batch, ref, region_start, region_loci, cfg):
    """run_reads + the full CalledAllele working set (allele strings) + TotalNumCalled."""
    refa = np.ascontiguousarray(ref, np.uint8)
    cap = region_loci * 5 + 16
    out = np.zeros(cap, dtype=_abi.CALLED_ALLELE_DTYPE)
    full = (OrcCalled * cap)()
    nloci, total = C.c_int64(0), C.c_int64(0)
    n = lib.orc_run_reads_full(C.byref(batch.c), refa.ctypes.data_as(C.POINTER(C.c_uint8)), len(refa), region_start,
                               region_loci, C.byref(cfg), out.ctypes.data, cap, C.byref(nloci), full, C.byref(total))
    assert n >= 0, n
    return out[:n], [(full[i].ref.decode("latin-1"), full[i].alt.decode("latin-1")) for i in range(n)], nloci.value, total.value


def run_reads_sharded(shards, ref, cfg, passes=1):
    """shards: list of (ReadBatch, region_start, region_loci); one host thread per shard inside the oracle library.
    Returns (list of record arrays in shard order, candidate loci summed over shards and passes)."""
    refa = np.ascontiguousarray(ref, np.uint8)
    jobs = (OrcShardJob * len(shards))()
    outs = []
    for j, (b, start, loci) in zip(jobs, shards):
        cap = loci * 5 + 16
        out = np.zeros(cap, dtype=_abi.CALLED_ALLELE_DTYPE)
        outs.append(out)
        j.batch = C.pointer(b.c)
        j.ref_bases = refa.ctypes.data_as(C.POINTER(C.c_uint8))
        j.ref_len = len(refa)
        j.region_start, j.region_loci = start, loci
        j.cfg = C.pointer(cfg)
        j.out, j.capacity, j.passes = out.ctypes.data, cap, passes
    rc = lib.orc_run_reads_sharded(jobs, len(shards))
    assert rc == 0, rc
    return [o[: j.n_out].copy() for o, j in zip(outs, jobs)], int(sum(j.n_loci for j in jobs))


def reallocate_failed_mnvs(failed, callable_, max_position=None):
    """MnvReallocator.ReallocateFailedMnvs on lists of dicts {position, ref, alt, support, dirs, category}; returns (callable, outside)."""
    def mk(d):
        v = OrcCalled()
        v.position, v.category = d["position"], d["category"]
        v.ref, v.alt = d["ref"].encode(), d["alt"].encode()
        v.allele_support = d["support"]
        for k in range(3):
            v.support_by_dir[k] = d["dirs"][k]
        return v

    def un(v):
        return {"position": v.position, "ref": v.ref.decode(), "alt": v.alt.decode(), "support": v.allele_support,
                "dirs": [v.support_by_dir[k] for k in range(3)], "category": v.category}
    cap = len(callable_) + 64 * max(len(failed), 1)
    f = (OrcCalled * max(len(failed), 1))(*[mk(d) for d in failed])
    c = (OrcCalled * cap)(*[mk(d) for d in callable_])
    o = (OrcCalled * cap)()
    no = C.c_int64(0)
    lib.orc_reallocate_failed_mnvs.restype = C.c_int64
    nc = lib.orc_reallocate_failed_mnvs(f, C.c_int64(len(failed)), c, C.c_int64(len(callable_)), C.c_int64(cap),
                                        C.c_int32(-1 if max_position is None else max_position), o, C.c_int64(cap), C.byref(no))
    assert nc <= cap and no.value <= cap
    return [un(c[i]) for i in range(nc)], [un(o[i]) for i in range(no.value)]


# ---- diploid (germline) pieces ----
lib.orc_diploid_gq.restype = C.c_int32
lib.orc_diploid_gq.argtypes = [C.c_int32] * 5
lib.orc_sb_populate_diploid_stats.restype = None
lib.orc_sb_populate_diploid_stats.argtypes = [C.c_double, C.c_double, C.c_double, C.POINTER(C.c_double)]
lib.orc_diploid_set_genotypes.restype = C.c_int32


def diploid_gq(genotype, total_coverage, allele_support, min_q=0, max_q=2147483647):
    return lib.orc_diploid_gq(genotype, total_coverage, allele_support, min_q, max_q)


def diploid_sb_stats(support, coverage, min_detectable):
    out = (C.c_double * 3)()
    lib.orc_sb_populate_diploid_stats(support, coverage, min_detectable, out)
    return list(out)


def diploid_set_genotypes(alleles, snv=(0.20, 0.70, 0.80), indel=(0.20, 0.70, 0.80), min_depth=100, min_gq=0, max_gq=0):
    """alleles: list of dicts {category, ref, alt, support, coverage, ref_support}; returns (locus genotype, prune flags, per-allele
    (genotype, gq, filters, phase))."""
    n = len(alleles)
    arr = (OrcCalled * max(n, 1))()
    for i, d in enumerate(alleles):
        arr[i].category = d["category"]
        arr[i].ref, arr[i].alt = d["ref"].encode(), d["alt"].encode()
        arr[i].allele_support, arr[i].total_coverage, arr[i].reference_support = d["support"], d["coverage"], d["ref_support"]
    phase = (C.c_int32 * max(n, 1))()
    prune = (C.c_uint8 * max(n, 1))()
    gt = lib.orc_diploid_set_genotypes(arr, n, (C.c_float * 3)(*snv), (C.c_float * 3)(*indel), min_depth, min_gq, max_gq, phase, prune)
    return gt, [int(prune[i]) for i in range(n)], [(arr[i].genotype, arr[i].genotype_qscore, arr[i].filters, phase[i]) for i in range(n)]


lib.orc_haploid_set_genotypes.restype = C.c_int32


def haploid_set_genotypes(alleles, minor_vf=0.20, major_vf=0.70, min_depth=100, min_gq=0, max_gq=100):
    n = len(alleles)
    arr = (OrcCalled * max(n, 1))()
    for i, d in enumerate(alleles):
        arr[i].category = d["category"]
        arr[i].ref, arr[i].alt = d["ref"].encode(), d["alt"].encode()
        arr[i].allele_support, arr[i].total_coverage, arr[i].reference_support = d["support"], d["coverage"], d["ref_support"]
    prune = (C.c_uint8 * max(n, 1))()
    gt = lib.orc_haploid_set_genotypes(arr, n, C.c_float(minor_vf), C.c_float(major_vf), min_depth, min_gq, max_gq, prune)
    return gt, [int(prune[i]) for i in range(n)], [(arr[i].genotype, arr[i].genotype_qscore) for i in range(n)]


def run_reads_blocks(batch, ref, region_start, region_loci, cfg):
    """run_reads_full with the block schedule (one batch per block of the cfg.block_size grid, MaxClearedPosition = its end)."""
    refa = np.ascontiguousarray(ref, np.uint8)
    cap = region_loci * 5 + 16
    out = np.zeros(cap, dtype=_abi.CALLED_ALLELE_DTYPE)
    full = (OrcCalled * cap)()
    total = C.c_int64(0)
    lib.orc_run_reads_blocks.restype = C.c_int64
    n = lib.orc_run_reads_blocks(C.byref(batch.c), refa.ctypes.data_as(C.POINTER(C.c_uint8)), C.c_int64(len(refa)), C.c_int32(region_start),
                                 C.c_int32(region_loci), C.byref(cfg), C.c_void_p(out.ctypes.data), C.c_int64(cap), full, C.byref(total))
    assert n >= 0, n
    return out[:n], [(full[i].ref.decode("latin-1"), full[i].alt.decode("latin-1")) for i in range(n)], total.value


def run_reads_schedule(batch, ref, region_start, region_loci, cfg, up_to_positions, forced=(), intervals=None, host_candidates=()):
    """run_reads_full with GetCandidatesToProcess(upTo) for every upTo of the list and then the final batch (RegionStateManager.cs:283-334,
    with AddCollapsableFromOtherBlocks); forced = [(position, ref, alt)] forced genotyping alleles of the chromosome; intervals = the
    ChrIntervalSet [(first, last)] (sorted, disjoint, inclusive) or None; host_candidates = candidate dicts (engine.AddCandidates' form) the
    host hands in behind the reads."""
    refa = np.ascontiguousarray(ref, np.uint8)
    cap = region_loci * 5 + 16
    out = np.zeros(cap, dtype=_abi.CALLED_ALLELE_DTYPE)
    full = (OrcCalled * cap)()
    total = C.c_int64(0)
    ups = np.ascontiguousarray(up_to_positions, np.int32)
    fa = (OrcCandidate * max(len(forced), 1))(*[make_candidate(p, 0, r, a) for (p, r, a) in forced])   # (the oracle derives the category)
    cap += len(forced)
    out = np.zeros(cap, dtype=_abi.CALLED_ALLELE_DTYPE)
    full = (OrcCalled * cap)()
    ivs = np.ascontiguousarray([a for a, _ in (intervals or [])], np.int32)
    ive = np.ascontiguousarray([b for _, b in (intervals or [])], np.int32)
    hc = (OrcCandidate * max(len(host_candidates), 1))(*[make_candidate(d["position"], d["category"], d["ref"], d["alt"], d["support_by_dir"], d["well_anchored_by_dir"])
                                                        for d in host_candidates])
    lib.orc_schedule_host_candidates(hc, C.c_int32(len(host_candidates)))   # (IStateManager.AddCandidates behind the reads, before the schedule)
    lib.orc_run_reads_schedule_intervals.restype = C.c_int64
    n = lib.orc_run_reads_schedule_intervals(C.byref(batch.c), refa.ctypes.data_as(C.POINTER(C.c_uint8)), C.c_int64(len(refa)), C.c_int32(region_start),
                                             C.c_int32(region_loci), C.byref(cfg), ups.ctypes.data_as(C.POINTER(C.c_int32)), C.c_int32(len(ups)),
                                             fa, C.c_int32(len(forced)), ivs.ctypes.data_as(C.POINTER(C.c_int32)), ive.ctypes.data_as(C.POINTER(C.c_int32)),
                                             C.c_int32(len(ivs)), C.c_void_p(out.ctypes.data), C.c_int64(cap), full, C.byref(total))
    lib.orc_schedule_host_candidates(None, C.c_int32(0))
    assert n >= 0, n
    return out[:n], [(full[i].ref.decode("latin-1"), full[i].alt.decode("latin-1")) for i in range(n)], total.value


def diploid_locus_process(alleles):
    """DiploidLocusProcessor.Process on [{category, genotype, gq, forced}]; returns [(genotype, gq)]."""
    n = len(alleles)
    arr = (OrcCalled * max(n, 1))()
    for i, d in enumerate(alleles):
        arr[i].category, arr[i].genotype, arr[i].genotype_qscore = d["category"], d["genotype"], d["gq"]
        arr[i].filters = (1 << _abi.FILTER_FORCED_REPORT) | (1 << _abi.FILTER_LOW_DEPTH) if d.get("forced") else 0
    lib.orc_diploid_locus_process.restype = None
    lib.orc_diploid_locus_process(arr, C.c_int32(n))
    return [(arr[i].genotype, arr[i].genotype_qscore) for i in range(n)]
