"""ctypes / numpy mirrors of include/pisces_hip.h (the C ABI of libpisceship.so).

Field order and widths must match the header exactly; tests/test_abi.py checks the
sizes against the compiled library.
"""
import ctypes as C

import numpy as np

ABI_VERSION = 8

# error codes
OK = 0
E_INVALID_ARG = -1
E_BUFFER_TOO_SMALL = -2
E_DEVICE = -3
E_UNMAPPED_BASE = -4
E_UNSUPPORTED = -5
E_STATE = -6

# src/lib/Pisces.Domain/Types/AlleleType.cs:3-11
ALLELE_A, ALLELE_G, ALLELE_C, ALLELE_T, ALLELE_N, ALLELE_DEL = range(6)
ALLELE_OF_BASE = {"A": ALLELE_A, "G": ALLELE_G, "C": ALLELE_C, "T": ALLELE_T, "N": ALLELE_N}
BASE_OF_ALLELE = "AGCTND"
# src/lib/Pisces.Domain/Types/DirectionType.cs:3-8
DIR_FORWARD, DIR_REVERSE, DIR_STITCHED = range(3)
# src/lib/Pisces.Domain/Types/CallType.cs:3-12
CAT_SNV, CAT_INSERTION, CAT_DELETION, CAT_MNV, CAT_REFERENCE = range(5)
# src/lib/Pisces.Domain/Types/Genotype.cs:3-18
(GT_HET_ALT1_ALT2, GT_ALT12_LIKE_NOCALL, GT_HET_ALT_REF, GT_HOM_ALT, GT_HOM_REF, GT_REF_LIKE_NOCALL,
 GT_ALT_LIKE_NOCALL, GT_REF_AND_NOCALL, GT_ALT_AND_NOCALL, GT_HEMI_REF, GT_HEMI_ALT, GT_HEMI_NOCALL, GT_OTHERS) = range(13)
GT_STRING = {GT_HET_ALT_REF: "0/1", GT_HOM_ALT: "1/1", GT_HOM_REF: "0/0", GT_REF_LIKE_NOCALL: "./.",
             GT_ALT_LIKE_NOCALL: "./.", GT_REF_AND_NOCALL: "0/.", GT_ALT_AND_NOCALL: "1/."}
# src/lib/Pisces.Domain/Types/FilterType.cs:3-19
(FILTER_STRAND_BIAS, FILTER_POOL_BIAS, FILTER_AMPLICON_BIAS, FILTER_LOW_VARIANT_QSCORE, FILTER_LOW_DEPTH,
 FILTER_LOW_VARIANT_FREQUENCY, FILTER_LOW_GENOTYPE_QUALITY, FILTER_INDEL_REPEAT_LENGTH,
 FILTER_MULTI_ALLELIC_SITE, FILTER_RMXN, FILTER_FORCED_REPORT, FILTER_OFF_TARGET, FILTER_NO_CALL) = range(13)
SB_POISSON, SB_EXTENDED, SB_DIPLOID = range(3)

ANCHOR_SIZE = 5
NUM_ANCHORS = 11
COUNTS_PER_LOCUS = 6 * 3 * NUM_ANCHORS
TUPLE_PAD = 0xFFFFFFFF


def tuple_column(locus, direction):
    """PISCES_TUPLE_COLUMN: the bank-spreading bijection of the locus-in-tile (include/pisces_hip.h)."""
    return ((((locus >> 2) & 15) | ((locus & 3) << 4)) ^ ((direction & 1) << 4))


def tuple_pack(locus, anchor, direction, allele, qual):
    """PISCES_TUPLE_PACK; works on ints, numpy arrays and torch tensors."""
    if isinstance(locus, np.ndarray):
        locus, anchor, direction, allele, qual = (np.asarray(x).astype(np.uint32) for x in (locus, anchor, direction, allele, qual))
        return ((tuple_column(locus, direction) << np.uint32(2)) | (direction << np.uint32(8)) | (allele << np.uint32(10))
                | (anchor << np.uint32(13)) | (qual << np.uint32(24)))
    v = (tuple_column(locus, direction) << 2) | (direction << 8) | (allele << 10) | (anchor << 13) | (qual << 24)
    return v & 0xFFFFFFFF if isinstance(v, int) else v


def tuple_fields(t):
    """(locus, anchor, direction, allele, qual) of packed tuples (numpy uint32 array or int)."""
    if isinstance(t, np.ndarray):
        t = t.astype(np.uint32)
    col, d = (t >> 2) & 63, (t >> 8) & 3
    c = col ^ ((d & 1) << 4)
    locus = ((c & 15) << 2) | (c >> 4)
    return locus, (t >> 13) & 15, d, (t >> 10) & 7, t >> 24


def tuple_with_locus(t, locus):
    """PISCES_TUPLE_WITH_LOCUS on numpy uint32 arrays."""
    t = np.asarray(t).astype(np.uint32)
    locus = np.asarray(locus).astype(np.uint32)
    return (t & ~np.uint32(0xFC)) | (tuple_column(locus, (t >> np.uint32(8)) & np.uint32(3)) << np.uint32(2))


class PiscesHipConfig(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32),
        ("min_base_call_quality", C.c_int32),
        ("noise_level", C.c_int32),
        ("max_variant_qscore", C.c_int32),
        ("min_variant_qscore", C.c_int32),
        ("variant_qscore_filter", C.c_int32),
        ("min_coverage", C.c_int32),
        ("low_depth_filter", C.c_int32),
        ("min_genotype_qscore", C.c_int32),
        ("max_genotype_qscore", C.c_int32),
        ("low_gq_filter", C.c_int32),
        ("strand_bias_model", C.c_int32),
        ("filter_single_strand", C.c_int32),
        ("include_reference_calls", C.c_int32),
        ("emit_zero_coverage_refs", C.c_int32),
        ("expect_stitched_reads", C.c_int32),
        ("tile_loci", C.c_int32),
        ("block_size", C.c_int32),
        ("min_frequency", C.c_float),
        ("variant_freq_filter", C.c_float),
        ("genotype_min_freq_filter", C.c_float),
        ("target_lod_frequency", C.c_float),
        ("strand_bias_threshold", C.c_float),
        ("no_call_filter_threshold", C.c_float),
        ("rmxn_max_repeat_length", C.c_int32),
        ("rmxn_min_repetitions", C.c_int32),
        ("rmxn_frequency_limit", C.c_float),
        ("collapse", C.c_int32),
        ("collapse_freq_threshold", C.c_float),
        ("collapse_freq_ratio_threshold", C.c_float),
        ("call_mnvs", C.c_int32),
        ("max_mnv_length", C.c_int32),
        ("max_gap_between_mnv", C.c_int32),
        ("noise_model", C.c_int32),
        ("ploidy", C.c_int32),
        ("diploid_snv_params", C.c_float * 3),
        ("diploid_indel_params", C.c_float * 3),
    ]


class PiscesTileBatch(C.Structure):
    _fields_ = [("d_tuples", C.c_void_p), ("d_tiles", C.c_void_p), ("n_tiles", C.c_int32), ("ref_start_position", C.c_int32),
                ("d_ref_bases", C.c_void_p), ("ref_length", C.c_int64), ("d_records", C.c_void_p), ("d_tile_results", C.c_void_p),
                ("record_capacity", C.c_int32), ("pad", C.c_int32)]


class PiscesVcfConfig(C.Structure):
    _fields_ = [("variant_quality_filter", C.c_int32), ("rmxn_max_repeat_length", C.c_int32), ("rmxn_min_repetitions", C.c_int32),
                ("noise_level", C.c_int32), ("output_strand_bias_and_noise_level", C.c_int32), ("output_no_call_fraction", C.c_int32),
                ("min_frequency_threshold", C.c_float), ("frequency_filter_threshold", C.c_float), ("crush", C.c_int32),
                ("noise_level_from_records", C.c_int32)]


class PiscesVcfPadState(C.Structure):
    _fields_ = [("last_variant_position_written", C.c_int32), ("last_padded_position", C.c_int32), ("last_cleared_interval_index", C.c_int32)]


def default_config(**overrides):
    """Reference defaults after VariantCallingParameters.Validate()
    (src/lib/Pisces.Domain/Options/VariantCallingParameters.cs:57-156)."""
    c = PiscesHipConfig()
    c.abi_version = ABI_VERSION
    c.min_base_call_quality = 20
    c.noise_level = 20
    c.max_variant_qscore = 100
    c.min_variant_qscore = 20
    c.variant_qscore_filter = 30
    c.min_coverage = 10
    c.low_depth_filter = 10
    c.min_genotype_qscore = 0
    c.max_genotype_qscore = 100
    c.low_gq_filter = -1
    c.strand_bias_model = SB_EXTENDED
    c.filter_single_strand = 0
    c.include_reference_calls = 1
    c.emit_zero_coverage_refs = 0
    c.expect_stitched_reads = 0
    c.tile_loci = 64
    c.block_size = 1000
    c.min_frequency = 0.01
    c.variant_freq_filter = 0.01
    c.genotype_min_freq_filter = 0.01
    c.target_lod_frequency = 0.01
    c.strand_bias_threshold = 0.5
    c.no_call_filter_threshold = 0.6
    c.rmxn_max_repeat_length = 5
    c.rmxn_min_repetitions = 9
    c.rmxn_frequency_limit = 0.35
    c.collapse = 1
    c.collapse_freq_threshold = 0.0
    c.collapse_freq_ratio_threshold = 0.5
    c.call_mnvs = 0
    c.max_mnv_length = 3
    c.max_gap_between_mnv = 1
    c.noise_model = 0
    c.ploidy = 0
    c.diploid_snv_params[:] = [0.20, 0.70, 0.80]
    c.diploid_indel_params[:] = [0.20, 0.70, 0.80]
    for k, v in overrides.items():
        if not hasattr(c, k):
            raise AttributeError(f"PiscesHipConfig has no field {k!r}")
        if isinstance(v, (list, tuple)):
            getattr(c, k)[:] = list(v)
        else:
            setattr(c, k, v)
    return c


CALLED_ALLELE_DTYPE = np.dtype([
    ("position", "<i4"),
    ("total_coverage", "<i4"),
    ("allele_support", "<i4"),
    ("reference_support", "<i4"),
    ("num_no_calls", "<i4"),
    ("coverage_by_dir", "<i4", (3,)),
    ("support_by_dir", "<i4", (3,)),
    ("variant_qscore", "<i4"),
    ("strand_bias_score", "<f8"),
    ("genotype_qscore", "<i2"),
    ("noise_level", "<i2"),
    ("filter_bits", "<u2"),
    ("info", "<u2"),
], align=True)
assert CALLED_ALLELE_DTYPE.itemsize == 64

TILE_DTYPE = np.dtype([("start_position", "<i4"), ("n_loci", "<i4"), ("tuple_begin", "<i8"), ("tuple_end", "<i8")],
                      align=True)
assert TILE_DTYPE.itemsize == 24
TILE_RESULT_DTYPE = np.dtype([("record_begin", "<i4"), ("n_records", "<i4"), ("n_candidate_loci", "<i4"),
                              ("n_called", "<i4"), ("valid", "<u4", (8,))], align=True)
assert TILE_RESULT_DTYPE.itemsize == 48
SLOTS_PER_TILE = 256


def records_in_order(records, tile_results):
    """Called alleles of a call_tiles launch in (position, ref, alt) order: the valid slots of every tile, ascending
    (slot of locus l, allele rank k = record_begin + 4*l + k; PiscesTileResult in include/pisces_hip.h)."""
    if len(tile_results) == 0:
        return records[:0]
    bits = np.unpackbits(tile_results["valid"].view(np.uint8).reshape(len(tile_results), 32), axis=1, bitorder="little")
    t, s = np.nonzero(bits)
    return records[tile_results["record_begin"][t].astype(np.int64) + s]


def info_genotype(info):
    return info & 0xF


def info_category(info):
    return (info >> 4) & 0x7


def info_ref(info):
    return (info >> 7) & 0x7


def info_alt(info):
    return (info >> 10) & 0x7


def info_sb_ok(info):
    return (info >> 13) & 1


def info_var_both(info):
    return (info >> 14) & 1


def info_cov_both(info):
    return (info >> 15) & 1


class PiscesReadBatch(C.Structure):
    _fields_ = [
        ("n_reads", C.c_int32),
        ("position", C.POINTER(C.c_int32)),
        ("flags", C.POINTER(C.c_uint8)),
        ("cigar_offset", C.POINTER(C.c_int32)),
        ("cigar_op", C.POINTER(C.c_uint8)),
        ("cigar_len", C.POINTER(C.c_uint32)),
        ("seq_offset", C.POINTER(C.c_int32)),
        ("bases", C.POINTER(C.c_uint8)),
        ("quals", C.POINTER(C.c_uint8)),
        ("directions", C.POINTER(C.c_uint8)),
        ("deletion_directions", C.POINTER(C.c_uint8)),
    ]


class PiscesGenotypeAllele(C.Structure):
    _fields_ = [
        ("category", C.c_int32), ("ref_len", C.c_int32), ("alt_len", C.c_int32),
        ("support", C.c_int32), ("coverage", C.c_int32), ("reference_support", C.c_int32),
        ("allele_offset", C.c_int64),
        ("genotype", C.c_int32), ("genotype_qscore", C.c_int32), ("phase_set_index", C.c_int32),
        ("multi_allelic", C.c_uint8), ("prune", C.c_uint8), ("pad", C.c_uint8 * 2),
    ]


class PiscesBgzfBlock(C.Structure):
    _fields_ = [
        ("in_offset", C.c_int64),
        ("out_offset", C.c_int64),
        ("in_length", C.c_int32),
        ("out_length", C.c_int32),
        ("crc32", C.c_uint32),
        ("reserved", C.c_int32),
    ]


class PiscesCandidate(C.Structure):
    _fields_ = [
        ("position", C.c_int32),
        ("category", C.c_int32),
        ("ref_len", C.c_int32),
        ("alt_len", C.c_int32),
        ("support_by_dir", C.c_int32 * 3),
        ("well_anchored_by_dir", C.c_int32 * 3),
        ("open_left", C.c_uint8),
        ("open_right", C.c_uint8),
        ("pad", C.c_uint8 * 2),
        ("allele_offset", C.c_int64),
    ]


DIR_UNTRACKED = 255
_READ_SPAN_OPS = "MIS=X"


def directions_from_xd(xd, cigar):
    """The two direction inputs of a stitched read from its XD tag (e.g. "3F4S3R", one direction per base of the EXPANDED CIGAR, deleted
    bases included): (dirs, del_dirs).  dirs = Read.CreateSequencedBaseDirectionMap (Read.cs:664-682): the directions of the read-span
    bases.  del_dirs = per CIGAR op, for a 'D' op the directions of its first and last deleted base (what
    GetDeletionDirectionForStitchedRead reads, CandidateVariantFinder.cs:468-487), (255, 255) for the other ops."""
    expanded, num = [], ""
    for ch in xd:
        if ch.isdigit():
            num += ch
        else:
            expanded += [{"F": DIR_FORWARD, "R": DIR_REVERSE, "S": DIR_STITCHED}[ch]] * int(num)
            num = ""
    dirs, del_dirs, e = [], [], 0
    for op, ln in cigar:
        seg = expanded[e:e + ln]
        e += ln
        if op in _READ_SPAN_OPS:
            dirs += seg
        del_dirs.append((seg[0], seg[-1]) if op == "D" and seg else (DIR_UNTRACKED, DIR_UNTRACKED))
    return dirs, del_dirs


def _ptr(arr, ctype):
    return arr.ctypes.data_as(C.POINTER(ctype))


class ReadBatch:
    """Structure-of-arrays read batch (what the C# shim pins per call). Keeps the numpy
    arrays alive next to the ctypes view."""

    def __init__(self, reads):
        """reads: iterable of dicts {pos, cigar:[(op,len)], seq:str|bytes, quals:bytes|list,
        reverse:bool, dirs:optional list, del_dirs:optional list of (first, last) per CIGAR op, or xd:"3F4S3R" for both}."""
        reads = list(reads)
        n = len(reads)
        self.position = np.array([r["pos"] for r in reads], dtype=np.int32).reshape(n)
        self.flags = np.array([1 if r.get("reverse") else 0 for r in reads], dtype=np.uint8).reshape(n)
        cig_off = [0]
        ops, lens = [], []
        seq_off = [0]
        bases, quals, dirs = [], [], []
        for r in reads:
            if r.get("xd") is not None and r.get("dirs") is None:
                r["dirs"], r["del_dirs"] = directions_from_xd(r["xd"], r["cigar"])
        any_dirs = any(r.get("dirs") is not None for r in reads)
        any_del = any(r.get("del_dirs") is not None for r in reads)
        del_dirs = []
        for r in reads:
            for op, ln in r["cigar"]:
                ops.append(ord(op))
                lens.append(ln)
            cig_off.append(len(ops))
            if any_del:
                dd = r.get("del_dirs")
                del_dirs += [tuple(x) for x in dd] if dd is not None else [(DIR_UNTRACKED, DIR_UNTRACKED)] * len(r["cigar"])
            s = r["seq"].encode() if isinstance(r["seq"], str) else bytes(r["seq"])
            q = bytes(r["quals"])
            assert len(s) == len(q), "sequence / quality length mismatch"
            bases.append(np.frombuffer(s, dtype=np.uint8))
            quals.append(np.frombuffer(q, dtype=np.uint8))
            if any_dirs:
                d = r.get("dirs")
                if d is None:
                    d = [DIR_REVERSE if r.get("reverse") else DIR_FORWARD] * len(s)
                dirs.append(np.array(d, dtype=np.uint8))
            seq_off.append(seq_off[-1] + len(s))
        self.cigar_offset = np.array(cig_off, dtype=np.int32)
        self.cigar_op = np.array(ops, dtype=np.uint8)
        self.cigar_len = np.array(lens, dtype=np.uint32)
        self.seq_offset = np.array(seq_off, dtype=np.int32)
        self.bases = np.concatenate(bases) if bases else np.zeros(0, np.uint8)
        self.quals = np.concatenate(quals) if quals else np.zeros(0, np.uint8)
        self.directions = np.concatenate(dirs) if any_dirs and dirs else None
        self.deletion_directions = np.array(del_dirs, dtype=np.uint8).reshape(-1) if any_del else None
        self._finish(n)

    @classmethod
    def from_arrays(cls, position, flags, cigar_offset, cigar_op, cigar_len, seq_offset, bases, quals,
                    directions=None, deletion_directions=None):
        self = cls.__new__(cls)
        self.position = np.ascontiguousarray(position, np.int32)
        self.flags = np.ascontiguousarray(flags, np.uint8)
        self.cigar_offset = np.ascontiguousarray(cigar_offset, np.int32)
        self.cigar_op = np.ascontiguousarray(cigar_op, np.uint8)
        self.cigar_len = np.ascontiguousarray(cigar_len, np.uint32)
        self.seq_offset = np.ascontiguousarray(seq_offset, np.int32)
        self.bases = np.ascontiguousarray(bases, np.uint8)
        self.quals = np.ascontiguousarray(quals, np.uint8)
        self.directions = None if directions is None else np.ascontiguousarray(directions, np.uint8)
        self.deletion_directions = None if deletion_directions is None else np.ascontiguousarray(deletion_directions, np.uint8).reshape(-1)
        self._finish(len(self.position))
        return self

    def _finish(self, n):
        b = PiscesReadBatch()
        b.n_reads = n
        b.position = _ptr(self.position, C.c_int32)
        b.flags = _ptr(self.flags, C.c_uint8)
        b.cigar_offset = _ptr(self.cigar_offset, C.c_int32)
        b.cigar_op = _ptr(self.cigar_op, C.c_uint8)
        b.cigar_len = _ptr(self.cigar_len, C.c_uint32)
        b.seq_offset = _ptr(self.seq_offset, C.c_int32)
        b.bases = _ptr(self.bases, C.c_uint8)
        b.quals = _ptr(self.quals, C.c_uint8)
        b.directions = _ptr(self.directions, C.c_uint8) if self.directions is not None else None
        dd = getattr(self, "deletion_directions", None)
        assert dd is None or len(dd) == 2 * len(self.cigar_op), "deletion_directions: two bytes per CIGAR op"
        b.deletion_directions = _ptr(dd, C.c_uint8) if dd is not None else None
        self.c = b
        self.n_reads = n
        self.n_bases = int(self.seq_offset[-1]) if n else 0
