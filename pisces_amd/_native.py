"""ctypes binding of libpisceship.so — the same entry points a C# [DllImport] shim binds
(INTEGRATION.md).  There is no CPU fallback: if the HIP library is missing or fails to load,
importing this module raises."""
import ctypes as C
import os

from . import _abi

_HERE = os.path.dirname(os.path.abspath(__file__))
# PISCES_HIP_LIB points at a development build of the same library (kernel experiments); default = in-tree product
LIB_PATH = os.environ.get("PISCES_HIP_LIB") or os.path.join(_HERE, "libpisceship.so")

EXPORTS = [
    "pisces_hip_abi_version", "pisces_hip_default_config", "pisces_hip_create", "pisces_hip_destroy", "pisces_hip_trim_memory",
    "pisces_hip_last_error", "pisces_hip_set_reference", "pisces_hip_set_intervals", "pisces_hip_add_reads", "pisces_hip_stage_reads", "pisces_hip_flush_begin", "pisces_hip_flush_end",
    "pisces_hip_add_observations", "pisces_hip_flush", "pisces_hip_get_counts", "pisces_hip_add_gapped_mnv_ref",
    "pisces_hip_get_candidates", "pisces_hip_stats", "pisces_hip_call_tiles", "pisces_hip_accumulate_tiles",
    "pisces_hip_synchronize", "pisces_hip_last_kernel_ms", "pisces_hip_expand_reads", "pisces_hip_device_totals",
    "pisces_hip_set_timing", "pisces_hip_kernel_time", "pisces_hip_set_chain_timing", "pisces_hip_chain_time", "pisces_hip_flush_ex", "pisces_hip_probe_read_bandwidth",
    "pisces_hip_vcf_default_config", "pisces_hip_format_vcf", "pisces_hip_format_vcf_padded", "pisces_hip_find_candidates", "pisces_hip_call_tiles_batched",
    "pisces_hip_find_indel_candidates", "pisces_hip_compact_records", "pisces_hip_bgzf_scan", "pisces_hip_bgzf_inflate",
    "pisces_hip_balanced_tile_loci", "pisces_hip_bam_decode", "pisces_hip_bam_fetch", "pisces_hip_bam_chain_mode", "pisces_hip_add_decoded_reads", "pisces_hip_find_candidates_device", "pisces_hip_get_base_quality_sums", "pisces_hip_get_gapped_mnv_ref", "pisces_hip_comm_unique_id", "pisces_hip_comm_init", "pisces_hip_reduce_summary", "pisces_hip_comm_destroy",
    "pisces_hip_add_candidates", "pisces_hip_set_forced_alleles", "pisces_hip_host_time", "pisces_hip_set_owned_range", "pisces_hip_bam_fetch_directions", "pisces_hip_call_tiles_graph_build", "pisces_hip_call_tiles_graph_launch", "pisces_hip_mark", "pisces_hip_marked_ms",
    "pisces_hip_flush_end_ex", "pisces_hip_device_count", "pisces_hip_flush_view", "pisces_hip_flush_end_view", "pisces_hip_transfer_bytes",
    "pisces_hip_add_device_reads", "pisces_hip_get_stream", "pisces_hip_comm_library", "pisces_hip_comm_ranks", "pisces_hip_set_known_variants", "pisces_hip_set_exclude_mnvs_from_collapsing", "pisces_hip_set_exact_total_called", "pisces_hip_reallocate_failed_mnvs", "pisces_hip_set_genotypes", "pisces_hip_diploid_genotype_qscore",
]


class PiscesHipError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"libpisceship error {code}: {message}")
        self.code = code
        self.message = message


def _share_hip_runtime_with_torch():
    """PyTorch-ROCm wheels bundle their own libamdhip64.so (SONAME libamdhip64.so.7, looked up by file name
    from torch/lib).  Device pointers and streams cross between torch and this library, so both must sit on
    ONE HIP runtime: load torch's copy first, then libpisceship.so's NEEDED libamdhip64.so.7 resolves to it
    by SONAME.  Without torch (the C# host) the system ROCm runtime is used."""
    import importlib.util
    spec = importlib.util.find_spec("torch")
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        C.CDLL(cand, mode=C.RTLD_GLOBAL)


def _load():
    _share_hip_runtime_with_torch()
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: the HIP extension has not been built. Run `python -c 'import __graft_entry__ as g; "
            "g.build()'` (needs hipcc). pisces_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    i32, i64, vp = C.c_int32, C.c_int64, C.c_void_p
    P = C.POINTER
    sig = {
        "pisces_hip_abi_version": (i32, []),
        "pisces_hip_default_config": (i32, [P(_abi.PiscesHipConfig)]),
        "pisces_hip_create": (i32, [P(_abi.PiscesHipConfig), i32, P(vp)]),
        "pisces_hip_destroy": (i32, [vp]),
        "pisces_hip_trim_memory": (C.c_int64, []),
        "pisces_hip_last_error": (C.c_char_p, [vp]),
        "pisces_hip_set_reference": (i32, [vp, vp, i64]),
        "pisces_hip_set_intervals": (i32, [vp, vp, vp, i32]),
        "pisces_hip_add_reads": (i32, [vp, P(_abi.PiscesReadBatch)]),
        "pisces_hip_add_device_reads": (i32, [vp, P(_abi.PiscesReadBatch), i64, i64]),
        "pisces_hip_reallocate_failed_mnvs": (i32, [vp, i64, vp, i64, vp, i64, i32, vp, i64, P(i64), vp, i64, P(i64), vp, i64, P(i64)]),
        "pisces_hip_set_genotypes": (i32, [P(_abi.PiscesHipConfig), P(_abi.PiscesGenotypeAllele), i32, vp, i64]),
        "pisces_hip_diploid_genotype_qscore": (i32, [i32, i32, i32, i32, i32]),
        "pisces_hip_stage_reads": (i32, [vp, i32, i64, i64, i32, i32, P(_abi.PiscesReadBatch)]),
        "pisces_hip_flush_begin": (i32, [vp, i32]),
        "pisces_hip_flush_end": (i32, [vp, vp, i64, P(i64)]),
        "pisces_hip_flush_end_ex": (i32, [vp, vp, i64, P(i64), vp, vp, i64, P(i64), vp, i64, P(i64)]),
        "pisces_hip_device_count": (i32, []),
        "pisces_hip_flush_view": (i32, [vp, i32, P(vp), P(i64), P(vp), P(vp), P(i64), P(vp), P(i64)]),
        "pisces_hip_flush_end_view": (i32, [vp, P(vp), P(i64), P(vp), P(vp), P(i64), P(vp), P(i64)]),
        "pisces_hip_transfer_bytes": (i32, [vp, P(i64), i32]),
        "pisces_hip_add_observations": (i32, [vp, vp, vp, i64]),
        "pisces_hip_flush": (i32, [vp, i32, vp, i64, P(i64)]),
        "pisces_hip_flush_ex": (i32, [vp, i32, vp, i64, P(i64), vp, vp, i64, P(i64), vp, i64, P(i64)]),
        "pisces_hip_find_indel_candidates": (i64, [P(_abi.PiscesReadBatch), vp, i64, i32, vp, i64, vp, i64, P(i64)]),
        "pisces_hip_call_tiles_batched": (i32, [vp, P(_abi.PiscesTileBatch), i32, vp]),
        "pisces_hip_find_candidates": (i64, [P(_abi.PiscesReadBatch), vp, i64, i32, i32, i32, i32, i32, vp, i64, vp, i64, P(i64)]),
        "pisces_hip_find_candidates_device": (i64, [vp, P(_abi.PiscesReadBatch), i32, i32, i32, i32, vp, i64, vp, i64, P(i64)]),
        "pisces_hip_bgzf_scan": (i64, [vp, i64, P(_abi.PiscesBgzfBlock), i64, P(i64)]),
        "pisces_hip_bgzf_inflate": (i32, [vp, vp, i64, P(_abi.PiscesBgzfBlock), i64, vp, i64, i32, P(C.c_float)]),
        "pisces_hip_get_counts": (i32, [vp, i32, i32, vp]),
        "pisces_hip_balanced_tile_loci": (i32, [vp, i64]),
        "pisces_hip_bam_decode": (i32, [vp, vp, i64, P(_abi.PiscesBgzfBlock), i64, i32, i32, i32, i32, P(i64)]),
        "pisces_hip_bam_fetch": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp]),
        "pisces_hip_bam_chain_mode": (i32, [vp]),
        "pisces_hip_add_decoded_reads": (i32, [vp]),
        "pisces_hip_get_base_quality_sums": (i32, [vp, i32, i32, vp]),
        "pisces_hip_get_gapped_mnv_ref": (i32, [vp, i32, P(i32)]),
        "pisces_hip_add_gapped_mnv_ref": (i32, [vp, vp, vp, i32]),
        "pisces_hip_get_candidates": (i32, [vp, i32, vp, i64, P(i64), vp, i64, P(i64)]),
        "pisces_hip_add_candidates": (i32, [vp, vp, i64, vp, i64]),
        "pisces_hip_set_forced_alleles": (i32, [vp, vp, i64, vp, i64]),
        "pisces_hip_set_known_variants": (i32, [vp, vp, i64, vp, i64]),
        "pisces_hip_set_exclude_mnvs_from_collapsing": (i32, [vp, i32]),
        "pisces_hip_set_exact_total_called": (i32, [vp, i32]),
        "pisces_hip_stats": (i32, [vp, P(i64)]),
        "pisces_hip_host_time": (i32, [vp, P(C.c_double), i32]),
        "pisces_hip_set_owned_range": (i32, [vp, i32, i32]),
        "pisces_hip_bam_fetch_directions": (i32, [vp, vp, vp]),
        "pisces_hip_call_tiles_graph_build": (i32, [vp, P(_abi.PiscesTileBatch), i32, P(i32)]),
        "pisces_hip_call_tiles_graph_launch": (i32, [vp, i32, vp]),
        "pisces_hip_mark": (i32, [vp, i32, vp]),
        "pisces_hip_marked_ms": (i32, [vp, P(C.c_float)]),
        "pisces_hip_comm_unique_id": (i32, [vp, i32]),
        "pisces_hip_comm_init": (i32, [vp, vp, i32, i32]),
        "pisces_hip_reduce_summary": (i32, [vp, P(i64)]),
        "pisces_hip_comm_destroy": (i32, [vp]),
        "pisces_hip_call_tiles": (i32, [vp, vp, vp, i32, vp, i32, i64, vp, i32, vp, vp]),
        "pisces_hip_compact_records": (i32, [vp, vp, vp, i32, vp, vp, i32, vp, vp]),
        "pisces_hip_accumulate_tiles": (i32, [vp, vp, vp, i32, vp, vp]),
        "pisces_hip_synchronize": (i32, [vp]),
        "pisces_hip_get_stream": (i32, [vp, P(vp)]),
        "pisces_hip_comm_library": (i32, [C.c_char_p, i32]),
        "pisces_hip_comm_ranks": (i32, [vp, P(i32)]),
        "pisces_hip_last_kernel_ms": (i32, [vp, P(C.c_float)]),
        "pisces_hip_expand_reads": (i64, [P(_abi.PiscesReadBatch), i32, vp, vp, i64]),
        "pisces_hip_device_totals": (i32, [vp, P(i64), i32]),
        "pisces_hip_set_timing": (i32, [vp, i32]),
        "pisces_hip_kernel_time": (i32, [vp, P(C.c_double), P(i64)]),
        "pisces_hip_set_chain_timing": (i32, [vp, i32]),
        "pisces_hip_chain_time": (i32, [vp, P(C.c_double)]),
        "pisces_hip_probe_read_bandwidth": (i32, [vp, i64, i32, P(C.c_double)]),
        "pisces_hip_vcf_default_config": (i32, [P(_abi.PiscesVcfConfig)]),
        "pisces_hip_format_vcf": (i64, [P(_abi.PiscesVcfConfig), C.c_char_p, vp, i64, vp, vp, vp, vp, i64]),
        "pisces_hip_format_vcf_padded": (i64, [P(_abi.PiscesVcfConfig), C.c_char_p, vp, i64, vp, vp, vp, vp, i64, vp, vp, i32,
                                               P(_abi.PiscesVcfPadState), i32, vp, i64]),
    }
    for name, (res, args) in sig.items():
        f = getattr(lib, name)   # AttributeError here = a declared symbol is not exported
        f.restype = res
        f.argtypes = args
    if lib.pisces_hip_abi_version() != _abi.ABI_VERSION:
        raise ImportError("libpisceship.so ABI version does not match pisces_amd._abi")
    return lib


lib = _load()
