// pisces_hip.hip — C ABI of libpisceship.so (include/pisces_hip.h): handle, device buffers,
// kernel launches, and the host mirror of the reference's streaming protocol
// (IStateManager.AddAlleleCounts / GetCandidatesToProcess / DoneProcessing around IAlleleCaller.Call,
// src/exe/Pisces/Logic/SmallVariantCaller.cs:79-189).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <dlfcn.h>

#include <algorithm>
#include <array>
#include <cctype>
#include <memory>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <set>
#include <string>
#include <atomic>
#include <mutex>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/pisces_hip.h"
#include "expander.h"
#include "diploid.h"
#include "finder.h"
#include "kernels.hip.h"
#include "stream_kernels.hip.h"
#include "finder_kernels.hip.h"
#include "bgzf_kernels.hip.h"
#include "bam_kernels.hip.h"

using namespace pisces;

static thread_local std::string g_create_error;

namespace {

constexpr int64_t kTimingRing = 4096;

struct BlockObs {   // the block's observations live in the device log (PiscesHip::d_log_*)
    // insertion / deletion candidates of the block (RegionState._candidateVariantsLookup), merged by
    // CandidateAllele.Equals (position, category, ref, alt) — the collapse-off rule of RegionState.AddCandidate
    std::vector<HostCandidate> cands;
    std::unordered_map<std::string, size_t> cand_index;
    int32_t max_allele_endpoint = 0;   // RegionState.MaxAlleleEndpoint
};

template <typename T>
struct DeviceBuf {
    T* p = nullptr;
    size_t cap = 0;
    DeviceBuf() = default;
    DeviceBuf(const DeviceBuf&) = delete;
    DeviceBuf& operator=(const DeviceBuf&) = delete;
    ~DeviceBuf() { release(); }   // a local buffer is freed on every early return
    hipError_t reserve(size_t n)
    {
        if (n <= cap) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        size_t want = n + n / 4 + 64;
        hipError_t e = hipMalloc((void**)&p, want * sizeof(T));
        if (e == hipSuccess) cap = want;
        return e;
    }
    // grow keeping the first n_keep elements (device-to-device copy on `stream`)
    hipError_t grow_keep(size_t n, size_t n_keep, hipStream_t stream)
    {
        if (n <= cap) return hipSuccess;
        T* old = p;
        size_t want = n + n / 2 + 1024;
        T* fresh = nullptr;
        hipError_t e = hipMalloc((void**)&fresh, want * sizeof(T));
        if (e != hipSuccess) return e;
        if (old && n_keep > 0) {
            e = hipMemcpyAsync(fresh, old, std::min(n_keep, cap) * sizeof(T), hipMemcpyDeviceToDevice, stream);
            if (e == hipSuccess) e = hipStreamSynchronize(stream);
        }
        if (old) (void)hipFree(old);
        p = fresh;
        cap = want;
        return e;
    }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

}  // namespace

struct PiscesHip {
    PiscesHipConfig cfg;
    DeviceParams P;
    int device = 0;
    hipStream_t stream = nullptr;
    // pisces_hip_call_tiles_batched: lanes that independent batches are spread over, created on first use
    static constexpr int kLanes = 3;
    hipStream_t lane[kLanes] = {nullptr, nullptr, nullptr};
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // per-launch timing window (pisces_hip_set_timing / pisces_hip_kernel_time)
    std::vector<hipEvent_t> ring;   // pairs: [2i] start, [2i+1] stop
    int32_t timing = 0;        // 0 = off (no events are recorded), n > 0 = every n-th launch is bracketed by events
    int64_t ring_used = 0;
    int64_t launches_seen = 0;
    DeviceBuf<unsigned long long> d_totals;
    DeviceBuf<double> d_qlut;
    DeviceBuf<ulonglong2> d_bq_lut;   // [256] Math.Pow(10, -1 * (int)q / 10f) in fixed point (two 38-bit halves): what a base of quality q adds to the sums
    DeviceBuf<unsigned long long> d_sumq_fix;   // the cells' fixed-point accumulators (accumulate_tiles_kernel)
    DeviceBuf<double> d_sumq;      // RegionState._sumOfAlleleBaseQualities of the tiles being called (NoiseModel.Window), made from them
    DeviceBuf<double> d_gq_tail;   // memo of the genotype-quality Poisson tail (DeviceParams::gq_tail)
    DeviceBuf<int16_t> d_vq_tab;   // memo tables of the call phase (DeviceParams::vq_tab / sb_tab / sb0_tab / gq_cap)
    DeviceBuf<double> d_sb_tab;
    DeviceBuf<double> d_sb0_tab;
    DeviceBuf<int16_t> d_gq_cap;
    DeviceBuf<DeviceParams> d_params;   // device copy of P (what the wave kernel's out-of-line cold path reads instead of a by-value copy)
    int n_cus = 256;
    void* comm = nullptr;           // ncclComm_t of the summary reduce (pisces_hip_comm_init), or nullptr
    int comm_world = 1;
    DeviceBuf<long long> d_summary;
    DeviceBuf<int32_t> d_offsets;
    DeviceBuf<PiscesCalledAllele> d_compact;
    int kernel_variant = 4;    // 4 = auto (two waves per tile while every tile of the launch is resident at once, else one),
                               // 2 = one wave per tile, 3 = two waves per tile, 0 = one 4-wave workgroup per tile
    int lds_pad = 0;           // development: extra dynamic LDS per workgroup (occupancy experiments)
    std::string err;

    DeviceBuf<uint8_t> d_ref;
    std::vector<uint8_t> h_ref;   // host copy for the indel candidate finder
    int64_t ref_len = 0;
    DeviceBuf<DevCandidate> d_cands;
    DeviceBuf<uint8_t> d_alleles;
    DeviceBuf<PiscesCalledAllele> d_cand_records;
    DeviceBuf<uint8_t> d_cand_callable;

    // ---- streaming state (RegionStateManager: _regionLookup, _lastUpToBlockKey) ----
    std::map<int32_t, BlockObs> blocks;   // key = GetBlockKey(position) (RegionStateManager.cs:385-391)
    int32_t last_block_key_cache = 0;
    BlockObs* last_block = nullptr;       // "performance improvement to remember last block" (:366)
    int32_t last_up_to_block_key = 0;
    std::unordered_map<int32_t, int32_t> gapped_mnv_ref;
    // forced genotyping alleles of this chromosome (pisces_hip_set_forced_alleles), in position order; the first n_forced_added have
    // been handed to the state as candidates (SmallVariantCaller.AddForcedAlleleAsCandidate)
    std::vector<HostCandidate> forced;
    size_t n_forced_added = 0;
    std::set<std::string> forced_keys;        // position|ref>alt (AlleleCaller.IsForcedAllele)
    std::set<int32_t> forced_positions;       // RegionState.CreateIntervalsFromAllels
    std::vector<std::pair<int32_t, int32_t>> intervals;   // sorted, disjoint [start, end]
    int64_t stats[4] = {0, 0, 0, 0};      // called, collapsed, reads, observations

    // cached result of a flush that did not fit the caller's buffer
    bool pending_valid = false;
    int32_t pending_up_to = 0;
    std::vector<PiscesCalledAllele> pending;
    std::vector<int32_t> pending_cand_index;        // per record: index into pending_cands, -1 for Reference / SNV rows
    std::vector<HostCandidate> pending_cands;       // called insertion / deletion candidates (their allele strings)
    std::vector<int32_t> pending_keys;
    int64_t pending_called = 0;
    bool pending_dropped = false;            // the flushed blocks' log entries are already gone from the other log buffer (kept entries there)
    unsigned long long pending_kept = 0;
    int64_t pending_collapsed = 0;

    // observation log on the device: (position, tuple) of every allele-count increment of the blocks not yet flushed,
    // appended by expand_reads_kernel / pisces_hip_add_observations, bucketed by tile at flush time
    DeviceBuf<int32_t> d_log_pos[2];
    DeviceBuf<uint32_t> d_log_tup[2];
    int log_cur = 0;
    DeviceBuf<unsigned long long> d_log_n;   // [0], [1]: entries kept by the last drop into log 0 / 1; [2]: observations ever made
    int64_t log_ub = 0;                      // entries of the current log, holes included (slots are reserved on the host)
    std::vector<long long> read_slots;
    std::vector<int32_t> found_slots_host;   // first candidate-record slot of every read of the batch being added
    DeviceBuf<int32_t> d_flags;              // [0] log overflow
    // staging of host input, double-buffered: a pinned host buffer, its device copy and an event that fires when the device
    // work reading them is done — add_reads returns without waiting for its own expansion kernel
    struct Stage {
        DeviceBuf<uint8_t> d;
        uint8_t* h = nullptr;
        size_t h_cap = 0;
        hipEvent_t done = nullptr;
        bool in_flight = false;
    } stage[2];
    int stage_cur = 0;
    uint8_t* h_stage = nullptr;              // = stage[stage_cur].h after stage_reserve
    DeviceBuf<uint8_t> d_stage_alias;        // unused placeholder (kept empty)
    DeviceBuf<int32_t> d_bucket;             // BucketMap tables
    std::vector<int32_t> bucket_host[4];
    int bucket_host_next = 0;
    uint64_t uploads_since_sync = 0;
    uint8_t* h_dl = nullptr;                 // pinned download buffer of flush
    // pinned arena of the small uploads of a flush (bucket tables, tile geometry, gapped-MNV counts): a copy from pageable memory makes
    // the host wait until the stream has caught up, i.e. it serialises the flush's enqueueing with the device; from pinned memory it is
    // asynchronous.  Bump-allocated, rewound when the stream is known to be idle (every flush ends with a synchronisation).
    uint8_t* h_meta = nullptr;
    size_t h_meta_cap = 0, h_meta_used = 0;
    size_t h_dl_cap = 0;
    bool drop_counter_cleared = false;   // d_log_n[log_cur ^ 1] was zeroed by the last bucket_scan_kernel and not used since
    DeviceBuf<long long> d_total;

    // candidate discovery on the device (finder_kernels.hip.h): records of the last add_reads, picked up when they are needed
    DeviceBuf<DevFound> d_found;
    DeviceBuf<uint8_t> d_found_pool;
    DeviceBuf<int32_t> d_found_slots, d_found_pool_first;
    DeviceBuf<unsigned int> d_found_misc;    // [0] pool cursor, [1] overflow flag
    DeviceBuf<long long> d_found_totals;
    struct FoundPending {
        uint8_t* h = nullptr;                // pinned: DevFound[n_slots], then the pool bytes, then misc[2]
        size_t h_cap = 0;
        hipEvent_t done = nullptr;
        bool in_flight = false;
        int64_t n_slots = 0, pool_bytes = 0;
    } found;

    // BAM decode on the device (bam_kernels.hip.h): file bytes -> inflated stream -> read batch, all handle-owned and grow-only
    struct BamState {
        DeviceBuf<uint8_t> d_file, d_stream;
        DeviceBuf<PiscesBgzfBlock> d_blocks;
        DeviceBuf<int32_t> d_status;
        DeviceBuf<uint16_t> d_exits;
        DeviceBuf<long long> d_header, d_entry;
        DeviceBuf<int32_t> d_n_reads, d_n_ops, d_n_bases, d_n_skipped, d_bstatus;
        // the read batch
        DeviceBuf<int32_t> position, cigar_offset, seq_offset;
        DeviceBuf<uint8_t> flags, cigar_op, bases, quals, op_quality, read_quality;
        DeviceBuf<uint32_t> cigar_len;
        DeviceBuf<long long> d_slots;      // log slots of the reads (pisces_hip_add_decoded_reads)
        DeviceBuf<int32_t> d_fslots;       // candidate-record slots of the reads
        int64_t n_reads = 0, n_ops = 0, n_bases = 0, n_skipped = 0;
        bool valid = false;
        int32_t min_bq = 0;
    } bam;

    // device scratch, grow-only
    DeviceBuf<uint32_t> d_tuples;
    DeviceBuf<PiscesTile> d_tiles;
    DeviceBuf<PiscesTileResult> d_tile_results;
    DeviceBuf<PiscesCalledAllele> d_records;
    DeviceBuf<int32_t> d_counts;
    DeviceBuf<uint32_t> d_gapped;
    DeviceBuf<int32_t> d_count;
};

#define PISCES_HIP_CHECK(h, expr)                                                              \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess) {                                                                \
            (h)->err = std::string(#expr) + ": " + hipGetErrorString(_e);                      \
            return PISCES_E_DEVICE;                                                            \
        }                                                                                      \
    } while (0)

static int32_t fail(PiscesHip* h, int32_t code, const std::string& msg)
{
    if (h) h->err = msg;
    else g_create_error = msg;
    return code;
}

static int32_t consume_found(PiscesHip* h);

// A flush that reported PISCES_E_BUFFER_TOO_SMALL has made its batch (collapsed candidates, MNV leftovers handed to later blocks, ...)
// and keeps it until the caller repeats the call with buffers that hold it.  Until then the state must not move: every entry that
// would change it refuses.
static int32_t refuse_while_batch_is_open(PiscesHip* h, const char* what);

static int32_t refuse_while_batch_is_open(PiscesHip* h, const char* what)
{
    if (!h->pending_valid) return PISCES_OK;
    return fail(h, PISCES_E_STATE, std::string(what) + ": a flush reported PISCES_E_BUFFER_TOO_SMALL; repeat it with larger buffers first");
}

// [256] what a base of quality q adds to the base-quality sums, RegionStateManager.cs:191: Math.Pow(10, -1 * (int)quality / 10f) (int /
// float is a float32 quotient, promoted for Pow), as the two 38-bit halves of its value cut at 2^-76 (kernels.hip.h)
static hipError_t ensure_quality_lut(PiscesHip* h)
{
    if (h->d_bq_lut.p) return hipSuccess;
    std::vector<ulonglong2> lut(256);
    for (int q = 0; q < 256; q++) {
        const double x = std::pow(10.0, (double)((float)(-1 * q) / 10.0f));   // in (0, 1]
        int e2 = 0;
        const double m = std::frexp(x, &e2);                                    // x = m * 2^e2, m in [0.5, 1)
        const unsigned long long mant = (unsigned long long)std::ldexp(m, 53);  // 53-bit integer
        const int shift = e2 - 53 + 2 * kSumqHalfBits;                          // x * 2^76 = mant * 2^shift
        const unsigned __int128 v = shift >= 0 ? (unsigned __int128)mant << shift : (shift > -64 ? (unsigned __int128)mant >> (-shift) : (unsigned __int128)0);
        lut[(size_t)q].x = (unsigned long long)(v >> kSumqHalfBits);
        lut[(size_t)q].y = (unsigned long long)(v & (((unsigned __int128)1 << kSumqHalfBits) - 1));
    }
    hipError_t e = h->d_bq_lut.reserve(256);
    if (e == hipSuccess) e = hipMemcpy(h->d_bq_lut.p, lut.data(), 256 * sizeof(ulonglong2), hipMemcpyHostToDevice);
    return e;
}

// tuples of `n_tiles` tiles -> anchor-resolved counts in d_counts (and, with_sums, the base-quality sums in d_sumq), on stream s
static hipError_t accumulate_tiles(PiscesHip* h, hipStream_t s, const uint32_t* d_tuples, const PiscesTile* d_tiles, int32_t n_tiles, bool with_sums)
{
    const size_t nc = (size_t)n_tiles * kTile * PISCES_COUNTS_PER_LOCUS;
    hipError_t e = h->d_counts.reserve(nc);
    if (e == hipSuccess && with_sums) e = h->d_sumq.reserve(nc);
    if (e == hipSuccess && with_sums) e = h->d_sumq_fix.reserve(2 * nc);
    if (e == hipSuccess && with_sums) e = ensure_quality_lut(h);
    if (e != hipSuccess) return e;
    (void)hipMemsetAsync(h->d_counts.p, 0, nc * sizeof(int32_t), s);
    if (with_sums) (void)hipMemsetAsync(h->d_sumq_fix.p, 0, 2 * nc * sizeof(unsigned long long), s);
    hipLaunchKernelGGL(accumulate_tiles_kernel, dim3((unsigned)n_tiles), dim3(kBlock), 0, s, d_tuples, d_tiles, n_tiles, h->d_counts.p,
                       h->cfg.min_base_call_quality, with_sums ? h->d_sumq_fix.p : (unsigned long long*)nullptr,
                       with_sums ? (const ulonglong2*)h->d_bq_lut.p : (const ulonglong2*)nullptr);
    if (with_sums)
        hipLaunchKernelGGL(finish_quality_sums_kernel, dim3((unsigned)((nc + 255) / 256)), dim3(256), 0, s, h->d_sumq_fix.p, h->d_sumq.p, (int64_t)nc);
    return hipGetLastError();
}

static DeviceParams make_params(const PiscesHipConfig& c)
{
    DeviceParams P;
    P.min_bq = c.min_base_call_quality;
    P.noise_level = c.noise_level;
    P.max_vq = c.max_variant_qscore;
    P.min_vq = c.min_variant_qscore;
    P.vq_filter = c.variant_qscore_filter;
    P.min_cov = c.min_coverage;
    P.low_depth_filter = c.low_depth_filter;
    P.min_gq = c.min_genotype_qscore;
    P.max_gq = c.max_genotype_qscore;
    P.low_gq_filter = c.low_gq_filter;
    P.sb_model = c.strand_bias_model;
    P.filter_single_strand = c.filter_single_strand;
    P.include_ref = c.include_reference_calls;
    P.emit_zero_cov = c.emit_zero_coverage_refs;
    P.rmxn_max_len = c.rmxn_max_repeat_length;
    P.rmxn_min_rep = c.rmxn_min_repetitions;
    P.min_freq = c.min_frequency;
    P.vf_filter = c.variant_freq_filter;
    P.gt_min_freq = c.genotype_min_freq_filter;
    P.target_lod = c.target_lod_frequency;
    P.nocall_thr = c.no_call_filter_threshold;
    P.rmxn_freq_limit = c.rmxn_frequency_limit;
    P.sb_threshold = (double)c.strand_bias_threshold;
    // MathOperations.QtoP: Math.Pow(10, -1 * q / 10f), q double  (stats/MathOperations.cs:7-10)
    P.err_q = std::pow(10.0, -1 * (double)c.noise_level / 10.0);
    // StrandBiasCalculator.cs:32: Math.Pow(10, -1*qNoise/10f), int / float -> float exponent
    P.err_sb = std::pow(10.0, (double)((float)(-1 * c.noise_level) / 10.0f));
    P.ln10 = std::log(10.0);
    P.totals = nullptr;
    P.q_to_p_lut = nullptr;
    P.q_to_p_n = 0;
    P.gq_tail = nullptr;
    P.gq_tail_a = 0;
    P.gq_tail_cov = 0;
    P.refs_only = c.call_mnvs ? 1 : 0;
    P.vq_tab = nullptr;
    P.sb_tab = nullptr;
    P.sb0_tab = nullptr;
    P.gq_cap = nullptr;
    P.vq_tab_k = P.sb_tab_k = P.tab_cov = 0;
    return P;
}

// Nothing crosses the C ABI as an exception: every exported entry runs inside this guard (std::bad_alloc from a host vector,
// std::length_error from a string built on caller data, ...) and reports PISCES_E_INTERNAL with the message in last_error.
template <typename R, typename F>
static R abi_guard(PiscesHip* h, F&& body)
{
    try {
        return body();
    } catch (const std::exception& e) {
        return (R)fail(h, PISCES_E_INTERNAL, std::string("exception inside the library: ") + e.what());
    } catch (...) {
        return (R)fail(h, PISCES_E_INTERNAL, "unknown exception inside the library");
    }
}

extern "C" {

int32_t pisces_hip_abi_version(void) { return PISCES_HIP_ABI_VERSION; }

int32_t pisces_hip_default_config(PiscesHipConfig* c)
{
    return abi_guard<int32_t>((PiscesHip*)nullptr, [&]() -> int32_t {
    if (!c) return PISCES_E_INVALID_ARG;
    std::memset(c, 0, sizeof(*c));
    c->abi_version = PISCES_HIP_ABI_VERSION;
    c->min_base_call_quality = 20;
    c->noise_level = 20;
    c->max_variant_qscore = 100;
    c->min_variant_qscore = 20;
    c->variant_qscore_filter = 30;
    c->min_coverage = 10;
    c->low_depth_filter = 10;
    c->min_genotype_qscore = 0;
    c->max_genotype_qscore = 100;
    c->low_gq_filter = -1;
    c->strand_bias_model = PISCES_SB_EXTENDED;
    c->filter_single_strand = 0;
    c->include_reference_calls = 1;
    c->emit_zero_coverage_refs = 0;
    c->expect_stitched_reads = 0;
    c->tile_loci = kTile;
    c->block_size = 1000;
    c->min_frequency = 0.01f;
    c->variant_freq_filter = 0.01f;
    c->genotype_min_freq_filter = 0.01f;
    c->target_lod_frequency = 0.01f;
    c->strand_bias_threshold = 0.5f;
    c->no_call_filter_threshold = 0.6f;
    c->rmxn_max_repeat_length = 5;
    c->rmxn_min_repetitions = 9;
    c->rmxn_frequency_limit = 0.35f;
    c->collapse = 1;
    c->collapse_freq_threshold = 0.0f;
    c->collapse_freq_ratio_threshold = 0.5f;
    c->call_mnvs = 0;
    c->max_mnv_length = 3;
    c->max_gap_between_mnv = 1;
    c->noise_model = PISCES_NOISE_FLAT;
    c->ploidy = PISCES_PLOIDY_SOMATIC;
    c->diploid_snv_params[0] = c->diploid_indel_params[0] = 0.20f;
    c->diploid_snv_params[1] = c->diploid_indel_params[1] = 0.70f;
    c->diploid_snv_params[2] = c->diploid_indel_params[2] = 0.80f;
    return PISCES_OK;
    });
}

int32_t pisces_hip_destroy(PiscesHip* h);

int32_t pisces_hip_create(const PiscesHipConfig* cfg, int32_t device, PiscesHip** out)
{
    return abi_guard<int32_t>((PiscesHip*)nullptr, [&]() -> int32_t {
    if (!cfg || !out) return fail(nullptr, PISCES_E_INVALID_ARG, "pisces_hip_create: null argument");
    *out = nullptr;
    if (cfg->abi_version != PISCES_HIP_ABI_VERSION)
        return fail(nullptr, PISCES_E_INVALID_ARG, "pisces_hip_create: config abi_version mismatch");
    if (cfg->tile_loci != 0 && cfg->tile_loci != kTile)
        return fail(nullptr, PISCES_E_UNSUPPORTED, "pisces_hip_create: tile_loci must be 64 in this build");
    if (cfg->strand_bias_model < PISCES_SB_POISSON || cfg->strand_bias_model > PISCES_SB_DIPLOID || cfg->ploidy < PISCES_PLOIDY_SOMATIC ||
        cfg->ploidy > PISCES_PLOIDY_HAPLOID || cfg->noise_model < PISCES_NOISE_FLAT || cfg->noise_model > PISCES_NOISE_WINDOW)
        return fail(nullptr, PISCES_E_INVALID_ARG, "pisces_hip_create: strand_bias_model / ploidy / noise_model out of range");
    if (cfg->block_size <= 0 || cfg->min_base_call_quality < 0 || cfg->min_base_call_quality > 254)
        return fail(nullptr, PISCES_E_INVALID_ARG, "pisces_hip_create: block_size / min_base_call_quality out of range");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(nullptr, PISCES_E_DEVICE, std::string("pisces_hip_create: no HIP device: ") + hipGetErrorString(e));
    if (device < 0 || device >= ndev) return fail(nullptr, PISCES_E_INVALID_ARG, "pisces_hip_create: device index out of range");
    PiscesHip* h = new PiscesHip();
    h->cfg = *cfg;
    h->cfg.tile_loci = kTile;
    h->P = make_params(h->cfg);
    h->device = device;
    if ((e = hipSetDevice(device)) != hipSuccess || (e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking)) != hipSuccess ||
        (e = hipEventCreate(&h->ev0)) != hipSuccess || (e = hipEventCreate(&h->ev1)) != hipSuccess) {
        g_create_error = std::string("pisces_hip_create: ") + hipGetErrorString(e);
        delete h;
        return PISCES_E_DEVICE;
    }
    if ((e = h->d_totals.reserve((size_t)kTotalShards * kTotalStride)) != hipSuccess ||
        (e = hipMemsetAsync(h->d_totals.p, 0, (size_t)kTotalShards * kTotalStride * sizeof(unsigned long long), h->stream)) != hipSuccess ||
        (e = hipStreamSynchronize(h->stream)) != hipSuccess) {
        g_create_error = std::string("pisces_hip_create: ") + hipGetErrorString(e);
        pisces_hip_destroy(h);
        return PISCES_E_DEVICE;
    }
    h->P.totals = h->d_totals.p;
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) h->n_cus = prop.multiProcessorCount;
        const char* kv = getenv("PISCES_HIP_KERNEL");   // development switch between the two forms of the hot kernel
        if (kv && std::string(kv) == "wave") h->kernel_variant = 2;
        if (kv && std::string(kv) == "wave2") h->kernel_variant = 3;
        if (kv && std::string(kv) == "auto") h->kernel_variant = 4;
        if (kv && std::string(kv) == "block") h->kernel_variant = 0;
        if (const char* lp = getenv("PISCES_HIP_LDS_PAD")) h->lds_pad = atoi(lp);
    }
    {
        // MathOperations.QtoP(q) = Math.Pow(10, -1 * q / 10f) for every integer q-score the caller can produce
        const int n = std::min(std::max(h->cfg.max_variant_qscore, 0), 4095) + 1;
        std::vector<double> lut((size_t)n);
        for (int q = 0; q < n; q++) lut[(size_t)q] = std::pow(10.0, -1 * (double)q / 10.0);
        if ((e = h->d_qlut.reserve((size_t)n)) != hipSuccess ||
            (e = hipMemcpy(h->d_qlut.p, lut.data(), (size_t)n * sizeof(double), hipMemcpyHostToDevice)) != hipSuccess) {
            g_create_error = std::string("pisces_hip_create: ") + hipGetErrorString(e);
            pisces_hip_destroy(h);
            return PISCES_E_DEVICE;
        }
        h->P.q_to_p_lut = h->d_qlut.p;
        h->P.q_to_p_n = n;
    }
    if (h->cfg.noise_model == PISCES_NOISE_WINDOW) {
        if ((e = ensure_quality_lut(h)) != hipSuccess) {
            g_create_error = std::string("pisces_hip_create: ") + hipGetErrorString(e);
            pisces_hip_destroy(h);
            return PISCES_E_DEVICE;
        }
    }
    if ((e = h->d_log_n.reserve(4)) != hipSuccess || (e = h->d_flags.reserve(4)) != hipSuccess ||
        (e = hipMemset(h->d_log_n.p, 0, 4 * sizeof(unsigned long long))) != hipSuccess ||
        (e = hipMemset(h->d_flags.p, 0, 4 * sizeof(int32_t))) != hipSuccess) {
        g_create_error = std::string("pisces_hip_create: ") + hipGetErrorString(e);
        pisces_hip_destroy(h);
        return PISCES_E_DEVICE;
    }
    if (!getenv("PISCES_HIP_NO_GQ_TABLE")) {
        // genotype-quality tail memo, evaluated on the device by the function it stands in for
        const int32_t n_a = 32, n_cov = 8192;
        if ((e = h->d_gq_tail.reserve((size_t)n_a * n_cov)) != hipSuccess) {
            g_create_error = std::string("pisces_hip_create: ") + hipGetErrorString(e);
            pisces_hip_destroy(h);
            return PISCES_E_DEVICE;
        }
        hipLaunchKernelGGL(build_gq_tail_kernel, dim3((unsigned)((n_a * n_cov + 255) / 256)), dim3(256), 0, h->stream, h->d_gq_tail.p,
                           n_a, n_cov, h->P.target_lod);
        if ((e = hipGetLastError()) != hipSuccess || (e = hipStreamSynchronize(h->stream)) != hipSuccess) {
            g_create_error = std::string("pisces_hip_create: ") + hipGetErrorString(e);
            pisces_hip_destroy(h);
            return PISCES_E_DEVICE;
        }
        h->P.gq_tail = h->d_gq_tail.p;
        h->P.gq_tail_a = n_a;
        h->P.gq_tail_cov = n_cov;
    }
    if (!getenv("PISCES_HIP_NO_CALL_TABLES") && h->cfg.noise_model == PISCES_NOISE_FLAT && h->cfg.strand_bias_model != PISCES_SB_DIPLOID &&
        h->cfg.max_variant_qscore <= 32767 && h->cfg.max_genotype_qscore <= 32767 && h->cfg.min_genotype_qscore >= -32768) {
        // Memo tables of the streaming-rate kernel's call phase, filled by the device with the functions they stand in for
        // (bit-identical by construction): variant q-score and strand-bias tail by (support, coverage), the support-0 power by
        // coverage, the capped genotype q-score by (non-allele observations, coverage).  ~22 MB and ~1 ms per handle.
        const int32_t n_k = 256, n_cov = 8192;
        if ((e = h->d_vq_tab.reserve((size_t)n_k * n_cov)) != hipSuccess || (e = h->d_sb_tab.reserve((size_t)n_k * n_cov)) != hipSuccess ||
            (e = h->d_sb0_tab.reserve((size_t)n_cov)) != hipSuccess ||
            (h->P.gq_tail && (e = h->d_gq_cap.reserve((size_t)h->P.gq_tail_a * h->P.gq_tail_cov)) != hipSuccess)) {
            g_create_error = std::string("pisces_hip_create: ") + hipGetErrorString(e);
            pisces_hip_destroy(h);
            return PISCES_E_DEVICE;
        }
        const unsigned nb = (unsigned)(((size_t)n_k * n_cov + 255) / 256);
        hipLaunchKernelGGL(build_vq_tab_kernel, dim3(nb), dim3(256), 0, h->stream, h->d_vq_tab.p, n_k, n_cov, h->P);
        hipLaunchKernelGGL(build_sb_tab_kernel, dim3(nb), dim3(256), 0, h->stream, h->d_sb_tab.p, h->d_sb0_tab.p, n_k, n_cov, h->P);
        if (h->P.gq_tail)
            hipLaunchKernelGGL(build_gq_cap_kernel, dim3((unsigned)((h->P.gq_tail_a * h->P.gq_tail_cov + 255) / 256)), dim3(256), 0, h->stream,
                               h->d_gq_cap.p, h->P.gq_tail, h->P.gq_tail_a, h->P.gq_tail_cov, h->P);
        if ((e = hipGetLastError()) != hipSuccess || (e = hipStreamSynchronize(h->stream)) != hipSuccess) {
            g_create_error = std::string("pisces_hip_create: ") + hipGetErrorString(e);
            pisces_hip_destroy(h);
            return PISCES_E_DEVICE;
        }
        h->P.vq_tab = h->d_vq_tab.p;
        h->P.sb_tab = h->d_sb_tab.p;
        h->P.sb0_tab = h->d_sb0_tab.p;
        h->P.gq_cap = h->P.gq_tail ? h->d_gq_cap.p : nullptr;
        h->P.vq_tab_k = h->P.sb_tab_k = n_k;
        h->P.tab_cov = n_cov;
    }
    if ((e = h->d_params.reserve(1)) != hipSuccess || (e = hipMemcpy(h->d_params.p, &h->P, sizeof(DeviceParams), hipMemcpyHostToDevice)) != hipSuccess) {
        g_create_error = std::string("pisces_hip_create: ") + hipGetErrorString(e);
        pisces_hip_destroy(h);
        return PISCES_E_DEVICE;
    }
    *out = h;
    return PISCES_OK;
    });
}

int32_t pisces_hip_destroy(PiscesHip* h)
{
    return abi_guard<int32_t>(h, [&]() -> int32_t {
    if (!h) return PISCES_OK;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    (void)pisces_hip_comm_destroy(h);
    h->d_summary.release();
    h->d_ref.release(); h->d_tuples.release(); h->d_tiles.release(); h->d_tile_results.release();
    h->d_records.release(); h->d_counts.release(); h->d_gapped.release(); h->d_count.release(); h->d_totals.release(); h->d_qlut.release(); h->d_bq_lut.release(); h->d_sumq_fix.release(); h->d_sumq.release(); h->d_gq_tail.release(); h->d_vq_tab.release(); h->d_sb_tab.release(); h->d_sb0_tab.release(); h->d_gq_cap.release(); h->d_params.release(); h->d_offsets.release(); h->d_compact.release();
    for (int i = 0; i < 2; i++) { h->d_log_pos[i].release(); h->d_log_tup[i].release(); }
    h->d_log_n.release(); h->d_flags.release(); h->d_bucket.release(); h->d_total.release();
    for (auto& st : h->stage) {
        st.d.release();
        if (st.h) (void)hipHostFree(st.h);
        st.h = nullptr;
        if (st.done) (void)hipEventDestroy(st.done);
        st.done = nullptr;
    }
    h->h_stage = nullptr;
    if (h->h_dl) (void)hipHostFree(h->h_dl);
    h->h_dl = nullptr;
    if (h->h_meta) (void)hipHostFree(h->h_meta);
    h->h_meta = nullptr;
    if (h->found.h) (void)hipHostFree(h->found.h);
    h->found.h = nullptr;
    if (h->found.done) (void)hipEventDestroy(h->found.done);
    h->found.done = nullptr;
    h->d_found.release(); h->d_found_pool.release(); h->d_found_slots.release(); h->d_found_pool_first.release();
    h->d_found_misc.release(); h->d_found_totals.release();
    h->d_cands.release(); h->d_alleles.release(); h->d_cand_records.release(); h->d_cand_callable.release();
    for (hipEvent_t ev : h->ring) (void)hipEventDestroy(ev);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    for (int k = 0; k < PiscesHip::kLanes; k++)
        if (h->lane[k]) (void)hipStreamDestroy(h->lane[k]);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return PISCES_OK;
    });
}

const char* pisces_hip_last_error(const PiscesHip* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int32_t pisces_hip_set_reference(PiscesHip* h, const uint8_t* bases, int64_t length)
{
    return abi_guard<int32_t>(h, [&]() -> int32_t {
    if (!h) return PISCES_E_INVALID_ARG;
    if (!bases || length <= 0) return fail(h, PISCES_E_INVALID_ARG, "set_reference: empty reference");
    PISCES_HIP_CHECK(h, hipSetDevice(h->device));
    { int32_t rcf = consume_found(h); if (rcf) return rcf; }   // candidates found against the previous reference take their strings from it
    PISCES_HIP_CHECK(h, hipStreamSynchronize(h->stream));
    PISCES_HIP_CHECK(h, h->d_ref.reserve((size_t)length));
    PISCES_HIP_CHECK(h, hipMemcpyAsync(h->d_ref.p, bases, (size_t)length, hipMemcpyHostToDevice, h->stream));
    PISCES_HIP_CHECK(h, hipStreamSynchronize(h->stream));
    h->h_ref.assign(bases, bases + length);
    h->ref_len = length;
    return PISCES_OK;
    });
}

int32_t pisces_hip_set_intervals(PiscesHip* h, const int32_t* starts, const int32_t* ends, int32_t n)
{
    return abi_guard<int32_t>(h, [&]() -> int32_t {
    if (!h || n < 0 || (n > 0 && (!starts || !ends))) return fail(h, PISCES_E_INVALID_ARG, "set_intervals: bad arguments");
    h->intervals.clear();
    for (int i = 0; i < n; i++) {
        if (starts[i] <= 0 || ends[i] < starts[i] || (i > 0 && starts[i] <= ends[i - 1]))
            return fail(h, PISCES_E_INVALID_ARG, "set_intervals: intervals must be positive, sorted and disjoint");
        h->intervals.emplace_back(starts[i], ends[i]);
    }
    return PISCES_OK;
    });
}

// ------------------------------------------------------------------------------------------------
// streaming surface
// ------------------------------------------------------------------------------------------------
static inline int32_t block_key(const PiscesHip* h, int32_t position)
{
    // GetBlockKey: (int)Math.Ceiling((double)position / _regionSize)
    return (position + h->cfg.block_size - 1) / h->cfg.block_size;
}

static inline BlockObs* get_block(PiscesHip* h, int32_t position)
{
    int32_t key = block_key(h, position);
    if (h->last_block && h->last_block_key_cache == key) return h->last_block;
    BlockObs* b = &h->blocks[key];
    h->last_block = b;
    h->last_block_key_cache = key;
    return b;
}

// room for `extra` more log entries (the log keeps its content when it grows)
static int32_t log_reserve(PiscesHip* h, int64_t extra)
{
    const size_t need = (size_t)(h->log_ub + extra);
    const int c = h->log_cur;
    PISCES_HIP_CHECK(h, h->d_log_pos[c].grow_keep(need, (size_t)h->log_ub, h->stream));
    PISCES_HIP_CHECK(h, h->d_log_tup[c].grow_keep(need, (size_t)h->log_ub, h->stream));
    return PISCES_OK;
}

// enqueues dst[0, bytes) = src[0, bytes) (device <- host) on h->stream through the pinned arena
static int32_t meta_upload(PiscesHip* h, void* dst, const void* src, size_t bytes)
{
    if (bytes == 0) return PISCES_OK;
    const size_t need = (bytes + 63) & ~(size_t)63;
    if (h->h_meta_used + need > h->h_meta_cap) {
        // copies out of the arena may be in flight: drain, rewind, and grow if this one upload is larger than the arena (rare)
        PISCES_HIP_CHECK(h, hipStreamSynchronize(h->stream));
        h->h_meta_used = 0;
        if (need > h->h_meta_cap) {
            const size_t want = std::max<size_t>(need * 2, (size_t)1 << 20);
            if (h->h_meta) (void)hipHostFree(h->h_meta);
            h->h_meta = nullptr;
            h->h_meta_cap = 0;
            PISCES_HIP_CHECK(h, hipHostMalloc((void**)&h->h_meta, want, hipHostMallocDefault));
            h->h_meta_cap = want;
        }
    }
    uint8_t* at = h->h_meta + h->h_meta_used;
    h->h_meta_used += need;
    std::memcpy(at, src, bytes);
    PISCES_HIP_CHECK(h, hipMemcpyAsync(dst, at, bytes, hipMemcpyHostToDevice, h->stream));
    return PISCES_OK;
}

// next staging pair with room for `bytes`; waits only for the work that used THIS pair two calls ago
static int32_t stage_reserve(PiscesHip* h, size_t bytes)
{
    h->stage_cur ^= 1;
    PiscesHip::Stage& st = h->stage[h->stage_cur];
    if (!st.done) PISCES_HIP_CHECK(h, hipEventCreateWithFlags(&st.done, hipEventDisableTiming));
    if (st.in_flight) {
        PISCES_HIP_CHECK(h, hipEventSynchronize(st.done));
        st.in_flight = false;
    }
    if (bytes > st.h_cap) {
        if (st.h) (void)hipHostFree(st.h);
        st.h = nullptr;
        st.h_cap = 0;
        const size_t want = bytes + bytes / 2 + 4096;
        PISCES_HIP_CHECK(h, hipHostMalloc((void**)&st.h, want, hipHostMallocDefault));
        st.h_cap = want;
    }
    PISCES_HIP_CHECK(h, st.d.reserve(bytes));
    h->h_stage = st.h;
    return PISCES_OK;
}
// call after the last device operation that reads the current staging pair has been enqueued
static int32_t stage_release(PiscesHip* h)
{
    PiscesHip::Stage& st = h->stage[h->stage_cur];
    PISCES_HIP_CHECK(h, hipEventRecord(st.done, h->stream));
    st.in_flight = true;
    return PISCES_OK;
}
#define D_STAGE(h) ((h)->stage[(h)->stage_cur].d.p)

namespace pisces {
// host-expanded observations: copied behind the current end of the log (its size is host-known: slots are reserved on the host)
__global__ __launch_bounds__(256) void log_append_kernel(const int32_t* __restrict__ src_pos, const uint32_t* __restrict__ src_tup, int64_t n,
                                                         int32_t* __restrict__ log_pos, uint32_t* __restrict__ log_tup, long long base,
                                                         unsigned long long* __restrict__ appended)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        log_pos[base + i] = src_pos[i];
        log_tup[base + i] = src_tup[i] & ~0xFCu;   // the column is set from the position when the log is bucketed
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(appended, (unsigned long long)n);
}
}  // namespace pisces

int32_t pisces_hip_add_observations(PiscesHip* h, const int32_t* positions, const uint32_t* tuples, int64_t n)
{
    return abi_guard<int32_t>(h, [&]() -> int32_t {
    if (!h) return PISCES_E_INVALID_ARG;
    if (n < 0 || (n > 0 && (!positions || !tuples))) return fail(h, PISCES_E_INVALID_ARG, "add_observations: null buffer");
    for (int64_t i = 0; i < n; i++)
        if (positions[i] <= 0) return fail(h, PISCES_E_INVALID_ARG, "Position must be greater than 0.");  // RegionStateManager.cs:363-364
    { int32_t rcp = refuse_while_batch_is_open(h, "add_observations"); if (rcp) return rcp; }
    if (n == 0) return PISCES_OK;
    PISCES_HIP_CHECK(h, hipSetDevice(h->device));
    for (int64_t i = 0; i < n; i++) (void)get_block(h, positions[i]);
    int32_t rc = log_reserve(h, n);
    if (rc) return rc;
    const size_t bytes = (size_t)n * 8;
    rc = stage_reserve(h, bytes);
    if (rc) return rc;
    std::memcpy(h->h_stage, positions, (size_t)n * 4);
    std::memcpy(h->h_stage + (size_t)n * 4, tuples, (size_t)n * 4);
    PISCES_HIP_CHECK(h, hipMemcpyAsync(D_STAGE(h), h->h_stage, bytes, hipMemcpyHostToDevice, h->stream));
    const int c = h->log_cur;
    hipLaunchKernelGGL(log_append_kernel, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 4096)), dim3(256), 0, h->stream,
                       (const int32_t*)D_STAGE(h), (const uint32_t*)(D_STAGE(h) + (size_t)n * 4), n, h->d_log_pos[c].p, h->d_log_tup[c].p,
                       (long long)h->log_ub, h->d_log_n.p + 2);
    PISCES_HIP_CHECK(h, hipGetLastError());
    { int32_t rcs = stage_release(h); if (rcs) return rcs; }
    h->log_ub += n;
    return PISCES_OK;
    });
}

namespace {
struct ArraySink : ObservationSink {
    int32_t* positions;
    uint32_t* tuples;
    int64_t capacity, n = 0;
    void emit(int32_t position, uint32_t tuple) override
    {
        if (n < capacity) { positions[n] = position; tuples[n] = tuple; }
        n++;
    }
};
}  // namespace

static int32_t validate_batch(const PiscesReadBatch* b)
{
    if (!b || b->n_reads < 0) return PISCES_E_INVALID_ARG;
    if (b->n_reads == 0) return PISCES_OK;
    if (!b->position || !b->flags || !b->cigar_offset || !b->cigar_op || !b->cigar_len || !b->seq_offset || !b->bases || !b->quals)
        return PISCES_E_INVALID_ARG;
    // BAM stores an operation length in 28 bits; anything larger would overflow the int arithmetic of the read walks
    const int64_t n_ops = (int64_t)b->cigar_offset[b->n_reads] - (int64_t)b->cigar_offset[0];
    if (n_ops < 0) return PISCES_E_INVALID_ARG;
    for (int64_t c = b->cigar_offset[0]; c < (int64_t)b->cigar_offset[b->n_reads]; c++)
        if (b->cigar_len[c] > 0x0FFFFFFFu) return PISCES_E_INVALID_ARG;
    return PISCES_OK;
}

// IStateManager.AddCandidates -> RegionState.AddCandidate (RegionState.cs:94-174): merge by CandidateAllele.Equals, and with the
// collapser on (trackOpenEnded) keep open-ended candidates apart (:114-137); UpdateMaxPosition (:205-223)
static void add_candidate(PiscesHip* h, const HostCandidate& cnd)
{
    BlockObs* b = get_block(h, cnd.position);
    std::string key = std::to_string(cnd.position) + "|" + std::to_string(cnd.category) + "|" + cnd.ref + ">" + cnd.alt;
    if (h->cfg.collapse) key += cnd.open_left ? (cnd.open_right ? "|LR" : "|L") : (cnd.open_right ? "|R" : "|");
    auto it = b->cand_index.find(key);
    if (it == b->cand_index.end()) {
        b->cand_index.emplace(std::move(key), b->cands.size());
        b->cands.push_back(cnd);
    } else {
        HostCandidate& e = b->cands[it->second];
        for (int d = 0; d < 3; d++) {
            e.support_by_dir[d] += cnd.support_by_dir[d];
            e.well_anchored_by_dir[d] += cnd.well_anchored_by_dir[d];
        }
    }
    int32_t other_end = 0;
    if (cnd.category == PISCES_CAT_DELETION) other_end = cnd.position + (int32_t)cnd.ref.size();
    else if (cnd.category == PISCES_CAT_INSERTION) other_end = cnd.position + 1;
    else if (cnd.category == PISCES_CAT_MNV) other_end = cnd.position + (int32_t)cnd.ref.size() - 1;
    if (other_end > b->max_allele_endpoint) b->max_allele_endpoint = other_end;
}

static std::string forced_key(int32_t position, const std::string& ref, const std::string& alt)
{
    return std::to_string(position) + "|" + ref + ">" + alt;
}
static bool is_forced_allele(const PiscesHip* h, const HostCandidate& c)   // AlleleCaller.IsForcedAllele (AlleleCaller.cs:179-184)
{
    return !h->forced_keys.empty() && h->forced_keys.count(forced_key(c.position, c.ref, c.alt)) != 0;
}

static int32_t host_candidates_of(PiscesHip* h, const PiscesCandidate* cands, int64_t n, const uint8_t* alleles, int64_t allele_bytes,
                                  std::vector<HostCandidate>& out, const char* what)
{
    if (n < 0 || (n > 0 && (!cands || !alleles))) return fail(h, PISCES_E_INVALID_ARG, std::string(what) + ": null input");
    for (int64_t i = 0; i < n; i++) {
        const PiscesCandidate& c = cands[i];
        if (c.position <= 0 || c.ref_len <= 0 || c.alt_len <= 0 || c.allele_offset < 0 || c.allele_offset + c.ref_len + c.alt_len > allele_bytes ||
            c.category < PISCES_CAT_SNV || c.category > PISCES_CAT_MNV)
            return fail(h, PISCES_E_INVALID_ARG, std::string(what) + ": bad candidate");
        HostCandidate hc;
        hc.position = c.position;
        hc.category = c.category;
        hc.ref.assign((const char*)alleles + c.allele_offset, (size_t)c.ref_len);
        hc.alt.assign((const char*)alleles + c.allele_offset + c.ref_len, (size_t)c.alt_len);
        for (int d = 0; d < 3; d++) { hc.support_by_dir[d] = c.support_by_dir[d]; hc.well_anchored_by_dir[d] = c.well_anchored_by_dir[d]; }
        hc.open_left = c.open_left != 0;
        hc.open_right = c.open_right != 0;
        out.push_back(std::move(hc));
    }
    return PISCES_OK;
}

// IStateManager.AddCandidates (IStateManager.cs; RegionStateManager.cs:83-116) for candidates the caller brings itself
int32_t pisces_hip_add_candidates(PiscesHip* h, const PiscesCandidate* cands, int64_t n, const uint8_t* alleles, int64_t allele_bytes)
{
    return abi_guard<int32_t>(h, [&]() -> int32_t {
    if (!h) return PISCES_E_INVALID_ARG;
    { int32_t rcp = refuse_while_batch_is_open(h, "add_candidates"); if (rcp) return rcp; }
    { int32_t rcf = consume_found(h); if (rcf) return rcf; }   // keep the arrival order: what the reads gave so far comes first
    std::vector<HostCandidate> list;
    int32_t rc = host_candidates_of(h, cands, n, alleles, allele_bytes, list, "add_candidates");
    if (rc) return rc;
    for (auto& c : list) add_candidate(h, c);
    return PISCES_OK;
    });
}

// -forcedalleles (Factory.GetForcedAlleles :56-96, SelectForcedAllele :270-286; SmallVariantCaller.CreateForcedAllelePos :49-77): the
// alleles to report whatever the reads say.  Categories are SmallVariantCaller.GetAlleleCategory's (:141-150), support is ignored.
int32_t pisces_hip_set_forced_alleles(PiscesHip* h, const PiscesCandidate* cands, int64_t n, const uint8_t* alleles, int64_t allele_bytes)
{
    return abi_guard<int32_t>(h, [&]() -> int32_t {
    if (!h) return PISCES_E_INVALID_ARG;
    if (h->n_forced_added > 0) return fail(h, PISCES_E_INVALID_ARG, "set_forced_alleles: some forced alleles are candidates already");
    std::vector<HostCandidate> list;
    int32_t rc = host_candidates_of(h, cands, n, alleles, allele_bytes, list, "set_forced_alleles");
    if (rc) return rc;
    h->forced.clear();
    h->forced_keys.clear();
    h->forced_positions.clear();
    for (auto& c : list) {
        // IsValidAlt :88-96
        if (c.ref == c.alt) continue;
        bool acgt = true;
        for (char ch : c.alt) acgt = acgt && (ch == 'A' || ch == 'C' || ch == 'G' || ch == 'T');
        if (!acgt) continue;
        if (!h->intervals.empty()) {   // SelectForcedAllele: inside the intervals only
            bool inside = false;
            for (auto& iv : h->intervals) inside = inside || (c.position >= iv.first && c.position <= iv.second);
            if (!inside) continue;
        }
        c.category = (c.ref.size() == 1 && c.alt.size() == 1) ? PISCES_CAT_SNV : c.ref.size() == c.alt.size() ? PISCES_CAT_MNV
                     : c.ref.size() > c.alt.size() ? PISCES_CAT_DELETION : PISCES_CAT_INSERTION;
        for (int d = 0; d < 3; d++) c.support_by_dir[d] = c.well_anchored_by_dir[d] = 0;
        c.open_left = c.open_right = false;
        if (!h->forced_keys.insert(forced_key(c.position, c.ref, c.alt)).second) continue;   // a HashSet
        h->forced_positions.insert(c.position);
        h->forced.push_back(c);
    }
    std::stable_sort(h->forced.begin(), h->forced.end(), [](const HostCandidate& a, const HostCandidate& b) { return a.position < b.position; });
    return PISCES_OK;
    });
}

// SmallVariantCaller.AddForcedAlleleAsCandidate :118-132, before GetCandidatesToProcess(upTo)
static void add_forced_as_candidates(PiscesHip* h, int32_t up_to_position)
{
    while (h->n_forced_added < h->forced.size()) {
        const HostCandidate& c = h->forced[h->n_forced_added];
        if (up_to_position >= 0 && c.position > up_to_position) break;
        add_candidate(h, c);
        h->n_forced_added++;
    }
}

// Candidate discovery for a read batch that is on the device (find_count / found_scan / find_emit kernels), enqueued on the handle's
// stream; its records come back into pinned memory and are merged by consume_found when they are needed.  d_slots: the record slots
// the host reserved per read from the CIGARs (MNV calling off), found_slots / found_pool their totals.
static int32_t enqueue_candidate_discovery(PiscesHip* h, const DevReadBatch& db, const uint8_t* d_deldirs, int32_t nr, const int32_t* d_slots_in,
                                           int64_t found_slots, int64_t found_pool)
{
    const int32_t minBQ = h->cfg.min_base_call_quality;
    const int32_t* d_slots = d_slots_in;
        const FinderParams FP = {minBQ, PISCES_ANCHOR_SIZE, h->cfg.call_mnvs ? 1 : 0, h->cfg.call_mnvs ? 1 : 0, h->cfg.max_mnv_length,
                                 h->cfg.max_gap_between_mnv};
        const unsigned grid = (unsigned)((nr + 255) / 256);
        PISCES_HIP_CHECK(h, h->d_found_misc.reserve(4));
        PISCES_HIP_CHECK(h, hipMemsetAsync(h->d_found_misc.p, 0, 4 * sizeof(unsigned int), h->stream));
        const int32_t* d_pool_first = nullptr;
        if (h->cfg.call_mnvs) {
            // count, scan (one more element than reads: the last one receives the total), then size the record buffer
            PISCES_HIP_CHECK(h, h->d_found_slots.reserve((size_t)nr + 1));
            PISCES_HIP_CHECK(h, h->d_found_pool_first.reserve((size_t)nr + 1));
            PISCES_HIP_CHECK(h, h->d_found_totals.reserve(2));
            PISCES_HIP_CHECK(h, hipMemsetAsync(h->d_found_slots.p + nr, 0, sizeof(int32_t), h->stream));
            PISCES_HIP_CHECK(h, hipMemsetAsync(h->d_found_pool_first.p + nr, 0, sizeof(int32_t), h->stream));
            hipLaunchKernelGGL(find_count_kernel, dim3(grid), dim3(256), 0, h->stream, db, d_deldirs, (const uint8_t*)h->d_ref.p, h->ref_len, FP,
                               h->d_found_slots.p, h->d_found_pool_first.p);
            hipLaunchKernelGGL(found_scan_kernel, dim3(1), dim3(1024), 0, h->stream, h->d_found_slots.p, h->d_found_pool_first.p, nr + 1,
                               h->d_found_totals.p);
            long long totals[2] = {0, 0};
            PISCES_HIP_CHECK(h, hipMemcpyAsync(totals, h->d_found_totals.p, sizeof(totals), hipMemcpyDeviceToHost, h->stream));
            PISCES_HIP_CHECK(h, hipStreamSynchronize(h->stream));
            if (totals[0] > 0x7FFFFFF0ll || totals[1] > 0x7FFFFFF0ll) return fail(h, PISCES_E_INVALID_ARG, "add_reads: too many candidates in one batch");
            found_slots = totals[0];
            found_pool = totals[1];
            d_slots = h->d_found_slots.p;
            d_pool_first = h->d_found_pool_first.p;
        }
        if (found_slots > 0) {
            PISCES_HIP_CHECK(h, h->d_found.reserve((size_t)found_slots));
            PISCES_HIP_CHECK(h, h->d_found_pool.reserve((size_t)found_pool + 16));
            hipLaunchKernelGGL(find_emit_kernel, dim3(grid), dim3(256), 0, h->stream, db, d_deldirs, (const uint8_t*)h->d_ref.p, h->ref_len, FP, d_slots,
                               d_pool_first, h->d_found.p, h->d_found_pool.p, h->d_found_misc.p, (int32_t)found_pool, (int32_t*)(h->d_found_misc.p + 1));
            PISCES_HIP_CHECK(h, hipGetLastError());
            // records + pool + {cursor, overflow} come back into pinned memory; consume_found waits for them when they are needed
            const size_t rec_bytes = (size_t)found_slots * sizeof(DevFound), pool_al = ((size_t)found_pool + 15) & ~(size_t)15;
            const size_t need = rec_bytes + pool_al + 16;
            if (need > h->found.h_cap) {
                if (h->found.h) (void)hipHostFree(h->found.h);
                h->found.h = nullptr;
                h->found.h_cap = 0;
                PISCES_HIP_CHECK(h, hipHostMalloc((void**)&h->found.h, need + need / 2, hipHostMallocDefault));
                h->found.h_cap = need + need / 2;
            }
            if (!h->found.done) PISCES_HIP_CHECK(h, hipEventCreateWithFlags(&h->found.done, hipEventDisableTiming));
            PISCES_HIP_CHECK(h, hipMemcpyAsync(h->found.h, h->d_found.p, rec_bytes, hipMemcpyDeviceToHost, h->stream));
            if (found_pool > 0)
                PISCES_HIP_CHECK(h, hipMemcpyAsync(h->found.h + rec_bytes, h->d_found_pool.p, (size_t)found_pool, hipMemcpyDeviceToHost, h->stream));
            PISCES_HIP_CHECK(h, hipMemcpyAsync(h->found.h + rec_bytes + pool_al, h->d_found_misc.p, 2 * sizeof(unsigned int), hipMemcpyDeviceToHost, h->stream));
            PISCES_HIP_CHECK(h, hipEventRecord(h->found.done, h->stream));
            h->found.n_slots = found_slots;
            h->found.pool_bytes = found_pool;
            h->found.in_flight = true;
        }
    return PISCES_OK;
}

// The candidates the device found for the last add_reads (find_emit_kernel), merged into their blocks in read order:
// IStateManager.AddCandidates (SmallVariantCaller.cs:92-96).  Called before anything that looks at the candidates.
static int32_t consume_found(PiscesHip* h)
{
    if (!h->found.in_flight) return PISCES_OK;
    h->found.in_flight = false;
    PISCES_HIP_CHECK(h, hipEventSynchronize(h->found.done));
    const DevFound* recs = (const DevFound*)h->found.h;
    const uint8_t* pool = h->found.h + (size_t)h->found.n_slots * sizeof(DevFound);
    const unsigned int* misc = (const unsigned int*)(pool + (((size_t)h->found.pool_bytes + 15) & ~(size_t)15));
    if (misc[1] != 0) return fail(h, PISCES_E_DEVICE, "add_reads: the candidate records of the device did not fit their reservation");
    for (int64_t i = 0; i < h->found.n_slots; i++) {
        const DevFound& f = recs[i];
        if (f.c.category == kFoundHole) continue;
        const uint8_t* bases = f.pool_offset >= 0 ? pool + f.pool_offset : f.alt;
        add_candidate(h, host_candidate_of(f.c, h->h_ref.data(), bases));
    }
    return PISCES_OK;
}

int32_t pisces_hip_add_reads(PiscesHip* h, const PiscesReadBatch* batch)
{
    return abi_guard<int32_t>(h, [&]() -> int32_t {
    if (!h) return PISCES_E_INVALID_ARG;
    if (validate_batch(batch) != PISCES_OK) return fail(h, PISCES_E_INVALID_ARG, "add_reads: malformed read batch");
    { int32_t rcp = refuse_while_batch_is_open(h, "add_reads"); if (rcp) return rcp; }
    if (batch->n_reads == 0) return PISCES_OK;
    PISCES_HIP_CHECK(h, hipSetDevice(h->device));
    { int32_t rcf = consume_found(h); if (rcf) return rcf; }
    const int32_t nr = batch->n_reads;
    const int32_t minBQ = h->cfg.min_base_call_quality;
    // ---- host pass over the CIGARs only (never over the bases): argument checks of the reference's walk, the insertion /
    // deletion candidates, the blocks the read touches, and an upper bound of its observations ----
    auto op_ref = [](uint8_t t) { return t == 'M' || t == 'D' || t == 'N' || t == '=' || t == 'X'; };
    auto op_read = [](uint8_t t) { return t == 'M' || t == 'I' || t == 'S' || t == '=' || t == 'X'; };
    int64_t ub = 0;
    std::vector<long long>& slots = h->read_slots;   // log slots reserved per read: [slots[i], slots[i + 1])
    slots.resize((size_t)nr + 1);
    for (int32_t i = 0; i < nr; i++) {
        ReadView r = read_view(batch, i);
        slots[(size_t)i] = (long long)(h->log_ub + ub);
        if (r.position <= 0) return fail(h, PISCES_E_INVALID_ARG, "Position must be greater than 0.");
        if (r.read_len < 0 || r.n_cigar < 0) return fail(h, PISCES_E_INVALID_ARG, "add_reads: malformed read batch");
        int64_t read_span = 0, ref_span = 0;
        for (int c = 0; c < r.n_cigar; c++) {
            const uint8_t t = r.cigar_op[c];
            if (op_read(t)) read_span += r.cigar_len[c];
            if (op_ref(t)) ref_span += r.cigar_len[c];   // mapped bases + every gap: one observation each at most
        }
        if (read_span > r.read_len) return fail(h, PISCES_E_INVALID_ARG, "add_reads: CIGAR does not match the read");
        if ((int64_t)r.position + ref_span > 0x7FFFFFFFll) return fail(h, PISCES_E_INVALID_ARG, "add_reads: read runs past position 2^31 - 1");
        if (r.dirs)
            for (int k = 0; k < r.read_len; k++)
                if (r.dirs[k] > 2) return fail(h, PISCES_E_INVALID_ARG, "add_reads: CIGAR does not match the read");
        if (r.del_dirs)
            for (int c = 0; c < r.n_cigar; c++)
                if (r.cigar_op[c] == 'D')
                    for (int k = 0; k < 2; k++)
                        if (r.del_dirs[2 * c + k] > 2 && r.del_dirs[2 * c + k] != PISCES_DIR_UNTRACKED)
                            return fail(h, PISCES_E_INVALID_ARG, "add_reads: deletion_directions holds a value that is no DirectionType");
        ub += ref_span;
    }
    slots[(size_t)nr] = (long long)(h->log_ub + ub);
    // ---- the read batch crosses PCIe once, packed; the walk runs on the device (expand_reads_kernel).  The transfer is started
    // BEFORE the second host pass over the CIGARs (block bookkeeping, candidate slots): that pass runs under it, and only its small
    // table of candidate slots follows in a transfer of its own ----
    const size_t n_cig = (size_t)batch->cigar_offset[nr], n_seq = (size_t)batch->seq_offset[nr];
    auto align16 = [](size_t x) { return (x + 15) & ~(size_t)15; };
    size_t off_pos = 0, off_flags = align16(off_pos + (size_t)nr * 4), off_coff = align16(off_flags + (size_t)nr),
           off_cop = align16(off_coff + ((size_t)nr + 1) * 4), off_clen = align16(off_cop + n_cig),
           off_soff = align16(off_clen + n_cig * 4), off_bases = align16(off_soff + ((size_t)nr + 1) * 4),
           off_quals = align16(off_bases + n_seq), off_dirs = align16(off_quals + n_seq),
           off_slots = align16(off_dirs + (batch->directions ? n_seq : 0)), off_deldirs = align16(off_slots + ((size_t)nr + 1) * 8),
           off_fslots = align16(off_deldirs + (batch->deletion_directions ? 2 * n_cig : 0)), total = align16(off_fslots + ((size_t)nr + 1) * 4);
    int32_t rc = stage_reserve(h, total);
    if (rc) return rc;
    rc = log_reserve(h, ub);
    if (rc) return rc;
    uint8_t* st = h->h_stage;
    std::memcpy(st + off_pos, batch->position, (size_t)nr * 4);
    std::memcpy(st + off_flags, batch->flags, (size_t)nr);
    std::memcpy(st + off_coff, batch->cigar_offset, ((size_t)nr + 1) * 4);
    std::memcpy(st + off_cop, batch->cigar_op, n_cig);
    std::memcpy(st + off_clen, batch->cigar_len, n_cig * 4);
    std::memcpy(st + off_soff, batch->seq_offset, ((size_t)nr + 1) * 4);
    std::memcpy(st + off_slots, slots.data(), ((size_t)nr + 1) * 8);
    if (batch->deletion_directions) std::memcpy(st + off_deldirs, batch->deletion_directions, 2 * n_cig);
    {
        // bases / qualities / directions are the bulk (2-3 bytes per aligned base).  Small batches: one copy into the pinned buffer,
        // one transfer.  Large ones: slices of 8 MB, each copied by a few threads and handed to the DMA engine as soon as it is
        // complete, so that the host copy of slice k+1 runs under the PCIe transfer of slice k.
        struct Seg { size_t dst; const uint8_t* src; size_t len; };
        const Seg segs[3] = {{off_bases, batch->bases, n_seq}, {off_quals, batch->quals, n_seq},
                             {off_dirs, batch->directions, batch->directions ? n_seq : 0}};
        const size_t bulk = 2 * n_seq + (batch->directions ? n_seq : 0);
        constexpr size_t kSlice = (size_t)8 << 20;
        if (bulk < 2 * kSlice) {
            for (const Seg& g : segs) if (g.len) std::memcpy(st + g.dst, g.src, g.len);
            PISCES_HIP_CHECK(h, hipMemcpyAsync(D_STAGE(h), st, off_fslots, hipMemcpyHostToDevice, h->stream));
        } else {
            // everything outside the bulk first (the descriptors before it, the slot table after it)
            PISCES_HIP_CHECK(h, hipMemcpyAsync(D_STAGE(h), st, off_bases, hipMemcpyHostToDevice, h->stream));
            PISCES_HIP_CHECK(h, hipMemcpyAsync(D_STAGE(h) + off_slots, st + off_slots, off_fslots - off_slots, hipMemcpyHostToDevice, h->stream));
            struct Slice { size_t dst; const uint8_t* src; size_t len; };
            std::vector<Slice> slices;
            for (const Seg& g : segs)
                for (size_t o = 0; o < g.len; o += kSlice) slices.push_back({g.dst + o, g.src + o, std::min(kSlice, g.len - o)});
            const int n_threads = (int)std::min<size_t>(4, std::max<unsigned>(1u, std::thread::hardware_concurrency()));
            std::vector<std::atomic<int>> parts_done(slices.size());
            for (auto& a : parts_done) a.store(0, std::memory_order_relaxed);
            auto worker = [&](int w) {
                for (size_t k = 0; k < slices.size(); k++) {
                    const size_t per = (slices[k].len + (size_t)n_threads - 1) / (size_t)n_threads, lo = std::min(slices[k].len, per * (size_t)w),
                                 hi = std::min(slices[k].len, lo + per);
                    if (hi > lo) std::memcpy(st + slices[k].dst + lo, slices[k].src + lo, hi - lo);
                    parts_done[k].fetch_add(1, std::memory_order_release);
                }
            };
            std::vector<std::thread> pool;
            for (int w = 1; w < n_threads; w++) pool.emplace_back(worker, w);
            hipError_t first_error = hipSuccess;
            {
                // this thread copies its share of a slice, then waits for the others' and enqueues the transfer
                for (size_t k = 0; k < slices.size(); k++) {
                    const size_t per = (slices[k].len + (size_t)n_threads - 1) / (size_t)n_threads, hi = std::min(slices[k].len, per);
                    if (hi) std::memcpy(st + slices[k].dst, slices[k].src, hi);
                    parts_done[k].fetch_add(1, std::memory_order_release);
                    while (parts_done[k].load(std::memory_order_acquire) < n_threads) std::this_thread::yield();
                    if (first_error == hipSuccess)
                        first_error = hipMemcpyAsync(D_STAGE(h) + slices[k].dst, st + slices[k].dst, slices[k].len, hipMemcpyHostToDevice, h->stream);
                }
            }
            for (auto& t : pool) t.join();
            PISCES_HIP_CHECK(h, first_error);
        }
    }
    // ICandidateVariantFinder.FindCandidates + IStateManager.AddCandidates (SmallVariantCaller.cs:92-96) run on the device
    // (find_emit_kernel, enqueued behind the read walk below).  With MNV calling off only insertions and deletions are discovered
    // (SNV candidates are implied by the allele counts): the host reserves one record slot per I / D operation here, from the CIGAR
    // alone; with it on the device counts its candidates itself.
    const bool find_on_device = !h->h_ref.empty();   // without a reference only the IStateManager half (allele counts) runs
    std::vector<int32_t>& fslots = h->found_slots_host;
    fslots.assign((size_t)nr + 1, 0);
    int64_t found_slots = 0, found_pool = 0;
    for (int32_t i = 0; i < nr; i++) {
        ReadView r = read_view(batch, i);
        fslots[(size_t)i] = (int32_t)found_slots;
        if (find_on_device && !h->cfg.call_mnvs)
            for (int c = 0; c < r.n_cigar; c++) {
                if (r.cigar_op[c] == 'I' || r.cigar_op[c] == 'D') found_slots++;
                if (r.cigar_op[c] == 'I' && r.cigar_len[c] > (uint32_t)kFoundInline) found_pool += r.cigar_len[c];
            }
        if (found_slots > 0x7FFFFFF0ll || found_pool > 0x7FFFFFF0ll) {
            (void)stage_release(h);   // (the batch's transfer is in flight out of the staging pair)
            return fail(h, PISCES_E_INVALID_ARG, "add_reads: too many insertions / deletions in one batch");
        }
        // GetBlock(position) for every position that receives a count (RegionStateManager.cs:361-383): the runs of mapped
        // bases always do; a gap (deletion / skip) does when its flanking qualities pass CheckDeletionQuality
        {
            auto touch = [&](int64_t from, int64_t to) {   // inclusive
                if (to < 1) return;
                if (from < 1) from = 1;
                for (int32_t k = block_key(h, (int32_t)from); k <= block_key(h, (int32_t)to); k++) (void)get_block(h, (k - 1) * h->cfg.block_size + 1);
            };
            auto delq = [&](int idx) {
                if (r.read_len == 0) return false;
                const int after = idx < r.read_len ? r.quals[idx] : r.quals[idx - 1];
                const int before = idx > 0 ? r.quals[idx - 1] : after;
                return before >= minBQ && after >= minBQ;
            };
            int64_t rp = r.position, last_mapped = (int64_t)r.position - 1;
            int ri = 0;
            for (int c = 0; c < r.n_cigar; c++) {
                const uint8_t t = r.cigar_op[c];
                const int64_t len = r.cigar_len[c];
                if (op_read(t) && op_ref(t) && len > 0) {
                    if (rp > last_mapped + 1 && ri < r.read_len && delq(ri)) touch(last_mapped + 1, rp - 1);
                    touch(rp, rp + len - 1);
                    last_mapped = rp + len - 1;
                }
                if (op_ref(t)) rp += len;
                if (op_read(t)) ri += (int)len;
            }
            const int nc = r.n_cigar;
            const bool ends_del = nc >= 1 && r.cigar_op[nc - 1] == 'D';
            const bool ends_del_soft = nc >= 2 && r.cigar_op[nc - 2] == 'D' && r.cigar_op[nc - 1] == 'S';
            if (ends_del && r.read_len > 0 && delq(r.read_len - 1)) touch(last_mapped + 1, last_mapped + r.cigar_len[nc - 1]);
            if (ends_del_soft) {
                const int idx = r.read_len - (int)r.cigar_len[nc - 1];
                if (idx >= 0 && idx < r.read_len && delq(idx)) touch(last_mapped + 1, last_mapped + r.cigar_len[nc - 2]);
            }
        }
        h->stats[2] += 1;
    }
    fslots[(size_t)nr] = (int32_t)found_slots;

    std::memcpy(st + off_fslots, fslots.data(), ((size_t)nr + 1) * 4);
    PISCES_HIP_CHECK(h, hipMemcpyAsync(D_STAGE(h) + off_fslots, st + off_fslots, total - off_fslots, hipMemcpyHostToDevice, h->stream));
    DevReadBatch db;
    const uint8_t* d = D_STAGE(h);
    db.position = (const int32_t*)(d + off_pos);
    db.flags = d + off_flags;
    db.cigar_offset = (const int32_t*)(d + off_coff);
    db.cigar_op = d + off_cop;
    db.cigar_len = (const uint32_t*)(d + off_clen);
    db.seq_offset = (const int32_t*)(d + off_soff);
    db.bases = d + off_bases;
    db.quals = d + off_quals;
    db.dirs = batch->directions ? d + off_dirs : nullptr;
    db.n_reads = nr;
    const int c = h->log_cur;
    hipLaunchKernelGGL(expand_reads_kernel, dim3((unsigned)((nr + 3) / 4)), dim3(256), 0, h->stream, db, (const long long*)(d + off_slots),
                       minBQ, h->d_log_pos[c].p, h->d_log_tup[c].p, h->d_log_n.p + 2);
    PISCES_HIP_CHECK(h, hipGetLastError());
    if (find_on_device && (h->cfg.call_mnvs || found_slots > 0)) {
        int32_t rcd = enqueue_candidate_discovery(h, db, batch->deletion_directions ? d + off_deldirs : nullptr, nr, (const int32_t*)(d + off_fslots),
                                                  found_slots, found_pool);
        if (rcd) return rcd;
    }
    { int32_t rcs = stage_release(h); if (rcs) return rcs; }
    h->log_ub += ub;
    return PISCES_OK;
    });
}

int64_t pisces_hip_expand_reads(const PiscesReadBatch* batch, int32_t min_bq, int32_t* positions, uint32_t* tuples, int64_t capacity)
{
    return abi_guard<int64_t>((PiscesHip*)nullptr, [&]() -> int64_t {
    if (validate_batch(batch) != PISCES_OK || capacity < 0 || (capacity > 0 && (!positions || !tuples))) return PISCES_E_INVALID_ARG;
    ArraySink sink;
    sink.positions = positions;
    sink.tuples = tuples;
    sink.capacity = capacity;
    for (int32_t i = 0; i < batch->n_reads; i++) {
        int32_t rc = expand_read(read_view(batch, i), min_bq, sink);
        if (rc != PISCES_OK) return rc;
    }
    return sink.n <= capacity ? sink.n : (int64_t)PISCES_E_BUFFER_TOO_SMALL;
    });
}

int64_t pisces_hip_find_candidates(const PiscesReadBatch* batch, const uint8_t* ref, int64_t ref_len, int32_t min_bq, int32_t snvs_and_mnvs,
                                   int32_t call_mnvs, int32_t max_mnv_length, int32_t max_gap_between_mnv, PiscesCandidate* out,
                                   int64_t capacity, uint8_t* alleles, int64_t allele_capacity, int64_t* allele_bytes)
{
    return abi_guard<int64_t>((PiscesHip*)nullptr, [&]() -> int64_t {
    if (validate_batch(batch) != PISCES_OK || !ref || ref_len <= 0 || capacity < 0 || (capacity > 0 && !out)) return PISCES_E_INVALID_ARG;
    std::vector<HostCandidate> found;
    try {
        for (int32_t i = 0; i < batch->n_reads; i++)
            find_candidates(read_view(batch, i), ref, ref_len, min_bq, PISCES_ANCHOR_SIZE, snvs_and_mnvs != 0, call_mnvs != 0, max_mnv_length,
                            max_gap_between_mnv, found);
    } catch (...) {   // nothing crosses the C ABI as an exception
        return PISCES_E_INVALID_ARG;
    }
    int64_t bytes = 0;
    for (size_t i = 0; i < found.size(); i++) {
        const HostCandidate& c = found[i];
        const int64_t need = (int64_t)(c.ref.size() + c.alt.size());
        if ((int64_t)i < capacity && (!alleles || bytes + need <= allele_capacity)) {
            PiscesCandidate& o = out[i];
            std::memset(&o, 0, sizeof(o));
            o.position = c.position; o.category = c.category;
            o.ref_len = (int32_t)c.ref.size(); o.alt_len = (int32_t)c.alt.size();
            for (int d = 0; d < 3; d++) { o.support_by_dir[d] = c.support_by_dir[d]; o.well_anchored_by_dir[d] = c.well_anchored_by_dir[d]; }
            o.open_left = c.open_left; o.open_right = c.open_right;
            o.allele_offset = bytes;
            if (alleles) {
                std::memcpy(alleles + bytes, c.ref.data(), c.ref.size());
                std::memcpy(alleles + bytes + c.ref.size(), c.alt.data(), c.alt.size());
            }
        }
        bytes += need;
    }
    if (allele_bytes) *allele_bytes = bytes;
    if ((int64_t)found.size() > capacity || (alleles && bytes > allele_capacity)) return PISCES_E_BUFFER_TOO_SMALL;
    return (int64_t)found.size();
    });
}

static int64_t export_candidates(const std::vector<HostCandidate>& found, PiscesCandidate* out, int64_t capacity, uint8_t* alleles,
                                 int64_t allele_capacity, int64_t* allele_bytes)
{
    int64_t bytes = 0;
    for (size_t i = 0; i < found.size(); i++) {
        const HostCandidate& c = found[i];
        const int64_t need = (int64_t)(c.ref.size() + c.alt.size());
        if ((int64_t)i < capacity && (!alleles || bytes + need <= allele_capacity)) {
            PiscesCandidate& o = out[i];
            std::memset(&o, 0, sizeof(o));
            o.position = c.position; o.category = c.category;
            o.ref_len = (int32_t)c.ref.size(); o.alt_len = (int32_t)c.alt.size();
            for (int d = 0; d < 3; d++) { o.support_by_dir[d] = c.support_by_dir[d]; o.well_anchored_by_dir[d] = c.well_anchored_by_dir[d]; }
            o.open_left = c.open_left; o.open_right = c.open_right;
            o.allele_offset = bytes;
            if (alleles) {
                std::memcpy(alleles + bytes, c.ref.data(), c.ref.size());
                std::memcpy(alleles + bytes + c.ref.size(), c.alt.data(), c.alt.size());
            }
        }
        bytes += need;
    }
    if (allele_bytes) *allele_bytes = bytes;
    if ((int64_t)found.size() > capacity || (alleles && bytes > allele_capacity)) return PISCES_E_BUFFER_TOO_SMALL;
    return (int64_t)found.size();
}

int64_t pisces_hip_find_candidates_device(PiscesHip* h, const PiscesReadBatch* batch, int32_t snvs_and_mnvs, int32_t call_mnvs,
                                          int32_t max_mnv_length, int32_t max_gap_between_mnv, PiscesCandidate* out, int64_t capacity,
                                          uint8_t* alleles, int64_t allele_capacity, int64_t* allele_bytes)
{
    return abi_guard<int64_t>(h, [&]() -> int64_t {
    if (!h) return PISCES_E_INVALID_ARG;
    if (validate_batch(batch) != PISCES_OK || capacity < 0 || (capacity > 0 && !out)) return fail(h, PISCES_E_INVALID_ARG, "find_candidates_device: malformed arguments");
    if (h->h_ref.empty()) return fail(h, PISCES_E_STATE, "find_candidates_device: set_reference has not been called");
    if (allele_bytes) *allele_bytes = 0;
    if (batch->n_reads == 0) return 0;
    PISCES_HIP_CHECK(h, hipSetDevice(h->device));
    const int32_t nr = batch->n_reads;
    for (int32_t i = 0; i < nr; i++) {
        const ReadView r = read_view(batch, i);
        int64_t read_span = 0;
        for (int c = 0; c < r.n_cigar; c++)
            if (r.cigar_op[c] == 'M' || r.cigar_op[c] == 'I' || r.cigar_op[c] == 'S' || r.cigar_op[c] == '=' || r.cigar_op[c] == 'X') read_span += r.cigar_len[c];
        if (r.position <= 0 || r.read_len < 0 || read_span > r.read_len) return fail(h, PISCES_E_INVALID_ARG, "find_candidates_device: CIGAR does not match the read");
    }
    // the batch on the device (local buffers: this entry is a test / tooling surface, not the streaming path)
    const size_t n_cig = (size_t)batch->cigar_offset[nr], n_seq = (size_t)batch->seq_offset[nr];
    DeviceBuf<int32_t> d_pos, d_coff, d_soff, d_cnt, d_pool_first;
    DeviceBuf<uint8_t> d_flags, d_cop, d_bases, d_quals, d_dirs, d_deldirs, d_pool;
    DeviceBuf<uint32_t> d_clen;
    DeviceBuf<long long> d_totals;
    DeviceBuf<unsigned int> d_misc;
    DeviceBuf<DevFound> d_out;
    auto up = [&](auto& buf, const void* src, size_t n_elems, size_t elem) -> hipError_t {
        hipError_t e = buf.reserve(std::max<size_t>(n_elems, 1));
        if (e != hipSuccess || n_elems == 0) return e;
        return hipMemcpyAsync(buf.p, src, n_elems * elem, hipMemcpyHostToDevice, h->stream);
    };
    PISCES_HIP_CHECK(h, up(d_pos, batch->position, (size_t)nr, 4));
    PISCES_HIP_CHECK(h, up(d_flags, batch->flags, (size_t)nr, 1));
    PISCES_HIP_CHECK(h, up(d_coff, batch->cigar_offset, (size_t)nr + 1, 4));
    PISCES_HIP_CHECK(h, up(d_cop, batch->cigar_op, n_cig, 1));
    PISCES_HIP_CHECK(h, up(d_clen, batch->cigar_len, n_cig, 4));
    PISCES_HIP_CHECK(h, up(d_soff, batch->seq_offset, (size_t)nr + 1, 4));
    PISCES_HIP_CHECK(h, up(d_bases, batch->bases, n_seq, 1));
    PISCES_HIP_CHECK(h, up(d_quals, batch->quals, n_seq, 1));
    if (batch->directions) PISCES_HIP_CHECK(h, up(d_dirs, batch->directions, n_seq, 1));
    if (batch->deletion_directions) PISCES_HIP_CHECK(h, up(d_deldirs, batch->deletion_directions, 2 * n_cig, 1));
    DevReadBatch db;
    db.position = d_pos.p; db.flags = d_flags.p; db.cigar_offset = d_coff.p; db.cigar_op = d_cop.p; db.cigar_len = d_clen.p;
    db.seq_offset = d_soff.p; db.bases = d_bases.p; db.quals = d_quals.p; db.dirs = batch->directions ? d_dirs.p : nullptr; db.n_reads = nr;
    const uint8_t* dd = batch->deletion_directions ? d_deldirs.p : nullptr;
    const FinderParams FP = {h->cfg.min_base_call_quality, PISCES_ANCHOR_SIZE, snvs_and_mnvs ? 1 : 0, call_mnvs ? 1 : 0, max_mnv_length, max_gap_between_mnv};
    const unsigned grid = (unsigned)((nr + 255) / 256);
    PISCES_HIP_CHECK(h, d_cnt.reserve((size_t)nr + 1));
    PISCES_HIP_CHECK(h, d_pool_first.reserve((size_t)nr + 1));
    PISCES_HIP_CHECK(h, d_totals.reserve(2));
    PISCES_HIP_CHECK(h, d_misc.reserve(4));
    PISCES_HIP_CHECK(h, hipMemsetAsync(d_misc.p, 0, 4 * sizeof(unsigned int), h->stream));
    PISCES_HIP_CHECK(h, hipMemsetAsync(d_cnt.p + nr, 0, sizeof(int32_t), h->stream));
    PISCES_HIP_CHECK(h, hipMemsetAsync(d_pool_first.p + nr, 0, sizeof(int32_t), h->stream));
    hipLaunchKernelGGL(find_count_kernel, dim3(grid), dim3(256), 0, h->stream, db, dd, (const uint8_t*)h->d_ref.p, h->ref_len, FP, d_cnt.p, d_pool_first.p);
    hipLaunchKernelGGL(found_scan_kernel, dim3(1), dim3(1024), 0, h->stream, d_cnt.p, d_pool_first.p, nr + 1, d_totals.p);
    long long totals[2] = {0, 0};
    PISCES_HIP_CHECK(h, hipMemcpyAsync(totals, d_totals.p, sizeof(totals), hipMemcpyDeviceToHost, h->stream));
    PISCES_HIP_CHECK(h, hipStreamSynchronize(h->stream));
    std::vector<HostCandidate> found;
    if (totals[0] > 0) {
        PISCES_HIP_CHECK(h, d_out.reserve((size_t)totals[0]));
        PISCES_HIP_CHECK(h, d_pool.reserve((size_t)totals[1] + 16));
        hipLaunchKernelGGL(find_emit_kernel, dim3(grid), dim3(256), 0, h->stream, db, dd, (const uint8_t*)h->d_ref.p, h->ref_len, FP, (const int32_t*)d_cnt.p,
                           (const int32_t*)d_pool_first.p, d_out.p, d_pool.p, d_misc.p, (int32_t)totals[1], (int32_t*)(d_misc.p + 1));
        PISCES_HIP_CHECK(h, hipGetLastError());
        std::vector<DevFound> recs((size_t)totals[0]);
        std::vector<uint8_t> pool((size_t)totals[1] + 1);
        unsigned int misc[2] = {0, 0};
        PISCES_HIP_CHECK(h, hipMemcpyAsync(recs.data(), d_out.p, recs.size() * sizeof(DevFound), hipMemcpyDeviceToHost, h->stream));
        if (totals[1] > 0) PISCES_HIP_CHECK(h, hipMemcpyAsync(pool.data(), d_pool.p, (size_t)totals[1], hipMemcpyDeviceToHost, h->stream));
        PISCES_HIP_CHECK(h, hipMemcpyAsync(misc, d_misc.p, sizeof(misc), hipMemcpyDeviceToHost, h->stream));
        PISCES_HIP_CHECK(h, hipStreamSynchronize(h->stream));
        if (misc[1]) return fail(h, PISCES_E_DEVICE, "find_candidates_device: record reservation exceeded");
        for (const DevFound& f : recs) {
            if (f.c.category == kFoundHole) continue;
            found.push_back(host_candidate_of(f.c, h->h_ref.data(), f.pool_offset >= 0 ? pool.data() + f.pool_offset : f.alt));
        }
    }
    return export_candidates(found, out, capacity, alleles, allele_capacity, allele_bytes);
    });
}

int64_t pisces_hip_find_indel_candidates(const PiscesReadBatch* batch, const uint8_t* ref, int64_t ref_len, int32_t min_bq,
                                         PiscesCandidate* out, int64_t capacity, uint8_t* alleles, int64_t allele_capacity,
                                         int64_t* allele_bytes)
{
    return abi_guard<int64_t>((PiscesHip*)nullptr, [&]() -> int64_t {
    return pisces_hip_find_candidates(batch, ref, ref_len, min_bq, 0, 0, 0, 0, out, capacity, alleles, allele_capacity, allele_bytes);
    });
}

// Builds tiles + tile-bucketed tuples for a set of blocks. Tiles follow the 1000-locus block grid
// (clipped to the interval set when one is given); every tile's tuple segment is padded to a
// multiple of 4 tuples so the kernel's 16-byte loads start aligned.
// Tile geometry of the blocks `keys` (ascending): the 64-locus grid of each block, clipped to the interval set when `clip`
// (ChrIntervalSet.GetClipped).  tile_of_locus (per key, block_size entries, relative to the key's first tile) is filled only
// when the grid is irregular, i.e. when intervals clip it.
static void tile_geometry(PiscesHip* h, const std::vector<int32_t>& keys, bool clip, std::vector<PiscesTile>& tiles,
                          std::vector<int32_t>& first_tile, std::vector<int32_t>& tol)
{
    tiles.clear();
    first_tile.clear();
    tol.clear();
    const int bs = h->cfg.block_size;
    const bool irregular = clip && !h->intervals.empty();
    if (irregular) tol.assign(keys.size() * (size_t)bs, -1);
    for (size_t ki = 0; ki < keys.size(); ki++) {
        const int32_t key = keys[ki];
        const int32_t bstart = (key - 1) * bs + 1, bend = key * bs;
        const size_t first = tiles.size();
        first_tile.push_back((int32_t)first);
        auto add_range = [&](int32_t s, int32_t e) {   // inclusive, inside the block
            for (int32_t p = s; p <= e; p += kTile) {
                PiscesTile t;
                t.start_position = p;
                t.n_loci = std::min<int32_t>(kTile, e - p + 1);
                t.tuple_begin = t.tuple_end = 0;
                if (irregular)
                    for (int32_t q = 0; q < t.n_loci; q++) tol[ki * (size_t)bs + (size_t)(p + q - bstart)] = (int32_t)(tiles.size() - first);
                tiles.push_back(t);
            }
        };
        if (!irregular) add_range(bstart, bend);
        else
            for (auto& iv : h->intervals) {
                int32_t s = std::max(iv.first, bstart), e = std::min(iv.second, bend);
                if (s <= e) add_range(s, e);
            }
    }
}

// uploads the BucketMap tables of `keys` and returns the device view
// (n_zero_tail > 0: that many zeroed 32-bit words ride behind the tables in the same transfer — the tile counters of the bucketing —
// and *zero_tail receives their device address)
static int32_t upload_bucket_map(PiscesHip* h, const std::vector<int32_t>& keys, const std::vector<int32_t>& first_tile,
                                 const std::vector<int32_t>& tol, BucketMap* m, size_t n_zero_tail = 0, unsigned int** zero_tail = nullptr)
{
    const int32_t kmin = keys.front(), kmax = keys.back();
    const size_t n_slot = (size_t)(kmax - kmin + 1);
    // the source of an asynchronous copy must stay untouched until the copy has left: a flush uploads at most three maps before its
    // one synchronisation, so a ring of four staging vectors never rewrites one that is still in flight
    std::vector<int32_t>& host = h->bucket_host[h->bucket_host_next];
    h->bucket_host_next = (h->bucket_host_next + 1) % 4;
    const size_t n_tables = n_slot + keys.size() + tol.size();
    host.assign(n_tables, -1);
    for (size_t i = 0; i < keys.size(); i++) host[(size_t)(keys[i] - kmin)] = (int32_t)i;
    std::copy(first_tile.begin(), first_tile.end(), host.begin() + (std::ptrdiff_t)n_slot);
    std::copy(tol.begin(), tol.end(), host.begin() + (std::ptrdiff_t)(n_slot + keys.size()));
    host.resize(n_tables + n_zero_tail, 0);
    PISCES_HIP_CHECK(h, h->d_bucket.reserve(host.size()));
    if (zero_tail) *zero_tail = (unsigned int*)(h->d_bucket.p + n_tables);
    { int32_t rcu = meta_upload(h, h->d_bucket.p, host.data(), host.size() * sizeof(int32_t)); if (rcu) return rcu; }
    m->key_slot = h->d_bucket.p;
    m->first_tile = h->d_bucket.p + n_slot;
    m->tile_of_locus = tol.empty() ? nullptr : h->d_bucket.p + n_slot + keys.size();
    m->key_min = kmin;
    m->key_max = kmax;
    m->block_size = h->cfg.block_size;
    return PISCES_OK;
}

static unsigned log_grid(const PiscesHip* h) { return (unsigned)std::max<int64_t>(1, (h->log_ub + kLogChunk - 1) / kLogChunk); }

// The observations of the blocks `keys`, bucketed by tile into h->d_tuples with the segments in h->d_tiles (device side of
// what a counting sort on the host used to do).  The log itself is left as it is.  `tiles` receives the geometry.
static int32_t bucket_blocks(PiscesHip* h, const std::vector<int32_t>& keys, bool clip, std::vector<PiscesTile>& tiles)
{
    std::vector<int32_t> first_tile, tol;
    tile_geometry(h, keys, clip, tiles, first_tile, tol);
    if (tiles.empty()) return PISCES_OK;
    const int32_t n_tiles = (int32_t)tiles.size();
    // Every stream operation of a flush costs ~4.5 us whatever its size (a 1000-locus block's whole flush is ~130 us of device time), so
    // there are as few as can be: the tile counters arrive zeroed behind the bucket tables, the drop's counter is cleared by the scan,
    // and the tuple buffer is not filled at all (a tile's segment is padded to a multiple of four tuples only so that the next segment
    // starts aligned: no kernel reads past tuple_end).
    BucketMap m;
    unsigned int* tile_cnt = nullptr;
    int32_t rc = upload_bucket_map(h, keys, first_tile, tol, &m, tiles.size(), &tile_cnt);
    if (rc) return rc;
    PISCES_HIP_CHECK(h, h->d_tiles.reserve(tiles.size()));
    PISCES_HIP_CHECK(h, h->d_tile_results.reserve(tiles.size()));
    PISCES_HIP_CHECK(h, h->d_count.reserve(4));
    PISCES_HIP_CHECK(h, h->d_total.reserve(2));
    const size_t tup_ub = (size_t)h->log_ub + 3 * tiles.size() + 4;
    PISCES_HIP_CHECK(h, h->d_tuples.reserve(tup_ub));
    { int32_t rcu = meta_upload(h, h->d_tiles.p, tiles.data(), tiles.size() * sizeof(PiscesTile)); if (rcu) return rcu; }
    const int c = h->log_cur;
    if (h->log_ub > 0) {
        hipLaunchKernelGGL(bucket_count_kernel, dim3(log_grid(h)), dim3(256), 0, h->stream, h->d_log_pos[c].p, (long long)h->log_ub, m,
                           tile_cnt);
    }
    hipLaunchKernelGGL(bucket_scan_kernel, dim3(1), dim3(1024), 0, h->stream, h->d_tiles.p, n_tiles, tile_cnt, h->d_total.p,
                       h->d_log_n.p + (c ^ 1));
    h->drop_counter_cleared = true;   // (by the scan above: enqueue_drop of the same submission needs no fill)
    if (h->log_ub > 0) {
        hipLaunchKernelGGL(bucket_scatter_kernel, dim3(log_grid(h)), dim3(256), 0, h->stream, h->d_log_pos[c].p, h->d_log_tup[c].p,
                           (long long)h->log_ub, m, h->d_tiles.p, tile_cnt, h->d_tuples.p);
    }
    PISCES_HIP_CHECK(h, hipGetLastError());
    return PISCES_OK;   // everything else of the handle is ordered behind this on h->stream
}

// DoneProcessing for the observation log: the entries of `keys` leave, the rest moves to the OTHER log buffer.  enqueue_drop only
// enqueues (the current log is left as it is, so a flush that has to be repeated loses nothing); commit_drop makes the other buffer
// the log once the number of entries it kept is known on the host.
static int32_t enqueue_drop(PiscesHip* h, const std::vector<int32_t>& keys)
{
    std::vector<int32_t> first_tile(keys.size(), 0), tol;
    BucketMap m;
    int32_t rc = upload_bucket_map(h, keys, first_tile, tol, &m);
    if (rc) return rc;
    const int c = h->log_cur, o = c ^ 1;
    PISCES_HIP_CHECK(h, h->d_log_pos[o].reserve((size_t)h->log_ub));
    PISCES_HIP_CHECK(h, h->d_log_tup[o].reserve((size_t)h->log_ub));
    if (!h->drop_counter_cleared) PISCES_HIP_CHECK(h, hipMemsetAsync(h->d_log_n.p + o, 0, sizeof(unsigned long long), h->stream));
    h->drop_counter_cleared = false;
    hipLaunchKernelGGL(log_drop_kernel, dim3(log_grid(h)), dim3(256), 0, h->stream, h->d_log_pos[c].p, h->d_log_tup[c].p, (long long)h->log_ub,
                       m, h->d_log_pos[o].p, h->d_log_tup[o].p, h->d_log_n.p + o);
    PISCES_HIP_CHECK(h, hipGetLastError());
    return PISCES_OK;
}
static void commit_drop(PiscesHip* h, unsigned long long kept)
{
    h->log_cur ^= 1;
    h->log_ub = (int64_t)kept;
}
static int32_t drop_blocks(PiscesHip* h, const std::vector<int32_t>& keys)
{
    if (keys.empty() || h->log_ub == 0) return PISCES_OK;
    int32_t rc = enqueue_drop(h, keys);
    if (rc) return rc;
    unsigned long long kept = 0;
    PISCES_HIP_CHECK(h, hipMemcpyAsync(&kept, h->d_log_n.p + (h->log_cur ^ 1), sizeof(kept), hipMemcpyDeviceToHost, h->stream));
    PISCES_HIP_CHECK(h, hipStreamSynchronize(h->stream));
    commit_drop(h, kept);
    return PISCES_OK;
}

// launches the fused tuples -> histogram -> call kernel on stream s
// e0 / e1 (optional): HIP events bound to the dispatch itself (hipExtLaunchKernel): their timestamps are the kernel's own
// start and end, not the arrival of separate marker packets before and after it.
static hipError_t launch_call_tiles(PiscesHip* h, hipStream_t s, const uint32_t* d_tuples, const PiscesTile* d_tiles, int32_t n_tiles,
                              const uint8_t* d_ref, int32_t ref_start, int64_t ref_len, PiscesCalledAllele* d_records,
                              PiscesTileResult* d_tr, hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr)
{
    const uint32_t lds = (uint32_t)h->lds_pad;
    if (h->cfg.noise_model == PISCES_NOISE_WINDOW) {
        // NoiseModel.Window needs the base-quality sums next to the counts, cell by cell (RegionState.cs:61): anchor-resolved counts and
        // sums go to HBM (accumulate_tiles_kernel) and the call phase reads them back (call_counts_kernel).  Not the streaming-rate
        // path; the reference's default is NoiseModel.Flat.
        if (e0) (void)hipEventRecord(e0, s);
        hipError_t er = accumulate_tiles(h, s, d_tuples, d_tiles, n_tiles, true);
        if (er != hipSuccess) return er;
        hipLaunchKernelGGL(call_counts_kernel, dim3((unsigned)n_tiles), dim3(kBlock), 0, s, h->d_counts.p, (const uint32_t*)nullptr, d_tiles, n_tiles,
                           d_ref, ref_start, ref_len, d_records, d_tr, h->P, h->d_sumq.p);
        if (e1) (void)hipEventRecord(e1, s);
        return hipSuccess;
    }
    // (the Diploid strand-bias model is compiled into call_tiles_kernel / call_counts_kernel / call_spanning_kernel only)
    if (h->kernel_variant >= 2 && h->cfg.min_base_call_quality <= 255 && h->cfg.strand_bias_model != PISCES_SB_DIPLOID) {   // the wave forms compare the quality byte in place
        // Two waves per tile shorten the call phase (Reference / q-score work and the strand-bias statistics run side by
        // side) and pay for it in registers (128 VGPRs for 8 tiles per CU).  Measured (tools/kbench.py, 500x): that wins up
        // to ~8 k tiles per launch (56 % vs 49 % at 2048 tiles, 62 % vs 60 % at 8192); beyond that tiles interleave on their
        // own and one wave per tile (no spills, 12 tiles per CU) streams better (68.5 % vs 66 % at 15 625 tiles).
        const bool two = h->kernel_variant == 3 || (h->kernel_variant == 4 && (int64_t)n_tiles <= (int64_t)h->n_cus * 32);
        if (!two)
            hipExtLaunchKernelGGL(call_tiles_wave_kernel<1>, dim3((unsigned)n_tiles), dim3(64), lds, s, e0, e1, 0u, d_tuples, d_tiles,
                                  n_tiles, d_ref, ref_start, ref_len, d_records, d_tr, h->P, (const DeviceParams*)h->d_params.p);
        else
            hipExtLaunchKernelGGL(call_tiles_wave_kernel<2>, dim3((unsigned)n_tiles), dim3(128), lds, s, e0, e1, 0u, d_tuples, d_tiles,
                                  n_tiles, d_ref, ref_start, ref_len, d_records, d_tr, h->P, (const DeviceParams*)h->d_params.p);
        return hipSuccess;
    }
    hipExtLaunchKernelGGL(call_tiles_kernel, dim3((unsigned)n_tiles), dim3(kBlock), lds, s, e0, e1, 0u, d_tuples, d_tiles, n_tiles, d_ref,
                          ref_start, ref_len, d_records, d_tr, h->P);
    return hipSuccess;
}

// scan + gather: d_out = called alleles in (position, allele) order, *d_count = how many
static void launch_compaction(hipStream_t s, const PiscesCalledAllele* d_records, const PiscesTileResult* d_tr, int32_t n_tiles,
                              int32_t* d_offsets, PiscesCalledAllele* d_out, int32_t cap, int32_t* d_count, int32_t* d_called = nullptr)
{
    hipLaunchKernelGGL(scan_tile_counts_kernel, dim3(1), dim3(1024), 0, s, d_tr, n_tiles, d_offsets, d_count, d_called);
    hipLaunchKernelGGL(gather_records_kernel, dim3((unsigned)n_tiles), dim3(64), 0, s, d_records, d_tr, n_tiles, d_offsets, d_out, cap);
}

// device work of one flush: returns called alleles of `keys` sorted by (position, ref, alt)
static int32_t call_blocks(PiscesHip* h, const std::vector<int32_t>& keys, std::vector<PiscesCalledAllele>& out, int64_t* n_called,
                           bool with_drop = false, bool* dropped = nullptr, unsigned long long* kept = nullptr)
{
    out.clear();   // (*n_called accumulates: the caller zeroes it)
    if (dropped) *dropped = false;
    if (keys.empty()) return PISCES_OK;
    if (!h->d_ref.p) return fail(h, PISCES_E_STATE, "flush: set_reference has not been called");
    std::vector<PiscesTile> tiles;
    int32_t rc = bucket_blocks(h, keys, true, tiles);
    if (rc) return rc;
    if (tiles.empty()) return PISCES_OK;
    const int32_t n_tiles = (int32_t)tiles.size();
    const size_t cap = (size_t)n_tiles * kSlotsPerTile;   // slot layout: 256 slots per tile
    PISCES_HIP_CHECK(h, h->d_records.reserve(cap));
    PISCES_HIP_CHECK(h, h->d_compact.reserve(cap));
    PISCES_HIP_CHECK(h, h->d_offsets.reserve((size_t)n_tiles));

    const bool window = h->cfg.noise_model == PISCES_NOISE_WINDOW;
    bool use_counts = false;
    for (auto& kv : h->gapped_mnv_ref)
        if (std::binary_search(keys.begin(), keys.end(), block_key(h, kv.first))) { use_counts = true; break; }

    std::vector<uint32_t> g;
    if (!use_counts && !window) {
        PISCES_HIP_CHECK(h, launch_call_tiles(h, h->stream, h->d_tuples.p, h->d_tiles.p, n_tiles, h->d_ref.p, 1, h->ref_len, h->d_records.p,
                                              h->d_tile_results.p));
    } else {
        // counts in HBM + AddGappedMnvRefCount adjustments (CoverageCalculator.cs:82-97)
        const size_t nc = (size_t)n_tiles * kTile * PISCES_COUNTS_PER_LOCUS;
        PISCES_HIP_CHECK(h, h->d_counts.reserve(nc));
        PISCES_HIP_CHECK(h, h->d_gapped.reserve((size_t)n_tiles * kTile));
        PISCES_HIP_CHECK(h, hipMemsetAsync(h->d_counts.p, 0, nc * sizeof(int32_t), h->stream));
        g.assign((size_t)n_tiles * kTile, 0u);
        for (int32_t t = 0; t < n_tiles; t++)
            for (int32_t l = 0; l < tiles[(size_t)t].n_loci; l++) {
                auto it = h->gapped_mnv_ref.find(tiles[(size_t)t].start_position + l);
                if (it != h->gapped_mnv_ref.end()) g[(size_t)t * kTile + (size_t)l] = (uint32_t)it->second;
            }
        { int32_t rcu = meta_upload(h, h->d_gapped.p, g.data(), g.size() * sizeof(uint32_t)); if (rcu) return rcu; }
        PISCES_HIP_CHECK(h, accumulate_tiles(h, h->stream, h->d_tuples.p, h->d_tiles.p, n_tiles, window));
        hipLaunchKernelGGL(call_counts_kernel, dim3((unsigned)n_tiles), dim3(kBlock), 0, h->stream, h->d_counts.p, h->d_gapped.p,
                           h->d_tiles.p, n_tiles, h->d_ref.p, 1, h->ref_len, h->d_records.p, h->d_tile_results.p, h->P,
                           window ? h->d_sumq.p : (const double*)nullptr);
    }
    // tiles were built in ascending position order: the ordered compaction is AlleleCaller.Call's (position, ref, alt) order
    launch_compaction(h->stream, h->d_records.p, h->d_tile_results.p, n_tiles, h->d_offsets.p, h->d_compact.p, (int32_t)cap, h->d_count.p,
                      h->d_count.p + 1);
    PISCES_HIP_CHECK(h, hipGetLastError());
    // one synchronisation in the usual case: the two counters and a speculative prefix of the sorted records (one per locus plus
    // a quarter) come back together into pinned memory; a second copy only when more alleles were called than that
    int64_t n_loci_total = 0;
    for (auto& t : tiles) n_loci_total += t.n_loci;
    const size_t spec = std::min<size_t>(cap, (size_t)(n_loci_total + n_loci_total / 4 + 64));
    // DoneProcessing's kernel rides in the same submission (it only writes the OTHER log buffer): one synchronisation per flush
    const bool drop_now = with_drop && h->log_ub > 0;
    if (drop_now) {
        int32_t rcd = enqueue_drop(h, keys);
        if (rcd) return rcd;
    }
    const size_t dl_bytes = 16 + cap * sizeof(PiscesCalledAllele);
    if (dl_bytes > h->h_dl_cap) {
        if (h->h_dl) (void)hipHostFree(h->h_dl);
        h->h_dl = nullptr;
        h->h_dl_cap = 0;
        PISCES_HIP_CHECK(h, hipHostMalloc((void**)&h->h_dl, dl_bytes + dl_bytes / 2, hipHostMallocDefault));
        h->h_dl_cap = dl_bytes + dl_bytes / 2;
    }
    int32_t* hdr = (int32_t*)h->h_dl;
    PiscesCalledAllele* hrec = (PiscesCalledAllele*)(h->h_dl + 16);
    PISCES_HIP_CHECK(h, hipMemcpyAsync(hdr, h->d_count.p, 2 * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
    if (drop_now)
        PISCES_HIP_CHECK(h, hipMemcpyAsync(hdr + 2, h->d_log_n.p + (h->log_cur ^ 1), sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream));
    PISCES_HIP_CHECK(h, hipMemcpyAsync(hrec, h->d_compact.p, spec * sizeof(PiscesCalledAllele), hipMemcpyDeviceToHost, h->stream));
    PISCES_HIP_CHECK(h, hipStreamSynchronize(h->stream));
    h->h_meta_used = 0;   // the stream is idle: nothing reads the arena any more
    const int32_t total = hdr[0];
    *n_called += hdr[1];
    if (drop_now) {
        if (dropped) *dropped = true;
        if (kept) std::memcpy(kept, hdr + 2, sizeof(unsigned long long));
    }
    if ((size_t)total > spec) {
        PISCES_HIP_CHECK(h, hipMemcpyAsync(hrec + spec, h->d_compact.p + spec, ((size_t)total - spec) * sizeof(PiscesCalledAllele),
                                           hipMemcpyDeviceToHost, h->stream));
        PISCES_HIP_CHECK(h, hipStreamSynchronize(h->stream));
    }
    out.assign(hrec, hrec + total);
    return PISCES_OK;
}

// ------------------------------------------------------------------------------------------------
// VariantCollapser.Collapse (exe/Pisces/Logic/VariantCalling/VariantCollapser.cs:31-79) for the host-side candidates of a batch:
// insertions / deletions, and with MNV calling on the SNV / MNV candidates too.  With it off SNV candidates need no pass here: an
// open-ended SNV collapses into its anchored twin, and the device counts are that sum already.  Frequencies come from the same
// coverage functions the device call uses, over a host copy of the anchor-resolved counts.
// ------------------------------------------------------------------------------------------------
namespace {
inline int cand_length(const HostCandidate& c)   // BaseAllele.Length
{
    return c.category == PISCES_CAT_INSERTION ? (int)c.alt.size() - 1 : c.category == PISCES_CAT_DELETION ? (int)c.ref.size() - 1 : (int)c.alt.size();
}
inline int cand_support(const HostCandidate& c) { return c.support_by_dir[0] + c.support_by_dir[1] + c.support_by_dir[2]; }
inline bool cand_fully_anchored(const HostCandidate& c) { return !c.open_left && !c.open_right; }
inline bool cand_equals(const HostCandidate& a, const HostCandidate& b)
{
    return a.position == b.position && a.alt == b.alt && a.category == b.category && a.ref == b.ref;
}
// CanCollapse :119-174
bool can_collapse(const HostCandidate& t, const HostCandidate& p)
{
    const bool ti = t.category == PISCES_CAT_INSERTION, pi = p.category == PISCES_CAT_INSERTION;
    const bool td = t.category == PISCES_CAT_DELETION, pd = p.category == PISCES_CAT_DELETION;
    if (ti != pi || td != pd || cand_length(t) > cand_length(p) || (cand_fully_anchored(t) && !cand_fully_anchored(p))) return false;
    const std::string& tb = td ? t.ref : t.alt;
    const std::string& pb = pd ? p.ref : p.alt;
    if (cand_fully_anchored(t) && cand_fully_anchored(p)) return cand_equals(t, p);
    if (td) {
        if (t.open_right) return p.position + 1 == t.position + 1;
        return p.position + (int)pb.size() - 1 == t.position + (int)tb.size() - 1;
    }
    if (t.open_right) return p.position == t.position && pb.size() >= tb.size() && pb.compare(0, tb.size(), tb) == 0;
    if (ti) {
        if (p.position + 1 != t.position + 1) return false;
        if (pb.size() + 1 < tb.size()) return false;
        return pb.compare(pb.size() - tb.size() + 1, std::string::npos, tb, 1, std::string::npos) == 0;
    }
    // SNV / MNV anchored on the right: same last position, the bases are a suffix
    return p.position + (int)p.alt.size() - 1 == t.position + (int)t.alt.size() - 1 && p.alt.size() >= t.alt.size() &&
           p.alt.compare(p.alt.size() - t.alt.size(), std::string::npos, t.alt) == 0;
}

// ---- MnvReallocator (exe/Pisces/Logic/VariantCalling/MnvReallocator.cs:12-261) over heap HostCandidate objects; AlleleSupport is the
// sum of support_by_dir throughout (AlleleHelper.Map and every CreateVariant on this path keep the two in step) ----
using CandPtr = HostCandidate*;
struct MnvArena {   // owns every object the reallocation creates
    std::vector<std::unique_ptr<HostCandidate>> objs;
    CandPtr make(int32_t position, const std::string& alt, const std::string& ref, const int32_t* dirs)   // CreateVariant :151-168
    {
        objs.emplace_back(new HostCandidate());
        CandPtr v = objs.back().get();
        bool same = alt.size() == ref.size();
        for (size_t i = 0; same && i < alt.size(); i++) same = std::toupper((unsigned char)alt[i]) == std::toupper((unsigned char)ref[i]);
        v->category = same ? PISCES_CAT_REFERENCE : (alt.size() > 1 ? PISCES_CAT_MNV : PISCES_CAT_SNV);
        v->position = position;
        v->alt = alt;
        v->ref = ref;
        if (dirs) for (int d = 0; d < 3; d++) v->support_by_dir[d] = dirs[d];
        return v;
    }
};
inline void list_remove(std::vector<CandPtr>& l, CandPtr v)
{
    auto it = std::find(l.begin(), l.end(), v);
    if (it != l.end()) l.erase(it);
}
inline bool overlap_before(CandPtr a, CandPtr b)   // OrderByDescending(alt.Length).ThenByDescending(AlleleSupport).ThenBy(alt).ThenBy(ref)
{
    if (a->alt.size() != b->alt.size()) return a->alt.size() > b->alt.size();
    if (cand_support(*a) != cand_support(*b)) return cand_support(*a) > cand_support(*b);
    if (a->alt != b->alt) return a->alt < b->alt;
    return a->ref < b->ref;
}
CandPtr mnv_break_off_edge_references(MnvArena& arena, CandPtr allele)   // :212-241
{
    if (allele->category != PISCES_CAT_MNV) return allele;
    const int n = (int)allele->ref.size();
    int leftAdjust = 0, rightAdjust = 0;
    for (int i = 0; i < n; i++) { if (allele->ref[(size_t)i] != allele->alt[(size_t)i]) break; leftAdjust++; }
    for (int i = 0; i < n; i++) { const int k = n - 1 - i; if (allele->ref[(size_t)k] != allele->alt[(size_t)k]) break; rightAdjust++; }
    return arena.make(allele->position + leftAdjust, allele->alt.substr((size_t)leftAdjust, allele->alt.size() - (size_t)(leftAdjust + rightAdjust)),
                      allele->ref.substr((size_t)leftAdjust, allele->ref.size() - (size_t)(leftAdjust + rightAdjust)), allele->support_by_dir);
}
void mnv_process_overlap(MnvArena& arena, bool hasMax, int32_t blockMaxPos, CandPtr overlap, CandPtr toReassign, std::vector<CandPtr>& remainderAlleles,
                         std::vector<CandPtr>& outsideThisBlock)   // :97-133
{
    for (int d = 0; d < 3; d++) overlap->support_by_dir[d] += toReassign->support_by_dir[d];
    list_remove(remainderAlleles, toReassign);
    // CreateAllelesFromRemainder :170-210
    std::vector<CandPtr> remainders;
    const int overlapIndexInFailedMnv = overlap->position - toReassign->position;
    const int rightSideOverlap = overlapIndexInFailedMnv + (int)overlap->alt.size();
    const int altLen = (int)toReassign->alt.size();
    if (altLen - rightSideOverlap > 0 && rightSideOverlap <= toReassign->position + altLen) {
        CandPtr r = arena.make(toReassign->position + rightSideOverlap, toReassign->alt.substr((size_t)rightSideOverlap),
                               toReassign->ref.substr((size_t)rightSideOverlap, (size_t)(altLen - rightSideOverlap)), toReassign->support_by_dir);
        if (r->category != PISCES_CAT_REFERENCE) remainders.push_back(r);
    }
    if (overlapIndexInFailedMnv > 0) {
        CandPtr l = arena.make(toReassign->position, toReassign->alt.substr(0, (size_t)overlapIndexInFailedMnv),
                               toReassign->ref.substr(0, (size_t)overlapIndexInFailedMnv), toReassign->support_by_dir);
        if (l->category != PISCES_CAT_REFERENCE) remainders.push_back(l);
    }
    for (auto& r : remainders) r = mnv_break_off_edge_references(arena, r);
    if (hasMax) {
        if (overlap->position > blockMaxPos) { list_remove(remainderAlleles, overlap); outsideThisBlock.push_back(overlap); }
        for (CandPtr r : remainders) (r->position <= blockMaxPos ? remainderAlleles : outsideThisBlock).push_back(r);
    } else {
        for (CandPtr r : remainders) remainderAlleles.push_back(r);
    }
}
// ReallocateFailedMnvs :12-95
void mnv_reallocate_failed(MnvArena& arena, const std::vector<CandPtr>& failed, std::vector<CandPtr>& callable, bool hasMax, int32_t blockMaxPos,
                           std::vector<CandPtr>& outsideThisBlock)
{
    std::vector<CandPtr> ordered(failed);
    std::stable_sort(ordered.begin(), ordered.end(), [](CandPtr a, CandPtr b) {
        if (a->position != b->position) return a->position < b->position;
        return overlap_before(a, b);
    });
    for (CandPtr failedMnv : ordered) {
        std::vector<CandPtr> remainderAlleles{failedMnv};
        while (!remainderAlleles.empty()) {
            CandPtr alleleToReassign = remainderAlleles.front();
            const int fl = (int)alleleToReassign->alt.size();
            std::vector<CandPtr> overlaps;
            for (CandPtr c : callable) {   // IsPotentialOverlap :250-261
                const int cl = (int)c->alt.size();
                if (c->position >= alleleToReassign->position && c->position <= alleleToReassign->position + fl && cl <= fl &&
                    c->position + cl <= alleleToReassign->position + fl &&
                    (c->category == PISCES_CAT_MNV || c->category == PISCES_CAT_SNV || c->category == PISCES_CAT_REFERENCE))
                    overlaps.push_back(c);
            }
            std::stable_sort(overlaps.begin(), overlaps.end(), overlap_before);
            CandPtr firstMatch = nullptr;
            bool anyLongMatch = false;
            for (CandPtr o : overlaps)   // OverlapMatches :243-248
                if (alleleToReassign->alt.compare((size_t)(o->position - alleleToReassign->position), o->alt.size(), o->alt) == 0) {
                    if (!firstMatch) firstMatch = o;
                    if (o->alt.size() > 1) anyLongMatch = true;
                }
            bool reallocated = false;
            if (hasMax) {
                const int distanceIntoNextBlock = alleleToReassign->position + (fl - 1) - blockMaxPos;
                if (distanceIntoNextBlock > 0 && !anyLongMatch) {
                    if (alleleToReassign->position <= blockMaxPos) {   // peel off into the next block
                        const int originalAlleleLength = (int)alleleToReassign->ref.size();
                        CandPtr next = arena.make(blockMaxPos + 1, alleleToReassign->alt.substr((size_t)(originalAlleleLength - distanceIntoNextBlock), (size_t)distanceIntoNextBlock),
                                                  alleleToReassign->ref.substr((size_t)(originalAlleleLength - distanceIntoNextBlock), (size_t)distanceIntoNextBlock), nullptr);
                        next = mnv_break_off_edge_references(arena, next);
                        mnv_process_overlap(arena, hasMax, blockMaxPos, next, alleleToReassign, remainderAlleles, outsideThisBlock);
                    } else {
                        list_remove(remainderAlleles, alleleToReassign);
                        outsideThisBlock.push_back(alleleToReassign);
                    }
                    reallocated = true;
                }
            }
            if (!reallocated && firstMatch) {
                mnv_process_overlap(arena, hasMax, blockMaxPos, firstMatch, alleleToReassign, remainderAlleles, outsideThisBlock);
                reallocated = true;
            }
            if (!reallocated) {   // BreakDownToSingleNucCalls :135-149
                for (int i = 0; i < fl; i++) {
                    CandPtr sn = arena.make(alleleToReassign->position + i, alleleToReassign->alt.substr((size_t)i, 1), alleleToReassign->ref.substr((size_t)i, 1),
                                            alleleToReassign->support_by_dir);
                    if (sn->category == PISCES_CAT_REFERENCE) continue;
                    if (hasMax && sn->position > blockMaxPos) outsideThisBlock.push_back(sn);
                    else callable.push_back(sn);
                }
                list_remove(remainderAlleles, alleleToReassign);
            }
        }
    }
}
}  // namespace

// cands is edited in place; freq(c) = CalledAllele.Frequency of the candidate against the current counts
extern "C++" {
template <typename FreqFn>
static int64_t collapse_candidates(std::vector<HostCandidate>& cands, float freq_threshold, float freq_ratio_threshold, FreqFn freq)
{
    const size_t n = cands.size();
    std::vector<uint8_t> removed(n, 0);
    std::vector<size_t> order;
    for (size_t i = 0; i < n; i++)
        if (cands[i].open_left || cands[i].open_right) order.push_back(i);
    // OrderByDescending(Length).ThenByDescending(both open).ThenByDescending(either).ThenBy(ref).ThenBy(alt).ThenBy(Support)
    // .ThenBy(OpenOnRight).ThenBy(OpenOnLeft) :41-46
    std::stable_sort(order.begin(), order.end(), [&](size_t x, size_t y) {
        const HostCandidate& a = cands[x];
        const HostCandidate& b = cands[y];
        if (cand_length(a) != cand_length(b)) return cand_length(a) > cand_length(b);
        const bool ba = a.open_left && a.open_right, bb = b.open_left && b.open_right;
        if (ba != bb) return ba;
        if (a.ref != b.ref) return a.ref < b.ref;
        if (a.alt != b.alt) return a.alt < b.alt;
        if (cand_support(a) != cand_support(b)) return cand_support(a) < cand_support(b);
        if (a.open_right != b.open_right) return !a.open_right;
        if (a.open_left != b.open_left) return !a.open_left;
        return false;
    });
    int64_t collapsed = 0;
    struct Row { size_t idx; float f; };
    std::vector<Row> rows;
    for (size_t oi : order) {
        HostCandidate& t = cands[oi];
        rows.clear();
        for (size_t j = 0; j < n; j++)
            if (j != oi && !removed[j] && can_collapse(t, cands[j])) rows.push_back({j, freq(cands[j])});
        if (rows.empty()) continue;
        const float tf = freq(t);
        // IComparer.Compare :214-244 (no known variants here); input order breaks the remaining ties
        std::stable_sort(rows.begin(), rows.end(), [&](const Row& x, const Row& y) {
            const HostCandidate& a = cands[x.idx];
            const HostCandidate& b = cands[y.idx];
            if (cand_fully_anchored(a) != cand_fully_anchored(b)) return cand_fully_anchored(a);
            if (cand_length(a) != cand_length(b)) return cand_length(a) > cand_length(b);
            if (std::fabs(x.f - y.f) > 0.0f) return x.f > y.f;
            if (a.position != b.position) return a.position < b.position;
            return a.alt < b.alt;
        });
        const Row* pick = nullptr;
        for (auto& r : rows)
            if (cand_equals(cands[r.idx], t) && cand_fully_anchored(cands[r.idx])) { pick = &r; break; }
        if (!pick)
            for (auto& r : rows)
                if (r.f >= freq_threshold && r.f / tf > freq_ratio_threshold) { pick = &r; break; }
        if (!pick) continue;
        HostCandidate& m = cands[pick->idx];
        collapsed++;
        for (int d = 0; d < 3; d++) {   // Collapse :81-90
            m.support_by_dir[d] += t.support_by_dir[d];
            m.well_anchored_by_dir[d] += t.well_anchored_by_dir[d];
        }
        m.open_left = m.open_left && t.open_left;
        m.open_right = m.open_right && t.open_right;
        removed[oi] = 1;
    }
    size_t w = 0;
    for (size_t i = 0; i < n; i++)
        if (!removed[i]) { if (w != i) cands[w] = std::move(cands[i]); w++; }
    cands.resize(w);
    return collapsed;
}
}  // extern "C++"

// IAlleleCaller.Call for the host-found candidates of `keys` (AlleleCaller.CallForPositions :60-141): anchor-resolved counts of every
// block a candidate touches -> collapser -> call_spanning_kernel -> callable candidates with their records.  With MNV calling on
// the candidates include the SNVs / MNVs of the read walk: MNV candidates are processed first, the ones that are not callable go
// through MnvReallocator on the host, leftovers past the last cleared block return to the state as candidates of the next block,
// reference support taken by gapped MNVs is registered (it reaches the Reference records through call_blocks, which runs after
// this), and every callable allele is processed again.  ref_overrides: Reference alleles that reallocation added support to
// (they replace the tile kernels' Reference record of that position).
static int32_t call_spanning(PiscesHip* h, const std::vector<int32_t>& keys, int32_t up_to_position, std::vector<PiscesCalledAllele>& recs,
                             std::vector<HostCandidate>& called, int64_t* n_called, int64_t* n_collapsed,
                             std::vector<PiscesCalledAllele>& ref_overrides)
{
    recs.clear();
    called.clear();
    ref_overrides.clear();
    *n_collapsed = 0;
    const bool mnv_mode = h->cfg.call_mnvs != 0;
    const bool window = h->cfg.noise_model == PISCES_NOISE_WINDOW;
    std::vector<HostCandidate> work;   // a copy: the blocks keep their candidates until DoneProcessing
    for (int32_t key : keys) {
        // RegionState.GetAllCandidates walks _candidateVariantsLookup by position, each position in arrival order (RegionState.cs:388-391)
        const size_t first = work.size();
        for (auto& c : h->blocks[key].cands) work.push_back(c);
        std::stable_sort(work.begin() + (std::ptrdiff_t)first, work.end(), [](const HostCandidate& x, const HostCandidate& y) { return x.position < y.position; });
    }
    const int bs = h->cfg.block_size;
    // AddCollapsableFromOtherBlocks (RegionStateManager.cs:321-324, 441-457): when an allele of the cleared blocks reaches past the last
    // cleared position and the collapser is on, the SNV / MNV candidates of the held blocks up to upTo that end at or before upTo and are
    // not open on the right (RegionState.ExtractCollapsable :470-490) leave their blocks and join this batch, where candidates of the
    // cleared blocks may collapse into them; whatever of them is left after collapsing goes back to the state (below)
    int32_t max_cleared = -1;
    if (!keys.empty() && up_to_position >= 0 && h->cfg.collapse) {
        int32_t max_endpoint = 0;
        for (int32_t key : keys) max_endpoint = std::max(max_endpoint, h->blocks[key].max_allele_endpoint);
        if (max_endpoint > keys.back() * bs) {
            max_cleared = keys.back() * bs;
            for (auto& kv : h->blocks) {   // ascending block order
                const int32_t start = (kv.first - 1) * bs + 1;
                if (start <= max_cleared || start > up_to_position) continue;
                std::vector<HostCandidate> kept;
                const size_t first = work.size();
                for (auto& c : kv.second.cands) {
                    const bool collapsable = (c.category == PISCES_CAT_MNV || c.category == PISCES_CAT_SNV) && !c.open_right &&
                                             c.position + (int32_t)c.alt.size() - 1 <= up_to_position;
                    (collapsable ? work : kept).push_back(c);
                }
                if (work.size() == first) continue;
                std::stable_sort(work.begin() + (std::ptrdiff_t)first, work.end(), [](const HostCandidate& x, const HostCandidate& y) { return x.position < y.position; });
                kv.second.cands.clear();   // (MaxAlleleEndpoint keeps its value: RegionState never lowers it)
                kv.second.cand_index.clear();
                for (auto& c : kept) add_candidate(h, c);
            }
        }
    }
    if (work.empty()) return PISCES_OK;
    // start / end points (CoverageCalculator.Compute :27-41)
    auto endpoints = [](const HostCandidate& c, int32_t& sp, int32_t& ep) {
        if (c.category == PISCES_CAT_DELETION) { sp = c.position + 1; ep = c.position + (int32_t)c.ref.size() - 1; }
        else if (c.category == PISCES_CAT_MNV) { sp = c.position; ep = c.position + (int32_t)c.alt.size() - 1; }
        else if (c.category == PISCES_CAT_INSERTION) { sp = c.position; ep = c.position + 1; }
        else { sp = c.position; ep = c.position; }
    };
    std::vector<int32_t> bkeys;
    for (auto& c : work) {
        int32_t sp, ep;
        endpoints(c, sp, ep);
        for (int32_t p : {sp, ep}) {
            const int32_t k = block_key(h, p);
            if (p > 0 && h->blocks.count(k)) bkeys.push_back(k);
        }
    }
    std::sort(bkeys.begin(), bkeys.end());
    bkeys.erase(std::unique(bkeys.begin(), bkeys.end()), bkeys.end());
    // counts over the whole block grid of those blocks (not the interval-clipped tiles)
    std::vector<PiscesTile> tiles;
    if (!bkeys.empty()) {
        int32_t rcb = bucket_blocks(h, bkeys, false, tiles);
        if (rcb) return rcb;
    }
    const int32_t n_tiles = (int32_t)tiles.size();
    const int tiles_per_block = (bs + kTile - 1) / kTile;
    auto locus_index = [&](int32_t p) -> int64_t {
        if (p <= 0) return -1;
        const int32_t k = block_key(h, p);
        auto it = std::lower_bound(bkeys.begin(), bkeys.end(), k);
        if (it == bkeys.end() || *it != k) return -1;
        const int64_t bi = it - bkeys.begin();
        const int32_t off = p - ((k - 1) * bs + 1);
        return (bi * tiles_per_block + off / kTile) * kTile + off % kTile;
    };
    if (n_tiles > 0) {
        PISCES_HIP_CHECK(h, accumulate_tiles(h, h->stream, h->d_tuples.p, h->d_tiles.p, n_tiles, window));
    } else {
        PISCES_HIP_CHECK(h, h->d_counts.reserve(PISCES_COUNTS_PER_LOCUS));
        if (window) PISCES_HIP_CHECK(h, h->d_sumq.reserve(PISCES_COUNTS_PER_LOCUS));
    }
    auto atype = [](char ch) { return ch == 'A' ? 0 : ch == 'G' ? 1 : ch == 'C' ? 2 : ch == 'T' ? 3 : 4; };
    auto gapped_at = [&](int32_t p) {
        auto it = h->gapped_mnv_ref.find(p);
        return it == h->gapped_mnv_ref.end() ? 0 : it->second;
    };
    auto to_dev = [&](const HostCandidate& c, DevCandidate& d) {
        std::memset(&d, 0, sizeof(d));
        d.position = c.position;
        d.category = c.category;
        d.ref_len = (int32_t)c.ref.size();
        d.alt_len = (int32_t)c.alt.size();
        for (int k = 0; k < 3; k++) { d.sup[k] = c.support_by_dir[k]; d.anch[k] = c.well_anchored_by_dir[k]; }
        d.first_base = d.last_base = PISCES_ALLELE_N;
        if (c.category == PISCES_CAT_INSERTION && c.alt.size() >= 2) {
            d.first_base = atype(c.alt[1]);
            d.last_base = atype(c.alt[c.alt.size() - 1]);
        }
        int32_t sp, ep;
        endpoints(c, sp, ep);
        d.start_idx = locus_index(sp);
        d.end_idx = locus_index(ep);
        d.gapped = (c.category == PISCES_CAT_SNV || c.category == PISCES_CAT_REFERENCE) ? gapped_at(c.position) : 0;
    };
    // the collapser's frequencies and the reallocator's Reference candidates read a host copy of the anchor-resolved counts
    std::vector<int32_t> host_counts;
    const bool have_forced = !h->forced.empty();
    if (h->cfg.collapse || mnv_mode || have_forced) {
        host_counts.assign((size_t)std::max(n_tiles, 1) * kTile * PISCES_COUNTS_PER_LOCUS, 0);
        if (n_tiles > 0) {
            PISCES_HIP_CHECK(h, hipMemcpyAsync(host_counts.data(), h->d_counts.p, host_counts.size() * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
            PISCES_HIP_CHECK(h, hipStreamSynchronize(h->stream));
        }
    }
    if (!mnv_mode && have_forced) {
        // MNV calling off: SNV candidates are the allele counts and never reach the host, so a forced SNV (added without support) takes
        // the support the merged candidate of the reference has: the reads that show the base at or above the quality threshold
        for (auto& c : work) {
            if (c.category != PISCES_CAT_SNV || cand_support(c) != 0 || !is_forced_allele(h, c)) continue;
            const int64_t li = locus_index(c.position);
            const int at = atype(c.alt[0]);
            if (li < 0 || at >= 4) continue;
            for (int d = 0; d < 3; d++) {
                const int32_t* row = host_counts.data() + li * PISCES_COUNTS_PER_LOCUS + (at * 3 + d) * PISCES_NUM_ANCHORS;
                for (int an = 0; an < PISCES_NUM_ANCHORS; an++) c.support_by_dir[d] += row[an];
            }
        }
    }
    if (h->cfg.collapse) {
        const int32_t stitched = h->cfg.expect_stitched_reads;
        *n_collapsed = collapse_candidates(work, h->cfg.collapse_freq_threshold, h->cfg.collapse_freq_ratio_threshold, [&](const HostCandidate& c) {
            DevCandidate d;
            to_dev(c, d);
            const int total = candidate_total_coverage(d, host_counts.data(), stitched);
            const int support = c.support_by_dir[0] + c.support_by_dir[1] + c.support_by_dir[2];
            if (total == 0) return 0.0f;                       // CalledAllele.Frequency (CalledAllele.cs:49-52)
            const float f = (float)support / (float)total;
            return f < 1.0f ? f : 1.0f;
        });
        // candidates past the last cleared position that could not be collapsed return to the state (VariantCollapser.cs:67-75): only the
        // ones AddCollapsableFromOtherBlocks brought in can lie there
        if (max_cleared >= 0) {
            size_t w = 0;
            for (size_t i = 0; i < work.size(); i++) {
                if (work[i].position > max_cleared && work[i].category != PISCES_CAT_REFERENCE) { add_candidate(h, work[i]); continue; }
                if (w != i) work[w] = std::move(work[i]);
                w++;
            }
            work.resize(w);
            if (work.empty()) return PISCES_OK;
        }
    }
    // one device pass over a list of candidates: records + IsCallable
    std::vector<PiscesCalledAllele> raw;
    std::vector<uint8_t> callable;
    bool second_pass = false;   // MNV mode: the pass over every callable allele, after the MNV-only pass
    auto device_pass = [&](const std::vector<const HostCandidate*>& list) -> int32_t {
        std::vector<DevCandidate> dc(list.size());
        std::vector<uint8_t> pool;
        for (size_t i = 0; i < list.size(); i++) {
            to_dev(*list[i], dc[i]);
            dc[i].reprocessed = (second_pass && list[i]->category == PISCES_CAT_MNV && !work.empty() && list[i] >= work.data() &&
                                 list[i] < work.data() + work.size()) ? 1 : 0;
            dc[i].allele_off = (int32_t)pool.size();
            pool.insert(pool.end(), list[i]->ref.begin(), list[i]->ref.end());
            pool.insert(pool.end(), list[i]->alt.begin(), list[i]->alt.end());
        }
        raw.assign(dc.size(), PiscesCalledAllele{});
        callable.assign(dc.size(), 0);
        if (dc.empty()) return PISCES_OK;
        const int32_t n = (int32_t)dc.size();
        PISCES_HIP_CHECK(h, h->d_cands.reserve(dc.size()));
        PISCES_HIP_CHECK(h, h->d_alleles.reserve(pool.size() + 16));
        PISCES_HIP_CHECK(h, h->d_cand_records.reserve(dc.size()));
        PISCES_HIP_CHECK(h, h->d_cand_callable.reserve(dc.size()));
        PISCES_HIP_CHECK(h, hipMemcpyAsync(h->d_cands.p, dc.data(), dc.size() * sizeof(DevCandidate), hipMemcpyHostToDevice, h->stream));
        PISCES_HIP_CHECK(h, hipMemcpyAsync(h->d_alleles.p, pool.data(), pool.size(), hipMemcpyHostToDevice, h->stream));
        hipLaunchKernelGGL(call_spanning_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, h->stream, h->d_cands.p, n, h->d_counts.p,
                           h->d_alleles.p, h->d_ref.p, h->ref_len, h->cfg.expect_stitched_reads, h->d_cand_records.p, h->d_cand_callable.p, h->P,
                           window ? h->d_sumq.p : (const double*)nullptr);
        PISCES_HIP_CHECK(h, hipGetLastError());
        PISCES_HIP_CHECK(h, hipMemcpyAsync(raw.data(), h->d_cand_records.p, raw.size() * sizeof(PiscesCalledAllele), hipMemcpyDeviceToHost, h->stream));
        PISCES_HIP_CHECK(h, hipMemcpyAsync(callable.data(), h->d_cand_callable.p, callable.size(), hipMemcpyDeviceToHost, h->stream));
        PISCES_HIP_CHECK(h, hipStreamSynchronize(h->stream));
        return PISCES_OK;
    };
    auto inside_intervals = [&](int32_t position) {   // ShouldReport (AlleleCaller.cs:260-263)
        if (h->intervals.empty()) return true;
        for (auto& iv : h->intervals)
            if (position >= iv.first && position <= iv.second) return true;
        return false;
    };

    std::vector<const HostCandidate*> final_list;
    MnvArena arena;
    std::vector<CandPtr> callable_alleles;          // AlleleCaller's callableAlleles (non-Reference ones and touched Reference ones)
    std::map<int32_t, CandPtr> touched_refs;         // Reference candidates created for the reallocator, by position
    if (!mnv_mode) {
        for (auto& c : work) final_list.push_back(&c);
    } else {
        // ---- MNV candidates first (AlleleCaller.cs:69-89)
        std::vector<const HostCandidate*> mnvs;
        for (auto& c : work)
            if (c.category == PISCES_CAT_MNV) mnvs.push_back(&c);
        int32_t rc1 = device_pass(mnvs);
        if (rc1) return rc1;
        std::vector<CandPtr> failed;
        {
            size_t mi = 0;
            for (auto& c : work) {
                if (c.category == PISCES_CAT_MNV) {
                    if (callable[mi]) { callable_alleles.push_back(&c); (*n_called)++; }   // IsCallable counts every pass (_totalNumCalled)
                    else failed.push_back(&c);
                    mi++;
                } else {
                    callable_alleles.push_back(&c);
                }
            }
        }
        if (!failed.empty()) {
            // Reference candidates of this batch that a failed MNV can reach: only those whose base equals the MNV's base there can
            // match (OverlapMatches); their AlleleSupport (the reference base's counts) decides the order among one-base overlaps
            const int32_t last_cleared = keys.back() * bs;
            auto ref_candidate_exists = [&](int32_t p, int32_t (&sup)[3]) {
                // (not a gVCF: Reference candidates exist at the positions of the forced alleles only, RegionState.cs:393-396)
                const bool forced_here = !h->cfg.include_reference_calls && h->forced_positions.count(p) != 0;
                if (!(h->cfg.include_reference_calls || forced_here) || p < 1 || p > h->ref_len || !inside_intervals(p)) return false;
                if (!std::binary_search(keys.begin(), keys.end(), block_key(h, p))) return false;
                const int64_t li = locus_index(p);
                const int rb = atype((char)h->h_ref[(size_t)p - 1]);
                int total = 0;
                sup[0] = sup[1] = sup[2] = 0;
                if (li >= 0)
                    for (int at = 0; at < PISCES_NUM_ALLELE_TYPES; at++)
                        for (int d = 0; d < 3; d++) {
                            int cnt = 0;
                            const int32_t* row = host_counts.data() + li * PISCES_COUNTS_PER_LOCUS + (at * 3 + d) * PISCES_NUM_ANCHORS;
                            for (int an = 0; an < PISCES_NUM_ANCHORS; an++) cnt += row[an];
                            if (at == rb) sup[d] = cnt;
                            total += cnt;
                        }
                return h->cfg.emit_zero_coverage_refs != 0 || forced_here || total > 0;   // RegionState.cs:446
            };
            for (CandPtr f : failed)
                for (size_t k = 0; k < f->alt.size(); k++) {
                    const int32_t p = f->position + (int32_t)k;
                    if (f->alt[k] != f->ref[k] || touched_refs.count(p)) continue;
                    int32_t sup[3];
                    if (!ref_candidate_exists(p, sup)) continue;
                    CandPtr rc = arena.make(p, std::string(1, f->ref[k]), std::string(1, f->ref[k]), sup);
                    touched_refs[p] = rc;
                }
            // GetAllCandidates appends the Reference candidates after the variant candidates of a block: the order only matters for
            // ties between alleles of equal length, support and bases, which Reference candidates (base == reference) cannot have with
            // a variant; among themselves they are in position order
            std::vector<CandPtr> ref_originals;
            std::vector<std::array<int32_t, 3>> ref_before;
            for (auto& kv : touched_refs) {
                callable_alleles.push_back(kv.second);
                ref_originals.push_back(kv.second);
                ref_before.push_back({kv.second->support_by_dir[0], kv.second->support_by_dir[1], kv.second->support_by_dir[2]});
            }
            std::vector<CandPtr> outside;
            mnv_reallocate_failed(arena, failed, callable_alleles, true, last_cleared, outside);
            for (CandPtr o : outside)   // source.AddCandidates(leftovers.Select(AlleleHelper.Map)) :92-93
                if (o->category != PISCES_CAT_REFERENCE && o->position > 0) {
                    HostCandidate c = *o;
                    c.well_anchored_by_dir[0] = c.well_anchored_by_dir[1] = c.well_anchored_by_dir[2] = 0;
                    c.open_left = c.open_right = false;
                    add_candidate(h, c);
                }
            // Reference candidates keep only what reallocation added: the kernel supplies their own counts
            for (size_t i = 0; i < ref_originals.size(); i++)
                for (int d = 0; d < 3; d++) ref_originals[i]->support_by_dir[d] -= ref_before[i][(size_t)d];
        }
        // GetRefSupportFromGappedMnvs :180-203 -> IAlleleSource.AddGappedMnvRefCount
        for (CandPtr a : callable_alleles) {
            if (a->category != PISCES_CAT_MNV) continue;
            const int support = cand_support(*a);
            for (size_t k = 0; k < a->ref.size() && k < a->alt.size(); k++)
                if (a->ref[k] == a->alt[k]) h->gapped_mnv_ref[a->position + (int32_t)k] += support;
        }
        // a failed MNV that is a forced allele is reported all the same (AlleleCaller.cs:98-107)
        if (have_forced)
            for (CandPtr f : failed)
                if (is_forced_allele(h, *f)) callable_alleles.push_back(f);
        for (CandPtr a : callable_alleles) {
            if (a->category == PISCES_CAT_REFERENCE && cand_support(*a) == 0) continue;   // untouched: the tile kernels' record stands
            final_list.push_back(a);
        }
    }
    // not a gVCF, forced alleles given: Reference candidates at the forced positions of the cleared blocks, with or without coverage
    // (RegionState.GetAllCandidates :393-450 with CreateIntervalsFromAllels); the candidate kernel makes their records from the counts
    if (have_forced && !h->cfg.include_reference_calls) {
        static const int32_t kNone[3] = {0, 0, 0};
        for (int32_t p : h->forced_positions) {
            if (p < 1 || p > h->ref_len || !inside_intervals(p) || !std::binary_search(keys.begin(), keys.end(), block_key(h, p))) continue;
            if (touched_refs.count(p)) {
                if (cand_support(*touched_refs[p]) != 0) continue;   // in the list already, with what reallocation added
            } else {
                touched_refs[p] = arena.make(p, std::string(1, (char)h->h_ref[(size_t)p - 1]), std::string(1, (char)h->h_ref[(size_t)p - 1]), kNone);
            }
            final_list.push_back(touched_refs[p]);
        }
    }

    second_pass = mnv_mode;
    int32_t rc2 = device_pass(final_list);
    if (rc2) return rc2;
    for (size_t i = 0; i < final_list.size(); i++) {
        if (final_list[i]->category == PISCES_CAT_REFERENCE) {   // counted as called by the tile kernels already (gVCF)
            ref_overrides.push_back(raw[i]);
            continue;
        }
        // AlleleCaller.cs:109-131: a forced allele is reported whether it is callable or not; IsCallable runs once in the test for
        // IsForcedToReport and once in the test for reporting, and counts a callable forced allele twice in TotalNumCalled
        const bool forced = have_forced && is_forced_allele(h, *final_list[i]);
        const bool reportable = callable[i] && inside_intervals(final_list[i]->position);
        if (callable[i]) (*n_called) += forced ? 2 : 1;
        if (forced && !mnv_mode && final_list[i]->category == PISCES_CAT_SNV && reportable) {   // MNV calling off: the tile kernels report it,
            (*n_called)--;                                                                        // and have counted it once
            continue;
        }
        if (!reportable && !forced) continue;
        PiscesCalledAllele r = raw[i];
        if (forced && !reportable) {
            // IsForcedToReport: the ForcedReport filter, and no genotyper sees the allele (:150): the genotype of a new CalledAllele
            // (CalledAllele.cs:151) and genotype q-score 0, against which AlleleCaller's LowGQ filter is taken (:166-170)
            uint32_t fb = (r.filter_bits | (1u << PISCES_FILTER_FORCED_REPORT)) & ~(1u << PISCES_FILTER_LOW_GENOTYPE_QUALITY);
            if (h->cfg.low_gq_filter >= 0 && 0.0f < (float)h->cfg.low_gq_filter) fb |= 1u << PISCES_FILTER_LOW_GENOTYPE_QUALITY;
            r.filter_bits = (uint16_t)fb;
            r.info = (uint16_t)((r.info & ~0xFu) | (uint32_t)PISCES_GT_HET_ALT_REF);
            r.genotype_qscore = 0;
        }
        recs.push_back(r);
        called.push_back(*final_list[i]);
    }
    return PISCES_OK;
}

int32_t pisces_hip_flush_ex(PiscesHip* h, int32_t up_to_position, PiscesCalledAllele* out, int64_t capacity, int64_t* n_out,
                            int32_t* cand_index_out, PiscesCandidate* cand_out, int64_t cand_capacity, int64_t* n_cand,
                            uint8_t* alleles_out, int64_t allele_capacity, int64_t* allele_bytes)
{
    return abi_guard<int32_t>(h, [&]() -> int32_t {
    if (!h) return PISCES_E_INVALID_ARG;
    if (!n_out || capacity < 0 || (capacity > 0 && !out)) return fail(h, PISCES_E_INVALID_ARG, "flush: null output");
    PISCES_HIP_CHECK(h, hipSetDevice(h->device));
    *n_out = 0;
    if (n_cand) *n_cand = 0;
    if (allele_bytes) *allele_bytes = 0;
    { int32_t rcf = consume_found(h); if (rcf) return rcf; }
    const bool final_flush = up_to_position < 0;
    const bool replay = h->pending_valid && h->pending_up_to == up_to_position;
    if (!replay) {
        { int32_t rcp = refuse_while_batch_is_open(h, "flush (another upToPosition)"); if (rcp) return rcp; }
        // GetCandidatesToProcess (RegionStateManager.cs:283-334): only build a batch when upTo has moved
        // onto another block; take blocks that lie wholly at or below upTo.
        add_forced_as_candidates(h, final_flush ? -1 : up_to_position);   // SmallVariantCaller.cs:101-108: before Call(upTo)
        if (!final_flush && block_key(h, up_to_position) == h->last_up_to_block_key) return PISCES_OK;
        std::vector<int32_t> keys;
        for (auto& kv : h->blocks) {   // std::map: ascending keys
            if (!(final_flush || (int64_t)kv.first * h->cfg.block_size <= up_to_position)) continue;
            // a block whose spanning alleles reach past upTo is held, and so is everything after it (:304-308)
            if (!final_flush && kv.second.max_allele_endpoint > up_to_position) break;
            keys.push_back(kv.first);
        }
        int64_t called = 0;
        std::vector<PiscesCalledAllele> point_recs, span_recs;
        std::vector<HostCandidate> span_cands;
        // host-side candidates first: with MNV calling on they register the reference support that gapped MNVs take, which the
        // Reference records of call_blocks must see (AlleleCaller.cs:95, CoverageCalculator.cs:82-97)
        int64_t collapsed = 0;
        std::vector<PiscesCalledAllele> ref_overrides;
        int32_t rc = call_spanning(h, keys, final_flush ? -1 : up_to_position, span_recs, span_cands, &called, &collapsed, ref_overrides);
        h->pending_collapsed = collapsed;
        if (rc) return rc;
        h->pending_dropped = false;
        rc = call_blocks(h, keys, point_recs, &called, true, &h->pending_dropped, &h->pending_kept);
        if (rc) return rc;
        if (!ref_overrides.empty()) {   // Reference alleles that MNV reallocation added support to
            std::map<int32_t, const PiscesCalledAllele*> by_pos;
            for (auto& r : ref_overrides) by_pos[r.position] = &r;
            for (auto& r : point_recs) {
                if (PISCES_INFO_CATEGORY(r.info) != PISCES_CAT_REFERENCE) continue;
                auto it = by_pos.find(r.position);
                if (it != by_pos.end()) { r = *it->second; by_pos.erase(it); }
            }
            // not a gVCF: the Reference alleles at forced positions have no tile-kernel record to replace; they are rows (and calls,
            // AlleleCaller.IsCallable) of their own
            if (!h->cfg.include_reference_calls)
                for (auto& kv : by_pos) { point_recs.push_back(*kv.second); called++; }
        }
        // per locus: drop the Reference row when a variant is reported there (AlleleCaller.cs:146-147), then order by
        // position, reference allele, alternate allele (:172-176; ordinal order of upper-case ASCII allele strings)
        h->pending.clear();
        h->pending_cand_index.clear();
        h->pending_cands = span_cands;
        const bool diploid = h->cfg.ploidy == PISCES_PLOIDY_DIPLOID || h->cfg.ploidy == PISCES_PLOIDY_HAPLOID;   // per-locus genotypers
        if (span_recs.empty() && !diploid && h->forced.empty()) {
            h->pending = std::move(point_recs);
            h->pending_cand_index.assign(h->pending.size(), -1);
        } else {
            struct Row { const PiscesCalledAllele* r; int32_t ci; std::string ref, alt; };
            static const char kBase[6] = {'A', 'G', 'C', 'T', 'N', 'D'};
            std::vector<Row> rows;
            // (a variant that is only there because it was forced prunes nothing: AlleleCaller.cs:146)
            auto forced_to_report = [](const PiscesCalledAllele& r) { return ((r.filter_bits >> PISCES_FILTER_FORCED_REPORT) & 1u) != 0; };
            std::vector<int32_t> variant_pos;
            for (auto& r : span_recs)
                if (!forced_to_report(r)) variant_pos.push_back(r.position);
            if (!h->forced.empty())   // forced alleles given: Reference rows can come from the candidate kernel, beside the tile kernels' SNV rows
                for (auto& r : point_recs)
                    if (PISCES_INFO_CATEGORY(r.info) != PISCES_CAT_REFERENCE) variant_pos.push_back(r.position);
            std::sort(variant_pos.begin(), variant_pos.end());
            for (auto& r : point_recs) {
                const bool is_ref = PISCES_INFO_CATEGORY(r.info) == PISCES_CAT_REFERENCE;
                if (is_ref && std::binary_search(variant_pos.begin(), variant_pos.end(), r.position)) continue;
                rows.push_back({&r, -1, std::string(1, kBase[PISCES_INFO_REF(r.info)]), std::string(1, kBase[PISCES_INFO_ALT(r.info)])});
            }
            for (size_t i = 0; i < span_recs.size(); i++) rows.push_back({&span_recs[i], (int32_t)i, span_cands[i].ref, span_cands[i].alt});
            std::stable_sort(rows.begin(), rows.end(), [](const Row& a, const Row& b) {
                if (a.r->position != b.r->position) return a.r->position < b.r->position;
                if (a.ref != b.ref) return a.ref < b.ref;
                return a.alt < b.alt;
            });
            if (!diploid) {
                for (auto& row : rows) { h->pending.push_back(*row.r); h->pending_cand_index.push_back(row.ci); }
            } else {
                // ComputeGenotypeAndFilterAllele :143-177 with DiploidThresholdingGenotyper: one genotype per locus, alleles beyond the
                // ploidy dropped, every kept allele gets its own diploid genotype q-score, LowGQ and MultiAllelicSite filters; the
                // device's somatic genotype fields are replaced.  (Reference rows at variant loci are gone already, rows are in
                // (ref, alt) order.)
                std::vector<DiploidAllele> at;
                std::vector<size_t> at_row;
                for (size_t i = 0; i < rows.size();) {
                    size_t j = i;
                    while (j < rows.size() && rows[j].r->position == rows[i].r->position) j++;
                    at.clear();
                    at_row.clear();
                    for (size_t k = i; k < j; k++) {
                        if (forced_to_report(*rows[k].r)) continue;   // the genotyper does not see alleles that are only there because they were forced (:150)
                        DiploidAllele a;
                        a.category = PISCES_INFO_CATEGORY(rows[k].r->info);
                        a.ref = rows[k].ref;
                        a.alt = rows[k].alt;
                        a.support = rows[k].r->allele_support;
                        a.coverage = rows[k].r->total_coverage;
                        a.ref_support = rows[k].r->reference_support;
                        at.push_back(std::move(a));
                        at_row.push_back(k);
                    }
                    if (h->cfg.ploidy == PISCES_PLOIDY_HAPLOID)
                        (void)haploid_set_genotypes(at, h->cfg.diploid_snv_params[0], h->cfg.diploid_snv_params[1], h->cfg.min_coverage,
                                                    h->cfg.min_genotype_qscore, h->cfg.max_genotype_qscore);
                    else
                        (void)diploid_set_genotypes(at, h->cfg.diploid_snv_params, h->cfg.diploid_indel_params, h->cfg.min_coverage,
                                                    h->cfg.min_genotype_qscore, h->cfg.max_genotype_qscore);
                    const size_t first_out = h->pending.size();
                    size_t ai = 0;
                    for (size_t k = i; k < j; k++) {
                        PiscesCalledAllele r = *rows[k].r;
                        if (ai < at_row.size() && at_row[ai] == k) {
                            const DiploidAllele& a = at[ai++];
                            // an allele beyond the ploidy goes, unless it is a forced allele (:155-163)
                            if (a.prune && !(!h->forced_keys.empty() && h->forced_keys.count(forced_key(r.position, rows[k].ref, rows[k].alt)))) continue;
                            r.info = (uint16_t)((r.info & ~0xFu) | ((uint32_t)a.genotype & 0xFu));
                            r.genotype_qscore = a.genotype_qscore;
                            uint32_t fb = r.filter_bits & ~(1u << PISCES_FILTER_LOW_GENOTYPE_QUALITY) & 0x3FFFu;
                            if (a.multi_allelic) fb |= 1u << PISCES_FILTER_MULTI_ALLELIC_SITE;
                            if (h->cfg.low_gq_filter >= 0 && (float)a.genotype_qscore < (float)h->cfg.low_gq_filter) fb |= 1u << PISCES_FILTER_LOW_GENOTYPE_QUALITY;
                            fb |= (uint32_t)(a.phase_set_index & 3) << 14;
                            r.filter_bits = (uint16_t)fb;
                        }
                        h->pending.push_back(r);
                        h->pending_cand_index.push_back(rows[k].ci);
                    }
                    if (h->cfg.ploidy == PISCES_PLOIDY_DIPLOID && !h->forced.empty()) {
                        // DiploidLocusProcessor.Process (DiploidLocusProcessor.cs:13-52): a forced allele takes the genotype the other alleles of
                        // the position imply, every allele the smallest genotype q-score among those others
                        bool any_forced = false, any_other = false, is_ref = false, is_no_call = false;
                        int min_gq = 0;
                        for (size_t q = first_out; q < h->pending.size(); q++) {
                            const PiscesCalledAllele& r = h->pending[q];
                            if (forced_to_report(r)) { any_forced = true; continue; }
                            const int g = PISCES_INFO_GENOTYPE(r.info);
                            if (PISCES_INFO_CATEGORY(r.info) == PISCES_CAT_REFERENCE) is_ref = true;
                            if (g == PISCES_GT_ALT12_LIKE_NOCALL || g == PISCES_GT_ALT_LIKE_NOCALL || g == PISCES_GT_HEMI_NOCALL || g == PISCES_GT_REF_LIKE_NOCALL) is_no_call = true;
                            if (!any_other || r.genotype_qscore < min_gq) min_gq = r.genotype_qscore;
                            any_other = true;
                        }
                        if (any_forced) {
                            if (!any_other) is_no_call = true;
                            const uint32_t genotype = is_no_call ? PISCES_GT_ALT_LIKE_NOCALL : is_ref ? PISCES_GT_HOM_REF : PISCES_GT_OTHERS;
                            for (size_t q = first_out; q < h->pending.size(); q++) {
                                PiscesCalledAllele& r = h->pending[q];
                                if (forced_to_report(r)) r.info = (uint16_t)((r.info & ~0xFu) | genotype);
                                r.genotype_qscore = (int16_t)(any_other ? min_gq : 0);
                            }
                        }
                    }
                    i = j;
                }
            }
        }
        h->pending_keys = keys;
        h->pending_called = called;
        h->pending_up_to = up_to_position;
        h->pending_valid = true;
    }
    int64_t pool_bytes = 0;
    for (auto& c : h->pending_cands) pool_bytes += (int64_t)(c.ref.size() + c.alt.size());
    if (n_cand) *n_cand = (int64_t)h->pending_cands.size();
    if (allele_bytes) *allele_bytes = pool_bytes;
    const bool cand_too_small = cand_out && ((int64_t)h->pending_cands.size() > cand_capacity || (alleles_out && pool_bytes > allele_capacity));
    if ((int64_t)h->pending.size() > capacity || cand_too_small) {
        *n_out = (int64_t)h->pending.size();
        return fail(h, PISCES_E_BUFFER_TOO_SMALL, "flush: output buffer too small");
    }
    if (!h->pending.empty()) std::memcpy(out, h->pending.data(), h->pending.size() * sizeof(PiscesCalledAllele));
    if (cand_index_out && !h->pending.empty()) std::memcpy(cand_index_out, h->pending_cand_index.data(), h->pending.size() * sizeof(int32_t));
    if (cand_out) {
        int64_t off = 0;
        for (size_t i = 0; i < h->pending_cands.size(); i++) {
            const HostCandidate& c = h->pending_cands[i];
            PiscesCandidate& o = cand_out[i];
            std::memset(&o, 0, sizeof(o));
            o.position = c.position; o.category = c.category;
            o.ref_len = (int32_t)c.ref.size(); o.alt_len = (int32_t)c.alt.size();
            for (int d = 0; d < 3; d++) { o.support_by_dir[d] = c.support_by_dir[d]; o.well_anchored_by_dir[d] = c.well_anchored_by_dir[d]; }
            o.open_left = c.open_left; o.open_right = c.open_right;
            o.allele_offset = off;
            if (alleles_out) {
                std::memcpy(alleles_out + off, c.ref.data(), c.ref.size());
                std::memcpy(alleles_out + off + c.ref.size(), c.alt.data(), c.alt.size());
            }
            off += (int64_t)(c.ref.size() + c.alt.size());
        }
    }
    *n_out = (int64_t)h->pending.size();
    // DoneProcessing (RegionStateManager.cs:336-353): the log entries of the flushed blocks left with call_blocks' submission when
    // there was one; what remains is to make that buffer the log
    if (h->pending_dropped) {
        commit_drop(h, h->pending_kept);
        h->pending_dropped = false;
    } else {
        int32_t rcd = drop_blocks(h, h->pending_keys);
        if (rcd) return rcd;
    }
    for (int32_t key : h->pending_keys) {
        h->blocks.erase(key);
        const int32_t bstart = (key - 1) * h->cfg.block_size + 1, bend = key * h->cfg.block_size;
        for (auto it = h->gapped_mnv_ref.begin(); it != h->gapped_mnv_ref.end();)
            it = (it->first >= bstart && it->first <= bend) ? h->gapped_mnv_ref.erase(it) : std::next(it);
    }
    h->last_block = nullptr;
    h->stats[0] += h->pending_called;
    h->stats[1] += h->pending_collapsed;
    h->last_up_to_block_key = final_flush ? -1 : block_key(h, up_to_position);
    h->pending_valid = false;
    h->pending.clear();
    h->pending_cand_index.clear();
    h->pending_cands.clear();
    h->pending_keys.clear();
    return PISCES_OK;
    });
}

int32_t pisces_hip_flush(PiscesHip* h, int32_t up_to_position, PiscesCalledAllele* out, int64_t capacity, int64_t* n_out)
{
    return abi_guard<int32_t>(h, [&]() -> int32_t {
    return pisces_hip_flush_ex(h, up_to_position, out, capacity, n_out, nullptr, nullptr, 0, nullptr, nullptr, 0, nullptr);
    });
}

int32_t pisces_hip_get_counts(PiscesHip* h, int32_t start_position, int32_t n, int32_t* out)
{
    return abi_guard<int32_t>(h, [&]() -> int32_t {
    if (!h) return PISCES_E_INVALID_ARG;
    if (n < 0 || (n > 0 && !out)) return fail(h, PISCES_E_INVALID_ARG, "get_counts: null output");
    if (start_position <= 0) return fail(h, PISCES_E_INVALID_ARG, "Position must be greater than 0.");
    PISCES_HIP_CHECK(h, hipSetDevice(h->device));
    std::memset(out, 0, (size_t)n * PISCES_COUNTS_PER_LOCUS * sizeof(int32_t));
    if (n == 0) return PISCES_OK;
    std::vector<int32_t> keys;
    for (int32_t k = block_key(h, start_position); k <= block_key(h, start_position + n - 1); k++)
        if (h->blocks.count(k)) keys.push_back(k);
    if (keys.empty()) return PISCES_OK;
    // counts are served over the whole block grid, not the interval-clipped tiles
    std::vector<PiscesTile> tiles;
    int32_t rc = bucket_blocks(h, keys, false, tiles);
    if (rc) return rc;
    const int32_t n_tiles = (int32_t)tiles.size();
    const size_t nc = (size_t)n_tiles * kTile * PISCES_COUNTS_PER_LOCUS;
    PISCES_HIP_CHECK(h, accumulate_tiles(h, h->stream, h->d_tuples.p, h->d_tiles.p, n_tiles, false));
    std::vector<int32_t> host(nc);
    PISCES_HIP_CHECK(h, hipMemcpyAsync(host.data(), h->d_counts.p, nc * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
    PISCES_HIP_CHECK(h, hipStreamSynchronize(h->stream));
    for (int32_t t = 0; t < n_tiles; t++)
        for (int32_t l = 0; l < tiles[(size_t)t].n_loci; l++) {
            int32_t p = tiles[(size_t)t].start_position + l;
            if (p < start_position || p >= start_position + n) continue;
            std::memcpy(out + (size_t)(p - start_position) * PISCES_COUNTS_PER_LOCUS,
                        host.data() + ((size_t)t * kTile + (size_t)l) * PISCES_COUNTS_PER_LOCUS,
                        PISCES_COUNTS_PER_LOCUS * sizeof(int32_t));
        }
    return PISCES_OK;
    });
}

// IAlleleSource.GetSumOfAlleleBaseQualities (RegionState._sumOfAlleleBaseQualities, RegionState.cs:61,233-239): the cells of
// [start_position, start_position + n), layout as pisces_hip_get_counts, accumulated on the device next to the counts from the
// observation log (Math.Pow(10, -(int)q / 10f) per base under its post-threshold allele, RegionStateManager.cs:191) in fixed point:
// the true sum rounded once, the same bits from run to run; the reference adds doubles in read order, equal to rounding.
int32_t pisces_hip_get_base_quality_sums(PiscesHip* h, int32_t start_position, int32_t n, double* out)
{
    return abi_guard<int32_t>(h, [&]() -> int32_t {
    if (!h) return PISCES_E_INVALID_ARG;
    if (n < 0 || (n > 0 && !out)) return fail(h, PISCES_E_INVALID_ARG, "get_base_quality_sums: null output");
    if (start_position <= 0) return fail(h, PISCES_E_INVALID_ARG, "Position must be greater than 0.");
    PISCES_HIP_CHECK(h, hipSetDevice(h->device));
    std::memset(out, 0, (size_t)n * PISCES_COUNTS_PER_LOCUS * sizeof(double));
    if (n == 0) return PISCES_OK;
    std::vector<int32_t> keys;
    for (int32_t k = block_key(h, start_position); k <= block_key(h, start_position + n - 1); k++)
        if (h->blocks.count(k)) keys.push_back(k);
    if (keys.empty()) return PISCES_OK;
    std::vector<PiscesTile> tiles;
    int32_t rc = bucket_blocks(h, keys, false, tiles);
    if (rc) return rc;
    const int32_t n_tiles = (int32_t)tiles.size();
    const size_t nc = (size_t)n_tiles * kTile * PISCES_COUNTS_PER_LOCUS;
    PISCES_HIP_CHECK(h, accumulate_tiles(h, h->stream, h->d_tuples.p, h->d_tiles.p, n_tiles, true));   // (any handle can serve the sums, not only NoiseModel.Window)
    std::vector<double> host(nc);
    PISCES_HIP_CHECK(h, hipMemcpyAsync(host.data(), h->d_sumq.p, nc * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    PISCES_HIP_CHECK(h, hipStreamSynchronize(h->stream));
    for (int32_t t = 0; t < n_tiles; t++)
        for (int32_t l = 0; l < tiles[(size_t)t].n_loci; l++) {
            int32_t p = tiles[(size_t)t].start_position + l;
            if (p < start_position || p >= start_position + n) continue;
            std::memcpy(out + (size_t)(p - start_position) * PISCES_COUNTS_PER_LOCUS,
                        host.data() + ((size_t)t * kTile + (size_t)l) * PISCES_COUNTS_PER_LOCUS, PISCES_COUNTS_PER_LOCUS * sizeof(double));
        }
    return PISCES_OK;
    });
}

// IAlleleSource.GetGappedMnvRefCount (RegionStateManager.cs: the lookup AddGappedMnvRefCount fills)
int32_t pisces_hip_get_gapped_mnv_ref(PiscesHip* h, int32_t position, int32_t* count)
{
    return abi_guard<int32_t>(h, [&]() -> int32_t {
    if (!h || !count) return PISCES_E_INVALID_ARG;
    auto it = h->gapped_mnv_ref.find(position);
    *count = it == h->gapped_mnv_ref.end() ? 0 : it->second;
    return PISCES_OK;
    });
}

int32_t pisces_hip_add_gapped_mnv_ref(PiscesHip* h, const int32_t* positions, const int32_t* counts, int32_t n)
{
    return abi_guard<int32_t>(h, [&]() -> int32_t {
    if (!h) return PISCES_E_INVALID_ARG;
    if (n < 0 || (n > 0 && (!positions || !counts))) return fail(h, PISCES_E_INVALID_ARG, "add_gapped_mnv_ref: null buffer");
    { int32_t rcp = refuse_while_batch_is_open(h, "add_gapped_mnv_ref"); if (rcp) return rcp; }
    for (int32_t i = 0; i < n; i++) {
        if (positions[i] <= 0) return fail(h, PISCES_E_INVALID_ARG, "Position must be greater than 0.");
        (void)get_block(h, positions[i]);   // GetBlock(position) creates the block (RegionStateManager.cs:78)
        h->gapped_mnv_ref[positions[i]] += counts[i];
    }
    return PISCES_OK;
    });
}

int32_t pisces_hip_get_candidates(PiscesHip* h, int32_t up_to_position, PiscesCandidate* out, int64_t capacity, int64_t* n_out,
                                  uint8_t* alleles, int64_t allele_capacity, int64_t* allele_bytes)
{
    return abi_guard<int32_t>(h, [&]() -> int32_t {
    if (!h || !n_out) return PISCES_E_INVALID_ARG;
    // the candidates collected so far: insertions / deletions, and with MNV calling on the SNVs / MNVs of the read walk (with it
    // off SNV candidates never leave the device: they are the allele counts)
    { int32_t rcf = consume_found(h); if (rcf) return rcf; }
    int64_t n = 0, bytes = 0;
    for (auto& kv : h->blocks)
        for (auto& c : kv.second.cands) {
            if (up_to_position >= 0 && c.position > up_to_position) continue;
            if (out && n < capacity && (!alleles || bytes + (int64_t)(c.ref.size() + c.alt.size()) <= allele_capacity)) {
                PiscesCandidate& o = out[n];
                std::memset(&o, 0, sizeof(o));
                o.position = c.position; o.category = c.category;
                o.ref_len = (int32_t)c.ref.size(); o.alt_len = (int32_t)c.alt.size();
                for (int d = 0; d < 3; d++) { o.support_by_dir[d] = c.support_by_dir[d]; o.well_anchored_by_dir[d] = c.well_anchored_by_dir[d]; }
                o.open_left = c.open_left; o.open_right = c.open_right;
                o.allele_offset = bytes;
                if (alleles) {
                    std::memcpy(alleles + bytes, c.ref.data(), c.ref.size());
                    std::memcpy(alleles + bytes + c.ref.size(), c.alt.data(), c.alt.size());
                }
            }
            n++;
            bytes += (int64_t)(c.ref.size() + c.alt.size());
        }
    *n_out = n;
    if (allele_bytes) *allele_bytes = bytes;
    if (out && (n > capacity || (alleles && bytes > allele_capacity))) return fail(h, PISCES_E_BUFFER_TOO_SMALL, "get_candidates: buffer too small");
    return PISCES_OK;
    });
}

int32_t pisces_hip_stats(PiscesHip* h, int64_t out[4])
{
    return abi_guard<int32_t>(h, [&]() -> int32_t {
    if (!h || !out) return PISCES_E_INVALID_ARG;
    for (int i = 0; i < 4; i++) out[i] = h->stats[i];
    unsigned long long appended = 0;   // observations: counted where they are made, on the device
    PISCES_HIP_CHECK(h, hipSetDevice(h->device));
    PISCES_HIP_CHECK(h, hipStreamSynchronize(h->stream));
    PISCES_HIP_CHECK(h, hipMemcpy(&appended, h->d_log_n.p + 2, sizeof(appended), hipMemcpyDeviceToHost));
    out[3] = (int64_t)appended;
    return PISCES_OK;
    });
}

// ------------------------------------------------------------------------------------------------
// device-resident surface
// ------------------------------------------------------------------------------------------------
int32_t pisces_hip_call_tiles(PiscesHip* h, const uint32_t* d_tuples, const PiscesTile* d_tiles, int32_t n_tiles,
                              const uint8_t* d_ref_bases, int32_t ref_start_position, int64_t ref_length,
                              PiscesCalledAllele* d_records, int32_t record_capacity, PiscesTileResult* d_tile_results, void* stream)
{
    return abi_guard<int32_t>(h, [&]() -> int32_t {
    if (!h) return PISCES_E_INVALID_ARG;
    if (n_tiles < 0 || record_capacity < 0 || ref_length < 0) return fail(h, PISCES_E_INVALID_ARG, "call_tiles: negative size");
    if (n_tiles > 0 && (!d_tiles || !d_ref_bases || !d_records || !d_tile_results))
        return fail(h, PISCES_E_INVALID_ARG, "call_tiles: null device pointer");
    if ((int64_t)record_capacity < (int64_t)n_tiles * kSlotsPerTile)
        return fail(h, PISCES_E_BUFFER_TOO_SMALL, "call_tiles: the slot layout needs record_capacity >= 256 * n_tiles");
    if (h->cfg.ploidy != PISCES_PLOIDY_SOMATIC)
        return fail(h, PISCES_E_STATE, "call_tiles: diploid / haploid genotyping is a per-locus pass of pisces_hip_flush (streaming surface)");
    PISCES_HIP_CHECK(h, hipSetDevice(h->device));
    hipStream_t s = stream ? (hipStream_t)stream : h->stream;
    // events only when asked for (pisces_hip_set_timing): an event record is a queue packet of its own, and two of them
    // per launch cost a few microseconds between back-to-back launches
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (h->timing > 0 && (h->launches_seen++ % h->timing) == 0) {
        const size_t slot = (size_t)(h->ring_used % kTimingRing);
        e0 = h->ring[2 * slot];
        e1 = h->ring[2 * slot + 1];
        h->ring_used++;
    }
    if (n_tiles > 0) {
        PISCES_HIP_CHECK(h, launch_call_tiles(h, s, d_tuples, d_tiles, n_tiles, d_ref_bases, ref_start_position, ref_length, d_records, d_tile_results, e0, e1));
    } else if (e0) {
        PISCES_HIP_CHECK(h, hipEventRecord(e0, s));
        PISCES_HIP_CHECK(h, hipEventRecord(e1, s));
    }
    PISCES_HIP_CHECK(h, hipGetLastError());
    return PISCES_OK;
    });
}

// Tile size for a launch of n_loci contiguous loci that keeps every CU equally loaded.  The hot kernel is HBM-bound and a CU streams
// at most ~1/256 of the chip's bandwidth, so a launch ends with the CU that holds the most tiles: 1563 tiles of 64 loci leave 27 CUs
// with 7 tiles and the rest with 6 (the launch takes 7/6.1 of the balanced time), 1786 tiles of 56 loci give every CU 7.  When the
// whole launch is resident at once (up to 8 two-wave tiles per CU) the tile count is made a multiple of the CU count; larger launches
// run in many rounds and balance themselves: 64.
int32_t pisces_hip_balanced_tile_loci(PiscesHip* h, int64_t n_loci)
{
    return abi_guard<int32_t>(h, [&]() -> int32_t {
    if (!h || n_loci <= 0) return kTile;
    const int64_t cus = std::max(1, h->n_cus);
    const int64_t per_cu = (n_loci + (int64_t)kTile * cus - 1) / ((int64_t)kTile * cus);   // tiles per CU at 64 loci
    if (per_cu > 8) return kTile;
    const int64_t n_tiles = per_cu * cus;
    return (int32_t)std::min<int64_t>(kTile, (n_loci + n_tiles - 1) / n_tiles);
    });
}

int32_t pisces_hip_call_tiles_batched(PiscesHip* h, const PiscesTileBatch* batches, int32_t n_batches, void* stream)
{
    return abi_guard<int32_t>(h, [&]() -> int32_t {
    if (!h) return PISCES_E_INVALID_ARG;
    if (n_batches < 0 || (n_batches > 0 && !batches)) return fail(h, PISCES_E_INVALID_ARG, "call_tiles_batched: null batch list");
    if (h->cfg.ploidy != PISCES_PLOIDY_SOMATIC)
        return fail(h, PISCES_E_STATE, "call_tiles: diploid / haploid genotyping is a per-locus pass of pisces_hip_flush (streaming surface)");
    if (h->cfg.noise_model == PISCES_NOISE_WINDOW)
        return fail(h, PISCES_E_STATE, "call_tiles_batched: NoiseModel.Window calls through the handle's one counts tensor; use pisces_hip_call_tiles");
    for (int32_t i = 0; i < n_batches; i++) {
        const PiscesTileBatch& b = batches[i];
        if (b.n_tiles < 0 || b.record_capacity < 0 || b.ref_length < 0) return fail(h, PISCES_E_INVALID_ARG, "call_tiles: negative size");
        if (b.n_tiles > 0 && (!b.d_tiles || !b.d_ref_bases || !b.d_records || !b.d_tile_results))
            return fail(h, PISCES_E_INVALID_ARG, "call_tiles: null device pointer");
        if ((int64_t)b.record_capacity < (int64_t)b.n_tiles * kSlotsPerTile)
            return fail(h, PISCES_E_BUFFER_TOO_SMALL, "call_tiles: the slot layout needs record_capacity >= 256 * n_tiles");
    }
    if (n_batches == 0) return PISCES_OK;
    PISCES_HIP_CHECK(h, hipSetDevice(h->device));
    if (!h->lane[0])
        for (int k = 0; k < PiscesHip::kLanes; k++) PISCES_HIP_CHECK(h, hipStreamCreateWithFlags(&h->lane[k], hipStreamNonBlocking));
    // Ordering is on the host, not through HIP events: a lane that has waited on an event of another stream runs every later kernel
    // ~5 us slower on this runtime (measured: 43 us per config-2 step with an event fork / join, 38 us without), which is most of
    // what the lanes are for.  So: inputs must be complete on `stream` - the call waits for it here - and the outputs are complete
    // after pisces_hip_synchronize.
    const int lanes = std::min<int>(PiscesHip::kLanes, n_batches);
    if (stream) PISCES_HIP_CHECK(h, hipStreamSynchronize((hipStream_t)stream));
    for (int32_t i = 0; i < n_batches; i++) {
        const PiscesTileBatch& b = batches[i];
        if (b.n_tiles == 0) continue;
        PISCES_HIP_CHECK(h, launch_call_tiles(h, h->lane[i % lanes], b.d_tuples, b.d_tiles, b.n_tiles, b.d_ref_bases, b.ref_start_position,
                                              b.ref_length, b.d_records, b.d_tile_results));
    }
    PISCES_HIP_CHECK(h, hipGetLastError());
    return PISCES_OK;
    });
}

int32_t pisces_hip_compact_records(PiscesHip* h, const PiscesCalledAllele* d_records, const PiscesTileResult* d_tile_results,
                                   int32_t n_tiles, int32_t* d_offsets, PiscesCalledAllele* d_out, int32_t out_capacity,
                                   int32_t* d_count, void* stream)
{
    return abi_guard<int32_t>(h, [&]() -> int32_t {
    if (!h) return PISCES_E_INVALID_ARG;
    if (n_tiles < 0 || out_capacity < 0) return fail(h, PISCES_E_INVALID_ARG, "compact_records: negative size");
    if (!d_count || (n_tiles > 0 && (!d_records || !d_tile_results || !d_offsets || !d_out)))
        return fail(h, PISCES_E_INVALID_ARG, "compact_records: null device pointer");
    PISCES_HIP_CHECK(h, hipSetDevice(h->device));
    hipStream_t s = stream ? (hipStream_t)stream : h->stream;
    if (n_tiles == 0) {
        PISCES_HIP_CHECK(h, hipMemsetAsync(d_count, 0, sizeof(int32_t), s));
        return PISCES_OK;
    }
    launch_compaction(s, d_records, d_tile_results, n_tiles, d_offsets, d_out, out_capacity, d_count);
    PISCES_HIP_CHECK(h, hipGetLastError());
    return PISCES_OK;
    });
}

int32_t pisces_hip_accumulate_tiles(PiscesHip* h, const uint32_t* d_tuples, const PiscesTile* d_tiles, int32_t n_tiles,
                                    int32_t* d_counts, void* stream)
{
    return abi_guard<int32_t>(h, [&]() -> int32_t {
    if (!h) return PISCES_E_INVALID_ARG;
    if (n_tiles < 0) return fail(h, PISCES_E_INVALID_ARG, "accumulate_tiles: negative size");
    if (n_tiles > 0 && (!d_tiles || !d_counts)) return fail(h, PISCES_E_INVALID_ARG, "accumulate_tiles: null device pointer");
    PISCES_HIP_CHECK(h, hipSetDevice(h->device));
    hipStream_t s = stream ? (hipStream_t)stream : h->stream;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (h->timing > 0 && (h->launches_seen++ % h->timing) == 0) {
        const size_t slot = (size_t)(h->ring_used % kTimingRing);
        e0 = h->ring[2 * slot];
        e1 = h->ring[2 * slot + 1];
        h->ring_used++;
    }
    if (n_tiles > 0) {
        hipExtLaunchKernelGGL(accumulate_tiles_kernel, dim3((unsigned)n_tiles), dim3(kBlock), 0u, s, e0, e1, 0u, d_tuples, d_tiles, n_tiles,
                              d_counts, h->cfg.min_base_call_quality, (unsigned long long*)nullptr, (const ulonglong2*)nullptr);
    } else if (e0) {
        PISCES_HIP_CHECK(h, hipEventRecord(e0, s));
        PISCES_HIP_CHECK(h, hipEventRecord(e1, s));
    }
    PISCES_HIP_CHECK(h, hipGetLastError());
    return PISCES_OK;
    });
}

int32_t pisces_hip_device_totals(PiscesHip* h, int64_t out[4], int32_t reset)
{
    return abi_guard<int32_t>(h, [&]() -> int32_t {
    if (!h || !out) return PISCES_E_INVALID_ARG;
    PISCES_HIP_CHECK(h, hipSetDevice(h->device));
    PISCES_HIP_CHECK(h, hipDeviceSynchronize());   // launches may sit on caller-supplied streams
    unsigned long long host[kTotalShards * kTotalStride];
    PISCES_HIP_CHECK(h, hipMemcpy(host, h->d_totals.p, sizeof(host), hipMemcpyDeviceToHost));
    for (int i = 0; i < 4; i++) {
        out[i] = 0;
        for (int sh = 0; sh < kTotalShards; sh++) out[i] += (int64_t)host[sh * kTotalStride + i];
    }
    if (reset) PISCES_HIP_CHECK(h, hipMemset(h->d_totals.p, 0, sizeof(host)));
    return PISCES_OK;
    });
}

int32_t pisces_hip_set_timing(PiscesHip* h, int32_t enable)
{
    return abi_guard<int32_t>(h, [&]() -> int32_t {
    if (!h) return PISCES_E_INVALID_ARG;
    PISCES_HIP_CHECK(h, hipSetDevice(h->device));
    if (enable && h->ring.empty()) {
        h->ring.resize((size_t)(2 * kTimingRing), nullptr);
        for (auto& ev : h->ring) PISCES_HIP_CHECK(h, hipEventCreate(&ev));
    }
    h->timing = enable > 0 ? enable : 0;
    h->ring_used = 0;
    h->launches_seen = 0;
    return PISCES_OK;
    });
}

int32_t pisces_hip_kernel_time(PiscesHip* h, double* total_ms, int64_t* launches)
{
    return abi_guard<int32_t>(h, [&]() -> int32_t {
    if (!h || !total_ms || !launches) return PISCES_E_INVALID_ARG;
    PISCES_HIP_CHECK(h, hipSetDevice(h->device));
    const int64_t n = std::min<int64_t>(h->ring_used, kTimingRing);
    double sum = 0.0;
    for (int64_t i = 0; i < n; i++) {
        float ms = 0.f;
        PISCES_HIP_CHECK(h, hipEventSynchronize(h->ring[(size_t)(2 * i + 1)]));
        PISCES_HIP_CHECK(h, hipEventElapsedTime(&ms, h->ring[(size_t)(2 * i)], h->ring[(size_t)(2 * i + 1)]));
        sum += ms;
    }
    *total_ms = sum;
    *launches = n;
    return PISCES_OK;
    });
}

int32_t pisces_hip_probe_read_bandwidth(PiscesHip* h, int64_t nbytes, int32_t reps, double* gb_per_s)
{
    return abi_guard<int32_t>(h, [&]() -> int32_t {
    if (!h || !gb_per_s || nbytes < (1 << 20) || reps < 1) return fail(h, PISCES_E_INVALID_ARG, "probe_read_bandwidth: bad arguments");
    PISCES_HIP_CHECK(h, hipSetDevice(h->device));
    DeviceBuf<uint32_t> buf;
    PISCES_HIP_CHECK(h, buf.reserve((size_t)(nbytes / 4) + 4));
    PISCES_HIP_CHECK(h, hipMemsetAsync(buf.p, 0x5A, (size_t)nbytes, h->stream));
    const int64_t n4 = nbytes / 16;
    const unsigned grid = (unsigned)std::min<int64_t>((n4 + 2047) / 2048, (int64_t)h->n_cus * 32);
    hipLaunchKernelGGL(read_probe_kernel, dim3(grid), dim3(256), 0, h->stream, (const u32x4*)buf.p, n4, buf.p + nbytes / 4);   // warm-up
    double best = 0.0;
    for (int r = 0; r < reps; r++) {
        hipExtLaunchKernelGGL(read_probe_kernel, dim3(grid), dim3(256), 0u, h->stream, h->ev0, h->ev1, 0u, (const u32x4*)buf.p, n4,
                              buf.p + nbytes / 4);
        PISCES_HIP_CHECK(h, hipEventSynchronize(h->ev1));
        float ms = 0.f;
        PISCES_HIP_CHECK(h, hipEventElapsedTime(&ms, h->ev0, h->ev1));
        if (ms > 0.f) best = std::max(best, (double)nbytes / ((double)ms * 1e-3) / 1e9);
    }
    buf.release();
    *gb_per_s = best;
    return PISCES_OK;
    });
}

// ---- BGZF (row f4, upstream of the read batch) ----
int64_t pisces_hip_bgzf_scan(const uint8_t* file, int64_t n_bytes, PiscesBgzfBlock* blocks, int64_t capacity, int64_t* inflated_bytes)
{
    return abi_guard<int64_t>((PiscesHip*)nullptr, [&]() -> int64_t {
    if (!file || n_bytes < 0 || capacity < 0 || (capacity > 0 && !blocks)) return PISCES_E_INVALID_ARG;
    int64_t pos = 0, n = 0, out = 0;
    while (pos < n_bytes) {
        // gzip member header (RFC 1952) with FEXTRA; BamConstants.BlockHeaderLength = 18 is the XLEN = 6 case (BamCommon.cs:989)
        if (pos + 12 > n_bytes) return PISCES_E_INVALID_ARG;
        const uint8_t* b = file + pos;
        if (b[0] != 31 || b[1] != 139 || b[2] != 8 || !(b[3] & 4)) return PISCES_E_INVALID_ARG;
        const int64_t xlen = b[10] | ((int64_t)b[11] << 8);
        if (pos + 12 + xlen > n_bytes) return PISCES_E_INVALID_ARG;
        int64_t bsize = -1;
        for (int64_t x = 0; x + 4 <= xlen;) {   // the BC subfield: total block size - 1 (BamReader.cs:622)
            const uint8_t* f = b + 12 + x;
            const int64_t slen = f[2] | ((int64_t)f[3] << 8);
            if (f[0] == 'B' && f[1] == 'C' && slen == 2 && x + 6 <= xlen) bsize = (f[4] | ((int64_t)f[5] << 8)) + 1;
            x += 4 + slen;
        }
        const int64_t header = 12 + xlen;
        if (bsize < header + 8 || pos + bsize > n_bytes) return PISCES_E_INVALID_ARG;
        const uint8_t* tr = b + bsize - 8;
        PiscesBgzfBlock blk;
        blk.in_offset = pos + header;
        blk.in_length = (int32_t)(bsize - header - 8);
        blk.crc32 = tr[0] | ((uint32_t)tr[1] << 8) | ((uint32_t)tr[2] << 16) | ((uint32_t)tr[3] << 24);
        const uint32_t isize = tr[4] | ((uint32_t)tr[5] << 8) | ((uint32_t)tr[6] << 16) | ((uint32_t)tr[7] << 24);
        if (isize > 65536u) return PISCES_E_INVALID_ARG;   // BgzfCommon.MaxBlockSize
        blk.out_length = (int32_t)isize;
        blk.out_offset = out;
        blk.reserved = 0;
        if (n < capacity) blocks[n] = blk;
        n++;
        out += isize;
        pos += bsize;
    }
    if (inflated_bytes) *inflated_bytes = out;
    return n;
    });
}

static uint32_t crc32_of(const uint8_t* p, size_t n)
{
    static uint32_t table[8][256];
    static std::once_flag once;
    std::call_once(once, [] {
        for (uint32_t i = 0; i < 256; i++) {
            uint32_t c = i;
            for (int k = 0; k < 8; k++) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            table[0][i] = c;
        }
        for (uint32_t i = 0; i < 256; i++)
            for (int t = 1; t < 8; t++) table[t][i] = (table[t - 1][i] >> 8) ^ table[0][table[t - 1][i] & 0xFF];
    });
    uint32_t c = 0xFFFFFFFFu;
    while (n >= 8) {   // slicing-by-8
        const uint32_t lo = (p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24)) ^ c;
        c = table[7][lo & 0xFF] ^ table[6][(lo >> 8) & 0xFF] ^ table[5][(lo >> 16) & 0xFF] ^ table[4][lo >> 24] ^ table[3][p[4]] ^
            table[2][p[5]] ^ table[1][p[6]] ^ table[0][p[7]];
        p += 8;
        n -= 8;
    }
    while (n--) c = table[0][(c ^ *p++) & 0xFF] ^ (c >> 8);
    return c ^ 0xFFFFFFFFu;
}

int32_t pisces_hip_bgzf_inflate(PiscesHip* h, const uint8_t* file, int64_t n_bytes, const PiscesBgzfBlock* blocks, int64_t n_blocks,
                                uint8_t* out, int64_t out_capacity, int32_t check_crc, float* kernel_ms)
{
    return abi_guard<int32_t>(h, [&]() -> int32_t {
    if (!h) return PISCES_E_INVALID_ARG;
    if (!file || n_bytes <= 0 || n_blocks < 0 || (n_blocks > 0 && !blocks) || out_capacity < 0) return fail(h, PISCES_E_INVALID_ARG, "bgzf_inflate: bad arguments");
    if (kernel_ms) *kernel_ms = 0.f;
    if (n_blocks == 0) return PISCES_OK;
    int64_t out_bytes = 0;
    for (int64_t i = 0; i < n_blocks; i++) {
        const PiscesBgzfBlock& b = blocks[i];
        if (b.in_offset < 0 || b.in_length < 0 || b.in_length > 65536 || b.in_offset + b.in_length > n_bytes || b.out_offset < 0 ||
            b.out_length < 0 || b.out_length > 65536 || b.out_offset + b.out_length > out_capacity)   // BgzfCommon.MaxBlockSize both ways
            return fail(h, PISCES_E_INVALID_ARG, "bgzf_inflate: block " + std::to_string(i) + " lies outside the file bytes or the output buffer");
        out_bytes = std::max(out_bytes, b.out_offset + b.out_length);
    }
    if (out_bytes > 0 && !out) return fail(h, PISCES_E_INVALID_ARG, "bgzf_inflate: bad arguments");
    PISCES_HIP_CHECK(h, hipSetDevice(h->device));
    DeviceBuf<uint8_t> d_in, d_out;
    DeviceBuf<PiscesBgzfBlock> d_blocks;
    DeviceBuf<int32_t> d_status;
    PISCES_HIP_CHECK(h, d_in.reserve((size_t)n_bytes + kInWindow + 32));   // the bit reader's LDS window is filled in whole: up to a window past a block's payload
    PISCES_HIP_CHECK(h, hipMemsetAsync(d_in.p + n_bytes, 0, 16, h->stream));
    PISCES_HIP_CHECK(h, d_out.reserve((size_t)std::max<int64_t>(out_bytes, 1)));
    PISCES_HIP_CHECK(h, d_blocks.reserve((size_t)n_blocks));
    PISCES_HIP_CHECK(h, d_status.reserve((size_t)n_blocks));
    PISCES_HIP_CHECK(h, hipMemcpyAsync(d_in.p, file, (size_t)n_bytes, hipMemcpyHostToDevice, h->stream));
    PISCES_HIP_CHECK(h, hipMemcpyAsync(d_blocks.p, blocks, (size_t)n_blocks * sizeof(PiscesBgzfBlock), hipMemcpyHostToDevice, h->stream));
    hipExtLaunchKernelGGL(bgzf_inflate_kernel, dim3((unsigned)n_blocks), dim3(64), 0u, h->stream, h->ev0, h->ev1, 0u,
                          (const uint8_t*)d_in.p, (const PiscesBgzfBlock*)d_blocks.p, n_blocks, d_out.p, d_status.p);
    PISCES_HIP_CHECK(h, hipGetLastError());
    std::vector<int32_t> status((size_t)n_blocks);
    PISCES_HIP_CHECK(h, hipMemcpyAsync(status.data(), d_status.p, status.size() * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
    if (out_bytes > 0) PISCES_HIP_CHECK(h, hipMemcpyAsync(out, d_out.p, (size_t)out_bytes, hipMemcpyDeviceToHost, h->stream));
    PISCES_HIP_CHECK(h, hipStreamSynchronize(h->stream));
    if (kernel_ms) PISCES_HIP_CHECK(h, hipEventElapsedTime(kernel_ms, h->ev0, h->ev1));
    d_in.release(); d_out.release(); d_blocks.release(); d_status.release();
    for (int64_t i = 0; i < n_blocks; i++)
        if (status[(size_t)i] != 0)
            return fail(h, PISCES_E_INVALID_ARG, "bgzf_inflate: block " + std::to_string(i) + " is not a valid DEFLATE stream of its ISIZE (code " +
                                                     std::to_string(status[(size_t)i]) + ")");
    if (check_crc) {
        // blocks are independent: a few host threads share them for large tables (slicing-by-8 runs at ~2 GB/s per core)
        (void)crc32_of(out, 0);   // the tables, once, before any thread needs them
        const int n_threads = n_blocks >= 256 ? (int)std::min<unsigned>(8u, std::max(1u, std::thread::hardware_concurrency())) : 1;
        std::atomic<int64_t> first_bad(n_blocks);
        auto check = [&](int w) {
            for (int64_t i = w; i < n_blocks; i += n_threads)
                if (crc32_of(out + blocks[i].out_offset, (size_t)blocks[i].out_length) != blocks[i].crc32) {
                    int64_t cur = first_bad.load();
                    while (i < cur && !first_bad.compare_exchange_weak(cur, i)) {}
                }
        };
        std::vector<std::thread> pool;
        for (int w = 1; w < n_threads; w++) pool.emplace_back(check, w);
        check(0);
        for (auto& t : pool) t.join();
        if (first_bad.load() < n_blocks)
            return fail(h, PISCES_E_INVALID_ARG, "bgzf_inflate: CRC-32 mismatch in block " + std::to_string(first_bad.load()));
    }
    return PISCES_OK;
    });
}

// ---- BAM bytes -> read batch on the device (row f4): only the compressed file crosses PCIe -------------------------------
int32_t pisces_hip_bam_decode(PiscesHip* h, const uint8_t* file, int64_t n_bytes, const PiscesBgzfBlock* blocks, int64_t n_blocks, int32_t ref_id,
                              int32_t min_map_quality, int32_t skip_duplicates, int32_t only_proper_pairs, int64_t counts[4])
{
    return abi_guard<int32_t>(h, [&]() -> int32_t {
    if (!h) return PISCES_E_INVALID_ARG;
    if (!file || n_bytes <= 0 || n_blocks <= 0 || !blocks) return fail(h, PISCES_E_INVALID_ARG, "bam_decode: bad arguments");
    h->bam.valid = false;
    int64_t out_bytes = 0;
    for (int64_t i = 0; i < n_blocks; i++) {
        const PiscesBgzfBlock& b = blocks[i];
        if (b.in_offset < 0 || b.in_length < 0 || b.in_length > 65536 || b.in_offset + b.in_length > n_bytes || b.out_offset < 0 ||
            b.out_length < 0 || b.out_length > 65536)
            return fail(h, PISCES_E_INVALID_ARG, "bam_decode: block " + std::to_string(i) + " lies outside the file bytes");
        out_bytes = std::max(out_bytes, b.out_offset + b.out_length);
    }
    if (out_bytes <= 0 || out_bytes > 0x7FFFFFFF00ll) return fail(h, PISCES_E_INVALID_ARG, "bam_decode: empty or oversized stream");
    PISCES_HIP_CHECK(h, hipSetDevice(h->device));
    auto& B = h->bam;
    PISCES_HIP_CHECK(h, B.d_file.reserve((size_t)n_bytes + kInWindow + 32));
    PISCES_HIP_CHECK(h, B.d_stream.reserve((size_t)out_bytes + 16));
    PISCES_HIP_CHECK(h, B.d_blocks.reserve((size_t)n_blocks));
    PISCES_HIP_CHECK(h, B.d_status.reserve((size_t)n_blocks));
    PISCES_HIP_CHECK(h, hipMemsetAsync(B.d_file.p + n_bytes, 0, 16, h->stream));
    PISCES_HIP_CHECK(h, hipMemcpyAsync(B.d_file.p, file, (size_t)n_bytes, hipMemcpyHostToDevice, h->stream));
    PISCES_HIP_CHECK(h, hipMemcpyAsync(B.d_blocks.p, blocks, (size_t)n_blocks * sizeof(PiscesBgzfBlock), hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(bgzf_inflate_kernel, dim3((unsigned)n_blocks), dim3(64), 0, h->stream, (const uint8_t*)B.d_file.p,
                       (const PiscesBgzfBlock*)B.d_blocks.p, n_blocks, B.d_stream.p, B.d_status.p);
    // record boundaries without a serial pass over the bytes
    const int64_t n_chunks = (out_bytes + kBamChunk - 1) / kBamChunk;
    PISCES_HIP_CHECK(h, B.d_exits.reserve((size_t)out_bytes));
    PISCES_HIP_CHECK(h, B.d_header.reserve(4));
    PISCES_HIP_CHECK(h, B.d_entry.reserve((size_t)n_chunks));
    PISCES_HIP_CHECK(h, B.d_bstatus.reserve(4));
    PISCES_HIP_CHECK(h, B.d_n_reads.reserve((size_t)n_chunks + 1));
    PISCES_HIP_CHECK(h, B.d_n_ops.reserve((size_t)n_chunks + 1));
    PISCES_HIP_CHECK(h, B.d_n_bases.reserve((size_t)n_chunks + 1));
    PISCES_HIP_CHECK(h, B.d_n_skipped.reserve((size_t)n_chunks + 1));
    PISCES_HIP_CHECK(h, hipMemsetAsync(B.d_bstatus.p, 0, 4 * sizeof(int32_t), h->stream));
    const BamFilter F = {ref_id, min_map_quality, skip_duplicates, only_proper_pairs, h->cfg.min_base_call_quality};
    hipLaunchKernelGGL(bam_header_kernel, dim3(1), dim3(1), 0, h->stream, (const uint8_t*)B.d_stream.p, out_bytes, B.d_header.p);
    hipLaunchKernelGGL(bam_chain_kernel, dim3((unsigned)n_chunks), dim3(1024), 0, h->stream, (const uint8_t*)B.d_stream.p, out_bytes, B.d_exits.p);
    hipLaunchKernelGGL(bam_entry_kernel, dim3(1), dim3(1), 0, h->stream, (const uint16_t*)B.d_exits.p, out_bytes, (const long long*)B.d_header.p,
                       n_chunks, B.d_entry.p, B.d_bstatus.p);
    hipLaunchKernelGGL(bam_count_kernel, dim3((unsigned)n_chunks), dim3(64), 0, h->stream, (const uint8_t*)B.d_stream.p, out_bytes,
                       (const long long*)B.d_entry.p, F, B.d_n_reads.p, B.d_n_ops.p, B.d_n_bases.p, B.d_n_skipped.p);
    hipLaunchKernelGGL(bam_scan3_kernel, dim3(1), dim3(1024), 0, h->stream, B.d_n_reads.p, B.d_n_ops.p, B.d_n_bases.p, (int32_t)n_chunks);
    PISCES_HIP_CHECK(h, hipGetLastError());
    std::vector<int32_t> status((size_t)n_blocks), skipped((size_t)n_chunks);
    int32_t totals[3] = {0, 0, 0}, bstatus[4] = {0, 0, 0, 0};
    PISCES_HIP_CHECK(h, hipMemcpyAsync(status.data(), B.d_status.p, status.size() * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
    PISCES_HIP_CHECK(h, hipMemcpyAsync(skipped.data(), B.d_n_skipped.p, skipped.size() * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
    PISCES_HIP_CHECK(h, hipMemcpyAsync(&totals[0], B.d_n_reads.p + n_chunks, sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
    PISCES_HIP_CHECK(h, hipMemcpyAsync(&totals[1], B.d_n_ops.p + n_chunks, sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
    PISCES_HIP_CHECK(h, hipMemcpyAsync(&totals[2], B.d_n_bases.p + n_chunks, sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
    PISCES_HIP_CHECK(h, hipMemcpyAsync(bstatus, B.d_bstatus.p, sizeof(bstatus), hipMemcpyDeviceToHost, h->stream));
    PISCES_HIP_CHECK(h, hipStreamSynchronize(h->stream));
    for (int64_t i = 0; i < n_blocks; i++)
        if (status[(size_t)i] != 0)
            return fail(h, PISCES_E_INVALID_ARG, "bam_decode: block " + std::to_string(i) + " is not a valid DEFLATE stream of its ISIZE");
    if (bstatus[0] == 1) return fail(h, PISCES_E_INVALID_ARG, "bam_decode: not a BAM stream (magic / header)");
    if (bstatus[0] != 0)
        return fail(h, PISCES_E_INVALID_ARG, "bam_decode: the record chain breaks in chunk " + std::to_string(bstatus[1]) +
                                                 " (corrupt block_size, or a record longer than 32 KiB)");
    B.n_reads = totals[0]; B.n_ops = totals[1]; B.n_bases = totals[2];
    B.n_skipped = 0;
    for (int32_t v : skipped) B.n_skipped += v;
    B.min_bq = h->cfg.min_base_call_quality;
    const size_t nr = (size_t)B.n_reads, no = (size_t)B.n_ops, nb = (size_t)B.n_bases;
    PISCES_HIP_CHECK(h, B.position.reserve(nr + 1)); PISCES_HIP_CHECK(h, B.flags.reserve(nr + 1));
    PISCES_HIP_CHECK(h, B.cigar_offset.reserve(nr + 1)); PISCES_HIP_CHECK(h, B.seq_offset.reserve(nr + 1));
    PISCES_HIP_CHECK(h, B.read_quality.reserve(nr + 1));
    PISCES_HIP_CHECK(h, B.cigar_op.reserve(no + 1)); PISCES_HIP_CHECK(h, B.cigar_len.reserve(no + 1)); PISCES_HIP_CHECK(h, B.op_quality.reserve(no + 1));
    PISCES_HIP_CHECK(h, B.bases.reserve(nb + 16)); PISCES_HIP_CHECK(h, B.quals.reserve(nb + 16));
    if (nr > 0)
        hipLaunchKernelGGL(bam_decode_kernel, dim3((unsigned)n_chunks), dim3(64), 0, h->stream, (const uint8_t*)B.d_stream.p, out_bytes,
                           (const long long*)B.d_entry.p, F, (const int32_t*)B.d_n_reads.p, (const int32_t*)B.d_n_ops.p, (const int32_t*)B.d_n_bases.p,
                           B.position.p, B.flags.p, B.cigar_offset.p, B.cigar_op.p, B.cigar_len.p, B.seq_offset.p, B.bases.p, B.quals.p,
                           B.op_quality.p, B.read_quality.p);
    // the closing offsets
    const int32_t end_ops = (int32_t)no, end_bases = (int32_t)nb;
    PISCES_HIP_CHECK(h, hipMemcpyAsync(B.cigar_offset.p + nr, &end_ops, sizeof(int32_t), hipMemcpyHostToDevice, h->stream));
    PISCES_HIP_CHECK(h, hipMemcpyAsync(B.seq_offset.p + nr, &end_bases, sizeof(int32_t), hipMemcpyHostToDevice, h->stream));
    PISCES_HIP_CHECK(h, hipGetLastError());
    PISCES_HIP_CHECK(h, hipStreamSynchronize(h->stream));
    B.valid = true;
    if (counts) { counts[0] = B.n_reads; counts[1] = B.n_skipped; counts[2] = B.n_ops; counts[3] = B.n_bases; }
    return PISCES_OK;
    });
}

int32_t pisces_hip_bam_fetch(PiscesHip* h, int32_t* position, uint8_t* flags, int32_t* cigar_offset, uint8_t* cigar_op, uint32_t* cigar_len,
                             int32_t* seq_offset, uint8_t* bases, uint8_t* quals)
{
    return abi_guard<int32_t>(h, [&]() -> int32_t {
    if (!h) return PISCES_E_INVALID_ARG;
    if (!h->bam.valid) return fail(h, PISCES_E_STATE, "bam_fetch: no decoded batch (pisces_hip_bam_decode first)");
    auto& B = h->bam;
    const size_t nr = (size_t)B.n_reads, no = (size_t)B.n_ops, nb = (size_t)B.n_bases;
    PISCES_HIP_CHECK(h, hipSetDevice(h->device));
    auto down = [&](void* dst, const void* src, size_t bytes) -> hipError_t {
        return (dst && bytes) ? hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, h->stream) : hipSuccess;
    };
    PISCES_HIP_CHECK(h, down(position, B.position.p, nr * 4));
    PISCES_HIP_CHECK(h, down(flags, B.flags.p, nr));
    PISCES_HIP_CHECK(h, down(cigar_offset, B.cigar_offset.p, (nr + 1) * 4));
    PISCES_HIP_CHECK(h, down(cigar_op, B.cigar_op.p, no));
    PISCES_HIP_CHECK(h, down(cigar_len, B.cigar_len.p, no * 4));
    PISCES_HIP_CHECK(h, down(seq_offset, B.seq_offset.p, (nr + 1) * 4));
    PISCES_HIP_CHECK(h, down(bases, B.bases.p, nb));
    PISCES_HIP_CHECK(h, down(quals, B.quals.p, nb));
    PISCES_HIP_CHECK(h, hipStreamSynchronize(h->stream));
    return PISCES_OK;
    });
}

// IStateManager.AddAlleleCounts + FindCandidates for the decoded batch: the bases and qualities stay on the device; the host sees
// only positions and CIGARs (about 20 bytes per read), from which it makes what pisces_hip_add_reads makes from its own pass
// (log slots, candidate-record slots, the blocks every read touches).
int32_t pisces_hip_add_decoded_reads(PiscesHip* h)
{
    return abi_guard<int32_t>(h, [&]() -> int32_t {
    if (!h) return PISCES_E_INVALID_ARG;
    if (!h->bam.valid) return fail(h, PISCES_E_STATE, "add_decoded_reads: no decoded batch (pisces_hip_bam_decode first)");
    auto& B = h->bam;
    if (B.min_bq != h->cfg.min_base_call_quality) return fail(h, PISCES_E_STATE, "add_decoded_reads: decoded with another minimum base quality");
    { int32_t rcp = refuse_while_batch_is_open(h, "add_decoded_reads"); if (rcp) return rcp; }
    const int32_t nr = (int32_t)B.n_reads;
    if (nr == 0) return PISCES_OK;
    PISCES_HIP_CHECK(h, hipSetDevice(h->device));
    { int32_t rcf = consume_found(h); if (rcf) return rcf; }
    const size_t no = (size_t)B.n_ops;
    std::vector<int32_t> position((size_t)nr), coff((size_t)nr + 1), soff((size_t)nr + 1);
    std::vector<uint8_t> cop(no), opq(no), rq((size_t)nr);
    std::vector<uint32_t> clen(no);
    PISCES_HIP_CHECK(h, hipMemcpyAsync(position.data(), B.position.p, (size_t)nr * 4, hipMemcpyDeviceToHost, h->stream));
    PISCES_HIP_CHECK(h, hipMemcpyAsync(coff.data(), B.cigar_offset.p, ((size_t)nr + 1) * 4, hipMemcpyDeviceToHost, h->stream));
    PISCES_HIP_CHECK(h, hipMemcpyAsync(soff.data(), B.seq_offset.p, ((size_t)nr + 1) * 4, hipMemcpyDeviceToHost, h->stream));
    PISCES_HIP_CHECK(h, hipMemcpyAsync(cop.data(), B.cigar_op.p, no, hipMemcpyDeviceToHost, h->stream));
    PISCES_HIP_CHECK(h, hipMemcpyAsync(clen.data(), B.cigar_len.p, no * 4, hipMemcpyDeviceToHost, h->stream));
    PISCES_HIP_CHECK(h, hipMemcpyAsync(opq.data(), B.op_quality.p, no, hipMemcpyDeviceToHost, h->stream));
    PISCES_HIP_CHECK(h, hipMemcpyAsync(rq.data(), B.read_quality.p, (size_t)nr, hipMemcpyDeviceToHost, h->stream));
    PISCES_HIP_CHECK(h, hipStreamSynchronize(h->stream));
    auto op_ref = [](uint8_t t) { return t == 'M' || t == 'D' || t == 'N' || t == '=' || t == 'X'; };
    auto op_read = [](uint8_t t) { return t == 'M' || t == 'I' || t == 'S' || t == '=' || t == 'X'; };
    const bool find_on_device = !h->h_ref.empty();
    std::vector<long long>& slots = h->read_slots;
    std::vector<int32_t>& fslots = h->found_slots_host;
    slots.resize((size_t)nr + 1);
    fslots.assign((size_t)nr + 1, 0);
    int64_t ub = 0, found_slots = 0, found_pool = 0;
    for (int32_t i = 0; i < nr; i++) {
        const int c0 = coff[(size_t)i], nc = coff[(size_t)i + 1] - c0, read_len = soff[(size_t)i + 1] - soff[(size_t)i];
        const uint8_t* ops = cop.data() + c0;
        const uint32_t* lens = clen.data() + c0;
        const uint8_t* oq = opq.data() + c0;
        slots[(size_t)i] = (long long)(h->log_ub + ub);
        fslots[(size_t)i] = (int32_t)found_slots;
        if (position[(size_t)i] <= 0) return fail(h, PISCES_E_INVALID_ARG, "Position must be greater than 0.");
        int64_t read_span = 0, ref_span = 0;
        for (int c = 0; c < nc; c++) {
            if (lens[c] > 0x0FFFFFFFu) return fail(h, PISCES_E_INVALID_ARG, "add_decoded_reads: CIGAR operation longer than 2^28 - 1");
            if (op_read(ops[c])) read_span += lens[c];
            if (op_ref(ops[c])) ref_span += lens[c];
            if (find_on_device && !h->cfg.call_mnvs) {
                if (ops[c] == 'I' || ops[c] == 'D') found_slots++;
                if (ops[c] == 'I' && lens[c] > (uint32_t)kFoundInline) found_pool += lens[c];
            }
        }
        if (read_span > read_len) return fail(h, PISCES_E_INVALID_ARG, "add_decoded_reads: CIGAR does not match the read");
        if ((int64_t)position[(size_t)i] + ref_span > 0x7FFFFFFFll) return fail(h, PISCES_E_INVALID_ARG, "add_decoded_reads: read runs past position 2^31 - 1");
        ub += ref_span;
        // the blocks the read touches (GetBlock for every position that receives a count, RegionStateManager.cs:361-383), as in
        // pisces_hip_add_reads, with CheckDeletionQuality taken from the bits the decode kernel left
        auto touch = [&](int64_t from, int64_t to) {
            if (to < 1) return;
            if (from < 1) from = 1;
            for (int32_t k = block_key(h, (int32_t)from); k <= block_key(h, (int32_t)to); k++) (void)get_block(h, (k - 1) * h->cfg.block_size + 1);
        };
        int64_t rp = position[(size_t)i], last_mapped = (int64_t)position[(size_t)i] - 1;
        int ri = 0;
        for (int c = 0; c < nc; c++) {
            const uint8_t t = ops[c];
            const int64_t len = lens[c];
            if (op_read(t) && op_ref(t) && len > 0) {
                if (rp > last_mapped + 1 && ri < read_len && oq[c]) touch(last_mapped + 1, rp - 1);
                touch(rp, rp + len - 1);
                last_mapped = rp + len - 1;
            }
            if (op_ref(t)) rp += len;
            if (op_read(t)) ri += (int)len;
        }
        const bool ends_del = nc >= 1 && ops[nc - 1] == 'D';
        const bool ends_del_soft = nc >= 2 && ops[nc - 2] == 'D' && ops[nc - 1] == 'S';
        if (ends_del && read_len > 0 && rq[(size_t)i]) touch(last_mapped + 1, last_mapped + lens[nc - 1]);
        if (ends_del_soft) {
            const int idx = read_len - (int)lens[nc - 1];
            if (idx >= 0 && idx < read_len && oq[nc - 1]) touch(last_mapped + 1, last_mapped + lens[nc - 2]);
        }
        h->stats[2] += 1;
    }
    slots[(size_t)nr] = (long long)(h->log_ub + ub);
    fslots[(size_t)nr] = (int32_t)found_slots;
    if (found_slots > 0x7FFFFFF0ll || found_pool > 0x7FFFFFF0ll) return fail(h, PISCES_E_INVALID_ARG, "add_decoded_reads: too many insertions / deletions in one batch");
    int32_t rc = log_reserve(h, ub);
    if (rc) return rc;
    PISCES_HIP_CHECK(h, B.d_slots.reserve((size_t)nr + 1));
    PISCES_HIP_CHECK(h, B.d_fslots.reserve((size_t)nr + 1));
    PISCES_HIP_CHECK(h, hipMemcpyAsync(B.d_slots.p, slots.data(), ((size_t)nr + 1) * sizeof(long long), hipMemcpyHostToDevice, h->stream));
    PISCES_HIP_CHECK(h, hipMemcpyAsync(B.d_fslots.p, fslots.data(), ((size_t)nr + 1) * sizeof(int32_t), hipMemcpyHostToDevice, h->stream));
    DevReadBatch db;
    db.position = B.position.p; db.flags = B.flags.p; db.cigar_offset = B.cigar_offset.p; db.cigar_op = B.cigar_op.p; db.cigar_len = B.cigar_len.p;
    db.seq_offset = B.seq_offset.p; db.bases = B.bases.p; db.quals = B.quals.p; db.dirs = nullptr; db.n_reads = nr;
    const int c = h->log_cur;
    hipLaunchKernelGGL(expand_reads_kernel, dim3((unsigned)((nr + 3) / 4)), dim3(256), 0, h->stream, db, (const long long*)B.d_slots.p,
                       h->cfg.min_base_call_quality, h->d_log_pos[c].p, h->d_log_tup[c].p, h->d_log_n.p + 2);
    PISCES_HIP_CHECK(h, hipGetLastError());
    if (find_on_device && (h->cfg.call_mnvs || found_slots > 0)) {
        int32_t rcd = enqueue_candidate_discovery(h, db, nullptr, nr, (const int32_t*)B.d_fslots.p, found_slots, found_pool);
        if (rcd) return rcd;
    }
    // slots / fslots on the host are reused by the next call: the copies above must have left first
    PISCES_HIP_CHECK(h, hipStreamSynchronize(h->stream));
    h->log_ub += ub;
    return PISCES_OK;
    });
}

// ---- RCCL, bound at run time: the library itself does not link librccl (a single-GPU host never loads it) ----
struct RcclId { char internal[PISCES_COMM_ID_BYTES]; };   // ncclUniqueId
namespace {
struct Rccl {
    void* lib = nullptr;
    int (*GetUniqueId)(void*) = nullptr;
    int (*CommInitRank)(void**, int, /* ncclUniqueId by value */ RcclId, int) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};
}  // namespace
static Rccl* rccl()
{
    static std::mutex mu;
    static Rccl r;
    std::lock_guard<std::mutex> lock(mu);
    if (r.lib) return &r;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
        r.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (r.lib) break;
    }
    if (!r.lib) return nullptr;
    r.GetUniqueId = (int (*)(void*))dlsym(r.lib, "ncclGetUniqueId");
    r.CommInitRank = (int (*)(void**, int, RcclId, int))dlsym(r.lib, "ncclCommInitRank");
    r.AllReduce = (int (*)(const void*, void*, size_t, int, int, void*, hipStream_t))dlsym(r.lib, "ncclAllReduce");
    r.CommDestroy = (int (*)(void*))dlsym(r.lib, "ncclCommDestroy");
    r.GetErrorString = (const char* (*)(int))dlsym(r.lib, "ncclGetErrorString");
    if (!r.GetUniqueId || !r.CommInitRank || !r.AllReduce || !r.CommDestroy) { dlclose(r.lib); r.lib = nullptr; return nullptr; }
    return &r;
}
static std::string rccl_error(Rccl* r, int code)
{
    return std::string("RCCL: ") + ((r && r->GetErrorString) ? r->GetErrorString(code) : "error") + " (" + std::to_string(code) + ")";
}

int32_t pisces_hip_comm_unique_id(uint8_t* id_out, int32_t capacity)
{
    return abi_guard<int32_t>((PiscesHip*)nullptr, [&]() -> int32_t {
    if (!id_out || capacity < PISCES_COMM_ID_BYTES) return fail(nullptr, PISCES_E_INVALID_ARG, "comm_unique_id: the id needs 128 bytes");
    Rccl* r = rccl();
    if (!r) return fail(nullptr, PISCES_E_DEVICE, "comm_unique_id: librccl could not be loaded");
    RcclId id;
    std::memset(&id, 0, sizeof(id));
    const int rc = r->GetUniqueId(&id);
    if (rc != 0) return fail(nullptr, PISCES_E_DEVICE, rccl_error(r, rc));
    std::memcpy(id_out, id.internal, PISCES_COMM_ID_BYTES);
    return PISCES_OK;
    });
}

int32_t pisces_hip_comm_init(PiscesHip* h, const uint8_t* id, int32_t rank, int32_t world)
{
    return abi_guard<int32_t>(h, [&]() -> int32_t {
    if (!h) return PISCES_E_INVALID_ARG;
    if (!id || world < 1 || rank < 0 || rank >= world) return fail(h, PISCES_E_INVALID_ARG, "comm_init: rank / world out of range");
    if (h->comm) return fail(h, PISCES_E_STATE, "comm_init: the handle already has a communicator");
    Rccl* r = rccl();
    if (!r) return fail(h, PISCES_E_DEVICE, "comm_init: librccl could not be loaded");
    PISCES_HIP_CHECK(h, hipSetDevice(h->device));
    PISCES_HIP_CHECK(h, h->d_summary.reserve(4));
    RcclId uid;
    std::memcpy(uid.internal, id, PISCES_COMM_ID_BYTES);
    void* comm = nullptr;
    const int rc = r->CommInitRank(&comm, world, uid, rank);
    if (rc != 0) return fail(h, PISCES_E_DEVICE, rccl_error(r, rc));
    h->comm = comm;
    h->comm_world = world;
    return PISCES_OK;
    });
}

int32_t pisces_hip_reduce_summary(PiscesHip* h, int64_t inout[4])
{
    return abi_guard<int32_t>(h, [&]() -> int32_t {
    if (!h || !inout) return PISCES_E_INVALID_ARG;
    if (!h->comm) return PISCES_OK;   // one shard: the sum is the value
    Rccl* r = rccl();
    if (!r) return fail(h, PISCES_E_DEVICE, "reduce_summary: librccl could not be loaded");
    PISCES_HIP_CHECK(h, hipSetDevice(h->device));
    long long v[4] = {inout[0], inout[1], inout[2], inout[3]};
    PISCES_HIP_CHECK(h, hipMemcpyAsync(h->d_summary.p, v, sizeof(v), hipMemcpyHostToDevice, h->stream));
    const int rc = r->AllReduce(h->d_summary.p, h->d_summary.p, 4, /* ncclInt64 */ 4, /* ncclSum */ 0, h->comm, h->stream);
    if (rc != 0) return fail(h, PISCES_E_DEVICE, rccl_error(r, rc));
    PISCES_HIP_CHECK(h, hipMemcpyAsync(v, h->d_summary.p, sizeof(v), hipMemcpyDeviceToHost, h->stream));
    PISCES_HIP_CHECK(h, hipStreamSynchronize(h->stream));
    for (int i = 0; i < 4; i++) inout[i] = v[i];
    return PISCES_OK;
    });
}

int32_t pisces_hip_comm_destroy(PiscesHip* h)
{
    return abi_guard<int32_t>(h, [&]() -> int32_t {
    if (!h) return PISCES_E_INVALID_ARG;
    if (!h->comm) return PISCES_OK;
    Rccl* r = rccl();
    if (r) (void)r->CommDestroy(h->comm);
    h->comm = nullptr;
    h->comm_world = 1;
    return PISCES_OK;
    });
}

int32_t pisces_hip_synchronize(PiscesHip* h)
{
    return abi_guard<int32_t>(h, [&]() -> int32_t {
    if (!h) return PISCES_E_INVALID_ARG;
    PISCES_HIP_CHECK(h, hipSetDevice(h->device));
    PISCES_HIP_CHECK(h, hipStreamSynchronize(h->stream));
    for (int k = 0; k < PiscesHip::kLanes; k++)
        if (h->lane[k]) PISCES_HIP_CHECK(h, hipStreamSynchronize(h->lane[k]));
    return PISCES_OK;
    });
}

int32_t pisces_hip_last_kernel_ms(PiscesHip* h, float* ms)
{
    return abi_guard<int32_t>(h, [&]() -> int32_t {
    if (!h || !ms) return PISCES_E_INVALID_ARG;
    if (h->timing <= 0 || h->ring_used == 0)
        return fail(h, PISCES_E_STATE, "last_kernel_ms: no timed launch (pisces_hip_set_timing first)");
    PISCES_HIP_CHECK(h, hipSetDevice(h->device));
    const size_t slot = (size_t)((h->ring_used - 1) % kTimingRing);
    PISCES_HIP_CHECK(h, hipEventSynchronize(h->ring[2 * slot + 1]));
    PISCES_HIP_CHECK(h, hipEventElapsedTime(ms, h->ring[2 * slot], h->ring[2 * slot + 1]));
    return PISCES_OK;
    });
}

}  // extern "C"
