// pisces_hip.hip — C ABI of libpisceship.so (include/pisces_hip.h): handle, device buffers,
// kernel launches, and the host mirror of the reference's streaming protocol
// (IStateManager.AddAlleleCounts / GetCandidatesToProcess / DoneProcessing around IAlleleCaller.Call,
// src/exe/Pisces/Logic/SmallVariantCaller.cs:79-189).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <dlfcn.h>
#include <link.h>

#include <algorithm>
#include <array>
#include <cctype>
#include <memory>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <ctime>
#include <map>
#include <set>
#include <string>
#include <atomic>
#include <chrono>
#include <mutex>
#include <thread>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "../../include/pisces_hip.h"
#include "expander.h"
#include "diploid.h"
#include "finder.h"
#include "kernels.hip.h"
#include "stream_kernels.hip.h"
#include "store_kernels.hip.h"
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
    // hash of (position, category, ref, alt[, open ends]) -> index of the first candidate with that hash; cand_next chains the (rare)
    // candidates that share a hash.  (A std::string key per record was 200 ns of every candidate record the device sends back.)
    std::unordered_map<uint64_t, uint32_t> cand_index;
    std::vector<uint32_t> cand_next;
    int32_t max_allele_endpoint = 0;   // RegionState.MaxAlleleEndpoint
    std::vector<std::pair<int32_t, int32_t>> x_spans;   // positions of the X operations of the block's reads (MNV calling on, split form): dirty loci
    // MNV calling off: the bases of X and = operations that the allele counts hold and no SNV candidate stands for (finder_walk.h
    // kFoundUnwalked), by (position, read base): the flush calls the SNVs of their loci from the counts LESS these (surface_flush.inc.h)
    struct Unwalked { int32_t position; uint8_t alt; int32_t sup[3]; };
    std::vector<Unwalked> unwalked;
};


// ---- device / pinned memory kept across handles ------------------------------------------------------------------------------------
// A caller that cuts a genome into pieces makes and destroys a handle per piece (one per chromosome in the reference,
// BaseGenomeProcessor.cs:40-90; one per (contig, interval range) in BASELINE config 4): hipFree and hipHostFree synchronise the device
// and unpin pages — 30 ms of a 75 ms piece were pisces_hip_destroy, 10 more the first touch of freshly pinned staging memory.  What a
// handle held when it was destroyed is therefore kept (by device; up to PISCES_HIP_ALLOC_CACHE_MB of device memory, default 8192 or a
// sixteenth of the device's memory if that is less, and a
// quarter of that pinned) and handed to the next allocation of about that size.  Only pisces_hip_destroy puts memory there — it has
// waited for the handle's streams — a buffer that is outgrown in mid-life is freed for real, as before.  pisces_hip_trim_memory()
// gives everything back.
constexpr int kCacheDevices = 16;
struct AllocCache {
    std::mutex m;
    std::multimap<size_t, void*> free_dev[kCacheDevices], free_host;
    std::unordered_map<void*, std::pair<size_t, int>> live;   // every allocation made here: bytes, device (-1: pinned host)
    size_t cached_dev[kCacheDevices] = {0}, cached_host = 0, limit_dev = 0, limit_host = 0;
    bool configured = false;
    void configure()
    {
        if (configured) return;
        configured = true;
        const char* e = getenv("PISCES_HIP_ALLOC_CACHE_MB");
        long long mb = e ? atoll(e) : 8192;
        if (!e) {   // the default never holds back more than a sixteenth of the device: processes that share a GPU cannot drop each other's caches
            size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && total_b > 0) mb = std::min<long long>(mb, (long long)(total_b >> 24));
            else (void)hipGetLastError();
        }
        limit_dev = (size_t)std::max(0ll, mb) << 20;
        limit_host = limit_dev / 4;
    }
};
AllocCache& alloc_cache() { static AllocCache* c = new AllocCache(); return *c; }   // (never destroyed: a static's destructor would run after the runtime's)
thread_local bool tl_keep_freed = false;   // pisces_hip_destroy, after it has waited for the handle's streams

void* cache_take(std::multimap<size_t, void*>& m, size_t& cached, size_t bytes)
{
    auto it = m.lower_bound(bytes);
    if (it == m.end() || it->first > bytes + bytes / 2 + ((size_t)1 << 20)) return nullptr;
    void* p = it->second;
    cached -= it->first;
    m.erase(it);
    return p;
}
size_t cache_drop(AllocCache& c, int device /* -1 host, -2 everything */)
{
    size_t freed = 0;
    for (int d = 0; d < kCacheDevices; d++) {
        if (device != -2 && device != d) continue;
        for (auto& kv : c.free_dev[d]) { (void)hipFree(kv.second); c.live.erase(kv.second); freed += kv.first; }
        c.free_dev[d].clear();
        c.cached_dev[d] = 0;
    }
    if (device == -2 || device == -1) {
        for (auto& kv : c.free_host) { (void)hipHostFree(kv.second); c.live.erase(kv.second); freed += kv.first; }
        c.free_host.clear();
        c.cached_host = 0;
    }
    return freed;
}
hipError_t dev_alloc(void** out, size_t bytes)
{
    AllocCache& c = alloc_cache();
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lock(c.m);
    c.configure();
    if (bytes == 0) bytes = 1;
    if (dev >= 0 && dev < kCacheDevices) {
        if (void* p = cache_take(c.free_dev[dev], c.cached_dev[dev], bytes)) { *out = p; return hipSuccess; }
    }
    hipError_t e = hipMalloc(out, bytes);
    if (e != hipSuccess && dev >= 0 && dev < kCacheDevices && !c.free_dev[dev].empty()) {   // out of memory with memory set aside: give it back, once more
        (void)hipGetLastError();
        (void)cache_drop(c, dev);
        e = hipMalloc(out, bytes);
    }
    if (e == hipSuccess) c.live[*out] = {bytes, dev};
    return e;
}
void dev_free(void* p)
{
    if (!p) return;
    AllocCache& c = alloc_cache();
    {
        std::lock_guard<std::mutex> lock(c.m);
        auto it = c.live.find(p);
        if (it != c.live.end()) {
            const size_t bytes = it->second.first;
            const int dev = it->second.second;
            if (tl_keep_freed && dev >= 0 && dev < kCacheDevices && c.cached_dev[dev] + bytes <= c.limit_dev) {
                c.free_dev[dev].insert({bytes, p});
                c.cached_dev[dev] += bytes;
                return;
            }
            c.live.erase(it);
        }
    }
    (void)hipFree(p);
}
hipError_t host_alloc(void** out, size_t bytes)
{
    AllocCache& c = alloc_cache();
    std::lock_guard<std::mutex> lock(c.m);
    c.configure();
    if (bytes == 0) bytes = 1;
    if (void* p = cache_take(c.free_host, c.cached_host, bytes)) { *out = p; return hipSuccess; }
    hipError_t e = hipHostMalloc(out, bytes, hipHostMallocDefault);
    if (e != hipSuccess && !c.free_host.empty()) {
        (void)hipGetLastError();
        (void)cache_drop(c, -1);
        e = hipHostMalloc(out, bytes, hipHostMallocDefault);
    }
    if (e == hipSuccess) c.live[*out] = {bytes, -1};
    return e;
}
void host_free(void* p)
{
    if (!p) return;
    AllocCache& c = alloc_cache();
    {
        std::lock_guard<std::mutex> lock(c.m);
        auto it = c.live.find(p);
        if (it != c.live.end()) {
            const size_t bytes = it->second.first;
            if (tl_keep_freed && c.cached_host + bytes <= c.limit_host) {
                c.free_host.insert({bytes, p});
                c.cached_host += bytes;
                return;
            }
            c.live.erase(it);
        }
    }
    (void)hipHostFree(p);
}

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
        if (p) dev_free(p);
        p = nullptr;
        cap = 0;
        size_t want = n + n / 4 + 64;
        hipError_t e = dev_alloc((void**)&p, want * sizeof(T));
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
        hipError_t e = dev_alloc((void**)&fresh, want * sizeof(T));
        if (e != hipSuccess) return e;
        if (old && n_keep > 0) {
            e = hipMemcpyAsync(fresh, old, std::min(n_keep, cap) * sizeof(T), hipMemcpyDeviceToDevice, stream);
            if (e == hipSuccess) e = hipStreamSynchronize(stream);
        }
        if (old) dev_free(old);
        p = fresh;
        cap = want;
        return e;
    }
    void release()
    {
        if (p) dev_free(p);
        p = nullptr;
        cap = 0;
    }
    void swap(DeviceBuf& o)
    {
        std::swap(p, o.p);
        std::swap(cap, o.cap);
    }
};

// One segment of the read store (store_kernels.hip.h): reads in position order with their descriptors.  Three ways to own the bytes:
// a batch uploaded in one piece (`blob`, laid out as the staging buffer is; the views point into it), a decoded BAM batch (the decode's
// own arrays moved in), or the OPEN segment that small batches are appended to (its arrays grow).
struct ReadSegment {
    DeviceBuf<uint8_t> blob;
    DeviceBuf<uint8_t> bases, quals, dirs, cop;
    DeviceBuf<uint8_t> codes;    // one byte per base: low-quality << 5 | AlleleType << 2, made when the batch joins (encode_rows: what the flush kernel walks)
    DeviceBuf<uint32_t> clen;
    DeviceBuf<ReadDesc> desc;
    DeviceBuf<ReadExt> ext;
    DeviceBuf<ReadDesc> frag;    // one per CIGAR operation: what the flush kernel walks
    DeviceBuf<int32_t> grid;     // the position grid (store_kernels.hip.h grid_cells): first fragment per position (cell) from cell grid_base on
    int64_t grid_base = 0, grid_n = 0;
    bool grid_ok = false;        // every batch so far could extend it (known first position, not before grid_base, a sane span)
    int32_t* state = nullptr;    // its four state words on the device (a slot of PiscesHip::state_pool, zero when handed out)
    const uint8_t *v_bases = nullptr, *v_quals = nullptr, *v_dirs = nullptr, *v_cop = nullptr, *v_codes = nullptr;
    const uint32_t* v_clen = nullptr;
    int64_t n_reads = 0, n_bases = 0, n_ops = 0;
    int64_t n_floored = 0, n_floored_ops = 0;   // reads / CIGAR operations that were there at the last flush ...
    int32_t floor = 0;         // ... have their positions below this counted already (DoneProcessing of the blocks below it)
    int32_t max_key = 0;       // highest block any of its reads touches: the segment is dropped once no block up to it is left
    bool open = false;         // accepts appended batches
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
    // pisces_hip_set_chain_timing / pisces_hip_chain_time: the device time of reads -> records, as two spans on the handle's stream — [0, 1]
    // around everything an add of reads enqueues, [2, 3] from a flush's first kernel to its compacted records in HBM (the transfer to the
    // host comes behind [3])
    bool chain_timing = false;
    hipEvent_t ev_chain[4] = {nullptr, nullptr, nullptr, nullptr};
    bool chain_have[2] = {false, false};
    bool chain_enqueued_since = false;        // ... and something was enqueued behind the span's provisional end (recorded behind the add's launch)
    bool chain_add_open = false;              // the add in progress recorded its first event (a batch in device memory)
    int64_t ring_used = 0;
    int64_t launches_seen = 0;
    DeviceBuf<unsigned long long> d_totals;
    unsigned long long* h_totals = nullptr;   // pinned: pisces_hip_device_totals
    bool foreign_stream_used = false;         // a call_tiles launch went to a stream that is not the handle's since the last pisces_hip_device_totals
    bool foreign_stream_ever = false;         // ... ever (pisces_hip_destroy)
    // pisces_hip_set_exact_total_called: IAlleleCaller.TotalNumCalled with an interval set and MNV calling off counts the callable SNVs of the
    // loci OUTSIDE the intervals too (AlleleCaller.cs:109-131: IsCallable counts, ShouldReport comes after): a second launch of the flush
    // kernel over those loci of the flushed blocks, its records dropped, its n_called added
    bool exact_total_called = false;
    DeviceBuf<PiscesTile> d_tiles_x;
    DeviceBuf<PiscesTileResult> d_tr_x;
    DeviceBuf<PiscesCalledAllele> d_rec_x;
    DeviceBuf<int32_t> d_cnt_x, d_off_x;
    int32_t* h_cnt_x = nullptr;   // pinned: {records, called} of the counting launch
    bool tables_shareable = false;            // the memo tables are the default ones of the configuration: they outlive the handle (table_cache)
    bool poisoned = false;                    // the deferred half of a batch's candidate discovery failed after the batch was committed (finish_candidate_discovery)
    std::string poison_why;
    uint8_t* h_prep = nullptr;                // pinned: add_fused_kernel's PrepVerdict + the keys of the touched blocks
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
    int32_t* h_counts = nullptr;           // pinned: the anchor-resolved counts of a batch's blocks for the collapser / reallocator (a copy into
    size_t h_counts_cap = 0;               // pageable memory is staged by the runtime and blocks: 0.6 ms a flush of 0.8 MB)

    // ---- streaming state (RegionStateManager: _regionLookup, _lastUpToBlockKey) ----
    std::map<int32_t, BlockObs> blocks;   // key = GetBlockKey(position) (RegionStateManager.cs:385-391)
    int32_t last_block_key_cache = 0;
    BlockObs* last_block = nullptr;       // "performance improvement to remember last block" (:366)
    int32_t last_up_to_block_key = 0;
    std::unordered_map<int32_t, int32_t> gapped_mnv_ref;
    std::vector<HostCandidate> known_variants;   // the chromosome's known (prior) variants (pisces_hip_set_known_variants): the collapser's AnnotateKnown
    bool exclude_mnvs_from_collapsing = false;   // PiscesApplicationOptions.ExcludeMNVsFromCollapsing (pisces_hip_set_exclude_mnvs_from_collapsing)
    // forced genotyping alleles of this chromosome (pisces_hip_set_forced_alleles), in position order; the first n_forced_added have
    // been handed to the state as candidates (SmallVariantCaller.AddForcedAlleleAsCandidate)
    std::vector<HostCandidate> forced;
    size_t n_forced_added = 0;
    std::set<std::string> forced_keys;        // position|ref>alt (AlleleCaller.IsForcedAllele)
    std::set<int32_t> forced_positions;       // RegionState.CreateIntervalsFromAllels
    std::vector<std::pair<int32_t, int32_t>> intervals;   // sorted, disjoint [start, end]
    int32_t own_lo = 1, own_hi = 0x7FFFFFFF;              // pisces_hip_set_owned_range
    int64_t stats[4] = {0, 0, 0, 0};      // called, collapsed, reads processed, reads skipped
    bool in_flush_begin = false;
    double prof[24] = {0};                 // development (PISCES_HIP_HOST_PROFILE=1): host seconds by phase of a flush, printed when the handle goes
    bool prof_on = false;
    int64_t pcie[4] = {0, 0, 0, 0};          // pisces_hip_transfer_bytes: H2D reads / file bytes, D2H records, D2H candidate records, D2H counts
    double host_time[4] = {0, 0, 0, 0};   // pisces_hip_host_time: seconds in add_reads, in flush, of that waiting for the device; flushes

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
    // pisces_hip_flush_begin / pisces_hip_flush_end: a flush whose device work is in flight while the host goes on (add_reads of the
    // next batch), or whose results are ready and wait to be taken
    struct AsyncFlush {
        int state = 0;                         // 0 none, 1 in flight, 2 results ready
        hipEvent_t done = nullptr;
        bool dropped = false;                  // the log was compacted behind the calls: [kept, bound) of the new log are holes
        int64_t bound = 0;
        int32_t* hdr = nullptr;                // {records, called, kept (8 bytes)} in the pinned download buffer
        PiscesCalledAllele* hrec = nullptr;
        size_t spec = 0;                       // records that come back with the header; more only with a second copy
        const PiscesCalledAllele* data = nullptr;   // state 2: the results
        size_t n = 0;
        std::vector<PiscesCalledAllele> owned; // state 2, when the flush had to run synchronously (host-side candidates, genotypers, ...)
        std::vector<int32_t> owned_index;      // ... and what pisces_hip_flush_ex returns beside the records (pisces_hip_flush_end_ex)
        std::vector<PiscesCandidate> owned_cands;
        std::vector<uint8_t> owned_alleles;
        size_t n_cands = 0, n_allele_bytes = 0;
    } async;
    struct FlushView {                        // pisces_hip_flush_view: what the last flush handed out in place
        bool wanted = false;
        const PiscesCalledAllele* data = nullptr;
        size_t n = 0;
        std::vector<PiscesCalledAllele> rows;  // the merged rows of a batch with host-side candidates (otherwise data points into h_dl)
        std::vector<int32_t> index;
        std::vector<PiscesCandidate> cands;
        std::vector<uint8_t> alleles;
    } view;
    int64_t log_known_holes = 0;             // slots of the log that the last asynchronous drop left as holes (0 after any other drop)
    size_t staged_total = 0;                 // bytes pisces_hip_stage_reads laid out in the current staging buffer (0: nothing staged)
    uint8_t* h_dl = nullptr;                 // pinned download buffer of flush
    const PiscesCalledAllele* pending_view = nullptr;   // the pending records when they are the download buffer's as they came (no host-side
    size_t pending_view_n = 0;                          // candidates, genotyper or forced alleles to merge in): no copy into `pending`
    // pinned arena of the small uploads of a flush (bucket tables, tile geometry, gapped-MNV counts): a copy from pageable memory makes
    // the host wait until the stream has caught up, i.e. it serialises the flush's enqueueing with the device; from pinned memory it is
    // asynchronous.  Bump-allocated, rewound when the stream is known to be idle (every flush ends with a synchronisation).
    uint8_t* h_cand_dl = nullptr;            // pinned: records + IsCallable of a call_spanning_kernel pass
    size_t h_cand_dl_cap = 0;
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
    DeviceBuf<int32_t> d_merge_tab, d_merge_acc;   // found_merge_kernel: the table of group owners, the accumulators (kMergeAcc int32 a record)
    int device_merge = -1;                   // PISCES_HIP_DEVICE_MERGE: 0 the records come back one per read event, 1 merged whatever their number, -1 (default) merged from 2048 records up
    struct FoundPending {
        bool merged = false;                 // h holds DevMerged[misc[2]] (any order) instead of DevFound[n_slots]
        uint8_t* h = nullptr;                // pinned: DevFound[n_slots] (or room for DevMerged[n_slots]), then the pool bytes, then misc[3]
        size_t h_cap = 0;
        hipEvent_t done = nullptr;
        bool in_flight = false;
        int64_t n_slots = 0, pool_bytes = 0;
        std::vector<int32_t> order;          // consume_found: group of each first-arrival record index
        uint32_t batch = 0;                  // the batch's sequence number (arrival stamps)
        int32_t min_position = 0;            // lowest read position of the batch, 0 = unknown (a flush below it need not wait for the records)
        bool split = false;                  // the plain SNV groups went to the SNV store (misc[3] of them)
        bool split_counted = false;          // ... and a sweep of the store has counted them since (snv_ub is exact: nothing to correct)
        // MNV calling on: the walk is counted at the add (find_count + scan), the records are written (find_emit, merge, gather) by the next
        // entry that comes along — by then the totals that size their buffers have arrived and nobody waits for them
        bool counted_only = false;
        hipEvent_t counted = nullptr;
        long long* h_totals = nullptr;       // pinned: {record slots, pool bytes}
        DevReadBatch db;
        const uint8_t* d_deldirs = nullptr;
        int32_t nr = 0;
        FinderParams fp = {};                // the walk's parameters (both halves)
    } found;
    bool eqx_in_batch = false;               // the batch being added has X or = operations (set by its checks, read by enqueue_candidate_discovery)

    // MNV calling on, SPLIT FORM (surface_flush.inc.h): the fully anchored SNV groups of the read walk stay in device memory (the SNV store,
    // finder_kernels.hip.h) until their block is flushed; the tile kernels call SNVs from the allele counts everywhere but on the dirty loci
    bool mnv_split = false;
    // The read walk makes the SNV candidates (its M-operation half runs): MNV calling on — or off with a collapser whose thresholds can keep
    // an open-ended SNV and its twin apart (VariantCollapser.GetMatches: a match that is not the fully anchored twin must reach
    // CollapseFreqThreshold and more than CollapseFreqRatioThreshold times the candidate's frequency).  Then two candidates of one allele
    // are called on their own support each, which the allele counts do not know; with the default thresholds (0, 0.5) twins always join
    // and the SNVs of MNV calling off are the allele counts (the tile kernels call them; a read that maps ONE base — an SNV open on both
    // sides — is the exception that stays with the counts).
    bool snv_walk = false;
    DeviceBuf<SnvGroup> d_snv[2];
    int snv_cur = 0;
    DeviceBuf<unsigned int> d_snv_n;          // [0], [1]: groups in d_snv[0] / [1]; [2], [3]: a sweep's {selected, kept}
    int64_t snv_ub = 0;                       // groups in d_snv[snv_cur], an upper bound while a batch's gather is in flight
    DeviceBuf<SnvGroup> d_snv_sel;
    SnvGroup* h_snv_sel = nullptr;            // pinned: {selected, kept} counts (16 bytes), then the first kSnvSpec selected groups
    DeviceBuf<uint32_t> d_dirty;
    std::vector<uint32_t> dirty_host;
    std::vector<HostCandidate> split_selected;   // the SNV groups of the current batch's dirty loci (split_prepare -> call_spanning), by position
    uint32_t batch_seq = 0, host_seq = 0;     // arrival stamps: (batch_seq << 32) | record index, host-side additions behind the batch's records
    DeviceBuf<int32_t> d_folded;              // the folded counts the flush's tile kernel leaves for the candidate kernel (DeviceParams::folded_out)
    struct { bool valid = false; int32_t lo = 0, hi = 0; } fold;   // the positions d_folded holds (the tile kernels of this flush went first)
    DeviceBuf<PiscesTile> d_span_tiles;       // the 64-locus tiles whose anchor-resolved counts the candidate kernel reads (call_spanning)
    DeviceBuf<long long> d_row_idx;           // gather_count_rows_kernel
    DeviceBuf<int32_t> d_rows;
    std::unordered_map<int64_t, int32_t> row_of_locus;
    int64_t split_stats[4] = {0, 0, 0, 0};    // development: groups appended to the store, selected by flushes, dropped, flushes that swept

    // BAM decode on the device (bam_kernels.hip.h): file bytes -> inflated stream -> read batch, all handle-owned and grow-only
    struct BamState {
        DeviceBuf<uint8_t> d_file, d_stream;
        DeviceBuf<PiscesBgzfBlock> d_blocks;
        DeviceBuf<int32_t> d_status;
        DeviceBuf<uint16_t> d_exits;
        DeviceBuf<uint32_t> d_shared_exit;
        DeviceBuf<long long> d_header, d_entry;
        DeviceBuf<int32_t> d_n_reads, d_n_ops, d_n_bases, d_n_skipped, d_bstatus, d_n_indels, d_n_pool;
        DeviceBuf<long long> d_n_span;
        DeviceBuf<uint32_t> d_block_map;          // one bit per block of the chromosome: a kept read touches it
        DeviceBuf<unsigned long long> d_first_error;
        std::vector<uint32_t> block_map;          // its host copy
        int64_t log_slots = 0, found_slots = 0, found_pool = 0;   // what the batch needs in the log / the candidate records
        unsigned long long first_error = ~0ull;   // read index * 8 + code of the first read add_reads would refuse (all ones: none)
        // the read batch
        DeviceBuf<int32_t> position, cigar_offset, seq_offset;
        DeviceBuf<uint8_t> flags, cigar_op, bases, quals, op_quality, read_quality, dirs, del_dirs;
        DeviceBuf<long long> d_totals64;
        bool has_eqx = false;     // some read of the batch has an X or = operation
        bool has_dirs = false;    // some read of the batch carries an XD tag (a stitched read): `dirs` / `del_dirs` are made
        DeviceBuf<uint32_t> cigar_len;
        DeviceBuf<long long> d_slots;      // log slots of the reads (pisces_hip_add_decoded_reads)
        DeviceBuf<int32_t> d_fslots;       // candidate-record slots of the reads
        int64_t n_reads = 0, n_ops = 0, n_bases = 0, n_skipped = 0;
        bool valid = false;
        bool moved = false;       // the batch's byte arrays went to the read store (pisces_hip_add_decoded_reads)
        bool added = false;       // pisces_hip_add_decoded_reads took the batch (moved or copied): it is not added twice
        int32_t chain_mode = 0;   // 0: every chunk's entry guessed and checked; 1: the serial hop ran
        int32_t min_bq = 0;
    } bam;

    // the read store (store_kernels.hip.h): the reads of the blocks not yet flushed stay in HBM as they came, a flush calls from them
    std::vector<std::unique_ptr<ReadSegment>> segments, segment_pool;
    int read_path = 1;                        // 1: read store (default); 0: observation log (PISCES_HIP_READ_PATH=log: the earlier chain, kept for comparison)
    size_t store_direct_bytes = (size_t)256 << 10;   // a batch of at least this many bytes becomes a segment of its own (no copy); smaller ones are appended to the open segment
    size_t store_seal_bytes = (size_t)4 << 20;       // the open segment stops accepting batches at this size
    std::vector<int32_t> touched_keys;
    // state words of the segments: slots of buffers that were zeroed in one piece (a fill per new segment is a stream operation per add_reads)
    std::vector<std::unique_ptr<DeviceBuf<int32_t>>> state_pool;
    size_t state_slots_used = 0;
    std::vector<hipGraphExec_t> graphs;       // pisces_hip_call_tiles_graph_build
    std::vector<hipGraph_t> graph_defs;
    int store_waves = 0;                      // development: waves per tile of call_store_tiles_kernel (PISCES_HIP_STORE_WAVES; 0 = by launch size)
    int tile_order = 2;                       // which tile a workgroup of call_store_tiles_kernel takes in a launch of several tiles a CU (PISCES_HIP_TILE_ORDER): 2 tiles
                                              // traded by price inside small groups (exchanged_tile), 1 tile_order_kernel's order (a launch in front), 0 position order
    DeviceBuf<int32_t> d_tile_order;
    bool store_prio = true;                   // PISCES_HIP_STORE_PRIO=0: no issue priority for the walking waves of call_store_tiles_kernel (the A / B)
    int finder_wave = 0;                      // PISCES_HIP_FINDER: the default is a lane a read, events first; =bases: a lane a read, base by base (round 3's);
                                              // =wave / =batch: a wave for one / for 64 reads (finder_kernels.hip.h; measured slower)
    DeviceBuf<long long> d_scan_sums;         // block sums of launch_found_scan
    bool device_genotyper = true;             // PISCES_HIP_DEVICE_GENOTYPER=0: diploid / haploid genotypes are always the host pass of the flush (the A / B of the tests)
    bool merge_in_place = true;               // PISCES_HIP_MERGE_IN_PLACE=0: the candidate kernel's rows and the tile kernels' are merged into a vector of their own (the A / B of the tests)
    int device_checks = -1;                   // PISCES_HIP_DEVICE_CHECKS: 1 every host batch is checked on the device (add_fused_kernel), 0 none, -1 (default) from 65 536 reads up
    bool prep_map_clean = false;              // the block map of add_fused_kernel is all zero
    size_t prep_map_copies = 0;               // copies the map is kept in (kPrepReplicas, or 1 when a small block size makes it large)
    DeviceBuf<uint32_t> d_prep_map;
    DeviceBuf<uint8_t> d_fused_words;         // add_fused_kernel's shared words (spans, first error, workgroups through, totals): set once, left clean by every launch
    DeviceBuf<unsigned long long> d_fused_scan;   // its look-back words, zero between launches
    DeviceBuf<unsigned long long> d_compact_state;   // compact_records_kernel's look-back words (epoch-tagged: never cleared between launches)
    uint32_t compact_epoch = 0;
    struct DeferredGrid { ShapeArgs S; unsigned blocks; };
    std::vector<DeferredGrid> deferred_grid;  // grid roles of read_shape_kernel kept back until something reads or changes a grid (store_run_deferred)
    int32_t tile_loci = 0;                    // PISCES_HIP_TILE_LOCI: loci a tile of a regular flush (development; 0 = 64)
    int32_t stream_wgs_per_cu = 0;            // add_fused_kernel: > 0 = that many persistent stream workgroups a CU in front of the read role (PISCES_HIP_STREAM_WGS_PER_CU)
    int32_t role_stride = 1;                  // add_fused_kernel: every n-th workgroup at the front of the launch is a read workgroup (PISCES_HIP_ROLE_STRIDE)
    bool defer_grid = true;                   // PISCES_HIP_DEFER_GRID=0: enqueued by the add itself (the A / B)
#ifdef PISCES_ADD_STAMPS
    DeviceBuf<long long> d_add_stamps;
#endif
    int32_t fused_seq = 0;                    // add_fused_kernel launches of this handle (the word the host polls for the verdict)
    int32_t compact_mode = 0;                 // PISCES_HIP_COMPACT: 0 = one launch (direct sums / look-back by size), 2 = "two" (scan + gather), 3 = "lookback" always

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

static inline double now_seconds()
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
// adds the time a scope took to one of PiscesHip::host_time's counters
struct HostTimer {
    double* into;
    double t0;
    explicit HostTimer(double* p) : into(p), t0(now_seconds()) {}
    ~HostTimer() { if (into) *into += now_seconds() - t0; }
};
// waits inside a flush are accounted apart from the host's own work
#define PISCES_TIMED_WAIT(h, expr)                                      \
    do {                                                                \
        HostTimer _w(&(h)->host_time[2]);                               \
        PISCES_HIP_CHECK(h, expr);                                      \
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
static void store_view(PiscesHip* h, StoreView* V);
// with_store: the reads of the handle's read store are walked as well (the streaming surface; the device-resident surface passes false)
static hipError_t accumulate_tiles(PiscesHip* h, hipStream_t s, const uint32_t* d_tuples, const PiscesTile* d_tiles, int32_t n_tiles, bool with_sums,
                                   bool with_store = false)
{
    const size_t nc = (size_t)n_tiles * kTile * PISCES_COUNTS_PER_LOCUS;
    hipError_t e = h->d_counts.reserve(nc);
    if (e == hipSuccess && with_sums) e = h->d_sumq.reserve(nc);
    if (e == hipSuccess && with_sums) e = h->d_sumq_fix.reserve(2 * nc);
    if (e == hipSuccess && with_sums) e = ensure_quality_lut(h);
    if (e != hipSuccess) return e;
    (void)hipMemsetAsync(h->d_counts.p, 0, nc * sizeof(int32_t), s);
    if (with_sums) (void)hipMemsetAsync(h->d_sumq_fix.p, 0, 2 * nc * sizeof(unsigned long long), s);
    if (with_store && h->read_path == 1) {
        StoreView V;
        store_view(h, &V);
        // a launch of few tiles: several workgroups a tile, so that ~2 workgroups a CU exist (the kernel)
        const int split = (int)std::min<int64_t>(16, std::max<int64_t>(1, 2 * (int64_t)h->n_cus / std::max(n_tiles, 1)));
        hipLaunchKernelGGL(accumulate_store_tiles_kernel, dim3((unsigned)n_tiles, (unsigned)split), dim3(kBlock), 0, s, V, d_tuples, d_tiles, n_tiles, h->d_counts.p,
                           h->cfg.min_base_call_quality, with_sums ? h->d_sumq_fix.p : (unsigned long long*)nullptr,
                           with_sums ? (const ulonglong2*)h->d_bq_lut.p : (const ulonglong2*)nullptr);
    } else
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
    P.variants_only = 0;
    P.vq_tab = nullptr;
    P.sb_tab = nullptr;
    P.sb0_tab = nullptr;
    P.gq_cap = nullptr;
    P.vq_tab_k = P.sb_tab_k = P.tab_cov = 0;
    P.dirty_bits = nullptr;
    P.dirty_first = P.dirty_n = 0;
    P.folded_out = nullptr;
    P.folded_first = P.folded_n = 0;
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

// The memo tables of the call phase depend on the configuration only (~22 MB, four kernels, ~0.6 ms of pisces_hip_create): a job per
// chromosome or per interval range makes and destroys a handle (BaseGenomeProcessor.cs:40-90), so a destroyed handle leaves its tables
// here and the next handle of the same configuration on the same device takes them over instead of building them again.
namespace {
struct TableSet {
    PiscesHipConfig cfg;
    int device = 0;
    DeviceBuf<double> gq_tail, sb_tab, sb0_tab;
    DeviceBuf<int16_t> vq_tab, gq_cap;
};
std::mutex g_tables_mu;
std::vector<TableSet*>& table_cache() { static std::vector<TableSet*>* v = new std::vector<TableSet*>(); return *v; }   // (never destroyed: the HIP runtime may be gone by then)
TableSet* take_tables(const PiscesHipConfig& cfg, int device)
{
    std::lock_guard<std::mutex> lock(g_tables_mu);
    auto& v = table_cache();
    for (size_t i = 0; i < v.size(); i++)
        if (v[i]->device == device && std::memcmp(&v[i]->cfg, &cfg, sizeof(cfg)) == 0) {
            TableSet* t = v[i];
            v.erase(v.begin() + (std::ptrdiff_t)i);
            return t;
        }
    return nullptr;
}
void leave_tables(TableSet* t)
{
    std::lock_guard<std::mutex> lock(g_tables_mu);
    auto& v = table_cache();
    v.push_back(t);
    if (v.size() > 8) { delete v.front(); v.erase(v.begin()); }
}
}  // namespace

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

int32_t pisces_hip_device_count(void)
{
    int ndev = 0;
    const hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess) { g_create_error = std::string("pisces_hip_device_count: ") + hipGetErrorString(e); return PISCES_E_DEVICE; }
    return ndev;
}

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
        h->prof_on = getenv("PISCES_HIP_HOST_PROFILE") != nullptr;
        const char* rp = getenv("PISCES_HIP_READ_PATH");   // "log": reads are expanded into the observation log and bucketed at flush time (the earlier chain)
        if (rp && std::string(rp) == "log") h->read_path = 0;
        if (const char* v = getenv("PISCES_HIP_STORE_DIRECT_BYTES")) h->store_direct_bytes = (size_t)std::max(0ll, atoll(v));
        if (const char* v = getenv("PISCES_HIP_STORE_WAVES")) h->store_waves = atoi(v);
        if (const char* v = getenv("PISCES_HIP_TILE_ORDER")) h->tile_order = atoi(v);
        if (const char* v = getenv("PISCES_HIP_STORE_PRIO")) h->store_prio = atoi(v) != 0;
        if (const char* v = getenv("PISCES_HIP_DEVICE_MERGE")) h->device_merge = atoi(v) != 0 ? 1 : 0;   // the A/B of tests/test_read_store.py
        // MNV calling on: the split form, unless the candidate records are asked to come back unmerged (PISCES_HIP_DEVICE_MERGE=0: the
        // earlier form, every candidate an object on the host, the tile kernels Reference records only) or PISCES_HIP_MNV_SPLIT=0
        h->snv_walk = h->cfg.call_mnvs != 0 || (h->cfg.collapse != 0 && (h->cfg.collapse_freq_threshold > 0.0f || h->cfg.collapse_freq_ratio_threshold >= 1.0f));
        h->mnv_split = h->snv_walk && h->device_merge != 0;
        if (const char* v = getenv("PISCES_HIP_MNV_SPLIT")) h->mnv_split = h->mnv_split && atoi(v) != 0;
        if (const char* v = getenv("PISCES_HIP_STORE_SEAL_BYTES")) h->store_seal_bytes = (size_t)std::max(0ll, atoll(v));
        if (const char* v = getenv("PISCES_HIP_DEVICE_CHECKS")) h->device_checks = atoi(v) != 0 ? 1 : 0;
        if (const char* v = getenv("PISCES_HIP_TILE_LOCI")) h->tile_loci = atoi(v);
        if (const char* v = getenv("PISCES_HIP_STREAM_WGS_PER_CU")) h->stream_wgs_per_cu = std::max(0, atoi(v));
        if (const char* v = getenv("PISCES_HIP_ROLE_STRIDE")) h->role_stride = std::max(1, atoi(v));
        if (const char* v = getenv("PISCES_HIP_DEFER_GRID")) h->defer_grid = atoi(v) != 0;
        if (const char* v = getenv("PISCES_HIP_COMPACT")) h->compact_mode = std::string(v) == "two" ? 2 : std::string(v) == "lookback" ? 3 : 0;
        if (const char* v = getenv("PISCES_HIP_MERGE_IN_PLACE")) h->merge_in_place = atoi(v) != 0;
        if (const char* v = getenv("PISCES_HIP_DEVICE_GENOTYPER")) h->device_genotyper = atoi(v) != 0;
        if (const char* v = getenv("PISCES_HIP_FINDER")) h->finder_wave = std::string(v) == "wave" ? 1 : std::string(v) == "batch" ? 2 : std::string(v) == "bases" ? 3 : 0;
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
    const bool tables_by_default = !getenv("PISCES_HIP_NO_GQ_TABLE") && !getenv("PISCES_HIP_NO_CALL_TABLES") && h->cfg.noise_model == PISCES_NOISE_FLAT &&
                                   h->cfg.strand_bias_model != PISCES_SB_DIPLOID && h->cfg.max_variant_qscore <= 32767 && h->cfg.max_genotype_qscore <= 32767 &&
                                   h->cfg.min_genotype_qscore >= -32768;
    std::unique_ptr<TableSet> kept(tables_by_default ? take_tables(h->cfg, h->device) : nullptr);
    if (kept) {   // the tables a destroyed handle of this configuration left (the sizes are the constants below)
        h->d_gq_tail.swap(kept->gq_tail); h->d_vq_tab.swap(kept->vq_tab); h->d_sb_tab.swap(kept->sb_tab); h->d_sb0_tab.swap(kept->sb0_tab); h->d_gq_cap.swap(kept->gq_cap);
        h->P.gq_tail = h->d_gq_tail.p; h->P.gq_tail_a = 32; h->P.gq_tail_cov = 8192;
        h->P.vq_tab = h->d_vq_tab.p; h->P.sb_tab = h->d_sb_tab.p; h->P.sb0_tab = h->d_sb0_tab.p; h->P.gq_cap = h->d_gq_cap.p;
        h->P.vq_tab_k = h->P.sb_tab_k = 256;
        h->P.tab_cov = 8192;
        h->tables_shareable = true;
    }
    if (!kept && !getenv("PISCES_HIP_NO_GQ_TABLE")) {
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
    if (!kept && !getenv("PISCES_HIP_NO_CALL_TABLES") && h->cfg.noise_model == PISCES_NOISE_FLAT && h->cfg.strand_bias_model != PISCES_SB_DIPLOID &&
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
        h->tables_shareable = tables_by_default && h->P.gq_tail != nullptr;
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

int64_t pisces_hip_trim_memory(void)
{
    {   // the memo tables kept for the next handle go too
        std::lock_guard<std::mutex> lock(g_tables_mu);
        for (TableSet* t : table_cache()) delete t;
        table_cache().clear();
    }
    AllocCache& c = alloc_cache();
    std::lock_guard<std::mutex> lock(c.m);
    int dev = 0;
    (void)hipGetDevice(&dev);
    const size_t freed = cache_drop(c, -2);
    (void)hipSetDevice(dev);
    return (int64_t)freed;
}

int32_t pisces_hip_destroy(PiscesHip* h)
{
    return abi_guard<int32_t>(h, [&]() -> int32_t {
    if (!h) return PISCES_OK;
    if (h->prof_on) {
        static const char* names[19] = {"consume_found", "spanning: candidates of the batch", "spanning: counts to the host", "collapse", "device pass (MNVs)",
                                        "reallocate", "device pass (all)", "call_blocks", "row merge / genotypers", "copy out + DoneProcessing", "split: dirty loci + SNV store", "  of reallocate: mnv_reallocate_failed",
                                        "add_device_reads: consume_found", "add_device_reads: copies + checks", "add_device_reads: shape + discovery + commit",
                                        "add_reads: place + reserve", "add_reads: staging + upload", "add_reads: pass over the CIGARs", "add_reads: launches + commit"};
        for (int i = 0; i < 19; i++) fprintf(stderr, "pisces_hip host profile: %-36s %9.3f ms\n", names[i], h->prof[i] * 1e3);
        if (h->mnv_split)
            fprintf(stderr, "pisces_hip split form: %lld SNV groups into the store, %lld taken by flushes (dirty loci), %lld dropped unseen, %lld sweeps\n",
                    (long long)h->split_stats[0], (long long)h->split_stats[1], (long long)h->split_stats[2], (long long)h->split_stats[3]);
    }
    (void)hipSetDevice(h->device);
    bool idle = !h->stream || hipStreamSynchronize(h->stream) == hipSuccess;
    for (int k = 0; k < PiscesHip::kLanes; k++)
        if (h->lane[k] && hipStreamSynchronize(h->lane[k]) != hipSuccess) idle = false;
    // launches that went to a caller's stream may still read the handle's tables: the buffers only go to the cache (and so to the next
    // handle) once the whole device is idle (hipFree used to wait implicitly; the cache does not)
    if (h->foreign_stream_ever && hipDeviceSynchronize() != hipSuccess) idle = false;
    (void)pisces_hip_comm_destroy(h);
    struct KeepFreed {   // nothing of the handle is in flight: what it held may go to the next handle as it is
        explicit KeepFreed(bool on) { tl_keep_freed = on; }
        ~KeepFreed() { tl_keep_freed = false; }
    } keep(idle);
    if (idle && h->tables_shareable && h->d_vq_tab.p && h->d_gq_tail.p && h->d_gq_cap.p) {   // the next handle of this configuration takes them over
        TableSet* t = new TableSet();
        t->cfg = h->cfg;
        t->device = h->device;
        t->gq_tail.swap(h->d_gq_tail); t->vq_tab.swap(h->d_vq_tab); t->sb_tab.swap(h->d_sb_tab); t->sb0_tab.swap(h->d_sb0_tab); t->gq_cap.swap(h->d_gq_cap);
        leave_tables(t);
    }
    h->d_summary.release();
    h->d_ref.release(); h->d_tuples.release(); h->d_tiles.release(); h->d_tile_results.release();
    h->d_records.release(); h->d_counts.release(); h->d_gapped.release(); h->d_count.release(); h->d_totals.release(); h->d_qlut.release(); h->d_bq_lut.release(); h->d_sumq_fix.release(); h->d_sumq.release(); h->d_gq_tail.release(); h->d_vq_tab.release(); h->d_sb_tab.release(); h->d_sb0_tab.release(); h->d_gq_cap.release(); h->d_params.release(); h->d_offsets.release(); h->d_compact.release(); h->d_tile_order.release();
    for (int i = 0; i < 2; i++) { h->d_log_pos[i].release(); h->d_log_tup[i].release(); }
    h->d_log_n.release(); h->d_flags.release(); h->d_bucket.release(); h->d_total.release();
    for (auto& st : h->stage) {
        st.d.release();
        if (st.h) host_free(st.h);
        st.h = nullptr;
        if (st.done) (void)hipEventDestroy(st.done);
        st.done = nullptr;
    }
    h->h_stage = nullptr;
    if (h->async.done) (void)hipEventDestroy(h->async.done);
    h->async.done = nullptr;
    if (h->h_dl) host_free(h->h_dl);
    h->h_dl = nullptr;
    if (h->h_cand_dl) host_free(h->h_cand_dl);
    h->h_cand_dl = nullptr;
    if (h->h_meta) host_free(h->h_meta);
    h->h_meta = nullptr;
    if (h->h_counts) host_free(h->h_counts);
    h->h_counts = nullptr;
    if (h->found.h) host_free(h->found.h);
    h->found.h = nullptr;
    if (h->found.done) (void)hipEventDestroy(h->found.done);
    h->found.done = nullptr;
    if (h->found.counted) (void)hipEventDestroy(h->found.counted);
    h->found.counted = nullptr;
    if (h->found.h_totals) host_free(h->found.h_totals);
    h->found.h_totals = nullptr;
    h->d_found.release(); h->d_found_pool.release(); h->d_found_slots.release(); h->d_found_pool_first.release();
    h->d_merge_tab.release(); h->d_merge_acc.release();
    h->d_scan_sums.release(); h->d_prep_map.release(); h->d_fused_words.release(); h->d_fused_scan.release(); h->d_compact_state.release(); h->d_folded.release(); h->d_span_tiles.release();
    h->d_snv[0].release(); h->d_snv[1].release(); h->d_snv_n.release(); h->d_snv_sel.release(); h->d_dirty.release(); h->d_row_idx.release(); h->d_rows.release();
    if (h->h_totals) host_free(h->h_totals);
    if (h->h_cnt_x) host_free(h->h_cnt_x);
    h->h_totals = nullptr;
    if (h->h_prep) host_free(h->h_prep);
    h->h_prep = nullptr;
    if (h->h_snv_sel) host_free(h->h_snv_sel);
    h->h_snv_sel = nullptr;
    h->d_found_misc.release(); h->d_found_totals.release();
    h->d_cands.release(); h->d_alleles.release(); h->d_cand_records.release(); h->d_cand_callable.release();
    h->segments.clear();
    h->segment_pool.clear();
    h->state_pool.clear();
    for (hipGraphExec_t g : h->graphs) (void)hipGraphExecDestroy(g);
    for (hipGraph_t g : h->graph_defs) (void)hipGraphDestroy(g);
    for (hipEvent_t ev : h->ring) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : h->ev_chain) if (ev) (void)hipEventDestroy(ev);
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

int32_t pisces_hip_set_owned_range(PiscesHip* h, int32_t lo, int32_t hi)
{
    return abi_guard<int32_t>(h, [&]() -> int32_t {
    if (!h || lo < 1 || hi < lo) return fail(h, PISCES_E_INVALID_ARG, "set_owned_range: bad range");
    h->own_lo = lo;
    h->own_hi = hi;
    return PISCES_OK;
    });
}

#include "surface_reads.inc.h"

#include "surface_store.inc.h"

#include "surface_flush.inc.h"

#include "surface_device.inc.h"

#include "surface_bam.inc.h"

#include "surface_comm.inc.h"

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

int32_t pisces_hip_get_stream(PiscesHip* h, void** stream)
{
    return abi_guard<int32_t>(h, [&]() -> int32_t {
    if (!h || !stream) return PISCES_E_INVALID_ARG;
    *stream = (void*)h->stream;
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
