// surface_reads.inc.h — part of pisces_hip.hip (included there, inside its extern "C" block; not a translation unit of its own).
// IStateManager.AddAlleleCounts / AddCandidates behind the C ABI: staging, pisces_hip_add_observations, candidates and forced alleles the
// caller brings, candidate discovery on the device, pisces_hip_add_reads, and the host-only walks (pisces_hip_expand_reads, pisces_hip_find_*).

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
            if (h->h_meta) host_free(h->h_meta);
            h->h_meta = nullptr;
            h->h_meta_cap = 0;
            PISCES_HIP_CHECK(h, host_alloc((void**)&h->h_meta, want));
            h->h_meta_cap = want;
        }
    }
    uint8_t* at = h->h_meta + h->h_meta_used;
    h->h_meta_used += need;
    std::memcpy(at, src, bytes);
    PISCES_HIP_CHECK(h, hipMemcpyAsync(dst, at, bytes, hipMemcpyHostToDevice, h->stream));
    return PISCES_OK;
}

static int32_t finish_candidate_discovery(PiscesHip* h);
// next staging pair with room for `bytes`; waits only for the work that used THIS pair two calls ago
static int32_t stage_reserve(PiscesHip* h, size_t bytes, bool with_device_half = true)
{
    // a deferred candidate walk (find_emit of the last batch, still to be enqueued) reads that batch's arrays where they lie — for a small
    // batch in the device half of a staging pair, which two more reservations would overwrite or a growing reserve free: it goes first
    { int32_t rcd = finish_candidate_discovery(h); if (rcd) return rcd; }
    h->stage_cur ^= 1;
    PiscesHip::Stage& st = h->stage[h->stage_cur];
    if (!st.done) PISCES_HIP_CHECK(h, hipEventCreateWithFlags(&st.done, hipEventDisableTiming));
    if (st.in_flight) {
        PISCES_HIP_CHECK(h, hipEventSynchronize(st.done));
        st.in_flight = false;
    }
    if (bytes > st.h_cap) {
        if (st.h) host_free(st.h);
        st.h = nullptr;
        st.h_cap = 0;
        const size_t want = bytes + bytes / 2 + 4096;
        PISCES_HIP_CHECK(h, host_alloc((void**)&st.h, want));
        st.h_cap = want;
    }
    if (with_device_half) PISCES_HIP_CHECK(h, st.d.reserve(bytes));
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
static inline uint64_t candidate_hash(const HostCandidate& c, bool with_open_ends)
{
    uint64_t x = 0xcbf29ce484222325ull;   // FNV-1a over the fields CandidateAllele.Equals compares
    auto mix = [&](uint64_t v) { x = (x ^ v) * 0x100000001b3ull; };
    mix((uint32_t)c.position);
    mix((uint32_t)c.category | (with_open_ends ? ((uint32_t)c.open_left << 8) | ((uint32_t)c.open_right << 9) | 0x10000u : 0u));
    for (char ch : c.ref) mix((uint8_t)ch);
    mix(0x3Eu);
    for (char ch : c.alt) mix((uint8_t)ch);
    return x;
}
// arrival stamp of a candidate the host adds itself (pisces_hip_add_candidates, forced alleles, what AlleleCaller.Call hands back to the
// state): behind every record of the batches added so far, before the next batch's
static uint64_t next_host_stamp(PiscesHip* h) { return ((uint64_t)h->batch_seq << 32) | 0x80000000ull | (uint64_t)(h->host_seq++ & 0x7FFFFFFFu); }

// the block's candidates back in order of first arrival (after candidates with earlier stamps were added late: the SNV groups a flush
// takes from the device store), and the hash index over them rebuilt
static inline uint64_t candidate_hash(const HostCandidate& c, bool with_open_ends);
static void reorder_block_candidates(PiscesHip* h, BlockObs* b)
{
    std::stable_sort(b->cands.begin(), b->cands.end(), [](const HostCandidate& x, const HostCandidate& y) { return x.stamp < y.stamp; });
    const bool track_open = h->cfg.collapse != 0;
    b->cand_index.clear();
    b->cand_next.assign(b->cands.size(), 0xFFFFFFFFu);
    for (uint32_t i = 0; i < (uint32_t)b->cands.size(); i++) {
        const uint64_t key = candidate_hash(b->cands[i], track_open);
        auto it = b->cand_index.find(key);
        if (it == b->cand_index.end()) { b->cand_index.emplace(key, i); continue; }
        uint32_t j = it->second;
        while (b->cand_next[j] != 0xFFFFFFFFu) j = b->cand_next[j];
        b->cand_next[j] = i;
    }
}

static void add_candidate(PiscesHip* h, const HostCandidate& cnd)
{
    BlockObs* b = get_block(h, cnd.position);
    const bool track_open = h->cfg.collapse != 0;
    const uint64_t key = candidate_hash(cnd, track_open);
    auto same = [&](const HostCandidate& e) {
        return e.position == cnd.position && e.category == cnd.category && e.ref == cnd.ref && e.alt == cnd.alt &&
               (!track_open || (e.open_left == cnd.open_left && e.open_right == cnd.open_right));
    };
    HostCandidate* found = nullptr;
    auto it = b->cand_index.find(key);
    uint32_t last = 0xFFFFFFFFu;
    if (it != b->cand_index.end())
        for (uint32_t i = it->second; i != 0xFFFFFFFFu; i = b->cand_next[i]) {
            if (same(b->cands[i])) { found = &b->cands[i]; break; }
            last = i;
        }
    if (!found) {
        const uint32_t idx = (uint32_t)b->cands.size();
        b->cands.push_back(cnd);
        b->cand_next.push_back(0xFFFFFFFFu);
        if (last != 0xFFFFFFFFu) b->cand_next[last] = idx;
        else b->cand_index.emplace(key, idx);
    } else {
        for (int d = 0; d < 3; d++) {
            found->support_by_dir[d] += cnd.support_by_dir[d];
            found->well_anchored_by_dir[d] += cnd.well_anchored_by_dir[d];
        }
        found->stamp = std::min(found->stamp, cnd.stamp);                // (the candidate keeps the place of its first arrival)
        found->from_reads = found->from_reads && cnd.from_reads;
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
    for (auto& c : list) { c.stamp = next_host_stamp(h); c.from_reads = false; add_candidate(h, c); }
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

int32_t pisces_hip_set_exact_total_called(PiscesHip* h, int32_t on)
{
    return abi_guard<int32_t>(h, [&]() -> int32_t {
    if (!h) return PISCES_E_INVALID_ARG;
    h->exact_total_called = on != 0;
    return PISCES_OK;
    });
}

// The chromosome's known (prior) variants (Factory.cs:204, 378-395: the priors file's insertions and MNVs of this chromosome): the
// collapser annotates the candidates that equal one (VariantCollapser.cs:16-24, 178-190) and prefers them among potential matches (:216-218).
int32_t pisces_hip_set_known_variants(PiscesHip* h, const PiscesCandidate* cands, int64_t n, const uint8_t* alleles, int64_t allele_bytes)
{
    return abi_guard<int32_t>(h, [&]() -> int32_t {
    if (!h) return PISCES_E_INVALID_ARG;
    std::vector<HostCandidate> list;
    int32_t rc = host_candidates_of(h, cands, n, alleles, allele_bytes, list, "set_known_variants");
    if (rc) return rc;
    for (auto& c : list)   // (CandidateAllele.Equals compares the type too: derived from the alleles as the reader of the priors file does)
        c.category = (c.ref.size() == 1 && c.alt.size() == 1) ? PISCES_CAT_SNV : c.ref.size() == c.alt.size() ? PISCES_CAT_MNV
                     : c.ref.size() > c.alt.size() ? PISCES_CAT_DELETION : PISCES_CAT_INSERTION;
    // (in position order: AnnotateKnown looks a candidate's position up instead of comparing it with every known variant)
    std::stable_sort(list.begin(), list.end(), [](const HostCandidate& a, const HostCandidate& b) { return a.position < b.position; });
    h->known_variants = std::move(list);
    return PISCES_OK;
    });
}

// PiscesApplicationOptions.ExcludeMNVsFromCollapsing (Options/PiscesApplicationOptions.cs:62 -> Factory.cs:204 -> VariantCollapser.cs:33)
int32_t pisces_hip_set_exclude_mnvs_from_collapsing(PiscesHip* h, int32_t on)
{
    return abi_guard<int32_t>(h, [&]() -> int32_t {
    if (!h) return PISCES_E_INVALID_ARG;
    h->exclude_mnvs_from_collapsing = on != 0;
    return PISCES_OK;
    });
}

// SmallVariantCaller.AddForcedAlleleAsCandidate :118-132, before GetCandidatesToProcess(upTo)
static void add_forced_as_candidates(PiscesHip* h, int32_t up_to_position)
{
    while (h->n_forced_added < h->forced.size()) {
        const HostCandidate& c = h->forced[h->n_forced_added];
        if (up_to_position >= 0 && c.position > up_to_position) break;
        HostCandidate fc = c;
        fc.stamp = next_host_stamp(h);
        fc.from_reads = false;
        add_candidate(h, fc);
        h->n_forced_added++;
    }
}

// the exclusive scans of the per-read record / pool-byte counts (n entries each, the last one zero: it receives the total), totals[0..1] = the sums
static int32_t launch_found_scan(PiscesHip* h, int32_t* a, int32_t* b, int32_t n, long long* d_totals)
{
    if (n <= 2 * kScanBlock) {
        hipLaunchKernelGGL(found_scan_kernel, dim3(1), dim3(1024), 0, h->stream, a, b, n, d_totals);
        return PISCES_OK;
    }
    const int32_t n_blocks = (n + kScanBlock - 1) / kScanBlock;
    PISCES_HIP_CHECK(h, h->d_scan_sums.reserve((size_t)2 * n_blocks));
    hipLaunchKernelGGL(scan_block_sums_kernel, dim3((unsigned)n_blocks), dim3(1024), 0, h->stream, (const int32_t*)a, (const int32_t*)b, n, h->d_scan_sums.p, n_blocks);
    hipLaunchKernelGGL(scan_block_offsets_kernel, dim3(1), dim3(1024), 0, h->stream, h->d_scan_sums.p, n_blocks, d_totals);
    hipLaunchKernelGGL(scan_apply_kernel, dim3((unsigned)n_blocks), dim3(1024), 0, h->stream, a, b, n, (const long long*)h->d_scan_sums.p, n_blocks);
    return PISCES_OK;
}
// the count / emit passes of the candidate walk: a wave per read when the M operations are walked (finder_kernels.hip.h, the wave form),
// a lane per read otherwise (insertions and deletions only: a read is a loop over its CIGAR) or when PISCES_HIP_FINDER=lane asks for it
static void launch_find_count(PiscesHip* h, const DevReadBatch& db, const uint8_t* d_deldirs, const FinderParams& FP, int32_t nr, int32_t* n_found, int32_t* n_pool)
{
    if (FP.snvs_and_mnvs && h->finder_wave == 2) {
        hipLaunchKernelGGL(find_batch_wave_kernel<false>, dim3((unsigned)((nr + 255) / 256)), dim3(256), 0, h->stream, db, d_deldirs, (const uint8_t*)h->d_ref.p, h->ref_len, FP,
                           n_found, n_pool, (const int32_t*)nullptr, (const int32_t*)nullptr, (DevFound*)nullptr, (uint8_t*)nullptr, (unsigned int*)nullptr, 0, (int32_t*)nullptr);
    } else if (FP.snvs_and_mnvs && h->finder_wave) {
        const unsigned waves = (unsigned)((nr + kReadsPerWave - 1) / kReadsPerWave);
        hipLaunchKernelGGL(find_count_wave_kernel, dim3((waves + 3) / 4), dim3(256), 0, h->stream, db, d_deldirs, (const uint8_t*)h->d_ref.p, h->ref_len, FP, n_found, n_pool);
    } else if (FP.snvs_and_mnvs && h->finder_wave != 3 && FP.min_bq <= 127) {
        hipLaunchKernelGGL(find_count_kernel<true>, dim3((unsigned)((nr + 255) / 256)), dim3(256), 0, h->stream, db, d_deldirs, (const uint8_t*)h->d_ref.p, h->ref_len, FP, n_found, n_pool);
    } else {
        hipLaunchKernelGGL(find_count_kernel<false>, dim3((unsigned)((nr + 255) / 256)), dim3(256), 0, h->stream, db, d_deldirs, (const uint8_t*)h->d_ref.p, h->ref_len, FP, n_found, n_pool);
    }
}
static void launch_find_emit(PiscesHip* h, const DevReadBatch& db, const uint8_t* d_deldirs, const FinderParams& FP, int32_t nr, const int32_t* d_slots,
                             const int32_t* d_pool_first, DevFound* out, uint8_t* pool, unsigned int* misc, int32_t pool_capacity)
{
    if (FP.snvs_and_mnvs && h->finder_wave == 2) {
        hipLaunchKernelGGL(find_batch_wave_kernel<true>, dim3((unsigned)((nr + 255) / 256)), dim3(256), 0, h->stream, db, d_deldirs, (const uint8_t*)h->d_ref.p, h->ref_len, FP,
                           (int32_t*)nullptr, (int32_t*)nullptr, d_slots, d_pool_first, out, pool, misc, pool_capacity, (int32_t*)(misc + 1));
    } else if (FP.snvs_and_mnvs && h->finder_wave) {
        const unsigned waves = (unsigned)((nr + kReadsPerWave - 1) / kReadsPerWave);
        hipLaunchKernelGGL(find_emit_wave_kernel, dim3((waves + 3) / 4), dim3(256), 0, h->stream, db, d_deldirs, (const uint8_t*)h->d_ref.p, h->ref_len, FP, d_slots,
                           d_pool_first, out, pool, misc, pool_capacity, (int32_t*)(misc + 1));
    } else if (FP.snvs_and_mnvs && h->finder_wave != 3 && FP.min_bq <= 127) {
        hipLaunchKernelGGL(find_emit_kernel<true>, dim3((unsigned)((nr + 255) / 256)), dim3(256), 0, h->stream, db, d_deldirs, (const uint8_t*)h->d_ref.p, h->ref_len, FP, d_slots,
                           d_pool_first, out, pool, misc, pool_capacity, (int32_t*)(misc + 1));
    } else {
        hipLaunchKernelGGL(find_emit_kernel<false>, dim3((unsigned)((nr + 255) / 256)), dim3(256), 0, h->stream, db, d_deldirs, (const uint8_t*)h->d_ref.p, h->ref_len, FP, d_slots,
                           d_pool_first, out, pool, misc, pool_capacity, (int32_t*)(misc + 1));
    }
}

// Candidate discovery for a read batch that is on the device (find_count / found_scan / find_emit kernels), enqueued on the handle's
// stream; its records come back into pinned memory and are merged by consume_found when they are needed.  d_slots: the record slots
// the host reserved per read from the CIGARs (MNV calling off), found_slots / found_pool their totals.
static int32_t enqueue_found_records(PiscesHip* h, const DevReadBatch& db, const uint8_t* d_deldirs, int32_t nr, const FinderParams& FP, const int32_t* d_slots,
                                     const int32_t* d_pool_first, int64_t found_slots, int64_t found_pool);
static FinderParams finder_params(const PiscesHip* h)
{
    const FinderParams FP = {h->cfg.min_base_call_quality, PISCES_ANCHOR_SIZE, h->snv_walk ? 1 : 0, h->cfg.call_mnvs ? 1 : 0, h->cfg.max_mnv_length,
                             h->cfg.max_gap_between_mnv, h->mnv_split ? 1 : 0};
    return FP;
}
// h->eqx_in_batch: some read of the batch has an X or = operation.  With MNV calling off those bases are allele counts that no SNV candidate
// stands for (ProcessCigarOps walks M operations only): the walk then leaves a record for each of them that an M operation would have
// made a candidate of (finder_walk.h kFoundUnwalked), counted on the device like the walk of MNV calling on — the slots the caller
// reserved from the CIGARs hold insertions and deletions only.
static int32_t enqueue_candidate_discovery(PiscesHip* h, const DevReadBatch& db, const uint8_t* d_deldirs, int32_t nr, const int32_t* d_slots_in,
                                           int64_t found_slots, int64_t found_pool)
{
    FinderParams FP = finder_params(h);
    const bool unwalked = !h->snv_walk && h->eqx_in_batch;
    if (unwalked) FP.mark_x_spans = 2;
    h->found.fp = FP;
    // (arrival stamps: this batch's records come behind everything the host added so far)
    h->batch_seq++;
    h->host_seq = 0;
    h->found.batch = h->batch_seq;
    h->found.split = false;
    h->found.split_counted = false;
    h->found.min_position = 0;
    h->found.counted_only = false;
    PISCES_HIP_CHECK(h, h->d_found_misc.reserve(4));
    PISCES_HIP_CHECK(h, hipMemsetAsync(h->d_found_misc.p, 0, 4 * sizeof(unsigned int), h->stream));
    if (!h->snv_walk && !unwalked) return enqueue_found_records(h, db, d_deldirs, nr, FP, d_slots_in, nullptr, found_slots, found_pool);
    // count, scan (one more element than reads: the last one receives the total); the totals size the record buffers: they travel to pinned
    // memory behind an event, and the second half (finish_candidate_discovery) is enqueued by whichever entry comes next — the batch's
    // arrays stay where they are until then (a segment's blob; or the staging pair, which only the next add reuses, behind that half)
    PISCES_HIP_CHECK(h, h->d_found_slots.reserve((size_t)nr + 1));
    PISCES_HIP_CHECK(h, h->d_found_pool_first.reserve((size_t)nr + 1));
    PISCES_HIP_CHECK(h, h->d_found_totals.reserve(2));
    if (!h->found.h_totals) PISCES_HIP_CHECK(h, host_alloc((void**)&h->found.h_totals, 64));
    if (!h->found.counted) PISCES_HIP_CHECK(h, hipEventCreateWithFlags(&h->found.counted, hipEventDisableTiming));
    PISCES_HIP_CHECK(h, hipMemsetAsync(h->d_found_slots.p + nr, 0, sizeof(int32_t), h->stream));
    PISCES_HIP_CHECK(h, hipMemsetAsync(h->d_found_pool_first.p + nr, 0, sizeof(int32_t), h->stream));
    launch_find_count(h, db, d_deldirs, FP, nr, h->d_found_slots.p, h->d_found_pool_first.p);
    { int32_t rcs = launch_found_scan(h, h->d_found_slots.p, h->d_found_pool_first.p, nr + 1, h->d_found_totals.p); if (rcs) return rcs; }
    PISCES_HIP_CHECK(h, hipGetLastError());
    PISCES_HIP_CHECK(h, hipMemcpyAsync(h->found.h_totals, h->d_found_totals.p, 2 * sizeof(long long), hipMemcpyDeviceToHost, h->stream));
    PISCES_HIP_CHECK(h, hipEventRecord(h->found.counted, h->stream));
    h->found.db = db;
    h->found.d_deldirs = d_deldirs;
    h->found.nr = nr;
    h->found.counted_only = true;
    h->found.in_flight = true;
    h->found.n_slots = 0;
    h->found.pool_bytes = 0;
    return PISCES_OK;
}
// the second half of a batch's candidate discovery (MNV calling on; or off, over reads with X / = operations), if it is still to come
static int32_t finish_candidate_discovery(PiscesHip* h)
{
    // A failure here comes AFTER the batch's reads were committed (its counts are in the store, its candidates are not): the handle no
    // longer holds what the reference's state manager would, and says so on every later call instead of calling without them.
    if (h->poisoned) return fail(h, PISCES_E_STATE, "the candidates of an earlier batch were lost (" + h->poison_why + "): the handle's state is unusable, destroy it");
    if (!h->found.counted_only) return PISCES_OK;
    h->found.counted_only = false;
    h->found.in_flight = false;
    PISCES_TIMED_WAIT(h, hipEventSynchronize(h->found.counted));
    const long long found_slots = h->found.h_totals[0], found_pool = h->found.h_totals[1];
    int32_t rc = PISCES_OK;
    if (found_slots > 0x7FFFFFF0ll || found_pool > 0x7FFFFFF0ll) rc = fail(h, PISCES_E_INVALID_ARG, "add_reads: too many candidates in one batch");
    else rc = enqueue_found_records(h, h->found.db, h->found.d_deldirs, h->found.nr, h->found.fp, h->d_found_slots.p, h->d_found_pool_first.p, found_slots, found_pool);
    if (rc) {
        h->poisoned = true;
        h->poison_why = h->err;
    }
    return rc;
}
static int32_t enqueue_found_records(PiscesHip* h, const DevReadBatch& db, const uint8_t* d_deldirs, int32_t nr, const FinderParams& FP, const int32_t* d_slots,
                                     const int32_t* d_pool_first, int64_t found_slots, int64_t found_pool)
{
        if (found_slots > 0) {
            PISCES_HIP_CHECK(h, h->d_found.reserve((size_t)found_slots));
            PISCES_HIP_CHECK(h, h->d_found_pool.reserve((size_t)found_pool + 16));
            launch_find_emit(h, db, d_deldirs, FP, nr, d_slots, d_pool_first, h->d_found.p, h->d_found_pool.p, h->d_found_misc.p, (int32_t)found_pool);
            PISCES_HIP_CHECK(h, hipGetLastError());
            // records + pool + {cursor, overflow, merged groups} come back into pinned memory; consume_found waits for them when they are needed
            const bool merge = h->mnv_split || h->device_merge == 1 || (h->device_merge < 0 && found_slots >= 2048);
            const size_t rec_bytes = (size_t)found_slots * (merge ? sizeof(DevMerged) : sizeof(DevFound)), pool_al = ((size_t)found_pool + 15) & ~(size_t)15;
            const size_t need = rec_bytes + pool_al + 16;
            if (need > h->found.h_cap) {
                if (h->found.h) host_free(h->found.h);
                h->found.h = nullptr;
                h->found.h_cap = 0;
                PISCES_HIP_CHECK(h, host_alloc((void**)&h->found.h, need + need / 2));
                h->found.h_cap = need + need / 2;
            }
            if (!h->found.done) PISCES_HIP_CHECK(h, hipEventCreateWithFlags(&h->found.done, hipEventDisableTiming));
            if (merge) {
                // RegionState.AddCandidate on the device (found_merge_kernel / found_gather_kernel): one record a candidate crosses PCIe, as the
                // gather kernel's own stores into the pinned buffer
                size_t cap = 1024;
                while (cap < 2 * (size_t)found_slots) cap <<= 1;
                PISCES_HIP_CHECK(h, h->d_merge_tab.reserve(cap));
                PISCES_HIP_CHECK(h, h->d_merge_acc.reserve((size_t)found_slots * kMergeAcc));
                PISCES_HIP_CHECK(h, hipMemsetAsync(h->d_merge_tab.p, 0xFF, cap * sizeof(int32_t), h->stream));
                PISCES_HIP_CHECK(h, hipMemsetAsync(h->d_merge_acc.p, 0, (size_t)found_slots * kMergeAcc * sizeof(int32_t), h->stream));
                const unsigned mgrid = (unsigned)((found_slots + 255) / 256);
                hipLaunchKernelGGL(found_merge_kernel, dim3(mgrid), dim3(256), 0, h->stream, (const DevFound*)h->d_found.p, (int32_t)found_slots,
                                   (const uint8_t*)h->d_found_pool.p, h->d_merge_tab.p, (uint32_t)(cap - 1), h->d_merge_acc.p, h->cfg.collapse != 0 ? 1 : 0);
                if (h->mnv_split) {
                    // the fully anchored SNV groups stay on the device (the SNV store); everything else goes to the host as before
                    if (!h->d_snv_n.p) {
                        PISCES_HIP_CHECK(h, h->d_snv_n.reserve(4));
                        PISCES_HIP_CHECK(h, hipMemsetAsync(h->d_snv_n.p, 0, 4 * sizeof(unsigned int), h->stream));
                    }
                    PISCES_HIP_CHECK(h, h->d_snv[h->snv_cur].grow_keep((size_t)(h->snv_ub + found_slots), (size_t)h->snv_ub, h->stream));
                    hipLaunchKernelGGL(found_gather_split_kernel, dim3(mgrid), dim3(256), 0, h->stream, (const DevFound*)h->d_found.p, (int32_t)found_slots,
                                       (const int32_t*)h->d_merge_acc.p, (DevMerged*)h->found.h, h->d_found_misc.p, h->d_snv[h->snv_cur].p,
                                       h->d_snv_n.p + h->snv_cur, (uint32_t)std::min<size_t>(h->d_snv[h->snv_cur].cap, 0xFFFFFFF0u), h->found.batch,
                                       h->cfg.collapse != 0 ? 1 : 0);
                    h->snv_ub += found_slots;   // (an upper bound until the batch's counts are in: consume_found)
                    h->found.split = true;
                } else
                hipLaunchKernelGGL(found_gather_kernel, dim3(mgrid), dim3(256), 0, h->stream, (const DevFound*)h->d_found.p, (int32_t)found_slots,
                                   (const int32_t*)h->d_merge_acc.p, (DevMerged*)h->found.h, h->d_found_misc.p + 2);
                PISCES_HIP_CHECK(h, hipGetLastError());
            } else {
            h->pcie[2] += (int64_t)rec_bytes + found_pool;
            PISCES_HIP_CHECK(h, hipMemcpyAsync(h->found.h, h->d_found.p, rec_bytes, hipMemcpyDeviceToHost, h->stream));
            }
            h->found.merged = merge;
            if (found_pool > 0)
                PISCES_HIP_CHECK(h, hipMemcpyAsync(h->found.h + rec_bytes, h->d_found_pool.p, (size_t)found_pool, hipMemcpyDeviceToHost, h->stream));
            PISCES_HIP_CHECK(h, hipMemcpyAsync(h->found.h + rec_bytes + pool_al, h->d_found_misc.p, 4 * sizeof(unsigned int), hipMemcpyDeviceToHost, h->stream));
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
    { int32_t rcd = finish_candidate_discovery(h); if (rcd) return rcd; }
    if (!h->found.in_flight) return PISCES_OK;
    HostTimer prof(h->prof_on ? &h->prof[0] : nullptr);
    h->found.in_flight = false;
    PISCES_TIMED_WAIT(h, hipEventSynchronize(h->found.done));
    const DevFound* recs = (const DevFound*)h->found.h;
    const uint8_t* pool = h->found.h + (size_t)h->found.n_slots * (h->found.merged ? sizeof(DevMerged) : sizeof(DevFound));
    const unsigned int* misc = (const unsigned int*)(pool + (((size_t)h->found.pool_bytes + 15) & ~(size_t)15));
    if (misc[1] != 0) return fail(h, PISCES_E_DEVICE, "add_reads: the candidate records of the device did not fit their reservation");
    if (h->found.merged) {
        // the groups in order of first arrival (a position's candidates keep that order, RegionState.cs:104-123): their first records'
        // indices are distinct numbers below n_slots
        const DevMerged* groups = (const DevMerged*)h->found.h;
        const int64_t n_groups = (int64_t)misc[2];
        if (n_groups > h->found.n_slots) return fail(h, PISCES_E_DEVICE, "add_reads: the merged candidate records of the device are inconsistent");
        if (h->found.split) {   // the plain SNV groups of the batch are in the SNV store: its size is exact again
            if ((int64_t)misc[3] + n_groups > h->found.n_slots) return fail(h, PISCES_E_DEVICE, "add_reads: the merged candidate records of the device are inconsistent");
            if (!h->found.split_counted) h->snv_ub -= h->found.n_slots - (int64_t)misc[3];
            h->split_stats[0] += (int64_t)misc[3];
        }
        h->pcie[2] += n_groups * (int64_t)sizeof(DevMerged) + h->found.pool_bytes;
        // (few groups among many record slots once the plain SNV groups stay on the device: ordered by a sort of the groups, not by a pass over the slots)
        std::vector<int32_t>& order = h->found.order;
        order.resize((size_t)n_groups);
        for (int64_t k = 0; k < n_groups; k++) {
            const int32_t first = groups[k].first;
            if (first < 0 || first >= h->found.n_slots) return fail(h, PISCES_E_DEVICE, "add_reads: the merged candidate records of the device are inconsistent");
            order[(size_t)k] = (int32_t)k;
        }
        std::sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return groups[a].first < groups[b].first; });
        for (int64_t k = 1; k < n_groups; k++)
            if (groups[order[(size_t)k]].first == groups[order[(size_t)k - 1]].first) return fail(h, PISCES_E_DEVICE, "add_reads: the merged candidate records of the device are inconsistent");
        for (int64_t oi = 0; oi < n_groups; oi++) {
            const DevMerged& m = groups[order[(size_t)oi]];
            const int64_t i = m.first;
            if (m.f.c.category == kFoundSpanMark) {   // the positions of an X operation: no candidate, dirty loci of their blocks (surface_flush.inc.h)
                for (int32_t k = block_key(h, m.f.c.position); k <= block_key(h, m.f.c.position + m.f.c.length - 1); k++)
                    get_block(h, (k - 1) * h->cfg.block_size + 1)->x_spans.emplace_back(m.f.c.position, m.f.c.position + m.f.c.length - 1);
                continue;
            }
            if (m.f.c.category == kFoundUnwalked) {   // bases of X / = operations: no candidate, support the allele counts hold and the walk does not
                get_block(h, m.f.c.position)->unwalked.push_back({m.f.c.position, m.f.alt[0], {m.sup[0], m.sup[1], m.sup[2]}});
                continue;
            }
            HostCandidate c = host_candidate_of(m.f.c, h->h_ref.data(), m.f.pool_offset >= 0 ? pool + m.f.pool_offset : m.f.alt);
            for (int d = 0; d < 3; d++) { c.support_by_dir[d] = m.sup[d]; c.well_anchored_by_dir[d] = m.anch[d]; }
            c.stamp = ((uint64_t)h->found.batch << 32) | (uint64_t)(uint32_t)i;
            c.from_reads = true;
            add_candidate(h, c);
        }
        return PISCES_OK;
    }
    for (int64_t i = 0; i < h->found.n_slots; i++) {
        const DevFound& f = recs[i];
        if (f.c.category == kFoundHole) continue;
        if (f.c.category == kFoundUnwalked) {
            BlockObs::Unwalked u = {f.c.position, f.alt[0], {0, 0, 0}};
            if (f.c.dir < 3) u.sup[f.c.dir] = 1;
            get_block(h, f.c.position)->unwalked.push_back(u);
            continue;
        }
        const uint8_t* bases = f.pool_offset >= 0 ? pool + f.pool_offset : f.alt;
        HostCandidate c = host_candidate_of(f.c, h->h_ref.data(), bases);
        c.stamp = ((uint64_t)h->found.batch << 32) | (uint64_t)(uint32_t)i;
        c.from_reads = true;
        add_candidate(h, c);
    }
    return PISCES_OK;
}

// where the arrays of a read batch lie in the staging pair (host pinned + device copy): the batch crosses PCIe packed, in one piece
struct StageLayout {
    size_t off_pos, off_flags, off_coff, off_cop, off_clen, off_soff, off_bases, off_quals, off_dirs, off_slots, off_deldirs, off_fslots, total;
};
static StageLayout stage_layout(size_t nr, size_t n_cig, size_t n_seq, bool with_dirs, bool with_deldirs)
{
    auto align16 = [](size_t x) { return (x + 15) & ~(size_t)15; };
    StageLayout L;
    L.off_pos = 0;
    L.off_flags = align16(L.off_pos + nr * 4);
    L.off_coff = align16(L.off_flags + nr);
    L.off_cop = align16(L.off_coff + (nr + 1) * 4);
    L.off_clen = align16(L.off_cop + n_cig);
    L.off_soff = align16(L.off_clen + n_cig * 4);
    L.off_bases = align16(L.off_soff + (nr + 1) * 4);
    L.off_quals = align16(L.off_bases + n_seq);
    L.off_dirs = align16(L.off_quals + n_seq);
    L.off_slots = align16(L.off_dirs + (with_dirs ? n_seq : 0));
    L.off_deldirs = align16(L.off_slots + (nr + 1) * 8);
    L.off_fslots = align16(L.off_deldirs + (with_deldirs ? 2 * n_cig : 0));
    L.total = align16(L.off_fslots + (nr + 1) * 4);
    return L;
}

// The arrays of a batch of n_reads reads inside the handle's pinned staging buffer: a host that marshals its reads anyway (the C#
// shim packs Read objects into arrays) writes them here and pisces_hip_add_reads sends them as they lie, without a copy of its own.
int32_t pisces_hip_stage_reads(PiscesHip* h, int32_t n_reads, int64_t n_cigar_ops, int64_t n_bases, int32_t with_directions,
                               int32_t with_deletion_directions, PiscesReadBatch* views)
{
    return abi_guard<int32_t>(h, [&]() -> int32_t {
    if (!h) return PISCES_E_INVALID_ARG;
    if (!views || n_reads < 0 || n_cigar_ops < 0 || n_bases < 0 || n_cigar_ops > 0x7FFFFFFFll || n_bases > 0x7FFFFFFFll)
        return fail(h, PISCES_E_INVALID_ARG, "stage_reads: bad arguments");
    { int32_t rcp = refuse_while_batch_is_open(h, "stage_reads"); if (rcp) return rcp; }
    PISCES_HIP_CHECK(h, hipSetDevice(h->device));
    const StageLayout L = stage_layout((size_t)n_reads, (size_t)n_cigar_ops, (size_t)n_bases, with_directions != 0, with_deletion_directions != 0);
    int32_t rc = stage_reserve(h, L.total);
    if (rc) return rc;
    h->staged_total = L.total;
    uint8_t* st = h->h_stage;
    views->n_reads = n_reads;
    views->position = (const int32_t*)(st + L.off_pos);
    views->flags = st + L.off_flags;
    views->cigar_offset = (const int32_t*)(st + L.off_coff);
    views->cigar_op = st + L.off_cop;
    views->cigar_len = (const uint32_t*)(st + L.off_clen);
    views->seq_offset = (const int32_t*)(st + L.off_soff);
    views->bases = st + L.off_bases;
    views->quals = st + L.off_quals;
    views->directions = with_directions ? st + L.off_dirs : nullptr;
    views->deletion_directions = with_deletion_directions ? st + L.off_deldirs : nullptr;
    return PISCES_OK;
    });
}

static int32_t add_reads_store(PiscesHip* h, const PiscesReadBatch* batch);

int32_t pisces_hip_add_reads(PiscesHip* h, const PiscesReadBatch* batch)
{
    return abi_guard<int32_t>(h, [&]() -> int32_t {
    if (!h) return PISCES_E_INVALID_ARG;
    HostTimer timer(&h->host_time[0]);
    if (validate_batch(batch) != PISCES_OK) return fail(h, PISCES_E_INVALID_ARG, "add_reads: malformed read batch");
    { int32_t rcp = refuse_while_batch_is_open(h, "add_reads"); if (rcp) return rcp; }
    if (batch->n_reads == 0) return PISCES_OK;
    PISCES_HIP_CHECK(h, hipSetDevice(h->device));
    { int32_t rcf = consume_found(h); if (rcf) return rcf; }
    {   // what crosses PCIe for this batch (pisces_hip_transfer_bytes): the arrays as they are handed over
        const int64_t n_ops = batch->cigar_offset[batch->n_reads], n_bases = batch->seq_offset[batch->n_reads];
        h->pcie[0] += (int64_t)batch->n_reads * 13 + 8 + n_ops * (batch->deletion_directions ? 7 : 5) + n_bases * (batch->directions ? 3 : 2);
    }
    if (h->read_path == 1) return add_reads_store(h, batch);
    const int32_t nr = batch->n_reads;
    const int32_t minBQ = h->cfg.min_base_call_quality;
    // ---- (PISCES_HIP_READ_PATH=log: the observation-log chain) host pass over the CIGARs only (never over the bases): argument checks of the reference's walk, the insertion /
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
        if (r.n_cigar > 0 && read_span != r.read_len) return fail(h, PISCES_E_INVALID_ARG, "add_reads: CIGAR does not match the read");   // Read.ValidateCigar (Read.cs:603-605)
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
    const StageLayout L = stage_layout((size_t)nr, n_cig, n_seq, batch->directions != nullptr, batch->deletion_directions != nullptr);
    const size_t off_pos = L.off_pos, off_flags = L.off_flags, off_coff = L.off_coff, off_cop = L.off_cop, off_clen = L.off_clen, off_soff = L.off_soff,
                 off_bases = L.off_bases, off_quals = L.off_quals, off_dirs = L.off_dirs, off_slots = L.off_slots, off_deldirs = L.off_deldirs,
                 off_fslots = L.off_fslots, total = L.total;
    // a batch that pisces_hip_stage_reads laid out lies in the staging buffer already: nothing to reserve, nothing to copy
    const bool staged = h->staged_total == total && h->h_stage && (const uint8_t*)batch->position == h->h_stage + off_pos &&
                        batch->bases == h->h_stage + off_bases && batch->quals == h->h_stage + off_quals;
    h->staged_total = 0;
    int32_t rc = staged ? PISCES_OK : stage_reserve(h, total);
    if (rc) return rc;
    rc = log_reserve(h, ub);
    if (rc) return rc;
    uint8_t* st = h->h_stage;
    auto place = [](uint8_t* dst, const void* src, size_t n) { if (n && (const void*)dst != src) std::memcpy(dst, src, n); };
    place(st + off_pos, batch->position, (size_t)nr * 4);
    place(st + off_flags, batch->flags, (size_t)nr);
    place(st + off_coff, batch->cigar_offset, ((size_t)nr + 1) * 4);
    place(st + off_cop, batch->cigar_op, n_cig);
    place(st + off_clen, batch->cigar_len, n_cig * 4);
    place(st + off_soff, batch->seq_offset, ((size_t)nr + 1) * 4);
    std::memcpy(st + off_slots, slots.data(), ((size_t)nr + 1) * 8);
    if (batch->deletion_directions) place(st + off_deldirs, batch->deletion_directions, 2 * n_cig);
    {
        // bases / qualities / directions are the bulk (2-3 bytes per aligned base).  Small batches: one copy into the pinned buffer,
        // one transfer.  Large ones: slices of 8 MB, each copied by a few threads and handed to the DMA engine as soon as it is
        // complete, so that the host copy of slice k+1 runs under the PCIe transfer of slice k.  A staged batch: one transfer.
        struct Seg { size_t dst; const uint8_t* src; size_t len; };
        const Seg segs[3] = {{off_bases, batch->bases, n_seq}, {off_quals, batch->quals, n_seq},
                             {off_dirs, batch->directions, batch->directions ? n_seq : 0}};
        const size_t bulk = 2 * n_seq + (batch->directions ? n_seq : 0);
        constexpr size_t kSlice = (size_t)8 << 20;
        if (staged || bulk < 2 * kSlice) {
            for (const Seg& g : segs) place(st + g.dst, g.src, g.len);
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
    h->eqx_in_batch = false;
    for (int32_t i = 0; i < nr; i++) {
        ReadView r = read_view(batch, i);
        fslots[(size_t)i] = (int32_t)found_slots;
        if (find_on_device && !h->snv_walk)
            for (int c = 0; c < r.n_cigar; c++) {
                if (r.cigar_op[c] == 'I' || r.cigar_op[c] == 'D') found_slots++;
                if (r.cigar_op[c] == 'I' && r.cigar_len[c] > (uint32_t)kFoundInline) found_pool += r.cigar_len[c];
                if (r.cigar_op[c] == 'X' || r.cigar_op[c] == '=') h->eqx_in_batch = true;
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
    hipLaunchKernelGGL(expand_reads_kernel, dim3(expand_reads_grid(nr)), dim3(256), 0, h->stream, db, (const long long*)(d + off_slots), 0ll,
                       minBQ, h->d_log_pos[c].p, h->d_log_tup[c].p, h->d_log_n.p + 2, expand_reads_per_wave(nr));
    { hipError_t el = hipGetLastError(); if (el != hipSuccess) { (void)stage_release(h); return fail(h, PISCES_E_DEVICE, std::string("add_reads: ") + hipGetErrorString(el)); } }
    if (find_on_device && (h->snv_walk || found_slots > 0 || h->eqx_in_batch)) {
        int32_t rcd = enqueue_candidate_discovery(h, db, batch->deletion_directions ? d + off_deldirs : nullptr, nr, (const int32_t*)(d + off_fslots),
                                                  found_slots, found_pool);
        if (rcd) { (void)stage_release(h); return rcd; }   // (transfers out of the staging pair are in flight)
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
    { int32_t rcd = finish_candidate_discovery(h); if (rcd) return rcd; }   // (this entry uses the discovery buffers)
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
    PISCES_HIP_CHECK(h, d_cnt.reserve((size_t)nr + 1));
    PISCES_HIP_CHECK(h, d_pool_first.reserve((size_t)nr + 1));
    PISCES_HIP_CHECK(h, d_totals.reserve(2));
    PISCES_HIP_CHECK(h, d_misc.reserve(4));
    PISCES_HIP_CHECK(h, hipMemsetAsync(d_misc.p, 0, 4 * sizeof(unsigned int), h->stream));
    PISCES_HIP_CHECK(h, hipMemsetAsync(d_cnt.p + nr, 0, sizeof(int32_t), h->stream));
    PISCES_HIP_CHECK(h, hipMemsetAsync(d_pool_first.p + nr, 0, sizeof(int32_t), h->stream));
    launch_find_count(h, db, dd, FP, nr, d_cnt.p, d_pool_first.p);
    { int32_t rcs = launch_found_scan(h, d_cnt.p, d_pool_first.p, nr + 1, d_totals.p); if (rcs) return rcs; }
    long long totals[2] = {0, 0};
    PISCES_HIP_CHECK(h, hipMemcpyAsync(totals, d_totals.p, sizeof(totals), hipMemcpyDeviceToHost, h->stream));
    PISCES_HIP_CHECK(h, hipStreamSynchronize(h->stream));
    std::vector<HostCandidate> found;
    if (totals[0] > 0) {
        PISCES_HIP_CHECK(h, d_out.reserve((size_t)totals[0]));
        PISCES_HIP_CHECK(h, d_pool.reserve((size_t)totals[1] + 16));
        launch_find_emit(h, db, dd, FP, nr, (const int32_t*)d_cnt.p, (const int32_t*)d_pool_first.p, d_out.p, d_pool.p, d_misc.p, (int32_t)totals[1]);
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

