// surface_flush.inc.h — part of pisces_hip.hip (included there, inside its extern "C" block; not a translation unit of its own).
// IAlleleCaller.Call behind the C ABI: tile geometry and bucketing of the observation log, DoneProcessing's log compaction, the device
// calls of a batch (call_blocks), VariantCollapser / MnvReallocator / the candidate kernel (call_spanning), pisces_hip_flush[_ex] with the
// block schedule, and the IAlleleSource read-backs (counts, base-quality sums, gapped-MNV reference counts, candidates, statistics).

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
        else {
            // (sorted, disjoint intervals: the first one that ends at or behind the block's start, then on while they start inside it)
            auto it = std::lower_bound(h->intervals.begin(), h->intervals.end(), bstart,
                                       [](const std::pair<int32_t, int32_t>& iv, int32_t p) { return iv.second < p; });
            for (; it != h->intervals.end() && it->first <= bend; ++it) {
                int32_t s = std::max(it->first, bstart), e = std::min(it->second, bend);
                if (s <= e) add_range(s, e);
            }
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
    const size_t n_tables = 2 * n_slot + keys.size() + tol.size();   // key_slot, key_first_tile, first_tile, tile_of_locus
    host.assign(n_tables, -1);
    for (size_t i = 0; i < keys.size(); i++) {
        host[(size_t)(keys[i] - kmin)] = (int32_t)i;
        host[n_slot + (size_t)(keys[i] - kmin)] = first_tile[i];
    }
    std::copy(first_tile.begin(), first_tile.end(), host.begin() + (std::ptrdiff_t)(2 * n_slot));
    std::copy(tol.begin(), tol.end(), host.begin() + (std::ptrdiff_t)(2 * n_slot + keys.size()));
    host.resize(n_tables + n_zero_tail, 0);
    PISCES_HIP_CHECK(h, h->d_bucket.reserve(host.size()));
    if (zero_tail) *zero_tail = (unsigned int*)(h->d_bucket.p + n_tables);
    { int32_t rcu = meta_upload(h, h->d_bucket.p, host.data(), host.size() * sizeof(int32_t)); if (rcu) return rcu; }
    m->key_slot = h->d_bucket.p;
    m->key_first_tile = h->d_bucket.p + n_slot;
    m->first_tile = h->d_bucket.p + 2 * n_slot;
    m->tile_of_locus = tol.empty() ? nullptr : h->d_bucket.p + 2 * n_slot + keys.size();
    m->inv_block_size = 1.0 / (double)h->cfg.block_size;
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
    if (h->log_ub == 0) {
        // nothing in the log (the reads are in the read store, or there are none): tiles without tuple segments, no bucketing
        PISCES_HIP_CHECK(h, h->d_tiles.reserve(tiles.size()));
        PISCES_HIP_CHECK(h, h->d_tile_results.reserve(tiles.size()));
        PISCES_HIP_CHECK(h, h->d_count.reserve(4));
        return meta_upload(h, h->d_tiles.p, tiles.data(), tiles.size() * sizeof(PiscesTile));
    }
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
static int32_t enqueue_drop(PiscesHip* h, const std::vector<int32_t>& keys, int64_t hole_bound = -1)
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
    if (hole_bound > 0)   // (pisces_hip_flush_begin: the host will not wait for the count; the new log is hole_bound slots long, the rest holes)
        hipLaunchKernelGGL(log_fill_holes_kernel, dim3((unsigned)((hole_bound + 255) / 256)), dim3(256), 0, h->stream, h->d_log_pos[o].p,
                           (const unsigned long long*)(h->d_log_n.p + o), (long long)hole_bound);
    PISCES_HIP_CHECK(h, hipGetLastError());
    return PISCES_OK;
}
static void commit_drop(PiscesHip* h, unsigned long long kept)
{
    h->log_cur ^= 1;
    h->log_ub = (int64_t)kept;
    h->log_known_holes = 0;
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
        // (without events the plain launch: it is what a stream capture records as a kernel node)
        if (!two && !e0 && !e1)
            hipLaunchKernelGGL(call_tiles_wave_kernel<1>, dim3((unsigned)n_tiles), dim3(64), lds, s, d_tuples, d_tiles, n_tiles, d_ref, ref_start, ref_len,
                               d_records, d_tr, h->P, (const DeviceParams*)h->d_params.p);
        else if (!two)
            hipExtLaunchKernelGGL(call_tiles_wave_kernel<1>, dim3((unsigned)n_tiles), dim3(64), lds, s, e0, e1, 0u, d_tuples, d_tiles,
                                  n_tiles, d_ref, ref_start, ref_len, d_records, d_tr, h->P, (const DeviceParams*)h->d_params.p);
        else if (!e0 && !e1)
            hipLaunchKernelGGL(call_tiles_wave_kernel<2>, dim3((unsigned)n_tiles), dim3(128), lds, s, d_tuples, d_tiles, n_tiles, d_ref, ref_start, ref_len,
                               d_records, d_tr, h->P, (const DeviceParams*)h->d_params.p);
        else
            hipExtLaunchKernelGGL(call_tiles_wave_kernel<2>, dim3((unsigned)n_tiles), dim3(128), lds, s, e0, e1, 0u, d_tuples, d_tiles,
                                  n_tiles, d_ref, ref_start, ref_len, d_records, d_tr, h->P, (const DeviceParams*)h->d_params.p);
        return hipSuccess;
    }
    hipExtLaunchKernelGGL(call_tiles_kernel, dim3((unsigned)n_tiles), dim3(kBlock), lds, s, e0, e1, 0u, d_tuples, d_tiles, n_tiles, d_ref,
                          ref_start, ref_len, d_records, d_tr, h->P);
    return hipSuccess;
}

// PloidyModel.DiploidByThresholding / Haploid over the record slots a tile kernel has just filled (genotype_loci_kernel), on stream s
static void launch_genotype_loci(PiscesHip* h, hipStream_t s, PiscesCalledAllele* d_records, PiscesTileResult* d_tr, int32_t n_tiles)
{
    GenotypeParams G;
    G.ploidy = h->cfg.ploidy;
    for (int k = 0; k < 3; k++) { G.snv[k] = h->cfg.diploid_snv_params[k]; G.indel[k] = h->cfg.diploid_indel_params[k]; }
    G.min_depth = h->cfg.min_coverage;
    G.min_gq = h->cfg.min_genotype_qscore;
    G.max_gq = h->cfg.max_genotype_qscore;
    G.low_gq_filter = h->cfg.low_gq_filter;
    if (n_tiles > 0) hipLaunchKernelGGL(genotype_loci_kernel, dim3((unsigned)n_tiles), dim3(64), 0, s, d_records, d_tr, n_tiles, G, h->P.totals);
}
static bool germline(const PiscesHip* h) { return h->cfg.ploidy == PISCES_PLOIDY_DIPLOID || h->cfg.ploidy == PISCES_PLOIDY_HAPLOID; }

// scan + gather: d_out = called alleles in (position, allele) order, *d_count = how many
// ONE launch: gather_direct_kernel up to kGatherDirectTiles tiles (a tile's wave adds up the counts before it), compact_records_kernel beyond
// (a decoupled look-back over the workgroups of sixteen tiles; its words carry the launch's epoch, so nothing is cleared between launches).
// PISCES_HIP_COMPACT=two: the scan + gather pair they replace (the A / B).
static int32_t launch_compaction(PiscesHip* h, hipStream_t s, const PiscesCalledAllele* d_records, const PiscesTileResult* d_tr, int32_t n_tiles,
                                 int32_t* d_offsets, PiscesCalledAllele* d_out, int32_t cap, int32_t* d_count, int32_t* d_called = nullptr)
{
    // (the look-back's words belong to the handle: launches on a caller's stream may overlap each other, and two launches sharing the words
    // would wait for each other's epochs for ever — those take the stateless pair)
    if (h->compact_mode == 2 || (n_tiles > kGatherDirectTiles && s != h->stream)) {
        hipLaunchKernelGGL(scan_tile_counts_kernel, dim3(1), dim3(1024), 0, s, d_tr, n_tiles, d_offsets, d_count, d_called);
        hipLaunchKernelGGL(gather_records_kernel, dim3((unsigned)n_tiles), dim3(64), 0, s, d_records, d_tr, n_tiles, d_offsets, d_out, cap);
        return PISCES_OK;
    }
    if (n_tiles <= kGatherDirectTiles && h->compact_mode != 3) {
        hipLaunchKernelGGL(gather_direct_kernel, dim3((unsigned)n_tiles), dim3(64), 0, s, d_records, d_tr, n_tiles, d_offsets, d_out, cap, d_count, d_called);
        return PISCES_OK;
    }
    const size_t groups = ((size_t)n_tiles + kCompactTiles - 1) / kCompactTiles;
    {   // the look-back's words: a new buffer starts zeroed (epoch 0 is never used); the epoch wraps after 2^30 launches: cleared again then
        const size_t before = h->d_compact_state.cap;
        PISCES_HIP_CHECK(h, h->d_compact_state.reserve(groups));
        if (h->d_compact_state.cap != before || h->compact_epoch >= (1u << 30) - 1u) {
            // (launches of earlier epochs on other streams are not waited for: a buffer only grows under the stream it is used on — the handle's)
            PISCES_HIP_CHECK(h, hipMemsetAsync(h->d_compact_state.p, 0, h->d_compact_state.cap * sizeof(unsigned long long), s));
            h->compact_epoch = 0;
        }
    }
    const uint32_t epoch = ++h->compact_epoch;
    hipLaunchKernelGGL(compact_records_kernel, dim3((unsigned)groups), dim3(256), 0, s, d_records, d_tr, n_tiles, h->d_compact_state.p, epoch, d_out, cap, d_count, d_called,
                       d_offsets);
    return PISCES_OK;
}

// device work of one flush: returns called alleles of `keys` sorted by (position, ref, alt)
// The device work of one flush in two halves: everything up to the copies into the pinned download buffer is enqueued by
// call_blocks_enqueue; call_blocks_finish reads what came back once the stream (or an event behind the copies) has been waited for.
struct CallBlocksInFlight {
    bool active = false;       // false: nothing was enqueued (no blocks, no tiles)
    bool drop_now = false;
    int32_t* hdr = nullptr;
    PiscesCalledAllele* hrec = nullptr;
    size_t spec = 0;
    const int32_t* extra = nullptr;   // {records, called} of the counting launch over the off-interval loci (exact TotalNumCalled), or nullptr
};
// pisces_hip_set_exact_total_called: the loci of `keys`' blocks that lie OUTSIDE the interval set, through the flush kernel once more with
// DeviceParams::variants_only (no Reference records; variants exactly as inside the intervals: dirty loci are skipped there and counted by
// the candidate path, as everywhere), the records dropped, the tiles' n_called summed into pinned memory behind the same wait.
static int32_t enqueue_off_interval_count(PiscesHip* h, const std::vector<int32_t>& keys, CallBlocksInFlight* st)
{
    const int bs = h->cfg.block_size;
    std::vector<PiscesTile> tiles;
    auto add_range = [&](int32_t s, int32_t e) {
        for (int32_t p = s; p <= e; p += kTile) {
            PiscesTile t;
            t.start_position = p;
            t.n_loci = std::min<int32_t>(kTile, e - p + 1);
            t.tuple_begin = t.tuple_end = 0;
            tiles.push_back(t);
        }
    };
    for (int32_t key : keys) {
        const int32_t bstart = (key - 1) * bs + 1, bend = (int32_t)std::min<int64_t>((int64_t)key * bs, h->ref_len);
        int32_t at = bstart;
        auto it = std::lower_bound(h->intervals.begin(), h->intervals.end(), bstart, [](const std::pair<int32_t, int32_t>& iv, int32_t p) { return iv.second < p; });
        for (; it != h->intervals.end() && it->first <= bend; ++it) {
            if (it->first > at) add_range(at, std::min(it->first - 1, bend));
            at = std::max(at, it->second + 1);
        }
        if (at <= bend) add_range(at, bend);
    }
    if (tiles.empty()) return PISCES_OK;
    if (tiles.size() > (size_t)(0x7FFFFF00ll / kSlotsPerTile))
        return fail(h, PISCES_E_UNSUPPORTED, "flush: pisces_hip_set_exact_total_called: more off-interval tiles in one flush than a launch's record slots hold (flush fewer blocks at a time)");
    const int32_t n = (int32_t)tiles.size();
    PISCES_HIP_CHECK(h, h->d_tiles_x.reserve(tiles.size()));
    PISCES_HIP_CHECK(h, h->d_tr_x.reserve(tiles.size()));
    PISCES_HIP_CHECK(h, h->d_rec_x.reserve(tiles.size() * kSlotsPerTile));
    PISCES_HIP_CHECK(h, h->d_off_x.reserve(tiles.size()));
    PISCES_HIP_CHECK(h, h->d_cnt_x.reserve(4));
    if (!h->h_cnt_x) PISCES_HIP_CHECK(h, host_alloc((void**)&h->h_cnt_x, 64));
    { int32_t rcu = meta_upload(h, h->d_tiles_x.p, tiles.data(), tiles.size() * sizeof(PiscesTile)); if (rcu) return rcu; }
    const DeviceParams saved = h->P;
    h->P.variants_only = 1;
    h->P.totals = nullptr;
    h->P.folded_out = nullptr;
    h->P.folded_first = h->P.folded_n = 0;
    const RegularTiles R = {0, bs, (bs + kTile - 1) / kTile, 0};
    const hipError_t e = launch_call_store_tiles(h, h->stream, nullptr, h->d_tiles_x.p, R, n, h->d_ref.p, 1, h->ref_len, h->d_rec_x.p, h->d_tr_x.p);
    h->P = saved;
    PISCES_HIP_CHECK(h, e);
    hipLaunchKernelGGL(scan_tile_counts_kernel, dim3(1), dim3(1024), 0, h->stream, (const PiscesTileResult*)h->d_tr_x.p, n, h->d_off_x.p, h->d_cnt_x.p, h->d_cnt_x.p + 1);
    PISCES_HIP_CHECK(h, hipGetLastError());
    PISCES_HIP_CHECK(h, hipMemcpyAsync(h->h_cnt_x, h->d_cnt_x.p, 2 * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
    st->extra = h->h_cnt_x;
    return PISCES_OK;
}

// hole_bound >= 0 (asynchronous flush): the compacted log is made hole_bound slots long (see enqueue_drop)
// with_folded: the launch — when it is the fused kernel over a run of whole blocks — also leaves the folded counts of every locus it walks
// in h->d_folded (h->fold says which positions), for the candidate kernel of the same flush
// genotype_on_device: a diploid / haploid handle's rows are genotyped where they lie (genotype_loci_kernel) before they are compacted:
// the caller knows that no row of the candidate kernel and no forced allele joins them
static int32_t call_blocks_enqueue(PiscesHip* h, const std::vector<int32_t>& keys, bool with_drop, int64_t hole_bound, CallBlocksInFlight* st, bool with_folded = false,
                                   bool genotype_on_device = false)
{
    *st = CallBlocksInFlight();
    h->fold.valid = false;
    h->pending_view = nullptr;
    h->pending_view_n = 0;
    if (keys.empty()) return PISCES_OK;
    if (!h->d_ref.p) return fail(h, PISCES_E_STATE, "flush: set_reference has not been called");
    const bool window = h->cfg.noise_model == PISCES_NOISE_WINDOW;
    bool use_counts = false;
    for (auto& kv : h->gapped_mnv_ref)
        if (std::binary_search(keys.begin(), keys.end(), block_key(h, kv.first))) { use_counts = true; break; }
    // the read store calls through call_store_tiles_kernel; configurations that kernel is not compiled for (the Diploid strand-bias model,
    // the 4-wave development form, a quality threshold above 127) go through the counts in HBM
    const bool store = h->read_path == 1;
    const bool store_fused = h->kernel_variant >= 2 && h->cfg.strand_bias_model != PISCES_SB_DIPLOID && h->cfg.min_base_call_quality <= 127;
    const bool fused = !use_counts && !window && store && store_fused;
    // A run of consecutive blocks that no interval set clips, nothing in the observation log: the tile geometry goes to the kernel by
    // value (RegularTiles) — no geometry is made on the host and none uploaded
    const int bs = h->cfg.block_size;
    int tiles_per_block = (bs + kTile - 1) / kTile;
    const bool regular = fused && h->log_ub == 0 && h->intervals.empty() && (int64_t)keys.back() - keys.front() + 1 == (int64_t)keys.size() &&
                         (int64_t)keys.size() * tiles_per_block < 0x7FFFFF00ll / kSlotsPerTile;
    // The tiles of a regular run may be of any size up to 64 (RegularTiles::tile_loci).  A launch ends with the CU that was dealt the most
    // loci — ceil(tiles / CUs) tiles — and 100 blocks on 256 CUs put 7 x 64 = 448 loci on 64 CUs with tiles of 64 but 7 x 59 = 413 with
    // tiles of 59; measured (round 6, config 2's flush, tools/tile_loci_ab.sh): 64 -> 50.8-51.3 us for the flush's span, 59 -> 51.4,
    // 56 / 60 / 62 -> 54.5 / 53.2 / 53.8: the trade of tiles inside the kernel has taken that imbalance already.  64 stays; PISCES_HIP_TILE_LOCI forces a size.
    int tile_loci = kTile;
    if (regular && h->tile_loci > 0 && h->tile_loci < kTile && (int64_t)keys.size() * ((bs + h->tile_loci - 1) / h->tile_loci) < 0x7FFFFF00ll / kSlotsPerTile) {
        tile_loci = h->tile_loci;
        tiles_per_block = (bs + tile_loci - 1) / tile_loci;
    }
    std::vector<PiscesTile> tiles;
    int32_t n_tiles = 0;
    int64_t n_loci_total = 0;
    RegularTiles R = {0, bs, tiles_per_block, tile_loci};
    if (regular) {
        R.first_key = keys.front();
        n_tiles = (int32_t)(keys.size() * (size_t)tiles_per_block);
        n_loci_total = (int64_t)keys.size() * bs;
        PISCES_HIP_CHECK(h, h->d_tile_results.reserve((size_t)n_tiles));
        PISCES_HIP_CHECK(h, h->d_count.reserve(4));
    } else {
        int32_t rc = bucket_blocks(h, keys, true, tiles);
        if (rc) return rc;
        if (tiles.empty()) return PISCES_OK;
        n_tiles = (int32_t)tiles.size();
        for (auto& t : tiles) n_loci_total += t.n_loci;
    }
    const size_t cap = (size_t)n_tiles * kSlotsPerTile;   // slot layout: 256 slots per tile
    PISCES_HIP_CHECK(h, h->d_records.reserve(cap));
    PISCES_HIP_CHECK(h, h->d_compact.reserve(cap + 1));   // ([0]: the header of a small launch's compaction)
    PISCES_HIP_CHECK(h, h->d_offsets.reserve((size_t)n_tiles));
    if (h->chain_timing) { h->chain_have[1] = false; PISCES_HIP_CHECK(h, hipEventRecord(h->ev_chain[2], h->stream)); }

    std::vector<uint32_t> g;
    if (fused) {
        // (pisces_hip_set_timing: the dispatch's own start / stop events, as for pisces_hip_call_tiles)
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (h->timing > 0 && (h->launches_seen++ % h->timing) == 0) {
            const size_t slot = (size_t)(h->ring_used % kTimingRing);
            e0 = h->ring[2 * slot];
            e1 = h->ring[2 * slot + 1];
            h->ring_used++;
        }
        if (with_folded && regular) {
            PISCES_HIP_CHECK(h, h->d_folded.reserve((size_t)n_loci_total * PISCES_FOLDED_PER_LOCUS));
            h->P.folded_out = h->d_folded.p;
            h->P.folded_first = (keys.front() - 1) * bs + 1;
            h->P.folded_n = (int32_t)n_loci_total;
            h->fold.valid = true;
            h->fold.lo = h->P.folded_first;
            h->fold.hi = h->P.folded_first + (int32_t)n_loci_total - 1;
        }
        const hipError_t el = launch_call_store_tiles(h, h->stream, regular ? nullptr : h->d_tuples.p, regular ? nullptr : h->d_tiles.p, R, n_tiles, h->d_ref.p, 1,
                                                      h->ref_len, h->d_records.p, h->d_tile_results.p, e0, e1);
        h->P.folded_out = nullptr;
        h->P.folded_first = h->P.folded_n = 0;
        PISCES_HIP_CHECK(h, el);
    } else if (!use_counts && !window && !store) {
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
        PISCES_HIP_CHECK(h, accumulate_tiles(h, h->stream, h->d_tuples.p, h->d_tiles.p, n_tiles, window, true));
        hipLaunchKernelGGL(call_counts_kernel, dim3((unsigned)n_tiles), dim3(kBlock), 0, h->stream, h->d_counts.p, h->d_gapped.p,
                           h->d_tiles.p, n_tiles, h->d_ref.p, 1, h->ref_len, h->d_records.p, h->d_tile_results.p, h->P,
                           window ? h->d_sumq.p : (const double*)nullptr);
    }
    if (genotype_on_device) launch_genotype_loci(h, h->stream, h->d_records.p, h->d_tile_results.p, n_tiles);
    // tiles were built in ascending position order: the ordered compaction is AlleleCaller.Call's (position, ref, alt) order.
    // The sorted records lie behind one header slot {records, called}.  A launch of up to 64 tiles (the blocks of one flush of the
    // streaming protocol) is compacted by one kernel that writes header and records into the pinned download buffer itself.
    const bool small = n_tiles <= 64;
    // one synchronisation in the usual case: the two counters and a speculative prefix of the sorted records (one per locus plus
    // a quarter) come back together into pinned memory; a second copy only when more alleles were called than that
    const size_t spec = small ? cap : std::min<size_t>(cap, (size_t)(n_loci_total + n_loci_total / 4 + 64));
    // DoneProcessing's kernel rides in the same submission (it only writes the OTHER log buffer): one synchronisation per flush
    const bool drop_now = with_drop && h->log_ub > 0;
    const size_t dl_bytes = (cap + 1) * sizeof(PiscesCalledAllele);
    if (dl_bytes > h->h_dl_cap) {
        if (h->async.state == 1) PISCES_HIP_CHECK(h, hipStreamSynchronize(h->stream));   // (never: a flush in flight owns the buffer; flush_begin refuses)
        if (h->h_dl) host_free(h->h_dl);
        h->h_dl = nullptr;
        h->h_dl_cap = 0;
        PISCES_HIP_CHECK(h, host_alloc((void**)&h->h_dl, dl_bytes + dl_bytes / 2));
        h->h_dl_cap = dl_bytes + dl_bytes / 2;
    }
    // the pinned download buffer mirrors d_compact: a header slot {records, called, kept (8 bytes)}, then the records
    int32_t* hdr = (int32_t*)h->h_dl;
    PiscesCalledAllele* hrec = (PiscesCalledAllele*)h->h_dl + 1;
    if (small) {
        // straight into the pinned buffer (host memory the device can write): the kernel's stores are the transfer
        hipLaunchKernelGGL(compact_small_kernel, dim3((unsigned)n_tiles), dim3(64), 0, h->stream, (const PiscesCalledAllele*)h->d_records.p,
                           (const PiscesTileResult*)h->d_tile_results.p, n_tiles, (PiscesCalledAllele*)h->h_dl, (int32_t)cap);
        if (h->chain_timing) { PISCES_HIP_CHECK(h, hipEventRecord(h->ev_chain[3], h->stream)); h->chain_have[1] = true; }
    } else {
        { int32_t rcc = launch_compaction(h, h->stream, h->d_records.p, h->d_tile_results.p, n_tiles, h->d_offsets.p, h->d_compact.p + 1, (int32_t)cap, h->d_count.p,
                                          h->d_count.p + 1); if (rcc) return rcc; }
        if (h->chain_timing) { PISCES_HIP_CHECK(h, hipEventRecord(h->ev_chain[3], h->stream)); h->chain_have[1] = true; }
        PISCES_HIP_CHECK(h, hipMemcpyAsync(hdr, h->d_count.p, 2 * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
        PISCES_HIP_CHECK(h, hipMemcpyAsync(hrec, h->d_compact.p + 1, spec * sizeof(PiscesCalledAllele), hipMemcpyDeviceToHost, h->stream));
    }
    PISCES_HIP_CHECK(h, hipGetLastError());
    if (drop_now) {
        int32_t rcd = enqueue_drop(h, keys, hole_bound);
        if (rcd) return rcd;
    }
    if (drop_now)
        PISCES_HIP_CHECK(h, hipMemcpyAsync(hdr + 2, h->d_log_n.p + (h->log_cur ^ 1), sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream));
    st->active = true;
    st->drop_now = drop_now;
    st->hdr = hdr;
    st->hrec = hrec;
    st->spec = spec;
    // pisces_hip_set_exact_total_called: with an interval set and the SNVs taken from the allele counts (MNV calling and collapsing off:
    // otherwise they are candidates, and the candidate path counts the callable ones outside the intervals itself) the off-interval loci
    // get a counting launch of the flush kernel.  A configuration that kernel does not serve cannot give the reference's number: refused
    // here, not answered with the smaller one.
    if (h->exact_total_called && !h->intervals.empty() && !h->snv_walk) {
        if (!(fused && h->log_ub == 0))
            return fail(h, PISCES_E_UNSUPPORTED, "flush: pisces_hip_set_exact_total_called is on, but this configuration does not go through the read store's "
                        "flush kernel (NoiseModel.Window, the Diploid strand-bias model, a gapped-MNV reference count in the flushed blocks, the "
                        "observation-log read path or a base-quality threshold above 127): the off-interval total cannot be counted; switch it off");
        int32_t rcx = enqueue_off_interval_count(h, keys, st);
        if (rcx) return rcx;
    }
    return PISCES_OK;
}
// Does a flush of `keys` go through the fused kernel over a run of whole blocks (call_blocks_enqueue's `fused && regular`)?  Then its
// launch can go first and leave the folded counts for the candidate kernel.
static bool fused_regular_applies(PiscesHip* h, const std::vector<int32_t>& keys)
{
    if (keys.empty() || !h->d_ref.p || h->cfg.noise_model == PISCES_NOISE_WINDOW || h->read_path != 1 || h->log_ub != 0 || !h->intervals.empty()) return false;
    if (germline(h)) return false;   // (whether the rows are genotyped on the device is only known once the candidates have been through)
    if (!(h->kernel_variant >= 2 && h->cfg.strand_bias_model != PISCES_SB_DIPLOID && h->cfg.min_base_call_quality <= 127)) return false;
    for (auto& kv : h->gapped_mnv_ref)
        if (std::binary_search(keys.begin(), keys.end(), block_key(h, kv.first))) return false;
    const int tiles_per_block = (h->cfg.block_size + kTile - 1) / kTile;
    return (int64_t)keys.back() - keys.front() + 1 == (int64_t)keys.size() && (int64_t)keys.size() * tiles_per_block < 0x7FFFFF00ll / kSlotsPerTile &&
           (int64_t)keys.size() * h->cfg.block_size < (0x7FFFFF00ll / PISCES_FOLDED_PER_LOCUS);
}

// (after the wait) total: the records; they lie in st.hrec, all of them
static int32_t call_blocks_finish(PiscesHip* h, const CallBlocksInFlight& st, int32_t* total, int64_t* n_called, unsigned long long* kept)
{
    *total = 0;
    if (!st.active) return PISCES_OK;
    *total = st.hdr[0];
    h->pcie[1] += (int64_t)*total * (int64_t)sizeof(PiscesCalledAllele) + 16;
    *n_called += st.hdr[1];
    if (st.extra) *n_called += st.extra[1];   // callable SNVs of the loci outside the intervals (pisces_hip_set_exact_total_called)
    if (st.drop_now && kept) std::memcpy(kept, st.hdr + 2, sizeof(unsigned long long));
    if ((size_t)*total > st.spec) {
        PISCES_HIP_CHECK(h, hipMemcpyAsync(st.hrec + st.spec, h->d_compact.p + 1 + st.spec, ((size_t)*total - st.spec) * sizeof(PiscesCalledAllele),
                                           hipMemcpyDeviceToHost, h->stream));
        PISCES_HIP_CHECK(h, hipStreamSynchronize(h->stream));
    }
    return PISCES_OK;
}

#ifdef PISCES_STORE_TIMING
// development: the fused kernel's clock stamps (every tile's last record slot) to the file PISCES_HIP_DUMP_TILE_RESULTS names (tools/store_timing.py)
static void dump_tile_stamps(PiscesHip* h)
{
    const char* path = getenv("PISCES_HIP_DUMP_TILE_RESULTS");
    if (!path) return;
    std::vector<PiscesCalledAllele> tr(h->d_tile_results.cap);
    (void)hipMemcpy2D(tr.data(), sizeof(PiscesCalledAllele), h->d_records.p + (kSlotsPerTile - 1), (size_t)kSlotsPerTile * sizeof(PiscesCalledAllele),
                      sizeof(PiscesCalledAllele), std::min(tr.size(), h->d_records.cap / kSlotsPerTile), hipMemcpyDeviceToHost);
    if (FILE* fp = fopen(path, "wb")) { fwrite(tr.data(), sizeof(PiscesCalledAllele), tr.size(), fp); fclose(fp); }
}
#endif
// device work of one flush: returns called alleles of `keys` sorted by (position, ref, alt)
// as_view: the records are not copied into `out`; h->pending_view points at them in the pinned download buffer (valid until the next
// call_blocks)
static int32_t call_blocks(PiscesHip* h, const std::vector<int32_t>& keys, std::vector<PiscesCalledAllele>& out, int64_t* n_called,
                           bool with_drop = false, bool* dropped = nullptr, unsigned long long* kept = nullptr, bool as_view = false, bool genotype_on_device = false)
{
    out.clear();   // (*n_called accumulates: the caller zeroes it)
    if (dropped) *dropped = false;
    CallBlocksInFlight st;
    int32_t rc = call_blocks_enqueue(h, keys, with_drop, -1, &st, false, genotype_on_device);
    if (rc || !st.active) return rc;
    PISCES_TIMED_WAIT(h, hipStreamSynchronize(h->stream));
    h->h_meta_used = 0;   // the stream is idle: nothing reads the arena any more
#ifdef PISCES_STORE_TIMING
    dump_tile_stamps(h);
#endif
    int32_t total = 0;
    rc = call_blocks_finish(h, st, &total, n_called, kept);
    if (rc) return rc;
    if (st.drop_now && dropped) *dropped = true;
    if (as_view) { h->pending_view = st.hrec; h->pending_view_n = (size_t)total; }
    else out.assign(st.hrec, st.hrec + total);
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
struct MnvArena {   // owns every object the reallocation creates (in chunks: a flush of thirty blocks at 2000x makes ~5 000 of them)
    static constexpr size_t kChunk = 512;
    std::vector<std::unique_ptr<HostCandidate[]>> chunks;
    size_t used = kChunk;
    CandPtr make(int32_t position, const std::string& alt, const std::string& ref, const int32_t* dirs)   // CreateVariant :151-168
    {
        if (used == kChunk) { chunks.emplace_back(new HostCandidate[kChunk]); used = 0; }
        CandPtr v = &chunks.back()[used++];
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
    // the callable alleles by position (their places in `callable`, ascending): IsPotentialOverlap only looks at alleles that start on
    // the failed allele's few positions, and a batch of thirty blocks holds ~10^5 callable alleles (a scan of all of them per failed
    // allele made reallocation 52 of the 65 ms of such a flush)
    // (a sorted array of (position, place) for the alleles there are at the start — thousands in a large batch, no allocation each —
    // and a second one for the SNVs that BreakDownToSingleNucCalls adds on the way)
    std::vector<std::pair<int32_t, uint32_t>> by_position(callable.size());
    for (size_t i = 0; i < callable.size(); i++) by_position[i] = {callable[i]->position, (uint32_t)i};
    std::sort(by_position.begin(), by_position.end());
    std::vector<std::pair<int32_t, uint32_t>> added;   // (position, place) of those, kept sorted: they arrive nearly in position order
    std::vector<uint32_t> places;
    std::vector<CandPtr> remainderAlleles, overlaps;   // (reused: a failed MNV is a handful of alleles, a batch hundreds of failed MNVs)
    for (CandPtr failedMnv : ordered) {
        remainderAlleles.assign(1, failedMnv);
        while (!remainderAlleles.empty()) {
            CandPtr alleleToReassign = remainderAlleles.front();
            const int fl = (int)alleleToReassign->alt.size();
            overlaps.clear();
            places.clear();
            {
                auto lo = std::lower_bound(by_position.begin(), by_position.end(), std::make_pair(alleleToReassign->position, 0u));
                for (; lo != by_position.end() && lo->first <= alleleToReassign->position + fl; ++lo) places.push_back(lo->second);
            }
            if (!added.empty()) {
                auto lo = std::lower_bound(added.begin(), added.end(), std::make_pair(alleleToReassign->position, 0u));
                for (; lo != added.end() && lo->first <= alleleToReassign->position + fl; ++lo) places.push_back(lo->second);
            }
            std::sort(places.begin(), places.end());   // the order of `callable`: what the stable sort below keeps among equals
            for (uint32_t i : places) {   // IsPotentialOverlap :250-261
                CandPtr c = callable[i];
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
                    else {
                        const std::pair<int32_t, uint32_t> e(sn->position, (uint32_t)callable.size());
                        added.insert(std::upper_bound(added.begin(), added.end(), e), e);
                        callable.push_back(sn);
                    }
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
static int64_t collapse_candidates(std::vector<HostCandidate>& cands, float freq_threshold, float freq_ratio_threshold, FreqFn freq,
                                   const std::vector<HostCandidate>* known_variants = nullptr, bool exclude_mnvs = false)
{
    const size_t n = cands.size();
    std::vector<uint8_t> removed(n, 0);
    // excludeMNVs (VariantCollapser.cs:33): MNV candidates are no targets — not annotated, not collapsed, nothing collapses into them
    auto excluded = [&](size_t i) { return exclude_mnvs && cands[i].category == PISCES_CAT_MNV; };
    // AnnotateKnown (VariantCollapser.cs:178-190): a candidate that equals a known (prior) variant of the chromosome is known, and anchored on
    // both sides whatever its reads said
    std::vector<uint8_t> known(n, 0);
    if (known_variants && !known_variants->empty()) {
        const bool sorted = std::is_sorted(known_variants->begin(), known_variants->end(),
                                           [](const HostCandidate& a, const HostCandidate& b) { return a.position < b.position; });   // (pisces_hip_set_known_variants sorts)
        for (size_t i = 0; i < n; i++) {
            if (excluded(i)) continue;
            auto k = known_variants->begin(), k_end = known_variants->end();
            if (sorted) {   // the known variants at the candidate's own position only
                k = std::lower_bound(k, k_end, cands[i].position, [](const HostCandidate& a, int32_t p) { return a.position < p; });
                k_end = std::upper_bound(k, k_end, cands[i].position, [](int32_t p, const HostCandidate& a) { return p < a.position; });
            }
            for (; k != k_end; ++k)
                if (cand_equals(cands[i], *k)) { known[i] = 1; cands[i].open_left = cands[i].open_right = false; break; }
        }
    }
    std::vector<size_t> order;
    for (size_t i = 0; i < n; i++)
        if (!excluded(i) && (cands[i].open_left || cands[i].open_right)) order.push_back(i);
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
            if (j != oi && !removed[j] && !excluded(j) && can_collapse(t, cands[j])) rows.push_back({j, freq(cands[j])});
        if (rows.empty()) continue;
        const float tf = freq(t);
        // IComparer.Compare :214-244; input order breaks the remaining ties
        std::stable_sort(rows.begin(), rows.end(), [&](const Row& x, const Row& y) {
            const HostCandidate& a = cands[x.idx];
            const HostCandidate& b = cands[y.idx];
            if (known[x.idx] != known[y.idx]) return known[x.idx] != 0;   // "return known one first" :216-218
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

// ------------------------------------------------------------------------------------------------
// MNV calling on, SPLIT FORM.  With -callmnvs every mismatch of every read is an SNV candidate of the read walk
// (CandidateVariantFinder.cs:90-168) — ~1.5 merged candidates a locus at 2000x, nearly all of them single-read errors that are never
// called — and the reference calls SNVs from those candidates because a base that an MNV candidate took is no SNV candidate any more.
// Where no such thing happened the candidate IS the allele count: support by direction = the quality-passing bases that show the allele,
// which is what the tile kernels call SNVs from when MNV calling is off.  So a flush parts the loci of its blocks:
//   DIRTY loci   an MNV candidate spans the locus; an open-ended SNV / MNV candidate sits on it (the collapser may move its support); a
//                candidate the host added itself lies there (what a failed MNV left for the next block, a caller's candidate); an X
//                operation of some read covers it (counted, but no candidate: ProcessCigarOps walks M operations only); it lies outside
//                the interval set (calls there are counted, not reported).  There every candidate goes the way it always went: host
//                objects, collapser, reallocator, candidate kernel — the SNV groups of those loci are taken from the device's SNV store
//                (snv_store_sweep_kernel), and the tile kernels emit the Reference record only (DeviceParams::dirty_bits).
//   the rest     SNVs from the allele counts in the tile kernels, as with MNV calling off; their groups in the store are dropped unseen.
// The host's work per flush follows the dirty loci (hundreds a block at most), not the candidates (thousands).
// ------------------------------------------------------------------------------------------------
constexpr size_t kSnvSpec = 4096;   // selected groups that come back with the sweep's counts; more only with a second copy

static inline bool candidate_is_plain_snv(const PiscesHip* h, const HostCandidate& c)
{
    return c.category == PISCES_CAT_SNV && c.from_reads && !(h->cfg.collapse != 0 && (c.open_left || c.open_right));
}
static inline bool split_dirty_at(const PiscesHip* h, int32_t p)
{
    const int64_t rel = (int64_t)p - h->P.dirty_first;
    if (!h->P.dirty_bits || rel < 0 || rel >= h->P.dirty_n) return false;
    return ((h->dirty_host[(size_t)(rel >> 5)] >> (rel & 31)) & 1u) != 0u;
}

// One pass over the SNV store: groups on set bits of `bm` (device, or nullptr) -> `selected`; groups at or below drop_hi are dropped; the
// rest is kept (compacted into the other buffer).  Waits for the device.
static int32_t snv_store_sweep(PiscesHip* h, const uint32_t* d_bm, int32_t bm_first, int32_t bm_n, int32_t drop_hi, std::vector<SnvGroup>& selected)
{
    selected.clear();
    if (h->snv_ub <= 0) return PISCES_OK;
    const int c = h->snv_cur, o = c ^ 1;
    PISCES_HIP_CHECK(h, h->d_snv[o].reserve((size_t)h->snv_ub));
    PISCES_HIP_CHECK(h, h->d_snv_sel.reserve((size_t)h->snv_ub));
    if (!h->h_snv_sel) PISCES_HIP_CHECK(h, host_alloc((void**)&h->h_snv_sel, 16 + kSnvSpec * sizeof(SnvGroup)));
    PISCES_HIP_CHECK(h, hipMemsetAsync(h->d_snv_n.p + 2, 0, 2 * sizeof(unsigned int), h->stream));
    hipLaunchKernelGGL(snv_store_sweep_kernel, dim3((unsigned)((h->snv_ub + 255) / 256)), dim3(256), 0, h->stream, (const SnvGroup*)h->d_snv[c].p,
                       (const unsigned int*)(h->d_snv_n.p + c), d_bm, bm_first, bm_n, drop_hi, h->d_snv_sel.p, (uint32_t)std::min<size_t>(h->d_snv_sel.cap, 0xFFFFFFF0u),
                       h->d_snv[o].p, h->d_snv_n.p + 2);
    PISCES_HIP_CHECK(h, hipGetLastError());
    // the kept count becomes the other buffer's count; the counts and a first stretch of the selected groups come back together
    PISCES_HIP_CHECK(h, hipMemcpyAsync(h->d_snv_n.p + o, h->d_snv_n.p + 3, sizeof(unsigned int), hipMemcpyDeviceToDevice, h->stream));
    unsigned int* counts = (unsigned int*)h->h_snv_sel;
    SnvGroup* first = (SnvGroup*)((uint8_t*)h->h_snv_sel + 16);
    const size_t spec = std::min<size_t>(kSnvSpec, (size_t)h->snv_ub);
    PISCES_HIP_CHECK(h, hipMemcpyAsync(counts, h->d_snv_n.p + 2, 2 * sizeof(unsigned int), hipMemcpyDeviceToHost, h->stream));
    if (d_bm) PISCES_HIP_CHECK(h, hipMemcpyAsync(first, h->d_snv_sel.p, spec * sizeof(SnvGroup), hipMemcpyDeviceToHost, h->stream));
    PISCES_TIMED_WAIT(h, hipStreamSynchronize(h->stream));
    const size_t n_sel = counts[0], n_kept = counts[1];
    if ((int64_t)(n_sel + n_kept) > h->snv_ub) return fail(h, PISCES_E_DEVICE, "flush: the SNV store's counts are inconsistent");
    selected.resize(n_sel);
    if (n_sel) std::memcpy(selected.data(), first, std::min(n_sel, spec) * sizeof(SnvGroup));
    if (n_sel > spec) {
        PISCES_HIP_CHECK(h, hipMemcpyAsync(selected.data() + spec, h->d_snv_sel.p + spec, (n_sel - spec) * sizeof(SnvGroup), hipMemcpyDeviceToHost, h->stream));
        PISCES_TIMED_WAIT(h, hipStreamSynchronize(h->stream));
    }
    h->pcie[2] += (int64_t)(n_sel * sizeof(SnvGroup)) + 16;
    h->split_stats[1] += (int64_t)n_sel;
    h->split_stats[2] += h->snv_ub - (int64_t)n_sel - (int64_t)n_kept;
    h->split_stats[3] += 1;
    h->snv_cur = o;
    h->snv_ub = (int64_t)n_kept;
    if (h->found.in_flight && h->found.split) h->found.split_counted = true;   // (a batch whose records are still to be taken: its groups are in n_kept)
    return PISCES_OK;
}

// The dirty loci of the batch `keys` (and, when alleles of the cleared blocks reach past the last cleared position, of the held blocks up
// to upTo whose candidates AddCollapsableFromOtherBlocks will bring in), as a bit map on host and device; the SNV groups on them join
// their blocks' candidates; DeviceParams is set for the tile kernels of this flush.  split_restore() undoes the latter.
// When the SNVs are the allele counts' (MNV calling off, PiscesHip::snv_walk false) the same bit map marks the few loci where they are
// not: bases of X / = operations (BlockObs::unwalked) and SNV candidates the host handed in; call_spanning makes those loci's candidates.
static void split_restore(PiscesHip* h)
{
    h->P.dirty_bits = nullptr;
    h->P.dirty_first = h->P.dirty_n = 0;
    h->P.refs_only = h->snv_walk ? 1 : 0;
}
static int32_t split_prepare(PiscesHip* h, const std::vector<int32_t>& keys, int32_t up_to_position)
{
    split_restore(h);
    h->split_selected.clear();
    // MNV calling off: the loci with bases of X / = operations that the allele counts hold and no SNV candidate stands for are dirty too
    // (their SNVs are called by call_spanning, from the counts less those bases)
    const bool unwalked_mode = !h->snv_walk;
    if (!h->mnv_split && !unwalked_mode) return PISCES_OK;
    // (and the loci of the SNV candidates the host handed in — pisces_hip_add_candidates: what the reads show joins them there)
    // (a forced SNV nobody gave support stays where it was: the tile kernels report it when it is callable, call_spanning when it is not)
    auto own_snv = [&](const HostCandidate& c) { return c.category == PISCES_CAT_SNV && (cand_support(c) > 0 || h->forced.empty() || !is_forced_allele(h, c)); };
    if (unwalked_mode) {
        bool some = false;
        for (int32_t key : keys) {
            const BlockObs& b = h->blocks[key];
            some = some || !b.unwalked.empty();
            for (auto& c : b.cands) some = some || own_snv(c);
        }
        if (!some) return PISCES_OK;
    } else {
        h->P.refs_only = 0;
    }
    if (keys.empty()) return PISCES_OK;
    const int bs = h->cfg.block_size;
    const int64_t lo = (int64_t)(keys.front() - 1) * bs + 1, hi = (int64_t)keys.back() * bs;
    int64_t hi_bm = hi;
    bool add_collapsable = false;
    if (up_to_position >= 0 && h->cfg.collapse && !unwalked_mode) {   // (the condition of call_spanning's AddCollapsableFromOtherBlocks step)
        int32_t max_endpoint = 0;
        for (int32_t key : keys) max_endpoint = std::max(max_endpoint, h->blocks[key].max_allele_endpoint);
        if ((int64_t)max_endpoint > hi) { add_collapsable = true; hi_bm = std::max<int64_t>(hi, up_to_position); }
    }
    const int64_t n = hi_bm - lo + 1;
    if (n > 0x7FFFFF00ll) return fail(h, PISCES_E_UNSUPPORTED, "flush: the blocks of one batch span more than 2^31 positions");
    std::vector<uint32_t>& bm = h->dirty_host;
    bm.assign((size_t)((n + 31) / 32), 0u);
    bool any = false;
    auto mark = [&](int64_t a, int64_t b) {   // inclusive positions
        a = std::max(a, lo); b = std::min(b, hi_bm);
        if (a > b) return;
        any = true;
        int64_t i = a - lo;
        const int64_t e = b - lo;
        while (i <= e) {
            const int64_t w = i >> 5;
            const int s0 = (int)(i & 31), s1 = (int)std::min<int64_t>(31, e - (w << 5));
            bm[(size_t)w] |= (s1 == 31 ? 0xFFFFFFFFu : ((1u << (s1 + 1)) - 1u)) & ~((1u << s0) - 1u);
            i = (w + 1) << 5;
        }
    };
    const bool track_open = h->cfg.collapse != 0;
    // A span that reaches past the last flushed position stays dirty for the block it reaches into: the candidate goes with this batch, the
    // bases it took (an MNV across a block edge) stay in that block's counts.  (Only blocks that exist: no reads, nothing counted.)
    auto carry = [&](int64_t a, int64_t b) {
        a = std::max(a, hi + 1);
        b = std::min<int64_t>(b, 0x7FFFFFFFll);
        if (a > b) return;
        for (int32_t k = block_key(h, (int32_t)a); k <= block_key(h, (int32_t)b); k++) {
            auto it = h->blocks.find(k);
            if (it != h->blocks.end()) it->second.x_spans.emplace_back((int32_t)a, (int32_t)b);
        }
    };
    auto mark_candidate = [&](const HostCandidate& c, bool of_the_batch) {
        if (c.category != PISCES_CAT_SNV && c.category != PISCES_CAT_MNV) return;   // (insertions and deletions touch no point allele)
        const bool interesting = c.category == PISCES_CAT_MNV || !c.from_reads || (track_open && (c.open_left || c.open_right));
        if (!interesting) return;
        const int64_t end = (int64_t)c.position + (int64_t)std::max<size_t>(c.alt.size(), c.ref.size()) - 1;
        mark(c.position, end);
        if (of_the_batch) carry(c.position, end);
    };
    if (unwalked_mode) {
        for (int32_t key : keys) {
            for (auto& u : h->blocks[key].unwalked) mark(u.position, u.position);
            for (auto& c : h->blocks[key].cands)
                if (own_snv(c)) mark(c.position, c.position);
        }
    } else if (!h->forced.empty()) {
        mark(lo, hi);   // forced alleles: every candidate of the batch is an object on the host, as before
    } else {
        for (int32_t key : keys) {
            const BlockObs& b = h->blocks[key];
            for (auto& c : b.cands) mark_candidate(c, true);
            for (auto& sp : b.x_spans) mark(sp.first, sp.second);   // (a span lies with every block it touches already)
        }
        if (!h->intervals.empty()) {   // outside the intervals a callable allele is counted and not reported (AlleleCaller.cs:236-263): the candidate path's rule
            int64_t at = lo;
            auto it = std::lower_bound(h->intervals.begin(), h->intervals.end(), (int32_t)lo, [](const std::pair<int32_t, int32_t>& iv, int32_t p) { return iv.second < p; });
            for (; it != h->intervals.end() && it->first <= hi; ++it) {
                if (it->first > at) mark(at, (int64_t)it->first - 1);
                at = std::max<int64_t>(at, (int64_t)it->second + 1);
            }
            if (at <= hi) mark(at, hi);
        }
    }
    if (add_collapsable)
        for (auto& kv : h->blocks) {
            const int64_t start = (int64_t)(kv.first - 1) * bs + 1;
            if (start <= hi || start > up_to_position) continue;
            for (auto& c : kv.second.cands) {
                const bool collapsable = (c.category == PISCES_CAT_MNV || c.category == PISCES_CAT_SNV) && !c.open_right &&
                                         c.position + (int32_t)c.alt.size() - 1 <= up_to_position;
                if (collapsable) mark_candidate(c, false);
            }
        }
    // the device's copy of the map, and the groups on its set bits
    if (any) {
        PISCES_HIP_CHECK(h, h->d_dirty.reserve(bm.size()));
        { int32_t rcu = meta_upload(h, h->d_dirty.p, bm.data(), bm.size() * sizeof(uint32_t)); if (rcu) return rcu; }
    }
    std::vector<SnvGroup> selected;
    if (h->snv_ub > 0) {
        int32_t rcs = snv_store_sweep(h, any ? h->d_dirty.p : (const uint32_t*)nullptr, (int32_t)lo, (int32_t)n, (int32_t)std::min<int64_t>(hi, 0x7FFFFFFFll), selected);
        if (rcs) return rcs;
    }
    // The groups of the batch's own blocks do not become block state: they are flushed with this batch, and call_spanning takes them from
    // h->split_selected (position order, then order of arrival).  Only the few of held blocks (the AddCollapsable case) join their blocks.
    h->split_selected.clear();
    if (!selected.empty()) {
        std::sort(selected.begin(), selected.end(), [](const SnvGroup& a, const SnvGroup& b) {
            if (a.position != b.position) return a.position < b.position;
            if (a.batch != b.batch) return a.batch < b.batch;
            return a.first < b.first;
        });
        std::vector<int32_t> touched;
        h->split_selected.reserve(selected.size());
        for (const SnvGroup& g : selected) {
            if (g.position < 1 || (int64_t)g.position > h->ref_len) continue;
            HostCandidate c;
            c.position = g.position;
            c.category = PISCES_CAT_SNV;
            c.ref.assign(1, (char)h->h_ref[(size_t)g.position - 1]);
            c.alt.assign(1, (char)g.alt);
            for (int d = 0; d < 3; d++) { c.support_by_dir[d] = g.sup[d]; c.well_anchored_by_dir[d] = g.anch[d]; }
            c.stamp = ((uint64_t)g.batch << 32) | (uint64_t)g.first;
            c.from_reads = true;
            if ((int64_t)g.position <= hi) { h->split_selected.push_back(std::move(c)); continue; }
            add_candidate(h, c);
            touched.push_back(block_key(h, g.position));
        }
        std::sort(touched.begin(), touched.end());
        touched.erase(std::unique(touched.begin(), touched.end()), touched.end());
        for (int32_t k : touched) reorder_block_candidates(h, &h->blocks[k]);
        h->last_block = nullptr;
    }
    if (any) {
        h->P.dirty_bits = h->d_dirty.p;
        h->P.dirty_first = (int32_t)lo;
        h->P.dirty_n = (int32_t)n;
    }
    return PISCES_OK;
}

// IAlleleCaller.Call for the host-found candidates of `keys` (AlleleCaller.CallForPositions :60-141): anchor-resolved counts of every
// block a candidate touches -> collapser -> call_spanning_kernel -> callable candidates with their records.  With MNV calling on
// the candidates include the SNVs / MNVs of the read walk: MNV candidates are processed first, the ones that are not callable go
// through MnvReallocator on the host, leftovers past the last cleared block return to the state as candidates of the next block,
// reference support taken by gapped MNVs is registered (it reaches the Reference records through call_blocks, which runs after
// this), and every callable allele is processed again.  ref_overrides: Reference alleles that reallocation added support to
// (they replace the tile kernels' Reference record of that position).
// blocks_first: the tile kernels of this flush are enqueued already (their Reference records do not know the reference support that
// gapped MNVs of THIS batch take: the Reference alleles of those positions come from here, as overrides), and h->fold says where their
// folded counts lie.
static int32_t call_spanning(PiscesHip* h, const std::vector<int32_t>& keys, int32_t up_to_position, std::vector<PiscesCalledAllele>& recs,
                             std::vector<HostCandidate>& called, int64_t* n_called, int64_t* n_collapsed,
                             std::vector<PiscesCalledAllele>& ref_overrides, bool blocks_first = false)
{
    recs.clear();
    called.clear();
    ref_overrides.clear();
    *n_collapsed = 0;
    const bool mnv_mode = h->cfg.call_mnvs != 0;
    const bool window = h->cfg.noise_model == PISCES_NOISE_WINDOW;
    std::unique_ptr<HostTimer> prof(new HostTimer(h->prof_on ? &h->prof[1] : nullptr));
    auto phase = [&](int i) { prof.reset(); prof.reset(new HostTimer(h->prof_on ? &h->prof[i] : nullptr)); };
    std::vector<HostCandidate> work;   // a copy: the blocks keep their candidates until DoneProcessing
    {
        size_t total = 0;
        for (int32_t key : keys) total += h->blocks[key].cands.size();
        work.reserve(total);
    }
    for (int32_t key : keys) {
        // RegionState.GetAllCandidates walks _candidateVariantsLookup by position, each position in arrival order (RegionState.cs:388-391)
        const size_t first = work.size();
        for (auto& c : h->blocks[key].cands) {
            // (split form of MNV calling: a fully anchored SNV of the read walk on a locus that is not dirty is the tile kernels' to call)
            if (h->mnv_split && candidate_is_plain_snv(h, c) && !split_dirty_at(h, c.position)) continue;
            work.push_back(c);
        }
        // the SNV groups of the block's dirty loci, taken from the device's store for this flush (split_prepare): they go where their
        // arrival puts them among the block's candidates, and a group equal to a candidate that is there already is merged into it
        // (RegionState.AddCandidate, RegionState.cs:114-137)
        bool extra = false;
        if (h->mnv_split && !h->split_selected.empty()) {
            const int32_t b0 = (key - 1) * h->cfg.block_size + 1, b1 = key * h->cfg.block_size;
            auto lo = std::lower_bound(h->split_selected.begin(), h->split_selected.end(), b0, [](const HostCandidate& c, int32_t p) { return c.position < p; });
            for (; lo != h->split_selected.end() && lo->position <= b1; ++lo) { work.push_back(*lo); extra = true; }
        }
        if (!extra) {
            std::stable_sort(work.begin() + (std::ptrdiff_t)first, work.end(), [](const HostCandidate& x, const HostCandidate& y) { return x.position < y.position; });
        } else {
            std::stable_sort(work.begin() + (std::ptrdiff_t)first, work.end(), [](const HostCandidate& x, const HostCandidate& y) {
                return x.position != y.position ? x.position < y.position : x.stamp < y.stamp;
            });
            const bool track_open = h->cfg.collapse != 0;
            size_t w = first;
            for (size_t i = first; i < work.size(); i++) {
                bool merged = false;
                for (size_t j = w; j-- > first && work[j].position == work[i].position;) {
                    HostCandidate& e = work[j];
                    if (e.category == work[i].category && e.ref == work[i].ref && e.alt == work[i].alt &&
                        (!track_open || (e.open_left == work[i].open_left && e.open_right == work[i].open_right))) {
                        for (int d = 0; d < 3; d++) { e.support_by_dir[d] += work[i].support_by_dir[d]; e.well_anchored_by_dir[d] += work[i].well_anchored_by_dir[d]; }
                        e.from_reads = e.from_reads && work[i].from_reads;
                        merged = true;
                        break;
                    }
                }
                if (merged) continue;
                if (w != i) work[w] = std::move(work[i]);
                w++;
            }
            work.resize(w);
        }
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
                kv.second.cand_next.clear();
                for (auto& c : kept) add_candidate(h, c);
            }
        }
    }
    // MNV calling off: the bases of X / = operations on the batch's loci that the allele counts hold and no SNV candidate stands for
    // (ProcessCigarOps walks M operations only, CandidateVariantFinder.cs:44-71), by (position, read base).  The tile kernels leave the
    // variants of those loci alone (split_prepare made them dirty); their SNV candidates are made here, after the collapser, with the
    // support the read walk's candidates have: the allele counts less these bases.
    std::vector<BlockObs::Unwalked> unw;
    if (!h->snv_walk) {
        for (int32_t key : keys) {
            const auto& u = h->blocks[key].unwalked;
            unw.insert(unw.end(), u.begin(), u.end());
        }
        for (auto& c : work)   // an SNV candidate the host handed in: its locus is dirty; the other bases of the locus are made below, from the counts
            if (c.category == PISCES_CAT_SNV && split_dirty_at(h, c.position) && c.alt.size() == 1 && std::binary_search(keys.begin(), keys.end(), block_key(h, c.position)))
                unw.push_back({c.position, (uint8_t)c.alt[0], {0, 0, 0}});
        if (max_cleared >= 0) {   // (SNV candidates of held blocks that joined the batch — forced ones: what the reads so far show at their positions)
            std::vector<int32_t> seen;
            for (auto& c : work) {
                if (c.category != PISCES_CAT_SNV || c.position <= max_cleared || std::find(seen.begin(), seen.end(), c.position) != seen.end()) continue;
                seen.push_back(c.position);
                auto it = h->blocks.find(block_key(h, c.position));
                if (it == h->blocks.end()) continue;
                for (auto& u : it->second.unwalked)
                    if (u.position == c.position) unw.push_back(u);
            }
        }
        std::sort(unw.begin(), unw.end(), [](const BlockObs::Unwalked& a, const BlockObs::Unwalked& b) { return a.position != b.position ? a.position < b.position : a.alt < b.alt; });
        size_t w = 0;
        for (size_t i = 0; i < unw.size(); i++) {
            if (w > 0 && unw[w - 1].position == unw[i].position && unw[w - 1].alt == unw[i].alt) { for (int d = 0; d < 3; d++) unw[w - 1].sup[d] += unw[i].sup[d]; continue; }
            unw[w++] = unw[i];
        }
        unw.resize(w);
    }
    auto unwalked_of = [&](int32_t position, char alt, int d) -> int32_t {
        auto it = std::lower_bound(unw.begin(), unw.end(), std::make_pair(position, (uint8_t)alt), [](const BlockObs::Unwalked& u, const std::pair<int32_t, uint8_t>& k) {
            return u.position != k.first ? u.position < k.first : u.alt < k.second; });
        return (it != unw.end() && it->position == position && it->alt == (uint8_t)alt) ? it->sup[d] : 0;
    };
    if (work.empty() && unw.empty()) return PISCES_OK;
    // start / end points (CoverageCalculator.Compute :27-41)
    auto endpoints = [](const HostCandidate& c, int32_t& sp, int32_t& ep) {
        if (c.category == PISCES_CAT_DELETION) { sp = c.position + 1; ep = c.position + (int32_t)c.ref.size() - 1; }
        else if (c.category == PISCES_CAT_MNV) { sp = c.position; ep = c.position + (int32_t)c.alt.size() - 1; }
        else if (c.category == PISCES_CAT_INSERTION) { sp = c.position; ep = c.position + 1; }
        else { sp = c.position; ep = c.position; }
    };
    // ---- where the candidates' counts come from (kernels.hip.h CountsView).  Point alleles, MNVs and deletions add up all anchor bins of a
    // cell: when the flush's tile kernel has run over these blocks already it left exactly those sums for every locus (h->fold), and no
    // second walk over the reads is needed for them.  What is left — insertions (their coverage looks at the bins), loci outside that
    // launch (an allele that ends in a held block), or everything when the tile kernels come later — is accumulated into the tensor, for
    // the 64-locus tiles those loci lie in only (the read store; with an observation log: the whole blocks, bucketed as they always were).
    const bool have_forced = !h->forced.empty();
    const bool fold_ok = blocks_first && h->fold.valid;
    auto in_fold = [&](int32_t p) { return fold_ok && p >= h->fold.lo && p <= h->fold.hi; };
    const bool sparse_tiles = h->read_path == 1 && h->log_ub == 0;
    const int tiles_per_block = (bs + kTile - 1) / kTile;
    auto tile_start_of = [&](int32_t p) { const int32_t b0 = (block_key(h, p) - 1) * bs + 1; return b0 + ((p - b0) / kTile) * kTile; };
    std::vector<int32_t> need_pos;   // positions whose counts must be in the tensor
    auto need = [&](int32_t p, bool may_fold) {
        if (p > 0 && !(may_fold && in_fold(p)) && h->blocks.count(block_key(h, p))) need_pos.push_back(p);
    };
    for (auto& c : work) {
        int32_t sp, ep;
        endpoints(c, sp, ep);
        const bool may_fold = c.category != PISCES_CAT_INSERTION;
        if (c.category == PISCES_CAT_MNV) { for (int32_t p = sp; p <= ep; p++) need(p, true); }   // (and what reallocation makes of it: SNVs, Reference alleles)
        else { need(sp, may_fold); need(ep, may_fold); }
    }
    if (have_forced)
        for (int32_t p : h->forced_positions)
            if (std::binary_search(keys.begin(), keys.end(), block_key(h, p))) need(p, true);
    for (auto& u : unw) need(u.position, true);
    phase(2);
    std::vector<PiscesTile> tiles;
    std::vector<int32_t> bkeys, tile_starts;
    if (sparse_tiles) {
        for (int32_t p : need_pos) tile_starts.push_back(tile_start_of(p));
        std::sort(tile_starts.begin(), tile_starts.end());
        tile_starts.erase(std::unique(tile_starts.begin(), tile_starts.end()), tile_starts.end());
        for (int32_t ts : tile_starts) {
            PiscesTile t;
            t.start_position = ts;
            t.n_loci = std::min<int32_t>(kTile, block_key(h, ts) * bs - ts + 1);
            t.tuple_begin = t.tuple_end = 0;
            tiles.push_back(t);
        }
        if (!tiles.empty()) {
            PISCES_HIP_CHECK(h, h->d_span_tiles.reserve(tiles.size()));
            { int32_t rcu = meta_upload(h, h->d_span_tiles.p, tiles.data(), tiles.size() * sizeof(PiscesTile)); if (rcu) return rcu; }
        }
    } else {
        for (int32_t p : need_pos) bkeys.push_back(block_key(h, p));
        std::sort(bkeys.begin(), bkeys.end());
        bkeys.erase(std::unique(bkeys.begin(), bkeys.end()), bkeys.end());
        // counts over the whole block grid of those blocks (not the interval-clipped tiles)
        if (!bkeys.empty()) {
            int32_t rcb = bucket_blocks(h, bkeys, false, tiles);
            if (rcb) return rcb;
        }
    }
    const int32_t n_tiles = (int32_t)tiles.size();
    const PiscesTile* const d_span_tiles = sparse_tiles ? h->d_span_tiles.p : h->d_tiles.p;
    bool counts_missing = false;
    auto locus_index = [&](int32_t p, bool may_fold = true) -> int64_t {
        if (p <= 0) return -1;
        if (may_fold && in_fold(p)) return -((int64_t)(p - h->fold.lo) + 2);   // the folded counts of the tile kernels' launch
        const int32_t k = block_key(h, p);
        if (!h->blocks.count(k)) return -1;   // no block: no counts (RegionStateManager.cs:222-226)
        if (sparse_tiles) {
            const int32_t ts = tile_start_of(p);
            auto it = std::lower_bound(tile_starts.begin(), tile_starts.end(), ts);
            if (it == tile_starts.end() || *it != ts) { counts_missing = true; return -1; }
            return (int64_t)(it - tile_starts.begin()) * kTile + (p - ts);
        }
        auto it = std::lower_bound(bkeys.begin(), bkeys.end(), k);
        if (it == bkeys.end() || *it != k) { counts_missing = true; return -1; }
        const int64_t bi = it - bkeys.begin();
        const int32_t off = p - ((k - 1) * bs + 1);
        return (bi * tiles_per_block + off / kTile) * kTile + off % kTile;
    };
    if (n_tiles > 0) {
        PISCES_HIP_CHECK(h, accumulate_tiles(h, h->stream, h->d_tuples.p, d_span_tiles, n_tiles, window, true));
    } else {
        PISCES_HIP_CHECK(h, h->d_counts.reserve(PISCES_COUNTS_PER_LOCUS));
        if (window) PISCES_HIP_CHECK(h, h->d_sumq.reserve(PISCES_COUNTS_PER_LOCUS));
    }
    const int32_t* const d_folded = fold_ok ? h->d_folded.p : (const int32_t*)nullptr;
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
        const bool may_fold = c.category != PISCES_CAT_INSERTION;
        d.start_idx = locus_index(sp, may_fold);
        d.end_idx = locus_index(ep, may_fold);
        d.gapped = (c.category == PISCES_CAT_SNV || c.category == PISCES_CAT_REFERENCE) ? gapped_at(c.position) : 0;
    };
    // The collapser's frequencies and the reallocator's Reference candidates read anchor-resolved counts on the host: the ROWS of the few
    // loci they can look at (gather_count_rows_kernel), never the tensor — every start / end point when some candidate is open-ended (the
    // collapser's candidates: CandidateAllele.Frequency of the open-ended one and of what it may join), the positions an MNV candidate
    // spans (a failed one's Reference candidates, MnvReallocator.cs:12-98), a forced SNV's position.
    const int32_t* host_counts_p = nullptr;
    std::unordered_map<int64_t, int32_t>& row_of = h->row_of_locus;
    row_of.clear();
    bool rows_missing = false;
    auto row_index = [&](int64_t li) -> int64_t {
        if (li == -1) return -1;
        auto it = row_of.find(li);
        if (it == row_of.end()) { rows_missing = true; return -1; }
        return it->second;
    };
    {
        bool any_open = false;
        if (h->cfg.collapse)
            for (auto& c : work) any_open = any_open || c.open_left || c.open_right;
        std::vector<long long> need;
        auto want = [&](int32_t p, bool may_fold) { const int64_t li = locus_index(p, may_fold); if (li != -1) need.push_back(li); };
        for (auto& c : work) {
            int32_t sp, ep;
            endpoints(c, sp, ep);
            const bool may_fold = c.category != PISCES_CAT_INSERTION;
            if (mnv_mode && c.category == PISCES_CAT_MNV)
                for (int32_t p = sp; p <= ep; p++) want(p, true);
            else if (any_open || ((have_forced || !h->snv_walk) && c.category == PISCES_CAT_SNV)) { want(sp, may_fold); want(ep, may_fold); }
        }
        for (auto& u : unw) want(u.position, true);
        std::sort(need.begin(), need.end());
        need.erase(std::unique(need.begin(), need.end()), need.end());
        const size_t n_rows = need.size(), n_counts = std::max<size_t>(n_rows, 1) * PISCES_COUNTS_PER_LOCUS;
        if (n_counts > h->h_counts_cap) {
            if (h->h_counts) host_free(h->h_counts);
            h->h_counts = nullptr;
            h->h_counts_cap = 0;
            PISCES_HIP_CHECK(h, host_alloc((void**)&h->h_counts, (n_counts + n_counts / 2) * sizeof(int32_t)));
            h->h_counts_cap = n_counts + n_counts / 2;
        }
        if (n_rows > 0) {
            for (size_t k = 0; k < n_rows; k++) row_of.emplace(need[k], (int32_t)k);
            PISCES_HIP_CHECK(h, h->d_row_idx.reserve(n_rows));
            PISCES_HIP_CHECK(h, h->d_rows.reserve(n_rows * PISCES_COUNTS_PER_LOCUS));
            { int32_t rcu = meta_upload(h, h->d_row_idx.p, need.data(), n_rows * sizeof(long long)); if (rcu) return rcu; }
            hipLaunchKernelGGL(gather_count_rows_kernel, dim3((unsigned)((n_rows * PISCES_COUNTS_PER_LOCUS + 255) / 256)), dim3(256), 0, h->stream,
                               (const int32_t*)h->d_counts.p, d_folded, (const long long*)h->d_row_idx.p, (int32_t)n_rows, h->d_rows.p);
            PISCES_HIP_CHECK(h, hipGetLastError());
            h->pcie[3] += (int64_t)(n_rows * PISCES_COUNTS_PER_LOCUS * sizeof(int32_t));
            PISCES_HIP_CHECK(h, hipMemcpyAsync(h->h_counts, h->d_rows.p, n_rows * PISCES_COUNTS_PER_LOCUS * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
            PISCES_TIMED_WAIT(h, hipStreamSynchronize(h->stream));
        }
        host_counts_p = h->h_counts;
    }
    struct { const int32_t* p; const int32_t* data() const { return p; } } host_counts = {host_counts_p};
    if (!h->snv_walk) {
        // MNV calling off: the reads' SNV candidates are the allele counts and never reach the host, so an SNV candidate that IS an object
        // here — a forced allele (added without support), one the host handed in (with its own) — takes the support the merged candidate of
        // the reference has: the reads that show the base at or above the quality threshold join it
        for (auto& c : work) {
            if (c.category != PISCES_CAT_SNV || c.alt.size() != 1 || c.counted_by_dir[0] + c.counted_by_dir[1] + c.counted_by_dir[2] != 0) continue;
            const int64_t li = row_index(locus_index(c.position));
            const int at = atype(c.alt[0]);
            if (li < 0 || at >= 4) continue;
            for (int d = 0; d < 3; d++) {
                const int32_t* row = host_counts.data() + li * PISCES_COUNTS_PER_LOCUS + (at * 3 + d) * PISCES_NUM_ANCHORS;
                const int32_t own = c.support_by_dir[d];
                for (int an = 0; an < PISCES_NUM_ANCHORS; an++) c.support_by_dir[d] += row[an];
                c.support_by_dir[d] = std::max(own, c.support_by_dir[d] - unwalked_of(c.position, c.alt[0], d));   // (less what no candidate stands for)
                c.counted_by_dir[d] = c.support_by_dir[d] - own;
            }
        }
    }
    phase(3);
    if (h->cfg.collapse) {
        const int32_t stitched = h->cfg.expect_stitched_reads;
        *n_collapsed = collapse_candidates(work, h->cfg.collapse_freq_threshold, h->cfg.collapse_freq_ratio_threshold, [&](const HostCandidate& c) {
            DevCandidate d;
            to_dev(c, d);
            d.start_idx = row_index(d.start_idx);   // (rows of the gathered loci, not of the tensor)
            d.end_idx = row_index(d.end_idx);
            const int total = candidate_total_coverage(d, host_counts.data(), stitched);
            const int support = c.support_by_dir[0] + c.support_by_dir[1] + c.support_by_dir[2];
            if (total == 0) return 0.0f;                       // CalledAllele.Frequency (CalledAllele.cs:49-52)
            const float f = (float)support / (float)total;
            return f < 1.0f ? f : 1.0f;
        }, &h->known_variants, h->exclude_mnvs_from_collapsing);
        // candidates past the last cleared position that could not be collapsed return to the state (VariantCollapser.cs:67-75): only the
        // ones AddCollapsableFromOtherBlocks brought in can lie there
        if (max_cleared >= 0) {
            size_t w = 0;
            for (size_t i = 0; i < work.size(); i++) {
                if (work[i].position > max_cleared && work[i].category != PISCES_CAT_REFERENCE) {
                    for (int d = 0; d < 3; d++) { work[i].support_by_dir[d] -= work[i].counted_by_dir[d]; work[i].counted_by_dir[d] = 0; }   // (the allele counts are asked again when its block is flushed)
                    work[i].stamp = next_host_stamp(h);   // (AddCandidates appends it to its position's list again)
                    add_candidate(h, work[i]);
                    continue;
                }
                if (w != i) work[w] = std::move(work[i]);
                w++;
            }
            work.resize(w);
            if (work.empty() && unw.empty()) return PISCES_OK;
        }
    }
    if (!unw.empty()) {
        // the SNV candidates of the loci with unwalked bases: what the reads show there (the counts) less those bases
        static const char kAcgt[4] = {'A', 'C', 'G', 'T'};
        const size_t n_before = work.size();
        std::unordered_set<uint64_t> snv_there;   // (position, base) of the SNV candidates that are objects already: forced alleles, the host's (support taken above)
        for (auto& c : work)
            if (c.category == PISCES_CAT_SNV && c.alt.size() == 1) snv_there.insert(((uint64_t)(uint32_t)c.position << 8) | (uint8_t)c.alt[0]);
        for (size_t i = 0; i < unw.size();) {
            const int32_t p = unw[i].position;
            while (i < unw.size() && unw[i].position == p) i++;
            if (p < 1 || (int64_t)p > h->ref_len || !std::binary_search(keys.begin(), keys.end(), block_key(h, p))) continue;
            const char rb = (char)h->h_ref[(size_t)p - 1];
            const int64_t li = row_index(locus_index(p));
            if (atype(rb) >= 4 || li < 0) continue;
            for (char ab : kAcgt) {
                if (ab == rb) continue;
                if (snv_there.count(((uint64_t)(uint32_t)p << 8) | (uint8_t)ab)) continue;
                HostCandidate c;
                c.position = p;
                c.category = PISCES_CAT_SNV;
                c.ref.assign(1, rb);
                c.alt.assign(1, ab);
                int32_t total = 0;
                for (int d = 0; d < 3; d++) {
                    const int32_t* row = host_counts.data() + li * PISCES_COUNTS_PER_LOCUS + (atype(ab) * 3 + d) * PISCES_NUM_ANCHORS;
                    int32_t n = 0;
                    for (int an = 0; an < PISCES_NUM_ANCHORS; an++) n += row[an];
                    n = std::max(0, n - unwalked_of(p, ab, d));
                    c.support_by_dir[d] = c.well_anchored_by_dir[d] = n;
                    total += n;
                }
                if (total <= 0) continue;   // no read walk made a candidate of this allele
                c.stamp = next_host_stamp(h);
                c.from_reads = true;
                work.push_back(std::move(c));
            }
        }
        if (work.size() != n_before)
            std::stable_sort(work.begin(), work.end(), [](const HostCandidate& x, const HostCandidate& y) { return x.position < y.position; });
        if (work.empty()) return PISCES_OK;
    }
    // one device pass over a list of candidates: records + IsCallable
    std::vector<PiscesCalledAllele> raw;
    std::vector<uint8_t> callable;
    bool second_pass = false;   // MNV mode: the pass over every callable allele, after the MNV-only pass
    // wanted: what the caller reads of the pass (call_spanning_kernel's kSpanning*): the records of alleles that are not callable only
    // matter for forced alleles, and the MNV pass reads IsCallable alone
    auto device_pass = [&](const std::vector<const HostCandidate*>& list, int32_t wanted) -> int32_t {
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
        // (through the pinned arena and into a pinned buffer: a transfer from or to pageable memory is staged by the runtime, and the host
        // waits for it)
        { int32_t rcu = meta_upload(h, h->d_cands.p, dc.data(), dc.size() * sizeof(DevCandidate)); if (rcu) return rcu; }
        { int32_t rcu = meta_upload(h, h->d_alleles.p, pool.data(), pool.size()); if (rcu) return rcu; }
        hipLaunchKernelGGL(call_spanning_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, h->stream, h->d_cands.p, n, h->d_counts.p,
                           h->d_alleles.p, h->d_ref.p, h->ref_len, h->cfg.expect_stitched_reads, h->d_cand_records.p, h->d_cand_callable.p, h->P,
                           window ? h->d_sumq.p : (const double*)nullptr, d_folded, wanted);
        PISCES_HIP_CHECK(h, hipGetLastError());
        const size_t rec_bytes = wanted == kSpanningFlagsOnly ? 0 : raw.size() * sizeof(PiscesCalledAllele), need = rec_bytes + callable.size();
        h->pcie[1] += (int64_t)need;
        if (need > h->h_cand_dl_cap) {
            if (h->h_cand_dl) host_free(h->h_cand_dl);
            h->h_cand_dl = nullptr;
            h->h_cand_dl_cap = 0;
            PISCES_HIP_CHECK(h, host_alloc((void**)&h->h_cand_dl, need + need / 2));
            h->h_cand_dl_cap = need + need / 2;
        }
        if (rec_bytes) PISCES_HIP_CHECK(h, hipMemcpyAsync(h->h_cand_dl, h->d_cand_records.p, rec_bytes, hipMemcpyDeviceToHost, h->stream));
        PISCES_HIP_CHECK(h, hipMemcpyAsync(h->h_cand_dl + rec_bytes, h->d_cand_callable.p, callable.size(), hipMemcpyDeviceToHost, h->stream));
        PISCES_TIMED_WAIT(h, hipStreamSynchronize(h->stream));
        std::memcpy(raw.data(), h->h_cand_dl, rec_bytes);
        std::memcpy(callable.data(), h->h_cand_dl + rec_bytes, callable.size());
        return PISCES_OK;
    };
    auto owned = [&](int32_t position) { return position >= h->own_lo && position <= h->own_hi; };   // (interval sharding: pisces_hip_set_owned_range)
    auto inside_intervals = [&](int32_t position) {   // ShouldReport (AlleleCaller.cs:260-263)
        if (!owned(position)) return false;
        if (h->intervals.empty()) return true;
        auto it = std::lower_bound(h->intervals.begin(), h->intervals.end(), position,
                                   [](const std::pair<int32_t, int32_t>& iv, int32_t p) { return iv.second < p; });
        return it != h->intervals.end() && it->first <= position;
    };

    phase(4);
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
        int32_t rc1 = device_pass(mnvs, kSpanningFlagsOnly);
        if (rc1) return rc1;
        phase(5);
        std::vector<CandPtr> failed;
        {
            size_t mi = 0;
            for (auto& c : work) {
                if (c.category == PISCES_CAT_MNV) {
                    if (callable[mi]) { callable_alleles.push_back(&c); if (owned(c.position)) (*n_called)++; }   // IsCallable counts every pass (_totalNumCalled)
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
                const int64_t li = row_index(locus_index(p));
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
            { HostTimer prof_r(h->prof_on ? &h->prof[11] : nullptr); mnv_reallocate_failed(arena, failed, callable_alleles, true, last_cleared, outside); }
            for (CandPtr o : outside)   // source.AddCandidates(leftovers.Select(AlleleHelper.Map)) :92-93
                if (o->category != PISCES_CAT_REFERENCE && o->position > 0) {
                    HostCandidate c = *o;
                    c.well_anchored_by_dir[0] = c.well_anchored_by_dir[1] = c.well_anchored_by_dir[2] = 0;
                    c.open_left = c.open_right = false;
                    c.stamp = next_host_stamp(h);
                    c.from_reads = false;   // (support that reallocation moved: no longer what the allele counts say)
                    add_candidate(h, c);
                }
            // Reference candidates keep only what reallocation added: the kernel supplies their own counts
            for (size_t i = 0; i < ref_originals.size(); i++)
                for (int d = 0; d < 3; d++) ref_originals[i]->support_by_dir[d] -= ref_before[i][(size_t)d];
        }
        // GetRefSupportFromGappedMnvs :180-203 -> IAlleleSource.AddGappedMnvRefCount
        std::vector<int32_t> gapped_now;
        for (CandPtr a : callable_alleles) {
            if (a->category != PISCES_CAT_MNV) continue;
            const int support = cand_support(*a);
            for (size_t k = 0; k < a->ref.size() && k < a->alt.size(); k++)
                if (a->ref[k] == a->alt[k]) { h->gapped_mnv_ref[a->position + (int32_t)k] += support; gapped_now.push_back(a->position + (int32_t)k); }
        }
        // The tile kernels of this flush ran before these counts existed: the Reference allele of such a position (its support less what the
        // gapped MNVs take, CoverageCalculator.cs:82-97) comes from the candidate kernel and replaces theirs.  (SNVs of the position are
        // candidates of this pass anyway: an MNV spans it.)
        std::vector<CandPtr> gapped_refs;
        if (blocks_first) {
            std::sort(gapped_now.begin(), gapped_now.end());
            gapped_now.erase(std::unique(gapped_now.begin(), gapped_now.end()), gapped_now.end());
            static const int32_t kNoSupport[3] = {0, 0, 0};
            for (int32_t p : gapped_now) {
                if (touched_refs.count(p)) { if (cand_support(*touched_refs[p]) == 0) gapped_refs.push_back(touched_refs[p]); continue; }
                if (!h->cfg.include_reference_calls || p < 1 || p > h->ref_len || !std::binary_search(keys.begin(), keys.end(), block_key(h, p))) continue;
                touched_refs[p] = arena.make(p, std::string(1, (char)h->h_ref[(size_t)p - 1]), std::string(1, (char)h->h_ref[(size_t)p - 1]), kNoSupport);
                gapped_refs.push_back(touched_refs[p]);
            }
        }
        // a failed MNV that is a forced allele is reported all the same (AlleleCaller.cs:98-107)
        if (have_forced)
            for (CandPtr f : failed)
                if (is_forced_allele(h, *f)) callable_alleles.push_back(f);
        for (CandPtr a : callable_alleles) {
            if (a->category == PISCES_CAT_REFERENCE && cand_support(*a) == 0) continue;   // untouched: the tile kernels' record stands
            final_list.push_back(a);
        }
        for (CandPtr a : gapped_refs) final_list.push_back(a);
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

    if (rows_missing || counts_missing) return fail(h, PISCES_E_INTERNAL, "flush: a candidate's counts were not among those made for the batch");
    phase(6);
    second_pass = mnv_mode;
    int32_t rc2 = device_pass(final_list, have_forced ? kSpanningEveryRecord : kSpanningCallableRecords);
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
        if (callable[i] && owned(final_list[i]->position)) (*n_called) += forced ? 2 : 1;
        if (forced && !h->snv_walk && final_list[i]->category == PISCES_CAT_SNV && reportable && !split_dirty_at(h, final_list[i]->position)) {   // MNV calling off: the tile kernels report it,
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
    if (rows_missing || counts_missing) return fail(h, PISCES_E_INTERNAL, "flush: a candidate's counts were not among those made for the batch");
    return PISCES_OK;
}

int32_t pisces_hip_flush_ex(PiscesHip* h, int32_t up_to_position, PiscesCalledAllele* out, int64_t capacity, int64_t* n_out,
                            int32_t* cand_index_out, PiscesCandidate* cand_out, int64_t cand_capacity, int64_t* n_cand,
                            uint8_t* alleles_out, int64_t allele_capacity, int64_t* allele_bytes)
{
    return abi_guard<int32_t>(h, [&]() -> int32_t {
    if (!h) return PISCES_E_INVALID_ARG;
    if (!n_out || capacity < 0 || (capacity > 0 && !out)) return fail(h, PISCES_E_INVALID_ARG, "flush: null output");
    HostTimer timer(h->in_flush_begin ? nullptr : &h->host_time[1]);   // (flush_begin times the synchronous flush it may run itself)
    PISCES_HIP_CHECK(h, hipSetDevice(h->device));
    *n_out = 0;
    if (n_cand) *n_cand = 0;
    if (allele_bytes) *allele_bytes = 0;
    if (h->async.state != 0) return fail(h, PISCES_E_STATE, "flush: pisces_hip_flush_begin is waiting for its pisces_hip_flush_end");
    // The candidates of the last batch are waited for unless every read of it starts MORE THAN ONE position behind upTo (a read whose first
    // operation is I or D puts its candidate at position - 1: finder_walk.h, CandidateVariantFinder.cs:52-76): then none of them lies in a
    // block this flush clears, nor among the candidates it may take from held blocks (SNVs / MNVs that end at or before upTo) — and a host that adds
    // the next stretch of reads before it calls up to their first position (SmallVariantCaller's LastClearedPosition) has the device
    // discover that stretch's candidates under this flush's host work.
    { int32_t rcd = finish_candidate_discovery(h); if (rcd) return rcd; }   // (the records of the last batch's walk go on their way before this flush's kernels)
    if (!(up_to_position >= 0 && h->found.in_flight && h->found.min_position - 1 > up_to_position && !h->pending_valid)) { int32_t rcf = consume_found(h); if (rcf) return rcf; }
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
        // MNV calling on, split form: the dirty loci of the batch, the SNV groups on them, the tile kernels' parameters for this flush
        struct SplitGuard { PiscesHip* h; ~SplitGuard() { split_restore(h); } } split_guard{h};
        { HostTimer prof_split(h->prof_on ? &h->prof[10] : nullptr); int32_t rcs = split_prepare(h, keys, final_flush ? -1 : up_to_position); if (rcs) return rcs; }
        // host-side candidates first: with MNV calling on they register the reference support that gapped MNVs take, which the
        // Reference records of call_blocks must see (AlleleCaller.cs:95, CoverageCalculator.cs:82-97)
        int64_t collapsed = 0;
        std::vector<PiscesCalledAllele> ref_overrides;
        // The tile kernels first when they are the fused kernel over a run of whole blocks: their launch leaves the folded counts of every
        // locus for the candidate kernel (no second walk over the reads for point alleles, MNVs and deletions), and it runs while the host
        // collects the candidates.  Otherwise (counts in HBM, NoiseModel.Window, an interval set, an observation log) the candidates go first,
        // as they always did: the Reference records then see what gapped MNVs take.
        CallBlocksInFlight blocks_st;
        const bool blocks_first = fused_regular_applies(h, keys);
        int32_t rc = PISCES_OK;
        if (blocks_first) {
            // (the folded counts only when a candidate can ask for them: 72 B a locus that an SNV-only flush need not write)
            bool want_folded = h->mnv_split || !h->forced.empty();
            if (!want_folded)
                for (auto& kv : h->blocks)
                    if (!kv.second.cands.empty() || !kv.second.unwalked.empty()) { want_folded = true; break; }
            rc = call_blocks_enqueue(h, keys, true, -1, &blocks_st, want_folded);
            if (rc) return rc;
        }
        rc = call_spanning(h, keys, final_flush ? -1 : up_to_position, span_recs, span_cands, &called, &collapsed, ref_overrides, blocks_first && blocks_st.active);
        h->fold.valid = false;
        h->pending_collapsed = collapsed;
        if (rc) return rc;
        h->pending_dropped = false;
        std::unique_ptr<HostTimer> prof(new HostTimer(h->prof_on ? &h->prof[7] : nullptr));
        // per-locus genotypers: on the device, over the tile kernels' slots, when nothing joins their rows; else the host pass over the merged rows
        const bool device_genotyper = germline(h) && h->device_genotyper && span_recs.empty() && h->forced.empty() && ref_overrides.empty();
        const bool diploid = germline(h) && !device_genotyper;
        // nothing to merge into the tile kernels' records: they go from the download buffer straight to the caller
        const bool plain = span_recs.empty() && !diploid && h->forced.empty() && ref_overrides.empty();
        // the tile kernels' rows are read where the last kernel left them (pinned memory) unless a per-locus genotyper or forced alleles rework them
        const bool fast_merge = !plain && !diploid && h->forced.empty();
        if (blocks_first) {
            // (enqueued above; the candidate passes have waited for the stream more than once since: this wait is short)
            if (blocks_st.active) {
                PISCES_TIMED_WAIT(h, hipStreamSynchronize(h->stream));
                h->h_meta_used = 0;
#ifdef PISCES_STORE_TIMING
                dump_tile_stamps(h);
#endif
                int32_t total = 0;
                rc = call_blocks_finish(h, blocks_st, &total, &called, &h->pending_kept);
                if (rc) return rc;
                if (blocks_st.drop_now) h->pending_dropped = true;
                if (plain || fast_merge) { h->pending_view = blocks_st.hrec; h->pending_view_n = (size_t)total; }
                else point_recs.assign(blocks_st.hrec, blocks_st.hrec + total);
            }
        } else {
        rc = call_blocks(h, keys, point_recs, &called, true, &h->pending_dropped, &h->pending_kept, plain || fast_merge, device_genotyper);
        if (rc) return rc;
        }
        prof.reset();
        prof.reset(new HostTimer(h->prof_on ? &h->prof[8] : nullptr));
        if (fast_merge) {
            // The candidate kernel's rows (a handful per block) go into the tile kernels' rows (position, ref, alt order, one per locus and
            // more): where each belongs is found by bisection, the rows between two such places are copied in one piece.  Three kinds of
            // places: a candidate row goes in FRONT of a row; a Reference row goes because a variant of the candidate kernel is reported on its
            // position (AlleleCaller.cs:146-147); a Reference row is replaced by the one MNV reallocation added support to.
            static const char kBaseF[6] = {'A', 'G', 'C', 'T', 'N', 'D'};
            const PiscesCalledAllele* const P = h->pending_view;
            const size_t np = h->pending_view_n;
            auto first_at = [&](int32_t position) {   // first row at or behind the position
                size_t a = 0, b = np;
                while (a < b) { const size_t m = (a + b) >> 1; if (P[m].position < position) a = m + 1; else b = m; }
                return a;
            };
            struct SRow { const PiscesCalledAllele* r; int32_t ci; const std::string* ref; const std::string* alt; };
            std::vector<SRow> rows;
            rows.reserve(span_recs.size());
            for (size_t i = 0; i < span_recs.size(); i++) rows.push_back({&span_recs[i], (int32_t)i, &span_cands[i].ref, &span_cands[i].alt});
            std::stable_sort(rows.begin(), rows.end(), [](const SRow& a, const SRow& b) {
                if (a.r->position != b.r->position) return a.r->position < b.r->position;
                if (*a.ref != *b.ref) return *a.ref < *b.ref;
                return *a.alt < *b.alt;
            });
            auto point_first = [&](const PiscesCalledAllele& p, const SRow& sr) {   // true: p goes before sr (or they are equal): one-base alleles against strings
                const char pr = kBaseF[PISCES_INFO_REF(p.info)], pa = kBaseF[PISCES_INFO_ALT(p.info)];
                const int cr = sr.ref->empty() ? 1 : (pr != (*sr.ref)[0] ? (pr < (*sr.ref)[0] ? -1 : 1) : (sr.ref->size() > 1 ? -1 : 0));
                if (cr != 0) return cr < 0;
                const int ca = sr.alt->empty() ? 1 : (pa != (*sr.alt)[0] ? (pa < (*sr.alt)[0] ? -1 : 1) : (sr.alt->size() > 1 ? -1 : 0));
                return ca <= 0;
            };
            struct Ev { size_t idx; int kind; const PiscesCalledAllele* row; int32_t ci; };   // kind 0: insert in front of idx, 1: drop idx, 2: replace idx
            std::vector<Ev> evs;
            evs.reserve(rows.size() * 2 + ref_overrides.size());
            int32_t last_dropped = -1;
            for (auto& sr : rows) {
                size_t i = first_at(sr.r->position);
                while (i < np && P[i].position == sr.r->position && point_first(P[i], sr)) i++;
                evs.push_back({i, 0, sr.r, sr.ci});
                const bool forced_row = ((sr.r->filter_bits >> PISCES_FILTER_FORCED_REPORT) & 1u) != 0;
                if (!forced_row && sr.r->position != last_dropped) {
                    last_dropped = sr.r->position;
                    for (size_t k = first_at(sr.r->position); k < np && P[k].position == sr.r->position; k++)
                        if (PISCES_INFO_CATEGORY(P[k].info) == PISCES_CAT_REFERENCE) evs.push_back({k, 1, nullptr, -1});
                }
            }
            for (auto& ov : ref_overrides)
                for (size_t k = first_at(ov.position); k < np && P[k].position == ov.position; k++)
                    if (PISCES_INFO_CATEGORY(P[k].info) == PISCES_CAT_REFERENCE) { evs.push_back({k, 2, &ov, -1}); break; }
            std::stable_sort(evs.begin(), evs.end(), [](const Ev& a, const Ev& b) { return a.idx != b.idx ? a.idx < b.idx : a.kind < b.kind; });
            h->pending.clear();
            h->pending_cand_index.clear();
            // In place when the rows lie in the download buffer and it has room for the ones that join them: a row that comes in pushes the
            // rows behind it one place along only until the Reference row of its position goes (the usual pair of events), so nearly all
            // of the buffer stays where the kernel wrote it.  (A copy of every row into h->pending was 5 of the 7 ms a flush of 230 000
            // loci of BASELINE config 4 spent on the host.)
            const size_t room = h->h_dl_cap / sizeof(PiscesCalledAllele);
            if (h->merge_in_place && h->h_dl && P == (const PiscesCalledAllele*)h->h_dl + 1 && room > 0 && np + rows.size() + 1 <= room) {
                PiscesCalledAllele* const M = (PiscesCalledAllele*)h->h_dl + 1;
                std::deque<PiscesCalledAllele> ahead;   // rows [cur, rp) of the input: read out of the way of the write cursor
                size_t cur = 0, rp = 0, w = 0;          // next input row; first input row still in its place (>= cur); next output place (<= rp)
                std::vector<std::pair<size_t, int32_t>> placed;
                placed.reserve(rows.size());
                auto emit = [&](const PiscesCalledAllele& row) {
                    if (w == rp && rp < np) { ahead.push_back(M[rp]); rp++; }
                    M[w++] = row;
                };
                auto skip_one = [&]() { if (!ahead.empty()) ahead.pop_front(); else rp++; cur++; };
                auto copy_to = [&](size_t end) {
                    while (cur < end && !ahead.empty()) {
                        const PiscesCalledAllele row = ahead.front();
                        ahead.pop_front();
                        cur++;
                        emit(row);
                    }
                    if (cur < end) {   // nothing read ahead: rp == cur, w <= cur
                        const size_t n = end - cur;
                        if (w != cur) std::memmove(M + w, M + cur, n * sizeof(PiscesCalledAllele));
                        w += n;
                        cur = rp = end;
                    }
                };
                for (size_t e = 0; e < evs.size(); e++) {
                    const Ev& ev = evs[e];
                    copy_to(ev.idx);
                    if (ev.kind == 0) { placed.push_back({w, ev.ci}); emit(*ev.row); continue; }
                    if (cur != ev.idx) continue;                       // (the row is gone already: dropped and replaced at once)
                    if (ev.kind == 2) {
                        bool dropped = false;
                        for (size_t q = e; q-- > 0 && evs[q].idx == ev.idx;) dropped = dropped || evs[q].kind == 1;
                        if (!dropped) emit(*ev.row);
                    }
                    skip_one();
                }
                copy_to(np);
                h->pending_view = M;
                h->pending_view_n = w;
                h->pending_cand_index.assign(w, -1);
                for (auto& pl : placed) h->pending_cand_index[pl.first] = pl.second;
                h->pending_cands = span_cands;
            } else {
            h->pending.reserve(np + rows.size());
            h->pending_cand_index.reserve(np + rows.size());
            size_t cur = 0;
            auto copy_to = [&](size_t end) {
                if (end > cur) {
                    h->pending.insert(h->pending.end(), P + cur, P + end);
                    h->pending_cand_index.insert(h->pending_cand_index.end(), end - cur, -1);
                    cur = end;
                }
            };
            for (size_t e = 0; e < evs.size(); e++) {
                const Ev& ev = evs[e];
                copy_to(ev.idx);
                if (ev.kind == 0) { h->pending.push_back(*ev.row); h->pending_cand_index.push_back(ev.ci); continue; }
                if (cur != ev.idx) continue;                       // (the row is gone already: dropped and replaced at once)
                if (ev.kind == 2) {
                    // a Reference row that a variant of its position removes is not brought back by its replacement
                    bool dropped = false;
                    for (size_t q = e; q-- > 0 && evs[q].idx == ev.idx;) dropped = dropped || evs[q].kind == 1;
                    if (!dropped) { h->pending.push_back(*ev.row); h->pending_cand_index.push_back(-1); }
                }
                cur = ev.idx + 1;
            }
            copy_to(np);
            h->pending_view = nullptr;
            h->pending_view_n = 0;
            h->pending_cands = span_cands;
            }
        } else {
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
        if (plain) {
            h->pending_cand_index.assign(h->pending_view_n, -1);
        } else if (span_recs.empty() && !diploid && h->forced.empty()) {
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
            auto row_before = [](const Row& a, const Row& b) {
                if (a.r->position != b.r->position) return a.r->position < b.r->position;
                if (a.ref != b.ref) return a.ref < b.ref;
                return a.alt < b.alt;
            };
            if (!diploid) {
                // The tile kernels' rows arrive in (position, ref, alt) order; only the candidate kernel's rows (a handful per block) need
                // sorting, and the two runs are merged — rows of the tile kernels first among equals, as a stable sort of the rows in
                // that order leaves them.  (Every row used to go through the sort with its two allele strings: 118 of the 151 ms of the
                // flushes of a 900 000-locus contig of BASELINE config 4.)
                for (size_t i = 0; i < span_recs.size(); i++) rows.push_back({&span_recs[i], (int32_t)i, span_cands[i].ref, span_cands[i].alt});
                std::stable_sort(rows.begin(), rows.end(), row_before);
                h->pending.reserve(point_recs.size() + rows.size());
                h->pending_cand_index.reserve(point_recs.size() + rows.size());
                // a tile kernel's row against a candidate row: its alleles are one base each
                auto point_first = [&](const PiscesCalledAllele& p, const Row& sr) {   // true: p goes before sr (or they are equal)
                    if (p.position != sr.r->position) return p.position < sr.r->position;
                    const char pr = kBase[PISCES_INFO_REF(p.info)], pa = kBase[PISCES_INFO_ALT(p.info)];
                    const int cr = sr.ref.empty() ? 1 : (pr != sr.ref[0] ? (pr < sr.ref[0] ? -1 : 1) : (sr.ref.size() > 1 ? -1 : 0));   // "X" against sr.ref
                    if (cr != 0) return cr < 0;
                    const int ca = sr.alt.empty() ? 1 : (pa != sr.alt[0] ? (pa < sr.alt[0] ? -1 : 1) : (sr.alt.size() > 1 ? -1 : 0));
                    return ca <= 0;
                };
                size_t si = 0;
                bool point_sorted = true;
                for (size_t i = 1; i < point_recs.size() && point_sorted; i++) point_sorted = point_recs[i - 1].position <= point_recs[i].position;
                if (point_sorted) {
                    for (auto& r : point_recs) {
                        const bool is_ref = PISCES_INFO_CATEGORY(r.info) == PISCES_CAT_REFERENCE;
                        if (is_ref && std::binary_search(variant_pos.begin(), variant_pos.end(), r.position)) continue;
                        while (si < rows.size() && !point_first(r, rows[si])) { h->pending.push_back(*rows[si].r); h->pending_cand_index.push_back(rows[si].ci); si++; }
                        h->pending.push_back(r);
                        h->pending_cand_index.push_back(-1);
                    }
                    for (; si < rows.size(); si++) { h->pending.push_back(*rows[si].r); h->pending_cand_index.push_back(rows[si].ci); }
                } else {
                    // (rows of the tile kernels that are not in position order — not something a flush produces: the general sort)
                    std::vector<Row> all;
                    for (auto& r : point_recs) {
                        const bool is_ref = PISCES_INFO_CATEGORY(r.info) == PISCES_CAT_REFERENCE;
                        if (is_ref && std::binary_search(variant_pos.begin(), variant_pos.end(), r.position)) continue;
                        all.push_back({&r, -1, std::string(1, kBase[PISCES_INFO_REF(r.info)]), std::string(1, kBase[PISCES_INFO_ALT(r.info)])});
                    }
                    for (auto& sr : rows) all.push_back(sr);
                    std::stable_sort(all.begin(), all.end(), row_before);
                    for (auto& row : all) { h->pending.push_back(*row.r); h->pending_cand_index.push_back(row.ci); }
                }
            } else {
            for (auto& r : point_recs) {
                const bool is_ref = PISCES_INFO_CATEGORY(r.info) == PISCES_CAT_REFERENCE;
                if (is_ref && std::binary_search(variant_pos.begin(), variant_pos.end(), r.position)) continue;
                rows.push_back({&r, -1, std::string(1, kBase[PISCES_INFO_REF(r.info)]), std::string(1, kBase[PISCES_INFO_ALT(r.info)])});
            }
            for (size_t i = 0; i < span_recs.size(); i++) rows.push_back({&span_recs[i], (int32_t)i, span_cands[i].ref, span_cands[i].alt});
            std::stable_sort(rows.begin(), rows.end(), row_before);
            {
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
        }
        }   // (!fast_merge)
        if (!keys.empty()) h->host_time[3] += 1.0;
        h->pending_keys = keys;
        h->pending_called = called;
        h->pending_up_to = up_to_position;
        h->pending_valid = true;
    }
    HostTimer prof_out(h->prof_on ? &h->prof[9] : nullptr);
    int64_t pool_bytes = 0;
    for (auto& c : h->pending_cands) pool_bytes += (int64_t)(c.ref.size() + c.alt.size());
    if (n_cand) *n_cand = (int64_t)h->pending_cands.size();
    if (allele_bytes) *allele_bytes = pool_bytes;
    const bool cand_too_small = cand_out && ((int64_t)h->pending_cands.size() > cand_capacity || (alleles_out && pool_bytes > allele_capacity));
    const PiscesCalledAllele* pending_data = h->pending_view ? h->pending_view : h->pending.data();
    const size_t pending_n = h->pending_view ? h->pending_view_n : h->pending.size();
    auto& W = h->view;
    if (W.wanted) {
        // pisces_hip_flush_view: the rows are handed out where they lie — the pinned download buffer as the kernels wrote it, or the
        // merged rows, which then move into a vector that lives until the next flush
        if (!h->pending_view) { W.rows.swap(h->pending); pending_data = W.rows.data(); }
        W.index.swap(h->pending_cand_index);
        W.cands.resize(h->pending_cands.size());
        W.alleles.resize((size_t)pool_bytes);
        cand_out = W.cands.data();
        alleles_out = W.alleles.data();
        W.data = pending_data;
        W.n = pending_n;
    } else {
    if ((int64_t)pending_n > capacity || cand_too_small) {
        *n_out = (int64_t)pending_n;
        return fail(h, PISCES_E_BUFFER_TOO_SMALL, "flush: output buffer too small");
    }
    if (pending_n) std::memcpy(out, pending_data, pending_n * sizeof(PiscesCalledAllele));
    if (cand_index_out && pending_n) std::memcpy(cand_index_out, h->pending_cand_index.data(), pending_n * sizeof(int32_t));
    }
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
    *n_out = (int64_t)pending_n;
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
    { int32_t rcs = store_commit_flush(h, h->pending_keys); if (rcs) return rcs; }
    h->stats[0] += h->pending_called;
    h->stats[1] += h->pending_collapsed;
    h->last_up_to_block_key = final_flush ? -1 : block_key(h, up_to_position);
    h->pending_valid = false;
    h->pending_view = nullptr;
    h->pending_view_n = 0;
    h->pending.clear();
    h->pending_cand_index.clear();
    h->pending_cands.clear();
    h->pending_keys.clear();
    return PISCES_OK;
    });
}

// The flush without the copy into a caller's array: the rows (and, for insertion / deletion / MNV rows, candidate index, candidates and
// allele strings) stay in memory of the handle — for a batch the device called alone that is the pinned buffer the last kernel wrote
// them to — and the caller reads them there until the next flush.
int32_t pisces_hip_flush_view(PiscesHip* h, int32_t up_to_position, const PiscesCalledAllele** rows, int64_t* n_rows, const int32_t** cand_index,
                              const PiscesCandidate** cands, int64_t* n_cand, const uint8_t** alleles, int64_t* allele_bytes)
{
    return abi_guard<int32_t>(h, [&]() -> int32_t {
    if (!h) return PISCES_E_INVALID_ARG;
    if (!rows || !n_rows) return fail(h, PISCES_E_INVALID_ARG, "flush_view: null output");
    *rows = nullptr; *n_rows = 0;
    auto& W = h->view;
    struct Wanted { bool& f; explicit Wanted(bool& x) : f(x) { f = true; } ~Wanted() { f = false; } } wanted(W.wanted);
    W.data = nullptr; W.n = 0;
    int64_t n = 0, nc = 0, nb = 0;
    PiscesCandidate none;   // (a non-null candidate output makes the flush fill W.cands / W.alleles)
    const int32_t rc = pisces_hip_flush_ex(h, up_to_position, nullptr, 0, &n, nullptr, &none, 0, &nc, nullptr, 0, &nb);
    if (rc) return rc;
    *rows = W.data;
    *n_rows = (int64_t)W.n;
    // a batch the device called alone has no index (every row is a Reference or SNV row): NULL then
    if (cand_index) *cand_index = W.index.size() == W.n && W.n ? W.index.data() : nullptr;
    if (cands) *cands = W.cands.empty() ? nullptr : W.cands.data();
    if (n_cand) *n_cand = (int64_t)W.cands.size();
    if (alleles) *alleles = W.alleles.empty() ? nullptr : W.alleles.data();
    if (allele_bytes) *allele_bytes = (int64_t)W.alleles.size();
    return PISCES_OK;
    });
}

// pisces_hip_flush_end the same way: the rows where pisces_hip_flush_begin's work left them
int32_t pisces_hip_flush_end_view(PiscesHip* h, const PiscesCalledAllele** rows, int64_t* n_rows, const int32_t** cand_index, const PiscesCandidate** cands,
                                  int64_t* n_cand, const uint8_t** alleles, int64_t* allele_bytes)
{
    return abi_guard<int32_t>(h, [&]() -> int32_t {
    if (!h) return PISCES_E_INVALID_ARG;
    if (!rows || !n_rows) return fail(h, PISCES_E_INVALID_ARG, "flush_end_view: null output");
    *rows = nullptr; *n_rows = 0;
    // (every optional output is defined on every successful return, as pisces_hip_flush_view leaves them)
    if (cand_index) *cand_index = nullptr;
    if (cands) *cands = nullptr;
    if (n_cand) *n_cand = 0;
    if (alleles) *alleles = nullptr;
    if (allele_bytes) *allele_bytes = 0;
    auto& A = h->async;
    if (A.state == 0) return fail(h, PISCES_E_STATE, "flush_end_view: no pisces_hip_flush_begin before it");
    if (A.state == 1) {   // wait as pisces_hip_flush_end does: a call with no room for rows completes the flush and reports the count
        int64_t need = 0;
        const int32_t rc = pisces_hip_flush_end(h, nullptr, 0, &need);
        if (rc == PISCES_OK) return PISCES_OK;   // no rows at all
        if (rc != PISCES_E_BUFFER_TOO_SMALL) return rc;
        h->err.clear();   // (the probe's "buffer too small" is not an error of this call)
    }
    *rows = A.data;
    *n_rows = (int64_t)A.n;
    // (a flush that ran inside flush_begin, with host-side candidates: its index, candidates and allele strings are kept with its rows)
    const bool with_cands = A.n && A.data == A.owned.data();
    if (cand_index) *cand_index = with_cands ? A.owned_index.data() : nullptr;
    if (cands) *cands = A.n_cands ? A.owned_cands.data() : nullptr;
    if (n_cand) *n_cand = (int64_t)A.n_cands;
    if (alleles) *alleles = A.n_allele_bytes ? A.owned_alleles.data() : nullptr;
    if (allele_bytes) *allele_bytes = (int64_t)A.n_allele_bytes;
    A.state = 0;
    A.data = nullptr;
    A.n = 0;
    A.n_cands = 0;
    A.n_allele_bytes = 0;
    return PISCES_OK;
    });
}

int32_t pisces_hip_flush(PiscesHip* h, int32_t up_to_position, PiscesCalledAllele* out, int64_t capacity, int64_t* n_out)
{
    return abi_guard<int32_t>(h, [&]() -> int32_t {
    return pisces_hip_flush_ex(h, up_to_position, out, capacity, n_out, nullptr, nullptr, 0, nullptr, nullptr, 0, nullptr);
    });
}

// ---- the flush as a pair: the device works on block k while the host prepares block k + 1 ---------------------------------------
// pisces_hip_flush_begin does what pisces_hip_flush does up to the point where it would wait for the device, and commits the state
// (DoneProcessing: the flushed blocks are gone, the log is the compacted one); pisces_hip_flush_end waits for what is still in flight
// and hands the alleles over.  Between the two the caller may add the next reads (pisces_hip_stage_reads / pisces_hip_add_reads /
// pisces_hip_add_observations / pisces_hip_add_decoded_reads): the compaction's entry count, which the host has not seen yet, is
// replaced by a bound -- the compacted log is made that long, holes behind the kept entries -- so appending needs nothing from the
// device.  A batch that needs the host between its device passes (host-side candidates: insertions, deletions, MNVs; forced alleles;
// the per-locus genotypers; NoiseModel.Window; gapped-MNV reference counts) is flushed synchronously inside begin, and end returns it.
int32_t pisces_hip_flush_begin(PiscesHip* h, int32_t up_to_position)
{
    return abi_guard<int32_t>(h, [&]() -> int32_t {
    if (!h) return PISCES_E_INVALID_ARG;
    if (h->async.state != 0) return fail(h, PISCES_E_STATE, "flush_begin: the flush before this one has not been taken (pisces_hip_flush_end)");
    HostTimer timer(&h->host_time[1]);
    struct InBegin { bool& f; explicit InBegin(bool& x) : f(x) { f = true; } ~InBegin() { f = false; } } in_begin(h->in_flush_begin);
    { int32_t rcp = refuse_while_batch_is_open(h, "flush_begin"); if (rcp) return rcp; }
    PISCES_HIP_CHECK(h, hipSetDevice(h->device));
    { int32_t rcd = finish_candidate_discovery(h); if (rcd) return rcd; }
    if (!(up_to_position >= 0 && h->found.in_flight && h->found.min_position - 1 > up_to_position)) { int32_t rcf = consume_found(h); if (rcf) return rcf; }
    const bool final_flush = up_to_position < 0;
    auto& A = h->async;
    // the batch GetCandidatesToProcess would build (as pisces_hip_flush_ex)
    std::vector<int32_t> keys;
    const bool same_block = !final_flush && block_key(h, up_to_position) == h->last_up_to_block_key;
    bool plain = h->forced.empty() && h->cfg.ploidy != PISCES_PLOIDY_DIPLOID && h->cfg.ploidy != PISCES_PLOIDY_HAPLOID &&
                 h->cfg.noise_model != PISCES_NOISE_WINDOW;
    if (!same_block)
        for (auto& kv : h->blocks) {
            if (!(final_flush || (int64_t)kv.first * h->cfg.block_size <= up_to_position)) continue;
            if (!final_flush && kv.second.max_allele_endpoint > up_to_position) break;
            keys.push_back(kv.first);
            if (!kv.second.cands.empty() || !kv.second.x_spans.empty() || !kv.second.unwalked.empty()) plain = false;
        }
    // MNV calling on, split form: a batch without dirty loci is the tile kernels' alone (SNVs from the allele counts); off-interval loci are
    // dirty, and a store that has grown large is swept by a synchronous flush (the groups of flushed blocks leave it there)
    if (h->mnv_split && (!h->intervals.empty() || h->snv_ub > (4ll << 20))) plain = false;
    for (auto& kv : h->gapped_mnv_ref)
        if (std::binary_search(keys.begin(), keys.end(), block_key(h, kv.first))) { plain = false; break; }
    if (!plain) {
        // the synchronous flush, its alleles kept for pisces_hip_flush_end
        // (with the candidates of its insertion / deletion / MNV rows and their allele strings, for pisces_hip_flush_end_ex)
        A.owned.resize(std::max<size_t>(A.owned.size(), 1024));
        A.owned_index.resize(A.owned.size());
        A.owned_cands.resize(std::max<size_t>(A.owned_cands.size(), 64));
        A.owned_alleles.resize(std::max<size_t>(A.owned_alleles.size(), 4096));
        for (;;) {
            int64_t n = 0, nc = 0, nb = 0;
            const int32_t rc = pisces_hip_flush_ex(h, up_to_position, A.owned.data(), (int64_t)A.owned.size(), &n, A.owned_index.data(), A.owned_cands.data(),
                                                   (int64_t)A.owned_cands.size(), &nc, A.owned_alleles.data(), (int64_t)A.owned_alleles.size(), &nb);
            if (rc == PISCES_E_BUFFER_TOO_SMALL) {
                if ((size_t)n > A.owned.size()) { A.owned.resize((size_t)n); A.owned_index.resize((size_t)n); }
                if ((size_t)nc > A.owned_cands.size()) A.owned_cands.resize((size_t)nc);
                if ((size_t)nb > A.owned_alleles.size()) A.owned_alleles.resize((size_t)nb);
                continue;
            }
            if (rc) return rc;
            A.data = A.owned.data();
            A.n = (size_t)n;
            A.n_cands = (size_t)nc;
            A.n_allele_bytes = (size_t)nb;
            break;
        }
        A.state = 2;
        return PISCES_OK;
    }
    if (!A.done) PISCES_HIP_CHECK(h, hipEventCreateWithFlags(&A.done, hipEventDisableTiming));
    if (same_block) { A.data = nullptr; A.n = 0; A.state = 2; return PISCES_OK; }
    // what the compacted log can hold at most: the entries that are not known holes
    const int64_t bound = std::max<int64_t>(0, h->log_ub - h->log_known_holes);
    CallBlocksInFlight st;
    if (h->mnv_split) h->P.refs_only = 0;   // (no dirty locus in this batch: every SNV is the allele counts')
    int32_t rc = call_blocks_enqueue(h, keys, true, bound, &st);
    split_restore(h);
    if (rc) return rc;
    if (st.active) PISCES_HIP_CHECK(h, hipEventRecord(A.done, h->stream));
    if (!keys.empty()) h->host_time[3] += 1.0;
    // DoneProcessing, now: the blocks leave, the other log buffer is the log, `bound` slots long
    if (st.active && st.drop_now) {
        h->log_cur ^= 1;
        h->log_ub = bound;
        h->log_known_holes = bound;   // until the kept count is known every slot may be a hole (pisces_hip_flush_end corrects this)
    } else if (!st.active) {
        // no tiles: nothing was launched; the log entries of these blocks (if any) leave the ordinary way
        int32_t rcd = drop_blocks(h, keys);
        if (rcd) return rcd;
    }
    for (int32_t key : keys) {
        h->blocks.erase(key);
        const int32_t bstart = (key - 1) * h->cfg.block_size + 1, bend = key * h->cfg.block_size;
        for (auto it = h->gapped_mnv_ref.begin(); it != h->gapped_mnv_ref.end();)
            it = (it->first >= bstart && it->first <= bend) ? h->gapped_mnv_ref.erase(it) : std::next(it);
    }
    h->last_block = nullptr;
    { int32_t rcs = store_commit_flush(h, keys); if (rcs) return rcs; }
    h->last_up_to_block_key = final_flush ? -1 : block_key(h, up_to_position);
    A.dropped = st.active && st.drop_now;
    A.bound = bound;
    A.hdr = st.hdr;
    A.hrec = st.hrec;
    A.spec = st.spec;
    A.data = nullptr;
    A.n = 0;
    A.state = st.active ? 1 : 2;
    return PISCES_OK;
    });
}

int32_t pisces_hip_flush_end(PiscesHip* h, PiscesCalledAllele* out, int64_t capacity, int64_t* n_out)
{
    return pisces_hip_flush_end_ex(h, out, capacity, n_out, nullptr, nullptr, 0, nullptr, nullptr, 0, nullptr);
}

// flush_end with what pisces_hip_flush_ex returns beside the records: the candidate of every insertion / deletion / MNV row and the
// allele strings.  A flush that ran on the device alone has none (every row is a Reference or SNV row: cand_index -1).
int32_t pisces_hip_flush_end_ex(PiscesHip* h, PiscesCalledAllele* out, int64_t capacity, int64_t* n_out, int32_t* cand_index_out, PiscesCandidate* cand_out,
                                int64_t cand_capacity, int64_t* n_cand, uint8_t* alleles_out, int64_t allele_capacity, int64_t* allele_bytes)
{
    return abi_guard<int32_t>(h, [&]() -> int32_t {
    if (!h) return PISCES_E_INVALID_ARG;
    if (!n_out || capacity < 0 || (capacity > 0 && !out)) return fail(h, PISCES_E_INVALID_ARG, "flush_end: null output");
    if (cand_capacity < 0 || allele_capacity < 0 || (cand_capacity > 0 && !cand_out) || (allele_capacity > 0 && !alleles_out))
        return fail(h, PISCES_E_INVALID_ARG, "flush_end_ex: null candidate output");
    *n_out = 0;
    if (n_cand) *n_cand = 0;
    if (allele_bytes) *allele_bytes = 0;
    auto& A = h->async;
    if (A.state == 0) return fail(h, PISCES_E_STATE, "flush_end: no pisces_hip_flush_begin before it");
    HostTimer timer(&h->host_time[1]);
    PISCES_HIP_CHECK(h, hipSetDevice(h->device));
    if (A.state == 1) {
        PISCES_TIMED_WAIT(h, hipEventSynchronize(A.done));
        h->h_meta_used = 0;   // (only a flush uploads through the arena, and this one's uploads lie before the event)
        CallBlocksInFlight st;
        st.active = true; st.drop_now = A.dropped; st.hdr = A.hdr; st.hrec = A.hrec; st.spec = A.spec;
        int32_t total = 0;
        int64_t called = 0;
        unsigned long long kept = 0;
        int32_t rc = call_blocks_finish(h, st, &total, &called, &kept);
        if (rc) return rc;
        // the holes the drop left are the bound minus what it kept; entries appended since lie behind the bound
        if (A.dropped) h->log_known_holes = A.bound - (int64_t)kept;
        h->stats[0] += called;
        A.data = A.hrec;
        A.n = (size_t)total;
        A.state = 2;
    }
    const bool want_cands = cand_out || alleles_out || n_cand || allele_bytes;
    if ((int64_t)A.n > capacity || (want_cands && ((int64_t)A.n_cands > cand_capacity || (int64_t)A.n_allele_bytes > allele_capacity))) {
        *n_out = (int64_t)A.n;
        if (n_cand) *n_cand = (int64_t)A.n_cands;
        if (allele_bytes) *allele_bytes = (int64_t)A.n_allele_bytes;
        return fail(h, PISCES_E_BUFFER_TOO_SMALL, "flush_end: output buffer too small");
    }
    if (A.n) std::memcpy(out, A.data, A.n * sizeof(PiscesCalledAllele));
    if (cand_index_out) {
        if (A.n_cands || A.data == A.owned.data()) { if (A.n) std::memcpy(cand_index_out, A.owned_index.data(), A.n * sizeof(int32_t)); }
        else std::fill(cand_index_out, cand_index_out + A.n, -1);
    }
    if (want_cands) {
        if (A.n_cands) std::memcpy(cand_out, A.owned_cands.data(), A.n_cands * sizeof(PiscesCandidate));
        if (A.n_allele_bytes) std::memcpy(alleles_out, A.owned_alleles.data(), A.n_allele_bytes);
        if (n_cand) *n_cand = (int64_t)A.n_cands;
        if (allele_bytes) *allele_bytes = (int64_t)A.n_allele_bytes;
    }
    *n_out = (int64_t)A.n;
    A.state = 0;
    A.data = nullptr;
    A.n = 0;
    A.n_cands = 0;
    A.n_allele_bytes = 0;
    return PISCES_OK;
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
    PISCES_HIP_CHECK(h, accumulate_tiles(h, h->stream, h->d_tuples.p, h->d_tiles.p, n_tiles, false, true));
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
    PISCES_HIP_CHECK(h, accumulate_tiles(h, h->stream, h->d_tuples.p, h->d_tiles.p, n_tiles, true, true));   // (any handle can serve the sums, not only NoiseModel.Window)
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
    // MNV calling on, split form: the fully anchored SNV groups of the read walk are in the device's SNV store.  They are read (and stay
    // where they are) and every block's candidates are listed as the state holds them: equal candidates merged, in order of first arrival.
    std::map<int32_t, std::vector<HostCandidate>> merged_blocks;
    if (h->mnv_split && h->snv_ub > 0) {
        PISCES_HIP_CHECK(h, hipSetDevice(h->device));
        unsigned int n_store = 0;
        PISCES_HIP_CHECK(h, hipMemcpyAsync(&n_store, h->d_snv_n.p + h->snv_cur, sizeof(n_store), hipMemcpyDeviceToHost, h->stream));
        PISCES_HIP_CHECK(h, hipStreamSynchronize(h->stream));
        std::vector<SnvGroup> groups(n_store);
        if (n_store) {
            PISCES_HIP_CHECK(h, hipMemcpyAsync(groups.data(), h->d_snv[h->snv_cur].p, (size_t)n_store * sizeof(SnvGroup), hipMemcpyDeviceToHost, h->stream));
            PISCES_HIP_CHECK(h, hipStreamSynchronize(h->stream));
        }
        const bool track_open = h->cfg.collapse != 0;
        for (const SnvGroup& g : groups) {
            if (g.position < 1 || (int64_t)g.position > h->ref_len) continue;
            const int32_t key = block_key(h, g.position);
            auto it = merged_blocks.find(key);
            if (it == merged_blocks.end()) {
                it = merged_blocks.emplace(key, std::vector<HostCandidate>()).first;
                auto bi = h->blocks.find(key);
                if (bi != h->blocks.end()) it->second = bi->second.cands;
            }
            HostCandidate c;
            c.position = g.position;
            c.category = PISCES_CAT_SNV;
            c.ref.assign(1, (char)h->h_ref[(size_t)g.position - 1]);
            c.alt.assign(1, (char)g.alt);
            for (int d = 0; d < 3; d++) { c.support_by_dir[d] = g.sup[d]; c.well_anchored_by_dir[d] = g.anch[d]; }
            c.stamp = ((uint64_t)g.batch << 32) | (uint64_t)g.first;
            c.from_reads = true;
            HostCandidate* same = nullptr;
            for (auto& e : it->second)   // (a listing for inspection: a scan of the block's candidates per group is fine)
                if (e.position == c.position && e.category == c.category && e.ref == c.ref && e.alt == c.alt &&
                    (!track_open || (!e.open_left && !e.open_right))) { same = &e; break; }
            if (same) {
                for (int d = 0; d < 3; d++) { same->support_by_dir[d] += c.support_by_dir[d]; same->well_anchored_by_dir[d] += c.well_anchored_by_dir[d]; }
                same->stamp = std::min(same->stamp, c.stamp);
            } else {
                it->second.push_back(c);
            }
        }
        for (auto& kv : merged_blocks)
            std::stable_sort(kv.second.begin(), kv.second.end(), [](const HostCandidate& x, const HostCandidate& y) { return x.stamp < y.stamp; });
    }
    int64_t n = 0, bytes = 0;
    for (auto& kv : h->blocks)
        for (auto& c : (merged_blocks.count(kv.first) ? merged_blocks[kv.first] : kv.second.cands)) {
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

int32_t pisces_hip_host_time(PiscesHip* h, double out[4], int32_t reset)
{
    return abi_guard<int32_t>(h, [&]() -> int32_t {
    if (!h || !out) return PISCES_E_INVALID_ARG;
    for (int i = 0; i < 4; i++) out[i] = h->host_time[i];
    if (reset)
        for (int i = 0; i < 4; i++) h->host_time[i] = 0.0;
    return PISCES_OK;
    });
}

int32_t pisces_hip_transfer_bytes(PiscesHip* h, int64_t out[4], int32_t reset)
{
    return abi_guard<int32_t>(h, [&]() -> int32_t {
    if (!h || !out) return PISCES_E_INVALID_ARG;
    for (int i = 0; i < 4; i++) out[i] = h->pcie[i];
    if (reset)
        for (int i = 0; i < 4; i++) h->pcie[i] = 0;
    return PISCES_OK;
    });
}

int32_t pisces_hip_stats(PiscesHip* h, int64_t out[4])
{
    return abi_guard<int32_t>(h, [&]() -> int32_t {
    if (!h || !out) return PISCES_E_INVALID_ARG;
    for (int i = 0; i < 4; i++) out[i] = h->stats[i];
    return PISCES_OK;
    });
}


// ---- the host half of IAlleleCaller.Call as functions of their own (pure CPU; include/pisces_hip.h) --------------------------------------
int32_t pisces_hip_reallocate_failed_mnvs(const PiscesCandidate* failed, int64_t n_failed, const PiscesCandidate* callable, int64_t n_callable,
                                          const uint8_t* alleles, int64_t allele_bytes, int32_t block_max_position,
                                          PiscesCandidate* callable_out, int64_t callable_capacity, int64_t* n_callable_out,
                                          PiscesCandidate* outside_out, int64_t outside_capacity, int64_t* n_outside_out,
                                          uint8_t* alleles_out, int64_t allele_capacity, int64_t* allele_bytes_out)
{
    return abi_guard<int32_t>((PiscesHip*)nullptr, [&]() -> int32_t {
    if (n_failed < 0 || n_callable < 0 || (n_failed > 0 && !failed) || (n_callable > 0 && !callable) || !n_callable_out || !n_outside_out || !allele_bytes_out ||
        callable_capacity < 0 || outside_capacity < 0 || allele_capacity < 0)
        return PISCES_E_INVALID_ARG;
    // (categories as given: the reference's tests hand over alleles typed Reference that are MNVs by their strings; IsPotentialOverlap
    // looks at the type, CreateVariant types what it makes)
    auto load = [&](const PiscesCandidate* in, int64_t n, std::vector<HostCandidate>& out) -> bool {
        out.resize((size_t)n);
        for (int64_t i = 0; i < n; i++) {
            const PiscesCandidate& c = in[i];
            if (c.ref_len < 0 || c.alt_len < 0 || c.allele_offset < 0 || c.allele_offset + c.ref_len + c.alt_len > allele_bytes || (c.ref_len + c.alt_len > 0 && !alleles)) return false;
            HostCandidate& h = out[(size_t)i];
            h.position = c.position;
            h.category = c.category;
            h.ref.assign((const char*)alleles + c.allele_offset, (size_t)c.ref_len);
            h.alt.assign((const char*)alleles + c.allele_offset + c.ref_len, (size_t)c.alt_len);
            for (int d = 0; d < 3; d++) { h.support_by_dir[d] = c.support_by_dir[d]; h.well_anchored_by_dir[d] = c.well_anchored_by_dir[d]; }
        }
        return true;
    };
    std::vector<HostCandidate> f, c;
    if (!load(failed, n_failed, f) || !load(callable, n_callable, c)) return PISCES_E_INVALID_ARG;
    MnvArena arena;
    std::vector<CandPtr> failed_p, callable_p, outside;
    for (auto& x : f) failed_p.push_back(&x);
    for (auto& x : c) callable_p.push_back(&x);
    mnv_reallocate_failed(arena, failed_p, callable_p, block_max_position >= 0, block_max_position, outside);
    int64_t bytes = 0;
    for (CandPtr x : callable_p) bytes += (int64_t)(x->ref.size() + x->alt.size());
    for (CandPtr x : outside) bytes += (int64_t)(x->ref.size() + x->alt.size());
    *n_callable_out = (int64_t)callable_p.size();
    *n_outside_out = (int64_t)outside.size();
    *allele_bytes_out = bytes;
    if ((int64_t)callable_p.size() > callable_capacity || (int64_t)outside.size() > outside_capacity || bytes > allele_capacity ||
        (!callable_p.empty() && !callable_out) || (!outside.empty() && !outside_out) || (bytes > 0 && !alleles_out))
        return PISCES_E_BUFFER_TOO_SMALL;
    int64_t off = 0;
    auto store = [&](const std::vector<CandPtr>& list, PiscesCandidate* out) {
        for (size_t i = 0; i < list.size(); i++) {
            const HostCandidate& x = *list[i];
            PiscesCandidate& o = out[i];
            std::memset(&o, 0, sizeof(o));
            o.position = x.position; o.category = x.category;
            o.ref_len = (int32_t)x.ref.size(); o.alt_len = (int32_t)x.alt.size();
            for (int d = 0; d < 3; d++) { o.support_by_dir[d] = x.support_by_dir[d]; o.well_anchored_by_dir[d] = x.well_anchored_by_dir[d]; }
            o.allele_offset = off;
            std::memcpy(alleles_out + off, x.ref.data(), x.ref.size());
            std::memcpy(alleles_out + off + x.ref.size(), x.alt.data(), x.alt.size());
            off += (int64_t)(x.ref.size() + x.alt.size());
        }
    };
    store(callable_p, callable_out);
    store(outside, outside_out);
    return PISCES_OK;
    });
}

int32_t pisces_hip_set_genotypes(const PiscesHipConfig* cfg, PiscesGenotypeAllele* a, int32_t n, const uint8_t* alleles, int64_t allele_bytes)
{
    return abi_guard<int32_t>((PiscesHip*)nullptr, [&]() -> int32_t {
    if (!cfg || n < 0 || (n > 0 && !a) || (cfg->ploidy != PISCES_PLOIDY_DIPLOID && cfg->ploidy != PISCES_PLOIDY_HAPLOID)) return PISCES_E_INVALID_ARG;
    std::vector<DiploidAllele> at((size_t)n);
    for (int32_t i = 0; i < n; i++) {
        const PiscesGenotypeAllele& x = a[i];
        if (x.ref_len < 0 || x.alt_len < 0 || x.allele_offset < 0 || x.allele_offset + x.ref_len + x.alt_len > allele_bytes || (x.ref_len + x.alt_len > 0 && !alleles)) return PISCES_E_INVALID_ARG;
        DiploidAllele& d = at[(size_t)i];
        d.category = x.category;
        d.ref.assign((const char*)alleles + x.allele_offset, (size_t)x.ref_len);
        d.alt.assign((const char*)alleles + x.allele_offset + x.ref_len, (size_t)x.alt_len);
        d.support = x.support; d.coverage = x.coverage; d.ref_support = x.reference_support;
    }
    const int32_t gt = cfg->ploidy == PISCES_PLOIDY_HAPLOID
                           ? haploid_set_genotypes(at, cfg->diploid_snv_params[0], cfg->diploid_snv_params[1], cfg->min_coverage, cfg->min_genotype_qscore, cfg->max_genotype_qscore)
                           : diploid_set_genotypes(at, cfg->diploid_snv_params, cfg->diploid_indel_params, cfg->min_coverage, cfg->min_genotype_qscore, cfg->max_genotype_qscore);
    for (int32_t i = 0; i < n; i++) {
        const DiploidAllele& d = at[(size_t)i];
        a[i].genotype = d.genotype; a[i].genotype_qscore = d.genotype_qscore; a[i].phase_set_index = d.phase_set_index;
        a[i].multi_allelic = d.multi_allelic ? 1 : 0; a[i].prune = d.prune ? 1 : 0;
    }
    return gt;
    });
}

int32_t pisces_hip_diploid_genotype_qscore(int32_t genotype, int32_t total_coverage, int32_t allele_support, int32_t min_qscore, int32_t max_qscore)
{
    return abi_guard<int32_t>((PiscesHip*)nullptr, [&]() -> int32_t { return diploid_genotype_qscore(genotype, total_coverage, allele_support, min_qscore, max_qscore); });
}
