// surface_store.inc.h — part of pisces_hip.hip (included there, inside its extern "C" block; not a translation unit of its own).
// The read store behind pisces_hip_add_reads / pisces_hip_add_decoded_reads: the reads of the blocks that
// are not flushed yet stay in HBM as they came (2 bytes per base + CIGAR), in SEGMENTS of position-sorted reads with a 16-byte
// descriptor each (store_kernels.hip.h); pisces_hip_flush calls straight from them (call_store_tiles_kernel).  The host keeps, per
// segment, how many reads it holds, which of them were there at the last flush (their positions below the flushed blocks' end are
// counted already: DoneProcessing, RegionStateManager.cs:336-353, is a floor, not a compaction) and the highest block any of its reads
// touches (the segment goes when no block up to it is left).

// The position grid of a batch whose descriptors, fragments and row codes add_fused_kernel made is all that is left of its add once the
// verdict is in — and a launch enqueued THEN starts 20-25 us after the kernel before it ended (the host's turn in between), with the device
// idle.  It is kept back and enqueued in front of whatever reads or changes a grid next: the flush's kernel (store_view), the next add
// (store_shape_launch), a failed add's clean-up.  Stream order does the rest.
static int32_t store_run_deferred(PiscesHip* h)
{
    if (h->deferred_grid.empty()) return PISCES_OK;
    for (auto& d : h->deferred_grid) hipLaunchKernelGGL(read_shape_kernel, dim3(d.blocks), dim3(256), 0, h->stream, d.S);
    h->deferred_grid.clear();
    PISCES_HIP_CHECK(h, hipGetLastError());
    return PISCES_OK;
}

static void store_view(PiscesHip* h, StoreView* V)
{
    (void)store_run_deferred(h);   // (a launch error stays with the stream: the kernel that follows reports it)
    std::memset(V, 0, sizeof(*V));
    int n = 0;
    for (auto& sp : h->segments) {
        const ReadSegment& g = *sp;
        if (g.n_reads == 0 || n == kMaxSegments) continue;
        SegmentView& v = V->seg[n++];
        v.frag = g.frag.p;
        v.desc = g.desc.p;
        v.ext = g.ext.p;
        v.bases = g.v_bases;
        v.quals = g.v_quals;
        v.codes = g.v_codes;
        v.dirs = g.v_dirs;
        v.cigar_op = g.v_cop;
        v.cigar_len = g.v_clen;
        v.state = g.state;
        v.n_reads = (int32_t)g.n_reads;
        v.n_floored = (int32_t)g.n_floored;
        v.floor = g.floor;
        v.n_frags = (int32_t)g.n_ops;
        v.n_floored_frags = (int32_t)g.n_floored_ops;
        v.grid = g.grid_ok && g.grid_n > 0 ? g.grid.p : nullptr;
        v.grid_base = (int32_t)g.grid_base;
        v.grid_n = (int32_t)g.grid_n;
    }
    V->n_segments = n;
}

static bool store_is_empty(const PiscesHip* h)
{
    for (auto& sp : h->segments)
        if (sp->n_reads > 0) return false;
    return true;
}

// four zeroed state words on the device
constexpr size_t kStateSlotsPerBuffer = 1024;
static int32_t store_state_slot(PiscesHip* h, int32_t** out)
{
    if (h->state_pool.empty() || h->state_slots_used == kStateSlotsPerBuffer) {
        std::unique_ptr<DeviceBuf<int32_t>> b(new DeviceBuf<int32_t>());
        PISCES_HIP_CHECK(h, b->reserve(4 * kStateSlotsPerBuffer));
        PISCES_HIP_CHECK(h, hipMemsetAsync(b->p, 0, 4 * kStateSlotsPerBuffer * sizeof(int32_t), h->stream));
        // (earlier buffers stay: segments that are alive point into them; 16 KB per 1024 segments)
        h->state_pool.push_back(std::move(b));
        h->state_slots_used = 0;
    }
    *out = h->state_pool.back()->p + 4 * h->state_slots_used++;
    return PISCES_OK;
}

// a segment without reads (buffers of a retired segment are reused)
static int32_t store_new_segment(PiscesHip* h, std::unique_ptr<ReadSegment>* out)
{
    std::unique_ptr<ReadSegment> g;
    if (!h->segment_pool.empty()) {
        g = std::move(h->segment_pool.back());
        h->segment_pool.pop_back();
    } else {
        g.reset(new ReadSegment());
    }
    g->n_reads = g->n_bases = g->n_ops = g->n_floored = g->n_floored_ops = 0;
    g->floor = 0;
    g->max_key = 0;
    g->open = false;
    g->grid_ok = false;
    g->grid_base = g->grid_n = 0;
    g->v_bases = g->v_quals = g->v_dirs = g->v_cop = g->v_codes = nullptr;
    g->v_clen = nullptr;
    { int32_t rc = store_state_slot(h, &g->state); if (rc) return rc; }
    *out = std::move(g);
    return PISCES_OK;
}

static void store_retire(PiscesHip* h, std::unique_ptr<ReadSegment> g)
{
    // (kernels that read the segment are ordered before whatever reuses its buffers: one stream)
    if (h->segment_pool.size() < 4) h->segment_pool.push_back(std::move(g));
}

// DoneProcessing for the read store: the blocks `keys` (a prefix of the blocks there were) are gone.  Every read that is in the store
// now has its positions up to the end of the last of them counted; segments none of whose reads touch a block that is left go.
static int32_t store_commit_flush(PiscesHip* h, const std::vector<int32_t>& keys)
{
    if (h->read_path != 1 || keys.empty()) return PISCES_OK;
    const int64_t floor64 = (int64_t)keys.back() * h->cfg.block_size + 1;
    const int32_t floor = (int32_t)std::min<int64_t>(floor64, 0x7FFFFFFFll);
    const bool none_left = h->blocks.empty();
    const int32_t min_key = none_left ? 0 : h->blocks.begin()->first;
    for (size_t i = 0; i < h->segments.size();) {
        ReadSegment& g = *h->segments[i];
        const bool dead = none_left || g.max_key < min_key;
        if (dead && g.open) {
            g.n_reads = g.n_bases = g.n_ops = g.n_floored = g.n_floored_ops = 0;
            g.floor = 0;
            g.max_key = 0;
            g.v_dirs = nullptr;
            g.grid_ok = false;
            g.grid_base = g.grid_n = 0;
            { int32_t rc = store_state_slot(h, &g.state); if (rc) return rc; }
            i++;
        } else if (dead) {
            store_retire(h, std::move(h->segments[i]));
            h->segments.erase(h->segments.begin() + (std::ptrdiff_t)i);
        } else {
            g.n_floored = g.n_reads;
            g.n_floored_ops = g.n_ops;
            g.floor = std::max(g.floor, floor);
            i++;
        }
    }
    return PISCES_OK;
}

// the batch in the pinned staging buffer (laid out by L) -> device, in one piece or, when it is large, in slices whose host copies
// run under the transfers of the slices before them
// (everything but the table of candidate-record slots, which the pass over the CIGARs makes while this is on its way)
static int32_t store_upload_batch(PiscesHip* h, const PiscesReadBatch* batch, const StageLayout& L, int32_t nr, size_t n_cig, size_t n_seq, bool staged,
                                  uint8_t* d_dst)
{
    uint8_t* st = h->h_stage;
    auto place = [](uint8_t* dst, const void* src, size_t n) { if (n && (const void*)dst != src) std::memcpy(dst, src, n); };
    place(st + L.off_pos, batch->position, (size_t)nr * 4);
    place(st + L.off_flags, batch->flags, (size_t)nr);
    place(st + L.off_coff, batch->cigar_offset, ((size_t)nr + 1) * 4);
    place(st + L.off_cop, batch->cigar_op, n_cig);
    place(st + L.off_clen, batch->cigar_len, n_cig * 4);
    place(st + L.off_soff, batch->seq_offset, ((size_t)nr + 1) * 4);
    if (batch->deletion_directions) place(st + L.off_deldirs, batch->deletion_directions, 2 * n_cig);
    struct Seg { size_t dst; const uint8_t* src; size_t len; };
    const Seg segs[3] = {{L.off_bases, batch->bases, n_seq}, {L.off_quals, batch->quals, n_seq}, {L.off_dirs, batch->directions, batch->directions ? n_seq : 0}};
    const size_t bulk = 2 * n_seq + (batch->directions ? n_seq : 0);
    constexpr size_t kSlice = (size_t)8 << 20;
    if (staged || bulk < 2 * kSlice) {
        for (const Seg& g : segs) place(st + g.dst, g.src, g.len);
        PISCES_HIP_CHECK(h, hipMemcpyAsync(d_dst, st, L.off_fslots, hipMemcpyHostToDevice, h->stream));
        return PISCES_OK;
    }
    PISCES_HIP_CHECK(h, hipMemcpyAsync(d_dst, st, L.off_bases, hipMemcpyHostToDevice, h->stream));
    PISCES_HIP_CHECK(h, hipMemcpyAsync(d_dst + L.off_slots, st + L.off_slots, L.off_fslots - L.off_slots, hipMemcpyHostToDevice, h->stream));
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
    for (size_t k = 0; k < slices.size(); k++) {
        const size_t per = (slices[k].len + (size_t)n_threads - 1) / (size_t)n_threads, hi = std::min(slices[k].len, per);
        if (hi) std::memcpy(st + slices[k].dst, slices[k].src, hi);
        parts_done[k].fetch_add(1, std::memory_order_release);
        while (parts_done[k].load(std::memory_order_acquire) < n_threads) std::this_thread::yield();
        if (first_error == hipSuccess)
            first_error = hipMemcpyAsync(d_dst + slices[k].dst, st + slices[k].dst, slices[k].len, hipMemcpyHostToDevice, h->stream);
    }
    for (auto& t : pool) t.join();
    PISCES_HIP_CHECK(h, first_error);
    return PISCES_OK;
}

// Where a batch goes: a segment of its own (`direct`: its device arrays ARE the segment, nothing is copied) or the open segment (small
// batches; their bytes are appended by segment_copy_kernel).  Only the last segment is ever open, and at most kMaxSegments - 1 are closed.
struct StorePlace {
    ReadSegment* seg = nullptr;
    bool direct = false;
    bool fresh = false;        // the segment was made for this batch (it leaves again if the batch fails)
};
static int32_t store_place_batch(PiscesHip* h, size_t bulk_bytes, StorePlace* out)
{
    ReadSegment* open = (!h->segments.empty() && h->segments.back()->open) ? h->segments.back().get() : nullptr;
    const size_t n_closed = h->segments.size() - (open ? 1 : 0);
    StorePlace pl;
    if (bulk_bytes >= h->store_direct_bytes && n_closed < (size_t)kMaxSegments - 1) {
        std::unique_ptr<ReadSegment> g;
        { int32_t rc = store_new_segment(h, &g); if (rc) return rc; }
        pl.seg = g.get();
        pl.direct = true;
        pl.fresh = true;
        h->segments.insert(h->segments.end() - (open ? 1 : 0), std::move(g));   // (the open segment stays last)
    } else if (open) {
        pl.seg = open;
    } else {
        std::unique_ptr<ReadSegment> g;
        { int32_t rc = store_new_segment(h, &g); if (rc) return rc; }
        g->open = true;
        pl.seg = g.get();
        pl.fresh = true;
        h->segments.push_back(std::move(g));
    }
    *out = pl;
    return PISCES_OK;
}
static void store_unplace(PiscesHip* h, const StorePlace& pl)
{
    if (!pl.fresh) return;
    for (size_t i = 0; i < h->segments.size(); i++)
        if (h->segments[i].get() == pl.seg) {
            store_retire(h, std::move(h->segments[i]));
            h->segments.erase(h->segments.begin() + (std::ptrdiff_t)i);
            return;
        }
}
// the open segment closes once it is large (and a closed one more is allowed)
static void store_maybe_seal(PiscesHip* h, ReadSegment* g)
{
    if (!g->open) return;
    const size_t n_closed = h->segments.size() - 1;
    if ((size_t)(2 * g->n_bases + 5 * g->n_ops) >= h->store_seal_bytes && n_closed < (size_t)kMaxSegments - 1) g->open = false;
}

// Appends the batch whose arrays (offsets relative to the batch) lie on the device at `src` to the segment: descriptors always, the
// bytes only when the batch joins the open segment.
struct StoreBatchArrays {
    const int32_t* position;
    const uint8_t* flags;
    const int32_t* cigar_offset;
    const uint8_t* cigar_op;
    const uint32_t* cigar_len;
    const int32_t* seq_offset;
    const uint8_t* bases;
    const uint8_t* quals;
    const uint8_t* dirs;   // or nullptr
};
// Two halves.  store_shape_begin: room in the segment (and, for a batch that joins the open segment, its bytes copied there), the arguments of
// the shape / row-code roles filled in.  store_shape_launch: the position grid sized from what the checks reported, then read_shape_kernel
// with the roles that are still to run (`shaped`: descriptors, fragments and row codes were made by add_fused_kernel already — the grid
// role alone is left).  store_append_arrays is both in turn.
static int32_t store_shape_begin(PiscesHip* h, const StorePlace& pl, const StoreBatchArrays& A, int32_t nr, size_t n_cig, size_t n_seq, ShapeArgs* S_out)
{
    ReadSegment& g = *pl.seg;
    if (g.n_reads + nr > 0x7FFFFF00ll || g.n_ops + (int64_t)n_cig > 0x7FFFFF00ll) return fail(h, PISCES_E_INVALID_ARG, "add_reads: too many reads held at once");
    // a fragment keeps the offset of its first base in 32 bits (kFragAoffMask): a segment's bases stay below 4 GB
    if ((unsigned long long)g.n_bases + n_seq + 2ull * kSegmentPad > 0xFFFFFFFFull)
        return fail(h, PISCES_E_INVALID_ARG, "add_reads: more than 4 GB of bases in one segment of the read store (flush, or hand the reads over in smaller batches)");
    ShapeArgs S;
    S.position = A.position; S.flags = A.flags; S.cigar_offset = A.cigar_offset; S.cigar_op = A.cigar_op; S.cigar_len = A.cigar_len;
    S.seq_offset = A.seq_offset;
    S.n_reads = nr;
    if (pl.direct) {
        PISCES_HIP_CHECK(h, g.desc.reserve((size_t)nr));
        PISCES_HIP_CHECK(h, g.ext.reserve((size_t)nr));
        PISCES_HIP_CHECK(h, g.frag.reserve(n_cig + 1));
        PISCES_HIP_CHECK(h, g.codes.reserve(n_seq + 2 * (size_t)kSegmentPad));
        g.v_bases = A.bases; g.v_quals = A.quals; g.v_dirs = A.dirs; g.v_cop = A.cigar_op; g.v_clen = A.cigar_len;
        g.v_codes = g.codes.p + kSegmentPad;
        S.n0 = 0; S.base0 = 0; S.ops0 = 0;
    } else {
        const size_t nb = (size_t)g.n_bases, no = (size_t)g.n_ops, n0 = (size_t)g.n_reads;
        constexpr size_t kPad = (size_t)kSegmentPad;   // room on both sides of the byte arrays (walk_segment loads whole words around a read's ends)
        PISCES_HIP_CHECK(h, g.desc.grow_keep(n0 + (size_t)nr, n0, h->stream));
        PISCES_HIP_CHECK(h, g.ext.grow_keep(n0 + (size_t)nr, n0, h->stream));
        PISCES_HIP_CHECK(h, g.frag.grow_keep(no + n_cig + 1, no, h->stream));
        PISCES_HIP_CHECK(h, g.bases.grow_keep(nb + n_seq + 2 * kPad, nb + kPad, h->stream));
        PISCES_HIP_CHECK(h, g.quals.grow_keep(nb + n_seq + 2 * kPad, nb + kPad, h->stream));
        PISCES_HIP_CHECK(h, g.codes.grow_keep(nb + n_seq + 2 * kPad, nb + kPad, h->stream));
        PISCES_HIP_CHECK(h, g.cop.grow_keep(no + n_cig + 16, no, h->stream));
        PISCES_HIP_CHECK(h, g.clen.grow_keep(no + n_cig + 16, no, h->stream));
        const bool tracks = g.v_dirs != nullptr;
        const bool wants = A.dirs != nullptr || tracks;
        if (wants) PISCES_HIP_CHECK(h, g.dirs.grow_keep(std::max(g.bases.cap, nb + n_seq + 2 * kPad), tracks ? nb + kPad : 0, h->stream));
        g.v_bases = g.bases.p + kPad; g.v_quals = g.quals.p + kPad; g.v_cop = g.cop.p; g.v_clen = g.clen.p;
        g.v_dirs = wants ? g.dirs.p + kPad : nullptr;
        g.v_codes = g.codes.p + kPad;
        uint8_t* const w_bases = g.bases.p + kPad;
        uint8_t* const w_quals = g.quals.p + kPad;
        uint8_t* const w_dirs = wants ? g.dirs.p + kPad : nullptr;
        CopyRanges C;
        std::memset(&C, 0, sizeof(C));
        C.dst[0] = w_bases + nb; C.src[0] = A.bases; C.n[0] = (int64_t)n_seq;
        C.dst[1] = w_quals + nb; C.src[1] = A.quals; C.n[1] = (int64_t)n_seq;
        C.dst[2] = g.cop.p + no; C.src[2] = A.cigar_op; C.n[2] = (int64_t)n_cig;
        C.dst[3] = (uint8_t*)(g.clen.p + no); C.src[3] = (const uint8_t*)A.cigar_len; C.n[3] = (int64_t)n_cig * 4;
        if (A.dirs) { C.dst[4] = w_dirs + nb; C.src[4] = A.dirs; C.n[4] = (int64_t)n_seq; }
        const int64_t most = std::max<int64_t>((int64_t)n_seq, (int64_t)n_cig * 4);
        if (most > 0)
            hipLaunchKernelGGL(segment_copy_kernel, dim3((unsigned)std::min<int64_t>((most + 255) / 256, 2048)), dim3(256), 0, h->stream, C);
        S.n0 = (int32_t)n0; S.base0 = (int64_t)nb; S.ops0 = (int64_t)no;
        if (A.dirs && !tracks && n0 > 0)   // the reads the segment held before its first batch with directions
            hipLaunchKernelGGL(segment_fill_dirs_kernel, dim3((unsigned)((n0 + 3) / 4)), dim3(256), 0, h->stream, (const ReadDesc*)g.desc.p,
                               (const ReadExt*)g.ext.p, 0, (int32_t)n0, w_dirs);
    }
    S.desc = g.desc.p;
    S.ext = g.ext.p;
    S.frag = g.frag.p;
    S.quals = A.quals;
    S.dirs = A.dirs;
    S.min_bq = h->cfg.min_base_call_quality;
    S.state = g.state;
    // the row codes of the batch's bases (what the flush kernel walks instead of bases + qualities), in the same launch: the workgroups
    // behind the shape's own
    S.enc_bases = A.bases;
    S.enc_quals = A.quals;
    S.enc_codes = const_cast<uint8_t*>(g.v_codes) + S.base0;
    S.enc_n = (int64_t)n_seq;
    S.enc_min_bq = (uint32_t)std::min(std::max(h->cfg.min_base_call_quality, 0), 127);
    S.shape_blocks = (nr + 255) / 256;
    S.enc_blocks = (int32_t)std::min<int64_t>(((int64_t)n_seq + 16 * 256 - 1) / (16 * 256), 8192);
    S.grid = nullptr;
    S.grid_base = S.grid_n = 0;
    *S_out = S;
    return PISCES_OK;
}
static int32_t store_shape_launch(PiscesHip* h, const StorePlace& pl, ShapeArgs S, bool batch_has_dirs, int32_t nr, size_t n_cig, int32_t min_position, int32_t max_key, bool shaped)
{
    ReadSegment& g = *pl.seg;
    if (shaped) S.shape_blocks = S.enc_blocks = 0;
    { int32_t rcd = store_run_deferred(h); if (rcd) return rcd; }   // (grids of earlier batches: this one's may be an extension of theirs, and may move)
    {   // the position grid (what gives a tile its fragment range): extended over the batch's span, or given up for this segment
        const bool first = g.n_reads == 0;
        bool ok = min_position > 0 && (first || g.grid_ok);
        int64_t base = g.grid_base, cells = g.grid_n;
        if (ok) {
            const int64_t c_lo = (int64_t)min_position;
            const int64_t c_hi = std::min<int64_t>(((int64_t)max_key + 2) * h->cfg.block_size + 2, 0x7FFFFFFFll);   // (past the last block a read of the batch touches)
            if (first) { base = c_lo; cells = 0; }
            // worth it for dense reads only: at most sixteen cells a CIGAR operation held (500x of 150-base reads: a cell per ~3.3; 10x: one per 0.07)
            if (c_lo < base || c_hi - base + 1 > std::max<int64_t>(1ll << 18, 16 * (g.n_ops + (int64_t)n_cig))) ok = false;
            else if (c_hi - base + 1 > cells) {
                const int64_t want = c_hi - base + 1;
                // (the new cells need no fill: the batch's first read fills from the read before it up to its own position, its last read
                // from its own position to the grid's end — grid_cells)
                PISCES_HIP_CHECK(h, g.grid.grow_keep((size_t)want, (size_t)cells, h->stream));
                cells = want;
            }
            // a batch that touches no block (every read soft-clipped away or without operations: max_key == 0, c_hi < base) leaves no cell; the
            // next batch of the segment would fill from its own first read on and the cells in between would keep the fill value (a tile
            // there would see an empty fragment range): such a segment goes without a grid
            if (cells == 0) ok = false;
        }
        g.grid_ok = ok;
        g.grid_base = ok ? base : 0;
        g.grid_n = ok ? cells : 0;
        if (ok) { S.grid = g.grid.p; S.grid_base = (int32_t)base; S.grid_n = (int32_t)cells; }
    }
    const unsigned grid_blocks = S.grid ? (unsigned)((nr + 255) / 256) : 0u;
    const unsigned n_blocks = (unsigned)S.shape_blocks + (unsigned)S.enc_blocks + grid_blocks;
    if (n_blocks && shaped && h->defer_grid) h->deferred_grid.push_back({S, n_blocks});   // (the grid role alone: enqueued with what comes next)
    else if (n_blocks) { hipLaunchKernelGGL(read_shape_kernel, dim3(n_blocks), dim3(256), 0, h->stream, S); h->chain_enqueued_since = true; }
    if (!pl.direct && !batch_has_dirs && g.v_dirs) h->chain_enqueued_since = true;
    if (!pl.direct && !batch_has_dirs && g.v_dirs)   // a batch without directions in a segment that tracks them
        hipLaunchKernelGGL(segment_fill_dirs_kernel, dim3((unsigned)((nr + 3) / 4)), dim3(256), 0, h->stream, (const ReadDesc*)g.desc.p,
                           (const ReadExt*)g.ext.p, S.n0, S.n0 + nr, const_cast<uint8_t*>(g.v_dirs));
    PISCES_HIP_CHECK(h, hipGetLastError());
    return PISCES_OK;
}
static int32_t store_append_arrays(PiscesHip* h, const StorePlace& pl, const StoreBatchArrays& A, int32_t nr, size_t n_cig, size_t n_seq,
                                   int32_t min_position = 0 /* lowest read position of the batch, 0 = unknown */, int32_t max_key = 0 /* highest block a read touches */)
{
    ShapeArgs S;
    { int32_t rc = store_shape_begin(h, pl, A, nr, n_cig, n_seq, &S); if (rc) return rc; }
    return store_shape_launch(h, pl, S, A.dirs != nullptr, nr, n_cig, min_position, max_key, false);
}

// The checks and the bookkeeping of a batch that lies in device memory, made THERE, in ONE launch (add_fused_kernel, store_kernels.hip.h):
// what add_reads_store's host pass over the CIGARs makes for a batch that came from the host.  `src` (or nullptr): the caller's device
// arrays — the launch then also copies them into the store's layout at d (laid out by L), reading bases and qualities once; otherwise the
// batch lies at d already (a host batch behind its upload).  `shape` (or nullptr; a batch that becomes a segment of its own): descriptors,
// fragments and row codes are made by the same launch (store_shape_begin filled the arguments).  One wait: the verdict, the span, the
// totals and the touched blocks arrive in pinned memory as the launch's own stores.  found_slots / found_pool: the candidate-record slots
// (MNV calling off), scanned by the launch itself, at d + L.off_fslots.
static int32_t store_device_checks(PiscesHip* h, uint8_t* d, const StageLayout& L, int32_t nr, size_t n_cig, size_t n_seq, bool has_dirs, bool has_deldirs,
                                   bool count_indels, int64_t* found_slots, int64_t* found_pool, std::vector<int32_t>& touched, int32_t* max_key, int32_t* min_position,
                                   const PiscesReadBatch* src = nullptr, const ShapeArgs* shape = nullptr)
{
    *found_slots = *found_pool = 0;
    *max_key = 0;
    *min_position = 0;
    touched.clear();
    if (nr <= 0) return PISCES_OK;
    const int32_t bs = h->cfg.block_size;
    const int64_t n_block_bits = (0x7FFFFFFFll + bs - 1) / bs + 2;
    if (n_block_bits > (1ll << 27)) return fail(h, PISCES_E_UNSUPPORTED, "add_reads: a batch on the device needs a block size of 16 positions or more");
    const size_t map_words = (size_t)((n_block_bits + 31) / 32);
    // the map in kPrepReplicas copies (prepare_reads: a workgroup sets bits in the copy of its index) while that stays small: 8.6 MB at
    // the default block size of 1000; a small block size would make it hundreds of MB to clear and to fold, so from 32 MB on the copies
    // alias one map (stride 0: the kernels are the same, the workgroups share the words again)
    const size_t map_copies = map_words * kPrepReplicas * sizeof(uint32_t) > (32u << 20) ? 1 : (size_t)kPrepReplicas;
    const size_t map_stride = map_copies == 1 ? 0 : map_words;
    if (h->prep_map_copies != map_copies) h->prep_map_clean = false;
    h->prep_map_copies = map_copies;
    { const size_t before = h->d_prep_map.cap; PISCES_HIP_CHECK(h, h->d_prep_map.reserve(map_words * map_copies)); if (h->d_prep_map.cap != before) h->prep_map_clean = false; }
    PISCES_HIP_CHECK(h, h->d_found_pool_first.reserve((size_t)nr + 1));
    // (the map is zero outside the span of the last batch that used it, and the collecting workgroup clears that: first use: all of it)
    if (!h->prep_map_clean) {
        PISCES_HIP_CHECK(h, hipMemsetAsync(h->d_prep_map.p, 0, map_words * map_copies * sizeof(uint32_t), h->stream));
        h->prep_map_clean = true;
    }
    // the launch's shared words — kPrepReplicas x {lowest key, highest key, lowest position, X / = seen}, the first-error word, the count of
    // workgroups that are through, the totals — are set once: the collecting workgroup of every launch leaves them as the next one expects
    constexpr size_t kWordsSpan = 0, kWordsError = 4 * kPrepReplicas * sizeof(int32_t), kWordsDone = kWordsError + 8, kWordsTotals = kWordsDone + 8 * ((1 + kPrepReplicas) * 4 / 8 + 1), kWordsBytes = kWordsTotals + 16;
    if (!h->d_fused_words.p) {
        PISCES_HIP_CHECK(h, h->d_fused_words.reserve(kWordsBytes));
        uint8_t init[kWordsBytes];
        std::memset(init, 0, sizeof(init));
        int32_t* sp = (int32_t*)init;
        for (int r = 0; r < kPrepReplicas; r++) { sp[4 * r] = 0x7FFFFFFF; sp[4 * r + 1] = 0; sp[4 * r + 2] = 0x7FFFFFFF; sp[4 * r + 3] = 0; }
        std::memset(init + kWordsError, 0xFF, 8);
        { int32_t rcu = meta_upload(h, h->d_fused_words.p, init, sizeof(init)); if (rcu) return rcu; }
    }
    const int32_t read_blocks = (nr + 255) / 256;
    {   // the look-back's words: zero between launches (cleared by the collecting workgroup); a new buffer starts zeroed
        const size_t before = h->d_fused_scan.cap;
        PISCES_HIP_CHECK(h, h->d_fused_scan.reserve((size_t)read_blocks + 1));
        if (h->d_fused_scan.cap != before) PISCES_HIP_CHECK(h, hipMemsetAsync(h->d_fused_scan.p, 0, h->d_fused_scan.cap * sizeof(unsigned long long), h->stream));
    }
    constexpr int32_t kPrepKeys = 8192;
    if (!h->h_prep) PISCES_HIP_CHECK(h, host_alloc((void**)&h->h_prep, sizeof(PrepVerdict) + (size_t)kPrepKeys * sizeof(int32_t) + 16));
    PrepVerdict* const verdict = (PrepVerdict*)h->h_prep;
    int32_t* const keys = (int32_t*)(h->h_prep + sizeof(PrepVerdict));
    int32_t* const bad_direction = keys + kPrepKeys;   // (set by the launch's stream role, which no other workgroup waits for)
    *bad_direction = 0;
    AddFusedArgs F;
    std::memset(&F, 0, sizeof(F));
    PrepareArgs& A = F.P;
    // (the checks read the batch where it lies when the launch starts: the caller's arrays while the copy is being made)
    A.position = src ? src->position : (const int32_t*)(d + L.off_pos);
    A.cigar_offset = src ? src->cigar_offset : (const int32_t*)(d + L.off_coff);
    A.cigar_op = src ? src->cigar_op : d + L.off_cop;
    A.cigar_len = src ? src->cigar_len : (const uint32_t*)(d + L.off_clen);
    A.seq_offset = src ? src->seq_offset : (const int32_t*)(d + L.off_soff);
    A.quals = src ? src->quals : d + L.off_quals;
    A.del_dirs = has_deldirs ? (src ? src->deletion_directions : d + L.off_deldirs) : nullptr;
    A.n_reads = nr; A.min_bq = h->cfg.min_base_call_quality; A.block_size = bs; A.count_indels = count_indels ? 1 : 0;
    A.n_ops_total = (int64_t)n_cig; A.n_bases_total = (int64_t)n_seq;
    A.block_bits = h->d_prep_map.p; A.n_block_bits = n_block_bits; A.map_stride = (int64_t)map_stride;
    A.n_found = count_indels ? (int32_t*)(d + L.off_fslots) : nullptr;
    A.n_pool = count_indels ? h->d_found_pool_first.p : nullptr;
    A.first_error = (unsigned long long*)(h->d_fused_words.p + kWordsError);
    A.key_span = (int32_t*)(h->d_fused_words.p + kWordsSpan);
    if (shape) {
        F.S = *shape;
        F.do_shape = 1;
        if (src) {   // (the shape role walks the same arrays as the checks)
            F.S.position = src->position; F.S.flags = src->flags; F.S.cigar_offset = src->cigar_offset; F.S.cigar_op = src->cigar_op; F.S.cigar_len = src->cigar_len;
            F.S.seq_offset = src->seq_offset; F.S.quals = src->quals; F.S.dirs = has_dirs ? src->directions : nullptr;
        }
        F.d_codes = shape->enc_codes;
    }
    F.s_bases = src ? src->bases : d + L.off_bases;
    F.s_quals = src ? src->quals : d + L.off_quals;
    F.s_dirs = has_dirs ? (src ? src->directions : d + L.off_dirs) : nullptr;
    if (src) { F.d_bases = d + L.off_bases; F.d_quals = d + L.off_quals; F.d_dirs = has_dirs ? d + L.off_dirs : nullptr; }
    F.n_seq = (int64_t)n_seq;
    F.enc_min_bq = (uint32_t)std::min(std::max(h->cfg.min_base_call_quality, 0), 127);
    int64_t misc_bytes = 0;
    if (src) {
        int k = 0;
        auto range = [&](size_t off, const void* from, size_t bytes) { F.C.dst[k] = d + off; F.C.src[k] = (const uint8_t*)from; F.C.n[k] = (int64_t)bytes; misc_bytes = std::max<int64_t>(misc_bytes, (int64_t)bytes); k++; };
        range(L.off_pos, src->position, (size_t)nr * 4);
        range(L.off_flags, src->flags, (size_t)nr);
        range(L.off_coff, src->cigar_offset, ((size_t)nr + 1) * 4);
        range(L.off_cop, src->cigar_op, n_cig);
        range(L.off_clen, src->cigar_len, n_cig * 4);
        range(L.off_soff, src->seq_offset, ((size_t)nr + 1) * 4);
        if (has_deldirs) range(L.off_deldirs, src->deletion_directions, 2 * n_cig);
    }
    F.read_blocks = read_blocks;
    // the stream role runs when there is something to copy, to encode or to check
    const bool stream = n_seq > 0 && (src || F.d_codes || has_dirs);
    // (four sixteen-byte pieces of each array a lane a trip — add_fused_kernel's kStreamPieces — and every lane the same number of trips)
    F.stream_blocks = stream ? (int32_t)std::min<int64_t>(((int64_t)n_seq + 4 * 16 * 256 - 1) / (4 * 16 * 256), 16384) : 0;
    // PISCES_HIP_STREAM_WGS_PER_CU = k > 0: at most k x CUs persistent stream workgroups, in FRONT of the read workgroups (add_fused_kernel)
    F.stream_first = 0;
    if (h->stream_wgs_per_cu > 0 && F.stream_blocks > 0) {
        F.stream_blocks = (int32_t)std::min<int64_t>(F.stream_blocks, (int64_t)h->stream_wgs_per_cu * h->n_cus);
        F.stream_first = 1;
    }
    F.misc_blocks = src ? (int32_t)std::min<int64_t>(std::max<int64_t>((misc_bytes / 16 + 255) / 256, 1), 64) : 0;
    // (1: the read workgroups in front.  Measured, XCD-aware, on config 2's batch: every 2nd / 3rd / 4th / 8th unit of eight workgroups a read
    // unit: 96 / 95 / 115 / 106 us against 83 with the read role in front; PISCES_HIP_ROLE_STRIDE for the A / B)
    F.role_stride = std::max<int32_t>(1, std::min<int32_t>(h->role_stride, (F.read_blocks + F.stream_blocks + F.misc_blocks) / std::max((F.read_blocks + 7) / 8 * 8, 1)));
    F.scan_state = h->d_fused_scan.p;
    F.done = (unsigned int*)(h->d_fused_words.p + kWordsDone);
    F.totals = (long long*)(h->d_fused_words.p + kWordsTotals);
    F.verdict = verdict;
    F.keys_out = keys;
    F.capacity = kPrepKeys;
    F.bad_direction = bad_direction;
    if (++h->fused_seq <= 0) h->fused_seq = 1;
    F.seq = h->fused_seq;
    verdict->ready = 0;
#ifdef PISCES_ADD_STAMPS
    const size_t n_wg = (size_t)(F.read_blocks + F.stream_blocks + F.misc_blocks);
    PISCES_HIP_CHECK(h, h->d_add_stamps.reserve(n_wg * 8));
    PISCES_HIP_CHECK(h, hipMemsetAsync(h->d_add_stamps.p, 0, n_wg * 8 * sizeof(long long), h->stream));
    PISCES_HIP_CHECK(h, hipStreamSynchronize(h->stream));
    F.stamps = h->d_add_stamps.p;
#endif
    // (pisces_hip_set_chain_timing: the add's device time starts with its first kernel — what the host does before it is not the device's)
    h->chain_have[0] = false;
    h->chain_add_open = false;
    if (h->chain_timing && src) { PISCES_HIP_CHECK(h, hipEventRecord(h->ev_chain[0], h->stream)); h->chain_add_open = true; }
    hipLaunchKernelGGL(add_fused_kernel, dim3((unsigned)(F.read_blocks + F.stream_blocks + F.misc_blocks)), dim3(256), 0, h->stream, F);
    PISCES_HIP_CHECK(h, hipGetLastError());
    // (the span's end goes in behind the launch at once: an event recorded after the verdict has come back would count the host's turn as
    // the device's; whatever the add enqueues later — candidate discovery, a grid that is not kept back — records it again, behind itself)
    if (h->chain_add_open) { PISCES_HIP_CHECK(h, hipEventRecord(h->ev_chain[1], h->stream)); h->chain_enqueued_since = false; }
    {   // The one wait of an add.  The collecting workgroup stores the launch's number behind the verdict: polling that word in pinned memory
        // sees it ~2 us after the store, where hipStreamSynchronize's wake-up takes 15-25 us during which the device has nothing to do (the
        // position grid and candidate discovery are enqueued behind the verdict).  The stream's other work is ordered by the stream itself.
        // A launch that never reports (a device fault) falls back to the stream's own wait and error after 0.2 s.
        const volatile int32_t* const ready = &verdict->ready;
        const auto t0 = std::chrono::steady_clock::now();
        bool seen = false;
        // (per-base directions are checked by the stream role, which the verdict does not wait for: such a batch waits for the whole launch)
        for (int64_t spin = 0; !has_dirs && !(seen = (*ready == F.seq)); spin++) {
            if ((spin & 1023) == 1023 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(200)) break;
            __builtin_ia32_pause();
        }
        std::atomic_thread_fence(std::memory_order_acquire);
        if (!seen) {
            PISCES_HIP_CHECK(h, hipStreamSynchronize(h->stream));
            // (the launch is over: its verdict is there, or the launch is broken — never a verdict of an earlier launch)
            if (*ready != F.seq) return fail(h, PISCES_E_DEVICE, "add_reads: the launch that checks the batch ended without its verdict");
        }
    }
#ifdef PISCES_ADD_STAMPS
    {   // slots of a workgroup: read role [0] start, [2] prepare done, [3] shape done, [4] scan done, [5] counted, [6] collector done; stream role [2] start, [3] end
        std::vector<long long> st(n_wg * 8);
        PISCES_HIP_CHECK(h, hipStreamSynchronize(h->stream));
        PISCES_HIP_CHECK(h, hipMemcpy(st.data(), h->d_add_stamps.p, st.size() * sizeof(long long), hipMemcpyDeviceToHost));
        long long t0 = 0x7FFFFFFFFFFFFFFFll;
        for (size_t w = 0; w < n_wg; w++) for (int k : {0, 2}) if (st[w * 8 + k]) t0 = std::min(t0, st[w * 8 + k]);
        double rs_min = 1e9, rs_max = 0, prep = 0, shape = 0, scan = 0, counted = 0, coll = 0, ss_min = 1e9, ss_max = 0, se_max = 0, dur_r = 0, dur_s = 0;
        int nr_ = 0, ns_ = 0;
        for (size_t w = 0; w < n_wg; w++) {
            const long long* q = &st[w * 8];
            auto us = [&](int k) { return (double)(q[k] - t0) / 100.0; };
            if (q[0]) {   // a read workgroup
                rs_min = std::min(rs_min, us(0)); rs_max = std::max(rs_max, us(0)); prep = std::max(prep, us(2)); shape = std::max(shape, us(3));
                scan = std::max(scan, us(4)); if (q[5]) counted = std::max(counted, us(5)); if (q[6]) coll = us(6);
                dur_r += us(q[5] ? 5 : 4) - us(0); nr_++;
            } else if (q[2]) {
                ss_min = std::min(ss_min, us(2)); ss_max = std::max(ss_max, us(2)); se_max = std::max(se_max, us(3)); dur_s += us(3) - us(2); ns_++;
            }
        }
        fprintf(stderr, "add_fused stamps (us from the first start): read role: starts %.1f .. %.1f, last prepare done %.1f, shape %.1f, scan %.1f, counted %.1f, collector done %.1f, "
                "mean workgroup %.1f us (%d) | stream role: starts %.1f .. %.1f, last end %.1f, mean workgroup %.1f us (%d)\n",
                rs_min, rs_max, prep, shape, scan, counted, coll, dur_r / std::max(nr_, 1), nr_, ss_min, ss_max, se_max, dur_s / std::max(ns_, 1), ns_);
    }
#endif
    h->h_meta_used = 0;
    const unsigned long long first_error = std::min<unsigned long long>(verdict->first_error, *bad_direction ? (unsigned long long)kPrepBadDirection : ~0ull);
    const int32_t span[3] = {verdict->span[0], verdict->span[1], verdict->span[2]};
    h->eqx_in_batch = count_indels && verdict->has_eqx != 0;
    const long long totals[2] = {count_indels ? verdict->totals[0] : 0, count_indels ? verdict->totals[1] : 0};
    if (verdict->n_keys <= kPrepKeys) {
        touched.assign(keys, keys + verdict->n_keys);
        std::sort(touched.begin(), touched.end());
        if (!touched.empty()) *max_key = touched.back();
    } else if (span[1] >= span[0] && span[1] > 0) {
        // (more touched blocks than the kernel had room for: the bits of [span[0], span[1]], which are cleared again behind the copy)
        std::vector<uint32_t> words;
        const size_t w0 = (size_t)span[0] >> 5, w1 = (size_t)span[1] >> 5;
        words.resize(w1 - w0 + 1);
        PISCES_HIP_CHECK(h, hipMemcpyAsync(words.data(), h->d_prep_map.p + w0, words.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream));
        PISCES_HIP_CHECK(h, hipMemsetAsync(h->d_prep_map.p + w0, 0, words.size() * sizeof(uint32_t), h->stream));
        PISCES_HIP_CHECK(h, hipStreamSynchronize(h->stream));
        for (size_t w = 0; w < words.size(); w++)
            for (uint32_t bits = words[w]; bits; bits &= bits - 1) touched.push_back((int32_t)((w0 + w) * 32 + (size_t)__builtin_ctz(bits)));
        if (!touched.empty()) *max_key = touched.back();
    }
    if (first_error != ~0ull) {
        const std::string read = " (read " + std::to_string((long long)(first_error >> 3)) + " of the batch)";
        switch ((int)(first_error & 7ull)) {
            case kPrepPositionNotPositive: return fail(h, PISCES_E_INVALID_ARG, "Position must be greater than 0." + read);
            case kPrepMalformed: return fail(h, PISCES_E_INVALID_ARG, "add_reads: malformed read batch" + read);
            case kPrepPastInt32: return fail(h, PISCES_E_INVALID_ARG, "add_reads: read runs past position 2^31 - 1" + read);
            case kPrepBadDeletionDirection: return fail(h, PISCES_E_INVALID_ARG, "add_reads: deletion_directions holds a value that is no DirectionType" + read);
            default: return fail(h, PISCES_E_INVALID_ARG, "add_reads: CIGAR does not match the read" + ((first_error & 7ull) == kPrepBadDirection ? std::string() : read));
        }
    }
    if (totals[0] > 0x7FFFFFF0ll || totals[1] > 0x7FFFFFF0ll) return fail(h, PISCES_E_INVALID_ARG, "add_reads: too many insertions / deletions in one batch");
    *found_slots = totals[0];
    *found_pool = totals[1];
    *min_position = span[2] == 0x7FFFFFFF ? 0 : span[2];
    return PISCES_OK;
}

// The tail of every add into the read store, once the batch's arrays lie on the device at d and its checks are in (rc_in: their verdict):
// descriptors and fragments (read_shape_kernel), candidate discovery, and — only now — the handle's state.  fslots_host: the candidate-record
// slots the host pass made (uploaded here), or nullptr when they were made on the device.
static int32_t store_finish_add(PiscesHip* h, StorePlace& pl, int32_t rc, uint8_t* d, const StageLayout& L, int32_t nr, size_t n_cig, size_t n_seq, bool has_dirs,
                                bool has_deldirs, bool find_on_device, int64_t found_slots, int64_t found_pool, const int32_t* fslots_host,
                                const std::vector<int32_t>& touched, int32_t max_key, int32_t min_position = 0, const ShapeArgs* shaped = nullptr /* add_fused_kernel made them */)
{
    const int32_t bs = h->cfg.block_size;
    DevReadBatch db;
    db.position = (const int32_t*)(d + L.off_pos);
    db.flags = d + L.off_flags;
    db.cigar_offset = (const int32_t*)(d + L.off_coff);
    db.cigar_op = d + L.off_cop;
    db.cigar_len = (const uint32_t*)(d + L.off_clen);
    db.seq_offset = (const int32_t*)(d + L.off_soff);
    db.bases = d + L.off_bases;
    db.quals = d + L.off_quals;
    db.dirs = has_dirs ? d + L.off_dirs : nullptr;
    db.n_reads = nr;
    if (rc == PISCES_OK) {
        const StoreBatchArrays A = {db.position, db.flags, db.cigar_offset, db.cigar_op, db.cigar_len, db.seq_offset, db.bases, db.quals, db.dirs};
        if (shaped) rc = store_shape_launch(h, pl, *shaped, A.dirs != nullptr, nr, n_cig, min_position, max_key, true);   // (the position grid is all that is left)
        else rc = store_append_arrays(h, pl, A, nr, n_cig, n_seq, min_position, max_key);
    }
    // ICandidateVariantFinder.FindCandidates + IStateManager.AddCandidates (SmallVariantCaller.cs:92-96) on the device, on the batch as it lies there
    if (rc == PISCES_OK && find_on_device && (h->snv_walk || found_slots > 0 || h->eqx_in_batch)) {
        if (fslots_host) {
            std::memcpy(h->h_stage + L.off_fslots, fslots_host, ((size_t)nr + 1) * 4);
            hipError_t e = hipMemcpyAsync(d + L.off_fslots, h->h_stage + L.off_fslots, L.total - L.off_fslots, hipMemcpyHostToDevice, h->stream);
            if (e != hipSuccess) rc = fail(h, PISCES_E_DEVICE, std::string("add_reads: ") + hipGetErrorString(e));
        }
        h->chain_enqueued_since = true;
        if (rc == PISCES_OK)
            rc = enqueue_candidate_discovery(h, db, has_deldirs ? d + L.off_deldirs : nullptr, nr, (const int32_t*)(d + L.off_fslots), found_slots, found_pool);
        h->found.min_position = min_position;   // (a flush up to a position below every read of this batch need not wait for its candidates)
    }
    { int32_t rcs = stage_release(h); if (rc == PISCES_OK) rc = rcs; }   // (transfers out of the pinned buffer may be in flight whatever happened after them)
    if (rc) {
        (void)store_run_deferred(h);   // (before the segment's buffers can go)
        (void)hipStreamSynchronize(h->stream);
        pl.seg->grid_ok = false;   // (the refused batch may have written cells of an open segment's position grid: the segment goes without)
        store_unplace(h, pl);
        return rc;
    }
    // (pisces_hip_set_chain_timing: behind the last thing the add enqueues; what follows is the host's bookkeeping)
    if (h->chain_timing && h->chain_add_open) {
        if (h->chain_enqueued_since) PISCES_HIP_CHECK(h, hipEventRecord(h->ev_chain[1], h->stream));
        h->chain_have[0] = true;
    }
    h->chain_add_open = false;
    // ---- commit
    ReadSegment& g = *pl.seg;
    g.n_reads += nr;
    g.n_bases += (int64_t)n_seq;
    g.n_ops += (int64_t)n_cig;
    g.max_key = std::max(g.max_key, max_key);
    store_maybe_seal(h, &g);
    for (int32_t k : touched) (void)get_block(h, (k - 1) * bs + 1);
    h->stats[2] += nr;
    return PISCES_OK;
}

// pisces_hip_add_reads with the read store: the batch goes across PCIe once, in one piece, and ONE host pass over the CIGARs runs while it
// is on its way (argument checks of the reference's walk, the blocks the reads touch, the candidate-record slots); descriptors and
// candidate discovery are made on the device.  Nothing of the handle's state changes before the whole batch has been checked.
static int32_t add_reads_store(PiscesHip* h, const PiscesReadBatch* batch)
{
    const int32_t nr = batch->n_reads;
    const int32_t minBQ = h->cfg.min_base_call_quality;
    const size_t n_cig = (size_t)batch->cigar_offset[nr], n_seq = (size_t)batch->seq_offset[nr];
    const StageLayout L = stage_layout((size_t)nr, n_cig, n_seq, batch->directions != nullptr, batch->deletion_directions != nullptr);
    const bool staged = h->staged_total == L.total && h->h_stage && (const uint8_t*)batch->position == h->h_stage + L.off_pos &&
                        batch->bases == h->h_stage + L.off_bases && batch->quals == h->h_stage + L.off_quals;
    h->staged_total = 0;
    for (int32_t i = 0; i < nr; i++)   // (what the upload itself relies on)
        if (batch->cigar_offset[i + 1] < batch->cigar_offset[i] || batch->seq_offset[i + 1] < batch->seq_offset[i])
            return fail(h, PISCES_E_INVALID_ARG, "add_reads: malformed read batch");
    // ---- where the batch goes, and across PCIe in one piece
    std::unique_ptr<HostTimer> prof_r(new HostTimer(h->prof_on ? &h->prof[15] : nullptr));
    const size_t bulk = 2 * n_seq + (batch->directions ? n_seq : 0) + 5 * n_cig;
    StorePlace pl;
    { int32_t rc = store_place_batch(h, bulk, &pl); if (rc) return rc; }
    int32_t rc = staged ? PISCES_OK : stage_reserve(h, L.total, !pl.direct);
    if (rc == PISCES_OK && pl.direct) {
        hipError_t e = pl.seg->blob.reserve(L.total);
        if (e != hipSuccess) rc = fail(h, PISCES_E_DEVICE, std::string("add_reads: ") + hipGetErrorString(e));
    }
    if (rc == PISCES_OK && !pl.direct && staged) {   // (stage_reads reserved the device half with the pinned one)
        hipError_t e = h->stage[h->stage_cur].d.reserve(L.total);
        if (e != hipSuccess) rc = fail(h, PISCES_E_DEVICE, std::string("add_reads: ") + hipGetErrorString(e));
    }
    if (rc) { store_unplace(h, pl); return rc; }
    uint8_t* const d = pl.direct ? pl.seg->blob.p : D_STAGE(h);
    prof_r.reset(new HostTimer(h->prof_on ? &h->prof[16] : nullptr));
    rc = store_upload_batch(h, batch, L, nr, n_cig, n_seq, staged, d);
    prof_r.reset(new HostTimer(h->prof_on ? &h->prof[17] : nullptr));
    const bool find_on_device = !h->h_ref.empty();   // without a reference only the IStateManager half (allele counts) runs
    const bool count_indels = find_on_device && !h->snv_walk;
    h->eqx_in_batch = false;
    std::vector<int32_t>& fslots = h->found_slots_host;
    std::vector<int32_t>& touched = h->touched_keys;
    touched.clear();
    int64_t found_slots = 0, found_pool = 0;
    int32_t max_key = 0, min_position = 0;
    // A large batch: the checks and the bookkeeping run on the device behind the upload (add_fused_kernel) — a host pass over tens of
    // millions of CIGARs is a second of one core, longer than the transfer it used to hide under.  (PISCES_HIP_DEVICE_CHECKS=0 / 1 forces either.)
    const bool checked_on_device = h->device_checks == 1 || (h->device_checks < 0 && nr >= (1 << 16));
    ShapeArgs shape;
    bool shaped = false;
    if (checked_on_device) {
        // a batch that becomes a segment of its own: descriptors, fragments and row codes in the same launch as the checks
        if (rc == PISCES_OK && pl.direct) {
            const StoreBatchArrays A = {(const int32_t*)(d + L.off_pos), d + L.off_flags, (const int32_t*)(d + L.off_coff), d + L.off_cop, (const uint32_t*)(d + L.off_clen),
                                        (const int32_t*)(d + L.off_soff), d + L.off_bases, d + L.off_quals, batch->directions ? d + L.off_dirs : nullptr};
            rc = store_shape_begin(h, pl, A, nr, n_cig, n_seq, &shape);
            shaped = rc == PISCES_OK;
        }
        if (rc == PISCES_OK) rc = store_device_checks(h, d, L, nr, n_cig, n_seq, batch->directions != nullptr, batch->deletion_directions != nullptr, count_indels,
                                                      &found_slots, &found_pool, touched, &max_key, &min_position, nullptr, shaped ? &shape : nullptr);
    } else {
    // ---- the pass over the CIGARs, under the transfer
    auto op_ref = [](uint8_t t) { return t == 'M' || t == 'D' || t == 'N' || t == '=' || t == 'X'; };
    auto op_read = [](uint8_t t) { return t == 'M' || t == 'I' || t == 'S' || t == '=' || t == 'X'; };
    fslots.assign((size_t)nr + 1, 0);
    const int32_t bs = h->cfg.block_size;
    int64_t in_lo = 1, in_hi = 0;   // positions of the block touched last: a read inside it needs no division
    const char* bad = nullptr;
    auto touch = [&](int64_t from, int64_t to) {   // inclusive: GetBlock(position) for every position that receives a count (RegionStateManager.cs:361-383)
        if (to < 1) return;
        if (from < 1) from = 1;
        if (from >= in_lo && to <= in_hi) return;
        const int32_t k0 = block_key(h, (int32_t)from), k1 = block_key(h, (int32_t)to);
        for (int32_t k = k0; k <= k1; k++) {
            if (touched.empty() || touched.back() != k) touched.push_back(k);
            max_key = std::max(max_key, k);
        }
        in_lo = (int64_t)(k1 - 1) * bs + 1;
        in_hi = (int64_t)k1 * bs;
    };
    // Nearly every batch is made of reads that are ONE aligned run spanning the read (<n>M): that is a property of the batch's arrays as they
    // lie — operation i is read i's, its length the read's — and is established in two loops over them that vectorise, where a view per
    // read (an out-of-line call, a dozen loads, the general walk's branches) cost 5.8 ns a read: 20 of the 32 us the host spent in the add of
    // one block's 3 500 staged reads, which is what bounds the per-block protocol.  A batch that fails any of it takes the general pass below,
    // which finds the read and the reason.
    bool plain_batch = false;
    if (!batch->directions && nr > 0 && batch->cigar_offset[0] >= 0 && batch->seq_offset[0] >= 0 && n_cig - (size_t)batch->cigar_offset[0] == (size_t)nr) {
        const int32_t* const co = batch->cigar_offset;
        const int32_t* const so = batch->seq_offset;
        const int32_t* const pos = batch->position;
        uint32_t odd = 0;
        for (int32_t i = 0; i < nr; i++) odd |= (uint32_t)((co[i + 1] - co[i]) ^ 1);
        if (odd == 0) {
            const uint8_t* const op = batch->cigar_op + co[0];
            const uint32_t* const ln = batch->cigar_len + co[0];
            for (int32_t i = 0; i < nr; i++) {
                const uint32_t read_len = (uint32_t)(so[i + 1] - so[i]);
                odd |= (uint32_t)(op[i] ^ (uint8_t)'M') | (ln[i] ^ read_len) | (uint32_t)(pos[i] <= 0) | (uint32_t)((int64_t)pos[i] + (int64_t)ln[i] > 0x7FFFFFFFll);
            }
            if (odd == 0) {
                plain_batch = true;   // (no candidate-record slots, no X / =: fslots stays zero)
                for (int32_t i = 0; i < nr; i++)
                    if (ln[i] > 0) touch(pos[i], (int64_t)pos[i] + (int64_t)ln[i] - 1);
            }
        }
    }
    for (int32_t i = 0; i < nr && rc == PISCES_OK && !bad && !plain_batch; i++) {
        const ReadView r = read_view(batch, i);
        fslots[(size_t)i] = (int32_t)found_slots;
        if (r.position <= 0) { bad = "Position must be greater than 0."; break; }
        if (r.read_len < 0 || r.n_cigar < 0) { bad = "add_reads: malformed read batch"; break; }
        if (r.dirs)
            for (int k = 0; k < r.read_len; k++)
                if (r.dirs[k] > 2) { bad = "add_reads: CIGAR does not match the read"; break; }
        if (r.n_cigar == 1 && (r.cigar_op[0] == 'M' || r.cigar_op[0] == '=' || r.cigar_op[0] == 'X')) {   // one aligned run: most reads
            const int64_t len = r.cigar_len[0];
            if (len != r.read_len) { bad = "add_reads: CIGAR does not match the read"; break; }   // Read.ValidateCigar (Read.cs:603-605)
            if ((int64_t)r.position + len > 0x7FFFFFFFll) { bad = "add_reads: read runs past position 2^31 - 1"; break; }
            if (len > 0) touch(r.position, r.position + len - 1);
            if (count_indels && r.cigar_op[0] != 'M') h->eqx_in_batch = true;
            continue;
        }
        auto delq = [&](int idx) {   // CandidateVariantFinder.CheckDeletionQuality (CandidateVariantFinder.cs:294-320)
            if (r.read_len == 0) return false;
            const int after = idx < r.read_len ? r.quals[idx] : r.quals[idx - 1];
            const int before = idx > 0 ? r.quals[idx - 1] : after;
            return before >= minBQ && after >= minBQ;
        };
        int64_t read_span = 0, ref_span = 0;
        for (int c = 0; c < r.n_cigar; c++) {
            const uint8_t t = r.cigar_op[c];
            if (op_read(t)) read_span += r.cigar_len[c];
            if (op_ref(t)) ref_span += r.cigar_len[c];
        }
        if (r.n_cigar > 0 && read_span != r.read_len) { bad = "add_reads: CIGAR does not match the read"; break; }   // Read.ValidateCigar (Read.cs:603-605)
        if ((int64_t)r.position + ref_span > 0x7FFFFFFFll) { bad = "add_reads: read runs past position 2^31 - 1"; break; }
        int64_t rp = r.position, last_mapped = (int64_t)r.position - 1;
        int ri = 0;
        for (int c = 0; c < r.n_cigar && !bad; c++) {
            const uint8_t t = r.cigar_op[c];
            const int64_t len = r.cigar_len[c];
            if (r.del_dirs && t == 'D')
                for (int k = 0; k < 2; k++)
                    if (r.del_dirs[2 * c + k] > 2 && r.del_dirs[2 * c + k] != PISCES_DIR_UNTRACKED) bad = "add_reads: deletion_directions holds a value that is no DirectionType";
            if (count_indels) {
                if (t == 'I' || t == 'D') found_slots++;
                if (t == 'I' && len > (int64_t)kFoundInline) found_pool += len;
                if (t == 'X' || t == '=') h->eqx_in_batch = true;
            }
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
        if (found_slots > 0x7FFFFFF0ll || found_pool > 0x7FFFFFF0ll) bad = "add_reads: too many insertions / deletions in one batch";
    }
    fslots[(size_t)nr] = (int32_t)found_slots;
    if (rc == PISCES_OK && bad) rc = fail(h, PISCES_E_INVALID_ARG, bad);
    }   // (the host's pass over the CIGARs)
    if (!checked_on_device && nr > 0) {
        min_position = batch->position[0];
        for (int32_t i = 1; i < nr; i++) min_position = std::min(min_position, batch->position[i]);
    }
    prof_r.reset(new HostTimer(h->prof_on ? &h->prof[18] : nullptr));
    return store_finish_add(h, pl, rc, d, L, nr, n_cig, n_seq, batch->directions != nullptr, batch->deletion_directions != nullptr, find_on_device, found_slots,
                            found_pool, checked_on_device ? nullptr : fslots.data(), touched, max_key, min_position, shaped ? &shape : nullptr);
}

// pisces_hip_add_reads for a batch in device memory: its arrays are copied into the segment's blob (or, for a small batch, the staging
// buffer's device half) in the layout an uploaded batch has, and everything else is add_reads_store's with the checks made on the device.
int32_t pisces_hip_add_device_reads(PiscesHip* h, const PiscesReadBatch* batch, int64_t n_cigar_ops, int64_t n_bases)
{
    return abi_guard<int32_t>(h, [&]() -> int32_t {
    if (!h) return PISCES_E_INVALID_ARG;
    HostTimer timer(&h->host_time[0]);
    if (!batch || batch->n_reads < 0 || n_cigar_ops < 0 || n_bases < 0 || n_cigar_ops > 0x7FFFFFF0ll || n_bases > 0x7FFFFFF0ll)
        return fail(h, PISCES_E_INVALID_ARG, "add_device_reads: malformed read batch");
    { int32_t rcp = refuse_while_batch_is_open(h, "add_device_reads"); if (rcp) return rcp; }
    const int32_t nr = batch->n_reads;
    if (nr == 0) return PISCES_OK;
    if (!batch->position || !batch->flags || !batch->cigar_offset || !batch->cigar_op || !batch->cigar_len || !batch->seq_offset || !batch->bases || !batch->quals)
        return fail(h, PISCES_E_INVALID_ARG, "add_device_reads: malformed read batch");
    if (h->read_path != 1) return fail(h, PISCES_E_UNSUPPORTED, "add_device_reads: the observation-log chain (PISCES_HIP_READ_PATH=log) takes host batches only");
    PISCES_HIP_CHECK(h, hipSetDevice(h->device));
    { HostTimer prof_c(h->prof_on ? &h->prof[12] : nullptr); int32_t rcf = consume_found(h); if (rcf) return rcf; }
    std::unique_ptr<HostTimer> prof_a(new HostTimer(h->prof_on ? &h->prof[13] : nullptr));
    const size_t n_cig = (size_t)n_cigar_ops, n_seq = (size_t)n_bases;
    const bool has_dirs = batch->directions != nullptr, has_deldirs = batch->deletion_directions != nullptr;
    const StageLayout L = stage_layout((size_t)nr, n_cig, n_seq, has_dirs, has_deldirs);
    h->staged_total = 0;
    StorePlace pl;
    { int32_t rc = store_place_batch(h, 2 * n_seq + (has_dirs ? n_seq : 0) + 5 * n_cig, &pl); if (rc) return rc; }
    int32_t rc = stage_reserve(h, pl.direct ? 64 : L.total, !pl.direct);   // (the pair's event orders the reuse of its device half; a segment's blob needs none of it)
    if (rc == PISCES_OK && pl.direct) {
        hipError_t e = pl.seg->blob.reserve(L.total);
        if (e != hipSuccess) rc = fail(h, PISCES_E_DEVICE, std::string("add_device_reads: ") + hipGetErrorString(e));
    }
    if (rc) { store_unplace(h, pl); return rc; }
    uint8_t* const d = pl.direct ? pl.seg->blob.p : D_STAGE(h);
    const bool find_on_device = !h->h_ref.empty();
    const bool count_indels = find_on_device && !h->snv_walk;
    h->eqx_in_batch = false;
    std::vector<int32_t>& touched = h->touched_keys;
    touched.clear();
    int64_t found_slots = 0, found_pool = 0;
    int32_t max_key = 0, min_position = 0;
    // ONE launch: the caller's arrays into the store's layout, the checks, the candidate-record slots and — a batch that becomes a segment of
    // its own — descriptors, fragments and row codes, the batch read once (add_fused_kernel)
    ShapeArgs shape;
    bool shaped = false;
    if (rc == PISCES_OK && pl.direct) {
        const StoreBatchArrays A = {(const int32_t*)(d + L.off_pos), d + L.off_flags, (const int32_t*)(d + L.off_coff), d + L.off_cop, (const uint32_t*)(d + L.off_clen),
                                    (const int32_t*)(d + L.off_soff), d + L.off_bases, d + L.off_quals, has_dirs ? d + L.off_dirs : nullptr};
        rc = store_shape_begin(h, pl, A, nr, n_cig, n_seq, &shape);
        shaped = rc == PISCES_OK;
    }
    if (rc == PISCES_OK) rc = store_device_checks(h, d, L, nr, n_cig, n_seq, has_dirs, has_deldirs, count_indels, &found_slots, &found_pool, touched, &max_key, &min_position,
                                                  batch, shaped ? &shape : nullptr);
    prof_a.reset(new HostTimer(h->prof_on ? &h->prof[14] : nullptr));
    return store_finish_add(h, pl, rc, d, L, nr, n_cig, n_seq, has_dirs, has_deldirs, find_on_device, found_slots, found_pool, nullptr, touched, max_key, min_position,
                            shaped ? &shape : nullptr);
    });
}

// pisces_hip_add_decoded_reads with the read store: a large decoded batch becomes a segment as it lies (its arrays change owner: the
// decode's next batch gets the buffers of a retired segment), a small one joins the open segment.
static int32_t add_decoded_reads_store(PiscesHip* h, int64_t found_slots, int64_t found_pool, bool find_on_device)
{
    auto& B = h->bam;
    const int32_t nr = (int32_t)B.n_reads;
    const size_t n_cig = (size_t)B.n_ops, n_seq = (size_t)B.n_bases;
    int32_t max_key = 0;
    for (size_t w = B.block_map.size(); w-- > 0;)
        if (B.block_map[w]) { max_key = (int32_t)(w * 32 + (31 - (size_t)__builtin_clz(B.block_map[w]))) + 1; break; }
    StorePlace pl;
    { int32_t rc = store_place_batch(h, 2 * n_seq + 5 * n_cig, &pl); if (rc) return rc; }
    ReadSegment& g = *pl.seg;
    if (pl.direct) {
        g.bases.swap(B.bases);
        g.quals.swap(B.quals);
        g.cop.swap(B.cigar_op);
        g.clen.swap(B.cigar_len);
        if (B.has_dirs) g.dirs.swap(B.dirs);
        B.moved = true;
    }
    // (the decode leaves kSegmentPad bytes in front of the bases and the qualities)
    const StoreBatchArrays A = {B.position.p, B.flags.p, B.cigar_offset.p, pl.direct ? g.cop.p : B.cigar_op.p, pl.direct ? g.clen.p : B.cigar_len.p,
                                B.seq_offset.p, (pl.direct ? g.bases.p : B.bases.p) + kSegmentPad, (pl.direct ? g.quals.p : B.quals.p) + kSegmentPad,
                                B.has_dirs ? (pl.direct ? g.dirs.p : B.dirs.p) + kSegmentPad : nullptr};
    int32_t rc = store_append_arrays(h, pl, A, nr, n_cig, n_seq);
    if (rc == PISCES_OK && find_on_device && (h->snv_walk || found_slots > 0 || h->eqx_in_batch)) {
        DevReadBatch db;
        db.position = A.position; db.flags = A.flags; db.cigar_offset = A.cigar_offset; db.cigar_op = A.cigar_op; db.cigar_len = A.cigar_len;
        db.seq_offset = A.seq_offset; db.bases = A.bases; db.quals = A.quals; db.dirs = A.dirs; db.n_reads = nr;
        rc = enqueue_candidate_discovery(h, db, B.has_dirs ? B.del_dirs.p : nullptr, nr, (const int32_t*)B.d_fslots.p, found_slots, found_pool);
    }
    if (rc) {
        (void)hipStreamSynchronize(h->stream);
        if (pl.direct) { g.bases.swap(B.bases); g.quals.swap(B.quals); g.cop.swap(B.cigar_op); g.clen.swap(B.cigar_len); if (B.has_dirs) g.dirs.swap(B.dirs); B.moved = false; }
        store_unplace(h, pl);
        return rc;
    }
    g.n_reads += nr;
    g.n_bases += (int64_t)n_seq;
    g.n_ops += (int64_t)n_cig;
    g.max_key = std::max(g.max_key, max_key);
    store_maybe_seal(h, &g);
    return PISCES_OK;
}

// the flush's kernel: reads of the store (+ the bucketed tuples of pisces_hip_add_observations) -> LDS histogram -> calls
static hipError_t launch_call_store_tiles(PiscesHip* h, hipStream_t s, const uint32_t* d_tuples, const PiscesTile* d_tiles /* or nullptr: R */,
                                          const RegularTiles& R, int32_t n_tiles, const uint8_t* d_ref, int32_t ref_start, int64_t ref_len, PiscesCalledAllele* d_records, PiscesTileResult* d_tr,
                                          hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr)
{
    StoreView V;
    store_view(h, &V);
    // waves per tile: enough of them that a small launch still puts its reads on many SIMDs (see the kernel)
    int nw = h->store_waves;
    if (nw == 0) nw = h->kernel_variant == 2 ? 1 : h->kernel_variant == 3 ? 2 : (int64_t)n_tiles <= (int64_t)h->n_cus ? 16 /* (a tile a CU at most: all sixteen waves; 235 tiles at 5000x: 69 us against 74 with eight) */
                      : (int64_t)n_tiles * 4 <= (int64_t)h->n_cus * 12 ? 4 : (int64_t)n_tiles <= (int64_t)h->n_cus * 32 ? 2 : 1;
    // several tiles a CU (store_kernels.hip.h): the workgroups trade tiles by price inside small groups (the default), or take them in
    // tile_order_kernel's order (PISCES_HIP_TILE_ORDER=1: a launch in front), or in position order (=0)
    const int32_t* order = nullptr;
    int32_t trade_cus = 0;
    // (from 4 to 8 tiles a CU: beyond, the CUs' shares even out by themselves and the trade's pricing only delays a tile's start —
    // 3 200 / 4 688 / 9 376 tiles: 60.3 / 48.3 / 92.3 us in position order, 61.0 / 49.2 / 93.0 traded, profiles/r05_tile_order.txt)
    if (h->tile_order == 2 && nw <= 2 && (int64_t)n_tiles >= 4 * (int64_t)h->n_cus && (int64_t)n_tiles <= 8 * (int64_t)h->n_cus && h->n_cus >= 8)
        trade_cus = (int32_t)(h->n_cus / 8);
    if (h->tile_order == 1 && nw <= 2 && (int64_t)n_tiles >= 4 * (int64_t)h->n_cus && h->n_cus >= 8) {
        if (h->d_tile_order.reserve((size_t)n_tiles) != hipSuccess) return hipErrorOutOfMemory;
        hipLaunchKernelGGL(tile_order_kernel, dim3(8), dim3(256), 0, s, V, d_tiles, R, n_tiles, (int32_t)(h->n_cus / 8), h->d_tile_order.p);
        order = h->d_tile_order.p;
    }
    // issue priority for walking waves: launches whose workgroups are all resident at once (8 tiles a CU at two waves a tile)
    const int32_t walk_prio = (h->store_prio && nw == 2 && (int64_t)n_tiles <= 8 * (int64_t)h->n_cus) ? 1 : 0;
#define PISCES_LAUNCH_STORE(NW)                                                                                                                     \
    hipExtLaunchKernelGGL(call_store_tiles_kernel<NW>, dim3((unsigned)n_tiles), dim3(64 * NW), 0u, s, e0, e1, 0u, V, d_tuples, d_tiles, R, n_tiles, order, trade_cus, walk_prio, d_ref, \
                          ref_start, ref_len, d_records, d_tr, h->P, (const DeviceParams*)h->d_params.p)
    if (nw >= 16) PISCES_LAUNCH_STORE(16);
    else if (nw >= 8) PISCES_LAUNCH_STORE(8);
    else if (nw >= 4) PISCES_LAUNCH_STORE(4);
    else if (nw >= 2) PISCES_LAUNCH_STORE(2);
    else PISCES_LAUNCH_STORE(1);
#undef PISCES_LAUNCH_STORE
    return hipGetLastError();
}
