// surface_device.inc.h — part of pisces_hip.hip (included there, inside its extern "C" block; not a translation unit of its own).
// The device-resident surface: pisces_hip_call_tiles[_batched], compaction, accumulation into a caller's tensor, totals, timing, the
// streaming-read probe.

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
    PISCES_HIP_CHECK(h, hipSetDevice(h->device));
    hipStream_t s = stream ? (hipStream_t)stream : h->stream;
    if (s != h->stream) h->foreign_stream_used = h->foreign_stream_ever = true;
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
        if (germline(h)) launch_genotype_loci(h, s, d_records, d_tile_results, n_tiles);   // PloidyModel.DiploidByThresholding / Haploid: a pass over the slots
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
        if (germline(h)) launch_genotype_loci(h, h->lane[i % lanes], b.d_records, b.d_tile_results, b.n_tiles);
    }
    PISCES_HIP_CHECK(h, hipGetLastError());
    return PISCES_OK;
    });
}

// The launches of pisces_hip_call_tiles as one HIP graph (stream capture of the very launch path, so the graph holds what a plain call
// would have launched: kernel form and geometry chosen by launch_call_tiles).
int32_t pisces_hip_call_tiles_graph_build(PiscesHip* h, const PiscesTileBatch* batches, int32_t n_batches, int32_t* graph_id)
{
    return abi_guard<int32_t>(h, [&]() -> int32_t {
    if (!h || !graph_id) return PISCES_E_INVALID_ARG;
    if (n_batches <= 0 || !batches) return fail(h, PISCES_E_INVALID_ARG, "call_tiles_graph_build: null batch list");
    if (h->cfg.noise_model == PISCES_NOISE_WINDOW)
        return fail(h, PISCES_E_STATE, "call_tiles_graph_build: NoiseModel.Flat only (see pisces_hip_call_tiles_batched)");
    for (int32_t i = 0; i < n_batches; i++) {
        const PiscesTileBatch& b = batches[i];
        if (b.n_tiles <= 0 || b.record_capacity < 0 || b.ref_length < 0 || !b.d_tiles || !b.d_ref_bases || !b.d_records || !b.d_tile_results)
            return fail(h, PISCES_E_INVALID_ARG, "call_tiles_graph_build: bad batch");
        if ((int64_t)b.record_capacity < (int64_t)b.n_tiles * kSlotsPerTile)
            return fail(h, PISCES_E_BUFFER_TOO_SMALL, "call_tiles: the slot layout needs record_capacity >= 256 * n_tiles");
    }
    PISCES_HIP_CHECK(h, hipSetDevice(h->device));
    PISCES_HIP_CHECK(h, hipStreamSynchronize(h->stream));
    hipGraph_t graph = nullptr;
    PISCES_HIP_CHECK(h, hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
    hipError_t e = hipSuccess;
    for (int32_t i = 0; i < n_batches && e == hipSuccess; i++) {
        const PiscesTileBatch& b = batches[i];
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (h->timing > 0 && (h->launches_seen++ % h->timing) == 0 && h->ring_used < kTimingRing) {
            const size_t slot = (size_t)h->ring_used;
            e0 = h->ring[2 * slot];
            e1 = h->ring[2 * slot + 1];
            h->ring_used++;
        }
        // (event records of a capture become event-record nodes: the stamps are taken when the replay reaches them)
        if (e0) e = hipEventRecord(e0, h->stream);
        if (e == hipSuccess)
            e = launch_call_tiles(h, h->stream, b.d_tuples, b.d_tiles, b.n_tiles, b.d_ref_bases, b.ref_start_position, b.ref_length, b.d_records,
                                  b.d_tile_results);
        if (e == hipSuccess && germline(h)) launch_genotype_loci(h, h->stream, b.d_records, b.d_tile_results, b.n_tiles);
        if (e == hipSuccess && e1) e = hipEventRecord(e1, h->stream);
    }
    const hipError_t ec = hipStreamEndCapture(h->stream, &graph);
    if (e == hipSuccess) e = ec;
    if (e == hipSuccess) e = hipGetLastError();
    hipGraphExec_t exec = nullptr;
    if (e == hipSuccess) e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    if (e != hipSuccess) {
        if (graph) (void)hipGraphDestroy(graph);
        return fail(h, PISCES_E_DEVICE, std::string("call_tiles_graph_build: ") + hipGetErrorString(e));
    }
    // (the executable graph's own upload now, not inside its first launch: a replay then costs what every later replay costs)
    if (hipGraphUpload(exec, h->stream) == hipSuccess) (void)hipStreamSynchronize(h->stream);
    else (void)hipGetLastError();
    h->graph_defs.push_back(graph);
    h->graphs.push_back(exec);
    *graph_id = (int32_t)h->graphs.size() - 1;
    return PISCES_OK;
    });
}

int32_t pisces_hip_call_tiles_graph_launch(PiscesHip* h, int32_t graph_id, void* stream)
{
    return abi_guard<int32_t>(h, [&]() -> int32_t {
    if (!h) return PISCES_E_INVALID_ARG;
    if (graph_id < 0 || (size_t)graph_id >= h->graphs.size()) return fail(h, PISCES_E_INVALID_ARG, "call_tiles_graph_launch: no such graph");
    PISCES_HIP_CHECK(h, hipSetDevice(h->device));
    if (stream && (hipStream_t)stream != h->stream) h->foreign_stream_used = h->foreign_stream_ever = true;
    PISCES_HIP_CHECK(h, hipGraphLaunch(h->graphs[(size_t)graph_id], stream ? (hipStream_t)stream : h->stream));
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
    if (s != h->stream) h->foreign_stream_used = h->foreign_stream_ever = true;
    if (n_tiles == 0) {
        PISCES_HIP_CHECK(h, hipMemsetAsync(d_count, 0, sizeof(int32_t), s));
        return PISCES_OK;
    }
    { int32_t rcc = launch_compaction(h, s, d_records, d_tile_results, n_tiles, d_offsets, d_out, out_capacity, d_count); if (rcc) return rcc; }
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
    if (s != h->stream) h->foreign_stream_used = h->foreign_stream_ever = true;
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
    // Launches may sit on the handle's lanes or on a caller's stream: those are waited for first (a device-wide wait only when a stream
    // that is not the handle's was used since the last call).  The totals then come back as totals_collect_kernel's own stores into
    // pinned host memory behind everything on the handle's stream: one stream wait, no copy operation (round 4: hipDeviceSynchronize +
    // hipMemcpy + hipMemset, ~75 us of fixed latency around bench.py's timed region).
    if (h->foreign_stream_used) {
        PISCES_HIP_CHECK(h, hipDeviceSynchronize());
        h->foreign_stream_used = false;
    } else {
        for (int k = 0; k < PiscesHip::kLanes; k++)
            if (h->lane[k]) PISCES_HIP_CHECK(h, hipStreamSynchronize(h->lane[k]));
    }
    if (!h->h_totals) PISCES_HIP_CHECK(h, host_alloc((void**)&h->h_totals, 64));
    hipLaunchKernelGGL(totals_collect_kernel, dim3(1), dim3(64), 0, h->stream, h->d_totals.p, h->h_totals, reset ? 1 : 0);
    PISCES_HIP_CHECK(h, hipGetLastError());
    PISCES_HIP_CHECK(h, hipStreamSynchronize(h->stream));
    for (int i = 0; i < 4; i++) out[i] = (int64_t)((volatile unsigned long long*)h->h_totals)[i];
    return PISCES_OK;
    });
}

int32_t pisces_hip_mark(PiscesHip* h, int32_t which, void* stream)
{
    return abi_guard<int32_t>(h, [&]() -> int32_t {
    if (!h || which < 0 || which > 1) return PISCES_E_INVALID_ARG;
    PISCES_HIP_CHECK(h, hipSetDevice(h->device));
    PISCES_HIP_CHECK(h, hipEventRecord(which ? h->ev1 : h->ev0, stream ? (hipStream_t)stream : h->stream));
    return PISCES_OK;
    });
}

int32_t pisces_hip_marked_ms(PiscesHip* h, float* ms)
{
    return abi_guard<int32_t>(h, [&]() -> int32_t {
    if (!h || !ms) return PISCES_E_INVALID_ARG;
    PISCES_HIP_CHECK(h, hipSetDevice(h->device));
    PISCES_HIP_CHECK(h, hipEventSynchronize(h->ev1));
    PISCES_HIP_CHECK(h, hipEventElapsedTime(ms, h->ev0, h->ev1));
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

int32_t pisces_hip_set_chain_timing(PiscesHip* h, int32_t enable)
{
    return abi_guard<int32_t>(h, [&]() -> int32_t {
    if (!h) return PISCES_E_INVALID_ARG;
    PISCES_HIP_CHECK(h, hipSetDevice(h->device));
    if (enable)
        for (auto& ev : h->ev_chain)
            if (!ev) PISCES_HIP_CHECK(h, hipEventCreate(&ev));
    h->chain_timing = enable != 0;
    h->chain_have[0] = h->chain_have[1] = false;
    return PISCES_OK;
    });
}

int32_t pisces_hip_chain_time(PiscesHip* h, double out_ms[2])
{
    return abi_guard<int32_t>(h, [&]() -> int32_t {
    if (!h || !out_ms) return PISCES_E_INVALID_ARG;
    if (!h->chain_timing || !h->chain_have[0] || !h->chain_have[1])
        return fail(h, PISCES_E_STATE, "chain_time: no timed add_device_reads + flush pair (pisces_hip_set_chain_timing first)");
    PISCES_HIP_CHECK(h, hipSetDevice(h->device));
    for (int k = 0; k < 2; k++) {
        float ms = 0.f;
        PISCES_HIP_CHECK(h, hipEventSynchronize(h->ev_chain[2 * k + 1]));
        PISCES_HIP_CHECK(h, hipEventElapsedTime(&ms, h->ev_chain[2 * k], h->ev_chain[2 * k + 1]));
        out_ms[k] = ms;
    }
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
    {   // calibration of the FETCH_SIZE counter on the read store kernel's load width (tools/fetch_calibration.py): dword loads instead
        const char* dw = std::getenv("PISCES_HIP_PROBE_DWORD");
        if (dw && dw[0] == '1') {
            double best = 0.0;
            for (int r = 0; r <= reps; r++) {
                hipExtLaunchKernelGGL(read_probe_dword_kernel, dim3((unsigned)h->n_cus * 32), dim3(256), 0u, h->stream, h->ev0, h->ev1, 0u, (const uint32_t*)buf.p,
                                      nbytes / 4, buf.p + nbytes / 4);
                PISCES_HIP_CHECK(h, hipEventSynchronize(h->ev1));
                float ms = 0.f;
                PISCES_HIP_CHECK(h, hipEventElapsedTime(&ms, h->ev0, h->ev1));
                if (r > 0 && ms > 0.f) best = std::max(best, (double)nbytes / ((double)ms * 1e-3) / 1e9);
            }
            buf.release();
            *gb_per_s = best;
            return PISCES_OK;
        }
    }
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

