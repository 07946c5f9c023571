// surface_bam.inc.h — part of pisces_hip.hip (included there, inside its extern "C" block; not a translation unit of its own).
// Row f4 behind the C ABI: BGZF block table and inflate, BAM bytes -> device-resident read batch, pisces_hip_add_decoded_reads.

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
    PISCES_HIP_CHECK(h, d_in.reserve((size_t)n_bytes + kInWindow + 256));   // the bit reader's LDS window is filled in whole: up to a window past a block's payload
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
    { int32_t rcd = finish_candidate_discovery(h); if (rcd) return rcd; }   // (the last batch's walk may still read the arrays this decode writes)
    h->bam.valid = false;
    h->bam.moved = false;
    h->bam.added = false;
    int64_t out_bytes = 0;
    for (int64_t i = 0; i < n_blocks; i++) {
        const PiscesBgzfBlock& b = blocks[i];
        if (b.in_offset < 0 || b.in_length < 0 || b.in_length > 65536 || b.in_offset + b.in_length > n_bytes || b.out_offset < 0 ||
            b.out_length < 0 || b.out_length > 65536)
            return fail(h, PISCES_E_INVALID_ARG, "bam_decode: block " + std::to_string(i) + " lies outside the file bytes");
        // the inflated stream is parsed as ONE run of BAM records: the blocks must tile it in order (what pisces_hip_bgzf_scan makes);
        // a gap would be parsed as records, overlapping blocks would race in the inflate
        if (b.out_offset != out_bytes)
            return fail(h, PISCES_E_INVALID_ARG, "bam_decode: block " + std::to_string(i) + " does not start where the block before it ends in the inflated stream");
        out_bytes = b.out_offset + b.out_length;
    }
    if (out_bytes <= 0 || out_bytes > 0x7FFFFFFF00ll) return fail(h, PISCES_E_INVALID_ARG, "bam_decode: empty or oversized stream");
    PISCES_HIP_CHECK(h, hipSetDevice(h->device));
    auto& B = h->bam;
    PISCES_HIP_CHECK(h, B.d_file.reserve((size_t)n_bytes + kInWindow + 256));
    PISCES_HIP_CHECK(h, B.d_stream.reserve((size_t)out_bytes + 16));
    PISCES_HIP_CHECK(h, B.d_blocks.reserve((size_t)n_blocks));
    PISCES_HIP_CHECK(h, B.d_status.reserve((size_t)n_blocks));
    PISCES_HIP_CHECK(h, hipMemsetAsync(B.d_file.p + n_bytes, 0, kInWindow + 256, h->stream));   // (the bit reader's window is filled in whole)
    PISCES_HIP_CHECK(h, hipMemcpyAsync(B.d_blocks.p, blocks, (size_t)n_blocks * sizeof(PiscesBgzfBlock), hipMemcpyHostToDevice, h->stream));
    // (Measured and not kept: the file in slices on a copy stream with every slice's blocks inflating on a stream of their own as the
    // slice arrives, from pageable and from pinned memory — 11.4-11.7 ms per 108 MB either way against 11.4 ms for one transfer in front
    // of one launch: the launches did not overlap the transfers on this runtime, and one launch per slice on ONE stream serialises at
    // 4.5 ms a launch, the time one block takes one wave.)
    PISCES_HIP_CHECK(h, hipMemcpyAsync(B.d_file.p, file, (size_t)n_bytes, hipMemcpyHostToDevice, h->stream));
    h->pcie[0] += n_bytes + n_blocks * (int64_t)sizeof(PiscesBgzfBlock);
    hipLaunchKernelGGL(bgzf_inflate_kernel, dim3((unsigned)n_blocks), dim3(64), 0, h->stream, (const uint8_t*)B.d_file.p,
                       (const PiscesBgzfBlock*)B.d_blocks.p, n_blocks, B.d_stream.p, B.d_status.p);
    // record boundaries without a serial pass over the bytes
    const int64_t n_chunks = (out_bytes + kBamChunk - 1) / kBamChunk;
    PISCES_HIP_CHECK(h, B.d_exits.reserve((size_t)n_chunks * kBamGuessWindow));
    PISCES_HIP_CHECK(h, B.d_header.reserve(4));
    PISCES_HIP_CHECK(h, B.d_entry.reserve((size_t)n_chunks));
    PISCES_HIP_CHECK(h, B.d_shared_exit.reserve((size_t)n_chunks));
    PISCES_HIP_CHECK(h, B.d_bstatus.reserve(4));
    PISCES_HIP_CHECK(h, B.d_n_reads.reserve((size_t)n_chunks + 1));
    PISCES_HIP_CHECK(h, B.d_n_ops.reserve((size_t)n_chunks + 1));
    PISCES_HIP_CHECK(h, B.d_n_bases.reserve((size_t)n_chunks + 1));
    PISCES_HIP_CHECK(h, B.d_n_skipped.reserve((size_t)n_chunks + 1));
    PISCES_HIP_CHECK(h, B.d_n_span.reserve((size_t)n_chunks + 1));
    PISCES_HIP_CHECK(h, B.d_n_indels.reserve((size_t)n_chunks + 1));
    PISCES_HIP_CHECK(h, B.d_n_pool.reserve((size_t)n_chunks + 1));
    PISCES_HIP_CHECK(h, B.d_first_error.reserve(1));
    PISCES_HIP_CHECK(h, B.d_totals64.reserve(8));
    PISCES_HIP_CHECK(h, hipMemsetAsync(B.d_bstatus.p, 0, 4 * sizeof(int32_t), h->stream));
    {   // diagnostics: PISCES_HIP_BAM_SERIAL_CHAIN=1 takes the serial hop whatever the guesses say (the two must agree: tests)
        const char* force = std::getenv("PISCES_HIP_BAM_SERIAL_CHAIN");
        if (force && force[0] == '1') PISCES_HIP_CHECK(h, hipMemsetAsync(B.d_bstatus.p + 3, 1, 1, h->stream));
    }
    const BamFilter F = {ref_id, min_map_quality, skip_duplicates, only_proper_pairs, h->cfg.min_base_call_quality, h->cfg.block_size};
    hipLaunchKernelGGL(bam_header_kernel, dim3(1), dim3(1), 0, h->stream, (const uint8_t*)B.d_stream.p, out_bytes, B.d_header.p, ref_id);
    hipLaunchKernelGGL(bam_chain_kernel, dim3((unsigned)n_chunks), dim3(256), 0, h->stream, (const uint8_t*)B.d_stream.p, out_bytes, B.d_exits.p,
                       B.d_shared_exit.p, (const long long*)B.d_header.p);
    // (d_bstatus[3]: some chunk needs the serial hop)
    hipLaunchKernelGGL(bam_entry_guess_kernel, dim3((unsigned)((n_chunks + 255) / 256)), dim3(256), 0, h->stream, (const uint32_t*)B.d_shared_exit.p,
                       (const uint16_t*)B.d_exits.p, out_bytes, (const long long*)B.d_header.p, n_chunks, B.d_entry.p, B.d_bstatus.p, B.d_bstatus.p + 3);
    hipLaunchKernelGGL(bam_entry_check_kernel, dim3((unsigned)((n_chunks + 255) / 256)), dim3(256), 0, h->stream, (const uint16_t*)B.d_exits.p,
                       out_bytes, (const long long*)B.d_header.p, n_chunks, (const long long*)B.d_entry.p, B.d_bstatus.p + 3);
    hipLaunchKernelGGL(bam_entry_kernel, dim3(1), dim3(1), 0, h->stream, (const uint8_t*)B.d_stream.p, (const uint16_t*)B.d_exits.p, out_bytes, (const long long*)B.d_header.p,
                       n_chunks, B.d_entry.p, B.d_bstatus.p, (const int32_t*)(B.d_bstatus.p + 3));
    hipLaunchKernelGGL(bam_count_kernel, dim3((unsigned)n_chunks), dim3(64), 0, h->stream, (const uint8_t*)B.d_stream.p, out_bytes,
                       (const long long*)B.d_entry.p, F, B.d_n_reads.p, B.d_n_ops.p, B.d_n_bases.p, B.d_n_skipped.p, B.d_n_span.p, B.d_n_indels.p,
                       B.d_n_pool.p, B.d_bstatus.p);
    hipLaunchKernelGGL(bam_scan3_kernel, dim3(1), dim3(1024), 0, h->stream, B.d_n_reads.p, B.d_n_ops.p, B.d_n_bases.p, (int32_t)n_chunks, B.d_totals64.p);
    hipLaunchKernelGGL(bam_scan3_kernel, dim3(1), dim3(1024), 0, h->stream, B.d_n_indels.p, B.d_n_pool.p, (int32_t*)nullptr, (int32_t)n_chunks, B.d_totals64.p + 3);
    hipLaunchKernelGGL(bam_scan_ll_kernel, dim3(1), dim3(1024), 0, h->stream, B.d_n_span.p, (int32_t)n_chunks);
    PISCES_HIP_CHECK(h, hipGetLastError());
    std::vector<int32_t> status((size_t)n_blocks), skipped((size_t)n_chunks);
    int32_t totals[5] = {0, 0, 0, 0, 0}, bstatus[4] = {0, 0, 0, 0};
    long long span_total = 0, header[4] = {0, 0, 0, 0}, totals64[6] = {0, 0, 0, 0, 0, 0};
    PISCES_HIP_CHECK(h, hipMemcpyAsync(status.data(), B.d_status.p, status.size() * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
    PISCES_HIP_CHECK(h, hipMemcpyAsync(skipped.data(), B.d_n_skipped.p, skipped.size() * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
    PISCES_HIP_CHECK(h, hipMemcpyAsync(&totals[0], B.d_n_reads.p + n_chunks, sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
    PISCES_HIP_CHECK(h, hipMemcpyAsync(&totals[1], B.d_n_ops.p + n_chunks, sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
    PISCES_HIP_CHECK(h, hipMemcpyAsync(&totals[2], B.d_n_bases.p + n_chunks, sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
    PISCES_HIP_CHECK(h, hipMemcpyAsync(&totals[3], B.d_n_indels.p + n_chunks, sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
    PISCES_HIP_CHECK(h, hipMemcpyAsync(&totals[4], B.d_n_pool.p + n_chunks, sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
    PISCES_HIP_CHECK(h, hipMemcpyAsync(&span_total, B.d_n_span.p + n_chunks, sizeof(long long), hipMemcpyDeviceToHost, h->stream));
    PISCES_HIP_CHECK(h, hipMemcpyAsync(header, B.d_header.p, sizeof(header), hipMemcpyDeviceToHost, h->stream));
    PISCES_HIP_CHECK(h, hipMemcpyAsync(bstatus, B.d_bstatus.p, sizeof(bstatus), hipMemcpyDeviceToHost, h->stream));
    PISCES_HIP_CHECK(h, hipMemcpyAsync(totals64, B.d_totals64.p, sizeof(totals64), hipMemcpyDeviceToHost, h->stream));
    PISCES_HIP_CHECK(h, hipStreamSynchronize(h->stream));
    for (int64_t i = 0; i < n_blocks; i++)
        if (status[(size_t)i] != 0)
            return fail(h, PISCES_E_INVALID_ARG, "bam_decode: block " + std::to_string(i) + " is not a valid DEFLATE stream of its ISIZE");
    if (bstatus[0] == 1) return fail(h, PISCES_E_INVALID_ARG, "bam_decode: not a BAM stream (magic / header)");
    if (bstatus[0] == 4)
        return fail(h, PISCES_E_INVALID_ARG, "bam_decode: a record in chunk " + std::to_string(bstatus[1]) + " is shorter than its name, CIGAR and bases");
    if (bstatus[0] != 0)
        return fail(h, PISCES_E_INVALID_ARG, "bam_decode: the record chain breaks in chunk " + std::to_string(bstatus[1]) +
                                                 " (corrupt block_size, or a record longer than 32 KiB)");
    // (the kernels index the batch with 32-bit offsets: a chromosome with more goes through in several regions)
    for (int k = 0; k < 5; k++)
        if (totals64[k] > 0x7FFFFFF0ll || span_total < 0)
            return fail(h, PISCES_E_INVALID_ARG, "bam_decode: more than 2^31 reads, CIGAR operations, bases or candidate slots in one call: decode the file in regions");
    B.chain_mode = bstatus[3] != 0 ? 1 : 0;
    B.has_dirs = (bstatus[2] & 1) != 0;
    B.has_eqx = (bstatus[2] & 2) != 0;
    B.n_reads = totals[0]; B.n_ops = totals[1]; B.n_bases = totals[2];
    B.found_slots = totals[3]; B.found_pool = totals[4]; B.log_slots = span_total;
    B.n_skipped = 0;
    for (int32_t v : skipped) B.n_skipped += v;
    B.min_bq = h->cfg.min_base_call_quality;
    const size_t nr = (size_t)B.n_reads, no = (size_t)B.n_ops, nb = (size_t)B.n_bases;
    PISCES_HIP_CHECK(h, B.position.reserve(nr + 1)); PISCES_HIP_CHECK(h, B.flags.reserve(nr + 1));
    PISCES_HIP_CHECK(h, B.cigar_offset.reserve(nr + 1)); PISCES_HIP_CHECK(h, B.seq_offset.reserve(nr + 1));
    PISCES_HIP_CHECK(h, B.read_quality.reserve(nr + 1));
    PISCES_HIP_CHECK(h, B.cigar_op.reserve(no + 1)); PISCES_HIP_CHECK(h, B.cigar_len.reserve(no + 1)); PISCES_HIP_CHECK(h, B.op_quality.reserve(no + 1));
    // (kSegmentPad bytes in front of the bases and the qualities: a batch that becomes a segment of the read store as it lies)
    PISCES_HIP_CHECK(h, B.bases.reserve(nb + 16 + 2 * kSegmentPad)); PISCES_HIP_CHECK(h, B.quals.reserve(nb + 16 + 2 * kSegmentPad));
    PISCES_HIP_CHECK(h, B.d_slots.reserve(nr + 1)); PISCES_HIP_CHECK(h, B.d_fslots.reserve(nr + 1));
    if (B.has_dirs) {   // some read is stitched (XD tag): per-base directions for all, the directions inside deletions for the finder
        PISCES_HIP_CHECK(h, B.dirs.reserve(nb + 16 + 2 * kSegmentPad));
        PISCES_HIP_CHECK(h, B.del_dirs.reserve(2 * no + 2));
    }
    // the blocks of the chromosome (its length from the header, and room for reads that hang over its end), one bit each
    const long long l_ref = std::min<long long>(std::max<long long>(header[3], 0), 0x7FFFFFFFll);
    const long long n_block_bits = std::min<long long>(l_ref + 70000, 0x7FFFFFFFll - h->cfg.block_size) / h->cfg.block_size + 1;
    const size_t map_words = (size_t)((n_block_bits + 31) / 32);
    PISCES_HIP_CHECK(h, B.d_block_map.reserve(map_words));
    PISCES_HIP_CHECK(h, hipMemsetAsync(B.d_block_map.p, 0, map_words * sizeof(uint32_t), h->stream));
    PISCES_HIP_CHECK(h, hipMemsetAsync(B.d_first_error.p, 0xFF, sizeof(unsigned long long), h->stream));
    if (nr > 0)
        hipLaunchKernelGGL(bam_decode_kernel, dim3((unsigned)n_chunks), dim3(256), 0, h->stream, (const uint8_t*)B.d_stream.p, out_bytes,
                           (const long long*)B.d_entry.p, F, (const int32_t*)B.d_n_reads.p, (const int32_t*)B.d_n_ops.p, (const int32_t*)B.d_n_bases.p,
                           B.position.p, B.flags.p, B.cigar_offset.p, B.cigar_op.p, B.cigar_len.p, B.seq_offset.p, B.bases.p + kSegmentPad, B.quals.p + kSegmentPad,
                           B.op_quality.p, B.read_quality.p, (const long long*)B.d_n_span.p, (const int32_t*)B.d_n_indels.p, B.d_slots.p,
                           B.d_fslots.p, B.d_block_map.p, n_block_bits, B.d_first_error.p, B.has_dirs ? B.dirs.p + kSegmentPad : (uint8_t*)nullptr,
                           B.has_dirs ? B.del_dirs.p : (uint8_t*)nullptr);
    // the closing offsets
    const int32_t end_ops = (int32_t)no, end_bases = (int32_t)nb, end_fslots = (int32_t)B.found_slots;
    const long long end_slots = B.log_slots;
    PISCES_HIP_CHECK(h, hipMemcpyAsync(B.cigar_offset.p + nr, &end_ops, sizeof(int32_t), hipMemcpyHostToDevice, h->stream));
    PISCES_HIP_CHECK(h, hipMemcpyAsync(B.seq_offset.p + nr, &end_bases, sizeof(int32_t), hipMemcpyHostToDevice, h->stream));
    PISCES_HIP_CHECK(h, hipMemcpyAsync(B.d_fslots.p + nr, &end_fslots, sizeof(int32_t), hipMemcpyHostToDevice, h->stream));
    PISCES_HIP_CHECK(h, hipMemcpyAsync(B.d_slots.p + nr, &end_slots, sizeof(long long), hipMemcpyHostToDevice, h->stream));
    B.block_map.assign(map_words, 0u);
    B.first_error = ~0ull;
    PISCES_HIP_CHECK(h, hipMemcpyAsync(B.block_map.data(), B.d_block_map.p, map_words * sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream));
    PISCES_HIP_CHECK(h, hipMemcpyAsync(&B.first_error, B.d_first_error.p, sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream));
    PISCES_HIP_CHECK(h, hipGetLastError());
    PISCES_HIP_CHECK(h, hipStreamSynchronize(h->stream));
    B.valid = true;
    if (counts) { counts[0] = B.n_reads; counts[1] = B.n_skipped; counts[2] = B.n_ops; counts[3] = B.n_bases; }
    return PISCES_OK;
    });
}

int32_t pisces_hip_bam_chain_mode(PiscesHip* h)
{
    return abi_guard<int32_t>(h, [&]() -> int32_t {
    if (!h) return PISCES_E_INVALID_ARG;
    if (!h->bam.valid) return fail(h, PISCES_E_STATE, "bam_chain_mode: no decoded batch (pisces_hip_bam_decode first)");
    return h->bam.chain_mode;
    });
}

int32_t pisces_hip_bam_fetch(PiscesHip* h, int32_t* position, uint8_t* flags, int32_t* cigar_offset, uint8_t* cigar_op, uint32_t* cigar_len,
                             int32_t* seq_offset, uint8_t* bases, uint8_t* quals)
{
    return abi_guard<int32_t>(h, [&]() -> int32_t {
    if (!h) return PISCES_E_INVALID_ARG;
    if (!h->bam.valid) return fail(h, PISCES_E_STATE, "bam_fetch: no decoded batch (pisces_hip_bam_decode first)");
    auto& B = h->bam;
    if (B.moved) return fail(h, PISCES_E_STATE, "bam_fetch: the decoded batch has been added to the read store (fetch before pisces_hip_add_decoded_reads)");
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
    PISCES_HIP_CHECK(h, down(bases, B.bases.p + kSegmentPad, nb));
    PISCES_HIP_CHECK(h, down(quals, B.quals.p + kSegmentPad, nb));
    PISCES_HIP_CHECK(h, hipStreamSynchronize(h->stream));
    return PISCES_OK;
    });
}

// IStateManager.AddAlleleCounts + FindCandidates for the decoded batch: nothing of it comes back to the host.  What pisces_hip_add_reads
// takes from a pass over the reads' CIGARs (log slots, candidate-record slots, the blocks every read touches, the reads it refuses)
// the decode kernel has made where the reads are; the host creates the blocks from a bit map and enqueues the read walk and the
// candidate discovery.
int32_t pisces_hip_bam_fetch_directions(PiscesHip* h, uint8_t* directions, uint8_t* deletion_directions)
{
    return abi_guard<int32_t>(h, [&]() -> int32_t {
    if (!h) return PISCES_E_INVALID_ARG;
    if (!h->bam.valid) return fail(h, PISCES_E_STATE, "bam_fetch_directions: no decoded batch (pisces_hip_bam_decode first)");
    auto& B = h->bam;
    if (B.moved) return fail(h, PISCES_E_STATE, "bam_fetch_directions: the decoded batch has been added to the read store");
    if (!B.has_dirs) return 0;
    PISCES_HIP_CHECK(h, hipSetDevice(h->device));
    if (directions && B.n_bases) PISCES_HIP_CHECK(h, hipMemcpyAsync(directions, B.dirs.p + kSegmentPad, (size_t)B.n_bases, hipMemcpyDeviceToHost, h->stream));
    if (deletion_directions && B.n_ops) PISCES_HIP_CHECK(h, hipMemcpyAsync(deletion_directions, B.del_dirs.p, 2 * (size_t)B.n_ops, hipMemcpyDeviceToHost, h->stream));
    PISCES_HIP_CHECK(h, hipStreamSynchronize(h->stream));
    return 1;
    });
}

int32_t pisces_hip_add_decoded_reads(PiscesHip* h)
{
    return abi_guard<int32_t>(h, [&]() -> int32_t {
    if (!h) return PISCES_E_INVALID_ARG;
    if (!h->bam.valid) return fail(h, PISCES_E_STATE, "add_decoded_reads: no decoded batch (pisces_hip_bam_decode first)");
    HostTimer timer(&h->host_time[0]);
    auto& B = h->bam;
    if (B.moved || B.added) return fail(h, PISCES_E_STATE, "add_decoded_reads: the decoded batch has been added already");
    if (B.min_bq != h->cfg.min_base_call_quality) return fail(h, PISCES_E_STATE, "add_decoded_reads: decoded with another minimum base quality");
    { int32_t rcp = refuse_while_batch_is_open(h, "add_decoded_reads"); if (rcp) return rcp; }
    const int32_t nr = (int32_t)B.n_reads;
    if (nr == 0) return PISCES_OK;
    PISCES_HIP_CHECK(h, hipSetDevice(h->device));
    { int32_t rcf = consume_found(h); if (rcf) return rcf; }
    if (B.first_error != ~0ull) {
        const std::string read = " (read " + std::to_string((long long)(B.first_error >> 3)) + " of the decoded batch)";
        switch ((int)(B.first_error & 7ull)) {
            case kBamReadPositionNotPositive: return fail(h, PISCES_E_INVALID_ARG, "Position must be greater than 0." + read);
            case kBamReadCigarLongerThanRead: return fail(h, PISCES_E_INVALID_ARG, "add_decoded_reads: CIGAR does not match the read" + read);
            case kBamReadPastInt32: return fail(h, PISCES_E_INVALID_ARG, "add_decoded_reads: read runs past position 2^31 - 1" + read);
            case kBamReadBadDirectionTag: return fail(h, PISCES_E_INVALID_ARG, "add_decoded_reads: unexpected format in a direction string (XD tag)" + read);
            default: return fail(h, PISCES_E_INVALID_ARG, "add_decoded_reads: read runs far past the end of its reference sequence" + read);
        }
    }
    const bool find_on_device = !h->h_ref.empty();
    const int64_t found_slots = (find_on_device && !h->snv_walk) ? B.found_slots : 0, found_pool = (find_on_device && !h->snv_walk) ? B.found_pool : 0;
    if (found_slots > 0x7FFFFFF0ll || found_pool > 0x7FFFFFF0ll) return fail(h, PISCES_E_INVALID_ARG, "add_decoded_reads: too many insertions / deletions in one batch");
    h->eqx_in_batch = find_on_device && !h->snv_walk && B.has_eqx;
    // commit: the blocks the reads touch (GetBlock, RegionStateManager.cs:361-383) and the totals — only once the batch is in the store / the log
    // (a failed add leaves neither empty blocks nor readsProcessed / readsSkipped that pisces_hip_reduce_summary would add up)
    auto commit = [&]() {
        for (size_t w = 0; w < B.block_map.size(); w++)
            for (uint32_t bits = B.block_map[w]; bits; bits &= bits - 1)
                (void)get_block(h, (int32_t)(((int64_t)w * 32 + __builtin_ctz(bits)) * h->cfg.block_size + 1));
        h->stats[2] += nr;
        h->stats[3] += B.n_skipped;
        B.added = true;   // consumed: a second add of the same decoded batch is refused, whichever way it went into the store
    };
    if (h->read_path == 1) {
        const int32_t rcs = add_decoded_reads_store(h, found_slots, found_pool, find_on_device);
        if (rcs == PISCES_OK) commit();
        return rcs;
    }
    int32_t rc = log_reserve(h, B.log_slots);
    if (rc) return rc;
    DevReadBatch db;
    db.position = B.position.p; db.flags = B.flags.p; db.cigar_offset = B.cigar_offset.p; db.cigar_op = B.cigar_op.p; db.cigar_len = B.cigar_len.p;
    db.seq_offset = B.seq_offset.p; db.bases = B.bases.p + kSegmentPad; db.quals = B.quals.p + kSegmentPad;
    db.dirs = B.has_dirs ? B.dirs.p + kSegmentPad : nullptr; db.n_reads = nr;
    const int c = h->log_cur;
    hipLaunchKernelGGL(expand_reads_kernel, dim3(expand_reads_grid(nr)), dim3(256), 0, h->stream, db, (const long long*)B.d_slots.p,
                       (long long)h->log_ub, h->cfg.min_base_call_quality, h->d_log_pos[c].p, h->d_log_tup[c].p, h->d_log_n.p + 2, expand_reads_per_wave(nr));
    PISCES_HIP_CHECK(h, hipGetLastError());
    if (find_on_device && (h->snv_walk || found_slots > 0 || h->eqx_in_batch)) {
        int32_t rcd = enqueue_candidate_discovery(h, db, B.has_dirs ? B.del_dirs.p : nullptr, nr, (const int32_t*)B.d_fslots.p, found_slots, found_pool);
        if (rcd) return rcd;
    }
    h->log_ub += B.log_slots;
    commit();
    return PISCES_OK;
    });
}

