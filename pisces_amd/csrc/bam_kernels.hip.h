// bam_kernels.hip.h — BAM records cut on the device (the stage between the BGZF inflate and the read
// batch): what BamReader.GetNextAlignment (src/lib/Alignment.IO/BamReader.cs:137 ff.) and AlignmentSource.ShouldSkipRead
// (src/exe/Pisces/Logic/Alignment/AlignmentsSource.cs:84-92) do one record at a time on the host.
//
// The inflated stream stays in HBM.  A BAM record is [block_size:int32][block_size bytes]; where a record starts is only known
// from the one before it, a serial chain over the whole file.  It is cut without a serial pass over the bytes:
//   bam_header_kernel   the header (magic, text, reference names) -> offset of the first record
//   bam_chain_kernel    per 32 KiB chunk, EVERY byte offset s is taken as a possible record start: next(s) = s + 4 + le32(s).  Pointer
//                       jumping in LDS (next <- next o next, 10 rounds) turns that into "where does the chain from s leave the chunk"
//                       for all 32768 offsets at once — no knowledge of the true starts needed.
//                       False chains die within a few records (a random word is no block_size), so the chains that are still alive
//                       when they leave a chunk have all merged into the true one: the chunk also reports the exit that every live
//                       chain from its first 4 KiB shares, when there is exactly one.
//   bam_entry_guess_kernel / bam_entry_check_kernel   entry(c + 1) taken from that shared exit of chunk c, for all chunks at once, and
//                       then every link checked against the pointers (entry(c + 1) must be where the chain from entry(c) leaves
//                       chunk c: by induction from the header the entries are then the true ones)
//   bam_entry_kernel    the fall-back when a chunk had no single shared exit or a link failed (long records, a first 4 KiB without a
//                       record start): one thread hops chunk to chunk, one dependent load per chunk
//   bam_count_kernel    one wave per chunk walks its records from the true entry: AlignmentSource.ShouldSkipRead, counts of kept reads,
//                       CIGAR operations and bases -> (after a scan) every chunk's place in the read batch
//   bam_decode_kernel   the same walk, all 64 lanes decoding each kept record into the SoA read batch the read walk takes
//                       (4-bit bases -> letters, CIGAR words -> operation + length, qualities), in file order.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pisces {

constexpr int kBamChunk = 32768;               // bytes of the inflated stream per workgroup
constexpr uint16_t kBamLeaves = 0x8000;        // pointer values >= this: the chain leaves the chunk, low 15 bits = bytes past its end
constexpr uint16_t kBamBroken = 0xFFFF;        // no record can start here (size out of range, or its chain runs into such a place)
constexpr int kBamGuessWindow = 4096;          // the first bytes of a chunk whose live chains vote for the chunk's exit
constexpr int kBamMinRecord = 32;              // fixed fields of a record after block_size
constexpr int kBamMaxRecord = kBamChunk - 8;   // a longer record would leave the chunk by more than 15 bits can say (long reads: not here)

struct BamFilter {   // AlignmentSourceConfig + the chromosome being called
    int32_t ref_id;
    int32_t min_map_quality, skip_duplicates, only_proper_pairs;
    int32_t min_base_quality;   // for the deletion-quality bits of the read metadata
};

struct BamCounts { long long reads, cigar_ops, bases, records, skipped; };

__device__ __forceinline__ int32_t bam_le32(const uint8_t* __restrict__ p)
{
    return (int32_t)((uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24));
}
__device__ __forceinline__ uint32_t bam_le16(const uint8_t* __restrict__ p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8); }

// out[0] = offset of the first record, out[1] = n_ref, out[2] = status (0 ok)
__global__ void bam_header_kernel(const uint8_t* __restrict__ s, int64_t n, long long* __restrict__ out)
{
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    out[0] = 0; out[1] = 0; out[2] = 1;
    if (n < 12 || s[0] != 'B' || s[1] != 'A' || s[2] != 'M' || s[3] != 1) return;
    const int64_t l_text = bam_le32(s + 4);
    if (l_text < 0 || 8 + l_text + 4 > n) return;
    int64_t p = 8 + l_text;
    const int64_t n_ref = bam_le32(s + p);
    p += 4;
    if (n_ref < 0) return;
    for (int64_t i = 0; i < n_ref; i++) {
        if (p + 4 > n) return;
        const int64_t l_name = bam_le32(s + p);
        if (l_name < 0 || p + 4 + l_name + 4 > n) return;
        p += 4 + l_name + 4;
    }
    out[0] = p; out[1] = n_ref; out[2] = 0;
}

// exits[s] for every byte offset s of the stream: kBamLeaves | (bytes past the end of s's chunk) once the chain from s leaves its chunk
__global__ __launch_bounds__(1024) void bam_chain_kernel(const uint8_t* __restrict__ s, int64_t n, uint16_t* __restrict__ exits, uint32_t* __restrict__ shared_exit,
                                                         const long long* __restrict__ header)
{
    __shared__ uint8_t bytes[kBamChunk + 4];
    __shared__ uint16_t ptr[kBamChunk];
    __shared__ uint32_t exit_lo, exit_hi;
    if (threadIdx.x == 0) { exit_lo = 0xFFFFFFFFu; exit_hi = 0u; }
    const int64_t c0 = (int64_t)blockIdx.x * kBamChunk;
    const int len = (int)min((int64_t)kBamChunk, n - c0);
    for (int i = threadIdx.x; i < kBamChunk + 4; i += 1024) bytes[i] = (c0 + i < n) ? s[c0 + i] : (uint8_t)0;
    __syncthreads();
    for (int i = threadIdx.x; i < kBamChunk; i += 1024) {
        uint16_t v = kBamBroken;
        if (i < len && c0 + i + 4 <= n) {
            const int32_t bs = bam_le32(bytes + i);
            if (bs >= kBamMinRecord && bs <= kBamMaxRecord) {
                const int64_t nxt = (int64_t)i + 4 + bs;          // relative to the chunk
                if (c0 + nxt <= n) v = nxt < len ? (uint16_t)nxt : (uint16_t)(kBamLeaves | (uint16_t)(nxt - len));
            }
        }
        ptr[i] = v;
    }
    __syncthreads();
    // a chain inside one chunk has at most 32768 / 36 = 910 hops: ten doublings reach its end (in-place updates only ever move a
    // pointer further along its own chain)
    for (int round = 0; round < 10; round++) {
        for (int i = threadIdx.x; i < len; i += 1024) {
            const uint16_t v = ptr[i];
            if (v < kBamLeaves) ptr[i] = ptr[v];
        }
        __syncthreads();
    }
    for (int i = threadIdx.x; i < len; i += 1024) exits[c0 + i] = ptr[i];
    // the exit all live chains from the chunk's first kBamGuessWindow bytes share (0: none alive, or more than one exit); in the chunk
    // the header ends in, the window starts at the first record
    const long long first = header[2] == 0 ? header[0] : 0;
    const int w0 = first >= c0 && first < c0 + len ? (int)(first - c0) : 0;
    for (int i = w0 + threadIdx.x; i < min(len, w0 + kBamGuessWindow); i += 1024) {
        const uint32_t v = ptr[i];
        if (v >= kBamLeaves && v != kBamBroken) { atomicMin(&exit_lo, v); atomicMax(&exit_hi, v); }
    }
    __syncthreads();
    if (threadIdx.x == 0) shared_exit[blockIdx.x] = exit_lo == exit_hi ? exit_lo : 0u;
}

// entry[c] for every chunk at once, from the shared exit of the chunk before it (fallback[0] = 1 when some chunk has none)
__global__ void bam_entry_guess_kernel(const uint32_t* __restrict__ shared_exit, int64_t n, const long long* __restrict__ header, int64_t n_chunks,
                                       long long* __restrict__ entry, int32_t* __restrict__ status, int32_t* __restrict__ fallback)
{
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_chunks) return;
    if (header[2] != 0) { if (c == 0) status[0] = 1; entry[c] = -1; return; }
    const int64_t first = header[0], c_first = first / kBamChunk;
    long long e = -1;
    if (first < n) {
        if (c == c_first) e = first;
        else if (c > c_first) {
            const uint32_t v = shared_exit[c - 1];
            if (v == 0u) { atomicOr(fallback, 1); }
            else {
                const int64_t at = min(c * (int64_t)kBamChunk, n) + (int64_t)(v & 0x7FFFu);
                if (at < n) {
                    if (at / kBamChunk == c) e = at; else atomicOr(fallback, 1);
                }
            }
        }
    }
    entry[c] = e;
}

// every link of the guessed entries against the pointers; fallback[0] = 1 on any doubt
__global__ void bam_entry_check_kernel(const uint16_t* __restrict__ exits, int64_t n, const long long* __restrict__ header, int64_t n_chunks,
                                       const long long* __restrict__ entry, int32_t* __restrict__ fallback)
{
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_chunks || header[2] != 0) return;
    const int64_t first = header[0], c_first = first / kBamChunk;
    if (first >= n) return;                   // no records at all: every entry is -1, nothing to check (the header alone is the stream)
    if (c < c_first) return;
    // where the chain from chunk k's entry leaves it (-1: not a live pointer)
    auto link = [&](int64_t k) -> int64_t {
        const uint16_t v = exits[entry[k]];
        if (v == kBamBroken || v < kBamLeaves) return -1;
        return min((k + 1) * (int64_t)kBamChunk, n) + (int64_t)(v & 0x7FFF);
    };
    const long long e = entry[c];
    if (e < 0) {
        // no record starts here: only the last chunk may say so, when the last record ends with the stream inside it
        if (!(c == n_chunks - 1 && c > c_first && entry[c - 1] >= 0 && link(c - 1) == n)) atomicOr(fallback, 1);
        return;
    }
    const int64_t at = link(c);
    if (at < 0) { atomicOr(fallback, 1); return; }
    if (at < n) {
        if (c + 1 >= n_chunks || at != entry[c + 1]) atomicOr(fallback, 1);
    } else if (at != n || (c + 1 < n_chunks && (c + 2 < n_chunks || entry[c + 1] != -1))) {
        atomicOr(fallback, 1);                // (the serial pass reports how a chain ends that does not end with the stream)
    }
}

// entry[c] = offset of the first record that STARTS in chunk c (-1: none); status[0] != 0 on a broken chain
__global__ void bam_entry_kernel(const uint16_t* __restrict__ exits, int64_t n, const long long* __restrict__ header, int64_t n_chunks,
                                 long long* __restrict__ entry, int32_t* __restrict__ status, const int32_t* __restrict__ fallback)
{
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    if (fallback && fallback[0] == 0) return;   // the guessed entries passed every check
    for (int64_t c = 0; c < n_chunks; c++) entry[c] = -1;
    if (header[2] != 0) { status[0] = 1; return; }
    int64_t at = header[0];
    while (at < n) {
        const int64_t c = at / kBamChunk;
        entry[c] = at;
        const uint16_t v = exits[at];
        if (v == kBamBroken || v < kBamLeaves) { status[0] = 2; status[1] = (int32_t)c; return; }   // (after ten doublings every live pointer leaves)
        const int64_t chunk_end = min((c + 1) * (int64_t)kBamChunk, n);
        at = chunk_end + (v & 0x7FFF);
    }
    if (at != n) { status[0] = 3; }
}

__device__ __forceinline__ bool bam_keep(const uint8_t* __restrict__ rec, const BamFilter& F)
{
    // rec points behind block_size: refID, pos, l_read_name, mapq, bin, n_cigar_op, flag, l_seq, ...
    const int32_t ref_id = bam_le32(rec);
    const uint32_t mapq = rec[9], n_cigar = bam_le16(rec + 12), flag = bam_le16(rec + 14);
    if (ref_id != F.ref_id) return false;
    // ShouldSkipRead: !IsMapped || !IsPrimaryAlignment || (OnlyUseProperPairs && !IsProperPair) || (SkipDuplicates && IsPcrDuplicate)
    //                 || MapQuality < MinimumMapQuality || !HasCigar
    if (flag & 0x4) return false;
    if (flag & 0x100) return false;
    if (F.only_proper_pairs && !(flag & 0x2)) return false;
    if (F.skip_duplicates && (flag & 0x400)) return false;
    if ((int32_t)mapq < F.min_map_quality) return false;
    if (n_cigar == 0) return false;
    return true;
}

// per chunk: {kept reads, CIGAR operations, bases, records of the chromosome that were skipped}
__global__ __launch_bounds__(64) void bam_count_kernel(const uint8_t* __restrict__ s, int64_t n, const long long* __restrict__ entry,
                                                       BamFilter F, int32_t* __restrict__ n_reads, int32_t* __restrict__ n_ops,
                                                       int32_t* __restrict__ n_bases, int32_t* __restrict__ n_skipped)
{
    const int64_t c = blockIdx.x;
    if (threadIdx.x != 0) return;
    int reads = 0, ops = 0, bases = 0, skipped = 0;
    int64_t at = entry[c];
    const int64_t chunk_end = min((c + 1) * (int64_t)kBamChunk, n);
    while (at >= 0 && at < chunk_end) {
        const int32_t bs = bam_le32(s + at);
        const uint8_t* rec = s + at + 4;
        if (bam_keep(rec, F)) {
            reads++;
            ops += (int)bam_le16(rec + 12);
            bases += bam_le32(rec + 16);
        } else if (bam_le32(rec) == F.ref_id) {
            skipped++;
        }
        at += 4 + (int64_t)bs;
    }
    n_reads[c] = reads; n_ops[c] = ops; n_bases[c] = bases; n_skipped[c] = skipped;
}

// in-place exclusive scans of three int32 arrays of n + 1 elements (the last receives the total) by one workgroup
__global__ __launch_bounds__(1024) void bam_scan3_kernel(int32_t* __restrict__ a, int32_t* __restrict__ b, int32_t* __restrict__ d, int32_t n)
{
    __shared__ long long sa[1024], sb[1024], sd[1024];
    __shared__ long long base[3];
    if (threadIdx.x == 0) { base[0] = base[1] = base[2] = 0; }
    __syncthreads();
    for (int32_t start = 0; start < n; start += 1024) {
        const int32_t i = start + (int32_t)threadIdx.x;
        const long long va = i < n ? a[i] : 0, vb = i < n ? b[i] : 0, vd = i < n ? d[i] : 0;
        sa[threadIdx.x] = va; sb[threadIdx.x] = vb; sd[threadIdx.x] = vd;
        __syncthreads();
        for (int k = 1; k < 1024; k <<= 1) {
            long long xa = 0, xb = 0, xd = 0;
            if ((int)threadIdx.x >= k) { xa = sa[threadIdx.x - k]; xb = sb[threadIdx.x - k]; xd = sd[threadIdx.x - k]; }
            __syncthreads();
            sa[threadIdx.x] += xa; sb[threadIdx.x] += xb; sd[threadIdx.x] += xd;
            __syncthreads();
        }
        if (i < n) {
            a[i] = (int32_t)(base[0] + sa[threadIdx.x] - va);
            b[i] = (int32_t)(base[1] + sb[threadIdx.x] - vb);
            d[i] = (int32_t)(base[2] + sd[threadIdx.x] - vd);
        }
        __syncthreads();
        if (threadIdx.x == 1023) { base[0] += sa[1023]; base[1] += sb[1023]; base[2] += sd[1023]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { a[n] = (int32_t)base[0]; b[n] = (int32_t)base[1]; d[n] = (int32_t)base[2]; }
}

// The read batch (the arrays PiscesReadBatch names), in file order.  read0 / op0 / base0: the scanned counts of bam_count_kernel.
// op_quality[k] bit 0: CheckDeletionQuality at the read index where CIGAR operation k starts (both flanking qualities >= minBQ);
// read_quality[r] bit 0: the same at the last read base — what the host needs to know which blocks a gap touches without the qualities.
__global__ __launch_bounds__(64) void bam_decode_kernel(const uint8_t* __restrict__ s, int64_t n, const long long* __restrict__ entry, BamFilter F,
                                                        const int32_t* __restrict__ read0, const int32_t* __restrict__ op0,
                                                        const int32_t* __restrict__ base0, int32_t* __restrict__ position,
                                                        uint8_t* __restrict__ flags, int32_t* __restrict__ cigar_offset,
                                                        uint8_t* __restrict__ cigar_op, uint32_t* __restrict__ cigar_len,
                                                        int32_t* __restrict__ seq_offset, uint8_t* __restrict__ bases, uint8_t* __restrict__ quals,
                                                        uint8_t* __restrict__ op_quality, uint8_t* __restrict__ read_quality)
{
    const int64_t c = blockIdx.x;
    const int lane = threadIdx.x;
    int64_t at = entry[c];
    const int64_t chunk_end = min((c + 1) * (int64_t)kBamChunk, n);
    int r = read0[c], o = op0[c], b = base0[c];
    while (at >= 0 && at < chunk_end) {   // (wave-uniform: every lane follows the same chain)
        const int32_t bs = bam_le32(s + at);
        const uint8_t* rec = s + at + 4;
        if (bam_keep(rec, F)) {
            const int l_name = rec[8], n_cigar = (int)bam_le16(rec + 12), l_seq = bam_le32(rec + 16);
            const uint32_t flag = bam_le16(rec + 14);
            const uint8_t* cig = rec + 32 + l_name;
            const uint8_t* seq = cig + 4 * n_cigar;
            const uint8_t* ql = seq + (l_seq + 1) / 2;
            if (lane == 0) {
                position[r] = bam_le32(rec + 4) + 1;          // BAM positions are 0-based
                flags[r] = (flag & 0x10) ? 1 : 0;             // bit 0 = reverse strand (PiscesReadBatch.flags)
                cigar_offset[r] = o;
                seq_offset[r] = b;
                const int lastq = l_seq > 0 ? ql[l_seq - 1] : 0, prevq = l_seq > 1 ? ql[l_seq - 2] : lastq;
                read_quality[r] = (l_seq > 0 && lastq >= F.min_base_quality && prevq >= F.min_base_quality) ? 1 : 0;
            }
            for (int k = lane; k < l_seq; k += 64) {
                const uint32_t nib = (seq[k >> 1] >> ((k & 1) ? 0 : 4)) & 0xFu;
                // "=ACMGRSVTWYHKDBN" (SAM specification 4.2.3; BamReader decodes with the same table), eight letters per constant
                const unsigned long long hi = 0x4E42444B48595754ull;   // T W Y H K D B N
                const unsigned long long lo8 = (unsigned long long)'=' | ((unsigned long long)'A' << 8) | ((unsigned long long)'C' << 16) |
                                               ((unsigned long long)'M' << 24) | ((unsigned long long)'G' << 32) | ((unsigned long long)'R' << 40) |
                                               ((unsigned long long)'S' << 48) | ((unsigned long long)'V' << 56);
                bases[b + k] = (uint8_t)(((nib < 8 ? lo8 : hi) >> (8 * (nib & 7))) & 0xFFu);
                quals[b + k] = ql[k];
            }
            if (lane == 0) {   // the CIGAR: a handful of operations, with the read index each one starts at
                int ri = 0;
                for (int k = 0; k < n_cigar; k++) {
                    const uint32_t v = (uint32_t)bam_le32(cig + 4 * k);
                    const uint32_t op = v & 0xFu, len = v >> 4;
                    const uint8_t letter = op == 0 ? 'M' : op == 1 ? 'I' : op == 2 ? 'D' : op == 3 ? 'N' : op == 4 ? 'S' : op == 5 ? 'H'
                                           : op == 6 ? 'P' : op == 7 ? '=' : op == 8 ? 'X' : '?';
                    cigar_op[o + k] = letter;
                    cigar_len[o + k] = len;
                    uint8_t ok = 0;
                    if (l_seq > 0) {
                        const int after = ri < l_seq ? ql[ri] : ql[l_seq - 1], before = ri > 0 ? ql[min(ri, l_seq) - 1] : after;
                        ok = (before >= F.min_base_quality && after >= F.min_base_quality) ? 1 : 0;
                    }
                    op_quality[o + k] = ok;
                    if (op == 0 || op == 1 || op == 4 || op == 7 || op == 8) ri += (int)len;
                }
            }
            r++;
            o += n_cigar;
            b += l_seq;
        }
        at += 4 + (int64_t)bs;
    }
}

}  // namespace pisces
