// bam_kernels.hip.h — BAM records cut on the device (the stage between the BGZF inflate and the read
// batch): what BamReader.GetNextAlignment (src/lib/Alignment.IO/BamReader.cs:137 ff.) and AlignmentSource.ShouldSkipRead
// (src/exe/Pisces/Logic/Alignment/AlignmentsSource.cs:84-92) do one record at a time on the host.
//
// The inflated stream stays in HBM.  A BAM record is [block_size:int32][block_size bytes]; where a record starts is only known
// from the one before it, a serial chain over the whole file.  It is cut without a serial pass over the bytes:
//   bam_header_kernel   the header (magic, text, reference names) -> offset of the first record
//   bam_chain_kernel    per 32 KiB chunk, every byte offset s of its first 4 KiB is taken as a possible record start and its chain
//                       (next(s) = s + 4 + le32(s)) followed through the chunk in LDS: "where does the chain from s leave the chunk",
//                       with no knowledge of the true starts.
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

#include "finder_kernels.hip.h"   // kFoundInline

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
    int32_t block_size;         // RegionStateManager's block grid: the decode marks the blocks every read touches
};

struct BamCounts { long long reads, cigar_ops, bases, records, skipped; };

// little-endian fields at any byte offset: ONE load each (gfx950 runs with unaligned access enabled; four byte loads and three shifts
// per field were most of what the record kernels issued)
__device__ __forceinline__ int32_t bam_le32(const uint8_t* __restrict__ p)
{
    int32_t v;
    __builtin_memcpy(&v, p, 4);
    return v;
}
__device__ __forceinline__ uint32_t bam_le16(const uint8_t* __restrict__ p)
{
    uint16_t v;
    __builtin_memcpy(&v, p, 2);
    return v;
}

// out[0] = offset of the first record, out[1] = n_ref, out[2] = status (0 ok), out[3] = l_ref of reference sequence ref_id (0: no such)
__global__ void bam_header_kernel(const uint8_t* __restrict__ s, int64_t n, long long* __restrict__ out, int32_t ref_id)
{
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    out[0] = 0; out[1] = 0; out[2] = 1; out[3] = 0;
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
        if (i == ref_id) out[3] = bam_le32(s + p + 4 + l_name);
        p += 4 + l_name + 4;
    }
    out[0] = p; out[1] = n_ref; out[2] = 0;
}

// The first offset of chunk c whose chain is followed (the window of kBamGuessWindow offsets starts there): the chunk's first byte, or
// the first record in the chunk the header ends in.
__device__ __forceinline__ int bam_window_start(int64_t c, const long long* __restrict__ header)
{
    const long long first = header[2] == 0 ? header[0] : 0;
    return first / kBamChunk == c ? (int)(first - c * (long long)kBamChunk) : 0;
}

// where the chain from stream offset `at` leaves its chunk: kBamLeaves | bytes past the chunk's end, kBamBroken, or kBamOutside when
// `at` does not lie in its chunk's window (the table only holds the window's offsets)
constexpr uint16_t kBamOutside = 0x7FFF;
__device__ __forceinline__ uint16_t bam_exit_of(const uint16_t* __restrict__ exits, int64_t at, const long long* __restrict__ header)
{
    const int64_t c = at / kBamChunk;
    const int i = (int)(at - c * kBamChunk) - bam_window_start(c, header);
    if (i < 0 || i >= kBamGuessWindow) return kBamOutside;
    return exits[c * kBamGuessWindow + i];
}

// one hop of a chain inside the stream: the offset behind the record at `at`, or -1 when no record can start there
__device__ __forceinline__ int64_t bam_next_record(const uint8_t* __restrict__ s, int64_t n, int64_t at)
{
    if (at + 4 > n) return -1;
    const int32_t bs = bam_le32(s + at);
    if (bs < kBamMinRecord || bs > kBamMaxRecord || at + 4 + bs > n) return -1;
    return at + 4 + bs;
}

// exits[c * kBamGuessWindow + k]: where the chain from offset k of chunk c's window leaves the chunk -- kBamLeaves | (bytes past the
// chunk's end), or kBamBroken.  Every offset of the window is taken as a possible record start (next(s) = s + 4 + le32(s)) and
// followed through the chunk's bytes in LDS; false chains die within a few hops (a random word is no block_size), the true chain is
// ~170 hops of 188-byte records.  The chunk also reports the exit all its live, record-like starts share (shared_exit).
__global__ __launch_bounds__(256) void bam_chain_kernel(const uint8_t* __restrict__ s, int64_t n, uint16_t* __restrict__ exits, uint32_t* __restrict__ shared_exit,
                                                        const long long* __restrict__ header)
{
    __shared__ __attribute__((aligned(16))) uint8_t bytes[kBamChunk + 16];
    __shared__ uint32_t exit_lo, exit_hi;
    if (threadIdx.x == 0) { exit_lo = 0xFFFFFFFFu; exit_hi = 0u; }
    const int64_t c = blockIdx.x, c0 = c * (int64_t)kBamChunk;
    const int len = (int)min((int64_t)kBamChunk, n - c0);
    // the chunk and the four bytes behind it (a block_size that starts in the chunk's last bytes), 16 bytes a thread and load where
    // the whole 16 lie inside the stream's buffer (it carries 16 bytes of slack), bytes otherwise
    for (int i = threadIdx.x * 16; i < kBamChunk + 16; i += 256 * 16) {
        if (c0 + i + 16 <= n + 16) *reinterpret_cast<uint4*>(bytes + i) = *reinterpret_cast<const uint4*>(s + c0 + i);
        else for (int k = 0; k < 16; k++) bytes[i + k] = 0;
    }
    __syncthreads();
    const int w0 = bam_window_start(c, header);
    const long long n_ref = header[2] == 0 ? header[1] : 0;
    for (int k = threadIdx.x; k < kBamGuessWindow; k += 256) {
        const int i = w0 + k;
        uint16_t v = kBamBroken;
        if (i < len) {
            int at = i;
            for (;;) {   // (at most 32768 / 36 = 910 hops)
                if (c0 + at + 4 > n) break;
                const int32_t bs = bam_le32(bytes + at);
                if (bs < kBamMinRecord || bs > kBamMaxRecord || c0 + at + 4 + (int64_t)bs > n) break;
                const int nxt = at + 4 + bs;
                if (nxt >= len) { v = (uint16_t)(kBamLeaves | (uint16_t)(nxt - len)); break; }
                at = nxt;
            }
            // the vote: only starts that look like a record (reference id inside the header's table, a block_size that holds its own
            // fixed fields, name, CIGAR and bases: an integer field that merely reads like a block_size does not).  The chains
            // themselves stay as permissive as BamReader is: the vote decides nothing that the check of the links does not confirm.
            if (v != kBamBroken && i + 4 + kBamMinRecord <= len) {
                const int32_t bs = bam_le32(bytes + i), ref_id = bam_le32(bytes + i + 4), l_seq = bam_le32(bytes + i + 20);
                const int l_name = bytes[i + 12], n_cigar = (int)bam_le16(bytes + i + 16);
                if (ref_id >= -1 && ref_id < n_ref && l_name >= 1 && l_seq >= 0 && (long long)bs >= 32ll + l_name + 4ll * n_cigar + (l_seq + 1ll) / 2 + l_seq) {
                    atomicMin(&exit_lo, (uint32_t)v);
                    atomicMax(&exit_hi, (uint32_t)v);
                }
            }
        }
        exits[c * kBamGuessWindow + k] = v;
    }
    __syncthreads();
    if (threadIdx.x == 0) shared_exit[c] = exit_lo == exit_hi ? exit_lo : 0u;   // (0: none alive, or more than one exit)
}

// entry[c] for every chunk at once, from the shared exit of the chunk before it.  Where that chunk has none (two live chains with
// different exits), the thread goes back to the nearest chunk whose entry is known and hops forward from there through the table (at
// most kBamGuessBack chunks; fallback[0] = 1 beyond, or when a hop starts outside its chunk's window).
constexpr int kBamGuessBack = 64;
__global__ void bam_entry_guess_kernel(const uint32_t* __restrict__ shared_exit, const uint16_t* __restrict__ exits, int64_t n,
                                       const long long* __restrict__ header, int64_t n_chunks, long long* __restrict__ entry,
                                       int32_t* __restrict__ status, int32_t* __restrict__ fallback)
{
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_chunks) return;
    if (header[2] != 0) { if (c == 0) status[0] = 1; entry[c] = -1; return; }
    const int64_t first = header[0], c_first = first / kBamChunk;
    long long e = -1;
    if (first < n && c >= c_first) {
        // the nearest chunk at or before c whose entry does not depend on another chunk's
        int64_t p = c;
        while (p > c_first && shared_exit[p - 1] == 0u && c - p < kBamGuessBack) p--;
        long long at;
        if (p == c_first) at = first;
        else if (shared_exit[p - 1] != 0u) at = min(p * (int64_t)kBamChunk, n) + (long long)(shared_exit[p - 1] & 0x7FFFu);
        else { atomicOr(fallback, 1); at = -1; }
        // forward through chunks p .. c - 1
        for (int64_t k = p; at >= 0 && k < c; k++) {
            if (at >= n || at / kBamChunk != k) { at = at >= n ? n : -1; break; }
            const uint16_t v = bam_exit_of(exits, at, header);
            if (v == kBamBroken || v < kBamLeaves) { at = -1; break; }
            at = min((k + 1) * (int64_t)kBamChunk, n) + (long long)(v & 0x7FFF);
        }
        if (at < 0) atomicOr(fallback, 1);
        else if (at < n) {
            if (at / kBamChunk == c) e = at; else atomicOr(fallback, 1);
        }
    }
    entry[c] = e;
}

// every link of the guessed entries against the table; fallback[0] = 1 on any doubt
__global__ void bam_entry_check_kernel(const uint16_t* __restrict__ exits, int64_t n, const long long* __restrict__ header, int64_t n_chunks,
                                       const long long* __restrict__ entry, int32_t* __restrict__ fallback)
{
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_chunks || header[2] != 0) return;
    const int64_t first = header[0], c_first = first / kBamChunk;
    if (first >= n) return;                   // no records at all: every entry is -1, nothing to check (the header alone is the stream)
    if (c < c_first) return;
    // where the chain from chunk k's entry leaves it (-1: not a live pointer, or an entry outside the window)
    auto link = [&](int64_t k) -> int64_t {
        const uint16_t v = bam_exit_of(exits, entry[k], header);
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

// The fall-back: entry[c] = offset of the first record that STARTS in chunk c (-1: none), hopping chunk to chunk from the end of the
// header; status[0] != 0 on a broken chain.  A hop is one look-up when the entry lies in its chunk's window, and a walk over the
// chunk's records otherwise (long records: few to a chunk).
__global__ void bam_entry_kernel(const uint8_t* __restrict__ s, const uint16_t* __restrict__ exits, int64_t n, const long long* __restrict__ header,
                                 int64_t n_chunks, long long* __restrict__ entry, int32_t* __restrict__ status, const int32_t* __restrict__ fallback)
{
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    if (fallback && fallback[0] == 0) return;   // the guessed entries passed every check
    for (int64_t c = 0; c < n_chunks; c++) entry[c] = -1;
    if (header[2] != 0) { status[0] = 1; return; }
    int64_t at = header[0];
    while (at < n) {
        const int64_t c = at / kBamChunk;
        entry[c] = at;
        const int64_t chunk_end = min((c + 1) * (int64_t)kBamChunk, n);
        const uint16_t v = bam_exit_of(exits, at, header);
        if (v == kBamOutside) {
            while (at >= 0 && at < chunk_end) at = bam_next_record(s, n, at);
            if (at < 0) { status[0] = 2; status[1] = (int32_t)c; return; }
        } else {
            if (v == kBamBroken || v < kBamLeaves) { status[0] = 2; status[1] = (int32_t)c; return; }
            at = chunk_end + (v & 0x7FFF);
        }
    }
    if (at != n) { status[0] = 3; }
}

// The value of the string tag t0 t1 (type 'Z') in a record's auxiliary fields [p, end): its first character and its length, or nullptr
// (BamAlignment.GetStringTag; SAM specification 4.2.4 for the value types that are skipped over).
__device__ inline const uint8_t* bam_find_string_tag(const uint8_t* p, const uint8_t* end, uint8_t t0, uint8_t t1, int* length)
{
    while (p + 3 <= end) {
        const uint8_t a = p[0], b = p[1], ty = p[2];
        p += 3;
        if (ty == 'Z' || ty == 'H') {
            const uint8_t* q = p;
            while (q < end && *q) q++;
            if (a == t0 && b == t1 && ty == 'Z') { *length = (int)(q - p); return p; }
            p = q + 1;
        } else if (ty == 'A' || ty == 'c' || ty == 'C') p += 1;
        else if (ty == 's' || ty == 'S') p += 2;
        else if (ty == 'i' || ty == 'I' || ty == 'f') p += 4;
        else if (ty == 'B') {
            if (p + 5 > end) return nullptr;
            const uint8_t sub = p[0];
            const long long count = (uint32_t)bam_le32(p + 1);
            const int size = (sub == 'c' || sub == 'C') ? 1 : (sub == 's' || sub == 'S') ? 2 : 4;
            if (count * size > end - (p + 5)) return nullptr;
            p += 5 + count * size;
        } else return nullptr;   // not a value type: the fields end here
    }
    return nullptr;
}
__device__ __forceinline__ const uint8_t* bam_aux_of(const uint8_t* rec)   // rec points behind block_size
{
    const int l_name = rec[8], n_cigar = (int)bam_le16(rec + 12), l_seq = bam_le32(rec + 16);
    return rec + 32 + l_name + 4 * n_cigar + (l_seq + 1) / 2 + l_seq;
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

// what pisces_hip_add_reads's host pass takes from a read's CIGAR, for the kept reads of the decoded batch: log slots (one per reference
// position the read spans: an upper bound of its observations), candidate-record slots (one per insertion / deletion), bytes of
// insertions too long for a record
struct BamCigarSums { long long ref_span; int read_span, indels, pool; bool eqx; };
__device__ __forceinline__ BamCigarSums bam_cigar_sums(const uint8_t* __restrict__ cig, int n_cigar)
{
    BamCigarSums t = {0, 0, 0, 0, false};
    for (int k = 0; k < n_cigar; k++) {
        const uint32_t v = (uint32_t)bam_le32(cig + 4 * k);
        const uint32_t op = v & 0xFu, len = v >> 4;
        if (op == 0 || op == 1 || op == 4 || op == 7 || op == 8) t.read_span += (int)len;    // M I S = X
        if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) t.ref_span += (long long)len;   // M D N = X
        if (op == 1 || op == 2) t.indels++;
        if (op == 1 && len > (uint32_t)kFoundInline) t.pool += (int)len;
        if (op == 7 || op == 8) t.eqx = true;
    }
    return t;
}

// per chunk: {kept reads, CIGAR operations, bases, records of the chromosome that were skipped, log slots, candidate slots, pool bytes}.
// Lane 0 walks the chunk's chain (one dependent load a record) and leaves the record offsets in LDS; the lanes then take a record each.
__global__ __launch_bounds__(64) void bam_count_kernel(const uint8_t* __restrict__ s, int64_t n, const long long* __restrict__ entry,
                                                       BamFilter F, int32_t* __restrict__ n_reads, int32_t* __restrict__ n_ops,
                                                       int32_t* __restrict__ n_bases, int32_t* __restrict__ n_skipped,
                                                       long long* __restrict__ n_span, int32_t* __restrict__ n_indels, int32_t* __restrict__ n_pool,
                                                       int32_t* __restrict__ status)
{
    __shared__ int32_t rec_at[kBamChunk / (4 + kBamMinRecord) + 2];   // relative to the chunk's first byte
    __shared__ int32_t n_rec_s;
    const int64_t c = blockIdx.x;
    const int64_t c0 = c * (int64_t)kBamChunk, chunk_end = min((c + 1) * (int64_t)kBamChunk, n);
    if (threadIdx.x == 0) {
        int k = 0;
        int64_t at = entry[c];
        while (at >= 0 && at < chunk_end) {
            rec_at[k++] = (int32_t)(at - c0);
            at += 4 + (int64_t)bam_le32(s + at);
        }
        n_rec_s = k;
    }
    __syncthreads();
    const int n_rec = n_rec_s;
    int reads = 0, ops = 0, bases = 0, skipped = 0, indels = 0, pool = 0;
    long long span = 0;
    for (int i = threadIdx.x; i < n_rec; i += 64) {
        const uint8_t* rec = s + c0 + rec_at[i] + 4;
        {
            // a record must hold what its own fields announce (name, CIGAR, packed bases, qualities): everything behind this kernel
            // reads those arrays by these lengths
            const long long bs = bam_le32(rec - 4), l_seq = bam_le32(rec + 16);
            if (l_seq < 0 || rec[8] < 1 || 32ll + rec[8] + 4ll * (long long)bam_le16(rec + 12) + (l_seq + 1) / 2 + l_seq > bs) {
                if (atomicCAS(status, 0, 4) == 0) status[1] = (int32_t)c;
                continue;
            }
        }
        if (bam_keep(rec, F)) {
            const int n_cigar = (int)bam_le16(rec + 12);
            int xd_len = 0;
            // status[2]: bit 0 stitched reads (the batch gets per-base directions), bit 1 reads with X or = operations
            if (!(status[2] & 1) && bam_find_string_tag(bam_aux_of(rec), rec + bam_le32(rec - 4), 'X', 'D', &xd_len)) atomicOr(&status[2], 1);
            reads++;
            ops += n_cigar;
            bases += bam_le32(rec + 16);
            const BamCigarSums t = bam_cigar_sums(rec + 32 + rec[8], n_cigar);
            span += t.ref_span;
            indels += t.indels;
            pool += t.pool;
            if (t.eqx && !(status[2] & 2)) atomicOr(&status[2], 2);
        } else if (bam_le32(rec) == F.ref_id) {
            skipped++;
        }
    }
    for (int d = 32; d >= 1; d >>= 1) {
        reads += __shfl_down(reads, d, 64); ops += __shfl_down(ops, d, 64); bases += __shfl_down(bases, d, 64);
        skipped += __shfl_down(skipped, d, 64); indels += __shfl_down(indels, d, 64); pool += __shfl_down(pool, d, 64);
        span += __shfl_down(span, d, 64);
    }
    if (threadIdx.x == 0) {
        n_reads[c] = reads; n_ops[c] = ops; n_bases[c] = bases; n_skipped[c] = skipped;
        n_span[c] = span; n_indels[c] = indels; n_pool[c] = pool;
    }
}

// in-place exclusive scans of up to three int32 arrays of n + 1 elements (the last receives the total) by one workgroup (d may be null)
// (totals64, optional: the three totals as they are, for the host to refuse a batch whose counts do not fit 31 bits before any of the
// wrapped 32-bit values is used as a size)
__global__ __launch_bounds__(1024) void bam_scan3_kernel(int32_t* __restrict__ a, int32_t* __restrict__ b, int32_t* __restrict__ d, int32_t n,
                                                         long long* __restrict__ totals64 = nullptr)
{
    __shared__ long long sa[1024], sb[1024], sd[1024];
    __shared__ long long base[3];
    if (threadIdx.x == 0) { base[0] = base[1] = base[2] = 0; }
    __syncthreads();
    for (int32_t start = 0; start < n; start += 1024) {
        const int32_t i = start + (int32_t)threadIdx.x;
        const long long va = i < n ? a[i] : 0, vb = i < n ? b[i] : 0, vd = (d && i < n) ? d[i] : 0;
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
            if (d) d[i] = (int32_t)(base[2] + sd[threadIdx.x] - vd);
        }
        __syncthreads();
        if (threadIdx.x == 1023) { base[0] += sa[1023]; base[1] += sb[1023]; base[2] += sd[1023]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        a[n] = (int32_t)base[0]; b[n] = (int32_t)base[1]; if (d) d[n] = (int32_t)base[2];
        if (totals64) { totals64[0] = base[0]; totals64[1] = base[1]; totals64[2] = base[2]; }
    }
}

// the same for one array of 64-bit counts
__global__ __launch_bounds__(1024) void bam_scan_ll_kernel(long long* __restrict__ a, int32_t n)
{
    __shared__ long long sa[1024];
    __shared__ long long base;
    if (threadIdx.x == 0) base = 0;
    __syncthreads();
    for (int32_t start = 0; start < n; start += 1024) {
        const int32_t i = start + (int32_t)threadIdx.x;
        const long long va = i < n ? a[i] : 0;
        sa[threadIdx.x] = va;
        __syncthreads();
        for (int k = 1; k < 1024; k <<= 1) {
            long long xa = 0;
            if ((int)threadIdx.x >= k) xa = sa[threadIdx.x - k];
            __syncthreads();
            sa[threadIdx.x] += xa;
            __syncthreads();
        }
        if (i < n) a[i] = base + sa[threadIdx.x] - va;
        __syncthreads();
        if (threadIdx.x == 1023) base += sa[1023];
        __syncthreads();
    }
    if (threadIdx.x == 0) a[n] = base;
}

// The read batch (the arrays PiscesReadBatch names), in file order.  read0 / op0 / base0: the scanned counts of bam_count_kernel.
// op_quality[k] bit 0: CheckDeletionQuality at the read index where CIGAR operation k starts (both flanking qualities >= minBQ);
// read_quality[r] bit 0: the same at the last read base.
// What pisces_hip_add_reads's host pass would make of the reads is made here as well, so that the batch can go into the handle's log
// without coming back: slots[r] / fslots[r] (exclusive sums of the reads' log slots and candidate-record slots, span0 / indel0 being
// the chunks' scanned sums), one bit per 1000-locus block a read touches (GetBlock for every position that receives a count,
// RegionStateManager.cs:361-383: the aligned segments, and a gap or terminal deletion when CheckDeletionQuality lets it count), and
// the first read the host pass would have refused (first_error = read index * 8 + code; codes below).
enum { kBamReadPositionNotPositive = 1, kBamReadCigarLongerThanRead = 2, kBamReadPastInt32 = 3, kBamReadPastBlockMap = 4, kBamReadBadDirectionTag = 5 };
constexpr int kBamMaxDirectionRuns = 64;   // runs of an XD tag ("3F4S3R": three); a tag with more is refused (kBamReadBadDirectionTag)
__global__ __launch_bounds__(256) void bam_decode_kernel(const uint8_t* __restrict__ s, int64_t n, const long long* __restrict__ entry, BamFilter F,
                                                        const int32_t* __restrict__ read0, const int32_t* __restrict__ op0,
                                                        const int32_t* __restrict__ base0, int32_t* __restrict__ position,
                                                        uint8_t* __restrict__ flags, int32_t* __restrict__ cigar_offset,
                                                        uint8_t* __restrict__ cigar_op, uint32_t* __restrict__ cigar_len,
                                                        int32_t* __restrict__ seq_offset, uint8_t* __restrict__ bases, uint8_t* __restrict__ quals,
                                                        uint8_t* __restrict__ op_quality, uint8_t* __restrict__ read_quality,
                                                        const long long* __restrict__ span0, const int32_t* __restrict__ indel0,
                                                        long long* __restrict__ slots, int32_t* __restrict__ fslots,
                                                        uint32_t* __restrict__ block_map, long long n_block_bits,
                                                        unsigned long long* __restrict__ first_error,
                                                        uint8_t* __restrict__ dirs /* per base, or nullptr: no read of the batch has an XD tag */,
                                                        uint8_t* __restrict__ del_dirs /* two per CIGAR operation, or nullptr */)
{
    // per wave: the XD tag of the record being decoded as (end in the expanded CIGAR, DirectionType) runs (CigarDirection, CigarDirection.cs:19-41)
    __shared__ int32_t xd_end[4][kBamMaxDirectionRuns];
    __shared__ uint8_t xd_dir[4][kBamMaxDirectionRuns];
    // Lane 0 walks the chunk's chain (one dependent load a record) and leaves the record offsets in LDS; the threads then take a record
    // each for what the records add to the batch's arrays (an exclusive scan over the chunk's records gives every record its place);
    // the four waves then decode a record each, in turn.
    __shared__ int32_t rec_at[kBamChunk / (4 + kBamMinRecord) + 2];   // relative to the chunk's first byte
    __shared__ int32_t rec_r[kBamChunk / (4 + kBamMinRecord) + 2], rec_o[kBamChunk / (4 + kBamMinRecord) + 2], rec_b[kBamChunk / (4 + kBamMinRecord) + 2],
        rec_f[kBamChunk / (4 + kBamMinRecord) + 2];
    __shared__ long long rec_s[kBamChunk / (4 + kBamMinRecord) + 2];
    __shared__ int32_t n_rec_s;
    const int64_t c = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t c0 = c * (int64_t)kBamChunk, chunk_end = min((c + 1) * (int64_t)kBamChunk, n);
    if (threadIdx.x == 0) {
        int k = 0;
        int64_t at = entry[c];
        while (at >= 0 && at < chunk_end) {
            rec_at[k++] = (int32_t)(at - c0);
            at += 4 + (int64_t)bam_le32(s + at);
        }
        n_rec_s = k;
    }
    __syncthreads();
    const int n_rec = n_rec_s;
    // what each record adds: 1 read (0 when it is not kept: rec_r then marks it), CIGAR operations, bases, log slots, candidate slots
    for (int i = threadIdx.x; i < n_rec; i += 256) {
        const uint8_t* rec = s + c0 + rec_at[i] + 4;
        int keep = 0, ops = 0, nb = 0, ind = 0;
        long long span = 0;
        if (bam_keep(rec, F)) {
            keep = 1;
            ops = (int)bam_le16(rec + 12);
            nb = bam_le32(rec + 16);
            const BamCigarSums t = bam_cigar_sums(rec + 32 + rec[8], ops);
            span = t.ref_span;
            ind = t.indels;
        }
        rec_r[i] = keep; rec_o[i] = ops; rec_b[i] = nb; rec_f[i] = ind; rec_s[i] = span;
    }
    __syncthreads();
    if (threadIdx.x == 0) {   // (at most 910 records: a serial scan of five counters is a few microseconds)
        int r = read0[c], o = op0[c], b = base0[c], f = indel0[c];
        long long sl = span0[c];
        for (int i = 0; i < n_rec; i++) {
            const int keep = rec_r[i], ops = rec_o[i], nb = rec_b[i], ind = rec_f[i];
            const long long span = rec_s[i];
            rec_r[i] = keep ? r : -1; rec_o[i] = o; rec_b[i] = b; rec_f[i] = f; rec_s[i] = sl;
            r += keep; o += ops; b += nb; f += ind; sl += span;
        }
    }
    __syncthreads();
    long long map_word = -1;      // lane 0 of each wave: the word of the block map its reads are setting bits in (reads come sorted:
    uint32_t map_bits = 0;        // one atomic a word instead of one a read)
    for (int rec_i = wave; rec_i < n_rec; rec_i += 4) {   // (wave-uniform)
        const int r = rec_r[rec_i], o = rec_o[rec_i], b = rec_b[rec_i], fslot = rec_f[rec_i];
        const long long slot = rec_s[rec_i];
        const uint8_t* rec = s + c0 + rec_at[rec_i] + 4;
        if (r >= 0) {   // (a kept record: AlignmentSource.ShouldSkipRead was applied when the records were counted above)
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
            // ---- stitched reads (Read.SequencedBaseDirectionMap / CigarDirections, Read.cs:340-400, 664-682): the XD tag gives a
            // DirectionType per base of the EXPANDED CIGAR (deleted bases included); without the tag every base has the strand's
            int n_runs = -1;   // -1: no XD tag
            if (dirs) {
                wave_lds_sync();   // (the record before this one is done with the runs)
                int xd_bad = 0;
                if (lane == 0) {
                    int xd_len = 0;
                    const uint8_t* xd = bam_find_string_tag(ql + l_seq, rec + bam_le32(rec - 4), 'X', 'D', &xd_len);
                    if (xd) {
                        int runs = 0, num = 0, end = 0, digits = 0;
                        for (int k = 0; k < xd_len && !xd_bad; k++) {
                            const uint8_t ch = xd[k];
                            if (ch >= '0' && ch <= '9') { num = num * 10 + (ch - '0'); digits++; continue; }
                            const int d = ch == 'F' ? PISCES_DIR_FORWARD : ch == 'R' ? PISCES_DIR_REVERSE : ch == 'S' ? PISCES_DIR_STITCHED : -1;
                            if (d < 0 || digits == 0 || runs == kBamMaxDirectionRuns) { xd_bad = 1; break; }   // (DirectionHelper.GetDirection / int.Parse throw)
                            end += num;
                            xd_end[wave][runs] = end;
                            xd_dir[wave][runs] = (uint8_t)d;
                            runs++;
                            num = 0;
                            digits = 0;
                        }
                        if (digits) xd_bad = 1;   // (CigarDirection: "Unexpected format in direction string")
                        n_runs = runs;
                        if (xd_bad) atomicMin(first_error, (unsigned long long)r * 8ull + (unsigned long long)kBamReadBadDirectionTag);
                    }
                }
                n_runs = __shfl(n_runs, 0, 64);
                wave_lds_sync();   // (lane 0's runs are in LDS for the other lanes)
            }
            auto direction_at = [&](int expanded_index) -> uint8_t {   // (bases the tag does not reach keep DirectionType's default)
                for (int j = 0; j < n_runs; j++)
                    if (expanded_index < xd_end[wave][j]) return xd_dir[wave][j];
                return (uint8_t)PISCES_DIR_FORWARD;
            };
            for (int k = lane; k < l_seq; k += 64) {
                if (dirs) {
                    uint8_t d = (flag & 0x10) ? PISCES_DIR_REVERSE : PISCES_DIR_FORWARD;
                    if (n_runs >= 0) {
                        // the base's index in the expanded CIGAR: every operation takes its length there, the read-span ones hold the bases
                        int e = 0, ri = 0, ex = -1;
                        for (int c = 0; c < n_cigar && ex < 0; c++) {
                            const uint32_t v = (uint32_t)bam_le32(cig + 4 * c);
                            const uint32_t op = v & 0xFu;
                            const int len = (int)(v >> 4);
                            if (op == 0 || op == 1 || op == 4 || op == 7 || op == 8) {
                                if (k < ri + len) ex = e + (k - ri);
                                ri += len;
                            }
                            e += len;
                        }
                        d = ex >= 0 ? direction_at(ex) : (uint8_t)PISCES_DIR_FORWARD;
                    }
                    dirs[b + k] = d;
                }
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
                const long long pos1 = (long long)bam_le32(rec + 4) + 1;
                int code = pos1 <= 0 ? kBamReadPositionNotPositive : 0;
                auto touch = [&](long long from, long long to) {   // positions [from, to] receive counts
                    if (to < 1) return;
                    if (from < 1) from = 1;
                    for (long long k = (from - 1) / F.block_size; k <= (to - 1) / F.block_size; k++) {
                        if (k >= n_block_bits) { code = code ? code : kBamReadPastBlockMap; return; }
                        if ((k >> 5) != map_word) {
                            if (map_bits) atomicOr(block_map + map_word, map_bits);
                            map_word = k >> 5;
                            map_bits = 0;
                        }
                        map_bits |= 1u << (k & 31);
                    }
                };
                int ri = 0, expanded = 0;
                long long rp = pos1, last_mapped = pos1 - 1;
                uint8_t ok_last = 0;
                uint32_t op_last = 99, len_last = 0, op_before = 99, len_before = 0;
                for (int k = 0; k < n_cigar; k++) {
                    const uint32_t v = (uint32_t)bam_le32(cig + 4 * k);
                    const uint32_t op = v & 0xFu, len = v >> 4;
                    const uint8_t letter = op == 0 ? 'M' : op == 1 ? 'I' : op == 2 ? 'D' : op == 3 ? 'N' : op == 4 ? 'S' : op == 5 ? 'H'
                                           : op == 6 ? 'P' : op == 7 ? '=' : op == 8 ? 'X' : '?';
                    cigar_op[o + k] = letter;
                    cigar_len[o + k] = len;
                    if (del_dirs) {   // a deletion's first and last deleted base (GetDeletionDirectionForStitchedRead, CandidateVariantFinder.cs:468-487)
                        const bool tracked = n_runs >= 0 && op == 2 && len > 0;
                        del_dirs[2 * (size_t)(o + k)] = tracked ? direction_at(expanded) : (uint8_t)PISCES_DIR_UNTRACKED;
                        del_dirs[2 * (size_t)(o + k) + 1] = tracked ? direction_at(expanded + (int)len - 1) : (uint8_t)PISCES_DIR_UNTRACKED;
                    }
                    expanded += (int)len;
                    uint8_t ok = 0;
                    if (l_seq > 0) {
                        const int after = ri < l_seq ? ql[ri] : ql[l_seq - 1], before = ri > 0 ? ql[min(ri, l_seq) - 1] : after;
                        ok = (before >= F.min_base_quality && after >= F.min_base_quality) ? 1 : 0;
                    }
                    op_quality[o + k] = ok;
                    const bool on_read = op == 0 || op == 1 || op == 4 || op == 7 || op == 8, on_ref = op == 0 || op == 2 || op == 3 || op == 7 || op == 8;
                    if (!code && on_read && on_ref && len > 0) {
                        if (rp > last_mapped + 1 && ri < l_seq && ok) touch(last_mapped + 1, rp - 1);   // the gap before this segment
                        touch(rp, rp + (long long)len - 1);
                        last_mapped = rp + (long long)len - 1;
                    }
                    if (on_ref) rp += (long long)len;
                    if (on_read) ri += (int)len;
                    op_before = op_last; len_before = len_last;
                    op_last = op; len_last = len; ok_last = ok;
                }
                if (!code && ri != l_seq) code = kBamReadCigarLongerThanRead;   // Read.ValidateCigar (Read.cs:603): the CIGAR's read span is the sequence length
                if (!code && rp > 0x7FFFFFFFll) code = kBamReadPastInt32;
                if (!code) {
                    // a terminal deletion counts at the anchor of the read's end (RegionStateManager.cs:195-210), a deletion before a
                    // terminal soft clip likewise
                    const int lastq = l_seq > 0 ? ql[l_seq - 1] : 0, prevq = l_seq > 1 ? ql[l_seq - 2] : lastq;
                    const bool end_ok = l_seq > 0 && lastq >= F.min_base_quality && prevq >= F.min_base_quality;
                    if (op_last == 2 && l_seq > 0 && end_ok) touch(last_mapped + 1, last_mapped + (long long)len_last);
                    if (n_cigar >= 2 && op_before == 2 && op_last == 4) {
                        const int idx = l_seq - (int)len_last;
                        if (idx >= 0 && idx < l_seq && ok_last) touch(last_mapped + 1, last_mapped + (long long)len_before);
                    }
                }
                if (code) atomicMin(first_error, (unsigned long long)r * 8ull + (unsigned long long)code);
                slots[r] = slot;
                fslots[r] = fslot;
            }
        }
    }
    if (lane == 0 && map_bits) atomicOr(block_map + map_word, map_bits);
}

}  // namespace pisces
