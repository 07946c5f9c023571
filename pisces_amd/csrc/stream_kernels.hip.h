// stream_kernels.hip.h — device side of the STREAMING surface (pisces_hip_add_reads / pisces_hip_flush), gfx950.
//
// the read walk of IStateManager.AddAlleleCounts (RegionStateManager.cs:118-220) runs on the
// device, so that what crosses PCIe is the read (1 byte of base + 1 byte of quality per aligned base) and not a 4-byte
// tuple per observation expanded by a host loop.
//
//   expand_reads_kernel   reads (SoA, device copy)       -> observation log  (position, tuple) appended in HBM
//   bucket_count_kernel   log entries of flushed blocks  -> per-tile counts
//   bucket_scan_kernel    per-tile counts                 -> PiscesTile::tuple_begin/end (segments padded to x4), cursors
//   bucket_scatter_kernel log entries                     -> the tile-bucketed tuple buffer the hot kernel streams
//   log_drop_kernel       DoneProcessing: keep the entries of blocks that were not flushed
//
// The order of tuples inside a tile is not defined (atomics) and does not matter: every consumer is a sum of counts.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

#include "../../include/pisces_hip.h"
#include "read_walk.h"

namespace pisces {

struct DevReadBatch {
    const int32_t* position;
    const uint8_t* flags;
    const int32_t* cigar_offset;
    const uint8_t* cigar_op;
    const uint32_t* cigar_len;
    const int32_t* seq_offset;
    const uint8_t* bases;
    const uint8_t* quals;
    const uint8_t* dirs;   // optional
    int32_t n_reads;
};

__device__ __forceinline__ int wave_exclusive_scan(int v, int lane, int* total)
{
    int incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(incl, d, 64);
        if (lane >= d) incl += o;
    }
    *total = __shfl(incl, 63, 64);
    return incl - v;
}

// One wave per read, lane = base within a chunk of 64 bases: every lane asks read_walk.h what its base adds (the same function the
// host form loops over), a wave scan turns the counts into log slots.  Positions <= 0 are not logged.
// Every read owns the log slots [read_slot[r], read_slot[r + 1]) reserved by the host from its CIGAR (mapped bases + gap
// lengths, an upper bound that is exact unless a deletion fails its quality test): no atomics on the log — same-address
// global atomics from 8 XCDs cost ~25-200 ns EACH and were the whole run time of the first version of this kernel.
// Slots a read does not use are written as position 0 ("hole"), which every consumer skips.
// (read_slot[r] + slot_base = the read's first slot in the log: a batch decoded on the device carries slots counted from 0)
// A wave takes reads_per_wave reads one after the other (expand_reads_grid: 1 for a block's worth of reads, more for large batches, so
// that the one statistics atomic a workgroup makes at its end -- same address for all of them -- stays a few thousand a launch: at one
// per four reads it was the whole run time of a 400 000-read launch).
__global__ __launch_bounds__(256) void expand_reads_kernel(DevReadBatch b, const long long* __restrict__ read_slot, long long slot_base, int32_t min_bq,
                                                           int32_t* __restrict__ log_pos, uint32_t* __restrict__ log_tup,
                                                           unsigned long long* __restrict__ appended, int32_t reads_per_wave)
{
    __shared__ unsigned int s_emitted[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int emitted = 0;
    for (int rr = 0; rr < reads_per_wave; rr++) {
    const int r = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * 4 + wave) * reads_per_wave + rr));
    if (r < b.n_reads) {
        const int c0 = b.cigar_offset[r], s0 = b.seq_offset[r];
        const ReadShape shape = read_shape(b.position[r], b.seq_offset[r + 1] - s0, b.cigar_offset[r + 1] - c0, b.cigar_op + c0, b.cigar_len + c0);
        const int n = shape.n;
        const uint32_t read_dir = (b.flags[r] & 1) ? PISCES_DIR_REVERSE : PISCES_DIR_FORWARD;
        const uint8_t* const quals = b.quals + s0;
        const uint8_t* const bases = b.bases + s0;
        const uint8_t* const dirs = b.dirs ? b.dirs + s0 : nullptr;
        const long long slot0 = read_slot[r] + slot_base, slot1 = read_slot[r + 1] + slot_base;
        const uint32_t lastAnchor = PISCES_NUM_ANCHORS - 1;
        if (shape.nc == 1 && walk_op_read_span(shape.ops[0]) && walk_op_ref_span(shape.ops[0]) && (int)shape.lens[0] == n && shape.pos0 >= 1 &&
            slot1 - slot0 >= n) {
            // one aligned segment and nothing else (most reads): base i sits on position pos0 + i and closes no gap, so it owns slot i
            // of the read -- no walk over the CIGAR per base, no scan
            for (int i = lane; i < n; i += 64) {
                const int p = shape.pos0 + i;
                const uint32_t dir = dirs ? (uint32_t)dirs[i] : read_dir;
                log_pos[slot0 + i] = p;
                log_tup[slot0 + i] = PISCES_TUPLE_PACK(0, (uint32_t)walk_anchor_type(shape.alignment_end, p, shape.pos0), dir, walk_allele_type(bases[i]),
                                                       (uint32_t)quals[i]);
                emitted++;
            }
            for (long long w = slot0 + n + lane; w < slot1; w += 64) log_pos[w] = 0;   // holes (none, by the host's bound)
        } else {

        long long w0 = slot0;   // next free slot of this read (wave-uniform)
        for (int base0 = 0; base0 < n; base0 += 64) {
            const int i = base0 + lane;
            const bool active = i < n;
            BaseWalk bw;
            bw.position = -1; bw.anchor = 0;
            bw.n_soft = bw.n_gap = bw.n_base = bw.n_end = 0;
            bw.soft_first = bw.gap_first = bw.end_first = 0;
            if (active) bw = walk_base(shape, i, quals, min_bq);
            const uint32_t dir = active ? (dirs ? (uint32_t)dirs[i] : read_dir) : 0u;
            const int cnt = bw.n_soft + bw.n_gap + bw.n_base + bw.n_end;
            int total;
            const int excl = wave_exclusive_scan(cnt, lane, &total);
            if (total == 0) continue;
            long long w = w0 + excl;
            const bool fits = w0 + total <= slot1;   // always, by the host's bound; never write outside the read's slots
            w0 += total;
            if (!fits) continue;
            emitted += cnt;
            for (int k = 0; k < bw.n_soft; k++, w++) {
                log_pos[w] = bw.soft_first + k;
                log_tup[w] = PISCES_TUPLE_PACK(0, lastAnchor, dir, PISCES_ALLELE_DEL, 255);
            }
            if (bw.position != -1) {
                const uint32_t anchor = (uint32_t)(bw.anchor < 0 ? 0 : bw.anchor);   // (inside the alignment the anchor is never negative)
                for (int k = 0; k < bw.n_gap; k++, w++) {
                    log_pos[w] = bw.gap_first + k;
                    log_tup[w] = PISCES_TUPLE_PACK(0, anchor, dir, PISCES_ALLELE_DEL, 255);
                }
                if (bw.n_base) {
                    log_pos[w] = bw.position;
                    log_tup[w] = PISCES_TUPLE_PACK(0, anchor, dir, walk_allele_type(bases[i]), (uint32_t)quals[i]);
                    w++;
                }
            }
            for (int k = 0; k < bw.n_end; k++, w++) {
                log_pos[w] = bw.end_first + k;
                log_tup[w] = PISCES_TUPLE_PACK(0, lastAnchor, dir, PISCES_ALLELE_DEL, 255);
            }
        }
        for (long long w = min(w0, slot1) + lane; w < slot1; w += 64) log_pos[w] = 0;   // holes
        }
    }
    }
    // observations made (IStateManager statistics): one global atomic per workgroup
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) emitted += __shfl_xor(emitted, d, 64);
    if (lane == 0) s_emitted[wave] = (unsigned int)emitted;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned int t = s_emitted[0] + s_emitted[1] + s_emitted[2] + s_emitted[3];
        if (t) atomicAdd(appended, (unsigned long long)t);
    }
}

// reads a wave takes / workgroups of a launch over n_reads reads
inline int32_t expand_reads_per_wave(int64_t n_reads) { return (int32_t)std::min<int64_t>(16, std::max<int64_t>(1, n_reads / 16384)); }
inline unsigned expand_reads_grid(int64_t n_reads)
{
    const int64_t per_group = 4 * (int64_t)expand_reads_per_wave(n_reads);
    return (unsigned)((n_reads + per_group - 1) / per_group);
}

// ---- bucketing of the log by tile --------------------------------------------------------------------------------
struct BucketMap {
    const int32_t* key_slot;        // [key_max - key_min + 1] -> slot of a block being bucketed, or -1
    const int32_t* key_first_tile;  // [key_max - key_min + 1] -> first tile of that block, or -1 (key_slot and first_tile in one look-up)
    const int32_t* first_tile;      // [slots] first tile of the block
    const int32_t* tile_of_locus;   // [slots * block_size] tile index relative to first_tile, or -1; nullptr = regular 64-locus grid
    int32_t key_min, key_max, block_size;
    double inv_block_size;          // 1 / block_size: the block of a position without an integer division per log entry
};

// GetBlockKey (RegionStateManager.cs:385-391): ceil(pos / block_size) for pos >= 1, exactly (the product is corrected by its remainder)
__device__ __forceinline__ int32_t bucket_key_of(const BucketMap& m, int32_t pos)
{
    const long long n = (long long)pos + m.block_size - 1;   // (64 bits: positions reach 2^31 - 1)
    int32_t q = (int32_t)((double)n * m.inv_block_size);
    const long long r = n - (long long)q * m.block_size;
    q += r >= m.block_size ? 1 : r < 0 ? -1 : 0;
    return q;
}

// tile of a position, or -1 (block not in this bucketing, or locus outside the interval set)
__device__ __forceinline__ int32_t bucket_tile_of(const BucketMap& m, int32_t pos)
{
    if (pos <= 0) return -1;
    const int32_t key = bucket_key_of(m, pos);
    if (key < m.key_min || key > m.key_max) return -1;
    const int32_t off = pos - ((key - 1) * m.block_size + 1);
    if (!m.tile_of_locus) {
        const int32_t first = m.key_first_tile[key - m.key_min];
        return first < 0 ? -1 : first + off / 64;
    }
    const int32_t slot = m.key_slot[key - m.key_min];
    if (slot < 0) return -1;
    const int32_t rel = m.tile_of_locus[(int64_t)slot * m.block_size + off];
    return rel < 0 ? -1 : m.first_tile[slot] + rel;
}

// A workgroup owns kLogChunk consecutive log entries.  Consecutive entries are a read's run of loci, i.e. a handful of tiles,
// so counts and fill cursors are privatized in LDS over the tile range [tmin, tmin + kLocalTiles) of the chunk and the
// global counters see one atomic per (workgroup, tile) instead of one per entry or per wave (same-address global atomics
// across XCDs serialize at ~0.1-0.2 us each on this part).
#ifndef PISCES_LOG_CHUNK
#define PISCES_LOG_CHUNK 4096
#endif
constexpr int kLogChunk = PISCES_LOG_CHUNK;
constexpr int kLogPerThread = kLogChunk / 256;
constexpr int kLocalTiles = 1024;

__device__ __forceinline__ int chunk_min_tile(const int32_t (&ti)[kLogPerThread], int* s_tmin)
{
    int tmin = 0x7FFFFFFF;
#pragma unroll
    for (int k = 0; k < kLogPerThread; k++)
        if (ti[k] >= 0) tmin = min(tmin, ti[k]);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) tmin = min(tmin, __shfl_xor(tmin, d, 64));
    if ((threadIdx.x & 63) == 0) atomicMin(s_tmin, tmin);
    __syncthreads();
    return *s_tmin;
}

__global__ __launch_bounds__(256) void bucket_count_kernel(const int32_t* __restrict__ log_pos, long long n, BucketMap m,
                                                           unsigned int* __restrict__ tile_count)
{
    __shared__ unsigned int s_cnt[kLocalTiles];
    __shared__ int s_tmin;
    const long long start = (long long)blockIdx.x * kLogChunk;
    if (start >= n) return;
    if (threadIdx.x == 0) s_tmin = 0x7FFFFFFF;
    for (int d = threadIdx.x; d < kLocalTiles; d += 256) s_cnt[d] = 0;
    __syncthreads();
    int32_t ti[kLogPerThread];
#pragma unroll
    for (int k = 0; k < kLogPerThread; k++) {
        const long long i = start + (long long)k * 256 + threadIdx.x;
        ti[k] = i < n ? bucket_tile_of(m, log_pos[i]) : -1;
    }
    const int tmin = chunk_min_tile(ti, &s_tmin);
    if (tmin == 0x7FFFFFFF) return;
#pragma unroll
    for (int k = 0; k < kLogPerThread; k++) {
        if (ti[k] < 0) continue;
        const int d = ti[k] - tmin;
        if (d < kLocalTiles) atomicAdd(&s_cnt[d], 1u);
        else atomicAdd(&tile_count[ti[k]], 1u);
    }
    __syncthreads();
    for (int d = threadIdx.x; d < kLocalTiles; d += 256)
        if (s_cnt[d]) atomicAdd(&tile_count[tmin + d], s_cnt[d]);
}

// after a drop whose count the host does not wait for: the slots [*kept, bound) of the new log become holes, so that `bound` (which
// the host knows) can be the log's length
__global__ __launch_bounds__(256) void log_fill_holes_kernel(int32_t* __restrict__ log_pos, const unsigned long long* __restrict__ kept, long long bound)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < bound && i >= (long long)*kept) log_pos[i] = 0;
}

// one workgroup: exclusive scan of the padded tile counts; fills the tile segments and resets the counters to cursors
// (zero_me: a counter a later kernel of the same submission starts from — log_drop_kernel's kept count — cleared here instead of by
// a fill of its own: every operation of a flush costs ~4.5 us of stream time whatever its size)
__global__ __launch_bounds__(1024) void bucket_scan_kernel(PiscesTile* __restrict__ tiles, int32_t n_tiles, unsigned int* __restrict__ tile_count,
                                                           long long* __restrict__ total_out, unsigned long long* __restrict__ zero_me)
{
    if (threadIdx.x == 0 && zero_me) *zero_me = 0ull;
    __shared__ long long s_part[1024];
    const int tid = threadIdx.x;
    const int per = (n_tiles + 1023) / 1024;
    const int b = tid * per, e = min(n_tiles, b + per);
    long long sum = 0;
    for (int t = b; t < e; t++) sum += ((long long)tile_count[t] + 3) & ~3ll;
    s_part[tid] = sum;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        long long v = tid >= d ? s_part[tid - d] : 0;
        __syncthreads();
        s_part[tid] += v;
        __syncthreads();
    }
    long long cursor = s_part[tid] - sum;
    for (int t = b; t < e; t++) {
        const long long c = tile_count[t];
        tiles[t].tuple_begin = cursor;
        tiles[t].tuple_end = cursor + c;
        cursor += (c + 3) & ~3ll;
        tile_count[t] = 0;   // reused as the fill cursor of the scatter
    }
    if (tid == 1023) *total_out = s_part[1023];
}

__global__ __launch_bounds__(256) void bucket_scatter_kernel(const int32_t* __restrict__ log_pos, const uint32_t* __restrict__ log_tup,
                                                             long long n, BucketMap m, const PiscesTile* __restrict__ tiles,
                                                             unsigned int* __restrict__ tile_fill, uint32_t* __restrict__ tuples)
{
    __shared__ unsigned int s_cnt[kLocalTiles];
    __shared__ long long s_dst[kLocalTiles];    // tuples index of this workgroup's first entry of the tile
    __shared__ int s_start[kLocalTiles];        // start_position of the tile
    __shared__ int s_tmin;
    const long long start = (long long)blockIdx.x * kLogChunk;
    if (start >= n) return;
    if (threadIdx.x == 0) s_tmin = 0x7FFFFFFF;
    for (int d = threadIdx.x; d < kLocalTiles; d += 256) s_cnt[d] = 0;
    __syncthreads();
    int32_t ti[kLogPerThread], pos[kLogPerThread];
    uint32_t tup[kLogPerThread];
#pragma unroll
    for (int k = 0; k < kLogPerThread; k++) {
        const long long i = start + (long long)k * 256 + threadIdx.x;
        pos[k] = i < n ? log_pos[i] : 0;
        tup[k] = i < n ? log_tup[i] : 0u;
        ti[k] = i < n ? bucket_tile_of(m, pos[k]) : -1;
    }
    const int tmin = chunk_min_tile(ti, &s_tmin);
    if (tmin == 0x7FFFFFFF) return;
#pragma unroll
    for (int k = 0; k < kLogPerThread; k++)
        if (ti[k] >= 0 && ti[k] - tmin < kLocalTiles) atomicAdd(&s_cnt[ti[k] - tmin], 1u);
    __syncthreads();
    for (int d = threadIdx.x; d < kLocalTiles; d += 256) {
        if (s_cnt[d]) {
            const PiscesTile t = tiles[tmin + d];
            s_dst[d] = t.tuple_begin + (long long)atomicAdd(&tile_fill[tmin + d], s_cnt[d]);
            s_start[d] = t.start_position;
            s_cnt[d] = 0;   // now the local cursor
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kLogPerThread; k++) {
        if (ti[k] < 0) continue;
        const int d = ti[k] - tmin;
        long long dst;
        int start_position;
        if (d < kLocalTiles) {
            dst = s_dst[d] + (long long)atomicAdd(&s_cnt[d], 1u);
            start_position = s_start[d];
        } else {
            const PiscesTile t = tiles[ti[k]];
            dst = t.tuple_begin + (long long)atomicAdd(&tile_fill[ti[k]], 1u);
            start_position = t.start_position;
        }
        tuples[dst] = PISCES_TUPLE_WITH_LOCUS(tup[k], (uint32_t)(pos[k] - start_position));
    }
}

// DoneProcessing (RegionStateManager.cs:336-353): entries of the blocks in `m` and holes are dropped, the rest is kept
// (one global atomic per workgroup; the order of the kept entries across workgroups is not defined)
__global__ __launch_bounds__(256) void log_drop_kernel(const int32_t* __restrict__ log_pos, const uint32_t* __restrict__ log_tup, long long n,
                                                       BucketMap m, int32_t* __restrict__ out_pos, uint32_t* __restrict__ out_tup,
                                                       unsigned long long* __restrict__ out_n)
{
    __shared__ unsigned int s_wave[4];
    __shared__ unsigned long long s_base;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long start = (long long)blockIdx.x * kLogChunk;
    if (start >= n) return;
    int32_t pos[kLogPerThread];
    uint32_t tup[kLogPerThread];
    unsigned int keep_bits = 0, mine = 0;
#pragma unroll
    for (int k = 0; k < kLogPerThread; k++) {
        const long long i = start + (long long)k * 256 + threadIdx.x;
        pos[k] = i < n ? log_pos[i] : 0;
        tup[k] = i < n ? log_tup[i] : 0u;
        bool keep = false;
        if (pos[k] > 0) {
            const int32_t key = bucket_key_of(m, pos[k]);
            keep = !(key >= m.key_min && key <= m.key_max && m.key_slot[key - m.key_min] >= 0);
        }
        if (keep) { keep_bits |= 1u << k; mine++; }
    }
    int total;
    const int excl = wave_exclusive_scan((int)mine, lane, &total);
    if (lane == 0) s_wave[wave] = (unsigned int)total;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned int t = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
        s_base = t ? atomicAdd(out_n, (unsigned long long)t) : 0ull;
    }
    __syncthreads();
    unsigned long long w = s_base + (unsigned long long)excl;
    for (int q = 0; q < wave; q++) w += s_wave[q];
#pragma unroll
    for (int k = 0; k < kLogPerThread; k++) {
        if (keep_bits & (1u << k)) {
            out_pos[w] = pos[k];
            out_tup[w] = tup[k];
            w++;
        }
    }
}

}  // namespace pisces
