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

#include "../../include/pisces_hip.h"

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

__device__ __forceinline__ bool dev_op_ref_span(uint8_t t) { return t == 'M' || t == 'D' || t == 'N' || t == '=' || t == 'X'; }
__device__ __forceinline__ bool dev_op_read_span(uint8_t t) { return t == 'M' || t == 'I' || t == 'S' || t == '=' || t == 'X'; }

// AlleleHelper.GetAlleleType (AlleleHelper.cs:13-32)
__device__ __forceinline__ uint32_t dev_allele_type(uint8_t c)
{
    return c == 'A' ? PISCES_ALLELE_A : c == 'C' ? PISCES_ALLELE_C : c == 'G' ? PISCES_ALLELE_G : c == 'T' ? PISCES_ALLELE_T : PISCES_ALLELE_N;
}

// RegionStateManager.GetAnchorType :83-116 with numAnchorTypes = 5 (positions here always lie inside the read's span)
__device__ __forceinline__ uint32_t dev_anchor_type(int alignmentEnd, int basePosition, int alignmentStart)
{
    const int leftAnchor = basePosition - alignmentStart, rightAnchor = alignmentEnd - basePosition;
    if (leftAnchor >= rightAnchor) {
        if (rightAnchor >= PISCES_ANCHOR_SIZE) return PISCES_ANCHOR_SIZE;
        return (uint32_t)max(PISCES_NUM_ANCHORS - rightAnchor - 1, 0);
    }
    if (leftAnchor >= PISCES_ANCHOR_SIZE) return PISCES_ANCHOR_SIZE;
    return (uint32_t)max(leftAnchor, 0);
}

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

// One wave per read, lane = read index within a chunk of 64 bases.  Follows expander.cpp (the host form, checked against
// the oracle's AddAlleleCounts) observation for observation; positions <= 0 are not logged, as there.
__global__ __launch_bounds__(256) void expand_reads_kernel(DevReadBatch b, int32_t min_bq, int32_t* __restrict__ log_pos,
                                                           uint32_t* __restrict__ log_tup, unsigned long long* __restrict__ log_n,
                                                           unsigned long long log_cap, int32_t* __restrict__ overflow,
                                                           unsigned long long* __restrict__ appended)
{
    const int lane = threadIdx.x & 63;
    const int r = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
    if (r >= b.n_reads) return;
    const int pos0 = b.position[r];
    const int c0 = b.cigar_offset[r], nc = b.cigar_offset[r + 1] - c0;
    const int s0 = b.seq_offset[r], n = b.seq_offset[r + 1] - s0;
    const uint32_t read_dir = (b.flags[r] & 1) ? PISCES_DIR_REVERSE : PISCES_DIR_FORWARD;
    const uint8_t* const ops = b.cigar_op + c0;
    const uint32_t* const lens = b.cigar_len + c0;
    const uint8_t* const quals = b.quals + s0;
    const uint8_t* const bases = b.bases + s0;
    const uint8_t* const dirs = b.dirs ? b.dirs + s0 : nullptr;
    if (n <= 0) return;

    int refSpan = 0, lastMappedOverall = pos0 - 1;
    {
        int rp = pos0;
        for (int c = 0; c < nc; c++) {
            const uint8_t t = ops[c];
            const int len = (int)lens[c];
            if (dev_op_ref_span(t)) {
                refSpan += len;
                if (dev_op_read_span(t) && len > 0) lastMappedOverall = rp + len - 1;
                rp += len;
            }
        }
    }
    const bool endsInDel = nc >= 1 && ops[nc - 1] == 'D';
    const bool endsInDelSoft = nc >= 2 && ops[nc - 2] == 'D' && ops[nc - 1] == 'S';
    int delLen = 0, lengthBeforeDeletion = n;
    if (endsInDel || endsInDelSoft) {
        delLen = (int)(endsInDelSoft ? lens[nc - 2] : lens[nc - 1]);
        lengthBeforeDeletion = endsInDelSoft ? n - (int)lens[nc - 1] : n;
    }
    const int alignmentEnd = pos0 + refSpan - 1;
    auto delq_ok = [&](int i) {   // CandidateVariantFinder.CheckDeletionQuality (:294-320), i < n
        const int after = quals[i], before = i > 0 ? quals[i - 1] : after;
        return before >= min_bq && after >= min_bq;
    };

    for (int base0 = 0; base0 < n; base0 += 64) {
        const int i = base0 + lane;
        const bool active = i < n;
        // Read.UpdatePositionMap (Read.cs:535-562) for index i, plus the position of the last mapped base before it
        int p = -1, lp = pos0 - 1;
        {
            int ri = 0, rp = pos0, lastm = pos0 - 1;
            for (int c = 0; c < nc; c++) {
                const uint8_t t = ops[c];
                const int len = (int)lens[c];
                const bool rs = dev_op_read_span(t), fs = dev_op_ref_span(t);
                if (rs) {
                    if (active && i >= ri && i < ri + len) {
                        if (fs) { p = rp + (i - ri); lp = (i == ri) ? lastm : p - 1; }
                        else lp = lastm;
                    }
                    if (fs && len > 0) { lastm = rp + len - 1; rp += len; }
                    ri += len;
                } else if (fs) {
                    rp += len;
                }
            }
        }
        const uint32_t dir = active ? (dirs ? (uint32_t)dirs[i] : read_dir) : 0u;
        const bool dq = active && delq_ok(i);
        // what this lane emits, in the host walk's order: terminal deletion before a soft clip, gap deletions, the base,
        // terminal deletion at the read end
        int n_soft = 0, n_gap = 0, n_base = 0, n_end = 0;
        int soft_first = 0, gap_first = 0, end_first = 0;
        if (active) {
            if (endsInDelSoft && i == lengthBeforeDeletion && dq) {
                soft_first = max(lp + 1, 1);
                n_soft = max(0, lp + delLen - soft_first + 1);
            }
            if (p != -1) {
                if (dq) {
                    gap_first = max(lp + 1, 1);
                    n_gap = max(0, p - 1 - gap_first + 1);
                }
                n_base = p > 0 ? 1 : 0;
            }
            if (endsInDel && i == n - 1 && dq) {
                end_first = max(lastMappedOverall + 1, 1);
                n_end = max(0, lastMappedOverall + delLen - end_first + 1);
            }
        }
        const int cnt = n_soft + n_gap + n_base + n_end;
        int total;
        const int excl = wave_exclusive_scan(cnt, lane, &total);
        if (total == 0) continue;
        unsigned long long base = 0;
        if (lane == 0) {
            base = atomicAdd(log_n, (unsigned long long)total);
            atomicAdd(appended, (unsigned long long)total);
        }
        base = ((unsigned long long)__shfl((int)(base >> 32), 0, 64) << 32) | (unsigned int)__shfl((int)(base & 0xFFFFFFFFull), 0, 64);
        if (base + (unsigned long long)total > log_cap) {
            if (lane == 0) *overflow = 1;
            continue;
        }
        unsigned long long w = base + (unsigned long long)excl;
        const uint32_t lastAnchor = PISCES_NUM_ANCHORS - 1;
        for (int k = 0; k < n_soft; k++, w++) {
            log_pos[w] = soft_first + k;
            log_tup[w] = PISCES_TUPLE_PACK(0, lastAnchor, dir, PISCES_ALLELE_DEL, 255);
        }
        if (p != -1) {
            const uint32_t anchor = dev_anchor_type(alignmentEnd, p, pos0);
            for (int k = 0; k < n_gap; k++, w++) {
                log_pos[w] = gap_first + k;
                log_tup[w] = PISCES_TUPLE_PACK(0, anchor, dir, PISCES_ALLELE_DEL, 255);
            }
            if (n_base) {
                log_pos[w] = p;
                log_tup[w] = PISCES_TUPLE_PACK(0, anchor, dir, dev_allele_type(bases[i]), (uint32_t)quals[i]);
                w++;
            }
        }
        for (int k = 0; k < n_end; k++, w++) {
            log_pos[w] = end_first + k;
            log_tup[w] = PISCES_TUPLE_PACK(0, lastAnchor, dir, PISCES_ALLELE_DEL, 255);
        }
    }
}

// ---- bucketing of the log by tile --------------------------------------------------------------------------------
struct BucketMap {
    const int32_t* key_slot;      // [key_max - key_min + 1] -> slot of a block being bucketed, or -1
    const int32_t* first_tile;    // [slots] first tile of the block
    const int32_t* tile_of_locus; // [slots * block_size] tile index relative to first_tile, or -1; nullptr = regular 64-locus grid
    int32_t key_min, key_max, block_size;
};

// tile of a position, or -1 (block not in this bucketing, or locus outside the interval set)
__device__ __forceinline__ int32_t bucket_tile_of(const BucketMap& m, int32_t pos)
{
    const int32_t key = (pos + m.block_size - 1) / m.block_size;   // GetBlockKey (RegionStateManager.cs:385-391)
    if (pos <= 0 || key < m.key_min || key > m.key_max) return -1;
    const int32_t slot = m.key_slot[key - m.key_min];
    if (slot < 0) return -1;
    const int32_t off = pos - ((key - 1) * m.block_size + 1);
    if (!m.tile_of_locus) return m.first_tile[slot] + off / 64;
    const int32_t rel = m.tile_of_locus[(int64_t)slot * m.block_size + off];
    return rel < 0 ? -1 : m.first_tile[slot] + rel;
}

// Consecutive log entries mostly fall into the same tile (a read's run of loci): one atomic per run of equal tiles in a
// wave instead of one per entry.  Returns the reserved base for this lane's run (valid in every lane of the run) and the
// lane's offset inside it.
__device__ __forceinline__ int64_t reserve_runs(int32_t ti, int lane, unsigned int* counters, int* offset_in_run)
{
    const int32_t prev = __shfl_up(ti, 1, 64);
    const bool head = lane == 0 || prev != ti;
    const unsigned long long heads = __ballot(head);
    // head lane of this lane's run = highest set bit of heads at or below lane
    const unsigned long long below = heads & ((lane == 63) ? ~0ull : ((1ull << (lane + 1)) - 1ull));
    const int head_lane = 63 - __clzll(below);
    // run length = distance from the head to the next head (or 64)
    const unsigned long long above = heads & ~((head_lane == 63) ? ~0ull : ((1ull << (head_lane + 1)) - 1ull));
    const int next_head = above ? __ffsll((long long)above) - 1 : 64;
    const int run_len = next_head - head_lane;
    unsigned int base = 0;
    if (head && ti >= 0) base = atomicAdd(&counters[ti], (unsigned int)run_len);
    base = (unsigned int)__shfl((int)base, head_lane, 64);
    *offset_in_run = lane - head_lane;
    return (int64_t)base;
}

__global__ __launch_bounds__(256) void bucket_count_kernel(const int32_t* __restrict__ log_pos, const unsigned long long* __restrict__ log_n,
                                                           BucketMap m, unsigned int* __restrict__ tile_count)
{
    const int lane = threadIdx.x & 63;
    const unsigned long long n = *log_n;
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    const unsigned long long first = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    for (unsigned long long base = first - lane; base < n; base += stride) {   // whole waves stay in the loop together
        const unsigned long long i = base + lane;
        const int32_t ti = i < n ? bucket_tile_of(m, log_pos[i]) : -1;
        int off;
        (void)reserve_runs(ti, lane, tile_count, &off);
    }
}

// one workgroup: exclusive scan of the padded tile counts; fills the tile segments and resets the counters to cursors
__global__ __launch_bounds__(1024) void bucket_scan_kernel(PiscesTile* __restrict__ tiles, int32_t n_tiles, unsigned int* __restrict__ tile_count,
                                                           long long* __restrict__ total_out)
{
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
                                                             const unsigned long long* __restrict__ log_n, BucketMap m,
                                                             const PiscesTile* __restrict__ tiles, unsigned int* __restrict__ tile_fill,
                                                             uint32_t* __restrict__ tuples)
{
    const int lane = threadIdx.x & 63;
    const unsigned long long n = *log_n;
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    const unsigned long long first = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    for (unsigned long long base = first - lane; base < n; base += stride) {
        const unsigned long long i = base + lane;
        int32_t pos = 0;
        uint32_t tup = 0;
        int32_t ti = -1;
        if (i < n) {
            pos = log_pos[i];
            tup = log_tup[i];
            ti = bucket_tile_of(m, pos);
        }
        int off;
        const int64_t run_base = reserve_runs(ti, lane, tile_fill, &off);
        if (ti >= 0) {
            const PiscesTile t = tiles[ti];
            const uint32_t locus = (uint32_t)(pos - t.start_position);
            tuples[t.tuple_begin + run_base + off] = (tup & ~0x7FFFu) | locus;
        }
    }
}

// DoneProcessing (RegionStateManager.cs:336-353): entries of the blocks in `m` are dropped, the rest is kept (order inside
// a wave is kept, across waves it is not defined)
__global__ __launch_bounds__(256) void log_drop_kernel(const int32_t* __restrict__ log_pos, const uint32_t* __restrict__ log_tup,
                                                       const unsigned long long* __restrict__ log_n, BucketMap m,
                                                       int32_t* __restrict__ out_pos, uint32_t* __restrict__ out_tup,
                                                       unsigned long long* __restrict__ out_n)
{
    const int lane = threadIdx.x & 63;
    const unsigned long long n = *log_n;
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    const unsigned long long first = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    for (unsigned long long base = first - lane; base < n; base += stride) {
        const unsigned long long i = base + lane;
        int32_t pos = 0;
        uint32_t tup = 0;
        bool keep = false;
        if (i < n) {
            pos = log_pos[i];
            tup = log_tup[i];
            const int32_t key = (pos + m.block_size - 1) / m.block_size;
            keep = !(key >= m.key_min && key <= m.key_max && m.key_slot[key - m.key_min] >= 0);
        }
        const unsigned long long mask = __ballot(keep);
        if (mask == 0ull) continue;
        unsigned long long wbase = 0;
        if (lane == 0) wbase = atomicAdd(out_n, (unsigned long long)__popcll(mask));
        wbase = ((unsigned long long)__shfl((int)(wbase >> 32), 0, 64) << 32) | (unsigned int)__shfl((int)(wbase & 0xFFFFFFFFull), 0, 64);
        if (keep) {
            const int rank = __popcll(mask & ((1ull << lane) - 1ull));
            out_pos[wbase + rank] = pos;
            out_tup[wbase + rank] = tup;
        }
    }
}

}  // namespace pisces
