// store_kernels.hip.h — the READ STORE of the streaming surface and the kernels that call straight from it (gfx950).
//
// FindCandidates + AddAlleleCounts as one pass over the reads.  pisces_hip_add_reads /
// pisces_hip_add_decoded_reads leave the reads where they are in HBM (bases, qualities, CIGARs: 2 bytes per aligned base) and make a
// 16-byte descriptor per read; a flush walks, per tile, the reads that overlap the tile and adds their bases straight into the LDS
// histogram the call phase reads (IStateManager.AddAlleleCounts, RegionStateManager.cs:118-220, then IAlleleCaller.Call): no
// observation log (8 bytes per observation written, then read three more times), no bucketing passes.
//
//   read_shape_kernel          add time: one lane per read, CIGAR -> ReadDesc / ReadExt, sortedness and longest reach of the segment
//   segment_copy_kernel        add time, small batches only: the batch's bytes appended to the open segment
//   segment_fill_dirs_kernel   a batch without per-base directions joining a segment that tracks them
//   call_store_tiles_kernel    THE FLUSH: per tile, reads (+ bucketed log tuples of pisces_hip_add_observations, if any) -> LDS
//                              histogram -> call_phase_wave (kernels.hip.h): HBM traffic ~ 2 B / observation + 16 B / read and tile + 64 B / record
//   accumulate_store_tiles_kernel  the same walk into the anchor-resolved tensor int32[locus][6][3][11] (+ base-quality sums) for the
//                              candidate kernel, the collapser, NoiseModel.Window and IAlleleSource.GetAlleleCount
//
// Reads of a segment are in position order (a BAM is); a tile's reads are then the index range [first read that can still reach the
// tile, first read that starts behind it), found by a 64-ary search over the descriptors (every lane probes one: 2-4 dependent loads).
// A segment that turned out not to be sorted (state[0]) is scanned in whole: slow, still exact.
// Lane = locus: for a read of one aligned run (soft clips allowed: most reads) lane l takes the base on locus l of the tile, one byte
// load each for base and quality, one ds_add — 64 consecutive bytes per load instruction, no two lanes on one LDS bank.  Reads with
// insertions, deletions or skips go through read_walk.h's per-base function (the walk the host form and the log path use).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.hip.h"
#include "read_walk.h"
#include "stream_kernels.hip.h"

namespace pisces {

struct ReadDesc {       // 16 bytes, one per read, in position order
    int32_t pos0;       // Read.Position
    uint32_t meta;      // kDesc*: aligned bases | flags
    int64_t aoff;       // index of the read's first ALIGNED base in the segment's bases / quals / dirs (complex reads: of its first base)
};
struct ReadExt {        // what only the general walk needs
    int64_t cig_off;    // first CIGAR operation in the segment's cigar_op / cigar_len
    int32_t n_cigar;
    int32_t n_bases;
};
static_assert(sizeof(ReadDesc) == 16 && sizeof(ReadExt) == 16, "descriptor layout");
constexpr uint32_t kDescLenMask = 0xFFFFFu;      // bases of the one aligned run (simple reads)
constexpr uint32_t kDescComplex = 1u << 20;      // insertions / deletions / skips / anything but clips around one aligned run: general walk
constexpr uint32_t kDescReverse = 1u << 21;      // flags bit 0 (direction of every base unless the segment tracks per-base directions)
constexpr int kMaxSegments = 8;
constexpr int kStateUnsorted = 0, kStateReach = 1;

struct SegmentView {
    const ReadDesc* desc;
    const ReadExt* ext;
    const uint8_t* bases;
    const uint8_t* quals;
    const uint8_t* dirs;          // per-base DirectionType of every read of the segment, or nullptr (direction = kDescReverse)
    const uint8_t* cigar_op;
    const uint32_t* cigar_len;
    const int32_t* state;         // [kStateUnsorted] != 0: not in position order; [kStateReach]: longest reference span of a read
    int32_t n_reads;
    int32_t n_floored;            // reads [0, n_floored) were there at the last flush: their positions below `floor` are counted already
    int32_t floor;
    int32_t pad;
};
struct StoreView {
    SegmentView seg[kMaxSegments];
    int32_t n_segments;
};

// ---- add time -------------------------------------------------------------------------------------------------------------------
struct ShapeArgs {
    // the batch as uploaded / decoded (offsets relative to the batch's own arrays)
    const int32_t* position;
    const uint8_t* flags;
    const int32_t* cigar_offset;
    const uint8_t* cigar_op;
    const uint32_t* cigar_len;
    const int32_t* seq_offset;
    int32_t n_reads;
    int32_t n0;            // index of the batch's first read in the segment
    int64_t base0, ops0;   // index of its first base / first CIGAR operation in the segment's arrays
    ReadDesc* desc;
    ReadExt* ext;
    int32_t* state;
};

__global__ __launch_bounds__(256) void read_shape_kernel(ShapeArgs A)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    int reach = 0;
    bool unsorted = false;
    if (r < A.n_reads) {
        const int c0 = A.cigar_offset[r], nc = A.cigar_offset[r + 1] - c0;
        const int s0 = A.seq_offset[r], n = A.seq_offset[r + 1] - s0;
        const int32_t pos0 = A.position[r];
        // one aligned run between clips?  phases: 0 leading clips, 1 the run (M = X), 2 trailing clips; H / P span nothing
        int phase = 0, lead = 0;
        long long run = 0, ref_span = 0;
        bool simple = true;
        for (int c = 0; c < nc; c++) {
            const uint8_t t = A.cigar_op[c0 + c];
            const long long len = A.cigar_len[c0 + c];
            if (walk_op_ref_span(t)) ref_span += len;
            if (t == 'M' || t == '=' || t == 'X') {
                if (phase == 2) simple = false;
                phase = 1;
                run += len;
            } else if (t == 'S') {
                if (phase == 0) lead += (int)len;
                else phase = 2;
            } else if (t == 'H' || t == 'P') {
                if (phase == 1 && t == 'H') phase = 2;
            } else {
                simple = false;
            }
        }
        if (run > (long long)kDescLenMask || run > n - lead) simple = false;   // (a CIGAR longer than the read is refused before it gets here)
        ReadDesc d;
        d.pos0 = pos0;
        d.meta = (simple ? (uint32_t)run : kDescComplex) | ((A.flags[r] & 1) ? kDescReverse : 0u);
        d.aoff = A.base0 + s0 + (simple ? lead : 0);
        A.desc[A.n0 + r] = d;
        ReadExt e;
        e.cig_off = A.ops0 + c0;
        e.n_cigar = nc;
        e.n_bases = n;
        A.ext[A.n0 + r] = e;
        reach = (int)(ref_span > 0x7FFFFFFFll ? 0x7FFFFFFFll : ref_span);
        // position order, the batch's first read against the read before it in the segment (written by an earlier launch)
        if (r > 0) unsorted = A.position[r - 1] > pos0;
        else if (A.n0 > 0) unsorted = A.desc[A.n0 - 1].pos0 > pos0;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) reach = max(reach, __shfl_xor(reach, d, 64));
    const bool any_unsorted = __ballot(unsorted) != 0ull;
    if ((threadIdx.x & 63) == 0) {
        if (reach > A.state[kStateReach]) atomicMax(&A.state[kStateReach], reach);
        if (any_unsorted) atomicOr(&A.state[kStateUnsorted], 1);
    }
}

// small batches: their bytes join the open segment (up to five ranges in one launch; byte-wise: destinations are not aligned)
struct CopyRanges {
    uint8_t* dst[5];
    const uint8_t* src[5];
    int64_t n[5];
};
__global__ __launch_bounds__(256) void segment_copy_kernel(CopyRanges C)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
#pragma unroll
    for (int k = 0; k < 5; k++)
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < C.n[k]; i += stride) C.dst[k][i] = C.src[k][i];
}

// per-base directions of reads [r0, r1) of a segment from their flags: a batch without directions joining a segment that tracks them,
// or the reads a segment held before its first batch with directions
__global__ __launch_bounds__(256) void segment_fill_dirs_kernel(const ReadDesc* __restrict__ desc, const ReadExt* __restrict__ ext, int32_t r0, int32_t r1,
                                                                uint8_t* __restrict__ dirs)
{
    const int r = r0 + (int)(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (r >= r1) return;
    const ReadDesc d = desc[r];
    const ReadExt e = ext[r];
    // (aoff of a simple read points at its first aligned base; the fill covers the aligned run, which is all the walk reads of it)
    const int n = (d.meta & kDescComplex) ? e.n_bases : (int)(d.meta & kDescLenMask);
    const uint8_t v = (d.meta & kDescReverse) ? PISCES_DIR_REVERSE : PISCES_DIR_FORWARD;
    for (int i = threadIdx.x & 63; i < n; i += 64) dirs[d.aoff + i] = v;
}

// ---- flush time -----------------------------------------------------------------------------------------------------------------
// first read of desc[0, n) whose position is >= x (n if none); desc in position order.  Every lane probes one descriptor a round.
__device__ __forceinline__ int wave_lower_bound(const ReadDesc* __restrict__ desc, int n, int x, int lane)
{
    int a = 0, b = n;   // the answer lies in [a, b]
    while (b > a) {
        const int step = (b - a + 63) >> 6;
        const long long idx = (long long)a + (long long)(lane + 1) * step - 1;
        const int v = idx < b ? desc[idx].pos0 : 0x7FFFFFFF;
        const int c = __popcll(__ballot(idx < b && v < x));   // a prefix of the lanes: probes below x
        const long long na = (long long)a + (long long)c * step;
        const long long nb = na + step - 1;                   // probe c is >= x (or past the end): the answer is at or before it
        a = (int)min(na, (long long)b);
        b = (int)max(min(nb, (long long)b), (long long)a);
    }
    return a;
}

__device__ __forceinline__ long long readlane64(long long v, int lane_index)
{
    const int lo = __builtin_amdgcn_readlane((int)(v & 0xFFFFFFFFll), lane_index);
    const int hi = __builtin_amdgcn_readlane((int)(v >> 32), lane_index);
    return ((long long)hi << 32) | (unsigned int)lo;
}

// The reads of one segment that can touch the tile [tile_start, tile_start + 64): simple reads eight at a time (their sixteen byte
// loads are in flight together), lane = locus.  on_base(k-th read of the group: base, quality, direction, valid, pos0, aligned bases).
// Then the complex reads of the same range, one at a time, through walk_base: on_obs(position, allele, direction, anchor, quality).
constexpr int kReadGroup = 8;
template <bool kDirs, typename OnBase, typename OnObs>
__device__ __forceinline__ void walk_segment(const SegmentView& G, int tile_start, int min_bq, int lane, int wid, int n_waves, OnBase on_base, OnObs on_obs)
{
    if (G.n_reads <= 0) return;
    const int tile_end = tile_start + kTile - 1;
    int lo = 0, hi = G.n_reads;
    if (G.state[kStateUnsorted] == 0) {
        const int reach = G.state[kStateReach];
        const long long x_lo = (long long)tile_start - reach + 1;
        lo = wave_lower_bound(G.desc, G.n_reads, (int)max(x_lo, -0x7FFFFFFFll), lane);
        hi = tile_end == 0x7FFFFFFF ? G.n_reads : wave_lower_bound(G.desc, G.n_reads, tile_end + 1, lane);
    }
    lo = __builtin_amdgcn_readfirstlane(lo);
    hi = __builtin_amdgcn_readfirstlane(hi);
    const int locus_pos = tile_start + lane;
    for (int base = lo + wid * 64; base < hi; base += n_waves * 64) {
        const int cnt = min(64, hi - base);
        const ReadDesc d = G.desc[base + min(lane, cnt - 1)];
        unsigned long long complex_mask = __ballot(lane < cnt && (d.meta & kDescComplex));
        for (int g0 = 0; g0 < cnt; g0 += kReadGroup) {
            uint32_t bb[kReadGroup], qq[kReadGroup], dd[kReadGroup];
            int p0[kReadGroup], nn[kReadGroup];
            uint32_t ok = 0;
#pragma unroll
            for (int k = 0; k < kReadGroup; k++) {
                const int u = min(g0 + k, cnt - 1);   // (a short last group repeats its last read with no bases)
                const int pos0 = __builtin_amdgcn_readlane(d.pos0, u);
                const uint32_t meta = (uint32_t)__builtin_amdgcn_readlane((int)d.meta, u);
                const long long aoff = readlane64(d.aoff, u);
                const int n = (g0 + k < cnt && !(meta & kDescComplex)) ? (int)(meta & kDescLenMask) : 0;
                const int floor_pos = base + u < G.n_floored ? G.floor : 0;
                const int i_min = max(floor_pos - pos0, 0);   // (floor <= 2^31 - 1, pos0 >= 1)
                const int i = locus_pos - pos0;
                if (i >= i_min && i < n) ok |= 1u << k;
                const uint32_t ic = (uint32_t)min(max(i, 0), max(n - 1, 0));
                bb[k] = (G.bases + aoff)[ic];
                qq[k] = (G.quals + aoff)[ic];
                dd[k] = kDirs ? (uint32_t)(G.dirs + aoff)[ic] : ((meta & kDescReverse) ? (uint32_t)PISCES_DIR_REVERSE : (uint32_t)PISCES_DIR_FORWARD);
                p0[k] = pos0;
                nn[k] = n;
            }
#pragma unroll
            for (int k = 0; k < kReadGroup; k++) on_base(bb[k], qq[k], dd[k], (ok >> k) & 1u, p0[k], nn[k]);
        }
        // the reads with insertions / deletions / skips: RegionStateManager.AddAlleleCounts base by base (read_walk.h)
        while (complex_mask) {
            const int u = __builtin_ctzll(complex_mask);
            complex_mask &= complex_mask - 1;
            const int r = base + u;
            const int pos0 = __builtin_amdgcn_readlane(d.pos0, u);
            const uint32_t meta = (uint32_t)__builtin_amdgcn_readlane((int)d.meta, u);
            const long long aoff = readlane64(d.aoff, u);
            const ReadExt e = G.ext[r];
            const ReadShape shape = read_shape(pos0, e.n_bases, e.n_cigar, G.cigar_op + e.cig_off, G.cigar_len + e.cig_off);
            if (pos0 > tile_end || (long long)pos0 + shape.ref_span - 1 < tile_start) continue;
            const int floor_pos = r < G.n_floored ? G.floor : 0;
            const int p_lo = max(tile_start, max(floor_pos, 1)), p_hi = tile_end;
            const uint8_t* const quals = G.quals + aoff;
            const uint8_t* const bases = G.bases + aoff;
            const uint32_t read_dir = (meta & kDescReverse) ? PISCES_DIR_REVERSE : PISCES_DIR_FORWARD;
            const int lastAnchor = PISCES_NUM_ANCHORS - 1;
            for (int base0 = 0; base0 < shape.n; base0 += 64) {
                const int i = base0 + lane;
                if (i >= shape.n) continue;
                const BaseWalk bw = walk_base(shape, i, quals, min_bq);
                const uint32_t dir = kDirs ? (uint32_t)(G.dirs + aoff)[i] : read_dir;
                auto deleted_run = [&](int first, int count, int anchor) {
                    const long long last = (long long)first + count - 1;
                    const int a = max(first, p_lo), b = (int)min(last, (long long)p_hi);
                    for (int p = a; p <= b; p++) on_obs(p, (uint32_t)PISCES_ALLELE_DEL, dir, anchor, 255u);
                };
                if (bw.n_soft) deleted_run(bw.soft_first, bw.n_soft, lastAnchor);
                if (bw.position != -1) {
                    const int anchor = bw.anchor < 0 ? 0 : bw.anchor;
                    if (bw.n_gap) deleted_run(bw.gap_first, bw.n_gap, anchor);
                    if (bw.n_base && bw.position >= p_lo && bw.position <= p_hi)
                        on_obs(bw.position, walk_allele_type(bases[i]), dir, anchor, (uint32_t)quals[i]);
                }
                if (bw.n_end) deleted_run(bw.end_first, bw.n_end, lastAnchor);
            }
        }
    }
}

template <typename OnBase, typename OnObs>
__device__ __forceinline__ void walk_store(const StoreView& S, int tile_start, int min_bq, int lane, int wid, int n_waves, OnBase on_base, OnObs on_obs)
{
    for (int s = 0; s < S.n_segments; s++) {
        const SegmentView& G = S.seg[s];
        if (G.dirs) walk_segment<true>(G, tile_start, min_bq, lane, wid, n_waves, on_base, on_obs);
        else walk_segment<false>(G, tile_start, min_bq, lane, wid, n_waves, on_base, on_obs);
    }
}

// consecutive tiles on one XCD (workgroups go round the 8 XCDs in dispatch order): neighbouring tiles read the same reads — their
// lines are then in that XCD's L2 the second time
__device__ __forceinline__ int xcd_tile_of_block(int b, int n)
{
    const int x = b & 7, j = b >> 3, q = n >> 3, rem = n & 7;
    return x * q + min(x, rem) + j;
}

// byte offset of an allele's first row in a histogram region, by read base (AlleleHelper.GetAlleleType, AlleleHelper.cs:13-32:
// anything but A, C, G, T is an N); row = allele * 4 + direction, 64 int32 columns a row
__device__ __forceinline__ uint32_t allele_row_bytes(uint32_t c)
{
    return (c == 'A' ? (uint32_t)PISCES_ALLELE_A : c == 'C' ? (uint32_t)PISCES_ALLELE_C : c == 'G' ? (uint32_t)PISCES_ALLELE_G
            : c == 'T' ? (uint32_t)PISCES_ALLELE_T : (uint32_t)PISCES_ALLELE_N) * (4u * kWaveRow * (uint32_t)sizeof(int));
}

template <int NW>
__global__ __launch_bounds__(64 * NW, NW == 1 ? PISCES_WAVE_OCC : PISCES_WAVE2_OCC) void call_store_tiles_kernel(
    StoreView S, const uint32_t* __restrict__ tuples, const PiscesTile* __restrict__ tiles, int32_t n_tiles, const uint8_t* __restrict__ ref,
    int32_t ref_start, int64_t ref_len, PiscesCalledAllele* __restrict__ records, PiscesTileResult* __restrict__ tile_results, DeviceParams P,
    const DeviceParams* __restrict__ Pd)
{
    __shared__ __attribute__((aligned(16))) int hist[2 * kWaveRegion];   // [region: quality-passing / low-quality][allele * 4 + direction][locus]
    __shared__ uint8_t s_refwin[kRefWin];
    __shared__ uint8_t s_vmask[kTile];
    __shared__ uint16_t s_row[256];   // allele_row_bytes by read base

    if ((int)blockIdx.x >= n_tiles) return;
    const int t = xcd_tile_of_block((int)blockIdx.x, n_tiles);
    const int l = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const PiscesTile tile = tiles[t];
    {
        int4* h4 = reinterpret_cast<int4*>(hist);
        for (int i = threadIdx.x; i < 2 * kWaveRegion / 4; i += 64 * NW) h4[i] = make_int4(0, 0, 0, 0);
        for (int i = threadIdx.x; i < kRefWin; i += 64 * NW) {
            const int64_t ri = (int64_t)tile.start_position - kRefMargin + i - ref_start;
            s_refwin[i] = (ri >= 0 && ri < ref_len) ? ref[ri] : (uint8_t)0;
        }
        for (int i = threadIdx.x; i < 256; i += 64 * NW) s_row[i] = (uint16_t)allele_row_bytes((uint32_t)i);
        __syncthreads();
    }
    const uint32_t min_bq = (uint32_t)min(max(P.min_bq, 0), 255);
    char* const hbytes = reinterpret_cast<char*>(hist);
    constexpr uint32_t kRegionBytes = (uint32_t)(kWaveRegion * sizeof(int));
    // pre-expanded observations (pisces_hip_add_observations), bucketed by tile: "qual < minBQ -> the low-quality region" as accumulate_wave
    for (int64_t i = tile.tuple_begin + threadIdx.x; i < tile.tuple_end; i += 64 * NW) {
        const uint32_t v = tuples[i];
        const uint32_t dir = PISCES_TUPLE_DIR(v), allele = PISCES_TUPLE_ALLELE(v);
        if (dir < 3u && allele < 6u)
            atomicAdd(reinterpret_cast<int*>(hbytes + ((PISCES_TUPLE_QUAL(v) < min_bq) ? kRegionBytes : 0u) +
                                             (uint32_t)HistLinear::idx((int)allele, (int)dir, (int)PISCES_TUPLE_LOCUS(v)) * (uint32_t)sizeof(int)), 1);
    }
    // the reads
    const uint32_t lane_bytes = (uint32_t)l * (uint32_t)sizeof(int);
    walk_store(S, tile.start_position, (int)min_bq, l, wid, NW,
               [&](uint32_t base, uint32_t qual, uint32_t dir, uint32_t valid, int, int) {
                   const uint32_t off = (uint32_t)s_row[base] + ((qual < min_bq) ? kRegionBytes : 0u) + dir * (uint32_t)(kWaveRow * sizeof(int)) + lane_bytes;
                   if (valid) atomicAdd(reinterpret_cast<int*>(hbytes + off), 1);
               },
               [&](int position, uint32_t allele, uint32_t dir, int, uint32_t qual) {
                   const uint32_t off = ((qual < min_bq) ? kRegionBytes : 0u) +
                                        (uint32_t)HistLinear::idx((int)allele, (int)dir, position - tile.start_position) * (uint32_t)sizeof(int);
                   atomicAdd(reinterpret_cast<int*>(hbytes + off), 1);
               });
    __syncthreads();
    call_phase_wave<NW, HistLinear>(hist, s_refwin, s_vmask, tile, t, l, wid, ref, ref_start, ref_len, records, tile_results, P
#ifdef PISCES_TIMING
                                    , 0ll, 0ll
#endif
                                    );
}

// The same walk into the anchor-resolved tensor (RegionState._alleleCounts, RegionState.cs:57) and, with sumq, the base-quality sums
// (RegionState._sumOfAlleleBaseQualities :61): what accumulate_tiles_kernel makes of tuples, here from the reads (and the tuples).
__global__ __launch_bounds__(kBlock) void accumulate_store_tiles_kernel(StoreView S, const uint32_t* __restrict__ tuples, const PiscesTile* __restrict__ tiles,
                                                                        int32_t n_tiles, int32_t* __restrict__ counts, int32_t min_bq_,
                                                                        unsigned long long* __restrict__ sumq, const ulonglong2* __restrict__ bq_lut)
{
    __shared__ int hist[kTile * kAnchStride];
    const int t = blockIdx.x;
    if (t >= n_tiles) return;
    const PiscesTile tile = tiles[t];
    for (int i = threadIdx.x; i < kTile * kAnchStride; i += kBlock) hist[i] = 0;
    __syncthreads();
    const uint32_t n_loci = (uint32_t)tile.n_loci, min_bq = (uint32_t)min_bq_;
    auto add = [&](uint32_t locus, uint32_t allele, uint32_t dir, uint32_t anchor, uint32_t qual) {
        if (allele < 4u && qual < min_bq) allele = 4u;   // RegionStateManager.cs:179-181
        if (locus < n_loci && dir < 3u && allele < 6u && anchor < (uint32_t)PISCES_NUM_ANCHORS) {
            atomicAdd(&hist[locus * kAnchStride + (allele * 3u + dir) * PISCES_NUM_ANCHORS + anchor], 1);
            if (sumq && allele < 4u) {
                const ulonglong2 q = bq_lut[qual];
                unsigned long long* cell = &sumq[2 * (((int64_t)t * kTile + locus) * PISCES_COUNTS_PER_LOCUS + (allele * 3u + dir) * PISCES_NUM_ANCHORS + anchor)];
                atomicAdd(cell, q.x);
                atomicAdd(cell + 1, q.y);
            }
        }
    };
    for (int64_t i = tile.tuple_begin + threadIdx.x; i < tile.tuple_end; i += kBlock) {
        const uint32_t v = tuples[i];
        add(PISCES_TUPLE_LOCUS(v), PISCES_TUPLE_ALLELE(v), PISCES_TUPLE_DIR(v), PISCES_TUPLE_ANCHOR(v), v >> 24);
    }
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    walk_store(S, tile.start_position, (int)min_bq, lane, wid, kBlock / 64,
               [&](uint32_t base, uint32_t qual, uint32_t dir, uint32_t valid, int pos0, int n_aligned) {
                   if (!valid) return;
                   // GetAnchorType (RegionStateManager.cs:83-116); EndPosition of a read of one aligned run = pos0 + run - 1
                   const int anchor = walk_anchor_type(pos0 + n_aligned - 1, tile.start_position + lane, pos0);
                   add((uint32_t)lane, walk_allele_type((uint8_t)base), dir, (uint32_t)(anchor < 0 ? 0 : anchor), qual);
               },
               [&](int position, uint32_t allele, uint32_t dir, int anchor, uint32_t qual) {
                   add((uint32_t)(position - tile.start_position), allele, dir, (uint32_t)anchor, qual);
               });
    __syncthreads();
    int32_t* __restrict__ dst = counts + (int64_t)t * kTile * PISCES_COUNTS_PER_LOCUS;
    const int n = tile.n_loci * PISCES_COUNTS_PER_LOCUS;
    for (int g = threadIdx.x; g < n; g += kBlock) {
        const int lo = g / PISCES_COUNTS_PER_LOCUS, c = g - lo * PISCES_COUNTS_PER_LOCUS;
        const int v = hist[lo * kAnchStride + c];
        if (v) dst[g] += v;
    }
}

}  // namespace pisces
