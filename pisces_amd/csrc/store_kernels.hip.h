// store_kernels.hip.h — the READ STORE of the streaming surface and the kernels that call straight from it (gfx950).
//
// FindCandidates + AddAlleleCounts as one pass over the reads.  pisces_hip_add_reads /
// pisces_hip_add_decoded_reads leave the reads where they are in HBM (bases, qualities, CIGARs: 2 bytes per aligned base) and make a
// 16-byte descriptor per read; a flush walks, per tile, the reads that overlap the tile and adds their bases straight into the LDS
// histogram the call phase reads (IStateManager.AddAlleleCounts, RegionStateManager.cs:118-220, then IAlleleCaller.Call): no
// observation log (8 bytes per observation written, then read three more times), no bucketing passes.
//
//   read_shape_kernel          add time: one lane per read, CIGAR -> ReadDesc / ReadExt, sortedness and longest reach of the segment;
//                              the workgroups behind those: the batch's row codes (encode_rows: 2 B read, 1 B written per base)
//   segment_copy_kernel        add time, small batches only: the batch's bytes appended to the open segment
//   segment_fill_dirs_kernel   a batch without per-base directions joining a segment that tracks them
//   call_store_tiles_kernel    THE FLUSH: per tile, reads (+ bucketed log tuples of pisces_hip_add_observations, if any) -> LDS
//                              histogram -> call_phase_wave (kernels.hip.h): HBM traffic ~ 1 B / observation (its row code) + 16 B / fragment and tile + 64 B / record
//   accumulate_store_tiles_kernel  the same walk into the anchor-resolved tensor int32[locus][6][3][11] (+ base-quality sums) for the
//                              candidate kernel, the collapser, NoiseModel.Window and IAlleleSource.GetAlleleCount
//
// Reads of a segment are in position order (a BAM is); a tile's reads are then the index range [first read that can still reach the
// tile, first read that starts behind it), found by a 64-ary search over the descriptors (every lane probes one: 2-4 dependent loads).
// A segment that turned out not to be sorted (state[0]) is scanned in whole: slow, still exact.
// The flush kernel walks ROW CODES (one byte per base, made at add time: encode_rows) by (fragment, half-tile) PAIRS, four lanes and
// eight bases a lane each (walk_segment_fast); the anchor-resolved walk_segment reads bases and qualities, sixteen lanes a fragment, four bases a lane;
// reads with insertions, deletions or skips go through read_walk.h's per-base function (the walk the host form and the log path use).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "kernels.hip.h"
#include "read_walk.h"
#include "stream_kernels.hip.h"

#ifndef PISCES_ADD_OCC
#define PISCES_ADD_OCC 6      // waves a SIMD of add_fused_kernel: 1 303 read workgroups of a 333 500-read batch are resident at once (5.1 waves a SIMD)
#endif
#ifdef PISCES_ADD_STAMPS   // development: when the roles of add_fused_kernel start and end (wall_clock64, 10 ns), min / max over workgroups
#define PISCES_STAMP(k, is_max) do { if (threadIdx.x == 0) A.stamps[(long long)blockIdx.x * 8 + ((k) == 0 ? 2 : (k) == 1 ? 3 : (k) - 2)] = (long long)wall_clock64(); } while (0)
#else
#define PISCES_STAMP(k, is_max) do { } while (0)
#endif
#ifndef PISCES_ADD_ABLATE
#define PISCES_ADD_ABLATE 0   // (development: ablations of add_fused_kernel, tools/add_ablate.sh)
#endif
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
constexpr uint32_t kDescGeneric = 1u << 22;      // a read whose fragments do not fit their fields (below): the flush kernel walks it base by base
// FRAGMENTS: what the flush kernel walks.  One 16-byte ReadDesc per CIGAR operation, in read order (so: in position order of the READS,
// pos0 = the read's position is the sort key of every one of them):
//   M = X        the operation's bases: length in meta, first base at aoff, first position at pos0 + delta
//   D N          the deleted positions, if the gap passes CandidateVariantFinder.CheckDeletionQuality at the base that closes it
//                (RegionStateManager.cs:131-176; a deletion at the read's end or before its final soft clip: :143-154, :199-213)
//   I S H P      nothing (length 0)
// aoff's top 16 bits hold delta, the 16 below them the read's reference span - 1 (a read that needs more, an operation of 2^20 positions
// or more, or two gaps with no aligned base between them, is kDescGeneric: its fragments are empty and the read goes through read_walk.h).  So a read with an insertion is two aligned fragments, one with a deletion two aligned
// fragments and a deletion fragment: the fast path takes them all.
constexpr uint32_t kFragDeletion = 1u << 20;
constexpr int kFragDirShift = 22;                // DirectionType of a deletion fragment's positions (the base that closes the gap), 2 bits
constexpr int kFragDeltaShift = 48;
constexpr long long kFragAoffMask = 0xFFFFFFFFll;   // bits 0..31: the offset (a segment's bases stay below 4 GB)
constexpr int kFragSpanShift = 32;               // bits 32..47: Read.EndPosition - Read.Position of the fragment's read (GetAnchorType needs both ends)
constexpr uint32_t kFragTerminal = 1u << 24;
// delta lives in the TOP 16 bits of the signed aoff: taken unsigned (an arithmetic shift would make 0x8000.. a negative offset, e.g. the
// second aligned run of 20M40000N30M)
__host__ __device__ __forceinline__ int frag_delta(long long aoff) { return (int)((unsigned long long)aoff >> 48); }     // a deletion at the read's end / before its final soft clip: its positions count in anchor bin 10
constexpr int kMaxSegments = 8;
constexpr int kStateUnsorted = 0, kStateReach = 1, kStateComplex = 2, kStateFrags = 3;   // [kStateFrags]: bit 0 some read is kDescGeneric, bit 1 some deletion fragment, bit 2 the position grid is not usable

struct SegmentView {
    const ReadDesc* frag;         // one per CIGAR operation (see above)
    const ReadDesc* desc;         // one per read
    const ReadExt* ext;
    const uint8_t* bases;
    const uint8_t* quals;
    const uint8_t* codes;         // per base: low-quality << 5 | AlleleType << 2 (encode_rows, made at add time: what the flush kernel walks)
    const uint8_t* dirs;          // per-base DirectionType of every read of the segment, or nullptr (direction = kDescReverse)
    const uint8_t* cigar_op;
    const uint32_t* cigar_len;
    const int32_t* state;         // [kStateUnsorted] != 0: not in position order; [kStateReach]: longest reference span of a read;
                                  // [kStateComplex] != 0: some read needs the general walk
    int32_t n_reads;
    int32_t n_floored;            // reads [0, n_floored) were there at the last flush: their positions below `floor` are counted already
    int32_t floor;
    int32_t n_frags, n_floored_frags;   // the same for the fragments
    int32_t grid_n;               // cells of `grid` (0: none)
    const int32_t* grid;          // grid[c - grid_base] = first fragment whose read starts at or behind position c (grid_cells), or nullptr
    int32_t grid_base, pad2;
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
    ReadDesc* frag;        // [ops0 + ...]: the fragments
    const uint8_t* quals;  // the batch's qualities / per-base directions (or nullptr), indexed by seq_offset: what a deletion fragment is gated by
    const uint8_t* dirs;
    int32_t min_bq;
    int32_t* state;
    // the second role of the launch (workgroups [shape_blocks, shape_blocks + enc_blocks)): the row codes of the batch's bases
    const uint8_t* enc_bases;
    const uint8_t* enc_quals;
    uint8_t* enc_codes;
    int64_t enc_n;
    uint32_t enc_min_bq;   // <= 127
    int32_t shape_blocks, enc_blocks;
    // the third role (the workgroups behind those, one lane a read; none when grid is nullptr): the segment's position grid
    int32_t* grid;
    int32_t grid_base, grid_n;
};

// ROW CODES.  What the flush kernel needs of a base is the row of the LDS histogram it counts in: low-quality << 5 | AlleleType << 2
// (| direction, which the fragment or the segment's per-base directions add).  Both depend on the base, its quality and the handle's
// minimum base-call quality only (AlleleHelper.GetAlleleType, AlleleHelper.cs:13-32: anything but A C G T is an N;
// RegionStateManager.cs:179-181: quality < minBQ), so they are made ONCE, when the batch joins the store — a streaming pass at 16 bytes a
// lane, every lane busy — instead of in every tile that walks the read (3.3 tiles of 64 loci for a read of 150 bases, at 71 % of the
// lanes).  The flush then loads one byte per base instead of two and classifies nothing.
__device__ __forceinline__ uint32_t row_codes_of(uint32_t bw, uint32_t qw, uint32_t qk4)
{
    const uint32_t idx4 = bw & 0x07070707u;
    const uint32_t letter4 = __builtin_amdgcn_perm(0x47000054u, 0x43004101u, idx4);   // the letter the low three bits stand for: 1 A, 3 C, 4 T, 7 G
    const uint32_t code4 = __builtin_amdgcn_perm(0x0410100Cu, 0x08100010u, idx4);     // its AlleleType << 2 (N for the rest)
    const uint32_t x4 = bw ^ letter4;                                                   // a zero byte: the base IS that letter
    const uint32_t nz = (((x4 & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x4) & 0x80808080u;
    const uint32_t nzff = (nz - (nz >> 7)) | nz;                                       // 0xFF where it is not
    const uint32_t allele4 = (nzff & 0x10101010u) | (~nzff & code4);
    // quality < minBQ: bit 7 of (0x7F + minBQ) - (q & 0x7F) is set iff (q & 0x7F) < minBQ; a quality >= 128 is never low
    const uint32_t low4 = ~qw & (qk4 - (qw & 0x7F7F7F7Fu));
    return ((low4 >> 2) & 0x20202020u) | allele4;
}
__device__ __forceinline__ void encode_rows(const ShapeArgs& A, int block, int n_blocks)
{
    const uint32_t qk4 = (0x7Fu + A.enc_min_bq) * 0x01010101u;
    const int64_t n16 = A.enc_n >> 4;
    for (int64_t i = (int64_t)block * 256 + threadIdx.x; i < n16; i += (int64_t)n_blocks * 256) {
        uint32_t b[4], q[4], c[4];
        __builtin_memcpy(b, A.enc_bases + 16 * i, 16);   // (any alignment: one global_load_dwordx4 each)
        __builtin_memcpy(q, A.enc_quals + 16 * i, 16);
#pragma unroll
        for (int k = 0; k < 4; k++) c[k] = row_codes_of(b[k], q[k], qk4);
        __builtin_memcpy(A.enc_codes + 16 * i, c, 16);
    }
    if (block == 0 && (int64_t)threadIdx.x < (A.enc_n & 15)) {   // the last bytes
        const int64_t i = (n16 << 4) + threadIdx.x;
        A.enc_codes[i] = (uint8_t)row_codes_of(A.enc_bases[i], A.enc_quals[i], qk4);
    }
}

// THE POSITION GRID of a segment: grid[p - grid_base] = index of the first fragment whose READ starts at or behind position p, written
// when a batch joins (grid_cells, one lane a read: the positions behind the read before it up to its own get the read's first fragment; positions
// behind the last read keep their fill value, which is above every index).  A tile's fragment range is then two entries of it — one
// round of two loads (wave_lower_bound2_hinted) where the 32-ary search over the whole segment takes four dependent rounds of ~1.1 us
// each.  (Cells of 8 positions with a probe round over the fragments of a cell, the first form, cost a second round and, where hundreds
// of reads start on one position — the first base of an amplicon —, a third: 7.6 us at the 90th percentile of tiles against 3.7.)
constexpr int kGridBadBit = 4;       // state[kStateFrags]
constexpr int kGridGapCells = 65536; // the widest gap between two reads that is filled (a wave's work: 64 cells a store)
// (the third role of read_shape_kernel's launch: the workgroups behind the row codes'; it reads the batch's own arrays — a read's first
// fragment is its first CIGAR operation — and, for the read before the batch's first, the descriptor an earlier launch wrote)
__device__ __forceinline__ void grid_cells(const ShapeArgs& A, int block)
{
    const int r = block * 256 + (int)threadIdx.x, lane = threadIdx.x & 63;
    int c0 = 0, c1 = -1, f0 = 0;   // this lane's read fills cells c0 .. c1 with f0 (none when the read starts where the one before it does)
    if (r < A.n_reads) {
        const long long p = A.position[r];
        const long long prev = r > 0 ? (long long)A.position[r - 1] : A.n0 > 0 ? (long long)A.desc[A.n0 - 1].pos0 : (long long)A.grid_base - 1;
        if (p > prev) {   // (else: the same position as the read before it; or out of order: the segment is then scanned, not searched)
            const long long a = prev + 1, b = p;   // the positions behind the read before it, up to its own
            // a gap wider than the grid is meant for (sparse reads: the segment goes without); or outside the span the host sized the grid
            // over (never expected)
            if (b - a > kGridGapCells || b - A.grid_base >= A.grid_n || a < A.grid_base) atomicOr(&A.state[kStateFrags], kGridBadBit);
            else { c0 = (int)(a - A.grid_base); c1 = (int)(b - A.grid_base); f0 = (int)(A.ops0 + A.cigar_offset[r]); }
        }
    }
    // a few cells: the lane's own stores; a gap (the positions between two amplicons: a thousand cells) is the whole wave's, 64 cells a store
    const bool wide = c1 - c0 >= 16;
    if (!wide)
        for (int k = c0; k <= c1; k++) A.grid[k] = f0;
    for (unsigned long long m = __ballot(wide); m != 0ull; m &= m - 1ull) {
        const int src = __builtin_ctzll(m);
        const int b0 = __builtin_amdgcn_readlane(c0, src), b1 = __builtin_amdgcn_readlane(c1, src), f = __builtin_amdgcn_readlane(f0, src);
        for (int k = b0 + lane; k <= b1; k += 64) A.grid[k] = f;
    }
    // the cells behind the batch's last read, to the grid's end: above every fragment index (no memset of the cells a batch adds; an
    // out-of-order batch leaves the grid unused)
    if (block == (A.n_reads - 1) / 256) {
        const int last_lane = (A.n_reads - 1) & 255;
        if ((int)(threadIdx.x >> 6) == (last_lane >> 6)) {
            const long long p_last = A.position[A.n_reads - 1];
            const long long from = max(p_last + 1 - (long long)A.grid_base, 0ll);
            for (long long k = from + lane; k < (long long)A.grid_n; k += 64) A.grid[k] = 0x7F7F7F7F;
        }
    }
}

// the shape role: reads [256 block, 256 block + 256), one lane each; `ok` false: the lane's read is not to be walked (refused by the checks
// made in the same lane just before, add_fused_kernel)
__device__ __forceinline__ void shape_reads(const ShapeArgs& A, const int block, const bool ok)
{
    const int r = block * 256 + (int)threadIdx.x;
    int reach = 0;
    bool unsorted = false, complex_read = false, generic_read = false, has_del = false;
    if (r < A.n_reads && ok) {
        const int c0 = A.cigar_offset[r], nc = A.cigar_offset[r + 1] - c0;
        const int s0 = A.seq_offset[r], n = A.seq_offset[r + 1] - s0;
        const int32_t pos0 = A.position[r];
        // one aligned run between clips?  phases: 0 leading clips, 1 the run (M = X), 2 trailing clips; H / P span nothing
        int phase = 0, lead = 0;
        long long run = 0, ref_span = 0;
        bool simple = true, prev_gap = false, gap_chain = false;
        for (int c = 0; c < nc; c++) {
            const uint8_t t = A.cigar_op[c0 + c];
            const long long len = A.cigar_len[c0 + c];
            if (walk_op_ref_span(t)) ref_span += len;
            if (t == 'D' || t == 'N') { gap_chain = gap_chain || prev_gap; prev_gap = true; }
            if (t == 'M' || t == '=' || t == 'X') {
                prev_gap = false;
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
        complex_read = !simple;
        // ---- the fragments, one per operation
        bool generic = ref_span > 0xFFFFll || gap_chain;   // (delta of a later fragment would not fit; the base that closes a gap would not sit right behind it)
        const long long span16 = (ref_span > 0 ? ref_span - 1 : 0) << kFragSpanShift;
        for (int c = 0; c < nc; c++) generic = generic || A.cigar_len[c0 + c] > kDescLenMask;
        if (generic) {
            d.meta |= kDescGeneric;
            A.desc[A.n0 + r] = d;
        }
        const uint8_t* const quals = A.quals + s0;
        const uint32_t rev = (A.flags[r] & 1) ? kDescReverse : 0u;
        auto dq = [&](int i) {   // CandidateVariantFinder.CheckDeletionQuality (CandidateVariantFinder.cs:294-320) at base i < n
            const int after = quals[i], before = i > 0 ? quals[i - 1] : after;
            return before >= A.min_bq && after >= A.min_bq;
        };
        auto dir_at = [&](int i) { return A.dirs ? (uint32_t)A.dirs[s0 + i] : (rev ? (uint32_t)PISCES_DIR_REVERSE : (uint32_t)PISCES_DIR_FORWARD); };
        const bool ends_in_del = nc >= 1 && A.cigar_op[c0 + nc - 1] == 'D';
        const bool ends_in_del_soft = nc >= 2 && A.cigar_op[c0 + nc - 2] == 'D' && A.cigar_op[c0 + nc - 1] == 'S';
        int ri = 0;
        long long rp = pos0, last_mapped = (long long)pos0 - 1;
        for (int c = 0; c < nc; c++) {
            const uint8_t t = A.cigar_op[c0 + c];
            const int len = (int)A.cigar_len[c0 + c];
            ReadDesc f;
            f.pos0 = pos0;
            f.meta = rev;
            f.aoff = A.base0 + s0;
            if (!generic && len > 0) {
                if (t == 'M' || t == '=' || t == 'X') {
                    const int have = max(min(len, n - ri), 0);   // (a CIGAR that runs past the read is refused before it gets here)
                    f.meta |= (uint32_t)have;
                    f.aoff = (A.base0 + s0 + ri) | span16 | ((rp - pos0) << kFragDeltaShift);
                } else if (t == 'D' || t == 'N') {
                    // the base that closes the gap: the first base of the next aligned operation (insertions and clips in between
                    // move the index, not the position); without one, only a deletion at the read's end counts, by its own rules
                    int idx = -1, first = (int)(rp - pos0), count = len;
                    uint32_t dir = 0;
                    {
                        int rj = ri;
                        for (int k = c + 1; k < nc && idx < 0; k++) {
                            const uint8_t tk = A.cigar_op[c0 + k];
                            const int lk = (int)A.cigar_len[c0 + k];
                            if ((tk == 'M' || tk == '=' || tk == 'X') && lk > 0) idx = rj;
                            else if (walk_op_read_span(tk)) rj += lk;
                        }
                    }
                    bool counted = false, terminal = false;
                    if (idx >= 0 && idx < n) {
                        counted = dq(idx);
                        dir = dir_at(idx);
                    } else if (t == 'D' && c == nc - 1 && ends_in_del && n > 0) {           // :199-213: the deleted positions follow the last mapped base
                        counted = dq(n - 1);
                        dir = dir_at(n - 1);
                        first = (int)(last_mapped + 1 - pos0);
                        terminal = true;
                    } else if (t == 'D' && c == nc - 2 && ends_in_del_soft) {              // :143-154
                        const int at = n - (int)A.cigar_len[c0 + nc - 1];
                        if (at >= 0 && at < n) {
                            counted = dq(at);
                            dir = dir_at(at);
                            first = (int)(last_mapped + 1 - pos0);
                            terminal = true;
                        }
                    }
                    if (counted && first >= 0 && first <= 0xFFFF) {
                        f.meta = rev | kFragDeletion | (uint32_t)count | (dir << kFragDirShift) | (terminal ? kFragTerminal : 0u);
                        f.aoff = (A.base0 + s0) | span16 | ((long long)first << kFragDeltaShift);
                        has_del = true;
                    }
                }
            }
            A.frag[A.ops0 + c0 + c] = f;
            if (walk_op_ref_span(t)) {
                if (walk_op_read_span(t) && len > 0) last_mapped = rp + len - 1;
                rp += len;
            }
            if (walk_op_read_span(t)) ri += len;
        }
        generic_read = generic;
        // position order, the batch's first read against the read before it in the segment (written by an earlier launch)
        if (r > 0) unsorted = A.position[r - 1] > pos0;
        else if (A.n0 > 0) unsorted = A.desc[A.n0 - 1].pos0 > pos0;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) reach = max(reach, __shfl_xor(reach, d, 64));
    const bool any_unsorted = __ballot(unsorted) != 0ull, any_complex = __ballot(complex_read) != 0ull;
    const int frag_bits = (__ballot(generic_read) != 0ull ? 1 : 0) | (__ballot(has_del) != 0ull ? 2 : 0);
    if ((threadIdx.x & 63) == 0 && PISCES_ADD_ABLATE != 9) {
        // (plain reads first: same-address atomics from the whole chip are what they cost, and after the first few waves none is needed)
        if (reach > A.state[kStateReach]) atomicMax(&A.state[kStateReach], reach);
        if (any_unsorted && A.state[kStateUnsorted] == 0) atomicOr(&A.state[kStateUnsorted], 1);
        if (any_complex && A.state[kStateComplex] == 0) atomicOr(&A.state[kStateComplex], 1);
        if (frag_bits & ~A.state[kStateFrags]) atomicOr(&A.state[kStateFrags], frag_bits);
    }
}
__global__ __launch_bounds__(256) void read_shape_kernel(ShapeArgs A)
{
    if ((int)blockIdx.x >= A.shape_blocks) {
        const int b = (int)blockIdx.x - A.shape_blocks;
        if (b < A.enc_blocks) encode_rows(A, b, A.enc_blocks);
        else grid_cells(A, b - A.enc_blocks);
        return;
    }
    shape_reads(A, (int)blockIdx.x, true);
}

// What pisces_hip_add_reads takes from a pass over the reads' CIGARs, for a batch that is in device memory (a batch handed over there,
// pisces_hip_add_device_reads, or a large host batch behind its upload): the argument checks of the reference's walk (Read.ValidateCigar,
// Read.cs:603-605; RegionStateManager.cs:363-364), one bit per block a read touches (GetBlock for every position that receives a count,
// RegionStateManager.cs:361-383: aligned runs, and gaps / terminal deletions that pass CheckDeletionQuality), and — MNV calling off —
// the candidate-record slots of every read (one per I / D operation).  One lane per read.
enum { kPrepPositionNotPositive = 1, kPrepMalformed = 2, kPrepCigarMismatch = 3, kPrepPastInt32 = 4, kPrepBadDeletionDirection = 5, kPrepBadDirection = 6 };
struct PrepareArgs {
    const int32_t* position;
    const int32_t* cigar_offset;
    const uint8_t* cigar_op;
    const uint32_t* cigar_len;
    const int32_t* seq_offset;
    const uint8_t* quals;
    const uint8_t* del_dirs;            // two per CIGAR operation, or nullptr
    int32_t n_reads, min_bq, block_size, count_indels;
    int64_t n_ops_total, n_bases_total;
    uint32_t* block_bits;               // bit k: block key k is touched; kPrepReplicas copies map_stride words apart (workgroup b writes copy b % kPrepReplicas)
    int64_t n_block_bits;
    int64_t map_stride;
    int32_t* n_found;                   // [n_reads + 1] (count_indels): candidate records / pool bytes of every read, scanned afterwards
    int32_t* n_pool;
    unsigned long long* first_error;    // read index * 8 + code of the first read that is refused (atomicMin; all ones: none)
    int32_t* key_span;                  // [0] lowest, [1] highest block key touched, [2] lowest read position, [3] X / = seen; kPrepReplicas copies of four
};
// Every wave of the launch starts with the same few words to set (reads come in position order: one or two block keys, one span), and an
// atomic on ONE address costs ~15 ns however many XCDs ask: 3 128 waves of a 200 000-read batch spent 47 us on them.  The words are
// kept in kPrepReplicas copies, a workgroup writes the copy of its index, prepare_collect folds the copies.
constexpr int kPrepReplicas = 32;
// bits [a, b] of the block map; a bit that is set already (seen through a load that goes past this XCD's L2, where a stale line would
// show zero for the rest of the launch) costs no atomic: same-address atomics from eight XCDs serialise at 0.1-0.2 us each
__device__ __forceinline__ void set_keys(const PrepareArgs& A, int replica, int64_t a, int64_t b)
{
#if PISCES_ADD_ABLATE == 7
    return;
#endif
    uint32_t* const map = A.block_bits + (int64_t)(replica & (kPrepReplicas - 1)) * A.map_stride;   // (any copy will do: the read workgroup's index)
    for (int64_t k = a; k <= b; k++) {
        if (k >= A.n_block_bits) continue;   // (cannot be: the map covers every int32 position)
        const uint32_t bit = 1u << (k & 31);
        uint32_t* const w = &map[k >> 5];
        // (no look before the atomic: a workgroup sets each of its distinct runs once, a launch a few atomics per word and copy — and a load
        // that must go past the L2 to see the others' bits is a round trip of its own in every workgroup's chain)
        (void)__hip_atomic_fetch_or(w, bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
__device__ __forceinline__ long long prep_shfl64(long long v, int src_lane)
{
    const int lo = __shfl((int)(v & 0xFFFFFFFFll), src_lane, 64);
    const int hi = __shfl((int)(v >> 32), src_lane, 64);
    return ((long long)hi << 32) | (unsigned int)lo;
}
// (a role of add_fused_kernel's launch, below; found_out / pool_out: the lane's read's candidate-record slots and pool bytes, 0 for a refused
// read; ok_out: the lane holds a read that passed every check — only then may its CIGAR be walked again, read_shape)
__device__ __forceinline__ void prepare_reads(const PrepareArgs& A, const int block, int& found_out, int& pool_out, bool& ok_out)
{
    // what the workgroup's waves found goes through LDS first: one lane of the workgroup touches the shared words (their loads go past the
    // XCD's L2 and queue on one memory channel when every wave of a large batch asks: 68 us for 500 000 reads before)
    constexpr int kRunSlots = 32;
    __shared__ int s_klo, s_khi, s_plo, s_eqx, s_nruns;
    __shared__ long long s_run_a[kRunSlots], s_run_b[kRunSlots];
    if (threadIdx.x == 0) { s_klo = 0x7FFFFFFF; s_khi = 0; s_plo = 0x7FFFFFFF; s_eqx = 0; s_nruns = 0; }
    __syncthreads();
    const int r = block * 256 + (int)threadIdx.x;
    found_out = pool_out = 0;
    ok_out = false;
    int k_lo = 0x7FFFFFFF, k_hi = 0, p_lo = 0x7FFFFFFF;
    int64_t run_a = 1, run_b = 0;   // the run of keys this read touches (empty)
    bool eqx = false;               // an X or = operation (MNV calling off: candidate discovery then walks them, surface_reads.inc.h)
    if (r < A.n_reads) {
        int code = 0;
        p_lo = A.position[r];
        const int64_t c0 = A.cigar_offset[r], c1 = A.cigar_offset[r + 1], s0 = A.seq_offset[r], s1 = A.seq_offset[r + 1];
        const int32_t pos0 = A.position[r];
        const int64_t nc = c1 - c0, n = s1 - s0;
        int found = 0, pool = 0;
        if (c0 < 0 || c1 < c0 || c1 > A.n_ops_total || s0 < 0 || s1 < s0 || s1 > A.n_bases_total) code = kPrepMalformed;
        else if (pos0 <= 0) code = kPrepPositionNotPositive;
        if (!code) {
            const uint8_t* const quals = A.quals + s0;
            auto delq = [&](int64_t idx) {   // CandidateVariantFinder.CheckDeletionQuality (CandidateVariantFinder.cs:294-320)
                if (n == 0) return false;
                const int after = idx < n ? quals[idx] : quals[idx - 1];
                const int before = idx > 0 ? quals[idx - 1] : after;
                return before >= A.min_bq && after >= A.min_bq;
            };
            // The keys a read touches are nearly always ONE run [run_a, run_b] (a read inside a block: one key); it is set behind the walk,
            // once per wave for all lanes that hold the same run.  A read whose keys have a hole (a skip across untouched blocks) sets the
            // run it leaves at once.
            auto touch = [&](int64_t from, int64_t to) {   // inclusive
                if (to < 1) return;
                if (from < 1) from = 1;
                const int64_t a = (from + A.block_size - 1) / A.block_size, b = (to + A.block_size - 1) / A.block_size;   // GetBlockKey
                if (run_b < run_a) { run_a = a; run_b = b; }
                else if (a <= run_b + 1 && b >= run_a - 1) { run_a = min(run_a, a); run_b = max(run_b, b); }
                else { set_keys(A, block, run_a, run_b); run_a = a; run_b = b; }
                k_lo = min(k_lo, (int)a);
                k_hi = max(k_hi, (int)min(b, (int64_t)0x7FFFFFFF));
            };
            int64_t read_span = 0, ref_span = 0;
            bool bad_del_dir = false;
            for (int64_t c = 0; c < nc && !code; c++) {
                const uint8_t t = A.cigar_op[c0 + c];
                const uint32_t len = A.cigar_len[c0 + c];
                if (len > 0x0FFFFFFFu) code = kPrepMalformed;   // (BAM keeps an operation's length in 28 bits)
                if (walk_op_read_span(t)) read_span += len;
                if (walk_op_ref_span(t)) ref_span += len;
                if (A.del_dirs && t == 'D')
                    for (int k = 0; k < 2; k++) {
                        const uint8_t d = A.del_dirs[2 * (c0 + c) + k];
                        if (d > 2 && d != PISCES_DIR_UNTRACKED) bad_del_dir = true;
                    }
                if (A.count_indels) {
                    if (t == 'I' || t == 'D') found++;
                    if (t == 'I' && len > 32u) pool += (int)len;   // (kFoundInline)
                    if (t == 'X' || t == '=') eqx = true;
                }
            }
            if (!code && nc > 0 && read_span != n) code = kPrepCigarMismatch;               // Read.ValidateCigar (Read.cs:603-605)
            if (!code && (int64_t)pos0 + ref_span > 0x7FFFFFFFll) code = kPrepPastInt32;
            if (!code && bad_del_dir) code = kPrepBadDeletionDirection;
            if (!code) {
                int64_t rp = pos0, last_mapped = (int64_t)pos0 - 1, ri = 0;
                for (int64_t c = 0; c < nc; c++) {
                    const uint8_t t = A.cigar_op[c0 + c];
                    const int64_t len = A.cigar_len[c0 + c];
                    if (walk_op_read_span(t) && walk_op_ref_span(t) && len > 0) {
                        if (rp > last_mapped + 1 && ri < n && delq(ri)) touch(last_mapped + 1, rp - 1);
                        touch(rp, rp + len - 1);
                        last_mapped = rp + len - 1;
                    }
                    if (walk_op_ref_span(t)) rp += len;
                    if (walk_op_read_span(t)) ri += len;
                }
                const bool ends_del = nc >= 1 && A.cigar_op[c0 + nc - 1] == 'D';
                const bool ends_del_soft = nc >= 2 && A.cigar_op[c0 + nc - 2] == 'D' && A.cigar_op[c0 + nc - 1] == 'S';
                if (ends_del && n > 0 && delq(n - 1)) touch(last_mapped + 1, last_mapped + (int64_t)A.cigar_len[c0 + nc - 1]);
                if (ends_del_soft) {
                    const int64_t idx = n - (int64_t)A.cigar_len[c0 + nc - 1];
                    if (idx >= 0 && idx < n && delq(idx)) touch(last_mapped + 1, last_mapped + (int64_t)A.cigar_len[c0 + nc - 2]);
                }
            }
        }
        found_out = code ? 0 : found;
        pool_out = code ? 0 : pool;
        ok_out = code == 0;
        if (code) atomicMin(A.first_error, (unsigned long long)r * 8ull + (unsigned long long)code);
    }
    {   // the lanes' runs, each distinct one set by one lane (reads come in position order: a wave holds one or two)
        unsigned long long todo = __ballot(run_b >= run_a);
        const int lane = threadIdx.x & 63;
        while (todo) {
            const int l0 = __builtin_ctzll(todo);
            const long long a0 = prep_shfl64((long long)run_a, l0), b0 = prep_shfl64((long long)run_b, l0);
            const unsigned long long same = __ballot(run_b >= run_a && run_a == a0 && run_b == b0);
            if (lane == l0) {
                const int slot = atomicAdd(&s_nruns, 1);
                if (slot < kRunSlots) { s_run_a[slot] = a0; s_run_b[slot] = b0; }
                else set_keys(A, block, a0, b0);
            }
            todo &= ~same;
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        k_lo = min(k_lo, __shfl_xor(k_lo, d, 64));
        k_hi = max(k_hi, __shfl_xor(k_hi, d, 64));
        p_lo = min(p_lo, __shfl_xor(p_lo, d, 64));
    }
    const bool any_eqx = __ballot(eqx) != 0ull;   // (all lanes: before the branch)
    if ((threadIdx.x & 63) == 0) {
        atomicMin(&s_plo, p_lo);
        if (any_eqx) s_eqx = 1;
        if (k_hi > 0) { atomicMin(&s_klo, k_lo); atomicMax(&s_khi, k_hi); }
    }
    __syncthreads();
    {   // the workgroup's distinct runs, each set once
        const int n_runs = min(s_nruns, kRunSlots), t = threadIdx.x;
        if (t < n_runs) {
            bool seen = false;
            for (int j = 0; j < t; j++) seen = seen || (s_run_a[j] == s_run_a[t] && s_run_b[j] == s_run_b[t]);
            if (!seen) set_keys(A, block, s_run_a[t], s_run_b[t]);
        }
    }
    if (threadIdx.x == 0 && PISCES_ADD_ABLATE != 8) {
        int32_t* const span = A.key_span + 4 * (int)(block & (kPrepReplicas - 1));
        // (no look before the atomics either: four of them a workgroup on the copy of its index, none waited for)
        (void)__hip_atomic_fetch_min(&span[2], s_plo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (s_eqx) (void)__hip_atomic_fetch_max(&span[3], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (s_khi > 0) {
            (void)__hip_atomic_fetch_min(&span[0], s_klo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            (void)__hip_atomic_fetch_max(&span[1], s_khi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}
// What the host needs of prepare_reads' results, into pinned host memory as the launch's own stores (one wait, no copy operations):
// the verdict, the span of touched blocks, the totals of the candidate-record slots, and the keys of the touched blocks (any order) — whose
// bits are cleared here for the next batch.  More touched blocks than `capacity`: nothing is cleared, the host reads the map's words itself.
struct PrepVerdict {
    unsigned long long first_error;
    long long totals[2];
    int32_t span[3];
    int32_t n_keys;      // touched blocks (all of them, also beyond capacity)
    int32_t has_eqx;     // some read has an X or = operation
    int32_t ready;       // the launch's sequence number, stored last (system-scope release): the host may poll it instead of waiting for the stream
};
static_assert(sizeof(PrepVerdict) == 48, "PrepVerdict layout");
// (run by the read workgroup of add_fused_kernel that is counted in last.  It runs while the streaming role has the memory system full — a
// round trip is 4-5 us then — so it is written for few of them: every word of the other workgroups is read with an agent-scope atomic
// load (it sees their agent-scope atomics without a fence; an acquire fence is an L2 invalidate in front of every load), two rounds of
// independent loads; what goes to the host goes as system-scope stores, acknowledged (s_waitcnt) before the word the host polls is
// stored — no release fence: that is a write-back of the XCD's L2, full of the streaming role's lines.  First form: 38 us; the launch
// ended with it.)
__device__ __forceinline__ void prepare_collect(uint32_t* __restrict__ block_bits, int64_t map_stride, int32_t* __restrict__ key_span,
                                                unsigned long long* __restrict__ first_error, long long* __restrict__ totals,
                                                PrepVerdict* __restrict__ out, int32_t* __restrict__ keys_out, int32_t capacity)
{
    __shared__ int s_n, s_at, s_span[4];
    __shared__ unsigned long long s_err;
    __shared__ long long s_tot[2];
    if (threadIdx.x == 0) { s_n = 0; s_at = 0; s_span[0] = 0x7FFFFFFF; s_span[1] = 0; s_span[2] = 0x7FFFFFFF; s_span[3] = 0; }
    __syncthreads();
    if (threadIdx.x < kPrepReplicas) {   // the copies of the span folded
        int32_t* const sp = key_span + 4 * threadIdx.x;
        const int v0 = __hip_atomic_load(&sp[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), v1 = __hip_atomic_load(&sp[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                  v2 = __hip_atomic_load(&sp[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), v3 = __hip_atomic_load(&sp[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        atomicMin(&s_span[0], v0); atomicMax(&s_span[1], v1); atomicMin(&s_span[2], v2); atomicMax(&s_span[3], v3);
    } else if (threadIdx.x == 64) {      // (another wave: the same round trip)
        s_err = __hip_atomic_load(first_error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_tot[0] = __hip_atomic_load(&totals[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_tot[1] = __hip_atomic_load(&totals[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    const int lo = s_span[0], hi = s_span[1];
    const bool any = hi >= lo && hi > 0;
    const int w0 = any ? lo >> 5 : 0, w1 = any ? hi >> 5 : -1;
    // the copies of the map folded into copy 0 (the others cleared for the next batch)
    int mine = 0;
    for (int w = w0 + (int)threadIdx.x; w <= w1; w += 256) {
        uint32_t v[kPrepReplicas], bits = 0;
#pragma unroll
        for (int r = 0; r < kPrepReplicas; r++) v[r] = __hip_atomic_load(&block_bits[(int64_t)r * map_stride + w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (all the loads first: one round trip)
#pragma unroll
        for (int r = 0; r < kPrepReplicas; r++) bits |= v[r];
#pragma unroll
        for (int r = 1; r < kPrepReplicas; r++)
            if (v[r] && map_stride) block_bits[(int64_t)r * map_stride + w] = 0;
        block_bits[w] = bits;
        mine += __popc(bits);
    }
    if (mine) atomicAdd(&s_n, mine);
    __syncthreads();
    const int n = s_n;
    if (n <= capacity) {
        for (int w = w0 + (int)threadIdx.x; w <= w1; w += 256) {
            uint32_t bits = block_bits[w];   // (this thread's own store of the pass above)
            if (!bits) continue;
            int at = atomicAdd(&s_at, __popc(bits));
            for (; bits; bits &= bits - 1) __hip_atomic_store(&keys_out[at++], w * 32 + (int)__builtin_ctz(bits), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            block_bits[w] = 0;
        }
    }
    if (threadIdx.x == 0) {
        __hip_atomic_store(&out->first_error, s_err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&out->totals[0], s_tot[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&out->totals[1], s_tot[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&out->span[0], s_span[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&out->span[1], s_span[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&out->span[2], s_span[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&out->n_keys, n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&out->has_eqx, s_span[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// the small arrays of a batch handed over in device memory, copied into the store's layout by add_fused_kernel's misc role: sixteen bytes
// a lane at any alignment of source and destination (gfx950 runs with unaligned global access enabled)
struct CopyRanges16 {
    uint8_t* dst[10];
    const uint8_t* src[10];
    int64_t n[10];
};
// ---- ONE LAUNCH PER ADD of a batch that lies in device memory (pisces_hip_add_device_reads, or a large host batch behind its upload) ----
// What used to be seven launches and three memsets (ranges_copy16 | read_prepare | check_directions | scan_block_sums | scan_block_offsets |
// scan_apply | prepare_collect, then read_shape behind the host's wait) — four streaming passes over the batch — is one launch that reads
// the batch ONCE.  Roles by workgroup index:
//   [0, read_blocks)            one lane a read: the checks and the bookkeeping (prepare_reads), the candidate-record slots scanned in the
//                               same launch (a decoupled look-back over the workgroups' sums: no scan kernels), and — a batch that becomes
//                               a segment of its own — descriptors and fragments from the same lane (shape_reads).  The workgroup that
//                               finishes last folds the copies of the block map and writes the verdict into pinned host memory
//                               (prepare_collect), then leaves every shared word as the next launch expects it: no memsets.
//   [.., + stream_blocks)       bases and qualities (and per-base directions), sixteen bytes a lane: loaded once, stored into the store's
//                               own arrays (a batch handed over in the caller's memory), encoded into row codes (encode_rows' arithmetic), the directions checked
//   [.., + misc_blocks)         the batch's small arrays into the store's layout (positions, flags, offsets, CIGARs, deletion directions)
struct AddFusedArgs {
    PrepareArgs P;              // (reads the batch where it lies NOW: the caller's arrays when the batch is being copied)
    ShapeArgs S;                // shape role (do_shape) — reads the same arrays; S.enc_* unused here
    int32_t do_shape;
    int32_t read_blocks, stream_blocks, misc_blocks;
    int32_t role_stride;        // every role_stride-th unit of eight workgroups is a read unit (>= 1), until there are read_blocks read workgroups
    int32_t stream_first;       // != 0: the stream and misc workgroups are the launch's first, the read workgroups follow
    const uint8_t* s_bases;     // stream role: sources
    const uint8_t* s_quals;
    const uint8_t* s_dirs;      // or nullptr
    uint8_t* d_bases;           // copies (nullptr: the batch is where it stays)
    uint8_t* d_quals;
    uint8_t* d_dirs;
    uint8_t* d_codes;           // row codes (nullptr: made later, read_shape_kernel's encode role)
    int64_t n_seq;
    uint32_t enc_min_bq;        // <= 127
    CopyRanges16 C;             // misc role
    unsigned long long* scan_state;   // [read_blocks], zero between launches: status << 62 | pool bytes << 31 | records
    unsigned int* done;               // [1 + kPrepReplicas], zero between launches: [1 + c] read workgroups of index class c that are through, [0] classes that are
    long long* totals;                // [2]: records, pool bytes of the batch
    PrepVerdict* verdict;             // pinned host memory
    int32_t* keys_out;
    int32_t capacity;
    int32_t* bad_direction;           // pinned host memory, zero at launch: set by the stream role when a per-base direction is none of 0 / 1 / 2
    int32_t seq;                      // this launch's number (nonzero): verdict->ready
    long long* stamps;                // development (PISCES_ADD_STAMPS): [16]
};
constexpr int kScanDirectBlocks = 4096;   // read workgroups up to which the candidate-slot scan adds up the words before its own (beyond: look-back)
constexpr unsigned long long kScanAggregate = 1ull << 62, kScanInclusive = 2ull << 62, kScanField = 0x7FFFFFFFull;

// the stream and misc roles of an add's launch: workgroup sb of stream_blocks + misc_blocks (add_fused_kernel's workgroups that are not
// read workgroups.  As a launch of their own on a second stream beside the read role's — the two roles want different register budgets, and
// in one kernel every wave is given the larger — the two launches did not run side by side but one after the other: 98-107 us for the
// add's span against 84-86, profiles/r06_add_fused.txt)
__device__ __forceinline__ void add_stream_roles(const AddFusedArgs& A, const int sb)
{
        if (sb < A.stream_blocks) {
            // ---- stream role
#if PISCES_ADD_ABLATE == 1
            return;
#endif
            PISCES_STAMP(0, false);
            const uint32_t qk4 = (0x7Fu + A.enc_min_bq) * 0x01010101u;
            const int64_t n16 = A.n_seq >> 4;
            bool bad_dir = false;
            // kStreamPieces sixteen-byte pieces of bases and of qualities a lane, ALL REQUESTED BEFORE THE FIRST IS USED: 8 KB a wave in flight.
            // While the read role holds most of the chip's wave slots (its workgroups are the launch's first: ~35 us of round trips, little
            // traffic) only 3-4 streaming waves a CU run; with one piece in flight each they moved a fifth of what the memory system takes and
            // the launch's bytes waited for the read role to end (stream role through at 61 us, of which the first 35 at a fifth of the rate).
            constexpr int kStreamPieces = 4;
            for (int64_t i0 = (int64_t)sb * (256 * kStreamPieces); i0 < n16; i0 += (int64_t)A.stream_blocks * (256 * kStreamPieces)) {
                uint32_t bw[kStreamPieces][4], qw[kStreamPieces][4];
#pragma unroll
                for (int p = 0; p < kStreamPieces; p++) {
                    const int64_t i = min(i0 + p * 256 + (int64_t)threadIdx.x, n16 - 1);   // (clamped: the load is unconditional, the stores are not)
                    __builtin_memcpy(bw[p], A.s_bases + 16 * i, 16);
                    __builtin_memcpy(qw[p], A.s_quals + 16 * i, 16);
                }
#pragma unroll
                for (int p = 0; p < kStreamPieces; p++) {
                    const int64_t i = i0 + p * 256 + (int64_t)threadIdx.x;
                    if (i >= n16) continue;
                    if (A.s_dirs) {   // (stitched reads' per-base directions: rare, fetched here)
                        uint32_t dw[4];
                        __builtin_memcpy(dw, A.s_dirs + 16 * i, 16);
#pragma unroll
                        for (int k = 0; k < 4; k++) bad_dir = bad_dir || ((dw[k] + 0x7D7D7D7Du) | dw[k]) & 0x80808080u;   // some byte > 2
                        if (A.d_dirs) __builtin_memcpy(A.d_dirs + 16 * i, dw, 16);
                    }
#if PISCES_ADD_ABLATE == 5
                    if (bw[p][0] == 0x12345678u && qw[p][1] == 0x9ABCDEF0u) __builtin_memcpy(A.d_codes + 16 * i, bw[p], 16);
                    continue;
#endif
#if PISCES_ADD_ABLATE != 4
                    if (A.d_bases) {
                        __builtin_memcpy(A.d_bases + 16 * i, bw[p], 16);
                        __builtin_memcpy(A.d_quals + 16 * i, qw[p], 16);
                    }
#endif
                    if (A.d_codes) {
                        uint32_t c[4];
#pragma unroll
                        for (int k = 0; k < 4; k++) c[k] = row_codes_of(bw[p][k], qw[p][k], qk4);
                        __builtin_memcpy(A.d_codes + 16 * i, c, 16);
                    }
                }
            }
            if (sb == 0 && (int64_t)threadIdx.x < (A.n_seq & 15)) {   // the last bytes
                const int64_t i = (n16 << 4) + threadIdx.x;
                const uint8_t bb = A.s_bases[i], qq = A.s_quals[i];
                if (A.s_dirs) { const uint8_t dd = A.s_dirs[i]; bad_dir = bad_dir || dd > 2; if (A.d_dirs) A.d_dirs[i] = dd; }
                if (A.d_bases) { A.d_bases[i] = bb; A.d_quals[i] = qq; }
                if (A.d_codes) A.d_codes[i] = (uint8_t)row_codes_of(bb, qq, qk4);
            }
            if (__ballot(bad_dir) != 0ull && (threadIdx.x & 63) == 0) __hip_atomic_store(A.bad_direction, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            PISCES_STAMP(1, true);
        } else {
            // ---- misc role: the small arrays, sixteen bytes a lane, one range after the other over this role's workgroups
            const int64_t stride = (int64_t)A.misc_blocks * 256, t = (int64_t)(sb - A.stream_blocks) * 256 + threadIdx.x;
#pragma unroll 1
            for (int k = 0; k < 10; k++) {
                const int64_t n = A.C.n[k], n16 = n >> 4;
                if (n <= 0) continue;
                const uint8_t* const src = A.C.src[k];
                uint8_t* const dst = A.C.dst[k];
                for (int64_t i = t; i < n16; i += stride) {
                    uint32_t v[4];
                    __builtin_memcpy(v, src + 16 * i, 16);
                    __builtin_memcpy(dst + 16 * i, v, 16);
                }
                if (t < (n & 15)) dst[(n16 << 4) + t] = src[(n16 << 4) + t];
            }
        }
}
__global__ __launch_bounds__(256, PISCES_ADD_OCC) void add_fused_kernel(AddFusedArgs A)
{
    // Roles by workgroup index.  stream_first > 0: the launch's FIRST workgroups are the stream (and misc) role — a few persistent ones a CU
    // that walk the batch's bytes from the launch's first microsecond to its last with 8 KB a wave in flight — and the read workgroups
    // follow in index order (which is dispatch order: their scan waits for lower read indices only) into the slots that are left.  With the
    // read role in front (stream_first == 0; every role_stride-th unit of eight workgroups — one per XCD — a read unit, until there are
    // read_blocks of them) its workgroups take 85 % of the chip's wave slots for ~35 us of round trips and the launch's bytes wait for them.
    const int raw = (int)blockIdx.x;
    bool read_role;
    int b;   // the index inside its role(s)
    if (A.stream_first) {
        const int front = A.stream_blocks + A.misc_blocks;
        read_role = raw >= front;
        b = read_role ? raw - front : raw;
    } else {
        const int unit = raw >> 3, in_unit = raw & 7;
        const int uq = unit / A.role_stride;
        const bool read_unit = unit - uq * A.role_stride == 0;
        const int reads_before = min(A.read_blocks, ((unit + A.role_stride - 1) / A.role_stride) * 8 + (read_unit ? in_unit : 0));
        read_role = read_unit && uq * 8 + in_unit < A.read_blocks;
        b = read_role ? uq * 8 + in_unit : raw - reads_before;
    }
    __shared__ int s_wf[4], s_wp[4], s_excl[2], s_last;
    if (!read_role) {
        add_stream_roles(A, b);
        // (the streaming roles leave nothing the collecting workgroup reads: no fence, no count — a release fence here is a write-back of the
        // XCD's whole L2, and 8 000 workgroups x 4 waves of them made the launch 1.2 ms where its bytes take 45 us)
        return;
    }
    // ---- read role
    {
    int found, pool;
    bool ok;
    PISCES_STAMP(2, false);
    PISCES_STAMP(3, true);
#if PISCES_ADD_ABLATE == 6
    found = pool = 0; ok = true;
#else
    prepare_reads(A.P, b, found, pool, ok);
#endif
    PISCES_STAMP(4, true);
#if PISCES_ADD_ABLATE != 2
    if (A.do_shape) shape_reads(A.S, b, ok);
#endif
    PISCES_STAMP(5, true);
    // the candidate-record slots: exclusive scan over the batch's reads, in this launch
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int incl_f = found, incl_p = pool;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int uf = __shfl_up(incl_f, d, 64), up = __shfl_up(incl_p, d, 64);
        if (lane >= d) { incl_f += uf; incl_p += up; }
    }
    if (lane == 63) { s_wf[wave] = incl_f; s_wp[wave] = incl_p; }
    __syncthreads();
    int off_f = 0, off_p = 0, agg_f = 0, agg_p = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) {
        if (w < wave) { off_f += s_wf[w]; off_p += s_wp[w]; }
        agg_f += s_wf[w]; agg_p += s_wp[w];
    }
    if (wave == 0) {
        const unsigned long long mine = ((unsigned long long)(unsigned)agg_p << 31) | (unsigned long long)(unsigned)agg_f;
        long long excl_f = 0, excl_p = 0;
        if (A.read_blocks <= kScanDirectBlocks) {
            // The read workgroups of a launch are resident together and reach this point together, so a look-back finds no inclusive word
            // near by and walks — 64 workgroups a round trip, the inclusive words coming to meet it at the same pace: ~10 dependent round
            // trips for workgroup 1 300, the launch's critical path.  Up to kScanDirectBlocks workgroups every one simply ADDS UP the
            // words before its own (lane i takes words i, i + 64, ...: 21 independent loads a lane for the last of 1 303), waiting only
            // for words that are not published yet: one round trip.
            if (lane == 0) __hip_atomic_store(&A.scan_state[b], kScanAggregate | mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (int i0 = 0; i0 < b; i0 += 64 * 16) {
                unsigned long long w[16];
                for (;;) {
                    bool all = true;
#pragma unroll
                    for (int k = 0; k < 16; k++) {
                        const int idx = i0 + 64 * k + lane;
                        w[k] = idx < b ? __hip_atomic_load(&A.scan_state[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : kScanAggregate;
                        all = all && (w[k] >> 62) != 0ull;
                    }
                    if (__ballot(!all) == 0ull) break;
                    __builtin_amdgcn_s_sleep(1);
                }
#pragma unroll
                for (int k = 0; k < 16; k++) { excl_f += (long long)(w[k] & kScanField); excl_p += (long long)((w[k] >> 31) & kScanField); }
            }
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) { excl_f += __shfl_xor(excl_f, d, 64); excl_p += __shfl_xor(excl_p, d, 64); }
        } else if (b == 0) {
            if (lane == 0) __hip_atomic_store(&A.scan_state[0], kScanInclusive | mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            if (lane == 0) __hip_atomic_store(&A.scan_state[b], kScanAggregate | mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int base = b - 1;   // lane l looks at workgroup base - l
#if PISCES_ADD_ABLATE == 3
            base = -1000000;
#endif
            for (;;) {
                const int idx = base - lane;
                const unsigned long long w = idx >= 0 ? __hip_atomic_load(&A.scan_state[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : kScanInclusive;
                const unsigned long long none = __ballot((w >> 62) == 0ull), incl = __ballot((w >> 62) == 2ull);
                const int first_incl = incl ? __builtin_ctzll(incl) : 64;
                const unsigned long long upto = first_incl >= 63 ? ~0ull : ((2ull << first_incl) - 1ull);   // lanes 0 .. first_incl
                if (none & upto) { __builtin_amdgcn_s_sleep(1); continue; }   // a workgroup in between has not published yet
                int f = (lane <= first_incl) ? (int)(w & kScanField) : 0, p = (lane <= first_incl) ? (int)((w >> 31) & kScanField) : 0;
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) { f += __shfl_xor(f, d, 64); p += __shfl_xor(p, d, 64); }
                excl_f += f; excl_p += p;
                if (first_incl < 64) break;
                base -= 64;
            }
            if (lane == 0)
                __hip_atomic_store(&A.scan_state[b], kScanInclusive | ((unsigned long long)(excl_p + agg_p) << 31) | (unsigned long long)(excl_f + agg_f), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
        }
        if (lane == 0) {
            s_excl[0] = (int)excl_f; s_excl[1] = (int)excl_p;
            if (b == A.read_blocks - 1) {
                __hip_atomic_store(&A.totals[0], excl_f + agg_f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&A.totals[1], excl_p + agg_p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    __syncthreads();
    PISCES_STAMP(6, true);
    if (A.P.n_found) {
        const int r = b * 256 + (int)threadIdx.x;
        if (r < A.P.n_reads) {
            A.P.n_found[r] = s_excl[0] + off_f + incl_f - found;
            A.P.n_pool[r] = s_excl[1] + off_p + incl_p - pool;
        }
        if (b == A.read_blocks - 1 && threadIdx.x == 0) { A.P.n_found[A.P.n_reads] = s_excl[0] + agg_f; A.P.n_pool[A.P.n_reads] = s_excl[1] + agg_p; }
    }
    }
    // The read workgroup that is through last collects.  What it reads of the others — block map, spans, first error, totals — was written
    // with agent-scope atomics, which are performed past the XCDs' L2s, and the barrier's s_waitcnt vmcnt(0) has every wave's atomics
    // acknowledged before thread 0 counts the workgroup in: the count is a RELAXED atomic.  (A release there is a write-back of the XCD's
    // whole L2 — while the stream role fills it — per read workgroup: 60 us of the launch's 127; a fence in every wave of every role: 1.2 ms.)
    // The collector reads with agent-scope atomic loads.
#if PISCES_ADD_ABLATE == 10
    return;
#endif
    __syncthreads();
    if (threadIdx.x == 0) {
        // counted in two levels — kPrepReplicas counters by workgroup index, then one for the counters that are full — so that no word sees
        // more than read_blocks / 32 (+ 32) atomics: same-address atomics from eight XCDs are served one at a time, ~0.1 us each
        const unsigned cls = (unsigned)b & (kPrepReplicas - 1u);
        const unsigned members = ((unsigned)A.read_blocks - cls + kPrepReplicas - 1u) / kPrepReplicas;
        int last = 0;
        if (__hip_atomic_fetch_add(&A.done[1 + cls], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == members - 1u) {
            const unsigned classes = min((unsigned)A.read_blocks, (unsigned)kPrepReplicas);
            last = __hip_atomic_fetch_add(&A.done[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == classes - 1u;
        }
        s_last = last;
    }
    __syncthreads();
    PISCES_STAMP(7, true);
    if (!s_last) return;
    prepare_collect(A.P.block_bits, A.P.map_stride, A.P.key_span, A.P.first_error, A.totals, A.verdict, A.keys_out, A.capacity);
    __syncthreads();
    // the shared words as the next launch expects them (plain stores: the next launch starts behind the end of this one)
    for (int i = threadIdx.x; i < A.read_blocks; i += 256) A.scan_state[i] = 0ull;
    if (threadIdx.x < kPrepReplicas) {
        int32_t* const sp = A.P.key_span + 4 * threadIdx.x;
        sp[0] = 0x7FFFFFFF; sp[1] = 0; sp[2] = 0x7FFFFFFF; sp[3] = 0;
    }
    if (threadIdx.x <= kPrepReplicas) A.done[threadIdx.x] = 0u;
    if (threadIdx.x == 0) *A.P.first_error = ~0ull;
    // the word the host polls, behind every thread's stores to the host's memory (verdict, keys): the barrier's workgroup-scope fence has
    // each wave wait for its stores' acknowledgements (s_waitcnt vmcnt(0)); device-to-host writes arrive in the order they were acknowledged in
    __syncthreads();
    PISCES_STAMP(8, true);
    if (threadIdx.x == 0) __hip_atomic_store(&A.verdict->ready, A.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
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

// Both ends of a tile's read range in the same rounds: lanes 0-31 look for the first read at or behind x_lo, lanes 32-63 for the first
// at or behind x_hi (32 probes each a round: as many rounds as one 64-ary search for the segment sizes that occur, half the latency of two).
__device__ __forceinline__ void wave_lower_bound2(const ReadDesc* __restrict__ desc, int n, int x_lo, int x_hi, int lane, int* lo_out, int* hi_out)
{
    const int half = lane >> 5, sub = lane & 31;
    const int x = half ? x_hi : x_lo;
    int a = 0, b = n;   // (per half; the same in all lanes of a half)
    bool more = n > 0;
    while (__ballot(more) != 0ull) {
        const int step = max((b - a + 31) >> 5, 1);
        const long long idx = (long long)a + (long long)(sub + 1) * step - 1;
        const int v = (more && idx < b) ? desc[idx].pos0 : 0x7FFFFFFF;
        const unsigned long long below = __ballot(more && idx < b && v < x);
        const int c = __popc((unsigned int)(half ? (below >> 32) : (below & 0xFFFFFFFFull)));
        const long long na = (long long)a + (long long)c * step;
        const long long nb = na + step - 1;
        if (more) {
            a = (int)min(na, (long long)b);
            b = (int)max(min(nb, (long long)b), (long long)a);
        }
        more = b > a;
    }
    *lo_out = __builtin_amdgcn_readlane(a, 0);
    *hi_out = __builtin_amdgcn_readlane(a, 32);
}

// Both ends of a tile's fragment range from the position grid: the entry of x_lo and the entry of x_hi (lane 0 loads one, lane 32 the
// other: one round trip).  A position before the grid's first is before every read (0), one behind its last behind every read (n).
__device__ __forceinline__ void wave_lower_bound2_hinted(int n, int x_lo, int x_hi, int lane, const int32_t* __restrict__ grid, int grid_base, int grid_n,
                                                         int* lo_out, int* hi_out)
{
    const int x = lane >> 5 ? x_hi : x_lo;
    const long long k = (long long)max(x, 0) - grid_base;
    int a = k < 0 ? 0 : n;
    if (k >= 0 && k < grid_n) a = min(grid[k], n);
    *lo_out = __builtin_amdgcn_readlane(a, 0);
    *hi_out = __builtin_amdgcn_readlane(a, 32);
}

__device__ __forceinline__ long long readlane64(long long v, int lane_index)
{
    const int lo = __builtin_amdgcn_readlane((int)(v & 0xFFFFFFFFll), lane_index);
    const int hi = __builtin_amdgcn_readlane((int)(v >> 32), lane_index);
    return ((long long)hi << 32) | (unsigned int)lo;
}

// The reads of desc[lo, hi) with insertions / deletions / skips, one at a time: RegionStateManager.AddAlleleCounts base by base
// (read_walk.h); on_obs(position, allele, direction, anchor, quality) for every observation inside the tile and at or above the floor.
template <bool kDirs, typename OnObs>
__device__ __forceinline__ void walk_segment_complex(const SegmentView& G, int lo, int hi, int tile_start, int min_bq, int lane, int wid, int n_waves, OnObs on_obs,
                                                     uint32_t which = kDescComplex /* the reads to take: kDescComplex, or kDescGeneric for the flush kernel */)
{
    const int tile_end = tile_start + kTile - 1;
    for (int base = lo + wid * 64; base < hi; base += n_waves * 64) {
        const int cnt = min(64, hi - base);
        const ReadDesc d = G.desc[base + min(lane, cnt - 1)];
        unsigned long long complex_mask = __ballot(lane < cnt && (d.meta & which));
        // the reads with insertions / deletions / skips: RegionStateManager.AddAlleleCounts base by base (read_walk.h)
        while (complex_mask) {
            const int u = __builtin_ctzll(complex_mask);
            complex_mask &= complex_mask - 1;
            const int r = base + u;
            const int pos0 = __builtin_amdgcn_readlane(d.pos0, u);
            const uint32_t meta = (uint32_t)__builtin_amdgcn_readlane((int)d.meta, u);
            long long aoff = readlane64(d.aoff, u);
            const ReadExt e = G.ext[r];
            const ReadShape shape = read_shape(pos0, e.n_bases, e.n_cigar, G.cigar_op + e.cig_off, G.cigar_len + e.cig_off);
            if (pos0 > tile_end || (long long)pos0 + shape.ref_span - 1 < tile_start) continue;
            if (!(meta & kDescComplex)) {   // (a read of one aligned run keeps the index of its first ALIGNED base: back to its first base)
                int lead = 0;
                for (int c = 0; c < e.n_cigar; c++) {
                    const uint8_t t = G.cigar_op[e.cig_off + c];
                    if (t == 'S') lead += (int)G.cigar_len[e.cig_off + c];
                    else if (t != 'H' && t != 'P') break;
                }
                aoff -= lead;
            }
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

// four bytes at any address (gfx950 runs with unaligned global access enabled: one global_load_dword)
__device__ __forceinline__ uint32_t load_u32_unaligned(const uint8_t* p)
{
    uint32_t v;
    __builtin_memcpy(&v, p, 4);
    return v;
}
__device__ __forceinline__ unsigned long long load_u64_unaligned(const uint8_t* p)   // eight bytes at any address: one global_load_dwordx2
{
    unsigned long long v;
    __builtin_memcpy(&v, p, 8);
    return v;
}
__device__ __forceinline__ long long shfl64(long long v, int src_lane)
{
    const int lo = __shfl((int)(v & 0xFFFFFFFFll), src_lane, 64);
    const int hi = __shfl((int)(v >> 32), src_lane, 64);
    return ((long long)hi << 32) | (unsigned int)lo;
}

// The reads of one segment that can touch the tile [tile_start, tile_start + 64).
// Simple reads (one aligned run): a wave takes 64 descriptors with one load, then SIXTEEN reads at a time: lane (g, j) = (lane >> 4,
// lane & 15) takes, of the reads 4u + g (u = 0..3), the four bases on the loci 4j .. 4j + 3 of the tile — one 4-byte load each for
// bases and qualities (any alignment), 256 bytes per load instruction, eight (twelve with per-base directions) in flight per wave.
// The four bytes are added in the order k = (s + g) & 3, s = 0..3: in step s the 64 lanes stand on 64 different loci, so a ds_add
// never has two lanes on one bank, whatever the reads.  on_base(locus, base, quality, direction, valid, pos0, aligned bases).
// Then the complex reads of the same descriptors, one at a time, through walk_base: on_obs(position, allele, direction, anchor, quality).
// Bytes up to 3 before a read's first aligned base and up to 3 behind its last one are loaded (and not used): the segment's arrays
// have that much room on both sides (kSegmentPad).
constexpr int kSegmentPad = 64;
template <bool kDirs, typename OnBase, typename OnObs>
__device__ __forceinline__ void walk_segment(const SegmentView& G, int tile_start, int min_bq, int lane, int wid, int n_waves, OnBase on_base, OnObs on_obs)
{
    if (G.n_frags <= 0) return;
    const int tile_end = tile_start + kTile - 1;
    const bool sorted = G.state[kStateUnsorted] == 0;
    const int x_lo = (int)max((long long)tile_start - G.state[kStateReach] + 1, -0x7FFFFFFFll), x_hi = tile_end == 0x7FFFFFFF ? 0x7FFFFFFF : tile_end + 1;
    int lo = 0, hi = G.n_frags;
    if (sorted) wave_lower_bound2(G.frag, G.n_frags, x_lo, x_hi, lane, &lo, &hi);
    const int g = lane >> 4, j4 = (lane & 15) * 4;
    const int lane_pos = tile_start + j4;
    // ---- simple reads.  Blocks of 64 descriptors (this wave takes the blocks wid, wid + n_waves, ...), four sub-chunks of sixteen reads
    // a block; the loads of sub-chunk f + 1 are issued before sub-chunk f is added (explicit ping-pong over two register sets, every
    // load unconditional with its index clamped, so that the loop body is straight-line code and the waits are counted, not drained).
    struct Sub {
        uint32_t bw[4], qw[4], dw[4];
        int rel[4], lim[4], p0[4], nn[4];
    };
    const int n_blocks = (hi - lo + 63) >> 6;
    const int my_blocks = (n_blocks - wid + n_waves - 1) / n_waves;
    if (my_blocks > 0) {
        auto block_base = [&](int b) { return lo + (wid + min(b, my_blocks - 1) * n_waves) * 64; };
        auto load_desc = [&](int b) {
            const int base = block_base(b);
            return G.frag[base + min(lane, min(64, hi - base) - 1)];
        };
        // the loads of sub-chunk f (of this wave's sequence) from the descriptors d of its block
        auto issue = [&](const ReadDesc& d, int f, Sub& S) {
            const bool live = f < my_blocks * 4;
            const int base = block_base(f >> 2), cnt = min(64, hi - base), q = f & 3;
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int src = 16 * q + 4 * u + g;
                const int srcl = min(src, cnt - 1);
                const int pos0 = __shfl(d.pos0, srcl, 64);
                const uint32_t meta = (uint32_t)__shfl((int)d.meta, srcl, 64);
                const long long aoff = shfl64(d.aoff, srcl);
                const int len = (meta & kFragDeletion) ? 0 : (int)(meta & kDescLenMask);   // (deletion fragments: their own pass below)
                const int n = (live && src < cnt) ? len : 0;
                const int first = pos0 + frag_delta(aoff);                   // the fragment's first position
                const int floor_pos = base + srcl < G.n_floored_frags ? G.floor : 0;
                const int i_min = max(floor_pos - first, 0);  // (floor <= 2^31 - 1, first >= 1)
                const int s0 = lane_pos - first;              // index, in the aligned run, of the base on the lane's first locus
                S.lim[u] = max(n - i_min, 0);
                S.rel[u] = s0 - i_min;                        // (may wrap when nothing is valid: the wrapped value is far above lim)
                const long long at = (aoff & kFragAoffMask) + min(max(s0, -3), max(len - 1, -3));
                S.bw[u] = load_u32_unaligned(G.bases + at);
                S.qw[u] = load_u32_unaligned(G.quals + at);
                S.dw[u] = kDirs ? load_u32_unaligned(G.dirs + at)
                                : ((meta & kDescReverse) ? (uint32_t)PISCES_DIR_REVERSE * 0x01010101u : (uint32_t)PISCES_DIR_FORWARD * 0x01010101u);
                S.p0[u] = pos0;                                               // the READ's ends, for GetAnchorType: Position and
                S.nn[u] = (int)((aoff >> kFragSpanShift) & 0xFFFF) + 1;       // EndPosition - Position + 1
            }
        };
        auto consume = [&](const Sub& S) {
#pragma unroll
            for (int u = 0; u < 4; u++) {
#pragma unroll
                for (int st = 0; st < 4; st++) {
                    const int k = (st + g) & 3;
                    const uint32_t sh = (uint32_t)k * 8u;
                    on_base(j4 + k, (S.bw[u] >> sh) & 0xFFu, (S.qw[u] >> sh) & 0xFFu, (S.dw[u] >> sh) & 0xFFu,
                            (uint32_t)(S.rel[u] + k) < (uint32_t)S.lim[u], S.p0[u], S.nn[u]);
                }
            }
        };
        const int n_f = my_blocks * 4;
        ReadDesc dc = load_desc(0), dn = load_desc(1);
        Sub A, B;
        issue(dc, 0, A);
        for (int f = 0; f < n_f; f += 2) {
            __builtin_amdgcn_sched_barrier(0);
            issue(dc, f + 1, B);                  // (same block as f: four sub-chunks a block)
            __builtin_amdgcn_sched_barrier(0);
            consume(A);
            __builtin_amdgcn_sched_barrier(0);
            const bool next_block = (f & 3) == 2;
            ReadDesc da;
            da.pos0 = next_block ? dn.pos0 : dc.pos0;
            da.meta = next_block ? dn.meta : dc.meta;
            da.aoff = next_block ? dn.aoff : dc.aoff;
            dn = load_desc(((f + 2) >> 2) + 1);   // (the same descriptors again three times out of four: straight-line code)
            issue(da, f + 2, A);                  // (past the end: a repeat of the last sub-chunk with nothing valid)
            dc = da;
            __builtin_amdgcn_sched_barrier(0);
            consume(B);
        }
    }
    const int frag_bits = G.state[kStateFrags];
    // ---- the deletion fragments of the range: lane = locus.  A gap's positions count as deletions in the direction and in the anchor
    // bin of the base that closes it (the first base behind the gap); a deletion at the read's end or before its final soft clip counts
    // in the last bin (RegionStateManager.cs:143-154, 170-176, 199-213)
    if (frag_bits & 2) {
        for (int base = lo + wid * 64; base < hi; base += n_waves * 64) {
            const int cnt = min(64, hi - base);
            const ReadDesc d = G.frag[base + min(lane, cnt - 1)];
            unsigned long long dels = __ballot(lane < cnt && (d.meta & kFragDeletion));
            while (dels) {
                const int u = __builtin_ctzll(dels);
                dels &= dels - 1;
                const uint32_t meta = (uint32_t)__builtin_amdgcn_readlane((int)d.meta, u);
                const int pos0 = __builtin_amdgcn_readlane(d.pos0, u);
                const long long aoff = readlane64(d.aoff, u);
                const int first = pos0 + frag_delta(aoff), count = (int)(meta & kDescLenMask);
                const int floor_pos = base + u < G.n_floored_frags ? G.floor : 0;
                const uint32_t dir = kDirs ? (meta >> kFragDirShift) & 3u : ((meta & kDescReverse) ? (uint32_t)PISCES_DIR_REVERSE : (uint32_t)PISCES_DIR_FORWARD);
                const int anchor = (meta & kFragTerminal) ? PISCES_NUM_ANCHORS - 1
                                                          : walk_anchor_type(pos0 + (int)((aoff >> kFragSpanShift) & 0xFFFF), first + count, pos0);
                const int p = tile_start + lane;
                if (p >= max(max(first, floor_pos), 1) && p - first < count) on_obs(p, (uint32_t)PISCES_ALLELE_DEL, dir, anchor, 0xFFu);
            }
        }
    }
    // ---- reads whose fragments did not fit their fields: base by base (read_walk.h)
    if (frag_bits & 1) {
        int rlo = 0, rhi = G.n_reads;
        if (sorted) wave_lower_bound2(G.desc, G.n_reads, x_lo, x_hi, lane, &rlo, &rhi);
        walk_segment_complex<kDirs>(G, rlo, rhi, tile_start, min_bq, lane, wid, n_waves, on_obs, kDescGeneric);
    }
}

// ---- the flush kernel's own form of the walk over simple reads -------------------------------------------------------------------
// The general form above hands every base to a callback (18 VALU instructions a base with the histogram update of the call kernel:
// 37.5 M a launch at BASELINE config 2, 4 cycles each — the kernel was VALU-bound at 3 x the tuple kernel's time).  Rounds 3-4 classified
// four bases at once here (v_perm tables, 12.0 M VALU a launch); round 5 moved the classification to add time (encode_rows: one ROW CODE
// per base, low-quality << 5 | allele << 2) and gives lanes only to the parts of fragments that lie on the tile:
//   * PAIRS.  A wave takes 64 fragments with one 16-byte load per lane, applies the floor, and lists — in LDS, by two ballots and
//     v_mbcnt, no scan — the (fragment, half of the tile) pairs that hold at least one base: first the fragments that reach the left 32
//     loci, then those that reach the right 32.  A fragment that only brushes the tile, lies on one half of it (an amplicon's edge
//     inside the tile: half of a tile's fragments end there, the other half start there) or is only in the range because its READ may
//     reach the tile costs one pair or none, where rounds 3-4 gave every fragment of the range eight lanes (71 % of the lane-bases of
//     BASELINE config 2 were on a read, and a tile that straddles two amplicons took twice the time of one inside an amplicon: the
//     launch ended 19 us after its median tile).
//   * UNITS.  Sixteen pairs a unit: lane (q, jj) = (lane >> 2, lane & 3) holds, of pair q, the eight bases on the loci 8 jj .. 8 jj + 7
//     of the pair's half — one 8-byte load of codes (bases and qualities are not read by the flush at all: half the bytes), the
//     on-the-read mask from two entries of a nine-entry LDS table of 64-bit masks (bytes >= a), row = code | direction on the read and
//     row 24 (a row nobody reads) elsewhere, then one v_perm_b32 (row byte, column byte -> LDS address) and one ds_add_u32 per base.
//   * BANKS.  A lane adds its eight bytes in the order (s + q) & 7 (the row bytes are rotated by q & 7 bytes once; selectors and column
//     bytes are then per-lane constants, the half's 128-byte column offset is or-ed into the column bytes).  A ds_add_u32 is serviced
//     in two halves of 32 lanes, bank = locus mod 32 = 8 jj + step byte whatever the half of the tile: the eight lanes of a half-wave
//     that share jj hold q = 0..7 (8..15), eight different bytes — no two lanes of a half-wave on one bank, whatever the pairs.
//     (Rounds 3-4: lanes j and j + 4 of a fragment shared a bank: 31 % of the kernel's LDS cycles were conflicts.)
//   * Four units are in flight per wave (a ring of four register sets refilled in order), in GROUPS of four: a block of 64 fragments
//     is one group (up to 64 pairs) or two; the loop body is straight-line code around one uniform branch (the second group), so the
//     compiler counts its waits instead of draining the queue.  The pairs of block b + 1 are listed (into the other of two LDS lists)
//     before block b's units are consumed — its first group is what block b's last group prefetches — from descriptors requested a
//     block earlier.
constexpr int kMaskTab = 81;       // s_masktab[9 a + e] = the bytes a .. e - 1 of eight (a, e = 0..8; none when a >= e)
constexpr int kPairSlots = 128 + 32;   // a block's pairs (<= 64 left + 64 right, the right ones from a multiple of 16) + the filler behind both
// a pair: x = first valid byte * 8 (bits 0-9, in loci of the tile x 8: 0..512) | one past the last * 8 (bits 10-19) | half (bit 20) |
// direction (bits 21-22); y = index, in the segment's arrays (+ kSegmentPad), of the base that would stand on the tile's first locus
template <bool kDirs, typename OnObs>
__device__ __forceinline__ void walk_segment_fast(const SegmentView& G, int tile_start, uint32_t min_bq, int lane, int wid, int n_waves, char* hbytes,
                                                  const unsigned long long* s_masktab, uint2* pairs /* LDS, this wave's [kPairSlots] */, OnObs on_obs,
                                                  int pre_lo /* >= 0: the tile's fragment range is known (exchanged_tile priced it) */, int pre_hi,
                                                  long long* stamps = nullptr /* development: PISCES_STORE_TIMING */)
{
    if (G.n_frags <= 0) return;
    const int tile_end = tile_start + kTile - 1;
    const bool sorted = pre_lo >= 0 || G.state[kStateUnsorted] == 0;
    const int reach = G.state[kStateReach];
    const int x_lo = (int)max((long long)tile_start - reach + 1, -0x7FFFFFFFll), x_hi = tile_end == 0x7FFFFFFF ? 0x7FFFFFFF : tile_end + 1;
    int lo = 0, hi = G.n_frags;
    if (pre_lo >= 0) { lo = pre_lo; hi = pre_hi; }
    else if (sorted) {
        if (G.grid && !(G.state[kStateFrags] & kGridBadBit)) wave_lower_bound2_hinted(G.n_frags, x_lo, x_hi, lane, G.grid, G.grid_base, G.grid_n, &lo, &hi);
        else wave_lower_bound2(G.frag, G.n_frags, x_lo, x_hi, lane, &lo, &hi);
    }
#ifdef PISCES_STORE_TIMING
    if (stamps) { stamps[0] = wall_clock64(); stamps[1] = hi - lo; }
#endif
    const int q = lane >> 2, jj = lane & 3;
    const int rot = q & 7;
    // per-lane constants: the column bytes of the eight steps (byte s: LDS byte offset, inside a row, of the locus this lane stands on
    // in step s) for a pair on the left / on the right half of the tile; the selectors that rotate the eight row bytes right by rot bytes
    uint32_t col_lo_l = 0, col_hi_l = 0, rot_lo = 0, rot_hi = 0;
#pragma unroll
    for (int st = 0; st < 8; st++) {
        const uint32_t c = (uint32_t)((8 * jj + ((st + rot) & 7)) * (int)sizeof(int));
        const uint32_t from = (uint32_t)((st + rot) & 7);   // v_perm_b32(r1, r0, .): 0-3 = bytes of r0, 4-7 = bytes of r1
        if (st < 4) { col_lo_l |= c << (8 * st); rot_lo |= from << (8 * st); }
        else { col_hi_l |= c << (8 * (st - 4)); rot_hi |= from << (8 * (st - 4)); }
    }
    const uint32_t col_lo_r = col_lo_l | 0x80808080u, col_hi_r = col_hi_l | 0x80808080u;   // the right half's columns: + 128 bytes
    const int rel8_l = 64 * jj, rel8_r = 64 * jj + 256;   // the lane's first locus in the tile, x 8
    const uint8_t* const codes = G.codes - kSegmentPad;
    const uint8_t* const dirs = kDirs ? G.dirs - kSegmentPad : nullptr;
    const ReadDesc* const frag = G.frag;
    const int seg_floor = G.floor, n_floored_frags = G.n_floored_frags;
    const int n_blocks = (hi - lo + 63) >> 6;
    const int my_blocks = (n_blocks - wid + n_waves - 1) / n_waves;
    if (my_blocks > 0) {
        struct Unit { uint32_t cw[2], dw[2], ex; unsigned long long m; };   // sixteen pairs of ONE half: this lane's eight codes of one of them
        auto block_base = [&](int b) { return lo + (wid + min(b, my_blocks - 1) * n_waves) * 64; };
        auto load_desc = [&](int b) {
            const int base = block_base(b);
            return frag[base + min(lane, min(64, hi - base) - 1)];
        };
        // The pairs of block b (descriptors d, lane = fragment; past the wave's last block: none) into `list`: the left-half pairs from
        // entry 0, the right-half pairs from the next multiple of 16 (a unit's sixteen pairs are all on one half: which half is then
        // uniform, and the lanes' constants are picked by a branch, not computed), filler behind both.  *units_l / *units: units of
        // left pairs / units in all (<= 4 + 4).
        auto list_pairs = [&](int b, const ReadDesc& d, uint2* list, int* units_l, int* units) {
            const int base = block_base(b), cnt = min(64, hi - base);
            const int n = (b < my_blocks && lane < cnt && !(d.meta & kFragDeletion)) ? (int)(d.meta & kDescLenMask) : 0;
            const int floor_pos = base + lane < n_floored_frags ? seg_floor : 0;
            const int first = d.pos0 + frag_delta(d.aoff);   // the fragment's first position
            const int pos = max(first, floor_pos);            // first position still to count
            const int cut = min(pos - first, n);              // bases below the floor (pos - first >= 0)
            const int end = pos + (n - cut);                  // one past the last
            const uint32_t aoff = (uint32_t)(d.aoff & kFragAoffMask) + (uint32_t)cut + (uint32_t)kSegmentPad;   // index of the base on `pos`
            const int pr = min(max(pos - tile_start, 0), kTile), er = min(max(end - tile_start, 0), kTile);     // (differences of positive ints: no overflow)
            const bool left = pr < min(er, 32), right = max(pr, 32) < er;
            const uint32_t dirbits = (d.meta & kDescReverse) ? (uint32_t)PISCES_DIR_REVERSE : (uint32_t)PISCES_DIR_FORWARD;
            uint2 e;
            e.x = (uint32_t)(pr * 8) | ((uint32_t)(er * 8) << 10) | (dirbits << 24);
            e.y = aoff + (uint32_t)(tile_start - pos);        // (wraps when the read starts behind the tile's first locus: added back per lane)
            const unsigned long long ml = __ballot(left), mr = __ballot(right);
            const int nl = __popcll(ml), nr = __popcll(mr);
            const int ul = (nl + 15) >> 4, ur = (nr + 15) >> 4;
            const uint2 filler = make_uint2(0u, (uint32_t)kSegmentPad);   // a pair with nothing on the read
            if (lane < 16) { list[nl + lane] = filler; list[16 * ul + nr + lane] = filler; }
            if (left) list[__builtin_amdgcn_mbcnt_hi((uint32_t)(ml >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)ml, 0u))] = e;
            if (right) list[16 * ul + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mr >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mr, 0u))] = e;
            __builtin_amdgcn_wave_barrier();                            // (the list is read by other lanes of this wave: LDS operations of a wave execute in order)
            *units_l = ul;
            *units = ul + ur;
        };
        // A unit: the pair reaches its four lanes by one ds_read_b64 (the same address in all four: a broadcast), read for all eight units
        // of a block before the first unit of the block before it is consumed (behind a consumed unit's eight ds_add it would wait for them).
        // `live` (uniform): the unit holds pairs; else only its load is issued (the same memory operations on every path: the compiler's
        // wait counts stay exact — a branch with a load on one side made it drain the queue at the join).
        auto issue = [&](const uint2 e, Unit& U, bool live, auto right) {   // right: std::true_type / std::false_type (two copies of the code, each with its own lane constants)
            uint32_t at = (uint32_t)kSegmentPad;
            if (live) {
                const int rel8 = decltype(right)::value ? rel8_r : rel8_l;
                const int a8 = min(max((int)(e.x & 0x3FFu) - rel8, 0), 64), e8 = min(max((int)((e.x >> 10) & 0x3FFu) - rel8, 0), 64);   // the lane's bytes a .. e - 1 are on the read
                U.m = *reinterpret_cast<const unsigned long long*>(reinterpret_cast<const char*>(s_masktab) + (a8 * 9 + e8));
                at = e.y + (uint32_t)(rel8 >> 3);   // the byte on the lane's first locus (up to 7 before the read's first, 63 behind its last: kSegmentPad)
                U.ex = e.x;
            }
#if defined(PISCES_STORE_ABLATE) && PISCES_STORE_ABLATE == 5
            const unsigned long long c8 = 0x0004080C0004080Cull + (at & 1u);   // development ablation: no loads of codes
#else
            const unsigned long long c8 = load_u64_unaligned(codes + at);
#endif
            U.cw[0] = (uint32_t)c8; U.cw[1] = (uint32_t)(c8 >> 32);
            if (kDirs) {
                const unsigned long long d8 = load_u64_unaligned(dirs + at);
                U.dw[0] = (uint32_t)d8; U.dw[1] = (uint32_t)(d8 >> 32);
            }
        };
        auto consume = [&](const Unit& U, auto right) {
            const uint32_t m0 = (uint32_t)U.m, m1 = (uint32_t)(U.m >> 32);   // the lane's bytes on the read
            const uint32_t d0 = kDirs ? U.dw[0] : __builtin_amdgcn_perm(U.ex, U.ex, 0x03030303u), d1 = kDirs ? U.dw[1] : d0;   // (the pair's direction in every byte)
            const uint32_t r0 = (m0 & (U.cw[0] | d0)) | (~m0 & 0x18181818u), r1 = (m1 & (U.cw[1] | d1)) | (~m1 & 0x18181818u);
#if defined(PISCES_STORE_ABLATE) && PISCES_STORE_ABLATE == 2
            if ((r0 ^ r1) == 0x12345u) *reinterpret_cast<volatile int*>(hbytes) = 1;   // development ablation: no histogram update
            return;
#endif
            const uint32_t cl = decltype(right)::value ? col_lo_r : col_lo_l, ch = decltype(right)::value ? col_hi_r : col_hi_l;
            const uint32_t lo8 = __builtin_amdgcn_perm(r1, r0, rot_lo), hi8 = __builtin_amdgcn_perm(r1, r0, rot_hi);   // the eight row bytes rotated right by rot bytes
#pragma unroll
            for (int st = 0; st < 4; st++)   // (row byte st and column byte st -> LDS address)
                atomicAdd(reinterpret_cast<int*>(hbytes + __builtin_amdgcn_perm(lo8, cl, 0x0C0C0000u | ((uint32_t)(4 + st) << 8) | (uint32_t)st)), 1);
#pragma unroll
            for (int st = 0; st < 4; st++)
                atomicAdd(reinterpret_cast<int*>(hbytes + __builtin_amdgcn_perm(hi8, ch, 0x0C0C0000u | ((uint32_t)(4 + st) << 8) | (uint32_t)st)), 1);
        };
        // (uniform branches: the half picks lane constants, nothing is computed from it)
#define PISCES_ISSUE(E, R, K, UL, UN) { if ((K) < (UL)) issue(E, R, true, std::false_type()); else issue(E, R, (K) < (UN), std::true_type()); }
#define PISCES_CONSUME(R, K, UL, UN) { if ((K) < (UL)) consume(R, std::false_type()); else if ((K) < (UN)) consume(R, std::true_type()); }
        // One step a block.  Invariant at the top of step b: A / B hold the units 0-3 / 4-7 of block b (loads in flight; ul_c of its un_c
        // units are of left pairs), `raw` the descriptors of block b + 1, requested a step ago.  The step lists block b + 1, requests the
        // descriptors of block b + 2, and issues block b + 1's units as it consumes block b's.
        Unit A0, A1, A2, A3, B0, B1, B2, B3;
        int ul_c, un_c;
        list_pairs(0, load_desc(0), pairs, &ul_c, &un_c);
        ReadDesc raw = load_desc(1);
        {
            const uint2* const at = pairs + q;
            PISCES_ISSUE(at[0], A0, 0, ul_c, un_c) PISCES_ISSUE(at[16], A1, 1, ul_c, un_c) PISCES_ISSUE(at[32], A2, 2, ul_c, un_c) PISCES_ISSUE(at[48], A3, 3, ul_c, un_c)
            PISCES_ISSUE(at[64], B0, 4, ul_c, un_c) PISCES_ISSUE(at[80], B1, 5, ul_c, un_c) PISCES_ISSUE(at[96], B2, 6, ul_c, un_c) PISCES_ISSUE(at[112], B3, 7, ul_c, un_c)
        }
#ifdef PISCES_STORE_TIMING
#define PISCES_TICK(k) { __builtin_amdgcn_sched_barrier(0); const long long now_ = clock64(); if (stamps) stamps[k] += now_ - tick_; tick_ = now_; __builtin_amdgcn_sched_barrier(0); }
        long long tick_ = clock64();
#else
#define PISCES_TICK(k)
#endif
        for (int b = 0; b < my_blocks; b++) {
            int ul_n, un_n;
            list_pairs(b + 1, raw, pairs, &ul_n, &un_n);
            PISCES_TICK(2)
            raw = load_desc(b + 2);
            const uint2* const at = pairs + q;
            const uint2 e0 = at[0], e1 = at[16], e2 = at[32], e3 = at[48], f0 = at[64], f1 = at[80], f2 = at[96], f3 = at[112];
#ifdef PISCES_STORE_TIMING
            if ((e0.x ^ e1.x ^ e2.x ^ e3.x ^ f0.x ^ f1.x ^ f2.x ^ f3.x) == 0xFFFFFFFFu) stamps[5] += 1;   // (the entries are in: the tick below sees the LDS round trip)
#endif
            PISCES_TICK(4)
            __builtin_amdgcn_sched_barrier(0);
            PISCES_CONSUME(A0, 0, ul_c, un_c) PISCES_ISSUE(e0, A0, 0, ul_n, un_n)
            __builtin_amdgcn_sched_barrier(0);
            PISCES_CONSUME(A1, 1, ul_c, un_c) PISCES_ISSUE(e1, A1, 1, ul_n, un_n)
            __builtin_amdgcn_sched_barrier(0);
            PISCES_CONSUME(A2, 2, ul_c, un_c) PISCES_ISSUE(e2, A2, 2, ul_n, un_n)
            __builtin_amdgcn_sched_barrier(0);
            PISCES_CONSUME(A3, 3, ul_c, un_c) PISCES_ISSUE(e3, A3, 3, ul_n, un_n)
            __builtin_amdgcn_sched_barrier(0);
            PISCES_CONSUME(B0, 4, ul_c, un_c) PISCES_ISSUE(f0, B0, 4, ul_n, un_n)
            __builtin_amdgcn_sched_barrier(0);
            PISCES_CONSUME(B1, 5, ul_c, un_c) PISCES_ISSUE(f1, B1, 5, ul_n, un_n)
            __builtin_amdgcn_sched_barrier(0);
            PISCES_CONSUME(B2, 6, ul_c, un_c) PISCES_ISSUE(f2, B2, 6, ul_n, un_n)
            __builtin_amdgcn_sched_barrier(0);
            PISCES_CONSUME(B3, 7, ul_c, un_c) PISCES_ISSUE(f3, B3, 7, ul_n, un_n)
            __builtin_amdgcn_sched_barrier(0);
            ul_c = ul_n;
            un_c = un_n;
            PISCES_TICK(3)
#ifdef PISCES_STORE_TIMING
            if (stamps) stamps[5] += 1;
#endif
        }
#undef PISCES_TICK
#undef PISCES_ISSUE
#undef PISCES_CONSUME
    }
    const int frag_bits = G.state[kStateFrags];
    // ---- the deletion fragments of the range (if the segment has any): lane = locus; a gap's positions count as deletions in the
    // direction of the base that closed it, whatever their own quality (RegionStateManager.cs:170-176)
    if (frag_bits & 2) {
        for (int base = lo + wid * 64; base < hi; base += n_waves * 64) {
            const int cnt = min(64, hi - base);
            const ReadDesc d = G.frag[base + min(lane, cnt - 1)];
            unsigned long long dels = __ballot(lane < cnt && (d.meta & kFragDeletion));
            while (dels) {
                const int u = __builtin_ctzll(dels);
                dels &= dels - 1;
                const uint32_t meta = (uint32_t)__builtin_amdgcn_readlane((int)d.meta, u);
                const int first = __builtin_amdgcn_readlane(d.pos0, u) + frag_delta(readlane64(d.aoff, u));
                const int count = (int)(meta & kDescLenMask);
                const int floor_pos = base + u < G.n_floored_frags ? G.floor : 0;
                const uint32_t dir = kDirs ? (meta >> kFragDirShift) & 3u : ((meta & kDescReverse) ? (uint32_t)PISCES_DIR_REVERSE : (uint32_t)PISCES_DIR_FORWARD);
                const int p = tile_start + lane;
                if (p >= max(first, floor_pos) && p - first < count)
                    atomicAdd(reinterpret_cast<int*>(hbytes) + HistLinear::idx(PISCES_ALLELE_DEL, (int)dir, lane), 1);
            }
        }
    }
    // ---- reads whose fragments did not fit their fields: base by base, as the general walk takes reads with insertions / deletions
    if (frag_bits & 1) {
        int rlo = 0, rhi = G.n_reads;
        if (sorted) wave_lower_bound2(G.desc, G.n_reads, x_lo, x_hi, lane, &rlo, &rhi);
        walk_segment_complex<kDirs>(G, rlo, rhi, tile_start, (int)min_bq, lane, wid, n_waves, on_obs, kDescGeneric);
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

// The tiles of a run of consecutive blocks with no interval clipping, by value: tile t is the (t % tiles_per_block)-th tile_loci-locus tile of
// block first_key + t / tiles_per_block (GetBlockKey grid, RegionStateManager.cs:385-391).  A flush of such a run uploads no geometry.
struct RegularTiles {
    int32_t first_key, block_size, tiles_per_block;
    int32_t tile_loci;   // loci a tile (<= 64; 0 = 64): tiles_per_block = ceil(block_size / tile_loci), a block's last tile takes what is left
};
__device__ __forceinline__ PiscesTile regular_tile(const RegularTiles& R, int t)
{
    const int b = t / R.tiles_per_block, i = t - b * R.tiles_per_block;
    const int step = R.tile_loci > 0 ? R.tile_loci : kTile;
    PiscesTile tile;
    tile.start_position = (R.first_key - 1 + b) * R.block_size + 1 + i * step;
    tile.n_loci = min(step, R.block_size - i * step);
    tile.tuple_begin = tile.tuple_end = 0;
    return tile;
}

// consecutive tiles on one XCD (workgroups go round the 8 XCDs in dispatch order): neighbouring tiles read the same reads — their
// lines are then in that XCD's L2 the second time
__device__ __forceinline__ int xcd_tile_of_block(int b, int n)
{
    const int x = b & 7, j = b >> 3, q = n >> 3, rem = n & 7;
    return x * q + min(x, rem) + j;
}

// THE ORDER OF A LAUNCH'S TILES (launches of several tiles a CU).  A launch ends with the CU that was dealt the most fragments: a tile
// across two amplicons lists twice the fragments of one inside an amplicon, and with the tiles taken in position order the tiles one
// CU is dealt lie a fixed distance apart — on a periodic panel they are all of one kind (config 2: 3 000 to 6 000 fragments a CU, mean
// 4 400; the launch lasted as long as the 6 000).  tile_order_kernel prices every tile (the fragments in its range: the two grid
// entries walk_segment_fast's search reads, + the bucketed tuples) and, inside each XCD's share of the tiles (xcd_tile_of_block: neighbours
// stay on one L2), orders them by falling cost, the rounds of `cus` workgroups alternately forwards and backwards: the workgroups
// of a round go to different CUs, so every CU gets one tile of every cost stratum.  order[b] = the tile of workgroup b.  Which tile a
// workgroup takes changes no result (a tile's records lie in the tile's own slots).  Measured (config 2, 1 600 tiles, profiles/
// r05_tile_order.txt): the flush kernel 32.0-32.7 us against 35.9-37.8 in position order (fragments of the fullest CU 5 000 against
// 6 000) — and 8.6-9.6 us for this launch in front of it (two dependent round trips on an idle chip and the launch itself), so a
// flush is slower by ~5 us: NOT the default (PISCES_HIP_TILE_ORDER=1 asks for it); the default is exchanged_tile below.
__device__ __forceinline__ int tile_cost(const StoreView& S, const PiscesTile& tile)
{
    int cost = (int)min((long long)(tile.tuple_end - tile.tuple_begin) >> 4, 0x3FFFFFll);
    const int tile_end = tile.start_position + kTile - 1;
    for (int sg = 0; sg < S.n_segments; sg++) {
        const SegmentView& G = S.seg[sg];
        if (G.n_frags <= 0 || G.state[kStateUnsorted] != 0 || !G.grid || (G.state[kStateFrags] & kGridBadBit)) continue;   // (the same for every tile)
        const int reach = G.state[kStateReach];
        const int x_lo = (int)max((long long)tile.start_position - reach + 1, -0x7FFFFFFFll), x_hi = tile_end == 0x7FFFFFFF ? 0x7FFFFFFF : tile_end + 1;
        int e[2];
#pragma unroll
        for (int k2 = 0; k2 < 2; k2++) {
            const long long k = (long long)max(k2 ? x_hi : x_lo, 0) - G.grid_base;
            e[k2] = k < 0 ? 0 : G.n_frags;
            if (k >= 0 && k < G.grid_n) e[k2] = min(G.grid[k], G.n_frags);
        }
        cost += min(max(e[1] - e[0], 0), 0x3FFFFF);
    }
    return cost;
}
constexpr int kTileFixedCost = 1264;   // a tile's search, pipeline fill / drain and call phase in units of one listed fragment (13.4 us against 10.6 ns, config 2)
constexpr int kOrderCached = 2048;      // tiles of an XCD whose costs tile_order_kernel keeps in LDS (beyond: priced again in each pass)
__global__ __launch_bounds__(256) void tile_order_kernel(StoreView S, const PiscesTile* __restrict__ tiles /* or nullptr: R */, RegularTiles R, int32_t n_tiles,
                                                         int32_t cus /* CUs of an XCD */, int32_t* __restrict__ order)
{
    __shared__ int s_max, s_cnt[256], s_at[256], s_wave[4], s_cost[kOrderCached];
    const int x = (int)blockIdx.x, q = n_tiles >> 3, rem = n_tiles & 7, tid = (int)threadIdx.x;
    const int first = x * q + min(x, rem), cnt = q + (x < rem ? 1 : 0);
    if (tid == 0) s_max = 1;
    s_cnt[tid] = 0;
    __syncthreads();
    auto price = [&](int j) { return kTileFixedCost + tile_cost(S, tiles ? tiles[first + j] : regular_tile(R, first + j)); };
    for (int j = tid; j < cnt; j += 256) {
        const int c = price(j);
        if (j < kOrderCached) s_cost[j] = c;
        atomicMax(&s_max, c);
    }
    __syncthreads();
    const long long mx = s_max;
    auto bucket_of = [&](int j) { const int c = j < kOrderCached ? s_cost[j] : price(j); return 255 - (int)((long long)c * 255 / mx); };   // the dearest tiles first
    for (int j = tid; j < cnt; j += 256) atomicAdd(&s_cnt[bucket_of(j)], 1);
    __syncthreads();
    {   // exclusive scan of the 256 bucket counts
        const int v = s_cnt[tid];
        int a = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int u = __shfl_up(a, d); if ((tid & 63) >= d) a += u; }
        if ((tid & 63) == 63) s_wave[tid >> 6] = a;
        __syncthreads();
        int base = 0;
        for (int w = 0; w < (tid >> 6); w++) base += s_wave[w];
        s_at[tid] = base + a - v;
    }
    __syncthreads();
    // workgroup j of the XCD runs on CU j % cus (the dispatcher deals them in turn): the r CUs that get one tile more than the others take
    // the cheapest tiles, the others share the dearest — each group by rounds that run alternately forwards and backwards
    const int s = cnt / cus, r = cnt - s * cus, n6 = cus - r, n_top = s * n6;
    for (int j = tid; j < cnt; j += 256) {
        const int rank = atomicAdd(&s_at[bucket_of(j)], 1);
        int k, c;
        if (rank < n_top) { k = rank / n6; const int i = rank - k * n6; c = r + ((k & 1) ? n6 - 1 - i : i); }
        else { const int rr = rank - n_top; k = rr / r; const int i = rr - k * r; c = (k & 1) ? r - 1 - i : i; }
        order[(k * cus + c) * 8 + x] = first + j;
    }
}

// A little of that without a launch in front: the workgroups of one round of an XCD trade tiles inside small groups.  Workgroup j of
// an XCD runs on CU j % cus (the dispatcher deals them in turn), so the first r = cnt % cus CUs get one tile more than the others; each
// of them forms a group with g - 1 of the others (g = cus / r, at most 8).  Every workgroup of a group prices the g tiles of the group
// (the 2 g grid entries are one load instruction of the wave, lanes 0 .. g - 1 the low ends, 32 .. 32 + g - 1 the high ends: a tile's
// start keeps its two dependent round trips, and the range of the tile taken goes to walk_segment_fast, which then has no search of
// its own) and all come to the same ranking: the CU with the extra tile takes the cheapest, the others the rest, rotated by the round so
// that over the rounds each CU gets every rank.  r = 0: groups of four, rotated.  Prices are the fragment counts of the first segment
// (one segment is the rule); no usable grid there: position order.  (Groups over all rounds of their CUs — up to 32 tiles priced by
// every workgroup, the dearest first on each CU — measured slower than position order: 38.6 against 36.0 us.)
__device__ __forceinline__ int exchanged_tile(const StoreView& S, const PiscesTile* __restrict__ tiles, const RegularTiles& R, int n_tiles, int cus, int b, int lane,
                                          int* pre_lo /* -1, or: the first segment's fragment range of the tile, both ends */, int* pre_hi)
{
    *pre_lo = -1;
    *pre_hi = 0;
    const int x = b & 7, j = b >> 3, q = n_tiles >> 3, rem = n_tiles & 7;
    const int first = x * q + min(x, rem), cnt = q + (x < rem ? 1 : 0), own = first + j;
    const int s = cnt / cus, r = cnt - s * cus, k = j / cus, c = j - k * cus;
    if (k >= s || S.n_segments < 1) return own;   // (the last, partial round keeps its tiles)
    int g, grp, m;
    if (r > 0) {
        g = min(cus / r, 8);
        if (g < 2) return own;
        if (c < r) { grp = c; m = 0; }
        else { const int d = c - r; grp = d / (g - 1); m = 1 + d - grp * (g - 1); if (grp >= r) return own; }
    } else {
        g = 4; grp = c >> 2; m = c & 3;
        if (grp * 4 + 4 > cus) return own;
    }
    const SegmentView& G = S.seg[0];
    if (G.n_frags <= 0 || G.state[kStateUnsorted] != 0 || !G.grid || (G.state[kStateFrags] & kGridBadBit)) return own;
    const int reach = G.state[kStateReach];
    const int i = lane & 31;   // the member this lane prices
    int a = 0, t_i = own;
    if (i < g) {
        const int c_i = r > 0 ? (i == 0 ? grp : r + grp * (g - 1) + i - 1) : grp * 4 + i;
        t_i = first + k * cus + c_i;
        const int start = tiles ? tiles[t_i].start_position : regular_tile(R, t_i).start_position;
        const int tile_end = start + kTile - 1;
        const int xq = (lane >> 5) ? (tile_end == 0x7FFFFFFF ? 0x7FFFFFFF : tile_end + 1) : (int)max((long long)start - reach + 1, -0x7FFFFFFFll);
        const long long kk = (long long)max(xq, 0) - G.grid_base;
        a = kk < 0 ? 0 : G.n_frags;
        if (kk >= 0 && kk < G.grid_n) a = min(G.grid[kk], G.n_frags);
    }
    const int cost = __shfl_down(a, 32) - a;   // (lanes 0 .. g - 1)
    int rank = 0;                               // by rising price
    for (int u = 0; u < g; u++) {
        const int cu = __builtin_amdgcn_readlane(cost, u);
        rank += (cu < cost || (cu == cost && u < i)) ? 1 : 0;
    }
    const int want = r > 0 ? (m == 0 ? 0 : 1 + (m - 1 + k) % (g - 1)) : (m + k) & 3;
    const unsigned long long hit = __ballot(lane < g && rank == want);
    if (hit == 0ull) return own;
    const int chosen = (int)__ffsll((long long)hit) - 1;
    *pre_lo = __builtin_amdgcn_readlane(a, chosen);
    *pre_hi = __builtin_amdgcn_readlane(a, chosen + 32);
    return __builtin_amdgcn_readlane(t_i, chosen);
}

// byte offset of an allele's first row in a histogram region, by read base (AlleleHelper.GetAlleleType, AlleleHelper.cs:13-32:
// anything but A, C, G, T is an N); row = allele * 4 + direction, 64 int32 columns a row
__device__ __forceinline__ uint32_t allele_row_bytes(uint32_t c)
{
    return (c == 'A' ? (uint32_t)PISCES_ALLELE_A : c == 'C' ? (uint32_t)PISCES_ALLELE_C : c == 'G' ? (uint32_t)PISCES_ALLELE_G
            : c == 'T' ? (uint32_t)PISCES_ALLELE_T : (uint32_t)PISCES_ALLELE_N) * (4u * kWaveRow * (uint32_t)sizeof(int));
}

// the same without a table: ((c >> 1) & 3) tells A, C, T, G apart (0, 1, 2, 3); the letter that index stands for is compared with c.
// (A look-up in LDS put a dependent LDS round trip in front of every ds_add: 74 of the kernel's first 91 us.)
__device__ __forceinline__ uint32_t allele_row_bytes_of_base(uint32_t c)
{
    const uint32_t t8 = (c << 2) & 0x18u;                  // 8 * ((c >> 1) & 3)
    const uint32_t letter = (0x47544341u >> t8) & 0xFFu;   // 'A', 'C', 'T', 'G'
    const uint32_t code = (0x01030200u >> t8) & 0xFFu;     // AlleleType of that letter: A 0, C 2, T 3, G 1
    return (c == letter ? code : (uint32_t)PISCES_ALLELE_N) * (4u * kWaveRow * (uint32_t)sizeof(int));
}

// NW waves walk a tile's reads (blocks of 64 descriptors in turn).  A launch that fills the chip anyway takes 2 (or 1 beyond ~8 k tiles);
// a launch of a few tiles — one 1000-locus block of the streaming protocol is 16 — takes 8, so that a tile's ~700 reads are eleven
// blocks side by side and not six one after the other per wave.  The call phase is the wave kernel's: wave 0 the Reference records and
// the directory, the last wave the variant records; the other waves are done when the histogram is.
template <int NW>
__global__ __launch_bounds__(64 * NW, NW == 1 ? PISCES_WAVE_OCC : PISCES_WAVE2_OCC) void call_store_tiles_kernel(
    StoreView S, const uint32_t* __restrict__ tuples, const PiscesTile* __restrict__ tiles /* or nullptr: R */, RegularTiles R, int32_t n_tiles,
    const int32_t* __restrict__ order /* tile_order_kernel's, or nullptr: position order */, int32_t trade_cus /* > 0: exchanged_tile */, int32_t walk_prio, const uint8_t* __restrict__ ref, int32_t ref_start, int64_t ref_len, PiscesCalledAllele* __restrict__ records,
    PiscesTileResult* __restrict__ tile_results, DeviceParams P, const DeviceParams* __restrict__ Pd)
{
    __shared__ __attribute__((aligned(16))) int hist[2 * kWaveRegion];   // [region: quality-passing / low-quality][allele * 4 + direction][locus]
    __shared__ uint8_t s_refwin[kRefWin];
    __shared__ uint8_t s_vmask[kTile];
    __shared__ unsigned long long s_masktab[kMaskTab];
    __shared__ uint2 s_pairs[NW][kPairSlots];

    if ((int)blockIdx.x >= n_tiles) return;
    int pre_lo = -1, pre_hi = 0;
#ifdef PISCES_STORE_NO_SWIZZLE
    const int t = (int)blockIdx.x;
#else
    const int t = order ? order[blockIdx.x] : trade_cus > 0 ? exchanged_tile(S, tiles, R, n_tiles, trade_cus, (int)blockIdx.x, (int)(threadIdx.x & 63), &pre_lo, &pre_hi)
                                            : xcd_tile_of_block((int)blockIdx.x, n_tiles);
#endif
    const int l = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const PiscesTile tile = tiles ? tiles[t] : regular_tile(R, t);
#ifdef PISCES_STORE_TIMING
    const long long tc0 = wall_clock64();   // 100 MHz, chip-global
    long long stamps[6] = {0, 0, 0, 0, 0, 0};   // [2..5]: shader-clock cycles this wave spent issuing / consuming / trimming, loop iterations
#endif
#if defined(PISCES_STORE_ABLATE) && PISCES_STORE_ABLATE == 4
    if (tile.n_loci > 0) { if (threadIdx.x == 0) { tile_results[t].record_begin = 0; tile_results[t].n_records = 0; tile_results[t].n_called = 0; tile_results[t].n_candidate_loci = 0; } return; }   // development ablation: nothing
#endif
    {
        int4* h4 = reinterpret_cast<int4*>(hist);
        for (int i = threadIdx.x; i < 2 * kWaveRegion / 4; i += 64 * NW) h4[i] = make_int4(0, 0, 0, 0);
        for (int i = threadIdx.x; i < kRefWin; i += 64 * NW) {
            const int64_t ri = (int64_t)tile.start_position - kRefMargin + i - ref_start;
            s_refwin[i] = (ri >= 0 && ri < ref_len) ? ref[ri] : (uint8_t)0;
        }
        for (int i = threadIdx.x; i < kMaskTab; i += 64 * NW) {
            const int a = i / 9, e = i - 9 * a;
            const unsigned long long ge_a = a >= 8 ? 0ull : ~0ull << (8 * a), ge_e = e >= 8 ? 0ull : ~0ull << (8 * e);
            s_masktab[i] = ge_a & ~ge_e;
        }
        __syncthreads();
    }
    // a launch whose workgroups are all resident at once: walking waves win the issue arbitration over waves in their call phase (as in
    // call_tiles_wave_kernel) — the launch ends with the slowest tile, and a tile that still walks has its whole call phase ahead of it.
    // Config 2, 1 600 tiles: 32.9-33.4 us against 35.5-37.4 (best launches 31.3-32.0 against 34.0-34.4); a launch of several rounds
    // (4 688 tiles) loses 2.4 us with it, so the host asks for it in single-round launches only
    if (walk_prio) __builtin_amdgcn_s_setprio(3);
    const uint32_t min_bq = (uint32_t)min(max(P.min_bq, 0), 127);   // (the host routes larger thresholds through the counts in HBM)
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
    auto on_obs = [&](int position, uint32_t allele, uint32_t dir, int, uint32_t qual) {
        const uint32_t off = ((qual < min_bq) ? kRegionBytes : 0u) +
                             (uint32_t)HistLinear::idx((int)allele, (int)dir, position - tile.start_position) * (uint32_t)sizeof(int);
        atomicAdd(reinterpret_cast<int*>(hbytes + off), 1);
    };
    for (int sg = 0; sg < S.n_segments; sg++) {
        const SegmentView& G = S.seg[sg];
#ifdef PISCES_STORE_TIMING
        if (G.dirs) walk_segment_fast<true>(G, tile.start_position, min_bq, l, wid, NW, hbytes, s_masktab, s_pairs[wid], on_obs, sg == 0 ? pre_lo : -1, pre_hi, stamps);
        else walk_segment_fast<false>(G, tile.start_position, min_bq, l, wid, NW, hbytes, s_masktab, s_pairs[wid], on_obs, sg == 0 ? pre_lo : -1, pre_hi, stamps);
#else
        if (G.dirs) walk_segment_fast<true>(G, tile.start_position, min_bq, l, wid, NW, hbytes, s_masktab, s_pairs[wid], on_obs, sg == 0 ? pre_lo : -1, pre_hi);
        else walk_segment_fast<false>(G, tile.start_position, min_bq, l, wid, NW, hbytes, s_masktab, s_pairs[wid], on_obs, sg == 0 ? pre_lo : -1, pre_hi);
#endif
    }
#ifdef PISCES_STORE_TIMING
    const long long tc_walk = wall_clock64();
#endif
    __syncthreads();
#if defined(PISCES_STORE_ABLATE) && (PISCES_STORE_ABLATE == 1 || PISCES_STORE_ABLATE == 2 || PISCES_STORE_ABLATE >= 5)
    if (threadIdx.x == 0) { tile_results[t].record_begin = 0; tile_results[t].n_records = hist[5 + l] & 0; tile_results[t].n_called = 0; tile_results[t].n_candidate_loci = 0; }   // development ablation: no call phase
    return;
#endif
    if (NW > 2 && wid != 0 && wid != NW - 1) return;
    if (walk_prio) __builtin_amdgcn_s_setprio(0);
    call_phase_wave<(NW == 1 ? 1 : 2), HistLinear>(hist, s_refwin, s_vmask, tile, t, l, NW == 1 ? 0 : (wid == NW - 1 ? 1 : 0), ref, ref_start, ref_len, records, tile_results, P
#ifdef PISCES_TIMING
                                    , 0ll, 0ll
#endif
                                    );
#ifdef PISCES_STORE_TIMING
    if (threadIdx.x == 0) {   // development instrumentation: chip-global clock stamps in the tile directory (the records are not usable then)
        const long long tc_end = wall_clock64();
        int* tr = reinterpret_cast<int*>(&records[(int64_t)t * kSlotsPerTile + kSlotsPerTile - 1]);   // the tile's last record slot (a T variant on locus 63, if there was one, is lost)
        tr[0] = (int)(tc0 & 0x3FFFFFFF);
        tr[1] = (int)(stamps[0] & 0x3FFFFFFF);      // search done (first segment)
        tr[2] = (int)(tc_walk & 0x3FFFFFFF);
        tr[3] = (int)(tc_end & 0x3FFFFFFF);
        tr[4] = (int)stamps[1];                     // reads in the tile's range
        tr[7] = (int)stamps[2]; tr[8] = (int)stamps[3]; tr[9] = (int)stamps[4]; tr[10] = (int)stamps[5];   // wave 0: cycles issuing / consuming / trimming, iterations
        tr[5] = (int)__builtin_amdgcn_s_getreg(63492);    // HW_REG_HW_ID
        tr[6] = (int)__builtin_amdgcn_s_getreg(63508);    // HW_REG_XCC_ID
        tr[11] = (int)blockIdx.x;
    }
#endif
}

// Ordered compaction of a small launch (<= 64 tiles: the blocks of one flush of the streaming protocol): one wave per tile; every
// wave scans the 64 record counts itself (no second kernel, no barrier) and copies the valid slots of its tile in order.  out[0] is a
// header {records, alleles called}, the records start at out[1] — and `out` may be PINNED HOST MEMORY: the sorted records of a block
// (~64 KB) then cross PCIe as the kernel's own stores, and the flush has no copy operation behind its kernels at all
// (scan_tile_counts_kernel + gather_records_kernel + two copies for any number of tiles are four stream operations).
__global__ __launch_bounds__(64) void compact_small_kernel(const PiscesCalledAllele* __restrict__ records, const PiscesTileResult* __restrict__ tr,
                                                           int32_t n_tiles, PiscesCalledAllele* __restrict__ out, int32_t capacity)
{
    const int l = threadIdx.x, t = blockIdx.x;
    const int v = l < n_tiles ? tr[l].n_records : 0;
    int incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int y = __shfl_up(incl, d, 64);
        if (l >= d) incl += y;
    }
    const int my_off = __shfl(incl - v, t, 64);
    if (t == 0) {
        int called = l < n_tiles ? tr[l].n_called : 0;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) called += __shfl_xor(called, d, 64);
        if (l == 63) reinterpret_cast<int4*>(out)[0] = make_int4(incl, called, 0, 0);
    }
    const uint32_t nib = (tr[t].valid[l >> 3] >> ((l & 7) * 4)) & 0xFu;
    int x = __popc(nib);
    const int mine = x;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int y = __shfl_up(x, d, 64);
        if (l >= d) x += y;
    }
    int64_t dst = (int64_t)my_off + (x - mine);
    const PiscesCalledAllele* src = records + (int64_t)tr[t].record_begin + l * 4;
    for (int k = 0; k < 4; k++) {
        if (!(nib & (1u << k))) continue;
        if (dst < capacity) copy_record(&out[1 + dst], &src[k]);
        dst++;
    }
}

// The same walk into the anchor-resolved tensor (RegionState._alleleCounts, RegionState.cs:57) and, with sumq, the base-quality sums
// (RegionState._sumOfAlleleBaseQualities :61): what accumulate_tiles_kernel makes of tuples, here from the reads (and the tuples).
// gridDim.y workgroups share a tile (each takes every gridDim.y-th block of 64 fragments into a histogram of its own and adds what it
// counted to the tensor with global atomics): the few tiles of one flush of the streaming protocol — 16 for a 1000-locus block, ~2 850
// reads each at 2000x — are then spread over the chip instead of sixteen CUs (281 us a block with one workgroup of 1024 threads a tile).
__global__ __launch_bounds__(1024) void accumulate_store_tiles_kernel(StoreView S, const uint32_t* __restrict__ tuples, const PiscesTile* __restrict__ tiles,
                                                                        int32_t n_tiles, int32_t* __restrict__ counts, int32_t min_bq_,
                                                                        unsigned long long* __restrict__ sumq, const ulonglong2* __restrict__ bq_lut)
{
    __shared__ int hist[kTile * kAnchStride];
    const int t = blockIdx.x;
    if (t >= n_tiles) return;
    const PiscesTile tile = tiles[t];
    for (int i = threadIdx.x; i < kTile * kAnchStride; i += (int)blockDim.x) hist[i] = 0;
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
    const int sub = (int)blockIdx.y, n_sub = (int)gridDim.y;
    for (int64_t i = tile.tuple_begin + (int64_t)sub * blockDim.x + threadIdx.x; i < tile.tuple_end; i += (int64_t)blockDim.x * n_sub) {
        const uint32_t v = tuples[i];
        add(PISCES_TUPLE_LOCUS(v), PISCES_TUPLE_ALLELE(v), PISCES_TUPLE_DIR(v), PISCES_TUPLE_ANCHOR(v), v >> 24);
    }
    const int lane = threadIdx.x & 63, waves = (int)blockDim.x / 64, wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) * n_sub + sub;
    walk_store(S, tile.start_position, (int)min_bq, lane, wid, waves * n_sub,
               [&](int locus, uint32_t base, uint32_t qual, uint32_t dir, bool valid, int pos0, int n_aligned) {
                   if (!valid) return;
                   // GetAnchorType (RegionStateManager.cs:83-116); EndPosition of a read of one aligned run = pos0 + run - 1
                   const int anchor = walk_anchor_type(pos0 + n_aligned - 1, tile.start_position + locus, pos0);
                   add((uint32_t)locus, walk_allele_type((uint8_t)base), dir, (uint32_t)(anchor < 0 ? 0 : anchor), qual);
               },
               [&](int position, uint32_t allele, uint32_t dir, int anchor, uint32_t qual) {
                   add((uint32_t)(position - tile.start_position), allele, dir, (uint32_t)anchor, qual);
               });
    __syncthreads();
    int32_t* __restrict__ dst = counts + (int64_t)t * kTile * PISCES_COUNTS_PER_LOCUS;
    const int n = tile.n_loci * PISCES_COUNTS_PER_LOCUS;
    for (int g = threadIdx.x; g < n; g += (int)blockDim.x) {
        const int lo = g / PISCES_COUNTS_PER_LOCUS, c = g - lo * PISCES_COUNTS_PER_LOCUS;
        const int v = hist[lo * kAnchStride + c];
        if (!v) continue;
        if (n_sub > 1) atomicAdd(dst + g, v);
        else dst[g] += v;
    }
}

}  // namespace pisces
