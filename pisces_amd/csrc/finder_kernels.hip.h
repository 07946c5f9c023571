// finder_kernels.hip.h — ICandidateVariantFinder.FindCandidates on the device, gfx950.
//
// The walk is finder_walk.h (the same source the host entry points compile): one lane per read, every candidate the read
// gives — insertions and deletions always, SNVs and MNVs when MNV calling is on — written as a 64-byte record in read
// order, with the read bases its ALT allele takes inline (longer ones in a byte pool).  Nothing of the read stays on the
// host: pisces_hip_add_reads enqueues this next to expand_reads_kernel and picks the records up when they are needed.
//
//   find_count_kernel   MNV calling on: the number of candidates (and pool bytes) of every read
//   found_scan_kernel   exclusive scan of those counts -> the first record slot of every read
//   find_emit_kernel    the records.  MNV calling off: the host reserved one slot per I / D operation of the read's CIGAR
//                       (an upper bound that is exact unless a quality gate fails); slots a read does not use become holes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "finder_walk.h"
#include "stream_kernels.hip.h"

namespace pisces {

constexpr int kFoundInline = 32;   // read bases of an ALT allele kept inside the record
constexpr uint8_t kFoundHole = 0xFF;

struct DevFound {   // what leaves the device: 64 bytes
    FoundCandidate c;
    int32_t read;             // index of the read in its batch (records are in read order, then in order of discovery)
    int32_t pool_offset;      // the read bases in the byte pool when c.length > kFoundInline (insertions / MNVs), else -1
    uint8_t alt[kFoundInline];
};
static_assert(sizeof(DevFound) == 64, "DevFound is 64 bytes");

__device__ __forceinline__ ReadView dev_read_view(const DevReadBatch& b, const uint8_t* del_dirs, int r)
{
    ReadView v;
    const int c0 = b.cigar_offset[r], s0 = b.seq_offset[r];
    v.position = b.position[r];
    v.n_cigar = b.cigar_offset[r + 1] - c0;
    v.cigar_op = b.cigar_op + c0;
    v.cigar_len = b.cigar_len + c0;
    v.read_len = b.seq_offset[r + 1] - s0;
    v.bases = b.bases + s0;
    v.quals = b.quals + s0;
    v.dirs = b.dirs ? b.dirs + s0 : nullptr;
    v.del_dirs = del_dirs ? del_dirs + 2 * (size_t)c0 : nullptr;
    v.is_reverse = (b.flags[r] & 1) ? 1 : 0;
    return v;
}

// The bases of an M operation four at a time: one dword each of read bases, qualities and reference bases (any alignment: gfx950 runs
// with unaligned global access on), the next word's loads issued before the state machine works through the current one — two
// register sets in turn, straight-line code, every load unconditional with its address pulled back inside the arrays at the
// operation's end (a guarded load becomes a branch with a wait for everything outstanding at the join, a register copy of a loaded
// value a wait where it stands).  With one lane per read the byte-wise walk is a chain of dependent loads: 450 a read of 150 bases,
// 215-230 us for 80 000 reads whatever the arithmetic.
struct WordBases {
    struct Word { uint32_t b, q, f; int sh; };   // sh: bits to shift right (the word was loaded `sh / 8` bytes early, at the end of the arrays)
    template <typename Step>
    __device__ __forceinline__ static void for_each(const ReadView& r, const uint8_t* ref, int op_read0, int op_ref0, int n, Step& step)
    {
        const uint8_t* pb = r.bases + op_read0; const uint8_t* pq = r.quals + op_read0; const uint8_t* pf = ref + op_ref0;
        if (n < 4) {   // (the arrays are only known to hold n bytes from here)
            for (int i = 0; i < n; i++) step(i, pb[i], pf[i], pq[i]);
            return;
        }
        const int n_words = (n + 3) >> 2;
        auto load = [&](int w, Word& W) {
            const int o = 4 * min(w, n_words - 1), oc = min(o, n - 4);
            __builtin_memcpy(&W.b, pb + oc, 4);
            __builtin_memcpy(&W.q, pq + oc, 4);
            __builtin_memcpy(&W.f, pf + oc, 4);
            W.sh = 8 * (o - oc);
        };
        auto walk = [&](int w, const Word& W) {
            const uint32_t b = W.b >> W.sh, q = W.q >> W.sh, f = W.f >> W.sh;
#pragma unroll 1
            for (int k = 0; k < 4; k++) {
                const int i = 4 * w + k;
                if (i < n) step(i, (uint8_t)(b >> (8 * k)), (uint8_t)(f >> (8 * k)), (uint8_t)(q >> (8 * k)));
            }
        };
        Word A, B;
        load(0, A);
        for (int w = 0; w < n_words; w += 2) {
            __builtin_amdgcn_sched_barrier(0);
            load(w + 1, B);
            __builtin_amdgcn_sched_barrier(0);
            walk(w, A);
            __builtin_amdgcn_sched_barrier(0);
            load(w + 2, A);
            __builtin_amdgcn_sched_barrier(0);
            if (w + 1 < n_words) walk(w + 1, B);
        }
    }
};

__device__ __forceinline__ bool found_needs_pool(const FoundCandidate& c)
{
    return c.category != PISCES_CAT_DELETION && c.length > kFoundInline;
}

__global__ __launch_bounds__(256) void find_count_kernel(DevReadBatch b, const uint8_t* __restrict__ del_dirs, const uint8_t* __restrict__ ref,
                                                         int64_t ref_len, FinderParams P, int32_t* __restrict__ n_found,
                                                         int32_t* __restrict__ n_pool)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= b.n_reads) return;
    const ReadView v = dev_read_view(b, del_dirs, r);
    int n = 0, bytes = 0;
    auto count = [&](const FoundCandidate& c) {
        if (c.position <= 0) return;
        n++;
        if (found_needs_pool(c)) bytes += c.length;
    };
    walk::walk_read<WordBases>(v, ref, ref_len, P, count);
    n_found[r] = n;
    n_pool[r] = bytes;
}

// in-place exclusive scans of two int32 arrays by one workgroup; totals[0], totals[1] = the sums.  Every wave owns a contiguous sixteenth
// of the arrays: it adds its part up, the sixteen sums are exchanged once through LDS, and the wave then scans its part 64 entries at a
// time with lane shuffles — one barrier in all (the first form scanned 1024 entries a round through LDS, twenty barriers a round:
// 228 us for 80 000 reads; this one 77 us: twelve dependent lane shuffles a round, 79 rounds a wave).
__global__ __launch_bounds__(1024) void found_scan_kernel(int32_t* __restrict__ a, int32_t* __restrict__ b, int32_t n, long long* __restrict__ totals)
{
    __shared__ long long s_tot[2][16];
    const int lane = (int)threadIdx.x & 63, w = (int)threadIdx.x >> 6;
    const int per = (((n + 15) / 16) + 63) & ~63;
    const int lo = min(w * per, n), hi = min(lo + per, n);
    long long sa = 0, sb = 0;
    for (int i = lo + lane; i < hi; i += 64) { sa += a[i]; sb += b[i]; }
    for (int d = 32; d; d >>= 1) { sa += __shfl_xor(sa, d); sb += __shfl_xor(sb, d); }
    if (lane == 0) { s_tot[0][w] = sa; s_tot[1][w] = sb; }
    __syncthreads();
    long long base_a = 0, base_b = 0;
    for (int k = 0; k < w; k++) { base_a += s_tot[0][k]; base_b += s_tot[1][k]; }
    int na = lo + lane < hi ? a[lo + lane] : 0, nb = lo + lane < hi ? b[lo + lane] : 0;
    for (int start = lo; start < hi; start += 64) {
        const int i = start + lane;
        const int va = na, vb = nb;
        // (the next 64 entries are requested before this round's are stored: a load behind a store to the same array is not moved up by
        // the compiler, and every round would be a full memory round trip)
        na = i + 64 < hi ? a[i + 64] : 0;
        nb = i + 64 < hi ? b[i + 64] : 0;
        int xa = va, xb = vb;
        for (int d = 1; d < 64; d <<= 1) {
            const int ya = __shfl_up(xa, d), yb = __shfl_up(xb, d);
            if (lane >= d) { xa += ya; xb += yb; }
        }
        if (i < hi) { a[i] = (int32_t)(base_a + xa - va); b[i] = (int32_t)(base_b + xb - vb); }
        base_a += __shfl(xa, 63);
        base_b += __shfl(xb, 63);
    }
    if (threadIdx.x == 0) {
        long long ta = 0, tb = 0;
        for (int k = 0; k < 16; k++) { ta += s_tot[0][k]; tb += s_tot[1][k]; }
        totals[0] = ta; totals[1] = tb;
    }
}

// slot_first[r] = first record slot of read r, slot_end = slot_first[r + 1] (host-made or scanned); pool_first[r] = first pool byte of
// read r when the pool offsets were scanned (MNV calling on), else nullptr: long insertions then take pool bytes from a cursor
// (rare: an insertion longer than kFoundInline bases).
__global__ __launch_bounds__(256) void find_emit_kernel(DevReadBatch b, const uint8_t* __restrict__ del_dirs, const uint8_t* __restrict__ ref,
                                                        int64_t ref_len, FinderParams P, const int32_t* __restrict__ slot_first,
                                                        const int32_t* __restrict__ pool_first, DevFound* __restrict__ out,
                                                        uint8_t* __restrict__ pool, unsigned int* __restrict__ pool_cursor,
                                                        int32_t pool_capacity, int32_t* __restrict__ overflow)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= b.n_reads) return;
    const ReadView v = dev_read_view(b, del_dirs, r);
    int slot = slot_first[r];
    const int slot_end = slot_first[r + 1];
    int pool_at = pool_first ? pool_first[r] : -1;
    auto write = [&](const FoundCandidate& c) {
        if (c.position <= 0) return;   // (pisces_hip_add_reads ignores candidates before the first base of the chromosome)
        if (slot >= slot_end) { atomicExch(overflow, 1); return; }
        DevFound f;
        f.c = c;
        f.read = r;
        f.pool_offset = -1;
        const int n_alt = c.category == PISCES_CAT_DELETION ? 0 : c.length;
        const uint8_t* src = v.bases + c.start_in_read;
        if (n_alt > kFoundInline) {
            int at;
            if (pool_first) { at = pool_at; pool_at += n_alt; }
            else at = (int)atomicAdd(pool_cursor, (unsigned int)n_alt);
            if (at + n_alt <= pool_capacity) {
                for (int k = 0; k < n_alt; k++) pool[at + k] = src[k];
                f.pool_offset = at;
            } else {
                atomicExch(overflow, 1);
            }
        }
#pragma unroll
        for (int k = 0; k < kFoundInline; k++) f.alt[k] = (k < n_alt && n_alt <= kFoundInline) ? src[k] : (uint8_t)0;
        out[slot++] = f;
    };
    walk::walk_read<WordBases>(v, ref, ref_len, P, write);
    for (; slot < slot_end; slot++) {   // reserved, unused: a hole
        DevFound f = {};
        f.c.category = kFoundHole;
        f.read = r;
        f.pool_offset = -1;
        out[slot] = f;
    }
}

}  // namespace pisces
