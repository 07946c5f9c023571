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

__device__ __forceinline__ uint32_t load_word(const uint8_t* p)   // four bytes at any address (unaligned global access is on)
{
    uint32_t v;
    __builtin_memcpy(&v, p, 4);
    return v;
}

// ---- the lane form, events first (round 4) -----------------------------------------------------------------------------------------
// walk_match_op hands every base of an M operation to the state machine; with a lane a read a wave then pays, at every base, for whatever
// any of its 64 reads does there — and with one base in a hundred a mismatch, some read closes a variant (Create + Annotate, the support
// direction: a hundred instructions) at every second base.  But the state machine only DOES something at an event — a base that cannot be
// called (N, low quality, N in the reference) or a callable mismatch — and between two events it is a closed form of the number of matches
// (a pending variant takes at most MaxGapBetweenMNV of them, then closes).  So here a lane first CLASSIFIES 64 bases of its operation at
// once, branch-free: twenty-four 8-byte loads issued together (bases, qualities, reference), four bases a step classified in SWAR form
// (v_perm letter table: is it exactly A C G T; one subtraction for quality < minBQ; a byte-wise xor for the mismatch) into two 64-bit
// masks.  Then it hops from event to event in a short loop that only advances (run, tail, open ends) and stops at a variant to emit;
// the long code (walk::finish_candidate and the record) runs once per EMITTED variant of the slowest lane, not once per base at which
// some lane emits.  Same candidates, same order, same records as walk::walk_match_op (tests/test_gpu_parity.py, tests/test_read_store.py
// against the host form and the oracle).  min_bq <= 127 (a quality byte >= 128 is never low then); the byte-wise form serves the rest.
__device__ __forceinline__ uint32_t swar_nonzero(uint32_t x) { return (((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x) & 0x80808080u; }   // bit 7 of every byte that is not 0
__device__ __forceinline__ uint32_t swar_not_acgt(uint32_t w)
{
    // the letter the low three bits of a base stand for (A 1, C 3, T 4, G 7; anything else: a byte no base with those bits equals)
    return swar_nonzero(w ^ __builtin_amdgcn_perm(0x47000054u, 0x43004101u, w & 0x07070707u));
}
__device__ __forceinline__ uint32_t swar_gather(uint32_t flags) { return (((flags >> 7) * 0x01020408u) >> 24) & 0xFu; }   // bit 7 of byte k -> bit k
__device__ __forceinline__ unsigned long long load_u64_at(const uint8_t* p)   // eight bytes at any address (unaligned global access is on)
{
    unsigned long long v;
    __builtin_memcpy(&v, p, 8);
    return v;
}

template <typename Emit>
__device__ __forceinline__ void walk_match_op_events(const ReadView& r, const walk::ReadFrame& f, const uint8_t* __restrict__ ref, int64_t ref_len,
                                                     const FinderParams& P, int op_read0, int op_len, int op_ref0, Emit& emit)
{
    int run = 0, tail = 0;
    bool open_left = false;
    // a variant that is closed waits here for the long code
    bool pend = false, p_ol = false, p_or = false;
    int p_at = 0, p_run = 0, p_len = 0;
    auto close = [&](int at, bool open_right) {   // walk::walk_match_op's close
        int len = run;
        if (tail >= 1) { len -= tail; open_right = false; }
        if (len < 1) return;
        pend = true; p_at = at; p_run = run; p_len = len; p_ol = open_left; p_or = open_right;
    };
    auto emit_pending = [&]() {
        const int start_read = op_read0 + p_at - p_run, start_ref = op_ref0 + p_at - p_run;
        walk::finish_candidate(r, f, P, p_len > 1 ? PISCES_CAT_MNV : PISCES_CAT_SNV, start_ref + 1, start_ref, start_read, p_len, p_len, -1, p_ol, p_or, emit);
        pend = false;
    };
    int64_t lim = op_len;
    if ((int64_t)r.read_len - op_read0 < lim) lim = (int64_t)r.read_len - op_read0;
    if (ref_len - (int64_t)op_ref0 < lim) lim = ref_len - (int64_t)op_ref0;
    const int walked = lim < 0 ? 0 : (int)lim;
    int cur = 0;   // bases [0, cur) of the operation have been through the state machine
    auto matches_until = [&](int upto) {   // the callable, matching bases [cur, upto) (walk_match_op_wave's, above)
        int g = upto - cur;
        if (g <= 0) return;
        if (run > 0) {
            if (P.call_mnvs) {
                const int take = max(0, min(g, min(P.max_mnv_length - run, P.max_gap - tail)));
                run += take; tail += take; g -= take;
            }
            if (g > 0) { close(upto - g, false); run = 0; tail = 0; }
        }
        if (g > 0) open_left = false;
        cur = upto;
    };
    auto event = [&](int i, bool callable) {
        matches_until(i);
        const bool alone_on_last_base = i == op_len - 1 && run == 0;
        const bool grows = callable && P.call_mnvs && run + 1 <= P.max_mnv_length && tail <= P.max_gap && !alone_on_last_base;
        if (grows) { run++; tail = 0; }
        else {
            close(i, !callable);   // (nothing pending from matches_until then: it left run == 0, or it closed nothing)
            run = callable ? 1 : 0;
            tail = 0;
            open_left = !callable;
        }
        cur = i + 1;
    };
    const uint8_t* const pb = r.bases + op_read0;
    const uint8_t* const pq = r.quals + op_read0;
    const uint8_t* const pf = ref + op_ref0;
    const uint32_t qk4 = (0x7Fu + (uint32_t)min(max(P.min_bq, 0), 127)) * 0x01010101u;
    for (int c0 = 0; c0 < walked; c0 += 64) {
        const int n_here = min(64, walked - c0);
        unsigned long long bw[8], qw[8], fw[8];
        if (walked >= 8) {
#pragma unroll
            for (int w = 0; w < 8; w++) {   // (a word that would reach past the operation's bases is taken early and shifted; one wholly past it: anything)
                const int o = c0 + 8 * w, oc = min(o, walked - 8), sh = 8 * min(o - oc, 7);
                bw[w] = load_u64_at(pb + oc) >> sh;
                qw[w] = load_u64_at(pq + oc) >> sh;
                fw[w] = load_u64_at(pf + oc) >> sh;
            }
        } else {
#pragma unroll
            for (int w = 0; w < 8; w++) bw[w] = qw[w] = fw[w] = 0;
            for (int k = 0; k < walked; k++) {
                bw[0] |= (unsigned long long)pb[k] << (8 * k); qw[0] |= (unsigned long long)pq[k] << (8 * k); fw[0] |= (unsigned long long)pf[k] << (8 * k);
            }
        }
        unsigned long long unc = 0, mis = 0;   // bit k: base c0 + k cannot be called / can, and differs from the reference
#pragma unroll
        for (int w = 0; w < 8; w++) {
#pragma unroll
            for (int hf = 0; hf < 2; hf++) {
                const uint32_t b4 = (uint32_t)(bw[w] >> (32 * hf)), q4 = (uint32_t)(qw[w] >> (32 * hf)), f4 = (uint32_t)(fw[w] >> (32 * hf));
                const uint32_t low4 = ~q4 & (qk4 - (q4 & 0x7F7F7F7Fu)) & 0x80808080u;   // quality < minBQ
                const uint32_t u4 = swar_not_acgt(b4) | swar_not_acgt(f4) | low4;
                const uint32_t m4 = swar_nonzero(b4 ^ f4) & ~u4;
                unc |= (unsigned long long)swar_gather(u4) << (8 * w + 4 * hf);
                mis |= (unsigned long long)swar_gather(m4) << (8 * w + 4 * hf);
            }
        }
        const unsigned long long here = n_here >= 64 ? ~0ull : ((1ull << n_here) - 1ull);
        unc &= here;
        unsigned long long ev = (unc | mis) & here;
        for (;;) {
            while (ev && !pend) {
                const int k = __builtin_ctzll(ev);
                ev &= ev - 1;
                event(c0 + k, !((unc >> k) & 1ull));
            }
            if (!pend) break;
            emit_pending();
        }
    }
    matches_until(walked);
    close(walked, false);
    if (pend) emit_pending();
}

// ProcessCigarOps (finder_walk.h walk_read) with the M operations in the event form
template <typename Emit>
__device__ __forceinline__ void walk_read_events(const ReadView& r, const uint8_t* __restrict__ ref, int64_t ref_len, const FinderParams& P, Emit& emit)
{
    const walk::ReadFrame f = walk::frame_of(r);
    int in_read = 0, in_ref = r.position - 1;
    for (int ci = 0; ci < r.n_cigar; ci++) {
        const uint8_t t = r.cigar_op[ci];
        const int len = (int)r.cigar_len[ci];
        if (t == 'M') {
            if (P.snvs_and_mnvs) walk_match_op_events(r, f, ref, ref_len, P, in_read, len, in_ref, emit);
        } else if (t == 'I') {
            const bool off_contig = (int64_t)in_ref - 1 >= ref_len || in_ref == 0;
            if (!off_contig && in_read < r.read_len && in_read + len <= r.read_len && r.quals[in_read] >= P.min_bq)
                walk::finish_candidate(r, f, P, PISCES_CAT_INSERTION, in_ref, in_ref - 1, in_read, len, len + 1, -1, false, false, emit);
        } else if (t == 'D') {
            bool flanks_ok = false;
            if (r.read_len > 0) {
                const int after = in_read < r.read_len ? r.quals[in_read] : r.quals[in_read - 1];
                const int before = in_read > 0 ? r.quals[in_read - 1] : after;
                flanks_ok = before >= P.min_bq && after >= P.min_bq;
            }
            if ((int64_t)in_ref + len < ref_len && in_ref >= 1 && flanks_ok)
                walk::finish_candidate(r, f, P, PISCES_CAT_DELETION, in_ref, in_ref - 1, in_read, len, 1, ci, false, false, emit);
        } else if ((t == 'X' || t == '=') && P.mark_x_spans && len > 0) {
            walk::mark_unwalked_span(r, t, len, in_read, in_ref, ref, ref_len, P, emit);
        }
        if (walk::spans_read(t)) in_read += len;
        if (walk::spans_ref(t)) in_ref += len;
    }
}

__device__ __forceinline__ bool found_needs_pool(const FoundCandidate& c)
{
    return c.category != PISCES_CAT_DELETION && c.category != kFoundSpanMark && c.length > kFoundInline;
}

template <bool kEvents>   // the event form of the M-operation walk (above), or finder_walk.h's base by base
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
    if (kEvents) walk_read_events(v, ref, ref_len, P, count);
    else walk::walk_read<WordBases>(v, ref, ref_len, P, count);
    n_found[r] = n;
    n_pool[r] = bytes;
}

// in-place exclusive scans of two int32 arrays by one workgroup; totals[0], totals[1] = the sums.  Every wave owns a contiguous sixteenth
// of the arrays: it adds its part up, the sixteen sums are exchanged once through LDS (the only barrier), and the wave then scans its
// part 512 entries a round: a lane takes eight neighbours (scanned in registers), the lanes' sums go through one shuffle scan.
// (The first form scanned 1024 entries a round through LDS, twenty barriers a round: 228 us for 80 000 reads; one entry a lane and
// round, twelve dependent shuffles for 64 entries: 77 us.)
__global__ __launch_bounds__(1024) void found_scan_kernel(int32_t* __restrict__ a, int32_t* __restrict__ b, int32_t n, long long* __restrict__ totals)
{
    __shared__ long long s_tot[2][16];
    constexpr int kPer = 8;
    const int lane = (int)threadIdx.x & 63, w = (int)threadIdx.x >> 6;
    const int per = (((n + 15) / 16) + 64 * kPer - 1) / (64 * kPer) * (64 * kPer);
    const int lo = min(w * per, n), hi = min(lo + per, n);
    long long sa = 0, sb = 0;
    for (int i = lo + lane; i < hi; i += 64) { sa += a[i]; sb += b[i]; }
    for (int d = 32; d; d >>= 1) { sa += __shfl_xor(sa, d); sb += __shfl_xor(sb, d); }
    if (lane == 0) { s_tot[0][w] = sa; s_tot[1][w] = sb; }
    __syncthreads();
    long long base_a = 0, base_b = 0;
    for (int k = 0; k < w; k++) { base_a += s_tot[0][k]; base_b += s_tot[1][k]; }
    for (int start = lo; start < hi; start += 64 * kPer) {
        const int i0 = start + lane * kPer;
        int va[kPer], vb[kPer];
        int ta = 0, tb = 0;   // the lane's eight entries: their values, then exclusive sums inside the lane
        const bool whole = start + 64 * kPer <= hi;   // (wave-uniform; the arrays and `start` are 16-byte aligned: two dwordx4 loads a lane)
        if (whole) {
            const int4 a0 = *reinterpret_cast<const int4*>(a + i0), a1 = *reinterpret_cast<const int4*>(a + i0 + 4);
            const int4 b0 = *reinterpret_cast<const int4*>(b + i0), b1 = *reinterpret_cast<const int4*>(b + i0 + 4);
            va[0] = a0.x; va[1] = a0.y; va[2] = a0.z; va[3] = a0.w; va[4] = a1.x; va[5] = a1.y; va[6] = a1.z; va[7] = a1.w;
            vb[0] = b0.x; vb[1] = b0.y; vb[2] = b0.z; vb[3] = b0.w; vb[4] = b1.x; vb[5] = b1.y; vb[6] = b1.z; vb[7] = b1.w;
        } else {
#pragma unroll
            for (int k = 0; k < kPer; k++) {
                va[k] = i0 + k < hi ? a[i0 + k] : 0;
                vb[k] = i0 + k < hi ? b[i0 + k] : 0;
            }
        }
#pragma unroll
        for (int k = 0; k < kPer; k++) {
            const int xa = va[k], xb = vb[k];
            va[k] = ta; vb[k] = tb;
            ta += xa; tb += xb;
        }
        int xa = ta, xb = tb;   // inclusive scan of the lanes' sums
        for (int d = 1; d < 64; d <<= 1) {
            const int ya = __shfl_up(xa, d), yb = __shfl_up(xb, d);
            if (lane >= d) { xa += ya; xb += yb; }
        }
        const long long la = base_a + xa - ta, lb = base_b + xb - tb;
        if (whole) {
            *reinterpret_cast<int4*>(a + i0) = make_int4((int)(la + va[0]), (int)(la + va[1]), (int)(la + va[2]), (int)(la + va[3]));
            *reinterpret_cast<int4*>(a + i0 + 4) = make_int4((int)(la + va[4]), (int)(la + va[5]), (int)(la + va[6]), (int)(la + va[7]));
            *reinterpret_cast<int4*>(b + i0) = make_int4((int)(lb + vb[0]), (int)(lb + vb[1]), (int)(lb + vb[2]), (int)(lb + vb[3]));
            *reinterpret_cast<int4*>(b + i0 + 4) = make_int4((int)(lb + vb[4]), (int)(lb + vb[5]), (int)(lb + vb[6]), (int)(lb + vb[7]));
        } else {
#pragma unroll
            for (int k = 0; k < kPer; k++)
                if (i0 + k < hi) { a[i0 + k] = (int32_t)(la + va[k]); b[i0 + k] = (int32_t)(lb + vb[k]); }
        }
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
template <bool kEvents>
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
        const int n_alt = (c.category == PISCES_CAT_DELETION || c.category == kFoundSpanMark) ? 0 : c.length;
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
    if (kEvents) walk_read_events(v, ref, ref_len, P, write);
    else walk::walk_read<WordBases>(v, ref, ref_len, P, write);
    for (; slot < slot_end; slot++) {   // reserved, unused: a hole
        DevFound f = {};
        f.c.category = kFoundHole;
        f.read = r;
        f.pool_offset = -1;
        out[slot] = f;
    }
}

// ---- the wave form of the walk (round 4) ------------------------------------------------------------------------------------------
// One lane per read runs the M-operation state machine base by base: ~1.2 waves a SIMD, and a wave pays for every branch any of its 64
// reads takes (106 + 119 us per 80 000 reads of BASELINE config 3's mix, whatever the memory system does).  But the state machine only
// ever DOES something at an EVENT — a base that cannot be called (N, low quality, N in the reference) or a mismatch: ~3 of a read's 150
// bases — and what it does between two events is a closed form of the number of matching bases between them (a pending variant grows by
// at most MaxGapBetweenMNV matches, then closes).  So here a WAVE takes a read: lane l compares the four bases 4 l .. 4 l + 3 of an M
// operation with the reference in one step (three 4-byte loads a lane for up to 256 bases), a ballot says which lanes hold an event,
// and the state machine — wave-uniform, i.e. scalar code — hops from event to event.  Same candidates, same order, same records as
// finder_walk.h's walk (the host form, and the oracle's restatement, check it: tests/test_gpu_parity.py).
//   find_count_wave_kernel / find_emit_wave_kernel   kReadsPerWave consecutive reads a wave
struct ChunkWords { uint32_t b, q, f; };   // a lane's four bases, qualities and reference bases of a chunk (already shifted into place)
// the lane's words of the first chunk of an operation of `walked` >= 4 bases (unconditional loads: a word that would reach past the
// operation's bases is taken 1-3 bytes early and shifted; a lane past the operation's end loads its last word)
__device__ __forceinline__ ChunkWords load_first_chunk(const uint8_t* pb, const uint8_t* pq, const uint8_t* pf, int walked, int lane)
{
    const int i0 = min(4 * lane, walked - 1);
    const int oc = min(i0, walked - 4), sh = 8 * (i0 - oc);
    ChunkWords w;
    w.b = load_word(pb + oc) >> sh; w.q = load_word(pq + oc) >> sh; w.f = load_word(pf + oc) >> sh;
    return w;
}

template <typename Emit>
__device__ __forceinline__ void walk_match_op_wave(const ReadView& r, const walk::ReadFrame& f, const uint8_t* __restrict__ ref, int64_t ref_len,
                                                   const FinderParams& P, int op_read0, int op_len, int op_ref0, int lane, Emit& emit,
                                                   const ChunkWords* first_chunk = nullptr /* load_first_chunk's words, requested ahead */)
{
    int run = 0, tail = 0;
    bool open_left = false;
    auto close = [&](int at, bool open_right) {   // finder_walk.h walk_match_op's close
        int len = run;
        if (tail >= 1) { len -= tail; open_right = false; }
        if (len < 1) return;
        const int start_read = op_read0 + at - run, start_ref = op_ref0 + at - run;
        walk::finish_candidate(r, f, P, len > 1 ? PISCES_CAT_MNV : PISCES_CAT_SNV, start_ref + 1, start_ref, start_read, len, len, -1, open_left, open_right, emit);
    };
    int64_t lim = op_len;
    if ((int64_t)r.read_len - op_read0 < lim) lim = (int64_t)r.read_len - op_read0;
    if (ref_len - (int64_t)op_ref0 < lim) lim = ref_len - (int64_t)op_ref0;
    const int walked = lim < 0 ? 0 : (int)lim;
    int cur = 0;   // bases [0, cur) of the operation have been through the state machine
    // the callable, matching bases [cur, upto): a pending variant takes them as trailing matches while ShouldBuildUpMNV lets it
    // (run + 1 <= MaxSizeMNV, tail + 1 <= MaxGapBetweenMNV), the first one it cannot take closes it, and a match with nothing pending
    // only clears open_left
    auto matches_until = [&](int upto) {
        int g = upto - cur;
        if (g <= 0) return;
        if (run > 0) {
            while (g > 0 && P.call_mnvs && run + 1 <= P.max_mnv_length && tail + 1 <= P.max_gap) { run++; tail++; g--; }
            if (g > 0) { close(upto - g, false); run = 0; tail = 0; }
        }
        if (g > 0) open_left = false;
        cur = upto;
    };
    auto event = [&](int i, bool callable) {   // a base that cannot be called, or a callable mismatch (finder_walk.h's step on such a base)
        matches_until(i);
        const bool alone_on_last_base = i == op_len - 1 && run == 0;
        const bool grows = callable && P.call_mnvs && run + 1 <= P.max_mnv_length && tail <= P.max_gap && !alone_on_last_base;
        if (grows) { run++; tail = 0; }
        else {
            close(i, !callable);
            run = callable ? 1 : 0;
            tail = 0;
            open_left = !callable;
        }
        cur = i + 1;
    };
    const uint8_t* const pb = r.bases + op_read0;
    const uint8_t* const pq = r.quals + op_read0;
    const uint8_t* const pf = ref + op_ref0;
    for (int c0 = 0; c0 < walked; c0 += 256) {
        const int i0 = c0 + 4 * lane;
        uint32_t packed = 0;   // bit k: base i0 + k cannot be called; bit 4 + k: it can, and differs from the reference
        if (i0 < walked) {
            uint32_t bw, qw, fw;
            if (c0 == 0 && first_chunk) {
                bw = first_chunk->b; qw = first_chunk->q; fw = first_chunk->f;
            } else if (walked >= 4) {   // a word that would reach past the operation's bases is taken 1-3 bytes early and shifted
                const int oc = min(i0, walked - 4), sh = 8 * (i0 - oc);
                bw = load_word(pb + oc) >> sh; qw = load_word(pq + oc) >> sh; fw = load_word(pf + oc) >> sh;
            } else {
                bw = qw = fw = 0;
                for (int k = 0; k < walked - i0; k++) { bw |= (uint32_t)pb[i0 + k] << (8 * k); qw |= (uint32_t)pq[i0 + k] << (8 * k); fw |= (uint32_t)pf[i0 + k] << (8 * k); }
            }
            const int n_here = min(4, walked - i0);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint8_t rb = (uint8_t)(bw >> (8 * k)), fb = (uint8_t)(fw >> (8 * k)), q = (uint8_t)(qw >> (8 * k));
                const bool callable = walk::is_acgt(rb) && walk::is_acgt(fb) && q >= P.min_bq;
                if (k < n_here) packed |= !callable ? (1u << k) : (rb != fb ? (16u << k) : 0u);
            }
        }
        unsigned long long ev = __ballot(packed != 0);
        while (ev) {
            const int L = __builtin_ctzll(ev);
            ev &= ev - 1;
            const uint32_t pk = (uint32_t)__builtin_amdgcn_readlane((int)packed, L);
#pragma unroll 1
            for (int k = 0; k < 4; k++)
                if ((pk >> k) & 0x11u) event(c0 + 4 * L + k, !((pk >> k) & 1u));
        }
    }
    matches_until(walked);
    close(walked, false);
}

// ProcessCigarOps (finder_walk.h walk_read) with the M operations walked by the wave
template <typename Emit>
__device__ __forceinline__ void walk_read_wave(const ReadView& r, const uint8_t* __restrict__ ref, int64_t ref_len, const FinderParams& P, int lane, Emit& emit)
{
    const walk::ReadFrame f = walk::frame_of(r);
    int in_read = 0, in_ref = r.position - 1;
    for (int ci = 0; ci < r.n_cigar; ci++) {
        const uint8_t t = r.cigar_op[ci];
        const int len = (int)r.cigar_len[ci];
        if (t == 'M') {
            if (P.snvs_and_mnvs) walk_match_op_wave(r, f, ref, ref_len, P, in_read, len, in_ref, lane, emit);
        } else if (t == 'I') {
            const bool off_contig = (int64_t)in_ref - 1 >= ref_len || in_ref == 0;
            if (!off_contig && in_read < r.read_len && in_read + len <= r.read_len && r.quals[in_read] >= P.min_bq)
                walk::finish_candidate(r, f, P, PISCES_CAT_INSERTION, in_ref, in_ref - 1, in_read, len, len + 1, -1, false, false, emit);
        } else if (t == 'D') {
            bool flanks_ok = false;
            if (r.read_len > 0) {
                const int after = in_read < r.read_len ? r.quals[in_read] : r.quals[in_read - 1];
                const int before = in_read > 0 ? r.quals[in_read - 1] : after;
                flanks_ok = before >= P.min_bq && after >= P.min_bq;
            }
            if ((int64_t)in_ref + len < ref_len && in_ref >= 1 && flanks_ok)
                walk::finish_candidate(r, f, P, PISCES_CAT_DELETION, in_ref, in_ref - 1, in_read, len, 1, ci, false, false, emit);
        } else if ((t == 'X' || t == '=') && P.mark_x_spans && len > 0) {
            walk::mark_unwalked_span(r, t, len, in_read, in_ref, ref, ref_len, P, emit);
        }
        if (walk::spans_read(t)) in_read += len;
        if (walk::spans_ref(t)) in_ref += len;
    }
}

constexpr int kReadsPerWave = 8;

__global__ __launch_bounds__(256) void find_count_wave_kernel(DevReadBatch b, const uint8_t* __restrict__ del_dirs, const uint8_t* __restrict__ ref,
                                                              int64_t ref_len, FinderParams P, int32_t* __restrict__ n_found, int32_t* __restrict__ n_pool)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)));
    for (int k = 0; k < kReadsPerWave; k++) {
        const int r = wave * kReadsPerWave + k;
        if (r >= b.n_reads) return;
        const ReadView v = dev_read_view(b, del_dirs, r);
        int n = 0, bytes = 0;
        auto count = [&](const FoundCandidate& c) {
            if (c.position <= 0) return;
            n++;
            if (found_needs_pool(c)) bytes += c.length;
        };
        walk_read_wave(v, ref, ref_len, P, lane, count);
        if (lane == 0) { n_found[r] = n; n_pool[r] = bytes; }
    }
}

__global__ __launch_bounds__(256) void find_emit_wave_kernel(DevReadBatch b, const uint8_t* __restrict__ del_dirs, const uint8_t* __restrict__ ref,
                                                             int64_t ref_len, FinderParams P, const int32_t* __restrict__ slot_first,
                                                             const int32_t* __restrict__ pool_first, DevFound* __restrict__ out, uint8_t* __restrict__ pool,
                                                             unsigned int* __restrict__ pool_cursor, int32_t pool_capacity, int32_t* __restrict__ overflow)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)));
    for (int k = 0; k < kReadsPerWave; k++) {
        const int r = wave * kReadsPerWave + k;
        if (r >= b.n_reads) return;
        const ReadView v = dev_read_view(b, del_dirs, r);
        int slot = slot_first[r];
        const int slot_end = slot_first[r + 1];
        int pool_at = pool_first ? pool_first[r] : -1;
        auto write = [&](const FoundCandidate& c) {   // (wave-uniform: every lane holds the same candidate; the lanes share the stores)
            if (c.position <= 0) return;
            if (slot >= slot_end) { if (lane == 0) atomicExch(overflow, 1); return; }
            const int n_alt = (c.category == PISCES_CAT_DELETION || c.category == kFoundSpanMark) ? 0 : c.length;
            const uint8_t* src = v.bases + c.start_in_read;
            int pool_offset = -1;
            if (n_alt > kFoundInline) {
                int at;
                if (pool_first) { at = pool_at; pool_at += n_alt; }
                else {
                    unsigned int got = 0;
                    if (lane == 0) got = atomicAdd(pool_cursor, (unsigned int)n_alt);
                    at = __builtin_amdgcn_readfirstlane((int)got);
                }
                if (at + n_alt <= pool_capacity) {
                    for (int q = lane; q < n_alt; q += 64) pool[at + q] = src[q];
                    pool_offset = at;
                } else if (lane == 0) {
                    atomicExch(overflow, 1);
                }
            }
            // the record's 64 bytes: FoundCandidate (24), read, pool_offset, alt[32]: lane 0 the head, lanes 0-31 one ALT byte each
            DevFound* const dst = out + slot;
            if (lane == 0) {
                dst->c = c;
                dst->read = r;
                dst->pool_offset = pool_offset;
            }
            if (lane < kFoundInline) dst->alt[lane] = (lane < n_alt && n_alt <= kFoundInline) ? src[lane] : (uint8_t)0;
            slot++;
        };
        walk_read_wave(v, ref, ref_len, P, lane, write);
        for (; slot < slot_end; slot++) {   // reserved, unused: a hole
            DevFound* const dst = out + slot;
            if (lane == 0) {
                FoundCandidate hole = {};
                hole.category = kFoundHole;
                dst->c = hole;
                dst->read = r;
                dst->pool_offset = -1;
            }
            if (lane < kFoundInline) dst->alt[lane] = 0;
        }
    }
}

// The form that is used: a wave takes SIXTY-FOUR reads.  First every lane reads the descriptors of one of them (position, CIGAR, offsets:
// the dependent scalar loads that made a read of the kernels above a chain of three memory round trips, here one pass for 64 reads);
// then the reads are taken in order, their values handed to all lanes with v_readlane.  A read of one M operation — nearly all — has the
// words of its bases requested while the read before it is walked (load_first_chunk: the round trip lies under the event loop of its
// predecessor); anything else goes through walk_read_wave as it is.
template <bool kEmit>
__global__ __launch_bounds__(256) void find_batch_wave_kernel(DevReadBatch b, const uint8_t* __restrict__ del_dirs, const uint8_t* __restrict__ ref,
                                                              int64_t ref_len, FinderParams P, int32_t* __restrict__ n_found, int32_t* __restrict__ n_pool,
                                                              const int32_t* __restrict__ slot_first, const int32_t* __restrict__ pool_first,
                                                              DevFound* __restrict__ out, uint8_t* __restrict__ pool, unsigned int* __restrict__ pool_cursor,
                                                              int32_t pool_capacity, int32_t* __restrict__ overflow)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)));
    const int r_mine = wave * 64 + lane;
    const bool valid = r_mine < b.n_reads;
    // ---- the lanes' own reads
    int pos = 0, c0 = 0, nc = 0, s0 = 0, n = 0, len0 = 0, rev = 0, slot0 = 0, slot1 = 0, pool0 = -1;
    uint8_t op0 = 0;
    if (valid) {
        pos = b.position[r_mine];
        c0 = b.cigar_offset[r_mine]; nc = b.cigar_offset[r_mine + 1] - c0;
        s0 = b.seq_offset[r_mine]; n = b.seq_offset[r_mine + 1] - s0;
        rev = b.flags[r_mine] & 1;
        if (nc > 0) { op0 = b.cigar_op[c0]; len0 = (int)b.cigar_len[c0]; }
        if (kEmit) { slot0 = slot_first[r_mine]; slot1 = slot_first[r_mine + 1]; pool0 = pool_first ? pool_first[r_mine] : -1; }
    }
    // the bases of the one M operation that lie on the read and on the contig (walk_match_op's `walked`)
    long long lim = len0;
    if ((long long)n < lim) lim = n;
    if (ref_len - (long long)(pos - 1) < lim) lim = ref_len - (long long)(pos - 1);
    const int walked = lim < 0 ? 0 : (int)lim;
    const bool simple = valid && P.snvs_and_mnvs && nc == 1 && op0 == 'M' && walked >= 4 && pos >= 1;
    const unsigned long long valid_mask = __ballot(valid), simple_mask = __ballot(simple);
    int my_n = 0, my_bytes = 0;
    auto issue = [&](int k) {   // the first chunk of (simple) read k
        const int kp = __builtin_amdgcn_readlane(pos, k), ks = __builtin_amdgcn_readlane(s0, k), kw = __builtin_amdgcn_readlane(walked, k);
        return load_first_chunk(b.bases + ks, b.quals + ks, ref + (kp - 1), kw, lane);
    };
    ChunkWords cur = {0, 0, 0};
    if (simple_mask) cur = issue(__builtin_ctzll(simple_mask));
    for (unsigned long long todo = valid_mask; todo; todo &= todo - 1) {
        const int k = __builtin_ctzll(todo);
        const int r = wave * 64 + k;
        const bool k_simple = (simple_mask >> k) & 1ull;
        // the next simple read's words: requested now, used when this read is done
        const unsigned long long later = simple_mask & ~((2ull << k) - 1ull);
        ChunkWords nxt = cur;
        if (k_simple && later) nxt = issue(__builtin_ctzll(later));
        __builtin_amdgcn_sched_barrier(0);
        int cnt = 0, bytes = 0;
        int slot = kEmit ? __builtin_amdgcn_readlane(slot0, k) : 0;
        const int slot_end = kEmit ? __builtin_amdgcn_readlane(slot1, k) : 0;
        int pool_at = kEmit ? __builtin_amdgcn_readlane(pool0, k) : -1;
        ReadView v;
        auto handle = [&](const FoundCandidate& c) {   // (wave-uniform)
            if (c.position <= 0) return;
            if (!kEmit) {
                cnt++;
                if (found_needs_pool(c)) bytes += c.length;
                return;
            }
            if (slot >= slot_end) { if (lane == 0) atomicExch(overflow, 1); return; }
            const int n_alt = (c.category == PISCES_CAT_DELETION || c.category == kFoundSpanMark) ? 0 : c.length;
            const uint8_t* src = v.bases + c.start_in_read;
            int pool_offset = -1;
            if (n_alt > kFoundInline) {
                int at;
                if (pool_first) { at = pool_at; pool_at += n_alt; }
                else {
                    unsigned int got = 0;
                    if (lane == 0) got = atomicAdd(pool_cursor, (unsigned int)n_alt);
                    at = __builtin_amdgcn_readfirstlane((int)got);
                }
                if (at + n_alt <= pool_capacity) {
                    for (int q = lane; q < n_alt; q += 64) pool[at + q] = src[q];
                    pool_offset = at;
                } else if (lane == 0) {
                    atomicExch(overflow, 1);
                }
            }
            DevFound* const dst = out + slot;
            if (lane == 0) {
                dst->c = c;
                dst->read = r;
                dst->pool_offset = pool_offset;
            }
            if (lane < kFoundInline) dst->alt[lane] = (lane < n_alt && n_alt <= kFoundInline) ? src[lane] : (uint8_t)0;
            slot++;
        };
        if (k_simple) {
            const int kp = __builtin_amdgcn_readlane(pos, k), ks = __builtin_amdgcn_readlane(s0, k), kn = __builtin_amdgcn_readlane(n, k);
            const int kc = __builtin_amdgcn_readlane(c0, k), kl = __builtin_amdgcn_readlane(len0, k);
            v.position = kp; v.n_cigar = 1; v.cigar_op = b.cigar_op + kc; v.cigar_len = b.cigar_len + kc; v.read_len = kn;
            v.bases = b.bases + ks; v.quals = b.quals + ks; v.dirs = b.dirs ? b.dirs + ks : nullptr;
            v.del_dirs = del_dirs ? del_dirs + 2 * (size_t)kc : nullptr;
            v.is_reverse = __builtin_amdgcn_readlane(rev, k);
            walk::ReadFrame f;   // frame_of for a read of one M operation
            f.end_position = kp + kl - 1;
            f.max_position = kl > 0 ? kp + kl - 1 : kp - 1;
            f.first_op = f.last_op = 'M';
            walk_match_op_wave(v, f, ref, ref_len, P, 0, kl, kp - 1, lane, handle, &cur);
        } else {
            v = dev_read_view(b, del_dirs, r);
            walk_read_wave(v, ref, ref_len, P, lane, handle);
        }
        if (!kEmit) {
            if (lane == k) { my_n = cnt; my_bytes = bytes; }
        } else {
            for (; slot < slot_end; slot++) {   // reserved, unused: a hole
                DevFound* const dst = out + slot;
                if (lane == 0) {
                    FoundCandidate hole = {};
                    hole.category = kFoundHole;
                    dst->c = hole;
                    dst->read = r;
                    dst->pool_offset = -1;
                }
                if (lane < kFoundInline) dst->alt[lane] = 0;
            }
        }
        if (k_simple) cur = nxt;
    }
    if (!kEmit && valid) { n_found[r_mine] = my_n; n_pool[r_mine] = my_bytes; }
}

// Exclusive scans of two int32 arrays of any length by many workgroups, in three launches (block sums, their scan, the blocks): the
// one-workgroup found_scan_kernel above takes 229 us for 400 000 reads — one workgroup's pace — where these take a few microseconds each.
constexpr int kScanBlock = 8192;   // entries a workgroup of 1024 threads takes: eight a thread
__global__ __launch_bounds__(1024) void scan_block_sums_kernel(const int32_t* __restrict__ a, const int32_t* __restrict__ b, int32_t n,
                                                               long long* __restrict__ sums /* [2][blocks] */, int32_t n_blocks)
{
    __shared__ long long s_part[2][16];
    const int base = blockIdx.x * kScanBlock, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    long long sa = 0, sb = 0;
    for (int i = base + (int)threadIdx.x; i < min(base + kScanBlock, n); i += 1024) { sa += a[i]; sb += b[i]; }
    for (int d = 32; d; d >>= 1) { sa += __shfl_xor(sa, d); sb += __shfl_xor(sb, d); }
    if (lane == 0) { s_part[0][w] = sa; s_part[1][w] = sb; }
    __syncthreads();
    if (threadIdx.x == 0) {
        long long ta = 0, tb = 0;
        for (int k = 0; k < 16; k++) { ta += s_part[0][k]; tb += s_part[1][k]; }
        sums[blockIdx.x] = ta;
        sums[n_blocks + blockIdx.x] = tb;
    }
}
__global__ __launch_bounds__(1024) void scan_block_offsets_kernel(long long* __restrict__ sums, int32_t n_blocks, long long* __restrict__ totals)
{
    // (one workgroup, two arrays of at most a few thousand block sums: thread t scans array t)
    if (threadIdx.x < 2) {
        long long* s = sums + (size_t)threadIdx.x * n_blocks;
        long long acc = 0;
        for (int i = 0; i < n_blocks; i++) { const long long v = s[i]; s[i] = acc; acc += v; }
        totals[threadIdx.x] = acc;
    }
}
__global__ __launch_bounds__(1024) void scan_apply_kernel(int32_t* __restrict__ a, int32_t* __restrict__ b, int32_t n, const long long* __restrict__ sums, int32_t n_blocks)
{
    __shared__ int s_wave[2][16];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int i0 = blockIdx.x * kScanBlock + (int)threadIdx.x * 8;
    int va[8], vb[8], ta = 0, tb = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const int xa = i0 + k < n ? a[i0 + k] : 0, xb = i0 + k < n ? b[i0 + k] : 0;
        va[k] = ta; vb[k] = tb;
        ta += xa; tb += xb;
    }
    int xa = ta, xb = tb;   // inclusive scan of the lanes' sums inside the wave
    for (int d = 1; d < 64; d <<= 1) {
        const int ya = __shfl_up(xa, d), yb = __shfl_up(xb, d);
        if (lane >= d) { xa += ya; xb += yb; }
    }
    if (lane == 63) { s_wave[0][w] = xa; s_wave[1][w] = xb; }
    __syncthreads();
    int wa = 0, wb = 0;
    for (int k = 0; k < w; k++) { wa += s_wave[0][k]; wb += s_wave[1][k]; }
    const long long la = sums[blockIdx.x] + wa + xa - ta, lb = sums[n_blocks + blockIdx.x] + wb + xb - tb;
#pragma unroll
    for (int k = 0; k < 8; k++)
        if (i0 + k < n) { a[i0 + k] = (int32_t)(la + va[k]); b[i0 + k] = (int32_t)(lb + vb[k]); }
}

// ---- RegionState.AddCandidate for the records of one batch, on the device ------------------------------------------------------
// find_emit_kernel leaves one record per read event; the reference merges equal candidates as they arrive (RegionState.cs:94-160: same
// position, type, alleles — and, when open ends are tracked, the same open ends — add their support and well-anchored support by
// direction; a position's candidates stay in order of first arrival).  Here the records of a batch are merged before they cross PCIe:
//   found_merge_kernel    lane = record.  Lanes of a wave that hold the same candidate are added up first (the reads that carry a
//                         variant are neighbours in read order, so most of a variant's ~VAF x depth records meet in a few waves); every
//                         distinct candidate of the wave then goes to an open-addressing table of record indices — the first to claim a
//                         slot owns the group, the others compare against the owner's record (the full key, no hash trust) and add
//                         their sums to the owner's accumulators, with the smallest record index of the group (= first arrival)
//   found_gather_kernel   the owners, each as the record of its group's FIRST arrival (whose open ends the merged candidate keeps
//                         when open ends are not part of the identity) with the group's sums, straight into pinned host memory
// The host puts the groups back into order of first arrival and merges them into the blocks (candidates of earlier batches included).
struct DevMerged {   // 96 bytes
    DevFound f;
    int32_t sup[3], anch[3];
    int32_t first;       // record index of the group's first arrival
    int32_t pad;
};
static_assert(sizeof(DevMerged) == 96, "DevMerged is 96 bytes");
constexpr int kMergeAcc = 8;   // int32 per record: sup[3], anch[3], 0x7FFFFFFF - first (atomicMax), owner flag

__device__ __forceinline__ int found_alt_len(const FoundCandidate& c) { return (c.category == PISCES_CAT_DELETION || c.category == kFoundSpanMark) ? 0 : c.length; }
__device__ __forceinline__ const uint8_t* found_alt_bytes(const DevFound& f, const uint8_t* pool)
{
    return found_alt_len(f.c) > kFoundInline ? pool + f.pool_offset : f.alt;
}
// CandidateAllele.Equals as RegionState.AddCandidate uses it: the REF allele follows from (position, category, length)
__device__ __forceinline__ bool found_same_key(const DevFound& a, const DevFound& b, const uint8_t* pool, int track_open)
{
    if (a.c.position != b.c.position || a.c.category != b.c.category || a.c.length != b.c.length) return false;
    if (track_open && (a.c.open_left != b.c.open_left || a.c.open_right != b.c.open_right)) return false;
    const int n = found_alt_len(a.c);
    if (n > kFoundInline && (a.pool_offset < 0 || b.pool_offset < 0)) return false;   // (a long allele that did not fit the pool: overflow is reported anyway)
    const uint8_t* pa = found_alt_bytes(a, pool);
    const uint8_t* pb = found_alt_bytes(b, pool);
    for (int k = 0; k < n; k++)
        if (pa[k] != pb[k]) return false;
    return true;
}
__device__ __forceinline__ uint64_t found_key_hash(const DevFound& f, const uint8_t* pool, int track_open)
{
    uint64_t x = 0xcbf29ce484222325ull;
    auto mix = [&](uint64_t v) { x = (x ^ v) * 0x100000001b3ull; x ^= x >> 29; };
    mix((uint32_t)f.c.position);
    mix((uint32_t)f.c.category | ((uint32_t)f.c.length << 8));
    if (track_open) mix(0x100u | (uint32_t)f.c.open_left | ((uint32_t)f.c.open_right << 1));
    const int n = found_alt_len(f.c);
    if (n <= kFoundInline || f.pool_offset >= 0) {
        const uint8_t* p = found_alt_bytes(f, pool);
        for (int k = 0; k < n; k++) mix(p[k]);
    }
    return x;
}

__global__ __launch_bounds__(256) void found_merge_kernel(const DevFound* __restrict__ rec, int32_t n, const uint8_t* __restrict__ pool,
                                                          int32_t* __restrict__ tab, uint32_t cap_mask, int32_t* __restrict__ acc, int32_t track_open)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63;
    DevFound f = {};
    bool live = false;
    if (i < n) {
        f = rec[i];
        live = f.c.category != kFoundHole;
    }
    const uint64_t key = live ? found_key_hash(f, pool, track_open) : 0ull;
    const int dir = live ? (int)f.c.dir : 3, anchored = live && f.c.well_anchored;
    // ---- the wave's records of one candidate, added up in the lowest lane that holds it
    int sup[3] = {0, 0, 0}, anch[3] = {0, 0, 0};
    bool leader = false;
    unsigned long long todo = __ballot(live);
    while (todo) {
        const int l0 = __builtin_ctzll(todo);
        const uint64_t k0 = ((uint64_t)(uint32_t)__shfl((int)(key >> 32), l0, 64) << 32) | (uint32_t)__shfl((int)(uint32_t)key, l0, 64);
        const int j = __shfl(i, l0, 64);
        bool same = ((todo >> lane) & 1ull) && key == k0;
        if (same && lane != l0) same = found_same_key(f, rec[j], pool, track_open);
        const unsigned long long grp = __ballot(same);
        const int s0 = __popcll(__ballot(same && dir == 0)), s1 = __popcll(__ballot(same && dir == 1)), s2 = __popcll(__ballot(same && dir == 2));
        const int a0 = __popcll(__ballot(same && anchored && dir == 0)), a1 = __popcll(__ballot(same && anchored && dir == 1)),
                  a2 = __popcll(__ballot(same && anchored && dir == 2));
        if (lane == l0) { leader = true; sup[0] = s0; sup[1] = s1; sup[2] = s2; anch[0] = a0; anch[1] = a1; anch[2] = a2; }
        todo &= ~grp;
    }
    if (!leader) return;
    // ---- the table: the first record to claim a slot owns its group
    uint32_t at = (uint32_t)(key ^ (key >> 31)) & cap_mask;
    int owner;
    for (;;) {
        const int cur = atomicCAS(&tab[at], -1, i);
        if (cur == -1) { owner = i; break; }
        if (found_same_key(f, rec[cur], pool, track_open)) { owner = cur; break; }
        at = (at + 1) & cap_mask;
    }
    int32_t* a = acc + (int64_t)owner * kMergeAcc;
    for (int d = 0; d < 3; d++) {
        if (sup[d]) atomicAdd(a + d, sup[d]);
        if (anch[d]) atomicAdd(a + 3 + d, anch[d]);
    }
    atomicMax(a + 6, 0x7FFFFFFF - i);        // (lanes of a wave hold ascending record indices: the leader's is the smallest of its records)
    if (owner == i) a[7] = 1;
}

// out: pinned host memory, out[0 .. *cursor) on return (any order: the host sorts by `first`)
__global__ __launch_bounds__(256) void found_gather_kernel(const DevFound* __restrict__ rec, int32_t n, const int32_t* __restrict__ acc,
                                                           DevMerged* __restrict__ out, unsigned int* __restrict__ cursor)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63;
    const int32_t* a = acc + (int64_t)min(i, n - 1) * kMergeAcc;
    const bool owner = i < n && a[7] == 1;
    const unsigned long long owners = __ballot(owner);
    if (!owners) return;
    unsigned int base = 0;
    if (lane == __builtin_ctzll(owners)) base = atomicAdd(cursor, (unsigned int)__popcll(owners));
    base = (unsigned int)__shfl((int)base, __builtin_ctzll(owners), 64);
    if (!owner) return;
    DevMerged m;
    m.first = 0x7FFFFFFF - a[6];
    m.f = rec[m.first];
    for (int d = 0; d < 3; d++) { m.sup[d] = a[d]; m.anch[d] = a[3 + d]; }
    m.pad = 0;
    out[base + (unsigned int)__popcll(owners & ((1ull << lane) - 1ull))] = m;
}

// ---- MNV calling on, split form: where a batch's merged groups go ----------------------------------------------------------------
// With MNV calling on every mismatch of every read is an SNV candidate of the read walk (bases an MNV candidate took are not): ~1.5
// groups a locus at 2000x, nearly all of them fully anchored SNVs of one or two reads that are never called.  The host needs a
// candidate object only where the read walk's candidates differ from what the allele counts say (surface_flush.inc.h, the dirty loci),
// so the groups part here:
//   fully anchored SNV groups ("plain")  -> the SNV STORE in device memory (SnvGroup, 40 bytes), where they stay until their block is
//                                           flushed; a flush takes the ones on dirty loci (snv_store_sweep_kernel) and drops the rest
//   everything else                      -> pinned host memory as before (DevMerged): MNVs, insertions, deletions, open-ended SNVs when
//                                           open ends are tracked, X-operation span marks
struct SnvGroup {
    int32_t position;
    uint8_t alt;          // the read base (ASCII); the reference base is the chromosome's
    uint8_t pad[3];
    int32_t sup[3], anch[3];
    uint32_t first;       // (batch, first): order of first arrival (RegionState.cs:104-123 keeps a position's candidates in that order)
    uint32_t batch;
};
static_assert(sizeof(SnvGroup) == 40, "SnvGroup is 40 bytes");

__device__ __forceinline__ bool found_is_plain_snv(const FoundCandidate& c, int track_open)
{
    return c.category == PISCES_CAT_SNV && !(track_open && (c.open_left || c.open_right));
}

// cursors: [2] groups written to out_host, [3] groups appended to the store (this batch); store_n: the store's running count (device)
__global__ __launch_bounds__(256) void found_gather_split_kernel(const DevFound* __restrict__ rec, int32_t n, const int32_t* __restrict__ acc,
                                                                 DevMerged* __restrict__ out_host, unsigned int* __restrict__ cursors,
                                                                 SnvGroup* __restrict__ store, unsigned int* __restrict__ store_n, uint32_t store_capacity,
                                                                 uint32_t batch, int32_t track_open)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63;
    const int32_t* a = acc + (int64_t)min(i, n - 1) * kMergeAcc;
    const bool owner = i < n && a[7] == 1;
    if (__ballot(owner) == 0ull) return;
    int32_t first = 0;
    DevFound f = {};
    if (owner) {
        first = 0x7FFFFFFF - a[6];
        f = rec[first];
    }
    const bool plain = owner && found_is_plain_snv(f.c, track_open);
    const bool other = owner && !plain;
    const unsigned long long plains = __ballot(plain), others = __ballot(other);
    unsigned int base_p = 0, base_o = 0;
    if (plains) {
        const int l0 = __builtin_ctzll(plains);
        if (lane == l0) { base_p = atomicAdd(store_n, (unsigned int)__popcll(plains)); atomicAdd(cursors + 3, (unsigned int)__popcll(plains)); }
        base_p = (unsigned int)__shfl((int)base_p, l0, 64);
    }
    if (others) {
        const int l0 = __builtin_ctzll(others);
        if (lane == l0) base_o = atomicAdd(cursors + 2, (unsigned int)__popcll(others));
        base_o = (unsigned int)__shfl((int)base_o, l0, 64);
    }
    const unsigned long long below = (1ull << lane) - 1ull;
    if (plain) {
        const unsigned int at = base_p + (unsigned int)__popcll(plains & below);
        if (at < store_capacity) {
            SnvGroup g;
            g.position = f.c.position;
            g.alt = f.alt[0];
            g.pad[0] = g.pad[1] = g.pad[2] = 0;
            for (int d = 0; d < 3; d++) { g.sup[d] = a[d]; g.anch[d] = a[3 + d]; }
            g.first = (uint32_t)first;
            g.batch = batch;
            store[at] = g;
        } else {
            atomicExch(cursors + 1, 1u);   // (the reservation is the batch's record count: cannot happen; reported as an overflow)
        }
    }
    if (other) {
        DevMerged m;
        m.first = first;
        m.f = f;
        for (int d = 0; d < 3; d++) { m.sup[d] = a[d]; m.anch[d] = a[3 + d]; }
        m.pad = 0;
        out_host[base_o + (unsigned int)__popcll(others & below)] = m;
    }
}

// A flush over the SNV store: the groups on dirty loci (bit (position - bm_first) of bm) go to pinned host memory, the groups of the
// flushed positions [1, drop_hi] that are not dirty are dropped (the tile kernels call those SNVs from the counts), the rest is kept,
// compacted into the other buffer.  counts_out: [0] selected, [1] kept (zeroed before the launch).
__global__ __launch_bounds__(256) void snv_store_sweep_kernel(const SnvGroup* __restrict__ in, const unsigned int* __restrict__ n_in, const uint32_t* __restrict__ bm,
                                                              int32_t bm_first, int32_t bm_n, int32_t drop_hi, SnvGroup* __restrict__ selected,
                                                              uint32_t selected_capacity, SnvGroup* __restrict__ kept, unsigned int* __restrict__ counts_out)
{
    const unsigned int n = *n_in;
    const unsigned int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    SnvGroup g = {};
    bool sel = false, keep = false;
    if (i < n) {
        g = in[i];
        const unsigned rel = (unsigned)(g.position - bm_first);
        sel = bm && rel < (unsigned)bm_n && ((bm[rel >> 5] >> (rel & 31u)) & 1u);
        keep = !sel && g.position > drop_hi;
    }
    const unsigned long long sels = __ballot(sel), keeps = __ballot(keep);
    const unsigned long long below = (1ull << lane) - 1ull;
    if (sels) {
        unsigned int base = 0;
        const int l0 = __builtin_ctzll(sels);
        if (lane == l0) base = atomicAdd(counts_out, (unsigned int)__popcll(sels));
        base = (unsigned int)__shfl((int)base, l0, 64);
        const unsigned int at = base + (unsigned int)__popcll(sels & below);
        if (sel && at < selected_capacity) selected[at] = g;
    }
    if (keeps) {
        unsigned int base = 0;
        const int l0 = __builtin_ctzll(keeps);
        if (lane == l0) base = atomicAdd(counts_out + 1, (unsigned int)__popcll(keeps));
        base = (unsigned int)__shfl((int)base, l0, 64);
        if (keep) kept[base + (unsigned int)__popcll(keeps & below)] = g;
    }
}

// rows[k][198] = counts[idx[k]][198] (a negative index: zeros): the anchor-resolved counts of the few loci the collapser's frequencies and
// the reallocator's Reference candidates read on the host (the tensor itself stays on the device)
// (an index <= -2 names a locus of the FOLDED counts, kernels.hip.h CountsView: its 18 sums go to the well-anchored bin of their cells —
// whoever reads such a row adds up all bins of a cell)
__global__ __launch_bounds__(256) void gather_count_rows_kernel(const int32_t* __restrict__ counts, const int32_t* __restrict__ folded,
                                                                const long long* __restrict__ idx, int32_t n_rows, int32_t* __restrict__ rows)
{
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t k = g / PISCES_COUNTS_PER_LOCUS;
    if (k >= n_rows) return;
    const int c = (int)(g - k * PISCES_COUNTS_PER_LOCUS);
    const long long li = idx[k];
    int v = 0;
    if (li >= 0) v = counts[li * PISCES_COUNTS_PER_LOCUS + c];
    else if (li <= -2 && folded && c % PISCES_NUM_ANCHORS == PISCES_ANCHOR_SIZE) v = folded[(-(li + 2)) * PISCES_FOLDED_PER_LOCUS + c / PISCES_NUM_ANCHORS];
    rows[g] = v;
}

}  // namespace pisces
