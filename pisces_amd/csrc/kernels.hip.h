// kernels.hip.h — gfx950 kernels of the pileup-and-likelihood path.
//
//   call_tiles_kernel        observation tuples -> LDS allele-count histogram (per 64-locus tile)
//                            -> coverage / Poisson q-score / strand bias / somatic genotype / filters
//                            for the Reference allele and every SNV candidate -> 64-byte records.
//                            Counts never leave LDS.  HBM traffic = 4 B/tuple + 1 B/locus + 64 B/record.
//   accumulate_tiles_kernel  tuples -> anchor-resolved int32[6][3][11] counts added to a global tensor
//                            (the IAlleleSource view: RegionState._alleleCounts, RegionState.cs:57).
//   call_counts_kernel       the same call phase fed from that global tensor (streaming surface after
//                            the host collapser has looked at the counts).
//
// One workgroup (256 threads = 4 wave64) per tile; grid = number of tiles (>> 256 CUs for any real
// interval set).  No MFMA: this is a scan + histogram + transcendental epilogue, HBM-bound.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_math.hip.h"

namespace pisces {

constexpr int kBlock = 256;
constexpr int kTile = 64;                 // loci per tile: kTile * 4 allele lanes = one workgroup pass
constexpr int kFolded = 18;               // 6 allele types x 3 directions (anchors folded)
constexpr int kUnroll = 4;                // 16-byte loads in flight per lane
constexpr int kSlotsPerTile = 4 * kTile;  // worst case: four alleles called at every locus
constexpr int kTotalShards = 64;          // running-total shards (one 128-byte line each)
constexpr int kTotalStride = 16;          // in 8-byte words
constexpr int kRefMargin = 32;            // reference bases staged in LDS on each side of a tile (RMxN scan reach)
constexpr int kRefWin = kTile + 2 * kRefMargin;
constexpr int kQLutLds = 128;             // QtoP(q), q < 128, staged in LDS (no global load inside the call phase)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));   // one dwordx4 load

// alphabetical allele order A, C, G, T (the per-locus output order, AlleleCaller.cs:172-176)
// expressed in AlleleType codes A=0, G=1, C=2, T=3
__device__ __forceinline__ int allele_of_rank(int k) { return k == 1 ? 2 : (k == 2 ? 1 : k); }

__device__ __forceinline__ int allele_type_of_base(uint8_t c)  // AlleleHelper.GetAlleleType, AlleleHelper.cs:13-32
{
    return c == 'A' ? 0 : c == 'G' ? 1 : c == 'C' ? 2 : c == 'T' ? 3 : 4;
}

// One observation into the folded LDS histogram hist[(allele*3+dir)*kTile + locus].
// "qual < minBQ -> N" is RegionStateManager.cs:179-181; deletion tuples carry qual 255.
// Padding tuples (0xFFFFFFFF) decode to allele 7 and fall out of the range test.
__device__ __forceinline__ void accumulate_folded(int* hist, uint32_t t, uint32_t n_loci, uint32_t min_bq)
{
    uint32_t locus = t & 0x7FFFu;
    uint32_t dir = (t >> 19) & 3u;
    uint32_t allele = (t >> 21) & 7u;
    uint32_t qual = t >> 24;
    if (allele < 4u && qual < min_bq) allele = 4u;
    if (locus < n_loci && dir < 3u && allele < 6u) atomicAdd(&hist[(allele * 3u + dir) * kTile + locus], 1);
}

// Streams tuples[begin, end) through `op(tuple)`: scalar head/tail up to 16-byte alignment, then
// kUnroll independent dwordx4 loads per lane per iteration (1 KiB per wave-instruction, coalesced).
template <typename Op>
__device__ __forceinline__ void stream_tuples(const uint32_t* __restrict__ tuples, int64_t begin, int64_t end, Op op)
{
    const int tid = threadIdx.x;
    int64_t abegin = (begin + 3) & ~(int64_t)3;
    int64_t aend = end & ~(int64_t)3;
    if (abegin > aend) { abegin = end; aend = end; }
    for (int64_t i = begin + tid; i < abegin; i += kBlock) op(tuples[i]);
    const u32x4* __restrict__ p4 = reinterpret_cast<const u32x4*>(tuples + abegin);
    const int64_t n4 = (aend - abegin) >> 2;
    for (int64_t i = tid; i < n4; i += (int64_t)kBlock * kUnroll) {
        u32x4 v[kUnroll];
#pragma unroll
        for (int u = 0; u < kUnroll; u++) {
            int64_t j = i + (int64_t)u * kBlock;
            // streamed exactly once: non-temporal so the tuples do not evict the reference / records from L2
            v[u] = j < n4 ? __builtin_nontemporal_load(&p4[j]) : (u32x4){~0u, ~0u, ~0u, ~0u};
        }
#pragma unroll
        for (int u = 0; u < kUnroll; u++) {
            op(v[u].x); op(v[u].y); op(v[u].z); op(v[u].w);
        }
    }
    for (int64_t i = aend + tid; i < end; i += kBlock) op(tuples[i]);
}

// exclusive prefix sum of a per-thread 0/1 flag over the 256-thread block (wave ballot + 4 wave totals)
__device__ __forceinline__ int block_exclusive_count(bool flag, int* wave_tot /* LDS[4] */, int* total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long m = __ballot(flag);
    int before = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) wave_tot[wave] = __popcll(m);
    __syncthreads();
    int off = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) off += (w < wave) ? wave_tot[w] : 0;
    *total = wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
    __syncthreads();
    return off + before;
}

// ---- the call phase: one lane per candidate allele ------------------------------------------
// PMC showed the call phase VALU-issue-bound (FP64 at 4 cycles per wave-instruction), so the layout that
// minimises wave-instructions wins: candidates compacted onto the low lanes, one lane runs one allele
// start to finish, and everything that cannot change the output is skipped:
//  * IsCallable (AlleleCaller.cs:236-258) tests coverage and frequency BEFORE the q-score, so a variant
//    candidate below MinFrequency (every sequencing-error allele at depth) is rejected on integer/float32
//    arithmetic alone, and one below MinVariantQscore right after its q-score; rejected alleles never reach
//    the strand-bias / genotype math because their CalledAllele is dropped by the reference too;
//  * the strand-bias tails of a well-supported allele are exactly 1.0 (poisson_cdf_sb).

struct PointCounts {
    int cov[3], sup[3];
    int total, nocalls, refsup, support;
};

// CoverageCalculator.CalculateSinglePoint (CoverageCalculator.cs:49-98) from the folded counts of one locus
__device__ __forceinline__ PointCounts point_counts(const int* hist, int l, int allele, bool isRef, int refType, int gapped)
{
    PointCounts c;
    int h[6][3];
#pragma unroll
    for (int k = 0; k < kFolded; k++) h[k / 3][k % 3] = hist[k * kTile + l];
    c.total = 0; c.nocalls = 0; c.refsup = 0;
    const int supAllele = isRef ? refType : allele;
#pragma unroll
    for (int d = 0; d < 3; d++) {
        c.cov[d] = h[PISCES_ALLELE_A][d] + h[PISCES_ALLELE_C][d] + h[PISCES_ALLELE_G][d] + h[PISCES_ALLELE_T][d] +
                   h[PISCES_ALLELE_DEL][d];
        c.total += c.cov[d];
        c.nocalls += h[PISCES_ALLELE_N][d];
        c.sup[d] = 0;
    }
#pragma unroll
    for (int a = 0; a < 6; a++) {
        if (a == refType && a < PISCES_ALLELE_N) c.refsup += h[a][0] + h[a][1] + h[a][2];
        if (a == supAllele) { c.sup[0] = h[a][0]; c.sup[1] = h[a][1]; c.sup[2] = h[a][2]; }
    }
    c.support = c.sup[0] + c.sup[1] + c.sup[2];   // AlleleHelper.Map: AlleleSupport = candidate.Support
    if (isRef) { c.support -= gapped; if (c.support < 0) c.support = 0; }    // CoverageCalculator.cs:94-97
    else { c.refsup -= gapped; if (c.refsup < 0) c.refsup = 0; }             // :90-93
    return c;
}

// IsCallable (AlleleCaller.cs:236-258), the tests that precede the q-score: coverage, then frequency.
__device__ __forceinline__ bool variant_passes_frequency(const PointCounts& c, const DeviceParams& P)
{
    if (c.total < P.min_cov && !P.include_ref) return false;
    if (c.total != 0 && frequency_f(c.support, c.total) < P.min_freq) return false;
    return true;
}

// Everything of AlleleCaller.ProcessVariant (AlleleCaller.cs:208-234) after the q-score and the strand-bias
// statistics: AlleleProcessor.ApplyFilters (AlleleProcessor.cs:25-71), SomaticGenotyper, record packing.
__device__ inline void finish_allele(const PointCounts& c, int pos, int a, bool isRef, int rt, int vq, const SbResult& sb,
                                     const uint8_t* __restrict__ ref, int64_t win_lo, int64_t win_hi, const DeviceParams& P,
                                     PiscesCalledAllele& r, const uint8_t* s_refwin, int s_refidx)
{
    const float freq = frequency_f(c.support, c.total);
    // SetFractionNoCalls (CalledAllele.cs:107-114) + ApplyFilters
    const float allReads = (float)(c.total + c.nocalls);
    const float fractionNoCalls = (allReads == 0.0f) ? 0.0f : ((float)c.nocalls / allReads);
    uint32_t filters = 0;
    if (P.low_depth_filter >= 0 && c.total < P.low_depth_filter) filters |= 1u << PISCES_FILTER_LOW_DEPTH;
    if (P.vq_filter >= 0 && vq < P.vq_filter && c.total != 0) filters |= 1u << PISCES_FILTER_LOW_VARIANT_QSCORE;
    if (!isRef) {
        if (P.nocall_thr >= 0.0f && fractionNoCalls > P.nocall_thr) filters |= 1u << PISCES_FILTER_NO_CALL;
        if (!sb.acceptable || (P.filter_single_strand && !sb.var_both)) filters |= 1u << PISCES_FILTER_STRAND_BIAS;
        const uint8_t bases[4] = {'A', 'G', 'C', 'T'};
        if (rt < 4 && a < 4) {
            // the reference window of the tile sits in LDS (global byte loads are dependent multi-microsecond
            // round trips while the chip is streaming); fall back to HBM only for an RMxN reach beyond the margin
            const bool hit = (s_refwin && P.rmxn_min_rep <= kRefMargin)
                                 ? rmxn_should_filter_snv_lds(s_refwin, kRefWin, s_refidx, bases[rt], bases[a], freq, P)
                                 : rmxn_should_filter_snv(ref, win_lo, win_hi, pos, bases[rt], bases[a], freq, P);
            if (hit) filters |= 1u << PISCES_FILTER_RMXN;
        }
        if (P.vf_filter >= 0.0f && freq < P.vf_filter) filters |= 1u << PISCES_FILTER_LOW_VARIANT_FREQUENCY;
    }
    const int gt = somatic_genotype(isRef, c.total, c.support, c.refsup, P);
#if !(defined(PISCES_ABLATE_MATH) && PISCES_ABLATE_MATH == 3)
    const int gq = somatic_gq(gt, vq, c.total, c.support, P);
#else
    const int gq = vq;
#endif
    if (P.low_gq_filter >= 0 && (float)gq < (float)P.low_gq_filter) filters |= 1u << PISCES_FILTER_LOW_GENOTYPE_QUALITY;

    r.position = pos;
    r.total_coverage = c.total;
    r.allele_support = c.support;
    r.reference_support = c.refsup;
    r.num_no_calls = c.nocalls;
    r.coverage_by_dir[0] = c.cov[0]; r.coverage_by_dir[1] = c.cov[1]; r.coverage_by_dir[2] = c.cov[2];
    r.support_by_dir[0] = c.sup[0]; r.support_by_dir[1] = c.sup[1]; r.support_by_dir[2] = c.sup[2];
    r.variant_qscore = vq;
    r.strand_bias_score = sb.bias_score;
    r.genotype_qscore = gq;
    r.filter_bits = (uint16_t)filters;
    r.info = PISCES_INFO_PACK(gt, isRef ? PISCES_CAT_REFERENCE : PISCES_CAT_SNV, rt, isRef ? rt : a, sb.acceptable, sb.var_both,
                              sb.cov_both);
}

// One lane, one allele, start to finish. Returns false (record untouched) when the reference would drop the allele.
__device__ inline bool process_point_allele(const PointCounts& c, int pos, int a, bool isRef, int rt,
                                             const uint8_t* __restrict__ ref, int64_t win_lo, int64_t win_hi,
                                             const DeviceParams& P, PiscesCalledAllele& r,
                                             const uint8_t* s_refwin = nullptr, int s_refidx = 0)
{
    if (!isRef && !variant_passes_frequency(c, P)) return false;
    int vq = 0;
#if !(defined(PISCES_ABLATE_MATH) && PISCES_ABLATE_MATH == 5)
    if (c.support > 0 && c.total != 0) vq = poisson_qscore(c.support, c.total, P);   // VariantQualityCalculator.Compute :11-24
#endif
    if (!isRef && vq < P.min_vq) return false;
    SbResult sb = {0.0, 0, 0, 0};
#if !(defined(PISCES_ABLATE_MATH) && PISCES_ABLATE_MATH == 4)
    if (c.support > 0) sb = strand_bias(c.cov, c.sup, P);                            // StrandBiasCalculator.Compute :10-15
#endif
    finish_allele(c, pos, a, isRef, rt, vq, sb, ref, win_lo, win_hi, P, r, s_refwin, s_refidx);
    return true;
}

// The call phase for one tile whose folded counts sit in LDS (hist[(allele*3+dir)*kTile + locus]).
// lane (locus = tid>>2, rank = tid&3) decides whether (locus, allele-of-rank) is a candidate:
//   Reference candidate per position  — RegionState.GetAllCandidates, RegionState.cs:414-447
//   SNV candidate                     — a quality-passing base != reference base, ref and read not N
//                                       (CandidateVariantFinder.cs:97-160 with callMNVs off; its
//                                       SupportByDirection equals the allele count by direction)
// candidates are compacted onto the low threads (at most 4*kTile = kBlock of them), processed one lane each,
// then the per-locus rule "drop the Reference allele when a variant is called" (AlleleCaller.cs:146-147)
// and the output order (position, then allele) are applied and the records written to HBM.
__device__ inline void call_phase(const int* hist, const uint32_t* gapped /* LDS[kTile] or nullptr */,
                                  const PiscesTile& tile, const uint8_t* __restrict__ ref, int32_t ref_start,
                                  int64_t ref_len, PiscesCalledAllele* __restrict__ records, int32_t capacity,
                                  int32_t* __restrict__ record_count, PiscesTileResult* __restrict__ tile_result,
                                  const DeviceParams& P, int* s_wave, uint8_t* s_work, uint8_t* s_callable, int* s_base)
{
    const int tid = threadIdx.x;
#ifdef PISCES_TIMING
    __shared__ long long s_dbg[8];
    if (tid == 0) s_dbg[0] = clock64();
#endif
    // Two work lists so that a wave is not held hostage by one divergent lane: Reference candidates (at most
    // one per locus, a cheap uniform path) are compacted onto threads 0..63 = wave 0, variant candidates
    // (rare, long data-dependent path) onto threads 64.. = waves 1-3.  s_work[thread] = key = 4*locus + rank.
    int n_ref, n_var;
    {
        const int locus = tid >> 2, rank = tid & 3;
        const int allele = allele_of_rank(rank);
        const int64_t ridx = (int64_t)tile.start_position + locus - ref_start;   // index into the reference window
        const bool in_ref = locus < tile.n_loci && ridx >= 0 && ridx < ref_len;
        const int refType = in_ref ? allele_type_of_base(ref[ridx]) : PISCES_ALLELE_N;
        bool is_ref_work = false, is_var_work = false;
        if (in_ref) {
            int mine = 0, all = 0;
#pragma unroll
            for (int c = 0; c < kFolded; c++) {
                int v = hist[c * kTile + locus];
                all += v;
                if (c / 3 == allele) mine += v;
            }
            const bool refLane = (refType < 4) ? (allele == refType) : (rank == 0);
            if (refLane) is_ref_work = P.include_ref && (P.emit_zero_cov || all > 0);
            else is_var_work = (refType < 4) && mine > 0;
        }
        const int rslot = block_exclusive_count(is_ref_work, s_wave, &n_ref);
        const int vslot = block_exclusive_count(is_var_work, s_wave, &n_var);
        if (is_ref_work) s_work[rslot] = (uint8_t)tid;
        if (is_var_work) s_work[kTile + vslot] = (uint8_t)tid;
        s_callable[tid] = 0;
    }
    __syncthreads();

#ifdef PISCES_TIMING
    if (tid == 0) s_dbg[1] = clock64();
#endif
    PiscesCalledAllele rec;
    bool callable = false;
    int item = 0;
    bool item_is_ref = false;
    if (tid < n_ref || (tid >= kTile && tid - kTile < n_var)) {
        item = s_work[tid];
        const int l = item >> 2;
        const int a = allele_of_rank(item & 3);
        const int pos = tile.start_position + l;
        const int rt = allele_type_of_base(ref[(int64_t)pos - ref_start]);
        item_is_ref = tid < kTile;
        const PointCounts c = point_counts(hist, l, a, item_is_ref, rt, gapped ? (int)gapped[l] : 0);
        callable = process_point_allele(c, pos, a, item_is_ref, rt, ref, (int64_t)ref_start - 1,
                                        (int64_t)ref_start - 1 + ref_len, P, rec);
        if (callable) s_callable[item] = item_is_ref ? 1 : 2;
    }
#ifdef PISCES_TIMING
    if (tid == 0) s_dbg[2] = clock64();     // wave 0 (Reference lanes) done
    if (tid == 64) s_dbg[4] = clock64();    // wave 1 (variant lanes) done
#endif
    __syncthreads();
#ifdef PISCES_TIMING
    if (tid == 0) s_dbg[3] = clock64();
#endif

    // per-locus pruning (AlleleCaller.cs:146-147) in key order: thread t looks at key t
    bool key_survives = false, key_first = false, key_callable = false;
    {
        const uint8_t mine = s_callable[tid];
        key_callable = mine != 0;
        if (key_callable) {
            const int q = tid & ~3;
            const bool any_variant = (s_callable[q] | s_callable[q + 1] | s_callable[q + 2] | s_callable[q + 3]) & 2;
            key_survives = !(mine == 1 && any_variant);
            if (key_survives) {
                bool earlier = false;   // an earlier surviving allele at this locus?
                for (int k = 0; k < (tid & 3); k++) {
                    uint8_t cc = s_callable[q + k];
                    if (cc == 2 || (cc == 1 && !any_variant)) earlier = true;
                }
                key_first = !earlier;
            }
        }
    }
    int n_callable, n_loci_called, n_surv;
    (void)block_exclusive_count(key_callable, s_wave, &n_callable);
    (void)block_exclusive_count(key_first, s_wave, &n_loci_called);
    const int key_idx = block_exclusive_count(key_survives, s_wave, &n_surv);
    // hand each surviving key its output index (position, then allele order); 0xFF = dropped
    s_work[tid] = key_survives ? (uint8_t)key_idx : (uint8_t)0xFF;   // n_surv <= 4*64 but idx 255 needs key 255 surviving with 255 before it: impossible (refs are pruned)
    if (tid == 0) {
        // record placement: fixed 256-slot stride per tile (no atomics, deterministic), or — when the caller
        // wants a compact buffer — one returning atomic per tile on a shared counter (same-address global
        // atomics serialize in L2 at ~12 ns each: measurable at thousands of tiles per launch)
        int base = record_count ? (n_surv > 0 ? atomicAdd(record_count, n_surv) : 0) : (int)blockIdx.x * kSlotsPerTile;
        *s_base = base;
        PiscesTileResult tr;
        tr.record_begin = base;
        tr.n_records = n_surv;
        tr.n_candidate_loci = n_loci_called;
        tr.reserved = n_callable;   // IsCallable == true count (IAlleleCaller.TotalNumCalled)
        *tile_result = tr;
        if (P.totals) {
            // running totals, sharded over kTotalShards cache lines so the adds do not serialize on one L2 line
            unsigned long long* tt = P.totals + (size_t)(blockIdx.x % kTotalShards) * kTotalStride;
            atomicAdd(&tt[0], (unsigned long long)n_surv);
            atomicAdd(&tt[1], (unsigned long long)n_loci_called);
            atomicAdd(&tt[2], (unsigned long long)n_callable);
            atomicAdd(&tt[3], 1ull);
        }
    }
    __syncthreads();
    if (callable && s_work[item] != 0xFF) {
        const int64_t dst = (int64_t)(*s_base) + s_work[item];
        if (dst < capacity) {
            const uint4* sp = reinterpret_cast<const uint4*>(&rec);
            uint4* dp = reinterpret_cast<uint4*>(&records[dst]);
            dp[0] = sp[0]; dp[1] = sp[1]; dp[2] = sp[2]; dp[3] = sp[3];
        }
    }
#ifdef PISCES_TIMING
    __syncthreads();
    if (tid == 0) {
        const long long now = clock64();
        tile_result->n_records = (int)(s_dbg[1] - s_dbg[0]);          // detection + compaction
        tile_result->n_candidate_loci = (int)(s_dbg[2] - s_dbg[1]);   // Reference lanes (wave 0)
        tile_result->reserved = (int)(now - s_dbg[0]);                // whole call phase
    }
#endif
}

// ------------------------------------------------------------------------------------------
// The hot kernel.  One workgroup per tile; four waves stream the tile's tuples into the LDS histogram, then
// waves 2 and 3 retire (their wave slots go to the next workgroup, whose streaming overlaps this tile's FP64
// call phase), wave 1 calls the variant candidates and wave 0 the Reference candidates, lane = locus.
// Measured on config 2: with all four waves held through a block-wide call phase the launch ran stream and
// call phases in lock-step across the chip (HBM idle ~30 us of 74); per-tile stamps: stream 45 k cycles,
// call 23 k of which 13 k Reference lanes, 10 k compaction / block barriers / waiting on the variant wave.
__device__ __forceinline__ int wave_exclusive_sum(int v, int* total)
{
    const int lane = threadIdx.x & 63;
    int x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int y = __shfl_up(x, d, 64);
        if (lane >= d) x += y;
    }
    *total = __shfl(x, 63, 64);
    return x - v;
}

__device__ __forceinline__ void copy_record(PiscesCalledAllele* dst, const PiscesCalledAllele* src)
{
    const uint4* sp = reinterpret_cast<const uint4*>(src);
    uint4* dp = reinterpret_cast<uint4*>(dst);
    dp[0] = sp[0]; dp[1] = sp[1]; dp[2] = sp[2]; dp[3] = sp[3];
}

// The call phase, lane = locus, one role per wave (wave-uniform code, no divergence between roles):
//   wave 0  Reference candidate of the locus, start to finish
//   wave 1  variant candidates: variant q-score            (then assembles the variant records)
//   wave 2  variant candidates: strand-bias overall + forward statistics
//   wave 3  variant candidates: strand-bias reverse statistics
// A called SNV is ~5 k dependent FP64 instructions on one lane (measured p99 26 us per tile with one variant
// wave against 6 us for the Reference wave); its q-score and three strand-bias tails are independent, so they
// run on three SIMDs at once and meet in LDS.
struct VarScratch {
    int vq[kTile * 4];
    double ov_var[kTile * 4], fw_var[kTile * 4], fw_fp[kTile * 4], rv_var[kTile * 4], rv_fp[kTile * 4];
};

__device__ inline void call_four_waves(const int* hist, const uint32_t* gapped, const PiscesTile& tile, int tile_index,
                                       const uint8_t* __restrict__ ref, int32_t ref_start, int64_t ref_len,
                                       PiscesCalledAllele* __restrict__ records, int32_t capacity,
                                       int32_t* __restrict__ record_count, PiscesTileResult* __restrict__ tile_result,
                                       const DeviceParams& P, PiscesCalledAllele* s_rec, uint8_t* s_mask,
                                       const uint8_t* s_refwin /* LDS[kRefWin], 0 = outside the reference */,
                                       VarScratch* vs)
{
    const int wave = threadIdx.x >> 6;
    const int l = threadIdx.x & 63;
    const int pos = tile.start_position + l;
    const uint8_t refb = s_refwin[kRefMargin + l];
    const bool in_ref = l < tile.n_loci && refb != 0;
    const int rt = in_ref ? allele_type_of_base(refb) : PISCES_ALLELE_N;
    const int64_t win_lo = (int64_t)ref_start - 1, win_hi = win_lo + ref_len;
    const int g = gapped ? (int)gapped[l] : 0;

    PiscesCalledAllele rec;      // wave 0: this locus' Reference record
    bool ref_emitted = false;
    if (wave == 0) {
        // Reference candidate (RegionState.GetAllCandidates, RegionState.cs:414-447)
        if (in_ref && P.include_ref) {
            int all = 0;
#pragma unroll
            for (int c = 0; c < kFolded; c++) all += hist[c * kTile + l];
            if (P.emit_zero_cov || all > 0) {
                const int a = (rt < 4) ? rt : PISCES_ALLELE_N;
                const PointCounts c = point_counts(hist, l, a, true, rt, g);
                (void)process_point_allele(c, pos, a, true, rt, ref, win_lo, win_hi, P, rec, s_refwin, kRefMargin + l);
                ref_emitted = true;
            }
        }
    } else if (in_ref && rt < 4) {
        // variant candidates (CandidateVariantFinder.cs:97-160, callMNVs off): quality-passing base != ref base
        for (int k = 0; k < 4; k++) {
            const int a = allele_of_rank(k);
            if (a == rt) continue;
            if (hist[(a * 3 + 0) * kTile + l] + hist[(a * 3 + 1) * kTile + l] + hist[(a * 3 + 2) * kTile + l] == 0) continue;
            const PointCounts c = point_counts(hist, l, a, false, rt, g);
            if (!variant_passes_frequency(c, P)) continue;
            const int slot = l * 4 + k;
            if (wave == 1) {
                vs->vq[slot] = (c.support > 0 && c.total != 0) ? poisson_qscore(c.support, c.total, P) : 0;
            } else if (c.support > 0) {
                if (wave == 2) {
                    const SbStats ov = sb_stats_of(0, c.cov, c.sup, P), fw = sb_stats_of(1, c.cov, c.sup, P);
                    vs->ov_var[slot] = ov.var_gt_zero;
                    vs->fw_var[slot] = fw.var_gt_zero;
                    vs->fw_fp[slot] = fw.false_pos;
                } else {
                    const SbStats rv = sb_stats_of(2, c.cov, c.sup, P);
                    vs->rv_var[slot] = rv.var_gt_zero;
                    vs->rv_fp[slot] = rv.false_pos;
                }
            }
        }
    }
    __syncthreads();   // four waves
    if (wave >= 2) return;
    if (wave == 1) {
        uint32_t mask = 0;
        if (in_ref && rt < 4) {
            for (int k = 0; k < 4; k++) {
                const int a = allele_of_rank(k);
                if (a == rt) continue;
                if (hist[(a * 3 + 0) * kTile + l] + hist[(a * 3 + 1) * kTile + l] + hist[(a * 3 + 2) * kTile + l] == 0) continue;
                const PointCounts c = point_counts(hist, l, a, false, rt, g);
                if (!variant_passes_frequency(c, P)) continue;
                const int slot = l * 4 + k;
                const int vq = vs->vq[slot];
                if (vq < P.min_vq) continue;                                   // IsCallable, last test
                SbResult sb = {0.0, 0, 0, 0};
                if (c.support > 0) {
                    SbStats ov, fw, rv;
                    ov.var_gt_zero = vs->ov_var[slot];
                    fw.var_gt_zero = vs->fw_var[slot]; fw.false_pos = vs->fw_fp[slot];
                    rv.var_gt_zero = vs->rv_var[slot]; rv.false_pos = vs->rv_fp[slot];
                    const int s2 = c.sup[2] / 2, c2 = c.cov[2] / 2;
                    fw.coverage = c.cov[0] + c2; fw.support = c.sup[0] + s2;
                    rv.coverage = c.cov[1] + c2; rv.support = c.sup[1] + s2;
                    ov.false_pos = 0; ov.coverage = 0; ov.support = 0;
                    sb = sb_combine(ov, fw, rv, P);
                }
                PiscesCalledAllele r;
                finish_allele(c, pos, a, false, rt, vq, sb, ref, win_lo, win_hi, P, r, s_refwin, kRefMargin + l);
                copy_record(&s_rec[slot], &r);
                mask |= 1u << k;
            }
        }
        s_mask[l] = (uint8_t)mask;
    }
    __syncthreads();   // waves 0 and 1
    if (wave == 1) return;

    // per-locus pruning (AlleleCaller.cs:146-147), output order (position, then allele), write-out
    const uint32_t vmask = s_mask[l];
    const int mine = vmask ? __popc(vmask) : (ref_emitted ? 1 : 0);
    const int n_callable = __popc(vmask) + (ref_emitted ? 1 : 0);   // IsCallable is always true for a Reference allele
    int n_surv, n_call_total;
    const int excl = wave_exclusive_sum(mine, &n_surv);
    (void)wave_exclusive_sum(n_callable, &n_call_total);
    const int n_loci_called = __popcll(__ballot(mine > 0));
    int base = 0;
    if (l == 0) {
        // record placement: fixed 256-slot stride per tile (no atomics, deterministic) or a compact buffer
        // handed out by one returning atomic per tile
        base = record_count ? (n_surv > 0 ? atomicAdd(record_count, n_surv) : 0) : tile_index * kSlotsPerTile;
        PiscesTileResult tr;
        tr.record_begin = base;
        tr.n_records = n_surv;
        tr.n_candidate_loci = n_loci_called;
        tr.reserved = n_call_total;   // IAlleleCaller.TotalNumCalled contribution
        *tile_result = tr;
        if (P.totals) {
            // running totals, sharded over kTotalShards cache lines so the adds do not serialize on one L2 line
            unsigned long long* tt = P.totals + (size_t)(tile_index % kTotalShards) * kTotalStride;
            atomicAdd(&tt[0], (unsigned long long)n_surv);
            atomicAdd(&tt[1], (unsigned long long)n_loci_called);
            atomicAdd(&tt[2], (unsigned long long)n_call_total);
            atomicAdd(&tt[3], 1ull);
        }
    }
    base = __shfl(base, 0, 64);
    if (vmask) {
        int j = 0;
        for (int k = 0; k < 4; k++) {
            if (!(vmask & (1u << k))) continue;
            const int64_t dst = (int64_t)base + excl + j;
            j++;
            if (dst < capacity) copy_record(&records[dst], &s_rec[l * 4 + k]);
        }
    } else if (ref_emitted) {
        const int64_t dst = (int64_t)base + excl;
        if (dst < capacity) copy_record(&records[dst], &rec);
    }
}

__global__ __launch_bounds__(kBlock, 5) void call_tiles_kernel(
    const uint32_t* __restrict__ tuples, const PiscesTile* __restrict__ tiles, int32_t n_tiles,
    const uint8_t* __restrict__ ref, int32_t ref_start, int64_t ref_len, PiscesCalledAllele* __restrict__ records,
    int32_t capacity, int32_t* __restrict__ record_count, PiscesTileResult* __restrict__ tile_results, DeviceParams P,
    int32_t* __restrict__ gate, int32_t gate_epoch, int32_t gate_width)
{
    __shared__ __attribute__((aligned(16))) PiscesCalledAllele s_rec[kTile * 4];
    __shared__ int hist[kFolded * kTile];
    __shared__ uint8_t s_mask[kTile];
    __shared__ uint8_t s_refwin[kRefWin];
    __shared__ VarScratch s_var;
    __shared__ double s_qlut[kQLutLds];

    const int t = blockIdx.x;
    if (t >= n_tiles) return;
    const PiscesTile tile = tiles[t];
#ifdef PISCES_TIMING
    const long long tc0 = wall_clock64();   // 100 MHz, chip-global
#endif
    for (int i = threadIdx.x; i < kFolded * kTile; i += kBlock) hist[i] = 0;
    if (threadIdx.x < kRefWin) {
        // the tile's reference bases (+ margin for the RMxN scan) into LDS now, under the stream
        const int64_t ri = (int64_t)tile.start_position - kRefMargin + threadIdx.x - ref_start;
        s_refwin[threadIdx.x] = (ri >= 0 && ri < ref_len) ? ref[ri] : (uint8_t)0;
    } else if (threadIdx.x >= 128 && threadIdx.x < 128 + kQLutLds) {
        const int q = threadIdx.x - 128;
        s_qlut[q] = (P.q_to_p_lut && q < P.q_to_p_n) ? P.q_to_p_lut[q] : q_to_p((double)q);
    }
    P.q_to_p_lut = s_qlut;   // generic pointer to LDS from here on
    P.q_to_p_n = kQLutLds;
    // Streaming window: tile t starts streaming once tile t - gate_width has finished streaming.  ~1300 waves with
    // 4 KiB in flight saturate HBM; letting every resident workgroup stream at once only makes all of them finish
    // together and then run their FP64 call phases together with HBM idle.  The window spreads the call phases
    // under the stream of later tiles.  It is a throttle, not a correctness dependency: the wait is bounded and a
    // workgroup proceeds regardless (dispatch order is observed, not guaranteed, to follow blockIdx).
    if (gate_width > 0 && t >= gate_width) {
        if (threadIdx.x == 0) {
            int spins = 0;
            while (__hip_atomic_load(&gate[t - gate_width], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != gate_epoch &&
                   ++spins < 20000)
                __builtin_amdgcn_s_sleep(4);
        }
    }
    __syncthreads();

    const uint32_t n_loci = (uint32_t)tile.n_loci, min_bq = (uint32_t)P.min_bq;
#if defined(PISCES_ABLATE) && PISCES_ABLATE == 2
    // development ablation: loads only (no LDS atomics, no call phase)
    uint32_t acc = 0;
    stream_tuples(tuples, tile.tuple_begin, tile.tuple_end, [&](uint32_t v) { acc ^= v; });
    if (acc == 0x12345u) hist[threadIdx.x] = (int)(n_loci + min_bq);
#else
    stream_tuples(tuples, tile.tuple_begin, tile.tuple_end,
                  [&](uint32_t v) { accumulate_folded(hist, v, n_loci, min_bq); });
#endif
    __syncthreads();
    if (gate_width > 0 && threadIdx.x == 0)
        __hip_atomic_store(&gate[t], gate_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#if defined(PISCES_ABLATE) && PISCES_ABLATE >= 1
    // development ablation: no call phase
    if (threadIdx.x == 0) { PiscesTileResult tr = {0, 0, hist[5], 0}; tile_results[t] = tr; }
#else
#ifdef PISCES_TIMING
    const long long tc1 = wall_clock64();
#endif
    call_four_waves(hist, nullptr, tile, t, ref, ref_start, ref_len, records, capacity, record_count, &tile_results[t], P,
                    s_rec, s_mask, s_refwin, &s_var);
#ifdef PISCES_TIMING
    if (threadIdx.x == 0) {   // development instrumentation: shader-clock stamps in the tile directory
        const long long tc2 = wall_clock64();
        tile_results[t].record_begin = (int)(tc1 - tc0);             // stream (10 ns ticks)
        tile_results[t].n_records = 0;
        tile_results[t].n_candidate_loci = 0;
        tile_results[t].reserved = (int)(tc2 - tc1);                 // call end
    }
#endif
#endif
}

// ------------------------------------------------------------------------------------------
// Software-pipelined persistent variant of call_tiles_kernel.
//
// In call_tiles_kernel every co-resident workgroup streams at the same time and then runs its FP64 call
// phase at the same time, so HBM sits idle during the call phase (measured: 41 us streaming + 32 us call).
// Here a workgroup is 5 waves and walks tiles t = blockIdx.x, +gridDim.x, ...:
//     waves 0-3  stream tile i+1's tuples into hist[(i+1)&1]      (HBM + LDS atomics)
//     wave  4    runs the call phase of tile i from hist[i&1]     (FP64 VALU), then clears that buffer
// with one workgroup barrier per tile.  The call wave keeps lane = locus: variant candidates first (at most
// three per lane, nearly all rejected by the integer frequency test), then the Reference allele unless a
// variant was called at the locus (AlleleCaller.cs:146-147); records are staged in LDS, ordered with a
// wave prefix sum and written as 64-byte rows.
constexpr int kStreamWaves = 4;
constexpr int kPipeBlock = (kStreamWaves + 1) * 64;

// One wave, lane = locus. hist: folded counts of the tile; s_rec: LDS staging [kTile][4] records.
__device__ inline void call_wave(const int* hist, const uint32_t* gapped, const PiscesTile& tile, int tile_index,
                                 const uint8_t* __restrict__ ref, int32_t ref_start, int64_t ref_len,
                                 PiscesCalledAllele* __restrict__ records, int32_t capacity,
                                 int32_t* __restrict__ record_count, PiscesTileResult* __restrict__ tile_result,
                                 const DeviceParams& P, PiscesCalledAllele* s_rec)
{
    const int l = threadIdx.x & 63;
    const int pos = tile.start_position + l;
    const int64_t ridx = (int64_t)pos - ref_start;
    const bool in_ref = l < tile.n_loci && ridx >= 0 && ridx < ref_len;
    const int rt = in_ref ? allele_type_of_base(ref[ridx]) : PISCES_ALLELE_N;
    const int64_t win_lo = (int64_t)ref_start - 1, win_hi = win_lo + ref_len;
    const int g = gapped ? (int)gapped[l] : 0;

    uint32_t called_mask = 0;   // bit k = allele of alphabetical rank k has a record in s_rec[l*4+k]
    int n_callable = 0;
    if (in_ref) {
        int cnt[4] = {0, 0, 0, 0}, all = 0;
#pragma unroll
        for (int c = 0; c < kFolded; c++) {
            int v = hist[c * kTile + l];
            all += v;
            if (c / 3 < 4) cnt[c / 3] += v;
        }
        // variant candidates (CandidateVariantFinder.cs:97-160, callMNVs off): quality-passing base != ref base
        if (rt < 4) {
            for (int k = 0; k < 4; k++) {
                const int a = allele_of_rank(k);
                if (a == rt || cnt[a] == 0) continue;
                const PointCounts c = point_counts(hist, l, a, false, rt, g);
                PiscesCalledAllele r;
                if (process_point_allele(c, pos, a, false, rt, ref, win_lo, win_hi, P, r)) {
                    const uint4* sp = reinterpret_cast<const uint4*>(&r);
                    uint4* dp = reinterpret_cast<uint4*>(&s_rec[l * 4 + k]);
                    dp[0] = sp[0]; dp[1] = sp[1]; dp[2] = sp[2]; dp[3] = sp[3];
                    called_mask |= 1u << k;
                    n_callable++;
                }
            }
        }
        // Reference candidate (RegionState.GetAllCandidates, RegionState.cs:414-447); IsCallable is always true
        // for it (TotalNumCalled counts it) but its record is dropped when a variant was called at the locus
        if (P.include_ref && (P.emit_zero_cov || all > 0)) {
            n_callable++;
            if (called_mask == 0) {
                const int kref = (rt < 4) ? (rt == 0 ? 0 : rt == 2 ? 1 : rt == 1 ? 2 : 3) : 0;
                const int a = (rt < 4) ? rt : PISCES_ALLELE_N;
                const PointCounts c = point_counts(hist, l, a, true, rt, g);
                PiscesCalledAllele r;
                (void)process_point_allele(c, pos, a, true, rt, ref, win_lo, win_hi, P, r);
                const uint4* sp = reinterpret_cast<const uint4*>(&r);
                uint4* dp = reinterpret_cast<uint4*>(&s_rec[l * 4 + kref]);
                dp[0] = sp[0]; dp[1] = sp[1]; dp[2] = sp[2]; dp[3] = sp[3];
                called_mask |= 1u << kref;
            }
        }
    }
    const int mine = __popc(called_mask);
    int n_surv, n_call_total;
    const int excl = wave_exclusive_sum(mine, &n_surv);
    (void)wave_exclusive_sum(n_callable, &n_call_total);
    const int n_loci_called = __popcll(__ballot(mine > 0));
    int base = 0;
    if (l == 0) {
        base = record_count ? (n_surv > 0 ? atomicAdd(record_count, n_surv) : 0) : tile_index * kSlotsPerTile;
        PiscesTileResult tr;
        tr.record_begin = base;
        tr.n_records = n_surv;
        tr.n_candidate_loci = n_loci_called;
        tr.reserved = n_call_total;
        *tile_result = tr;
        if (P.totals) {
            unsigned long long* tt = P.totals + (size_t)(tile_index % kTotalShards) * kTotalStride;
            atomicAdd(&tt[0], (unsigned long long)n_surv);
            atomicAdd(&tt[1], (unsigned long long)n_loci_called);
            atomicAdd(&tt[2], (unsigned long long)n_call_total);
            atomicAdd(&tt[3], 1ull);
        }
    }
    base = __shfl(base, 0, 64);
    int j = 0;
    for (int k = 0; k < 4; k++) {
        if (!(called_mask & (1u << k))) continue;
        const int64_t dst = (int64_t)base + excl + j;
        j++;
        if (dst < capacity) {
            const uint4* sp = reinterpret_cast<const uint4*>(&s_rec[l * 4 + k]);
            uint4* dp = reinterpret_cast<uint4*>(&records[dst]);
            dp[0] = sp[0]; dp[1] = sp[1]; dp[2] = sp[2]; dp[3] = sp[3];
        }
    }
}

__global__ __launch_bounds__(kPipeBlock, 5) void call_tiles_pipelined_kernel(
    const uint32_t* __restrict__ tuples, const PiscesTile* __restrict__ tiles, int32_t n_tiles,
    const uint8_t* __restrict__ ref, int32_t ref_start, int64_t ref_len, PiscesCalledAllele* __restrict__ records,
    int32_t capacity, int32_t* __restrict__ record_count, PiscesTileResult* __restrict__ tile_results, DeviceParams P)
{
    __shared__ __attribute__((aligned(16))) PiscesCalledAllele s_rec[kTile * 4];
    __shared__ int hist[2][kFolded * kTile];

    const int wave = threadIdx.x >> 6;
    const bool streamer = wave < kStreamWaves;
    int t = blockIdx.x;
    if (t >= n_tiles) return;
    for (int i = threadIdx.x; i < 2 * kFolded * kTile; i += kPipeBlock) (&hist[0][0])[i] = 0;
    __syncthreads();
    const uint32_t min_bq = (uint32_t)P.min_bq;
    if (streamer) {
        const PiscesTile t0 = tiles[t];
        int* hb = hist[0];
        const uint32_t n_loci = (uint32_t)t0.n_loci;
        stream_tuples(tuples, t0.tuple_begin, t0.tuple_end, [&](uint32_t v) { accumulate_folded(hb, v, n_loci, min_bq); });
    }
    __syncthreads();
    for (int it = 0; t < n_tiles; t += gridDim.x, it++) {
        const int cur = it & 1;
        const int t_next = t + gridDim.x;
        if (streamer) {
            if (t_next < n_tiles) {
                const PiscesTile tn = tiles[t_next];
                int* hb = hist[cur ^ 1];
                const uint32_t n_loci = (uint32_t)tn.n_loci;
                stream_tuples(tuples, tn.tuple_begin, tn.tuple_end, [&](uint32_t v) { accumulate_folded(hb, v, n_loci, min_bq); });
            }
        } else {
            const PiscesTile tc = tiles[t];
            call_wave(hist[cur], nullptr, tc, t, ref, ref_start, ref_len, records, capacity, record_count, &tile_results[t], P, s_rec);
            int* hb = hist[cur];
            for (int i = threadIdx.x & 63; i < kFolded * kTile; i += 64) hb[i] = 0;
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------
// Anchor-resolved accumulation: LDS [locus][199] (odd stride: consecutive loci hit distinct banks),
// added into counts[(tile*kTile + locus)][6][3][11] — RegionState._alleleCounts layout.
constexpr int kAnchStride = PISCES_COUNTS_PER_LOCUS + 1;

__global__ __launch_bounds__(kBlock) void accumulate_tiles_kernel(
    const uint32_t* __restrict__ tuples, const PiscesTile* __restrict__ tiles, int32_t n_tiles,
    int32_t* __restrict__ counts, int32_t min_bq_)
{
    __shared__ int hist[kTile * kAnchStride];
    const int t = blockIdx.x;
    if (t >= n_tiles) return;
    const PiscesTile tile = tiles[t];
    for (int i = threadIdx.x; i < kTile * kAnchStride; i += kBlock) hist[i] = 0;
    __syncthreads();
    const uint32_t n_loci = (uint32_t)tile.n_loci, min_bq = (uint32_t)min_bq_;
    stream_tuples(tuples, tile.tuple_begin, tile.tuple_end, [&](uint32_t v) {
        uint32_t locus = v & 0x7FFFu;
        uint32_t anchor = (v >> 15) & 0xFu;
        uint32_t dir = (v >> 19) & 3u;
        uint32_t allele = (v >> 21) & 7u;
        uint32_t qual = v >> 24;
        if (allele < 4u && qual < min_bq) allele = 4u;
        if (locus < n_loci && dir < 3u && allele < 6u && anchor < (uint32_t)PISCES_NUM_ANCHORS)
            atomicAdd(&hist[locus * kAnchStride + (allele * 3u + dir) * PISCES_NUM_ANCHORS + anchor], 1);
    });
    __syncthreads();
    int32_t* __restrict__ dst = counts + (int64_t)t * kTile * PISCES_COUNTS_PER_LOCUS;
    const int n = tile.n_loci * PISCES_COUNTS_PER_LOCUS;
    for (int g = threadIdx.x; g < n; g += kBlock) {
        int l = g / PISCES_COUNTS_PER_LOCUS, c = g - l * PISCES_COUNTS_PER_LOCUS;
        int v = hist[l * kAnchStride + c];
        if (v) dst[g] += v;
    }
}

// Call phase fed from the global anchor-resolved counts (+ gapped-MNV reference counts).
__global__ __launch_bounds__(kBlock) void call_counts_kernel(
    const int32_t* __restrict__ counts, const uint32_t* __restrict__ gapped_mnv_ref,
    const PiscesTile* __restrict__ tiles, int32_t n_tiles, const uint8_t* __restrict__ ref, int32_t ref_start,
    int64_t ref_len, PiscesCalledAllele* __restrict__ records, int32_t capacity, int32_t* __restrict__ record_count,
    PiscesTileResult* __restrict__ tile_results, DeviceParams P)
{
    __shared__ int hist[kFolded * kTile];
    __shared__ uint32_t s_gapped[kTile];
    __shared__ int s_wave[4];
    __shared__ int s_base;
    __shared__ uint8_t s_work[kBlock];
    __shared__ uint8_t s_callable[kBlock];
    const int t = blockIdx.x;
    if (t >= n_tiles) return;
    const PiscesTile tile = tiles[t];
    const int32_t* __restrict__ src = counts + (int64_t)t * kTile * PISCES_COUNTS_PER_LOCUS;
    // fold the 11 anchor bins: one (locus, allele*3+dir) cell per thread-iteration
    for (int i = threadIdx.x; i < kFolded * kTile; i += kBlock) {
        int c = i / kTile, l = i - c * kTile;
        int s = 0;
        if (l < tile.n_loci) {
            const int32_t* p = src + (int64_t)l * PISCES_COUNTS_PER_LOCUS + c * PISCES_NUM_ANCHORS;
#pragma unroll
            for (int a = 0; a < PISCES_NUM_ANCHORS; a++) s += p[a];
        }
        hist[i] = s;
    }
    if (threadIdx.x < kTile)
        s_gapped[threadIdx.x] = (gapped_mnv_ref && threadIdx.x < tile.n_loci) ? gapped_mnv_ref[(int64_t)t * kTile + threadIdx.x] : 0u;
    __syncthreads();
    call_phase(hist, s_gapped, tile, ref, ref_start, ref_len, records, capacity, record_count, &tile_results[t], P,
               s_wave, s_work, s_callable, &s_base);
}

}  // namespace pisces

// ==========================================================================================
// Spanning candidates (insertions / deletions found by the host finder): CoverageCalculator.CalculateSpanning
// (lib/Pisces.Calculators/CoverageCalculator.cs:162-321) from the anchor-resolved counts in HBM, then the same
// q-score / strand-bias / filter / genotype chain.  One lane per candidate; these are rare (a handful per block).
// ==========================================================================================
namespace pisces {

struct DevCandidate {
    int32_t position, category, ref_len, alt_len;
    int32_t sup[3];
    int32_t anch[3];
    int32_t first_base, last_base;   // AlleleType of AlternateAllele[1] / [last] (insertions, :180-186)
    int64_t start_idx, end_idx;      // locus index of the start / end point in the counts tensor, -1 = no block (count 0)
    int32_t allele_off;              // ref bytes then alt bytes in the allele pool
    int32_t pad;
};

// AlleleCountHelper.GetAnchorAdjustedAlleleCount (lib/Pisces.Processing/RegionState/AlleleCountHelper.cs:21-85)
// over one [11] row; maxAnchor < 0 = null; symmetric is never set on this path
__device__ inline int anchor_adjusted_count(const int32_t* __restrict__ row, int minAnchor, int maxAnchor, bool fromEnd)
{
    const int wellAnchoredIndex = PISCES_ANCHOR_SIZE, numAnchorIndexes = PISCES_NUM_ANCHORS;
    const int trueMinAnchor = wellAnchoredIndex < minAnchor ? wellAnchoredIndex : minAnchor;
    int initialMaxAnchor = wellAnchoredIndex;
    if (maxAnchor >= 0) {
        if (maxAnchor >= wellAnchoredIndex) initialMaxAnchor = wellAnchoredIndex - 1;
        if (maxAnchor < wellAnchoredIndex) initialMaxAnchor = maxAnchor;
    }
    int tot = 0;
    if (fromEnd) {
        for (int i = trueMinAnchor; i <= initialMaxAnchor; i++) tot += row[numAnchorIndexes - i - 1];
        if (maxAnchor < 0)
            for (int i = 0; i < initialMaxAnchor; i++) tot += row[i];
    } else {
        for (int i = trueMinAnchor; i <= initialMaxAnchor; i++) tot += row[i];
        if (maxAnchor < 0)
            for (int i = initialMaxAnchor + 1; i < numAnchorIndexes; i++) tot += row[i];
    }
    return tot;
}

__device__ inline int get_allele_count(const int32_t* __restrict__ counts, int64_t idx, int allele, int dir, int minAnchor,
                                       int maxAnchor, bool fromEnd)
{
    if (idx < 0) return 0;   // RegionStateManager.GetAlleleCount: no block -> 0 (RegionStateManager.cs:222-226)
    return anchor_adjusted_count(counts + idx * PISCES_COUNTS_PER_LOCUS + (allele * 3 + dir) * PISCES_NUM_ANCHORS, minAnchor,
                                 maxAnchor, fromEnd);
}

// RMxNCalculator.ComputeRMxNLengthForIndel (lib/Pisces.Calculators/RMxNCalculator.cs:50-94); ref[i] = string index i
__device__ inline int rmxn_length_for_indel(int variantPosition, const uint8_t* __restrict__ bases, int length,
                                            const uint8_t* __restrict__ ref, int64_t ref_len, int maxRepeatUnitLength)
{
    int maxRepeatsFound = 0;
    const int lo = length - (maxRepeatUnitLength < length ? maxRepeatUnitLength : length);
    for (int pass = 0; pass < 2; pass++) {       // prefixes, then suffixes
        for (int i = lo; i < length; i++) {
            const int blen = length - i;
            const uint8_t* bookend = pass == 0 ? bases : bases + i;
            auto matches = [&](int64_t at) {
                for (int k = 0; k < blen; k++)
                    if (ref[at + k] != bookend[k]) return false;
                return true;
            };
            int64_t backPeekPosition = variantPosition;
            while (true) {
                const int64_t nb = backPeekPosition - blen;
                if (nb < 0) break;
                if (nb + blen > ref_len || !matches(nb)) break;
                backPeekPosition = nb;
            }
            int repeatCount = 0;
            int64_t currentPosition = backPeekPosition;
            while (true) {
                if (currentPosition + blen > ref_len) break;
                if (!matches(currentPosition)) break;
                repeatCount++;
                currentPosition += blen;
            }
            if (repeatCount > maxRepeatsFound) maxRepeatsFound = repeatCount;
        }
    }
    return maxRepeatsFound;
}

__global__ __launch_bounds__(64) void call_spanning_kernel(
    const DevCandidate* __restrict__ cands, int32_t n, const int32_t* __restrict__ counts, const uint8_t* __restrict__ alleles,
    const uint8_t* __restrict__ ref, int64_t ref_len /* ref[i] = position i+1 */, int32_t expect_stitched,
    PiscesCalledAllele* __restrict__ out, uint8_t* __restrict__ callable_out, DeviceParams P)
{
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    const DevCandidate c = cands[i];
    const int length = c.category == PISCES_CAT_INSERTION ? c.alt_len - 1 : c.ref_len - 1;   // BaseAllele.Length
    const int support = c.sup[0] + c.sup[1] + c.sup[2];
    const int wellAnchored = c.anch[0] + c.anch[1] + c.anch[2];
    const bool presumeAnchoredForExactCov = c.category == PISCES_CAT_INSERTION ? (expect_stitched != 0) : true;   // :31-41
    const bool bePicky = c.category == PISCES_CAT_INSERTION;   // considerAnchorInformation (TrackedAnchorSize 5 > 0), :179

    int startPointCoverage[3] = {0, 0, 0}, endPointCoverage[3] = {0, 0, 0};
    int startUnanch[3] = {0, 0, 0}, endUnanch[3] = {0, 0, 0};
    int confidentLeft = 0, confidentRight = 0, suspiciousLeft = 0, suspiciousRight = 0;
    const int unanchoredSupport = support - wellAnchored;
    const int cca[5] = {PISCES_ALLELE_A, PISCES_ALLELE_C, PISCES_ALLELE_G, PISCES_ALLELE_T, PISCES_ALLELE_DEL};
    for (int d = 0; d < 3; d++) {
        for (int k = 0; k < 5; k++) {
            const int at = cca[k];
            const int minAnchorEnd = (bePicky && at == c.first_base) ? length : 0;
            const int minAnchorStart = (bePicky && at == c.last_base) ? length : 0;
            const int sc = get_allele_count(counts, c.start_idx, at, d, minAnchorStart, -1, false);
            const int ec = get_allele_count(counts, c.end_idx, at, d, minAnchorEnd, -1, true);
            startPointCoverage[d] += sc;
            endPointCoverage[d] += ec;
            confidentLeft += sc;
            confidentRight += ec;
            if (bePicky && unanchoredSupport > 0) {
                if (minAnchorStart > 0) {
                    const int u = get_allele_count(counts, c.start_idx, at, d, 0, minAnchorStart - 1, false);
                    startUnanch[d] += u;
                    suspiciousLeft += u;
                }
                if (minAnchorEnd > 0) {
                    const int u = get_allele_count(counts, c.end_idx, at, d, 0, minAnchorEnd - 1, true);
                    endUnanch[d] += u;
                    suspiciousRight += u;
                }
            }
        }
    }
    if (bePicky) {   // :261-293, float32 arithmetic as written
        const float trulyAnchoredCoverage = (((confidentLeft - suspiciousRight) + (confidentRight - suspiciousLeft)) / 2.0f);
        const float anchoredVariantFreq = trulyAnchoredCoverage <= 0 ? 0.0f : (float)wellAnchored / trulyAnchoredCoverage;
        const int totalSuspicious = suspiciousLeft + suspiciousRight;
        const float unanchoredVariantFreq = totalSuspicious == 0 ? 0.0f : unanchoredSupport / ((float)totalSuspicious);
        float w = anchoredVariantFreq == 0 ? 1.0f : fminf(1.0f, unanchoredVariantFreq / anchoredVariantFreq);
        if (!(w > 0.0f)) w = 0.0f;
        const double weight = w;
        for (int d = 0; d < 3; d++) {
            startPointCoverage[d] += (int)(startUnanch[d] * weight);
            endPointCoverage[d] += (int)(endUnanch[d] * weight);
        }
    }
    // RedistributeStitchedCoverage :324-331
    startPointCoverage[0] += (int)ceilf((float)startPointCoverage[2] / 2);
    startPointCoverage[1] += (int)floorf((float)startPointCoverage[2] / 2);
    endPointCoverage[0] += (int)ceilf((float)endPointCoverage[2] / 2);
    endPointCoverage[1] += (int)floorf((float)endPointCoverage[2] / 2);
    int cov[3] = {0, 0, 0};
    float exactTotalCoverage = 0.0f;
    for (int d = 0; d < 2; d++) {
        const float e = presumeAnchoredForExactCov
                            ? (startPointCoverage[d] + endPointCoverage[d]) / 2.0f
                            : (float)(startPointCoverage[d] < endPointCoverage[d] ? startPointCoverage[d] : endPointCoverage[d]);
        cov[d] = (int)e;
        exactTotalCoverage += e;
    }
    const int total = (int)exactTotalCoverage;
    int refsup = total - support;
    if (refsup < 0) refsup = 0;

    // ProcessVariant (AlleleCaller.cs:208-234)
    int vq = 0;
    SbResult sb = {0.0, 0, 0, 0};
    if (support > 0) {
        if (total != 0) vq = poisson_qscore(support, total, P);
        sb = strand_bias(cov, c.sup, P);
    }
    const float freq = frequency_f(support, total);
    uint32_t filters = 0;   // NumNoCalls stays 0 for spanning alleles -> FractionNoCalls 0
    if (P.low_depth_filter >= 0 && total < P.low_depth_filter) filters |= 1u << PISCES_FILTER_LOW_DEPTH;
    if (P.vq_filter >= 0 && vq < P.vq_filter && total != 0) filters |= 1u << PISCES_FILTER_LOW_VARIANT_QSCORE;
    if (!sb.acceptable || (P.filter_single_strand && !sb.var_both)) filters |= 1u << PISCES_FILTER_STRAND_BIAS;
    if (P.rmxn_max_len >= 0 && !(freq >= P.rmxn_freq_limit)) {   // RMxNCalculator.ShouldFilter :19-38
        const uint8_t* vb = alleles + c.allele_off + (c.category == PISCES_CAT_INSERTION ? c.ref_len + 1 : 1);
        const int c1 = rmxn_length_for_indel(c.position, vb, length, ref, ref_len, P.rmxn_max_len);
        if (c1 >= P.rmxn_min_rep) filters |= 1u << PISCES_FILTER_RMXN;
    }
    if (P.vf_filter >= 0.0f && freq < P.vf_filter) filters |= 1u << PISCES_FILTER_LOW_VARIANT_FREQUENCY;
    if (expect_stitched) {   // AlleleProcessor.cs:64-68
        for (int k = 0; k < c.alt_len; k++)
            if (alleles[c.allele_off + c.ref_len + k] == 'N') filters |= 1u << PISCES_FILTER_STRAND_BIAS;
    }
    bool callable = true;   // IsCallable (AlleleCaller.cs:236-258)
    if (total < P.min_cov && !P.include_ref) callable = false;
    else if (total != 0 && freq < P.min_freq) callable = false;
    else if (vq < P.min_vq) callable = false;

    const int gt = somatic_genotype(false, total, support, refsup, P);
    const int gq = somatic_gq(gt, vq, total, support, P);
    if (P.low_gq_filter >= 0 && (float)gq < (float)P.low_gq_filter) filters |= 1u << PISCES_FILTER_LOW_GENOTYPE_QUALITY;

    PiscesCalledAllele r;
    r.position = c.position;
    r.total_coverage = total;
    r.allele_support = support;
    r.reference_support = refsup;
    r.num_no_calls = 0;
    r.coverage_by_dir[0] = cov[0]; r.coverage_by_dir[1] = cov[1]; r.coverage_by_dir[2] = 0;
    r.support_by_dir[0] = c.sup[0]; r.support_by_dir[1] = c.sup[1]; r.support_by_dir[2] = c.sup[2];
    r.variant_qscore = vq;
    r.strand_bias_score = sb.bias_score;
    r.genotype_qscore = gq;
    r.filter_bits = (uint16_t)filters;
    const int rt = allele_type_of_base(alleles[c.allele_off]);
    r.info = PISCES_INFO_PACK(gt, c.category, rt, PISCES_ALLELE_N, sb.acceptable, sb.var_both, sb.cov_both);
    out[i] = r;
    callable_out[i] = callable ? 1 : 0;
}

}  // namespace pisces
