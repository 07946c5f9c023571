// kernels.hip.h — gfx950 kernels of the pileup-and-likelihood path.
//
//   call_tiles_wave_kernel<NW>  THE HOT KERNEL: one tile (<= 64 loci) per wave, NW waves per workgroup.  Observation tuples stream in
//                            as 16-byte loads; the tuple IS the LDS address of its counter (3 VALU + 1 ds_add per observation, two
//                            [32][64] int32 regions: quality-passing / low-quality bases); the call phase — coverage, Poisson
//                            q-score, strand bias, somatic genotype, filters for the Reference allele and every SNV candidate —
//                            reads memo tables first (DeviceParams::vq_tab / sb_tab / gq_cap, filled by the functions they stand for)
//                            and falls into the full FP64 functions only on a miss.  Counts never leave LDS.
//                            HBM traffic = 4 B/tuple + 1 B/locus + 64 B/record.
//   call_tiles_kernel        the earlier form (one 4-wave workgroup per tile, roles per wave); germline / Window configurations
//   accumulate_tiles_kernel  tuples -> anchor-resolved int32[6][3][11] counts (+ base-quality sums) added to a global tensor
//                            (the IAlleleSource view: RegionState._alleleCounts / _sumOfAlleleBaseQualities, RegionState.cs:57,61).
//   call_counts_kernel       the same call phase fed from that global tensor (+ gapped-MNV reference counts).
//   scan_tile_counts_kernel / gather_records_kernel
//                            optional ordered compaction of the per-tile record slots.
//   call_spanning_kernel     candidates with allele strings (insertions, deletions, MNVs; SNV / Reference alleles of the MNV path and
//                            forced alleles): spanning coverage from the global tensor, then the same q-score / strand-bias /
//                            filter / genotype chain; records are made for alleles that are not callable too (forced alleles).
//   build_*_kernel           the memo tables, once per handle.
//
// Grid = number of tiles / NW (>> 256 CUs for any real interval set; pisces_hip_balanced_tile_loci picks the tile size that gives every
// CU the same number of tiles).  No MFMA: this is a scan + histogram + transcendental epilogue, HBM-bound.
// The read walk, the bucketing and candidate discovery are in stream_kernels.hip.h / finder_kernels.hip.h, BGZF / BAM in
// bgzf_kernels.hip.h / bam_kernels.hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_math.hip.h"
#include "genotype_core.h"

namespace pisces {

constexpr int kBlock = 256;
constexpr int kTile = 64;                 // loci per tile (lane = locus in the call phase)
constexpr int kFolded = 18;               // 6 allele types x 3 directions (anchors folded)
constexpr int kUnroll = 8;                // 16-byte loads in flight per lane
constexpr int kSlotsPerTile = 4 * kTile;  // record slot of (locus, allele rank) = 256 * tile + 4 * locus + rank
#ifndef PISCES_TOTAL_SHARDS
#define PISCES_TOTAL_SHARDS 64
#endif
constexpr int kTotalShards = PISCES_TOTAL_SHARDS;   // running-total shards (one 128-byte line each)
constexpr int kTotalStride = 16;          // in 8-byte words
constexpr int kRefMargin = 32;            // reference bases staged in LDS on each side of a tile (RMxN scan reach)
constexpr int kRefWin = kTile + 2 * kRefMargin;
constexpr int kWaveRows = 32;             // wave-kernel histogram row = the tuple's 5-bit allele:direction field as it is
constexpr int kWaveRow = kTile;           // 64 columns (PISCES_TUPLE_COLUMN): the LDS bank of a counter depends on its column only
constexpr int kWaveRegion = kWaveRows * kWaveRow;   // counters per region; two regions: quality-passing bases, low-quality bases
constexpr int kQLutLds = 128;             // QtoP(q), q < 128, staged in LDS (no global load inside the call phase)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));   // one dwordx4 load

// alphabetical allele order A, C, G, T (the per-locus output order, AlleleCaller.cs:172-176)
// expressed in AlleleType codes A=0, G=1, C=2, T=3
__device__ __forceinline__ int allele_of_rank(int k) { return k == 1 ? 2 : (k == 2 ? 1 : k); }
__device__ __forceinline__ int rank_of_allele(int a) { return a == 1 ? 2 : (a == 2 ? 1 : a); }

__device__ __forceinline__ int allele_type_of_base(uint8_t c)  // AlleleHelper.GetAlleleType, AlleleHelper.cs:13-32
{
    return c == 'A' ? 0 : c == 'G' ? 1 : c == 'C' ? 2 : c == 'T' ? 3 : 4;
}

// One observation into the folded LDS histogram hist[(allele*3+dir)*kTile + locus].
// "qual < minBQ -> N" is RegionStateManager.cs:179-181; deletion tuples carry qual 255.
// Padding tuples (0xFFFFFFFF) decode to allele 7 and fall out of the range test.
__device__ __forceinline__ void accumulate_folded(int* hist, uint32_t t, uint32_t n_loci, uint32_t min_bq)
{
    uint32_t locus = PISCES_TUPLE_LOCUS(t);
    uint32_t dir = PISCES_TUPLE_DIR(t);
    uint32_t allele = PISCES_TUPLE_ALLELE(t);
    uint32_t qual = PISCES_TUPLE_QUAL(t);
    if (allele < 4u && qual < min_bq) allele = 4u;
    if (locus < n_loci && dir < 3u && allele < 6u) atomicAdd(&hist[(allele * 3u + dir) * kTile + locus], 1);
}

// Wave-kernel form.  The tuple is laid out for this (include/pisces_hip.h): bits 2..12 ARE the byte offset of the (allele, direction,
// column) counter in a [32][64] int32 region, so an observation is 3 VALU + 1 DS instruction: one compare-and-select for "qual <
// minBQ" (RegionStateManager.cs:179-181; it picks the second region, whose counts the call phase adds to N, or to the deletion
// count for a deletion tuple, exactly where the remapped allele would have gone), one and-or for the address, ds_add_u32.  No
// branch, no exec masking, no range test: six column bits cannot leave the tile, padding tuples (all ones) and invalid allele /
// direction codes select rows nobody reads.
__device__ __forceinline__ void accumulate_wave(int* hist, uint32_t t, uint32_t min_bq_shifted)
{
    const uint32_t region = (t < min_bq_shifted) ? (uint32_t)(kWaveRegion * sizeof(int)) : 0u;   // qual is the top byte: qual < minBQ <=> t < minBQ << 24
    const uint32_t off = (t & 0x1FFCu) | region;
    atomicAdd(reinterpret_cast<int*>(reinterpret_cast<char*>(hist) + off), 1);
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

// The record goes out as four dwordx4 stores assembled from its fields in registers.  (Copying it as uint4 words through a pointer
// keeps the whole record a stack object: fields stored to scratch, read back, then stored to HBM — a dependent scratch round trip in
// front of every record while the chip is streaming.)
__device__ __forceinline__ void copy_record(PiscesCalledAllele* dst, const PiscesCalledAllele* src)
{
#ifdef PISCES_ABLATE_STORE
    if (src->position != -12345) return;   // development ablation: no record stores
#endif
    const unsigned long long sb = (unsigned long long)__double_as_longlong(src->strand_bias_score);
    uint4* dp = reinterpret_cast<uint4*>(dst);
    dp[0] = make_uint4((uint32_t)src->position, (uint32_t)src->total_coverage, (uint32_t)src->allele_support, (uint32_t)src->reference_support);
    dp[1] = make_uint4((uint32_t)src->num_no_calls, (uint32_t)src->coverage_by_dir[0], (uint32_t)src->coverage_by_dir[1], (uint32_t)src->coverage_by_dir[2]);
    dp[2] = make_uint4((uint32_t)src->support_by_dir[0], (uint32_t)src->support_by_dir[1], (uint32_t)src->support_by_dir[2], (uint32_t)src->variant_qscore);
    dp[3] = make_uint4((uint32_t)sb, (uint32_t)(sb >> 32), (uint32_t)(uint16_t)src->genotype_qscore | ((uint32_t)(uint16_t)src->noise_level << 16), (uint32_t)src->filter_bits | ((uint32_t)src->info << 16));
}

struct PointCounts {
    int cov[3], sup[3];
    int total, nocalls, refsup, support;
};

// LDS layouts of the folded histogram: count of (allele a, direction d) at locus l
struct HistBlock {   // call_tiles_kernel / call_counts_kernel: 18 rows of kTile
    static __device__ __forceinline__ int idx(int a, int d, int l) { return (a * 3 + d) * kTile + l; }
};
struct HistWave {    // call_tiles_wave_kernel: counter of (allele a, direction d) at locus l in the quality-passing region
    static __device__ __forceinline__ int idx(int a, int d, int l) { return (a * 4 + d) * kWaveRow + (int)PISCES_TUPLE_COLUMN(l, d); }
};

struct HistLinear {  // call_store_tiles_kernel (store_kernels.hip.h): the same rows, column = locus (lane = locus when a read is added: no two lanes of
                     // an instruction share a bank)
    static __device__ __forceinline__ int idx(int a, int d, int l) { return (a * 4 + d) * kWaveRow + l; }
};

struct LocusCounts { int h[6][3]; };

template <typename H = HistBlock>
__device__ __forceinline__ LocusCounts load_counts(const int* hist, int l)
{
    LocusCounts lc;
#pragma unroll
    for (int k = 0; k < kFolded; k++) lc.h[k / 3][k % 3] = hist[H::idx(k / 3, k % 3, l)];
    return lc;
}

// the wave kernel's two regions folded into the reference's counts: a low-quality A/C/G/T/N base is an N, a deletion stays one
template <typename HistWave = pisces::HistWave>
__device__ __forceinline__ LocusCounts load_counts_wave(const int* hist, int l)
{
    LocusCounts lc;
#pragma unroll
    for (int d = 0; d < 3; d++) {
        int low = 0;
#pragma unroll
        for (int a = 0; a < 5; a++) low += hist[kWaveRegion + HistWave::idx(a, d, l)];
#pragma unroll
        for (int a = 0; a < 4; a++) lc.h[a][d] = hist[HistWave::idx(a, d, l)];
        lc.h[PISCES_ALLELE_N][d] = hist[HistWave::idx(PISCES_ALLELE_N, d, l)] + low;
        lc.h[PISCES_ALLELE_DEL][d] = hist[HistWave::idx(PISCES_ALLELE_DEL, d, l)] + hist[kWaveRegion + HistWave::idx(PISCES_ALLELE_DEL, d, l)];
    }
    return lc;
}

// CoverageCalculator.CalculateSinglePoint (CoverageCalculator.cs:49-98) from the folded counts of one locus
__device__ __forceinline__ PointCounts point_counts_of(const LocusCounts& lc, int allele, bool isRef, int refType, int gapped)
{
    PointCounts c;
    const int (&h)[6][3] = lc.h;
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

template <typename H = HistBlock>
__device__ __forceinline__ PointCounts point_counts(const int* hist, int l, int allele, bool isRef, int refType, int gapped)
{
    return point_counts_of(load_counts<H>(hist, l), allele, isRef, refType, gapped);
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
struct GqPre { int32_t idx, val; };   // memo index of the genotype q-score (somatic_gq_index) and P.gq_cap[idx], fetched ahead by the caller
template <bool kMemo = false>   // kMemo: table-first forms only; false = a table missed, the record is not made (the caller takes the long way)
__device__ inline bool finish_allele(const PointCounts& c, int pos, int a, bool isRef, int rt, int vq, const SbResult& sb,
                                     const uint8_t* __restrict__ ref, int64_t win_lo, int64_t win_hi, const DeviceParams& P,
                                     PiscesCalledAllele& r, const uint8_t* s_refwin, int s_refidx,
                                     const GqTail* pre_tail = nullptr, const GqPre* pre_gq = nullptr,
                                     const int32_t* window_level_ = nullptr /* NoiseModel.Window: the allele's own noise level */)
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
        // base letter of an allele code (A0 G1 C2 T3) from a packed constant: a local array indexed at run time would live in scratch
        auto base_of = [](int code) { return (uint8_t)((0x54434741u >> (8 * code)) & 0xFFu); };
        if (rt < 4 && a < 4) {
            // the reference window of the tile sits in LDS (global byte loads are dependent multi-microsecond
            // round trips while the chip is streaming); fall back to HBM only for an RMxN reach beyond the margin
            const bool hit = (s_refwin && P.rmxn_min_rep <= kRefMargin)
                                 ? rmxn_should_filter_snv_lds(s_refwin, kRefWin, s_refidx, base_of(rt), base_of(a), freq, P)
                                 : rmxn_should_filter_snv(ref, win_lo, win_hi, pos, base_of(rt), base_of(a), freq, P);
            if (hit) filters |= 1u << PISCES_FILTER_RMXN;
        }
        if (P.vf_filter >= 0.0f && freq < P.vf_filter) filters |= 1u << PISCES_FILTER_LOW_VARIANT_FREQUENCY;
    }
    const int gt = somatic_genotype(isRef, c.total, c.support, c.refsup, P);
    int gq;
    if (kMemo) {
        GqPre g;
        if (pre_gq) g = *pre_gq;
        else {
            g.idx = somatic_gq_index(gt, c.total, c.support, P);
            g.val = (g.idx >= 0 && P.gq_cap) ? (int32_t)P.gq_cap[g.idx] : 0;
        }
        if (!somatic_gq_try(gt, vq, c.total, c.support, P, g.idx, g.val, gq)) return false;
    } else {
        gq = pre_tail ? somatic_gq_finish(gt, vq, c.total, *pre_tail, P) : somatic_gq(gt, vq, c.total, c.support, P);
    }
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
    r.genotype_qscore = (int16_t)gq;
    // NoiseLevelApplied is assigned where the q-score is computed (VariantQualityCalculator.cs:13): alleles with support only
    r.noise_level = c.support > 0 ? (window_level_ ? noise_level_field(*window_level_) : (int16_t)P.noise_level) : (int16_t)0;
    r.filter_bits = (uint16_t)filters;
    r.info = PISCES_INFO_PACK(gt, isRef ? PISCES_CAT_REFERENCE : PISCES_CAT_SNV, rt, isRef ? rt : a, sb.acceptable, sb.var_both,
                              sb.cov_both);
    return true;
}

// One lane, one allele, start to finish. Returns false (record untouched) when the reference would drop the allele.
// kFinishAnyway: the record is made even then (a forced allele is reported whether it is callable or not, AlleleCaller.cs:109-131).
template <bool kDiploidOk = false, bool kFinishAnyway = false>
__device__ inline bool process_point_allele(const PointCounts& c, int pos, int a, bool isRef, int rt,
                                             const uint8_t* __restrict__ ref, int64_t win_lo, int64_t win_hi,
                                             const DeviceParams& P, PiscesCalledAllele& r,
                                             const uint8_t* s_refwin = nullptr, int s_refidx = 0, const GqTail* pre_tail = nullptr,
                                             const int32_t* window_level_ = nullptr /* NoiseModel.Window: the allele's own noise level (kNoLevel = q-score 0) */)
{
    bool callable = true;
    if (!isRef && !variant_passes_frequency(c, P)) {
        if (!kFinishAnyway) return false;
        callable = false;
    }
    int vq = 0;
    if (window_level_) {
        const double werr = window_err_of_level(*window_level_, P);
        if (c.support > 0 && c.total != 0 && werr >= 0.0) vq = poisson_qscore_e(c.support, c.total, werr, P);
    } else
#if !(defined(PISCES_ABLATE_MATH) && (PISCES_ABLATE_MATH == 5 || PISCES_ABLATE_MATH == 9))
    if (c.support > 0 && c.total != 0) vq = poisson_qscore(c.support, c.total, P);   // VariantQualityCalculator.Compute :11-24
#endif
    if (!isRef && vq < P.min_vq) {
        if (!kFinishAnyway) return false;
        callable = false;
    }
    SbResult sb = {0.0, 0, 0, 0};
#if !(defined(PISCES_ABLATE_MATH) && (PISCES_ABLATE_MATH == 4 || PISCES_ABLATE_MATH == 9))
    if (c.support > 0) sb = strand_bias<kDiploidOk>(c.cov, c.sup, P);                // StrandBiasCalculator.Compute :10-15
#endif
    finish_allele(c, pos, a, isRef, rt, vq, sb, ref, win_lo, win_hi, P, r, s_refwin, s_refidx, pre_tail, nullptr, window_level_);
    return callable;
}

// ------------------------------------------------------------------------------------------
// The call phase of one tile whose folded counts sit in LDS, lane = locus, one role per wave (wave-uniform code):
//   wave 0  Reference candidate of the locus (RegionState.GetAllCandidates, RegionState.cs:414-447), start to finish
//   wave 1  variant candidates: variant q-score, then filters / genotype / record for the callable ones
//   wave 2  variant candidates: the three strand-bias statistics
//   wave 3  retires at once
// SNV candidates are the quality-passing bases that differ from the reference base (CandidateVariantFinder.cs:97-160
// with callMNVs off; SupportByDirection = allele count by direction).  A called SNV is ~5 k dependent FP64
// instructions; its q-score and strand-bias tails are independent, so they run on two SIMDs and meet in LDS.
// Waves leave as soon as their role is done: their wave slots start the next workgroup's stream while the
// last one or two waves finish the FP64 tail (measured timeline in DESIGN.md section 4).
//
// Output: the record of (locus l, allele rank k) goes straight to slot record_begin + 4*l + k; the tile directory
// carries a validity bit per slot.  Ascending valid slots are (position, allele) order, and the Reference slot of a
// locus is simply not marked valid when a variant is called there (AlleleCaller.cs:146-147).  No staging, no
// prefix sums, no allocation atomics, placement independent of scheduling.
struct VarScratch {
    int vq[kTile * 4];
    double ov_var[kTile * 4], fw_var[kTile * 4], fw_fp[kTile * 4], rv_var[kTile * 4], rv_fp[kTile * 4];
};

__device__ inline void call_roles(const int* hist, const uint32_t* gapped /* LDS[kTile] or nullptr */, const int32_t* s_lvl /* LDS[kTile]: NoiseModel.Window noise level of the locus' point alleles, or nullptr */, const PiscesTile& tile,
                                  int tile_index, const uint8_t* __restrict__ ref, int32_t ref_start, int64_t ref_len,
                                  PiscesCalledAllele* __restrict__ records, PiscesTileResult* __restrict__ tile_result,
                                  const DeviceParams& P, uint8_t* s_mask /* LDS[kTile] */,
                                  const uint8_t* s_refwin /* LDS[kRefWin], 0 = outside the reference */, VarScratch* vs)
{
    const int wave = threadIdx.x >> 6;
    if (wave == 3) return;
    const int l = threadIdx.x & 63;
    const int pos = tile.start_position + l;
    const uint8_t refb = s_refwin[kRefMargin + l];
    const bool in_ref = l < tile.n_loci && refb != 0;
    const int rt = in_ref ? allele_type_of_base(refb) : PISCES_ALLELE_N;
    const int64_t win_lo = (int64_t)ref_start - 1, win_hi = win_lo + ref_len;
    const int g = gapped ? (int)gapped[l] : 0;
    PiscesCalledAllele* const slots = records + (int64_t)tile_index * kSlotsPerTile + l * 4;

    // Which variant candidates of this locus survive the integer / float32 half of IsCallable (coverage, frequency)?
    // Every wave evaluates the same data, so the "no variant work in this tile" decision is wave-uniform across the
    // workgroup: waves 1 and 2 then retire at once and wave 0 never meets a barrier.
    uint32_t pass_mask = 0;
    if (in_ref && rt < 4 && !P.refs_only && !locus_is_dirty(P, pos)) {
        for (int k = 0; k < 4; k++) {
            const int a = allele_of_rank(k);
            if (a == rt) continue;
            if (hist[(a * 3 + 0) * kTile + l] + hist[(a * 3 + 1) * kTile + l] + hist[(a * 3 + 2) * kTile + l] == 0) continue;
            const PointCounts c = point_counts(hist, l, a, false, rt, g);
            if (variant_passes_frequency(c, P)) pass_mask |= 1u << k;
        }
    }
    const bool tile_has_variants = __ballot(pass_mask != 0) != 0ull;
    if (!tile_has_variants && wave != 0) return;

    bool ref_emitted = false;
    int ref_rank = 0;
    if (wave == 0) {
        if (in_ref && P.include_ref) {
            int all = 0;
#pragma unroll
            for (int c = 0; c < kFolded; c++) all += hist[c * kTile + l];
            if (P.emit_zero_cov || all > 0) {
                const int a = (rt < 4) ? rt : PISCES_ALLELE_N;
                ref_rank = (rt < 4) ? rank_of_allele(rt) : 0;
                const PointCounts c = point_counts(hist, l, a, true, rt, g);
                PiscesCalledAllele rec;
                (void)process_point_allele<true>(c, pos, a, true, rt, ref, win_lo, win_hi, P, rec, s_refwin, kRefMargin + l, nullptr, s_lvl ? &s_lvl[l] : nullptr);
                // written now; it only counts if no variant turns out callable at this locus
                copy_record(&slots[ref_rank], &rec);
                ref_emitted = true;
            }
        }
    } else {
        for (int k = 0; k < 4; k++) {
            if (!(pass_mask & (1u << k))) continue;
            const int a = allele_of_rank(k);
            const PointCounts c = point_counts(hist, l, a, false, rt, g);
            const int slot = l * 4 + k;
            if (wave == 1) {
#if defined(PISCES_ABLATE_MATH) && (PISCES_ABLATE_MATH == 9 || PISCES_ABLATE_MATH == 6)
                vs->vq[slot] = 100;
#else
                vs->vq[slot] = !(c.support > 0 && c.total != 0) ? 0 : !s_lvl ? poisson_qscore(c.support, c.total, P)
                               : s_lvl[l] != kNoLevel ? poisson_qscore_e(c.support, c.total, window_err_of_level(s_lvl[l], P), P) : 0;
#endif
#if defined(PISCES_ABLATE_MATH) && (PISCES_ABLATE_MATH == 9 || PISCES_ABLATE_MATH == 7)
            } else if (c.support < 0) {
#else
            } else if (c.support > 0) {
#endif
                const SbStats ov = sb_stats_of<true>(0, c.cov, c.sup, P), fw = sb_stats_of<true>(1, c.cov, c.sup, P),
                              rv = sb_stats_of<true>(2, c.cov, c.sup, P);
                vs->ov_var[slot] = ov.var_gt_zero;
                vs->fw_var[slot] = fw.var_gt_zero; vs->fw_fp[slot] = fw.false_pos;
                vs->rv_var[slot] = rv.var_gt_zero; vs->rv_fp[slot] = rv.false_pos;
            }
        }
    }
    uint32_t vmask = 0;
    if (tile_has_variants) {
        __syncthreads();   // waves 0, 1, 2
        if (wave == 2) return;
        if (wave == 1) {
            uint32_t mask = 0;
            for (int k = 0; k < 4; k++) {
                if (!(pass_mask & (1u << k))) continue;
                const int a = allele_of_rank(k);
                const int slot = l * 4 + k;
                const int vq = vs->vq[slot];
                if (vq < P.min_vq) continue;                                   // IsCallable, last test
                const PointCounts c = point_counts(hist, l, a, false, rt, g);
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
                finish_allele(c, pos, a, false, rt, vq, sb, ref, win_lo, win_hi, P, r, s_refwin, kRefMargin + l, nullptr, nullptr, s_lvl ? &s_lvl[l] : nullptr);
                copy_record(&slots[k], &r);
                mask |= 1u << k;
            }
            s_mask[l] = (uint8_t)mask;
        }
        __syncthreads();   // waves 0 and 1
        if (wave == 1) return;
        vmask = s_mask[l];
    }

    // wave 0: validity bits, counts, tile directory
    const uint32_t valid = vmask ? vmask : (ref_emitted ? (1u << ref_rank) : 0u);
    const int n_callable = __popc(vmask) + (ref_emitted ? 1 : 0);   // IsCallable is always true for a Reference allele
    int n_surv = __popc(valid), n_call_total = n_callable;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        n_surv += __shfl_xor(n_surv, d, 64);
        n_call_total += __shfl_xor(n_call_total, d, 64);
    }
    const int n_loci_called = __popcll(__ballot(valid != 0));
    // nibble of lane l -> bit 4*l.. of the 256-bit mask: 8 lanes per 32-bit word
    uint32_t word = valid << ((l & 7) * 4);
    word |= __shfl_xor(word, 1, 64);
    word |= __shfl_xor(word, 2, 64);
    word |= __shfl_xor(word, 4, 64);
    if ((l & 7) == 0) tile_result->valid[l >> 3] = word;
    if (l == 0) {
        tile_result->record_begin = tile_index * kSlotsPerTile;
        tile_result->n_records = n_surv;
        tile_result->n_candidate_loci = n_loci_called;
        tile_result->n_called = n_call_total;   // IAlleleCaller.TotalNumCalled contribution
        if (P.totals) {
            // running totals, sharded over kTotalShards cache lines so the adds do not serialize on one L2 line
            unsigned long long* tt = P.totals + (size_t)(tile_index % kTotalShards) * kTotalStride;
            atomicAdd(&tt[0], (unsigned long long)n_surv);
            atomicAdd(&tt[1], (unsigned long long)n_loci_called);
            atomicAdd(&tt[2], (unsigned long long)n_call_total);
            atomicAdd(&tt[3], 1ull);
        }
    }
}

// ------------------------------------------------------------------------------------------
// The hot kernel: stream -> LDS histogram -> call_roles.
__global__ __launch_bounds__(kBlock, 4) void call_tiles_kernel(
    const uint32_t* __restrict__ tuples, const PiscesTile* __restrict__ tiles, int32_t n_tiles,
    const uint8_t* __restrict__ ref, int32_t ref_start, int64_t ref_len, PiscesCalledAllele* __restrict__ records,
    PiscesTileResult* __restrict__ tile_results, DeviceParams P)
{
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
        // the tile's reference bases (+ margin for the RMxN scan) into LDS now, under the stream: a global byte load
        // inside the call phase is a dependent multi-microsecond round trip while the chip is streaming
        const int64_t ri = (int64_t)tile.start_position - kRefMargin + threadIdx.x - ref_start;
        s_refwin[threadIdx.x] = (ri >= 0 && ri < ref_len) ? ref[ri] : (uint8_t)0;
    } else if (threadIdx.x >= 128 && threadIdx.x < 128 + kQLutLds) {
        const int q = threadIdx.x - 128;
        s_qlut[q] = (P.q_to_p_lut && q < P.q_to_p_n) ? P.q_to_p_lut[q] : q_to_p((double)q);
    }
    P.q_to_p_lut = s_qlut;   // generic pointer to LDS from here on
    P.q_to_p_n = kQLutLds;
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
#if defined(PISCES_ABLATE) && PISCES_ABLATE >= 1
    // development ablation: no call phase
    if (threadIdx.x == 0) { tile_results[t].record_begin = 0; tile_results[t].n_records = hist[5]; }
#else
#ifdef PISCES_TIMING
    const long long tc1 = wall_clock64();
#endif
    call_roles(hist, nullptr, nullptr, tile, t, ref, ref_start, ref_len, records, &tile_results[t], P, s_mask, s_refwin, &s_var);
#ifdef PISCES_TIMING
    if (threadIdx.x == 0) {   // development instrumentation: chip-global clock stamps in the tile directory
        const long long tc2 = wall_clock64();
        tile_results[t].record_begin = (int)(tc0 & 0x3FFFFFFF);      // start (10 ns ticks)
        tile_results[t].n_records = (int)(tc1 & 0x3FFFFFFF);         // stream end
        tile_results[t].n_candidate_loci = (int)(tc2 & 0x3FFFFFFF);  // call end
    }
#endif
#endif
}

// ------------------------------------------------------------------------------------------
// Wave-per-tile form of the hot kernel: one wave64 owns one tile from its first tuple to its directory entry.
//
// Why: a tile is a fixed-latency chain (stream ~130 KB, then a call phase of dependent FP64 work).  With four waves
// per tile only 1024 tiles are resident (128 VGPRs), a BASELINE-config-2 launch (1563 tiles) runs as two lock-step
// rounds and exposes two call phases (measured timeline).  One wave per tile makes every tile of
// such a launch resident at once (4096 wave slots), needs no workgroup barrier, and lets larger launches interleave
// freely.  The call phase is arranged so that the single wave never runs the same long function twice in a row:
//   1. Reference candidate of every locus (lane = locus).  Its q-score and strand-bias tails are the provable
//      early-outs, its genotype-quality tail comes from the handle's memo table (DeviceParams::gq_tail).
//   2. variant q-scores, lane = locus, one converged pass per allele rank that has a candidate in the tile.
//   3. strand bias of the callable variants, lane = (variant, statistic): the three Poisson tails of a variant are
//      independent, so they run side by side on three lanes instead of one after the other.
//   4. filters / genotype / record of the callable variants, lane = locus again.
#ifndef PISCES_WAVE_UNROLL
#define PISCES_WAVE_UNROLL 8
#endif
#ifndef PISCES_WAVE_OCC
#define PISCES_WAVE_OCC 3    // waves per SIMD the one-wave-per-tile form is compiled for (168 VGPRs)
#endif
#ifndef PISCES_WAVE2_OCC
#define PISCES_WAVE2_OCC 4   // two waves per tile: 8 tiles per CU resident, 2048 on the chip
#endif
constexpr int kWaveUnroll = PISCES_WAVE_UNROLL;   // 16-byte loads per lane per batch; two batches are in flight

// One wave streams tuples[begin, end) through op(): two register batches of kWaveUnroll dwordx4 loads per lane in
// ping-pong, so that the loads of the next batch are in flight while the current one is decoded (a lone wave has no
// neighbour to hide its decode time behind).
template <int NW, typename Op, typename Pre>
__device__ __forceinline__ void stream_tuples_wave(const uint32_t* __restrict__ tuples, int64_t begin, int64_t end, int lane, int wid,
                                                   Op op, Pre pre)
{
    int64_t abegin = (begin + 3) & ~(int64_t)3;
    int64_t aend = end & ~(int64_t)3;
    if (abegin > aend) { abegin = end; aend = end; }
    const int tid = wid * 64 + lane;
    const u32x4* __restrict__ p4 = reinterpret_cast<const u32x4*>(tuples + abegin);
    const uint32_t n4 = (uint32_t)((aend - abegin) >> 2);   // a tile's segment is < 2^32 dwordx4
    constexpr uint32_t kBatch = 64u * kWaveUnroll;
    // wave-uniform loop control in scalar registers: a divergent-looking `break` would put the loop exit on the latch
    // path, and the waitcnt pass would then drain the batch in flight at every loop header
    const uint32_t nb = (uint32_t)__builtin_amdgcn_readfirstlane((int)((n4 + kBatch - 1) / kBatch));   // batches wid, wid + NW, ...
    u32x4 cur[kWaveUnroll], nxt[kWaveUnroll];
    // every load is unconditional (index clamped into the segment) so that the batches stay straight-line code and
    // the waits are counted, not drained; only the last, partial batch masks its out-of-range lanes afterwards
    auto load = [&](u32x4* v, uint32_t b) {
#pragma unroll
        for (int u = 0; u < kWaveUnroll; u++) {
            const uint32_t j = min(b * kBatch + (uint32_t)u * 64u + (uint32_t)lane, n4 - 1u);
            // streamed exactly once: non-temporal so the tuples do not evict the reference / records from L2
            v[u] = __builtin_nontemporal_load(&p4[j]);
        }
    };
    auto consume = [&](const u32x4* v) {
#pragma unroll
        for (int u = 0; u < kWaveUnroll; u++) { op(v[u].x); op(v[u].y); op(v[u].z); op(v[u].w); }
    };
    // the first batch is requested before pre() (the workgroup's LDS set-up and its barrier): it is in flight while the histogram
    // is cleared and the reference window arrives, instead of one memory latency after them
    // (single-round launches, NW = 2: 51.5 -> 50.4 us at config 2; the one-wave form of multi-round launches shows no difference
    // beyond run-to-run noise and keeps the set-up first)
    const bool streams = (uint32_t)wid < nb;
    if (NW == 2) {
        if (streams) load(cur, (uint32_t)__builtin_amdgcn_readfirstlane(wid));
        __builtin_amdgcn_sched_barrier(0);
        pre();
    } else {
        pre();
        if (streams) load(cur, (uint32_t)__builtin_amdgcn_readfirstlane(wid));
    }
    for (int64_t i = begin + tid; i < abegin; i += 64 * NW) op(tuples[i]);
    if (streams) {
        uint32_t b = (uint32_t)__builtin_amdgcn_readfirstlane(wid);
        const uint32_t m = (nb - b + NW - 1) / NW;   // this wave's batches
        // explicit ping-pong over pairs of batches with a trip count known up front: no register copy between the
        // halves (it would wait for the batch in flight) and no mid-loop exit (its latch path would make the waitcnt
        // pass drain the batch in flight at every loop header)
        for (uint32_t p = (m - 1) / 2; p > 0; p--) {
            load(nxt, b + NW);
            __builtin_amdgcn_sched_barrier(0);   // keep the next batch's loads ahead of this batch's decode
            consume(cur);
            __builtin_amdgcn_sched_barrier(0);
            load(cur, b + 2 * NW);
            __builtin_amdgcn_sched_barrier(0);
            consume(nxt);
            __builtin_amdgcn_sched_barrier(0);
            b += 2 * NW;
        }
        if (((m - 1) & 1u) != 0) {
            load(nxt, b + NW);
            __builtin_amdgcn_sched_barrier(0);
            consume(cur);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < kWaveUnroll; u++) cur[u] = nxt[u];
            b += NW;
        }
        if (b == nb - 1 && (n4 % kBatch)) {
#pragma unroll
            for (int u = 0; u < kWaveUnroll; u++)
                if (b * kBatch + (uint32_t)u * 64u + (uint32_t)lane >= n4) cur[u] = (u32x4){~0u, ~0u, ~0u, ~0u};
        }
        consume(cur);
    }
    for (int64_t i = aend + tid; i < end; i += 64 * NW) op(tuples[i]);
}

// LDS traffic of ONE wave is processed in order; the fence keeps the compiler from moving it and drains lgkmcnt
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __builtin_amdgcn_wave_barrier();
}

// Cold path of the wave kernel: alleles whose count or coverage lies beyond the memo tables go through the evaluations the tables
// were filled with (out-of-line leaves, device_math.hip.h), counts re-read from LDS: it runs after the fast pass, when almost nothing
// is live.  slow: bit 4 = the Reference candidate, bits 0..3 = variant ranks; returns the variant ranks that were called.
template <typename H = HistWave>
__device__ __forceinline__ uint32_t slow_alleles(const int* hist, const uint8_t* s_refwin, uint32_t slow, int l, int pos, int rt, int ref_rank,
                                                 const uint8_t* __restrict__ ref, int64_t win_lo, int64_t win_hi, PiscesCalledAllele* slots,
                                                 const DeviceParams& P)
{
    uint32_t called = 0;
    const int ref_a = (rt < 4) ? rt : PISCES_ALLELE_N;
#pragma unroll 1
    for (int k = 0; k < 5; k++) {
        if (!((slow >> k) & 1u)) continue;
        const bool isRef = k == 4;
        const int a = isRef ? ref_a : allele_of_rank(k);
        const LocusCounts lc = load_counts_wave<H>(hist, l);
        const PointCounts c = point_counts_of(lc, a, isRef, rt, 0);
        int vq = 0;
        if (c.support > 0 && c.total != 0) vq = poisson_qscore_out_of_line(c.support, c.total, P.err_q, P.max_vq, P.ln10);
        if (!isRef && vq < P.min_vq) continue;
        SbResult sb = {0.0, 0, 0, 0};
        if (c.support > 0) sb = strand_bias_out_of_line(c.cov, c.sup, P);
        PiscesCalledAllele r;
        (void)finish_allele<false>(c, pos, a, isRef, rt, vq, sb, ref, win_lo, win_hi, P, r, s_refwin, kRefMargin + l);
        copy_record(&slots[isRef ? ref_rank : k], &r);
        if (!isRef) called |= 1u << k;
    }
    return called;
}

// The call phase of one tile whose histogram sits in LDS: lane = locus.  NW = 2: the two waves of the tile's workgroup share it (wave 0
// makes the Reference records and the tile directory, wave 1 the variant records; they meet once in LDS).  NW = 1: one wave does all of it.
template <int NW, typename H = HistWave>
__device__ __forceinline__ void call_phase_wave(const int* hist, const uint8_t* s_refwin, uint8_t* s_vmask, const PiscesTile& tile, const int t,
                                                const int l, const int wid, const uint8_t* __restrict__ ref, const int32_t ref_start,
                                                const int64_t ref_len, PiscesCalledAllele* __restrict__ records,
                                                PiscesTileResult* __restrict__ tile_results, const DeviceParams& P
#ifdef PISCES_TIMING
                                                , const long long tc0, const long long tc1
#endif
                                                )
{
    // ---- call phase: lane = locus; with two waves per tile wave 0 makes the Reference records and the tile directory, wave 1 the
    // variant records, and they meet once in LDS (a variant called at a locus drops its Reference record, AlleleCaller.cs:146-147).
    // Every FP64 tail comes from the handle's memo tables (DeviceParams), keyed by the integer counts: no incomplete gamma,
    // no logarithm and no division chain runs here unless a count lies beyond the tables (out-of-line evaluation then). ----
    const int pos = tile.start_position + l;
    const uint8_t refb = s_refwin[kRefMargin + l];
    const bool in_ref = l < tile.n_loci && refb != 0;
    const int rt = in_ref ? allele_type_of_base(refb) : PISCES_ALLELE_N;
    const int64_t win_lo = (int64_t)ref_start - 1, win_hi = win_lo + ref_len;
    auto slot_of = [&](int k) { return records + ((int64_t)t * kSlotsPerTile + l * 4 + k); };
    const bool dirty = locus_is_dirty(P, pos);   // (one dword per 32 loci, requested before the counts are read)
    const LocusCounts lc = load_counts_wave<H>(hist, l);
    const bool ref_wave = wid == 0, var_wave = wid == NW - 1;
    if (P.folded_out && ref_wave && l < tile.n_loci) {   // the locus' 18 counts for the candidate kernel of this flush
        const unsigned rel = (unsigned)(pos - P.folded_first);
        if (rel < (unsigned)P.folded_n) {
            int32_t* const dst = P.folded_out + (size_t)rel * PISCES_FOLDED_PER_LOCUS;
#pragma unroll
            for (int a = 0; a < 6; a++)
#pragma unroll
                for (int d = 0; d < 3; d++) dst[a * 3 + d] = lc.h[a][d];
        }
    }

    // One allele through the table-first forms; false = a table missed (the record is not made).  With the handle's tables in place
    // (always, for the configurations routed here) every entry the allele can need is requested up front: one round trip.
    const bool tabs = tables_complete(P);
    auto fast_allele = [&](const PointCounts& c, int a, bool isRef, PiscesCalledAllele& rec) -> bool {
        int vq = 0;
        double sb_score = 0.0;
        int sb_ok = 0, sb_var = 0, sb_cov = 0;
        if (!tabs) return false;   // (a handle without tables: everything takes the long way)
        const AlleleTables tb = request_allele_tables(isRef, c.support, c.total, c.refsup, c.cov, c.sup, P);
        if (c.support > 0 && c.total != 0 && !poisson_qscore_try(c.support, c.total, P, tb, vq)) return false;   // VariantQualityCalculator.Compute :11-24
        if (!isRef && vq < P.min_vq) return true;                                                                 // IsCallable, last test: not callable, nothing to make
        if (c.support > 0 && !strand_bias_try(c.cov, c.sup, P, tb, sb_score, sb_ok, sb_var, sb_cov)) return false;   // StrandBiasCalculator.Compute :10-15
        const GqPre g = {tb.gq_idx, tb.gq_cap};
        const SbResult sb = {sb_score, sb_ok, sb_var, sb_cov};
        if (!finish_allele<true>(c, pos, a, isRef, rt, vq, sb, ref, win_lo, win_hi, P, rec, s_refwin, kRefMargin + l, nullptr, &g)) return false;
        return true;
    };
    bool ref_emitted = false;
    int ref_rank = 0;
    uint32_t slow = 0;    // alleles of this lane that need the long evaluation: bit 4 = Reference, bits 0..3 = variant ranks
    uint32_t vmask = 0;
    const int ref_a = (rt < 4) ? rt : PISCES_ALLELE_N;
    if (ref_wave) {
        // Reference candidate (RegionState.GetAllCandidates, RegionState.cs:414-447); written now, it only counts if no variant turns
        // out callable at the locus
        const PointCounts c = point_counts_of(lc, ref_a, true, rt, 0);
        if (in_ref && P.include_ref && !P.variants_only && (P.emit_zero_cov || c.total + c.nocalls > 0)) {
            ref_rank = (rt < 4) ? rank_of_allele(rt) : 0;
            ref_emitted = true;
            PiscesCalledAllele rec;
            if (fast_allele(c, ref_a, true, rec)) copy_record(slot_of(ref_rank), &rec);
            else slow |= 16u;
        }
    }
#ifdef PISCES_TIMING
    const long long tcA = wall_clock64();
#endif
    if (var_wave && in_ref && rt < 4 && !P.refs_only && !dirty) {
        // SNV candidates: quality-passing bases that differ from the reference base (CandidateVariantFinder.cs:97-160 with MNV
        // calling off); IsCallable (AlleleCaller.cs:236-258) in its own order: coverage, frequency, q-score
#pragma unroll 1
        for (int k = 0; k < 4; k++) {
            const int a = allele_of_rank(k);
            if (a == rt) continue;
            const PointCounts c = point_counts_of(lc, a, false, rt, 0);   // (selects over the unrolled allele loop: no run-time index into the counts)
            if (c.support == 0) continue;
            if (!variant_passes_frequency(c, P)) continue;
            PiscesCalledAllele r;
            r.position = 0;
            if (!fast_allele(c, a, false, r)) { slow |= 1u << k; continue; }
            if (r.position == 0) continue;   // not callable
            copy_record(slot_of(k), &r);
            vmask |= 1u << k;
        }
    }
    if (__builtin_expect(__ballot(slow != 0) != 0ull, 0))
        vmask |= slow_alleles<H>(hist, s_refwin, slow, l, pos, rt, ref_rank, ref, win_lo, win_hi, records + ((int64_t)t * kSlotsPerTile + l * 4), P);
    if (NW == 2) {
        if (var_wave) s_vmask[l] = (uint8_t)vmask;
        __syncthreads();
        if (!ref_wave) return;
        vmask = s_vmask[l];
    }
#ifdef PISCES_TIMING
    const long long tcB = wall_clock64();
#endif

    // validity bits, counts, tile directory
    const uint32_t valid = vmask ? vmask : (ref_emitted ? (1u << ref_rank) : 0u);
    // counts from ballots (scalar) instead of cross-lane reductions
    int n_surv = 0, n_var = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        n_surv += __popcll(__ballot((valid >> k) & 1u));
        n_var += __popcll(__ballot((vmask >> k) & 1u));
    }
    const int n_call_total = n_var + __popcll(__ballot(ref_emitted));   // IsCallable is always true for a Reference allele
    const int n_loci_called = __popcll(__ballot(valid != 0));
    // nibble of lane l -> bits 4*(l&7).. of word l>>3 of the 256-bit mask
    uint32_t word = valid << ((l & 7) * 4);
    word |= __shfl_xor(word, 1, 64);
    word |= __shfl_xor(word, 2, 64);
    word |= __shfl_xor(word, 4, 64);
    PiscesTileResult* const tile_result = &tile_results[t];
    if ((l & 7) == 0) tile_result->valid[l >> 3] = word;
    if (l == 0) {
        tile_result->record_begin = t * kSlotsPerTile;
        tile_result->n_records = n_surv;
        tile_result->n_candidate_loci = n_loci_called;
        tile_result->n_called = n_call_total;   // IAlleleCaller.TotalNumCalled contribution
        if (P.totals) {
            unsigned long long* tt = P.totals + (size_t)(t % kTotalShards) * kTotalStride;
            atomicAdd(&tt[0], (unsigned long long)n_surv);
            atomicAdd(&tt[1], (unsigned long long)n_loci_called);
            atomicAdd(&tt[2], (unsigned long long)n_call_total);
            atomicAdd(&tt[3], 1ull);
        }
#ifdef PISCES_TIMING
        const long long tc2 = wall_clock64();
        tile_result->record_begin = (int)(tc0 & 0x3FFFFFFF);
        tile_result->n_records = (int)(tc1 & 0x3FFFFFFF);
        tile_result->n_candidate_loci = (int)(tc2 & 0x3FFFFFFF);
        tile_result->n_called = (int)__builtin_amdgcn_s_getreg(63492);   // HW_REG_HW_ID
        tile_result->valid[0] = __builtin_amdgcn_s_getreg(63508);        // HW_REG_XCC_ID
        tile_result->valid[1] = 0;
        tile_result->valid[2] = (uint32_t)(tcA - tc1);   // counts + Reference pass
        tile_result->valid[3] = (uint32_t)(tcB - tcA);   // variants + meeting
        tile_result->valid[4] = 0;
        tile_result->valid[5] = (uint32_t)(tc2 - tcB);   // directory
#endif
    }
}

template <int NW>
__global__ __launch_bounds__(64 * NW, NW == 1 ? PISCES_WAVE_OCC : PISCES_WAVE2_OCC) void call_tiles_wave_kernel(
    const uint32_t* __restrict__ tuples, const PiscesTile* __restrict__ tiles, int32_t n_tiles,
    const uint8_t* __restrict__ ref, int32_t ref_start, int64_t ref_len, PiscesCalledAllele* __restrict__ records,
    PiscesTileResult* __restrict__ tile_results, DeviceParams P, const DeviceParams* __restrict__ Pd /* the same in device memory, for the cold path */)
{
    __shared__ __attribute__((aligned(16))) int hist[2 * kWaveRegion];   // 16 KiB: [region][allele:direction row][column]
    __shared__ uint8_t s_refwin[kRefWin];
    __shared__ uint8_t s_vmask[kTile];

    const int t = blockIdx.x;
    if (t >= n_tiles) return;
    const int l = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const PiscesTile tile = tiles[t];
#ifdef PISCES_TIMING
    const long long tc0 = wall_clock64();
#endif
    auto setup = [&]() {
        int4* h4 = reinterpret_cast<int4*>(hist);
        for (int i = threadIdx.x; i < 2 * kWaveRegion / 4; i += 64 * NW) h4[i] = make_int4(0, 0, 0, 0);
        for (int i = threadIdx.x; i < kRefWin; i += 64 * NW) {
            const int64_t ri = (int64_t)tile.start_position - kRefMargin + i - ref_start;
            s_refwin[i] = (ri >= 0 && ri < ref_len) ? ref[ri] : (uint8_t)0;
        }
        __syncthreads();
    };

    {
        // single-round launches (NW = 2): streaming waves win the issue arbitration over waves in their call phase — the launch
        // ends with the slowest tile, and a tile that is still streaming has its whole call phase ahead of it
#ifndef PISCES_NO_PRIO
        if (NW == 2) __builtin_amdgcn_s_setprio(3);
#endif
        const uint32_t min_bq_shifted = (uint32_t)min(max(P.min_bq, 0), 255) << 24;
#if defined(PISCES_ABLATE) && PISCES_ABLATE == 2
        uint32_t acc = 0;   // development ablation: loads only
        stream_tuples_wave<NW>(tuples, tile.tuple_begin, tile.tuple_end, l, wid, [&](uint32_t v) { acc ^= v; }, setup);
        if (acc == 0x12345u) hist[l] = (int)min_bq_shifted;
#else
        stream_tuples_wave<NW>(tuples, tile.tuple_begin, tile.tuple_end, l, wid, [&](uint32_t v) { accumulate_wave(hist, v, min_bq_shifted); }, setup);
#endif
    }
    __syncthreads();
    if (NW == 2) __builtin_amdgcn_s_setprio(0);
#if defined(PISCES_ABLATE) && PISCES_ABLATE >= 1
    if (threadIdx.x == 0) { tile_results[t].record_begin = 0; tile_results[t].n_records = hist[5 + l]; }
    return;
#endif
#ifdef PISCES_TIMING
    const long long tc1 = wall_clock64();
#endif

    call_phase_wave<NW>(hist, s_refwin, s_vmask, tile, t, l, wid, ref, ref_start, ref_len, records, tile_results, P
#ifdef PISCES_TIMING
                        , tc0, tc1
#endif
                        );
}

// The handle's memo tables (DeviceParams), filled with the very functions the call phase would otherwise run.
__global__ __launch_bounds__(256) void build_vq_tab_kernel(int16_t* __restrict__ tab, int32_t n_k, int32_t n_cov, DeviceParams P)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)n_k * n_cov) return;
    tab[i] = (int16_t)poisson_qscore((int32_t)(i / n_cov), (int32_t)(i % n_cov), P);
}
__global__ __launch_bounds__(256) void build_sb_tab_kernel(double* __restrict__ tab, double* __restrict__ tab0, int32_t n_k, int32_t n_cov, DeviceParams P)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)n_k * n_cov) return;
    const int k = (int)(i / n_cov), cov = (int)(i % n_cov);
    // sb_create_stats with support >= 1 (Poisson / Extended): Math.Max(0, Poisson.Cdf(support - 1, coverage * noiseFreq))
    tab[i] = k >= 1 ? fmax(0.0, poisson_cdf_sb((double)k - 1, (double)cov * P.err_sb)) : 0.0;
    if (k == 0) tab0[cov] = pow(1 - P.err_sb, (double)cov);   // support == 0, Extended model
}
__global__ __launch_bounds__(256) void build_gq_cap_kernel(int16_t* __restrict__ tab, const double* __restrict__ gq_tail, int32_t n_a, int32_t n_cov, DeviceParams P)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)n_a * n_cov) return;
    // somatic_gq_finish for a hom-ref / hom-alt call whose variant q-score is the cap
    if (i < n_cov) { tab[i] = 0; return; }   // row a = 0 is never addressed
    double rawQ = p_to_q(q_to_p_int(P.max_vq, P) + gq_tail[i]);
    double qScore = fmin((double)P.max_gq, rawQ);
    qScore = fmax(qScore, (double)P.min_gq);
    tab[i] = (int16_t)rint(qScore);
}

// streaming-read probe: the hot kernel's load pattern (non-temporal dwordx4, 8 per lane in flight) and nothing else
// pisces_hip_device_totals: the running totals' shards added up into host-visible memory as this kernel's own stores (no copy operation,
// one stream wait on the host), and cleared when asked to
__global__ __launch_bounds__(64) void totals_collect_kernel(unsigned long long* __restrict__ shards, unsigned long long* __restrict__ out_host, int reset)
{
    const int l = threadIdx.x;
    if (l < 4) {
        unsigned long long v = 0;
        for (int sh = 0; sh < kTotalShards; sh++) v += shards[sh * kTotalStride + l];
        out_host[l] = v;
    }
    if (reset)
        for (int i = l; i < kTotalShards * kTotalStride; i += 64) shards[i] = 0ull;
    __threadfence_system();
}

__global__ __launch_bounds__(256) void read_probe_kernel(const u32x4* __restrict__ p, int64_t n4, uint32_t* __restrict__ sink)
{
    uint32_t acc = 0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x * 8;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x * 8 + threadIdx.x; i < n4; i += stride) {
        u32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int64_t j = i + (int64_t)u * blockDim.x;
            v[u] = j < n4 ? __builtin_nontemporal_load(&p[j]) : (u32x4){0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int u = 0; u < 8; u++) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    if (acc == 0x9E3779B9u) *sink = acc;
}

// the same with the read store kernel's loads: one dword a lane (256 bytes a wave instruction), eight in flight.  Only for calibrating
// FETCH_SIZE on that width (PISCES_HIP_PROBE_DWORD=1; tools/fetch_calibration.py).
__global__ __launch_bounds__(256) void read_probe_dword_kernel(const uint32_t* __restrict__ p, int64_t n, uint32_t* __restrict__ sink)
{
    uint32_t acc = 0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x * 8;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x * 8 + threadIdx.x; i < n; i += stride) {
        uint32_t v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int64_t j = i + (int64_t)u * blockDim.x;
            v[u] = j < n ? p[j] : 0u;
        }
#pragma unroll
        for (int u = 0; u < 8; u++) acc ^= v[u];
    }
    if (acc == 0x9E3779B9u) *sink = acc;
}

// fills DeviceParams::gq_tail with the function the call phase would evaluate (bit-identical by construction)
__global__ void build_gq_tail_kernel(double* __restrict__ table, int32_t n_a, int32_t n_cov, float target_lod)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)n_a * n_cov) return;
    const int a = (int)(i / n_cov), cov = (int)(i % n_cov);
    const float expectedF = target_lod * (float)cov;
    table[i] = a >= 1 ? incomplete_gamma_function((double)a, (double)expectedF) : -1.0;
}

// ------------------------------------------------------------------------------------------
// Anchor-resolved accumulation: LDS [locus][199] (odd stride: consecutive loci hit distinct banks),
// added into counts[(tile*kTile + locus)][6][3][11] — RegionState._alleleCounts layout.
// AlleleCountHelper.GetAnchorAdjustedAlleleCount (lib/Pisces.Processing/RegionState/AlleleCountHelper.cs:21-85) over one [11] row of
// counts (int) or of base-quality sums (double, GetAnchorAdjustedTotalQuality: same walk, same order of additions); maxAnchor < 0 =
// null; symmetric is never set on this path
template <typename T, typename R>
__host__ __device__ inline R anchor_adjusted(const T* __restrict__ row, int minAnchor, int maxAnchor, bool fromEnd)
{
    const int wellAnchoredIndex = PISCES_ANCHOR_SIZE, numAnchorIndexes = PISCES_NUM_ANCHORS;
    const int trueMinAnchor = wellAnchoredIndex < minAnchor ? wellAnchoredIndex : minAnchor;
    int initialMaxAnchor = wellAnchoredIndex;
    if (maxAnchor >= 0) {
        if (maxAnchor >= wellAnchoredIndex) initialMaxAnchor = wellAnchoredIndex - 1;
        if (maxAnchor < wellAnchoredIndex) initialMaxAnchor = maxAnchor;
    }
    R tot = 0;
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
__host__ __device__ inline double get_base_quality_sum(const double* __restrict__ sumq, int64_t idx, int allele, int dir, int minAnchor,
                                                       int maxAnchor, bool fromEnd)
{
    if (idx < 0) return 0.0;
    return anchor_adjusted<double, double>(sumq + idx * PISCES_COUNTS_PER_LOCUS + (allele * 3 + dir) * PISCES_NUM_ANCHORS, minAnchor, maxAnchor,
                                           fromEnd);
}

constexpr int kAnchStride = PISCES_COUNTS_PER_LOCUS + 1;

// The base-quality sums are accumulated in FIXED POINT: what a base of quality q adds, Math.Pow(10, -1 * (int)q / 10f)
// (RegionStateManager.cs:191), is cut at 2^-76 and carried as two 38-bit halves that go to two 64-bit integer counters of the cell.
// Integer addition does not care about the order the atomics arrive in, so the sums are the same bits from run to run (FP64 atomics in
// arrival order were not), and exact to 2^-76 per term; finish_quality_sums_kernel turns a cell's pair into the double the call
// phase and pisces_hip_get_base_quality_sums read.  (The reference adds doubles in read order; its result and this one are both within
// n * 2^-53 relative of the true sum, which is what SURVEY promises for this mode.)
constexpr int kSumqHalfBits = 38;
__global__ __launch_bounds__(256) void finish_quality_sums_kernel(const unsigned long long* __restrict__ fix, double* __restrict__ sumq, int64_t n_cells)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_cells) return;
    const unsigned long long hi = fix[2 * i], lo = fix[2 * i + 1];
    const unsigned long long hi_total = hi + (lo >> kSumqHalfBits), lo_rest = lo & ((1ull << kSumqHalfBits) - 1ull);
    sumq[i] = (double)hi_total * 0x1p-38 + (double)lo_rest * 0x1p-76;
}

__global__ __launch_bounds__(kBlock) void accumulate_tiles_kernel(
    const uint32_t* __restrict__ tuples, const PiscesTile* __restrict__ tiles, int32_t n_tiles,
    int32_t* __restrict__ counts, int32_t min_bq_,
    unsigned long long* __restrict__ sumq = nullptr /* NoiseModel.Window: RegionState._sumOfAlleleBaseQualities, two fixed-point halves per cell */,
    const ulonglong2* __restrict__ bq_lut = nullptr /* [256] the halves of Math.Pow(10, -1 * (int)q / 10f) */)
{
    __shared__ int hist[kTile * kAnchStride];
    const int t = blockIdx.x;
    if (t >= n_tiles) return;
    const PiscesTile tile = tiles[t];
    for (int i = threadIdx.x; i < kTile * kAnchStride; i += kBlock) hist[i] = 0;
    __syncthreads();
    const uint32_t n_loci = (uint32_t)tile.n_loci, min_bq = (uint32_t)min_bq_;
    stream_tuples(tuples, tile.tuple_begin, tile.tuple_end, [&](uint32_t v) {
        uint32_t locus = PISCES_TUPLE_LOCUS(v);
        uint32_t anchor = PISCES_TUPLE_ANCHOR(v);
        uint32_t dir = PISCES_TUPLE_DIR(v);
        uint32_t allele = PISCES_TUPLE_ALLELE(v);
        uint32_t qual = v >> 24;
        if (allele < 4u && qual < min_bq) allele = 4u;
        if (locus < n_loci && dir < 3u && allele < 6u && anchor < (uint32_t)PISCES_NUM_ANCHORS) {
            atomicAdd(&hist[locus * kAnchStride + (allele * 3u + dir) * PISCES_NUM_ANCHORS + anchor], 1);
            // the quality of a base goes under its post-threshold allele type; only A/C/G/T are ever read back (CoverageCalculator.cs:62)
            if (sumq && allele < 4u) {
                const ulonglong2 add = bq_lut[qual];
                unsigned long long* cell = &sumq[2 * (((int64_t)t * kTile + locus) * PISCES_COUNTS_PER_LOCUS + (allele * 3u + dir) * PISCES_NUM_ANCHORS + anchor)];
                atomicAdd(cell, add.x);
                atomicAdd(cell + 1, add.y);
            }
        }
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

// Call phase fed from the global anchor-resolved counts (+ gapped-MNV reference counts, CoverageCalculator.cs:82-97).
__global__ __launch_bounds__(kBlock, 4) void call_counts_kernel(
    const int32_t* __restrict__ counts, const uint32_t* __restrict__ gapped_mnv_ref,
    const PiscesTile* __restrict__ tiles, int32_t n_tiles, const uint8_t* __restrict__ ref, int32_t ref_start,
    int64_t ref_len, PiscesCalledAllele* __restrict__ records, PiscesTileResult* __restrict__ tile_results, DeviceParams P,
    const double* __restrict__ sumq = nullptr /* NoiseModel.Window */)
{
    __shared__ int hist[kFolded * kTile];
    __shared__ uint32_t s_gapped[kTile];
    __shared__ int32_t s_lvl[kTile];
    __shared__ uint8_t s_mask[kTile];
    __shared__ uint8_t s_refwin[kRefWin];
    __shared__ VarScratch s_var;
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
    if (threadIdx.x >= 128 && threadIdx.x < 128 + kRefWin) {
        const int j = threadIdx.x - 128;
        const int64_t ri = (int64_t)tile.start_position - kRefMargin + j - ref_start;
        s_refwin[j] = (ri >= 0 && ri < ref_len) ? ref[ri] : (uint8_t)0;
    }
    __syncthreads();
    if (sumq && threadIdx.x < kTile) {
        // CalculateSinglePoint :49-98: SumOfBaseQuality and TotalCoverage of a point allele of this locus, in the reference's order
        const int l = threadIdx.x;
        double sum = 0.0;
        int total = 0;
        if (l < tile.n_loci) {
            const int cca[5] = {PISCES_ALLELE_A, PISCES_ALLELE_C, PISCES_ALLELE_G, PISCES_ALLELE_T, PISCES_ALLELE_DEL};
            for (int d = 0; d < 3; d++)
                for (int k = 0; k < 5; k++) {
                    total += hist[HistBlock::idx(cca[k], d, l)];
                    sum += get_base_quality_sum(sumq, (int64_t)t * kTile + l, cca[k], d, 0, -1, false);
                }
        }
        s_lvl[l] = window_level(sum, total);
    }
    __syncthreads();
    call_roles(hist, s_gapped, sumq ? s_lvl : nullptr, tile, t, ref, ref_start, ref_len, records, &tile_results[t], P, s_mask, s_refwin, &s_var);
}

// ------------------------------------------------------------------------------------------
// Ordered compaction of the slot layout: offsets[t] = sum of n_records of tiles < t (one workgroup scans the
// directory), then each tile's valid slots are copied, in slot order, to out[offsets[t] ..].  The result is the
// called alleles of the whole launch sorted by (position, allele) in one contiguous buffer.
__global__ __launch_bounds__(1024) void scan_tile_counts_kernel(const PiscesTileResult* __restrict__ tr, int32_t n_tiles,
                                                                int32_t* __restrict__ offsets, int32_t* __restrict__ total,
                                                                int32_t* __restrict__ called_out /* optional */)
{
    __shared__ int s_wave[2][16];   // the sixteen waves' totals of a chunk of 1024 tiles (two sets: one barrier a chunk)
    __shared__ int s_called;
    if (threadIdx.x == 0) s_called = 0;
    int called = 0;   // IAlleleCaller.TotalNumCalled of the launch: total[1]
    int carry = 0;    // records of the chunks before this one (the same in every thread)
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int set = 0;
    for (int base = 0; base < n_tiles; base += 1024, set ^= 1) {
        const int i = base + threadIdx.x;
        const int v = i < n_tiles ? tr[i].n_records : 0;
        if (i < n_tiles) called += tr[i].n_called;
        int x = v;   // inclusive scan inside the wave (shuffles), then the waves' totals through LDS
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int y = __shfl_up(x, d, 64);
            if (lane >= d) x += y;
        }
        if (lane == 63) s_wave[set][w] = x;
        __syncthreads();
        int before = 0, chunk = 0;
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const int t = s_wave[set][k];
            before += k < w ? t : 0;
            chunk += t;
        }
        if (i < n_tiles) offsets[i] = carry + before + x - v;
        carry += chunk;
    }
    __syncthreads();
    const int s_carry = carry;
    if (called) atomicAdd(&s_called, called);
    __syncthreads();
    if (threadIdx.x == 0) {
        *total = s_carry;
        if (called_out) *called_out = s_called;
    }
}

__global__ __launch_bounds__(64) void gather_records_kernel(const PiscesCalledAllele* __restrict__ records,
                                                            const PiscesTileResult* __restrict__ tr, int32_t n_tiles,
                                                            const int32_t* __restrict__ offsets,
                                                            PiscesCalledAllele* __restrict__ out, int32_t capacity)
{
    const int t = blockIdx.x;
    if (t >= n_tiles) return;
    const int l = threadIdx.x;
    const uint32_t nib = (tr[t].valid[l >> 3] >> ((l & 7) * 4)) & 0xFu;
    int x = __popc(nib);
    const int mine = x;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int y = __shfl_up(x, d, 64);
        if (l >= d) x += y;
    }
    int64_t dst = (int64_t)offsets[t] + (x - mine);
    const PiscesCalledAllele* src = records + (int64_t)tr[t].record_begin + l * 4;
    for (int k = 0; k < 4; k++) {
        if (!(nib & (1u << k))) continue;
        if (dst < capacity) copy_record(&out[dst], &src[k]);
        dst++;
    }
}


// The two launches above as ONE for launches of up to kGatherDirectTiles tiles (the flush of a batch: BASELINE config 2 is 1 563-1 786): the
// directory is a few KB that stay in L2, so a tile's wave simply ADDS UP the record counts of the tiles before it — lane i takes tiles i,
// i + 64, ... (28 independent loads a lane for the last tile of 1 786) — while its own validity mask is on the way: two dependent round
// trips (directory, records) where scan + gather took four and a launch gap.  The last tile's wave writes the totals.
constexpr int kGatherDirectTiles = 4096;
__global__ __launch_bounds__(64) void gather_direct_kernel(const PiscesCalledAllele* __restrict__ records, const PiscesTileResult* __restrict__ tr, int32_t n_tiles,
                                                           int32_t* __restrict__ offsets /* optional */, PiscesCalledAllele* __restrict__ out, int32_t capacity,
                                                           int32_t* __restrict__ total, int32_t* __restrict__ called_out /* optional */)
{
    const int t = blockIdx.x, l = threadIdx.x;
    if (t >= n_tiles) return;
    const uint32_t nib = (tr[t].valid[l >> 3] >> ((l & 7) * 4)) & 0xFu;
    const int32_t begin = tr[t].record_begin;
    const bool last = t == n_tiles - 1;
    int before = 0, called = 0;
    for (int i0 = 0; i0 < t; i0 += 64 * 32) {   // 32 independent loads a lane a round: one round trip up to 2 048 tiles (the loop's round trips are what the kernel costs)
        int v[32];
#pragma unroll
        for (int k = 0; k < 32; k++) v[k] = tr[min(i0 + 64 * k + l, t - 1)].n_records;
#pragma unroll
        for (int k = 0; k < 32; k++) before += i0 + 64 * k + l < t ? v[k] : 0;
    }
    if (last && called_out)   // IAlleleCaller.TotalNumCalled of the launch (one wave of the launch)
        for (int i = l; i < t; i += 64) called += tr[i].n_called;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { before += __shfl_xor(before, d, 64); called += __shfl_xor(called, d, 64); }
    if (l == 0) {
        if (offsets) offsets[t] = before;
        if (last) {
            *total = before + tr[t].n_records;
            if (called_out) *called_out = called + tr[t].n_called;
        }
    }
    int x = __popc(nib);
    const int mine = x;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int y = __shfl_up(x, d, 64);
        if (l >= d) x += y;
    }
    int64_t dst = (int64_t)before + (x - mine);
    const PiscesCalledAllele* src = records + (int64_t)begin + l * 4;
    for (int k = 0; k < 4; k++) {
        if (!(nib & (1u << k))) continue;
        if (dst < capacity) copy_record(&out[dst], &src[k]);
        dst++;
    }
}

// The two launches above as ONE (large launches: the flush of many blocks, the device-resident surface): a workgroup takes sixteen tiles,
// publishes the number of their records, finds the records of all tiles before its own by a decoupled look-back over the workgroups'
// words (64 predecessors a round; a word = launch epoch << 34 | status << 32 | count, so words of earlier launches read as "not yet" and
// nothing is cleared between launches), and copies its tiles' valid slots to their final places, a wave four tiles.  Workgroups are
// dispatched in index order and wait for lower indices only.  The last workgroup writes the totals.
constexpr int kCompactTiles = 16;
__global__ __launch_bounds__(256) void compact_records_kernel(const PiscesCalledAllele* __restrict__ records, const PiscesTileResult* __restrict__ tr, int32_t n_tiles,
                                                              unsigned long long* __restrict__ state, uint32_t epoch, PiscesCalledAllele* __restrict__ out, int32_t capacity,
                                                              int32_t* __restrict__ total, int32_t* __restrict__ called_out /* optional */, int32_t* __restrict__ offsets /* optional */)
{
    __shared__ int s_excl, s_tile[kCompactTiles];
    const int b = (int)blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int t0 = b * kCompactTiles;
    const unsigned long long tag = (unsigned long long)epoch << 34;
    if (wave == 0) {
        const int t = t0 + lane;
        const int v = (lane < kCompactTiles && t < n_tiles) ? tr[t].n_records : 0;
        int x = v;   // inclusive scan over the workgroup's tiles
#pragma unroll
        for (int d = 1; d < kCompactTiles; d <<= 1) {
            const int y = __shfl_up(x, d, 64);
            if (lane >= d) x += y;
        }
        const int agg = __shfl(x, kCompactTiles - 1, 64);
        if (lane < kCompactTiles) s_tile[lane] = x - v;
        long long excl = 0;
        if (b == 0) {
            if (lane == 0) __hip_atomic_store(&state[0], tag | (2ull << 32) | (unsigned long long)(unsigned)agg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            if (lane == 0) __hip_atomic_store(&state[b], tag | (1ull << 32) | (unsigned long long)(unsigned)agg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int base = b - 1;
            for (;;) {
                const int idx = base - lane;
                unsigned long long w = idx >= 0 ? __hip_atomic_load(&state[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (tag | (2ull << 32));
                const int status = (w >> 34) == (unsigned long long)epoch ? (int)((w >> 32) & 3ull) : 0;
                const unsigned long long none = __ballot(status == 0), incl = __ballot(status == 2);
                const int first_incl = incl ? __builtin_ctzll(incl) : 64;
                const unsigned long long upto = first_incl >= 63 ? ~0ull : ((2ull << first_incl) - 1ull);
                if (none & upto) { __builtin_amdgcn_s_sleep(1); continue; }
                int f = lane <= first_incl ? (int)(uint32_t)w : 0;
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) f += __shfl_xor(f, d, 64);
                excl += f;
                if (first_incl < 64) break;
                base -= 64;
            }
            if (lane == 0) __hip_atomic_store(&state[b], tag | (2ull << 32) | (unsigned long long)(unsigned)(excl + agg), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (lane == 0) {
            s_excl = (int)excl;
            if (b == (int)gridDim.x - 1) *total = (int)excl + agg;
        }
    }
    if (called_out && b == (int)gridDim.x - 1 && wave == 1) {   // IAlleleCaller.TotalNumCalled of the launch (the tiles' own counts: no other workgroup is waited for)
        int called = 0;
        for (int i = lane; i < n_tiles; i += 64) called += tr[i].n_called;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) called += __shfl_xor(called, d, 64);
        if (lane == 0) *called_out = called;
    }
    __syncthreads();
    const int excl = s_excl;
#pragma unroll 1
    for (int k = 0; k < kCompactTiles / 4; k++) {
        const int j = wave * (kCompactTiles / 4) + k, t = t0 + j;
        if (t >= n_tiles) break;
        const int at = excl + s_tile[j];
        if (offsets && lane == 0) offsets[t] = at;
        const uint32_t nib = (tr[t].valid[lane >> 3] >> ((lane & 7) * 4)) & 0xFu;
        int x = __popc(nib);
        const int mine = x;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int y = __shfl_up(x, d, 64);
            if (lane >= d) x += y;
        }
        int64_t dst = (int64_t)at + (x - mine);
        const PiscesCalledAllele* src = records + (int64_t)tr[t].record_begin + lane * 4;
        for (int q = 0; q < 4; q++) {
            if (!(nib & (1u << q))) continue;
            if (dst < capacity) copy_record(&out[dst], &src[q]);
            dst++;
        }
    }
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
    int32_t gapped;                  // GetGappedMnvRefCount at the position (SNV: off the reference support, Reference: off the support)
    int32_t reprocessed;             // an MNV candidate on its second ProcessVariant (AlleleCaller.cs:74 then :111): CalledAllele.SumOfBaseQuality
                                     // keeps what the first pass added (CoverageCalculator.cs:226-227 use +=), which NoiseModel.Window then sees twice
    int32_t pad;
};

// AlleleCountHelper.GetAnchorAdjustedAlleleCount (lib/Pisces.Processing/RegionState/AlleleCountHelper.cs:21-85)
// over one [11] row; maxAnchor < 0 = null; symmetric is never set on this path
__host__ __device__ inline int anchor_adjusted_count(const int32_t* __restrict__ row, int minAnchor, int maxAnchor, bool fromEnd)
{
    return anchor_adjusted<int32_t, int>(row, minAnchor, maxAnchor, fromEnd);
}

// Where a candidate's counts come from: the anchor-resolved tensor int32[locus][6][3][11] (accumulate_store_tiles_kernel), or — for
// point alleles, MNVs and deletions, which add up all eleven anchor bins of a cell anyway (only an insertion's coverage looks at the bins,
// CoverageCalculator.cs:179-259) — the FOLDED counts int32[locus][6][3] that the flush's tile kernel leaves behind for every locus it
// walks (DeviceParams::folded_out): no second walk over the reads for them.  A locus index >= 0 is the tensor's, -1 is "no block"
// (RegionStateManager.GetAlleleCount -> 0, RegionStateManager.cs:222-226), <= -2 is the folded array's, kept as -(index + 2).
struct CountsView {
    const int32_t* tensor;
    const int32_t* folded;
    __host__ __device__ CountsView(const int32_t* t, const int32_t* f = nullptr) : tensor(t), folded(f) {}
};
__host__ __device__ inline int64_t folded_locus(int64_t index) { return -(index + 2); }

__host__ __device__ inline int get_allele_count(const CountsView counts, int64_t idx, int allele, int dir, int minAnchor,
                                       int maxAnchor, bool fromEnd)
{
    if (idx == -1) return 0;   // RegionStateManager.GetAlleleCount: no block -> 0 (RegionStateManager.cs:222-226)
    if (idx < -1) return counts.folded ? counts.folded[folded_locus(idx) * PISCES_FOLDED_PER_LOCUS + allele * 3 + dir] : 0;   // (all anchors: minAnchor 0, no maxAnchor)
    return anchor_adjusted_count(counts.tensor + idx * PISCES_COUNTS_PER_LOCUS + (allele * 3 + dir) * PISCES_NUM_ANCHORS, minAnchor,
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

// CoverageCalculator.CalculateSpanning (CoverageCalculator.cs:162-321) for an insertion / deletion / MNV candidate over the
// anchor-resolved counts tensor: coverage by direction, total coverage.  Host and device: the host-side collapser needs the
// same number (CandidateAllele.Frequency) the device call uses.
struct SpanningCoverage { int cov[3]; int total; };
__host__ __device__ inline SpanningCoverage spanning_coverage(const DevCandidate& c, const CountsView counts, int32_t expect_stitched,
                                                             const double* __restrict__ sumq = nullptr, double* sum_of_base_quality = nullptr)
{
    const int length = c.category == PISCES_CAT_INSERTION ? c.alt_len - 1 : c.category == PISCES_CAT_DELETION ? c.ref_len - 1 : c.alt_len;   // BaseAllele.Length
    const int support = c.sup[0] + c.sup[1] + c.sup[2];
    const int wellAnchored = c.anch[0] + c.anch[1] + c.anch[2];
    const bool presumeAnchoredForExactCov = c.category == PISCES_CAT_INSERTION ? (expect_stitched != 0) : true;   // :31-41
    const bool bePicky = c.category == PISCES_CAT_INSERTION;   // considerAnchorInformation (TrackedAnchorSize 5 > 0), :179

    int startPointCoverage[3] = {0, 0, 0}, endPointCoverage[3] = {0, 0, 0};
    int startUnanch[3] = {0, 0, 0}, endUnanch[3] = {0, 0, 0};
    int confidentLeft = 0, confidentRight = 0, suspiciousLeft = 0, suspiciousRight = 0;
    const int unanchoredSupport = support - wellAnchored;
    const int cca[5] = {PISCES_ALLELE_A, PISCES_ALLELE_C, PISCES_ALLELE_G, PISCES_ALLELE_T, PISCES_ALLELE_DEL};
    double sumQ = 0.0, unanchoredStartQ = 0.0, unanchoredEndQ = 0.0;   // SumOfBaseQuality :226-227,245,254 (NoiseModel.Window)
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
            if (sumq) {
                sumQ += get_base_quality_sum(sumq, c.start_idx, at, d, minAnchorStart, -1, false);
                sumQ += get_base_quality_sum(sumq, c.end_idx, at, d, minAnchorEnd, -1, true);
            }
            if (bePicky && unanchoredSupport > 0) {
                if (minAnchorStart > 0) {
                    const int u = get_allele_count(counts, c.start_idx, at, d, 0, minAnchorStart - 1, false);
                    startUnanch[d] += u;
                    suspiciousLeft += u;
                    if (sumq) unanchoredStartQ += get_base_quality_sum(sumq, c.start_idx, at, d, 0, minAnchorStart - 1, false);
                }
                if (minAnchorEnd > 0) {
                    const int u = get_allele_count(counts, c.end_idx, at, d, 0, minAnchorEnd - 1, true);
                    endUnanch[d] += u;
                    suspiciousRight += u;
                    // the reference reads the START point here (CoverageCalculator.cs:254) - reproduced
                    if (sumq) unanchoredEndQ += get_base_quality_sum(sumq, c.start_idx, at, d, 0, minAnchorEnd - 1, true);
                }
            }
        }
    }
    if (bePicky) {   // :261-293, float32 arithmetic as written
        const float trulyAnchoredCoverage = (((confidentLeft - suspiciousRight) + (confidentRight - suspiciousLeft)) / 2.0f);
        const float anchoredVariantFreq = trulyAnchoredCoverage <= 0 ? 0.0f : (float)wellAnchored / trulyAnchoredCoverage;
        const int totalSuspicious = suspiciousLeft + suspiciousRight;
        const float unanchoredVariantFreq = totalSuspicious == 0 ? 0.0f : unanchoredSupport / ((float)totalSuspicious);
        float w = anchoredVariantFreq == 0 ? 1.0f : ((unanchoredVariantFreq / anchoredVariantFreq) < 1.0f ? (unanchoredVariantFreq / anchoredVariantFreq) : 1.0f);
        if (!(w > 0.0f)) w = 0.0f;
        const double weight = w;
        for (int d = 0; d < 3; d++) {
            startPointCoverage[d] += (int)(startUnanch[d] * weight);
            endPointCoverage[d] += (int)(endUnanch[d] * weight);
            sumQ += unanchoredStartQ * weight;   // inside the direction loop in the reference too (:284-292)
            sumQ += unanchoredEndQ * weight;
        }
    }
    if (sum_of_base_quality) *sum_of_base_quality = sumQ;
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
    SpanningCoverage r;
    r.cov[0] = cov[0]; r.cov[1] = cov[1]; r.cov[2] = cov[2];
    r.total = (int)exactTotalCoverage;
    return r;
}


// TotalCoverage of a candidate of any category against the counts tensor (the collapser's frequencies, CandidateAllele -> CalledAllele
// -> CoverageCalculator.Compute): point alleles sum the five coverage-contributing allele types over directions and anchors.
__host__ __device__ inline int candidate_total_coverage(const DevCandidate& c, const CountsView counts, int32_t expect_stitched)
{
    if (c.category == PISCES_CAT_SNV || c.category == PISCES_CAT_REFERENCE) {
        const int cca[5] = {PISCES_ALLELE_A, PISCES_ALLELE_C, PISCES_ALLELE_G, PISCES_ALLELE_T, PISCES_ALLELE_DEL};
        int total = 0;
        for (int d = 0; d < 3; d++)
            for (int k = 0; k < 5; k++) total += get_allele_count(counts, c.start_idx, cca[k], d, 0, -1, false);
        return total;
    }
    return spanning_coverage(c, counts, expect_stitched).total;
}

// PloidyModel.DiploidByThresholding / Haploid over the tile kernels' record slots, lane = locus (AlleleCaller.ComputeGenotypeAndFilterAllele,
// AlleleCaller.cs:143-177 with genotype_core.h): the rows of a locus are its valid slots — variants in rank order A C G T, which is their
// (REF, ALT) order, or the one Reference row —, every kept row gets the locus genotype, its own genotype q-score, the LowGQ and
// MultiAllelicSite filters and its phase-set index; rows beyond the ploidy lose their validity bit (the compaction that follows never
// sees them).  TotalNumCalled counts callable alleles before this (AlleleCaller.cs:236-258): n_called stays.  The device-resident
// surface (pisces_hip_call_tiles*) runs it behind every tile kernel of a diploid / haploid handle; a flush runs it when no row of the
// candidate kernel and no forced allele joins the tile kernels' rows (then the host pass over the merged rows does the same: diploid.cpp).
struct GenotypeParams {
    int32_t ploidy;
    float snv[3], indel[3];
    int32_t min_depth, min_gq, max_gq, low_gq_filter;
};
__global__ __launch_bounds__(64) void genotype_loci_kernel(PiscesCalledAllele* __restrict__ records, PiscesTileResult* __restrict__ tile_results, int32_t n_tiles,
                                                           GenotypeParams G, unsigned long long* __restrict__ totals)
{
    const int t = blockIdx.x;
    if (t >= n_tiles) return;
    const int l = threadIdx.x;
    PiscesTileResult* const tr = &tile_results[t];
    const uint32_t nib = (tr->valid[l >> 3] >> ((l & 7) * 4)) & 0xFu;
    uint32_t keep = nib;
    if (nib) {
        PiscesCalledAllele* const rows = records + (int64_t)t * kSlotsPerTile + l * 4;
        genotype::Allele a[4];
        int slot[4], order[5], n = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (!((nib >> k) & 1u)) continue;
            const PiscesCalledAllele& r = rows[k];
            a[n].category = (int32_t)PISCES_INFO_CATEGORY(r.info);
            a[n].support = r.allele_support; a[n].coverage = r.total_coverage; a[n].ref_support = r.reference_support;
            a[n].genotype = 0; a[n].genotype_qscore = 0; a[n].phase_set_index = 0; a[n].multi_allelic = false; a[n].prune = false;
            slot[n] = k;
            n++;
        }
        auto before = [](int x, int y) { return x < y; };   // slot rank = ordinal order of the ALT base (one REF base a locus)
        if (G.ploidy == PISCES_PLOIDY_HAPLOID) (void)genotype::haploid_set(a, n, order, G.snv[0], G.snv[1], G.min_depth, G.min_gq, G.max_gq, before);
        else (void)genotype::diploid_set(a, n, order, G.snv, G.indel, G.min_depth, G.min_gq, G.max_gq, before);
        for (int i = 0; i < n; i++) {
            if (a[i].prune) { keep &= ~(1u << slot[i]); continue; }
            PiscesCalledAllele& r = rows[slot[i]];
            r.info = (uint16_t)((r.info & ~0xFu) | ((uint32_t)a[i].genotype & 0xFu));
            r.genotype_qscore = (int16_t)a[i].genotype_qscore;
            uint32_t fb = r.filter_bits & ~(1u << PISCES_FILTER_LOW_GENOTYPE_QUALITY) & 0x3FFFu;
            if (a[i].multi_allelic) fb |= 1u << PISCES_FILTER_MULTI_ALLELIC_SITE;
            if (G.low_gq_filter >= 0 && (float)a[i].genotype_qscore < (float)G.low_gq_filter) fb |= 1u << PISCES_FILTER_LOW_GENOTYPE_QUALITY;
            fb |= (uint32_t)(a[i].phase_set_index & 3) << 14;
            r.filter_bits = (uint16_t)fb;
        }
    }
    int n_was = 0, n_now = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        n_was += __popcll(__ballot((nib >> k) & 1u));
        n_now += __popcll(__ballot((keep >> k) & 1u));
    }
    const int loci_was = __popcll(__ballot(nib != 0)), loci_now = __popcll(__ballot(keep != 0));
    uint32_t word = keep << ((l & 7) * 4);
    word |= __shfl_xor(word, 1, 64);
    word |= __shfl_xor(word, 2, 64);
    word |= __shfl_xor(word, 4, 64);
    if ((l & 7) == 0) tr->valid[l >> 3] = word;
    if (l == 0) {
        tr->n_records = n_now;
        tr->n_candidate_loci = loci_now;
        if (totals && (n_now != n_was || loci_now != loci_was)) {   // (the tile kernel has added its own counts: take back what went)
            unsigned long long* tt = totals + (size_t)(t % kTotalShards) * kTotalStride;
            atomicAdd(&tt[0], (unsigned long long)(long long)(n_now - n_was));
            atomicAdd(&tt[1], (unsigned long long)(long long)(loci_now - loci_was));
        }
    }
}

enum { kSpanningEveryRecord = 0, kSpanningCallableRecords = 1, kSpanningFlagsOnly = 2 };   // call_spanning_kernel's `wanted`

// IAlleleCaller's ProcessVariant + IsCallable for host-supplied candidates (one lane each): insertions / deletions always; with MNV
// calling on also the SNV and MNV candidates of the read walk and the Reference alleles that MNV reallocation touched
// (AlleleCaller.cs:60-141).  Point alleles (SNV, Reference) go through the tile kernels' own process_point_allele.
__global__ __launch_bounds__(64) void call_spanning_kernel(
    const DevCandidate* __restrict__ cands, int32_t n, const int32_t* __restrict__ counts_tensor, const uint8_t* __restrict__ alleles,
    const uint8_t* __restrict__ ref, int64_t ref_len /* ref[i] = position i+1 */, int32_t expect_stitched,
    PiscesCalledAllele* __restrict__ out, uint8_t* __restrict__ callable_out, DeviceParams P,
    const double* __restrict__ sumq = nullptr /* NoiseModel.Window */, const int32_t* __restrict__ counts_folded = nullptr,
    int32_t wanted = kSpanningEveryRecord)
{
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    const DevCandidate c = cands[i];
    const CountsView counts(counts_tensor, counts_folded);
    // What the caller reads back: IsCallable alone (the MNV pass: a failed MNV goes to the reallocator with its candidate, not its record),
    // or the records of the callable alleles (nothing looks at the record of an allele that is not callable unless it is a forced one).
    // An allele stops at the first IsCallable test it fails then: most candidates of a deep batch are one- and two-read errors below the
    // frequency threshold, and the q-score and strand-bias evaluations they would run are the long ones.
    const bool flags_only = wanted == kSpanningFlagsOnly, every_record = wanted == kSpanningEveryRecord;
    // the memo tables of the tile kernels' call phase (flat noise, Poisson / Extended strand bias), every entry requested up front; a
    // miss takes the evaluation the table was filled with
    const bool tabs = !sumq && tables_complete(P);
    if (c.category == PISCES_CAT_SNV || c.category == PISCES_CAT_REFERENCE) {
        // CalculateSinglePoint :49-98 over the anchor-resolved counts; AlleleSupport is the candidate's (AlleleHelper.Map)
        const bool isRef = c.category == PISCES_CAT_REFERENCE;
        LocusCounts lc;
        for (int a = 0; a < 6; a++)
            for (int d = 0; d < 3; d++) lc.h[a][d] = get_allele_count(counts, c.start_idx, a, d, 0, -1, false);
        const int rt = allele_type_of_base(alleles[c.allele_off]);
        const int a = isRef ? rt : allele_type_of_base(alleles[c.allele_off + c.ref_len]);
        PointCounts pc = point_counts_of(lc, a, isRef, rt, 0);
        if (isRef) {   // the Reference candidate's own support (RegionState.cs:414-447) + what reallocated MNVs added to it
            for (int d = 0; d < 3; d++) pc.sup[d] += c.sup[d];
            pc.support = pc.sup[0] + pc.sup[1] + pc.sup[2] - c.gapped;
            if (pc.support < 0) pc.support = 0;
        } else {
            for (int d = 0; d < 3; d++) pc.sup[d] = c.sup[d];
            pc.support = c.sup[0] + c.sup[1] + c.sup[2];
            pc.refsup -= c.gapped;
            if (pc.refsup < 0) pc.refsup = 0;
        }
        PiscesCalledAllele r;
        r.position = c.position; r.total_coverage = pc.total; r.allele_support = pc.support; r.reference_support = pc.refsup;
        r.num_no_calls = pc.nocalls;
        for (int d = 0; d < 3; d++) { r.coverage_by_dir[d] = pc.cov[d]; r.support_by_dir[d] = pc.sup[d]; }
        r.variant_qscore = 0; r.strand_bias_score = 0.0; r.genotype_qscore = 0; r.noise_level = 0; r.filter_bits = 0;
        r.info = PISCES_INFO_PACK(PISCES_GT_HET_ALT_REF, c.category, rt, a, 0, 0, 0);
        int32_t wlevel = kNoLevel;
        if (sumq) {   // CalculateSinglePoint: SumOfBaseQuality over the five coverage-contributing allele types, directions outermost
            const int cca[5] = {PISCES_ALLELE_A, PISCES_ALLELE_C, PISCES_ALLELE_G, PISCES_ALLELE_T, PISCES_ALLELE_DEL};
            double sum = 0.0;
            for (int d = 0; d < 3; d++)
                for (int k = 0; k < 5; k++) sum += get_base_quality_sum(sumq, c.start_idx, cca[k], d, 0, -1, false);
            wlevel = window_level(sum, pc.total);
        }
        if (!tabs) {
            const bool ok = process_point_allele<true, true>(pc, c.position, a, isRef, rt, ref, 0, ref_len, P, r, nullptr, 0, nullptr, sumq ? &wlevel : nullptr);
            out[i] = r;
            callable_out[i] = ok ? 1 : 0;
            return;
        }
        bool ok = isRef || variant_passes_frequency(pc, P);
        if (!ok && !every_record) { callable_out[i] = 0; return; }
        const AlleleTables tb = request_allele_tables(isRef, pc.support, pc.total, pc.refsup, pc.cov, pc.sup, P);
        int vq = 0;
        if (pc.support > 0 && pc.total != 0 && !poisson_qscore_try(pc.support, pc.total, P, tb, vq)) vq = poisson_qscore(pc.support, pc.total, P);
        if (!isRef && vq < P.min_vq) ok = false;
        callable_out[i] = ok ? 1 : 0;
        if (flags_only || (!ok && !every_record)) return;
        SbResult sb = {0.0, 0, 0, 0};
        if (pc.support > 0 && !strand_bias_try(pc.cov, pc.sup, P, tb, sb.bias_score, sb.acceptable, sb.var_both, sb.cov_both))
            sb = strand_bias<true>(pc.cov, pc.sup, P);
        const GqPre g = {tb.gq_idx, tb.gq_cap};
        if (!finish_allele<true>(pc, c.position, a, isRef, rt, vq, sb, ref, 0, ref_len, P, r, nullptr, 0, nullptr, &g))
            finish_allele<false>(pc, c.position, a, isRef, rt, vq, sb, ref, 0, ref_len, P, r, nullptr, 0);
        out[i] = r;
        return;
    }
    const int length = c.category == PISCES_CAT_INSERTION ? c.alt_len - 1 : c.category == PISCES_CAT_DELETION ? c.ref_len - 1 : c.alt_len;
    const int support = c.sup[0] + c.sup[1] + c.sup[2];
    double sumQ = 0.0;
    const SpanningCoverage sc = spanning_coverage(c, counts, expect_stitched, sumq, &sumQ);
    const int cov[3] = {sc.cov[0], sc.cov[1], sc.cov[2]};
    const int total = sc.total;
    int refsup = total - support;
    if (refsup < 0) refsup = 0;

    // IsCallable (AlleleCaller.cs:236-258), the tests that precede the q-score
    const float freq = frequency_f(support, total);
    bool callable = true;
    if (total < P.min_cov && !P.include_ref) callable = false;
    else if (total != 0 && freq < P.min_freq) callable = false;
    if (!callable && !every_record) { callable_out[i] = 0; return; }

    // ProcessVariant (AlleleCaller.cs:208-234)
    AlleleTables tb;
    if (tabs) tb = request_allele_tables(false, support, total, refsup, cov, c.sup, P);
    int vq = 0;
    int16_t noise_level = 0;   // NoiseLevelApplied: assigned with the q-score, i.e. for alleles with support
    if (support > 0) {
        noise_level = (int16_t)P.noise_level;
        if (sumq) {
            const int32_t level = window_level(c.reprocessed ? sumQ + sumQ : sumQ, total);
            noise_level = noise_level_field(level);
            if (total != 0 && level != kNoLevel) vq = poisson_qscore_e(support, total, window_err_of_level(level, P), P);
        } else if (total != 0) {
            if (!(tabs && poisson_qscore_try(support, total, P, tb, vq))) vq = poisson_qscore(support, total, P);
        }
    }
    if (vq < P.min_vq) callable = false;
    callable_out[i] = callable ? 1 : 0;
    if (flags_only || (!callable && !every_record)) return;
    SbResult sb = {0.0, 0, 0, 0};
    if (support > 0 && !(tabs && strand_bias_try(cov, c.sup, P, tb, sb.bias_score, sb.acceptable, sb.var_both, sb.cov_both)))
        sb = strand_bias<true>(cov, c.sup, P);
    uint32_t filters = 0;   // NumNoCalls stays 0 for spanning alleles -> FractionNoCalls 0
    if (P.low_depth_filter >= 0 && total < P.low_depth_filter) filters |= 1u << PISCES_FILTER_LOW_DEPTH;
    if (P.vq_filter >= 0 && vq < P.vq_filter && total != 0) filters |= 1u << PISCES_FILTER_LOW_VARIANT_QSCORE;
    if (!sb.acceptable || (P.filter_single_strand && !sb.var_both)) filters |= 1u << PISCES_FILTER_STRAND_BIAS;
    if (P.rmxn_max_len >= 0 && !(freq >= P.rmxn_freq_limit)) {   // RMxNCalculator.ShouldFilter :19-38
        int c1, c2 = 2147483647;
        if (c.category == PISCES_CAT_MNV) {   // :30-36: the reference allele's repeat against the alternate allele's at either end
            const uint8_t* rb = alleles + c.allele_off;
            const uint8_t* ab = alleles + c.allele_off + c.ref_len;
            c1 = rmxn_length_for_indel(c.position - 1, rb, c.ref_len, ref, ref_len, P.rmxn_max_len);
            const int i1 = rmxn_length_for_indel(c.position + c.ref_len - 1, ab, c.alt_len, ref, ref_len, P.rmxn_max_len);
            const int i2 = rmxn_length_for_indel(c.position - 1, ab, c.alt_len, ref, ref_len, P.rmxn_max_len);
            c2 = i1 > i2 ? i1 : i2;
        } else {
            const uint8_t* vb = alleles + c.allele_off + (c.category == PISCES_CAT_INSERTION ? c.ref_len + 1 : 1);
            c1 = rmxn_length_for_indel(c.position, vb, length, ref, ref_len, P.rmxn_max_len);
        }
        if ((c1 < c2 ? c1 : c2) >= P.rmxn_min_rep) filters |= 1u << PISCES_FILTER_RMXN;
    }
    if (P.vf_filter >= 0.0f && freq < P.vf_filter) filters |= 1u << PISCES_FILTER_LOW_VARIANT_FREQUENCY;
    if (expect_stitched) {   // AlleleProcessor.cs:64-68
        for (int k = 0; k < c.alt_len; k++)
            if (alleles[c.allele_off + c.ref_len + k] == 'N') filters |= 1u << PISCES_FILTER_STRAND_BIAS;
    }
    const int gt = somatic_genotype(false, total, support, refsup, P);
    int gq;
    if (!(tabs && somatic_gq_try(gt, vq, total, support, P, tb.gq_idx, tb.gq_cap, gq))) gq = somatic_gq(gt, vq, total, support, P);
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
    r.genotype_qscore = (int16_t)gq;
    r.noise_level = noise_level;
    r.filter_bits = (uint16_t)filters;
    const int rt = allele_type_of_base(alleles[c.allele_off]);
    r.info = PISCES_INFO_PACK(gt, c.category, rt, PISCES_ALLELE_N, sb.acceptable, sb.var_both, sb.cov_both);
    out[i] = r;
}

}  // namespace pisces
