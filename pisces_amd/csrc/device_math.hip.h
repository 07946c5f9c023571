// device_math.hip.h — FP64/FP32 device functions of the call phase for gfx950.
//
// These are the per-allele calculators of the Pisces path, written for one lane per
// allele: data-dependent trip counts, no LDS, no cross-lane traffic.  Arithmetic order and
// types follow the reference so that integer outputs are bit-exact and q-scores land on
// the same integer (compile with -ffp-contract=off; ocml exp/log/pow are within an ulp or
// two of the CLR's, which only matters within 1e-13 of a rounding boundary).
// Citations: paths relative to /root/reference/src.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/pisces_hip.h"

namespace pisces {

// Constants the host evaluates once per handle (so host and device agree on them bit for bit).
struct DeviceParams {
    int32_t min_bq, noise_level, max_vq, min_vq, vq_filter, min_cov, low_depth_filter;
    int32_t min_gq, max_gq, low_gq_filter, sb_model, filter_single_strand, include_ref, emit_zero_cov;
    int32_t rmxn_max_len, rmxn_min_rep;
    float min_freq, vf_filter, gt_min_freq, target_lod, nocall_thr, rmxn_freq_limit;
    double sb_threshold;  // (double)StrandBiasFilterThreshold
    double err_q;         // MathOperations.QtoP(NL)                       stats/MathOperations.cs:7
    double err_sb;        // Math.Pow(10, -1*NL/10f) (float exponent)      StrandBiasCalculator.cs:32
    double ln10;          // Math.Log(10.0)
    unsigned long long* totals;  // device int64[4] running totals of the handle, or nullptr
    const double* q_to_p_lut;    // MathOperations.QtoP(q) for integer q in [0, q_to_p_n), evaluated on the host
    int32_t q_to_p_n;
    // Memo of the genotype-quality tail: gq_tail[a * gq_tail_cov + cov] = incomplete_gamma_function(a, target_lod * cov)
    // for a in [1, gq_tail_a), cov in [0, gq_tail_cov), filled once per handle BY THE DEVICE with the very function
    // the call phase would run (build_gq_tail_kernel), so a hit is bit-identical to the evaluation it replaces.
    const double* gq_tail;
    int32_t gq_tail_a, gq_tail_cov;
    int32_t refs_only;   // MNV calling on: SNV candidates come from the read walk, the tile kernels emit Reference records only
    int32_t variants_only;   // a counting launch over loci OUTSIDE the interval set (pisces_hip_set_exact_total_called): no Reference records, the variants' IsCallable only
    // Memo tables of the call phase, all filled once per handle BY THE DEVICE with the very functions they stand in for
    // (build_call_tables_kernel), so a hit is bit-identical to the evaluation it replaces; `tab_cov` columns (coverage) each:
    //   vq_tab[k * tab_cov + cov]  = poisson_qscore(k, cov)                                    1 <= k < vq_tab_k   (NoiseModel.Flat)
    //   sb_tab[k * tab_cov + cov]  = fmax(0, poisson_cdf_sb(k - 1, cov * err_sb))              1 <= k < sb_tab_k   (ChanceVarFreqGreaterThanZero)
    //   sb0_tab[cov]               = pow(1 - err_sb, cov)                                      support == 0, Extended model
    //   gq_cap[a * gq_tail_cov + cov] = the hom-ref / hom-alt genotype q-score of an allele whose variant q-score is max_vq
    const int16_t* vq_tab;
    const double* sb_tab;
    const double* sb0_tab;
    const int16_t* gq_cap;
    int32_t vq_tab_k, sb_tab_k, tab_cov;
    // MNV calling on, split form (surface_flush.inc.h): SNV candidates are the allele counts — as with MNV calling off — everywhere but on
    // the DIRTY loci, where candidates of the read walk decide (an MNV candidate spans the locus and has taken bases out of the SNVs there,
    // an open-ended SNV candidate sits on it, a failed MNV's leftovers landed on it, ...): there the tile kernels emit the Reference
    // record only and the variants come from the candidate kernel.  Bit (position - dirty_first) of dirty_bits; nullptr: no locus is dirty.
    const uint32_t* dirty_bits;
    int32_t dirty_first, dirty_n;
    // The folded counts int32[position - folded_first][6][3] of every locus the launch walks (anchor bins added up, low-quality bases
    // under N: what the call phase itself reads), for the candidate kernel of the same flush; nullptr: not wanted.
    int32_t* folded_out;
    int32_t folded_first, folded_n;
};

__device__ __forceinline__ bool locus_is_dirty(const DeviceParams& P, int pos)
{
    if (!P.dirty_bits) return false;
    const unsigned rel = (unsigned)(pos - P.dirty_first);
    return rel < (unsigned)P.dirty_n && ((P.dirty_bits[rel >> 5] >> (rel & 31u)) & 1u) != 0u;
}

// ------------------------------------------------------------------------------------------
// lib/Pisces.Calculators/stats/Poisson.cs — in-repo regularized incomplete gamma
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ double lanczos_approximation(double p)  // Poisson.cs:106-120
{
    double x = p;
    double tmp = x + 5.5;
    tmp = tmp - (x + 0.5) * log(tmp);
    double ser = 1.000000000190015 + 76.18009172947146 / (p + 1.0);
    ser -= 86.50532032941678 / (p + 2.0);
    ser += 24.01409824083091 / (p + 3.0);
    ser -= 1.231739572450155 / (p + 4.0);
    ser += 0.001208650973866179 / (p + 5.0);
    ser -= 5.395239384953E-06 / (p + 6.0);
    return (log(2.506628274631001 * ser / x) - tmp);
}

__device__ __forceinline__ double stirling_approximation(double n)  // Poisson.cs:125-128
{
    return (0.5 * log(2.0 * 3.14159265358979323846) + (0.5 + n) * log(n) - n);
}

__device__ inline double gamma_continued_fraction(double a, double x, double g)  // Poisson.cs:49-74
{
    const double kFpmin = 1.0E-50, kEpsilon = 1.0E-20;
    double b = x + 1.0 - a;
    double c = 1.0 / kFpmin;
    double d = 1.0 / b;
    double h = d;
    int i;
    for (i = 1; i <= 300; i++) {
        double an = i * (a - i);
        b += 2.0;
        d = an * d + b;
        if (fabs(d) < kFpmin) d = kFpmin;
        c = b + an / c;
        if (fabs(c) < kFpmin) c = kFpmin;
        d = 1.0 / d;
        double del = d * c;
        h *= del;
        if (fabs(del - 1.0) < kEpsilon) break;
    }
    if (i > 300) return -1.0;
    return exp(a * log(x) - x - g) * h;
}

__device__ inline double gamma_series(double a, double x, double g)  // Poisson.cs:76-101
{
    // Same operations in the same order as the reference loop; only the scheduling differs: the quotients
    // x / (a + i) do not depend on the running term, so four of them are formed together (independent FP64
    // division sequences overlap in the pipeline) and then consumed one by one with the reference's
    // convergence test after each.  a is an integer-valued double here, so a + i is exactly the
    // reference's repeatedly incremented `ap`.
    const double kEpsilon = 1.0E-20;
    double retval = -1.0;
    if (x == 0.0) return 0.0;
    if (x < 0.0) return retval;
    double sum = 1.0 / a;
    double del = sum;
    bool done = false;
#if defined(PISCES_SERIES_ILP) && PISCES_SERIES_ILP == 2
    for (int i = 1; i <= 300 && !done; i += 2) {
        const double ap0 = a + (double)i;
        const double q0 = x / ap0, q1 = x / (ap0 + 1.0);
        del *= q0; sum += del;
        if (fabs(del) < fabs(sum) * kEpsilon) { done = true; break; }
        del *= q1; sum += del;
        if (fabs(del) < fabs(sum) * kEpsilon) { done = true; break; }
    }
#else
    for (int i = 1; i <= 300 && !done; i += 4) {
        const double ap0 = a + (double)i;
        const double q0 = x / ap0, q1 = x / (ap0 + 1.0), q2 = x / (ap0 + 2.0), q3 = x / (ap0 + 3.0);
        del *= q0; sum += del;
        if (fabs(del) < fabs(sum) * kEpsilon) { done = true; break; }
        del *= q1; sum += del;
        if (fabs(del) < fabs(sum) * kEpsilon) { done = true; break; }
        del *= q2; sum += del;
        if (fabs(del) < fabs(sum) * kEpsilon) { done = true; break; }
        del *= q3; sum += del;
        if (fabs(del) < fabs(sum) * kEpsilon) { done = true; break; }
    }
#endif
    if (done) retval = sum * exp(a * log(x) - x - g);
    return retval;
}

__device__ inline double incomplete_gamma_function(double a, double x)  // Poisson.cs:34-44
{
    if ((x < 0) || (a <= 0)) return -1.0;
    double g = (a >= 700.0 ? stirling_approximation(a) : lanczos_approximation(a));
    if (x >= a + 1.0) return gamma_continued_fraction(a, x, g);
    if ((g = gamma_series(a, x, g)) < 0) return g;
    return 1.0 - g;
}

__device__ __forceinline__ double poisson_cdf(double num_occurrences, double expected)  // Poisson.cs:26-29
{
    return incomplete_gamma_function((double)(int)(num_occurrences + 1.0), expected);
}

// ln(r) >= ilogb(r) * ln 2 for r >= 1: a logarithm-free lower bound (one v_frexp_exp) for the early-out tests
__device__ __forceinline__ double ln_lower_bound(double r) { return (double)ilogb(r) * 0.6931471805; }

// Same value as poisson_cdf, bit for bit, with an exact early-out for the dominant case of the strand-bias
// statistics (a well-supported allele against the noise rate): when 2x <= a the series branch is taken and
// converges (every ratio x/(a+i) < 1/2, so |del| < |sum|*1e-20 within 70 of the 300 iterations), its sum is
// < 2/a <= 2, and lgamma(a) >= (a-1/2)ln a - a + ln sqrt(2 pi) (also true of the Lanczos / Stirling forms the
// reference uses, to 1e-9).  So if E = a ln x - x - [(a-1/2)ln a - a + 0.9189385] < -40 the series value is
// < 2e^(-39.9) < 2^-54 and the reference's `1.0 - g` rounds to exactly 1.0.
__device__ inline double poisson_cdf_sb(double num_occurrences, double expected)
{
    const double a = (double)(int)(num_occurrences + 1.0), x = expected;
    if (x > 0.0 && a >= 1.0 && 2.0 * x <= a) {
        // E = -a ln(a/x) + a - x + ln(a)/2 - 0.9189..., and ln(a)/2 < 10.75 for any int32 count:
        // one logarithm decides E < -40
        const double r = a / x;
        if (a * (ln_lower_bound(r) - 1.0) + x > 51.0) return 1.0;   // usual case: decided without a logarithm
        if (a * (log(r) - 1.0) + x > 51.0) return 1.0;
    }
    return incomplete_gamma_function(a, x);
}


__device__ __forceinline__ double q_to_p(double q) { return pow(10.0, -1 * q / 10.0); }  // MathOperations.cs:7
__device__ __forceinline__ double q_to_p_int(int q, const DeviceParams& P)
{
    return (P.q_to_p_lut && q >= 0 && q < P.q_to_p_n) ? P.q_to_p_lut[q] : q_to_p((double)q);
}
__device__ __forceinline__ double p_to_q(double p) { return (-10 * log10(p)); }          // MathOperations.cs:12

// ------------------------------------------------------------------------------------------
// MathNet.Numerics 4.5.1 (NuGet dependency of Pisces.Calculators): GammaLn (Lanczos g=10.900511),
// GammaLowerRegularized (Cephes igam/igamc), Poisson CDF / ln PMF — the algorithm behind
// VariantQualityCalculator.cs:36,38,47.
// ------------------------------------------------------------------------------------------
__device__ inline double mathnet_gamma_ln(double z)
{
    const double dk[11] = {2.48574089138753565546e-5,  1.05142378581721974210,    -3.45687097222016235469,
                           4.51227709466894823700,     -2.98285225323576655721,   1.05639711577126713077,
                           -1.95428773191645869583e-1, 1.70970543404441224307e-2, -5.71926117404305781283e-4,
                           4.63399473359905636708e-6,  -2.71994908488607703910e-9};
    const double r = 10.900511;
    const double log_two_sqrt_e_over_pi = 0.6207822376352452223455184457816472122518527279025978;
    const double e = 2.7182818284590452354;
    // arguments here are call counts >= 1, so only the z >= 0.5 branch is reachable
    double s = dk[0];
#if defined(PISCES_GAMMALN_UNROLL) && PISCES_GAMMALN_UNROLL == 1
#pragma unroll 1
#elif defined(PISCES_GAMMALN_UNROLL) && PISCES_GAMMALN_UNROLL == 2
#pragma unroll 2
#else
#pragma unroll
#endif
    for (int i = 1; i <= 10; i++) s += dk[i] / (z + i - 1.0);
    return log(s) + log_two_sqrt_e_over_pi + ((z - 0.5) * log((z - 0.5 + r) / e));
}

__device__ inline double mathnet_factorial_ln(int x)
{
    if (x <= 1) return 0.0;
    if (x < 171) {
        // SpecialFunctions' factorial cache: c[i] = c[i-1] * i in double, then Math.Log
        double c = 1.0;
        for (int i = 2; i <= x; i++) c = c * i;
        return log(c);
    }
    return mathnet_gamma_ln(x + 1.0);
}

__device__ inline double mathnet_gamma_lower_regularized(double a, double x, double* gamma_ln_a)
{
    const double epsilon = 0.000000000000001;
    const double big = 4503599627370496.0;
    const double bigInv = 2.22044604925031308085e-16;
    *gamma_ln_a = __builtin_nan("");
    if (fabs(a) < 1e-15) return (fabs(x) < 1e-15) ? __builtin_nan("") : 1.0;
    if (fabs(x) < 1e-15) return 0.0;

    const double gl = mathnet_gamma_ln(a);
    *gamma_ln_a = gl;
    double ax = (a * log(x)) - x - gl;
    if (ax < -709.78271289338399) return a < x ? 1.0 : 0.0;

    if (x <= 1 || x <= a) {
        double r2 = a, c2 = 1, ans2 = 1;
        do {
            r2 = r2 + 1;
            c2 = c2 * x / r2;
            ans2 += c2;
        } while ((c2 / ans2) > epsilon);
        return exp(ax) * ans2 / a;
    }

    int c = 0;
    double y = 1 - a;
    double z = x + y + 1;
    double p3 = 1, q3 = x, p2 = x + 1, q2 = z * x;
    double ans = p2 / q2;
    double error;
    do {
        c++;
        y += 1;
        z += 2;
        double yc = y * c;
        double p = (p2 * z) - (p3 * yc);
        double q = (q2 * z) - (q3 * yc);
        if (q != 0) {
            double nextans = p / q;
            error = fabs((ans - nextans) / nextans);
            ans = nextans;
        } else {
            error = 1;
        }
        p3 = p2; p2 = p; q3 = q2; q2 = q;
        if (fabs(p) > big) {
            p3 *= bigInv; p2 *= bigInv; q3 *= bigInv; q2 *= bigInv;
        }
    } while (error > epsilon);
    return 1.0 - (exp(ax) * ans);
}

// ------------------------------------------------------------------------------------------
// lib/Pisces.Calculators/VariantQualityCalculator.cs:27-65
// ------------------------------------------------------------------------------------------
// err = MathOperations.QtoP(noise level): P.err_q with NoiseModel.Flat, the allele's own with NoiseModel.Window
struct QParams { int32_t max_vq; double ln10; };
__device__ inline int32_t poisson_qscore_core(int32_t callCount, int32_t coverage, double err, const QParams P)
{
    if ((callCount <= 0) || (coverage <= 0)) return 0;
    double callCountMinusOne = callCount - 1;
    double callCountDouble = callCount;
    double lambda = err * coverage;
    // Exact early-out at the cap (the dominant case: a well-supported allele, MaximumVariantQScore = 100).
    // With k-1 >= e*lambda and k >= 2*lambda both the true tail P(X >= k) <= (e lambda/k)^k and the value the
    // reference's log-space branch uses, pmf(k-1) * k/(2(k-lambda)) <= (e lambda/(k-1))^(k-1), are bounded by
    // B = (e lambda/(k-1))^(k-1).  If B <= 10^-((cap+1)/10) (cap <= 110 so that B <= 7.9e-12 dwarfs the 2^-53
    // cancellation error of `1 - CDF`), either branch yields rawQ >= cap + 0.5 and the clamp returns cap.
    if (P.max_vq <= 110 && callCount >= 3 && callCountDouble >= 2.0 * lambda) {
        const double km1 = callCountMinusOne;
        const double need = ((double)P.max_vq + 1.0) * 0.23025850929940458 + 1e-3;
        const double r = km1 / lambda;
        if (km1 * (ln_lower_bound(r) - 1.0) >= need) return P.max_vq;   // usual case: decided without a logarithm
        if (km1 * (log(r) - 1.0) >= need) return P.max_vq;
    }
    // Poisson.CumulativeDistribution(k-1) = 1 - GammaLowerRegularized(k, lambda)
    double gamma_ln_k;   // GammaLn(k) from the CDF; FactorialLn(k-1) = GammaLn(k) is the same evaluation
    double pValue = 1 - (1.0 - mathnet_gamma_lower_regularized(callCountMinusOne + 1, lambda, &gamma_ln_k));
    double rawQ;
    if (pValue > 0) {
        rawQ = p_to_q(pValue);
    } else {
        int k = (int)callCountMinusOne;
        const double fl = (k >= 171 && gamma_ln_k == gamma_ln_k) ? gamma_ln_k : mathnet_factorial_ln(k);
        double A = -lambda + (k * log(lambda)) - fl;  // Poisson.ProbabilityLn
        double correction = (callCountDouble - lambda) / callCountDouble;
        rawQ = -10.0 * (A - log(2.0 * correction)) / P.ln10;
    }
    double qScore = fmin((double)P.max_vq, rawQ);
    qScore = fmax(qScore, 0.0);
    return (int32_t)rint(qScore);  // Math.Round: ties to even
}
__device__ inline int32_t poisson_qscore_e(int32_t callCount, int32_t coverage, double err, const DeviceParams& P)
{
    const QParams q = {P.max_vq, P.ln10};
    return poisson_qscore_core(callCount, coverage, err, q);
}
__device__ inline int32_t poisson_qscore(int32_t callCount, int32_t coverage, const DeviceParams& P)
{
    return poisson_qscore_e(callCount, coverage, P.err_q, P);
}
// ln(a / x) >= (ilogb(a) - ilogb(x) - 1) ln 2 for a >= x > 0: the logarithm-free bound of ln_lower_bound without the division
__device__ __forceinline__ double ln_ratio_lower_bound(double a, double x) { return (double)(ilogb(a) - ilogb(x) - 1) * 0.6931471805; }

// poisson_qscore for the streaming-rate kernel (NoiseModel.Flat): the memo table; beyond the table the cap early-out of
// poisson_qscore_core in a division-free form (a weaker bound of the same quantity, so still a proof).  false = neither
// decides: the caller evaluates the long way (a cold path shared by every miss of the tile).
__device__ __forceinline__ bool poisson_qscore_try(int32_t callCount, int32_t coverage, const DeviceParams& P, int32_t& vq)
{
    if ((callCount <= 0) || (coverage <= 0)) { vq = 0; return true; }
    if (P.vq_tab && callCount < P.vq_tab_k && coverage < P.tab_cov) {
        vq = P.vq_tab[(uint32_t)callCount * (uint32_t)P.tab_cov + (uint32_t)coverage];   // (tables hold < 2^31 entries: 32-bit index arithmetic)
        return true;
    }
    const double lambda = P.err_q * coverage;
    if (P.max_vq <= 110 && callCount >= 3 && (double)callCount >= 2.0 * lambda) {
        const double km1 = callCount - 1;
        const double need = ((double)P.max_vq + 1.0) * 0.23025850929940458 + 1e-3;
        if (km1 * (ln_ratio_lower_bound(km1, lambda) - 1.0) >= need) { vq = P.max_vq; return true; }
    }
    return false;
}

// NoiseModel.Window (AlleleCaller.cs:215-218): the allele's noise level (int)PtoQ(SumOfBaseQuality / TotalCoverage), or kNoLevel when
// the mean error is not a positive finite number (the C# cast then gives int.MinValue and the reference's arithmetic ends in a
// q-score of 0; see oracle/pisces_oracle.c), and QtoP of it (negative for kNoLevel).
constexpr int32_t kNoLevel = (int32_t)0x80000000;
__device__ inline int32_t window_level(double sumOfBaseQuality, int totalCoverage)
{
    if (totalCoverage == 0) return kNoLevel;
    const double mean = sumOfBaseQuality / (double)totalCoverage;
    if (!(mean > 0.0) || isinf(mean)) return kNoLevel;
    return (int32_t)p_to_q(mean);
}
__device__ inline double window_err_of_level(int32_t level, const DeviceParams& P) { return level == kNoLevel ? -1.0 : q_to_p_int(level, P); }
__device__ inline double window_err(double sumOfBaseQuality, int totalCoverage, const DeviceParams& P)
{
    return window_err_of_level(window_level(sumOfBaseQuality, totalCoverage), P);
}
// CalledAllele.NoiseLevelApplied as the 64-byte record carries it
__device__ __forceinline__ int16_t noise_level_field(int32_t level) { return level == kNoLevel ? (int16_t)-32768 : (int16_t)level; }

// ------------------------------------------------------------------------------------------
// lib/Pisces.Calculators/StrandBiasCalculator.cs:21-231 (Poisson and Extended models)
// ------------------------------------------------------------------------------------------
struct SbStats { double var_gt_zero, false_pos, coverage, support; };

// MathNet.Numerics 4.5.1 SpecialFunctions.BetaRegularized (continued fraction with the symmetry transformation, eps = 2^-53, at most
// 50000 rounds) and Binomial(p, n).CumulativeDistribution(x) = BetaRegularized(n - k, k + 1, 1 - p): the Diploid strand-bias model
// (StrandBiasCalculator.cs:150-173).  Kept out of line: only germline runs reach it.
__device__ __noinline__ double mathnet_beta_regularized(double a, double b, double x)
{
    const double bt = (x == 0.0 || x == 1.0) ? 0.0
                      : exp(mathnet_gamma_ln(a + b) - mathnet_gamma_ln(a) - mathnet_gamma_ln(b) + (a * log(x)) + (b * log(1.0 - x)));
    const bool symmetryTransformation = x >= (a + 1.0) / (a + b + 2.0);
    const double eps = 1.1102230246251565e-16;
    const double fpmin = 4.9406564584124654e-324 / eps;
    if (symmetryTransformation) { x = 1.0 - x; const double swap = a; a = b; b = swap; }
    const double qab = a + b, qap = a + 1.0, qam = a - 1.0;
    double c = 1.0, d = 1.0 - (qab * x / qap);
    if (fabs(d) < fpmin) d = fpmin;
    d = 1.0 / d;
    double h = d;
    for (int m = 1, m2 = 2; m <= 50000; m++, m2 += 2) {
        double aa = m * (b - m) * x / ((qam + m2) * (a + m2));
        d = 1.0 + (aa * d); if (fabs(d) < fpmin) d = fpmin;
        c = 1.0 + (aa / c); if (fabs(c) < fpmin) c = fpmin;
        d = 1.0 / d;
        h *= d * c;
        aa = -(a + m) * (qab + m) * x / ((a + m2) * (qap + m2));
        d = 1.0 + (aa * d); if (fabs(d) < fpmin) d = fpmin;
        c = 1.0 + (aa / c); if (fabs(c) < fpmin) c = fpmin;
        d = 1.0 / d;
        const double del = d * c;
        h *= del;
        if (fabs(del - 1.0) <= eps) break;
    }
    return symmetryTransformation ? 1.0 - (bt * h / a) : bt * h / a;
}

// kDiploidOk: the kernels of the streaming-rate path instantiate the Poisson / Extended models only; the Diploid model
// (PopulateDiploidStats :150-173) is compiled into the counts-fed and candidate kernels, where germline configurations are routed.
template <bool kDiploidOk = false>
__device__ inline SbStats sb_create_stats(double support, double coverage, double noiseFreq, double minVariantFreq, int model)
{
    // CreateStats :137-148 — minDetectableSNP = noiseFreq for every non-Diploid model;
    // ChanceFalseNeg (:213) does not enter the bias score and is not part of the record.
    SbStats st;
    st.support = support;
    st.coverage = coverage;
    const double minDetectableSNP = (kDiploidOk && model == PISCES_SB_DIPLOID) ? minVariantFreq : noiseFreq;
    if (support == 0) {
        if (model == PISCES_SB_POISSON) {
            st.false_pos = 1;
            st.var_gt_zero = 0;
        } else {
            st.var_gt_zero = pow(1 - minDetectableSNP, coverage);
            st.false_pos = 1 - st.var_gt_zero;
        }
    } else if (kDiploidOk && model == PISCES_SB_DIPLOID) {
        const double frequency = coverage == 0 ? 0.0 : support / coverage;
        if (frequency >= minDetectableSNP) {
            st.var_gt_zero = 1;
            st.false_pos = 0;
        } else {
            const int n = (int)coverage;
            const double k = floor(support);
            const double cdf = support < 0.0 ? 0.0 : support > n ? 1.0 : mathnet_beta_regularized(n - k, k + 1, 1 - minDetectableSNP);
            st.var_gt_zero = fmax(cdf, 0.0);   // ChanceVarFreqGreaterThanZero = ChanceFalseNeg
            st.false_pos = fmax(0.0, 1 - poisson_cdf_sb(support, coverage * 0.1));
        }
    } else {
        st.var_gt_zero = fmax(0.0, poisson_cdf_sb(support - 1, coverage * noiseFreq));
        st.false_pos = fmax(0.0, 1 - st.var_gt_zero);
    }
    return st;
}

struct SbResult { double bias_score; int acceptable, var_both, cov_both; };

// which: 0 overall (F+R+S), 1 forward (F + S/2), 2 reverse (R + S/2); the stitched halves use integer division (:36-41)
template <bool kDiploidOk = false>
__device__ __forceinline__ SbStats sb_stats_of(int which, const int32_t cov[3], const int32_t sup[3], const DeviceParams& P)
{
    const int s = which == 0 ? sup[0] + sup[1] + sup[2] : (which == 1 ? sup[0] + sup[2] / 2 : sup[1] + sup[2] / 2);
    const int c = which == 0 ? cov[0] + cov[1] + cov[2] : (which == 1 ? cov[0] + cov[2] / 2 : cov[1] + cov[2] / 2);
    return sb_create_stats<kDiploidOk>((double)s, (double)c, P.err_sb, (double)P.min_freq, P.sb_model);
}

// AssignBiasScore (:89-105) + the both-strands rules (:57-69) from the three statistics
__device__ __forceinline__ SbResult sb_combine(const SbStats& overall, const SbStats& fwd, const SbStats& rev, const DeviceParams& P)
{
    // StitchedStats (:55-56) feed only the optional strand-bias report file, not the score.
    double forwardBias = (fwd.var_gt_zero * rev.false_pos) / overall.var_gt_zero;
    double reverseBias = (rev.var_gt_zero * fwd.false_pos) / overall.var_gt_zero;
    if (overall.var_gt_zero == 0) {
        forwardBias = 1;
        reverseBias = 1;
    }
    SbResult r;
    r.bias_score = forwardBias > reverseBias ? forwardBias : reverseBias;
    r.cov_both = (fwd.coverage > 0) && (rev.coverage > 0);
    r.var_both = (fwd.support > 0) && (rev.support > 0);
    if (!r.cov_both) r.bias_score = 0;
    r.acceptable = r.bias_score < P.sb_threshold;
    return r;
}

template <bool kDiploidOk = false>
__device__ inline SbResult strand_bias(const int32_t cov[3], const int32_t sup[3], const DeviceParams& P)
{
    // the three evaluations are independent; interleaving them costs ~3x the registers of one (scratch spills at the
    // 128-VGPR forms of the kernel) and buys nothing on the usual early-out path, so keep them apart
    const SbStats overall = sb_stats_of<kDiploidOk>(0, cov, sup, P);
    __builtin_amdgcn_sched_barrier(0);
    const SbStats fwd = sb_stats_of<kDiploidOk>(1, cov, sup, P);
    __builtin_amdgcn_sched_barrier(0);
    const SbStats rev = sb_stats_of<kDiploidOk>(2, cov, sup, P);
    __builtin_amdgcn_sched_barrier(0);
    return sb_combine(overall, fwd, rev, P);
}

// ------------------------------------------------------------------------------------------
// CalledAllele.Frequency (CalledAllele.cs:49-52) and lib/Pisces.Genotyping/Somatic
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float frequency_f(int32_t support, int32_t coverage)
{
    if (coverage == 0) return 0.0f;
    float f = (float)support / (float)coverage;
    return f < 1.0f ? f : 1.0f;
}

__device__ inline int32_t somatic_genotype(bool isReference, int32_t cov, int32_t support, int32_t refSupport,
                                           const DeviceParams& P)  // SomaticGenotyper.cs:65-100
{
    if (cov < P.min_cov) return isReference ? PISCES_GT_REF_LIKE_NOCALL : PISCES_GT_ALT_LIKE_NOCALL;
    float freq = frequency_f(support, cov);
    if (!isReference) {
        if (frequency_f(refSupport, cov) < P.gt_min_freq) {
            if ((1 - freq) > P.gt_min_freq) return PISCES_GT_ALT_AND_NOCALL;
            return PISCES_GT_HOM_ALT;
        }
        return PISCES_GT_HET_ALT_REF;
    }
    if (freq < P.gt_min_freq) return PISCES_GT_REF_LIKE_NOCALL;
    if ((1 - freq) > P.gt_min_freq) return PISCES_GT_REF_AND_NOCALL;
    return PISCES_GT_HOM_REF;
}

// The memo miss path (coverage or count beyond the table, or no table): the same evaluation, kept out of line so that
// its registers are not part of every caller's budget (it was the VGPR peak of the whole call phase).
__device__ __noinline__ double poisson_cdf_out_of_line(double num_occurrences, double expected)
{
    return poisson_cdf(num_occurrences, expected);
}

// The Poisson tail of SomaticGenotypeQualityCalculator.cs:30-41 for a hom-ref / hom-alt call: a pure function of
// (coverage, support, targetLODFrequency).  `skip` = the reference returns MinGenotypeQScore before using it.
struct GqTail { double p2; bool used, floor; };
__device__ __forceinline__ GqTail somatic_gq_tail(int32_t genotype, int32_t cov, int32_t support, const DeviceParams& P)
{
    GqTail t = {0.0, false, false};
    if (cov == 0) return t;
    if ((genotype == PISCES_GT_HOM_REF) || (genotype == PISCES_GT_HOM_ALT)) {
        const float nonAlleleObservationsF = (1.0f - frequency_f(support, cov)) * (float)cov;
        const float expectedNonAllelObservationsF = P.target_lod * (float)cov;
        t.used = true;
        if (nonAlleleObservationsF >= expectedNonAllelObservationsF) { t.floor = true; return t; }
        const int ai = (int)((double)nonAlleleObservationsF + 1.0);   // Poisson.Cdf's (int)(numOccurrences + 1)
        if (P.gq_tail && ai >= 1 && ai < P.gq_tail_a && cov < P.gq_tail_cov)
            t.p2 = P.gq_tail[(size_t)ai * (size_t)P.gq_tail_cov + (size_t)cov];
        else
            t.p2 = poisson_cdf_out_of_line(nonAlleleObservationsF, expectedNonAllelObservationsF);
    }
    return t;
}

__device__ __forceinline__ int32_t somatic_gq_finish(int32_t genotype, int32_t variantQ, int32_t cov, const GqTail& t,
                                                     const DeviceParams& P)  // SomaticGenotypeQualityCalculator.cs:10-48
{
    double rawQ = variantQ;
    bool noCall = (genotype == PISCES_GT_ALT12_LIKE_NOCALL || genotype == PISCES_GT_ALT_LIKE_NOCALL ||
                   genotype == PISCES_GT_REF_LIKE_NOCALL);
    if ((cov == 0) || noCall) return P.min_gq;
    if (t.used) {
        if (t.floor) return P.min_gq;
        double p1 = q_to_p_int(variantQ, P);
        rawQ = p_to_q(p1 + t.p2);
    }
    double qScore = fmin((double)P.max_gq, rawQ);
    qScore = fmax(qScore, (double)P.min_gq);
    return (int32_t)rint(qScore);
}

__device__ inline int32_t somatic_gq(int32_t genotype, int32_t variantQ, int32_t cov, int32_t support,
                                     const DeviceParams& P)
{
    const GqTail t = somatic_gq_tail(genotype, cov, support, P);
    return somatic_gq_finish(genotype, variantQ, cov, t, P);
}

// ------------------------------------------------------------------------------------------
// Table-first forms of the strand-bias statistics and the genotype q-score for the streaming-rate kernel.  Every table is filled
// by the device with the function it stands in for (build_*_kernel in kernels.hip.h), keyed by the integer counts the function is
// called with.  Each returns false when neither a table nor a provable early-out decides (count or coverage beyond the table):
// the caller then takes the allele through the long evaluation, a cold path shared by every miss of the tile.
// ------------------------------------------------------------------------------------------
// CreateStats / PopulateStats (:137-231) for the Poisson and Extended models, integer support and coverage
__device__ __forceinline__ bool sb_stats_try(int32_t support, int32_t coverage, const DeviceParams& P, SbStats& st)
{
    st.support = (double)support;
    st.coverage = (double)coverage;
    if (support == 0) {
        if (P.sb_model == PISCES_SB_POISSON) {
            st.false_pos = 1;
            st.var_gt_zero = 0;
            return true;
        }
        if (!(P.sb0_tab && coverage >= 0 && coverage < P.tab_cov)) return false;
        st.var_gt_zero = P.sb0_tab[coverage];
        st.false_pos = 1 - st.var_gt_zero;
        return true;
    }
    if (P.sb_tab && support > 0 && support < P.sb_tab_k && coverage >= 0 && coverage < P.tab_cov) {
        st.var_gt_zero = P.sb_tab[(uint32_t)support * (uint32_t)P.tab_cov + (uint32_t)coverage];
    } else {
        // poisson_cdf_sb's proof of "exactly 1.0" with ln(a / x) bounded from below without the division
        const double a = (double)support, x = (double)coverage * P.err_sb;
        if (!(x > 0.0 && 2.0 * x <= a && a * (ln_ratio_lower_bound(a, x) - 1.0) + x > 51.0)) return false;
        st.var_gt_zero = 1.0;
    }
    st.false_pos = fmax(0.0, 1 - st.var_gt_zero);
    return true;
}

__device__ __forceinline__ bool strand_bias_try(const int32_t cov[3], const int32_t sup[3], const DeviceParams& P, double& bias_score,
                                                int& acceptable, int& var_both, int& cov_both)
{
    const int s2 = sup[2] / 2, c2 = cov[2] / 2;   // the stitched halves use integer division (:36-41)
    const int sf = sup[0] + s2, sr = sup[1] + s2, cf = cov[0] + c2, cr = cov[1] + c2;
    SbStats overall, fwd, rev;
    const bool ok0 = sb_stats_try(sup[0] + sup[1] + sup[2], cov[0] + cov[1] + cov[2], P, overall);
    const bool ok1 = sb_stats_try(sf, cf, P, fwd);
    const bool ok2 = sb_stats_try(sr, cr, P, rev);
    // AssignBiasScore (:89-105) + the both-strands rules (:57-69), as sb_combine; with both products exactly +0 the quotients are
    // (x * 0) / y = 0 for finite x >= 0, y > 0 — no division needed
    double forwardBias = 0.0, reverseBias = 0.0;
    if (!(fwd.false_pos == 0.0 && rev.false_pos == 0.0 && overall.var_gt_zero > 0.0)) {
        forwardBias = (fwd.var_gt_zero * rev.false_pos) / overall.var_gt_zero;
        reverseBias = (rev.var_gt_zero * fwd.false_pos) / overall.var_gt_zero;
        if (overall.var_gt_zero == 0) {
            forwardBias = 1;
            reverseBias = 1;
        }
    }
    double score = forwardBias > reverseBias ? forwardBias : reverseBias;
    cov_both = (cf > 0) && (cr > 0);
    var_both = (sf > 0) && (sr > 0);
    if (!cov_both) score = 0;
    bias_score = score;
    acceptable = score < P.sb_threshold;
    return ok0 && ok1 && ok2;
}

// SomaticGenotypeQualityCalculator.Compute (:10-48).  somatic_gq_index = the memo index of a hom-ref / hom-alt call (or -1), so that
// a caller can issue the load of P.gq_cap[index] ahead of time.
__device__ __forceinline__ int32_t somatic_gq_index(int32_t genotype, int32_t cov, int32_t support, const DeviceParams& P)
{
    if (cov == 0 || !((genotype == PISCES_GT_HOM_REF) || (genotype == PISCES_GT_HOM_ALT))) return -1;
    const float nonAlleleObservationsF = (1.0f - frequency_f(support, cov)) * (float)cov;
    const float expectedNonAllelObservationsF = P.target_lod * (float)cov;
    if (nonAlleleObservationsF >= expectedNonAllelObservationsF) return -1;
    const int ai = (int)((double)nonAlleleObservationsF + 1.0);   // Poisson.Cdf's (int)(numOccurrences + 1)
    if (ai >= 1 && ai < P.gq_tail_a && cov < P.gq_tail_cov) return ai * P.gq_tail_cov + cov;
    return -1;
}
__device__ __forceinline__ bool somatic_gq_try(int32_t genotype, int32_t variantQ, int32_t cov, int32_t support, const DeviceParams& P,
                                               int32_t gq_index, int32_t gq_cap_value /* P.gq_cap[gq_index] when gq_index >= 0 */, int32_t& gq)
{
    const bool noCall = (genotype == PISCES_GT_ALT12_LIKE_NOCALL || genotype == PISCES_GT_ALT_LIKE_NOCALL || genotype == PISCES_GT_REF_LIKE_NOCALL);
    if ((cov == 0) || noCall) { gq = P.min_gq; return true; }
    if ((genotype == PISCES_GT_HOM_REF) || (genotype == PISCES_GT_HOM_ALT)) {
        if (gq_index >= 0 && variantQ == P.max_vq && P.gq_cap) { gq = gq_cap_value; return true; }
        const float nonAlleleObservationsF = (1.0f - frequency_f(support, cov)) * (float)cov;
        const float expectedNonAllelObservationsF = P.target_lod * (float)cov;
        if (nonAlleleObservationsF >= expectedNonAllelObservationsF) { gq = P.min_gq; return true; }
        return false;   // a logarithm of QtoP(variantQ) + tail: the long way
    }
    double qScore = fmin((double)P.max_gq, (double)variantQ);
    qScore = fmax(qScore, (double)P.min_gq);
    gq = (int32_t)rint(qScore);
    return true;
}

// ------------------------------------------------------------------------------------------
// The same table-first forms with every table entry an allele can need requested AT ONCE (one memory round trip instead of a
// chain of five to eight: in the call phase of the streaming kernel a dependent global load queues behind the CU's streaming loads
// and costs a microsecond).  The addresses depend on the integer counts only; an entry that is out of range is requested at
// index 0 and ignored.  The decisions and the values are those of poisson_qscore_try / sb_stats_try / somatic_gq_try.
// ------------------------------------------------------------------------------------------
struct AlleleTables {
    int32_t vq;            // P.vq_tab[support][coverage]
    double sb[3];          // overall / forward / reverse: P.sb_tab[support][coverage], or P.sb0_tab[coverage] when support == 0
    int32_t gq_cap;        // P.gq_cap[gq_idx]
    int32_t gq_idx;        // somatic_gq_index (-1 = none)
    uint32_t have;         // bit 0: vq, bits 1..3: sb[0..2]
};
__device__ __forceinline__ bool tables_complete(const DeviceParams& P) { return P.vq_tab && P.sb_tab && P.sb0_tab && P.gq_cap; }

__device__ __forceinline__ AlleleTables request_allele_tables(bool isRef, int32_t support, int32_t total, int32_t refsup, const int32_t cov[3],
                                                              const int32_t sup[3], const DeviceParams& P)
{
    AlleleTables t;
    const uint32_t tab_cov = (uint32_t)P.tab_cov;
    const bool vq_in = support > 0 && total > 0 && support < P.vq_tab_k && total < P.tab_cov;
    const int16_t vq_v = P.vq_tab[vq_in ? (uint32_t)support * tab_cov + (uint32_t)total : 0u];
    const int s2 = sup[2] / 2, c2 = cov[2] / 2;
    const int ss[3] = {sup[0] + sup[1] + sup[2], sup[0] + s2, sup[1] + s2};
    const int cc[3] = {cov[0] + cov[1] + cov[2], cov[0] + c2, cov[1] + c2};
    double sb_v[3];
    uint32_t have = vq_in ? 1u : 0u;
#pragma unroll
    for (int w = 0; w < 3; w++) {
        const bool zero = ss[w] == 0;
        const bool in = zero ? (P.sb_model != PISCES_SB_POISSON && cc[w] >= 0 && cc[w] < P.tab_cov)
                             : (ss[w] > 0 && ss[w] < P.sb_tab_k && cc[w] >= 0 && cc[w] < P.tab_cov);
        const double* base = zero ? P.sb0_tab : P.sb_tab;
        const uint32_t idx = in ? (zero ? (uint32_t)cc[w] : (uint32_t)ss[w] * tab_cov + (uint32_t)cc[w]) : 0u;
        sb_v[w] = base[idx];
        have |= in ? (2u << w) : 0u;
    }
    t.gq_idx = somatic_gq_index(somatic_genotype(isRef, total, support, refsup, P), total, support, P);
    const int16_t gq_v = P.gq_cap[t.gq_idx >= 0 ? t.gq_idx : 0];
    t.vq = vq_v;
    t.sb[0] = sb_v[0]; t.sb[1] = sb_v[1]; t.sb[2] = sb_v[2];
    t.gq_cap = gq_v;
    t.have = have;
    return t;
}

__device__ __forceinline__ bool poisson_qscore_try(int32_t callCount, int32_t coverage, const DeviceParams& P, const AlleleTables& t, int32_t& vq)
{
    if ((callCount <= 0) || (coverage <= 0)) { vq = 0; return true; }
    if (t.have & 1u) { vq = t.vq; return true; }
    const double lambda = P.err_q * coverage;
    if (P.max_vq <= 110 && callCount >= 3 && (double)callCount >= 2.0 * lambda) {
        const double km1 = callCount - 1;
        const double need = ((double)P.max_vq + 1.0) * 0.23025850929940458 + 1e-3;
        if (km1 * (ln_ratio_lower_bound(km1, lambda) - 1.0) >= need) { vq = P.max_vq; return true; }
    }
    return false;
}

__device__ __forceinline__ bool sb_stats_try(int32_t support, int32_t coverage, const DeviceParams& P, bool have, double value, SbStats& st)
{
    st.support = (double)support;
    st.coverage = (double)coverage;
    if (support == 0) {
        if (P.sb_model == PISCES_SB_POISSON) {
            st.false_pos = 1;
            st.var_gt_zero = 0;
            return true;
        }
        if (!have) return false;
        st.var_gt_zero = value;
        st.false_pos = 1 - st.var_gt_zero;
        return true;
    }
    if (have) {
        st.var_gt_zero = value;
    } else {
        const double a = (double)support, x = (double)coverage * P.err_sb;
        if (!(x > 0.0 && 2.0 * x <= a && a * (ln_ratio_lower_bound(a, x) - 1.0) + x > 51.0)) return false;
        st.var_gt_zero = 1.0;
    }
    st.false_pos = fmax(0.0, 1 - st.var_gt_zero);
    return true;
}

__device__ __forceinline__ bool strand_bias_try(const int32_t cov[3], const int32_t sup[3], const DeviceParams& P, const AlleleTables& t,
                                                double& bias_score, int& acceptable, int& var_both, int& cov_both)
{
    const int s2 = sup[2] / 2, c2 = cov[2] / 2;   // the stitched halves use integer division (:36-41)
    const int sf = sup[0] + s2, sr = sup[1] + s2, cf = cov[0] + c2, cr = cov[1] + c2;
    SbStats overall, fwd, rev;
    const bool ok0 = sb_stats_try(sup[0] + sup[1] + sup[2], cov[0] + cov[1] + cov[2], P, (t.have & 2u) != 0, t.sb[0], overall);
    const bool ok1 = sb_stats_try(sf, cf, P, (t.have & 4u) != 0, t.sb[1], fwd);
    const bool ok2 = sb_stats_try(sr, cr, P, (t.have & 8u) != 0, t.sb[2], rev);
    double forwardBias = 0.0, reverseBias = 0.0;
    if (!(fwd.false_pos == 0.0 && rev.false_pos == 0.0 && overall.var_gt_zero > 0.0)) {
        forwardBias = (fwd.var_gt_zero * rev.false_pos) / overall.var_gt_zero;
        reverseBias = (rev.var_gt_zero * fwd.false_pos) / overall.var_gt_zero;
        if (overall.var_gt_zero == 0) {
            forwardBias = 1;
            reverseBias = 1;
        }
    }
    double score = forwardBias > reverseBias ? forwardBias : reverseBias;
    cov_both = (cf > 0) && (cr > 0);
    var_both = (sf > 0) && (sr > 0);
    if (!cov_both) score = 0;
    bias_score = score;
    acceptable = score < P.sb_threshold;
    return ok0 && ok1 && ok2;
}

// Out-of-line leaves of the wave kernel's cold path (an allele beyond the memo tables): the evaluations the tables were filled
// with, as calls, so that neither their registers nor their code weigh on the call phase proper.
__device__ __noinline__ int32_t poisson_qscore_out_of_line(int32_t callCount, int32_t coverage, double err, int32_t max_vq, double ln10)
{
    const QParams q = {max_vq, ln10};
    return poisson_qscore_core(callCount, coverage, err, q);
}
struct SbPair { double var_gt_zero, false_pos; };
__device__ __noinline__ SbPair sb_stats_out_of_line(double support, double coverage, double noiseFreq, int model)
{
    const SbStats st = sb_create_stats<false>(support, coverage, noiseFreq, 0.0, model);
    SbPair p = {st.var_gt_zero, st.false_pos};
    return p;
}
__device__ inline SbResult strand_bias_out_of_line(const int32_t cov[3], const int32_t sup[3], const DeviceParams& P)
{
    const int s2 = sup[2] / 2, c2 = cov[2] / 2;
    SbStats st[3];
#pragma unroll 1
    for (int which = 0; which < 3; which++) {
        const int s = which == 0 ? sup[0] + sup[1] + sup[2] : (which == 1 ? sup[0] + s2 : sup[1] + s2);
        const int c = which == 0 ? cov[0] + cov[1] + cov[2] : (which == 1 ? cov[0] + c2 : cov[1] + c2);
        const SbPair p = sb_stats_out_of_line((double)s, (double)c, P.err_sb, P.sb_model);
        st[which].var_gt_zero = p.var_gt_zero;
        st[which].false_pos = p.false_pos;
        st[which].support = (double)s;
        st[which].coverage = (double)c;
    }
    return sb_combine(st[0], st[1], st[2], P);
}

// ------------------------------------------------------------------------------------------
// RMxNCalculator (lib/Pisces.Calculators/RMxNCalculator.cs:19-131) for a single-base SNV: every
// prefix/suffix "bookend" of a one-base string is that base, so each component is the length of
// the homopolymer run found by ratcheting back from `start` and reading forward.
// ref[i] is position ref_start_position + i (so string index s <-> i = s + 1 - ref_start_position).
// ------------------------------------------------------------------------------------------
__device__ inline int rmxn_run(const uint8_t* ref, int64_t lo, int64_t hi, int64_t start, uint8_t base)
{
    // indices are 0-based string indices of the chromosome clipped to the resident window [lo, hi)
    int64_t back = start;
    while (back - 1 >= lo && ref[back - 1 - lo] == base) back--;   // ComputeRMxNLengthForIndel :66-76
    int n = 0;
    int64_t cur = back;
    while (cur < hi && ref[cur - lo] == base) { n++; cur++; }      // :79-89
    return n;
}

__device__ inline bool rmxn_should_filter_snv(const uint8_t* ref, int64_t win_lo, int64_t win_hi, int32_t position,
                                              uint8_t refBase, uint8_t altBase, float freq, const DeviceParams& P)
{
    if (P.rmxn_max_len < 1) return false;   // a one-base unit needs maxRepeatUnitLength >= 1
    if (freq >= P.rmxn_freq_limit) return false;
    // string index of `position` is position-1
    int c1 = rmxn_run(ref, win_lo, win_hi, (int64_t)position - 1, refBase);
    int i1 = rmxn_run(ref, win_lo, win_hi, (int64_t)position, altBase);      // ReferencePosition + refLen - 1
    int i2 = rmxn_run(ref, win_lo, win_hi, (int64_t)position - 1, altBase);
    int c2 = i1 > i2 ? i1 : i2;
    return (c1 < c2 ? c1 : c2) >= P.rmxn_min_rep;
}

// The same scan over a window of the reference staged in LDS: win[j] is string index win_lo + j, j in [0, n).
// A run clipped by the window edge is at least kRefMargin long, so comparing against rmxn_min_rep <= kRefMargin
// gives the reference's decision.
__device__ inline int rmxn_run_lds(const uint8_t* win, int n, int start, uint8_t base)
{
    int back = start;
    while (back - 1 >= 0 && win[back - 1] == base) back--;
    int cnt = 0;
    int cur = back;
    while (cur < n && win[cur] == base) { cnt++; cur++; }
    return cnt;
}

__device__ inline bool rmxn_should_filter_snv_lds(const uint8_t* win, int n, int idx /* window index of the SNV */,
                                                  uint8_t refBase, uint8_t altBase, float freq, const DeviceParams& P)
{
    if (P.rmxn_max_len < 1) return false;
    if (freq >= P.rmxn_freq_limit) return false;
    // min(c1, c2) >= m  <=>  c1 >= m and c2 >= m: the reference-base run decides first, and almost always at once.  Its eight
    // neighbours are read together (one LDS round trip instead of a dependent chain); a run that stays inside them is exact.
    if (idx >= 4 && idx + 4 < n) {
        const bool l1 = win[idx - 1] == refBase, l2 = win[idx - 2] == refBase, l3 = win[idx - 3] == refBase, l4 = win[idx - 4] == refBase;
        const bool r1 = win[idx + 1] == refBase, r2 = win[idx + 2] == refBase, r3 = win[idx + 3] == refBase, r4 = win[idx + 4] == refBase;
        const int left = !l1 ? 0 : !l2 ? 1 : !l3 ? 2 : !l4 ? 3 : 4;
        const int right = !r1 ? 0 : !r2 ? 1 : !r3 ? 2 : !r4 ? 3 : 4;
        if (left < 4 && right < 4 && 1 + left + right < P.rmxn_min_rep) return false;
    }
    int c1 = rmxn_run_lds(win, n, idx, refBase);
    if (c1 < P.rmxn_min_rep) return false;
    int i1 = rmxn_run_lds(win, n, idx + 1, altBase);
    int i2 = rmxn_run_lds(win, n, idx, altBase);
    int c2 = i1 > i2 ? i1 : i2;
    return (c1 < c2 ? c1 : c2) >= P.rmxn_min_rep;
}

}  // namespace pisces
