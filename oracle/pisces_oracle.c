/*
 * pisces_oracle.c — TEST INFRASTRUCTURE ONLY (see pisces_oracle.h).
 * CPU restatement of the Pisces pileup-and-likelihood path.  Compile with
 * -ffp-contract=off: the reference is C# double/float arithmetic without fused ops.
 * Citations are relative to /root/reference/src.
 */
#include "pisces_oracle.h"

#include <ctype.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* =====================================================================================
 * lib/Pisces.Calculators/stats/Poisson.cs  (in-repo incomplete gamma, Numerical-Recipes style)
 * ===================================================================================== */
static const double kEpsilon = 1.0E-20;      /* Poisson.cs:15 */
static const double kFpmin = 1.0E-50;        /* :16 */
static const double kLanczCutoff = 700.0;    /* :17 */
#define kItmax 300                            /* :18 */

/* Poisson.cs:106-120 */
static double lanczos_approximation(double p)
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

/* Poisson.cs:125-128 */
static double stirling_approximation(double n)
{
    return (0.5 * log(2.0 * M_PI) + (0.5 + n) * log(n) - n);
}

/* Poisson.cs:49-74 */
static double gamma_continued_fraction(double a, double x, double g)
{
    double b = x + 1.0 - a;
    double c = 1.0 / kFpmin;
    double d = 1.0 / b;
    double h = d;
    int i;
    for (i = 1; i <= kItmax; i++) {
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
    if (i > kItmax) return -1.0;
    return exp(a * log(x) - x - g) * h;
}

/* Poisson.cs:76-101 */
static double gamma_series(double a, double x, double g)
{
    double retval = -1.0;
    if (x == 0.0) return 0.0;
    if (x < 0.0) return retval;
    double ap = a;
    double sum = 1.0 / a;
    double del = sum;
    for (int i = 1; i <= kItmax; i++) {
        ap += 1.0;
        del *= x / ap;
        sum += del;
        if (fabs(del) < fabs(sum) * kEpsilon) {
            retval = sum * exp(a * log(x) - x - g);
            break;
        }
    }
    return retval;
}

/* Poisson.cs:34-44 */
static double incomplete_gamma_function(double a, double x)
{
    if ((x < 0) || (a <= 0)) return -1.0;
    double g = (a >= kLanczCutoff ? stirling_approximation(a) : lanczos_approximation(a));
    if (x >= a + 1.0) return gamma_continued_fraction(a, x, g);
    if ((g = gamma_series(a, x, g)) < 0) return g;
    return 1.0 - g;
}

/* Poisson.cs:26-29 */
double orc_poisson_cdf(double num_occurrences, double expected)
{
    return incomplete_gamma_function((double)(int)(num_occurrences + 1.0), expected);
}

/* stats/MathOperations.cs:7-10: Math.Pow(10, -1 * q / 10f) with q double -> double division */
double orc_q_to_p(double q) { return pow(10.0, -1 * q / (double)10.0f); }
/* stats/MathOperations.cs:12-15 */
double orc_p_to_q(double p) { return (-10 * log10(p)); }

/* =====================================================================================
 * MathNet.Numerics 4.5.1 (NuGet; lib/Pisces.Calculators/Pisces.Calculators.csproj:24) —
 * restated from the published algorithm: SpecialFunctions.GammaLn (Lanczos, g = 10.900511,
 * 11 terms), GammaLowerRegularized (Cephes igam/igamc), Poisson.CumulativeDistribution and
 * ProbabilityLn.  Call sites: VariantQualityCalculator.cs:36,38,47.
 * ===================================================================================== */
static const double kGammaDk[11] = {
    2.48574089138753565546e-5, 1.05142378581721974210, -3.45687097222016235469,
    4.51227709466894823700, -2.98285225323576655721, 1.05639711577126713077,
    -1.95428773191645869583e-1, 1.70970543404441224307e-2, -5.71926117404305781283e-4,
    4.63399473359905636708e-6, -2.71994908488607703910e-9};
static const double kGammaR = 10.900511;
static const double kLogTwoSqrtEOverPi = 0.6207822376352452223455184457816472122518527279025978;
static const double kLnPi = 1.1447298858494001741434273513530587116472948129153;

double orc_mathnet_gamma_ln(double z)
{
    if (z < 0.5) {
        double s = kGammaDk[0];
        for (int i = 1; i <= 10; i++) s += kGammaDk[i] / (i - z);
        return kLnPi - log(sin(M_PI * z)) - log(s) - kLogTwoSqrtEOverPi -
               ((0.5 - z) * log((0.5 - z + kGammaR) / M_E));
    } else {
        double s = kGammaDk[0];
        for (int i = 1; i <= 10; i++) s += kGammaDk[i] / (z + i - 1.0);
        return log(s) + kLogTwoSqrtEOverPi + ((z - 0.5) * log((z - 0.5 + kGammaR) / M_E));
    }
}

static double mathnet_factorial_ln(int x)
{
    /* SpecialFunctions.FactorialLn: cache of 171 factorials built by repeated multiplication */
    static double cache[171];
    static int init = 0;
    if (!init) {
        cache[0] = 1.0;
        for (int i = 1; i < 171; i++) cache[i] = cache[i - 1] * i;
        init = 1;
    }
    if (x <= 1) return 0.0;
    if (x < 171) return log(cache[x]);
    return orc_mathnet_gamma_ln(x + 1.0);
}

static int almost_zero(double v) { return fabs(v) < 1e-15; }

double orc_mathnet_gamma_lower_regularized(double a, double x)
{
    const double epsilon = 0.000000000000001;
    const double big = 4503599627370496.0;
    const double bigInv = 2.22044604925031308085e-16;

    if (almost_zero(a)) {
        if (almost_zero(x)) return NAN;
        return 1.0;
    }
    if (almost_zero(x)) return 0.0;

    double ax = (a * log(x)) - x - orc_mathnet_gamma_ln(a);
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

/* Poisson.CumulativeDistribution(x) = 1 - GammaLowerRegularized(x + 1, lambda) */
double orc_mathnet_poisson_cdf(double lambda, double x)
{
    return 1.0 - orc_mathnet_gamma_lower_regularized(x + 1, lambda);
}
/* Poisson.ProbabilityLn(k) = -lambda + k ln(lambda) - FactorialLn(k) */
double orc_mathnet_poisson_ln_pmf(double lambda, int32_t k)
{
    return -lambda + (k * log(lambda)) - mathnet_factorial_ln(k);
}

/* =====================================================================================
 * lib/Pisces.Calculators/VariantQualityCalculator.cs
 * ===================================================================================== */
/* :67-74 */
double orc_assign_pvalue(int32_t observed, int32_t coverage, int32_t nl)
{
    double errorRate = orc_q_to_p(nl);
    if (observed == 0) return 1.0;
    return (1 - orc_poisson_cdf(observed - 1.0, coverage * errorRate));
}

/* :27-52 */
double orc_raw_poisson_qscore(int32_t callCount, int32_t coverage, int32_t nl)
{
    double errorRate = orc_q_to_p(nl);
    double callCountMinusOne = callCount - 1;
    double callCountDouble = callCount;
    double lambda = errorRate * coverage;
    double pValue = 1 - orc_mathnet_poisson_cdf(lambda, callCountMinusOne);
    if (pValue > 0) {
        return orc_p_to_q(pValue);
    } else {
        double A = orc_mathnet_poisson_ln_pmf(lambda, (int)callCountMinusOne);
        double correction = (callCountDouble - lambda) / callCountDouble;
        double qScore = -10.0 * (A - log(2.0 * correction)) / log(10.0);
        return qScore;
    }
}

/* :54-65; Math.Round = banker's rounding = rint() in the default FP environment */
int32_t orc_poisson_qscore(int32_t callCount, int32_t coverage, int32_t nl, int32_t maxQ)
{
    if ((callCount <= 0) || (coverage <= 0)) return 0;
    double rawQ = orc_raw_poisson_qscore(callCount, coverage, nl);
    double qScore = fmin((double)maxQ, rawQ);
    qScore = fmax(qScore, 0);
    return (int32_t)rint(qScore);
}

/* =====================================================================================
 * lib/Pisces.Calculators/StrandBiasCalculator.cs
 * ===================================================================================== */
/* MathNet.Numerics 4.5.1 SpecialFunctions.BetaRegularized (Beta.cs): the continued fraction of Numerical Recipes' betacf with the
 * symmetry transformation, eps = 2^-53 (Precision.DoublePrecision), fpmin = double.Epsilon / eps, at most 50000 rounds. */
double orc_mathnet_beta_regularized(double a, double b, double x)
{
    const double bt = (x == 0.0 || x == 1.0) ? 0.0
                      : exp(orc_mathnet_gamma_ln(a + b) - orc_mathnet_gamma_ln(a) - orc_mathnet_gamma_ln(b) + (a * log(x)) + (b * log(1.0 - x)));
    const int symmetryTransformation = x >= (a + 1.0) / (a + b + 2.0);
    const double eps = 1.1102230246251565e-16;
    const double fpmin = 4.9406564584124654e-324 / eps;
    if (symmetryTransformation) { x = 1.0 - x; double swap = a; a = b; b = swap; }
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
        if (fabs(del - 1.0) <= eps) return symmetryTransformation ? 1.0 - (bt * h / a) : bt * h / a;
    }
    return symmetryTransformation ? 1.0 - (bt * h / a) : bt * h / a;
}
/* Binomial(p, n).CumulativeDistribution(x) (Distributions/Binomial.cs CDF): BetaRegularized(n - k, k + 1, 1 - p), k = floor(x) */
double orc_mathnet_binomial_cdf(double p, int n, double x)
{
    if (x < 0.0) return 0.0;
    if (x > n) return 1.0;
    const double k = floor(x);
    return orc_mathnet_beta_regularized(n - k, k + 1, 1 - p);
}
/* Binomial(p, n).ProbabilityLn(k) (PMFLn): BinomialLn(n, k) + k ln p + (n - k) ln(1 - p) */
double orc_mathnet_binomial_lnpmf(double p, int n, int k)
{
    if (k < 0 || k > n) return -INFINITY;
    if (p == 0.0) return k == 0 ? 0.0 : -INFINITY;
    if (p == 1.0) return k == n ? 0.0 : -INFINITY;
    const double binomialLn = mathnet_factorial_ln(n) - mathnet_factorial_ln(k) - mathnet_factorial_ln(n - k);
    return binomialLn + (k * log(p)) + ((n - k) * log(1.0 - p));
}

/* PopulateDiploidStats :150-173 */
void orc_sb_populate_diploid_stats(double support, double coverage, double minDetectableSNP, double out3[3] /* FN, FP, P(var > 0) */)
{
    const double frequency = coverage == 0 ? 0 : support / coverage;
    if (frequency >= minDetectableSNP) { out3[0] = 1; out3[1] = 0; out3[2] = 1; return; }
    out3[0] = fmax(orc_mathnet_binomial_cdf(minDetectableSNP, (int)coverage, support), 0);
    out3[1] = fmax(0.0, 1 - orc_poisson_cdf(support, coverage * 0.1));
    out3[2] = out3[0];
}

/* PopulateStats :175-231 */
static void sb_create_stats(OrcSbStats* st, double support, double coverage, double noiseFreq,
                            double minDetectableSNP, int32_t model)
{
    if (model != PISCES_SB_DIPLOID) minDetectableSNP = noiseFreq; /* CreateStats :141-142 */
    st->frequency = support / coverage;                           /* StrandBiasStats ctor */
    st->support = support;
    st->coverage = coverage;
    if (coverage == 0) st->frequency = 0;
    st->chance_false_neg = st->chance_false_pos = st->chance_var_freq_gt_zero = 0;

    if (st->support == 0) {
        if (model == PISCES_SB_POISSON) {
            st->chance_false_pos = 1;
            st->chance_var_freq_gt_zero = 0;
            st->chance_false_neg = 0;
        } else {
            st->chance_var_freq_gt_zero = pow(1 - minDetectableSNP, st->coverage);
            st->chance_false_pos = 1 - st->chance_var_freq_gt_zero;
            st->chance_false_neg = st->chance_var_freq_gt_zero;
        }
    } else if (model == PISCES_SB_DIPLOID) {
        double o[3];
        orc_sb_populate_diploid_stats(st->support, st->coverage, minDetectableSNP, o);
        st->chance_false_neg = o[0]; st->chance_false_pos = o[1]; st->chance_var_freq_gt_zero = o[2];
    } else {
        st->chance_var_freq_gt_zero = fmax(0, orc_poisson_cdf(st->support - 1, st->coverage * noiseFreq));
        st->chance_false_pos = fmax(0, 1 - st->chance_var_freq_gt_zero);
        st->chance_false_neg = fmax(0, orc_poisson_cdf(st->support, st->coverage * minDetectableSNP));
    }
}

/* CalculateStrandBiasResults :21-72, AssignBiasScore :89-105 */
void orc_strand_bias(const int32_t cov[3], const int32_t sup[3], int32_t qNoise, double minVariantFreq,
                     double acceptance, int32_t model, OrcBiasResults* r)
{
    int forwardSupport = sup[PISCES_DIR_FORWARD], forwardCoverage = cov[PISCES_DIR_FORWARD];
    int reverseSupport = sup[PISCES_DIR_REVERSE], reverseCoverage = cov[PISCES_DIR_REVERSE];
    int stitchedSupport = sup[PISCES_DIR_STITCHED], stitchedCoverage = cov[PISCES_DIR_STITCHED];

    /* Math.Pow(10, -1*qNoise/10f): int / float -> float exponent */
    double errorRate = pow(10.0, (double)((float)(-1 * qNoise) / 10.0f));

    sb_create_stats(&r->overall, forwardSupport + reverseSupport + stitchedSupport,
                    forwardCoverage + reverseCoverage + stitchedCoverage, errorRate, minVariantFreq, model);
    sb_create_stats(&r->forward, forwardSupport + stitchedSupport / 2, forwardCoverage + stitchedCoverage / 2,
                    errorRate, minVariantFreq, model);
    sb_create_stats(&r->reverse, reverseSupport + stitchedSupport / 2, reverseCoverage + stitchedCoverage / 2,
                    errorRate, minVariantFreq, model);
    sb_create_stats(&r->stitched, stitchedSupport, stitchedCoverage, errorRate, minVariantFreq, model);

    double forwardBias = (r->forward.chance_var_freq_gt_zero * r->reverse.chance_false_pos) /
                         r->overall.chance_var_freq_gt_zero;
    double reverseBias = (r->reverse.chance_var_freq_gt_zero * r->forward.chance_false_pos) /
                         r->overall.chance_var_freq_gt_zero;
    if (r->overall.chance_var_freq_gt_zero == 0) {
        forwardBias = 1;
        reverseBias = 1;
    }
    /* Math.Max: NaN-propagating; operands here are never NaN after the guard above */
    double p = forwardBias > reverseBias ? forwardBias : reverseBias;
    r->bias_score = p;
    r->gatk_bias_score = 10 * log10(p);
    r->cov_present_on_both = ((r->forward.coverage > 0) && (r->reverse.coverage > 0));
    r->var_present_on_both = ((r->forward.support > 0) && (r->reverse.support > 0));
    if (!r->cov_present_on_both) {
        r->bias_score = 0;
        r->gatk_bias_score = -INFINITY;
    }
    r->bias_acceptable = (r->bias_score < acceptance);
}

/* =====================================================================================
 * lib/Pisces.Genotyping/Somatic
 * ===================================================================================== */
/* CalledAllele.Frequency / RefFrequency (CalledAllele.cs:49-52,121-124): float32 */
static float frequency_f(int32_t support, int32_t coverage)
{
    if (coverage == 0) return 0.0f;
    float f = (float)support / (float)coverage;
    return f < 1.0f ? f : 1.0f;
}

/* SomaticGenotyper.CalculateSomaticGenotype :65-100 */
int32_t orc_somatic_genotype(int32_t category, int32_t totalCoverage, int32_t alleleSupport,
                             int32_t referenceSupport, float minFrequencyFilter, int32_t minDepthToGenotype)
{
    if (totalCoverage < minDepthToGenotype)
        return (category == PISCES_CAT_REFERENCE) ? PISCES_GT_REF_LIKE_NOCALL : PISCES_GT_ALT_LIKE_NOCALL;
    float freq = frequency_f(alleleSupport, totalCoverage);
    if (category != PISCES_CAT_REFERENCE) {
        float refFreq = frequency_f(referenceSupport, totalCoverage);
        if (refFreq < minFrequencyFilter) {
            if ((1 - freq) > minFrequencyFilter) return PISCES_GT_ALT_AND_NOCALL;
            return PISCES_GT_HOM_ALT;
        }
        return PISCES_GT_HET_ALT_REF;
    } else {
        if (freq < minFrequencyFilter) return PISCES_GT_REF_LIKE_NOCALL;
        if ((1 - freq) > minFrequencyFilter) return PISCES_GT_REF_AND_NOCALL;
    }
    return PISCES_GT_HOM_REF;
}

/* =====================================================================================
 * lib/Pisces.Genotyping/Thresholding (PloidyModel.DiploidByThresholding) + GenotypeCalculatorUtilities.cs
 * ===================================================================================== */
/* DiploidGenotypeQualityCalculator.Compute :12-105 */
int32_t orc_diploid_gq(int32_t calledGT, int32_t totalCoverage, int32_t alleleSupport, int32_t minQScore, int32_t maxQScore)
{
    if (totalCoverage == 0) return minQScore;
    const float noiseHomRef = 0.05f, noiseHomAlt = 0.075f, noiseHetAlt = 0.10f, expectedHetFreq = 0.40f;
    const float depth = (float)totalCoverage;
    const float frequency = frequency_f(alleleSupport, totalCoverage);
    const double lamHomRef = (double)(noiseHomRef * depth), lamHomAlt = (double)(noiseHomAlt * depth);   /* float products, as written */
    const int nonAlleleCalls = totalCoverage - alleleSupport > 0 ? totalCoverage - alleleSupport : 0;
    double LnPofH0GT = 0, LnPofH1GT = 0;
    switch (calledGT) {
    case PISCES_GT_HOM_REF:
        LnPofH0GT = orc_mathnet_poisson_ln_pmf(lamHomRef, nonAlleleCalls);
        LnPofH1GT = orc_mathnet_binomial_lnpmf((double)expectedHetFreq, totalCoverage, nonAlleleCalls);
        break;
    case PISCES_GT_HOM_ALT:
        LnPofH0GT = orc_mathnet_poisson_ln_pmf(lamHomAlt, nonAlleleCalls);
        LnPofH1GT = orc_mathnet_binomial_lnpmf((double)expectedHetFreq, totalCoverage, alleleSupport);
        break;
    case PISCES_GT_HET_ALT1_ALT2:
    case PISCES_GT_HET_ALT_REF: {
        const int k = (int)(depth * frequency);
        LnPofH0GT = orc_mathnet_binomial_lnpmf((double)expectedHetFreq, totalCoverage, k);
        if (frequency >= 0.50) LnPofH1GT = orc_mathnet_binomial_lnpmf((double)(1 - noiseHetAlt), totalCoverage, k);
        else LnPofH1GT = orc_mathnet_binomial_lnpmf((double)noiseHetAlt, totalCoverage, k);
        break;
    }
    default: return minQScore;
    }
    /* (int)Math.Floor(...): a value outside int (or NaN) converts to int.MinValue */
    const double v = floor(10.0 * 0.4342944819032518 * (LnPofH0GT - LnPofH1GT));
    const int32_t qScore = (v > -2147483649.0 && v < 2147483648.0) ? (int32_t)v : INT32_MIN;
    if ((LnPofH1GT <= (double)INT32_MIN) && (LnPofH0GT > LnPofH1GT)) return maxQScore;
    if ((LnPofH0GT <= (double)INT32_MIN) && (LnPofH0GT < LnPofH1GT)) return minQScore;
    const int32_t capped = qScore < maxQScore ? qScore : maxQScore;
    return capped > minQScore ? capped : minQScore;
}

/* DiploidThresholdingGenotyper.SetGenotypes :54-75 + CalculateDiploidGenotype :77-103 over the alleles of one locus (already without
 * pruned Reference rows).  Sets genotype, genotype q-score, phase set index and the MultiAllelicSite filter; prune[i] = 1 for the
 * alleles the genotyper asks the caller to drop.  Returns the locus genotype. */
int32_t orc_diploid_set_genotypes(OrcCalled* alleles, int n, const float snv[3], const float indel[3], int32_t minDepthToGenotype,
                                  int32_t minGQ, int32_t maxGQ, int32_t* phase_set_index, uint8_t* prune)
{
    int order[64];
    int nv = 0;
    for (int i = 0; i < n; i++) prune[i] = 0;
    /* FilterAndOrderAllelesByFrequency: variants at or above the SNV MinorVF, by descending frequency then (ref, alt) */
    const double minFreqThreshold = (double)snv[0];
    for (int i = 0; i < n && nv < 64; i++) {
        if (alleles[i].category == PISCES_CAT_REFERENCE) continue;
        if ((double)frequency_f(alleles[i].allele_support, alleles[i].total_coverage) >= minFreqThreshold) order[nv++] = i;
        else prune[i] = 1;
    }
    for (int a = 1; a < nv; a++) {   /* stable insertion sort */
        int x = order[a], b = a;
        while (b > 0) {
            const OrcCalled* p = &alleles[order[b - 1]];
            const OrcCalled* q = &alleles[x];
            const float fp = frequency_f(p->allele_support, p->total_coverage), fq = frequency_f(q->allele_support, q->total_coverage);
            int after = 0;
            if (fp != fq) after = fp < fq;
            else { int r = strcmp(p->ref, q->ref); if (!r) r = strcmp(p->alt, q->alt); after = r > 0; }
            if (!after) break;
            order[b] = order[b - 1];
            b--;
        }
        order[b] = x;
    }
    /* GetReferenceFrequency */
    double referenceFrequency = 0;
    if (n == 1) referenceFrequency = frequency_f(alleles[0].reference_support, alleles[0].total_coverage);
    else if (n > 1) {
        double refFrequencyCountBySNP = 0, indelFrequencyCount = 0;
        int returned = 0;
        for (int i = 0; i < n && !returned; i++) {
            const float f = frequency_f(alleles[i].allele_support, alleles[i].total_coverage);
            if (alleles[i].category == PISCES_CAT_REFERENCE) { referenceFrequency = f; returned = 1; break; }
            if (alleles[i].category == PISCES_CAT_SNV) refFrequencyCountBySNP = frequency_f(alleles[i].reference_support, alleles[i].total_coverage);
            else indelFrequencyCount += f;
        }
        if (!returned) referenceFrequency = fmax(refFrequencyCountBySNP - indelFrequencyCount, 0.0);
    }
    const int refExists = referenceFrequency >= (double)snv[0];
    int depthIssue = 0;
    for (int i = 0; i < n; i++) depthIssue |= alleles[i].total_coverage < minDepthToGenotype;
    const float f0 = nv ? frequency_f(alleles[order[0]].allele_support, alleles[order[0]].total_coverage) : 0.0f;
    const int refCall = nv == 0 || f0 < snv[0];
    const float* par = (!refCall && alleles[order[0]].category != PISCES_CAT_SNV) ? indel : snv;   /* SelectParameters */
    /* GetPreliminaryGenotype: 0 HomozygousRef, 1 HeterozygousAltRef, 2 HomozygousAlt */
    int prelim = 0;
    if (!refCall) prelim = (f0 >= par[0] && f0 <= par[1]) ? 1 : (f0 > par[1]) ? 2 : 0;
    /* ConvertSimpleGenotypeToComplexGenotype */
    int32_t gt;
    if (depthIssue) gt = refCall ? PISCES_GT_REF_LIKE_NOCALL : PISCES_GT_ALT_LIKE_NOCALL;
    else if (prelim == 0) {
        if (!refExists) gt = PISCES_GT_REF_LIKE_NOCALL;
        else {
            const float first = frequency_f(alleles[0].allele_support, alleles[0].total_coverage);
            gt = (alleles[0].category == PISCES_CAT_REFERENCE && (1 - first) > par[0]) ? PISCES_GT_REF_AND_NOCALL : PISCES_GT_HOM_REF;
        }
    } else if (prelim == 1) {
        if (nv == 1) gt = refExists ? PISCES_GT_HET_ALT_REF : PISCES_GT_ALT_AND_NOCALL;
        else {
            /* CheckForTriAllelicIssue */
            int fail;
            if (alleles[order[nv - 1]].category != PISCES_CAT_SNV) fail = 0;
            else if (refExists && ((double)f0 + referenceFrequency) < (double)par[2]) fail = 1;
            else {
                const float f1 = frequency_f(alleles[order[1]].allele_support, alleles[order[1]].total_coverage);
                fail = (f0 + f1) < par[2];
            }
            if (fail) {
                for (int i = 0; i < n; i++) alleles[i].filters |= 1u << PISCES_FILTER_MULTI_ALLELIC_SITE;   /* SetMultiAllelicFilter */
                gt = refExists ? PISCES_GT_ALT_LIKE_NOCALL : PISCES_GT_ALT12_LIKE_NOCALL;
            } else {
                gt = refExists ? PISCES_GT_HET_ALT_REF : PISCES_GT_HET_ALT1_ALT2;
            }
        }
    } else gt = PISCES_GT_HOM_ALT;
    /* GetAllelesToPruneBasedOnGTCall */
    int allowed = 0;
    if (gt == PISCES_GT_ALT_AND_NOCALL || gt == PISCES_GT_ALT_LIKE_NOCALL || gt == PISCES_GT_HOM_ALT || gt == PISCES_GT_HET_ALT_REF) allowed = 1;
    else if (gt == PISCES_GT_ALT12_LIKE_NOCALL || gt == PISCES_GT_HET_ALT1_ALT2) allowed = 2;
    for (int k = allowed; k < nv; k++) prune[order[k]] = 1;
    /* SetGenotypes: every allele gets the locus genotype, its own q-score, a phase set index in list order */
    int phase = 1;
    for (int i = 0; i < n; i++) {
        alleles[i].genotype = gt;
        alleles[i].genotype_qscore = orc_diploid_gq(gt, alleles[i].total_coverage, alleles[i].allele_support, minGQ, maxGQ);
        if (alleles[i].category == PISCES_CAT_REFERENCE) phase_set_index[i] = 0;
        else phase_set_index[i] = phase++;
    }
    return gt;
}

/* HaploidGenotypeQualityCalculator.Compute :10-59 */
int32_t orc_haploid_gq(int32_t calledGT, int32_t totalCoverage, int32_t alleleSupport, int32_t minQScore, int32_t maxQScore)
{
    if (totalCoverage == 0) return minQScore;
    const float noiseHomRef = 0.05f, noiseHomAlt = 0.075f, expectedHetFreq = 0.40f;
    const float depth = (float)totalCoverage;
    const int nonAlleleCalls = totalCoverage - alleleSupport > 0 ? totalCoverage - alleleSupport : 0;
    double h0, h1;
    if (calledGT == PISCES_GT_HEMI_REF) {
        h0 = orc_mathnet_poisson_ln_pmf((double)(noiseHomRef * depth), nonAlleleCalls);
        h1 = orc_mathnet_binomial_lnpmf((double)expectedHetFreq, totalCoverage, nonAlleleCalls);
    } else if (calledGT == PISCES_GT_HEMI_ALT) {
        h0 = orc_mathnet_poisson_ln_pmf((double)(noiseHomAlt * depth), nonAlleleCalls);
        h1 = orc_mathnet_binomial_lnpmf((double)expectedHetFreq, totalCoverage, alleleSupport);
    } else return minQScore;
    const double v = floor(10.0 * 0.4342944819032518 * (h0 - h1));
    const int32_t qScore = (v > -2147483649.0 && v < 2147483648.0) ? (int32_t)v : INT32_MIN;
    const int32_t capped = qScore < maxQScore ? qScore : maxQScore;
    return capped > minQScore ? capped : minQScore;
}

/* HaploidGenotyper.SetGenotypes :36-48 + CalculateHaploidGenotype :50-83 */
int32_t orc_haploid_set_genotypes(OrcCalled* alleles, int n, float minorVF, float majorVF, int32_t minDepthToGenotype, int32_t minGQ,
                                  int32_t maxGQ, uint8_t* prune)
{
    int order[64];
    int nv = 0;
    for (int i = 0; i < n; i++) prune[i] = 0;
    for (int i = 0; i < n && nv < 64; i++) {
        if (alleles[i].category == PISCES_CAT_REFERENCE) continue;
        if ((double)frequency_f(alleles[i].allele_support, alleles[i].total_coverage) >= (double)minorVF) order[nv++] = i;
        else prune[i] = 1;
    }
    for (int a = 1; a < nv; a++) {
        int x = order[a], b = a;
        while (b > 0) {
            const OrcCalled* p = &alleles[order[b - 1]];
            const OrcCalled* q = &alleles[x];
            const float fp = frequency_f(p->allele_support, p->total_coverage), fq = frequency_f(q->allele_support, q->total_coverage);
            int after;
            if (fp != fq) after = fp < fq;
            else { int r = strcmp(p->ref, q->ref); if (!r) r = strcmp(p->alt, q->alt); after = r > 0; }
            if (!after) break;
            order[b] = order[b - 1];
            b--;
        }
        order[b] = x;
    }
    double referenceFrequency = 0;
    if (n == 1) referenceFrequency = frequency_f(alleles[0].reference_support, alleles[0].total_coverage);
    else if (n > 1) {
        double refBySNP = 0, indelCount = 0;
        int returned = 0;
        for (int i = 0; i < n; i++) {
            const float f = frequency_f(alleles[i].allele_support, alleles[i].total_coverage);
            if (alleles[i].category == PISCES_CAT_REFERENCE) { referenceFrequency = f; returned = 1; break; }
            if (alleles[i].category == PISCES_CAT_SNV) refBySNP = frequency_f(alleles[i].reference_support, alleles[i].total_coverage);
            else indelCount += f;
        }
        if (!returned) referenceFrequency = fmax(refBySNP - indelCount, 0.0);
    }
    const int refExists = referenceFrequency >= (double)minorVF;
    int depthIssue = 0;
    for (int i = 0; i < n; i++) depthIssue |= alleles[i].total_coverage < minDepthToGenotype;
    const float f0 = nv ? frequency_f(alleles[order[0]].allele_support, alleles[order[0]].total_coverage) : 0.0f;
    const int refCall = nv == 0 || f0 < minorVF;
    int32_t gt = PISCES_GT_HEMI_NOCALL;
    if (!depthIssue && refCall && refExists && referenceFrequency > (double)majorVF) gt = PISCES_GT_HEMI_REF;
    if (!depthIssue && !refCall && !refExists && f0 > majorVF) gt = PISCES_GT_HEMI_ALT;
    for (int k = (gt == PISCES_GT_HEMI_ALT ? 1 : 0); k < nv; k++) prune[order[k]] = 1;
    for (int i = 0; i < n; i++) {
        alleles[i].genotype = gt;
        alleles[i].genotype_qscore = orc_haploid_gq(gt, alleles[i].total_coverage, alleles[i].allele_support, minGQ, maxGQ);
    }
    return gt;
}

/* SomaticGenotypeQualityCalculator.Compute :10-48 */
int32_t orc_somatic_gq(int32_t genotype, int32_t variantQ, int32_t totalCoverage, int32_t alleleSupport,
                       float targetLod, int32_t minGQ, int32_t maxGQ)
{
    double rawQ = variantQ;
    int isNoCall = (genotype == PISCES_GT_ALT12_LIKE_NOCALL || genotype == PISCES_GT_ALT_LIKE_NOCALL ||
                    genotype == PISCES_GT_REF_LIKE_NOCALL || genotype == 11 /* HemizygousNoCall */);
    if ((totalCoverage == 0) || isNoCall) return minGQ;
    if ((genotype == PISCES_GT_HOM_REF) || (genotype == PISCES_GT_HOM_ALT)) {
        double p1 = orc_q_to_p(variantQ);
        float nonAlleleObservationsF = (1.0f - frequency_f(alleleSupport, totalCoverage)) * (float)totalCoverage;
        float expectedNonAllelObservationsF = targetLod * (float)totalCoverage;
        if (nonAlleleObservationsF >= expectedNonAllelObservationsF) return minGQ;
        double p2 = orc_poisson_cdf(nonAlleleObservationsF, expectedNonAllelObservationsF);
        rawQ = orc_p_to_q(p1 + p2);
    }
    double qScore = fmin((double)maxGQ, rawQ);
    qScore = fmax(qScore, (double)minGQ);
    return (int32_t)rint(qScore);
}

/* =====================================================================================
 * Read geometry: Read.cs:535-562 (UpdatePositionMap), BamCommon.cs:119,560-585
 * ===================================================================================== */
static int op_is_ref_span(uint8_t t) { return t == 'M' || t == 'D' || t == 'N' || t == '=' || t == 'X'; }
static int op_is_read_span(uint8_t t) { return t == 'M' || t == 'I' || t == 'S' || t == '=' || t == 'X'; }

static int32_t read_ref_span(const OrcRead* r)
{
    int32_t n = 0;
    for (int i = 0; i < r->n_cigar; i++)
        if (op_is_ref_span(r->cigar_op[i])) n += (int32_t)r->cigar_len[i];
    return n;
}
/* Read.EndPosition = BamAlignment.EndPosition + 1 = Position(1-based) + refSpan - 1 (Read.cs:88-91) */
static int32_t read_end_position(const OrcRead* r) { return r->position + read_ref_span(r) - 1; }

static void build_position_map(const OrcRead* r, int32_t* posmap)
{
    if (r->posmap_override) {
        memcpy(posmap, r->posmap_override, sizeof(int32_t) * (size_t)r->read_len);
        return;
    }
    for (int i = 0; i < r->read_len; i++) posmap[i] = -1;
    int readIndex = 0;
    int referencePosition = r->position;
    for (int c = 0; c < r->n_cigar; c++) {
        int readSpan = op_is_read_span(r->cigar_op[c]);
        int refSpan = op_is_ref_span(r->cigar_op[c]);
        for (uint32_t k = 0; k < r->cigar_len[c]; k++) {
            if (readSpan) {
                if (readIndex < r->read_len) posmap[readIndex] = refSpan ? referencePosition++ : -1;
                readIndex++;
            } else if (refSpan) {
                referencePosition++;
            }
        }
    }
}

static int32_t read_dir(const OrcRead* r, int idx)
{
    if (r->dirs) return r->dirs[idx];
    return r->is_reverse ? PISCES_DIR_REVERSE : PISCES_DIR_FORWARD; /* Read.cs:394-401 */
}

/* CigarExtensions.HasOperationAtOpIndex (Utility/CigarExtensions.cs:38-44) */
static int has_op_at(const OrcRead* r, int index, uint8_t type, int fromEnd)
{
    int opIndex = fromEnd ? r->n_cigar - index - 1 : index;
    return r->n_cigar > opIndex && opIndex >= 0 && r->cigar_op[opIndex] == type;
}

/* AlleleHelper.GetAlleleType (Utility/AlleleHelper.cs:13-32) */
static int32_t allele_type_of(uint8_t c)
{
    switch (c) {
    case 'A': return PISCES_ALLELE_A;
    case 'C': return PISCES_ALLELE_C;
    case 'G': return PISCES_ALLELE_G;
    case 'T': return PISCES_ALLELE_T;
    default: return PISCES_ALLELE_N;
    }
}

/* CandidateVariantFinder.CheckDeletionQuality :294-320 */
int32_t orc_check_deletion_quality(const OrcRead* r, int32_t opStartIndexInRead, int32_t minBQ)
{
    if (r->read_len == 0) return 0;
    int after = (opStartIndexInRead < r->read_len) ? r->quals[opStartIndexInRead] : r->quals[opStartIndexInRead - 1];
    int before = after;
    if (opStartIndexInRead > 0) before = r->quals[opStartIndexInRead - 1];
    return (before >= minBQ) && (after >= minBQ);
}

/* =====================================================================================
 * lib/Pisces.Processing/RegionState
 * ===================================================================================== */
struct OrcState {
    int32_t start_position, n_loci, min_bq, num_anchor_types, n_anchor_idx, track_open_ended;
    int32_t* counts;   /* [n_loci][6][3][n_anchor_idx]  RegionState.cs:57 */
    double* sumq;      /* RegionState.cs:61 */
    int32_t* gapped;   /* RegionState.cs:58 */
    int32_t* cand_head;
    int32_t* cand_tail;
    OrcCandidate* cands;
    int32_t n_cands, cap_cands;
    /* the block grid of RegionStateManager over the window, when a schedule is run (orc_track_blocks): MaxAlleleEndpoint per block */
    int32_t block_size, first_block_key, n_blocks, next_block_key, last_up_to_block_key, have_last_up_to;
    int32_t* max_allele_endpoint;
    /* forced genotyping alleles (-forcedalleles): sorted by position; [0, n_forced_added) have been handed in as candidates */
    OrcCandidate* forced;
    int32_t n_forced, n_forced_added;
    /* ChrIntervalSet (sorted, disjoint, inclusive), or none: Reference candidates are made inside it only (RegionState.cs:414-447) and a
     * callable allele outside it is counted and not reported (AlleleCaller.ShouldReport :260-263).  None: the state's window is the interval. */
    int32_t* iv_start;
    int32_t* iv_end;
    int32_t n_intervals;
};

OrcState* orc_state_create(int32_t start, int32_t n_loci, int32_t min_bq, int32_t num_anchor_types,
                           int32_t track_open_ended)
{
    OrcState* s = (OrcState*)calloc(1, sizeof(OrcState));
    s->start_position = start;
    s->n_loci = n_loci;
    s->min_bq = min_bq;
    s->num_anchor_types = num_anchor_types;
    s->n_anchor_idx = num_anchor_types * 2 + 1;
    s->track_open_ended = track_open_ended;
    size_t per = (size_t)6 * 3 * (size_t)s->n_anchor_idx;
    s->counts = (int32_t*)calloc((size_t)n_loci * per, sizeof(int32_t));
    s->sumq = (double*)calloc((size_t)n_loci * per, sizeof(double));
    s->gapped = (int32_t*)calloc((size_t)n_loci, sizeof(int32_t));
    s->cand_head = (int32_t*)malloc(sizeof(int32_t) * (size_t)n_loci);
    s->cand_tail = (int32_t*)malloc(sizeof(int32_t) * (size_t)n_loci);
    for (int i = 0; i < n_loci; i++) s->cand_head[i] = s->cand_tail[i] = -1;
    return s;
}

void orc_state_destroy(OrcState* s)
{
    if (!s) return;
    free(s->counts); free(s->sumq); free(s->gapped); free(s->cand_head); free(s->cand_tail); free(s->cands);
    free(s->max_allele_endpoint);
    free(s->forced);
    free(s->iv_start); free(s->iv_end);
    free(s);
}

void orc_set_intervals(OrcState* s, const int32_t* starts, const int32_t* ends, int32_t n)
{
    free(s->iv_start); free(s->iv_end);
    s->iv_start = s->iv_end = NULL;
    s->n_intervals = 0;
    if (n <= 0) return;
    s->iv_start = (int32_t*)malloc(sizeof(int32_t) * (size_t)n);
    s->iv_end = (int32_t*)malloc(sizeof(int32_t) * (size_t)n);
    memcpy(s->iv_start, starts, sizeof(int32_t) * (size_t)n);
    memcpy(s->iv_end, ends, sizeof(int32_t) * (size_t)n);
    s->n_intervals = n;
}
/* ChrIntervalSet.ContainsPosition; without an interval set every position of the window */
static int inside_intervals(const OrcState* s, int32_t position)
{
    if (s->n_intervals == 0) return 1;
    int lo = 0, hi = s->n_intervals - 1;
    while (lo <= hi) {
        const int m = (lo + hi) / 2;
        if (position < s->iv_start[m]) hi = m - 1;
        else if (position > s->iv_end[m]) lo = m + 1;
        else return 1;
    }
    return 0;
}

const int32_t* orc_counts_ptr(const OrcState* s) { return s->counts; }
int32_t orc_num_anchor_indexes(const OrcState* s) { return s->n_anchor_idx; }

static inline size_t cidx(const OrcState* s, int32_t pos, int allele, int dir, int anchor)
{
    return (((size_t)(pos - s->start_position) * 6 + (size_t)allele) * 3 + (size_t)dir) * (size_t)s->n_anchor_idx +
           (size_t)anchor;
}
static inline int in_region(const OrcState* s, int32_t pos)
{
    return pos >= s->start_position && pos < s->start_position + s->n_loci;
}

/* RegionState.AddAlleleCount :225-231 (positions outside the oracle's dense window are dropped,
 * the analogue of a block nobody asks for) */
static void add_allele_count(OrcState* s, int32_t pos, int allele, int dir, int anchor)
{
    if (in_region(s, pos)) s->counts[cidx(s, pos, allele, dir, anchor)]++;
}
static void add_base_quality(OrcState* s, int32_t pos, int allele, int dir, double q, int anchor)
{
    if (in_region(s, pos)) s->sumq[cidx(s, pos, allele, dir, anchor)] += q;
}

/* RegionStateManager.GetAnchorType :83-116; returns -1 where the reference throws */
static int get_anchor_type(const OrcState* s, int alignmentEndPosition, int basePosition, int alignmentStartPosition)
{
    int leftAnchor = basePosition - alignmentStartPosition;
    int rightAnchor = alignmentEndPosition - basePosition;
    int minAnchor;
    if (leftAnchor >= rightAnchor) {
        if (rightAnchor >= s->num_anchor_types) return s->num_anchor_types; /* WellAnchoredIndex */
        minAnchor = s->n_anchor_idx - rightAnchor - 1;
    } else {
        if (leftAnchor >= s->num_anchor_types) return s->num_anchor_types;
        minAnchor = leftAnchor;
    }
    if (minAnchor < 0) return -1;
    return minAnchor;
}

/* RegionStateManager.AddAlleleCounts :118-220 */
int32_t orc_add_allele_counts(OrcState* s, const OrcRead* r)
{
    int32_t stack_map[512];
    int32_t* posmap = r->read_len <= 512 ? stack_map : (int32_t*)malloc(sizeof(int32_t) * (size_t)r->read_len);
    build_position_map(r, posmap);
    int32_t rc = 0;

    int lastPosition = r->position - 1;
    int deletionLength = 0;
    int lengthBeforeDeletion = r->read_len;
    int endsInDeletion = has_op_at(r, 0, 'D', 1);
    int endsInDeletionBeforeSoftclip = has_op_at(r, 1, 'D', 1) && has_op_at(r, 0, 'S', 1);
    if (endsInDeletion || endsInDeletionBeforeSoftclip) {
        deletionLength = (int)(endsInDeletionBeforeSoftclip ? r->cigar_len[r->n_cigar - 2] : r->cigar_len[r->n_cigar - 1]);
        lengthBeforeDeletion = (int)(endsInDeletionBeforeSoftclip ? r->read_len - (int)r->cigar_len[r->n_cigar - 1] : r->read_len);
    }
    int positionMapLength = r->read_len;
    int alignmentEndPosition = read_end_position(r);
    int alignmentStartPosition = r->position;
    int nAnchorIdx = s->n_anchor_idx;

    for (int positionMapIndex = 0; positionMapIndex < positionMapLength; positionMapIndex++) {
        int directionType = read_dir(r, positionMapIndex);

        if ((endsInDeletionBeforeSoftclip) && positionMapIndex == lengthBeforeDeletion) {
            if (orc_check_deletion_quality(r, positionMapIndex, s->min_bq)) {
                for (int j = 1; j < deletionLength + 1; j++) {
                    int anchorIndex = nAnchorIdx - 1;
                    add_allele_count(s, j + lastPosition, PISCES_ALLELE_DEL, directionType, anchorIndex);
                }
            }
        }

        int position = posmap[positionMapIndex];
        if (position == -1) continue;

        int anchorType = get_anchor_type(s, alignmentEndPosition, position, alignmentStartPosition);
        if (anchorType < 0) { rc = PISCES_E_UNMAPPED_BASE; goto done; }

        if (orc_check_deletion_quality(r, positionMapIndex, s->min_bq)) {
            for (int j = lastPosition + 1; j < position; j++)
                add_allele_count(s, j, PISCES_ALLELE_DEL, directionType, anchorType);
        }

        int alleleType = allele_type_of(r->bases[positionMapIndex]);
        if (r->quals[positionMapIndex] < s->min_bq) alleleType = PISCES_ALLELE_N;

        add_allele_count(s, position, alleleType, directionType, anchorType);

        /* Math.Pow(10, -1 * (int)q / 10f) :191 — int / float -> float exponent */
        double bq = pow(10.0, (double)((float)(-1 * (int)r->quals[positionMapIndex]) / 10.0f));
        add_base_quality(s, position, alleleType, directionType, bq, anchorType);
        lastPosition = position;
    }

    if (endsInDeletion) {
        if (orc_check_deletion_quality(r, r->read_len - 1, s->min_bq)) {
            for (int j = 1; j < deletionLength + 1; j++) {
                int directionType = read_dir(r, r->read_len - 1);
                int anchorIndex = nAnchorIdx - 1;
                add_allele_count(s, j + lastPosition, PISCES_ALLELE_DEL, directionType, anchorIndex);
            }
        }
    }
done:
    if (posmap != stack_map) free(posmap);
    return rc;
}

/* AlleleCountHelper.GetAnchorAdjustedAlleleCount :21-85 (generic over int / double storage) */
#define ANCHOR_ADJUSTED(TYPE, ARR)                                                                        \
    int wellAnchoredIndex = s->num_anchor_types, numAnchorIndexes = s->n_anchor_idx;                      \
    int trueMinAnchor = wellAnchoredIndex < minAnchor ? wellAnchoredIndex : minAnchor;                    \
    int initialMaxAnchor = wellAnchoredIndex;                                                             \
    if (maxAnchor >= 0) {                                                                                 \
        if (maxAnchor >= wellAnchoredIndex) initialMaxAnchor = wellAnchoredIndex - 1;                     \
        if (maxAnchor < wellAnchoredIndex) initialMaxAnchor = maxAnchor;                                  \
    }                                                                                                     \
    TYPE totCount = 0;                                                                                    \
    if (fromEnd) {                                                                                        \
        for (int i = trueMinAnchor; i <= initialMaxAnchor; i++)                                           \
            totCount += ARR[cidx(s, position, allele, dir, numAnchorIndexes - i - 1)];                    \
        if (maxAnchor < 0)                                                                                \
            for (int i = symmetric ? trueMinAnchor : 0; i < initialMaxAnchor; i++)                        \
                totCount += ARR[cidx(s, position, allele, dir, i)];                                       \
    } else {                                                                                              \
        for (int i = trueMinAnchor; i <= initialMaxAnchor; i++)                                           \
            totCount += ARR[cidx(s, position, allele, dir, i)];                                           \
        if (maxAnchor < 0)                                                                                \
            for (int i = initialMaxAnchor + 1; i < (symmetric ? numAnchorIndexes - trueMinAnchor : numAnchorIndexes); i++) \
                totCount += ARR[cidx(s, position, allele, dir, i)];                                       \
    }                                                                                                     \
    return totCount;

/* RegionStateManager.GetAlleleCount :222-226: no block -> 0 */
int32_t orc_get_allele_count(const OrcState* s, int32_t position, int32_t allele, int32_t dir, int32_t minAnchor,
                             int32_t maxAnchor, int32_t fromEnd, int32_t symmetric)
{
    if (!in_region(s, position)) return 0;
    ANCHOR_ADJUSTED(int32_t, s->counts)
}
double orc_get_sum_base_quality(const OrcState* s, int32_t position, int32_t allele, int32_t dir, int32_t minAnchor,
                                int32_t maxAnchor, int32_t fromEnd, int32_t symmetric)
{
    if (!in_region(s, position)) return 0;
    ANCHOR_ADJUSTED(double, s->sumq)
}

void orc_add_gapped_mnv_ref(OrcState* s, int32_t position, int32_t count)
{
    if (in_region(s, position)) s->gapped[position - s->start_position] += count;
}
static int32_t get_gapped_mnv_ref(const OrcState* s, int32_t position)
{
    return in_region(s, position) ? s->gapped[position - s->start_position] : 0;
}

/* CandidateAllele.Equals (CandidateAllele.cs:58-68) */
static int candidate_equals(const OrcCandidate* a, const OrcCandidate* b)
{
    return a->position == b->position && strcmp(a->alt, b->alt) == 0 && a->category == b->category &&
           strcmp(a->ref, b->ref) == 0;
}

/* RegionState.AddCandidate :94-174 */
/* Blocks of block_size positions, block k = [(k-1)*block_size + 1, k*block_size] (RegionStateManager.GetBlockKey :404-407); call before
 * candidates are added.  Every block of the window counts as existing. */
void orc_track_blocks(OrcState* s, int32_t block_size)
{
    s->block_size = block_size;
    s->first_block_key = (s->start_position - 1) / block_size + 1;
    s->n_blocks = (s->start_position + s->n_loci - 2) / block_size + 1 - s->first_block_key + 1;
    free(s->max_allele_endpoint);
    s->max_allele_endpoint = (int32_t*)calloc((size_t)s->n_blocks, sizeof(int32_t));
}

int32_t orc_add_candidate(OrcState* s, const OrcCandidate* c)
{
    if (c->category == PISCES_CAT_REFERENCE) return PISCES_E_INVALID_ARG;
    if (!in_region(s, c->position)) return PISCES_E_INVALID_ARG;
    int li = c->position - s->start_position;
    if (s->block_size > 0) {   /* RegionState.UpdateMaxPosition :203-223 (before the merge, for every candidate handed in) */
        int otherEnd = 0;
        if (c->category == PISCES_CAT_DELETION) otherEnd = c->position + (int)strlen(c->ref);
        else if (c->category == PISCES_CAT_INSERTION) otherEnd = c->position + 1;
        else if (c->category == PISCES_CAT_MNV) otherEnd = c->position + (int)strlen(c->ref) - 1;
        int32_t* m = &s->max_allele_endpoint[(c->position - 1) / s->block_size + 1 - s->first_block_key];
        if (otherEnd > *m) *m = otherEnd;
    }
    for (int i = s->cand_head[li]; i >= 0; i = s->cands[i].next) {
        OrcCandidate* e = &s->cands[i];
        int match = candidate_equals(e, c);
        if (match && s->track_open_ended) match = (e->open_left == c->open_left && e->open_right == c->open_right);
        if (match) {
            for (int d = 0; d < 3; d++) {
                e->support_by_dir[d] += c->support_by_dir[d];
                e->well_anchored_by_dir[d] += c->well_anchored_by_dir[d];
            }
            return 0;
        }
    }
    if (s->n_cands == s->cap_cands) {
        s->cap_cands = s->cap_cands ? s->cap_cands * 2 : 64;
        s->cands = (OrcCandidate*)realloc(s->cands, sizeof(OrcCandidate) * (size_t)s->cap_cands);
    }
    int id = s->n_cands++;
    s->cands[id] = *c;
    s->cands[id].next = -1;
    if (s->cand_tail[li] >= 0) s->cands[s->cand_tail[li]].next = id;
    else s->cand_head[li] = id;
    s->cand_tail[li] = id;
    return 0;
}

/* The forced alleles of this chromosome (Factory.SelectForcedAllele, Factory.cs:270-286; SmallVariantCaller.CreateForcedAllelePos :49-77):
 * (position, ref, alt) with the category of SmallVariantCaller.GetAlleleCategory :141-150, kept in position order (a SortedList), alleles
 * of one position in the order given. */
void orc_set_forced_alleles(OrcState* s, const OrcCandidate* list, int32_t n)
{
    free(s->forced);
    s->forced = (OrcCandidate*)calloc((size_t)(n > 0 ? n : 1), sizeof(OrcCandidate));
    s->n_forced = n;
    s->n_forced_added = 0;
    for (int i = 0; i < n; i++) {
        OrcCandidate c;
        memset(&c, 0, sizeof(c));
        c.position = list[i].position;
        strcpy(c.ref, list[i].ref);
        strcpy(c.alt, list[i].alt);
        const size_t rl = strlen(c.ref), al = strlen(c.alt);
        c.category = (rl == 1 && al == 1) ? PISCES_CAT_SNV : rl == al ? PISCES_CAT_MNV : rl > al ? PISCES_CAT_DELETION : PISCES_CAT_INSERTION;
        c.next = -1;
        int k = i;   /* stable insertion by position */
        while (k > 0 && s->forced[k - 1].position > c.position) { s->forced[k] = s->forced[k - 1]; k--; }
        s->forced[k] = c;
    }
}

/* SmallVariantCaller.AddForcedAlleleAsCandidate :118-132: the forced alleles up to upToPosition (< 0 = null: all that are left) become
 * candidates without support (IStateManager.AddCandidates), once */
void orc_add_forced_as_candidates(OrcState* s, int32_t up_to_position)
{
    while (s->n_forced_added < s->n_forced) {
        const OrcCandidate* c = &s->forced[s->n_forced_added];
        if (up_to_position >= 0 && c->position > up_to_position) break;
        (void)orc_add_candidate(s, c);
        s->n_forced_added++;
    }
}

static int is_forced_allele(const OrcState* s, const OrcCalled* a)   /* AlleleCaller.IsForcedAllele :179-184 */
{
    for (int i = 0; i < s->n_forced; i++)
        if (s->forced[i].position == a->position && strcmp(s->forced[i].ref, a->ref) == 0 && strcmp(s->forced[i].alt, a->alt) == 0) return 1;
    return 0;
}

int32_t orc_num_candidates(const OrcState* s) { return s->n_cands; }

int32_t orc_get_candidates(const OrcState* s, OrcCandidate* out, int32_t capacity)
{
    int n = 0;
    for (int li = 0; li < s->n_loci; li++)
        for (int i = s->cand_head[li]; i >= 0; i = s->cands[i].next) {
            if (n < capacity) out[n] = s->cands[i];
            n++;
        }
    return n;
}

/* =====================================================================================
 * lib/Pisces.Domain/Logic/CandidateVariantFinder.cs
 * ===================================================================================== */
typedef struct FinderCtx {
    const OrcRead* r;
    const uint8_t* ref;
    int64_t ref_len;
    int min_bq, max_mnv, max_gap, call_mnvs, anchor_size;
    OrcCandidate* out;
    int n, cap;
    int overflow;
} FinderCtx;

/* Read.SequencedIndexesToExpandedIndexes (Read.cs:432-476), one index: the position of sequenced base idx in the expanded CIGAR */
static int sequenced_index_to_expanded(const OrcRead* r, int idx)
{
    int extendedIndex = 0, sequencedBaseIndex = 0;
    for (int c = 0; c < r->n_cigar; c++)
        for (uint32_t k = 0; k < r->cigar_len[c]; k++, extendedIndex++)
            if (op_is_read_span(r->cigar_op[c])) {
                if (sequencedBaseIndex == idx) return extendedIndex;
                sequencedBaseIndex++;
            }
    return -1;
}

/* GetDeletionDirectionForStitchedRead :468-487: the directions one step inside the deletion from either anchor, read from the
 * expanded direction map.  -1 where the reference throws (indexes outside the map). */
int32_t orc_deletion_direction_for_stitched_read(const OrcRead* r, int32_t leftAnchorIndexInSequencedRead, int32_t rightAnchorIndexInSequencedRead)
{
    int first = sequenced_index_to_expanded(r, leftAnchorIndexInSequencedRead) + 1;
    int last = sequenced_index_to_expanded(r, rightAnchorIndexInSequencedRead) - 1;
    if (first >= 0 && first < r->n_expanded && last >= 0 && last < r->n_expanded) {
        int startDirection = r->expanded_dirs[first], endDirection = r->expanded_dirs[last];
        return startDirection == PISCES_DIR_STITCHED ? endDirection : startDirection;
    }
    return -1;
}

/* GetSupportDirection :396-445 */
static int get_support_direction(const FinderCtx* f, int category, int length, int startIndexInRead)
{
    const OrcRead* r = f->r;
    if (category == PISCES_CAT_SNV || category == PISCES_CAT_REFERENCE) return read_dir(r, startIndexInRead);
    int leftAnchorIndex = startIndexInRead - 1;
    int rightAnchorIndex = category == PISCES_CAT_DELETION ? startIndexInRead : startIndexInRead + length;
    int lastIndex = r->read_len - 1;
    if (rightAnchorIndex == 0) return read_dir(r, rightAnchorIndex);
    if (leftAnchorIndex == lastIndex) return read_dir(r, lastIndex);
    if (leftAnchorIndex == rightAnchorIndex - 1) {
        if (r->expanded_dirs) {
            int d = orc_deletion_direction_for_stitched_read(r, leftAnchorIndex, rightAnchorIndex);
            if (d >= 0 && d <= 2) return d;
        }
        int startDirection = read_dir(r, leftAnchorIndex);
        int endDirection = read_dir(r, rightAnchorIndex);
        return startDirection == PISCES_DIR_STITCHED ? endDirection : startDirection;
    }
    int direction = PISCES_DIR_FORWARD;
    for (int i = leftAnchorIndex + 1; i < rightAnchorIndex; i++) {
        direction = read_dir(r, i);
        if (direction == PISCES_DIR_STITCHED) return PISCES_DIR_STITCHED;
    }
    return direction;
}

static int allele_length(int category, const char* ref, const char* alt)
{
    /* BaseAllele.Length (BaseAllele.cs:26-46) */
    switch (category) {
    case PISCES_CAT_MNV:
    case PISCES_CAT_SNV: return (int)strlen(alt);
    case PISCES_CAT_INSERTION: return (int)strlen(alt) - 1;
    case PISCES_CAT_DELETION: return (int)strlen(ref) - 1;
    default: return (int)strlen(ref);
    }
}

/* CandidateVariantFinder.Create :334-387 */
static OrcCandidate* finder_create(FinderCtx* f, int category, int coordinate, const char* ref, int ref_n,
                                   const char* alt, int alt_n, int startIndexInRead)
{
    if (f->n >= f->cap || ref_n >= ORC_MAX_ALLELE || alt_n >= ORC_MAX_ALLELE) { f->overflow = 1; return NULL; }
    OrcCandidate* c = &f->out[f->n++];
    memset(c, 0, sizeof(*c));
    c->position = coordinate;
    c->category = category;
    memcpy(c->ref, ref, (size_t)ref_n); c->ref[ref_n] = 0;
    memcpy(c->alt, alt, (size_t)alt_n); c->alt[alt_n] = 0;
    c->next = -1;
    int dir = get_support_direction(f, category, allele_length(category, c->ref, c->alt), startIndexInRead);
    c->support_by_dir[dir]++;
    int endPos = read_end_position(f->r);
    int a1 = coordinate - f->r->position, a2 = endPos - coordinate;
    int anchor = a1 < a2 ? a1 : a2;
    int lim = (f->anchor_size - 1) < (alt_n - 1) ? (f->anchor_size - 1) : (alt_n - 1);
    if (anchor > lim) c->well_anchored_by_dir[dir]++;
    return c;
}

/* FlushVariant :183-203 */
static void flush_variant(FinderCtx* f, int variantStartIndexInRead, int variantStartIndexInReference,
                          int variantLengthSoFar, int interveningRefLengthSoFar, int openLeft, int openRight)
{
    if (interveningRefLengthSoFar >= 1) {
        variantLengthSoFar -= interveningRefLengthSoFar;
        openRight = 0;
    }
    if (variantLengthSoFar >= 1) {
        int cat = variantLengthSoFar > 1 ? PISCES_CAT_MNV : PISCES_CAT_SNV;
        OrcCandidate* c = finder_create(f, cat, variantStartIndexInReference + 1,
                                        (const char*)f->ref + variantStartIndexInReference, variantLengthSoFar,
                                        (const char*)f->r->bases + variantStartIndexInRead, variantLengthSoFar,
                                        variantStartIndexInRead);
        if (c) { c->open_left = openLeft; c->open_right = openRight; }
    }
}

/* ShouldBuildUpMNV :170-181 */
static int should_build_up_mnv(const FinderCtx* f, int mnvLengthSoFar, int interveningRefLengthSoFar, int refCallNext)
{
    if (!f->call_mnvs) return 0;
    if (refCallNext && mnvLengthSoFar == 0) return 0;
    if ((mnvLengthSoFar + 1) > f->max_mnv) return 0;
    if ((interveningRefLengthSoFar + (refCallNext ? 1 : 0)) > f->max_gap) return 0;
    return 1;
}

/* ExtractSnvsFromOperation :90-168 */
static void extract_snvs(FinderCtx* f, int opStartIndexInRead, uint32_t operationLength, int opStartIndexInReference)
{
    const OrcRead* r = f->r;
    int variantLengthSoFar = 0, interveningRefLengthSoFar = 0, openLeft = 0;
    /* An M operation that runs past the end of the contig stops there (:103-104).  The reference then flushes a pending variant
     * from `operationLength`, i.e. from indices beyond both strings, and Substring throws; here the flush uses the bases actually
     * walked (n_done), so the pending variant comes out at its own coordinates and nothing is read out of range. */
    int n_done = (int)operationLength;
    for (int i = 0; i < (int)operationLength; i++) {
        int qualityGoodEnough = r->quals[opStartIndexInRead + i] >= f->min_bq;
        uint8_t readBase = r->bases[opStartIndexInRead + i];
        if (opStartIndexInReference + i >= f->ref_len) { n_done = i; break; }
        uint8_t refBase = f->ref[opStartIndexInReference + i];
        int atEndOfOperation = i == ((int)operationLength - 1);
        int startingMnvAtEndOfOperation = (atEndOfOperation && variantLengthSoFar == 0);

        if ((allele_type_of(readBase) == PISCES_ALLELE_N) || (allele_type_of(refBase) == PISCES_ALLELE_N) || !qualityGoodEnough) {
            flush_variant(f, opStartIndexInRead + i - variantLengthSoFar, opStartIndexInReference + i - variantLengthSoFar,
                          variantLengthSoFar, interveningRefLengthSoFar, openLeft, 1);
            variantLengthSoFar = 0; interveningRefLengthSoFar = 0; openLeft = 1;
        } else if (refBase == readBase) {
            if (should_build_up_mnv(f, variantLengthSoFar, interveningRefLengthSoFar, 1) && !startingMnvAtEndOfOperation) {
                variantLengthSoFar++; interveningRefLengthSoFar++;
            } else {
                flush_variant(f, opStartIndexInRead + i - variantLengthSoFar, opStartIndexInReference + i - variantLengthSoFar,
                              variantLengthSoFar, interveningRefLengthSoFar, openLeft, 0);
                variantLengthSoFar = 0; interveningRefLengthSoFar = 0; openLeft = 0;
            }
        } else {
            if (should_build_up_mnv(f, variantLengthSoFar, interveningRefLengthSoFar, 0) && !startingMnvAtEndOfOperation) {
                variantLengthSoFar++; interveningRefLengthSoFar = 0;
            } else {
                flush_variant(f, opStartIndexInRead + i - variantLengthSoFar, opStartIndexInReference + i - variantLengthSoFar,
                              variantLengthSoFar, interveningRefLengthSoFar, openLeft, 0);
                variantLengthSoFar = 1; interveningRefLengthSoFar = 0; openLeft = 0;
            }
        }
    }
    flush_variant(f, opStartIndexInRead + n_done - variantLengthSoFar,
                  opStartIndexInReference + n_done - variantLengthSoFar, variantLengthSoFar,
                  interveningRefLengthSoFar, openLeft, 0);
}

/* ExtractInsertionFromOperation :234-260 */
static void extract_insertion(FinderCtx* f, int opStartIndexInRead, uint32_t operationLength, int opStartIndexInReference)
{
    if (opStartIndexInReference - 1 >= f->ref_len || opStartIndexInReference == 0) return;
    int n = (int)operationLength;
    if (n + 1 >= ORC_MAX_ALLELE) { f->overflow = 1; return; }
    char alt[ORC_MAX_ALLELE];
    alt[0] = (char)f->ref[opStartIndexInReference - 1];
    memcpy(alt + 1, f->r->bases + opStartIndexInRead, (size_t)n);
    if (!(f->r->quals[opStartIndexInRead] >= f->min_bq)) return;
    finder_create(f, PISCES_CAT_INSERTION, opStartIndexInReference, (const char*)f->ref + opStartIndexInReference - 1, 1,
                  alt, n + 1, opStartIndexInRead);
}

/* ExtractDeletionFromOperation :262-292 */
static void extract_deletion(FinderCtx* f, int opStartIndexInRead, uint32_t operationLength, int opStartIndexInReference)
{
    if ((int64_t)opStartIndexInReference + (int64_t)operationLength >= f->ref_len) return;
    if (opStartIndexInReference - 1 < 0) return; /* C# Substring(-1, ..) would throw; reads never start with D at ref index 0 */
    if (!orc_check_deletion_quality(f->r, opStartIndexInRead, f->min_bq)) return;
    finder_create(f, PISCES_CAT_DELETION, opStartIndexInReference, (const char*)f->ref + opStartIndexInReference - 1,
                  (int)operationLength + 1, (const char*)f->ref + opStartIndexInReference - 1, 1, opStartIndexInRead);
}

/* PositionMap.MaxPosition */
static int posmap_max(const int32_t* posmap, int n)
{
    int m = -1;
    for (int i = 0; i < n; i++) if (posmap[i] > m) m = posmap[i];
    return m;
}

/* ProcessCigarOps :36-83, Annotate :496-553 */
int32_t orc_find_candidates(const OrcRead* r, const uint8_t* ref, int64_t ref_len, int32_t min_bq, int32_t max_mnv,
                            int32_t max_gap, int32_t call_mnvs, int32_t anchor_size, OrcCandidate* out, int32_t cap)
{
    FinderCtx f = {r, ref, ref_len, min_bq, max_mnv, max_gap, call_mnvs, anchor_size, out, 0, cap, 0};
    int startIndexInRead = 0;
    int startIndexInReference = r->position - 1;
    for (int c = 0; c < r->n_cigar; c++) {
        uint8_t t = r->cigar_op[c];
        uint32_t len = r->cigar_len[c];
        switch (t) {
        case 'S': break;
        case 'M': extract_snvs(&f, startIndexInRead, len, startIndexInReference); break;
        case 'I': extract_insertion(&f, startIndexInRead, len, startIndexInReference); break;
        case 'D': extract_deletion(&f, startIndexInRead, len, startIndexInReference); break;
        default: break;
        }
        if (op_is_read_span(t)) startIndexInRead += (int)len;
        if (op_is_ref_span(t)) startIndexInReference += (int)len;
    }
    if (f.overflow) return PISCES_E_BUFFER_TOO_SMALL;
    if (f.n == 0) return 0;

    /* Annotate */
    int fi = 0, li = r->n_cigar - 1;
    if (r->cigar_op[fi] == 'S') fi = 1;
    if (r->cigar_op[li] == 'S') li = r->n_cigar - 2;
    if (fi >= r->n_cigar || li < 0) return f.n;
    int32_t stack_map[512];
    int32_t* posmap = r->read_len <= 512 ? stack_map : (int32_t*)malloc(sizeof(int32_t) * (size_t)r->read_len);
    build_position_map(r, posmap);
    int maxPosition = posmap_max(posmap, r->read_len);
    if (posmap != stack_map) free(posmap);
    if (maxPosition == -1) maxPosition = r->position - 1;
    uint8_t firstOp = r->cigar_op[fi], lastOp = r->cigar_op[li];
    for (int i = 0; i < f.n; i++) {
        OrcCandidate* c = &out[i];
        int isSnvMnv = (c->category == PISCES_CAT_MNV || c->category == PISCES_CAT_SNV);
        switch (firstOp) {
        case 'M': if (c->position == r->position && isSnvMnv) c->open_left = 1; break;
        case 'I': if (c->position == r->position - 1 && c->category == PISCES_CAT_INSERTION) c->open_left = 1; break;
        case 'D': if (c->position == r->position - 1 && c->category == PISCES_CAT_DELETION) c->open_left = 1; break;
        default: break;
        }
        switch (lastOp) {
        case 'M': if (c->position + (int)strlen(c->alt) - 1 == maxPosition && isSnvMnv) c->open_right = 1; break;
        case 'I': if (c->position == maxPosition && c->category == PISCES_CAT_INSERTION) c->open_right = 1; break;
        case 'D': if (c->position == maxPosition && c->category == PISCES_CAT_DELETION) c->open_right = 1; break;
        default: break;
        }
    }
    return f.n;
}

/* =====================================================================================
 * lib/Pisces.Calculators/CoverageCalculator.cs
 * ===================================================================================== */
static const int kCoverageContributing[5] = {PISCES_ALLELE_A, PISCES_ALLELE_C, PISCES_ALLELE_G, PISCES_ALLELE_T,
                                              PISCES_ALLELE_DEL}; /* Constants.cs:41-44 */

/* CalculateSinglePoint :49-98 */
static void coverage_single_point(OrcCalled* a, const OrcState* s)
{
    int refType = allele_type_of((uint8_t)a->ref[0]); /* AlleleHelper.GetAlleleType(string) needs length 1 */
    for (int direction = 0; direction < 3; direction++) {
        for (int k = 0; k < 5; k++) {
            int alleleType = kCoverageContributing[k];
            a->coverage_by_dir[direction] += orc_get_allele_count(s, a->position, alleleType, direction, 0, -1, 0, 0);
            a->sum_of_base_quality += orc_get_sum_base_quality(s, a->position, alleleType, direction, 0, -1, 0, 0);
            if (alleleType != refType) continue;
            a->reference_support += orc_get_allele_count(s, a->position, alleleType, direction, 0, -1, 0, 0);
        }
        a->total_coverage += a->coverage_by_dir[direction];
        a->confident_start += a->coverage_by_dir[direction];
        a->confident_end += a->coverage_by_dir[direction];
        a->num_no_calls += orc_get_allele_count(s, a->position, PISCES_ALLELE_N, direction, 0, -1, 0, 0);
    }
    int gappedRefCounts = get_gapped_mnv_ref(s, a->position);
    if (a->category == PISCES_CAT_SNV) {
        int v = a->reference_support - gappedRefCounts;
        a->reference_support = v > 0 ? v : 0;
    } else if (a->category == PISCES_CAT_REFERENCE) {
        int v = a->allele_support - gappedRefCounts;
        a->allele_support = v > 0 ? v : 0;
    }
}

/* RedistributeStitchedCoverage :324-331 */
static void redistribute_stitched(int dataPoint[3])
{
    int stitchedCoverage = dataPoint[PISCES_DIR_STITCHED];
    dataPoint[PISCES_DIR_FORWARD] += (int)ceilf((float)stitchedCoverage / 2);
    dataPoint[PISCES_DIR_REVERSE] += (int)floorf((float)stitchedCoverage / 2);
    dataPoint[PISCES_DIR_STITCHED] = 0;
}

/* CalculateSpanning :162-321 */
static void coverage_spanning(OrcCalled* a, const OrcState* s, int startPointPosition, int endPointPosition,
                              int presumeAnchoredForExactCov, int considerAnchorInformation)
{
    int startPointCoverage[3] = {0, 0, 0}, endPointCoverage[3] = {0, 0, 0};
    float exactTotalCoverage = 0.0f;
    int confidentCoverageLeft = 0, confidentCoverageRight = 0, suspiciousCoverageLeft = 0, suspiciousCoverageRight = 0;
    int firstBase = PISCES_ALLELE_N, lastBase = PISCES_ALLELE_N;
    int bePickyAboutAnchors = considerAnchorInformation && a->category == PISCES_CAT_INSERTION;
    int alleleLength = allele_length(a->category, a->ref, a->alt);
    if (bePickyAboutAnchors) {
        firstBase = allele_type_of((uint8_t)a->alt[1]);
        lastBase = allele_type_of((uint8_t)a->alt[strlen(a->alt) - 1]);
    }
    int startPointCoverageUnanchored[3] = {0, 0, 0}, endPointCoverageUnanchored[3] = {0, 0, 0};
    double unanchoredCoverageStartQuality = 0, unanchoredCoverageEndQuality = 0;
    int unanchoredSupport = a->allele_support - a->well_anchored_support;

    for (int directionIndex = 0; directionIndex < 3; directionIndex++) {
        for (int k = 0; k < 5; k++) {
            int alleleType = kCoverageContributing[k];
            int anchoredCoverageOnlyEnd = bePickyAboutAnchors && alleleType == firstBase;
            int anchoredCoverageOnlyStart = bePickyAboutAnchors && alleleType == lastBase;
            int minAnchorEnd = anchoredCoverageOnlyEnd ? alleleLength : 0;
            int minAnchorStart = anchoredCoverageOnlyStart ? alleleLength : 0;

            int startCov = orc_get_allele_count(s, startPointPosition, alleleType, directionIndex, minAnchorStart, -1, 0, 0);
            startPointCoverage[directionIndex] += startCov;
            int endCov = orc_get_allele_count(s, endPointPosition, alleleType, directionIndex, minAnchorEnd, -1, 1, 0);
            endPointCoverage[directionIndex] += endCov;
            confidentCoverageLeft += startCov;
            confidentCoverageRight += endCov;
            a->sum_of_base_quality += orc_get_sum_base_quality(s, startPointPosition, alleleType, directionIndex, minAnchorStart, -1, 0, 0);
            a->sum_of_base_quality += orc_get_sum_base_quality(s, endPointPosition, alleleType, directionIndex, minAnchorEnd, -1, 1, 0);

            if (bePickyAboutAnchors && unanchoredSupport > 0) {
                if (minAnchorStart > 0) {
                    int u = orc_get_allele_count(s, startPointPosition, alleleType, directionIndex, 0, minAnchorStart - 1, 0, 0);
                    startPointCoverageUnanchored[directionIndex] += u;
                    suspiciousCoverageLeft += u;
                    unanchoredCoverageStartQuality += orc_get_sum_base_quality(s, startPointPosition, alleleType, directionIndex, 0, minAnchorStart - 1, 0, 0);
                }
                if (minAnchorEnd > 0) {
                    int u = orc_get_allele_count(s, endPointPosition, alleleType, directionIndex, 0, minAnchorEnd - 1, 1, 0);
                    endPointCoverageUnanchored[directionIndex] += u;
                    suspiciousCoverageRight += u;
                    /* reference reads startPointPosition here (:254) — reproduced */
                    unanchoredCoverageEndQuality += orc_get_sum_base_quality(s, startPointPosition, alleleType, directionIndex, 0, minAnchorEnd - 1, 1, 0);
                }
            }
        }
    }

    if (bePickyAboutAnchors) {
        float trulyAnchoredCoverage = (((confidentCoverageLeft - suspiciousCoverageRight) +
                                        (confidentCoverageRight - suspiciousCoverageLeft)) / 2.0f);
        float anchoredVariantFreq = trulyAnchoredCoverage <= 0 ? 0 : (float)a->well_anchored_support / trulyAnchoredCoverage;
        int totalSuspiciousCoverage = suspiciousCoverageLeft + suspiciousCoverageRight;
        float unanchoredVariantFreq = totalSuspiciousCoverage == 0 ? 0 : unanchoredSupport / ((float)totalSuspiciousCoverage);
        /* Math.Max(0, anchoredVariantFreq == 0 ? 1 : Math.Min(1, unanchoredVariantFreq / anchoredVariantFreq)) in float */
        float w = anchoredVariantFreq == 0 ? 1.0f : fminf(1.0f, unanchoredVariantFreq / anchoredVariantFreq);
        if (!(w > 0.0f)) w = 0.0f;
        double variantSpecificUnanchoredWeight = w;
        a->unanchored_weight = variantSpecificUnanchoredWeight;
        for (int d = 0; d < 3; d++) {
            startPointCoverage[d] += (int)(startPointCoverageUnanchored[d] * variantSpecificUnanchoredWeight);
            endPointCoverage[d] += (int)(endPointCoverageUnanchored[d] * variantSpecificUnanchoredWeight);
            a->sum_of_base_quality += unanchoredCoverageStartQuality * variantSpecificUnanchoredWeight;
            a->sum_of_base_quality += unanchoredCoverageEndQuality * variantSpecificUnanchoredWeight;
        }
    }

    redistribute_stitched(startPointCoverage);
    redistribute_stitched(endPointCoverage);

    for (int d = 0; d < 2; d++) {
        float exactCoverageForDir = presumeAnchoredForExactCov
            ? (startPointCoverage[d] + endPointCoverage[d]) / 2.0f
            : (float)(startPointCoverage[d] < endPointCoverage[d] ? startPointCoverage[d] : endPointCoverage[d]);
        a->coverage_by_dir[d] = (int)exactCoverageForDir;
        exactTotalCoverage += exactCoverageForDir;
    }
    a->total_coverage = (int)exactTotalCoverage;
    int rs = a->total_coverage - a->allele_support;
    a->reference_support = rs > 0 ? rs : 0;
    a->suspicious_start = suspiciousCoverageLeft;
    a->confident_start = confidentCoverageLeft;
    a->suspicious_end = suspiciousCoverageRight;
    a->confident_end = confidentCoverageRight;
}

/* Compute :19-47 */
void orc_coverage_compute(OrcCalled* a, const OrcState* s, int32_t considerAnchors, int32_t expectStitched)
{
    int len = allele_length(a->category, a->ref, a->alt);
    switch (a->category) {
    case PISCES_CAT_REFERENCE: coverage_single_point(a, s); break;
    case PISCES_CAT_DELETION: coverage_spanning(a, s, a->position + 1, a->position + len, 1, considerAnchors); break;
    case PISCES_CAT_MNV: coverage_spanning(a, s, a->position, a->position + len - 1, 1, considerAnchors); break;
    case PISCES_CAT_INSERTION: coverage_spanning(a, s, a->position, a->position + 1, expectStitched, considerAnchors); break;
    default: coverage_single_point(a, s); break;
    }
}

/* AlleleHelper.Map(CandidateAllele) :51-85 */
void orc_called_from_candidate(OrcCalled* v, const OrcCandidate* c)
{
    memset(v, 0, sizeof(*v));
    v->position = c->position;
    v->category = c->category;
    strcpy(v->ref, c->ref);
    strcpy(v->alt, c->alt);
    for (int d = 0; d < 3; d++) {
        v->support_by_dir[d] = c->support_by_dir[d];
        v->well_anchored_by_dir[d] = c->well_anchored_by_dir[d];
        v->allele_support += c->support_by_dir[d];
        v->well_anchored_support += c->well_anchored_by_dir[d];
    }
    /* CalledAllele(AlleleCategory) ctor, CalledAllele.cs:142-155 */
    v->genotype = (c->category == PISCES_CAT_REFERENCE) ? PISCES_GT_HOM_REF : PISCES_GT_HET_ALT_REF;
}

/* RMxNCalculator.ComputeRMxNLengthForIndel (RMxNCalculator.cs:50-94) */
static int rmxn_length_for_indel(int variantPosition, const char* variantBases, int length, const uint8_t* ref,
                                 int64_t ref_len, int maxRepeatUnitLength)
{
    int maxRepeatsFound = 0;
    int lo = length - (maxRepeatUnitLength < length ? maxRepeatUnitLength : length);
    for (int pass = 0; pass < 2; pass++) {
        for (int i = lo; i < length; i++) {
            int blen = length - i;
            const char* bookend = pass == 0 ? variantBases : variantBases + i;
            int64_t backPeekPosition = variantPosition;
            while (1) {
                int64_t nb = backPeekPosition - blen;
                if (nb < 0) break;
                if (nb + blen > ref_len || memcmp(bookend, ref + nb, (size_t)blen) != 0) break;
                backPeekPosition = nb;
            }
            int repeatCount = 0;
            int64_t currentPosition = backPeekPosition;
            while (1) {
                if (currentPosition + blen > ref_len) break;
                if (memcmp(bookend, ref + currentPosition, (size_t)blen) != 0) break;
                repeatCount++;
                currentPosition += blen;
            }
            if (repeatCount > maxRepeatsFound) maxRepeatsFound = repeatCount;
        }
    }
    return maxRepeatsFound;
}

/* RMxNCalculator.ShouldFilter :19-38 with the default settings 5 x 9 @ 0.35f
 * (VariantCallingParameters.cs:76-80) carried in the config */
static int rmxn_should_filter(const OrcCalled* a, const uint8_t* ref, int64_t ref_len, const PiscesHipConfig* cfg)
{
    const int maxLen = cfg->rmxn_max_repeat_length, minRep = cfg->rmxn_min_repetitions;
    const float freqLimit = cfg->rmxn_frequency_limit;
    if (!ref || maxLen < 0) return 0;
    if (frequency_f(a->allele_support, a->total_coverage) >= freqLimit) return 0;
    int c1, c2 = 2147483647;
    if (a->category == PISCES_CAT_INSERTION) {
        c1 = rmxn_length_for_indel(a->position, a->alt + 1, (int)strlen(a->alt) - 1, ref, ref_len, maxLen);
    } else if (a->category == PISCES_CAT_DELETION) {
        c1 = rmxn_length_for_indel(a->position, a->ref + 1, (int)strlen(a->ref) - 1, ref, ref_len, maxLen);
    } else {
        int rl = (int)strlen(a->ref), al = (int)strlen(a->alt);
        c1 = rmxn_length_for_indel(a->position - 1, a->ref, rl, ref, ref_len, maxLen);
        int i1 = rmxn_length_for_indel(a->position + rl - 1, a->alt, al, ref, ref_len, maxLen);
        int i2 = rmxn_length_for_indel(a->position - 1, a->alt, al, ref, ref_len, maxLen);
        c2 = i1 > i2 ? i1 : i2;
    }
    return (c1 < c2 ? c1 : c2) >= minRep;
}

static const uint8_t* g_rmxn_ref = NULL; /* set per orc_call_all; oracle is single-threaded per call */
static int64_t g_rmxn_ref_len = 0;

/* AlleleCaller.ProcessVariant :208-234 + AlleleProcessor.Process/ApplyFilters (AlleleProcessor.cs:16-71) */
void orc_process_variant(OrcCalled* v, const OrcState* s, const PiscesHipConfig* cfg)
{
    /* coverage accumulates with += : re-processing a variant double counts, like the reference does
     * for MNVs (AlleleCaller.cs:74,111); callers reset what they need. */
    orc_coverage_compute(v, s, /*considerAnchorInformation: TrackedAnchorSize > 0*/ s->num_anchor_types > 0,
                         cfg->expect_stitched_reads);

    if (v->allele_support > 0) {
        /* VariantQualityCalculator.Compute :11-24 */
        if (cfg->noise_model == PISCES_NOISE_WINDOW) {
            /* AlleleCaller.cs:215-218: (int)MathOperations.PtoQ(variant.SumOfBaseQuality / variant.TotalCoverage).  A mean error that
             * is not a positive finite number (coverage made of deletions only: no base qualities) makes PtoQ +inf, the C# cast gives
             * int.MinValue, QtoP of that +inf, and MathNet's Poisson(+inf).CumulativeDistribution is 0: p = 1, Q = 0. */
            double mean = v->sum_of_base_quality / v->total_coverage;
            if (v->total_coverage == 0 || !(mean > 0.0) || isinf(mean)) {
                v->noise_level_applied = INT32_MIN;
                v->variant_qscore = 0;
            } else {
                v->noise_level_applied = (int32_t)(-10.0 * log10(mean));
                v->variant_qscore = orc_poisson_qscore(v->allele_support, v->total_coverage, v->noise_level_applied, cfg->max_variant_qscore);
            }
        } else {
        v->noise_level_applied = cfg->noise_level;
        if (v->total_coverage == 0) v->variant_qscore = 0;
        else v->variant_qscore = orc_poisson_qscore(v->allele_support, v->total_coverage, cfg->noise_level, cfg->max_variant_qscore);
        }
        orc_strand_bias(v->coverage_by_dir, v->support_by_dir, cfg->noise_level, (double)cfg->min_frequency,
                        (double)cfg->strand_bias_threshold, cfg->strand_bias_model, &v->sb);
        v->has_sb = 1;
    }

    /* SetFractionNoCalls CalledAllele.cs:107-114 */
    float allReads = (float)(v->total_coverage + v->num_no_calls);
    v->fraction_no_calls = allReads == 0 ? 0.0f : ((float)v->num_no_calls / allReads);

    /* ApplyFilters :25-71 */
    v->filters = 0;
    if (cfg->low_depth_filter >= 0 && v->total_coverage < cfg->low_depth_filter) v->filters |= 1u << PISCES_FILTER_LOW_DEPTH;
    if (cfg->variant_qscore_filter >= 0 && v->variant_qscore < cfg->variant_qscore_filter && (v->total_coverage != 0))
        v->filters |= 1u << PISCES_FILTER_LOW_VARIANT_QSCORE;
    if (v->category != PISCES_CAT_REFERENCE) {
        if (cfg->no_call_filter_threshold >= 0 && v->fraction_no_calls > cfg->no_call_filter_threshold)
            v->filters |= 1u << PISCES_FILTER_NO_CALL;
        int biasAcceptable = v->has_sb ? v->sb.bias_acceptable : 0; /* new BiasResults(): false */
        int varBoth = v->has_sb ? v->sb.var_present_on_both : 0;
        if (!biasAcceptable || (cfg->filter_single_strand && !varBoth)) v->filters |= 1u << PISCES_FILTER_STRAND_BIAS;
        if (rmxn_should_filter(v, g_rmxn_ref, g_rmxn_ref_len, cfg)) v->filters |= 1u << PISCES_FILTER_RMXN;
        if (cfg->variant_freq_filter >= 0 && frequency_f(v->allele_support, v->total_coverage) < cfg->variant_freq_filter)
            v->filters |= 1u << PISCES_FILTER_LOW_VARIANT_FREQUENCY;
        if (cfg->expect_stitched_reads && strchr(v->alt, 'N')) v->filters |= 1u << PISCES_FILTER_STRAND_BIAS;
    }
}

/* AlleleCaller.IsCallable :236-258 */
static int is_callable(const OrcCalled* a, const PiscesHipConfig* cfg, int64_t* totalNumCalled)
{
    if (a->category == PISCES_CAT_REFERENCE) { (*totalNumCalled)++; return 1; }
    if (a->total_coverage < cfg->min_coverage && !cfg->include_reference_calls) return 0;
    if (a->total_coverage != 0 && frequency_f(a->allele_support, a->total_coverage) < cfg->min_frequency) return 0;
    if (a->variant_qscore < cfg->min_variant_qscore) return 0;
    (*totalNumCalled)++;
    return 1;
}

static int called_cmp(const void* pa, const void* pb)
{
    const OrcCalled* a = (const OrcCalled*)pa;
    const OrcCalled* b = (const OrcCalled*)pb;
    if (a->position != b->position) return a->position < b->position ? -1 : 1;
    int r = strcmp(a->ref, b->ref);
    return r ? r : strcmp(a->alt, b->alt);
}

static void to_record(PiscesCalledAllele* o, const OrcCalled* v)
{
    memset(o, 0, sizeof(*o));
    o->position = v->position;
    o->total_coverage = v->total_coverage;
    o->allele_support = v->allele_support;
    o->reference_support = v->reference_support;
    o->num_no_calls = v->num_no_calls;
    for (int d = 0; d < 3; d++) { o->coverage_by_dir[d] = v->coverage_by_dir[d]; o->support_by_dir[d] = v->support_by_dir[d]; }
    o->variant_qscore = v->variant_qscore;
    o->strand_bias_score = v->has_sb ? v->sb.bias_score : 0.0;
    o->genotype_qscore = (int16_t)v->genotype_qscore;
    o->noise_level = v->noise_level_applied == INT32_MIN ? INT16_MIN : (int16_t)v->noise_level_applied;
    o->filter_bits = (uint16_t)v->filters;
    int single = (strlen(v->ref) == 1 && strlen(v->alt) == 1);
    int refc = allele_type_of((uint8_t)v->ref[0]);
    int altc = single ? allele_type_of((uint8_t)v->alt[0]) : PISCES_ALLELE_N;
    o->info = PISCES_INFO_PACK(v->genotype, v->category, refc, altc, v->has_sb ? v->sb.bias_acceptable : 0,
                               v->has_sb ? v->sb.var_present_on_both : 0, v->has_sb ? v->sb.cov_present_on_both : 0);
}

/* =====================================================================================
 * exe/Pisces/Logic/VariantCalling/VariantCollapser.cs
 * ===================================================================================== */
static int cand_length(const OrcCandidate* c) { return allele_length(c->category, c->ref, c->alt); }
static int cand_support(const OrcCandidate* c) { return c->support_by_dir[0] + c->support_by_dir[1] + c->support_by_dir[2]; }
static int cand_fully_anchored(const OrcCandidate* c) { return !c->open_left && !c->open_right; }

/* CanCollapse :119-174 */
static int can_collapse(const OrcCandidate* toCollapse, const OrcCandidate* potentialMatch)
{
    const int ti = toCollapse->category == PISCES_CAT_INSERTION, pi = potentialMatch->category == PISCES_CAT_INSERTION;
    const int td = toCollapse->category == PISCES_CAT_DELETION, pd = potentialMatch->category == PISCES_CAT_DELETION;
    if ((ti && !pi) || (!ti && pi) || (td && !pd) || (!td && pd) || cand_length(toCollapse) > cand_length(potentialMatch) ||
        (cand_fully_anchored(toCollapse) && !cand_fully_anchored(potentialMatch)))
        return 0;
    const char* tb = td ? toCollapse->ref : toCollapse->alt;
    const char* pb = pd ? potentialMatch->ref : potentialMatch->alt;
    const int tl = (int)strlen(tb), pl = (int)strlen(pb);
    if (cand_fully_anchored(toCollapse) && cand_fully_anchored(potentialMatch)) return candidate_equals(toCollapse, potentialMatch);
    if (td) {
        if (toCollapse->open_right) return potentialMatch->position + 1 == toCollapse->position + 1;
        return potentialMatch->position + pl - 1 == toCollapse->position + tl - 1;
    }
    if (toCollapse->open_right) return potentialMatch->position == toCollapse->position && pl >= tl && strncmp(pb, tb, (size_t)tl) == 0;
    if (ti) {
        /* anchored on right: Substring(pl - tl + 1) == toCollapseBases.Substring(1); throws (never reached in practice) when negative */
        if (potentialMatch->position + 1 != toCollapse->position + 1) return 0;
        return pl - tl + 1 >= 0 && strcmp(pb + (pl - tl + 1), tb + 1) == 0;
    }
    const int tal = (int)strlen(toCollapse->alt), pal = (int)strlen(potentialMatch->alt);
    return potentialMatch->position + pal - 1 == toCollapse->position + tal - 1 && pal >= tal &&
           strcmp(potentialMatch->alt + (pal - tal), toCollapse->alt) == 0;
}

/* Frequency of a candidate: AlleleHelper.Map + CoverageCalculator.Compute + CalledAllele.Frequency (:196-207) */
static float cand_frequency(const OrcCandidate* c, const OrcState* src, int considerAnchors, int expectStitched)
{
    OrcCalled v;
    orc_called_from_candidate(&v, c);
    orc_coverage_compute(&v, src, considerAnchors, expectStitched);
    return frequency_f(v.allele_support, v.total_coverage);
}

typedef struct { const OrcCandidate* c; float freq; int idx; int known; } MatchRow;

/* the chromosome's known (prior) variants: Factory.cs:204 hands VariantCollapser the list of the priors file (Factory.cs:378-395); the
 * next orc_collapse / schedule run annotates with them (AnnotateKnown :178-190); n = 0 clears */
static const OrcCandidate* g_known = NULL;
static int32_t g_known_n = 0;
void orc_set_known_variants(const OrcCandidate* list, int32_t n) { g_known = list; g_known_n = n; }
/* PiscesApplicationOptions.ExcludeMNVsFromCollapsing (Factory.cs:204 -> VariantCollapser's excludeMNVs, VariantCollapser.cs:33): what the
 * schedule runs hand orc_collapse; 0 (the option's default) until set */
static int32_t g_exclude_mnvs = 0;
void orc_set_exclude_mnvs_from_collapsing(int32_t on) { g_exclude_mnvs = on != 0; }

/* IComparer<CandidateAllele>.Compare :214-244 */
static int match_cmp(const void* pa, const void* pb)
{
    const MatchRow* a = (const MatchRow*)pa;
    const MatchRow* b = (const MatchRow*)pb;
    if (a->known && !b->known) return -1;   /* return known one first :216-218 */
    if (!a->known && b->known) return 1;
    const int fa = cand_fully_anchored(a->c), fb = cand_fully_anchored(b->c);
    if (fa && !fb) return -1;
    if (!fa && fb) return 1;
    const int la = cand_length(a->c), lb = cand_length(b->c);
    if (la != lb) return la > lb ? -1 : 1;
    if (fabsf(a->freq - b->freq) > 0.0f) return a->freq > b->freq ? -1 : 1;
    if (a->c->position != b->c->position) return a->c->position < b->c->position ? -1 : 1;
    int r = strcmp(a->c->alt, b->c->alt);
    if (r != 0) return r;
    return a->idx - b->idx;   /* List.Sort is unstable there; input order keeps this deterministic */
}

typedef struct { int idx; int len, both, either, sup, openr, openl; const char* ref; const char* alt; } OrderRow;
/* OrderByDescending(Length).ThenByDescending(both open).ThenByDescending(either open).ThenBy(ref).ThenBy(alt).ThenBy(Support)
 * .ThenBy(OpenOnRight).ThenBy(OpenOnLeft) :41-46 (stable) */
static int order_cmp(const void* pa, const void* pb)
{
    const OrderRow* a = (const OrderRow*)pa;
    const OrderRow* b = (const OrderRow*)pb;
    if (a->len != b->len) return a->len > b->len ? -1 : 1;
    if (a->both != b->both) return a->both > b->both ? -1 : 1;
    if (a->either != b->either) return a->either > b->either ? -1 : 1;
    int r = strcmp(a->ref, b->ref);
    if (r) return r;
    r = strcmp(a->alt, b->alt);
    if (r) return r;
    if (a->sup != b->sup) return a->sup < b->sup ? -1 : 1;
    if (a->openr != b->openr) return a->openr < b->openr ? -1 : 1;
    if (a->openl != b->openl) return a->openl < b->openl ? -1 : 1;
    return a->idx - b->idx;
}

/* VariantCollapser.Collapse :31-79.  cands[0..n) is edited in place (collapsed entries removed, order kept); returns the new
 * count.  Candidates past max_cleared_position (>= 0) that could not be collapsed are moved to added_back (source.AddCandidates). */
int32_t orc_collapse(OrcCandidate* cands, int32_t n, const OrcState* src, float freq_threshold, float freq_ratio_threshold,
                     int32_t exclude_mnvs, int32_t consider_anchors, int32_t expect_stitched, int32_t max_cleared_position,
                     int32_t* n_collapsed, OrcCandidate* added_back, int32_t* n_added_back)
{
    uint8_t* removed = (uint8_t*)calloc((size_t)(n > 0 ? n : 1), 1);
    OrderRow* order = (OrderRow*)malloc(sizeof(OrderRow) * (size_t)(n > 0 ? n : 1));
    MatchRow* rows = (MatchRow*)malloc(sizeof(MatchRow) * (size_t)(n > 0 ? n : 1));
    int no = 0, collapsed = 0;
    /* AnnotateKnown :178-190: a target that equals a known variant is known and anchored on both sides, whatever its reads said */
    uint8_t* known = (uint8_t*)calloc((size_t)(n > 0 ? n : 1), 1);
    for (int i = 0; i < n; i++) {
        if (exclude_mnvs && cands[i].category == PISCES_CAT_MNV) continue;
        for (int k = 0; k < g_known_n; k++)
            if (candidate_equals(&cands[i], &g_known[k])) { known[i] = 1; cands[i].open_left = cands[i].open_right = 0; break; }
    }
    for (int i = 0; i < n; i++) {
        const OrcCandidate* c = &cands[i];
        if (exclude_mnvs && c->category == PISCES_CAT_MNV) continue;
        if (!(c->open_left || c->open_right)) continue;
        OrderRow r = {i, cand_length(c), c->open_left && c->open_right, c->open_left || c->open_right, cand_support(c), c->open_right,
                      c->open_left, c->ref, c->alt};
        order[no++] = r;
    }
    qsort(order, (size_t)no, sizeof(OrderRow), order_cmp);
    for (int k = 0; k < no; k++) {
        OrcCandidate* toCollapse = &cands[order[k].idx];
        int nm = 0;
        for (int j = 0; j < n; j++) {
            if (j == order[k].idx || removed[j]) continue;
            if (exclude_mnvs && cands[j].category == PISCES_CAT_MNV) continue;
            if (!can_collapse(toCollapse, &cands[j])) continue;
            rows[nm].c = &cands[j];
            rows[nm].idx = j;
            rows[nm].known = known[j];
            rows[nm].freq = cand_frequency(&cands[j], src, consider_anchors, expect_stitched);
            nm++;
        }
        if (nm == 0) continue;
        const float toFreq = cand_frequency(toCollapse, src, consider_anchors, expect_stitched);
        qsort(rows, (size_t)nm, sizeof(MatchRow), match_cmp);
        int pick = -1;
        for (int m = 0; m < nm && pick < 0; m++)
            if (candidate_equals(rows[m].c, toCollapse) && cand_fully_anchored(rows[m].c)) pick = m;
        for (int m = 0; m < nm && pick < 0; m++)
            if (rows[m].freq >= freq_threshold && rows[m].freq / toFreq > freq_ratio_threshold) pick = m;
        if (pick < 0) continue;
        OrcCandidate* match = &cands[rows[pick].idx];
        collapsed++;
        for (int d = 0; d < 3; d++) {   /* Collapse :81-90 */
            match->support_by_dir[d] += toCollapse->support_by_dir[d];
            match->well_anchored_by_dir[d] += toCollapse->well_anchored_by_dir[d];
        }
        match->open_left = match->open_left && toCollapse->open_left;
        match->open_right = match->open_right && toCollapse->open_right;
        removed[order[k].idx] = 1;
    }
    int nab = 0;
    if (max_cleared_position >= 0)
        for (int i = 0; i < n; i++)
            if (!removed[i] && cands[i].position > max_cleared_position && cands[i].category != PISCES_CAT_REFERENCE) {
                if (added_back) added_back[nab] = cands[i];
                nab++;
                removed[i] = 1;
            }
    int w = 0;
    for (int i = 0; i < n; i++)
        if (!removed[i]) { if (w != i) cands[w] = cands[i]; w++; }
    if (n_collapsed) *n_collapsed = collapsed;
    if (n_added_back) *n_added_back = nab;
    free(removed); free(order); free(rows); free(known);
    return w;
}

/* =====================================================================================
 * exe/Pisces/Logic/VariantCalling/MnvReallocator.cs
 * ===================================================================================== */
typedef struct PtrList { OrcCalled** p; int64_t n, cap; } PtrList;
static void pl_push(PtrList* l, OrcCalled* v)
{
    if (l->n == l->cap) { l->cap = l->cap ? l->cap * 2 : 16; l->p = (OrcCalled**)realloc(l->p, sizeof(OrcCalled*) * (size_t)l->cap); }
    l->p[l->n++] = v;
}
static void pl_remove(PtrList* l, const OrcCalled* v)   /* List<T>.Remove: first element that is this object */
{
    for (int64_t i = 0; i < l->n; i++)
        if (l->p[i] == v) { memmove(&l->p[i], &l->p[i + 1], sizeof(OrcCalled*) * (size_t)(l->n - i - 1)); l->n--; return; }
}
/* stable insertion sort (LINQ OrderBy is stable) */
static void pl_sort(PtrList* l, int (*cmp)(const OrcCalled*, const OrcCalled*))
{
    for (int64_t i = 1; i < l->n; i++) {
        OrcCalled* x = l->p[i];
        int64_t j = i;
        while (j > 0 && cmp(l->p[j - 1], x) > 0) { l->p[j] = l->p[j - 1]; j--; }
        l->p[j] = x;
    }
}
/* .OrderByDescending(alt.Length).ThenByDescending(AlleleSupport).ThenBy(alt).ThenBy(ref) :27 */
static int overlap_order(const OrcCalled* a, const OrcCalled* b)
{
    int la = (int)strlen(a->alt), lb = (int)strlen(b->alt);
    if (la != lb) return la > lb ? -1 : 1;
    if (a->allele_support != b->allele_support) return a->allele_support > b->allele_support ? -1 : 1;
    int r = strcmp(a->alt, b->alt);
    return r ? r : strcmp(a->ref, b->ref);
}
/* .OrderBy(position) then the same keys :17 */
static int failed_order(const OrcCalled* a, const OrcCalled* b)
{
    if (a->position != b->position) return a->position < b->position ? -1 : 1;
    return overlap_order(a, b);
}

/* CreateVariant :151-168 */
static OrcCalled* mnv_create_variant(int coordinate, int alleleSupport, const char* alt, int alt_n, const char* ref, int ref_n,
                                     const int32_t* supportByDirection)
{
    OrcCalled* v = (OrcCalled*)calloc(1, sizeof(OrcCalled));
    int same = (alt_n == ref_n);
    for (int i = 0; same && i < alt_n; i++)
        if (toupper((unsigned char)alt[i]) != toupper((unsigned char)ref[i])) same = 0;
    v->category = same ? PISCES_CAT_REFERENCE : (alt_n > 1 ? PISCES_CAT_MNV : PISCES_CAT_SNV);
    v->genotype = same ? PISCES_GT_HOM_REF : PISCES_GT_HET_ALT_REF;   /* CalledAllele() / CalledAllele(category) ctors */
    v->position = coordinate;
    v->allele_support = alleleSupport;
    memcpy(v->alt, alt, (size_t)alt_n); v->alt[alt_n] = 0;
    memcpy(v->ref, ref, (size_t)ref_n); v->ref[ref_n] = 0;
    if (supportByDirection) for (int d = 0; d < 3; d++) v->support_by_dir[d] = supportByDirection[d];
    return v;
}

/* BreakOffEdgeReferences :212-241 (one element) */
static OrcCalled* mnv_break_off_edge_references(OrcCalled* allele)
{
    if (allele->category != PISCES_CAT_MNV) return allele;
    int n = (int)strlen(allele->ref), leftAdjust = 0, rightAdjust = 0;
    for (int i = 0; i < n; i++) { if (allele->ref[i] != allele->alt[i]) break; leftAdjust++; }
    for (int i = 0; i < n; i++) { int k = n - 1 - i; if (allele->ref[k] != allele->alt[k]) break; rightAdjust++; }
    int len = (int)strlen(allele->alt) - (leftAdjust + rightAdjust), rlen = n - (leftAdjust + rightAdjust);
    OrcCalled* rest = mnv_create_variant(allele->position + leftAdjust, allele->allele_support, allele->alt + leftAdjust, len,
                                         allele->ref + leftAdjust, rlen, allele->support_by_dir);
    free(allele);
    return rest;
}

/* CreateAllelesFromRemainder :170-210 */
static void mnv_remainders(const OrcCalled* overlap, const OrcCalled* toReassign, PtrList* out)
{
    int overlapIndexInFailedMnv = overlap->position - toReassign->position;
    int overlapAlleleLength = (int)strlen(overlap->alt);
    int rightSideOverlap = overlapIndexInFailedMnv + overlapAlleleLength;
    int altLen = (int)strlen(toReassign->alt);
    PtrList tmp = {0};
    if (altLen - rightSideOverlap > 0 && rightSideOverlap <= toReassign->position + altLen) {
        OrcCalled* r = mnv_create_variant(toReassign->position + rightSideOverlap, toReassign->allele_support,
                                          toReassign->alt + rightSideOverlap, altLen - rightSideOverlap,
                                          toReassign->ref + rightSideOverlap, altLen - rightSideOverlap, toReassign->support_by_dir);
        if (r->category != PISCES_CAT_REFERENCE) pl_push(&tmp, r); else free(r);
    }
    if (overlapIndexInFailedMnv > 0) {
        OrcCalled* l = mnv_create_variant(toReassign->position, toReassign->allele_support, toReassign->alt, overlapIndexInFailedMnv,
                                          toReassign->ref, overlapIndexInFailedMnv, toReassign->support_by_dir);
        if (l->category != PISCES_CAT_REFERENCE) pl_push(&tmp, l); else free(l);
    }
    for (int64_t i = 0; i < tmp.n; i++) pl_push(out, mnv_break_off_edge_references(tmp.p[i]));
    free(tmp.p);
}

/* IsPotentialOverlap :250-261 (one chromosome) */
static int mnv_is_potential_overlap(const OrcCalled* callable, const OrcCalled* failed)
{
    int fl = (int)strlen(failed->alt), cl = (int)strlen(callable->alt);
    return callable->position >= failed->position && callable->position <= failed->position + fl && cl <= fl &&
           callable->position + cl <= failed->position + fl &&
           (callable->category == PISCES_CAT_MNV || callable->category == PISCES_CAT_SNV || callable->category == PISCES_CAT_REFERENCE);
}
/* OverlapMatches :243-248 */
static int mnv_overlap_matches(const OrcCalled* overlap, const OrcCalled* toReassign)
{
    int idx = overlap->position - toReassign->position, n = (int)strlen(overlap->alt);
    return strncmp(overlap->alt, toReassign->alt + idx, (size_t)n) == 0;
}

/* ProcessOverlap :97-133 */
static void mnv_process_overlap(int hasMax, int blockMaxPos, OrcCalled* overlap, OrcCalled* toReassign, PtrList* remainderAlleles,
                                PtrList* outsideThisBlock)
{
    overlap->allele_support += toReassign->allele_support;
    for (int d = 0; d < 3; d++) overlap->support_by_dir[d] += toReassign->support_by_dir[d];
    pl_remove(remainderAlleles, toReassign);
    PtrList remainders = {0};
    mnv_remainders(overlap, toReassign, &remainders);
    if (hasMax) {
        if (overlap->position > blockMaxPos) { pl_remove(remainderAlleles, overlap); pl_push(outsideThisBlock, overlap); }
        for (int64_t i = 0; i < remainders.n; i++) {
            if (remainders.p[i]->position <= blockMaxPos) pl_push(remainderAlleles, remainders.p[i]);
            else pl_push(outsideThisBlock, remainders.p[i]);
        }
    } else {
        for (int64_t i = 0; i < remainders.n; i++) pl_push(remainderAlleles, remainders.p[i]);
    }
    free(remainders.p);
}

/* ReallocateFailedMnvs :12-95.  failed / callable hold heap OrcCalled objects; new objects are appended to callable or
 * outsideThisBlock (which the caller maps back to candidates of the next block, AlleleCaller.cs:92-93).  The failed MNVs themselves
 * stay owned by `failed`. */
static void mnv_reallocate_failed(PtrList* failed, PtrList* callable, int hasMax, int blockMaxPos, PtrList* outsideThisBlock)
{
    PtrList ordered = {0};
    for (int64_t i = 0; i < failed->n; i++) pl_push(&ordered, failed->p[i]);
    pl_sort(&ordered, failed_order);
    for (int64_t fi = 0; fi < ordered.n; fi++) {
        PtrList remainderAlleles = {0};
        pl_push(&remainderAlleles, ordered.p[fi]);
        while (remainderAlleles.n > 0) {
            OrcCalled* alleleToReassign = remainderAlleles.p[0];
            PtrList overlaps = {0};
            for (int64_t i = 0; i < callable->n; i++)
                if (mnv_is_potential_overlap(callable->p[i], alleleToReassign)) pl_push(&overlaps, callable->p[i]);
            pl_sort(&overlaps, overlap_order);   /* longest sub-MNVs first, support as tie-breaker */
            OrcCalled* firstMatch = NULL;
            int anyLongMatch = 0;
            for (int64_t i = 0; i < overlaps.n; i++)
                if (mnv_overlap_matches(overlaps.p[i], alleleToReassign)) {
                    if (!firstMatch) firstMatch = overlaps.p[i];
                    if (strlen(overlaps.p[i]->alt) > 1) anyLongMatch = 1;
                }
            free(overlaps.p);
            int reallocated = 0;
            if (hasMax) {
                int altLen = (int)strlen(alleleToReassign->alt);
                int distanceIntoNextBlock = alleleToReassign->position + (altLen - 1) - blockMaxPos;
                if (distanceIntoNextBlock > 0 && !anyLongMatch) {
                    if (alleleToReassign->position <= blockMaxPos) {
                        /* peel off into the next block */
                        int originalAlleleLength = (int)strlen(alleleToReassign->ref);
                        OrcCalled* nextBlockVariant = mnv_create_variant(blockMaxPos + 1, 0,
                            alleleToReassign->alt + (originalAlleleLength - distanceIntoNextBlock), distanceIntoNextBlock,
                            alleleToReassign->ref + (originalAlleleLength - distanceIntoNextBlock), distanceIntoNextBlock, NULL);
                        nextBlockVariant = mnv_break_off_edge_references(nextBlockVariant);
                        mnv_process_overlap(hasMax, blockMaxPos, nextBlockVariant, alleleToReassign, &remainderAlleles, outsideThisBlock);
                    } else {
                        pl_remove(&remainderAlleles, alleleToReassign);
                        pl_push(outsideThisBlock, alleleToReassign);
                    }
                    reallocated = 1;
                }
            }
            if (!reallocated && firstMatch) {
                mnv_process_overlap(hasMax, blockMaxPos, firstMatch, alleleToReassign, &remainderAlleles, outsideThisBlock);
                reallocated = 1;
            }
            if (!reallocated) {
                /* BreakDownToSingleNucCalls :135-149 */
                int altLen = (int)strlen(alleleToReassign->alt);
                for (int i = 0; i < altLen; i++) {
                    OrcCalled* sn = mnv_create_variant(alleleToReassign->position + i, alleleToReassign->allele_support,
                                                       alleleToReassign->alt + i, 1, alleleToReassign->ref + i, 1,
                                                       alleleToReassign->support_by_dir);
                    if (sn->category == PISCES_CAT_REFERENCE) { free(sn); continue; }
                    if (hasMax && sn->position > blockMaxPos) pl_push(outsideThisBlock, sn);
                    else pl_push(callable, sn);
                }
                pl_remove(&remainderAlleles, alleleToReassign);
            }
        }
        free(remainderAlleles.p);
    }
    free(ordered.p);
}

/* test hook: MnvReallocator.ReallocateFailedMnvs over arrays.  callable[0..n_callable) is updated in place and grown (capacity
 * cap_callable); outside[0..) receives the leftovers for the next block.  max_position < 0 = null.  Returns the new n_callable,
 * *n_outside = number written. */
int64_t orc_reallocate_failed_mnvs(const OrcCalled* failed, int64_t n_failed, OrcCalled* callable, int64_t n_callable, int64_t cap_callable,
                                   int32_t max_position, OrcCalled* outside, int64_t cap_outside, int64_t* n_outside)
{
    PtrList f = {0}, c = {0}, o = {0};
    for (int64_t i = 0; i < n_failed; i++) { OrcCalled* v = (OrcCalled*)malloc(sizeof(OrcCalled)); *v = failed[i]; pl_push(&f, v); }
    for (int64_t i = 0; i < n_callable; i++) { OrcCalled* v = (OrcCalled*)malloc(sizeof(OrcCalled)); *v = callable[i]; pl_push(&c, v); }
    mnv_reallocate_failed(&f, &c, max_position >= 0, max_position, &o);
    int64_t nc = c.n, no = o.n;
    for (int64_t i = 0; i < c.n && i < cap_callable; i++) callable[i] = *c.p[i];
    for (int64_t i = 0; i < o.n && i < cap_outside; i++) outside[i] = *o.p[i];
    if (n_outside) *n_outside = no;
    /* objects may sit in several lists (a failed MNV that lies wholly in the next block is also in `outside`): free each once */
    for (int64_t i = 0; i < o.n; i++) {
        int dup = 0;
        for (int64_t k = 0; k < f.n; k++) if (f.p[k] == o.p[i]) dup = 1;
        for (int64_t k = 0; k < c.n; k++) if (c.p[k] == o.p[i]) dup = 1;
        for (int64_t k = 0; k < i; k++) if (o.p[k] == o.p[i]) dup = 1;
        if (!dup) free(o.p[i]);
    }
    for (int64_t i = 0; i < f.n; i++) free(f.p[i]);
    for (int64_t i = 0; i < c.n; i++) free(c.p[i]);
    free(f.p); free(c.p); free(o.p);
    return nc;
}

/* DiploidLocusProcessor.Process (exe/Pisces/Logic/VariantCalling/DiploidLocusProcessor.cs:13-52) over the alleles of one position: a
 * forced allele (ForcedReport filter) takes the genotype the other alleles imply, every allele the smallest genotype q-score of the others */
void orc_diploid_locus_process(OrcCalled* at, int32_t n)
{
    int anyForced = 0, anyOther = 0, isRef = 0, isNoCall = 0, minGq = 0;
    for (int k = 0; k < n; k++) {
        if ((at[k].filters >> PISCES_FILTER_FORCED_REPORT) & 1u) { anyForced = 1; continue; }
        const int g = at[k].genotype;
        if (at[k].category == PISCES_CAT_REFERENCE) isRef = 1;
        if (g == PISCES_GT_ALT12_LIKE_NOCALL || g == PISCES_GT_ALT_LIKE_NOCALL || g == PISCES_GT_HEMI_NOCALL || g == PISCES_GT_REF_LIKE_NOCALL)
            isNoCall = 1;   /* CalledAllele.IsNocall, CalledAllele.cs:54-62 */
        if (!anyOther || at[k].genotype_qscore < minGq) minGq = at[k].genotype_qscore;
        anyOther = 1;
    }
    if (!anyForced) return;
    if (!anyOther) isNoCall = 1;
    const int genotype = isNoCall ? PISCES_GT_ALT_LIKE_NOCALL : isRef ? PISCES_GT_HOM_REF : PISCES_GT_OTHERS;
    for (int k = 0; k < n; k++) {
        if ((at[k].filters >> PISCES_FILTER_FORCED_REPORT) & 1u) at[k].genotype = genotype;
        at[k].genotype_qscore = anyOther ? minGq : 0;
    }
}

/* AlleleCaller.CallForPositions :60-141 (collapser applied by the caller; forced alleles: orc_set_forced_alleles) over an explicit batch of
 * candidates (ICandidateBatch.GetCandidates): MNV candidates are processed first, the ones that are not callable are handed to
 * MnvReallocator, what it pushes past max_cleared_position goes back to the state as candidates (the next block's), reference
 * support taken by gapped MNVs is registered, then ProcessVariant + IsCallable per callable allele, and per position the
 * reference pruning, genotype, LowGQ filter and the (ref, alt) order of ComputeGenotypeAndFilterAllele :143-177. */
int64_t orc_call_candidates_max(OrcState* s, const OrcCandidate* list, int64_t n_list, const uint8_t* ref_bases, int64_t ref_len,
                                const PiscesHipConfig* cfg, int32_t max_cleared_position, PiscesCalledAllele* out, int64_t capacity,
                                OrcCalled* full_out, int64_t* total_num_called)
{
    g_rmxn_ref = ref_bases;
    g_rmxn_ref_len = ref_len;
    int64_t totalNumCalled = 0;
    PtrList callable = {0}, failedMnvs = {0}, outside = {0};
    for (int64_t i = 0; i < n_list; i++) {
        OrcCalled* v = (OrcCalled*)malloc(sizeof(OrcCalled));
        orc_called_from_candidate(v, &list[i]);
        if (v->category == PISCES_CAT_MNV) {
            orc_process_variant(v, s, cfg);
            if (is_callable(v, cfg, &totalNumCalled)) pl_push(&callable, v);
            else pl_push(&failedMnvs, v);
        } else {
            pl_push(&callable, v);
        }
    }
    if (failedMnvs.n > 0) {
        {   /* (PiscesApplicationOptions.UseMNVReallocation is declared but read nowhere: the caller always reallocates) */
            mnv_reallocate_failed(&failedMnvs, &callable, max_cleared_position >= 0, max_cleared_position, &outside);
            for (int64_t i = 0; i < outside.n; i++) {   /* source.AddCandidates(leftovers.Select(AlleleHelper.Map)) */
                OrcCandidate c;
                memset(&c, 0, sizeof(c));
                c.position = outside.p[i]->position;
                c.category = outside.p[i]->category;
                strcpy(c.ref, outside.p[i]->ref);
                strcpy(c.alt, outside.p[i]->alt);
                for (int d = 0; d < 3; d++) c.support_by_dir[d] = outside.p[i]->support_by_dir[d];
                if (c.category != PISCES_CAT_REFERENCE) (void)orc_add_candidate(s, &c);   /* outside the window: dropped */
            }
        }
    }
    /* GetRefSupportFromGappedMnvs :180-203 */
    for (int64_t i = 0; i < callable.n; i++) {
        const OrcCalled* a = callable.p[i];
        if (a->category != PISCES_CAT_MNV) continue;
        for (int k = 0; a->ref[k]; k++)
            if (a->ref[k] == a->alt[k]) orc_add_gapped_mnv_ref(s, a->position + k, a->allele_support);
    }

    /* a failed MNV that is a forced allele is reported all the same: back among the callable alleles (:98-107) */
    int64_t n_spiked = 0;
    for (int64_t i = 0; i < failedMnvs.n; i++)
        if (is_forced_allele(s, failedMnvs.p[i])) { pl_push(&callable, failedMnvs.p[i]); n_spiked++; }

    int64_t n = 0, cap = callable.n + 1;
    OrcCalled* called = (OrcCalled*)malloc(sizeof(OrcCalled) * (size_t)cap);
    for (int64_t i = 0; i < callable.n; i++) {
        OrcCalled* v = callable.p[i];
        orc_process_variant(v, s, cfg);
        /* :109-131.  (ShouldReport: the window is the interval; forced alleles are inside the intervals by Factory.SelectForcedAllele.)
         * IsCallable runs once in each condition, and counts a callable forced allele twice in TotalNumCalled. */
        const int forced = s->n_forced > 0 && is_forced_allele(s, v);
        if (forced && !(is_callable(v, cfg, &totalNumCalled) && inside_intervals(s, v->position))) v->filters |= 1u << PISCES_FILTER_FORCED_REPORT;   /* IsForcedToReport */
        if ((is_callable(v, cfg, &totalNumCalled) && inside_intervals(s, v->position)) || forced) called[n++] = *v;   /* IsCallable && ShouldReport */
    }
    callable.n -= n_spiked;   /* (owned by failedMnvs) */
    {   /* free every object once */
        for (int64_t i = 0; i < outside.n; i++) {
            int dup = 0;
            for (int64_t k = 0; k < failedMnvs.n && !dup; k++) dup = failedMnvs.p[k] == outside.p[i];
            for (int64_t k = 0; k < i && !dup; k++) dup = outside.p[k] == outside.p[i];
            if (!dup) free(outside.p[i]);
        }
        for (int64_t i = 0; i < failedMnvs.n; i++) free(failedMnvs.p[i]);
        for (int64_t i = 0; i < callable.n; i++) free(callable.p[i]);
        free(outside.p); free(failedMnvs.p); free(callable.p);
    }

    /* SortedList by position; per position sort by (ref, alt) :172-176. Stable enough: keys are unique
     * unless open-ended twins survive (only with track_open_ended). */
    qsort(called, (size_t)n, sizeof(OrcCalled), called_cmp);

    /* ComputeGenotypeAndFilterAllele :143-177 per position, then ILocusProcessor.Process */
#define FORCED_TO_REPORT(a) (((a).filters >> PISCES_FILTER_FORCED_REPORT) & 1u)
    const int per_locus_genotyper = cfg->ploidy == PISCES_PLOIDY_DIPLOID || cfg->ploidy == PISCES_PLOIDY_HAPLOID;
    int64_t w = 0;
    for (int64_t i = 0; i < n;) {
        int64_t j = i;
        int anyNonRef = 0;   /* a variant that is not there only because it was forced */
        while (j < n && called[j].position == called[i].position) {
            if (called[j].category != PISCES_CAT_REFERENCE && !FORCED_TO_REPORT(called[j])) anyNonRef = 1;
            j++;
        }
        OrcCalled at[64];   /* allelesAtPosition, Reference rows pruned */
        int m = 0;
        for (int64_t k = i; k < j && m < 64; k++) {
            if (anyNonRef && called[k].category == PISCES_CAT_REFERENCE) continue;
            at[m++] = called[k];
        }
        uint8_t prune[64] = {0};
        int32_t phase[64] = {0};
        if (per_locus_genotyper) {
            /* the genotyper sees the alleles that were not forced in, and names the ones beyond the ploidy */
            OrcCalled sub[64];
            int idx[64], ms = 0;
            uint8_t sub_prune[64];
            int32_t sub_phase[64];
            for (int q = 0; q < m; q++)
                if (!FORCED_TO_REPORT(at[q])) { idx[ms] = q; sub[ms++] = at[q]; }
            if (cfg->ploidy == PISCES_PLOIDY_HAPLOID) {
                orc_haploid_set_genotypes(sub, ms, cfg->diploid_snv_params[0], cfg->diploid_snv_params[1], cfg->min_coverage, cfg->min_genotype_qscore,
                                          cfg->max_genotype_qscore, sub_prune);
                for (int q = 0; q < ms; q++) sub_phase[q] = 0;   /* HaploidGenotyper leaves PhaseSetIndex alone */
            } else {
                orc_diploid_set_genotypes(sub, ms, cfg->diploid_snv_params, cfg->diploid_indel_params, cfg->min_coverage, cfg->min_genotype_qscore,
                                          cfg->max_genotype_qscore, sub_phase, sub_prune);
            }
            for (int q = 0; q < ms; q++) {
                at[idx[q]] = sub[q];
                phase[idx[q]] = sub_phase[q];
                /* a forced allele stays even when the genotyper would drop it :155-163 */
                prune[idx[q]] = sub_prune[q] && !(s->n_forced > 0 && is_forced_allele(s, &sub[q]));
            }
        } else {
            for (int q = 0; q < m; q++) {
                OrcCalled* a = &at[q];
                if (FORCED_TO_REPORT(*a)) continue;
                a->genotype = orc_somatic_genotype(a->category, a->total_coverage, a->allele_support, a->reference_support,
                                                   cfg->genotype_min_freq_filter, cfg->min_coverage);
                a->genotype_qscore = orc_somatic_gq(a->genotype, a->variant_qscore, a->total_coverage, a->allele_support,
                                                    cfg->target_lod_frequency, cfg->min_genotype_qscore, cfg->max_genotype_qscore);
            }
        }
        const int64_t w0 = w;
        for (int q = 0; q < m; q++) {
            if (prune[q]) continue;
            if (cfg->low_gq_filter >= 0 && (float)at[q].genotype_qscore < (float)cfg->low_gq_filter)
                at[q].filters |= 1u << PISCES_FILTER_LOW_GENOTYPE_QUALITY;
            at[q].filters |= (uint32_t)phase[q] << 14;   /* PhaseSetIndex rides in filter_bits 14..15 */
            called[w++] = at[q];
        }
        if (cfg->ploidy == PISCES_PLOIDY_DIPLOID) orc_diploid_locus_process(called + w0, (int32_t)(w - w0));   /* Factory.cs:145-147: the diploid model only */
        i = j;
    }
#undef FORCED_TO_REPORT
    n = w;
    if (total_num_called) *total_num_called = totalNumCalled;
    if (n > capacity) { free(called); return -n; }
    for (int64_t i = 0; i < n; i++) {
        to_record(&out[i], &called[i]);
        if (full_out) full_out[i] = called[i];
    }
    free(called);
    return n;
}

int64_t orc_call_candidates(OrcState* s, const OrcCandidate* list, int64_t n_list, const uint8_t* ref_bases, int64_t ref_len,
                            const PiscesHipConfig* cfg, PiscesCalledAllele* out, int64_t capacity, OrcCalled* full_out,
                            int64_t* total_num_called)
{
    /* one dense window = one block: MaxClearedPosition is its last position (RegionStateManager.cs:318-320) */
    return orc_call_candidates_max(s, list, n_list, ref_bases, ref_len, cfg, s->start_position + s->n_loci - 1, out, capacity, full_out,
                                   total_num_called);
}

/* CandidateBatch of GetCandidatesToProcess (RegionStateManager.cs:283-334) for cleared blocks [first_position, last_position]: their
 * candidates by position, each position in arrival order (RegionState.GetAllCandidates :388-391; they stay in the state until
 * DoneProcessing), and with up_to_position >= 0, open-ended tracking on and an allele of those blocks reaching past last_position, the
 * collapsable SNV / MNV candidates of the blocks that start in (last_position, up_to_position], which leave the state
 * (AddCollapsableFromOtherBlocks :441-457, RegionState.ExtractCollapsable :470-490).  Returns the count. */
int32_t orc_batch_candidates(OrcState* s, int32_t first_position, int32_t last_position, int32_t up_to_position, OrcCandidate* out,
                             int32_t capacity, int32_t* from_other_blocks)
{
    int32_t n = 0;
    if (from_other_blocks) *from_other_blocks = 0;
    for (int li = 0; li < s->n_loci; li++) {
        const int position = s->start_position + li;
        if (position < first_position || position > last_position) continue;
        for (int i = s->cand_head[li]; i >= 0; i = s->cands[i].next) { if (n < capacity) out[n] = s->cands[i]; n++; }
    }
    if (!(up_to_position >= 0 && s->track_open_ended && s->block_size > 0)) return n;
    int maxEndpoint = 0;   /* blocks.Max(b => b.MaxAlleleEndpoint) :321 */
    for (int k = (first_position - 1) / s->block_size + 1; k <= (last_position - 1) / s->block_size + 1; k++)
        if (k >= s->first_block_key && k < s->first_block_key + s->n_blocks && s->max_allele_endpoint[k - s->first_block_key] > maxEndpoint)
            maxEndpoint = s->max_allele_endpoint[k - s->first_block_key];
    if (maxEndpoint <= last_position) return n;
    const int lastOfThose = ((up_to_position - 1) / s->block_size + 1) * s->block_size;   /* end of the block that holds upTo */
    for (int li = 0; li < s->n_loci; li++) {
        const int position = s->start_position + li;
        if (position <= last_position || position > lastOfThose) continue;
        int prev = -1;
        for (int i = s->cand_head[li]; i >= 0;) {
            OrcCandidate* c = &s->cands[i];
            const int next = c->next;
            if (c->position + (int)strlen(c->alt) - 1 <= up_to_position && !c->open_right &&
                (c->category == PISCES_CAT_MNV || c->category == PISCES_CAT_SNV)) {
                if (n < capacity) out[n] = *c;
                n++;
                if (from_other_blocks) *from_other_blocks = 1;
                if (prev >= 0) s->cands[prev].next = next; else s->cand_head[li] = next;   /* lookup.Remove(collapsable) */
                if (s->cand_tail[li] == i) s->cand_tail[li] = prev;
            } else {
                prev = i;
            }
            i = next;
        }
    }
    return n;
}

/* Which blocks GetCandidatesToProcess(upToPosition) clears (RegionStateManager.cs:283-334; up_to_position < 0 = null, the final batch).
 * Returns -1: no batch (upTo is still in the block of the previous call); 0: a batch without cleared blocks; 1: blocks
 * [*first_position, *last_position].  Every block of the window from the first one not yet done counts as existing. */
int32_t orc_next_batch(OrcState* s, int32_t up_to_position, int32_t* first_position, int32_t* last_position)
{
    const int bs = s->block_size;
    const int final_batch = up_to_position < 0;
    if (s->next_block_key < s->first_block_key) s->next_block_key = s->first_block_key;
    if (!final_batch) {
        const int key = (up_to_position - 1) / bs + 1;   /* GetBlockKey :404-407 */
        if (s->have_last_up_to && key == s->last_up_to_block_key) return -1;
        s->last_up_to_block_key = key;
        s->have_last_up_to = 1;
    } else {
        s->have_last_up_to = 0;
    }
    int last_key = s->next_block_key - 1;
    for (int k = s->next_block_key; k < s->first_block_key + s->n_blocks; k++) {
        if (!final_batch && (int64_t)k * bs > up_to_position) break;                                          /* keys.Where(k * size <= upTo) */
        if (!final_batch && s->max_allele_endpoint[k - s->first_block_key] > up_to_position) break;          /* a held block, and all after it */
        last_key = k;
    }
    if (last_key < s->next_block_key) return 0;
    *first_position = (s->next_block_key - 1) * bs + 1;
    *last_position = last_key * bs;
    return 1;
}

/* DoneProcessing :336-360 for the blocks up to last_position */
void orc_done_processing(OrcState* s, int32_t last_position)
{
    for (int li = 0; li < s->n_loci; li++)
        if (s->start_position + li <= last_position) s->cand_head[li] = s->cand_tail[li] = -1;
    const int key = (last_position - 1) / s->block_size + 1;
    if (key + 1 > s->next_block_key) s->next_block_key = key + 1;
}

/* The same over RegionState.GetAllCandidates :383-453: every candidate of the state plus (gVCF) a Reference candidate per
 * position with support = the reference base's counts by direction. */
int64_t orc_call_range(OrcState* s, const uint8_t* ref_bases, int64_t ref_len, const PiscesHipConfig* cfg, int32_t first_position,
                       int32_t last_position, PiscesCalledAllele* out, int64_t capacity, OrcCalled* full_out, int64_t* total_num_called)
{
    return orc_call_range_up_to(s, ref_bases, ref_len, cfg, first_position, last_position, -1, out, capacity, full_out, total_num_called);
}

/* up_to_position >= 0: a batch made while reads are still arriving (GetCandidatesToProcess(upToPosition)); when an allele of the cleared
 * blocks reaches past last_position the collapsable candidates of the following blocks join it (AddCollapsableFromOtherBlocks). */
int64_t orc_call_range_up_to(OrcState* s, const uint8_t* ref_bases, int64_t ref_len, const PiscesHipConfig* cfg, int32_t first_position,
                             int32_t last_position, int32_t up_to_position, PiscesCalledAllele* out, int64_t capacity, OrcCalled* full_out,
                             int64_t* total_num_called)
{
    /* one batch of the block schedule: the candidates and Reference candidates of [first_position, last_position] (whole blocks),
     * MaxClearedPosition = last_position (RegionStateManager.cs:283-334).  What the batch pushes past it (MNV leftovers) goes back to
     * the state and is found by the next range; processed candidates are removed (DoneProcessing). */
    orc_add_forced_as_candidates(s, up_to_position);   /* SmallVariantCaller.cs:101-108: before Call(upTo) */
    int64_t cap = (int64_t)s->n_cands + 64;
    OrcCandidate* list = (OrcCandidate*)malloc(sizeof(OrcCandidate) * (size_t)cap);
    int32_t from_other_blocks = 0;
    int64_t n = orc_batch_candidates(s, first_position, last_position, up_to_position, list, (int32_t)cap, &from_other_blocks);
    for (int li = 0; li < s->n_loci; li++) {   /* DoneProcessing :336-360 (what the batch hands back goes to later blocks) */
        const int position = s->start_position + li;
        if (position >= first_position && position <= last_position) s->cand_head[li] = s->cand_tail[li] = -1;
    }
    if (cfg->collapse) {   /* AlleleCaller.Call :50-58: candidates = _collapser.Collapse(batch.GetCandidates(), source, MaxClearedPosition) */
        OrcCandidate* back = from_other_blocks ? (OrcCandidate*)malloc(sizeof(OrcCandidate) * (size_t)(n > 0 ? n : 1)) : NULL;
        int32_t n_back = 0;
        n = orc_collapse(list, (int32_t)n, s, cfg->collapse_freq_threshold, cfg->collapse_freq_ratio_threshold, g_exclude_mnvs, 1,
                         cfg->expect_stitched_reads, from_other_blocks ? last_position : -1, NULL, back, &n_back);
        for (int i = 0; i < n_back; i++) orc_add_candidate(s, &back[i]);   /* source.AddCandidates(notClearedVariants) :67-75 */
        free(back);
    }
    /* RegionState.GetAllCandidates :393-450: Reference candidates over the block (gVCF; the window stands for the intervals), or — not a
     * gVCF, forced alleles given — at the positions of the forced alleles (CreateIntervalsFromAllels :455-468), there with or without
     * coverage (IntervalsInUse != null) */
    const int refs_at_forced_only = !cfg->include_reference_calls && s->n_forced > 0;
    if ((cfg->include_reference_calls || refs_at_forced_only) && ref_bases) {
        for (int li = 0; li < s->n_loci; li++) {
            int position = s->start_position + li;
            if (position < first_position || position > last_position) continue;
            if (position > ref_len) break;
            if (!inside_intervals(s, position)) continue;
            if (refs_at_forced_only) {
                int here = 0;
                for (int f = 0; f < s->n_forced && !here; f++) here = s->forced[f].position == position;
                if (!here) continue;
            }
            uint8_t refBase = ref_bases[position - 1];
            int refBaseIndex = allele_type_of(refBase);
            OrcCandidate rc;
            memset(&rc, 0, sizeof(rc));
            rc.position = position;
            rc.category = PISCES_CAT_REFERENCE;
            rc.ref[0] = rc.alt[0] = (char)refBase;
            int totalSupport = 0;
            for (int at = 0; at < 6; at++)
                for (int d = 0; d < 3; d++) {
                    int count = 0;
                    for (int an = 0; an < s->n_anchor_idx; an++) count += s->counts[cidx(s, position, at, d, an)];
                    if (at == refBaseIndex) rc.support_by_dir[d] = count;
                    totalSupport += count;
                }
            if (cfg->emit_zero_coverage_refs || refs_at_forced_only || totalSupport > 0) {
                if (n == cap) { cap *= 2; list = (OrcCandidate*)realloc(list, sizeof(OrcCandidate) * (size_t)cap); }
                list[n++] = rc;
            }
        }
    }
    int64_t r = orc_call_candidates_max(s, list, n, ref_bases, ref_len, cfg, last_position, out, capacity, full_out, total_num_called);
    free(list);
    return r;
}

int64_t orc_call_all(OrcState* s, const uint8_t* ref_bases, int64_t ref_len, const PiscesHipConfig* cfg,
                     PiscesCalledAllele* out, int64_t capacity, OrcCalled* full_out, int64_t* total_num_called)
{
    return orc_call_range(s, ref_bases, ref_len, cfg, s->start_position, s->start_position + s->n_loci - 1, out, capacity, full_out,
                          total_num_called);
}

/* =====================================================================================
 * Whole path: SmallVariantCaller.Execute (exe/Pisces/Logic/SmallVariantCaller.cs:79-116).
 * One dense window instead of 1000-locus blocks: without collapser / MNV spill-over the
 * block schedule (RegionStateManager.cs:283-334) only changes WHEN alleles are emitted.
 * ===================================================================================== */
static int64_t count_candidate_loci(const PiscesCalledAllele* out, int64_t n)
{
    int64_t loci = 0;
    for (int64_t i = 0; i < n; i++)
        if (i == 0 || out[i].position != out[i - 1].position) loci++;
    return loci;
}

int64_t orc_run_reads_full(const PiscesReadBatch* b, const uint8_t* ref_bases, int64_t ref_len, int32_t region_start,
                           int32_t region_loci, const PiscesHipConfig* cfg, PiscesCalledAllele* out, int64_t capacity,
                           int64_t* n_candidate_loci, OrcCalled* full_out, int64_t* total_num_called);

/* One read of a batch as the oracle's Read.  A batch that tracks directions inside deletions (deletion_directions) is turned back
 * into the expanded direction map the reference reads them from: sequenced bases carry their own direction, a deletion its first /
 * last pair (first direction up to the last base, which takes the second). */
static void read_from_batch(const PiscesReadBatch* b, int i, OrcRead* r, uint8_t** expanded, int64_t* expanded_cap)
{
    r->position = b->position[i];
    r->n_cigar = b->cigar_offset[i + 1] - b->cigar_offset[i];
    r->cigar_op = b->cigar_op + b->cigar_offset[i];
    r->cigar_len = b->cigar_len + b->cigar_offset[i];
    r->read_len = b->seq_offset[i + 1] - b->seq_offset[i];
    r->bases = b->bases + b->seq_offset[i];
    r->quals = b->quals + b->seq_offset[i];
    r->dirs = b->directions ? b->directions + b->seq_offset[i] : NULL;
    r->is_reverse = b->flags[i] & 1;
    r->posmap_override = NULL;
    r->expanded_dirs = NULL;
    r->n_expanded = 0;
    if (!b->deletion_directions) return;
    const uint8_t* dd = b->deletion_directions + 2 * (size_t)b->cigar_offset[i];
    int tracked = 0;
    int64_t total = 0;
    for (int c = 0; c < r->n_cigar; c++) {
        total += r->cigar_len[c];
        if (r->cigar_op[c] == 'D' && dd[2 * c] != PISCES_DIR_UNTRACKED) tracked = 1;
    }
    if (!tracked) return;
    if (total > *expanded_cap) {
        free(*expanded);
        *expanded_cap = total * 2;
        *expanded = (uint8_t*)malloc((size_t)*expanded_cap);
    }
    int64_t e = 0;
    int seq = 0;
    for (int c = 0; c < r->n_cigar; c++)
        for (uint32_t k = 0; k < r->cigar_len[c]; k++, e++) {
            if (op_is_read_span(r->cigar_op[c])) {
                (*expanded)[e] = seq < r->read_len ? (uint8_t)read_dir(r, seq) : (uint8_t)PISCES_DIR_FORWARD;
                seq++;
            } else if (r->cigar_op[c] == 'D') {
                (*expanded)[e] = k + 1 == r->cigar_len[c] ? dd[2 * c + 1] : dd[2 * c];
            } else {
                (*expanded)[e] = PISCES_DIR_UNTRACKED;
            }
        }
    r->expanded_dirs = *expanded;
    r->n_expanded = (int32_t)total;
}

int64_t orc_run_reads(const PiscesReadBatch* b, const uint8_t* ref_bases, int64_t ref_len, int32_t region_start,
                      int32_t region_loci, const PiscesHipConfig* cfg, PiscesCalledAllele* out, int64_t capacity,
                      int64_t* n_candidate_loci)
{
    return orc_run_reads_full(b, ref_bases, ref_len, region_start, region_loci, cfg, out, capacity, n_candidate_loci, NULL, NULL);
}

int64_t orc_run_reads_full(const PiscesReadBatch* b, const uint8_t* ref_bases, int64_t ref_len, int32_t region_start,
                           int32_t region_loci, const PiscesHipConfig* cfg, PiscesCalledAllele* out, int64_t capacity,
                           int64_t* n_candidate_loci, OrcCalled* full_out, int64_t* total_num_called)
{
    /* the state manager tracks open-ended candidates apart when the collapser is on (Factory.cs:209-227) */
    OrcState* s = orc_state_create(region_start, region_loci, cfg->min_base_call_quality, PISCES_ANCHOR_SIZE, cfg->collapse ? 1 : 0);
    OrcCandidate cands[256];
    uint8_t* expanded = NULL;
    int64_t expanded_cap = 0;
    for (int i = 0; i < b->n_reads; i++) {
        OrcRead r;
        read_from_batch(b, i, &r, &expanded, &expanded_cap);
        /* FindCandidates -> AddCandidates -> AddAlleleCounts (SmallVariantCaller.cs:92-98) */
        int nc = orc_find_candidates(&r, ref_bases, ref_len, cfg->min_base_call_quality, cfg->max_mnv_length, cfg->max_gap_between_mnv,
                                     cfg->call_mnvs, PISCES_ANCHOR_SIZE, cands, 256);
        for (int k = 0; k < nc; k++)
            if (cands[k].position >= region_start && cands[k].position < region_start + region_loci)
                orc_add_candidate(s, &cands[k]);
        int rc = orc_add_allele_counts(s, &r);
        if (rc) { free(expanded); orc_state_destroy(s); return rc; }
    }
    free(expanded);
    int64_t n = orc_call_all(s, ref_bases, ref_len, cfg, out, capacity, full_out, total_num_called);
    orc_state_destroy(s);
    if (n >= 0 && n_candidate_loci) *n_candidate_loci = count_candidate_loci(out, n);
    return n;
}

/* The same with the block schedule: every block of the cfg->block_size grid is its own batch, in order, each with MaxClearedPosition = its
 * last position, as SmallVariantCaller drives the caller when reads arrive in position order and every block is flushed once the reads
 * have moved past it.  Differs from the single window only where alleles interact across a block edge (MNV reallocation peels). */
int64_t orc_run_reads_blocks(const PiscesReadBatch* b, const uint8_t* ref_bases, int64_t ref_len, int32_t region_start, int32_t region_loci,
                             const PiscesHipConfig* cfg, PiscesCalledAllele* out, int64_t capacity, OrcCalled* full_out, int64_t* total_num_called)
{
    OrcState* s = orc_state_create(region_start, region_loci, cfg->min_base_call_quality, PISCES_ANCHOR_SIZE, cfg->collapse ? 1 : 0);
    OrcCandidate cands[256];
    uint8_t* expanded = NULL;
    int64_t expanded_cap = 0;
    for (int i = 0; i < b->n_reads; i++) {
        OrcRead r;
        read_from_batch(b, i, &r, &expanded, &expanded_cap);
        int nc = orc_find_candidates(&r, ref_bases, ref_len, cfg->min_base_call_quality, cfg->max_mnv_length, cfg->max_gap_between_mnv,
                                     cfg->call_mnvs, PISCES_ANCHOR_SIZE, cands, 256);
        for (int k = 0; k < nc; k++)
            if (cands[k].position >= region_start && cands[k].position < region_start + region_loci) orc_add_candidate(s, &cands[k]);
        int rc = orc_add_allele_counts(s, &r);
        if (rc) { free(expanded); orc_state_destroy(s); return rc; }
    }
    free(expanded);
    const int bs = cfg->block_size;
    int64_t n = 0, total = 0;
    const int region_end = region_start + region_loci - 1;
    for (int first = ((region_start - 1) / bs) * bs + 1; first <= region_end; first += bs) {
        int64_t t = 0;
        int64_t k = orc_call_range(s, ref_bases, ref_len, cfg, first, first + bs - 1, out + n, capacity - n, full_out ? full_out + n : NULL, &t);
        if (k < 0) { orc_state_destroy(s); return k; }
        n += k;
        total += t;
    }
    if (total_num_called) *total_num_called = total;
    orc_state_destroy(s);
    return n;
}

/* SmallVariantCaller's loop with the reads given first and then a list of upToPosition values, the last batch being the final one
 * (GetCandidatesToProcess(null)): RegionStateManager.GetCandidatesToProcess :283-334 decides which blocks each batch clears.  Every
 * block of the window exists (the window is dense); a batch is skipped while upTo stays in the block of the previous call. */
/* the candidates the next orc_run_reads_schedule* adds to its state before the schedule runs (a host's own AddCandidates); n = 0 clears */
static const OrcCandidate* g_host_candidates = NULL;
static int32_t g_host_candidates_n = 0;
void orc_schedule_host_candidates(const OrcCandidate* list, int32_t n) { g_host_candidates = list; g_host_candidates_n = n; }

int64_t orc_run_reads_schedule_intervals(const PiscesReadBatch* b, const uint8_t* ref_bases, int64_t ref_len, int32_t region_start, int32_t region_loci,
                                         const PiscesHipConfig* cfg, const int32_t* up_to_positions, int32_t n_up_to, const OrcCandidate* forced,
                                         int32_t n_forced, const int32_t* iv_starts, const int32_t* iv_ends, int32_t n_intervals,
                                         PiscesCalledAllele* out, int64_t capacity, OrcCalled* full_out, int64_t* total_num_called);
int64_t orc_run_reads_schedule(const PiscesReadBatch* b, const uint8_t* ref_bases, int64_t ref_len, int32_t region_start, int32_t region_loci,
                               const PiscesHipConfig* cfg, const int32_t* up_to_positions, int32_t n_up_to, const OrcCandidate* forced,
                               int32_t n_forced, PiscesCalledAllele* out, int64_t capacity, OrcCalled* full_out, int64_t* total_num_called)
{
    return orc_run_reads_schedule_intervals(b, ref_bases, ref_len, region_start, region_loci, cfg, up_to_positions, n_up_to, forced, n_forced, NULL, NULL, 0,
                                            out, capacity, full_out, total_num_called);
}

/* ... with a ChrIntervalSet (n_intervals inclusive ranges, sorted and disjoint; 0: none) */
int64_t orc_run_reads_schedule_intervals(const PiscesReadBatch* b, const uint8_t* ref_bases, int64_t ref_len, int32_t region_start, int32_t region_loci,
                                         const PiscesHipConfig* cfg, const int32_t* up_to_positions, int32_t n_up_to, const OrcCandidate* forced,
                                         int32_t n_forced, const int32_t* iv_starts, const int32_t* iv_ends, int32_t n_intervals,
                                         PiscesCalledAllele* out, int64_t capacity, OrcCalled* full_out, int64_t* total_num_called)
{
    OrcState* s = orc_state_create(region_start, region_loci, cfg->min_base_call_quality, PISCES_ANCHOR_SIZE, cfg->collapse ? 1 : 0);
    orc_track_blocks(s, cfg->block_size);
    orc_set_intervals(s, iv_starts, iv_ends, n_intervals);
    /* candidates the host hands in beside the reads' (IStateManager.AddCandidates, SmallVariantCaller.cs:92-96): orc_schedule_host_candidates */
    for (int i = 0; i < g_host_candidates_n; i++)
        if (g_host_candidates[i].position >= region_start && g_host_candidates[i].position < region_start + region_loci) orc_add_candidate(s, &g_host_candidates[i]);
    if (n_forced > 0) orc_set_forced_alleles(s, forced, n_forced);
    OrcCandidate cands[256];
    uint8_t* expanded = NULL;
    int64_t expanded_cap = 0;
    for (int i = 0; i < b->n_reads; i++) {
        OrcRead r;
        read_from_batch(b, i, &r, &expanded, &expanded_cap);
        int nc = orc_find_candidates(&r, ref_bases, ref_len, cfg->min_base_call_quality, cfg->max_mnv_length, cfg->max_gap_between_mnv,
                                     cfg->call_mnvs, PISCES_ANCHOR_SIZE, cands, 256);
        for (int k = 0; k < nc; k++)
            if (cands[k].position >= region_start && cands[k].position < region_start + region_loci) orc_add_candidate(s, &cands[k]);
        int rc = orc_add_allele_counts(s, &r);
        if (rc) { free(expanded); orc_state_destroy(s); return rc; }
    }
    free(expanded);
    int64_t n = 0, total = 0;
    for (int u = 0; u <= n_up_to; u++) {
        const int upTo = u == n_up_to ? -1 : up_to_positions[u];
        int32_t first = 0, last = 0;
        orc_add_forced_as_candidates(s, upTo);
        if (orc_next_batch(s, upTo, &first, &last) != 1) continue;
        int64_t t = 0;
        int64_t k = orc_call_range_up_to(s, ref_bases, ref_len, cfg, first, last, upTo, out + n, capacity - n, full_out ? full_out + n : NULL, &t);
        if (k < 0) { orc_state_destroy(s); return k; }
        n += k;
        total += t;
        orc_done_processing(s, last);
    }
    if (total_num_called) *total_num_called = total;
    orc_state_destroy(s);
    return n;
}

int64_t orc_run_observations(const int32_t* positions, const uint32_t* tuples, int64_t n_obs, const uint8_t* ref_bases,
                             int64_t ref_len, int32_t region_start, int32_t region_loci, const PiscesHipConfig* cfg,
                             PiscesCalledAllele* out, int64_t capacity, int64_t* n_candidate_loci)
{
    OrcState* s = orc_state_create(region_start, region_loci, cfg->min_base_call_quality, PISCES_ANCHOR_SIZE, 0);
    static const char kBase[6] = {'A', 'G', 'C', 'T', 'N', 'D'};
    for (int64_t i = 0; i < n_obs; i++) {
        uint32_t t = tuples[i];
        if (t == PISCES_TUPLE_PAD) continue;
        int pos = positions[i];
        int allele = (int)PISCES_TUPLE_ALLELE(t), dir = (int)PISCES_TUPLE_DIR(t), anchor = (int)PISCES_TUPLE_ANCHOR(t);
        int qual = (int)PISCES_TUPLE_QUAL(t);
        int raw = allele;
        /* RegionStateManager.cs:179-181 */
        if (allele < PISCES_ALLELE_N && qual < cfg->min_base_call_quality) allele = PISCES_ALLELE_N;
        add_allele_count(s, pos, allele, dir, anchor);
        /* the SNV candidate this observation implies (CandidateVariantFinder.cs:97-160, callMNVs off) */
        if (in_region(s, pos) && allele < PISCES_ALLELE_N && pos <= ref_len) {
            uint8_t rb = ref_bases[pos - 1];
            int rt = allele_type_of(rb);
            if (rt != PISCES_ALLELE_N && rt != allele) {
                OrcCandidate c;
                memset(&c, 0, sizeof(c));
                c.position = pos;
                c.category = PISCES_CAT_SNV;
                c.ref[0] = (char)rb;
                c.alt[0] = kBase[raw];
                c.support_by_dir[dir] = 1;
                c.well_anchored_by_dir[dir] = (anchor != 0 && anchor != PISCES_NUM_ANCHORS - 1);
                c.next = -1;
                orc_add_candidate(s, &c);
            }
        }
    }
    int64_t n = orc_call_all(s, ref_bases, ref_len, cfg, out, capacity, NULL, NULL);
    orc_state_destroy(s);
    if (n >= 0 && n_candidate_loci) *n_candidate_loci = count_candidate_loci(out, n);
    return n;
}

/* VariantCallingParameters.cs:57-156 defaults after Validate() */
void orc_default_config(PiscesHipConfig* c)
{
    memset(c, 0, sizeof(*c));
    c->abi_version = PISCES_HIP_ABI_VERSION;
    c->min_base_call_quality = 20;
    c->noise_level = 20;
    c->max_variant_qscore = 100;
    c->min_variant_qscore = 20;
    c->variant_qscore_filter = 30;
    c->min_coverage = 10;
    c->low_depth_filter = 10;
    c->min_genotype_qscore = 0;
    c->max_genotype_qscore = 100;
    c->low_gq_filter = -1;
    c->strand_bias_model = PISCES_SB_EXTENDED;
    c->filter_single_strand = 0;
    c->include_reference_calls = 1;
    c->emit_zero_coverage_refs = 0;
    c->expect_stitched_reads = 0;
    c->tile_loci = 64;
    c->block_size = 1000;
    c->min_frequency = 0.01f;
    c->variant_freq_filter = 0.01f;
    c->genotype_min_freq_filter = 0.01f;
    c->target_lod_frequency = 0.01f;
    c->strand_bias_threshold = 0.5f;
    c->no_call_filter_threshold = 0.6f;
    c->rmxn_max_repeat_length = 5;
    c->rmxn_min_repetitions = 9;
    c->rmxn_frequency_limit = 0.35f;
    c->collapse = 1;
    c->collapse_freq_threshold = 0.0f;
    c->collapse_freq_ratio_threshold = 0.5f;
    c->call_mnvs = 0;
    c->max_mnv_length = 3;
    c->max_gap_between_mnv = 1;
    c->noise_model = PISCES_NOISE_FLAT;
    c->ploidy = PISCES_PLOIDY_SOMATIC;
    c->diploid_snv_params[0] = c->diploid_indel_params[0] = 0.20f;
    c->diploid_snv_params[1] = c->diploid_indel_params[1] = 0.70f;
    c->diploid_snv_params[2] = c->diploid_indel_params[2] = 0.80f;
}


/* ---- interval shards on host threads: the -threadbychr analogue of the reference (one job thread per chromosome,
 * BaseGenomeProcessor.cs:49-71, JobManager.cs:70-73), used only as bench.py's threaded CPU baseline ---- */
#include <pthread.h>

static void* orc_shard_main(void* arg)
{
    OrcShardJob* j = (OrcShardJob*)arg;
    j->n_out = 0;
    j->n_loci = 0;
    for (int32_t p = 0; p < j->passes; p++) {
        int64_t nl = 0;
        int64_t n = orc_run_reads(j->batch, j->ref_bases, j->ref_len, j->region_start, j->region_loci, j->cfg, j->out, j->capacity, &nl);
        if (n < 0) { j->n_out = n; return NULL; }
        j->n_out = n;
        j->n_loci += nl;
    }
    return NULL;
}

int32_t orc_run_reads_sharded(OrcShardJob* jobs, int32_t n_jobs)
{
    pthread_t* th = (pthread_t*)calloc((size_t)(n_jobs > 0 ? n_jobs : 1), sizeof(pthread_t));
    int32_t started = 0, rc = 0;
    for (; started < n_jobs; started++)
        if (pthread_create(&th[started], NULL, orc_shard_main, &jobs[started]) != 0) { rc = -1; break; }
    for (int32_t i = 0; i < started; i++) pthread_join(th[i], NULL);
    free(th);
    for (int32_t i = 0; i < n_jobs && rc == 0; i++)
        if (jobs[i].n_out < 0) rc = -2;
    return rc;
}
