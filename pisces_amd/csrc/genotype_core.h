// genotype_core.h — PloidyModel.DiploidByThresholding / Haploid for the alleles of ONE locus, one source for the host and the device
// (src/lib/Pisces.Genotyping/Thresholding/DiploidThresholdingGenotyper.cs:54-141, GenotypeCalculatorUtilities.cs:11-237,
// DiploidGenotypeQualityCalculator.cs:12-105, Haploid/HaploidGenotyper.cs:36-83, HaploidGenotypeQualityCalculator.cs:10-59; the ln PMFs are
// MathNet.Numerics 4.5.1's Poisson.ProbabilityLn / Binomial.ProbabilityLn with its GammaLn / FactorialLn).
//   diploid.cpp                 the host form: the per-locus pass of a flush whose rows come from two kernels, pisces_hip_set_genotypes
//   genotype_loci_kernel        (kernels.hip.h) lane = locus over the tile kernels' record slots: the device-resident surface and the
//                               flushes that have no candidate rows
// The alleles are plain records; the order of two alleles of equal frequency (ordinal order of their REF, then ALT strings) comes from the
// caller: strings on the host, the slot rank (A C G T) on the device.
#pragma once
#include <math.h>
#include <stdint.h>

#include "../../include/pisces_hip.h"

#if defined(__HIPCC__)
#define PISCES_GHD __host__ __device__
#else
#define PISCES_GHD
#endif

namespace pisces {
namespace genotype {

struct Allele {
    int32_t category;            // PISCES_CAT_*
    int32_t support, coverage, ref_support;
    // results
    int32_t genotype, genotype_qscore, phase_set_index;
    bool multi_allelic;          // FilterType.MultiAllelicSite
    bool prune;                  // the genotyper asks the caller to drop this allele
};

// MathNet.Numerics 4.5.1 SpecialFunctions.GammaLn (Lanczos, g = 10.900511) for z >= 0.5 (arguments here are counts + 1)
PISCES_GHD inline double gamma_ln(double z)
{
    const double dk[11] = {2.48574089138753565546e-5,  1.05142378581721974210,    -3.45687097222016235469,
                           4.51227709466894823700,     -2.98285225323576655721,   1.05639711577126713077,
                           -1.95428773191645869583e-1, 1.70970543404441224307e-2, -5.71926117404305781283e-4,
                           4.63399473359905636708e-6,  -2.71994908488607703910e-9};
    const double r = 10.900511, log_two_sqrt_e_over_pi = 0.6207822376352452223455184457816472122518527279025978, e = 2.7182818284590452354;
    double s = dk[0];
    for (int i = 1; i <= 10; i++) s += dk[i] / (z + i - 1.0);
    return log(s) + log_two_sqrt_e_over_pi + ((z - 0.5) * log((z - 0.5 + r) / e));
}
// SpecialFunctions.FactorialLn: the log of the cached factorial below 171, GammaLn(x + 1) from there
PISCES_GHD inline double factorial_ln(int x)
{
    if (x <= 1) return 0.0;
    if (x < 171) {
        double c = 1.0;
        for (int i = 2; i <= x; i++) c = c * i;
        return log(c);
    }
    return gamma_ln(x + 1.0);
}
PISCES_GHD inline double poisson_ln_pmf(double lambda, int k) { return -lambda + (k * log(lambda)) - factorial_ln(k); }   // Poisson.ProbabilityLn
PISCES_GHD inline double binomial_ln_pmf(double p, int n, int k)   // Binomial.ProbabilityLn
{
    const double ninf = -INFINITY;
    if (k < 0 || k > n) return ninf;
    if (p == 0.0) return k == 0 ? 0.0 : ninf;
    if (p == 1.0) return k == n ? 0.0 : ninf;
    return (factorial_ln(n) - factorial_ln(k) - factorial_ln(n - k)) + (k * log(p)) + ((n - k) * log(1.0 - p));
}
// CalledAllele.Frequency / RefFrequency (CalledAllele.cs:49-52,121-124)
PISCES_GHD inline float frequency_of(int32_t support, int32_t coverage)
{
    if (coverage == 0) return 0.0f;
    const float f = (float)support / (float)coverage;
    return f < 1.0f ? f : 1.0f;
}
PISCES_GHD inline int32_t clamp_q(double v, int32_t lo, int32_t hi)   // C#: (int) of a double, then Math.Max(Math.Min(q, max), min)
{
    const int32_t q = (v > -2147483649.0 && v < 2147483648.0) ? (int32_t)v : INT32_MIN;
    const int32_t a = q < hi ? q : hi;
    return a > lo ? a : lo;
}

// DiploidGenotypeQualityCalculator.Compute
PISCES_GHD inline int32_t diploid_qscore(int32_t calledGT, int32_t totalCoverage, int32_t alleleSupport, int32_t minQScore, int32_t maxQScore)
{
    if (totalCoverage == 0) return minQScore;
    const float noiseHomRef = 0.05f, noiseHomAlt = 0.075f, noiseHetAlt = 0.10f, expectedHetFreq = 0.40f;
    const float depth = (float)totalCoverage;
    const float frequency = frequency_of(alleleSupport, totalCoverage);
    const int nonAlleleCalls = totalCoverage - alleleSupport > 0 ? totalCoverage - alleleSupport : 0;
    double h0 = 0, h1 = 0;
    switch (calledGT) {
    case PISCES_GT_HOM_REF:
        h0 = poisson_ln_pmf((double)(noiseHomRef * depth), nonAlleleCalls);
        h1 = binomial_ln_pmf((double)expectedHetFreq, totalCoverage, nonAlleleCalls);
        break;
    case PISCES_GT_HOM_ALT:
        h0 = poisson_ln_pmf((double)(noiseHomAlt * depth), nonAlleleCalls);
        h1 = binomial_ln_pmf((double)expectedHetFreq, totalCoverage, alleleSupport);
        break;
    case PISCES_GT_HET_ALT1_ALT2:
    case PISCES_GT_HET_ALT_REF: {
        const int k = (int)(depth * frequency);
        h0 = binomial_ln_pmf((double)expectedHetFreq, totalCoverage, k);
        h1 = frequency >= 0.50 ? binomial_ln_pmf((double)(1 - noiseHetAlt), totalCoverage, k) : binomial_ln_pmf((double)noiseHetAlt, totalCoverage, k);
        break;
    }
    default: return minQScore;
    }
    const double v = floor(10.0 * 0.4342944819032518 * (h0 - h1));   // Math.Log10(Math.E)
    if (h1 <= (double)INT32_MIN && h0 > h1) return maxQScore;
    if (h0 <= (double)INT32_MIN && h0 < h1) return minQScore;
    return clamp_q(v, minQScore, maxQScore);
}

// FilterAndOrderAllelesByFrequency: the variant alleles at or above the minor frequency, by descending frequency; `order` has room for n
// indices; before(x, y): allele x goes before allele y among equals (ordinal order of REF, then ALT).  Returns how many.
template <typename Before>
PISCES_GHD inline int order_by_frequency(Allele* a, int n, int* order, float minorVF, Before before)
{
    int nv = 0;
    for (int i = 0; i < n; i++) {
        a[i].prune = false;
        if (a[i].category == PISCES_CAT_REFERENCE) continue;
        const float fi = frequency_of(a[i].support, a[i].coverage);
        if (!((double)fi >= (double)minorVF)) { a[i].prune = true; continue; }
        int at = nv;   // insertion behind every allele that goes before this one (a stable sort of the arrival order)
        while (at > 0) {
            const int j = order[at - 1];
            const float fj = frequency_of(a[j].support, a[j].coverage);
            const bool j_first = fj != fi ? fj > fi : !before(i, j);
            if (j_first) break;
            order[at] = j;
            at--;
        }
        order[at] = i;
        nv++;
    }
    return nv;
}
// GetReferenceFrequency
PISCES_GHD inline double reference_frequency(const Allele* a, int n)
{
    if (n == 1) return frequency_of(a[0].ref_support, a[0].coverage);
    double refBySNP = 0, indelCount = 0;
    for (int i = 0; i < n; i++) {
        if (a[i].category == PISCES_CAT_REFERENCE) return frequency_of(a[i].support, a[i].coverage);
        if (a[i].category == PISCES_CAT_SNV) refBySNP = frequency_of(a[i].ref_support, a[i].coverage);
        else indelCount += frequency_of(a[i].support, a[i].coverage);
    }
    if (n < 1) return 0.0;
    return refBySNP - indelCount > 0.0 ? refBySNP - indelCount : 0.0;
}

// DiploidThresholdingGenotyper.SetGenotypes over the alleles of one locus (Reference rows already gone when a variant is there);
// snv / indel = {MinorVF, MajorVF, SumVFforMultiAllelicSite}.  Returns the locus genotype.
template <typename Before>
PISCES_GHD inline int32_t diploid_set(Allele* a, int n, int* order, const float snv[3], const float indel[3], int32_t minDepthToGenotype, int32_t minGQ,
                                      int32_t maxGQ, Before before)
{
    const int nv = order_by_frequency(a, n, order, snv[0], before);
    const double referenceFrequency = reference_frequency(a, n);
    const bool refExists = referenceFrequency >= (double)snv[0];
    bool depthIssue = false;
    for (int i = 0; i < n; i++) depthIssue |= a[i].coverage < minDepthToGenotype;
    const float f0 = nv ? frequency_of(a[order[0]].support, a[order[0]].coverage) : 0.0f;
    const bool refCall = nv == 0 || f0 < snv[0];
    const float* par = (!refCall && a[order[0]].category != PISCES_CAT_SNV) ? indel : snv;   // SelectParameters
    int prelim = 0;   // GetPreliminaryGenotype: 0 HomozygousRef, 1 HeterozygousAltRef, 2 HomozygousAlt
    if (!refCall) prelim = (f0 >= par[0] && f0 <= par[1]) ? 1 : (f0 > par[1]) ? 2 : 0;
    // ConvertSimpleGenotypeToComplexGenotype
    int32_t gt;
    if (depthIssue) gt = refCall ? PISCES_GT_REF_LIKE_NOCALL : PISCES_GT_ALT_LIKE_NOCALL;
    else if (prelim == 0) {
        if (!refExists) gt = PISCES_GT_REF_LIKE_NOCALL;
        else gt = (n > 0 && a[0].category == PISCES_CAT_REFERENCE && (1 - frequency_of(a[0].support, a[0].coverage)) > par[0]) ? PISCES_GT_REF_AND_NOCALL : PISCES_GT_HOM_REF;
    } else if (prelim == 1) {
        if (nv == 1) gt = refExists ? PISCES_GT_HET_ALT_REF : PISCES_GT_ALT_AND_NOCALL;
        else {
            bool fail;   // CheckForTriAllelicIssue
            if (a[order[nv - 1]].category != PISCES_CAT_SNV) fail = false;
            else if (refExists && ((double)f0 + referenceFrequency) < (double)par[2]) fail = true;
            else fail = (f0 + frequency_of(a[order[1]].support, a[order[1]].coverage)) < par[2];
            if (fail) {
                for (int i = 0; i < n; i++) a[i].multi_allelic = true;
                gt = refExists ? PISCES_GT_ALT_LIKE_NOCALL : PISCES_GT_ALT12_LIKE_NOCALL;
            } else {
                gt = refExists ? PISCES_GT_HET_ALT_REF : PISCES_GT_HET_ALT1_ALT2;
            }
        }
    } else gt = PISCES_GT_HOM_ALT;
    // GetAllelesToPruneBasedOnGTCall
    int allowed = 0;
    if (gt == PISCES_GT_ALT_AND_NOCALL || gt == PISCES_GT_ALT_LIKE_NOCALL || gt == PISCES_GT_HOM_ALT || gt == PISCES_GT_HET_ALT_REF) allowed = 1;
    else if (gt == PISCES_GT_ALT12_LIKE_NOCALL || gt == PISCES_GT_HET_ALT1_ALT2) allowed = 2;
    for (int k = allowed; k < nv; k++) a[order[k]].prune = true;
    // SetGenotypes
    int phase = 1;
    for (int i = 0; i < n; i++) {
        a[i].genotype = gt;
        a[i].genotype_qscore = diploid_qscore(gt, a[i].coverage, a[i].support, minGQ, maxGQ);
        a[i].phase_set_index = a[i].category == PISCES_CAT_REFERENCE ? 0 : phase++;
    }
    return gt;
}

// HaploidGenotyper.SetGenotypes with HaploidGenotypeQualityCalculator; minorVF / majorVF are the SNV thresholding parameters
// (GenotypeCreator.cs:21-22)
template <typename Before>
PISCES_GHD inline int32_t haploid_set(Allele* a, int n, int* order, float minorVF, float majorVF, int32_t minDepthToGenotype, int32_t minGQ, int32_t maxGQ,
                                      Before before)
{
    const int nv = order_by_frequency(a, n, order, minorVF, before);
    const double referenceFrequency = reference_frequency(a, n);
    const bool refExists = referenceFrequency >= (double)minorVF;
    bool depthIssue = false;
    for (int i = 0; i < n; i++) depthIssue |= a[i].coverage < minDepthToGenotype;
    const float f0 = nv ? frequency_of(a[order[0]].support, a[order[0]].coverage) : 0.0f;
    const bool refCall = nv == 0 || f0 < minorVF;
    int32_t gt = PISCES_GT_HEMI_NOCALL;
    if (!depthIssue && refCall && refExists && referenceFrequency > (double)majorVF) gt = PISCES_GT_HEMI_REF;
    if (!depthIssue && !refCall && !refExists && f0 > majorVF) gt = PISCES_GT_HEMI_ALT;
    for (int k = gt == PISCES_GT_HEMI_ALT ? 1 : 0; k < nv; k++) a[order[k]].prune = true;
    for (int i = 0; i < n; i++) {
        a[i].genotype = gt;
        a[i].phase_set_index = 0;
        a[i].multi_allelic = false;
        // HaploidGenotypeQualityCalculator.Compute
        int32_t gq = minGQ;
        if (a[i].coverage != 0 && (gt == PISCES_GT_HEMI_REF || gt == PISCES_GT_HEMI_ALT)) {
            const float depth = (float)a[i].coverage;
            const int nonAlleleCalls = a[i].coverage - a[i].support > 0 ? a[i].coverage - a[i].support : 0;
            const double h0 = poisson_ln_pmf((double)((gt == PISCES_GT_HEMI_REF ? 0.05f : 0.075f) * depth), nonAlleleCalls);
            const double h1 = binomial_ln_pmf((double)0.40f, a[i].coverage, gt == PISCES_GT_HEMI_REF ? nonAlleleCalls : a[i].support);
            gq = clamp_q(floor(10.0 * 0.4342944819032518 * (h0 - h1)), minGQ, maxGQ);
        }
        a[i].genotype_qscore = gq;
    }
    return gt;
}

}  // namespace genotype
}  // namespace pisces
