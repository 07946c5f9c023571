// diploid.cpp — see diploid.h.
#include "diploid.h"

#include <algorithm>
#include <cmath>
#include <limits>

#include "../../include/pisces_hip.h"

namespace pisces {
namespace {

// MathNet.Numerics 4.5.1 SpecialFunctions.GammaLn (Lanczos, g = 10.900511) for z >= 0.5 (arguments here are counts + 1)
double gamma_ln(double z)
{
    static const double dk[11] = {2.48574089138753565546e-5,  1.05142378581721974210,    -3.45687097222016235469,
                                  4.51227709466894823700,     -2.98285225323576655721,   1.05639711577126713077,
                                  -1.95428773191645869583e-1, 1.70970543404441224307e-2, -5.71926117404305781283e-4,
                                  4.63399473359905636708e-6,  -2.71994908488607703910e-9};
    const double r = 10.900511, log_two_sqrt_e_over_pi = 0.6207822376352452223455184457816472122518527279025978, e = 2.7182818284590452354;
    double s = dk[0];
    for (int i = 1; i <= 10; i++) s += dk[i] / (z + i - 1.0);
    return std::log(s) + log_two_sqrt_e_over_pi + ((z - 0.5) * std::log((z - 0.5 + r) / e));
}
// SpecialFunctions.FactorialLn: the log of the cached factorial below 171, GammaLn(x + 1) from there
double factorial_ln(int x)
{
    if (x <= 1) return 0.0;
    if (x < 171) {
        double c = 1.0;
        for (int i = 2; i <= x; i++) c = c * i;
        return std::log(c);
    }
    return gamma_ln(x + 1.0);
}
double poisson_ln_pmf(double lambda, int k) { return -lambda + (k * std::log(lambda)) - factorial_ln(k); }   // Poisson.ProbabilityLn
double binomial_ln_pmf(double p, int n, int k)   // Binomial.ProbabilityLn
{
    const double ninf = -std::numeric_limits<double>::infinity();
    if (k < 0 || k > n) return ninf;
    if (p == 0.0) return k == 0 ? 0.0 : ninf;
    if (p == 1.0) return k == n ? 0.0 : ninf;
    return (factorial_ln(n) - factorial_ln(k) - factorial_ln(n - k)) + (k * std::log(p)) + ((n - k) * std::log(1.0 - p));
}
// CalledAllele.Frequency / RefFrequency (CalledAllele.cs:49-52,121-124)
float frequency_f(int32_t support, int32_t coverage)
{
    if (coverage == 0) return 0.0f;
    const float f = (float)support / (float)coverage;
    return f < 1.0f ? f : 1.0f;
}

}  // namespace

int32_t diploid_genotype_qscore(int32_t calledGT, int32_t totalCoverage, int32_t alleleSupport, int32_t minQScore, int32_t maxQScore)
{
    if (totalCoverage == 0) return minQScore;
    const float noiseHomRef = 0.05f, noiseHomAlt = 0.075f, noiseHetAlt = 0.10f, expectedHetFreq = 0.40f;
    const float depth = (float)totalCoverage;
    const float frequency = frequency_f(alleleSupport, totalCoverage);
    const int nonAlleleCalls = std::max(totalCoverage - alleleSupport, 0);
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
    const double v = std::floor(10.0 * 0.4342944819032518 * (h0 - h1));   // Math.Log10(Math.E)
    const int32_t qScore = (v > -2147483649.0 && v < 2147483648.0) ? (int32_t)v : INT32_MIN;   // C# (int) of an out-of-range double
    if (h1 <= (double)INT32_MIN && h0 > h1) return maxQScore;
    if (h0 <= (double)INT32_MIN && h0 < h1) return minQScore;
    return std::max(std::min(qScore, maxQScore), minQScore);
}

int32_t diploid_set_genotypes(std::vector<DiploidAllele>& alleles, const float snv[3], const float indel[3], int32_t minDepthToGenotype,
                              int32_t minGQ, int32_t maxGQ)
{
    const int n = (int)alleles.size();
    auto freq = [&](int i) { return frequency_f(alleles[(size_t)i].support, alleles[(size_t)i].coverage); };
    // FilterAndOrderAllelesByFrequency
    std::vector<int> order;
    for (int i = 0; i < n; i++) {
        alleles[(size_t)i].prune = false;
        if (alleles[(size_t)i].category == PISCES_CAT_REFERENCE) continue;
        if ((double)freq(i) >= (double)snv[0]) order.push_back(i);
        else alleles[(size_t)i].prune = true;
    }
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) {
        if (freq(x) != freq(y)) return freq(x) > freq(y);
        if (alleles[(size_t)x].ref != alleles[(size_t)y].ref) return alleles[(size_t)x].ref < alleles[(size_t)y].ref;
        return alleles[(size_t)x].alt < alleles[(size_t)y].alt;
    });
    const int nv = (int)order.size();
    // GetReferenceFrequency
    double referenceFrequency = 0;
    if (n == 1) referenceFrequency = frequency_f(alleles[0].ref_support, alleles[0].coverage);
    else if (n > 1) {
        double refFrequencyCountBySNP = 0, indelFrequencyCount = 0;
        bool returned = false;
        for (int i = 0; i < n; i++) {
            if (alleles[(size_t)i].category == PISCES_CAT_REFERENCE) { referenceFrequency = freq(i); returned = true; break; }
            if (alleles[(size_t)i].category == PISCES_CAT_SNV) refFrequencyCountBySNP = frequency_f(alleles[(size_t)i].ref_support, alleles[(size_t)i].coverage);
            else indelFrequencyCount += freq(i);
        }
        if (!returned) referenceFrequency = std::max(refFrequencyCountBySNP - indelFrequencyCount, 0.0);
    }
    const bool refExists = referenceFrequency >= (double)snv[0];
    bool depthIssue = false;
    for (int i = 0; i < n; i++) depthIssue |= alleles[(size_t)i].coverage < minDepthToGenotype;
    const float f0 = nv ? freq(order[0]) : 0.0f;
    const bool refCall = nv == 0 || f0 < snv[0];
    const float* par = (!refCall && alleles[(size_t)order[0]].category != PISCES_CAT_SNV) ? indel : snv;   // SelectParameters
    int prelim = 0;   // GetPreliminaryGenotype: 0 HomozygousRef, 1 HeterozygousAltRef, 2 HomozygousAlt
    if (!refCall) prelim = (f0 >= par[0] && f0 <= par[1]) ? 1 : (f0 > par[1]) ? 2 : 0;
    // ConvertSimpleGenotypeToComplexGenotype
    int32_t gt;
    if (depthIssue) gt = refCall ? PISCES_GT_REF_LIKE_NOCALL : PISCES_GT_ALT_LIKE_NOCALL;
    else if (prelim == 0) {
        if (!refExists) gt = PISCES_GT_REF_LIKE_NOCALL;
        else gt = (n > 0 && alleles[0].category == PISCES_CAT_REFERENCE && (1 - freq(0)) > par[0]) ? PISCES_GT_REF_AND_NOCALL : PISCES_GT_HOM_REF;
    } else if (prelim == 1) {
        if (nv == 1) gt = refExists ? PISCES_GT_HET_ALT_REF : PISCES_GT_ALT_AND_NOCALL;
        else {
            bool fail;   // CheckForTriAllelicIssue
            if (alleles[(size_t)order[(size_t)nv - 1]].category != PISCES_CAT_SNV) fail = false;
            else if (refExists && ((double)f0 + referenceFrequency) < (double)par[2]) fail = true;
            else fail = (f0 + freq(order[1])) < par[2];
            if (fail) {
                for (auto& a : alleles) a.multi_allelic = true;
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
    for (int k = allowed; k < nv; k++) alleles[(size_t)order[(size_t)k]].prune = true;
    // SetGenotypes
    int phase = 1;
    for (auto& a : alleles) {
        a.genotype = gt;
        a.genotype_qscore = diploid_genotype_qscore(gt, a.coverage, a.support, minGQ, maxGQ);
        a.phase_set_index = a.category == PISCES_CAT_REFERENCE ? 0 : phase++;
    }
    return gt;
}

int32_t haploid_set_genotypes(std::vector<DiploidAllele>& alleles, float minorVF, float majorVF, int32_t minDepthToGenotype, int32_t minGQ,
                              int32_t maxGQ)
{
    const int n = (int)alleles.size();
    auto freq = [&](int i) { return frequency_f(alleles[(size_t)i].support, alleles[(size_t)i].coverage); };
    std::vector<int> order;
    for (int i = 0; i < n; i++) {
        alleles[(size_t)i].prune = false;
        if (alleles[(size_t)i].category == PISCES_CAT_REFERENCE) continue;
        if ((double)freq(i) >= (double)minorVF) order.push_back(i);
        else alleles[(size_t)i].prune = true;
    }
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) {
        if (freq(x) != freq(y)) return freq(x) > freq(y);
        if (alleles[(size_t)x].ref != alleles[(size_t)y].ref) return alleles[(size_t)x].ref < alleles[(size_t)y].ref;
        return alleles[(size_t)x].alt < alleles[(size_t)y].alt;
    });
    const int nv = (int)order.size();
    double referenceFrequency = 0;
    if (n == 1) referenceFrequency = frequency_f(alleles[0].ref_support, alleles[0].coverage);
    else if (n > 1) {
        double refBySNP = 0, indelCount = 0;
        bool returned = false;
        for (int i = 0; i < n; i++) {
            if (alleles[(size_t)i].category == PISCES_CAT_REFERENCE) { referenceFrequency = freq(i); returned = true; break; }
            if (alleles[(size_t)i].category == PISCES_CAT_SNV) refBySNP = frequency_f(alleles[(size_t)i].ref_support, alleles[(size_t)i].coverage);
            else indelCount += freq(i);
        }
        if (!returned) referenceFrequency = std::max(refBySNP - indelCount, 0.0);
    }
    const bool refExists = referenceFrequency >= (double)minorVF;
    bool depthIssue = false;
    for (int i = 0; i < n; i++) depthIssue |= alleles[(size_t)i].coverage < minDepthToGenotype;
    const float f0 = nv ? freq(order[0]) : 0.0f;
    const bool refCall = nv == 0 || f0 < minorVF;
    int32_t gt = PISCES_GT_HEMI_NOCALL;
    if (!depthIssue && refCall && refExists && referenceFrequency > (double)majorVF) gt = PISCES_GT_HEMI_REF;
    if (!depthIssue && !refCall && !refExists && f0 > majorVF) gt = PISCES_GT_HEMI_ALT;
    for (int k = gt == PISCES_GT_HEMI_ALT ? 1 : 0; k < nv; k++) alleles[(size_t)order[(size_t)k]].prune = true;
    for (auto& a : alleles) {
        a.genotype = gt;
        a.phase_set_index = 0;
        a.multi_allelic = false;
        // HaploidGenotypeQualityCalculator.Compute
        int32_t gq = minGQ;
        if (a.coverage != 0 && (gt == PISCES_GT_HEMI_REF || gt == PISCES_GT_HEMI_ALT)) {
            const float depth = (float)a.coverage;
            const int nonAlleleCalls = std::max(a.coverage - a.support, 0);
            const double h0 = poisson_ln_pmf((double)((gt == PISCES_GT_HEMI_REF ? 0.05f : 0.075f) * depth), nonAlleleCalls);
            const double h1 = binomial_ln_pmf((double)0.40f, a.coverage, gt == PISCES_GT_HEMI_REF ? nonAlleleCalls : a.support);
            const double v = std::floor(10.0 * 0.4342944819032518 * (h0 - h1));
            const int32_t q = (v > -2147483649.0 && v < 2147483648.0) ? (int32_t)v : INT32_MIN;
            gq = std::max(std::min(q, maxGQ), minGQ);
        }
        a.genotype_qscore = gq;
    }
    return gt;
}

}  // namespace pisces
