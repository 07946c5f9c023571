// diploid.cpp — see diploid.h: the host form of genotype_core.h (alleles with their strings; the order among equal frequencies is the
// ordinal order of REF, then ALT).
#include "diploid.h"

#include <vector>

#include "genotype_core.h"

namespace pisces {
namespace {

struct Scratch {
    std::vector<genotype::Allele> a;
    std::vector<int> order;
    explicit Scratch(const std::vector<DiploidAllele>& in) : a(in.size()), order(in.size() + 1)
    {
        for (size_t i = 0; i < in.size(); i++) {
            genotype::Allele& g = a[i];
            g.category = in[i].category; g.support = in[i].support; g.coverage = in[i].coverage; g.ref_support = in[i].ref_support;
            g.genotype = 0; g.genotype_qscore = 0; g.phase_set_index = 0;
            g.multi_allelic = in[i].multi_allelic; g.prune = false;
        }
    }
    void back(std::vector<DiploidAllele>& out) const
    {
        for (size_t i = 0; i < out.size(); i++) {
            out[i].genotype = a[i].genotype; out[i].genotype_qscore = a[i].genotype_qscore; out[i].phase_set_index = a[i].phase_set_index;
            out[i].multi_allelic = a[i].multi_allelic; out[i].prune = a[i].prune;
        }
    }
};

}  // namespace

int32_t diploid_genotype_qscore(int32_t calledGT, int32_t totalCoverage, int32_t alleleSupport, int32_t minQScore, int32_t maxQScore)
{
    return genotype::diploid_qscore(calledGT, totalCoverage, alleleSupport, minQScore, maxQScore);
}

int32_t diploid_set_genotypes(std::vector<DiploidAllele>& alleles, const float snv[3], const float indel[3], int32_t minDepthToGenotype,
                              int32_t minGQ, int32_t maxGQ)
{
    Scratch s(alleles);
    auto before = [&](int x, int y) {
        if (alleles[(size_t)x].ref != alleles[(size_t)y].ref) return alleles[(size_t)x].ref < alleles[(size_t)y].ref;
        return alleles[(size_t)x].alt < alleles[(size_t)y].alt;
    };
    const int32_t gt = genotype::diploid_set(s.a.data(), (int)alleles.size(), s.order.data(), snv, indel, minDepthToGenotype, minGQ, maxGQ, before);
    s.back(alleles);
    return gt;
}

int32_t haploid_set_genotypes(std::vector<DiploidAllele>& alleles, float minorVF, float majorVF, int32_t minDepthToGenotype, int32_t minGQ,
                              int32_t maxGQ)
{
    Scratch s(alleles);
    auto before = [&](int x, int y) {
        if (alleles[(size_t)x].ref != alleles[(size_t)y].ref) return alleles[(size_t)x].ref < alleles[(size_t)y].ref;
        return alleles[(size_t)x].alt < alleles[(size_t)y].alt;
    };
    const int32_t gt = genotype::haploid_set(s.a.data(), (int)alleles.size(), s.order.data(), minorVF, majorVF, minDepthToGenotype, minGQ, maxGQ, before);
    s.back(alleles);
    return gt;
}

}  // namespace pisces
