// diploid.h — PloidyModel.DiploidByThresholding on the host: one genotype per locus from the alleles' frequencies
// (src/lib/Pisces.Genotyping/Thresholding/DiploidThresholdingGenotyper.cs:54-141, GenotypeCalculatorUtilities.cs:11-237) and the
// genotype q-score of each allele (DiploidGenotypeQualityCalculator.cs:12-105, MathNet.Numerics 4.5.1 Poisson / Binomial ln PMF).
// A per-locus decision over a handful of records that are on the host already when pisces_hip_flush assembles its output.
#pragma once
#include <stdint.h>

#include <string>
#include <vector>

namespace pisces {

struct DiploidAllele {
    int32_t category = 0;            // PISCES_CAT_*
    std::string ref, alt;
    int32_t support = 0, coverage = 0, ref_support = 0;
    // results
    int32_t genotype = 0, genotype_qscore = 0, phase_set_index = 0;
    bool multi_allelic = false;      // FilterType.MultiAllelicSite
    bool prune = false;              // the genotyper asks the caller to drop this allele
};

// DiploidGenotypeQualityCalculator.Compute
int32_t diploid_genotype_qscore(int32_t genotype, int32_t total_coverage, int32_t allele_support, int32_t min_q, int32_t max_q);
// DiploidThresholdingGenotyper.SetGenotypes over the alleles of one locus (Reference rows already gone when a variant is there);
// params = {MinorVF, MajorVF, SumVFforMultiAllelicSite}.  Returns the locus genotype.
int32_t diploid_set_genotypes(std::vector<DiploidAllele>& alleles, const float snv[3], const float indel[3], int32_t min_depth_to_genotype,
                              int32_t min_gq, int32_t max_gq);

// HaploidGenotyper.SetGenotypes (src/lib/Pisces.Genotyping/Haploid/HaploidGenotyper.cs:36-83) with HaploidGenotypeQualityCalculator
// (:10-59); minor_vf / major_vf are the SNV thresholding parameters (GenotypeCreator.cs:21-22)
int32_t haploid_set_genotypes(std::vector<DiploidAllele>& alleles, float minor_vf, float major_vf, int32_t min_depth_to_genotype, int32_t min_gq,
                              int32_t max_gq);

}  // namespace pisces
