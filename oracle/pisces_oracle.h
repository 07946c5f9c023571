/*
 * pisces_oracle.h — TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C, FP contraction off) of the reference's per-locus
 * pileup-and-likelihood path (Illumina/Pisces v5.2.11, C#).  It exists to check the
 * HIP path; nothing in the product (pisces_amd/, libpisceship.so) may link, import or
 * call it.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it.
 *
 * Parity pinning: the reference cannot be compiled or run here (no dotnet/mono), so the
 * restatement is pinned by the reference's own known-answer tests, transcribed as
 * fixtures under tests/golden/ (see tests/golden/README.md for file:line of each).
 * MathNet.Numerics 4.5.1 (NuGet dependency, not in /root/reference) is restated from its
 * published algorithm (Cephes igam/igamc + Lanczos GammaLn) and pinned by
 * QualityCalculatorTests.cs:65-75.
 *
 * Every function cites the reference file:line it follows (paths relative to
 * /root/reference/src).
 */
#ifndef PISCES_ORACLE_H
#define PISCES_ORACLE_H

#include <stdint.h>
#include "../include/pisces_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAX_ALLELE 320

typedef struct OrcRead {
    int32_t position;            /* Read.Position (1-based) */
    int32_t n_cigar;
    const uint8_t* cigar_op;
    const uint32_t* cigar_len;
    int32_t read_len;
    const uint8_t* bases;
    const uint8_t* quals;
    const uint8_t* dirs;         /* per-base DirectionType or NULL */
    int32_t is_reverse;
    const int32_t* posmap_override; /* tests poke PositionMap directly; NULL = from CIGAR */
    const uint8_t* expanded_dirs;   /* Read.CigarDirections.Expand(): one DirectionType per base of the expanded CIGAR (deleted bases
                                     * included); NULL = CigarDirections == null */
    int32_t n_expanded;
} OrcRead;

typedef struct OrcCandidate {
    int32_t position;
    int32_t category;
    char ref[ORC_MAX_ALLELE];
    char alt[ORC_MAX_ALLELE];
    int32_t support_by_dir[3];
    int32_t well_anchored_by_dir[3];
    int32_t open_left, open_right;
    int32_t next;                /* per-locus chain inside OrcState */
} OrcCandidate;

/* StrandBiasStats / BiasResults (lib/Pisces.Domain/Models/StrandBiasStats.cs) */
typedef struct OrcSbStats {
    double chance_false_neg, chance_false_pos, chance_var_freq_gt_zero, coverage, frequency, support;
} OrcSbStats;
typedef struct OrcBiasResults {
    double bias_score, gatk_bias_score;
    int32_t bias_acceptable, var_present_on_both, cov_present_on_both;
    OrcSbStats forward, reverse, overall, stitched;
} OrcBiasResults;

/* full CalledAllele working set (superset of the 64-byte record) */
typedef struct OrcCalled {
    int32_t position, category;
    char ref[ORC_MAX_ALLELE], alt[ORC_MAX_ALLELE];
    int32_t support_by_dir[3], well_anchored_by_dir[3];
    int32_t allele_support, well_anchored_support;
    int32_t total_coverage, reference_support, num_no_calls;
    int32_t coverage_by_dir[3];
    int32_t confident_start, confident_end, suspicious_start, suspicious_end;
    double unanchored_weight;
    double sum_of_base_quality;
    int32_t variant_qscore, noise_level_applied;
    float fraction_no_calls;
    OrcBiasResults sb;
    int32_t has_sb;
    uint32_t filters;
    int32_t genotype, genotype_qscore;
} OrcCalled;

typedef struct OrcState OrcState;

/* ---- math (lib/Pisces.Calculators/stats) ---- */
double orc_poisson_cdf(double num_occurrences, double expected);           /* stats/Poisson.cs:26 */
double orc_q_to_p(double q);                                               /* stats/MathOperations.cs:7 */
double orc_p_to_q(double p);                                               /* stats/MathOperations.cs:12 */
double orc_mathnet_gamma_ln(double z);                                     /* MathNet SpecialFunctions.GammaLn */
double orc_mathnet_gamma_lower_regularized(double a, double x);            /* MathNet SpecialFunctions.GammaLowerRegularized */
double orc_mathnet_poisson_cdf(double lambda, double x);                   /* MathNet Poisson.CumulativeDistribution */
double orc_mathnet_poisson_ln_pmf(double lambda, int32_t k);               /* MathNet Poisson.ProbabilityLn */

/* ---- calculators ---- */
double  orc_assign_pvalue(int32_t call_count, int32_t coverage, int32_t nl);              /* VariantQualityCalculator.cs:67-74 */
double  orc_raw_poisson_qscore(int32_t call_count, int32_t coverage, int32_t nl);          /* :27-52 */
int32_t orc_poisson_qscore(int32_t call_count, int32_t coverage, int32_t nl, int32_t max_q);/* :54-65 */
void    orc_strand_bias(const int32_t cov_by_dir[3], const int32_t sup_by_dir[3], int32_t q_noise,
                        double min_variant_freq, double acceptance, int32_t model, OrcBiasResults* out); /* StrandBiasCalculator.cs:21-72 */
int32_t orc_somatic_genotype(int32_t category, int32_t total_coverage, int32_t allele_support,
                             int32_t reference_support, float min_freq_filter, int32_t min_depth); /* SomaticGenotyper.cs:65-100 */
int32_t orc_somatic_gq(int32_t genotype, int32_t variant_q, int32_t total_coverage, int32_t allele_support,
                       float target_lod, int32_t min_gq, int32_t max_gq);                  /* SomaticGenotypeQualityCalculator.cs:10-48 */

/* ---- region state (lib/Pisces.Processing/RegionState) ---- */
OrcState* orc_state_create(int32_t start_position, int32_t n_loci, int32_t min_bq,
                           int32_t num_anchor_types, int32_t track_open_ended);
void      orc_state_destroy(OrcState* s);
int32_t   orc_add_allele_counts(OrcState* s, const OrcRead* r);            /* RegionStateManager.cs:118-220 */
int32_t   orc_get_allele_count(const OrcState* s, int32_t position, int32_t allele, int32_t dir,
                               int32_t min_anchor, int32_t max_anchor /* -1 = null */, int32_t from_end,
                               int32_t symmetric);                          /* AlleleCountHelper.cs:21-85 */
double    orc_get_sum_base_quality(const OrcState* s, int32_t position, int32_t allele, int32_t dir,
                               int32_t min_anchor, int32_t max_anchor, int32_t from_end, int32_t symmetric);
const int32_t* orc_counts_ptr(const OrcState* s);                           /* [n_loci][6][3][n_anchor_idx] */
int32_t   orc_num_anchor_indexes(const OrcState* s);
void      orc_add_gapped_mnv_ref(OrcState* s, int32_t position, int32_t count);
int32_t   orc_add_candidate(OrcState* s, const OrcCandidate* c);            /* RegionState.cs:94-174 */
int32_t   orc_num_candidates(const OrcState* s);
int32_t   orc_get_candidates(const OrcState* s, OrcCandidate* out, int32_t capacity); /* position order */

/* ---- candidate finder (lib/Pisces.Domain/Logic/CandidateVariantFinder.cs) ---- */
int32_t orc_find_candidates(const OrcRead* r, const uint8_t* ref_bases, int64_t ref_len,
                            int32_t min_bq, int32_t max_mnv_len, int32_t max_gap, int32_t call_mnvs,
                            int32_t anchor_size, OrcCandidate* out, int32_t capacity); /* :31-83 */
int32_t orc_check_deletion_quality(const OrcRead* r, int32_t op_start_index, int32_t min_bq); /* :294-320 */
int32_t orc_deletion_direction_for_stitched_read(const OrcRead* r, int32_t left_anchor_index, int32_t right_anchor_index); /* :468-487 */

/* ---- coverage + caller ---- */
void    orc_coverage_compute(OrcCalled* allele, const OrcState* s, int32_t consider_anchors,
                             int32_t expect_stitched);                      /* CoverageCalculator.cs:19-47 */
void    orc_called_from_candidate(OrcCalled* out, const OrcCandidate* c);   /* AlleleHelper.Map, AlleleHelper.cs:51-85 */
void    orc_process_variant(OrcCalled* v, const OrcState* s, const PiscesHipConfig* cfg); /* AlleleCaller.cs:208-234 */
/* AlleleCaller.CallForPositions (AlleleCaller.cs:60-141) without collapser / MNV reallocation,
 * over every candidate in the state plus (gVCF) a Reference candidate per position
 * (RegionState.GetAllCandidates, RegionState.cs:383-453).  Sorted output; returns count or
 * -(needed) when capacity is too small. */
int64_t orc_call_all(OrcState* s, const uint8_t* ref_bases, int64_t ref_len, const PiscesHipConfig* cfg,
                     PiscesCalledAllele* out, int64_t capacity, OrcCalled* full_out /* optional, same capacity */,
                     int64_t* total_num_called);

/* VariantCollapser.Collapse (exe/Pisces/Logic/VariantCalling/VariantCollapser.cs:31-79) on a candidate list, in place;
 * returns the new count.  max_cleared_position < 0 = null. */
/* the known (prior) variants the next orc_collapse / schedule run annotates with (VariantCollapser.cs:16-24, 178-190); n = 0 clears */
void orc_set_known_variants(const OrcCandidate* list, int32_t n);
/* ExcludeMNVsFromCollapsing for the schedule runs (VariantCollapser.cs:33; orc_collapse itself takes it as an argument) */
void orc_set_exclude_mnvs_from_collapsing(int32_t on);
int32_t orc_collapse(OrcCandidate* cands, int32_t n, const OrcState* src, float freq_threshold, float freq_ratio_threshold,
                     int32_t exclude_mnvs, int32_t consider_anchors, int32_t expect_stitched, int32_t max_cleared_position,
                     int32_t* n_collapsed, OrcCandidate* added_back, int32_t* n_added_back);

/* the same call over an explicit batch of candidates (ICandidateBatch.GetCandidates; Reference candidates allowed) */
int64_t orc_call_candidates(OrcState* s, const OrcCandidate* list, int64_t n_list, const uint8_t* ref_bases, int64_t ref_len,
                            const PiscesHipConfig* cfg, PiscesCalledAllele* out, int64_t capacity, OrcCalled* full_out,
                            int64_t* total_num_called);

/* ---- whole path on a read batch: the CPU baseline (SmallVariantCaller.Execute loop,
 * exe/Pisces/Logic/SmallVariantCaller.cs:79-116) ---- */
/* MathNet.Numerics 4.5.1 Beta / Binomial and the diploid (germline) pieces: DiploidGenotypeQualityCalculator, DiploidThresholdingGenotyper,
 * StrandBiasCalculator.PopulateDiploidStats */
double orc_mathnet_beta_regularized(double a, double b, double x);
double orc_mathnet_binomial_cdf(double p, int n, double x);
double orc_mathnet_binomial_lnpmf(double p, int n, int k);
void orc_sb_populate_diploid_stats(double support, double coverage, double minDetectableSNP, double out3[3]);
int32_t orc_diploid_gq(int32_t calledGT, int32_t totalCoverage, int32_t alleleSupport, int32_t minQScore, int32_t maxQScore);
int32_t orc_haploid_gq(int32_t calledGT, int32_t totalCoverage, int32_t alleleSupport, int32_t minQScore, int32_t maxQScore);
int32_t orc_haploid_set_genotypes(OrcCalled* alleles, int n, float minorVF, float majorVF, int32_t minDepthToGenotype, int32_t minGQ,
                                  int32_t maxGQ, uint8_t* prune);
int32_t orc_diploid_set_genotypes(OrcCalled* alleles, int n, const float snv[3], const float indel[3], int32_t minDepthToGenotype,
                                  int32_t minGQ, int32_t maxGQ, int32_t* phase_set_index, uint8_t* prune);
/* MnvReallocator.ReallocateFailedMnvs over arrays (test hook; max_position < 0 = null) */
int64_t orc_reallocate_failed_mnvs(const OrcCalled* failed, int64_t n_failed, OrcCalled* callable, int64_t n_callable, int64_t cap_callable,
                                   int32_t max_position, OrcCalled* outside, int64_t cap_outside, int64_t* n_outside);
/* one batch of the block schedule over whole blocks [first_position, last_position]; MaxClearedPosition = last_position */
int64_t orc_call_range(OrcState* s, const uint8_t* ref_bases, int64_t ref_len, const PiscesHipConfig* cfg, int32_t first_position,
                       int32_t last_position, PiscesCalledAllele* out, int64_t capacity, OrcCalled* full_out, int64_t* total_num_called);
int64_t orc_call_candidates_max(OrcState* s, const OrcCandidate* list, int64_t n_list, const uint8_t* ref_bases, int64_t ref_len,
                                const PiscesHipConfig* cfg, int32_t max_cleared_position, PiscesCalledAllele* out, int64_t capacity,
                                OrcCalled* full_out, int64_t* total_num_called);
int64_t orc_run_reads(const PiscesReadBatch* batch, const uint8_t* ref_bases, int64_t ref_len,
                      int32_t region_start, int32_t region_loci, const PiscesHipConfig* cfg,
                      PiscesCalledAllele* out, int64_t capacity, int64_t* n_candidate_loci);
int64_t orc_run_reads_full(const PiscesReadBatch* batch, const uint8_t* ref_bases, int64_t ref_len,
                      int32_t region_start, int32_t region_loci, const PiscesHipConfig* cfg,
                      PiscesCalledAllele* out, int64_t capacity, int64_t* n_candidate_loci,
                      OrcCalled* full_out /* optional: allele strings etc. */, int64_t* total_num_called);
int64_t orc_run_reads_blocks(const PiscesReadBatch* batch, const uint8_t* ref_bases, int64_t ref_len, int32_t region_start, int32_t region_loci,
                             const PiscesHipConfig* cfg, PiscesCalledAllele* out, int64_t capacity, OrcCalled* full_out, int64_t* total_num_called);
void    orc_track_blocks(OrcState* s, int32_t block_size);                  /* RegionState.MaxAlleleEndpoint per block, RegionState.cs:203-223 */
int32_t orc_batch_candidates(OrcState* s, int32_t first_position, int32_t last_position, int32_t up_to_position, OrcCandidate* out,
                             int32_t capacity, int32_t* from_other_blocks);   /* RegionStateManager.cs:283-334, 441-457; RegionState.cs:470-490 */
int32_t orc_next_batch(OrcState* s, int32_t up_to_position, int32_t* first_position, int32_t* last_position);   /* RegionStateManager.cs:283-334 */
void    orc_done_processing(OrcState* s, int32_t last_position);           /* RegionStateManager.cs:336-360 */
int64_t orc_call_range_up_to(OrcState* s, const uint8_t* ref_bases, int64_t ref_len, const PiscesHipConfig* cfg, int32_t first_position,
                             int32_t last_position, int32_t up_to_position, PiscesCalledAllele* out, int64_t capacity, OrcCalled* full_out,
                             int64_t* total_num_called);                    /* + AddCollapsableFromOtherBlocks, RegionStateManager.cs:321-324,441-457 */
int64_t orc_run_reads_schedule(const PiscesReadBatch* batch, const uint8_t* ref_bases, int64_t ref_len, int32_t region_start, int32_t region_loci,
                               const PiscesHipConfig* cfg, const int32_t* up_to_positions, int32_t n_up_to, const OrcCandidate* forced,
                               int32_t n_forced, PiscesCalledAllele* out, int64_t capacity, OrcCalled* full_out,
                               int64_t* total_num_called);
/* ... with a ChrIntervalSet (n_intervals inclusive ranges, sorted and disjoint; 0 = none): Reference candidates inside the intervals only
 * (RegionState.GetAllCandidates, RegionState.cs:414-447), a callable allele outside them counted and not reported (AlleleCaller.ShouldReport,
 * AlleleCaller.cs:260-263) */
int64_t orc_run_reads_schedule_intervals(const PiscesReadBatch* batch, const uint8_t* ref_bases, int64_t ref_len, int32_t region_start, int32_t region_loci,
                                         const PiscesHipConfig* cfg, const int32_t* up_to_positions, int32_t n_up_to, const OrcCandidate* forced,
                                         int32_t n_forced, const int32_t* iv_starts, const int32_t* iv_ends, int32_t n_intervals,
                                         PiscesCalledAllele* out, int64_t capacity, OrcCalled* full_out, int64_t* total_num_called);
void orc_set_intervals(OrcState* s, const int32_t* starts, const int32_t* ends, int32_t n);
void orc_schedule_host_candidates(const OrcCandidate* list, int32_t n);   /* IStateManager.AddCandidates beside the reads', for the next orc_run_reads_schedule* (n = 0 clears) */                  /* RegionStateManager.cs:283-334; forced alleles: SmallVariantCaller.cs:49-77,118-150 */
void    orc_diploid_locus_process(OrcCalled* alleles_at_position, int32_t n);             /* DiploidLocusProcessor.cs:13-52 */
void    orc_set_forced_alleles(OrcState* s, const OrcCandidate* list, int32_t n);       /* Factory.cs:56-96,270-286 */
void    orc_add_forced_as_candidates(OrcState* s, int32_t up_to_position);              /* SmallVariantCaller.cs:118-132 */
/* same, from packed observations (position, tuple) instead of reads */
int64_t orc_run_observations(const int32_t* positions, const uint32_t* tuples, int64_t n_obs,
                      const uint8_t* ref_bases, int64_t ref_len, int32_t region_start, int32_t region_loci,
                      const PiscesHipConfig* cfg, PiscesCalledAllele* out, int64_t capacity,
                      int64_t* n_candidate_loci);

/* interval shards, one host thread each (bench.py's threaded CPU baseline); every job has its own output buffer */
typedef struct OrcShardJob {
    const PiscesReadBatch* batch;
    const uint8_t* ref_bases;
    int64_t ref_len;
    int32_t region_start, region_loci;
    const PiscesHipConfig* cfg;
    PiscesCalledAllele* out;
    int64_t capacity;
    int32_t passes;
    int32_t pad;
    int64_t n_out;    /* records of the last pass, or < 0 on error */
    int64_t n_loci;   /* candidate loci summed over the passes */
} OrcShardJob;
int32_t orc_run_reads_sharded(OrcShardJob* jobs, int32_t n_jobs);

void orc_default_config(PiscesHipConfig* cfg);

#ifdef __cplusplus
}
#endif
#endif
