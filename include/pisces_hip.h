/*
 * pisces_hip.h — C ABI of libpisceship.so: the MI355X (gfx950) pileup-and-likelihood
 * engine that sits behind Pisces' ICandidateVariantFinder / IStateManager / IAlleleCaller.
 *
 * Conventions (mirror the only P/Invoke precedent in the reference,
 * src/lib/Common.IO/FileCompression.cs:10-35): cdecl, every function returns int32
 * (0 = ok, <0 = error, see PISCES_E_*), plain pointers + sizes, caller-owned buffers that
 * are only read/written for the duration of the call, opaque handle owning all device
 * memory.  No function throws or aborts; pisces_hip_last_error() returns the message the
 * C# shim turns into an exception (caught per job at
 * src/lib/Pisces.Processing/Logic/BaseGenomeProcessor.cs:121-128).
 *
 * Each entry point cites the reference interface member it replaces.
 */
#ifndef PISCES_HIP_H
#define PISCES_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PISCES_HIP_ABI_VERSION 8

/* ---- error codes -------------------------------------------------------- */
#define PISCES_OK                 0
#define PISCES_E_INVALID_ARG     -1  /* reference: ArgumentException (RegionStateManager.cs:363-364, RegionState.cs:315-316) */
#define PISCES_E_BUFFER_TOO_SMALL -2 /* caller grows the output buffer and repeats the call */
#define PISCES_E_DEVICE          -3  /* HIP runtime error, message in last_error */
#define PISCES_E_UNMAPPED_BASE   -4  /* reference: RegionStateManager.cs:109-113 */
#define PISCES_E_UNSUPPORTED     -5
#define PISCES_E_STATE           -6  /* call protocol violated */
#define PISCES_E_INTERNAL        -7  /* an exception inside the library (e.g. out of host memory): caught at the boundary, message in last_error */

/* ---- enums: numeric values are the reference's ------------------------- */
/* src/lib/Pisces.Domain/Types/AlleleType.cs:3-11 */
enum { PISCES_ALLELE_A = 0, PISCES_ALLELE_G = 1, PISCES_ALLELE_C = 2, PISCES_ALLELE_T = 3,
       PISCES_ALLELE_N = 4, PISCES_ALLELE_DEL = 5, PISCES_NUM_ALLELE_TYPES = 6 };
/* src/lib/Pisces.Domain/Types/DirectionType.cs:3-8 */
enum { PISCES_DIR_FORWARD = 0, PISCES_DIR_REVERSE = 1, PISCES_DIR_STITCHED = 2, PISCES_NUM_DIRECTIONS = 3 };
/* src/lib/Pisces.Domain/Types/CallType.cs:3-12 */
enum { PISCES_CAT_SNV = 0, PISCES_CAT_INSERTION = 1, PISCES_CAT_DELETION = 2, PISCES_CAT_MNV = 3,
       PISCES_CAT_REFERENCE = 4 };
/* src/lib/Pisces.Domain/Types/Genotype.cs:3-18 */
enum { PISCES_GT_HET_ALT1_ALT2 = 0, PISCES_GT_ALT12_LIKE_NOCALL = 1, PISCES_GT_HET_ALT_REF = 2,
       PISCES_GT_HOM_ALT = 3, PISCES_GT_HOM_REF = 4, PISCES_GT_REF_LIKE_NOCALL = 5,
       PISCES_GT_ALT_LIKE_NOCALL = 6, PISCES_GT_REF_AND_NOCALL = 7, PISCES_GT_ALT_AND_NOCALL = 8,
       PISCES_GT_HEMI_REF = 9, PISCES_GT_HEMI_ALT = 10, PISCES_GT_HEMI_NOCALL = 11 /* PloidyModel.Haploid */,
       PISCES_GT_OTHERS = 12 /* a forced allele next to other variants, DiploidLocusProcessor.cs:36-38 */ };
/* src/lib/Pisces.Domain/Types/FilterType.cs:3-19 — bit i of filter_bits = enum value i */
enum { PISCES_FILTER_STRAND_BIAS = 0, PISCES_FILTER_POOL_BIAS = 1, PISCES_FILTER_AMPLICON_BIAS = 2,
       PISCES_FILTER_LOW_VARIANT_QSCORE = 3, PISCES_FILTER_LOW_DEPTH = 4,
       PISCES_FILTER_LOW_VARIANT_FREQUENCY = 5, PISCES_FILTER_LOW_GENOTYPE_QUALITY = 6,
       PISCES_FILTER_INDEL_REPEAT_LENGTH = 7, PISCES_FILTER_MULTI_ALLELIC_SITE = 8,
       PISCES_FILTER_RMXN = 9, PISCES_FILTER_FORCED_REPORT = 10, PISCES_FILTER_OFF_TARGET = 11,
       PISCES_FILTER_NO_CALL = 12 };
/* src/lib/Pisces.Domain/Types (StrandBiasModel): Poisson, Extended, Diploid */
enum { PISCES_SB_POISSON = 0, PISCES_SB_EXTENDED = 1, PISCES_SB_DIPLOID = 2 };
enum { PISCES_NOISE_FLAT = 0, PISCES_NOISE_WINDOW = 1 };
enum { PISCES_PLOIDY_SOMATIC = 0, PISCES_PLOIDY_DIPLOID = 1, PISCES_PLOIDY_HAPLOID = 2 };   /* PloidyModel.Somatic / DiploidByThresholding /
                                                                                         Haploid (Types/ModelTypes.cs) */   /* Pisces.Domain/Types/ModelTypes.cs:13 */

/* Anchor bins: NumAnchorIndexes = 2*trackedAnchorSize+1 (RegionStateManager.cs:30-31), default 5 -> 11 */
#define PISCES_ANCHOR_SIZE   5
#define PISCES_NUM_ANCHORS   11
#define PISCES_COUNTS_PER_LOCUS (PISCES_NUM_ALLELE_TYPES * PISCES_NUM_DIRECTIONS * PISCES_NUM_ANCHORS) /* 198 */
#define PISCES_FOLDED_PER_LOCUS (PISCES_NUM_ALLELE_TYPES * PISCES_NUM_DIRECTIONS)                      /* 18  */

/* ---- packed observation tuple (4 bytes) ---------------------
 * Laid out so that the hot kernel's LDS counter address is a mask of the tuple: bits 2..12 are the byte
 * offset of the (allele, direction, column) counter in a [32 rows][64 columns] int32 histogram.
 * bit  0..1   zero
 * bit  2..7   column = PISCES_TUPLE_COLUMN(locus-in-tile, direction): a bank-spreading bijection of the locus
 *             (0..63).  A dwordx4 load hands each lane four consecutive loci of a read, so one LDS instruction sees the
 *             loci c + 4q of a few reads: the column puts q in the low four bank bits and the direction parity in
 *             the fifth, so a forward and a reverse read of the same loci never meet on a bank.
 * bit  8..9   direction 0..2
 * bit 10..12  raw allele code 0..5 (before the min-base-quality test)
 * bit 13..16  anchor bin 0..10  (GetAnchorType, RegionStateManager.cs:83-116)
 * bit 17..23  zero
 * bit 24..31  base quality (deletion tuples carry 255: their quality gate,
 *             CheckDeletionQuality, was applied when the read was expanded)
 * The kernel applies "qual < minBQ -> N" (RegionStateManager.cs:179-181).
 * A tile has at most 64 loci, so every column value is a locus of the tile: no tuple can address memory outside its
 * tile's histogram; allele codes 6, 7 and direction 3 select rows nobody reads. */
#define PISCES_TUPLE_MAX_TILE   64
#define PISCES_TUPLE_COLUMN(locus, dir) \
    (((((uint32_t)(locus) >> 2) & 15u) | (((uint32_t)(locus) & 3u) << 4)) ^ (((uint32_t)(dir) & 1u) << 4))
#define PISCES_TUPLE_PACK(locus, anchor, dir, allele, qual) \
    ((PISCES_TUPLE_COLUMN(locus, dir) << 2) | ((uint32_t)(dir) << 8) | ((uint32_t)(allele) << 10) | \
     ((uint32_t)(anchor) << 13) | ((uint32_t)(qual) << 24))
#define PISCES_TUPLE_DIR(t)    (((t) >> 8) & 0x3u)
#define PISCES_TUPLE_ALLELE(t) (((t) >> 10) & 0x7u)
#define PISCES_TUPLE_ANCHOR(t) (((t) >> 13) & 0xFu)
#define PISCES_TUPLE_QUAL(t)   ((t) >> 24)
#define PISCES_TUPLE_COL(t)    (((t) >> 2) & 63u)
/* inverse of PISCES_TUPLE_COLUMN */
#define PISCES_TUPLE_LOCUS_OF(col, dir) \
    ((((((uint32_t)(col)) ^ (((uint32_t)(dir) & 1u) << 4)) & 15u) << 2) | ((((uint32_t)(col)) ^ (((uint32_t)(dir) & 1u) << 4)) >> 4))
#define PISCES_TUPLE_LOCUS(t)  PISCES_TUPLE_LOCUS_OF(PISCES_TUPLE_COL(t), PISCES_TUPLE_DIR(t))
/* the tuple with another locus (same observation) */
#define PISCES_TUPLE_WITH_LOCUS(t, locus) (((t) & ~0xFCu) | (PISCES_TUPLE_COLUMN(locus, PISCES_TUPLE_DIR(t)) << 2))
/* A tuple with all bits set is padding and is ignored by every kernel. */
#define PISCES_TUPLE_PAD 0xFFFFFFFFu

/* ---- configuration: VariantCallerConfig (src/exe/Pisces/Logic/VariantCalling/AlleleCaller.cs:266-291)
 * + the state-manager/finder settings wired in Factory.cs:123-227.
 * Defaults (pisces_hip_default_config) are the reference's
 * (src/lib/Pisces.Domain/Options/VariantCallingParameters.cs:57-156). */
typedef struct PiscesHipConfig {
    int32_t abi_version;              /* PISCES_HIP_ABI_VERSION */
    int32_t min_base_call_quality;    /* BamFilterParameters.MinimumBaseCallQuality, 20 */
    int32_t noise_level;              /* NoiseLevelUsedForQScoring, = minBQ unless forced */
    int32_t max_variant_qscore;       /* 100 */
    int32_t min_variant_qscore;       /* 20 */
    int32_t variant_qscore_filter;    /* VariantQscoreFilterThreshold, 30; -1 = null */
    int32_t min_coverage;             /* MinCoverage, 10 */
    int32_t low_depth_filter;         /* LowDepthFilter, 10; -1 = null */
    int32_t min_genotype_qscore;      /* 0 */
    int32_t max_genotype_qscore;      /* 100 */
    int32_t low_gq_filter;            /* LowGTqFilter; -1 = null */
    int32_t strand_bias_model;        /* PISCES_SB_EXTENDED */
    int32_t filter_single_strand;     /* FilterSingleStrandVariants, 0 */
    int32_t include_reference_calls;  /* gVCF, 1 */
    int32_t emit_zero_coverage_refs;  /* 1 when an interval set is supplied (RegionState.cs:446) */
    int32_t expect_stitched_reads;    /* IAlleleSource.ExpectStitchedReads */
    int32_t tile_loci;                /* device tile, power of two <= 1024; 0 = default 64 */
    int32_t block_size;               /* GlobalConstants.RegionSize, 1000 */
    float   min_frequency;            /* MinFrequency, 0.01f */
    float   variant_freq_filter;      /* VariantFreqFilter (MinimumFrequencyFilter), 0.01f; <0 = null */
    float   genotype_min_freq_filter; /* SomaticGenotyper._minVariantFrequencyFilter, 0.01f */
    float   target_lod_frequency;     /* TargetLODFrequency, 0.01f */
    float   strand_bias_threshold;    /* StrandBiasFilterThreshold, 0.5f */
    float   no_call_filter_threshold; /* NoCallFilterThreshold, 0.6f; <0 = null */
    int32_t rmxn_max_repeat_length;   /* RMxNFilterMaxLengthRepeat, 5; <0 = filter off */
    int32_t rmxn_min_repetitions;     /* RMxNFilterMinRepetitions, 9 */
    float   rmxn_frequency_limit;     /* RMxNFilterFrequencyLimit, 0.35f */
    int32_t collapse;                 /* PiscesApplicationOptions.Collapse (VariantCollapser on insertion / deletion candidates; SNV twins are
                                         the device counts already); reference default true, see pisces_hip_default_config */
    float   collapse_freq_threshold;        /* CollapseFreqThreshold, 0f */
    float   collapse_freq_ratio_threshold;  /* CollapseFreqRatioThreshold, 0.5f */
    int32_t call_mnvs;                /* PiscesApplicationOptions.CallMNVs, 0: with it on, SNV / MNV candidates come from the read walk
                                         (CandidateVariantFinder.cs:90-232), no longer from the allele counts */
    int32_t max_mnv_length;           /* MaxSizeMNV, 3 */
    int32_t max_gap_between_mnv;      /* MaxGapBetweenMNV, 1 */
    int32_t noise_model;              /* VariantCallingParameters.NoiseModel: PISCES_NOISE_FLAT (default) or PISCES_NOISE_WINDOW, where the
                                         variant q-score of an allele uses (int)PtoQ(SumOfBaseQuality / TotalCoverage) as its noise level
                                         (AlleleCaller.cs:215-218, RegionStateManager.cs:191) */
    int32_t ploidy;                   /* PISCES_PLOIDY_SOMATIC (default), PISCES_PLOIDY_DIPLOID or PISCES_PLOIDY_HAPLOID: one genotype per locus from the
                                         variant frequencies, alleles beyond the ploidy pruned (DiploidThresholdingGenotyper.cs:54-141,
                                         HaploidGenotyper.cs:36-83, which takes MinorVF / MajorVF from the SNV parameters below).  Made on the
                                         device over the tile kernels' record slots (pisces_hip_call_tiles*, and a flush whose rows are
                                         the tile kernels' alone); by the host pass of the flush when rows of insertions / deletions / MNVs
                                         or forced alleles join them.  Both are csrc/genotype_core.h */
    float   diploid_snv_params[3];    /* DiploidSNVThresholdingParameters {MinorVF, MajorVF, SumVFforMultiAllelicSite}: 0.20, 0.70, 0.80 */
    float   diploid_indel_params[3];  /* DiploidINDELThresholdingParameters, same defaults */
} PiscesHipConfig;

/* ---- one called allele (64 bytes; what CalledAllele carries to the VCF writer,
 * src/lib/Pisces.Domain/Models/Alleles/CalledAllele.cs:7-140, VcfFormatter.cs:224) */
typedef struct PiscesCalledAllele {
    int32_t position;            /* ReferencePosition, 1-based */
    int32_t total_coverage;      /* TotalCoverage */
    int32_t allele_support;      /* AlleleSupport */
    int32_t reference_support;   /* ReferenceSupport */
    int32_t num_no_calls;        /* NumNoCalls */
    int32_t coverage_by_dir[3];  /* EstimatedCoverageByDirection */
    int32_t support_by_dir[3];   /* SupportByDirection */
    int32_t variant_qscore;      /* VariantQscore */
    double  strand_bias_score;   /* StrandBiasResults.BiasScore; GATKBiasScore = 10*log10 of it */
    int16_t genotype_qscore;     /* GenotypeQscore */
    int16_t noise_level;         /* NoiseLevelApplied: set where the q-score is computed (VariantQualityCalculator.cs:13), so 0 for an
                                    allele without support; the configured level with NoiseModel.Flat, (int)PtoQ(SumOfBaseQuality /
                                    TotalCoverage) with NoiseModel.Window (AlleleCaller.cs:215-218); -32768 stands for int.MinValue (the
                                    C# cast of a non-finite PtoQ) */
    uint16_t filter_bits;        /* bit i = FilterType i */
    uint16_t info;               /* see PISCES_INFO_* */
} PiscesCalledAllele;

/* info: genotype[0..3] | category[4..6] | ref allele code[7..9] | alt allele code[10..12] |
 *       BiasAcceptable[13] | VarPresentOnBothStrands[14] | CovPresentOnBothStrands[15]
 * ref/alt codes are AlleleType values (single-base alleles).  For host-supplied spanning
 * candidates the strings stay with the caller; alt code then holds PISCES_ALLELE_N. */
#define PISCES_INFO_GENOTYPE(i)  ((i) & 0xF)
#define PISCES_INFO_CATEGORY(i)  (((i) >> 4) & 0x7)
#define PISCES_INFO_REF(i)       (((i) >> 7) & 0x7)
#define PISCES_INFO_ALT(i)       (((i) >> 10) & 0x7)
#define PISCES_INFO_SB_OK(i)     (((i) >> 13) & 1)
#define PISCES_INFO_VAR_BOTH(i)  (((i) >> 14) & 1)
#define PISCES_INFO_COV_BOTH(i)  (((i) >> 15) & 1)
/* filter_bits 14..15: CalledAllele.PhaseSetIndex of a diploid call (0 reference, 1 / 2 the variant alleles in frequency order) */
#define PISCES_FILTERBITS_PHASE(f) (((f) >> 14) & 3)
#define PISCES_INFO_PACK(gt, cat, ref, alt, sbok, varboth, covboth) \
    ((uint16_t)((gt) | ((cat) << 4) | ((ref) << 7) | ((alt) << 10) | ((sbok) << 13) | \
                ((varboth) << 14) | ((covboth) << 15)))

/* ---- tile descriptor: a run of <= tile_loci consecutive reference positions ------------
 * The unit of device work and of interval sharding.  Tiles of one call must
 * not overlap; tuples [tuple_begin, tuple_end) of the tuple buffer belong to this tile. */
typedef struct PiscesTile {
    int32_t start_position;   /* 1-based position of locus 0 */
    int32_t n_loci;           /* 1..tile_loci */
    int64_t tuple_begin;
    int64_t tuple_end;
} PiscesTile;

/* per-tile output directory entry written by the device (48 bytes).
 * The record of (locus l of the tile, allele of alphabetical rank k in A,C,G,T) lives in slot
 * record_begin + 4*l + k of the record buffer; bit (4*l + k) of valid[] says whether that slot holds a called
 * allele.  Walking the valid slots in ascending order yields the tile's alleles sorted by (position, ref, alt);
 * a Reference allele sits at the rank of its base and is not marked valid when a variant is called at its locus. */
typedef struct PiscesTileResult {
    int32_t record_begin;     /* 256 * tile index */
    int32_t n_records;        /* number of valid slots */
    int32_t n_candidate_loci; /* positions with >= 1 called allele */
    int32_t n_called;         /* alleles for which IsCallable was true (IAlleleCaller.TotalNumCalled) */
    uint32_t valid[8];        /* 256 slot-validity bits */
} PiscesTileResult;

/* ---- a batch of reads, structure-of-arrays (what the C# shim pins per call) ------------
 * replaces the Read object walked by FindCandidates/AddAlleleCounts
 * (src/lib/Pisces.Domain/Models/Read.cs:74-96, 535-562). Reads must already have passed
 * AlignmentSource.ShouldSkipRead (src/exe/Pisces/Logic/Alignment/AlignmentsSource.cs:84-92). */
typedef struct PiscesReadBatch {
    int32_t        n_reads;
    const int32_t* position;      /* [n_reads] Read.Position (1-based) */
    const uint8_t* flags;         /* [n_reads] bit0 = reverse strand */
    const int32_t* cigar_offset;  /* [n_reads+1] into cigar_op / cigar_len */
    const uint8_t* cigar_op;      /* 'M','I','D','S','N','=','X','H','P' */
    const uint32_t* cigar_len;
    const int32_t* seq_offset;    /* [n_reads+1] into bases / quals / directions */
    const uint8_t* bases;         /* upper-case ASCII */
    const uint8_t* quals;         /* phred */
    const uint8_t* directions;    /* optional per-base DirectionType (stitched reads, XD tag); NULL = from flags */
    const uint8_t* deletion_directions; /* optional, 2 bytes per CIGAR op (indexed like cigar_op): for a 'D' op the DirectionType of
                                     * its first and of its last deleted base in Read.CigarDirections.Expand() (the XD tag covers
                                     * deleted bases too) - what GetDeletionDirectionForStitchedRead reads,
                                     * CandidateVariantFinder.cs:417-420, 468-487; 255 = this read tracks no directions inside
                                     * deletions (CigarDirections == null, :422-428); ignored for other ops.  NULL = 255 throughout */
} PiscesReadBatch;
#define PISCES_DIR_UNTRACKED 255

/* ---- a candidate allele crossing the boundary (CandidateAllele.cs:8-49) ---- */
typedef struct PiscesCandidate {
    int32_t position;
    int32_t category;             /* PISCES_CAT_* */
    int32_t ref_len, alt_len;
    int32_t support_by_dir[3];
    int32_t well_anchored_by_dir[3];
    uint8_t open_left, open_right;
    uint8_t pad[2];
    int64_t allele_offset;        /* ref bytes then alt bytes at this offset of the allele byte pool */
} PiscesCandidate;

typedef struct PiscesHip PiscesHip;

/* ---- lifecycle ----------------------------------------------------------- */
/* fills *cfg with the reference defaults */
int32_t pisces_hip_default_config(PiscesHipConfig* cfg);
/* Factory.CreateStateManager / CreateVariantCaller / CreateVariantFinder (Factory.cs:123,128,209):
 * one handle per (BAM, chromosome) job; owns one HIP stream; no global mutable state. */
int32_t pisces_hip_create(const PiscesHipConfig* cfg, int32_t device, PiscesHip** out);
/* The number of HIP devices this process sees (what `device` above ranges over), or PISCES_E_DEVICE.  With -threadbychr the reference
 * runs its (BAM, chromosome) jobs on a thread pool (src/lib/Pisces.Processing/Logic/BaseGenomeProcessor.cs:40-90, JobManager.cs:70-73):
 * a host gives job j the device j % count, handles on different devices share nothing. */
int32_t pisces_hip_device_count(void);
int32_t pisces_hip_destroy(PiscesHip* h);
/* Device and pinned host memory of destroyed handles is kept for the handles that follow (a job per chromosome, or per interval range,
 * makes and destroys one each: BaseGenomeProcessor.cs:40-90): up to PISCES_HIP_ALLOC_CACHE_MB of device memory per device (default
 * 8192; 0 = keep nothing) and a quarter of that pinned.  This gives it all back; returns the bytes released. */
int64_t pisces_hip_trim_memory(void);
const char* pisces_hip_last_error(const PiscesHip* h);   /* h may be NULL: message of the last failed create */
int32_t pisces_hip_abi_version(void);

/* ChrReference.Sequence (upper-cased whole chromosome, src/lib/Pisces.IO/Genome.cs:84-96).
 * bases[i] is position i+1. Copied to the device. */
int32_t pisces_hip_set_reference(PiscesHip* h, const uint8_t* upper_bases, int64_t length);

/* ChrIntervalSet after SortAndCollapse (src/lib/Pisces.Domain/Models/IntervalSet.cs:38-74): sorted,
 * disjoint, inclusive [start, end].  With intervals, calls are made only inside them
 * (AlleleCaller.ShouldReport, AlleleCaller.cs:260-263; RegionState.GetAllCandidates clips the
 * reference candidates, RegionState.cs:406-408); set emit_zero_coverage_refs = 1 alongside. n = 0 clears. */
int32_t pisces_hip_set_intervals(PiscesHip* h, const int32_t* starts, const int32_t* ends, int32_t n);

/* Interval sharding: the positions [lo, hi] this handle OWNS.  A shard is fed the reads that overlap its range plus
 * a halo, so that counts and spanning alleles at its edges are complete; candidates the halo reads bring that lie outside the range
 * belong to the neighbouring shard: they are neither reported nor counted in IAlleleCaller.TotalNumCalled here, so that the shards'
 * totals add up to the unsharded job's (the reference has no such notion: one job per chromosome, BaseGenomeProcessor.cs:40-90).
 * Default: everything.  Give the shard's own intervals to pisces_hip_set_intervals as well. */
int32_t pisces_hip_set_owned_range(PiscesHip* h, int32_t lo, int32_t hi);

/* ---- streaming surface: IStateManager -------------------------------------- */
/* ICandidateVariantFinder.FindCandidates + IStateManager.AddCandidates + AddAlleleCounts
 * (SmallVariantCaller.cs:88-98) for a batch of reads.  The batch crosses PCIe once, packed (2 bytes per base), and the
 * read walk of AddAlleleCounts (RegionStateManager.cs:118-220) runs on the device into an observation log in HBM; the
 * host only reads the CIGARs: argument checks, the blocks a read touches, and the insertion / deletion candidates
 * (needs set_reference first; without a reference only the AddAlleleCounts half runs).  The whole batch is checked
 * before any of it is committed. */
int32_t pisces_hip_add_reads(PiscesHip* h, const PiscesReadBatch* batch);
/* The arrays of a batch of n_reads reads (n_cigar_ops CIGAR operations, n_bases bases in all) inside the handle's pinned staging
 * buffer: `views` receives pointers the caller may WRITE through (the const of PiscesReadBatch's members is for add_reads).  A host
 * that marshals its reads anyway (the C# shim packs Read objects into arrays) fills them -- cigar_offset[n_reads] and
 * seq_offset[n_reads] included -- and hands `views` to pisces_hip_add_reads, which then sends the batch as it lies instead of
 * copying it into that buffer first (the copy is the larger part of pisces_hip_add_reads for a large batch).  The views are valid
 * until the next call on the handle that is not pisces_hip_add_reads(views). */
int32_t pisces_hip_stage_reads(PiscesHip* h, int32_t n_reads, int64_t n_cigar_ops, int64_t n_bases, int32_t with_directions,
                               int32_t with_deletion_directions, PiscesReadBatch* views);
/* pisces_hip_add_reads for a batch that lies in DEVICE memory of the handle's device already: every pointer of `device_batch` is a device
 * pointer (the struct itself is host memory), n_cigar_ops = cigar_offset[n_reads] and n_bases = seq_offset[n_reads] (which the host
 * cannot read).  What an upstream stage on the device hands over — a decoder, an aligner, the bench's generator — and the form in which
 * the reads -> records rate of the device is measured (inputs resident in HBM when the timed region starts).  The arrays must
 * be complete when the call is made (synchronise the stream that made them) and are copied: the caller may reuse them when the call
 * returns.  The checks pisces_hip_add_reads makes in a host pass over the CIGARs (Read.ValidateCigar, Read.cs:603-605; position > 0,
 * RegionStateManager.cs:363-364; the blocks a read touches, :361-383) run on the device (read_prepare_kernel); same refusals, same
 * messages, the state unchanged when a read is refused. */
int32_t pisces_hip_add_device_reads(PiscesHip* h, const PiscesReadBatch* device_batch, int64_t n_cigar_ops, int64_t n_bases);
/* Pre-expanded observations for the block grid: positions[i] is the 1-based locus of tuples[i]
 * (the tuple's locus field is ignored). */
int32_t pisces_hip_add_observations(PiscesHip* h, const int32_t* positions, const uint32_t* tuples, int64_t n);
/* IStateManager.GetCandidatesToProcess(upTo) + IAlleleCaller.Call + DoneProcessing
 * (SmallVariantCaller.cs:157-189).  up_to_position < 0 = final flush (Call(null)).
 * Writes called alleles sorted by (position, ref, alt) into out[0..capacity); *n_out = number
 * produced.  Returns PISCES_E_BUFFER_TOO_SMALL (and *n_out = required) without consuming the
 * batch when capacity is insufficient: grow and repeat.  The batch is made at that point (what AlleleCaller.Call hands back to the state
 * — MNV leftovers, candidates that did not collapse — has been handed back) and waits for the repeated call: until then every entry that
 * would change the state (add_reads, add_observations, add_candidates, add_decoded_reads, add_gapped_mnv_ref, a flush with another
 * up_to_position) returns PISCES_E_STATE. */
int32_t pisces_hip_flush(PiscesHip* h, int32_t up_to_position, PiscesCalledAllele* out,
                         int64_t capacity, int64_t* n_out);
/* The same call, also returning the allele strings of the called insertions / deletions:
 * cand_index_out[i] (capacity entries, may be NULL) is -1 for Reference / SNV rows (their alleles are the
 * info base codes) or the index into cand_out of row i's candidate; alleles_out receives ref then alt bytes of each
 * candidate at cand_out[j].allele_offset. PISCES_E_BUFFER_TOO_SMALL if any buffer is short (counts reported). */
int32_t pisces_hip_flush_ex(PiscesHip* h, int32_t up_to_position, PiscesCalledAllele* out, int64_t capacity, int64_t* n_out,
                            int32_t* cand_index_out, PiscesCandidate* cand_out, int64_t cand_capacity, int64_t* n_cand,
                            uint8_t* alleles_out, int64_t allele_capacity, int64_t* allele_bytes);
/* The flush as a pair, for a host that wants the device to work on block k while it prepares block k + 1:
 * pisces_hip_flush_begin(upTo) does what pisces_hip_flush does up to the point where it would wait for the device and commits
 * DoneProcessing (the flushed blocks are gone, the observation log is the compacted one); pisces_hip_flush_end waits for what is
 * still in flight and returns the same alleles pisces_hip_flush would have returned (PISCES_E_BUFFER_TOO_SMALL with *n_out = the
 * count needed: repeat flush_end with a larger buffer, nothing is lost).  Between the two the caller may stage and add the next
 * reads (pisces_hip_stage_reads, pisces_hip_add_reads, pisces_hip_add_observations, pisces_hip_bam_decode +
 * pisces_hip_add_decoded_reads) and read counts; pisces_hip_flush[_ex] and another pisces_hip_flush_begin return PISCES_E_STATE
 * until flush_end has been called.  A batch that needs the host between its device passes (insertion / deletion / MNV candidates,
 * forced alleles, the diploid / haploid genotypers, NoiseModel.Window, gapped-MNV reference counts) is flushed synchronously inside
 * flush_begin; flush_end returns it all the same (pisces_hip_flush_end_ex: with the candidates' allele strings). */
int32_t pisces_hip_flush_begin(PiscesHip* h, int32_t up_to_position);
int32_t pisces_hip_flush_end(PiscesHip* h, PiscesCalledAllele* out, int64_t capacity, int64_t* n_out);
/* pisces_hip_flush_end with pisces_hip_flush_ex's candidate outputs (same meaning, same PISCES_E_BUFFER_TOO_SMALL protocol: all three
 * counts are reported, repeat with larger buffers).  A flush that needed no host-side candidates returns cand_index -1 everywhere. */
int32_t pisces_hip_flush_end_ex(PiscesHip* h, PiscesCalledAllele* out, int64_t capacity, int64_t* n_out, int32_t* cand_index_out,
                                PiscesCandidate* cand_out, int64_t cand_capacity, int64_t* n_cand, uint8_t* alleles_out,
                                int64_t allele_capacity, int64_t* allele_bytes);
/* The flushes without the copy into the caller's array, for a host that reads the rows where they lie (a managed host marshals them into
 * its own objects anyway; 2.8 M rows are 179 MB): *rows points at *n_rows called alleles in memory of the handle — for a batch the device
 * called alone, the pinned host buffer the last kernel wrote them to — valid until the next pisces_hip_flush* / pisces_hip_flush_begin on the
 * handle.  cand_index / cands / alleles (any may be NULL) as pisces_hip_flush_ex returns them; *cand_index is NULL when no row has a
 * candidate (every row is a Reference or SNV row).  No PISCES_E_BUFFER_TOO_SMALL: nothing is copied. */
int32_t pisces_hip_flush_view(PiscesHip* h, int32_t up_to_position, const PiscesCalledAllele** rows, int64_t* n_rows, const int32_t** cand_index,
                              const PiscesCandidate** cands, int64_t* n_cand, const uint8_t** alleles, int64_t* allele_bytes);
int32_t pisces_hip_flush_end_view(PiscesHip* h, const PiscesCalledAllele** rows, int64_t* n_rows, const int32_t** cand_index,
                                  const PiscesCandidate** cands, int64_t* n_cand, const uint8_t** alleles, int64_t* allele_bytes);
/* IAlleleSource.GetAlleleCount for a run of positions: out[n][6][3][11] int32
 * (RegionState.cs:57); blocks never touched read as zero (RegionStateManager.cs:222-226). */
int32_t pisces_hip_get_counts(PiscesHip* h, int32_t start_position, int32_t n, int32_t* out);
/* IAlleleSource.GetSumOfAlleleBaseQualities (src/lib/Pisces.Domain/Interfaces/IAlleleSource.cs:16; RegionState.cs:61,233-239): the
 * base-quality sums double[n][6][3][11] of [start_position, start_position + n), same layout and rules as pisces_hip_get_counts (the
 * caller applies the anchor window, AlleleCountHelper.GetAnchorAdjustedTotalQuality).  Served by any handle, not only NoiseModel.Window.
 * The device accumulates in fixed point: each cell is the true sum rounded once, identical from run to run. */
int32_t pisces_hip_get_base_quality_sums(PiscesHip* h, int32_t start_position, int32_t n, double* out);
/* IAlleleSource.GetGappedMnvRefCount (IAlleleSource.cs:19): what pisces_hip_add_gapped_mnv_ref registered for the position, else 0 */
int32_t pisces_hip_get_gapped_mnv_ref(PiscesHip* h, int32_t position, int32_t* count);
/* IAlleleSource.AddGappedMnvRefCount (RegionStateManager.cs:74-81) */
int32_t pisces_hip_add_gapped_mnv_ref(PiscesHip* h, const int32_t* positions, const int32_t* counts, int32_t n);
/* host-side candidates (insertion / deletion) found so far with position <= up_to (< 0 = all)
 * (IStateManager.GetCandidatesToProcess for the host collapser). alleles = byte pool; out may be NULL to count. */
int32_t pisces_hip_get_candidates(PiscesHip* h, int32_t up_to_position, PiscesCandidate* out,
                                  int64_t capacity, int64_t* n_out, uint8_t* alleles,
                                  int64_t allele_capacity, int64_t* allele_bytes);
/* IStateManager.AddCandidates (src/lib/Pisces.Domain/Interfaces/IStateManager.cs; RegionStateManager.cs:83-116) for candidates the caller
 * brings itself: merged with the candidates the library finds in the reads by RegionState.AddCandidate's rules (RegionState.cs:94-174).
 * cands[i].allele_offset / ref_len / alt_len index `alleles` (ref bytes then alt bytes), as pisces_hip_get_candidates writes them. */
int32_t pisces_hip_add_candidates(PiscesHip* h, const PiscesCandidate* cands, int64_t n, const uint8_t* alleles, int64_t allele_bytes);
/* Forced genotyping (-forcedalleles; Factory.GetForcedAlleles :56-96 + SelectForcedAllele :270-286, src/exe/Pisces/Logic/Factory.cs): the
 * alleles of this chromosome that are reported whatever the reads say.  Only position and the allele strings are read (the category
 * is SmallVariantCaller.GetAlleleCategory's, SmallVariantCaller.cs:141-150); alleles equal to the reference, with an ALT outside
 * A/C/G/T or outside the intervals of pisces_hip_set_intervals are dropped as the reference drops them.  From then on every flush first
 * adds the forced alleles up to upTo as candidates without support (AddForcedAlleleAsCandidate :118-132), reports a forced allele that
 * is not callable with PISCES_FILTER_FORCED_REPORT, genotype 0/1 and genotype q-score 0 (AlleleCaller.cs:98-131, 143-170), keeps the
 * Reference row beside it, and when include_reference_calls is off makes Reference rows at the forced positions
 * (RegionState.GetAllCandidates :393-450).  With the diploid model DiploidLocusProcessor's rules apply (PISCES_GT_OTHERS).  Call it
 * after pisces_hip_set_intervals and before the first flush. */
int32_t pisces_hip_set_forced_alleles(PiscesHip* h, const PiscesCandidate* alleles_of, int64_t n, const uint8_t* alleles, int64_t allele_bytes);
/* IAlleleCaller.TotalNumCalled with an interval set and MNV calling off: AlleleCaller.Call counts in IsCallable, BEFORE ShouldReport
 * (AlleleCaller.cs:109-131), so the reference's total includes the callable SNVs of loci OUTSIDE the intervals, which it does not report.
 * By default the library evaluates the intervals' loci only (the total then counts the callable alleles inside them; every reported row
 * is the reference's either way).  on != 0: every flush also runs its kernel over the off-interval loci of the flushed blocks, drops the
 * records and adds their callable alleles — the reference's number, at the price of a second launch per flush.  With MNV calling on (or collapser
 * thresholds that keep an open-ended SNV and its twin apart) the SNVs are candidates and the candidate path counts the callable ones outside the intervals itself (the switch changes
 * nothing).  The counting launch is the read store's flush kernel: a flush of a configuration that kernel does not serve (NoiseModel.Window,
 * the Diploid strand-bias model, a gapped-MNV reference count in the flushed blocks, the observation-log read path, minimum base quality
 * above 127) fails with PISCES_E_UNSUPPORTED while the switch is on instead of reporting the smaller total. */
int32_t pisces_hip_set_exact_total_called(PiscesHip* h, int32_t on);
/* The chromosome's known (prior) variants: what Factory.cs:204 hands VariantCollapser (the priors file's insertions and MNVs, Factory.cs:378-395).
 * With the collapser on, a candidate that equals one (position, reference allele, alternate allele, type) is anchored on both sides and
 * preferred among the potential matches of an open-ended candidate (VariantCollapser.cs:16-24, 178-190, 216-218).  Same arguments as
 * pisces_hip_set_forced_alleles (position, ref_len, alt_len, allele_offset of every entry; the rest is ignored); n = 0 clears.  Any time
 * before the flush that should see them. */
int32_t pisces_hip_set_known_variants(PiscesHip* h, const PiscesCandidate* variants, int64_t n, const uint8_t* alleles, int64_t allele_bytes);
/* PiscesApplicationOptions.ExcludeMNVsFromCollapsing (Options/PiscesApplicationOptions.cs:62; Factory.cs:204 hands it to VariantCollapser):
 * on != 0: MNV candidates are left out of the collapser's targets (VariantCollapser.cs:33) — an open-ended MNV is not collapsed, and no
 * open-ended SNV / MNV collapses INTO an MNV.  Off by default, as the option.  Any time before the flush that should see it. */
int32_t pisces_hip_set_exclude_mnvs_from_collapsing(PiscesHip* h, int32_t on);
/* totals lines: {allelesCalled (IAlleleCaller.TotalNumCalled), variantsCollapsed, readsProcessed, readsSkipped}
 * (SmallVariantCaller.cs:114-115; readsSkipped = AlignmentSource's count of the reads ShouldSkipRead dropped, AlignmentsSource.cs:63,84-92:
 * the reads pisces_hip_bam_decode dropped from the batches that pisces_hip_add_decoded_reads added; reads a host hands over through
 * pisces_hip_add_reads have passed that filter already).  This is the vector pisces_hip_reduce_summary adds up over the interval shards. */
int32_t pisces_hip_stats(PiscesHip* h, int64_t out[4]);

/* Where the host's time went inside the streaming surface since the handle was made (or since the last call with reset != 0), seconds:
 * {pisces_hip_add_reads / pisces_hip_add_decoded_reads, pisces_hip_flush / _flush_ex / _flush_begin / _flush_end in all, of that waiting
 * for the device (stream / event synchronisation), number of flushes that made a batch}.  (flush - waiting) / flushes is the serial host
 * work per flush: block bookkeeping, candidate merge, VariantCollapser, MnvReallocator, the diploid genotyper, record assembly
 * (what runs on one core of the host next to the device; IAlleleCaller.Call's host half, AlleleCaller.cs:60-141). */
int32_t pisces_hip_host_time(PiscesHip* h, double out[4], int32_t reset);
/* Bytes that crossed PCIe for this handle since the last reset: {host -> device: read batches as handed to pisces_hip_add_reads, or the
 * compressed file bytes and block table of pisces_hip_bam_decode; device -> host: called-allele records; device -> host: the candidate
 * records of the device finder (64 B per read event + long ALT alleles); device -> host: allele counts for the collapser / reallocator}. */
int32_t pisces_hip_transfer_bytes(PiscesHip* h, int64_t out[4], int32_t reset);

/* ---- multi-GPU: the per-chromosome summary across interval shards --------------------------------------------------
 * Loci shard by genomic interval, one process (or one handle) per GPU, no data-path exchange; the only collective is the sum of
 * the int64[4] totals above over the shards (the reference concatenates per-chromosome jobs and prints their totals,
 * src/lib/Pisces.Processing/Logic/BaseGenomeProcessor.cs:40-90, SmallVariantCaller.cs:114-115).  RCCL over xGMI, bound at run time
 * (librccl is loaded when comm_init is first called, so a single-GPU host needs no RCCL):
 *   rank 0:      pisces_hip_comm_unique_id(id)            -> hands the 128 bytes to the other processes (file, pipe, MPI, ...)
 *   every rank:  pisces_hip_comm_init(h, id, rank, world) -> ncclCommInitRank on the handle's device
 *   every rank:  pisces_hip_reduce_summary(h, totals)     -> one ncclAllReduce(sum) of int64[4] on the handle's stream, in place
 * Without a communicator (never initialised, or world == 1) reduce_summary leaves `inout` as it is.  Handles of ONE process on
 * several GPUs can simply add their pisces_hip_stats on the host. */
#define PISCES_COMM_ID_BYTES 128
int32_t pisces_hip_comm_unique_id(uint8_t* id_out, int32_t capacity);
int32_t pisces_hip_comm_init(PiscesHip* h, const uint8_t* id, int32_t rank, int32_t world);
int32_t pisces_hip_reduce_summary(PiscesHip* h, int64_t inout[4]);
int32_t pisces_hip_comm_destroy(PiscesHip* h);
/* Which librccl the library binds, and how it found it — in this order, the first that applies deciding: (1) the file PISCES_HIP_RCCL_PATH
 * names (an error naming it if it cannot be bound); (2) a librccl.so the process has mapped already (a PyTorch host brings its own under
 * torch/lib: binding THAT copy keeps one RCCL per process); (3) librccl.so.1 / librccl.so on the loader's path, then /opt/rocm/lib.
 * Writes "<source>: <path>" (source = PISCES_HIP_RCCL_PATH | mapped | default) and returns its length, or PISCES_E_DEVICE. */
int32_t pisces_hip_comm_library(char* out, int32_t capacity);
/* ncclCommCount of the handle's communicator (1 without one). */
int32_t pisces_hip_comm_ranks(PiscesHip* h, int32_t* ranks);

/* ---- device-resident surface (bench / multi-GPU shards) --------------------
 * All d_* pointers are device pointers on the handle's device; stream is a hipStream_t
 * (NULL = the handle's own stream).  One launch: per tile, observation tuples -> LDS
 * allele-count histogram -> coverage, Poisson q-score, strand bias, somatic genotype and
 * filters for the reference allele and every SNV candidate -> 64-byte records.
 * d_ref_bases[i] is the reference base of position ref_start_position+i.
 * d_records needs record_capacity >= 256 * n_tiles slots (PiscesTileResult explains the slot layout: no
 * allocation atomics, placement independent of scheduling); d_tile_results[n_tiles] is the directory.
 * Asynchronous; launches of one handle must be stream-ordered with respect to each other. */
/* The tile size (<= 64 loci) at which a launch over n_loci contiguous loci loads every CU of the handle's device equally: a launch
 * that is resident at once ends with the CU that holds the most tiles (100 000 loci: 56, i.e. 1786 tiles = 7 per CU, instead of 1563
 * tiles of 64 = 6 or 7 per CU).  64 for launches of many rounds.  A hint for whoever buckets tuples by tile. */
int32_t pisces_hip_balanced_tile_loci(PiscesHip* h, int64_t n_loci);
int32_t pisces_hip_call_tiles(PiscesHip* h, const uint32_t* d_tuples, const PiscesTile* d_tiles,
                              int32_t n_tiles, const uint8_t* d_ref_bases, int32_t ref_start_position,
                              int64_t ref_length, PiscesCalledAllele* d_records, int32_t record_capacity,
                              PiscesTileResult* d_tile_results, void* stream);
/* Several independent batches at once: the launches are spread over the handle's own HIP streams ("lanes"), so that the call phase
 * that ends one launch runs under the streaming phase of the next (at BASELINE config 2 a step takes ~37 us this way instead of ~52 us
 * one after the other).  Every batch in flight needs its own d_records / d_tile_results.  Ordering is on the host: the call first
 * waits for `stream` (NULL = nothing to wait for: the inputs are ready), returns once everything is enqueued, and the outputs are
 * complete after pisces_hip_synchronize (a lane that has waited on another stream's event runs its later kernels ~5 us slower on
 * this runtime, so no event fork / join).  The process should run with GPU_MAX_HW_QUEUES >= 8 so that each lane owns a hardware
 * queue. */
typedef struct PiscesTileBatch {
    const uint32_t*     d_tuples;
    const PiscesTile*   d_tiles;
    int32_t             n_tiles;
    int32_t             ref_start_position;
    const uint8_t*      d_ref_bases;
    int64_t             ref_length;
    PiscesCalledAllele* d_records;
    PiscesTileResult*   d_tile_results;
    int32_t             record_capacity;
    int32_t             pad;
} PiscesTileBatch;
int32_t pisces_hip_call_tiles_batched(PiscesHip* h, const PiscesTileBatch* batches, int32_t n_batches, void* stream);

/* The same launches as ONE HIP graph: pisces_hip_call_tiles_graph_build captures the n_batches launches (in order, on one stream: what
 * pisces_hip_call_tiles n_batches times does) into a graph the handle keeps and returns its id; pisces_hip_call_tiles_graph_launch replays
 * it on `stream` (NULL = the handle's) with one submission — the launches then follow each other at the device's own pace instead of the
 * host's (a launch call per step leaves ~10 us between 38 us kernels at BASELINE config 2).  With pisces_hip_set_timing(n) in force when
 * the graph is built, every n-th launch is bracketed by event records inside the graph, read back by pisces_hip_kernel_time after a
 * replay.  The buffers named by `batches` must stay where they are while the graph is in use; graphs go with the handle. */
int32_t pisces_hip_call_tiles_graph_build(PiscesHip* h, const PiscesTileBatch* batches, int32_t n_batches, int32_t* graph_id);
int32_t pisces_hip_call_tiles_graph_launch(PiscesHip* h, int32_t graph_id, void* stream);

/* Ordered compaction of a call_tiles result: d_out[0 .. *d_count) receives every called allele of the launch
 * sorted by (position, ref, alt) (tiles must be in ascending position order); d_offsets[n_tiles] (int32 scratch)
 * receives each tile's first index in d_out. Asynchronous. */
int32_t pisces_hip_compact_records(PiscesHip* h, const PiscesCalledAllele* d_records,
                                   const PiscesTileResult* d_tile_results, int32_t n_tiles, int32_t* d_offsets,
                                   PiscesCalledAllele* d_out, int32_t out_capacity, int32_t* d_count, void* stream);
/* tuples -> anchor-resolved counts added into d_counts[n_tiles*tile_loci][6][3][11]
 * (the IAlleleSource view for host-side collapsing / spanning coverage). Asynchronous. */
int32_t pisces_hip_accumulate_tiles(PiscesHip* h, const uint32_t* d_tuples, const PiscesTile* d_tiles,
                                    int32_t n_tiles, int32_t* d_counts, void* stream);
/* Device-side running totals bumped by every call_tiles launch of this handle:
 * {records written, candidate loci, IsCallable==true alleles (IAlleleCaller.TotalNumCalled), tiles}.
 * Synchronizes the handle's stream; reset != 0 zeroes them afterwards.  This is the per-chromosome
 * summary a multi-GPU job all-reduces (SmallVariantCaller.cs:114-115 totals line). */
int32_t pisces_hip_device_totals(PiscesHip* h, int64_t out[4], int32_t reset);
/* Kernel timing with HIP events bound to the kernel's own dispatch on the launch stream (start and stop events of
 * hipExtLaunchKernel).  Off by default.
 * enable = n > 0 starts a fresh measurement window in which every n-th call_tiles / accumulate_tiles launch is timed
 * (up to 4096 timed launches are kept); enable = 0 ends it. */
int32_t pisces_hip_set_timing(PiscesHip* h, int32_t enable);
/* A span on the launch stream: pisces_hip_mark(h, 0, stream) before the first of a run of launches, pisces_hip_mark(h, 1, stream) behind the
 * last (HIP events on `stream`, NULL = the handle's own stream: the stream the launches go to); pisces_hip_marked_ms waits for the
 * second mark and returns the time between the two.  Two event records for any number of launches: what bench.py brackets its timed
 * region with (events of every dispatch, pisces_hip_set_timing, put 5-10 us between launches). */
int32_t pisces_hip_mark(PiscesHip* h, int32_t which, void* stream);
int32_t pisces_hip_marked_ms(PiscesHip* h, float* ms);
/* Sum of the timed kernel durations (ms) and number of timed launches since set_timing(n). Waits for them. */
int32_t pisces_hip_kernel_time(PiscesHip* h, double* total_ms, int64_t* launches);
/* The device time of the streaming surface's chain reads -> records (bench.py's roofline_chain).  enable != 0: every
 * pisces_hip_add_device_reads records an event on the handle's stream before the first thing it enqueues and one behind the last, every
 * flush one in front of its first kernel and one behind the kernel that leaves the compacted records in HBM (the transfer to the host
 * follows that event).  pisces_hip_chain_time waits for the last add's and the last flush's events: out_ms[0] = the add's span
 * (IStateManager.AddAlleleCounts + ICandidateVariantFinder.FindCandidates of the batch, RegionStateManager.cs:118-220,
 * CandidateVariantFinder.cs:36-83), out_ms[1] = the flush's (GetCandidatesToProcess + IAlleleCaller.Call); PISCES_E_STATE when either has
 * not happened since timing was switched on. */
int32_t pisces_hip_set_chain_timing(PiscesHip* h, int32_t enable);
int32_t pisces_hip_chain_time(PiscesHip* h, double out_ms[2]);
/* Streaming-read bandwidth of the handle's device, GB/s: best of `reps` timed passes of a kernel that only reads `nbytes`
 * with the hot kernel's load pattern.  Reported beside the spec peak in bench.py; allocates and frees nbytes. */
int32_t pisces_hip_probe_read_bandwidth(PiscesHip* h, int64_t nbytes, int32_t reps, double* gb_per_s);
/* waits for the handle's stream */
int32_t pisces_hip_synchronize(PiscesHip* h);
/* The handle's own stream (a hipStream_t), i.e. what a NULL `stream` argument of the calls above stands for.  It is created
 * hipStreamNonBlocking: it does NOT order against HIP's null stream.  A host whose runtime fills or allocates-and-zeroes device
 * buffers on the null stream (PyTorch's default stream is the null stream) must therefore either enqueue that work on this stream
 * (torch.cuda.ExternalStream over the value returned here) or wait for it before handing the buffers to the library; passing its
 * own null-stream handle (0) as `stream` selects THIS stream, not the null stream, and orders nothing. */
int32_t pisces_hip_get_stream(PiscesHip* h, void** stream);
/* duration of the most recent timed launch of the current window, in milliseconds (PISCES_E_STATE when timing is off) */
int32_t pisces_hip_last_kernel_ms(PiscesHip* h, float* ms);

/* ---- host-side helpers (pure CPU, no device needed) -------------------------- */
/* Expands reads to (position, tuple) observations exactly as AddAlleleCounts walks a read
 * (RegionStateManager.cs:118-220).  Returns the number written or PISCES_E_BUFFER_TOO_SMALL. */
int64_t pisces_hip_expand_reads(const PiscesReadBatch* batch, int32_t min_base_call_quality,
                                int32_t* positions, uint32_t* tuples, int64_t capacity);

/* ICandidateVariantFinder.FindCandidates restricted to insertions and deletions
 * (CandidateVariantFinder.cs:234-292, Create :334-345, Annotate :496-553), one entry per read event, unmerged,
 * in read order. ref[i] is position i+1. Returns the count or PISCES_E_BUFFER_TOO_SMALL. */
int64_t pisces_hip_find_indel_candidates(const PiscesReadBatch* batch, const uint8_t* ref, int64_t ref_len,
                                         int32_t min_base_call_quality, PiscesCandidate* out, int64_t capacity,
                                         uint8_t* alleles, int64_t allele_capacity, int64_t* allele_bytes);
/* The whole of CandidateVariantFinder.FindCandidates (CandidateVariantFinder.cs:36-387,496-553): with snvs_and_mnvs the M operations are
 * walked too (ExtractSnvsFromOperation :90-232; call_mnvs / max_mnv_length / max_gap_between_mnv are ShouldBuildUpMNV's :170-181).
 * Same outputs as above.  Host form of the walk (no handle, no device); pisces_hip_add_reads runs the device form below. */
int64_t pisces_hip_find_candidates(const PiscesReadBatch* batch, const uint8_t* ref, int64_t ref_len, int32_t min_base_call_quality,
                                   int32_t snvs_and_mnvs, int32_t call_mnvs, int32_t max_mnv_length, int32_t max_gap_between_mnv,
                                   PiscesCandidate* out, int64_t capacity, uint8_t* alleles, int64_t allele_capacity, int64_t* allele_bytes);
/* The same walk on the handle's device (find_count_kernel / find_emit_kernel: one lane per read), against the reference given to
 * pisces_hip_set_reference and with the handle's minimum base quality: the candidate discovery pisces_hip_add_reads enqueues for
 * every batch, here with its records returned per read event, unmerged, in read order — what the parity tests compare with
 * the reference's own finder cases.  Same outputs and return values as pisces_hip_find_candidates. */
int64_t pisces_hip_find_candidates_device(PiscesHip* h, const PiscesReadBatch* batch, int32_t snvs_and_mnvs, int32_t call_mnvs,
                                          int32_t max_mnv_length, int32_t max_gap_between_mnv, PiscesCandidate* out, int64_t capacity,
                                          uint8_t* alleles, int64_t allele_capacity, int64_t* allele_bytes);

/* ---- the host half of IAlleleCaller.Call as functions of their own (pure CPU) ------------------------------------------------------
 * What pisces_hip_flush runs on the host between its device passes, callable on alleles the caller brings — the reference's own unit
 * tests drive MnvReallocator and the genotypers this way (MNVReallocatorTests.cs, GenotypeCalculatorTest.cs), and so do this repository's
 * (tests/test_product_goldens.py: the reference's known answers through the PRODUCT's code, not through the oracle's restatement).
 *
 * MnvReallocator.ReallocateFailedMnvs(failedMnvs, callableAlleles, blockMaxPos) (src/exe/Pisces/Logic/VariantCalling/MnvReallocator.cs:12-98):
 * the support of MNVs that are not callable goes to the callable alleles inside them (longest first, then most support), what is left
 * becomes smaller MNVs / SNVs, and — with block_max_position >= 0 — what lies past the block goes to `outside`.  An allele's
 * AlleleSupport is the sum of support_by_dir (the product keeps no second number).  Candidates in, candidates out (same pool layout as
 * pisces_hip_get_candidates); returns PISCES_E_BUFFER_TOO_SMALL with the three counts set when an output is short. */
int32_t pisces_hip_reallocate_failed_mnvs(const PiscesCandidate* failed, int64_t n_failed, const PiscesCandidate* callable, int64_t n_callable,
                                          const uint8_t* alleles, int64_t allele_bytes, int32_t block_max_position,
                                          PiscesCandidate* callable_out, int64_t callable_capacity, int64_t* n_callable_out,
                                          PiscesCandidate* outside_out, int64_t outside_capacity, int64_t* n_outside_out,
                                          uint8_t* alleles_out, int64_t allele_capacity, int64_t* allele_bytes_out);
/* The per-locus genotypers of the germline modes over the alleles of ONE locus (Reference row gone when a variant is there, rows in
 * (ref, alt) order — what pisces_hip_flush hands them): DiploidThresholdingGenotyper.SetGenotypes
 * (src/lib/Pisces.Genotyping/Thresholding/DiploidThresholdingGenotyper.cs:54-141) with cfg->ploidy == PISCES_PLOIDY_DIPLOID,
 * HaploidGenotyper.SetGenotypes (Haploid/HaploidGenotyper.cs:36-83) with PISCES_PLOIDY_HAPLOID; thresholds, minimum depth (min_coverage)
 * and the q-score range come from cfg.  Fills the result fields of every allele; returns the locus' genotype (PISCES_GT_*) or < 0. */
typedef struct PiscesGenotypeAllele {
    int32_t category;                 /* PISCES_CAT_* */
    int32_t ref_len, alt_len;
    int32_t support, coverage, reference_support;
    int64_t allele_offset;            /* ref bytes then alt bytes in the allele pool */
    int32_t genotype, genotype_qscore, phase_set_index;   /* results */
    uint8_t multi_allelic, prune, pad[2];                 /* results: FilterType.MultiAllelicSite; the genotyper drops the allele */
} PiscesGenotypeAllele;
int32_t pisces_hip_set_genotypes(const PiscesHipConfig* cfg, PiscesGenotypeAllele* alleles_of_one_locus, int32_t n, const uint8_t* alleles,
                                 int64_t allele_bytes);
/* DiploidGenotypeQualityCalculator.Compute (Thresholding/DiploidGenotypeQualityCalculator.cs:17-103) for one allele */
int32_t pisces_hip_diploid_genotype_qscore(int32_t genotype, int32_t total_coverage, int32_t allele_support, int32_t min_qscore, int32_t max_qscore);

/* ---- VCF body lines (pure CPU) ---------------------------------------
 * What the writer needs of VcfWriterConfig (src/lib/Pisces.IO/VcfFileWriter.cs:264-330). */
typedef struct PiscesVcfConfig {
    int32_t variant_quality_filter;             /* VariantQualityFilterThreshold (names the q<N> filter); -1 = null */
    int32_t rmxn_max_repeat_length;             /* RMxNFilterMaxLengthRepeat, names R<M>x<N>; -1 = null */
    int32_t rmxn_min_repetitions;               /* RMxNFilterMinRepetitions */
    int32_t noise_level;                        /* EstimatedBaseCallQuality = NoiseLevelUsedForQScoring (the NL field) */
    int32_t output_strand_bias_and_noise_level; /* ShouldOutputStrandBiasAndNoiseLevel */
    int32_t output_no_call_fraction;            /* ShouldOutputNoCallFraction (-reportnocalls) */
    float   min_frequency_threshold;            /* MinFrequencyThreshold: sets the number of VF decimals */
    float   frequency_filter_threshold;         /* FrequencyFilterThreshold; < 0 = null */
    int32_t crush;                              /* !AllowMultipleVcfLinesPerLoci (-crushvcf): co-located alleles share one line */
    int32_t noise_level_from_records;           /* 1: the NL column is the record's noise_level (CalledAllele.NoiseLevelApplied: records made by
                                                   this library, exact with NoiseModel.Window too); 0: noise_level above for every allele
                                                   with support (records assembled elsewhere) */
} PiscesVcfConfig;
int32_t pisces_hip_vcf_default_config(PiscesVcfConfig* cfg);
/* One VCF body line per record, or with cfg->crush one per position (VcfFileWriter.WriteListOfColocatedAlleles, VcfFileWriter.cs:206-262
 * with VcfFormatter.cs:52-495): CHROM POS . REF ALT QUAL FILTER DP=<n> GT:GQ:AD:DP:VF[:NL:SB][:NC] <sample>.  cand_index / cands /
 * alleles are what pisces_hip_flush_ex returned (may be NULL when no row is an insertion / deletion / MNV).  Returns the number of
 * bytes of text; when that exceeds `capacity` nothing is written and the call is repeated with a larger buffer; < 0 = error. */
int64_t pisces_hip_format_vcf(const PiscesVcfConfig* cfg, const char* chrom, const PiscesCalledAllele* recs, int64_t n,
                              const int32_t* cand_index, const PiscesCandidate* cands, const uint8_t* alleles, char* out,
                              int64_t capacity);
/* The same with RegionMapper's padding (src/lib/Pisces.IO/RegionMapper.cs:31-84, VcfFileWriter.PadIfNeeded / WriteRemaining :124-172):
 * positions of the interval set that no allele covers get a no-call row (./., LowDP, DP=0, NL = cfg->noise_level) before the next
 * written position, and with `finish` up to the end of the last interval.  `state` carries the writer's and the mapper's cursors from
 * one call to the next (zero-initialise, then {0, 0, -1}: see PiscesVcfPadState); it is only advanced when the text fitted. */
typedef struct PiscesVcfPadState {
    int32_t last_variant_position_written;   /* VcfFileWriter._lastVariantPositionWritten, starts at 0 */
    int32_t last_padded_position;            /* RegionMapper._lastPaddedPosition, starts at 0 */
    int32_t last_cleared_interval_index;     /* RegionMapper._lastClearedIntervalIndex, starts at -1 */
} PiscesVcfPadState;
int64_t pisces_hip_format_vcf_padded(const PiscesVcfConfig* cfg, const char* chrom, const PiscesCalledAllele* recs, int64_t n,
                                     const int32_t* cand_index, const PiscesCandidate* cands, const uint8_t* alleles,
                                     const uint8_t* ref_bases, int64_t ref_len, const int32_t* interval_starts,
                                     const int32_t* interval_ends, int32_t n_intervals, PiscesVcfPadState* state, int32_t finish,
                                     char* out, int64_t capacity);

/* ---- BGZF inflate on the device (the stage upstream of the read batch) ------------------------------------
 * A BAM file is a chain of BGZF blocks, gzip members of <= 64 KiB whose 'BC' extra subfield holds the block size; each is an
 * independent DEFLATE stream.  Replaces BamReader.ReadBlock's per-block call into the native zlib binding
 * (src/lib/Alignment.IO/BamReader.cs:603-645 -> SafeNativeMethods.UncompressBlock, src/lib/Common.IO/FileCompression.cs:14-16):
 * the host walks the block headers (pisces_hip_bgzf_scan), the device inflates every block of the table at once, one wave per block. */
typedef struct PiscesBgzfBlock {
    int64_t  in_offset;    /* first byte of the DEFLATE payload in the file bytes */
    int64_t  out_offset;   /* where the block's bytes go in the inflated stream (running sum of ISIZE) */
    int32_t  in_length;    /* payload bytes (block size - header - 8 trailer bytes) */
    int32_t  out_length;   /* ISIZE */
    uint32_t crc32;        /* CRC-32 of the inflated bytes, from the trailer */
    int32_t  reserved;
} PiscesBgzfBlock;
/* Host: the block table of file[0, n_bytes).  Returns the number of blocks (the empty end-of-file block included); fills at most
 * `capacity` of them; *inflated_bytes = sum of ISIZE.  PISCES_E_INVALID_ARG: not a BGZF block chain (bad magic, no BC subfield, a
 * block running past the end - what BamReader.ReadBlock throws InvalidDataException for). */
int64_t pisces_hip_bgzf_scan(const uint8_t* file, int64_t n_bytes, PiscesBgzfBlock* blocks, int64_t capacity, int64_t* inflated_bytes);
/* Device: inflates blocks[0, n_blocks) of the file bytes into out (host memory, out_capacity >= the blocks' last out_offset +
 * out_length).  check_crc: the trailer CRC-32s are verified on the host afterwards.  PISCES_E_INVALID_ARG with the index of the first
 * bad block in the message when a stream is corrupt (UncompressBlock < 0 -> ReadBlock returns -1).  kernel_ms (optional): the inflate
 * kernel's own duration. */
int32_t pisces_hip_bgzf_inflate(PiscesHip* h, const uint8_t* file, int64_t n_bytes, const PiscesBgzfBlock* blocks, int64_t n_blocks,
                                uint8_t* out, int64_t out_capacity, int32_t check_crc, float* kernel_ms);

/* ---- BAM records cut on the device (the rest of row f4): the compressed file is the only thing that crosses PCIe --------------------
 * pisces_hip_bam_decode: the BGZF blocks of `file` are inflated into HBM (as above, without the copy back), the BAM record chain is cut
 * there (no serial pass: every byte offset of the first 4 KiB of a 32 KiB chunk is tried as a record start and its chain followed
 * to where it leaves the chunk), AlignmentSource.ShouldSkipRead (src/exe/Pisces/Logic/Alignment/AlignmentsSource.cs:84-92: unmapped, secondary,
 * optionally improper pairs and duplicates, MAPQ below the minimum, no CIGAR) drops what the reference drops, and the alignments of
 * reference sequence `ref_id` are decoded into the PiscesReadBatch arrays in device memory, in file order — what BamReader.GetNextAlignment
 * (src/lib/Alignment.IO/BamReader.cs:137) + Read's constructor do per record on the host.  counts = {reads kept, reads of the chromosome
 * skipped, CIGAR operations, bases}.  Records longer than 32 KiB (long reads) are reported as a broken chain.
 * Where the record chain enters each chunk is first taken, for all chunks at once, from the one exit that the live chains of the chunk
 * before share, and every link checked; a file where that fails somewhere (records of several KiB, ...) gets the serial hop chunk to
 * chunk instead.  pisces_hip_bam_chain_mode says which it was for the last decode (0 = all at once, 1 = serial hop): diagnostics
 * (the environment variable PISCES_HIP_BAM_SERIAL_CHAIN=1 forces the serial hop).
 * pisces_hip_bam_fetch: the decoded batch to host arrays sized from `counts` (any pointer may be NULL) — for inspection and tests.
 * pisces_hip_add_decoded_reads: pisces_hip_add_reads for the decoded batch without anything of it leaving the device: what
 * pisces_hip_add_reads takes from a host pass over the CIGARs (log slots, candidate-record slots, the blocks every read touches, the
 * reads it refuses: position <= 0, a CIGAR longer than the read, a read past 2^31 - 1 or far past the end of its reference sequence)
 * the decode has made where the reads are (the blocks as a bit map that came back with `counts`); the call enqueues the read walk and
 * the candidate discovery and returns without waiting. */
int32_t pisces_hip_bam_decode(PiscesHip* h, const uint8_t* file, int64_t n_bytes, const PiscesBgzfBlock* blocks, int64_t n_blocks, int32_t ref_id,
                              int32_t min_map_quality, int32_t skip_duplicates, int32_t only_proper_pairs, int64_t counts[4]);
int32_t pisces_hip_bam_fetch(PiscesHip* h, int32_t* position, uint8_t* flags, int32_t* cigar_offset, uint8_t* cigar_op, uint32_t* cigar_len,
                             int32_t* seq_offset, uint8_t* bases, uint8_t* quals);
int32_t pisces_hip_add_decoded_reads(PiscesHip* h);
int32_t pisces_hip_bam_chain_mode(PiscesHip* h);
/* Stitched reads (the Stitcher's XD tag, one DirectionType per base of the expanded CIGAR; Read.CigarDirections / SequencedBaseDirectionMap,
 * src/lib/Pisces.Domain/Models/Read.cs:340-400, 664-682): when a record of the decoded chromosome carries the tag, the decode makes
 * PiscesReadBatch.directions (per base; reads without the tag: their strand's direction) and .deletion_directions (two per CIGAR
 * operation) for the batch, and pisces_hip_add_decoded_reads counts and discovers candidates with them.  This entry copies them to the
 * host (either pointer may be NULL): returns 1 when the batch has them, 0 when no read of it was stitched, < 0 on error.  A malformed tag
 * (CigarDirection's "Unexpected format in direction string") makes pisces_hip_add_decoded_reads refuse the batch. */
int32_t pisces_hip_bam_fetch_directions(PiscesHip* h, uint8_t* directions, uint8_t* deletion_directions);

#ifdef __cplusplus
}
#endif
#endif /* PISCES_HIP_H */
