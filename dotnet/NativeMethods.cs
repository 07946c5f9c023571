// NativeMethods.cs — P/Invoke declarations for libpisceship.so (include/pisces_hip.h).
// Mirrors the convention of the only native binding in Pisces, src/lib/Common.IO/FileCompression.cs:10-35
// (cdecl, int return code, pinned blittable arrays).  Source only: this image has no dotnet/mono to compile it.
using System;
using System.Runtime.InteropServices;

namespace Pisces.Hip
{
    [StructLayout(LayoutKind.Sequential)]
    public struct PiscesVcfConfig   // include/pisces_hip.h PiscesVcfConfig; filled from VcfWriterConfig (VcfFileWriter.cs:264-330), -1 = null
    {
        public int VariantQualityFilter, RMxNMaxRepeatLength, RMxNMinRepetitions, NoiseLevel;
        public int OutputStrandBiasAndNoiseLevel, OutputNoCallFraction;
        public float MinFrequencyThreshold, FrequencyFilterThreshold;
        public int Crush;   // !AllowMultipleVcfLinesPerLoci
        public int NoiseLevelFromRecords;   // 1: the NL column is the record's NoiseLevelApplied (records made by this library)
    }

    [StructLayout(LayoutKind.Sequential)]
    public struct PiscesVcfPadState { public int LastVariantPositionWritten, LastPaddedPosition, LastClearedIntervalIndex; }   // start at {0, 0, -1}

    [StructLayout(LayoutKind.Sequential)]
    public struct PiscesHipConfig
    {
        public int AbiVersion, MinBaseCallQuality, NoiseLevel, MaxVariantQscore, MinVariantQscore, VariantQscoreFilter,
            MinCoverage, LowDepthFilter, MinGenotypeQscore, MaxGenotypeQscore, LowGqFilter, StrandBiasModel,
            FilterSingleStrand, IncludeReferenceCalls, EmitZeroCoverageRefs, ExpectStitchedReads, TileLoci, BlockSize;
        public float MinFrequency, VariantFreqFilter, GenotypeMinFreqFilter, TargetLodFrequency, StrandBiasThreshold,
            NoCallFilterThreshold;
        public int RmxnMaxRepeatLength, RmxnMinRepetitions;
        public float RmxnFrequencyLimit;
        public int Collapse;
        public float CollapseFreqThreshold, CollapseFreqRatioThreshold;
        public int CallMnvs, MaxMnvLength, MaxGapBetweenMnv, NoiseModel, Ploidy;
        public float DiploidSnvMinorVF, DiploidSnvMajorVF, DiploidSnvSumVF, DiploidIndelMinorVF, DiploidIndelMajorVF, DiploidIndelSumVF;   // PiscesApplicationOptions.CallMNVs / MaxSizeMNV / MaxGapBetweenMNV, VariantCallingParameters.NoiseModel (0 Flat, 1 Window)
    }

    [StructLayout(LayoutKind.Sequential, Pack = 8, Size = 64)]
    public struct PiscesCalledAllele
    {
        public int Position, TotalCoverage, AlleleSupport, ReferenceSupport, NumNoCalls;
        public int CovF, CovR, CovS, SupF, SupR, SupS;
        public int VariantQscore;
        public double StrandBiasScore;
        public short GenotypeQscore;
        public short NoiseLevel;   // CalledAllele.NoiseLevelApplied; -32768 = int.MinValue (the C# cast of a non-finite PtoQ)
        public ushort FilterBits, Info;
    }

    [StructLayout(LayoutKind.Sequential)]
    public unsafe struct PiscesReadBatch
    {
        public int NReads;
        public int* Position; public byte* Flags; public int* CigarOffset; public byte* CigarOp; public uint* CigarLen;
        public int* SeqOffset; public byte* Bases; public byte* Quals; public byte* Directions;
        public byte* DeletionDirections;   // 2 per CIGAR op: first / last deleted base of a D op in CigarDirections.Expand(), 255 = untracked
    }

    [StructLayout(LayoutKind.Sequential)]
    public struct PiscesGenotypeAllele   // one allele of a locus for pisces_hip_set_genotypes (the germline genotypers as a function)
    {
        public int Category, RefLen, AltLen, Support, Coverage, ReferenceSupport;
        public long AlleleOffset;
        public int Genotype, GenotypeQscore, PhaseSetIndex;
        public byte MultiAllelic, Prune, Pad0, Pad1;
    }

    [StructLayout(LayoutKind.Sequential, Size = 32)]
    public struct PiscesBgzfBlock
    {
        public long InOffset, OutOffset;
        public int InLength, OutLength;
        public uint Crc32;
        public int Reserved;
    }

    [StructLayout(LayoutKind.Sequential, Size = 56)]
    public struct PiscesCandidate
    {
        public int Position, Category, RefLen, AltLen;
        public int SupF, SupR, SupS, AnchoredF, AnchoredR, AnchoredS;
        public byte OpenLeft, OpenRight, Pad0, Pad1;
        public long AlleleOffset;   // ref bytes then alt bytes in the allele pool
    }

    // device-resident surface: tile descriptor and per-tile result directory entry (slot layout, pisces_hip.h)
    [StructLayout(LayoutKind.Sequential, Size = 24)]
    public struct PiscesTile { public int StartPosition, NLoci; public long TupleBegin, TupleEnd; }

    [StructLayout(LayoutKind.Sequential, Size = 48)]
    public unsafe struct PiscesTileResult
    {
        public int RecordBegin, NRecords, NCandidateLoci, NCalled;
        public fixed uint Valid[8];
    }

    internal static unsafe class NativeMethods
    {
        private const string Lib = "pisceship";   // libpisceship.so next to Pisces.dll (like libFileCompression.so)

        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int pisces_hip_abi_version();
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int pisces_hip_default_config(out PiscesHipConfig cfg);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int pisces_hip_create(ref PiscesHipConfig cfg, int device, out IntPtr handle);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int pisces_hip_destroy(IntPtr handle);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern long pisces_hip_trim_memory();
        /// the HIP devices of this process; job j of a -threadbychr run takes device j % count (HipFactory)
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int pisces_hip_device_count();
        /// the positions this handle reports calls and totals for, when it is one interval shard of a chromosome (halo reads beyond it only feed the counts)
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int pisces_hip_set_owned_range(IntPtr handle, int lo, int hi);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern IntPtr pisces_hip_last_error(IntPtr handle);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int pisces_hip_set_reference(IntPtr handle, byte[] upperBases, long length);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int pisces_hip_set_intervals(IntPtr handle, int[] starts, int[] ends, int n);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int pisces_hip_add_reads(IntPtr handle, ref PiscesReadBatch batch);
        /// the same for a batch whose arrays lie in device memory already (every pointer of the struct a device pointer)
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int pisces_hip_add_device_reads(IntPtr handle, ref PiscesReadBatch deviceBatch, long nCigarOps, long nBases);
        /// the arrays of a batch inside the handle's pinned staging buffer: fill them, then pisces_hip_add_reads(views) sends them as they lie
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int pisces_hip_stage_reads(IntPtr handle, int nReads, long nCigarOps, long nBases, int withDirections, int withDeletionDirections, ref PiscesReadBatch views);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int pisces_hip_flush(IntPtr handle, int upToPosition, [Out] PiscesCalledAllele[] output, long capacity, out long nOut);
        /// the flush as a pair: begin enqueues it and commits DoneProcessing, end waits and returns the alleles pisces_hip_flush would have
        /// returned; the next reads may be staged and added in between (the device works on block k while the host marshals block k + 1)
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int pisces_hip_flush_begin(IntPtr handle, int upToPosition);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int pisces_hip_flush_end(IntPtr handle, [Out] PiscesCalledAllele[] output, long capacity, out long nOut);
        /// flush_end with pisces_hip_flush_ex's candidate outputs (what a host that writes VCF rows takes the allele strings from)
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int pisces_hip_flush_end_ex(IntPtr handle, [Out] PiscesCalledAllele[] output, long capacity, out long nOut, [Out] int[] candIndex, [Out] PiscesCandidate[] cands, long candCapacity, out long nCand, [Out] byte[] alleles, long alleleCapacity, out long alleleBytes);
        /// the flushes without the copy: the rows where they lie (pinned memory of the handle), valid until the next flush on the handle
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int pisces_hip_flush_view(IntPtr handle, int upToPosition, out PiscesCalledAllele* rows, out long nRows, out int* candIndex, out PiscesCandidate* cands, out long nCand, out byte* alleles, out long alleleBytes);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int pisces_hip_flush_end_view(IntPtr handle, out PiscesCalledAllele* rows, out long nRows, out int* candIndex, out PiscesCandidate* cands, out long nCand, out byte* alleles, out long alleleBytes);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int pisces_hip_flush_ex(IntPtr handle, int upToPosition, [Out] PiscesCalledAllele[] output, long capacity, out long nOut, [Out] int[] candIndex, [Out] PiscesCandidate[] cands, long candCapacity, out long nCand, [Out] byte[] alleles, long alleleCapacity, out long alleleBytes);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int pisces_hip_get_candidates(IntPtr handle, int upToPosition, [Out] PiscesCandidate[] cands, long capacity, out long nOut, [Out] byte[] alleles, long alleleCapacity, out long alleleBytes);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int pisces_hip_add_candidates(IntPtr handle, PiscesCandidate[] cands, long n, byte[] alleles, long alleleBytes);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int pisces_hip_set_forced_alleles(IntPtr handle, PiscesCandidate[] alleles_of, long n, byte[] alleles, long alleleBytes);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int pisces_hip_set_exact_total_called(IntPtr handle, int on);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int pisces_hip_set_known_variants(IntPtr handle, PiscesCandidate[] variants, long n, byte[] alleles, long alleleBytes);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int pisces_hip_set_exclude_mnvs_from_collapsing(IntPtr handle, int on);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern long pisces_hip_find_indel_candidates(ref PiscesReadBatch batch, byte[] reference, long refLen, int minBaseCallQuality, [Out] PiscesCandidate[] cands, long capacity, [Out] byte[] alleles, long alleleCapacity, out long alleleBytes);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern long pisces_hip_find_candidates(ref PiscesReadBatch batch, byte[] reference, long refLen, int minBaseCallQuality, int snvsAndMnvs, int callMnvs, int maxMnvLength, int maxGapBetweenMnv, [Out] PiscesCandidate[] cands, long capacity, [Out] byte[] alleles, long alleleCapacity, out long alleleBytes);
        [StructLayout(LayoutKind.Sequential)]
        public struct PiscesTileBatch { public IntPtr DTuples, DTiles; public int NTiles, RefStartPosition; public IntPtr DRefBases; public long RefLength; public IntPtr DRecords, DTileResults; public int RecordCapacity, Pad; }
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int pisces_hip_call_tiles_batched(IntPtr handle, PiscesTileBatch[] batches, int nBatches, IntPtr stream);
        // device-resident surface (raw device pointers + hipStream_t as IntPtr)
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int pisces_hip_call_tiles(IntPtr handle, IntPtr dTuples, IntPtr dTiles, int nTiles, IntPtr dRefBases, int refStartPosition, long refLength, IntPtr dRecords, int recordCapacity, IntPtr dTileResults, IntPtr stream);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int pisces_hip_compact_records(IntPtr handle, IntPtr dRecords, IntPtr dTileResults, int nTiles, IntPtr dOffsets, IntPtr dOut, int outCapacity, IntPtr dCount, IntPtr stream);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int pisces_hip_accumulate_tiles(IntPtr handle, IntPtr dTuples, IntPtr dTiles, int nTiles, IntPtr dCounts, IntPtr stream);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int pisces_hip_device_totals(IntPtr handle, [Out] long[] totals4, int reset);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int pisces_hip_synchronize(IntPtr handle);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int pisces_hip_get_stream(IntPtr handle, out IntPtr stream);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int pisces_hip_comm_library([Out] byte[] text, int capacity);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int pisces_hip_comm_ranks(IntPtr handle, out int ranks);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int pisces_hip_get_counts(IntPtr handle, int startPosition, int n, [Out] int[] counts);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int pisces_hip_add_gapped_mnv_ref(IntPtr handle, int[] positions, int[] counts, int n);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int pisces_hip_get_gapped_mnv_ref(IntPtr handle, int position, out int count);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int pisces_hip_get_base_quality_sums(IntPtr handle, int startPosition, int n, [Out] double[] sums);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern long pisces_hip_find_candidates_device(IntPtr handle, ref PiscesReadBatch batch, int snvsAndMnvs, int callMnvs, int maxMnvLength, int maxGapBetweenMnv, [Out] PiscesCandidate[] cands, long capacity, [Out] byte[] alleles, long alleleCapacity, out long alleleBytes);
        // multi-GPU summary (one process per GPU): RCCL is bound by the library at run time
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int pisces_hip_comm_unique_id([Out] byte[] id128, int capacity);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int pisces_hip_comm_init(IntPtr handle, byte[] id128, int rank, int world);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int pisces_hip_reduce_summary(IntPtr handle, [In, Out] long[] totals4);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int pisces_hip_comm_destroy(IntPtr handle);
        /// {allelesCalled, variantsCollapsed, readsProcessed, readsSkipped} (SmallVariantCaller.cs:114-115, AlignmentsSource.cs:63)
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int pisces_hip_stats(IntPtr handle, [Out] long[] stats4);
        /// host seconds inside the library: {add_reads / add_decoded_reads, flushes, of those waiting for the device, flushes counted}
        /// bytes over PCIe: {host -> device reads / file bytes, device -> host records, candidate records, allele counts}
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int pisces_hip_transfer_bytes(IntPtr handle, [Out] long[] out4, int reset);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int pisces_hip_host_time(IntPtr handle, [Out] double[] out4, int reset);

        // the BAM surface: BamReader.GetNextAlignment + AlignmentSource's filters + Read construction for one chromosome's records
        // (src/lib/Alignment.IO/Sequencing/BamReader.cs:287-420, Pisces.Processing/Logic/AlignmentsSource.cs:33-92), on the device, from the
        // compressed file bytes.  counts4 = {records of the chromosome, reads kept, bases, CIGAR operations}.
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int pisces_hip_bam_decode(IntPtr handle, byte[] file, long nBytes, PiscesBgzfBlock[] blocks, long nBlocks, int refId, int minMapQuality, int skipDuplicates, int onlyProperPairs, [Out] long[] counts4);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int pisces_hip_bam_decode(IntPtr handle, IntPtr file, long nBytes, PiscesBgzfBlock[] blocks, long nBlocks, int refId, int minMapQuality, int skipDuplicates, int onlyProperPairs, [Out] long[] counts4);   // (a memory-mapped file)
        /// the decoded batch to the host, array by array (any may be null): for hosts that want the Read objects as well
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int pisces_hip_bam_fetch(IntPtr handle, [Out] int[] position, [Out] byte[] flags, [Out] int[] cigarOffset, [Out] byte[] cigarOp, [Out] uint[] cigarLen, [Out] int[] seqOffset, [Out] byte[] bases, [Out] byte[] quals);
        /// 1: the batch has stitched reads (XD tags) and their per-base / per-deletion directions were copied out; 0: none
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int pisces_hip_bam_fetch_directions(IntPtr handle, [Out] byte[] directions, [Out] byte[] deletionDirections);
        /// IStateManager.AddAlleleCounts + the candidate finder for every read of the decoded batch, without the reads leaving the device
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int pisces_hip_add_decoded_reads(IntPtr handle);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int pisces_hip_bam_chain_mode(IntPtr handle);

        // measurement helpers (bench / diagnostics)
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int pisces_hip_set_timing(IntPtr handle, int everyNth);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int pisces_hip_last_kernel_ms(IntPtr handle, out float ms);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int pisces_hip_kernel_time(IntPtr handle, out double totalMs, out long launches);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int pisces_hip_set_chain_timing(IntPtr handle, int enable);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int pisces_hip_chain_time(IntPtr handle, [Out] double[] outMs);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int pisces_hip_mark(IntPtr handle, int which, IntPtr stream);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int pisces_hip_marked_ms(IntPtr handle, out float ms);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int pisces_hip_balanced_tile_loci(IntPtr handle, long nLoci);
        // a run of call_tiles launches captured once (hipGraph) and replayed
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int pisces_hip_call_tiles_graph_build(IntPtr handle, PiscesTileBatch[] batches, int nBatches, out int graphId);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int pisces_hip_call_tiles_graph_launch(IntPtr handle, int graphId, IntPtr stream);
        // observations a host expanded itself (pisces_hip_expand_reads is the host-side walk of RegionStateManager.AddAlleleCounts)
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int pisces_hip_add_observations(IntPtr handle, int[] positions, uint[] tuples, long n);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern long pisces_hip_expand_reads(ref PiscesReadBatch batch, int minBaseCallQuality, [Out] int[] positions, [Out] uint[] tuples, long capacity);
        // BGZF: the batched counterpart of Common.IO.SafeNativeMethods.UncompressBlock (FileCompression.cs:14-16)
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern long pisces_hip_bgzf_scan(byte[] file, long nBytes, [Out] PiscesBgzfBlock[] blocks, long capacity, out long inflatedBytes);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern long pisces_hip_bgzf_scan(IntPtr file, long nBytes, [Out] PiscesBgzfBlock[] blocks, long capacity, out long inflatedBytes);   // (a memory-mapped file)
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int pisces_hip_bgzf_inflate(IntPtr handle, byte[] file, long nBytes, PiscesBgzfBlock[] blocks, long nBlocks, [Out] byte[] output, long outCapacity, int checkCrc, out float kernelMs);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int pisces_hip_probe_read_bandwidth(IntPtr handle, long nBytes, int reps, out double gbPerSecond);
        // VCF body lines straight from the records (what VcfFileWriter.WriteListOfColocatedAlleles writes per allele, Pisces.IO/VcfFileWriter.cs:206-262)
        /// the host half of IAlleleCaller.Call as functions: MnvReallocator.ReallocateFailedMnvs and the germline genotypers on alleles the caller brings
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int pisces_hip_reallocate_failed_mnvs(PiscesCandidate[] failed, long nFailed, PiscesCandidate[] callable, long nCallable, byte[] alleles, long alleleBytes, int blockMaxPosition, [Out] PiscesCandidate[] callableOut, long callableCapacity, out long nCallableOut, [Out] PiscesCandidate[] outsideOut, long outsideCapacity, out long nOutsideOut, [Out] byte[] allelesOut, long alleleCapacity, out long alleleBytesOut);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int pisces_hip_set_genotypes(ref PiscesHipConfig cfg, [In, Out] PiscesGenotypeAllele[] allelesOfOneLocus, int n, byte[] alleles, long alleleBytes);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int pisces_hip_diploid_genotype_qscore(int genotype, int totalCoverage, int alleleSupport, int minQscore, int maxQscore);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int pisces_hip_vcf_default_config(out PiscesVcfConfig cfg);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern long pisces_hip_format_vcf(ref PiscesVcfConfig cfg, [MarshalAs(UnmanagedType.LPStr)] string chrom, PiscesCalledAllele[] records, long n, int[] candIndex, PiscesCandidate[] cands, byte[] alleles, [Out] byte[] text, long capacity);

        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern long pisces_hip_format_vcf_padded(ref PiscesVcfConfig cfg, [MarshalAs(UnmanagedType.LPStr)] string chrom, PiscesCalledAllele[] records, long n, int[] candIndex, PiscesCandidate[] cands, byte[] alleles, byte[] referenceBases, long refLen, int[] intervalStarts, int[] intervalEnds, int nIntervals, ref PiscesVcfPadState state, int finish, [Out] byte[] text, long capacity);

        public static void Check(IntPtr handle, int rc)
        {
            if (rc == 0) return;
            var msg = Marshal.PtrToStringAnsi(pisces_hip_last_error(handle));
            // surfaces through BaseGenomeProcessor's per-job catch (Pisces.Processing/Logic/BaseGenomeProcessor.cs:121-128)
            if (rc == -1 || rc == -4) throw new ArgumentException(msg);
            throw new Exception("libpisceship error " + rc + ": " + msg);
        }
    }
}
