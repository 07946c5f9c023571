// HipEngine.cs — the managed half of the drop-in: one native handle per (BAM, chromosome) job, the read staging buffers, the option ->
// PiscesHipConfig mapping and the record -> CalledAllele conversion.  Source only: this image has no dotnet toolchain (INTEGRATION.md).
// It is written against the reference's types (Pisces.Domain / Pisces.Interfaces, v5.2.11) and include/pisces_hip.h (ABI 4).
using System;
using System.Collections.Generic;
using System.Runtime.InteropServices;
using System.Text;
using Pisces.Domain.Models;
using Pisces.Domain.Models.Alleles;
using Pisces.Domain.Options;
using Pisces.Domain.Types;

namespace Pisces.Hip
{
    internal sealed class HipBatch : Pisces.Interfaces.ICandidateBatch
    {
        public readonly int? UpToPosition;
        public HipBatch(int? upTo) { UpToPosition = upTo; }
        public bool HasCandidates { get { return true; } }                       // the native flush decides (block rule of RegionStateManager.cs:283-334)
        public int? MaxClearedPosition { get; set; }
        public List<Region> ClearedRegions { get; set; }
        public List<CandidateAllele> GetCandidates() { return new List<CandidateAllele>(); }
        public void Add(CandidateAllele candidate) { }
        public void Add(IEnumerable<CandidateAllele> candidates) { }
    }

    public sealed class HipEngine : IDisposable
    {
        private IntPtr _h;
        private readonly PiscesHipConfig _cfg;
        private bool _referenceSet;
        // staged reads (SoA, grown geometrically; handed to pisces_hip_add_reads when the caller asks for candidates)
        private readonly List<int> _pos = new List<int>(), _cigOff = new List<int> { 0 }, _seqOff = new List<int> { 0 };
        private readonly List<byte> _flags = new List<byte>(), _cigOp = new List<byte>(), _bases = new List<byte>(), _quals = new List<byte>(), _dirs = new List<byte>(), _delDirs = new List<byte>();
        private readonly List<uint> _cigLen = new List<uint>();
        private bool _anyStitched, _anyDelDirs;
        // BlocksPerFlush > 1 holds the native flush back until upTo has moved that many blocks on: records come out later, in the same
        // order (the VCF writer does not care), and the per-flush latency (~0.26 ms) is paid once per group (DESIGN.md section 8)
        public int BlocksPerFlush = 1;
        private int _lastFlushedBlock = -1;

        public bool ExpectStitchedReads { get { return _cfg.ExpectStitchedReads != 0; } }
        public int BlockKey(int position) { return (position - 1) / _cfg.BlockSize + 1; }   // RegionStateManager.GetBlockKey

        /// Factory.CreateVariantCaller's VariantCallerConfig (Factory.cs:149-179) + the state-manager / finder settings (:123,209-227)
        public static PiscesHipConfig ConfigFrom(PiscesApplicationOptions o, bool expectStitchedReads, bool hasIntervals)
        {
            PiscesHipConfig c;
            NativeMethods.Check(IntPtr.Zero, NativeMethods.pisces_hip_default_config(out c));
            var v = o.VariantCallingParameters;
            c.MinBaseCallQuality = o.BamFilterParameters.MinimumBaseCallQuality;
            c.NoiseLevel = v.NoiseLevelUsedForQScoring;
            c.MaxVariantQscore = v.MaximumVariantQScore; c.MinVariantQscore = v.MinimumVariantQScore; c.VariantQscoreFilter = v.MinimumVariantQScoreFilter;
            c.MinCoverage = v.MinimumCoverage; c.LowDepthFilter = v.LowDepthFilter ?? -1;
            c.MinGenotypeQscore = v.MinimumGenotypeQScore; c.MaxGenotypeQscore = v.MaximumGenotypeQScore; c.LowGqFilter = v.LowGenotypeQualityFilter ?? -1;
            c.StrandBiasModel = v.StrandBiasModel == StrandBiasModel.Poisson ? 0 : v.StrandBiasModel == StrandBiasModel.Extended ? 1 : 2;
            c.StrandBiasThreshold = v.StrandBiasAcceptanceCriteria; c.FilterSingleStrand = v.FilterOutVariantsPresentOnlyOneStrand ? 1 : 0;
            c.IncludeReferenceCalls = o.VcfWritingParameters.OutputGvcfFile ? 1 : 0;
            c.EmitZeroCoverageRefs = hasIntervals ? 1 : 0;                         // RegionState.cs:446
            c.ExpectStitchedReads = expectStitchedReads ? 1 : 0;
            c.NoCallFilterThreshold = v.NoCallFilterThreshold; c.TargetLodFrequency = v.TargetLODFrequency;
            c.RmxnMaxRepeatLength = v.RMxNFilterMaxLengthRepeat ?? -1; c.RmxnMinRepetitions = v.RMxNFilterMinRepetitions ?? -1; c.RmxnFrequencyLimit = v.RMxNFilterFrequencyLimit;
            c.Collapse = o.Collapse ? 1 : 0; c.CollapseFreqThreshold = o.CollapseFreqThreshold; c.CollapseFreqRatioThreshold = o.CollapseFreqRatioThreshold;
            c.CallMnvs = o.CallMNVs ? 1 : 0; c.MaxMnvLength = o.MaxSizeMNV; c.MaxGapBetweenMnv = o.MaxGapBetweenMNV;
            c.NoiseModel = v.NoiseModel == NoiseModel.Window ? 1 : 0;
            c.Ploidy = v.PloidyModel == PloidyModel.DiploidByThresholding ? 1 : v.PloidyModel == PloidyModel.Haploid ? 2 : 0;
            var snv = v.DiploidSNVThresholdingParameters; var indel = v.DiploidINDELThresholdingParameters;
            c.DiploidSnvMinorVF = snv.MinorVF; c.DiploidSnvMajorVF = snv.MajorVF; c.DiploidSnvSumVF = snv.SumVFforMultiAllelicSite;
            c.DiploidIndelMinorVF = indel.MinorVF; c.DiploidIndelMajorVF = indel.MajorVF; c.DiploidIndelSumVF = indel.SumVFforMultiAllelicSite;
            // MinFrequency / VariantFreqFilter come from the genotyper (Factory.cs:160,167; IGenotypeCalculator.MinVarFrequency[Filter])
            float minVarFrequency = c.Ploidy == 0 ? v.MinimumFrequency : snv.MinorVF;
            c.MinFrequency = minVarFrequency;
            c.VariantFreqFilter = Math.Max(v.MinimumFrequencyFilter, minVarFrequency);   // SetMinFreqFilter
            c.GenotypeMinFreqFilter = c.VariantFreqFilter;
            return c;
        }

        public HipEngine(PiscesHipConfig cfg, int device)
        {
            _cfg = cfg;
            var c = cfg;
            NativeMethods.Check(IntPtr.Zero, NativeMethods.pisces_hip_create(ref c, device, out _h));
        }

        public void SetIntervals(ChrIntervalSet set)
        {
            var s = new int[set.Intervals.Count]; var e = new int[set.Intervals.Count];
            for (int i = 0; i < s.Length; i++) { s[i] = set.Intervals[i].StartPosition; e[i] = set.Intervals[i].EndPosition; }
            NativeMethods.Check(_h, NativeMethods.pisces_hip_set_intervals(_h, s, e, s.Length));
        }

        /// IStateManager.AddAlleleCounts: copy the read out (the Read object is reused by the source, AlignmentsSource.cs:21,61)
        public void StageRead(Read read)
        {
            _pos.Add(read.Position);
            _flags.Add((byte)(read.BamAlignment.IsReverseStrand() ? 1 : 0));
            // directions inside deletions (CandidateVariantFinder.GetDeletionDirectionForStitchedRead reads them from the expanded XD map)
            var expanded = read.CigarDirections != null && read.CigarDirections.Directions.Count > 0 ? read.CigarDirections.Expand() : null;
            int e = 0;
            foreach (var op in read.CigarData)
            {
                _cigOp.Add((byte)op.Type); _cigLen.Add(op.Length);
                bool tracked = expanded != null && op.Type == 'D' && op.Length > 0 && e + (int)op.Length <= expanded.Count;
                _delDirs.Add(tracked ? (byte)expanded[e] : (byte)255);
                _delDirs.Add(tracked ? (byte)expanded[e + (int)op.Length - 1] : (byte)255);
                _anyDelDirs |= tracked;
                e += (int)op.Length;
            }
            _cigOff.Add(_cigOp.Count);
            _bases.AddRange(Encoding.ASCII.GetBytes(read.Sequence));
            _quals.AddRange(read.Qualities);
            var map = read.SequencedBaseDirectionMap;                               // per-base DirectionType (stitched reads: XD tag)
            for (int i = 0; i < map.Length; i++) { _dirs.Add((byte)map[i]); _anyStitched |= map[i] == DirectionType.Stitched; }
            _seqOff.Add(_bases.Count);
        }

        /// IStateManager.AddCandidates: the library finds the candidates of the reads itself (NoCandidates is the managed finder), so what
        /// arrives here is what the caller makes on its own: the forced alleles of SmallVariantCaller.AddForcedAlleleAsCandidate
        /// (SmallVariantCaller.cs:118-139).  They go to the state through pisces_hip_add_candidates (ahead of the reads still staged: a
        /// forced allele carries no support, so its place among the candidates of a position decides nothing).
        public void AddCandidates(IEnumerable<CandidateAllele> candidates)
        {
            var list = new List<CandidateAllele>(candidates);
            if (list.Count == 0) return;
            PiscesCandidate[] c; byte[] pool;
            Pack(list.ConvertAll(a => Tuple.Create(a.ReferencePosition, a.ReferenceAllele, a.AlternateAllele)), list, out c, out pool);
            NativeMethods.Check(_h, NativeMethods.pisces_hip_add_candidates(_h, c, c.LongLength, pool, pool.LongLength));
        }

        /// AlleleCaller.AddForcedGtAlleles (Factory.cs:187) + RegionState's reference rows at forced positions: the (chr, pos, ref, alt) set
        /// Factory.SelectForcedAllele picked for this chromosome.  After SetIntervals, before the first flush.
        public void SetForcedAlleles(HashSet<Tuple<string, int, string, string>> forced)
        {
            if (forced == null || forced.Count == 0) return;
            var triples = new List<Tuple<int, string, string>>();
            foreach (var f in forced) triples.Add(Tuple.Create(f.Item2, f.Item3, f.Item4));
            PiscesCandidate[] c; byte[] pool;
            Pack(triples, null, out c, out pool);
            NativeMethods.Check(_h, NativeMethods.pisces_hip_set_forced_alleles(_h, c, c.LongLength, pool, pool.LongLength));
        }

        private static void Pack(List<Tuple<int, string, string>> alleles, List<CandidateAllele> full, out PiscesCandidate[] c, out byte[] pool)
        {
            c = new PiscesCandidate[alleles.Count];
            var bytes = new List<byte>();
            for (int i = 0; i < c.Length; i++)
            {
                c[i].Position = alleles[i].Item1; c[i].RefLen = alleles[i].Item2.Length; c[i].AltLen = alleles[i].Item3.Length;
                c[i].AlleleOffset = bytes.Count;
                bytes.AddRange(Encoding.ASCII.GetBytes(alleles[i].Item2)); bytes.AddRange(Encoding.ASCII.GetBytes(alleles[i].Item3));
                if (full == null) continue;
                var a = full[i];
                c[i].Category = a.Type == AlleleCategory.Snv ? 0 : a.Type == AlleleCategory.Insertion ? 1 : a.Type == AlleleCategory.Deletion ? 2 : 3;   // PISCES_CAT_*
                c[i].SupF = a.SupportByDirection[0]; c[i].SupR = a.SupportByDirection[1]; c[i].SupS = a.SupportByDirection[2];
                c[i].AnchoredF = a.WellAnchoredSupportByDirection[0]; c[i].AnchoredR = a.WellAnchoredSupportByDirection[1]; c[i].AnchoredS = a.WellAnchoredSupportByDirection[2];
                c[i].OpenLeft = (byte)(a.OpenOnLeft ? 1 : 0); c[i].OpenRight = (byte)(a.OpenOnRight ? 1 : 0);
            }
            pool = bytes.ToArray();
        }

        public unsafe void FlushStagedReads(ChrReference chrReference)
        {
            if (!_referenceSet && chrReference != null)
            {
                var bytes = Encoding.ASCII.GetBytes(chrReference.Sequence);         // upper case already (Genome.cs:84-96)
                NativeMethods.Check(_h, NativeMethods.pisces_hip_set_reference(_h, bytes, bytes.LongLength));
                _referenceSet = true;
            }
            if (_pos.Count == 0) return;
            // the staged lists go straight into the library's pinned staging buffer (pisces_hip_stage_reads): one copy, the one a managed
            // host has to make anyway, and pisces_hip_add_reads sends the batch as it lies
            var v = new PiscesReadBatch();
            NativeMethods.Check(_h, NativeMethods.pisces_hip_stage_reads(_h, _pos.Count, _cigOp.Count, _bases.Count, _anyStitched ? 1 : 0, _anyDelDirs ? 1 : 0, ref v));
            for (int i = 0; i < _pos.Count; i++) { v.Position[i] = _pos[i]; v.Flags[i] = _flags[i]; }
            for (int i = 0; i < _cigOff.Count; i++) v.CigarOffset[i] = _cigOff[i];
            for (int i = 0; i < _seqOff.Count; i++) v.SeqOffset[i] = _seqOff[i];
            for (int i = 0; i < _cigOp.Count; i++) { v.CigarOp[i] = _cigOp[i]; v.CigarLen[i] = _cigLen[i]; }
            for (int i = 0; i < _bases.Count; i++) { v.Bases[i] = _bases[i]; v.Quals[i] = _quals[i]; }
            if (_anyStitched) for (int i = 0; i < _dirs.Count; i++) v.Directions[i] = _dirs[i];
            if (_anyDelDirs) for (int i = 0; i < _delDirs.Count; i++) v.DeletionDirections[i] = _delDirs[i];
            NativeMethods.Check(_h, NativeMethods.pisces_hip_add_reads(_h, ref v));
            _pos.Clear(); _flags.Clear(); _cigOp.Clear(); _cigLen.Clear(); _bases.Clear(); _quals.Clear(); _dirs.Clear(); _delDirs.Clear(); _anyDelDirs = false;
            _cigOff.Clear(); _cigOff.Add(0); _seqOff.Clear(); _seqOff.Add(0); _anyStitched = false;
        }

        /// IAlleleCaller.Call: pisces_hip_flush_view — the rows are read where the library left them (for a batch the device called alone:
        /// the pinned buffer its last kernel wrote them to), straight into CalledAllele objects; no intermediate managed arrays
        public unsafe SortedList<int, List<CalledAllele>> Flush(int? upToPosition, ChrReference chr)
        {
            if (HeldBack(upToPosition)) return new SortedList<int, List<CalledAllele>>();
            PiscesCalledAllele* rows; int* idx; PiscesCandidate* cands; byte* pool; long n, nc, nb;
            NativeMethods.Check(_h, NativeMethods.pisces_hip_flush_view(_h, upToPosition ?? -1, out rows, out n, out idx, out cands, out nc, out pool, out nb));
            return ToCalledAlleles(chr, rows, n, idx, cands, pool);
        }

        /// The flush as a pair (pisces_hip_flush_begin / pisces_hip_flush_end_view): FlushBegin enqueues the device work of the batch and
        /// commits DoneProcessing; the host goes on staging the reads of the next block; FlushEnd hands over the alleles.  False when the
        /// flush is held back (BlocksPerFlush) and nothing was begun.
        public bool FlushBegin(int? upToPosition)
        {
            if (HeldBack(upToPosition)) return false;
            NativeMethods.Check(_h, NativeMethods.pisces_hip_flush_begin(_h, upToPosition ?? -1));
            return true;
        }

        public unsafe SortedList<int, List<CalledAllele>> FlushEnd(ChrReference chr)
        {
            PiscesCalledAllele* rows; int* idx; PiscesCandidate* cands; byte* pool; long n, nc, nb;
            NativeMethods.Check(_h, NativeMethods.pisces_hip_flush_end_view(_h, out rows, out n, out idx, out cands, out nc, out pool, out nb));
            return ToCalledAlleles(chr, rows, n, idx, cands, pool);
        }

        private bool HeldBack(int? upToPosition)
        {
            if (!upToPosition.HasValue || BlocksPerFlush <= 1) return false;
            int block = (upToPosition.Value - 1) / _cfg.BlockSize;
            if (block < _lastFlushedBlock + BlocksPerFlush) return true;
            _lastFlushedBlock = block;
            return false;
        }

        /// rows (valid until the next flush on the handle) -> CalledAllele objects; idx == null: no row has a candidate (Reference / SNV rows only)
        private static unsafe SortedList<int, List<CalledAllele>> ToCalledAlleles(ChrReference chr, PiscesCalledAllele* recs, long n, int* idx, PiscesCandidate* cands, byte* pool)
        {
            var result = new SortedList<int, List<CalledAllele>>();
            const string baseOf = "AGCTND";
            for (long i = 0; i < n; i++)
            {
                var r = recs[i];
                var category = (AlleleCategory)((r.Info >> 4) & 7);   // the native codes follow Pisces.Domain.Types.AlleleCategory
                string refAllele, altAllele;
                if (idx != null && idx[i] >= 0)
                {
                    var c = cands[idx[i]];
                    refAllele = Encoding.ASCII.GetString(pool + c.AlleleOffset, c.RefLen);
                    altAllele = Encoding.ASCII.GetString(pool + c.AlleleOffset + c.RefLen, c.AltLen);
                }
                else { refAllele = baseOf[(r.Info >> 7) & 7].ToString(); altAllele = baseOf[(r.Info >> 10) & 7].ToString(); }
                var a = new CalledAllele(category)
                {
                    Chromosome = chr.Name, ReferencePosition = r.Position, ReferenceAllele = refAllele, AlternateAllele = altAllele,
                    TotalCoverage = r.TotalCoverage, AlleleSupport = r.AlleleSupport, ReferenceSupport = r.ReferenceSupport, NumNoCalls = r.NumNoCalls,
                    VariantQscore = r.VariantQscore, GenotypeQscore = r.GenotypeQscore, Genotype = MapGenotype(r.Info & 15),
                    NoiseLevelApplied = r.NoiseLevel == short.MinValue ? int.MinValue : r.NoiseLevel, PhaseSetIndex = (r.FilterBits >> 14) & 3
                };
                a.EstimatedCoverageByDirection = new[] { r.CovF, r.CovR, r.CovS };
                a.SupportByDirection = new[] { r.SupF, r.SupR, r.SupS };
                a.SetFractionNoCalls();
                a.StrandBiasResults.BiasScore = r.StrandBiasScore;
                a.StrandBiasResults.GATKBiasScore = r.AlleleSupport > 0 ? 10 * Math.Log10(r.StrandBiasScore) : 0;   // MathOperations.PtoGATKBiasScale
                a.StrandBiasResults.BiasAcceptable = ((r.Info >> 13) & 1) != 0;
                a.StrandBiasResults.VarPresentOnBothStrands = ((r.Info >> 14) & 1) != 0;
                a.StrandBiasResults.CovPresentOnBothStrands = ((r.Info >> 15) & 1) != 0;
                foreach (var f in FilterOrder) if ((r.FilterBits & (1 << (int)f.Item1)) != 0) a.AddFilter(f.Item2);
                List<CalledAllele> at;
                if (!result.TryGetValue(r.Position, out at)) { at = new List<CalledAllele>(); result.Add(r.Position, at); }
                at.Add(a);                                              // rows arrive sorted by position, then (ref, alt)
            }
            return result;
        }

        // native filter bit -> FilterType, in the order AlleleProcessor / the genotyper / AlleleCaller add them
        private static readonly Tuple<int, FilterType>[] FilterOrder = {
            Tuple.Create(4, FilterType.LowDepth), Tuple.Create(3, FilterType.LowVariantQscore), Tuple.Create(12, FilterType.NoCall),
            Tuple.Create(0, FilterType.StrandBias), Tuple.Create(9, FilterType.RMxN), Tuple.Create(5, FilterType.LowVariantFrequency),
            Tuple.Create(8, FilterType.MultiAllelicSite), Tuple.Create(6, FilterType.LowGenotypeQuality) };

        private static Genotype MapGenotype(int code)   // PISCES_GT_* (include/pisces_hip.h)
        {
            switch (code)
            {
                case 0: return Genotype.HeterozygousAlt1Alt2; case 1: return Genotype.Alt12LikeNoCall; case 2: return Genotype.HeterozygousAltRef;
                case 3: return Genotype.HomozygousAlt; case 4: return Genotype.HomozygousRef; case 5: return Genotype.RefLikeNoCall;
                case 6: return Genotype.AltLikeNoCall; case 7: return Genotype.RefAndNoCall; case 8: return Genotype.AltAndNoCall;
                case 9: return Genotype.HemizygousRef; case 10: return Genotype.HemizygousAlt; default: return Genotype.HemizygousNoCall;
            }
        }

        /// IAlleleSource.GetAlleleCount: the anchor-resolved counts of one position + AlleleCountHelper.GetAnchorAdjustedAlleleCount
        public int GetAlleleCount(int position, int allele, int direction, int minAnchor, int? maxAnchor, bool fromEnd, bool symmetric)
        {
            var flat = new int[6 * 3 * 11];
            NativeMethods.Check(_h, NativeMethods.pisces_hip_get_counts(_h, position, 1, flat));
            var counts = new int[1, 6, 3, 11];                                  // RegionState._alleleCounts layout for one position
            Buffer.BlockCopy(flat, 0, counts, 0, flat.Length * sizeof(int));
            return Pisces.Processing.RegionState.AlleleCountHelper.GetAnchorAdjustedAlleleCount(minAnchor, fromEnd, 5, 11, counts, 0, allele, direction,
                5, maxAnchor, symmetric);                                       // (AlleleCountHelper.cs:21-85, TrackedAnchorSize 5)
        }

        /// IAlleleSource.GetSumOfAlleleBaseQualities: the base-quality sums of one position + AlleleCountHelper.GetAnchorAdjustedTotalQuality
        public double GetSumOfAlleleBaseQualities(int position, int allele, int direction, int minAnchor, int? maxAnchor, bool fromEnd, bool symmetric)
        {
            var flat = new double[6 * 3 * 11];
            NativeMethods.Check(_h, NativeMethods.pisces_hip_get_base_quality_sums(_h, position, 1, flat));
            var sums = new double[1, 6, 3, 11];                                 // RegionState._sumOfAlleleBaseQualities layout for one position
            Buffer.BlockCopy(flat, 0, sums, 0, flat.Length * sizeof(double));
            return Pisces.Processing.RegionState.AlleleCountHelper.GetAnchorAdjustedTotalQuality(minAnchor, fromEnd, 5, 11, sums, 0, allele, direction,
                5, maxAnchor, symmetric);                                       // (AlleleCountHelper.cs:87-166)
        }

        public int GetGappedMnvRefCount(int position)
        {
            int count;
            NativeMethods.Check(_h, NativeMethods.pisces_hip_get_gapped_mnv_ref(_h, position, out count));
            return count;
        }

        /// The per-chromosome totals summed over the interval shards of a multi-GPU job (one process per GPU; SmallVariantCaller.cs:114-115):
        /// InitSummaryReduce once per handle with the id rank 0 made (MakeSummaryReduceId), then ReduceSummary(Stats()).
        public static byte[] MakeSummaryReduceId() { var id = new byte[128]; NativeMethods.Check(IntPtr.Zero, NativeMethods.pisces_hip_comm_unique_id(id, id.Length)); return id; }
        public void InitSummaryReduce(byte[] id, int rank, int world) { NativeMethods.Check(_h, NativeMethods.pisces_hip_comm_init(_h, id, rank, world)); }
        public long[] ReduceSummary(long[] totals4) { NativeMethods.Check(_h, NativeMethods.pisces_hip_reduce_summary(_h, totals4)); return totals4; }

        public void AddGappedMnvRefCount(Dictionary<int, int> lookup)
        {
            var p = new int[lookup.Count]; var c = new int[lookup.Count]; int i = 0;
            foreach (var kv in lookup) { p[i] = kv.Key; c[i++] = kv.Value; }
            NativeMethods.Check(_h, NativeMethods.pisces_hip_add_gapped_mnv_ref(_h, p, c, p.Length));
        }

        /// The BAM surface: one chromosome's records from the compressed file bytes to counts and candidates without the reads leaving
        /// the device (pisces_hip_bgzf_scan -> pisces_hip_bam_decode -> pisces_hip_add_decoded_reads).  What AlignmentSource + BamReader +
        /// the AddAlleleCounts loop do per read (AlignmentsSource.cs:33-92, BamReader.cs:287-420, SmallVariantCaller.cs:88-98) for a whole
        /// batch; returns {records of the chromosome, reads kept, bases, CIGAR operations}.  `file` = a run of whole BGZF blocks (the
        /// blocks of one chromosome from the .bai's chunk list, or the file).
        public long[] AddBamBlocks(byte[] file, int refId, ChrReference chrReference, int minMapQuality, bool skipDuplicates, bool onlyProperPairs)
        {
            if (!_referenceSet && chrReference != null)
            {
                var bytes = Encoding.ASCII.GetBytes(chrReference.Sequence);
                NativeMethods.Check(_h, NativeMethods.pisces_hip_set_reference(_h, bytes, bytes.LongLength));
                _referenceSet = true;
            }
            long inflated;
            long nBlocks = NativeMethods.pisces_hip_bgzf_scan(file, file.LongLength, null, 0, out inflated);   // the count; fills at most `capacity` entries
            if (nBlocks < 0) throw new System.IO.InvalidDataException("not a chain of BGZF blocks (" + nBlocks + ")");   // BamReader.ReadBlock's InvalidDataException
            var blocks = new PiscesBgzfBlock[Math.Max(1, nBlocks)];
            nBlocks = NativeMethods.pisces_hip_bgzf_scan(file, file.LongLength, blocks, blocks.LongLength, out inflated);
            var counts = new long[4];
            NativeMethods.Check(_h, NativeMethods.pisces_hip_bam_decode(_h, file, file.LongLength, blocks, nBlocks, refId, minMapQuality, skipDuplicates ? 1 : 0, onlyProperPairs ? 1 : 0, counts));
            NativeMethods.Check(_h, NativeMethods.pisces_hip_add_decoded_reads(_h));
            return counts;
        }

        /// The same over bytes that are not a managed array: a view of a memory-mapped BAM (HipBamSource) — no copy of the file on the
        /// managed heap, no 2 GB limit of byte[].
        public long[] AddBamBlocks(IntPtr file, long nBytes, int refId, ChrReference chrReference, int minMapQuality, bool skipDuplicates, bool onlyProperPairs)
        {
            if (!_referenceSet && chrReference != null)
            {
                var bytes = Encoding.ASCII.GetBytes(chrReference.Sequence);
                NativeMethods.Check(_h, NativeMethods.pisces_hip_set_reference(_h, bytes, bytes.LongLength));
                _referenceSet = true;
            }
            long inflated;
            long nBlocks = NativeMethods.pisces_hip_bgzf_scan(file, nBytes, null, 0, out inflated);
            if (nBlocks < 0) throw new System.IO.InvalidDataException("not a chain of BGZF blocks (" + nBlocks + ")");
            var blocks = new PiscesBgzfBlock[Math.Max(1, nBlocks)];
            nBlocks = NativeMethods.pisces_hip_bgzf_scan(file, nBytes, blocks, blocks.LongLength, out inflated);
            var counts = new long[4];
            NativeMethods.Check(_h, NativeMethods.pisces_hip_bam_decode(_h, file, nBytes, blocks, nBlocks, refId, minMapQuality, skipDuplicates ? 1 : 0, onlyProperPairs ? 1 : 0, counts));
            NativeMethods.Check(_h, NativeMethods.pisces_hip_add_decoded_reads(_h));
            return counts;
        }

        /// one interval shard of a chromosome: calls and totals only inside [lo, hi] (the reads of the halo still feed the counts)
        public void SetOwnedRange(int lo, int hi) { NativeMethods.Check(_h, NativeMethods.pisces_hip_set_owned_range(_h, lo, hi)); }

        /// host seconds spent inside the library: {adding reads, flushes, of those waiting for the device, number of flushes}
        public double[] HostTime(bool reset = false) { var t = new double[4]; NativeMethods.Check(_h, NativeMethods.pisces_hip_host_time(_h, t, reset ? 1 : 0)); return t; }

        public static int DeviceCount() { int n = NativeMethods.pisces_hip_device_count(); if (n <= 0) throw new Exception("libpisceship: no HIP device (" + n + ")"); return n; }

        /// {allelesCalled, variantsCollapsed, readsProcessed, readsSkipped}
        public long[] Stats() { var s = new long[4]; NativeMethods.Check(_h, NativeMethods.pisces_hip_stats(_h, s)); return s; }

        public void Dispose() { if (_h != IntPtr.Zero) { NativeMethods.pisces_hip_destroy(_h); _h = IntPtr.Zero; } }
    }
}
