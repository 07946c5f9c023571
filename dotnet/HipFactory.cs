// HipFactory.cs — the injection point: subclass Pisces' Factory and override the three protected virtual
// Create* hooks (src/exe/Pisces/Logic/Factory.cs:123,128,209), exactly as the reference's own test double
// MockFactoryWithDefaults does (src/test/Pisces.Tests/MockBehaviors/MockFactoryWithDefaults.cs:36-49).
// Program.ProgramExecution (src/exe/Pisces/Program.cs:39) is the single `new Factory(...)` site to switch.
// Source only (no dotnet toolchain in the build image).
using System;
using System.Collections.Generic;
using Pisces.Domain.Interfaces;
using Pisces.Domain.Models;
using Pisces.Domain.Models.Alleles;
using Pisces.Domain.Options;
using Pisces.Interfaces;
using Pisces.Processing.Interfaces;

namespace Pisces.Hip
{
    public class HipFactory : Pisces.Logic.Factory
    {
        private HipEngine _engine;   // one per (BAM, chromosome) job, created in CreateStateManager
        private HashSet<Tuple<string, int, string, string>> _forcedGtAlleles;   // of that job (Factory.cs:260-262 makes the caller first)

        public HipFactory(PiscesApplicationOptions options) : base(options) { }

        private static int _jobsMade = -1;   // across the factory instances of the process (one per BAM, Program.cs:39)
        private static int NextDevice()
        {
            var pinned = Environment.GetEnvironmentVariable("PISCES_HIP_DEVICE");
            int device;
            if (!string.IsNullOrEmpty(pinned) && int.TryParse(pinned, out device)) return device;
            return (int)((uint)System.Threading.Interlocked.Increment(ref _jobsMade) % (uint)HipEngine.DeviceCount());
        }

        // Candidates are found by the library itself, on the device, from the reads handed to pisces_hip_add_reads (find_emit_kernel):
        // insertions / deletions always, SNV / MNV candidates of the M walk when CallMNVs is set (PiscesHipConfig.call_mnvs); with it off SNV
        // candidates are implied by the device counts.  No managed finder runs: a finder that yields nothing keeps SmallVariantCaller's loop unchanged.
        protected override ICandidateVariantFinder CreateVariantFinder() { return new NoCandidates(); }

        // PISCES_HIP_BAM_SURFACE=1: the reads of the job never become Read objects.  The alignment source hands the compressed file bytes to
        // the library (inflate, record cut, AlignmentSource's filters, XD tags, read store, candidate discovery: all on the device) the first
        // time SmallVariantCaller asks for a read, and then reports the end of the file; the loop of SmallVariantCaller.Execute
        // (SmallVariantCaller.cs:88-104) falls through to its final Call(), which flushes everything.  The managed extractor is opened
        // once for what only the BAM header says (stitched / collapsed source, reference order).
        protected override IAlignmentSource CreateAlignmentSource(ChrReference chrReference, string bamFilePath, bool commandLineSaysStitched,
            List<string> chrsToProcess = null)
        {
            if (Environment.GetEnvironmentVariable("PISCES_HIP_BAM_SURFACE") != "1")
                return base.CreateAlignmentSource(chrReference, bamFilePath, commandLineSaysStitched, chrsToProcess);
            bool stitched, collapsed; int refId;
            using (var extractor = new Pisces.IO.BamFileAlignmentExtractor(bamFilePath, commandLineSaysStitched, chrReference.Name))
            {
                stitched = extractor.SourceIsStitched; collapsed = extractor.SourceIsCollapsed;
                refId = extractor.SourceReferenceList.IndexOf(chrReference.Name);
            }
            var f = _options.BamFilterParameters;
            return new HipBamSource(() => _engine, bamFilePath, refId, chrReference, f.MinimumMapQuality, f.RemoveDuplicates, f.OnlyUseProperPairs, stitched, collapsed);
        }

        protected override IStateManager CreateStateManager(ChrIntervalSet intervalSet, bool expectStitchedReads = false,
            bool expectCollapsedReads = true)
        {
            // one handle per (BAM, chromosome) job.  With -threadbychr the jobs of a BAM run on a thread pool (BaseGenomeProcessor.cs:40-90,
            // JobManager.cs:70-73): job j takes device j % count, so that the jobs spread over the GPUs of the node (each handle owns its
            // own HIP stream and buffers; handles share nothing).  PISCES_HIP_DEVICE pins one device (a process per GPU, as bench.py runs).
            _engine = new HipEngine(HipEngine.ConfigFrom(_options, expectStitchedReads, intervalSet != null), NextDevice());
            if (intervalSet != null) _engine.SetIntervals(intervalSet);
            _engine.SetForcedAlleles(_forcedGtAlleles);   // -forcedalleles: ForcedReport rows, reference rows at forced positions
            return new HipStateManager(_engine);
        }

        protected override IAlleleCaller CreateVariantCaller(ChrReference chrReference, ChrIntervalSet intervalSet,
            IAlignmentSource alignmentSource, HashSet<Tuple<string, int, string, string>> forceGtAlleles = null)
        {
            _forcedGtAlleles = forceGtAlleles;
            // the flush as a pair unless PISCES_HIP_SYNC_FLUSH is set: the device calls block k while the managed side stages block k + 1
            return new HipAlleleCaller(() => _engine, chrReference, Environment.GetEnvironmentVariable("PISCES_HIP_SYNC_FLUSH") == null);
        }
    }

    /// IAlignmentSource over the library's BAM surface: no Read ever reaches the managed side (HipFactory.CreateAlignmentSource).
    /// The file is memory-mapped and handed over as it lies (no managed copy, no byte[] size limit; jobs of one process share the page
    /// cache); the library keeps the records of refId.  Still per job: the whole file is inflated and cut to find one chromosome's
    /// records — a production reader hands over the chromosome's chunks from the .bai, in position-ordered slices with a flush between them.
    internal sealed class HipBamSource : IAlignmentSource
    {
        private readonly Func<HipEngine> _engine; private readonly string _path; private readonly int _refId; private readonly ChrReference _chr;
        private readonly int _minMapQuality; private readonly bool _skipDuplicates, _onlyProperPairs;
        private bool _handedOver;
        public HipBamSource(Func<HipEngine> engine, string path, int refId, ChrReference chr, int minMapQuality, bool skipDuplicates, bool onlyProperPairs,
            bool stitched, bool collapsed)
        {
            _engine = engine; _path = path; _refId = refId; _chr = chr; _minMapQuality = minMapQuality; _skipDuplicates = skipDuplicates;
            _onlyProperPairs = onlyProperPairs; SourceIsStitched = stitched; SourceIsCollapsed = collapsed;
        }
        public unsafe Read GetNextRead()
        {
            if (!_handedOver && _refId >= 0)
            {
                _handedOver = true;
                long nBytes = new System.IO.FileInfo(_path).Length;
                using (var map = System.IO.MemoryMappedFiles.MemoryMappedFile.CreateFromFile(_path, System.IO.FileMode.Open, null, 0,
                                                                                             System.IO.MemoryMappedFiles.MemoryMappedFileAccess.Read))
                using (var view = map.CreateViewAccessor(0, 0, System.IO.MemoryMappedFiles.MemoryMappedFileAccess.Read))
                {
                    byte* p = null;
                    view.SafeMemoryMappedViewHandle.AcquirePointer(ref p);
                    try
                    {
                        // {records of the chromosome, reads kept, bases, CIGAR operations}; the skipped reads are in HipEngine.Stats()[3]
                        _engine().AddBamBlocks((IntPtr)(p + view.PointerOffset), nBytes, _refId, _chr, _minMapQuality, _skipDuplicates, _onlyProperPairs);
                    }
                    finally { view.SafeMemoryMappedViewHandle.ReleasePointer(); }
                }
            }
            return null;   // end of file: SmallVariantCaller goes on to its final Call()
        }
        public int? LastClearedPosition { get { return null; } }
        public bool SourceIsStitched { get; private set; }
        public bool SourceIsCollapsed { get; private set; }
    }

    internal sealed class NoCandidates : ICandidateVariantFinder
    {
        private static readonly CandidateAllele[] None = new CandidateAllele[0];
        public IEnumerable<CandidateAllele> FindCandidates(Read read, string refChromosome, string chromosomeName) { return None; }
    }

    /// IStateManager over the native handle: AddAlleleCounts batches reads into a pinned SoA and calls
    /// pisces_hip_add_reads; GetCandidatesToProcess returns a batch token carrying upToPosition;
    /// DoneProcessing is a no-op (the native flush already retired the blocks); GetAlleleCount is served by
    /// pisces_hip_get_counts + the AlleleCountHelper window arithmetic (kept in C#, it is 40 lines of ints).
    public class HipStateManager : IStateManager
    {
        private readonly HipEngine _e;
        public HipStateManager(HipEngine e) { _e = e; }
        public void AddAlleleCounts(Read read) { _e.StageRead(read); }                       // copies out: the Read object is reused (AlignmentsSource.cs:21,61)
        // SmallVariantCaller hands the forced alleles in here itself (AddForcedAlleleAsCandidate); the library also adds them at flush
        // time from pisces_hip_set_forced_alleles, and a second copy without support merges into the first (RegionState.AddCandidate)
        public void AddCandidates(IEnumerable<CandidateAllele> candidates) { _e.AddCandidates(candidates); }
        private int _lastUpToBlockKey = -1;
        public ICandidateBatch GetCandidatesToProcess(int? upToPosition, ChrReference chrReference = null,
            HashSet<Tuple<string, int, string, string>> forcedGtAlleles = null)
        {
            // RegionStateManager.cs:287-291: only make a batch when upTo has moved onto another block (SmallVariantCaller asks after every
            // read); until then the reads just accumulate in the staging arrays and cross PCIe once per block
            if (upToPosition.HasValue && _e.BlockKey(upToPosition.Value) == _lastUpToBlockKey) return null;
            _lastUpToBlockKey = upToPosition.HasValue ? _e.BlockKey(upToPosition.Value) : -1;
            _e.FlushStagedReads(chrReference);
            return new HipBatch(upToPosition);
        }
        public void DoneProcessing(ICandidateBatch batch) { }
        public int GetAlleleCount(int position, Pisces.Domain.Types.AlleleType a, Pisces.Domain.Types.DirectionType d,
            int minAnchor = 0, int? maxAnchor = null, bool fromEnd = false, bool symmetric = false)
        { return _e.GetAlleleCount(position, (int)a, (int)d, minAnchor, maxAnchor, fromEnd, symmetric); }
        public void AddGappedMnvRefCount(Dictionary<int, int> lookup) { _e.AddGappedMnvRefCount(lookup); }
        public int GetGappedMnvRefCount(int position) { return _e.GetGappedMnvRefCount(position); }
        public double GetSumOfAlleleBaseQualities(int position, Pisces.Domain.Types.AlleleType a, Pisces.Domain.Types.DirectionType d,
            int minAnchor = 0, int? maxAnchor = null, bool fromEnd = false, bool symmetric = false)
        { return _e.GetSumOfAlleleBaseQualities(position, (int)a, (int)d, minAnchor, maxAnchor, fromEnd, symmetric); }
        // Read-collapsing (UMI) counts are kept by CollapsedRegionStateManager only, for BAMs made by the read collapser
        // (CollapedRegionStateManager.cs:33); RegionStateManager itself returns 0 (RegionStateManager.cs: the virtual no-op
        // AddCollapsedReadCount), and so does this state manager: Factory.CreateStateManager is asked for the plain one here.
        public int GetCollapsedReadCount(int position, Pisces.Domain.Types.ReadCollapsedType type) { return 0; }
        // Consumers: ExactCoverageCalculator (only with the exact-coverage option, which HipAlleleCaller does not run) and the amplicon-bias
        // calculator (off by default; AmpliconBiasFilterThreshold null).  The native path carries neither, as RegionStateManager
        // carries none without an amplicon-tagged BAM.
        public List<ReadCoverageSummary> GetSpanningReadSummaries(int startPosition, int endPosition) { return new List<ReadCoverageSummary>(); }
        public AmpliconCounts GetCoverageByAmplicon(int position) { return AmpliconCounts.GetEmptyAmpliconCounts(); }
        public bool ExpectStitchedReads { get { return _e.ExpectStitchedReads; } }
    }

    /// IAlleleCaller: Call(batch, source) = the native flush of upTo -> PiscesCalledAllele[] (+ the allele strings of the called
    /// insertions / deletions) -> CalledAllele objects in a SortedList<int, List<CalledAllele>> (already sorted by position, then
    /// ref/alt).  No managed VariantCollapser is passed down: PiscesHipConfig.Collapse = options.Collapse does it natively.
    ///
    /// Pipelined (the default): Call(k) begins the flush of batch k (pisces_hip_flush_begin: the device work is enqueued, DoneProcessing
    /// committed) and returns the alleles of batch k - 1 (pisces_hip_flush_end_ex), so the device works on block k while
    /// SmallVariantCaller's loop reads and stages the reads of block k + 1 (SmallVariantCaller.cs:88-104).  The final Call(null) returns
    /// what is outstanding and its own batch.  The VCF writer sees the same alleles in the same order (batches are disjoint, ascending
    /// runs of positions; SmallVariantCaller.cs:170-178 writes whatever Call returns), one block later.
    public class HipAlleleCaller : IAlleleCaller
    {
        private readonly Func<HipEngine> _engine; private readonly ChrReference _chr; private readonly bool _pipelined;
        private bool _inFlight;
        public HipAlleleCaller(Func<HipEngine> engine, ChrReference chr, bool pipelined = true) { _engine = engine; _chr = chr; _pipelined = pipelined; }
        public int TotalNumCollapsed { get { return (int)_engine().Stats()[1]; } }   // the library collapses insertion / deletion candidates (PiscesHipConfig.Collapse)
        public int TotalNumCalled { get { return (int)_engine().Stats()[0]; } }      // (read after the final Call: nothing is in flight then)
        public SortedList<int, List<CalledAllele>> Call(ICandidateBatch batch, IAlleleSource source)
        {
            var upTo = ((HipBatch)batch).UpToPosition;
            var e = _engine();
            if (!_pipelined) return e.Flush(upTo, _chr);
            var ready = _inFlight ? e.FlushEnd(_chr) : new SortedList<int, List<CalledAllele>>();
            _inFlight = false;
            if (upTo.HasValue) { _inFlight = e.FlushBegin(upTo); return ready; }
            foreach (var kv in e.Flush(null, _chr))              // the last batch: positions above everything returned so far
            {
                List<CalledAllele> at;
                if (ready.TryGetValue(kv.Key, out at)) at.AddRange(kv.Value); else ready.Add(kv.Key, kv.Value);
            }
            return ready;
        }
    }
}
