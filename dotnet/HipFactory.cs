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

        // Candidates are found by the library itself, on the device, from the reads handed to pisces_hip_add_reads (find_emit_kernel):
        // insertions / deletions always, SNV / MNV candidates of the M walk when CallMNVs is set (PiscesHipConfig.call_mnvs); with it off SNV
        // candidates are implied by the device counts.  No managed finder runs: a finder that yields nothing keeps SmallVariantCaller's loop unchanged.
        protected override ICandidateVariantFinder CreateVariantFinder() { return new NoCandidates(); }

        protected override IStateManager CreateStateManager(ChrIntervalSet intervalSet, bool expectStitchedReads = false,
            bool expectCollapsedReads = true)
        {
            _engine = new HipEngine(HipEngine.ConfigFrom(_options, expectStitchedReads, intervalSet != null), device: 0);
            if (intervalSet != null) _engine.SetIntervals(intervalSet);
            _engine.SetForcedAlleles(_forcedGtAlleles);   // -forcedalleles: ForcedReport rows, reference rows at forced positions
            return new HipStateManager(_engine);
        }

        protected override IAlleleCaller CreateVariantCaller(ChrReference chrReference, ChrIntervalSet intervalSet,
            IAlignmentSource alignmentSource, HashSet<Tuple<string, int, string, string>> forceGtAlleles = null)
        {
            _forcedGtAlleles = forceGtAlleles;
            return new HipAlleleCaller(() => _engine, chrReference);
        }
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

    /// IAlleleCaller: Call(batch, source) = pisces_hip_flush_ex(upTo) -> PiscesCalledAllele[] (+ the allele strings of the called
    /// insertions / deletions) -> CalledAllele objects in a SortedList<int, List<CalledAllele>> (already sorted by position, then
    /// ref/alt).  No managed VariantCollapser is passed down: PiscesHipConfig.Collapse = options.Collapse does it natively.
    public class HipAlleleCaller : IAlleleCaller
    {
        private readonly Func<HipEngine> _engine; private readonly ChrReference _chr;
        public HipAlleleCaller(Func<HipEngine> engine, ChrReference chr) { _engine = engine; _chr = chr; }
        public int TotalNumCollapsed { get { return (int)_engine().Stats()[1]; } }   // the library collapses insertion / deletion candidates (PiscesHipConfig.Collapse)
        public int TotalNumCalled { get { return (int)_engine().Stats()[0]; } }
        public SortedList<int, List<CalledAllele>> Call(ICandidateBatch batch, IAlleleSource source)
        { return _engine().Flush(((HipBatch)batch).UpToPosition, _chr); }
    }
}
