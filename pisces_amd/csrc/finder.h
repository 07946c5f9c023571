// finder.h — ICandidateVariantFinder.FindCandidates (src/lib/Pisces.Domain/Logic/CandidateVariantFinder.cs:36-387,496-553) as host
// calls; the walk itself is finder_walk.h (one source for the host and the device).  With MNV calling off (the reference default) SNV
// candidates are implied by the device allele counts and only insertions and deletions are discovered (M operations are not
// walked); with it on, SNV and MNV candidates come from the M walk too, because bases absorbed into an MNV are no SNV candidates.
#pragma once
#include <stdint.h>

#include <string>
#include <vector>

#include "expander.h"

namespace pisces {

struct HostCandidate {
    int32_t position = 0;     // coordinate of the anchor base (one before the inserted / deleted bases)
    int32_t category = 0;     // PISCES_CAT_*
    std::string ref, alt;
    int32_t support_by_dir[3] = {0, 0, 0};
    int32_t well_anchored_by_dir[3] = {0, 0, 0};
    bool open_left = false, open_right = false;
    // bookkeeping of the streaming surface (surface_reads.inc.h): order of first arrival (RegionState.AddCandidate keeps a position's
    // candidates in that order, RegionState.cs:104-123), and whether every read event behind the candidate came from the device's read walk
    uint64_t stamp = 0;
    bool from_reads = false;
    // MNV calling off, a forced SNV: the support it took from the allele counts at a flush (surface_flush.inc.h) — taken back when the
    // candidate returns to the state (a candidate of a held block that joined a batch for the collapser), where more reads may reach it
    int32_t counted_by_dir[3] = {0, 0, 0};
};

// Appends the read's indel candidates. ref[i] is position i+1 of the chromosome (upper case).
void find_indel_candidates(const ReadView& read, const uint8_t* ref, int64_t ref_len, int32_t min_base_call_quality,
                           int32_t well_anchored_anchor_size, std::vector<HostCandidate>& out);

// The full walk: snvs_and_mnvs adds the candidates of the M operations; call_mnvs / max_mnv_length / max_gap are ShouldBuildUpMNV's.
void find_candidates(const ReadView& read, const uint8_t* ref, int64_t ref_len, int32_t min_base_call_quality,
                     int32_t well_anchored_anchor_size, bool snvs_and_mnvs, bool call_mnvs, int32_t max_mnv_length, int32_t max_gap,
                     std::vector<HostCandidate>& out);

struct FoundCandidate;
// REF / ALT strings of a record of the walk (finder_walk.h); read_bases = the read's bases from FoundCandidate::start_in_read on
void candidate_strings(const FoundCandidate& c, const uint8_t* ref, const uint8_t* read_bases, std::string& ref_allele, std::string& alt_allele);
HostCandidate host_candidate_of(const FoundCandidate& c, const uint8_t* ref, const uint8_t* read_bases);

}  // namespace pisces
