// finder.h — host side of ICandidateVariantFinder.FindCandidates for the candidates that do NOT fall out of
// the device allele counts: insertions and deletions (src/lib/Pisces.Domain/Logic/CandidateVariantFinder.cs:234-292).
// SNV candidates are implied by the counts (callMNVs off, the reference default), so M operations are not walked.
#pragma once
#include <stdint.h>

#include <string>
#include <vector>

#include "expander.h"

namespace pisces {

struct HostCandidate {
    int32_t position = 0;     // coordinate of the anchor base (one before the inserted / deleted bases)
    int32_t category = 0;     // PISCES_CAT_INSERTION / PISCES_CAT_DELETION
    std::string ref, alt;
    int32_t support_by_dir[3] = {0, 0, 0};
    int32_t well_anchored_by_dir[3] = {0, 0, 0};
    bool open_left = false, open_right = false;
};

// Appends the read's indel candidates. ref[i] is position i+1 of the chromosome (upper case).
void find_indel_candidates(const ReadView& read, const uint8_t* ref, int64_t ref_len, int32_t min_base_call_quality,
                           int32_t well_anchored_anchor_size, std::vector<HostCandidate>& out);

}  // namespace pisces
