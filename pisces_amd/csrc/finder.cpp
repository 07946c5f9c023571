// finder.cpp — the candidate walk on the host (finder_walk.h is the walk itself, shared with the device kernel): what
// pisces_hip_find_candidates / pisces_hip_find_indel_candidates return.  The streaming surface (pisces_hip_add_reads) discovers its
// candidates on the device (finder_kernels.hip.h) and uses only candidate_strings() from here.
#include "finder.h"

#include "finder_walk.h"

namespace pisces {

// The REF / ALT strings of a found candidate (CandidateVariantFinder.Create's arguments): SNV / MNV = the reference bases and the
// read bases; insertion = the anchor base, and the anchor base + the inserted bases; deletion = the anchor base + the deleted
// bases, and the anchor base.  `read_bases` points at the candidate's own read bases (FoundCandidate::start_in_read onwards).
void candidate_strings(const FoundCandidate& c, const uint8_t* ref, const uint8_t* read_bases, std::string& ref_allele, std::string& alt_allele)
{
    const char* rp = (const char*)ref + c.ref_index;
    if (c.category == PISCES_CAT_INSERTION) {
        ref_allele.assign(rp, 1);
        alt_allele.assign(rp, 1);
        alt_allele.append((const char*)read_bases, (size_t)c.length);
    } else if (c.category == PISCES_CAT_DELETION) {
        ref_allele.assign(rp, (size_t)c.length + 1);
        alt_allele.assign(rp, 1);
    } else {
        ref_allele.assign(rp, (size_t)c.length);
        alt_allele.assign((const char*)read_bases, (size_t)c.length);
    }
}

HostCandidate host_candidate_of(const FoundCandidate& c, const uint8_t* ref, const uint8_t* read_bases)
{
    HostCandidate h;
    h.position = c.position;
    h.category = c.category;
    candidate_strings(c, ref, read_bases, h.ref, h.alt);
    h.support_by_dir[c.dir]++;
    if (c.well_anchored) h.well_anchored_by_dir[c.dir]++;
    h.open_left = c.open_left != 0;
    h.open_right = c.open_right != 0;
    return h;
}

void find_candidates(const ReadView& r, const uint8_t* ref, int64_t ref_len, int32_t minBQ, int32_t anchorSize, bool snvs_and_mnvs,
                     bool call_mnvs, int32_t max_mnv_length, int32_t max_gap, std::vector<HostCandidate>& out)
{
    const FinderParams P = {minBQ, anchorSize, snvs_and_mnvs ? 1 : 0, call_mnvs ? 1 : 0, max_mnv_length, max_gap};
    auto emit = [&](const FoundCandidate& c) { out.push_back(host_candidate_of(c, ref, r.bases + c.start_in_read)); };
    walk::walk_read(r, ref, ref_len, P, emit);
}

void find_indel_candidates(const ReadView& r, const uint8_t* ref, int64_t ref_len, int32_t minBQ, int32_t anchorSize,
                           std::vector<HostCandidate>& out)
{
    find_candidates(r, ref, ref_len, minBQ, anchorSize, false, false, 0, 0, out);
}

}  // namespace pisces
