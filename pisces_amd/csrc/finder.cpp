// finder.cpp — candidate discovery on the host (see finder.h): insertions / deletions always, SNVs / MNVs when MNV calling is on.
#include "finder.h"

#include <algorithm>

namespace pisces {

static inline bool op_is_ref_span(uint8_t t) { return t == 'M' || t == 'D' || t == 'N' || t == '=' || t == 'X'; }
static inline bool op_is_read_span(uint8_t t) { return t == 'M' || t == 'I' || t == 'S' || t == '=' || t == 'X'; }

static inline int dir_at(const ReadView& r, int i)
{
    return r.dirs ? r.dirs[i] : (r.is_reverse ? PISCES_DIR_REVERSE : PISCES_DIR_FORWARD);
}

// CandidateVariantFinder.GetSupportDirection :396-445. A deletion of a read whose XD tag tracks directions inside deletions
// (ReadView::del_dirs) takes GetDeletionDirectionForStitchedRead :468-487: the directions of its first and last deleted base.
static int support_direction(const ReadView& r, int category, int length, int startIndexInRead, int cigarIndex)
{
    if (category == PISCES_CAT_SNV || category == PISCES_CAT_REFERENCE) return dir_at(r, startIndexInRead);
    const int leftAnchorIndex = startIndexInRead - 1;
    const int rightAnchorIndex = category == PISCES_CAT_DELETION ? startIndexInRead : startIndexInRead + length;
    const int lastIndex = r.read_len - 1;
    if (rightAnchorIndex == 0) return dir_at(r, rightAnchorIndex);
    if (leftAnchorIndex == lastIndex) return dir_at(r, lastIndex);
    if (leftAnchorIndex == rightAnchorIndex - 1) {
        if (r.del_dirs && cigarIndex >= 0 && r.del_dirs[2 * cigarIndex] != PISCES_DIR_UNTRACKED) {
            const int startDirection = r.del_dirs[2 * cigarIndex], endDirection = r.del_dirs[2 * cigarIndex + 1];
            return startDirection == PISCES_DIR_STITCHED ? endDirection : startDirection;
        }
        const int startDirection = dir_at(r, leftAnchorIndex), endDirection = dir_at(r, rightAnchorIndex);
        return startDirection == PISCES_DIR_STITCHED ? endDirection : startDirection;
    }
    int direction = PISCES_DIR_FORWARD;
    for (int i = leftAnchorIndex + 1; i < rightAnchorIndex; i++) {
        direction = dir_at(r, i);
        if (direction == PISCES_DIR_STITCHED) return PISCES_DIR_STITCHED;
    }
    return direction;
}

// CandidateVariantFinder.CheckDeletionQuality :294-320
static bool deletion_quality_ok(const ReadView& r, int opStartIndexInRead, int minBQ)
{
    if (r.read_len == 0) return false;
    const int after = (opStartIndexInRead < r.read_len) ? r.quals[opStartIndexInRead] : r.quals[opStartIndexInRead - 1];
    int before = after;
    if (opStartIndexInRead > 0) before = r.quals[opStartIndexInRead - 1];
    return before >= minBQ && after >= minBQ;
}

void find_indel_candidates(const ReadView& r, const uint8_t* ref, int64_t ref_len, int32_t minBQ, int32_t anchorSize,
                           std::vector<HostCandidate>& out)
{
    find_candidates(r, ref, ref_len, minBQ, anchorSize, false, false, 0, 0, out);
}

static inline int allele_code(uint8_t b) { return b == 'A' ? 0 : b == 'G' ? 1 : b == 'C' ? 2 : b == 'T' ? 3 : 4; }

void find_candidates(const ReadView& r, const uint8_t* ref, int64_t ref_len, int32_t minBQ, int32_t anchorSize, bool snvs_and_mnvs,
                     bool call_mnvs, int32_t max_mnv_length, int32_t max_gap, std::vector<HostCandidate>& out)
{
    const size_t first = out.size();
    int refSpan = 0;
    for (int c = 0; c < r.n_cigar; c++)
        if (op_is_ref_span(r.cigar_op[c])) refSpan += (int)r.cigar_len[c];
    const int endPosition = r.position + refSpan - 1;   // Read.EndPosition

    // CandidateVariantFinder.Create :334-345
    auto create = [&](int category, int coordinate, std::string refAllele, std::string altAllele, int startIndexInRead, int cigarIndex = -1) {
        HostCandidate c;
        c.position = coordinate;
        c.category = category;
        const int length = category == PISCES_CAT_INSERTION ? (int)altAllele.size() - 1
                           : category == PISCES_CAT_DELETION ? (int)refAllele.size() - 1 : (int)altAllele.size();   // BaseAllele.Length
        const int dir = support_direction(r, category, length, startIndexInRead, cigarIndex);
        c.support_by_dir[dir]++;
        const int anchor = std::min(coordinate - r.position, endPosition - coordinate);
        if (anchor > std::min(anchorSize - 1, (int)altAllele.size() - 1)) c.well_anchored_by_dir[dir]++;
        c.ref = std::move(refAllele);
        c.alt = std::move(altAllele);
        out.push_back(std::move(c));
    };
    // FlushVariant :183-203
    auto flush_variant = [&](int variantStartIndexInRead, int variantStartIndexInReference, int variantLengthSoFar,
                             int interveningRefLengthSoFar, bool openLeft, bool openRight) {
        if (interveningRefLengthSoFar >= 1) { variantLengthSoFar -= interveningRefLengthSoFar; openRight = false; }
        if (variantLengthSoFar < 1) return;
        create(variantLengthSoFar > 1 ? PISCES_CAT_MNV : PISCES_CAT_SNV, variantStartIndexInReference + 1,
               std::string((const char*)ref + variantStartIndexInReference, (size_t)variantLengthSoFar),
               std::string((const char*)r.bases + variantStartIndexInRead, (size_t)variantLengthSoFar), variantStartIndexInRead);
        out.back().open_left = openLeft;
        out.back().open_right = openRight;
    };
    // ShouldBuildUpMNV :170-181
    auto should_build_up = [&](int mnvLengthSoFar, int interveningRefLengthSoFar, bool refCallNext) {
        if (!call_mnvs) return false;
        if (refCallNext && mnvLengthSoFar == 0) return false;
        if (mnvLengthSoFar + 1 > max_mnv_length) return false;
        if (interveningRefLengthSoFar + (refCallNext ? 1 : 0) > max_gap) return false;
        return true;
    };
    // ExtractSnvsFromOperation :90-168
    auto extract_snvs = [&](int opStartIndexInRead, int operationLength, int opStartIndexInReference) {
        int variantLengthSoFar = 0, interveningRefLengthSoFar = 0;
        bool openLeft = false;
        // An M operation that runs past the contig end stops there; the pending variant is flushed from the bases actually walked
        // (the reference flushes from operationLength and its Substring throws): nothing is read beyond the reference.
        int n_done = operationLength;
        for (int i = 0; i < operationLength; i++) {
            if (opStartIndexInRead + i >= r.read_len) { n_done = i; break; }
            const bool qualityGoodEnough = r.quals[opStartIndexInRead + i] >= minBQ;
            const uint8_t readBase = r.bases[opStartIndexInRead + i];
            if (opStartIndexInReference + i >= ref_len) { n_done = i; break; }
            const uint8_t refBase = ref[opStartIndexInReference + i];
            const bool atEndOfOperation = i == operationLength - 1;
            const bool startingMnvAtEndOfOperation = atEndOfOperation && variantLengthSoFar == 0;
            if (allele_code(readBase) == 4 || allele_code(refBase) == 4 || !qualityGoodEnough) {
                flush_variant(opStartIndexInRead + i - variantLengthSoFar, opStartIndexInReference + i - variantLengthSoFar, variantLengthSoFar,
                              interveningRefLengthSoFar, openLeft, true);
                variantLengthSoFar = 0; interveningRefLengthSoFar = 0; openLeft = true;
            } else if (refBase == readBase) {
                if (should_build_up(variantLengthSoFar, interveningRefLengthSoFar, true) && !startingMnvAtEndOfOperation) {
                    variantLengthSoFar++; interveningRefLengthSoFar++;
                } else {
                    flush_variant(opStartIndexInRead + i - variantLengthSoFar, opStartIndexInReference + i - variantLengthSoFar,
                                  variantLengthSoFar, interveningRefLengthSoFar, openLeft, false);
                    variantLengthSoFar = 0; interveningRefLengthSoFar = 0; openLeft = false;
                }
            } else {
                if (should_build_up(variantLengthSoFar, interveningRefLengthSoFar, false) && !startingMnvAtEndOfOperation) {
                    variantLengthSoFar++; interveningRefLengthSoFar = 0;
                } else {
                    flush_variant(opStartIndexInRead + i - variantLengthSoFar, opStartIndexInReference + i - variantLengthSoFar,
                                  variantLengthSoFar, interveningRefLengthSoFar, openLeft, false);
                    variantLengthSoFar = 1; interveningRefLengthSoFar = 0; openLeft = false;
                }
            }
        }
        flush_variant(opStartIndexInRead + n_done - variantLengthSoFar, opStartIndexInReference + n_done - variantLengthSoFar,
                      variantLengthSoFar, interveningRefLengthSoFar, openLeft, false);
    };

    int startIndexInRead = 0;
    int startIndexInReference = r.position - 1;
    for (int ci = 0; ci < r.n_cigar; ci++) {   // ProcessCigarOps :36-83
        const uint8_t t = r.cigar_op[ci];
        const int len = (int)r.cigar_len[ci];
        if (t == 'M' && snvs_and_mnvs) {
            extract_snvs(startIndexInRead, len, startIndexInReference);
        } else if (t == 'I') {   // ExtractInsertionFromOperation :234-260
            if (!(startIndexInReference - 1 >= ref_len || startIndexInReference == 0) && startIndexInRead + len <= r.read_len &&
                r.quals[startIndexInRead] >= minBQ) {
                std::string refAllele(1, (char)ref[startIndexInReference - 1]);
                std::string altAllele = refAllele + std::string((const char*)r.bases + startIndexInRead, (size_t)len);
                create(PISCES_CAT_INSERTION, startIndexInReference, refAllele, altAllele, startIndexInRead);
            }
        } else if (t == 'D') {   // ExtractDeletionFromOperation :262-292
            if (!((int64_t)startIndexInReference + len >= ref_len) && startIndexInReference >= 1 &&
                deletion_quality_ok(r, startIndexInRead, minBQ)) {
                std::string refAllele((const char*)ref + startIndexInReference - 1, (size_t)len + 1);
                std::string altAllele(1, (char)ref[startIndexInReference - 1]);
                create(PISCES_CAT_DELETION, startIndexInReference, refAllele, altAllele, startIndexInRead, ci);
            }
        }
        if (op_is_read_span(t)) startIndexInRead += len;
        if (op_is_ref_span(t)) startIndexInReference += len;
    }
    if (out.size() == first || r.n_cigar == 0) return;

    // Annotate :496-553 — open-endedness only at the unclipped ends of the read
    int fi = 0, li = r.n_cigar - 1;
    if (r.cigar_op[fi] == 'S') fi = 1;
    if (r.cigar_op[li] == 'S') li = r.n_cigar - 2;
    if (fi >= r.n_cigar || li < 0) return;
    // PositionMap.MaxPosition: the last mapped read base
    int maxPosition = -1;
    {
        int refPos = r.position, lastMapped = -1;
        for (int c = 0; c < r.n_cigar; c++) {
            const bool rs = op_is_read_span(r.cigar_op[c]), fs = op_is_ref_span(r.cigar_op[c]);
            if (rs && fs) lastMapped = refPos + (int)r.cigar_len[c] - 1;
            if (fs) refPos += (int)r.cigar_len[c];
        }
        maxPosition = lastMapped;
    }
    if (maxPosition == -1) maxPosition = r.position - 1;
    const uint8_t firstOp = r.cigar_op[fi], lastOp = r.cigar_op[li];
    for (size_t i = first; i < out.size(); i++) {
        HostCandidate& c = out[i];
        const bool isSnvMnv = c.category == PISCES_CAT_SNV || c.category == PISCES_CAT_MNV;
        if (firstOp == 'M' && c.position == r.position && isSnvMnv) c.open_left = true;
        if (lastOp == 'M' && c.position + (int)c.alt.size() - 1 == maxPosition && isSnvMnv) c.open_right = true;
        if (firstOp == 'I' && c.position == r.position - 1 && c.category == PISCES_CAT_INSERTION) c.open_left = true;
        if (firstOp == 'D' && c.position == r.position - 1 && c.category == PISCES_CAT_DELETION) c.open_left = true;
        if (lastOp == 'I' && c.position == maxPosition && c.category == PISCES_CAT_INSERTION) c.open_right = true;
        if (lastOp == 'D' && c.position == maxPosition && c.category == PISCES_CAT_DELETION) c.open_right = true;
    }
}

}  // namespace pisces
