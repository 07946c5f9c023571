// expander.cpp — host side of IStateManager.AddAlleleCounts: walks a read exactly as
// RegionStateManager.AddAlleleCounts does (src/lib/Pisces.Processing/RegionState/RegionStateManager.cs:118-220)
// and emits one packed observation tuple per allele-count increment instead of bumping a
// dense int[1000,6,3,11] block.  Pure C++ (no device); the tuples are what the kernels stream.
#include "expander.h"

#include <vector>

namespace pisces {

static inline bool op_is_ref_span(uint8_t t) { return t == 'M' || t == 'D' || t == 'N' || t == '=' || t == 'X'; }
static inline bool op_is_read_span(uint8_t t) { return t == 'M' || t == 'I' || t == 'S' || t == '=' || t == 'X'; }

// AlleleHelper.GetAlleleType (src/lib/Pisces.Domain/Utility/AlleleHelper.cs:13-32)
static inline uint32_t allele_type_of(uint8_t c)
{
    switch (c) {
    case 'A': return PISCES_ALLELE_A;
    case 'C': return PISCES_ALLELE_C;
    case 'G': return PISCES_ALLELE_G;
    case 'T': return PISCES_ALLELE_T;
    default: return PISCES_ALLELE_N;
    }
}

// RegionStateManager.GetAnchorType :83-116 with numAnchorTypes = 5; -1 where the reference throws
static inline int anchor_type(int alignmentEnd, int basePosition, int alignmentStart)
{
    const int numAnchorTypes = PISCES_ANCHOR_SIZE, numAnchorIndexes = PISCES_NUM_ANCHORS;
    int leftAnchor = basePosition - alignmentStart;
    int rightAnchor = alignmentEnd - basePosition;
    int minAnchor;
    if (leftAnchor >= rightAnchor) {
        if (rightAnchor >= numAnchorTypes) return numAnchorTypes;
        minAnchor = numAnchorIndexes - rightAnchor - 1;
    } else {
        if (leftAnchor >= numAnchorTypes) return numAnchorTypes;
        minAnchor = leftAnchor;
    }
    return minAnchor < 0 ? -1 : minAnchor;
}

// CandidateVariantFinder.CheckDeletionQuality (src/lib/Pisces.Domain/Logic/CandidateVariantFinder.cs:294-320)
static inline bool deletion_quality_ok(const uint8_t* quals, int readLen, int opStartIndexInRead, int minBQ)
{
    if (readLen == 0) return false;
    int after = (opStartIndexInRead < readLen) ? quals[opStartIndexInRead] : quals[opStartIndexInRead - 1];
    int before = after;
    if (opStartIndexInRead > 0) before = quals[opStartIndexInRead - 1];
    return before >= minBQ && after >= minBQ;
}

int32_t expand_read(const ReadView& r, int32_t minBQ, ObservationSink& sink)
{
    const int n = r.read_len;
    // Read.UpdatePositionMap (src/lib/Pisces.Domain/Models/Read.cs:535-562)
    std::vector<int32_t>& posmap = sink.scratch;
    posmap.assign((size_t)n, -1);
    int refSpan = 0;
    {
        int readIndex = 0, referencePosition = r.position;
        for (int c = 0; c < r.n_cigar; c++) {
            const bool readSpan = op_is_read_span(r.cigar_op[c]), refSpanOp = op_is_ref_span(r.cigar_op[c]);
            const uint32_t len = r.cigar_len[c];
            if (refSpanOp) refSpan += (int)len;
            if (readSpan) {
                for (uint32_t k = 0; k < len; k++) {
                    if (readIndex < n) posmap[(size_t)readIndex] = refSpanOp ? referencePosition++ : -1;
                    readIndex++;
                }
                if (readIndex > n) return PISCES_E_INVALID_ARG;  // ValidateCigar: CIGAR longer than the read
            } else if (refSpanOp) {
                referencePosition += (int)len;
            }
        }
    }
    auto has_op_from_end = [&](int index, uint8_t type) {  // CigarExtensions.HasOperationAtOpIndex(.., fromEnd: true)
        int opIndex = r.n_cigar - index - 1;
        return r.n_cigar > opIndex && opIndex >= 0 && r.cigar_op[opIndex] == type;
    };
    auto dir_at = [&](int i) -> uint32_t {
        return r.dirs ? r.dirs[i] : (r.is_reverse ? PISCES_DIR_REVERSE : PISCES_DIR_FORWARD);
    };

    int lastPosition = r.position - 1;
    int deletionLength = 0;
    int lengthBeforeDeletion = n;
    const bool endsInDeletion = has_op_from_end(0, 'D');
    const bool endsInDeletionBeforeSoftclip = has_op_from_end(1, 'D') && has_op_from_end(0, 'S');
    if (endsInDeletion || endsInDeletionBeforeSoftclip) {
        deletionLength = (int)(endsInDeletionBeforeSoftclip ? r.cigar_len[r.n_cigar - 2] : r.cigar_len[r.n_cigar - 1]);
        lengthBeforeDeletion = endsInDeletionBeforeSoftclip ? n - (int)r.cigar_len[r.n_cigar - 1] : n;
    }
    const int alignmentEnd = r.position + refSpan - 1;   // Read.EndPosition (Read.cs:88-91, BamCommon.cs:119)
    const int alignmentStart = r.position;
    const uint32_t lastAnchor = PISCES_NUM_ANCHORS - 1;

    for (int i = 0; i < n; i++) {
        const uint32_t dir = dir_at(i);
        if (dir > 2) return PISCES_E_INVALID_ARG;
        if (endsInDeletionBeforeSoftclip && i == lengthBeforeDeletion) {
            if (deletion_quality_ok(r.quals, n, i, minBQ))
                for (int j = 1; j < deletionLength + 1; j++)
                    sink.emit(j + lastPosition, PISCES_TUPLE_PACK(0, lastAnchor, dir, PISCES_ALLELE_DEL, 255));
        }
        const int position = posmap[(size_t)i];
        if (position == -1) continue;
        const int anchor = anchor_type(alignmentEnd, position, alignmentStart);
        if (anchor < 0) return PISCES_E_UNMAPPED_BASE;
        if (deletion_quality_ok(r.quals, n, i, minBQ))
            for (int j = lastPosition + 1; j < position; j++)
                sink.emit(j, PISCES_TUPLE_PACK(0, (uint32_t)anchor, dir, PISCES_ALLELE_DEL, 255));
        // the quality test "qual < minBQ -> N" (:179-181) is applied by the kernel from the tuple's qual
        sink.emit(position, PISCES_TUPLE_PACK(0, (uint32_t)anchor, dir, allele_type_of(r.bases[i]), r.quals[i]));
        lastPosition = position;
    }
    if (endsInDeletion && n > 0) {
        if (deletion_quality_ok(r.quals, n, n - 1, minBQ)) {
            const uint32_t dir = dir_at(n - 1);
            for (int j = 1; j < deletionLength + 1; j++)
                sink.emit(j + lastPosition, PISCES_TUPLE_PACK(0, lastAnchor, dir, PISCES_ALLELE_DEL, 255));
        }
    }
    return PISCES_OK;
}

ReadView read_view(const PiscesReadBatch* b, int32_t i)
{
    ReadView r;
    r.position = b->position[i];
    r.n_cigar = b->cigar_offset[i + 1] - b->cigar_offset[i];
    r.cigar_op = b->cigar_op + b->cigar_offset[i];
    r.cigar_len = b->cigar_len + b->cigar_offset[i];
    r.read_len = b->seq_offset[i + 1] - b->seq_offset[i];
    r.bases = b->bases + b->seq_offset[i];
    r.quals = b->quals + b->seq_offset[i];
    r.dirs = b->directions ? b->directions + b->seq_offset[i] : nullptr;
    r.del_dirs = b->deletion_directions ? b->deletion_directions + 2 * (size_t)b->cigar_offset[i] : nullptr;
    r.is_reverse = b->flags[i] & 1;
    return r;
}

}  // namespace pisces
