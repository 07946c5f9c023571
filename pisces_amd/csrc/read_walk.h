// read_walk.h — IStateManager.AddAlleleCounts (src/lib/Pisces.Processing/RegionState/RegionStateManager.cs:118-220) as a per-base
// function that runs on the device and on the host.
//
// The reference walks a read base by base and carries state (the position map of Read.UpdatePositionMap, Read.cs:535-562, the last
// mapped position, the deletion pending at the end).  Here every base index answers for itself from the CIGAR alone: which position
// it sits on, which run of deleted positions it closes, and whether it carries one of the two terminal-deletion cases — so that 64
// lanes can take 64 bases of a read at once (expand_reads_kernel: lane = base, a wave scan gives every lane its log slots) and a host
// loop over the same function gives the same observations in the same order (expander.cpp behind pisces_hip_expand_reads).
#pragma once
#include <stdint.h>

#include "../../include/pisces_hip.h"

#if defined(__HIPCC__)
#define PISCES_HD __host__ __device__
#else
#define PISCES_HD
#endif

namespace pisces {

PISCES_HD inline bool walk_op_ref_span(uint8_t t) { return t == 'M' || t == 'D' || t == 'N' || t == '=' || t == 'X'; }
PISCES_HD inline bool walk_op_read_span(uint8_t t) { return t == 'M' || t == 'I' || t == 'S' || t == '=' || t == 'X'; }

// AlleleHelper.GetAlleleType (src/lib/Pisces.Domain/Utility/AlleleHelper.cs:13-32)
PISCES_HD inline uint32_t walk_allele_type(uint8_t c)
{
    return c == 'A' ? PISCES_ALLELE_A : c == 'C' ? PISCES_ALLELE_C : c == 'G' ? PISCES_ALLELE_G : c == 'T' ? PISCES_ALLELE_T : PISCES_ALLELE_N;
}

// RegionStateManager.GetAnchorType :83-116 with numAnchorTypes = 5; -1 where the reference throws (a position outside the alignment)
PISCES_HD inline int walk_anchor_type(int alignmentEnd, int basePosition, int alignmentStart)
{
    const int leftAnchor = basePosition - alignmentStart, rightAnchor = alignmentEnd - basePosition;
    int minAnchor;
    if (leftAnchor >= rightAnchor) {
        if (rightAnchor >= PISCES_ANCHOR_SIZE) return PISCES_ANCHOR_SIZE;
        minAnchor = PISCES_NUM_ANCHORS - rightAnchor - 1;
    } else {
        if (leftAnchor >= PISCES_ANCHOR_SIZE) return PISCES_ANCHOR_SIZE;
        minAnchor = leftAnchor;
    }
    return minAnchor < 0 ? -1 : minAnchor;
}

// What is the same for every base of a read
struct ReadShape {
    int32_t pos0, n, nc;          // Read.Position, bases, CIGAR operations
    const uint8_t* ops;
    const uint32_t* lens;
    int32_t ref_span;             // reference positions the alignment covers
    int32_t alignment_end;        // Read.EndPosition (Read.cs:88-91, BamCommon.cs:119)
    int32_t last_mapped;          // position of the last mapped base (pos0 - 1: none)
    int32_t read_span;            // bases the CIGAR accounts for (> n: ValidateCigar fails)
    // a deletion at the end of the read, or before a final soft clip (:131-141): its length and the index of the base that carries it
    bool ends_in_del, ends_in_del_soft;
    int32_t del_len, length_before_deletion;
};

PISCES_HD inline ReadShape read_shape(int32_t pos0, int32_t n, int32_t nc, const uint8_t* ops, const uint32_t* lens)
{
    ReadShape s;
    s.pos0 = pos0; s.n = n; s.nc = nc; s.ops = ops; s.lens = lens;
    s.ref_span = 0;
    s.read_span = 0;
    s.last_mapped = pos0 - 1;
    int rp = pos0;
    for (int c = 0; c < nc; c++) {
        const uint8_t t = ops[c];
        const int len = (int)lens[c];
        if (walk_op_read_span(t)) s.read_span += len;
        if (walk_op_ref_span(t)) {
            s.ref_span += len;
            if (walk_op_read_span(t) && len > 0) s.last_mapped = rp + len - 1;
            rp += len;
        }
    }
    s.alignment_end = pos0 + s.ref_span - 1;
    s.ends_in_del = nc >= 1 && ops[nc - 1] == 'D';
    s.ends_in_del_soft = nc >= 2 && ops[nc - 2] == 'D' && ops[nc - 1] == 'S';
    s.del_len = 0;
    s.length_before_deletion = n;
    if (s.ends_in_del || s.ends_in_del_soft) {
        s.del_len = (int)(s.ends_in_del_soft ? lens[nc - 2] : lens[nc - 1]);
        s.length_before_deletion = s.ends_in_del_soft ? n - (int)lens[nc - 1] : n;
    }
    return s;
}

// What base i of the read adds to the counts, in the order the reference adds it (:143-213): the deleted positions of a terminal
// deletion in front of a soft clip (anchor index 10), the deleted positions of the gap the base closes (the base's own anchor), the
// base, the deleted positions of a terminal deletion at the read's end (anchor index 10).  Positions below 1 are never counted.
struct BaseWalk {
    int32_t position;             // -1: the base sits on no reference position (insertion, soft clip)
    int32_t anchor;               // of the base and of the gap it closes; -1 = the reference throws here
    int32_t n_soft, soft_first;
    int32_t n_gap, gap_first;
    int32_t n_base;
    int32_t n_end, end_first;
};

PISCES_HD inline BaseWalk walk_base(const ReadShape& s, int32_t i, const uint8_t* quals, int32_t min_bq)
{
    BaseWalk w;
    w.position = -1; w.anchor = 0;
    w.n_soft = w.n_gap = w.n_base = w.n_end = 0;
    w.soft_first = w.gap_first = w.end_first = 0;
    // Read.UpdatePositionMap for index i, plus the position of the last mapped base before it
    int p = -1, lp = s.pos0 - 1;
    {
        int ri = 0, rp = s.pos0, lastm = s.pos0 - 1;
        for (int c = 0; c < s.nc; c++) {
            const uint8_t t = s.ops[c];
            const int len = (int)s.lens[c];
            const bool rs = walk_op_read_span(t), fs = walk_op_ref_span(t);
            if (rs) {
                if (i >= ri && i < ri + len) {
                    if (fs) { p = rp + (i - ri); lp = (i == ri) ? lastm : p - 1; }
                    else lp = lastm;
                }
                if (fs && len > 0) { lastm = rp + len - 1; rp += len; }
                ri += len;
            } else if (fs) {
                rp += len;
            }
        }
    }
    // CandidateVariantFinder.CheckDeletionQuality (CandidateVariantFinder.cs:294-320) at index i < n
    const int after = quals[i], before = i > 0 ? quals[i - 1] : after;
    const bool dq = before >= min_bq && after >= min_bq;
    if (s.ends_in_del_soft && i == s.length_before_deletion && dq) {
        w.soft_first = lp + 1 > 1 ? lp + 1 : 1;
        const int cnt = lp + s.del_len - w.soft_first + 1;
        w.n_soft = cnt > 0 ? cnt : 0;
    }
    w.position = p;
    if (p != -1) {
        w.anchor = walk_anchor_type(s.alignment_end, p, s.pos0);
        if (dq) {
            w.gap_first = lp + 1 > 1 ? lp + 1 : 1;
            const int cnt = p - 1 - w.gap_first + 1;
            w.n_gap = cnt > 0 ? cnt : 0;
        }
        w.n_base = p > 0 ? 1 : 0;
    }
    if (s.ends_in_del && i == s.n - 1 && dq) {
        w.end_first = s.last_mapped + 1 > 1 ? s.last_mapped + 1 : 1;
        const int cnt = s.last_mapped + s.del_len - w.end_first + 1;
        w.n_end = cnt > 0 ? cnt : 0;
    }
    return w;
}

}  // namespace pisces
