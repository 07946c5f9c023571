// expander.cpp — host side of IStateManager.AddAlleleCounts (src/lib/Pisces.Processing/RegionState/RegionStateManager.cs:118-220): one
// packed observation tuple per allele-count increment, from the per-base walk the device kernel uses (read_walk.h).  Pure C++ (no
// device): what pisces_hip_expand_reads returns and what the CPU tests check against the oracle.
#include "expander.h"
#include "read_walk.h"

#include <vector>

namespace pisces {

// The host form of the read walk: a loop over read_walk.h's per-base function, observation for observation what the device's
// expand_reads_kernel logs.  Positions below 1 are not emitted.
int32_t expand_read(const ReadView& r, int32_t minBQ, ObservationSink& sink)
{
    const ReadShape shape = read_shape(r.position, r.read_len, r.n_cigar, r.cigar_op, r.cigar_len);
    if (shape.read_span > r.read_len) return PISCES_E_INVALID_ARG;   // ValidateCigar: the CIGAR is longer than the read
    const uint32_t lastAnchor = PISCES_NUM_ANCHORS - 1;
    for (int i = 0; i < r.read_len; i++) {
        const uint32_t dir = r.dirs ? r.dirs[i] : (r.is_reverse ? PISCES_DIR_REVERSE : PISCES_DIR_FORWARD);
        if (dir > 2) return PISCES_E_INVALID_ARG;
        const BaseWalk w = walk_base(shape, i, r.quals, minBQ);
        for (int k = 0; k < w.n_soft; k++) sink.emit(w.soft_first + k, PISCES_TUPLE_PACK(0, lastAnchor, dir, PISCES_ALLELE_DEL, 255));
        if (w.position != -1) {
            if (w.anchor < 0) return PISCES_E_UNMAPPED_BASE;   // GetAnchorType throws (RegionStateManager.cs:109-113)
            for (int k = 0; k < w.n_gap; k++) sink.emit(w.gap_first + k, PISCES_TUPLE_PACK(0, (uint32_t)w.anchor, dir, PISCES_ALLELE_DEL, 255));
            // the quality test "qual < minBQ -> N" (:179-181) is applied by the kernel from the tuple's qual
            if (w.n_base) sink.emit(w.position, PISCES_TUPLE_PACK(0, (uint32_t)w.anchor, dir, walk_allele_type(r.bases[i]), r.quals[i]));
        }
        for (int k = 0; k < w.n_end; k++) sink.emit(w.end_first + k, PISCES_TUPLE_PACK(0, lastAnchor, dir, PISCES_ALLELE_DEL, 255));
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
