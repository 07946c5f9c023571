// expander.h — read -> observation-tuple expansion (host side of IStateManager.AddAlleleCounts).
#pragma once
#include <stdint.h>

#include <vector>

#include "../../include/pisces_hip.h"

namespace pisces {

struct ReadView {
    int32_t position;   // Read.Position, 1-based
    int32_t n_cigar;
    const uint8_t* cigar_op;
    const uint32_t* cigar_len;
    int32_t read_len;
    const uint8_t* bases;
    const uint8_t* quals;
    const uint8_t* dirs;   // per-base DirectionType or nullptr
    const uint8_t* del_dirs = nullptr;   // 2 per CIGAR op (first / last deleted base of a D op, 255 = untracked) or nullptr
    int32_t is_reverse;
};

// receives (position, tuple-with-locus-0) pairs in read order
struct ObservationSink {
    virtual void emit(int32_t position, uint32_t tuple) = 0;
    virtual ~ObservationSink() {}
};

ReadView read_view(const PiscesReadBatch* batch, int32_t i);
// returns PISCES_OK or a PISCES_E_* code (reads are walked atomically: on error nothing of that
// read has been committed only if the sink buffers per read)
int32_t expand_read(const ReadView& read, int32_t min_base_call_quality, ObservationSink& sink);

}  // namespace pisces
