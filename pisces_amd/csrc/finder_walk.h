// finder_walk.h — ICandidateVariantFinder.FindCandidates as ONE walk that runs on the host and on the device
// (src/lib/Pisces.Domain/Logic/CandidateVariantFinder.cs:31-83 ProcessCigarOps, :90-203 the M-operation state machine,
// :234-292 insertions / deletions, :334-345 Create, :396-487 support direction, :496-553 Annotate).
//
// The walk emits plain records: a candidate is (coordinate, category, where its bases sit in the read / the reference, length,
// support direction, well-anchored bit, open ends).  Allele strings are NOT built here: the device kernel writes the records (and
// the few read bases an ALT allele needs) and the host turns them into strings only for what survives merging.
//   find_candidates_kernel (finder_kernels.hip.h)  lane = read, count pass + write pass of this walk
//   find_candidates        (finder.cpp)            the same walk on the host behind pisces_hip_find_candidates
#pragma once
#include <stdint.h>

#include "expander.h"

#if defined(__HIPCC__)
#define PISCES_HD __host__ __device__
#else
#define PISCES_HD
#endif

namespace pisces {

struct FoundCandidate {
    int32_t position;        // coordinate: the variant's first base (SNV / MNV), the anchor base before the event (insertion / deletion)
    int32_t ref_index;       // 0-based chromosome index of the first base of the REF allele string
    int32_t start_in_read;   // read index of the first read base the ALT allele takes (SNV / MNV: its bases, insertion: the inserted bases)
    int32_t length;          // BaseAllele.Length: SNV / MNV bases, inserted bases, deleted bases
    uint8_t category;        // PISCES_CAT_*
    uint8_t dir;             // support direction
    uint8_t well_anchored;
    uint8_t open_left, open_right;
    uint8_t pad[3];
};

struct FinderParams {
    int32_t min_bq, anchor_size;
    int32_t snvs_and_mnvs;   // walk the M operations (MNV calling on); off: insertions and deletions only
    int32_t call_mnvs, max_mnv_length, max_gap;
    int32_t mark_x_spans;    // what the X and = operations leave (below): 0 nothing, 1 a span mark an operation (the streaming surface's split form
                             // of MNV calling), 2 a record for every base a walk of the operation would have made an SNV of (MNV calling off)
};

// Not a candidate: the positions of an X operation, or of the differing bases of an = operation (position, length).  ProcessCigarOps
// (:44-71) walks M operations only, so such bases are allele counts without SNV candidates; the flush takes the loci of such a span from
// the read walk's candidates instead of from the counts (surface_flush.inc.h, the dirty loci).
constexpr uint8_t kFoundSpanMark = 0x40;
// Not a candidate either: ONE base of an X or = operation that an M operation would have made an SNV candidate of (position, the read base,
// its direction).  MNV calling off: SNVs are called from the allele counts, which hold such bases (AddAlleleCounts walks every operation
// that spans read and reference) though no candidate stands for them; the flush takes them off the allele's support again.
constexpr uint8_t kFoundUnwalked = 0x41;

namespace walk {

PISCES_HD inline bool spans_ref(uint8_t t) { return t == 'M' || t == 'D' || t == 'N' || t == '=' || t == 'X'; }
PISCES_HD inline bool spans_read(uint8_t t) { return t == 'M' || t == 'I' || t == 'S' || t == '=' || t == 'X'; }
PISCES_HD inline bool is_acgt(uint8_t b) { return b == 'A' || b == 'C' || b == 'G' || b == 'T'; }
PISCES_HD inline int dir_of_base(const ReadView& r, int i)
{
    return r.dirs ? r.dirs[i] : (r.is_reverse ? PISCES_DIR_REVERSE : PISCES_DIR_FORWARD);
}

// GetSupportDirection :396-445; a deletion of a read whose XD tag tracks directions inside deletions (del_dirs) takes
// GetDeletionDirectionForStitchedRead :468-487
PISCES_HD inline int support_direction(const ReadView& r, int category, int length, int start_in_read, int cigar_index)
{
    if (category == PISCES_CAT_SNV || category == PISCES_CAT_REFERENCE) return dir_of_base(r, start_in_read);
    const int left = start_in_read - 1;
    const int right = category == PISCES_CAT_DELETION ? start_in_read : start_in_read + length;
    const int last = r.read_len - 1;
    if (right == 0) return dir_of_base(r, right);
    if (left == last) return dir_of_base(r, last);
    if (left == right - 1) {
        int first_dir, second_dir;
        if (r.del_dirs && cigar_index >= 0 && r.del_dirs[2 * cigar_index] != PISCES_DIR_UNTRACKED) {
            first_dir = r.del_dirs[2 * cigar_index];
            second_dir = r.del_dirs[2 * cigar_index + 1];
        } else {
            first_dir = dir_of_base(r, left);
            second_dir = dir_of_base(r, right);
        }
        return first_dir == PISCES_DIR_STITCHED ? second_dir : first_dir;
    }
    int d = PISCES_DIR_FORWARD;
    for (int i = left + 1; i < right; i++) {
        d = dir_of_base(r, i);
        if (d == PISCES_DIR_STITCHED) return PISCES_DIR_STITCHED;
    }
    return d;
}

// per-read constants of Create / Annotate
struct ReadFrame {
    int32_t end_position;    // Read.EndPosition
    int32_t max_position;    // PositionMap.MaxPosition: the last mapped read base (or position - 1)
    uint8_t first_op, last_op;   // the unclipped end operations, 0 = no annotation
};

PISCES_HD inline ReadFrame frame_of(const ReadView& r)
{
    ReadFrame f;
    int ref_span = 0, ref_pos = r.position, last_mapped = -1;
    for (int c = 0; c < r.n_cigar; c++) {
        const uint8_t t = r.cigar_op[c];
        const int len = (int)r.cigar_len[c];
        if (spans_ref(t)) {
            if (spans_read(t)) last_mapped = ref_pos + len - 1;
            ref_span += len;
            ref_pos += len;
        }
    }
    f.end_position = r.position + ref_span - 1;
    f.max_position = last_mapped == -1 ? r.position - 1 : last_mapped;
    f.first_op = f.last_op = 0;
    if (r.n_cigar > 0) {
        int fi = 0, li = r.n_cigar - 1;
        if (r.cigar_op[fi] == 'S') fi = 1;
        if (r.cigar_op[li] == 'S') li = r.n_cigar - 2;
        if (fi < r.n_cigar && li >= 0) { f.first_op = r.cigar_op[fi]; f.last_op = r.cigar_op[li]; }
    }
    return f;
}

// Create :334-345 + Annotate :496-553 for one candidate, then emit(candidate)
template <typename Emit>
PISCES_HD inline void finish_candidate(const ReadView& r, const ReadFrame& f, const FinderParams& P, int category, int coordinate, int ref_index,
                                       int start_in_read, int length, int alt_allele_len, int cigar_index, bool open_left, bool open_right,
                                       Emit& emit)
{
    FoundCandidate c;
    c.position = coordinate;
    c.ref_index = ref_index;
    c.start_in_read = start_in_read;
    c.length = length;
    c.category = (uint8_t)category;
    c.dir = (uint8_t)support_direction(r, category, length, start_in_read, cigar_index);
    const int to_start = coordinate - r.position, to_end = f.end_position - coordinate;
    const int anchor = to_start < to_end ? to_start : to_end;
    const int need = (P.anchor_size - 1) < (alt_allele_len - 1) ? (P.anchor_size - 1) : (alt_allele_len - 1);
    c.well_anchored = anchor > need;
    const bool snv_mnv = category == PISCES_CAT_SNV || category == PISCES_CAT_MNV;
    if (f.first_op == 'M' && snv_mnv && coordinate == r.position) open_left = true;
    if (f.last_op == 'M' && snv_mnv && coordinate + length - 1 == f.max_position) open_right = true;
    if (f.first_op == 'I' && category == PISCES_CAT_INSERTION && coordinate == r.position - 1) open_left = true;
    if (f.first_op == 'D' && category == PISCES_CAT_DELETION && coordinate == r.position - 1) open_left = true;
    if (f.last_op == 'I' && category == PISCES_CAT_INSERTION && coordinate == f.max_position) open_right = true;
    if (f.last_op == 'D' && category == PISCES_CAT_DELETION && coordinate == f.max_position) open_right = true;
    c.open_left = open_left;
    c.open_right = open_right;
    c.pad[0] = c.pad[1] = c.pad[2] = 0;
    emit(c);
}

// Where the bases of an M operation come from: step(i, read base, reference base, quality) for i = 0 .. n - 1 (n bases of the operation
// lie on the read and on the contig).  One byte at a time here; the device kernels take them four at a time, the next word requested
// while the state machine works through the current one (WordBases, finder_kernels.hip.h).
struct ByteBases {
    template <typename Step>
    PISCES_HD static void for_each(const ReadView& r, const uint8_t* ref, int op_read0, int op_ref0, int n, Step& step)
    {
        const uint8_t* pb = r.bases + op_read0; const uint8_t* pq = r.quals + op_read0; const uint8_t* pf = ref + op_ref0;
        for (int i = 0; i < n; i++) step(i, pb[i], pf[i], pq[i]);
    }
};

// The M-operation state machine (ExtractSnvsFromOperation :90-168 with ShouldBuildUpMNV :170-181 and FlushVariant :183-203).
// `run` = bases of the variant being built, trailing reference matches included; `tail` = those trailing matches, which are
// given back when the variant is closed; a variant that was closed by a base that cannot be called (N, low quality) is open on
// that side.  A walk that reaches the end of the contig (or of the read) closes the pending variant where it stopped.
template <typename Src, typename Emit>
PISCES_HD inline void walk_match_op(const ReadView& r, const ReadFrame& f, const uint8_t* ref, int64_t ref_len, const FinderParams& P,
                                    int op_read0, int op_len, int op_ref0, Emit& emit)
{
    int run = 0, tail = 0;
    bool open_left = false;
    auto close = [&](int at, bool open_right) {   // `at` = bases of the operation walked so far
        int len = run;
        if (tail >= 1) { len -= tail; open_right = false; }
        if (len < 1) return;
        const int start_read = op_read0 + at - run, start_ref = op_ref0 + at - run;
        finish_candidate(r, f, P, len > 1 ? PISCES_CAT_MNV : PISCES_CAT_SNV, start_ref + 1, start_ref, start_read, len, len, -1, open_left,
                         open_right, emit);
    };
    auto may_grow = [&](bool matches) {
        if (!P.call_mnvs) return false;
        if (matches && run == 0) return false;            // never start on a reference base
        if (run + 1 > P.max_mnv_length) return false;
        return tail + (matches ? 1 : 0) <= P.max_gap;
    };
    // the bases of the operation that lie on the read and on the contig (the walk stops at the end of either)
    int64_t lim = op_len;
    if ((int64_t)r.read_len - op_read0 < lim) lim = (int64_t)r.read_len - op_read0;
    if (ref_len - (int64_t)op_ref0 < lim) lim = ref_len - (int64_t)op_ref0;
    const int walked = lim < 0 ? 0 : (int)lim;
    // One base: what happens to the variant being built is decided first (selects), the variant is closed in ONE place — on the device
    // a wave pays for every branch any of its 64 reads takes, and three copies of the closing code in three branches were most of
    // what a step cost.
    auto step = [&](int i, uint8_t rb, uint8_t fb, uint8_t q) {
        const bool callable = is_acgt(rb) && is_acgt(fb) && q >= P.min_bq;
        const bool alone_on_last_base = i == op_len - 1 && run == 0;   // no MNV is started on the last base of an operation
        const bool matches = rb == fb;
        const bool grows = callable && may_grow(matches) && !alone_on_last_base;
        if (grows) {
            run++;
            tail = matches ? tail + 1 : 0;
            return;
        }
        close(i, !callable);   // (a base that cannot be called leaves the variant open on that side)
        run = (callable && !matches) ? 1 : 0;
        tail = 0;
        open_left = !callable;
    };
    Src::for_each(r, ref, op_read0, op_ref0, walked, step);
    close(walked, false);
}

// ProcessCigarOps walks M operations only (:44-71): the bases of an X operation — and those of an = operation that differ from the
// reference after all (a CIGAR is not checked against the reference) — are allele counts that no SNV candidate stands for.  With
// mark_x_spans = 1 the walk leaves a span mark over them (all of an X operation; from the first to the last differing base of an = operation),
// with mark_x_spans = 2 a kFoundUnwalked record for each of them that an M operation would have made an SNV candidate of.
template <typename Emit>
PISCES_HD inline void mark_unwalked_span(const ReadView& r, uint8_t t, int len, int in_read, int in_ref, const uint8_t* ref, int64_t ref_len,
                                         const FinderParams& P, Emit& emit)
{
    FoundCandidate c;
    c.dir = 0;
    c.well_anchored = c.open_left = c.open_right = 0;
    c.pad[0] = c.pad[1] = c.pad[2] = 0;
    if (P.mark_x_spans == 2) {   // the callable mismatches of the operation, one by one (walk_match_op's `callable && !matches`)
        for (int i = 0; i < len && in_read + i < r.read_len && (int64_t)in_ref + i < ref_len; i++) {
            const uint8_t rb = r.bases[in_read + i], fb = ref[in_ref + i];
            if (rb == fb || !is_acgt(rb) || !is_acgt(fb) || r.quals[in_read + i] < P.min_bq) continue;
            c.position = in_ref + i + 1;
            c.ref_index = in_ref + i;
            c.start_in_read = in_read + i;
            c.length = 1;
            c.category = kFoundUnwalked;
            c.dir = (uint8_t)dir_of_base(r, in_read + i);
            emit(c);
        }
        return;
    }
    int lo = 0, hi = len - 1;
    if (t == '=') {
        lo = -1;
        for (int i = 0; i < len && in_read + i < r.read_len && (int64_t)in_ref + i < ref_len; i++)
            if (r.bases[in_read + i] != ref[in_ref + i]) { if (lo < 0) lo = i; hi = i; }
        if (lo < 0) return;
    }
    c.position = in_ref + lo + 1;
    c.ref_index = in_ref + lo;
    c.start_in_read = in_read + lo;
    c.length = hi - lo + 1;
    c.category = kFoundSpanMark;
    emit(c);
}

// ProcessCigarOps :36-83: every candidate of one read, in the reference's order of discovery
template <typename Src = ByteBases, typename Emit>
PISCES_HD inline void walk_read(const ReadView& r, const uint8_t* ref, int64_t ref_len, const FinderParams& P, Emit& emit)
{
    const ReadFrame f = frame_of(r);
    int in_read = 0, in_ref = r.position - 1;
    for (int ci = 0; ci < r.n_cigar; ci++) {
        const uint8_t t = r.cigar_op[ci];
        const int len = (int)r.cigar_len[ci];
        if (t == 'M') {
            if (P.snvs_and_mnvs) walk_match_op<Src>(r, f, ref, ref_len, P, in_read, len, in_ref, emit);
        } else if (t == 'I') {   // ExtractInsertionFromOperation :234-260: anchored on the base before, gated on the first inserted base
            const bool off_contig = (int64_t)in_ref - 1 >= ref_len || in_ref == 0;
            if (!off_contig && in_read < r.read_len && in_read + len <= r.read_len && r.quals[in_read] >= P.min_bq)
                finish_candidate(r, f, P, PISCES_CAT_INSERTION, in_ref, in_ref - 1, in_read, len, len + 1, -1, false, false, emit);
        } else if (t == 'D') {   // ExtractDeletionFromOperation :262-292: both flanking qualities (CheckDeletionQuality :294-320)
            bool flanks_ok = false;
            if (r.read_len > 0) {
                const int after = in_read < r.read_len ? r.quals[in_read] : r.quals[in_read - 1];
                const int before = in_read > 0 ? r.quals[in_read - 1] : after;
                flanks_ok = before >= P.min_bq && after >= P.min_bq;
            }
            if ((int64_t)in_ref + len < ref_len && in_ref >= 1 && flanks_ok)
                finish_candidate(r, f, P, PISCES_CAT_DELETION, in_ref, in_ref - 1, in_read, len, 1, ci, false, false, emit);
        }
        else if ((t == 'X' || t == '=') && P.mark_x_spans && len > 0) mark_unwalked_span(r, t, len, in_read, in_ref, ref, ref_len, P, emit);
        if (spans_read(t)) in_read += len;
        if (spans_ref(t)) in_ref += len;
    }
}

}  // namespace walk
}  // namespace pisces
