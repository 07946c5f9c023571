// bgzf_kernels.hip.h — SURVEY row f4 (upstream stage): BGZF blocks inflated on the device.
//
// A BAM file is a chain of BGZF blocks: gzip members of at most 64 KiB of payload, each an independent RFC 1951 DEFLATE stream
// (BamReader.ReadBlock, src/lib/Alignment.IO/BamReader.cs:603-645, hands each to the native zlib binding UncompressBlock,
// src/lib/Common.IO/FileCompression.cs:14-16).  Independent streams are the parallelism: one wave inflates one block, a
// launch inflates every block of a file region.  Inside a block a code's position depends on every code before it; what the 64 lanes
// share of that serial chain is described at inflate_codes.  Stored, fixed and dynamic blocks, any number of them per stream.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/pisces_hip.h"

namespace pisces {

enum : int32_t {
    kInflateOk = 0,
    kInflateInputExhausted = 1,   // the stream wants more bytes than the block holds
    kInflateBadBlockType = 2,
    kInflateBadStoredLength = 3,
    kInflateBadCodeLengths = 4,   // over-subscribed / incomplete code, bad repeat, too many lengths, no end-of-block code
    kInflateBadSymbol = 5,        // invalid literal / length or distance symbol, or a code that is not in the table
    kInflateDistanceTooFar = 6,
    kInflateOutputOverflow = 7,   // more bytes than the block's ISIZE announces
    kInflateLengthMismatch = 8,   // the stream ended before ISIZE bytes were produced
};

// ---- one wave per BGZF block ----
// Block headers and code lengths are read by all 64 lanes alike through one bit buffer (identical scalar registers, broadcast LDS
// reads: no divergence), the decode tables are filled by all lanes, and the symbols themselves are decoded by the lanes side by side,
// one bit offset each (inflate_codes).  Output goes straight to HBM.  ~8.5 KB of LDS per wave, so a CU holds as many waves as it has
// slots for and a file's thousands of blocks are all in flight.

// The payload reaches the bit buffer through a window in LDS: kInWindow bytes of the file copied by all 64 lanes at once (16 bytes a
// lane per load instruction), so that topping up the bit buffer is an LDS read (~100 cycles) and not a global load on the decode's
// dependent chain (a microsecond while the chip is busy: with a few thousand blocks in flight that wait was most of a symbol's time).
#ifndef PISCES_INFLATE_WINDOW
#define PISCES_INFLATE_WINDOW 1024
#endif
constexpr int kInWindow = PISCES_INFLATE_WINDOW;   // bytes; the window starts on a 16-byte boundary of the file copy
static_assert(kInWindow % 1024 == 0 && kInWindow >= 1024, "the window is filled 16 bytes a lane at a time by 64 lanes");

struct InflateStream {
    const uint8_t* in;        // first byte of the payload
    int32_t in_len, in_pos;   // in_pos: next byte to load into the bit buffer, relative to `in`; in + in_pos is 4-byte aligned
    uint64_t bitbuf;
    int32_t bitcnt;
    uint8_t* out;
    int32_t out_len, out_pos;
    int32_t err;
    int lane;
    uint32_t* window;         // LDS [kInWindow / 4]
    int32_t win_pos;          // in_pos of the window's first byte (or a value that no in_pos lies in)
};

__device__ __forceinline__ void wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ void inflate_refill(InflateStream& s)
{
    if (s.bitcnt <= 32) {
        // the file bytes continue past the payload (trailer, next header; the device copy is padded), so the loads themselves are always
        // in bounds; consuming bits that lie past the payload is the error
        if (s.in_pos >= s.in_len + 8) { s.err = s.err ? s.err : kInflateInputExhausted; return; }
        if ((uint32_t)(s.in_pos - s.win_pos) >= (uint32_t)kInWindow) {
            // slide the window: it starts at the 16-byte boundary at or below in_pos
            const int32_t skew = (int32_t)((uintptr_t)(s.in + s.in_pos) & 15u);
            s.win_pos = s.in_pos - skew;
            const uint4* src = reinterpret_cast<const uint4*>(s.in + s.win_pos);
            uint4* dst = reinterpret_cast<uint4*>(s.window);
            wave_lds_fence();   // (earlier reads of the window are done)
#pragma unroll
            for (int k = 0; k < kInWindow / 16 / 64; k++) dst[k * 64 + s.lane] = src[k * 64 + s.lane];
            wave_lds_fence();
        }
        // every lane reads the same word; telling the compiler so keeps the bit buffer and all that follows from it (symbols, lengths,
        // the branches on them) in scalar registers and scalar branches instead of 64-wide copies under exec masks
        const uint32_t w = (uint32_t)__builtin_amdgcn_readfirstlane((int)s.window[(uint32_t)(s.in_pos - s.win_pos) >> 2]);
        s.bitbuf |= (uint64_t)w << s.bitcnt;
        s.bitcnt += 32;
        s.in_pos += 4;
    }
}

// puts the bit buffer at byte `pos` of the payload (the start of the stream, the byte after a stored block)
__device__ __forceinline__ void inflate_seek(InflateStream& s, int32_t pos)
{
    const int32_t skew = (int32_t)((uintptr_t)(s.in + pos) & 3u);
    s.in_pos = pos - skew;
    s.bitbuf = 0;
    s.bitcnt = 0;
    if (skew) {
        inflate_refill(s);
        if (s.bitcnt >= 8 * skew) { s.bitbuf >>= 8 * skew; s.bitcnt -= 8 * skew; }
    }
}

__device__ __forceinline__ uint32_t inflate_bits(InflateStream& s, int need)   // need <= 16
{
    inflate_refill(s);
    const uint32_t v = (uint32_t)s.bitbuf & ((1u << need) - 1u);
    s.bitbuf >>= need;
    s.bitcnt -= need;
    return v;
}

// bits of the payload consumed so far must not exceed the payload
__device__ __forceinline__ bool inflate_overran(const InflateStream& s) { return (int64_t)s.in_pos * 8 - s.bitcnt > (int64_t)s.in_len * 8; }

struct HuffmanTable {
    int16_t* count;    // [16] codes of each length
    int16_t* symbol;   // symbols in canonical code order
    uint32_t* lut;     // [1 << lut_bits], indexed by the next bits of the stream; 0 = a code longer than lut_bits (canonical walk)
    int lut_bits;
};
// What a table entry holds depends on what the code is for:
//   kPlainCode (the code-length code): (symbol << 4) | code length
//   kLengthCode (literals and lengths): bits 0-6 = how far the walk over a group's literals moves on: a literal's code length (bits
//       8-15 = the byte), 64 for everything else (which ends the walk: a group has 64 offsets).  Everything else: bits 20-23 = code
//       length (0: the code is longer than the index), bit 28 = end of block, bit 7 = a length code with its base in bits 8-16 and its
//       number of extra bits in bits 17-19 (a code length and neither bit: an invalid symbol)
//   kDistanceCode: bits 0-3 = bits to consume, bits 4-7 = extra bits, bits 8-22 = base, bit 31 = a valid distance symbol
// A length or distance needs no table of bases: the look-up is all there is to a symbol.
enum { kPlainCode = 0, kLengthCode = 1, kDistanceCode = 2 };
#ifndef PISCES_INFLATE_LEN_BITS
#define PISCES_INFLATE_LEN_BITS 10
#endif
#ifndef PISCES_INFLATE_DIST_BITS
#define PISCES_INFLATE_DIST_BITS 9
#endif
constexpr int kLenLutBits = PISCES_INFLATE_LEN_BITS, kDistLutBits = PISCES_INFLATE_DIST_BITS;

__device__ const int16_t kLenBase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
__device__ const int16_t kLenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
__device__ const int16_t kDistBase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145,
                                          8193, 12289, 16385, 24577};
__device__ const int16_t kDistExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
__device__ const uint8_t kCodeLengthOrder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

// canonical code: the codes of one length are consecutive integers, shorter codes first (RFC 1951 3.2.2).  The bit-by-bit walk over
// `bits` (first bit of the stream lowest), at most max_len of them: the symbol and its code length, or length 0 when no code ends there.
__device__ __forceinline__ int inflate_walk(const HuffmanTable& h, uint32_t bits, int max_len, int& symbol)
{
    int code = 0, first = 0, index = 0;
    for (int len = 1; len <= max_len; len++) {
        code |= (int)(bits & 1u);
        bits >>= 1;
        const int count = h.count[len];
        if (code - count < first) {
            symbol = h.symbol[index + (code - first)];
            return len;
        }
        index += count;
        first += count;
        first <<= 1;
        code <<= 1;
    }
    return 0;
}

__device__ __forceinline__ void inflate_consume(InflateStream& s, int n)
{
    s.bitbuf >>= n;
    s.bitcnt -= n;
}

// a code that is longer than the table's index: the walk, with the state kept scalar
__device__ __forceinline__ int inflate_decode_long(InflateStream& s, const HuffmanTable& h)
{
    int symbol = 0;
    const int len = __builtin_amdgcn_readfirstlane(inflate_walk(h, (uint32_t)s.bitbuf, 15, symbol));
    if (len == 0) { s.err = kInflateBadSymbol; return -1; }
    inflate_consume(s, len);
    return __builtin_amdgcn_readfirstlane(symbol);
}

// one symbol of a kPlainCode table
__device__ __forceinline__ int inflate_decode(InflateStream& s, const HuffmanTable& h)
{
    inflate_refill(s);
    const uint32_t e = (uint32_t)__builtin_amdgcn_readfirstlane((int)h.lut[(uint32_t)s.bitbuf & ((1u << h.lut_bits) - 1u)]);
    if (e) {
        inflate_consume(s, (int)(e & 15u));
        return (int)(e >> 4);
    }
    return inflate_decode_long(s, h);
}

// returns 0 for a complete code, > 0 for an incomplete one (that many codes unused), < 0 when over-subscribed.
// The canonical order (shorter codes first, the symbols of one length in symbol order) without a serial pass: every lane holds the
// lengths of symbols lane, lane + 64, ...; for each length in turn a ballot over each of those rows says which symbols have it, and a
// symbol's place is the running count plus the number of such lanes below it.
__device__ inline int inflate_construct(const InflateStream& s, HuffmanTable& h, const int16_t* length, int n, int kind)
{
    constexpr int kRows = 5;   // 5 x 64 >= 288 literal / length symbols (the most a table has)
    int v[kRows];
#pragma unroll
    for (int k = 0; k < kRows; k++) v[k] = s.lane + 64 * k < n ? (int)length[s.lane + 64 * k] : 0;
    int base = 0, left = 1;
    for (int len = 1; len <= 15; len++) {
        const int first = base;
#pragma unroll
        for (int k = 0; k < kRows; k++) {
            if (64 * k >= n) break;
            const bool mine = v[k] == len;
            const uint64_t b = __builtin_amdgcn_ballot_w64(mine);
            if (b) {
                if (mine) h.symbol[base + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0u))] = (int16_t)(s.lane + 64 * k);
                base += __builtin_popcountll(b);
            }
        }
        if (s.lane == 0) h.count[len] = (int16_t)(base - first);
        left = (left << 1) - (base - first);   // (negative stays negative: an over-subscribed code is known at the end as well)
        if (left < -(1 << 20)) left = -(1 << 20);
    }
    if (s.lane == 0) h.count[0] = (int16_t)(n - base);
    wave_lds_fence();
    const bool usable = left >= 0 && base != 0;
    const int size = 1 << h.lut_bits;
    for (int i = s.lane; i < size; i += 64) {
        uint32_t e = kind == kLengthCode ? 0x40u : 0u;
        int sym = 0;
        const int len = usable ? inflate_walk(h, (uint32_t)i, h.lut_bits, sym) : 0;
        if (len) {
            if (kind == kPlainCode) {
                e = ((uint32_t)sym << 4) | (uint32_t)len;
            } else if (kind == kDistanceCode) {
                e = (uint32_t)len;
                if (sym < 30) e |= ((uint32_t)kDistExtra[sym] << 4) | ((uint32_t)kDistBase[sym] << 8) | 0x80000000u;
            } else if (sym < 256) {
                e = ((uint32_t)sym << 8) | (uint32_t)len;   // a literal
            } else if (sym == 256) {
                e = ((uint32_t)len << 20) | 0x40u | (1u << 28);
            } else {
                e = ((uint32_t)len << 20) | 0x40u;
                if (sym <= 285) e |= 0x80u | ((uint32_t)kLenBase[sym - 257] << 8) | ((uint32_t)kLenExtra[sym - 257] << 17);
            }
        }
        h.lut[i] = e;
    }
    wave_lds_fence();
    if (base == 0) return 0;   // no codes: complete, but decoding will fail
    return left;
}

// literals and length / distance pairs until the end-of-block code (RFC 1951 3.2.3, 3.2.5).
//
// Where a symbol starts is only known from the symbol before it, but what a symbol WOULD be if it started at a given bit is not: the
// payload is taken in groups of 64 bits, and lane j of the wave decodes the symbol that would start at bit offset j of the group --
// both tables looked up at that offset (one LDS read each for all 64 offsets: the stream bits of a group are three words every lane
// shifts by its own offset), a length code's extra bits, and the distance that follows it, fetched from the lane at the offset where
// the distance code would start (ds_bpermute).  Every lane then knows where the chain would go on from it (`nxt`), and the serial
// part that is left is a walk over that one register: v_readlane at the offset the chain stands on, mark the lane, until the chain
// leaves the group (four instructions a symbol, literal or length / distance pair alike).  The marked lanes then write the group's
// output: literals with one store instruction (a byte's place is the number of marked literal lanes below it plus the lengths of the
// marked pairs below it), the pairs one after the other with all lanes copying.  The next two groups are looked up while this one is
// walked, so no table look-up waits on the dependent chain.  What the lanes cannot decode by themselves -- the end-of-block code, a
// code longer than a table's index, an invalid symbol -- stops the walk and is taken by the scalar path below, from the same registers.
struct BitGroup {
    uint32_t raw;   // the 32 stream bits from this lane's offset on
    uint32_t le;    // lencode.lut at those bits
    uint32_t de;    // distcode.lut at those bits
    uint32_t pk;    // the distance that would start here: bits 0-15 = distance, bits 16-20 = bits it takes (code + extra), bit 31 = valid
};

__device__ __forceinline__ uint64_t bitset64(uint64_t mask, int bit)   // mask | 1 << bit, both wave-uniform: one scalar instruction
{
    asm("s_bitset1_b64 %0, %1" : "+s"(mask) : "s"(bit));
    return mask;
}

__device__ __forceinline__ uint32_t group_lane(uint32_t cur, uint32_t next, int off)   // off < 128, wave-uniform
{
    return off < 64 ? (uint32_t)__builtin_amdgcn_readlane((int)cur, off) : (uint32_t)__builtin_amdgcn_readlane((int)next, off - 64);
}

constexpr int kWalkStop = 255;   // `nxt` of an offset whose symbol the lanes cannot decode

__device__ inline void inflate_codes(InflateStream& s, const HuffmanTable& lencode, const HuffmanTable& distcode)
{
    // bit positions are counted from the 16-byte boundary at or below the payload's first byte; a group is 64 of them (two words)
    const int32_t skew16 = (int32_t)((uintptr_t)s.in & 15u);
    const int32_t start = s.in_pos * 8 - s.bitcnt + 8 * skew16;
    int32_t G = start >> 6;
    int pos = start & 63;
    int32_t wb = s.win_pos + skew16;   // the window's first byte (a multiple of 16), counted from the same boundary
    const uint32_t lmask = (1u << lencode.lut_bits) - 1u, dmask = (1u << distcode.lut_bits) - 1u;
    const bool upper = s.lane >= 32;
    const uint32_t shift = (uint32_t)s.lane & 31u;

    // words [d_lo, d_hi] (counted from the boundary) are in the window afterwards
    auto ensure = [&](int32_t d_lo, int32_t d_hi) {
        if (d_lo * 4 < wb || d_hi * 4 + 4 > wb + kInWindow) {
            wb = (d_lo * 4) & ~15;
            const uint4* src = reinterpret_cast<const uint4*>(s.in - skew16 + wb);
            uint4* dst = reinterpret_cast<uint4*>(s.window);
            wave_lds_fence();
#pragma unroll
            for (int k = 0; k < kInWindow / 16 / 64; k++) dst[k * 64 + s.lane] = src[k * 64 + s.lane];
            wave_lds_fence();
        }
    };
    auto word = [&](int32_t d) { return s.window[d - (wb >> 2)]; };   // every lane the same word
    auto uniform = [](uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); };
    auto make = [&](uint32_t w0, uint32_t w1, uint32_t w2) {
        BitGroup g;
        g.raw = __builtin_amdgcn_alignbit(upper ? w2 : w1, upper ? w1 : w0, shift);
        g.le = lencode.lut[g.raw & lmask];
        g.de = distcode.lut[g.raw & dmask];
        g.pk = 0;
        return g;
    };
    // (when the group's distance entries have arrived)
    auto pack_distance = [&](BitGroup& g) {
        const uint32_t dl = g.de & 15u, deb = (g.de >> 4) & 15u;
        g.pk = (g.de & 0x80000000u) | ((dl + deb) << 16) | (((g.de >> 8) & 0x7FFFu) + ((g.raw >> dl) & ((1u << deb) - 1u)));
    };

    // the group the chain stands in (X), the two behind it (Y, Z), and the words of the one after those, asked for a group ahead
    // (uniform, but left in vector registers until they are used)
    ensure(2 * G, 2 * G + 8);
    BitGroup X, Y, Z;
    uint32_t last_w;
    {
        const uint32_t w0 = uniform(word(2 * G)), w1 = uniform(word(2 * G + 1)), w2 = uniform(word(2 * G + 2)), w3 = uniform(word(2 * G + 3)),
                       w4 = uniform(word(2 * G + 4)), w5 = uniform(word(2 * G + 5)), w6 = uniform(word(2 * G + 6));
        X = make(w0, w1, w2);
        Y = make(w2, w3, w4);
        Z = make(w4, w5, w6);
        last_w = w6;
    }
    uint32_t pre1 = word(2 * G + 7), pre2 = word(2 * G + 8);
    pack_distance(X);
    pack_distance(Y);

    // what every lane of X would decode at its offset
    uint32_t nxt, mlen, mdist;
    uint64_t lit_lanes, pair_lanes;
    auto prepare = [&]() {
        const uint32_t e = X.le;
        const bool lit = !(e & 0x40u);
        const uint32_t cl = (e >> 20) & 15u, eb = (e >> 17) & 7u;
        mlen = ((e >> 8) & 0x1FFu) + ((X.raw >> cl) & ((1u << eb) - 1u));
        const uint32_t t = (uint32_t)s.lane + cl + eb;   // where the distance code would start: in X or in Y
        const uint32_t from_x = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((t & 63u) << 2), (int)X.pk),
                       from_y = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((t & 63u) << 2), (int)Y.pk);
        const uint32_t g = t < 64u ? from_x : from_y;
        const bool pair = (e & 0x80u) && (g >> 31);
        mdist = g & 0xFFFFu;
        nxt = lit ? (uint32_t)s.lane + (e & 127u) : pair ? t + ((g >> 16) & 31u) : (uint32_t)kWalkStop;
        lit_lanes = __builtin_amdgcn_ballot_w64(lit);
        pair_lanes = __builtin_amdgcn_ballot_w64(pair);
    };
    prepare();

    // the output of the lanes the chain went through
    auto emit = [&](uint64_t chain) {
        const uint64_t lits = chain & lit_lanes, pairs = chain & pair_lanes;
        if (!(lits | pairs)) return;
        uint32_t before = 0;   // lane k: the bytes the pairs below offset k write
        int total = __builtin_popcountll(lits);
        for (uint64_t mm = pairs; mm; mm &= mm - 1) {
            const int m = __builtin_ctzll(mm);
            const uint32_t len = (uint32_t)__builtin_amdgcn_readlane((int)mlen, m);
            before += s.lane > m ? len : 0u;
            total += (int)len;
        }
        if (s.out_pos + total > s.out_len) { s.err = s.err ? s.err : kInflateOutputOverflow; return; }
        // where this lane's symbol writes: the bytes of the marked literals and pairs below it behind out_pos
        const int place = s.out_pos + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(lits >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)lits, 0u)) + (int)before;
#ifndef PISCES_INFLATE_ABLATE_LIT
        if ((lits >> s.lane) & 1u) s.out[place] = (uint8_t)(X.le >> 8);
#endif
        // a distance that reaches in front of the output (checked for all pairs at once)
        if (__builtin_amdgcn_ballot_w64(((pairs >> s.lane) & 1u) && (int)mdist > place)) { s.err = s.err ? s.err : kInflateDistanceTooFar; return; }
        // The pairs of most groups copy a few bytes each from output that older groups wrote: when no source reaches into this group's
        // own output and the pairs' bytes are 64 at most, lane k takes byte k of them all -- one load and one store for the group's
        // pairs, no loop, branch or exec mask per pair.
        const int pair_bytes = total - __builtin_popcountll(lits);
        const bool is_pair = (pairs >> s.lane) & 1u;
        if (pairs && pair_bytes <= 64 && !__builtin_amdgcn_ballot_w64(is_pair && place - (int)mdist + (int)mlen > s.out_pos)) {
            int src_at = 0, dst_at = 0, start = 0;
            for (uint64_t mm = pairs; mm; mm &= mm - 1) {
                const int m = __builtin_ctzll(mm);
                const int len = __builtin_amdgcn_readlane((int)mlen, m), dist = __builtin_amdgcn_readlane((int)mdist, m);
                const int at = __builtin_amdgcn_readlane(place, m);
                const int k = s.lane - start;
                const bool mine = (uint32_t)k < (uint32_t)len;
                dst_at = mine ? at + k : dst_at;
                src_at = mine ? at - dist + k : src_at;
                start += len;
            }
#ifndef PISCES_INFLATE_ABLATE_COPY
            // (Measured and not kept: the load here, its store when the NEXT group is emitted, so that the load's round trip lies under
            // the next group's walk — 6.81 ms against 6.09 for 268 MB, 5.59 against 4.95 for 101 MB: the wait is not what the group
            // stands on, and the value and its place carried across the loop cost more than they hide.)
            if (s.lane < pair_bytes) s.out[dst_at] = s.out[src_at];
#endif
        } else
        for (uint64_t mm = pairs; mm; mm &= mm - 1) {
            const int m = __builtin_ctzll(mm);
            const int len = __builtin_amdgcn_readlane((int)mlen, m), dist = __builtin_amdgcn_readlane((int)mdist, m);
            const int at = __builtin_amdgcn_readlane(place, m);
#ifndef PISCES_INFLATE_ABLATE_COPY
            // every source byte lies before `at`, also when the pair overlaps itself (dist < len: a run of period dist).  The bytes may
            // be this wave's own stores of a moment ago: a wave's memory instructions reach the cache in program order, no wait is needed.
            const uint8_t* src = s.out + at - dist;
            uint8_t* dst = s.out + at;
            if (dist >= len) {
                for (int k = s.lane; k < len; k += 64) dst[k] = src[k];
            } else {
                for (int k = s.lane; k < len; k += 64) dst[k] = src[k % dist];
            }
#endif
        }
        s.out_pos += total;
    };
    // the stream continues at bit `pos` of group G: hand it back to the bit buffer (block headers are read through it)
    auto leave = [&]() {
        const int32_t at = G * 64 + pos - 8 * skew16;
        s.win_pos = wb - skew16;
        if (at > s.in_len * 8) { s.err = s.err ? s.err : kInflateInputExhausted; return; }
        inflate_seek(s, at >> 3);
        (void)inflate_bits(s, at & 7);
    };

    // a symbol the lanes could not decode, at offset `at` of X (the chain so far: `chain`); true = the block is over (end-of-block code,
    // or an error)
    auto special = [&](uint64_t chain, int at) -> bool {
        pos = at;
        emit(chain & ~(1ull << pos));
        if (s.err) return true;
        const uint32_t e = (uint32_t)__builtin_amdgcn_readlane((int)X.le, pos);
        const uint32_t raw = (uint32_t)__builtin_amdgcn_readlane((int)X.raw, pos);
        int len, t;   // t: where the distance code starts
        const int cl = (int)((e >> 20) & 15u);
        if (cl) {
            if (e >> 28) {   // end of block
                pos += cl;
                leave();
                return true;
            }
            if (!(e & 0x80u)) { s.err = kInflateBadSymbol; return true; }
            const int eb = (int)((e >> 17) & 7u);
            len = (int)((e >> 8) & 0x1FFu) + (int)((raw >> cl) & ((1u << eb) - 1u));
            t = pos + cl + eb;
        } else {   // a code longer than the table's index: the canonical walk over the bits at this offset
            int symbol = 0;
            const int wl = __builtin_amdgcn_readfirstlane(inflate_walk(lencode, raw, 15, symbol));
            symbol = __builtin_amdgcn_readfirstlane(symbol);
            if (wl == 0) { s.err = kInflateBadSymbol; return true; }
            if (symbol < 256) {
                if (s.out_pos >= s.out_len) { s.err = kInflateOutputOverflow; return true; }
#ifndef PISCES_INFLATE_ABLATE_LIT
                if (s.lane == 0) s.out[s.out_pos] = (uint8_t)symbol;
#endif
                s.out_pos++;
                pos += wl;
                return false;
            }
            if (symbol == 256) {
                pos += wl;
                leave();
                return true;
            }
            symbol -= 257;
            if (symbol >= 29) { s.err = kInflateBadSymbol; return true; }
            const int eb = kLenExtra[symbol];
            len = kLenBase[symbol] + (int)((raw >> wl) & ((1u << eb) - 1u));
            t = pos + wl + eb;
        }
        const uint32_t ed = group_lane(X.de, Y.de, t), rawd = group_lane(X.raw, Y.raw, t);
        int dist;
        if (ed) {
            if (!(ed >> 31)) { s.err = kInflateBadSymbol; return true; }
            const int dl = (int)(ed & 15u), deb = (int)((ed >> 4) & 15u);
            dist = (int)((ed >> 8) & 0x7FFFu) + (int)((rawd >> dl) & ((1u << deb) - 1u));
            pos = t + dl + deb;
        } else {
            int symbol = 0;
            const int dl = __builtin_amdgcn_readfirstlane(inflate_walk(distcode, rawd, 15, symbol));
            symbol = __builtin_amdgcn_readfirstlane(symbol);
            if (dl == 0 || symbol >= 30) { s.err = kInflateBadSymbol; return true; }
            const int deb = kDistExtra[symbol];
            dist = kDistBase[symbol] + (int)((rawd >> dl) & ((1u << deb) - 1u));
            pos = t + dl + deb;
        }
        if (dist > s.out_pos) { s.err = kInflateDistanceTooFar; return true; }
        if (s.out_pos + len > s.out_len) { s.err = kInflateOutputOverflow; return true; }
#ifndef PISCES_INFLATE_ABLATE_COPY
        const uint8_t* src = s.out + s.out_pos - dist;
        uint8_t* dst = s.out + s.out_pos;
        if (dist >= len) {
            for (int k = s.lane; k < len; k += 64) dst[k] = src[k];
        } else {
            for (int k = s.lane; k < len; k += 64) dst[k] = src[k % dist];
        }
#endif
        s.out_pos += len;
        return false;
    };

    for (;;) {
        // the walk (four steps per turn of the loop: a step that leaves the group branches forward, the loop branches back once in four)
        uint64_t chain = 0;
        int at = pos;   // the offset of the last symbol the walk read
        if (pos < 64) {   // (a length / distance pair can end beyond the next group's first offsets)
            for (;;) {
#define PISCES_WALK_STEP                               \
    at = pos;                                          \
    chain = bitset64(chain, pos);                      \
    pos = __builtin_amdgcn_readlane((int)nxt, pos);    \
    if (pos >= 64) break;
                PISCES_WALK_STEP
                PISCES_WALK_STEP
                PISCES_WALK_STEP
                PISCES_WALK_STEP
#undef PISCES_WALK_STEP
            }
        }
        if (pos != kWalkStop) {
            // the chain has left the group
            emit(chain);
            if (s.err) return;
            X = Y;
            Y = Z;
            G++;
            pos -= 64;
            if (G * 8 - skew16 > s.in_len + 16) { s.err = kInflateInputExhausted; return; }   // (a valid stream ends inside its payload)
            const uint32_t w1 = uniform(pre1), w2 = uniform(pre2);
            Z = make(last_w, w1, w2);
            last_w = w2;
            pack_distance(Y);
            prepare();
            ensure(2 * G + 7, 2 * G + 8);
            pre1 = word(2 * G + 7);
            pre2 = word(2 * G + 8);
        } else if (special(chain, at)) {
            return;
        }
    }
}

// LDS workspace of a wave: the small tables in int16 units, and the two look-up tables
constexpr int kInflateTableWords = 16 + 288 + 16 + 30 + 320;
__device__ inline void inflate_stream(InflateStream& s, int16_t* tables, uint32_t* llut, uint32_t* dlut)
{
    int16_t* const lcount = tables;
    int16_t* const lsymbol = lcount + 16;
    int16_t* const dcount = lsymbol + 288;
    int16_t* const dsymbol = dcount + 16;
    int16_t* const lengths = dsymbol + 30;
    HuffmanTable lencode = {lcount, lsymbol, llut, kLenLutBits}, distcode = {dcount, dsymbol, dlut, kDistLutBits};
    int last;
    do {
        last = (int)inflate_bits(s, 1);
        const int type = (int)inflate_bits(s, 2);
        if (s.err || inflate_overran(s)) { s.err = s.err ? s.err : kInflateInputExhausted; return; }
        if (type == 0) {   // stored: to the byte boundary, LEN, ~LEN, bytes
            const int drop = s.bitcnt & 7;
            s.bitbuf >>= drop;
            s.bitcnt -= drop;
            const uint32_t len = inflate_bits(s, 16);
            const uint32_t nlen = inflate_bits(s, 16);
            if (s.err) return;
            if (len != (~nlen & 0xFFFFu)) { s.err = kInflateBadStoredLength; return; }
            const int32_t from = s.in_pos - s.bitcnt / 8;   // whole bytes are left in the bit buffer: back to the byte position
            if (from + (int32_t)len > s.in_len) { s.err = kInflateInputExhausted; return; }
            if (s.out_pos + (int32_t)len > s.out_len) { s.err = kInflateOutputOverflow; return; }
            for (int32_t k = s.lane; k < (int32_t)len; k += 64) s.out[s.out_pos + k] = s.in[from + k];
            s.out_pos += (int32_t)len;
            inflate_seek(s, from + (int32_t)len);
        } else if (type == 1) {   // fixed code (RFC 1951 3.2.6)
            for (int sym = s.lane; sym < 288; sym += 64) lengths[sym] = (int16_t)(sym < 144 ? 8 : sym < 256 ? 9 : sym < 280 ? 7 : 8);
            wave_lds_fence();
            (void)inflate_construct(s, lencode, lengths, 288, kLengthCode);
            for (int sym = s.lane; sym < 30; sym += 64) lengths[sym] = 5;
            wave_lds_fence();
            (void)inflate_construct(s, distcode, lengths, 30, kDistanceCode);
            inflate_codes(s, lencode, distcode);
        } else if (type == 2) {   // dynamic code (RFC 1951 3.2.7)
            const int nlen = (int)inflate_bits(s, 5) + 257, ndist = (int)inflate_bits(s, 5) + 1, ncode = (int)inflate_bits(s, 4) + 4;
            if (s.err) return;
            if (nlen > 286 || ndist > 30) { s.err = kInflateBadCodeLengths; return; }
            for (int index = 0; index < 19; index++) {
                const int16_t v = index < ncode ? (int16_t)inflate_bits(s, 3) : (int16_t)0;
                if (s.lane == 0) lengths[kCodeLengthOrder[index]] = v;
            }
            if (s.err) return;
            wave_lds_fence();
            if (inflate_construct(s, lencode, lengths, 19, kPlainCode) != 0) { s.err = kInflateBadCodeLengths; return; }   // the code-length code is complete
            int index = 0, prev = 0;
            while (index < nlen + ndist) {
                int symbol = inflate_decode(s, lencode);
                if (s.err) return;
                if (symbol < 16) {
                    if (s.lane == 0) lengths[index] = (int16_t)symbol;
                    index++;
                    prev = symbol;
                } else {
                    int len = 0, rep;
                    if (symbol == 16) {
                        if (index == 0) { s.err = kInflateBadCodeLengths; return; }
                        len = prev;
                        rep = 3 + (int)inflate_bits(s, 2);
                    } else if (symbol == 17) {
                        rep = 3 + (int)inflate_bits(s, 3);
                    } else {
                        rep = 11 + (int)inflate_bits(s, 7);
                    }
                    if (s.err) return;
                    if (index + rep > nlen + ndist) { s.err = kInflateBadCodeLengths; return; }
                    if (s.lane < rep) lengths[index + s.lane] = (int16_t)len;
                    if (s.lane + 64 < rep) lengths[index + s.lane + 64] = (int16_t)len;
                    if (s.lane + 128 < rep) lengths[index + s.lane + 128] = (int16_t)len;
                    index += rep;
                    prev = len;
                }
                if (inflate_overran(s)) { s.err = kInflateInputExhausted; return; }
            }
            wave_lds_fence();
            if (lengths[256] == 0) { s.err = kInflateBadCodeLengths; return; }   // no end-of-block code
            int e = inflate_construct(s, lencode, lengths, nlen, kLengthCode);
            if (e && (e < 0 || nlen != lencode.count[0] + lencode.count[1])) { s.err = kInflateBadCodeLengths; return; }   // incomplete only as a single code
            e = inflate_construct(s, distcode, lengths + nlen, ndist, kDistanceCode);
            if (e && (e < 0 || ndist != distcode.count[0] + distcode.count[1])) { s.err = kInflateBadCodeLengths; return; }
            inflate_codes(s, lencode, distcode);
        } else {
            s.err = kInflateBadBlockType;
        }
        if (s.err) return;
    } while (!last);
    if (s.out_pos != s.out_len) s.err = kInflateLengthMismatch;
}

// in: the file bytes as they are (the device copy carries a window (kInWindow + 256 bytes) of slack behind the last one); blocks: payload offset / length and
// output offset / length (ISIZE) per block.
__global__ __launch_bounds__(64) void bgzf_inflate_kernel(const uint8_t* __restrict__ in, const PiscesBgzfBlock* __restrict__ blocks, int64_t n_blocks,
                                                          uint8_t* out, int32_t* __restrict__ status)
{
    __shared__ __attribute__((aligned(16))) uint32_t window[kInWindow / 4];
    __shared__ uint32_t llut[1 << kLenLutBits], dlut[1 << kDistLutBits];
    __shared__ int16_t tables[kInflateTableWords];
    const int64_t i = (int64_t)blockIdx.x;
    if (i >= n_blocks) return;
    const PiscesBgzfBlock b = blocks[i];
    InflateStream s;
    s.in = in + b.in_offset;
    s.in_len = b.in_length;
    s.window = window;
    s.win_pos = -2 * kInWindow;
    s.out = out + b.out_offset;
    s.out_len = b.out_length;
    s.out_pos = 0;
    s.err = kInflateOk;
    s.lane = (int)threadIdx.x;
    inflate_seek(s, 0);
    inflate_stream(s, tables, llut, dlut);
    if (threadIdx.x == 0) status[i] = s.err;
}

}  // namespace pisces
