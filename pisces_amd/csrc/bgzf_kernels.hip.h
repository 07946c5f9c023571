// bgzf_kernels.hip.h — SURVEY row f4 (upstream stage): BGZF blocks inflated on the device.
//
// A BAM file is a chain of BGZF blocks: gzip members of at most 64 KiB of payload, each an independent RFC 1951 DEFLATE stream
// (BamReader.ReadBlock, src/lib/Alignment.IO/BamReader.cs:603-645, hands each to the native zlib binding UncompressBlock,
// src/lib/Common.IO/FileCompression.cs:14-16).  Independent streams are the parallelism: one wave inflates one block, a
// launch inflates every block of a file region.  Inside a block DEFLATE is serial (a code's position depends on every code
// before it), so one lane walks it bit by bit with canonical-Huffman decoding from (count per length, symbols in code order)
// tables in LDS; stored, fixed and dynamic blocks, any number of them per stream.
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

// ---- one wave per BGZF block, every lane carrying the same decoder state ----
// DEFLATE is serial inside a stream, so the 64 lanes of the wave run the SAME decode (identical registers, broadcast LDS reads: no
// divergence, no extra cost) and split what can be split: a length / distance pair is copied by all lanes at once, the decode tables
// are filled by all lanes, and a literal is stored by lane 0.  Symbols are looked up in a 10-bit table in LDS (code, length) with the
// canonical bit-by-bit walk as the fall-back for longer codes.  Output goes straight to HBM; a copy waits for the wave's outstanding
// stores first (its source bytes may be among them).  LDS per wave is 5 KB, so a CU holds as many waves as it has slots for and a
// file's thousands of blocks are all in flight: the serial chains hide one another's latency.
constexpr int kLutBits = 10;
constexpr int kLutSize = 1 << kLutBits;

struct InflateStream {
    const uint8_t* in;
    int32_t in_len, in_pos;   // in_pos: next byte to load into the bit buffer
    uint64_t bitbuf;
    int32_t bitcnt;
    uint8_t* out;
    int32_t out_len, out_pos;
    int32_t err;
    int lane;
};

__device__ __forceinline__ void inflate_refill(InflateStream& s)
{
    if (s.bitcnt <= 32) {
        // the file bytes continue past the payload (trailer, next header; the device copy is padded), so the load itself is always
        // in bounds; consuming bits that lie past the payload is the error
        if (s.in_pos >= s.in_len + 8) { s.err = s.err ? s.err : kInflateInputExhausted; return; }
        const uint8_t* p = s.in + s.in_pos;
        // every lane loads the same word; telling the compiler so keeps the bit buffer and all that follows from it (symbols, lengths,
        // the branches on them) in scalar registers and scalar branches instead of 64-wide copies under exec masks
        const uint32_t w = (uint32_t)__builtin_amdgcn_readfirstlane(
            (int)((uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24)));
        s.bitbuf |= (uint64_t)w << s.bitcnt;
        s.bitcnt += 32;
        s.in_pos += 4;
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
    uint16_t* lut;     // [kLutSize] (symbol << 4) | length for codes of <= kLutBits bits, 0 = longer code
};

__device__ __forceinline__ void wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __builtin_amdgcn_wave_barrier();
}

// canonical code: the codes of one length are consecutive integers, shorter codes first (RFC 1951 3.2.2)
__device__ __forceinline__ int inflate_decode(InflateStream& s, const HuffmanTable& h)
{
    inflate_refill(s);
    const uint32_t e = (uint32_t)__builtin_amdgcn_readfirstlane((int)h.lut[(uint32_t)s.bitbuf & (kLutSize - 1)]);
    if (e) {
        const int n = (int)(e & 15u);
        s.bitbuf >>= n;
        s.bitcnt -= n;
        return (int)(e >> 4);
    }
    int code = 0, first = 0, index = 0;
    uint32_t bits = (uint32_t)s.bitbuf;
    for (int len = 1; len <= 15; len++) {
        code |= (int)(bits & 1u);
        bits >>= 1;
        const int count = __builtin_amdgcn_readfirstlane((int)h.count[len]);
        if (code - count < first) {
            s.bitbuf >>= len;
            s.bitcnt -= len;
            return __builtin_amdgcn_readfirstlane((int)h.symbol[index + (code - first)]);
        }
        index += count;
        first += count;
        first <<= 1;
        code <<= 1;
    }
    s.err = kInflateBadSymbol;
    return -1;
}

// returns 0 for a complete code, > 0 for an incomplete one (that many codes unused), < 0 when over-subscribed.  Lane 0 counts and
// sorts (serial), all lanes fill the look-up table.
__device__ inline int inflate_construct(const InflateStream& s, HuffmanTable& h, const int16_t* length, int n, int16_t* offs /* [16] LDS */)
{
    if (s.lane == 0) {
        for (int len = 0; len <= 15; len++) h.count[len] = 0;
        for (int sym = 0; sym < n; sym++) h.count[length[sym]]++;
        offs[1] = 0;
        for (int len = 1; len < 15; len++) offs[len + 1] = (int16_t)(offs[len] + h.count[len]);
        for (int sym = 0; sym < n; sym++)
            if (length[sym] != 0) h.symbol[offs[length[sym]]++] = (int16_t)sym;   // offs[len] ends as the END of that length's run
    }
    wave_lds_fence();
    int left = 1, total = 0;
    for (int len = 1; len <= 15; len++) {
        left <<= 1;
        left -= h.count[len];
        total += h.count[len];
        if (left < 0) break;
    }
    for (int k = s.lane; k < kLutSize; k += 64) h.lut[k] = 0;
    wave_lds_fence();
    if (left >= 0 && h.count[0] != n) {
        // entry i of symbol[] has length len where run_begin(len) <= i < run_end(len), code = first_code(len) + i - run_begin(len)
        for (int i = s.lane; i < total; i += 64) {
            int len = 1, begin = 0, code0 = 0;
            while (len <= 15 && i >= begin + h.count[len]) { code0 = (code0 + h.count[len]) << 1; begin += h.count[len]; len++; }
            if (len <= kLutBits) {
                const uint32_t code = (uint32_t)(code0 + (i - begin));
                const uint32_t rev = __brev(code) >> (32 - len);   // the stream carries a code's most significant bit first
                const uint16_t e = (uint16_t)(((uint32_t)h.symbol[i] << 4) | (uint32_t)len);
                for (uint32_t k = rev; k < (uint32_t)kLutSize; k += 1u << len) h.lut[k] = e;
            }
        }
    }
    wave_lds_fence();
    if (h.count[0] == n) return 0;   // no codes: complete, but decoding will fail
    return left;
}

__device__ const int16_t kLenBase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
__device__ const int16_t kLenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
__device__ const int16_t kDistBase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145,
                                          8193, 12289, 16385, 24577};
__device__ const int16_t kDistExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
__device__ const uint8_t kCodeLengthOrder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

// literals and length / distance pairs until the end-of-block code (RFC 1951 3.2.3, 3.2.5)
__device__ inline void inflate_codes(InflateStream& s, const HuffmanTable& lencode, const HuffmanTable& distcode)
{
    for (;;) {
        int symbol = inflate_decode(s, lencode);
        if (s.err) return;
        if (symbol < 256) {
            if (s.out_pos >= s.out_len) { s.err = kInflateOutputOverflow; return; }
            if (s.lane == 0) s.out[s.out_pos] = (uint8_t)symbol;
            s.out_pos++;
        } else if (symbol == 256) {
            return;
        } else {
            symbol -= 257;
            if (symbol >= 29) { s.err = kInflateBadSymbol; return; }
            const int len = kLenBase[symbol] + (int)inflate_bits(s, kLenExtra[symbol]);
            symbol = inflate_decode(s, distcode);
            if (s.err) return;
            if (symbol >= 30) { s.err = kInflateBadSymbol; return; }
            const int dist = kDistBase[symbol] + (int)inflate_bits(s, kDistExtra[symbol]);
            if (s.err) return;
            if (dist > s.out_pos) { s.err = kInflateDistanceTooFar; return; }
            if (s.out_pos + len > s.out_len) { s.err = kInflateOutputOverflow; return; }
            // the source bytes may still be on their way to memory (this wave's own earlier stores)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            // every source byte lies before out_pos, also when the pair overlaps itself (dist < len: a run of period dist)
            const uint8_t* src = s.out + s.out_pos - dist;
            uint8_t* dst = s.out + s.out_pos;
            for (int k = s.lane; k < len; k += 64) dst[k] = src[dist >= len ? k : k % dist];
            s.out_pos += len;
        }
        if (inflate_overran(s)) { s.err = kInflateInputExhausted; return; }
    }
}

// LDS workspace of a wave, in int16 units
constexpr int kInflateTableWords = 16 + 288 + 16 + 30 + 320 + 16 + 2 * kLutSize;
__device__ inline void inflate_stream(InflateStream& s, int16_t* tables)
{
    int16_t* const lcount = tables;
    int16_t* const lsymbol = lcount + 16;
    int16_t* const dcount = lsymbol + 288;
    int16_t* const dsymbol = dcount + 16;
    int16_t* const lengths = dsymbol + 30;
    int16_t* const offs = lengths + 320;
    uint16_t* const llut = (uint16_t*)(offs + 16);
    uint16_t* const dlut = llut + kLutSize;
    HuffmanTable lencode = {lcount, lsymbol, llut}, distcode = {dcount, dsymbol, dlut};
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
            s.in_pos = from + (int32_t)len;
            s.bitbuf = 0;
            s.bitcnt = 0;
        } else if (type == 1) {   // fixed code (RFC 1951 3.2.6)
            for (int sym = s.lane; sym < 288; sym += 64) lengths[sym] = (int16_t)(sym < 144 ? 8 : sym < 256 ? 9 : sym < 280 ? 7 : 8);
            wave_lds_fence();
            (void)inflate_construct(s, lencode, lengths, 288, offs);
            for (int sym = s.lane; sym < 30; sym += 64) lengths[sym] = 5;
            wave_lds_fence();
            (void)inflate_construct(s, distcode, lengths, 30, offs);
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
            if (inflate_construct(s, lencode, lengths, 19, offs) != 0) { s.err = kInflateBadCodeLengths; return; }   // the code-length code is complete
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
            int e = inflate_construct(s, lencode, lengths, nlen, offs);
            if (e && (e < 0 || nlen != lencode.count[0] + lencode.count[1])) { s.err = kInflateBadCodeLengths; return; }   // incomplete only as a single code
            e = inflate_construct(s, distcode, lengths + nlen, ndist, offs);
            if (e && (e < 0 || ndist != distcode.count[0] + distcode.count[1])) { s.err = kInflateBadCodeLengths; return; }
            inflate_codes(s, lencode, distcode);
        } else {
            s.err = kInflateBadBlockType;
        }
        if (s.err) return;
    } while (!last);
    if (s.out_pos != s.out_len) s.err = kInflateLengthMismatch;
}

// in: the file bytes as they are (the device copy carries 16 spare bytes behind the last one); blocks: payload offset / length and
// output offset / length (ISIZE) per block.
__global__ __launch_bounds__(64) void bgzf_inflate_kernel(const uint8_t* __restrict__ in, const PiscesBgzfBlock* __restrict__ blocks, int64_t n_blocks,
                                                          uint8_t* out, int32_t* __restrict__ status)
{
    __shared__ int16_t tables[kInflateTableWords];
    const int64_t i = (int64_t)blockIdx.x;
    if (i >= n_blocks) return;
    const PiscesBgzfBlock b = blocks[i];
    InflateStream s;
    s.in = in + b.in_offset;
    s.in_len = b.in_length;
    s.in_pos = 0;
    s.bitbuf = 0;
    s.bitcnt = 0;
    s.out = out + b.out_offset;
    s.out_len = b.out_length;
    s.out_pos = 0;
    s.err = kInflateOk;
    s.lane = (int)threadIdx.x;
    inflate_stream(s, tables);
    if (threadIdx.x == 0) status[i] = s.err;
}

}  // namespace pisces
