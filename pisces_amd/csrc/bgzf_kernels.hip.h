// bgzf_kernels.hip.h — SURVEY row f4 (upstream stage): BGZF blocks inflated on the device.
//
// A BAM file is a chain of BGZF blocks: gzip members of at most 64 KiB of payload, each an independent RFC 1951 DEFLATE stream
// (BamReader.ReadBlock, src/lib/Alignment.IO/BamReader.cs:603-645, hands each to the native zlib binding UncompressBlock,
// src/lib/Common.IO/FileCompression.cs:14-16).  Independent streams are the parallelism: one lane inflates one block, a
// launch inflates every block of a file region.  Inside a block DEFLATE is serial (a code's position depends on every code
// before it), so the lane walks it bit by bit with canonical-Huffman decoding from (count per length, symbols in code order)
// tables in private memory; stored, fixed and dynamic blocks, any number of them per stream.
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

struct InflateStream {
    const uint8_t* in;
    int32_t in_len, in_pos;
    uint32_t bitbuf;
    int32_t bitcnt;
    uint8_t* out;
    int32_t out_len, out_pos;
    int32_t err;
};

__device__ __forceinline__ uint32_t inflate_bits(InflateStream& s, int need)   // need <= 16
{
    uint32_t val = s.bitbuf;
    while (s.bitcnt < need) {
        if (s.in_pos >= s.in_len) { s.err = s.err ? s.err : kInflateInputExhausted; return 0; }
        val |= (uint32_t)s.in[s.in_pos++] << s.bitcnt;
        s.bitcnt += 8;
    }
    s.bitbuf = val >> need;
    s.bitcnt -= need;
    return val & ((1u << need) - 1u);
}

struct HuffmanTable {
    int16_t* count;    // [16] codes of each length
    int16_t* symbol;   // symbols in canonical code order
};

// canonical code: the codes of one length are consecutive integers, shorter codes first (RFC 1951 3.2.2)
__device__ __forceinline__ int inflate_decode(InflateStream& s, const HuffmanTable& h)
{
    int code = 0, first = 0, index = 0;
    for (int len = 1; len <= 15; len++) {
        code |= (int)inflate_bits(s, 1);
        if (s.err) return -1;
        const int count = h.count[len];
        if (code - count < first) return h.symbol[index + (code - first)];
        index += count;
        first += count;
        first <<= 1;
        code <<= 1;
    }
    s.err = kInflateBadSymbol;
    return -1;
}

// returns 0 for a complete code, > 0 for an incomplete one (that many codes unused), < 0 when over-subscribed
__device__ inline int inflate_construct(HuffmanTable& h, const int16_t* length, int n)
{
    int16_t offs[16];
    for (int len = 0; len <= 15; len++) h.count[len] = 0;
    for (int sym = 0; sym < n; sym++) h.count[length[sym]]++;
    if (h.count[0] == n) return 0;   // no codes: complete, but decoding will fail
    int left = 1;
    for (int len = 1; len <= 15; len++) {
        left <<= 1;
        left -= h.count[len];
        if (left < 0) return left;
    }
    offs[1] = 0;
    for (int len = 1; len < 15; len++) offs[len + 1] = (int16_t)(offs[len] + h.count[len]);
    for (int sym = 0; sym < n; sym++)
        if (length[sym] != 0) h.symbol[offs[length[sym]]++] = (int16_t)sym;
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
            s.out[s.out_pos++] = (uint8_t)symbol;
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
            // byte by byte: source and destination overlap when dist < len (a run)
            for (int k = 0; k < len; k++, s.out_pos++) s.out[s.out_pos] = s.out[s.out_pos - dist];
        }
    }
}

__device__ inline void inflate_stream(InflateStream& s)
{
    int16_t lcount[16], lsymbol[288], dcount[16], dsymbol[30], lengths[320];
    HuffmanTable lencode = {lcount, lsymbol}, distcode = {dcount, dsymbol};
    int last;
    do {
        last = (int)inflate_bits(s, 1);
        const int type = (int)inflate_bits(s, 2);
        if (s.err) return;
        if (type == 0) {   // stored: to the byte boundary, LEN, ~LEN, bytes
            s.bitbuf = 0;
            s.bitcnt = 0;
            if (s.in_pos + 4 > s.in_len) { s.err = kInflateInputExhausted; return; }
            const uint32_t len = s.in[s.in_pos] | ((uint32_t)s.in[s.in_pos + 1] << 8);
            const uint32_t nlen = s.in[s.in_pos + 2] | ((uint32_t)s.in[s.in_pos + 3] << 8);
            s.in_pos += 4;
            if (len != (~nlen & 0xFFFFu)) { s.err = kInflateBadStoredLength; return; }
            if (s.in_pos + (int32_t)len > s.in_len) { s.err = kInflateInputExhausted; return; }
            if (s.out_pos + (int32_t)len > s.out_len) { s.err = kInflateOutputOverflow; return; }
            for (uint32_t k = 0; k < len; k++) s.out[s.out_pos++] = s.in[s.in_pos++];
        } else if (type == 1) {   // fixed code (RFC 1951 3.2.6)
            int sym = 0;
            for (; sym < 144; sym++) lengths[sym] = 8;
            for (; sym < 256; sym++) lengths[sym] = 9;
            for (; sym < 280; sym++) lengths[sym] = 7;
            for (; sym < 288; sym++) lengths[sym] = 8;
            (void)inflate_construct(lencode, lengths, 288);
            for (sym = 0; sym < 30; sym++) lengths[sym] = 5;
            (void)inflate_construct(distcode, lengths, 30);
            inflate_codes(s, lencode, distcode);
        } else if (type == 2) {   // dynamic code (RFC 1951 3.2.7)
            const int nlen = (int)inflate_bits(s, 5) + 257, ndist = (int)inflate_bits(s, 5) + 1, ncode = (int)inflate_bits(s, 4) + 4;
            if (s.err) return;
            if (nlen > 286 || ndist > 30) { s.err = kInflateBadCodeLengths; return; }
            int index = 0;
            for (; index < ncode; index++) lengths[kCodeLengthOrder[index]] = (int16_t)inflate_bits(s, 3);
            for (; index < 19; index++) lengths[kCodeLengthOrder[index]] = 0;
            if (s.err) return;
            if (inflate_construct(lencode, lengths, 19) != 0) { s.err = kInflateBadCodeLengths; return; }   // the code-length code is complete
            index = 0;
            while (index < nlen + ndist) {
                int symbol = inflate_decode(s, lencode);
                if (s.err) return;
                if (symbol < 16) {
                    lengths[index++] = (int16_t)symbol;
                } else {
                    int len = 0, rep;
                    if (symbol == 16) {
                        if (index == 0) { s.err = kInflateBadCodeLengths; return; }
                        len = lengths[index - 1];
                        rep = 3 + (int)inflate_bits(s, 2);
                    } else if (symbol == 17) {
                        rep = 3 + (int)inflate_bits(s, 3);
                    } else {
                        rep = 11 + (int)inflate_bits(s, 7);
                    }
                    if (s.err) return;
                    if (index + rep > nlen + ndist) { s.err = kInflateBadCodeLengths; return; }
                    while (rep--) lengths[index++] = (int16_t)len;
                }
            }
            if (lengths[256] == 0) { s.err = kInflateBadCodeLengths; return; }   // no end-of-block code
            int e = inflate_construct(lencode, lengths, nlen);
            if (e && (e < 0 || nlen != lencode.count[0] + lencode.count[1])) { s.err = kInflateBadCodeLengths; return; }   // incomplete only as a single code
            e = inflate_construct(distcode, lengths + nlen, ndist);
            if (e && (e < 0 || ndist != distcode.count[0] + distcode.count[1])) { s.err = kInflateBadCodeLengths; return; }
            inflate_codes(s, lencode, distcode);
        } else {
            s.err = kInflateBadBlockType;
        }
        if (s.err) return;
    } while (!last);
    if (s.out_pos != s.out_len) s.err = kInflateLengthMismatch;
}

// lane = BGZF block.  in: the file bytes as they are; blocks: payload offset / length and output offset / length (ISIZE) per block.
__global__ __launch_bounds__(64) void bgzf_inflate_kernel(const uint8_t* __restrict__ in, const PiscesBgzfBlock* __restrict__ blocks, int64_t n_blocks,
                                                          uint8_t* __restrict__ out, int32_t* __restrict__ status)
{
    const int64_t i = (int64_t)blockIdx.x * 64 + threadIdx.x;
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
    inflate_stream(s);
    status[i] = s.err;
}

}  // namespace pisces
