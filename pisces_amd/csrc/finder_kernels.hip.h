// finder_kernels.hip.h — ICandidateVariantFinder.FindCandidates on the device, gfx950.
//
// The walk is finder_walk.h (the same source the host entry points compile): one lane per read, every candidate the read
// gives — insertions and deletions always, SNVs and MNVs when MNV calling is on — written as a 64-byte record in read
// order, with the read bases its ALT allele takes inline (longer ones in a byte pool).  Nothing of the read stays on the
// host: pisces_hip_add_reads enqueues this next to expand_reads_kernel and picks the records up when they are needed.
//
//   find_count_kernel   MNV calling on: the number of candidates (and pool bytes) of every read
//   found_scan_kernel   exclusive scan of those counts -> the first record slot of every read
//   find_emit_kernel    the records.  MNV calling off: the host reserved one slot per I / D operation of the read's CIGAR
//                       (an upper bound that is exact unless a quality gate fails); slots a read does not use become holes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "finder_walk.h"
#include "stream_kernels.hip.h"

namespace pisces {

constexpr int kFoundInline = 32;   // read bases of an ALT allele kept inside the record
constexpr uint8_t kFoundHole = 0xFF;

struct DevFound {   // what leaves the device: 64 bytes
    FoundCandidate c;
    int32_t read;             // index of the read in its batch (records are in read order, then in order of discovery)
    int32_t pool_offset;      // the read bases in the byte pool when c.length > kFoundInline (insertions / MNVs), else -1
    uint8_t alt[kFoundInline];
};
static_assert(sizeof(DevFound) == 64, "DevFound is 64 bytes");

__device__ __forceinline__ ReadView dev_read_view(const DevReadBatch& b, const uint8_t* del_dirs, int r)
{
    ReadView v;
    const int c0 = b.cigar_offset[r], s0 = b.seq_offset[r];
    v.position = b.position[r];
    v.n_cigar = b.cigar_offset[r + 1] - c0;
    v.cigar_op = b.cigar_op + c0;
    v.cigar_len = b.cigar_len + c0;
    v.read_len = b.seq_offset[r + 1] - s0;
    v.bases = b.bases + s0;
    v.quals = b.quals + s0;
    v.dirs = b.dirs ? b.dirs + s0 : nullptr;
    v.del_dirs = del_dirs ? del_dirs + 2 * (size_t)c0 : nullptr;
    v.is_reverse = (b.flags[r] & 1) ? 1 : 0;
    return v;
}

__device__ __forceinline__ bool found_needs_pool(const FoundCandidate& c)
{
    return c.category != PISCES_CAT_DELETION && c.length > kFoundInline;
}

__global__ __launch_bounds__(256) void find_count_kernel(DevReadBatch b, const uint8_t* __restrict__ del_dirs, const uint8_t* __restrict__ ref,
                                                         int64_t ref_len, FinderParams P, int32_t* __restrict__ n_found,
                                                         int32_t* __restrict__ n_pool)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= b.n_reads) return;
    const ReadView v = dev_read_view(b, del_dirs, r);
    int n = 0, bytes = 0;
    auto count = [&](const FoundCandidate& c) {
        if (c.position <= 0) return;
        n++;
        if (found_needs_pool(c)) bytes += c.length;
    };
    walk::walk_read(v, ref, ref_len, P, count);
    n_found[r] = n;
    n_pool[r] = bytes;
}

// in-place exclusive scans of two int32 arrays by one workgroup; totals[0], totals[1] = the sums
__global__ __launch_bounds__(1024) void found_scan_kernel(int32_t* __restrict__ a, int32_t* __restrict__ b, int32_t n, long long* __restrict__ totals)
{
    __shared__ long long s_a[1024], s_b[1024];
    __shared__ long long s_base[2];
    if (threadIdx.x == 0) { s_base[0] = 0; s_base[1] = 0; }
    __syncthreads();
    for (int32_t start = 0; start < n; start += 1024) {
        const int32_t i = start + (int32_t)threadIdx.x;
        const long long va = i < n ? a[i] : 0, vb = i < n ? b[i] : 0;
        s_a[threadIdx.x] = va;
        s_b[threadIdx.x] = vb;
        __syncthreads();
        for (int d = 1; d < 1024; d <<= 1) {
            long long xa = 0, xb = 0;
            if ((int)threadIdx.x >= d) { xa = s_a[threadIdx.x - d]; xb = s_b[threadIdx.x - d]; }
            __syncthreads();
            s_a[threadIdx.x] += xa;
            s_b[threadIdx.x] += xb;
            __syncthreads();
        }
        if (i < n) {
            a[i] = (int32_t)(s_base[0] + s_a[threadIdx.x] - va);
            b[i] = (int32_t)(s_base[1] + s_b[threadIdx.x] - vb);
        }
        __syncthreads();
        if (threadIdx.x == 1023) { s_base[0] += s_a[1023]; s_base[1] += s_b[1023]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { totals[0] = s_base[0]; totals[1] = s_base[1]; }
}

// slot_first[r] = first record slot of read r, slot_end = slot_first[r + 1] (host-made or scanned); pool_first[r] = first pool byte of
// read r when the pool offsets were scanned (MNV calling on), else nullptr: long insertions then take pool bytes from a cursor
// (rare: an insertion longer than kFoundInline bases).
__global__ __launch_bounds__(256) void find_emit_kernel(DevReadBatch b, const uint8_t* __restrict__ del_dirs, const uint8_t* __restrict__ ref,
                                                        int64_t ref_len, FinderParams P, const int32_t* __restrict__ slot_first,
                                                        const int32_t* __restrict__ pool_first, DevFound* __restrict__ out,
                                                        uint8_t* __restrict__ pool, unsigned int* __restrict__ pool_cursor,
                                                        int32_t pool_capacity, int32_t* __restrict__ overflow)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= b.n_reads) return;
    const ReadView v = dev_read_view(b, del_dirs, r);
    int slot = slot_first[r];
    const int slot_end = slot_first[r + 1];
    int pool_at = pool_first ? pool_first[r] : -1;
    auto write = [&](const FoundCandidate& c) {
        if (c.position <= 0) return;   // (pisces_hip_add_reads ignores candidates before the first base of the chromosome)
        if (slot >= slot_end) { atomicExch(overflow, 1); return; }
        DevFound f;
        f.c = c;
        f.read = r;
        f.pool_offset = -1;
        const int n_alt = c.category == PISCES_CAT_DELETION ? 0 : c.length;
        const uint8_t* src = v.bases + c.start_in_read;
        if (n_alt > kFoundInline) {
            int at;
            if (pool_first) { at = pool_at; pool_at += n_alt; }
            else at = (int)atomicAdd(pool_cursor, (unsigned int)n_alt);
            if (at + n_alt <= pool_capacity) {
                for (int k = 0; k < n_alt; k++) pool[at + k] = src[k];
                f.pool_offset = at;
            } else {
                atomicExch(overflow, 1);
            }
        }
#pragma unroll
        for (int k = 0; k < kFoundInline; k++) f.alt[k] = (k < n_alt && n_alt <= kFoundInline) ? src[k] : (uint8_t)0;
        out[slot++] = f;
    };
    walk::walk_read(v, ref, ref_len, P, write);
    for (; slot < slot_end; slot++) {   // reserved, unused: a hole
        DevFound f = {};
        f.c.category = kFoundHole;
        f.read = r;
        f.pool_offset = -1;
        out[slot] = f;
    }
}

}  // namespace pisces
