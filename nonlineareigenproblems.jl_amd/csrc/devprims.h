// libnepmi355: the two device-wide primitives the one-off plan enumeration of the device LU needs (csrc/lufac.hip), written for
// 64-wide wavefronts -- an exclusive prefix sum over 64-bit items produced by a functor, and a stable least-significant-digit radix
// sort of (64-bit key, 64-bit value) pairs.  They replace hipCUB's DeviceScan::ExclusiveSum / DeviceRadixSort::SortPairs (rocPRIM
// kernels compiled into the library until round 5): nothing behind the C ABI is a vendor primitive any more.  Both are
// deterministic (fixed summation / ranking order) and sized for what the enumeration hands them (10^5 .. 10^8 items, once per
// sparsity pattern); neither is on a solve or factorisation path.
//
//   scan   reduce-then-scan in three launches: tile sums (2048 items per workgroup) -> one workgroup scans the tile sums ->
//          every tile rescans itself from its base.  Two reads of the input, one write: the decoupled-lookback single pass would
//          save one read of a 4-byte-per-item array that sits in L2 / Infinity Cache at these sizes.
//   sort   8-bit digits.  Per pass: digit histogram per tile (LDS atomics: counts are order-independent), the scan above over the
//          digit-major [256][tiles] count table = every tile's base address per digit, then the scatter: a tile walks its 2048
//          items in index order, 256 at a time; a lane's rank among the lanes of its wave that hold the same digit comes from
//          eight wave-wide ballots (one per digit bit -- gfx950 has no match_any), ranks across waves and rounds from a running
//          per-digit counter in LDS.  Equal keys keep their input order (stable), so the result does not depend on timing.
#pragma once
#include "common.h"

namespace nepprim {

constexpr int SCAN_NT = 256, SCAN_IPT = 8, SCAN_TILE = SCAN_NT * SCAN_IPT;

__device__ __forceinline__ unsigned long long shfl_up_u64(unsigned long long v, int d) {
    unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32);
    lo = __shfl_up(lo, d, 64); hi = __shfl_up(hi, d, 64);
    return ((unsigned long long)hi << 32) | lo;
}
// exclusive scan of one value per thread over a workgroup of NT threads (NT / 64 waves); *total = sum over the workgroup
template <int NT>
__device__ __forceinline__ unsigned long long block_excl_scan(unsigned long long v, unsigned long long* wsum /* NT/64 + 1 in LDS */,
                                                              unsigned long long* total) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    unsigned long long inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const unsigned long long o = shfl_up_u64(inc, d); if (lane >= d) inc += o; }
    if (lane == 63) wsum[wv] = inc;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long run = 0;
        for (int w = 0; w < NT / 64; ++w) { const unsigned long long t = wsum[w]; wsum[w] = run; run += t; }
        wsum[NT / 64] = run;
    }
    __syncthreads();
    const unsigned long long r = wsum[wv] + inc - v;
    *total = wsum[NT / 64];
    __syncthreads();                                    // wsum is reused by the caller's next round
    return r;
}

template <class In>
__global__ __launch_bounds__(SCAN_NT) void k_scan_tile_sums(In in, int64_t n, unsigned long long* __restrict__ part) {
    __shared__ unsigned long long wsum[SCAN_NT / 64 + 1];
    const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_IPT;
    unsigned long long s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_IPT; ++i) if (base + i < n) s += in(base + i);
    unsigned long long tot;
    (void)block_excl_scan<SCAN_NT>(s, wsum, &tot);
    if (threadIdx.x == 0) part[blockIdx.x] = tot;
}
// one workgroup: part[0 .. nb) -> its exclusive scan, in place
__global__ __launch_bounds__(1024) void k_scan_partials(unsigned long long* __restrict__ part, int64_t nb) {
    __shared__ unsigned long long wsum[1024 / 64 + 1];
    unsigned long long carry = 0;
    for (int64_t c = 0; c < nb; c += 1024) {
        const int64_t i = c + threadIdx.x;
        const unsigned long long v = i < nb ? part[i] : 0ull;
        unsigned long long tot;
        const unsigned long long e = block_excl_scan<1024>(v, wsum, &tot);
        if (i < nb) part[i] = carry + e;
        carry += tot;
    }
}
template <class In>
__global__ __launch_bounds__(SCAN_NT) void k_scan_tiles(In in, int64_t n, const unsigned long long* __restrict__ part,
                                                         unsigned long long* __restrict__ out) {
    __shared__ unsigned long long wsum[SCAN_NT / 64 + 1];
    const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_IPT;
    unsigned long long v[SCAN_IPT], s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_IPT; ++i) { v[i] = base + i < n ? in(base + i) : 0ull; s += v[i]; }
    unsigned long long tot;
    unsigned long long run = part[blockIdx.x] + block_excl_scan<SCAN_NT>(s, wsum, &tot);
#pragma unroll
    for (int i = 0; i < SCAN_IPT; ++i) { if (base + i < n) out[base + i] = run; run += v[i]; }
}

inline size_t scan_temp_bytes(int64_t n) { return (size_t)((n + SCAN_TILE - 1) / SCAN_TILE + 1) * sizeof(unsigned long long); }
// out[i] = sum_{j < i} in(j), i < n.  d_tmp: scan_temp_bytes(n).  Asynchronous on st.
template <class In>
inline int exclusive_sum_u64(In in, unsigned long long* out, int64_t n, void* d_tmp, hipStream_t st) {
    if (n <= 0) return NEP_OK;
    const int64_t nb = (n + SCAN_TILE - 1) / SCAN_TILE;
    unsigned long long* part = (unsigned long long*)d_tmp;
    hipLaunchKernelGGL((k_scan_tile_sums<In>), dim3((unsigned)nb), dim3(SCAN_NT), 0, st, in, n, part);
    hipLaunchKernelGGL(k_scan_partials, dim3(1), dim3(1024), 0, st, part, nb);
    hipLaunchKernelGGL((k_scan_tiles<In>), dim3((unsigned)nb), dim3(SCAN_NT), 0, st, in, n, (const unsigned long long*)part, out);
    LAUNCHCHK();
    return NEP_OK;
}

// ---- radix sort of pairs ---------------------------------------------------------------------------------------------------
constexpr int RS_NT = 256, RS_ROUNDS = 8, RS_TILE = RS_NT * RS_ROUNDS;
struct CountIn {                                         // the [256][tiles] count table as a scan input
    const uint32_t* c;
    __device__ __forceinline__ unsigned long long operator()(int64_t i) const { return c[i]; }
};

__global__ __launch_bounds__(RS_NT) void k_rs_hist(const unsigned long long* __restrict__ keys, int64_t n, int shift, int64_t nb,
                                                    uint32_t* __restrict__ counts) {
    __shared__ uint32_t h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * RS_TILE;
#pragma unroll
    for (int r = 0; r < RS_ROUNDS; ++r) {
        const int64_t i = base + (int64_t)r * RS_NT + threadIdx.x;
        if (i < n) atomicAdd(&h[(unsigned)(keys[i] >> shift) & 255u], 1u);
    }
    __syncthreads();
    counts[(int64_t)threadIdx.x * nb + blockIdx.x] = h[threadIdx.x];
}
__global__ __launch_bounds__(RS_NT) void k_rs_scatter(const unsigned long long* __restrict__ kin, const unsigned long long* __restrict__ vin,
                                                       int64_t n, int shift, int64_t nb, const unsigned long long* __restrict__ offs,
                                                       unsigned long long* __restrict__ kout, unsigned long long* __restrict__ vout) {
    __shared__ unsigned long long dbase[256];            // next free output slot of digit d for this tile
    __shared__ uint32_t whist[RS_NT / 64][256];          // items of digit d held by wave w in this round
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    dbase[threadIdx.x] = offs[(int64_t)threadIdx.x * nb + blockIdx.x];
    const int64_t base = (int64_t)blockIdx.x * RS_TILE;
    for (int r = 0; r < RS_ROUNDS; ++r) {
#pragma unroll
        for (int w = 0; w < RS_NT / 64; ++w) whist[w][threadIdx.x] = 0;
        __syncthreads();
        const int64_t i = base + (int64_t)r * RS_NT + threadIdx.x;
        const bool on = i < n;
        unsigned long long key = 0, val = 0;
        if (on) { key = kin[i]; val = vin[i]; }
        const unsigned d = (unsigned)(key >> shift) & 255u;
        unsigned long long peers = __ballot(on);         // lanes of this wave with a valid item and the same digit
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const unsigned long long m = __ballot(on && ((d >> b) & 1u));
            peers &= ((d >> b) & 1u) ? m : ~m;
        }
        const int rank = __popcll(peers & ((1ull << lane) - 1ull));
        if (on && rank == 0) whist[wv][d] = (uint32_t)__popcll(peers);
        __syncthreads();
        if (on) {
            unsigned long long pos = dbase[d] + (unsigned)rank;
            for (int w = 0; w < wv; ++w) pos += whist[w][d];
            kout[pos] = key; vout[pos] = val;
        }
        __syncthreads();
        {
            uint32_t t = 0;
#pragma unroll
            for (int w = 0; w < RS_NT / 64; ++w) t += whist[w][threadIdx.x];
            dbase[threadIdx.x] += t;
        }
        __syncthreads();
    }
}

inline size_t sort_temp_bytes(int64_t n) {
    const int64_t nb = (n + RS_TILE - 1) / RS_TILE;
    return (size_t)256 * nb * (sizeof(uint32_t) + sizeof(unsigned long long)) + scan_temp_bytes(256 * nb) + 256;
}
// Stable sort of n (key, value) pairs by the key bits [0, nbits).  (k0, v0) hold the input and are overwritten; (k1, v1) are scratch
// of the same size.  *sorted_in_0 tells where the result is: 1 = (k0, v0), 0 = (k1, v1).  d_tmp: sort_temp_bytes(n).  Asynchronous.
inline int radix_sort_pairs_u64(unsigned long long* k0, unsigned long long* v0, unsigned long long* k1, unsigned long long* v1, int64_t n,
                                int nbits, void* d_tmp, hipStream_t st, int* sorted_in_0) {
    *sorted_in_0 = 1;
    if (n <= 1 || nbits <= 0) return NEP_OK;
    const int64_t nb = (n + RS_TILE - 1) / RS_TILE;
    uint32_t* counts = (uint32_t*)d_tmp;
    unsigned long long* offs = (unsigned long long*)((char*)d_tmp + (((size_t)256 * nb * sizeof(uint32_t) + 255) & ~(size_t)255));
    void* stmp = (void*)(offs + (size_t)256 * nb);
    int in0 = 1;
    for (int shift = 0; shift < nbits; shift += 8) {
        unsigned long long* ki = in0 ? k0 : k1; unsigned long long* vi = in0 ? v0 : v1;
        unsigned long long* ko = in0 ? k1 : k0; unsigned long long* vo = in0 ? v1 : v0;
        hipLaunchKernelGGL(k_rs_hist, dim3((unsigned)nb), dim3(RS_NT), 0, st, (const unsigned long long*)ki, n, shift, nb, counts);
        int rc = exclusive_sum_u64(CountIn{counts}, offs, 256 * nb, stmp, st);
        if (rc) return rc;
        hipLaunchKernelGGL(k_rs_scatter, dim3((unsigned)nb), dim3(RS_NT), 0, st, (const unsigned long long*)ki, (const unsigned long long*)vi, n,
                           shift, nb, (const unsigned long long*)offs, ko, vo);
        LAUNCHCHK();
        in0 = !in0;
    }
    *sorted_in_0 = in0;
    return NEP_OK;
}

}  // namespace nepprim
