// K7 design microbenchmark (round 3): variants of the B-resident tall-skinny complex GEMM  Y = Z * B  on v_mfma_f64_16x16x4_f64,
// timed with HIP events on the waveguide shape (rows = 1 003 995, k = p = 60) and checked against a plain kernel.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/ub_k7 scripts/ub_k7.hip && /tmp/ub_k7 [rows k p reps]
// Variants (what the numbers in DESIGN.md section 3 K7 refer to):
//   base   the library's resident kernel of round 2 (ping-pong rings of 4 k-steps, k-steps padded to a multiple of 8)
//   v3     k-steps as a template parameter (no padding), ONE register buffer of a whole strip refilled k-step by k-step for the
//          wave's next strip (prefetch distance = a whole strip), fragments of the next k-step read from LDS ahead of the MFMAs
//   flags  S16: 16-byte paired stores (row-major Y); NOPF: no fragment prefetch
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cmath>
#include <cstring>
#include <vector>
#include <string>

typedef double2 cplx;
typedef double d4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__device__ __forceinline__ double shfl_xor_d(double v, int m) { return __shfl_xor(v, m, 64); }

__global__ void k_fill(cplx* Z, int64_t n, uint64_t seed) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        uint64_t h = (uint64_t)i * 0x9E3779B97F4A7C15ull + seed;
        h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
        uint64_t g = h * 0x94D049BB133111EBull; g ^= g >> 31;
        // full-entropy mantissas (the matrix pipe's power draw, and with it the sustained clock, depends on the operand bits:
        // 16-bit fractions ran 20 % faster than these); NEP_UB_LOWENT=1 restores the low-entropy fill for that comparison
        if (seed & (1ull << 63)) Z[i] = make_double2((double)(h & 0xffff) / 65536.0 - 0.5, (double)((h >> 16) & 0xffff) / 65536.0 - 0.5);
        else if (seed & (1ull << 62)) {      // standard normal (Box-Muller), what torch.randn fills the library benchmarks with
            const double u1 = ((double)(h >> 11) + 1.0) * (1.0 / 9007199254740993.0), u2 = (double)(g >> 11) * (1.0 / 9007199254740992.0);
            const double r = sqrt(-2.0 * log(u1));
            Z[i] = make_double2(r * cos(6.283185307179586 * u2), r * sin(6.283185307179586 * u2));
        }
        else Z[i] = make_double2((double)(h >> 11) * (1.0 / 9007199254740992.0) - 0.5, (double)(g >> 11) * (1.0 / 9007199254740992.0) - 0.5);
    }
}

__global__ void k_expand_B(const cplx* __restrict__ B, int64_t ldb, int k, int pp, int nks, int nt, double* __restrict__ frag) {
    const int64_t total = (int64_t)nks * nt * 64;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int l = (int)(i & 63);
        const int64_t kt = i >> 6;
        const int t = (int)(kt % nt), ks = (int)(kt / nt);
        const int q = l >> 4, n = l & 15;
        const int c = 4 * ks + q, jc = 8 * t + (n >> 1);
        double b0 = 0.0, b1 = 0.0;
        if (c < k && jc < pp) {
            const cplx b = B[(int64_t)jc * ldb + c];
            if ((n & 1) == 0) { b0 = b.x; b1 = -b.y; } else { b0 = b.y; b1 = b.x; }
        }
        double* f = frag + kt * 128;
        f[l] = b0;
        f[64 + l] = b1;
    }
}

// plain check kernel: one thread per (row, column)
__global__ void k_check(const cplx* Z, int64_t ldz, int64_t rows, int k, const cplx* B, int p, const cplx* Y, int64_t ldy, int rm,
                        unsigned long long* maxerr) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= rows * p) return;
    const int64_t r = i / p; const int j = (int)(i % p);
    double sr = 0, si = 0;
    for (int c = 0; c < k; ++c) {
        const cplx z = Z[(int64_t)c * ldz + r], b = B[(int64_t)j * k + c];
        sr += z.x * b.x - z.y * b.y; si += z.x * b.y + z.y * b.x;
    }
    const cplx y = rm ? Y[r * ldy + j] : Y[(int64_t)j * ldy + r];
    const double e = fmax(fabs(y.x - sr), fabs(y.y - si));
    atomicMax(maxerr, (unsigned long long)__double_as_longlong(e));
}

// ---------------------------------------------------------------------------------------------- base (round 2)
template <int NT, bool ROWMAJOR>
__global__ __launch_bounds__(512) void k_base(const cplx* __restrict__ Z, int64_t ldz, int64_t rows, int k,
                                              const double* __restrict__ Bfrag, int nks, int p, cplx* __restrict__ Y, int64_t ldy) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int PER_KS = NT * 2 * 64;
    double* bs = (double*)smem_raw;
    {
        const double2* src = (const double2*)Bfrag;
        double2* dst = (double2*)bs;
        const int n2 = nks * PER_KS / 2;
        for (int t = threadIdx.x; t < n2; t += 512) dst[t] = src[t];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const int m = lane & 15, q = lane >> 4;
    const int64_t nstrips = (rows + 15) / 16;
    const int64_t stride = (int64_t)gridDim.x * 8;
    constexpr int RES_D = 4;
    cplx ra[RES_D], rb[RES_D];
    int64_t strip = blockIdx.x * 8LL + wv;
    auto zload = [&](int ks, int64_t arow, int64_t nrow) -> cplx {
        const bool same = ks < nks;
        int col = 4 * (same ? ks : ks - nks) + q;
        if (col >= k) col = k - 1;
        return Z[(int64_t)col * ldz + (same ? arow : nrow)];
    };
    {
        int64_t arow = (strip < nstrips ? strip : nstrips - 1) * 16 + m;
        if (arow >= rows) arow = rows - 1;
#pragma unroll
        for (int j = 0; j < RES_D; ++j) ra[j] = zload(j, arow, arow);
    }
    for (; strip < nstrips; strip += stride) {
        const int64_t next = strip + stride < nstrips ? strip + stride : strip;
        int64_t arow = strip * 16 + m, nrow = next * 16 + m;
        if (arow >= rows) arow = rows - 1;
        if (nrow >= rows) nrow = rows - 1;
        d4 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = (d4){0.0, 0.0, 0.0, 0.0};
        for (int kb = 0; kb < nks; kb += 2 * RES_D) {
#pragma unroll
            for (int j = 0; j < RES_D; ++j) rb[j] = zload(kb + RES_D + j, arow, nrow);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < RES_D; ++j) {
                const double* bk = bs + (size_t)(kb + j) * PER_KS + lane;
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(ra[j].x, bk[(t * 2 + 0) * 64], acc[t], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(ra[j].y, bk[(t * 2 + 1) * 64], acc[t], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < RES_D; ++j) ra[j] = zload(kb + 2 * RES_D + j, arow, nrow);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < RES_D; ++j) {
                const double* bk = bs + (size_t)(kb + RES_D + j) * PER_KS + lane;
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(rb[j].x, bk[(t * 2 + 0) * 64], acc[t], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(rb[j].y, bk[(t * 2 + 1) * 64], acc[t], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        const int64_t row0 = strip * 16;
        const int n = lane & 15, g = lane >> 4;
        if (ROWMAJOR) {
            double* Yd = (double*)Y;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int jc = 8 * t + (n >> 1);
                if (jc < p) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int64_t r = row0 + g + 4 * i;
                        if (r < rows) Yd[(r * ldy + jc) * 2 + (n & 1)] = acc[t][i];
                    }
                }
            }
        } else {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int jc = 8 * t + (n >> 1);
                const bool odd = n & 1;
                const double s0 = odd ? acc[t][0] : acc[t][1];
                const double s1 = odd ? acc[t][2] : acc[t][3];
                const double r0 = shfl_xor_d(s0, 1);
                const double r1 = shfl_xor_d(s1, 1);
                cplx v0, v1;
                int i0, i1;
                if (!odd) { v0 = make_double2(acc[t][0], r0); v1 = make_double2(acc[t][2], r1); i0 = 0; i1 = 2; }
                else      { v0 = make_double2(r0, acc[t][1]); v1 = make_double2(r1, acc[t][3]); i0 = 1; i1 = 3; }
                if (jc < p) {
                    const int64_t ra_ = row0 + g + 4 * i0, rb_ = row0 + g + 4 * i1;
                    cplx* col = Y + (int64_t)jc * ldy;
                    if (ra_ < rows) col[ra_] = v0;
                    if (rb_ < rows) col[rb_] = v1;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------- base_exact: the round-2 ring kernel, padded k-steps skipped by wave-uniform branches
template <int NT, bool ROWMAJOR>
__global__ __launch_bounds__(512) void k_exact(const cplx* __restrict__ Z, int64_t ldz, int64_t rows, int k,
                                              const double* __restrict__ Bfrag, int nks, int nks_true, int p, cplx* __restrict__ Y, int64_t ldy) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int PER_KS = NT * 2 * 64;
    double* bs = (double*)smem_raw;
    {
        const double2* src = (const double2*)Bfrag;
        double2* dst = (double2*)bs;
        const int n2 = nks_true * PER_KS / 2;
        for (int t = threadIdx.x; t < n2; t += 512) dst[t] = src[t];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const int m = lane & 15, q = lane >> 4;
    const int64_t nstrips = (rows + 15) / 16;
    const int64_t stride = (int64_t)gridDim.x * 8;
    constexpr int RES_D = 4;
    cplx ra[RES_D], rb[RES_D];
    int64_t strip = blockIdx.x * 8LL + wv;
    auto zload = [&](int ks, int64_t arow, int64_t nrow) -> cplx {
        const bool same = ks < nks;
        int col = 4 * (same ? ks : ks - nks) + q;
        if (col >= k) col = k - 1;
        return Z[(int64_t)col * ldz + (same ? arow : nrow)];
    };
    {
        int64_t arow = (strip < nstrips ? strip : nstrips - 1) * 16 + m;
        if (arow >= rows) arow = rows - 1;
#pragma unroll
        for (int j = 0; j < RES_D; ++j) ra[j] = zload(j, arow, arow);
    }
    for (; strip < nstrips; strip += stride) {
        const int64_t next = strip + stride < nstrips ? strip + stride : strip;
        int64_t arow = strip * 16 + m, nrow = next * 16 + m;
        if (arow >= rows) arow = rows - 1;
        if (nrow >= rows) nrow = rows - 1;
        d4 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = (d4){0.0, 0.0, 0.0, 0.0};
        for (int kb = 0; kb < nks; kb += 2 * RES_D) {
#pragma unroll
            for (int j = 0; j < RES_D; ++j) rb[j] = zload(kb + RES_D + j, arow, nrow);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < RES_D; ++j) {
                if (kb + j < nks_true) {
                const double* bk = bs + (size_t)(kb + j) * PER_KS + lane;
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(ra[j].x, bk[(t * 2 + 0) * 64], acc[t], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(ra[j].y, bk[(t * 2 + 1) * 64], acc[t], 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < RES_D; ++j) ra[j] = zload(kb + 2 * RES_D + j, arow, nrow);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < RES_D; ++j) {
                if (kb + RES_D + j < nks_true) {
                const double* bk = bs + (size_t)(kb + RES_D + j) * PER_KS + lane;
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(rb[j].x, bk[(t * 2 + 0) * 64], acc[t], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(rb[j].y, bk[(t * 2 + 1) * 64], acc[t], 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        const int64_t row0 = strip * 16;
        const int n = lane & 15, g = lane >> 4;
        if (ROWMAJOR) {
            double* Yd = (double*)Y;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int jc = 8 * t + (n >> 1);
                if (jc < p) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int64_t r = row0 + g + 4 * i;
                        if (r < rows) Yd[(r * ldy + jc) * 2 + (n & 1)] = acc[t][i];
                    }
                }
            }
        } else {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int jc = 8 * t + (n >> 1);
                const bool odd = n & 1;
                const double s0 = odd ? acc[t][0] : acc[t][1];
                const double s1 = odd ? acc[t][2] : acc[t][3];
                const double r0 = shfl_xor_d(s0, 1);
                const double r1 = shfl_xor_d(s1, 1);
                cplx v0, v1;
                int i0, i1;
                if (!odd) { v0 = make_double2(acc[t][0], r0); v1 = make_double2(acc[t][2], r1); i0 = 0; i1 = 2; }
                else      { v0 = make_double2(r0, acc[t][1]); v1 = make_double2(r1, acc[t][3]); i0 = 1; i1 = 3; }
                if (jc < p) {
                    const int64_t ra_ = row0 + g + 4 * i0, rb_ = row0 + g + 4 * i1;
                    cplx* col = Y + (int64_t)jc * ldy;
                    if (ra_ < rows) col[ra_] = v0;
                    if (rb_ < rows) col[rb_] = v1;
                }
            }
        }
    }
}

__device__ __forceinline__ cplx ntload(const cplx* p) {
    const double* d = (const double*)p;
    typedef double d2v __attribute__((ext_vector_type(2)));
    const d2v v = __builtin_nontemporal_load((const d2v*)d);
    return make_double2(v.x, v.y);
}
__device__ __forceinline__ void ntstore(cplx* p, cplx v) {
    typedef double d2v __attribute__((ext_vector_type(2)));
    d2v w; w.x = v.x; w.y = v.y;
    __builtin_nontemporal_store(w, (d2v*)p);
}
__device__ unsigned long long g_clk[2 * 1024];
__device__ unsigned long long g_wave[4 * 8 * 1024];
__device__ unsigned long long g_span[2];       // absolute 100 MHz ticks: min wave entry, max wave exit of the last launch  // per wave: loop start, loop end (100 MHz ticks since kernel start of WG), strips     // per workgroup: shader-clock ticks and 100 MHz ticks of wave 0's strip loop

// ---------------------------------------------------------------------------------------------- v3
// FLAGS bit 0: 16-byte paired stores (row-major Y)   bit 1: no fragment prefetch   bit 2: skip Z loads (ablation)
// bit 3: skip Y stores (ablation)
template <int NT, int NKS, bool ROWMAJOR, int FLAGS>
__global__ __launch_bounds__(512) void k_v3(const cplx* __restrict__ Z, int64_t ldz, int64_t rows, int k,
                                            const double* __restrict__ Bfrag, int p, cplx* __restrict__ Y, int64_t ldy) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int PER_KS = NT * 2 * 64;
    const unsigned long long wstart = wall_clock64();
    constexpr bool S16 = FLAGS & 1, NOPF = FLAGS & 2, NOLOAD = FLAGS & 4, NOSTORE = FLAGS & 8, NOPRED = FLAGS & 16, NTST = FLAGS & 32, NTLD = FLAGS & 64;
    double* bs = (double*)smem_raw;
    {
        const double2* src = (const double2*)Bfrag;
        double2* dst = (double2*)bs;
        constexpr int n2 = NKS * PER_KS / 2;
        for (int t = threadIdx.x; t < n2; t += 512) dst[t] = src[t];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const int m = lane & 15, q = lane >> 4;
    const int64_t nstrips = (rows + 15) / 16;
    const int64_t stride = (int64_t)gridDim.x * 8;
    int64_t strip = blockIdx.x * 8LL + wv;
    if (strip >= nstrips) return;
    // per-lane column offsets (in elements) of the NKS loads of a strip; the last k-step may run past k: clamp (zero B rows)
    cplx z[NKS];
    const cplx* zq = Z + (int64_t)q * ldz;                   // column q, advanced by 4*ks*ldz per k-step
    auto colptr = [&](int ks) -> const cplx* {
        if (4 * ks + 3 < 4 * NKS - 3 || true) {
            int col = 4 * ks + q;
            if (4 * ks + 3 >= k) col = col < k ? col : k - 1;   // compile-time ks: only the tail k-step pays the clamp when k % 4
            return Z + (int64_t)col * ldz;
        }
        return zq;
    };
    {
        int64_t arow = strip * 16 + m;
        if (arow >= rows) arow = rows - 1;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) { z[ks] = NTLD ? ntload(colptr(ks) + arow) : colptr(ks)[arow]; __builtin_amdgcn_sched_barrier(0); }   // in k-step order
    }
    const double* bl = bs + lane;
    // the body of one strip; the first strip is peeled off the loop so that the loop header sees the same queue of outstanding
    // loads and stores from both of its predecessors (the compiler's s_waitcnt counts are exact then, not the minimum of two paths)
    auto body = [&](int64_t strip) __attribute__((always_inline)) {
        const int64_t next = strip + stride < nstrips ? strip + stride : strip;
        int64_t nrow = next * 16 + m;
        if (nrow >= rows) nrow = rows - 1;
        d4 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = (d4){0.0, 0.0, 0.0, 0.0};
        double f[2][2 * NT];
        if (!NOPF) {
#pragma unroll
            for (int i = 0; i < 2 * NT; ++i) f[0][i] = bl[i * 64];
        }
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const cplx a = z[ks];
            if (NOPF) {
#pragma unroll
                for (int i = 0; i < 2 * NT; ++i) f[ks & 1][i] = bl[ks * PER_KS + i * 64];
            } else if (ks + 1 < NKS) {
#pragma unroll
                for (int i = 0; i < 2 * NT; ++i) f[(ks + 1) & 1][i] = bl[(ks + 1) * PER_KS + i * 64];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a.x, f[ks & 1][t * 2 + 0], acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a.y, f[ks & 1][t * 2 + 1], acc[t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (!NOLOAD) z[ks] = NTLD ? ntload(colptr(ks) + nrow) : colptr(ks)[nrow];            // the same k-step of the wave's next strip: a whole strip ahead
            __builtin_amdgcn_sched_barrier(0);
        }
        if (NOSTORE) {
            double s = 0;
#pragma unroll
            for (int t = 0; t < NT; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
            if (s == 1.2345e300) Y[0] = make_double2(s, s);
            return;
        }
        const int64_t row0 = strip * 16;
        const int n = lane & 15, g = lane >> 4;
        if (ROWMAJOR && !S16) {
            double* Yd = (double*)Y;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int jc = 8 * t + (n >> 1);
                if (jc < p) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int64_t r = row0 + g + 4 * i;
                        if (r < rows) Yd[(r * ldy + jc) * 2 + (n & 1)] = acc[t][i];
                    }
                }
            }
        } else {
            const bool odd = n & 1;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int jc = 8 * t + (n >> 1);
                const double s0 = odd ? acc[t][0] : acc[t][1];
                const double s1 = odd ? acc[t][2] : acc[t][3];
                const double r0 = shfl_xor_d(s0, 1);
                const double r1 = shfl_xor_d(s1, 1);
                cplx v0, v1;
                if (!odd) { v0 = make_double2(acc[t][0], r0); v1 = make_double2(acc[t][2], r1); }
                else      { v0 = make_double2(r0, acc[t][1]); v1 = make_double2(r1, acc[t][3]); }
                const int64_t ra_ = row0 + g + (odd ? 4 : 0), rb_ = ra_ + 8;
                if (NOPRED) {
                    if (ROWMAJOR) {
                        if (NTST) { ntstore(Y + ra_ * ldy + jc, v0); ntstore(Y + rb_ * ldy + jc, v1); }
                        else { Y[ra_ * ldy + jc] = v0; Y[rb_ * ldy + jc] = v1; }
                    } else { cplx* col = Y + (int64_t)jc * ldy; if (NTST) { ntstore(col + ra_, v0); ntstore(col + rb_, v1); } else { col[ra_] = v0; col[rb_] = v1; } }
                } else if (jc < p) {
                    if (ROWMAJOR) {
                        if (ra_ < rows) Y[ra_ * ldy + jc] = v0;
                        if (rb_ < rows) Y[rb_ * ldy + jc] = v1;
                    } else {
                        cplx* col = Y + (int64_t)jc * ldy;
                        if (ra_ < rows) col[ra_] = v0;
                        if (rb_ < rows) col[rb_] = v1;
                    }
                }
            }
        }
    };
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    int nstr = 1;
    body(strip);
    for (strip += stride; strip < nstrips; strip += stride) { body(strip); ++nstr; }
    if (lane == 0) { unsigned long long* g = g_wave + 4 * (blockIdx.x * 8 + wv); g[0] = w0 - wstart; g[1] = wall_clock64() - wstart; g[2] = nstr;
                     atomicMin(&g_span[0], wstart); atomicMax(&g_span[1], wall_clock64()); }
    if (threadIdx.x == 0) { g_clk[2 * blockIdx.x] = clock64() - c0; g_clk[2 * blockIdx.x + 1] = wall_clock64() - w0; }
}

// ---------------------------------------------------------------------------------------------- v4
// v3 + (a) the strips of a workgroup's contiguous chunk are handed out by an LDS counter (the two waves of a SIMD do not finish
// 5 % apart any more, and a chunk is 245 or 246 strips instead of 30 or 31 per wave), (b) the first strip's Z loads are in flight
// during the fragment copy.  FLAGS bit 0: 16-byte paired stores, bit 5: non-temporal stores, bit 6: non-temporal loads
template <int NT, int NKS, bool ROWMAJOR, int FLAGS>
__global__ __launch_bounds__(512) void k_v4(const cplx* __restrict__ Z, int64_t ldz, int64_t rows, int k,
                                            const double* __restrict__ Bfrag, int p, cplx* __restrict__ Y, int64_t ldy, cplx* __restrict__ dump) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    __shared__ int s_next;
    constexpr int PER_KS = NT * 2 * 64;
    const unsigned long long wstart = wall_clock64();
    constexpr bool S16 = FLAGS & 1, NTST = FLAGS & 32, NTLD = FLAGS & 64;
    double* bs = (double*)smem_raw;
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const int m = lane & 15, q = lane >> 4;
    const int64_t nstrips = (rows + 15) / 16;
    // chunk of this workgroup: strips [c0, c1)
    const int64_t per = nstrips / gridDim.x, rem = nstrips % gridDim.x;
    const int64_t c0 = blockIdx.x * per + (blockIdx.x < rem ? blockIdx.x : rem);
    const int64_t c1 = c0 + per + (blockIdx.x < rem ? 1 : 0);
    int64_t cur = c0 + wv;
    const bool have = cur < c1;
    cplx z[NKS];
    auto colptr = [&](int ks) -> const cplx* {
        int col = 4 * ks + q;
        if (4 * ks + 3 >= k) col = col < k ? col : k - 1;
        return Z + (int64_t)col * ldz;
    };
    if (have) {
        int64_t arow = cur * 16;
        if (arow > rows - 16) arow = rows - 16;              // the last strip is shifted up: its rows overlap the one before (same values)
        arow += m;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) { z[ks] = NTLD ? ntload(colptr(ks) + arow) : colptr(ks)[arow]; __builtin_amdgcn_sched_barrier(0); }
    }
    {
        const double2* src = (const double2*)Bfrag;
        double2* dst = (double2*)bs;
        constexpr int n2 = NKS * PER_KS / 2;
        for (int t = threadIdx.x; t < n2; t += 512) dst[t] = src[t];
        if (threadIdx.x == 0) s_next = (int)(c0 + 8 - c0) ;      // offsets within the chunk; the first 8 are taken
    }
    __syncthreads();
    if (!have) return;
    const int nchunk = (int)(c1 - c0);
    const double* bl = bs + lane;
    const int n = lane & 15, g = lane >> 4;
    const bool odd = n & 1;
    int nxt_off;
    auto body = [&](int64_t strip) __attribute__((always_inline)) {
        {   // draw the wave's next strip
            int v = 0;
            if (lane == 0) v = atomicAdd(&s_next, 1);
            nxt_off = __builtin_amdgcn_readfirstlane(v);
        }
        const int64_t next = nxt_off < nchunk ? c0 + nxt_off : strip;
        int64_t nrow = next * 16;
        if (nrow > rows - 16) nrow = rows - 16;
        nrow += m;
        d4 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = (d4){0.0, 0.0, 0.0, 0.0};
        double f[2][2 * NT];
#pragma unroll
        for (int i = 0; i < 2 * NT; ++i) f[0][i] = bl[i * 64];
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const cplx a = z[ks];
            if (ks + 1 < NKS) {
#pragma unroll
                for (int i = 0; i < 2 * NT; ++i) f[(ks + 1) & 1][i] = bl[(ks + 1) * PER_KS + i * 64];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a.x, f[ks & 1][t * 2 + 0], acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a.y, f[ks & 1][t * 2 + 1], acc[t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            z[ks] = NTLD ? ntload(colptr(ks) + nrow) : colptr(ks)[nrow];
            __builtin_amdgcn_sched_barrier(0);
        }
        int64_t row0 = strip * 16;
        if (row0 > rows - 16) row0 = rows - 16;
        // unconditional stores (lanes of columns >= p write to a dump line): every path through the strip issues the same number of
        // memory operations, so the compiler's vmcnt waits are exact and never drain the stores
        if (ROWMAJOR && !S16) {
            double* Yd = (double*)Y;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int jc = 8 * t + (n >> 1);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int64_t r = row0 + g + 4 * i;
                    double* pd = jc < p ? Yd + (r * ldy + jc) * 2 + (n & 1) : (double*)dump + lane;
                    *pd = acc[t][i];
                }
            }
        } else {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int jc = 8 * t + (n >> 1);
                const double s0 = odd ? acc[t][0] : acc[t][1];
                const double s1 = odd ? acc[t][2] : acc[t][3];
                const double r0 = shfl_xor_d(s0, 1);
                const double r1 = shfl_xor_d(s1, 1);
                cplx v0, v1;
                if (!odd) { v0 = make_double2(acc[t][0], r0); v1 = make_double2(acc[t][2], r1); }
                else      { v0 = make_double2(r0, acc[t][1]); v1 = make_double2(r1, acc[t][3]); }
                const int64_t ra_ = row0 + g + (odd ? 4 : 0), rb_ = ra_ + 8;
                cplx* pa = ROWMAJOR ? Y + ra_ * ldy + jc : Y + (int64_t)jc * ldy + ra_;
                cplx* pb = ROWMAJOR ? Y + rb_ * ldy + jc : Y + (int64_t)jc * ldy + rb_;
                if (jc >= p) { pa = dump + lane; pb = dump + 64 + lane; }
                if (NTST) { ntstore(pa, v0); ntstore(pb, v1); } else { *pa = v0; *pb = v1; }
            }
        }
    };
    const unsigned long long cc0 = clock64(), w0 = wall_clock64();
    int nstr = 1;
    body(cur);
    while (nxt_off < nchunk) { cur = c0 + nxt_off; body(cur); ++nstr; }
    if (lane == 0) { unsigned long long* gw = g_wave + 4 * (blockIdx.x * 8 + wv); gw[0] = w0 - wstart; gw[1] = wall_clock64() - wstart; gw[2] = nstr;
                     atomicMin(&g_span[0], wstart); atomicMax(&g_span[1], wall_clock64()); }
    if (threadIdx.x == 0) { g_clk[2 * blockIdx.x] = clock64() - cc0; g_clk[2 * blockIdx.x + 1] = wall_clock64() - w0; }
}

// ---------------------------------------------------------------------------------------------- v5
// v3 with a RUN-TIME number of k-steps: the register buffer has MAXKS slots, all MAXKS loads of a strip are issued (columns
// clamped to k - 1: the surplus ones hit L1), only the MFMA blocks of k-steps >= nks are skipped by wave-uniform branches -- the
// number of memory operations per strip stays a compile-time constant, which is what keeps the compiler's vmcnt waits exact
template <int NT, int MAXKS, bool ROWMAJOR>
__global__ __launch_bounds__(512) void k_v5(const cplx* __restrict__ Z, int64_t ldz, int64_t rows, int k,
                                            const double* __restrict__ Bfrag, int nks, int p, cplx* __restrict__ Y, int64_t ldy) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int PER_KS = NT * 2 * 64;
    const unsigned long long wstart = wall_clock64();
    double* bs = (double*)smem_raw;
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const int m = lane & 15, q = lane >> 4;
    const int64_t nstrips = (rows + 15) / 16;
    const int64_t stride = (int64_t)gridDim.x * 8;
    int64_t strip = blockIdx.x * 8LL + wv;
    const bool have = strip < nstrips;
    cplx z[MAXKS];
    auto colptr = [&](int ks) -> const cplx* {
        int col = 4 * ks + q;
        col = col < k ? col : k - 1;
        return Z + (int64_t)col * ldz;
    };
    if (have) {
        int64_t arow = strip * 16 + m;
        if (arow >= rows) arow = rows - 1;
#pragma unroll
        for (int ks = 0; ks < MAXKS; ++ks) { z[ks] = colptr(ks)[arow]; __builtin_amdgcn_sched_barrier(0); }
    }
    {
        const double2* src = (const double2*)Bfrag;
        double2* dst = (double2*)bs;
        const int n2 = nks * PER_KS / 2;
        for (int t = threadIdx.x; t < n2; t += 512) dst[t] = src[t];
    }
    __syncthreads();
    if (!have) return;
    const double* bl = bs + lane;
    auto body = [&](int64_t strip) __attribute__((always_inline)) {
        const int64_t next = strip + stride < nstrips ? strip + stride : strip;
        int64_t nrow = next * 16 + m;
        if (nrow >= rows) nrow = rows - 1;
        d4 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int ks = 0; ks < MAXKS; ++ks) {
            if (ks < nks) {
                const cplx a = z[ks];
                double f[2 * NT];
#pragma unroll
                for (int i = 0; i < 2 * NT; ++i) f[i] = bl[ks * PER_KS + i * 64];
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a.x, f[t * 2 + 0], acc[t], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a.y, f[t * 2 + 1], acc[t], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            z[ks] = colptr(ks)[nrow];
            __builtin_amdgcn_sched_barrier(0);
        }
        const int64_t row0 = strip * 16;
        const int n = lane & 15, g = lane >> 4;
        if (ROWMAJOR) {
            double* Yd = (double*)Y;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int jc = 8 * t + (n >> 1);
                if (jc < p) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int64_t r = row0 + g + 4 * i;
                        if (r < rows) Yd[(r * ldy + jc) * 2 + (n & 1)] = acc[t][i];
                    }
                }
            }
        } else {
            const bool odd = n & 1;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int jc = 8 * t + (n >> 1);
                const double s0 = odd ? acc[t][0] : acc[t][1];
                const double s1 = odd ? acc[t][2] : acc[t][3];
                const double r0 = shfl_xor_d(s0, 1);
                const double r1 = shfl_xor_d(s1, 1);
                cplx v0, v1;
                if (!odd) { v0 = make_double2(acc[t][0], r0); v1 = make_double2(acc[t][2], r1); }
                else      { v0 = make_double2(r0, acc[t][1]); v1 = make_double2(r1, acc[t][3]); }
                const int64_t ra_ = row0 + g + (odd ? 4 : 0), rb_ = ra_ + 8;
                if (jc < p) {
                    cplx* col = Y + (int64_t)jc * ldy;
                    if (ra_ < rows) col[ra_] = v0;
                    if (rb_ < rows) col[rb_] = v1;
                }
            }
        }
    };
    const unsigned long long cc0 = clock64(), w0 = wall_clock64();
    body(strip);
    for (strip += stride; strip < nstrips; strip += stride) body(strip);
    if (lane == 0) { atomicMin(&g_span[0], wstart); atomicMax(&g_span[1], wall_clock64()); }
    if (threadIdx.x == 0) { g_clk[2 * blockIdx.x] = clock64() - cc0; g_clk[2 * blockIdx.x + 1] = wall_clock64() - w0; }
}

struct Ctx {
    int64_t rows; int k, p, reps; cplx *Z, *B, *Y, *dump; double* frag; unsigned long long* d_err; int ncu;
};

template <typename F>
static void run_case(const char* name, Ctx& c, int rm, int nks_frag, int nt, F launch) {
    const int64_t ldy = rm ? c.p : c.rows;
    CK(hipMemset(c.Y, 0, (size_t)c.rows * c.p * sizeof(cplx)));
    hipLaunchKernelGGL(k_expand_B, dim3(256), dim3(256), 0, 0, c.B, (int64_t)c.k, c.k, c.p, nks_frag, nt, c.frag);
    launch();
    CK(hipGetLastError());
    CK(hipDeviceSynchronize());
    CK(hipMemset(c.d_err, 0, 8));
    const int64_t tot = c.rows * c.p;
    hipLaunchKernelGGL(k_check, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, 0, c.Z, c.rows, c.rows, c.k, c.B, c.p, c.Y, ldy, rm, c.d_err);
    unsigned long long e; CK(hipMemcpy(&e, c.d_err, 8, hipMemcpyDeviceToHost));
    double err; memcpy(&err, &e, 8);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) launch();
    CK(hipDeviceSynchronize());
    float best = 1e30f, tot_ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < c.reps; ++i) launch();
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        ms /= c.reps; tot_ms += ms; if (ms < best) best = ms;
    }
    unsigned long long sp[2] = {~0ull, 0ull};
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_span), sp, 16));
    launch();
    CK(hipDeviceSynchronize());
    CK(hipMemcpyFromSymbol(sp, HIP_SYMBOL(g_span), 16));
    const double span_ms = sp[1] > sp[0] ? (double)(sp[1] - sp[0]) * 1e-5 : 0.0;
    std::vector<unsigned long long> hc(2 * c.ncu);
    CK(hipMemcpyFromSymbol(hc.data(), HIP_SYMBOL(g_clk), hc.size() * 8));
    double sc = 0, sw = 0;
    for (int i = 0; i < c.ncu; ++i) { sc += (double)hc[2 * i]; sw += (double)hc[2 * i + 1]; }
    const double ghz = sw > 0 ? sc / sw * 0.1 : 0.0;      // wall_clock64 ticks at 100 MHz
    if (getenv("NEP_UB_WAVES") && sw > 0) {
        std::vector<unsigned long long> hw(4 * 8 * c.ncu);
        CK(hipMemcpyFromSymbol(hw.data(), HIP_SYMBOL(g_wave), hw.size() * 8));
        double st[8] = {0}, en[8] = {0}, ns[8] = {0}; double enmax = 0;
        for (int b = 0; b < c.ncu; ++b) for (int w = 0; w < 8; ++w) { const unsigned long long* g = &hw[4 * (b * 8 + w)]; st[w] += g[0]; en[w] += g[1]; ns[w] += g[2]; if (g[1] > enmax) enmax = g[1]; }
        printf("  %s per-wave (avg over WGs) start/end us:", name);
        for (int w = 0; w < 8; ++w) printf(" w%d %.1f/%.1f(%.1f)", w, st[w] / c.ncu * 0.01, en[w] / c.ncu * 0.01, ns[w] / c.ncu);
        printf("  max end %.1f us\n", enmax * 0.01);
    }
    CK(hipMemset(c.d_err, 0, 8));
    { std::vector<unsigned long long> z(2 * 1024, 0ull); CK(hipMemcpyToSymbol(HIP_SYMBOL(g_clk), z.data(), z.size() * 8)); }
    const double fl = 8.0 * c.rows * c.k * c.p;
    printf("{\"variant\": \"%s\", \"rowmajor\": %d, \"rows\": %lld, \"k\": %d, \"p\": %d, \"ms_best\": %.4f, \"ms_avg\": %.4f, \"TFLOPs_best\": %.2f, "
           "\"TFLOPs_avg\": %.2f, \"frac_peak_avg\": %.3f, \"sclk_GHz\": %.3f, \"loop_ms_wave0\": %.4f, \"span_ms_single\": %.4f, \"maxerr\": %.3e}\n",
           name, rm, (long long)c.rows, c.k, c.p, best, tot_ms / 3, fl / best / 1e9, fl / (tot_ms / 3) / 1e9, fl / (tot_ms / 3) / 1e9 / 78.6, ghz, sw / c.ncu * 1e-5, span_ms, err);
    fflush(stdout);
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
}

template <int NT, int NKS, bool RM, int FLAGS>
static void case_v3(const char* name, Ctx& c) {
    if ((c.k + 3) / 4 != NKS || (c.p + 7) / 8 != NT) return;
    const size_t shm = (size_t)NKS * NT * 2 * 64 * sizeof(double);
    CK(hipFuncSetAttribute((const void*)k_v3<NT, NKS, RM, FLAGS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
    run_case(name, c, RM, NKS, NT, [&] {
        hipLaunchKernelGGL((k_v3<NT, NKS, RM, FLAGS>), dim3(c.ncu), dim3(512), shm, 0, c.Z, c.rows, c.rows, c.k, c.frag, c.p, c.Y,
                           RM ? (int64_t)c.p : c.rows);
    });
}

template <int NT, int NKS, bool RM, int FLAGS>
static void case_v4(const char* name, Ctx& c) {
    if ((c.k + 3) / 4 != NKS || (c.p + 7) / 8 != NT) return;
    const size_t shm = (size_t)NKS * NT * 2 * 64 * sizeof(double);
    CK(hipFuncSetAttribute((const void*)k_v4<NT, NKS, RM, FLAGS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
    run_case(name, c, RM, NKS, NT, [&] {
        hipLaunchKernelGGL((k_v4<NT, NKS, RM, FLAGS>), dim3(c.ncu), dim3(512), shm, 0, c.Z, c.rows, c.rows, c.k, c.frag, c.p, c.Y,
                           RM ? (int64_t)c.p : c.rows, c.dump);
    });
}

template <int NT, int MAXKS, bool RM>
static void case_v5(const char* name, Ctx& c) {
    const int nks = (c.k + 3) / 4;
    if ((c.p + 7) / 8 != NT || nks > MAXKS || nks + 8 <= MAXKS) return;
    const size_t shm = (size_t)nks * NT * 2 * 64 * sizeof(double);
    CK(hipFuncSetAttribute((const void*)k_v5<NT, MAXKS, RM>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
    run_case(name, c, RM, nks, NT, [&] {
        hipLaunchKernelGGL((k_v5<NT, MAXKS, RM>), dim3(c.ncu), dim3(512), shm, 0, c.Z, c.rows, c.rows, c.k, c.frag, nks, c.p, c.Y,
                           RM ? (int64_t)c.p : c.rows);
    });
}

template <int NT, bool RM>
static void case_exact(const char* name, Ctx& c) {
    if ((c.p + 7) / 8 != NT) return;
    const int nkt = (c.k + 3) / 4, nks = (nkt + 7) & ~7;
    const size_t shm = (size_t)nkt * NT * 2 * 64 * sizeof(double);
    if (shm > 147456) return;
    CK(hipFuncSetAttribute((const void*)k_exact<NT, RM>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
    run_case(name, c, RM, nkt, NT, [&] {
        hipLaunchKernelGGL((k_exact<NT, RM>), dim3(c.ncu), dim3(512), shm, 0, c.Z, c.rows, c.rows, c.k, c.frag, nks, nkt, c.p, c.Y,
                           RM ? (int64_t)c.p : c.rows);
    });
}

template <int NT, bool RM>
static void case_base(const char* name, Ctx& c) {
    if ((c.p + 7) / 8 != NT) return;
    const int nks = (((c.k + 3) / 4) + 7) & ~7;
    const size_t shm = (size_t)nks * NT * 2 * 64 * sizeof(double);
    if (shm > 147456) return;
    CK(hipFuncSetAttribute((const void*)k_base<NT, RM>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
    run_case(name, c, RM, nks, NT, [&] {
        hipLaunchKernelGGL((k_base<NT, RM>), dim3(c.ncu), dim3(512), shm, 0, c.Z, c.rows, c.rows, c.k, c.frag, nks, c.p, c.Y,
                           RM ? (int64_t)c.p : c.rows);
    });
}

int main(int argc, char** argv) {
    Ctx c;
    c.rows = argc > 1 ? atoll(argv[1]) : 1003995;
    c.k = argc > 2 ? atoi(argv[2]) : 60;
    c.p = argc > 3 ? atoi(argv[3]) : 60;
    c.reps = argc > 4 ? atoi(argv[4]) : 20;
    hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
    c.ncu = pr.multiProcessorCount;
    CK(hipMalloc(&c.Z, (size_t)c.rows * c.k * sizeof(cplx)));
    CK(hipMalloc(&c.B, (size_t)c.k * c.p * sizeof(cplx)));
    CK(hipMalloc(&c.Y, (size_t)c.rows * c.p * sizeof(cplx)));
    CK(hipMalloc(&c.frag, (size_t)32 * 16 * 128 * sizeof(double)));
    CK(hipMalloc(&c.d_err, 8));
    CK(hipMalloc(&c.dump, 4096));
    const uint64_t lowent = (getenv("NEP_UB_LOWENT") && atoi(getenv("NEP_UB_LOWENT")) ? (1ull << 63) : 0ull) |
                            (getenv("NEP_UB_NORMAL") && atoi(getenv("NEP_UB_NORMAL")) ? (1ull << 62) : 0ull);
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, c.Z, c.rows * c.k, 1ull | lowent);
    hipLaunchKernelGGL(k_fill, dim3(16), dim3(256), 0, 0, c.B, (int64_t)c.k * c.p, 77ull | lowent);
    CK(hipDeviceSynchronize());
    case_base<8, true>("base_rm", c);
    case_base<8, false>("base_cm", c);
    case_exact<8, true>("exact_rm", c);
    case_exact<8, false>("exact_cm", c);
    case_v5<8, 16, true>("v5_rm_max16", c);
    case_v5<8, 16, false>("v5_cm_max16", c);
    case_v5<8, 24, true>("v5_rm_max24", c);
    case_v5<8, 24, false>("v5_cm_max24", c);
    case_v5<4, 16, true>("v5_rm_nt4_max16", c);
    case_v5<4, 24, true>("v5_rm_nt4_max24", c);
    case_v3<8, 15, true, 0>("v3_rm", c);
    case_v3<8, 15, true, 1>("v3_rm_s16", c);
    case_v3<8, 15, true, 2>("v3_rm_nopf", c);
    case_v3<8, 15, true, 3>("v3_rm_s16_nopf", c);
    case_v3<8, 15, false, 0>("v3_cm", c);
    case_v4<8, 15, true, 0>("v4_rm", c);
    case_v4<8, 15, true, 1>("v4_rm_s16", c);
    case_v4<8, 15, true, 1 | 32>("v4_rm_s16_ntst", c);
    case_v4<8, 15, true, 1 | 32 | 64>("v4_rm_s16_ntst_ntld", c);
    case_v4<8, 15, false, 0>("v4_cm", c);
    case_v4<8, 15, false, 32>("v4_cm_ntst", c);
    case_v4<8, 16, true, 0>("v4_rm", c);
    case_v4<8, 16, true, 1 | 32>("v4_rm_s16_ntst", c);
    case_v4<8, 16, false, 32>("v4_cm_ntst", c);
    case_v3<8, 15, true, 1 | 4>("v3_rm_s16_NOLOAD", c);
    case_v3<8, 15, true, 1 | 8>("v3_rm_s16_NOSTORE", c);
    case_v3<8, 15, true, 1 | 4 | 8>("v3_rm_s16_NOLOAD_NOSTORE", c);
    case_v3<8, 16, true, 1>("v3_rm_s16", c);        // k = p = 64
    case_v3<8, 16, false, 0>("v3_cm", c);
    if (c.rows % 16 == 0 && c.p % 8 == 0) {
        case_v3<8, 16, true, 1 | 16>("v3_rm_s16_nopred", c);
        case_v3<8, 16, true, 1 | 16 | 32>("v3_rm_s16_nopred_ntst", c);
        case_v3<8, 16, true, 1 | 16 | 64>("v3_rm_s16_nopred_ntld", c);
        case_v3<8, 16, true, 1 | 16 | 32 | 64>("v3_rm_s16_nopred_ntst_ntld", c);
        case_v3<8, 16, true, 1 | 16 | 4>("v3_rm_s16_nopred_NOLOAD", c);
        case_v3<8, 16, true, 1 | 16 | 8>("v3_rm_s16_nopred_NOSTORE", c);
        case_v3<8, 16, true, 1 | 16 | 4 | 8>("v3_rm_s16_nopred_NOLOAD_NOSTORE", c);
        case_v3<8, 16, false, 16>("v3_cm_nopred", c);
    }
    return 0;
}
