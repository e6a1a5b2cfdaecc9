// libnepmi355: K7 tall-skinny complex GEMM  Y = Z * B  on the gfx950 FP64 matrix cores.
//
// Complex arithmetic is mapped onto v_mfma_f64_16x16x4_f64 through the real embedding
//   [Yre Yim] = [Zre Zim] * [[Bre Bim],[-Bim Bre]]
// with a K-ordering chosen so that every lane loads ONE full complex128 (16 B) of Z per k-step:
// MFMA #0 consumes the real parts of 4 complex columns of Z, MFMA #1 the imaginary parts.
//   A operand (16 x 4): lane l -> row (l & 15), complex column 4*ks + (l >> 4)
//   B operand (4 x 16): lane l -> k index (l >> 4), real output column n = l & 15
//                       (complex output column 8*nt + n/2, part n & 1)
//   C/D (16 x 16, 4 regs): reg i, lane l -> row (l >> 4) + 4 i, real column l & 15
// The B operands are pre-expanded on the host into exactly this lane order, so a wave fetches a
// fragment with one conflict-free 512-byte LDS read.  Z is streamed from HBM exactly once; the
// accumulators for ALL p output columns of a 16-row strip live in registers.
#include "common.h"
#include <cstring>
#include <vector>
#include <algorithm>

typedef double d4 __attribute__((ext_vector_type(4)));

#define GEMM_KCH 4  // k-steps (of 4 complex columns) staged in LDS per chunk

template <int NT, bool ROWMAJOR>
__global__ __launch_bounds__(512) void k_gemm_ts(const cplx* __restrict__ Z, int64_t ldz, int64_t rows, int k,
                                                 const double* __restrict__ Bfrag, int nks, int p, int j0,
                                                 cplx* __restrict__ Y, int64_t ldy) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int PER_KS = NT * 2 * 64;               // doubles per k-step
    constexpr int CH = GEMM_KCH * PER_KS;             // doubles per LDS stage
    constexpr int BPT = (CH / 2 + 511) / 512;         // double2 per thread per stage (NT*2*64*4/2/512 = NT/2 rounded up)
    double* bs = (double*)smem_raw;                   // [2][GEMM_KCH][NT][2][64]  (double buffered)
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const int64_t row0 = (blockIdx.x * 8LL + wv) * 16;
    const int m = lane & 15, q = lane >> 4;
    int64_t arow = row0 + m;
    if (arow >= rows) arow = rows - 1;
    d4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = (d4){0.0, 0.0, 0.0, 0.0};

    const int nch = (nks + GEMM_KCH - 1) / GEMM_KCH;
    cplx a[GEMM_KCH], an[GEMM_KCH];
    double2 bn[BPT];

    auto load_a = [&](int c, cplx* dst) {
#pragma unroll
        for (int s = 0; s < GEMM_KCH; ++s) {
            const int ks = c * GEMM_KCH + s;
            int col = 4 * ks + q;
            if (col >= k) col = k - 1;
            dst[s] = (ks < nks) ? Z[(int64_t)col * ldz + arow] : cmake(0.0, 0.0);
        }
    };
    auto load_b = [&](int c) {
        const int nk = min(GEMM_KCH, nks - c * GEMM_KCH);
        const double2* src = (const double2*)(Bfrag + (int64_t)c * CH);
        const int n2 = nk * PER_KS / 2;
#pragma unroll
        for (int i = 0; i < BPT; ++i) {
            const int t = threadIdx.x + 512 * i;
            bn[i] = (t < n2) ? src[t] : make_double2(0.0, 0.0);
        }
    };
    auto store_b = [&](int buf) {
        double2* dst = (double2*)(bs + (size_t)buf * CH);
#pragma unroll
        for (int i = 0; i < BPT; ++i) {
            const int t = threadIdx.x + 512 * i;
            if (t < CH / 2) dst[t] = bn[i];
        }
    };

    // prologue: stage chunk 0
    load_a(0, a);
    load_b(0);
    store_b(0);
    __syncthreads();
    for (int c = 0; c < nch; ++c) {
        const bool more = c + 1 < nch;
        if (more) { load_a(c + 1, an); load_b(c + 1); }        // global loads in flight during the MFMAs below
        const int nk = min(GEMM_KCH, nks - c * GEMM_KCH);
        const double* bk0 = bs + (size_t)(c & 1) * CH + lane;
#pragma unroll
        for (int s = 0; s < GEMM_KCH; ++s) {
            if (s < nk) {
                const double* bk = bk0 + s * PER_KS;
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s].x, bk[(t * 2 + 0) * 64], acc[t], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s].y, bk[(t * 2 + 1) * 64], acc[t], 0, 0, 0);
            }
        }
        if (more) {
            store_b((c + 1) & 1);                               // the other stage: nobody reads it during chunk c
#pragma unroll
            for (int s = 0; s < GEMM_KCH; ++s) a[s] = an[s];
        }
        __syncthreads();
    }
    // ---- epilogue
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
                    if (r < rows) Yd[(r * ldy + j0 + jc) * 2 + (n & 1)] = acc[t][i];
                }
            }
        }
    } else {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int jc = 8 * t + (n >> 1);
            // even lanes keep regs {0,2}, odd lanes regs {1,3}; partner supplies the other part
            const bool odd = n & 1;
            const double s0 = odd ? acc[t][0] : acc[t][1];
            const double s1 = odd ? acc[t][2] : acc[t][3];
            const double r0 = shfl_xor_d(s0, 1);
            const double r1 = shfl_xor_d(s1, 1);
            cplx v0, v1;
            int i0, i1;
            if (!odd) { v0 = cmake(acc[t][0], r0); v1 = cmake(acc[t][2], r1); i0 = 0; i1 = 2; }
            else      { v0 = cmake(r0, acc[t][1]); v1 = cmake(r1, acc[t][3]); i0 = 1; i1 = 3; }
            if (jc < p) {
                const int64_t ra = row0 + g + 4 * i0, rb = row0 + g + 4 * i1;
                cplx* col = Y + (int64_t)(j0 + jc) * ldy;
                if (ra < rows) col[ra] = v0;
                if (rb < rows) col[rb] = v1;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// B-resident persistent variant for tall blocks: when ALL B fragments of a column panel fit in LDS (nks * NT KiB <= 144 KiB, e.g.
// k = p = 60: 120 KiB) they are loaded once per workgroup and every wave walks 16-row strips on its own -- no __syncthreads in
// the main loop.  The Z values of a WHOLE strip live in a register buffer of MAXKS slots; slot ks is refilled for the wave's next
// strip right behind the MFMAs of k-step ks, so every load has a full strip of MFMAs (about 16 us) to land.  The number of k-steps
// is a run-time argument: all MAXKS loads of a strip are issued (columns clamped to k - 1: the surplus ones hit L1) and only the
// MFMA blocks of k-steps >= nks are skipped, by wave-uniform branches.  That keeps the number of memory operations per strip a
// compile-time constant -- the condition under which the compiler's s_waitcnt vmcnt(N) for "Z of k-step ks has landed" does not
// also drain the younger loads and the stores of the previous strip (gfx950 has ONE counter for loads and stores).  For the same
// reason the prologue loads are pinned in k-step order and the first strip is peeled off the loop (both predecessors of the loop
// header then carry the same queue).  Round 3; the ring-of-four kernel of round 2 padded the k-steps to a multiple of 8 (k = 60:
// 16 instead of 15, k = 52: 16 instead of 13) and waited for its ring once per 4 k-steps: 48 -> 52 TFLOP/s at k = p = 60, 43 -> 50
// at k = 52 (scripts/ub_k7.hip, full-entropy operands).  One workgroup per CU (LDS-bound), 2 waves per SIMD.
#define GEMM_RES_MAXKS 24
#define GEMM_TALL_ROWS (16LL * 16 * 8 * 256)   // >= 16 strips per wave of a full grid: the resident kernel is worth its set-up
static inline int gemm_nks(int k, int64_t rows) { (void)rows; return (k + 3) / 4; }
template <int NT, int MAXKS, bool ROWMAJOR>
__global__ __launch_bounds__(512) void k_gemm_ts_res(const cplx* __restrict__ Z, int64_t ldz, int64_t rows, int k,
                                                     const double* __restrict__ Bfrag, int nks, int p, int j0,
                                                     cplx* __restrict__ Y, int64_t ldy) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int PER_KS = NT * 2 * 64;
    double* bs = (double*)smem_raw;                       // [nks][NT][2][64]
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
        col = col < k ? col : k - 1;                      // the matching B rows are zero / the k-step is skipped
        return Z + (int64_t)col * ldz;
    };
    if (have) {                                           // the first strip's loads are in flight during the fragment copy
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
            z[ks] = colptr(ks)[nrow];                     // the same k-step of the wave's next strip: a whole strip ahead
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
                        if (r < rows) Yd[(r * ldy + j0 + jc) * 2 + (n & 1)] = acc[t][i];
                    }
                }
            }
        } else {
            const bool odd = n & 1;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int jc = 8 * t + (n >> 1);
                // even lanes keep regs {0,2}, odd lanes regs {1,3}; the partner supplies the other part
                const double s0 = odd ? acc[t][0] : acc[t][1];
                const double s1 = odd ? acc[t][2] : acc[t][3];
                const double r0 = shfl_xor_d(s0, 1);
                const double r1 = shfl_xor_d(s1, 1);
                cplx v0, v1;
                if (!odd) { v0 = cmake(acc[t][0], r0); v1 = cmake(acc[t][2], r1); }
                else      { v0 = cmake(r0, acc[t][1]); v1 = cmake(r1, acc[t][3]); }
                const int64_t ra = row0 + g + (odd ? 4 : 0), rb = ra + 8;
                if (jc < p) {
                    cplx* col = Y + (int64_t)(j0 + jc) * ldy;
                    if (ra < rows) col[ra] = v0;
                    if (rb < rows) col[rb] = v1;
                }
            }
        }
    };
    body(strip);
    for (strip += stride; strip < nstrips; strip += stride) body(strip);
}

static inline int gemm_nt(int pp) {   // N-tiles of 8 complex output columns
    return pp <= 8 ? 1 : pp <= 16 ? 2 : pp <= 32 ? 4 : pp <= 48 ? 6 : pp <= 64 ? 8 : pp <= 80 ? 10 : 13;
}

static thread_local NepScratch g_gemm_scratch;
static thread_local PinnedRing g_gemm_ring;

template <int NT>
static int gemm_launch(bool rowmajor, const cplx* Z, int64_t ldz, int64_t rows, int k, const double* dB, int nks,
                       int p, int j0, cplx* Y, int64_t ldy, hipStream_t st) {
    // B-resident persistent kernel: all fragments in LDS (<= 144 KiB), tall blocks only (>= 16 strips per wave)
    static const int res_mode = getenv("NEP_GEMM_RES") ? atoi(getenv("NEP_GEMM_RES")) : 1;
    const size_t res_bytes = (size_t)nks * NT * 2 * 64 * sizeof(double);
    // (NT = 8 with more than 16 k-steps would spill: 96 buffer + 64 accumulator registers)
    if constexpr (NT <= 8) if (res_mode && nks <= (NT == 8 ? 16 : GEMM_RES_MAXKS) && res_bytes <= 147456 && rows >= GEMM_TALL_ROWS) {
        static thread_local int ncu = 0;
        if (!ncu) { hipDeviceProp_t pr; int dev = 0; (void)hipGetDevice(&dev); ncu = (hipGetDeviceProperties(&pr, dev) == hipSuccess) ? pr.multiProcessorCount : 256; }
        const dim3 grid((unsigned)ncu), block(512);
#define RES_LAUNCH(MAXKS_, RM_)                                                                                               \
        do {                                                                                                                  \
            static thread_local bool set_ = false;                                                                            \
            if (!set_) { HIPCHK(hipFuncSetAttribute((const void*)k_gemm_ts_res<NT, MAXKS_, RM_>, hipFuncAttributeMaxDynamicSharedMemorySize, 147456)); set_ = true; } \
            hipLaunchKernelGGL((k_gemm_ts_res<NT, MAXKS_, RM_>), grid, block, res_bytes, st, Z, ldz, rows, k, dB, nks, p, j0, Y, ldy); \
        } while (0)
        if (nks <= 16) { if (rowmajor) RES_LAUNCH(16, true); else RES_LAUNCH(16, false); }
        else if constexpr (NT < 8) { if (rowmajor) RES_LAUNCH(24, true); else RES_LAUNCH(24, false); }
#undef RES_LAUNCH
        LAUNCHCHK();
        return NEP_OK;
    }
    const dim3 grid((unsigned)((rows + 127) / 128)), block(512);
    const size_t shm = (size_t)2 * GEMM_KCH * NT * 2 * 64 * sizeof(double);      // double-buffered B stage
    if (rowmajor)
        hipLaunchKernelGGL((k_gemm_ts<NT, true>), grid, block, shm, st, Z, ldz, rows, k, dB, nks, p, j0, Y, ldy);
    else
        hipLaunchKernelGGL((k_gemm_ts<NT, false>), grid, block, shm, st, Z, ldz, rows, k, dB, nks, p, j0, Y, ldy);
    LAUNCHCHK();
    return NEP_OK;
}

// builds the lane-ordered B fragments of one column panel from a DEVICE matrix
// B[c, j] = b_rowmajor ? dB[c*ldb + j] : dB[j*ldb + c]
__global__ void k_expand_B(const cplx* __restrict__ B, int64_t ldb, int b_rowmajor, int k, int pp, int j0, int nks,
                           int nt, double* __restrict__ frag) {
    const int total = nks * nt * 64;          // (32-bit index arithmetic: the 64-bit divisions cost more than the rest of this kernel)
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int l = i & 63;
        const int kt = i >> 6;                // ks*nt + t
        const int ks = kt / nt, t = kt - ks * nt;
        const int q = l >> 4, n = l & 15;
        const int c = 4 * ks + q, jc = 8 * t + (n >> 1);
        double b0 = 0.0, b1 = 0.0;
        if (c < k && jc < pp) {
            const cplx b = b_rowmajor ? B[(int64_t)c * ldb + (j0 + jc)] : B[(int64_t)(j0 + jc) * ldb + c];
            if ((n & 1) == 0) { b0 = b.x; b1 = -b.y; } else { b0 = b.y; b1 = b.x; }
        }
        double* f = frag + (int64_t)kt * 128;
        f[l] = b0;
        f[64 + l] = b1;
    }
}

static int gemm_dispatch(int nt, bool y_rowmajor, const cplx* Z, int64_t ldz, int64_t rows, int k, const double* dB,
                         int nks, int pp, int j0, cplx* Y, int64_t ldy, hipStream_t st) {
    switch (nt) {
        case 1: return gemm_launch<1>(y_rowmajor, Z, ldz, rows, k, dB, nks, pp, j0, Y, ldy, st);
        case 2: return gemm_launch<2>(y_rowmajor, Z, ldz, rows, k, dB, nks, pp, j0, Y, ldy, st);
        case 4: return gemm_launch<4>(y_rowmajor, Z, ldz, rows, k, dB, nks, pp, j0, Y, ldy, st);
        case 6: return gemm_launch<6>(y_rowmajor, Z, ldz, rows, k, dB, nks, pp, j0, Y, ldy, st);
        case 8: return gemm_launch<8>(y_rowmajor, Z, ldz, rows, k, dB, nks, pp, j0, Y, ldy, st);
        case 10: return gemm_launch<10>(y_rowmajor, Z, ldz, rows, k, dB, nks, pp, j0, Y, ldy, st);
        default: return gemm_launch<13>(y_rowmajor, Z, ldz, rows, k, dB, nks, pp, j0, Y, ldy, st);
    }
}

extern "C" int32_t nep_gemm_ts_dev(const nep_cdouble* dZ, int64_t ldz, int64_t rows, int32_t k,
                                   const nep_cdouble* dB, int64_t ldb, int32_t b_rowmajor, int32_t p,
                                   nep_cdouble* dY, int64_t ldy, int32_t y_rowmajor, nep_stream stream);

// B arrives from the host: the k x p block itself is staged through the pinned ring (16 k p bytes; iar's Ritz block at
// k = p = 100: 160 KB) and the lane-ordered MFMA fragments are built on the device by k_expand_B.  (Round 1 expanded the
// fragments on the host: a 20 800-iteration triple loop + 333 KB upload per call cost more than the GEMM at gun size.)
extern "C" int32_t nep_gemm_ts(const nep_cdouble* dZ, int64_t ldz, int64_t rows, int32_t k,
                               const nep_cdouble* hB, int64_t ldb, int32_t p, nep_cdouble* dY, int64_t ldy,
                               int32_t y_rowmajor, nep_stream stream) {
    ARGCHK(dZ && hB && dY);
    ARGCHK(rows > 0 && k >= 1 && p >= 1 && ldz >= rows && ldb >= k);
    ARGCHK(y_rowmajor ? ldy >= p : ldy >= rows);
    hipStream_t st = as_stream(stream);
    const size_t bytes = (size_t)k * p * sizeof(cplx);
    int rc = g_gemm_scratch.ensure(bytes);
    if (rc) return rc;
    if (ldb == k) {
        rc = g_gemm_ring.upload(g_gemm_scratch.dptr, hB, bytes, st);
    } else {                                  // pack the columns (leading dimension > k)
        static thread_local std::vector<nep_cdouble> pack;
        pack.resize((size_t)k * p);
        for (int j = 0; j < p; ++j) memcpy(pack.data() + (size_t)j * k, hB + (size_t)j * ldb, (size_t)k * sizeof(nep_cdouble));
        rc = g_gemm_ring.upload(g_gemm_scratch.dptr, pack.data(), bytes, st);
    }
    if (rc) return rc;
    return nep_gemm_ts_dev(dZ, ldz, rows, k, (const nep_cdouble*)g_gemm_scratch.dptr, k, 0, p, dY, ldy, y_rowmajor, stream);
}

static thread_local NepScratch g_gemm_scratch_dev;

extern "C" int32_t nep_gemm_ts_dev(const nep_cdouble* dZ, int64_t ldz, int64_t rows, int32_t k,
                                   const nep_cdouble* dB, int64_t ldb, int32_t b_rowmajor, int32_t p,
                                   nep_cdouble* dY, int64_t ldy, int32_t y_rowmajor, nep_stream stream) {
    ARGCHK(dZ && dB && dY);
    ARGCHK(rows > 0 && k >= 1 && p >= 1 && ldz >= rows);
    ARGCHK(b_rowmajor ? ldb >= p : ldb >= k);
    ARGCHK(y_rowmajor ? ldy >= p : ldy >= rows);
    hipStream_t st = as_stream(stream);
    const int nks = gemm_nks(k, rows);
    size_t total = 0;
    for (int j0 = 0; j0 < p; j0 += 104) total += (size_t)nks * gemm_nt(std::min(104, p - j0)) * 128;
    int rc = g_gemm_scratch_dev.ensure(total * sizeof(double));
    if (rc) return rc;
    size_t off = 0;
    for (int j0 = 0; j0 < p; j0 += 104) {
        const int pp = std::min(104, p - j0);
        const int nt = gemm_nt(pp);
        double* frag = (double*)g_gemm_scratch_dev.dptr + off;
        const int64_t work = (int64_t)nks * nt * 64;
        hipLaunchKernelGGL(k_expand_B, dim3((unsigned)std::min<int64_t>((work + 255) / 256, 2048)), dim3(256), 0, st,
                           (const cplx*)dB, ldb, (int)b_rowmajor, (int)k, pp, j0, nks, nt, frag);
        LAUNCHCHK();
        rc = gemm_dispatch(nt, y_rowmajor != 0, (const cplx*)dZ, ldz, rows, k, frag, nks, pp, j0, (cplx*)dY, ldy, st);
        if (rc) return rc;
        off += (size_t)nks * nt * 128;
    }
    return NEP_OK;
}


// ------------------------------------------------------------------------------------------------
// K9  C = W^H Y  (k x p) for two tall row-major blocks WT (rows x k), YT (rows x p): the reduction GEMM behind
// B_i = W^H (A_i V) of Proj_SPMF_NEP (src/NEPTypes.jl:733) and Gram matrices.  Row-major operands ARE the MFMA
// operand layout of v_mfma_f64_16x16x4_f64 with the row index as contraction index: lane l loads
// WT[r0 + (l>>4)][i0 + (l&15)] (A operand, 16 contiguous complex per row) and YT[r0 + (l>>4)][j0 + (l&15)] (B operand),
// no LDS staging.  conj(a) b = (ar br + ai bi) + i (ar bi - ai br): four real MFMAs per 4-row step and 16 x 16 tile.
// One wave per tile, the waves of a workgroup share the rows through L1/L2; per-workgroup partial tiles are summed
// in a fixed order by k_gemm_h_reduce (deterministic).  Bound: HBM (16 rows (k+p) bytes) for k, p <~ 64, MFMA above.
__global__ __launch_bounds__(1024) void k_gemm_h_rm(const cplx* __restrict__ WT, int64_t ldw, const cplx* __restrict__ YT,
                                                    int64_t ldy, int64_t rows, int k, int p, int rows_per_wg,
                                                    cplx* __restrict__ partial) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int ti = (k + 15) / 16, tj = (p + 15) / 16;
    const int64_t rbeg = (int64_t)blockIdx.x * rows_per_wg;
    const int64_t rend = rbeg + rows_per_wg < rows ? rbeg + rows_per_wg : rows;
    const int li = lane & 15, lk = lane >> 4;
    cplx* out = partial + (int64_t)blockIdx.x * ti * tj * 256;
    for (int t = wv; t < ti * tj; t += nw) {
        const int i0 = (t / tj) * 16, j0 = (t % tj) * 16;
        const bool ia = i0 + li < k, jb = j0 + li < p;
        d4 cre = {0.0, 0.0, 0.0, 0.0}, cim = {0.0, 0.0, 0.0, 0.0};
        // 16 rows per trip: the 8 operand loads are issued before the 16 MFMAs that consume them
        for (int64_t r = rbeg; r < rend; r += 16) {
            cplx a[4], b[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t rr = r + 4 * u + lk;
                a[u] = cmake(0.0, 0.0); b[u] = cmake(0.0, 0.0);
                if (rr < rend) {
                    if (ia) a[u] = WT[rr * ldw + i0 + li];
                    if (jb) b[u] = YT[rr * ldy + j0 + li];
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                cre = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u].x, b[u].x, cre, 0, 0, 0);
                cim = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u].x, b[u].y, cim, 0, 0, 0);
                cre = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u].y, b[u].y, cre, 0, 0, 0);
                cim = __builtin_amdgcn_mfma_f64_16x16x4f64(-a[u].y, b[u].x, cim, 0, 0, 0);
            }
        }
        // D[row = (l>>4) + 4q][col = l&15]
        cplx* tile = out + (int64_t)t * 256;
#pragma unroll
        for (int q = 0; q < 4; ++q) tile[(lk + 4 * q) * 16 + li] = cmake(cre[q], cim[q]);
    }
}

// C[i + j*k] = sum_b partial[b][tile(i,j)][i%16][j%16]   (column-major k x p).  One workgroup per 16 x 16 tile and
// split of the partial range: 4 groups of 256 threads (one per tile element, coalesced 4 KB reads per partial), fixed
// summation order; the splits are summed by the same kernel in a second call (nsplit = 1).
__global__ __launch_bounds__(1024) void k_gemm_h_reduce(int nb, int ntile, int k, int p, int tj,
                                                        const cplx* __restrict__ partial, int64_t pstride,
                                                        cplx* __restrict__ out, int final_layout) {
    __shared__ cplx sm[4][256];
    const int t = blockIdx.x, sp = blockIdx.y, nsp = gridDim.y;
    const int e = threadIdx.x & 255, g = threadIdx.x >> 8;
    const int b0 = (int)((int64_t)nb * sp / nsp), b1 = (int)((int64_t)nb * (sp + 1) / nsp);
    cplx acc = cmake(0.0, 0.0);
    for (int b = b0 + g; b < b1; b += 4) acc = cadd(acc, partial[(int64_t)b * pstride + (int64_t)t * 256 + e]);
    sm[g][e] = acc;
    __syncthreads();
    if (g == 0) {
        const cplx v = cadd(cadd(sm[0][e], sm[1][e]), cadd(sm[2][e], sm[3][e]));
        if (!final_layout) {
            out[((int64_t)sp * ntile + t) * 256 + e] = v;
        } else {
            const int i = (t / tj) * 16 + e / 16, j = (t % tj) * 16 + e % 16;
            if (i < k && j < p) out[i + (int64_t)j * k] = v;
        }
    }
}

static thread_local NepScratch g_gemmh_scratch;

extern "C" int32_t nep_gemm_h_rm(const nep_cdouble* dWT, int64_t ldw, const nep_cdouble* dYT, int64_t ldy, int64_t rows,
                                 int32_t k, int32_t p, nep_cdouble* h_C, nep_stream stream) {
    ARGCHK(dWT && dYT && h_C);
    ARGCHK(rows > 0 && k >= 1 && p >= 1 && k <= 256 && p <= 256 && ldw >= k && ldy >= p);
    hipStream_t st = as_stream(stream);
    const int ti = (k + 15) / 16, tj = (p + 15) / 16;
    int rows_per_wg = rows >= 262144 ? 1024 : (rows >= 16384 ? 128 : 64);
    const int nb = (int)((rows + rows_per_wg - 1) / rows_per_wg);
    const int ntile = ti * tj;
    const int nsplit = nb >= 64 ? 16 : 1;
    const size_t pbytes = (size_t)nb * ntile * 256 * sizeof(cplx);
    const size_t p2bytes = (size_t)nsplit * ntile * 256 * sizeof(cplx);
    int rc = g_gemmh_scratch.ensure(pbytes + p2bytes + (size_t)k * p * sizeof(cplx));
    if (rc) return rc;
    cplx* partial = (cplx*)g_gemmh_scratch.dptr;
    cplx* partial2 = (cplx*)((char*)g_gemmh_scratch.dptr + pbytes);
    cplx* dC = (cplx*)((char*)g_gemmh_scratch.dptr + pbytes + p2bytes);
    const int nw = std::min(16, ntile);
    hipLaunchKernelGGL(k_gemm_h_rm, dim3(nb), dim3(64 * nw), 0, st, (const cplx*)dWT, ldw, (const cplx*)dYT, ldy, rows,
                       (int)k, (int)p, rows_per_wg, partial);
    LAUNCHCHK();
    if (nsplit > 1) {
        hipLaunchKernelGGL(k_gemm_h_reduce, dim3(ntile, nsplit), dim3(1024), 0, st, nb, ntile, (int)k, (int)p, tj,
                           (const cplx*)partial, (int64_t)ntile * 256, partial2, 0);
        LAUNCHCHK();
        hipLaunchKernelGGL(k_gemm_h_reduce, dim3(ntile, 1), dim3(1024), 0, st, nsplit, ntile, (int)k, (int)p, tj,
                           (const cplx*)partial2, (int64_t)ntile * 256, dC, 1);
    } else {
        hipLaunchKernelGGL(k_gemm_h_reduce, dim3(ntile, 1), dim3(1024), 0, st, nb, ntile, (int)k, (int)p, tj,
                           (const cplx*)partial, (int64_t)ntile * 256, dC, 1);
    }
    LAUNCHCHK();
    std::vector<nep_cdouble> tmp((size_t)k * p);
    HIPCHK(hipMemcpyAsync(tmp.data(), dC, (size_t)k * p * sizeof(cplx), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    memcpy(h_C, tmp.data(), (size_t)k * p * sizeof(cplx));
    return NEP_OK;
}


// ------------------------------------------------------------------------------------------------
// Plain dense GEMM  C = alpha op(A) op(B) + beta C  (column-major; op = none / transpose / conjugate transpose), complex128
// (nep_zgemm) and float64 (nep_dgemm).  Utility entry points: the dense-transform form of the waveguide Sylvester solver for
// shapes the prime-factor DFT + tridiagonal-scan kernels of csrc/wep.hip do not take (nx > 2048), region means / expansions
// of its A/B reference path, and hosts that want a dense product next to their device blocks.  Until round 3 these two called
// rocBLAS; they are this library's own kernel now, so that no vendor GEMM sits behind the C ABI.
// Kernel: 64 x 64 tile of C per workgroup (256 threads, 4 x 4 outputs per thread), 16-deep panels of op(A) and op(B) staged in
// LDS with the transposition / conjugation applied on the way in, FP64 FMAs on the vector pipe (FP64 MFMA and FP64 vector FMA
// have the same peak on gfx950; the tall-skinny MFMA kernels above serve the hot shapes).
namespace {
__device__ __forceinline__ double gconj(double v, bool) { return v; }
__device__ __forceinline__ cplx gconj(cplx v, bool c) { return c ? cmake(v.x, -v.y) : v; }
__device__ __forceinline__ double gzero(double*) { return 0.0; }
__device__ __forceinline__ cplx gzero(cplx*) { return cmake(0.0, 0.0); }
__device__ __forceinline__ void gfma(double& acc, double a, double b) { acc = fma(a, b, acc); }
__device__ __forceinline__ void gfma(cplx& acc, cplx a, cplx b) { cfma(acc, a, b); }
__device__ __forceinline__ double gmul(double a, double b) { return a * b; }
__device__ __forceinline__ cplx gmul(cplx a, cplx b) { return cmul(a, b); }
__device__ __forceinline__ double gadd(double a, double b) { return a + b; }
__device__ __forceinline__ cplx gadd(cplx a, cplx b) { return cadd(a, b); }
__device__ __forceinline__ bool gnonzero(double v) { return v != 0.0; }
__device__ __forceinline__ bool gnonzero(cplx v) { return v.x != 0.0 || v.y != 0.0; }

template <typename T>
__global__ __launch_bounds__(256) void k_gemm_general(int ta, int tb, int m, int n, int k, T alpha, const T* __restrict__ A,
                                                      int64_t lda, const T* __restrict__ B, int64_t ldb, T beta, T* __restrict__ C,
                                                      int64_t ldc, const int32_t* __restrict__ krange = nullptr, int kchunk = 0,
                                                      T* __restrict__ P = nullptr) {
    // krange (per 64-row tile of C: [k_lo, k_hi)): op(A) is zero outside that range on the tile's rows (block-diagonal A: the
    // diagonal-block inverses of the dense apex build, trsv_ml.hip).  kchunk > 0: split-K -- workgroup z sums k in [z kchunk,
    // (z+1) kchunk) and stores the raw partial tile into P + z m n (ld = m); k_gemm_splitk_reduce adds the slices in order.
    constexpr int TM = 64, TN = 64, TK = 16;
    int k_lo = 0, k_hi = k;
    if (krange) { k_lo = krange[2 * blockIdx.x]; k_hi = krange[2 * blockIdx.x + 1]; }
    if (kchunk > 0) { k_lo = (int)blockIdx.z * kchunk; k_hi = min(k, k_lo + kchunk); }
    __shared__ T As[TK][TM + 1];          // As[kk][i] = op(A)[i0 + i, k0 + kk]
    __shared__ T Bs[TK][TN + 1];          // Bs[kk][j] = op(B)[k0 + kk, j0 + j]
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int i0 = blockIdx.x * TM, j0 = blockIdx.y * TN;
    T acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = gzero((T*)nullptr);
    for (int k0 = k_lo; k0 < k_hi; k0 += TK) {
        // stage: 64 x 16 elements each, 4 per thread; the fast index follows the operand's storage order
        for (int t = threadIdx.x; t < TM * TK; t += 256) {
            int i, kk;
            if (ta == 0) { i = t % TM; kk = t / TM; } else { kk = t % TK; i = t / TK; }
            const int gi = i0 + i, gk = k0 + kk;
            T v = gzero((T*)nullptr);
            if (gi < m && gk < k_hi) v = ta == 0 ? A[gi + (int64_t)gk * lda] : gconj(A[gk + (int64_t)gi * lda], ta == 2);
            As[kk][i] = v;
        }
        for (int t = threadIdx.x; t < TN * TK; t += 256) {
            int j, kk;
            if (tb == 0) { kk = t % TK; j = t / TK; } else { j = t % TN; kk = t / TN; }
            const int gj = j0 + j, gk = k0 + kk;
            T v = gzero((T*)nullptr);
            if (gj < n && gk < k_hi) v = tb == 0 ? B[gk + (int64_t)gj * ldb] : gconj(B[gj + (int64_t)gk * ldb], tb == 2);
            Bs[kk][j] = v;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < TK; ++kk) {
            T av[4], bv[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) av[a] = As[kk][tx + 16 * a];
#pragma unroll
            for (int b = 0; b < 4; ++b) bv[b] = Bs[kk][ty + 16 * b];
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) gfma(acc[a][b], av[a], bv[b]);
        }
        __syncthreads();
    }
    if (kchunk > 0) {                                   // split-K: raw partial tile
        T* Pz = P + (int64_t)blockIdx.z * m * n;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int gj = j0 + ty + 16 * b;
            if (gj >= n) continue;
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const int gi = i0 + tx + 16 * a;
                if (gi < m) Pz[gi + (int64_t)gj * m] = acc[a][b];
            }
        }
        return;
    }
    const bool use_c = gnonzero(beta);
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const int gj = j0 + ty + 16 * b;
        if (gj >= n) continue;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int gi = i0 + tx + 16 * a;
            if (gi >= m) continue;
            T v = gmul(alpha, acc[a][b]);
            if (use_c) v = gadd(v, gmul(beta, C[gi + (int64_t)gj * ldc]));       // C is not read when beta == 0 (may hold NaN)
            C[gi + (int64_t)gj * ldc] = v;
        }
    }
}
// C = alpha (P_0 + P_1 + ... in this order) + beta C
__global__ __launch_bounds__(256) void k_gemm_splitk_reduce(int m, int n, int nz_, cplx alpha, const cplx* __restrict__ P, cplx beta,
                                                            cplx* __restrict__ C, int64_t ldc) {
    const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (t >= (int64_t)m * n) return;
    const int i = (int)(t % m), j = (int)(t / m);
    cplx sacc = P[t];
    for (int z = 1; z < nz_; ++z) sacc = cadd(sacc, P[(int64_t)z * m * n + t]);
    cplx v = cmul(alpha, sacc);
    if (beta.x != 0.0 || beta.y != 0.0) v = cadd(v, cmul(beta, C[i + (int64_t)j * ldc]));
    C[i + (int64_t)j * ldc] = v;
}
}  // namespace

// internal (trsv_ml.hip, dense apex build): complex GEMM without transposition with (a) a K range per 64-row tile for a block-diagonal
// A (h_krange on the HOST, 2 ints per tile, uploaded through dWork) or (b) deterministic split-K for products with few tiles and a
// long K (dWork: ksplit * m * n complex).  ksplit <= 1 and h_krange == NULL: plain nep_zgemm.
extern "C" int32_t nep_zgemm_ex(int32_t m, int32_t n, int32_t k, nep_cdouble alpha, const nep_cdouble* dA, int64_t lda,
                                const nep_cdouble* dB, int64_t ldb, nep_cdouble beta, nep_cdouble* dC, int64_t ldc,
                                const int32_t* d_krange, int32_t ksplit, nep_cdouble* dWork, nep_stream stream) {
    ARGCHK(dA && dB && dC && m >= 1 && n >= 1 && k >= 1 && lda >= m && ldb >= k && ldc >= m);
    cplx al, be; al.x = alpha.re; al.y = alpha.im; be.x = beta.re; be.y = beta.im;
    const dim3 g((unsigned)((m + 63) / 64), (unsigned)((n + 63) / 64), 1);
    if (ksplit > 1 && dWork) {
        const int kchunk = ((k + ksplit - 1) / ksplit + 15) / 16 * 16;
        const int nzs = (k + kchunk - 1) / kchunk;
        hipLaunchKernelGGL(k_gemm_general<cplx>, dim3(g.x, g.y, (unsigned)nzs), dim3(256), 0, as_stream(stream), 0, 0, (int)m, (int)n, (int)k, al,
                           (const cplx*)dA, lda, (const cplx*)dB, ldb, be, (cplx*)dC, ldc, (const int32_t*)nullptr, kchunk, (cplx*)dWork);
        LAUNCHCHK();
        hipLaunchKernelGGL(k_gemm_splitk_reduce, dim3((unsigned)(((int64_t)m * n + 255) / 256)), dim3(256), 0, as_stream(stream), (int)m, (int)n, nzs,
                           al, (const cplx*)dWork, be, (cplx*)dC, ldc);
        LAUNCHCHK();
        return NEP_OK;
    }
    hipLaunchKernelGGL(k_gemm_general<cplx>, g, dim3(256), 0, as_stream(stream), 0, 0, (int)m, (int)n, (int)k, al, (const cplx*)dA, lda,
                       (const cplx*)dB, ldb, be, (cplx*)dC, ldc, d_krange, 0, (cplx*)nullptr);
    LAUNCHCHK();
    return NEP_OK;
}

// C = alpha op(A) op(B) + beta C with the K range split over `ksplit` workgroups per tile (deterministic: the slices are summed in
// order): products of a few tiles with a long reduction -- the k x k Gram block Q^H A1 of Beyn's method (one 32 x 32 tile, K = n).
// dWork: ksplit * m * n complex.
// ---- dense complex inverse on the device: in-place Gauss-Jordan with partial pivoting ------------------------------------------
// (the mm x mm Sylvester-SMW matrix of the waveguide preconditioner, waveguide_preconditioner.jl:221-313: mm = 1517, 0.14 s of
// numpy.linalg.inv per tiar run until round 4.)  Column-major A, n steps of two launches:
//   k_gj_pivot  (one workgroup): p = argmax_{i >= j} |A[i, j]| (ties: smallest i), rows j <-> p, f = column j, column j <- e_j,
//               row j <- row j / pivot
//   k_gj_update (grid):          A[i, c] -= f[i] A[j, c]  for i != j
// and at the end the column exchanges in reverse order.  Deterministic; a zero / non-finite pivot is reported through *info.
namespace {
__global__ __launch_bounds__(256) void k_conjt_addi(int n, const cplx* __restrict__ M, int64_t ldm, double add, cplx* __restrict__ out, int64_t ldo) {
    // out[r, c] = conj(M[c, r]) + add (r == c): 32 x 32 tiles through LDS
    __shared__ cplx t[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int r0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    for (int q = ty; q < 32; q += 8) {            // read M[c0 + tx, r0 + q]  (row index fast)
        const int rr = c0 + tx, cc = r0 + q;
        t[q][tx] = (rr < n && cc < n) ? M[rr + (int64_t)cc * ldm] : cmake(0.0, 0.0);
    }
    __syncthreads();
    for (int q = ty; q < 32; q += 8) {            // write out[r0 + tx, c0 + q] = conj(M[c0 + q, r0 + tx]) = conj(t[tx][q])
        const int r = r0 + tx, c = c0 + q;
        if (r < n && c < n) { cplx v = t[tx][q]; out[r + (int64_t)c * ldo] = cmake(v.x + (r == c ? add : 0.0), -v.y); }
    }
}
__global__ __launch_bounds__(1024) void k_gj_pivot(int n, int j, cplx* __restrict__ A, int64_t lda, cplx* __restrict__ f, int* __restrict__ piv,
                                                   int* __restrict__ info) {
    __shared__ double sv[16]; __shared__ int si[16]; __shared__ int sp;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    double best = -1.0; int bi = n;
    for (int i = j + tid; i < n; i += 1024) {
        const cplx a = A[i + (int64_t)j * lda];
        const double m = a.x * a.x + a.y * a.y;
        if (m > best || (m == best && i < bi)) { best = m; bi = i; }
    }
    for (int off = 32; off > 0; off >>= 1) {
        const double ob = __shfl_xor(best, off, 64); const int oi = __shfl_xor(bi, off, 64);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    if (lane == 0) { sv[wv] = best; si[wv] = bi; }
    __syncthreads();
    if (tid == 0) {
        double b = sv[0]; int ix = si[0];
        for (int q = 1; q < 16; ++q) if (sv[q] > b || (sv[q] == b && si[q] < ix)) { b = sv[q]; ix = si[q]; }
        if (!(b > 0.0) || !isfinite(b)) { if (*info == 0) *info = j + 1; ix = j; }
        sp = ix; piv[j] = ix;
    }
    __syncthreads();
    const int p = sp;
    if (p != j)
        for (int c = tid; c < n; c += 1024) {
            const cplx a = A[j + (int64_t)c * lda], b = A[p + (int64_t)c * lda];
            A[j + (int64_t)c * lda] = b; A[p + (int64_t)c * lda] = a;
        }
    __syncthreads();
    const cplx pv = A[j + (int64_t)j * lda];
    const double den = pv.x * pv.x + pv.y * pv.y;
    const cplx inv = den > 0.0 ? cmake(pv.x / den, -pv.y / den) : cmake(0.0, 0.0);
    __syncthreads();
    for (int i = tid; i < n; i += 1024) {                    // f = column j (0 at the pivot row), column j <- e_j
        const cplx a = A[i + (int64_t)j * lda];
        f[i] = i == j ? cmake(0.0, 0.0) : a;
        A[i + (int64_t)j * lda] = i == j ? cmake(1.0, 0.0) : cmake(0.0, 0.0);
    }
    __syncthreads();
    for (int c = tid; c < n; c += 1024) A[j + (int64_t)c * lda] = cmul(A[j + (int64_t)c * lda], inv);
}
__global__ __launch_bounds__(256) void k_gj_update(int n, int j, cplx* __restrict__ A, int64_t lda, const cplx* __restrict__ f) {
    // thread = row i of a 256-row strip, workgroup y = a group of 8 columns
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int c0 = blockIdx.y * 8;
    if (i >= n || i == j) return;
    const cplx fi = f[i];
    if (fi.x == 0.0 && fi.y == 0.0) return;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int c = c0 + q;
        if (c < n) {
            const cplx r = A[j + (int64_t)c * lda];
            cplx a = A[i + (int64_t)c * lda];
            a.x -= fi.x * r.x - fi.y * r.y; a.y -= fi.x * r.y + fi.y * r.x;
            A[i + (int64_t)c * lda] = a;
        }
    }
}
__global__ __launch_bounds__(256) void k_gj_unpermute(int n, cplx* __restrict__ A, int64_t lda, const int* __restrict__ piv) {
    const int i = blockIdx.x * 256 + threadIdx.x;           // row i: the column exchanges of the pivoting, last first
    if (i >= n) return;
    for (int j = n - 1; j >= 0; --j) {
        const int p = piv[j];
        if (p != j) { const cplx a = A[i + (int64_t)j * lda], b = A[i + (int64_t)p * lda]; A[i + (int64_t)j * lda] = b; A[i + (int64_t)p * lda] = a; }
    }
}
}  // namespace
// dOut (n x n, column-major, ld ldo) = inv(M + add_identity I)^H  = inv((M + add_identity I)^H), M n x n column-major (ld ldm) on the
// device; dWork: n complex + (n + 1) int32 of device scratch ((n + 1) complex + ... : 2 n + 2 complex is enough).  *h_info = 0, or
// 1 + the step whose pivot was zero / not finite (host value: the call synchronises the stream once at its end).
extern "C" int32_t nep_zinv_h_dev(int32_t n, const nep_cdouble* dM, int64_t ldm, double add_identity, nep_cdouble* dOut, int64_t ldo,
                                  nep_cdouble* dWork, int32_t* h_info, nep_stream stream) {
    ARGCHK(dM && dOut && dWork && h_info && n >= 1 && ldm >= n && ldo >= n);
    hipStream_t st = as_stream(stream);
    cplx* f = (cplx*)dWork; int* piv = (int*)(f + n); int* info = piv + n;
    HIPCHK(hipMemsetAsync(info, 0, sizeof(int), st));
    hipLaunchKernelGGL(k_conjt_addi, dim3((unsigned)((n + 31) / 32), (unsigned)((n + 31) / 32)), dim3(256), 0, st, (int)n, (const cplx*)dM, ldm,
                       add_identity, (cplx*)dOut, ldo);
    LAUNCHCHK();
    const dim3 gu((unsigned)((n + 255) / 256), (unsigned)((n + 7) / 8));
    for (int j = 0; j < n; ++j) {
        hipLaunchKernelGGL(k_gj_pivot, dim3(1), dim3(1024), 0, st, (int)n, j, (cplx*)dOut, ldo, f, piv, info);
        hipLaunchKernelGGL(k_gj_update, gu, dim3(256), 0, st, (int)n, j, (cplx*)dOut, ldo, (const cplx*)f);
    }
    LAUNCHCHK();
    hipLaunchKernelGGL(k_gj_unpermute, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (int)n, (cplx*)dOut, ldo, (const int*)piv);
    LAUNCHCHK();
    int hi = 0;
    HIPCHK(hipMemcpyAsync(&hi, info, sizeof(int), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    *h_info = hi;
    return NEP_OK;
}

extern "C" int32_t nep_zgemm_sk(int32_t transa, int32_t transb, int32_t m, int32_t n, int32_t k, nep_cdouble alpha,
                                const nep_cdouble* dA, int64_t lda, const nep_cdouble* dB, int64_t ldb, nep_cdouble beta,
                                nep_cdouble* dC, int64_t ldc, int32_t ksplit, nep_cdouble* dWork, nep_stream stream) {
    ARGCHK(dA && dB && dC && dWork && m >= 1 && n >= 1 && k >= 1 && ksplit >= 1);
    ARGCHK(transa >= 0 && transa <= 2 && transb >= 0 && transb <= 2);
    ARGCHK(lda >= (transa ? k : m) && ldb >= (transb ? n : k) && ldc >= m);
    cplx al, be; al.x = alpha.re; al.y = alpha.im; be.x = beta.re; be.y = beta.im;
    const int kchunk = ((k + ksplit - 1) / ksplit + 15) / 16 * 16;
    const int nzs = (k + kchunk - 1) / kchunk;
    hipLaunchKernelGGL(k_gemm_general<cplx>, dim3((unsigned)((m + 63) / 64), (unsigned)((n + 63) / 64), (unsigned)nzs), dim3(256), 0,
                       as_stream(stream), (int)transa, (int)transb, (int)m, (int)n, (int)k, al, (const cplx*)dA, lda, (const cplx*)dB, ldb, be,
                       (cplx*)dC, ldc, (const int32_t*)nullptr, kchunk, (cplx*)dWork);
    LAUNCHCHK();
    hipLaunchKernelGGL(k_gemm_splitk_reduce, dim3((unsigned)(((int64_t)m * n + 255) / 256)), dim3(256), 0, as_stream(stream), (int)m, (int)n, nzs,
                       al, (const cplx*)dWork, be, (cplx*)dC, ldc);
    LAUNCHCHK();
    return NEP_OK;
}

extern "C" int32_t nep_zgemm(int32_t transa, int32_t transb, int32_t m, int32_t n, int32_t k, nep_cdouble alpha,
                             const nep_cdouble* dA, int64_t lda, const nep_cdouble* dB, int64_t ldb, nep_cdouble beta,
                             nep_cdouble* dC, int64_t ldc, nep_stream stream) {
    ARGCHK(dA && dB && dC && m >= 1 && n >= 1 && k >= 1);
    ARGCHK(transa >= 0 && transa <= 2 && transb >= 0 && transb <= 2);
    ARGCHK(lda >= (transa ? k : m) && ldb >= (transb ? n : k) && ldc >= m);
    cplx al, be; al.x = alpha.re; al.y = alpha.im; be.x = beta.re; be.y = beta.im;
    hipLaunchKernelGGL(k_gemm_general<cplx>, dim3((unsigned)((m + 63) / 64), (unsigned)((n + 63) / 64)), dim3(256), 0, as_stream(stream),
                       (int)transa, (int)transb, (int)m, (int)n, (int)k, al, (const cplx*)dA, lda, (const cplx*)dB, ldb, be, (cplx*)dC, ldc);
    LAUNCHCHK();
    return NEP_OK;
}

// real counterpart: a complex column-major m x n block IS a real 2m x n block (re / im interleaved along the rows), so X * W
// with a REAL W -- the sine transform of the waveguide Sylvester solver -- costs half the flops of a complex product.
extern "C" int32_t nep_dgemm(int32_t transa, int32_t transb, int32_t m, int32_t n, int32_t k, double alpha,
                             const double* dA, int64_t lda, const double* dB, int64_t ldb, double beta,
                             double* dC, int64_t ldc, nep_stream stream) {
    ARGCHK(dA && dB && dC && m >= 1 && n >= 1 && k >= 1);
    ARGCHK(transa >= 0 && transa <= 1 && transb >= 0 && transb <= 1);
    ARGCHK(lda >= (transa ? k : m) && ldb >= (transb ? n : k) && ldc >= m);
    hipLaunchKernelGGL(k_gemm_general<double>, dim3((unsigned)((m + 63) / 64), (unsigned)((n + 63) / 64)), dim3(256), 0, as_stream(stream),
                       (int)transa, (int)transb, (int)m, (int)n, (int)k, alpha, dA, lda, dB, ldb, beta, dC, ldc);
    LAUNCHCHK();
    return NEP_OK;
}
