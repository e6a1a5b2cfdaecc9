// libnepmi355: the Sylvester solve of the waveguide preconditioner (config C5) without any library GEMM or FFT.
//
// replaces: solve_wg_sylvester_fft! src/gallery_extra/waveguide/waveguide_preconditioner.jl:120-219 -- the reference
//           diagonalises A(sigma) X + X B = C with an FFT along z (circulant A, length nz) and a sine transform along x
//           (tridiagonal Toeplitz B, realised there as an FFT of length 2(nx+1)).  Round 1 of this backend applied both transforms as
//           dense GEMMs (4 x 4-8 GFLOP per solve on rocBLAS, 0.41 ms at nz = 999, nx = 1003).
//
// Here only the z direction is diagonalised; what is left is, for every z-mode i, the TRIDIAGONAL system
//           (d_i I + B) x_i = c_i,        B = tridiag(1, -2, 1) / hx^2,   d_i = eigenvalue i of A(sigma) + (sigma^2 + k_bar)
// which is mathematically the same operator as W diag(1 / (d_i + s_j)) W with the sine matrix W (s_j = eigenvalues of B).
//   1. k_dft_cols<forward>   DFT along z of every column with the Good-Thomas prime-factor algorithm: nz = N1 N2 with
//      gcd(N1, N2) = 1 (999 = 27 * 37, 299 = 13 * 23, 105 = 15 * 7) is a TWIDDLE-FREE N1 x N2 two-dimensional DFT under the
//      Ruritanian / CRT index maps; both small DFTs are dense matrix products out of LDS (27 + 37 = 64 complex multiplies per
//      entry instead of 999).  A workgroup takes 4 columns and writes the result transposed (x fastest), 64 bytes per mode.
//   2. k_tridiag_modes       one wave per mode: Thomas forward / backward substitution with precomputed pivots as two
//      first-order affine recurrences, each evaluated by a segmented scan inside the wave (16 consecutive x per lane).
//   3. k_dft_cols<inverse>   reads the transposed block back, inverse DFT, writes z fastest.
// Data moved per solve: about 10 passes over the 16 nx nz byte block (HBM/L2-bound, ~0.03 ms) instead of 24 GFLOP.
#include "common.h"
#include <vector>
#include <cmath>
#include <numeric>

struct nep_wep_sylv {
    int nz = 0, nx = 0, N1 = 0, N2 = 0, cols = 4;
    int ldt = 0, lseg = 0;        // row length / log2(SEG) of the x-fastest blocks below (TLay)
    int32_t* d_in = nullptr;      // nz: input position of (n1, n2)
    int32_t* d_out = nullptr;     // nz: output position of (k1, k2)
    int32_t* d_in_inv = nullptr;  // nz: (n1, n2) slot of input position z (coalesced loads, scattered LDS writes)
    cplx* d_w1 = nullptr;         // N1 roots exp(-2 pi i j / N1)
    cplx* d_w2 = nullptr;         // N2 roots
    cplx* d_m = nullptr;          // nz x ldt (x fastest, TLay order): forward multipliers m_j = b / dtilde_{j-1}
    cplx* d_dinv = nullptr;       // nz x nx: 1 / dtilde_j
    cplx* d_T = nullptr;          // nz x nx work (x fastest)
    cplx* d_T2 = nullptr;         // second block of that shape + the small vectors of nep_wep_smw_apply (allocated on first use)
    int smw_N = 0;                // region count the small vectors were sized for
    double b = 0.0;
};

// Layout of the transposed block T (and of the Thomas factors read with it): mode i at T + i * ldt.  The tridiagonal kernel gives every
// lane SEG = 2^lseg consecutive x; with x stored in order a wave-wide 16-byte load touched 64 different lines (lane stride 16 SEG bytes)
// and the 16 loads of a lane came back to each line after 3 x 16 KB per wave had passed through the 32 KB L1.  For SEG >= 4 the row is
// stored in pieces of four x as [piece within the segment][lane][4]: the four x a DFT workgroup writes stay one 64-byte piece, and the
// tridiagonal kernel's loads are contiguous over the wave (ldt = 64 SEG).  SEG < 4 (nx <= 128): plain order, ldt = nx.
struct TLay { int ldt, lseg; };
__device__ __forceinline__ int tpos(int x, const TLay tl) {
    if (tl.lseg < 2) return x;
    const int seg_mask = (1 << tl.lseg) - 1;
    return (((x & seg_mask) >> 2) << 8) + ((x >> tl.lseg) << 2) + (x & 3);
}

// workgroup -> column group, XCD-contiguous (workgroup id % 8 = XCD): the groups an XCD works on at one time are neighbours in x, so
// the 64-byte pieces they write to (read from) one mode row of the transposed block are neighbours too -- its L2 sees 2 KB runs per
// row instead of every other 64-byte piece of a line belonging to another XCD (NEP_WEP_DFT_XCD=0: group = workgroup id as before)
__device__ __forceinline__ int dft_group(int xcd_order) {
    const int b = blockIdx.x, nb = gridDim.x;
    if (!xcd_order) return b;
    const int x = b & 7, i = b >> 3, per = nb >> 3, rem = nb & 7;
    return x * per + (x < rem ? x : rem) + i;
}

// ---- prime-factor DFT of COLS columns per workgroup ---------------------------------------------------------------------
// FWD = true : in  X (z fastest, column x at X + x*nz), out T (x fastest, mode i at T + i*nx), exponent sign `sgn`
// FWD = false: in  T (x fastest), out X (z fastest)
template <bool FWD, int COLS>
__global__ __launch_bounds__(1024) void k_dft_cols(int nz, int nx, int N1, int N2, const int32_t* __restrict__ in_idx,
                                                   const int32_t* __restrict__ out_idx, const cplx* __restrict__ w1,
                                                   const cplx* __restrict__ w2, double sgn, double scale,
                                                   const cplx* __restrict__ src, cplx* __restrict__ dst, int xcd_order, TLay tl) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    cplx* xs = (cplx*)smem_raw;                 // [q][COLS], natural (n1, n2) order q = n1 N2 + n2
    cplx* ts = xs + (size_t)COLS * nz;          // [q][COLS], q = k1 N2 + n2
    cplx* r1 = ts + (size_t)COLS * nz;          // N1 roots
    cplx* r2 = r1 + N1;                         // N2 roots
    const int x0 = dft_group(xcd_order) * COLS;
    const int nc = min(COLS, nx - x0);
    const int nt = blockDim.x;
    for (int t = threadIdx.x; t < N1; t += nt) r1[t] = cmake(w1[t].x, sgn * w1[t].y);
    for (int t = threadIdx.x; t < N2; t += nt) r2[t] = cmake(w2[t].x, sgn * w2[t].y);
    // load element (n1, n2) of the COLS columns (the COLS values of one q sit side by side: one root read serves COLS MACs)
    if (FWD) {
        for (int t = threadIdx.x; t < COLS * nz; t += nt) {
            const int c = t / nz, q = t - c * nz;
            xs[q * COLS + c] = c < nc ? src[(int64_t)(x0 + c) * nz + in_idx[q]] : cmake(0.0, 0.0);
        }
    } else {
        for (int t = threadIdx.x; t < COLS * nz; t += nt) {      // consecutive threads -> consecutive x of one mode
            const int q = t / COLS, c = t - q * COLS;
            xs[t] = c < nc ? src[(int64_t)in_idx[q] * tl.ldt + tpos(x0 + c, tl)] : cmake(0.0, 0.0);
        }
    }
    __syncthreads();
    // stage 1: ts[k1, n2] = sum_n1 xs[n1, n2] r1^(n1 k1)
    for (int q = threadIdx.x; q < nz; q += nt) {
        const int k1 = q / N2, n2 = q - k1 * N2;
        const cplx* xc = xs + (size_t)n2 * COLS;
        cplx acc[COLS];
#pragma unroll
        for (int c = 0; c < COLS; ++c) acc[c] = cmake(0.0, 0.0);
        int e = 0;
        for (int n1 = 0; n1 < N1; ++n1) {
            const cplx w = r1[e];
            const cplx* xp = xc + (size_t)n1 * N2 * COLS;
#pragma unroll
            for (int c = 0; c < COLS; ++c) cfma(acc[c], xp[c], w);
            e += k1; if (e >= N1) e -= N1;
        }
#pragma unroll
        for (int c = 0; c < COLS; ++c) ts[(size_t)q * COLS + c] = acc[c];
    }
    __syncthreads();
    // stage 2: out[k1, k2] = sum_n2 ts[k1, n2] r2^(n2 k2)
    for (int q = threadIdx.x; q < nz; q += nt) {
        const int k1 = q / N2, k2 = q - k1 * N2;
        const cplx* tc = ts + (size_t)k1 * N2 * COLS;
        cplx acc[COLS];
#pragma unroll
        for (int c = 0; c < COLS; ++c) acc[c] = cmake(0.0, 0.0);
        int e = 0;
        for (int n2 = 0; n2 < N2; ++n2) {
            const cplx w = r2[e];
#pragma unroll
            for (int c = 0; c < COLS; ++c) cfma(acc[c], tc[(size_t)n2 * COLS + c], w);
            e += k2; if (e >= N2) e -= N2;
        }
        const int o = out_idx[q];
#pragma unroll
        for (int c = 0; c < COLS; ++c)
            if (c < nc) {
                const cplx v = cmake(scale * acc[c].x, scale * acc[c].y);
                if (FWD) dst[(int64_t)o * tl.ldt + tpos(x0 + c, tl)] = v;          // COLS consecutive x of one mode: COLS * 16 bytes
                else dst[(int64_t)(x0 + c) * nz + o] = v;
            }
    }
}

// Register-blocked form of k_dft_cols (COLS = 4): a thread owns KB = 3 outputs that share their inputs -- stage 1 (k1, k1+G1,
// k1+2 G1) of one n2, stage 2 (k2, k2+G2, k2+2 G2) of one k1 -- so one 64-byte read of the four columns feeds 12 complex
// multiply-adds instead of 4 (the one-output-per-thread form moves 80 bytes of LDS per 32 flops and is LDS-bound at 35 us;
// this one moves 112 bytes per 96 flops).  Measured at 999 x 1003: 31.5 / 33.5 us (forward / inverse) against 35.8 / 34.0 us -- the kernel
// is bound by exposed latency (one 128 KB workgroup per CU, six waves), not by LDS bytes; `NEP_WEP_DFT_RB=0` selects the old form.
#ifdef WEP_PROF
__device__ unsigned long long g_dft_prof[16];
extern "C" int32_t nep_wep_prof_read(unsigned long long* out, int32_t reset) {
    HIPCHK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dft_prof), sizeof(unsigned long long) * 16));
    if (reset) { unsigned long long z[16] = {0}; HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_dft_prof), z, sizeof(z))); }
    return NEP_OK;
}
#define WP(x) const long long x = clock64()
#else
#define WP(x)
#endif
template <bool FWD>
__global__ __launch_bounds__(384) void k_dft_cols_rb(int nz, int nx, int N1, int N2, const int32_t* __restrict__ in_idx,
                                                      const int32_t* __restrict__ in_inv,
                                                      const int32_t* __restrict__ out_idx, const cplx* __restrict__ w1,
                                                      const cplx* __restrict__ w2, double sgn, double scale,
                                                      const cplx* __restrict__ src, cplx* __restrict__ dst, int xcd_order, TLay tl) {
    constexpr int COLS = 4, KB = 3;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    cplx* xs = (cplx*)smem_raw;                 // [q][COLS], q = n1 N2 + n2
    cplx* ts = xs + (size_t)COLS * nz;          // [q][COLS], q = k1 N2 + n2
    cplx* r1 = ts + (size_t)COLS * nz;
    cplx* r2 = r1 + N1;
    WP(tp0);
    const int x0 = dft_group(xcd_order) * COLS;
    const int nc = min(COLS, nx - x0);
    const int nt = blockDim.x;
    for (int t = threadIdx.x; t < N1; t += nt) r1[t] = cmake(w1[t].x, sgn * w1[t].y);
    for (int t = threadIdx.x; t < N2; t += nt) r2[t] = cmake(w2[t].x, sgn * w2[t].y);
    if (FWD) {
        // consecutive threads read consecutive z (coalesced) and scatter into the (n1, n2) slot in LDS -- reading THROUGH the
        // index map instead touched a different 64-byte line with every 16-byte load
        // (four trips' loads are issued together, unconditionally with a clamped column: a trip's index + value round trip was
        // waited for before the next trip's loads went out -- eleven serial round trips per workgroup)
        for (int t = threadIdx.x; t < COLS * nz; t += 4 * nt) {
            int slot[4]; cplx v[4]; bool on[4], live[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int tt = t + u * nt;
                live[u] = tt < COLS * nz;
                const int tc = live[u] ? tt : t;
                const int c = tc / nz, z = tc - c * nz;
                on[u] = c < nc;
                slot[u] = in_inv[z] * COLS + c;
                v[u] = src[(int64_t)(x0 + (on[u] ? c : 0)) * nz + z];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) if (live[u]) xs[slot[u]] = on[u] ? v[u] : cmake(0.0, 0.0);
        }
    } else {
        for (int t = threadIdx.x; t < COLS * nz; t += 4 * nt) {
            cplx v[4]; bool on[4], live[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int tt = t + u * nt;
                live[u] = tt < COLS * nz;
                const int tc = live[u] ? tt : t;
                const int q = tc / COLS, c = tc - q * COLS;
                on[u] = c < nc;
                v[u] = src[(int64_t)in_idx[q] * tl.ldt + tpos(x0 + (on[u] ? c : 0), tl)];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) if (live[u]) xs[t + u * nt] = on[u] ? v[u] : cmake(0.0, 0.0);
        }
    }
    __syncthreads();
    WP(tp1);
    // stage 1: ts[k1, n2] = sum_n1 xs[n1, n2] r1^(n1 k1), three k1 per thread
    const int G1 = (N1 + KB - 1) / KB;
    for (int t = threadIdx.x; t < G1 * N2; t += nt) {
        const int g = t / N2, n2 = t - g * N2;
        int kk[KB], e[KB];
        cplx acc[KB][COLS];
#pragma unroll
        for (int b = 0; b < KB; ++b) {
            kk[b] = g + b * G1; e[b] = 0;
#pragma unroll
            for (int c = 0; c < COLS; ++c) acc[b][c] = cmake(0.0, 0.0);
        }
        const cplx* xc = xs + (size_t)n2 * COLS;
        for (int n1 = 0; n1 < N1; ++n1) {
            const cplx* xp = xc + (size_t)n1 * N2 * COLS;
            const cplx x0v = xp[0], x1v = xp[1], x2v = xp[2], x3v = xp[3];
#pragma unroll
            for (int b = 0; b < KB; ++b) {
                const cplx w = r1[e[b] < N1 ? e[b] : 0];            // (groups beyond N1 are computed and dropped)
                cfma(acc[b][0], x0v, w); cfma(acc[b][1], x1v, w); cfma(acc[b][2], x2v, w); cfma(acc[b][3], x3v, w);
                e[b] += kk[b]; if (e[b] >= N1) e[b] -= N1;
            }
        }
#pragma unroll
        for (int b = 0; b < KB; ++b)
            if (kk[b] < N1) {
                cplx* o = ts + ((size_t)kk[b] * N2 + n2) * COLS;
                o[0] = acc[b][0]; o[1] = acc[b][1]; o[2] = acc[b][2]; o[3] = acc[b][3];
            }
    }
    __syncthreads();
    WP(tp2);
    // stage 2: out[k1, k2] = sum_n2 ts[k1, n2] r2^(n2 k2), three k2 per thread
    const int G2 = (N2 + KB - 1) / KB;
    for (int t = threadIdx.x; t < N1 * G2; t += nt) {
        const int k1 = t / G2, g = t - k1 * G2;
        int kk[KB], e[KB];
        cplx acc[KB][COLS];
#pragma unroll
        for (int b = 0; b < KB; ++b) {
            kk[b] = g + b * G2; e[b] = 0;
#pragma unroll
            for (int c = 0; c < COLS; ++c) acc[b][c] = cmake(0.0, 0.0);
        }
        const cplx* tc = ts + (size_t)k1 * N2 * COLS;
        int oidx[KB];                                   // output positions: requested before the n2 loop, not behind it
#pragma unroll
        for (int b = 0; b < KB; ++b) oidx[b] = out_idx[k1 * N2 + (kk[b] < N2 ? kk[b] : 0)];
        for (int n2 = 0; n2 < N2; ++n2) {
            const cplx* tp = tc + (size_t)n2 * COLS;
            const cplx t0v = tp[0], t1v = tp[1], t2v = tp[2], t3v = tp[3];
#pragma unroll
            for (int b = 0; b < KB; ++b) {
                const int eb = e[b] < N2 ? e[b] : 0;
                const cplx w = r2[eb];
                cfma(acc[b][0], t0v, w); cfma(acc[b][1], t1v, w); cfma(acc[b][2], t2v, w); cfma(acc[b][3], t3v, w);
                e[b] += kk[b]; if (e[b] >= N2) e[b] -= N2;
            }
        }
#pragma unroll
        for (int b = 0; b < KB; ++b)
            if (kk[b] < N2) {
                const int o = oidx[b];
                if (FWD) {
#pragma unroll
                    for (int c = 0; c < COLS; ++c)
                        if (c < nc) dst[(int64_t)o * tl.ldt + tpos(x0 + c, tl)] = cmake(scale * acc[b][c].x, scale * acc[b][c].y);
                } else {
                    // (staging the result in LDS for a coalesced store was measured: 35.3 us against 33.5 us for this direct store)
#pragma unroll
                    for (int c = 0; c < COLS; ++c)
                        if (c < nc) dst[(int64_t)(x0 + c) * nz + o] = cmake(scale * acc[b][c].x, scale * acc[b][c].y);
                }
            }
    }
#ifdef WEP_PROF
    __syncthreads();
    if (threadIdx.x == 0) {
        const long long tp3 = clock64();
        const int o = FWD ? 0 : 8;
        atomicAdd(&g_dft_prof[o + 0], (unsigned long long)(tp1 - tp0)); atomicAdd(&g_dft_prof[o + 1], (unsigned long long)(tp2 - tp1));
        atomicAdd(&g_dft_prof[o + 2], (unsigned long long)(tp3 - tp2)); atomicAdd(&g_dft_prof[o + 3], 1ull);
        atomicMax(&g_dft_prof[o + 4], (unsigned long long)(tp3 - tp0));
    }
#endif
}

// ---- the two dense stages by the symmetry of the roots (N1, N2 odd) --------------------------------------------------------------
// For odd N the roots pair up: w^(n k) and w^((N-n) k) are conjugates, and so are w^(n k) and w^(n (N-k)).  With
//   s_n = x_n + x_(N-n),  d_n = x_n - x_(N-n)   (n = 1 .. H, H = (N-1)/2),
//   A_k = x_0 + sum_n cos(2 pi n k / N) s_n,     B_k = sum_n sin(2 pi n k / N) d_n       (REAL coefficients, complex data)
// the outputs are X_k = A_k - i sgn B_k and X_(N-k) = A_k + i sgn B_k (exponent -i sgn theta) and X_0 = x_0 + sum_n s_n:
// H^2 real-by-complex pairs (4 FMA) make 2 H outputs where the plain dense stage spends N^2 complex multiply-adds (4 FMA each) on
// N -- 3.9x fewer flops at N = 37 and 27, the same two stages, index maps and stores otherwise.  One LDS buffer: stage 1 keeps its
// results in registers across a barrier and overwrites its input (every work item is in flight at once: the launch sizes the
// workgroup to the number of items), which also leaves room for a second workgroup per CU.  Layout [column][q]: consecutive lanes
// (consecutive n2 / k1) read consecutive 16-byte words.  A thread owns KB values of k (both members of each pair) of one n2 / k1.
// dft_sym_stages: both stages on xs ([c][q], q = n1 N2 + n2, overwritten), roots c1 / c2 as (cos, sin)(2 pi e / N); the result
// X[k1, k2] of column c is handed to put(out_idx[k1 N2 + k2], c, re, im).  blockDim.x >= max(G1 N2, G2 N1) items.
template <int COLS, int KB, class Put>
__device__ __forceinline__ void dft_sym_stages(cplx* __restrict__ xs, const cplx* __restrict__ c1, const cplx* __restrict__ c2, int nz,
                                               int N1, int N2, double sgn, const int32_t* __restrict__ out_idx, Put put) {
    cplx A[KB][COLS], B[KB][COLS], S0[COLS];
    int kk[KB], e[KB];
    // ---- stage 1 (over n1, for one n2): item = (g, n2), k1 = 1 + g + b G1
    const int H1 = (N1 - 1) / 2, G1 = (H1 + KB - 1) / KB;
    const int t1 = threadIdx.x;
    const bool act1 = t1 < G1 * N2;
    const int g1 = act1 ? t1 / N2 : 0, n2 = act1 ? t1 - g1 * N2 : 0;
    if (act1) {
#pragma unroll
        for (int c = 0; c < COLS; ++c) S0[c] = xs[(size_t)c * nz + n2];
#pragma unroll
        for (int b = 0; b < KB; ++b) {
            kk[b] = 1 + g1 + b * G1; if (kk[b] > H1) kk[b] = 0;
            e[b] = kk[b];
#pragma unroll
            for (int c = 0; c < COLS; ++c) { A[b][c] = S0[c]; B[b][c] = cmake(0.0, 0.0); }
        }
        for (int n = 1; n <= H1; ++n) {
            const cplx* pa = xs + n * N2 + n2;
            const cplx* pb = xs + (N1 - n) * N2 + n2;
            cplx sv[COLS], dv[COLS];
#pragma unroll
            for (int c = 0; c < COLS; ++c) {
                const cplx xa = pa[(size_t)c * nz], xb = pb[(size_t)c * nz];
                sv[c] = cadd(xa, xb); dv[c] = csub(xa, xb);
                S0[c] = cadd(S0[c], sv[c]);
            }
#pragma unroll
            for (int b = 0; b < KB; ++b) {
                const cplx w = c1[e[b]];
#pragma unroll
                for (int c = 0; c < COLS; ++c) {
                    A[b][c].x = fma(w.x, sv[c].x, A[b][c].x); A[b][c].y = fma(w.x, sv[c].y, A[b][c].y);
                    B[b][c].x = fma(w.y, dv[c].x, B[b][c].x); B[b][c].y = fma(w.y, dv[c].y, B[b][c].y);
                }
                e[b] += kk[b]; if (e[b] >= N1) e[b] -= N1;
            }
        }
    }
    __syncthreads();                                   // every input has been read: the results go where the inputs were
    if (act1) {
        if (g1 == 0) {
#pragma unroll
            for (int c = 0; c < COLS; ++c) xs[(size_t)c * nz + n2] = S0[c];
        }
#pragma unroll
        for (int b = 0; b < KB; ++b)
            if (kk[b]) {
#pragma unroll
                for (int c = 0; c < COLS; ++c) {
                    const double bx = sgn * B[b][c].y, by = -sgn * B[b][c].x;          // -i sgn B
                    xs[(size_t)c * nz + kk[b] * N2 + n2] = cmake(A[b][c].x + bx, A[b][c].y + by);
                    xs[(size_t)c * nz + (N1 - kk[b]) * N2 + n2] = cmake(A[b][c].x - bx, A[b][c].y - by);
                }
            }
    }
    __syncthreads();
    // ---- stage 2 (over n2, for one k1): item = (g, k1), k2 = 1 + g + b G2
    const int H2 = (N2 - 1) / 2, G2 = (H2 + KB - 1) / KB;
    const int t2 = threadIdx.x;
    if (t2 < G2 * N1) {
        const int g2 = t2 / N1, k1 = t2 - g2 * N1;
        const cplx* row = xs + k1 * N2;
        int oa[KB], ob[KB];
#pragma unroll
        for (int b = 0; b < KB; ++b) {
            kk[b] = 1 + g2 + b * G2; if (kk[b] > H2) kk[b] = 0;
            e[b] = kk[b];
            oa[b] = out_idx[k1 * N2 + kk[b]]; ob[b] = out_idx[k1 * N2 + (kk[b] ? N2 - kk[b] : 0)];
        }
        const int o0 = out_idx[k1 * N2];
#pragma unroll
        for (int c = 0; c < COLS; ++c) S0[c] = row[(size_t)c * nz];
#pragma unroll
        for (int b = 0; b < KB; ++b)
#pragma unroll
            for (int c = 0; c < COLS; ++c) { A[b][c] = S0[c]; B[b][c] = cmake(0.0, 0.0); }
        for (int n = 1; n <= H2; ++n) {
            const cplx* pa = row + n;
            const cplx* pb = row + (N2 - n);
            cplx sv[COLS], dv[COLS];
#pragma unroll
            for (int c = 0; c < COLS; ++c) {
                const cplx xa = pa[(size_t)c * nz], xb = pb[(size_t)c * nz];
                sv[c] = cadd(xa, xb); dv[c] = csub(xa, xb);
                S0[c] = cadd(S0[c], sv[c]);
            }
#pragma unroll
            for (int b = 0; b < KB; ++b) {
                const cplx w = c2[e[b]];
#pragma unroll
                for (int c = 0; c < COLS; ++c) {
                    A[b][c].x = fma(w.x, sv[c].x, A[b][c].x); A[b][c].y = fma(w.x, sv[c].y, A[b][c].y);
                    B[b][c].x = fma(w.y, dv[c].x, B[b][c].x); B[b][c].y = fma(w.y, dv[c].y, B[b][c].y);
                }
                e[b] += kk[b]; if (e[b] >= N2) e[b] -= N2;
            }
        }
        if (g2 == 0) {
#pragma unroll
            for (int c = 0; c < COLS; ++c) put(o0, c, S0[c].x, S0[c].y);
        }
#pragma unroll
        for (int b = 0; b < KB; ++b)
            if (kk[b]) {
#pragma unroll
                for (int c = 0; c < COLS; ++c) {
                    const double bx = sgn * B[b][c].y, by = -sgn * B[b][c].x;
                    put(oa[b], c, A[b][c].x + bx, A[b][c].y + by);
                }
#pragma unroll
                for (int c = 0; c < COLS; ++c) {
                    const double bx = sgn * B[b][c].y, by = -sgn * B[b][c].x;
                    put(ob[b], c, A[b][c].x - bx, A[b][c].y - by);
                }
            }
    }
}

// EXPAND (forward only): the input block is not read but formed on the fly -- the expansion sum_k alpha_k E_k of the Sylvester-SMW
// correction (k_region_expand + the boundary pieces): Y[z, x] = alpha[region(z), region(x)] Ksc[z, x], minus pb[z] in the first and
// pb[nz + z] in the last grid column; `src` is Ksc
// EXPAND = 2 (set-up of the SMW matrix): a BATCH of unit expansions E_kappa, kappa = kap0 + blockIdx.y = rz + N rx of an interior
// region (2 <= rx < N + 2): K_scaled on the L x L block of region (rz, rx), zero elsewhere, no boundary pieces.  Only the L columns
// of region rx are transformed (grid.x = ceil(L / COLS)); batch item b writes to dst + b tstride (cleared by the caller).
struct DftExpand { const cplx* alpha; const cplx* pb; int N, L; int kap0; int64_t tstride; };
template <bool FWD, int COLS, int KB, int EXPAND = 0>
__global__ __launch_bounds__(512) void k_dft_cols_sym(int nz, int nx, int N1, int N2, const int32_t* __restrict__ in_idx,
                                                       const int32_t* __restrict__ in_inv, const int32_t* __restrict__ out_idx,
                                                       const cplx* __restrict__ w1, const cplx* __restrict__ w2, double sgn,
                                                       double scale, const cplx* __restrict__ src, cplx* __restrict__ dst,
                                                       int xcd_order, TLay tl, DftExpand ex = DftExpand{nullptr, nullptr, 0, 0, 0, 0}) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    cplx* xs = (cplx*)smem_raw;                 // [c][q], q = n1 N2 + n2, then q = k1 N2 + n2
    cplx* c1 = xs + (size_t)COLS * nz;          // (cos, sin)(2 pi e / N1)
    cplx* c2 = c1 + N1;
    WP(tp0);
    int x0, nc, rzb = 0;
    if (EXPAND == 2) {
        const int kap = ex.kap0 + (int)blockIdx.y;
        const int rxb = kap / ex.N;
        rzb = kap - rxb * ex.N;
        const int xs0 = 2 + (rxb - 2) * ex.L;
        x0 = xs0 + (int)blockIdx.x * COLS;
        nc = min(COLS, xs0 + ex.L - x0);
        dst += (int64_t)blockIdx.y * ex.tstride;
    } else {
        x0 = dft_group(xcd_order) * COLS;
        nc = min(COLS, nx - x0);
    }
    const int nt = blockDim.x;
    for (int t = threadIdx.x; t < N1; t += nt) c1[t] = cmake(w1[t].x, -w1[t].y);
    for (int t = threadIdx.x; t < N2; t += nt) c2[t] = cmake(w2[t].x, -w2[t].y);
    if (FWD) {
        for (int t = threadIdx.x; t < COLS * nz; t += 4 * nt) {
            int slot[4]; cplx v[4]; bool on[4], live[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int tt = t + u * nt;
                live[u] = tt < COLS * nz;
                const int tc = live[u] ? tt : t;
                const int c = tc / nz, z = tc - c * nz;
                on[u] = c < nc;
                slot[u] = c * nz + in_inv[z];
                const int x = x0 + (on[u] ? c : 0);
                v[u] = src[(int64_t)x * nz + z];
                if (EXPAND == 2) {
                    if (z / ex.L != rzb) v[u] = cmake(0.0, 0.0);
                }
                if (EXPAND == 1) {
                    const int rz = z / ex.L;
                    const int rx = x < 2 ? x : (x >= nx - 2 ? ex.N + 2 + (x - (nx - 2)) : 2 + (x - 2) / ex.L);
                    v[u] = cmul(ex.alpha[(int64_t)rx * ex.N + rz], v[u]);
                    if (x == 0) v[u] = csub(v[u], ex.pb[z]);
                    if (x == nx - 1) v[u] = csub(v[u], ex.pb[nz + z]);
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) if (live[u]) xs[slot[u]] = on[u] ? v[u] : cmake(0.0, 0.0);
        }
    } else {
        for (int t = threadIdx.x; t < COLS * nz; t += 4 * nt) {
            int slot[4]; cplx v[4]; bool on[4], live[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int tt = t + u * nt;
                live[u] = tt < COLS * nz;
                const int tc = live[u] ? tt : t;
                const int q = tc / COLS, c = tc - q * COLS;
                on[u] = c < nc;
                slot[u] = c * nz + q;
                v[u] = src[(int64_t)in_idx[q] * tl.ldt + tpos(x0 + (on[u] ? c : 0), tl)];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) if (live[u]) xs[slot[u]] = on[u] ? v[u] : cmake(0.0, 0.0);
        }
    }
    __syncthreads();
    WP(tp1);
    dft_sym_stages<COLS, KB>(xs, c1, c2, nz, N1, N2, sgn, out_idx, [&](int o, int c, double re, double im) {
        if (c < nc) {
            if (FWD) dst[(int64_t)o * tl.ldt + tpos(x0 + c, tl)] = cmake(scale * re, scale * im);
            else dst[(int64_t)(x0 + c) * nz + o] = cmake(scale * re, scale * im);
        }
    });
#ifdef WEP_PROF
    __syncthreads();
    if (threadIdx.x == 0) {
        const long long tp3 = clock64();
        const int o = FWD ? 0 : 8;
        atomicAdd(&g_dft_prof[o + 0], (unsigned long long)(tp1 - tp0)); atomicAdd(&g_dft_prof[o + 1], (unsigned long long)(tp3 - tp1));
        atomicAdd(&g_dft_prof[o + 3], 1ull);
        atomicMax(&g_dft_prof[o + 4], (unsigned long long)(tp3 - tp0));
    }
#endif
}

// (Round 4: the same two stages were also written for the FP64 matrix cores -- complex products as four v_mfma_f64_16x16x4_f64 on
// the halves of one 16-byte LDS read, results of stage 1 kept in registers across a barrier so that ONE 64 KB buffer serves
// four columns -- correct against the NumPy reference, and exactly as fast: 28 / 30 us per transform against 28 / 30.  A
// -DWEP_PROF build (scripts/diag/wep_prof_run.py) shows why: a workgroup spends 8-10 k cycles loading, 28-29 k in stage 1 and
// 21 k in stage 2 + store with either kernel.  The two dense DFTs are 2.0 MFLOP per workgroup = 16 k cycles at the CU's
// 128 flop/clk, vector or matrix pipe alike on gfx950 (the FP64 MFMA issues one 16x16x4 per 64 cycles and SIMD); the matrix form
// pays 1.4x padding (27 -> 32, 37 -> 48) and its dependent accumulator chains, the vector form its issue rate at 1.5 wavefronts
// per SIMD.  The kernel is bound by the arithmetic of the prime-factor scheme at ~35 % of the FP64 peak, not by its transposing
// stores; what would lower it is fewer flops (27 = 3^3 by radix-3 steps: stage 1 from 27 to ~9 multiply-adds per entry), not
// another pipe.  The matrix-core version was removed again.)

// ---- Thomas pivots of (d_i I + B), one thread per mode (one-off per shift) -------------------------------------------------------
__global__ void k_tridiag_factor(int nz, int nx, const cplx* __restrict__ d, double b, cplx* __restrict__ mfac,
                                 cplx* __restrict__ dinv, TLay tl) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nz) return;
    const cplx a = cmake(d[i].x - 2.0 * b, d[i].y);
    cplx piv = a;
    auto inv = [](cplx z) { const double s = 1.0 / (z.x * z.x + z.y * z.y); return cmake(z.x * s, -z.y * s); };
    cplx pinv = inv(piv);
    mfac[(int64_t)i * tl.ldt] = cmake(0.0, 0.0);
    dinv[(int64_t)i * tl.ldt] = pinv;
    for (int j = 1; j < nx; ++j) {
        const cplx m = cmake(b * pinv.x, b * pinv.y);            // m_j = b / dtilde_{j-1}
        piv = cmake(a.x - b * m.x, a.y - b * m.y);               // dtilde_j = a - b m_j
        pinv = inv(piv);
        mfac[(int64_t)i * tl.ldt + tpos(j, tl)] = m;
        dinv[(int64_t)i * tl.ldt + tpos(j, tl)] = pinv;
    }
}

// affine map v -> A v + B ; composition "first f then g": (g.A f.A, g.A f.B + g.B)
struct Aff { cplx A, B; };
__device__ __forceinline__ Aff aff_then(const Aff f, const Aff g) {
    Aff r; r.A = cmul(g.A, f.A); r.B = cmul(g.A, f.B); r.B.x += g.B.x; r.B.y += g.B.y; return r;
}
__device__ __forceinline__ cplx shfl_c(cplx v, int src) { return cmake(__shfl(v.x, src, 64), __shfl(v.y, src, 64)); }

// ---- (d_i I + B) x = c for every mode i: one wave per mode, SEG consecutive x per lane -----------------------------------------
// forward  y_j = c_j - m_j y_{j-1}            (y_{-1} = 0)
// backward x_j = (y_j - b x_{j+1}) dinv_j     (x_{nx} = 0)
// MODE 0: in place on T.
// MODE 1: in place on T, and the x-region sums of the solution of every mode, S[rx nz + i] = wx(rx) sum_{x in region rx} x_i[x]
//         (regions as in k_region_means: columns 0, 1, nx-2, nx-1 alone with weight 1, N blocks of L columns with weight 1/L) -- the
//         region means of the Sylvester-SMW correction are taken in MODE space from these (nep_wep_smw_apply).
// MODE 2: right-hand sides from T2, result T <- T - solution (the second solve of nep_wep_smw_apply, subtracted in mode space).
template <int SEG, int MODE>
__global__ __launch_bounds__(256) void k_tridiag_modes(int nz, int nx, const cplx* __restrict__ mfac, const cplx* __restrict__ dinv,
                                                       double b, cplx* __restrict__ T, const cplx* __restrict__ T2 = nullptr,
                                                       cplx* __restrict__ S = nullptr, int N = 0, int L = 0, int64_t tstride = 0,
                                                       int64_t sstride = 0) {
    extern __shared__ __attribute__((aligned(16))) char tri_smem[];
    T += (int64_t)blockIdx.y * tstride;              // batch of right-hand side blocks (MODE 1, set-up of the SMW matrix)
    if (MODE == 1) S += (int64_t)blockIdx.y * sstride;
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const int iraw = blockIdx.x * 4 + wv;
    if (MODE != 1 && iraw >= nz) return;
    const bool valid = iraw < nz;                    // MODE 1 meets at a barrier: a wave past the last mode works on a copy of it
    const int i = valid ? iraw : nz - 1;
    const int64_t base = (int64_t)i * (SEG >= 4 ? 64 * SEG : nx);
    const cplx* __restrict__ Tin = MODE == 2 ? T2 : T;
    const int j0 = lane * SEG;
    cplx c[SEG], al[SEG];           // c: right-hand side, then y, then h;  al: forward multipliers, then g
    // (unconditional loads with a clamped index, masked afterwards: behind a per-entry `if (j < nx)` the 2 x SEG loads of a lane sat in
    // SEG exec-masked branches and were waited for one pair at a time)
    cplx dv[SEG];                   // dinv of the segment, fetched with the other two arrays (used by the backward sweep)
#pragma unroll
    for (int t = 0; t < SEG; ++t) {
        // SEG >= 4: entry lane * SEG + t sits at (t / 4) * 256 + lane * 4 + t % 4 of a row of 64 SEG entries (TLay); entries beyond nx are
        // allocated padding, read and then masked
        const int j = SEG >= 4 ? ((t >> 2) << 8) + (lane << 2) + (t & 3) : (j0 + t < nx ? j0 + t : nx - 1);
        c[t] = Tin[base + j]; al[t] = mfac[base + j]; dv[t] = dinv[base + j];
    }
#pragma unroll
    for (int t = 0; t < SEG; ++t) {
        if (j0 + t < nx) al[t] = cmake(-al[t].x, -al[t].y);
        else { c[t] = cmake(0.0, 0.0); al[t] = cmake(0.0, 0.0); }     // padding maps everything to 0 (never used downstream)
    }
    // ---- forward: segment composite, inclusive scan over lanes, apply
    Aff seg; seg.A = cmake(1.0, 0.0); seg.B = cmake(0.0, 0.0);
#pragma unroll
    for (int t = 0; t < SEG; ++t) { Aff e; e.A = al[t]; e.B = c[t]; seg = aff_then(seg, e); }
    Aff inc = seg;                                   // Hillis-Steele: inc(l) = seg(0) then ... then seg(l)
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int srcl = lane >= off ? lane - off : 0;
        Aff o; o.A = shfl_c(inc.A, srcl); o.B = shfl_c(inc.B, srcl);
        if (lane >= off) inc = aff_then(o, inc);
    }
    cplx carry = shfl_c(inc.B, lane > 0 ? lane - 1 : 0);   // y at the end of the previous lane's segment (start value 0: only B counts)
    if (lane == 0) carry = cmake(0.0, 0.0);
#pragma unroll
    for (int t = 0; t < SEG; ++t) { cplx v = c[t]; cfma(v, al[t], carry); c[t] = v; carry = v; }      // c = y
    // ---- backward: x_j = g_j x_{j+1} + h_j,  g_j = -b dinv_j, h_j = y_j dinv_j ; scan from the high end
#pragma unroll
    for (int t = 0; t < SEG; ++t) {
        const int j = j0 + t;
        if (j < nx) { const cplx di = dv[t]; al[t] = cmake(-b * di.x, -b * di.y); c[t] = cmul(c[t], di); }
        else { al[t] = cmake(0.0, 0.0); c[t] = cmake(0.0, 0.0); }      // beyond the end: x = 0
    }
    Aff segb; segb.A = cmake(1.0, 0.0); segb.B = cmake(0.0, 0.0);      // maps x_{j0+SEG} to x_{j0}: entry SEG-1 acts first
#pragma unroll
    for (int t = SEG - 1; t >= 0; --t) { Aff e; e.A = al[t]; e.B = c[t]; segb = aff_then(segb, e); }
    Aff incb = segb;                                  // inclusive scan from lane 63 downwards
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int srcl = lane + off < 64 ? lane + off : 63;
        Aff o; o.A = shfl_c(incb.A, srcl); o.B = shfl_c(incb.B, srcl);
        if (lane + off < 64) incb = aff_then(o, incb);
    }
    cplx carryb = shfl_c(incb.B, lane < 63 ? lane + 1 : 63);          // x at the start of the next lane's segment
    if (lane == 63) carryb = cmake(0.0, 0.0);
    cplx* rowbuf = (cplx*)tri_smem + (size_t)wv * nx;          // MODE 1: the solution of this wave's mode in plain x order
    cplx prev[SEG];
    if (MODE == 2) {
#pragma unroll
        for (int t = 0; t < SEG; ++t)
            prev[t] = T[base + (SEG >= 4 ? ((t >> 2) << 8) + (lane << 2) + (t & 3) : (j0 + t < nx ? j0 + t : nx - 1))];
    }
#pragma unroll
    for (int t = SEG - 1; t >= 0; --t) {
        cplx v = c[t]; cfma(v, al[t], carryb); carryb = v;
        const int j = j0 + t;
        if (MODE == 1) { if (j < nx) rowbuf[j] = v; }
        if (MODE == 2) v = csub(prev[t], v);
        if (j < nx && (MODE != 1 || valid)) T[base + (SEG >= 4 ? ((t >> 2) << 8) + (lane << 2) + (t & 3) : j)] = v;
    }
    if (MODE == 1) {
        __syncthreads();
        for (int r = lane; r < N + 4; r += 64) {
            int x0, len; double w;
            if (r < 2) { x0 = r; len = 1; w = 1.0; }
            else if (r >= N + 2) { x0 = nx - 2 + (r - (N + 2)); len = 1; w = 1.0; }
            else { x0 = 2 + (r - 2) * L; len = L; w = 1.0 / L; }
            cplx acc = cmake(0.0, 0.0);
            for (int x = x0; x < x0 + len; ++x) { acc.x += rowbuf[x].x; acc.y += rowbuf[x].y; }
            if (valid) S[(int64_t)r * nz + i] = cmake(w * acc.x, w * acc.y);
        }
    }
}

// f[rx N + rz] = sum_i G[rz nz + i] S[rx nz + i]: the region means of F U (U: mode-space solution, F: the inverse transform) from the
// x-region sums S of the modes; G[rz, i] = mean over the z of region rz of F[z, i] (host, once per grid).  One wave per entry.
__global__ __launch_bounds__(256) void k_wep_mode_means(int nz, int N, const cplx* __restrict__ G, const cplx* __restrict__ S,
                                                        cplx* __restrict__ f, int64_t sstride = 0, int64_t fstride = 0) {
    S += (int64_t)blockIdx.y * sstride; f += (int64_t)blockIdx.y * fstride;      // batch (set-up of the SMW matrix)
    const int lane = threadIdx.x & 63;
    const int o = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (o >= N * (N + 4)) return;
    const int rx = o / N, rz = o - rx * N;
    const cplx* g = G + (int64_t)rz * nz;
    const cplx* sc = S + (int64_t)rx * nz;
    cplx acc = cmake(0.0, 0.0), acc2 = cmake(0.0, 0.0);        // eight trips in flight, two accumulators, fixed order
    int i = lane;
    for (; i + 7 * 64 < nz; i += 8 * 64) {
        cplx gv[8], sv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { gv[u] = g[i + u * 64]; sv[u] = sc[i + u * 64]; }
#pragma unroll
        for (int u = 0; u < 8; u += 2) { cfma(acc, gv[u], sv[u]); cfma(acc2, gv[u + 1], sv[u + 1]); }
    }
    for (; i < nz; i += 64) cfma(acc, g[i], sc[i]);
    acc = cadd(acc, acc2);
    acc = group_reduce_sum<64>(acc);
    if (lane == 0) f[o] = acc;
}

// the boundary pieces of the expansion (k_region_expand's eb) alone: eb[:, 0] = dd1 a[rz, 0] + dd2 a[rz, 1],
// eb[:, 1] = dd2 a[rz, N+2] + dd1 a[rz, N+3]
__global__ void k_region_eb(int nz, int N, int L, const cplx* __restrict__ alpha, double dd1, double dd2, cplx* __restrict__ eb) {
    const int z = blockIdx.x * blockDim.x + threadIdx.x;
    if (z >= nz) return;
    const int rz = z / L;
    const cplx a0 = alpha[rz], a1 = alpha[(int64_t)N + rz], a2 = alpha[(int64_t)(N + 2) * N + rz], a3 = alpha[(int64_t)(N + 3) * N + rz];
    eb[z] = cmake(dd1 * a0.x + dd2 * a1.x, dd1 * a0.y + dd2 * a1.y);
    eb[nz + z] = cmake(dd2 * a2.x + dd1 * a3.x, dd2 * a2.y + dd1 * a3.y);
}

// ---- region means / expansion of the SMW correction (waveguide_preconditioner.jl:263-304) -------------------------------------
// out (N x (N+4), column-major): mean over the region (zi, xk) of X (nz x nx, z fastest).  x regions: columns 0, 1, nx-2, nx-1 are
// single-column regions 0, 1, N+2, N+3; region 2+j covers columns 2 + jL .. 2 + (j+1)L - 1.  z regions: L consecutive rows.
__global__ __launch_bounds__(256) void k_region_means(int nz, int nx, int N, int L, const cplx* __restrict__ X, cplx* __restrict__ out) {
    __shared__ cplx sm[4];
    const int zi = blockIdx.x, xk = blockIdx.y;
    int c0, c1; double wx;
    if (xk < 2) { c0 = xk; c1 = xk + 1; wx = 1.0; }
    else if (xk >= N + 2) { c0 = nx - 2 + (xk - (N + 2)); c1 = c0 + 1; wx = 1.0; }
    else { c0 = 2 + (xk - 2) * L; c1 = c0 + L; wx = 1.0 / L; }
    const int nrow = L, ncol = c1 - c0;
    cplx acc = cmake(0.0, 0.0);
    for (int t = threadIdx.x; t < nrow * ncol; t += 256) {
        const int cc = t / nrow, rr = t - cc * nrow;
        const cplx v = X[(int64_t)(c0 + cc) * nz + zi * L + rr];
        acc.x += v.x; acc.y += v.y;
    }
    acc = group_reduce_sum<64>(acc);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        cplx s = sm[0];
        for (int q = 1; q < 4; ++q) { s.x += sm[q].x; s.y += sm[q].y; }
        const double w = wx / L;
        out[(int64_t)xk * N + zi] = cmake(w * s.x, w * s.y);
    }
}
// Y[z, x] = alpha[region_z(z), region_x(x)] * Ksc[z, x];  eb (nz x 2) = the boundary pieces
//   eb[:, 0] = dd1 * alpha[rz, 0] + dd2 * alpha[rz, 1],  eb[:, 1] = dd2 * alpha[rz, N+2] + dd1 * alpha[rz, N+3]
__global__ void k_region_expand(int nz, int nx, int N, int L, const cplx* __restrict__ alpha, const cplx* __restrict__ Ksc,
                                double dd1, double dd2, cplx* __restrict__ Y, cplx* __restrict__ eb) {
    const int64_t total = (int64_t)nz * nx;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(t / nz), z = (int)(t - (int64_t)x * nz);
        const int rz = z / L;
        const int rx = x < 2 ? x : (x >= nx - 2 ? N + 2 + (x - (nx - 2)) : 2 + (x - 2) / L);
        Y[t] = cmul(alpha[(int64_t)rx * N + rz], Ksc[t]);
        if (x == 0) {
            const cplx a0 = alpha[rz], a1 = alpha[(int64_t)N + rz], a2 = alpha[(int64_t)(N + 2) * N + rz], a3 = alpha[(int64_t)(N + 3) * N + rz];
            eb[z] = cmake(dd1 * a0.x + dd2 * a1.x, dd1 * a0.y + dd2 * a1.y);
            eb[nz + z] = cmake(dd2 * a2.x + dd1 * a3.x, dd2 * a2.y + dd1 * a3.y);
        }
    }
}

// ---- boundary operator P(lam)^{-1} = R diag(1 / s(lam)) R^H / nz,  R x = reverse(bb .* fft(x))  (Waveguide.jl:53-65,159-162) ------
// one workgroup per half (minus / plus block): x -> reverse -> conj(bb) .* -> inverse-direction DFT -> scale by sinv = 1/(nz s) ->
// DFT -> bb .* -> reverse, both DFTs by the same prime-factor scheme out of LDS.  Replaces four nz x nz dense GEMVs per application.
// entry i of the half's input: x[i], or -- gathered from the interior block X (nz x gnx, z fastest) -- the boundary functional C2T of
// generate_fd_boundary_mat: gd1 X[i, 0] + gd2 X[i, 1] (minus half), gd1 X[i, nx-1] + gd2 X[i, nx-2] (plus half)
// ... or, gnx < 0 (nep_wep_smw_apply): the boundary pieces of the expansion formed on the fly from the region coefficients alpha = gX
// (N x (N+4), N = -gnx):  minus half gd1 a[rz, 0] + gd2 a[rz, 1],  plus half gd2 a[rz, N+2] + gd1 a[rz, N+3]  (k_region_eb)
__device__ __forceinline__ cplx pinv_in(const cplx* __restrict__ xh, const cplx* __restrict__ gX, int gnx, double gd1, double gd2,
                                        int half, int nz, int i) {
    if (!gX) return xh[i];
    if (gnx < 0) {
        const int N = -gnx, rz = i / (nz / N);
        const cplx a = gX[(int64_t)(half ? N + 2 : 0) * N + rz], b = gX[(int64_t)(half ? N + 3 : 1) * N + rz];
        return half ? cmake(fma(gd2, a.x, gd1 * b.x), fma(gd2, a.y, gd1 * b.y)) : cmake(fma(gd1, a.x, gd2 * b.x), fma(gd1, a.y, gd2 * b.y));
    }
    const cplx u = gX[(int64_t)(half ? gnx - 1 : 0) * nz + i], v = gX[(int64_t)(half ? gnx - 2 : 1) * nz + i];
    return cmake(fma(gd1, u.x, gd2 * v.x), fma(gd1, u.y, gd2 * v.y));
}
__global__ __launch_bounds__(1024) void k_wep_pinv(int nz, int N1, int N2, const int32_t* __restrict__ in_idx,
                                                   const int32_t* __restrict__ out_idx, const cplx* __restrict__ w1,
                                                   const cplx* __restrict__ w2, const cplx* __restrict__ bb,
                                                   const cplx* __restrict__ sinv, const cplx* __restrict__ x, cplx* __restrict__ out,
                                                      const cplx* __restrict__ gX = nullptr, int gnx = 0, double gd1 = 0.0, double gd2 = 0.0) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    cplx* a = (cplx*)smem_raw;        // nz
    cplx* t = a + nz;                 // nz
    cplx* v = t + nz;                 // nz (natural order)
    cplx* r1 = v + nz; cplx* r2 = r1 + N1;
    const int half = blockIdx.x;
    const cplx* xh = x + (int64_t)half * nz;
    const cplx* sh = sinv + (int64_t)half * nz;
    cplx* oh = out + (int64_t)half * nz;
    const int nt = blockDim.x;
    for (int pass = 0; pass < 2; ++pass) {
        const double sgn = pass == 0 ? -1.0 : 1.0;           // pass 0: F^H (exponent +), pass 1: F
        __syncthreads();
        for (int q = threadIdx.x; q < N1; q += nt) r1[q] = cmake(w1[q].x, sgn * w1[q].y);
        for (int q = threadIdx.x; q < N2; q += nt) r2[q] = cmake(w2[q].x, sgn * w2[q].y);
        for (int q = threadIdx.x; q < nz; q += nt) {
            const int m = in_idx[q];
            if (pass == 0) { const cplx b = bb[m]; a[q] = cmul(cmake(b.x, -b.y), pinv_in(xh, gX, gnx, gd1, gd2, half, nz, nz - 1 - m)); }
            else a[q] = v[m];
        }
        __syncthreads();
        for (int q = threadIdx.x; q < nz; q += nt) {
            const int k1 = q / N2, n2 = q - k1 * N2;
            cplx acc = cmake(0.0, 0.0);
            int e = 0;
            for (int n1 = 0; n1 < N1; ++n1) { cfma(acc, a[n1 * N2 + n2], r1[e]); e += k1; if (e >= N1) e -= N1; }
            t[q] = acc;
        }
        __syncthreads();
        for (int q = threadIdx.x; q < nz; q += nt) {
            const int k1 = q / N2, k2 = q - k1 * N2;
            cplx acc = cmake(0.0, 0.0);
            int e = 0;
            for (int n2 = 0; n2 < N2; ++n2) { cfma(acc, t[k1 * N2 + n2], r2[e]); e += k2; if (e >= N2) e -= N2; }
            const int k = out_idx[q];
            if (pass == 0) v[k] = cmul(acc, sh[k]);
            else oh[nz - 1 - k] = cmul(bb[k], acc);
        }
    }
}

// k_wep_pinv with the symmetric-half stages (N1, N2 odd): the plain form above is bound by the LDS bandwidth of its one CU per half
// (two 16-byte reads per complex multiply-add, 13.6 us of reads at 999 points); this one reads 12 bytes per product of the plain form
__global__ __launch_bounds__(512) void k_wep_pinv_sym(int nz, int N1, int N2, const int32_t* __restrict__ in_idx,
                                                      const int32_t* __restrict__ out_idx, const cplx* __restrict__ w1,
                                                      const cplx* __restrict__ w2, const cplx* __restrict__ bb,
                                                      const cplx* __restrict__ sinv, const cplx* __restrict__ x, cplx* __restrict__ out,
                                                      const cplx* __restrict__ gX = nullptr, int gnx = 0, double gd1 = 0.0, double gd2 = 0.0) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    cplx* a = (cplx*)smem_raw;        // nz: stage buffer
    cplx* v = a + nz;                 // nz: result of pass 0, natural order
    cplx* c1 = v + nz; cplx* c2 = c1 + N1;
    const int half = blockIdx.x;
    const cplx* xh = x + (int64_t)half * nz;
    const cplx* sh = sinv + (int64_t)half * nz;
    cplx* oh = out + (int64_t)half * nz;
    const int nt = blockDim.x;
    for (int q = threadIdx.x; q < N1; q += nt) c1[q] = cmake(w1[q].x, -w1[q].y);
    for (int q = threadIdx.x; q < N2; q += nt) c2[q] = cmake(w2[q].x, -w2[q].y);
    for (int q = threadIdx.x; q < nz; q += nt) {
        const int m = in_idx[q];
        const cplx b = bb[m];
        a[q] = cmul(cmake(b.x, -b.y), pinv_in(xh, gX, gnx, gd1, gd2, half, nz, nz - 1 - m));
    }
    __syncthreads();
    dft_sym_stages<1, 1>(a, c1, c2, nz, N1, N2, -1.0, out_idx, [&](int k, int, double re, double im) { v[k] = cmul(cmake(re, im), sh[k]); });
    __syncthreads();
    for (int q = threadIdx.x; q < nz; q += nt) a[q] = v[in_idx[q]];
    __syncthreads();
    dft_sym_stages<1, 1>(a, c1, c2, nz, N1, N2, 1.0, out_idx, [&](int k, int, double re, double im) { oh[nz - 1 - k] = cmul(bb[k], cmake(re, im)); });
}

static int egcd_inv(int a, int m) {            // a^{-1} mod m (gcd = 1)
    int t = 0, nt = 1, r = m, nr = a % m;
    while (nr) { const int q = r / nr; int tmp = t - q * nt; t = nt; nt = tmp; tmp = r - q * nr; r = nr; nr = tmp; }
    return t < 0 ? t + m : t;
}

extern "C" {

int32_t nep_wep_sylv_destroy(nep_wep_sylv* s) {
    if (!s) return NEP_OK;
    nep_pool_free(s->d_in); nep_pool_free(s->d_out); nep_pool_free(s->d_in_inv); nep_pool_free(s->d_w1); nep_pool_free(s->d_w2);
    nep_pool_free(s->d_m); nep_pool_free(s->d_dinv); nep_pool_free(s->d_T); nep_pool_free(s->d_T2);
    delete s;
    return NEP_OK;
}

int32_t nep_wep_sylv_create(int32_t nz, int32_t nx, const nep_cdouble* h_d, double b, nep_wep_sylv** out) {
    ARGCHK(out != nullptr);
    *out = nullptr;
    ARGCHK(nz >= 1 && nx >= 2 && h_d != nullptr && nx <= 64 * 32);
    // coprime factorisation with the smallest N1 + N2 (N2 = 1: plain dense DFT of length nz)
    int N1 = nz, N2 = 1;
    for (int a = 2; a * a <= nz; ++a)
        if (nz % a == 0 && std::gcd(a, nz / a) == 1 && a + nz / a < N1 + N2) { N1 = nz / a; N2 = a; }
    int cols = getenv("NEP_WEP_DFT_COLS") ? std::max(1, std::min(4, atoi(getenv("NEP_WEP_DFT_COLS")))) : 4;
    if (cols == 3) cols = 2;
    while (cols > 1 && ((size_t)2 * cols * nz + N1 + N2) * sizeof(cplx) > 150 * 1024) cols >>= 1;
    if (((size_t)2 * cols * nz + N1 + N2) * sizeof(cplx) > 150 * 1024) {
        nep_set_error("nep_wep_sylv_create: nz = %d does not fit the LDS staging of the DFT kernel", nz);
        return NEP_ERR_UNSUPPORTED;
    }
    nep_wep_sylv* s = new nep_wep_sylv();
    s->nz = nz; s->nx = nx; s->N1 = N1; s->N2 = N2; s->cols = cols; s->b = b;
    std::vector<int32_t> in_idx(nz), out_idx(nz);
    const int i2 = N1 > 1 && N2 > 1 ? egcd_inv(N2 % N1, N1) : 0, i1 = N1 > 1 && N2 > 1 ? egcd_inv(N1 % N2, N2) : 0;
    for (int a = 0; a < N1; ++a)
        for (int c = 0; c < N2; ++c) {
            if (N2 == 1) { in_idx[a] = a; out_idx[a] = a; continue; }
            in_idx[a * N2 + c] = (int32_t)(((int64_t)N2 * a + (int64_t)N1 * c) % nz);
            out_idx[a * N2 + c] = (int32_t)(((int64_t)N2 * i2 % nz * a + (int64_t)N1 * i1 % nz * c) % nz);
        }
    std::vector<nep_cdouble> w1(N1), w2(N2);
    for (int j = 0; j < N1; ++j) { const double th = -2.0 * M_PI * j / N1; w1[j].re = cos(th); w1[j].im = sin(th); }
    for (int j = 0; j < N2; ++j) { const double th = -2.0 * M_PI * j / N2; w2[j].re = cos(th); w2[j].im = sin(th); }
    {   // SEG of k_tridiag_modes (nep_wep_sylv_solve): the smallest power of two with 64 SEG >= nx
        int seg = 1, lseg = 0;
        while (64 * seg < nx) { seg <<= 1; ++lseg; }
        s->lseg = lseg; s->ldt = seg >= 4 ? 64 * seg : nx;
    }
    const size_t ldt = (size_t)s->ldt;
    int rc = nep_pool_alloc((void**)&s->d_in, (size_t)nz * 4);
    if (!rc) rc = nep_pool_alloc((void**)&s->d_out, (size_t)nz * 4);
    if (!rc) rc = nep_pool_alloc((void**)&s->d_in_inv, (size_t)nz * 4);
    if (!rc) rc = nep_pool_alloc((void**)&s->d_w1, (size_t)N1 * 16);
    if (!rc) rc = nep_pool_alloc((void**)&s->d_w2, (size_t)N2 * 16);
    if (!rc) rc = nep_pool_alloc((void**)&s->d_m, (size_t)nz * ldt * 16);
    if (!rc) rc = nep_pool_alloc((void**)&s->d_dinv, (size_t)nz * ldt * 16);
    if (!rc) rc = nep_pool_alloc((void**)&s->d_T, (size_t)nz * ldt * 16);
    cplx* d_d = nullptr;
    if (!rc) rc = nep_pool_alloc((void**)&d_d, (size_t)nz * 16);
    if (rc) { nep_wep_sylv_destroy(s); return rc; }
    std::vector<int32_t> in_inv(nz);
    for (int q = 0; q < nz; ++q) in_inv[in_idx[q]] = q;
    hipError_t e = hipMemcpy(s->d_in, in_idx.data(), (size_t)nz * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(s->d_out, out_idx.data(), (size_t)nz * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(s->d_in_inv, in_inv.data(), (size_t)nz * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(s->d_w1, w1.data(), (size_t)N1 * 16, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(s->d_w2, w2.data(), (size_t)N2 * 16, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_d, h_d, (size_t)nz * 16, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_tridiag_factor, dim3((nz + 63) / 64), dim3(64), 0, nullptr, (int)nz, (int)nx, (const cplx*)d_d, b, s->d_m, s->d_dinv, TLay{s->ldt, s->lseg});
        e = hipGetLastError();
        if (e == hipSuccess) e = hipDeviceSynchronize();
    }
    nep_pool_free(d_d);
    if (e != hipSuccess) { nep_set_error("nep_wep_sylv_create: %s", hipGetErrorString(e)); nep_wep_sylv_destroy(s); return NEP_ERR_HIP; }
    *out = s;
    return NEP_OK;
}

// ---- P(lam)^{-1}: plan (depends on nz and bb only) ----------------------------------------------------------------------------
struct nep_wep_pinv { int nz = 0, N1 = 0, N2 = 0; int32_t *d_in = nullptr, *d_out = nullptr; cplx *d_w1 = nullptr, *d_w2 = nullptr, *d_bb = nullptr; };

int32_t nep_wep_pinv_destroy(nep_wep_pinv* p) {
    if (!p) return NEP_OK;
    nep_pool_free(p->d_in); nep_pool_free(p->d_out); nep_pool_free(p->d_w1); nep_pool_free(p->d_w2); nep_pool_free(p->d_bb);
    delete p;
    return NEP_OK;
}

int32_t nep_wep_pinv_create(int32_t nz, const nep_cdouble* h_bb, nep_wep_pinv** out) {
    ARGCHK(out != nullptr);
    *out = nullptr;
    ARGCHK(nz >= 1 && h_bb != nullptr);
    if (((size_t)3 * nz + 2 * (size_t)nz) * sizeof(cplx) > 150 * 1024) { nep_set_error("nep_wep_pinv_create: nz = %d too large for the LDS staging", nz); return NEP_ERR_UNSUPPORTED; }
    int N1 = nz, N2 = 1;
    for (int a = 2; a * a <= nz; ++a)
        if (nz % a == 0 && std::gcd(a, nz / a) == 1 && a + nz / a < N1 + N2) { N1 = nz / a; N2 = a; }
    nep_wep_pinv* p = new nep_wep_pinv();
    p->nz = nz; p->N1 = N1; p->N2 = N2;
    std::vector<int32_t> in_idx(nz), out_idx(nz);
    const int i2 = N2 > 1 ? egcd_inv(N2 % N1, N1) : 0, i1 = N2 > 1 ? egcd_inv(N1 % N2, N2) : 0;
    for (int a = 0; a < N1; ++a)
        for (int c = 0; c < N2; ++c) {
            if (N2 == 1) { in_idx[a] = a; out_idx[a] = a; continue; }
            in_idx[a * N2 + c] = (int32_t)(((int64_t)N2 * a + (int64_t)N1 * c) % nz);
            out_idx[a * N2 + c] = (int32_t)(((int64_t)N2 * i2 % nz * a + (int64_t)N1 * i1 % nz * c) % nz);
        }
    std::vector<nep_cdouble> w1(N1), w2(N2);
    for (int j = 0; j < N1; ++j) { const double th = -2.0 * M_PI * j / N1; w1[j].re = cos(th); w1[j].im = sin(th); }
    for (int j = 0; j < N2; ++j) { const double th = -2.0 * M_PI * j / N2; w2[j].re = cos(th); w2[j].im = sin(th); }
    int rc = nep_pool_alloc((void**)&p->d_in, (size_t)nz * 4);
    if (!rc) rc = nep_pool_alloc((void**)&p->d_out, (size_t)nz * 4);
    if (!rc) rc = nep_pool_alloc((void**)&p->d_w1, (size_t)N1 * 16);
    if (!rc) rc = nep_pool_alloc((void**)&p->d_w2, (size_t)N2 * 16);
    if (!rc) rc = nep_pool_alloc((void**)&p->d_bb, (size_t)nz * 16);
    if (rc) { nep_wep_pinv_destroy(p); return rc; }
    hipError_t e = hipMemcpy(p->d_in, in_idx.data(), (size_t)nz * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(p->d_out, out_idx.data(), (size_t)nz * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(p->d_w1, w1.data(), (size_t)N1 * 16, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(p->d_w2, w2.data(), (size_t)N2 * 16, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(p->d_bb, h_bb, (size_t)nz * 16, hipMemcpyHostToDevice);
    if (e != hipSuccess) { nep_set_error("nep_wep_pinv_create: %s", hipGetErrorString(e)); nep_wep_pinv_destroy(p); return NEP_ERR_HIP; }
    *out = p;
    return NEP_OK;
}

// dOut (2 nz) = blkdiag(R, R) diag(d_sinv) blkdiag(R, R)^H dX, d_sinv = 1 / (nz s_j(lam)) (2 nz device entries); dOut may alias dX
static int32_t pinv_apply_impl(nep_wep_pinv* p, const nep_cdouble* d_sinv, const nep_cdouble* dX, nep_cdouble* dOut, nep_stream stream,
                               const cplx* gX, int gnx, double gd1, double gd2);
int32_t nep_wep_pinv_apply(nep_wep_pinv* p, const nep_cdouble* d_sinv, const nep_cdouble* dX, nep_cdouble* dOut, nep_stream stream) {
    ARGCHK(p && d_sinv && dX && dOut);
    return pinv_apply_impl(p, d_sinv, dX, dOut, stream, nullptr, 0, 0.0, 0.0);
}
static int32_t pinv_apply_impl(nep_wep_pinv* p, const nep_cdouble* d_sinv, const nep_cdouble* dX, nep_cdouble* dOut, nep_stream stream,
                               const cplx* gX, int gnx, double gd1, double gd2) {
    static thread_local bool attr_set = false;
    if (!attr_set) { HIPCHK(hipFuncSetAttribute((const void*)k_wep_pinv, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024)); attr_set = true; }
    static const bool sym_on = !(getenv("NEP_WEP_PINV_SYM") && atoi(getenv("NEP_WEP_PINV_SYM")) == 0);
    if (sym_on && (p->N1 & 1) && (p->N2 & 1) && p->N1 >= 3 && p->N2 >= 3) {
        const int items = std::max((p->N1 - 1) / 2 * p->N2, (p->N2 - 1) / 2 * p->N1);
        if (items <= 512) {
            hipLaunchKernelGGL(k_wep_pinv_sym, dim3(2), dim3((items + 63) / 64 * 64), ((size_t)2 * p->nz + p->N1 + p->N2) * sizeof(cplx),
                               as_stream(stream), p->nz, p->N1, p->N2, (const int32_t*)p->d_in, (const int32_t*)p->d_out, (const cplx*)p->d_w1,
                               (const cplx*)p->d_w2, (const cplx*)p->d_bb, (const cplx*)d_sinv, (const cplx*)dX, (cplx*)dOut, gX, gnx, gd1, gd2);
            LAUNCHCHK();
            return NEP_OK;
        }
    }
    const size_t shm = ((size_t)3 * p->nz + p->N1 + p->N2) * sizeof(cplx);
    const int threads = p->nz >= 768 ? 1024 : (p->nz >= 256 ? 512 : 256);
    hipLaunchKernelGGL(k_wep_pinv, dim3(2), dim3(threads), shm, as_stream(stream), p->nz, p->N1, p->N2, (const int32_t*)p->d_in,
                       (const int32_t*)p->d_out, (const cplx*)p->d_w1, (const cplx*)p->d_w2, (const cplx*)p->d_bb, (const cplx*)d_sinv,
                       (const cplx*)dX, (cplx*)dOut, gX, gnx, gd1, gd2);
    LAUNCHCHK();
    return NEP_OK;
}

// ---- Schur complement of the waveguide, matrix-free (SchurMatVec, Waveguide.jl:394-425) ------------------------------------------
//   out = vec(A(lam) X + X B + K .* X) - C1 P(lam)^{-1} C2T v,   X = reshape(v, nz, nx)
// with A(lam) = Dzz + 2 lam Dz + lam^2 I (periodic second / central first difference along z), B = Dxx (Dirichlet along x):
// a five-point stencil with constant off-diagonal weights cp (z+1), cm (z-1), cx (x-1, x+1) and the diagonal D0 = K + lam^2 -
// 2/hz^2 - 2/hx^2 as an array; C2T v is gathered inside the boundary kernel (pinv_in), C1 adds c1s * P^{-1}(..) to the first and
// the last column.  One read of X and D0, one write: 48 bytes per unknown where the assembled operator (three stacked sparse terms
// through K1, a copy in front and a CSR pass behind it) moved 148.
__global__ __launch_bounds__(256) void k_wep_schur_stencil(int nz, int nx, const cplx* __restrict__ X, const cplx* __restrict__ D0,
                                                           cplx cp, cplx cm, double cx, double c1s, const cplx* __restrict__ pb,
                                                           cplx* __restrict__ out) {
    const int x = blockIdx.y;
    const int z = blockIdx.x * 256 + threadIdx.x;
    if (z >= nz) return;
    const int64_t i = (int64_t)x * nz + z;
    const cplx c = X[i];
    const cplx up = X[(int64_t)x * nz + (z + 1 < nz ? z + 1 : 0)];
    const cplx dn = X[(int64_t)x * nz + (z > 0 ? z - 1 : nz - 1)];
    cplx r = cmul(D0[i], c);
    cfma(r, cp, up); cfma(r, cm, dn);
    if (x > 0) { const cplx l = X[i - nz]; r.x = fma(cx, l.x, r.x); r.y = fma(cx, l.y, r.y); }
    if (x + 1 < nx) { const cplx rr = X[i + nz]; r.x = fma(cx, rr.x, r.x); r.y = fma(cx, rr.y, r.y); }
    if (x == 0) { const cplx q = pb[z]; r.x = fma(-c1s, q.x, r.x); r.y = fma(-c1s, q.y, r.y); }
    if (x == nx - 1) { const cplx q = pb[nz + z]; r.x = fma(-c1s, q.x, r.x); r.y = fma(-c1s, q.y, r.y); }
    out[i] = r;
}

int32_t nep_wep_schur_matvec(nep_wep_pinv* p, const nep_cdouble* d_sinv, int32_t nx, const nep_cdouble* dV, const nep_cdouble* dD0,
                             nep_cdouble cp, nep_cdouble cm, double cx, double d1, double d2, double c1s, nep_cdouble* dP,
                             nep_cdouble* dOut, nep_stream stream) {
    ARGCHK(p && d_sinv && dV && dD0 && dP && dOut && nx >= 2 && dV != dOut);
    int rc = pinv_apply_impl(p, d_sinv, dV, dP, stream, (const cplx*)dV, nx, d1, d2);
    if (rc) return rc;
    cplx cpv, cmv; cpv.x = cp.re; cpv.y = cp.im; cmv.x = cm.re; cmv.y = cm.im;
    hipLaunchKernelGGL(k_wep_schur_stencil, dim3((p->nz + 255) / 256, nx), dim3(256), 0, as_stream(stream), p->nz, (int)nx,
                       (const cplx*)dV, (const cplx*)dD0, cpv, cmv, cx, c1s, (const cplx*)dP, (cplx*)dOut);
    LAUNCHCHK();
    return NEP_OK;
}

int32_t nep_wep_sylv_info(const nep_wep_sylv* s, int32_t out[4]) {
    ARGCHK(s && out);
    out[0] = s->N1; out[1] = s->N2; out[2] = s->cols; out[3] = (s->nx + 63) / 64;
    return NEP_OK;
}

// X (nz x nx, column-major = z fastest, device) <- solution of A(sigma) X + X B = X, in place
int32_t nep_wep_sylv_solve(nep_wep_sylv* s, nep_cdouble* dX, nep_stream stream) {
    ARGCHK(s && dX);
    hipStream_t st = as_stream(stream);
    const int nz = s->nz, nx = s->nx;
    const size_t shm = ((size_t)2 * s->cols * nz + s->N1 + s->N2) * sizeof(cplx);
    static thread_local bool attr_set = false;
    if (!attr_set) {
#define DFT_ATTR(F_, C_) HIPCHK(hipFuncSetAttribute((const void*)k_dft_cols<F_, C_>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024))
        DFT_ATTR(true, 4); DFT_ATTR(false, 4); DFT_ATTR(true, 2); DFT_ATTR(false, 2); DFT_ATTR(true, 1); DFT_ATTR(false, 1);
        HIPCHK(hipFuncSetAttribute((const void*)k_dft_cols_rb<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
        HIPCHK(hipFuncSetAttribute((const void*)k_dft_cols_rb<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
#undef DFT_ATTR
        attr_set = true;
    }
    const double scale = 1.0 / sqrt((double)nz);
    const TLay tl{s->ldt, s->lseg};
    const dim3 grid((unsigned)((nx + s->cols - 1) / s->cols));
    const int threads = nz >= 768 ? 1024 : (nz >= 384 ? 512 : 256);
#define DFT_LAUNCH(F_, C_, SGN_, SRC_, DST_)                                                                               \
    hipLaunchKernelGGL((k_dft_cols<F_, C_>), grid, dim3(threads), shm, st, nz, nx, s->N1, s->N2, (const int32_t*)s->d_in,    \
                       (const int32_t*)s->d_out, (const cplx*)s->d_w1, (const cplx*)s->d_w2, SGN_, scale, SRC_, DST_, xcd_order, tl)
    static const int xcd_order = getenv("NEP_WEP_DFT_XCD") ? atoi(getenv("NEP_WEP_DFT_XCD")) : 1;
    static const int rb = getenv("NEP_WEP_DFT_RB") ? atoi(getenv("NEP_WEP_DFT_RB")) : 1;
#define DFT_BY_COLS(F_, SGN_, SRC_, DST_)                                                                                  \
    do { if (s->cols == 4 && rb)                                                                                            \
             hipLaunchKernelGGL((k_dft_cols_rb<F_>), grid, dim3(384), shm, st, nz, nx, s->N1, s->N2, (const int32_t*)s->d_in, \
                                (const int32_t*)s->d_in_inv, (const int32_t*)s->d_out, (const cplx*)s->d_w1, (const cplx*)s->d_w2, SGN_, scale, SRC_, DST_, xcd_order, tl); \
         else if (s->cols == 4) DFT_LAUNCH(F_, 4, SGN_, SRC_, DST_); else if (s->cols == 2) DFT_LAUNCH(F_, 2, SGN_, SRC_, DST_);  \
         else DFT_LAUNCH(F_, 1, SGN_, SRC_, DST_); } while (0)
    // symmetric-half form of the two dense stages (odd N1, N2): NEP_WEP_DFT_SYM = "cols*10 + kb" (42, 43, 22, 23) or 0 = off
    static const int symcfg = getenv("NEP_WEP_DFT_SYM") ? atoi(getenv("NEP_WEP_DFT_SYM")) : 22;
    int sym_cols = symcfg / 10, sym_kb = symcfg % 10, sym_threads = 0;
    if (symcfg && (s->N1 & 1) && (s->N2 & 1) && s->N1 >= 3 && s->N2 >= 3 && (sym_cols == 2 || sym_cols == 4) && (sym_kb == 2 || sym_kb == 3)) {
        const int H1 = (s->N1 - 1) / 2, H2 = (s->N2 - 1) / 2;
        const int items = std::max(((H1 + sym_kb - 1) / sym_kb) * s->N2, ((H2 + sym_kb - 1) / sym_kb) * s->N1);
        sym_threads = (items + 63) / 64 * 64;
        if (sym_threads > 512 || ((size_t)sym_cols * nz + s->N1 + s->N2) * sizeof(cplx) > 150 * 1024) sym_threads = 0;
    }
    if (sym_threads) {
        static thread_local bool sym_attr = false;
        if (!sym_attr) {
#define SYM_ATTR(F_, C_, K_) HIPCHK(hipFuncSetAttribute((const void*)k_dft_cols_sym<F_, C_, K_>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024))
            SYM_ATTR(true, 4, 2); SYM_ATTR(false, 4, 2); SYM_ATTR(true, 4, 3); SYM_ATTR(false, 4, 3);
            SYM_ATTR(true, 2, 2); SYM_ATTR(false, 2, 2); SYM_ATTR(true, 2, 3); SYM_ATTR(false, 2, 3);
#undef SYM_ATTR
            sym_attr = true;
        }
    }
    const size_t sym_shm = ((size_t)sym_cols * nz + s->N1 + s->N2) * sizeof(cplx);
    const dim3 sym_grid((unsigned)((nx + std::max(sym_cols, 1) - 1) / std::max(sym_cols, 1)));
#define SYM_LAUNCH(F_, C_, K_, SGN_, SRC_, DST_)                                                                             \
    hipLaunchKernelGGL((k_dft_cols_sym<F_, C_, K_>), sym_grid, dim3(sym_threads), sym_shm, st, nz, nx, s->N1, s->N2,          \
                       (const int32_t*)s->d_in, (const int32_t*)s->d_in_inv, (const int32_t*)s->d_out, (const cplx*)s->d_w1,  \
                       (const cplx*)s->d_w2, SGN_, scale, SRC_, DST_, xcd_order, tl)
#define DFT_SYM(F_, SGN_, SRC_, DST_)                                                                                        \
    do { if (sym_cols == 4 && sym_kb == 2) SYM_LAUNCH(F_, 4, 2, SGN_, SRC_, DST_);                                            \
         else if (sym_cols == 4) SYM_LAUNCH(F_, 4, 3, SGN_, SRC_, DST_);                                                      \
         else if (sym_kb == 2) SYM_LAUNCH(F_, 2, 2, SGN_, SRC_, DST_);                                                        \
         else SYM_LAUNCH(F_, 2, 3, SGN_, SRC_, DST_); } while (0)
    // F^H X : exponent +, result transposed into T
    if (sym_threads) DFT_SYM(true, -1.0, (const cplx*)dX, s->d_T);
    else DFT_BY_COLS(true, -1.0, (const cplx*)dX, s->d_T);
    LAUNCHCHK();
    const int seg = (nx + 63) / 64;
    const dim3 g2((unsigned)((nz + 3) / 4));
#define TRI(S_) hipLaunchKernelGGL((k_tridiag_modes<S_, 0>), g2, dim3(256), 0, st, nz, nx, (const cplx*)s->d_m, (const cplx*)s->d_dinv, s->b, s->d_T)
    if (seg <= 1) TRI(1); else if (seg <= 2) TRI(2); else if (seg <= 4) TRI(4); else if (seg <= 8) TRI(8); else if (seg <= 16) TRI(16); else TRI(32);
#undef TRI
    LAUNCHCHK();
    // F T : exponent -, back to z fastest
    if (sym_threads) DFT_SYM(false, 1.0, (const cplx*)s->d_T, (cplx*)dX);
    else DFT_BY_COLS(false, 1.0, (const cplx*)s->d_T, (cplx*)dX);
    LAUNCHCHK();
#undef DFT_SYM
#undef SYM_LAUNCH
#undef DFT_BY_COLS
#undef DFT_LAUNCH
    return NEP_OK;
}

// ---- the whole Sylvester-SMW preconditioner application in THREE transforms (solve_smw, waveguide_preconditioner.jl:323-421) --------
//   r <- Linv r - Linv(sum_k alpha_k E_k),   alpha = M^{-1} f(Linv r),   Linv = F Tsolve F^H
// As issued through the pieces above this is four transforms, two tridiagonal sweeps and seven smaller kernels (region means of the
// back-transformed block, expansion into a full block, boundary pieces, three axpy).  Linv is linear and the region means are a
// linear functional of the MODE-space solution U1 = Tsolve(F^H r):  f = G S  with S the x-region sums of the modes (taken by the
// tridiagonal kernel from the rows it has just solved) and G the z-region means of the columns of F.  So
//   r <- F ( U1 - Tsolve(F^H E alpha) ):
//   1. T  = F^H r                       k_dft_cols_sym<forward>
//   2. U1 = Tsolve(T), S                k_tridiag_modes<.., 1>
//   3. f = G S, alpha = M^{-1} f        k_wep_mode_means, nep_gemv_hd
//   4. pb = P^{-1}(sigma) eb(alpha)     k_wep_pinv(_sym), the boundary pieces eb formed in its loader
//   5. T2 = F^H (E alpha)               k_dft_cols_sym<forward, EXPAND>: the expansion is formed in the loader from K_scaled, never stored
//   6. T  = U1 - Tsolve(T2)             k_tridiag_modes<.., 2>
//   7. r  = F T                         k_dft_cols_sym<inverse>
// 8 launches and ~11 passes over the 16 nx nz byte block instead of 17 launches and ~24 passes.  Needs the symmetric-half DFT form (odd
// coprime factors); NEP_ERR_UNSUPPORTED otherwise (the caller keeps the piecewise route).
// dMinvH: (M^{-1})^H, mm x mm column-major (nep_gemv_hd applies its conjugate transpose); dG: N x nz, row rz at dG + rz nz.
}  // extern "C"
struct SymCfg { int cols = 0, kb = 0, threads = 0; size_t shm = 0; unsigned grid = 0; };
static bool sylv_sym_cfg(const nep_wep_sylv* s, SymCfg& c) {
    static const int symcfg = getenv("NEP_WEP_DFT_SYM") ? atoi(getenv("NEP_WEP_DFT_SYM")) : 22;
    c.cols = symcfg / 10; c.kb = symcfg % 10;
    if (!(symcfg && (s->N1 & 1) && (s->N2 & 1) && s->N1 >= 3 && s->N2 >= 3 && (c.cols == 2 || c.cols == 4) && (c.kb == 2 || c.kb == 3)))
        return false;
    const int H1 = (s->N1 - 1) / 2, H2 = (s->N2 - 1) / 2;
    const int items = std::max(((H1 + c.kb - 1) / c.kb) * s->N2, ((H2 + c.kb - 1) / c.kb) * s->N1);
    c.threads = (items + 63) / 64 * 64;
    c.shm = ((size_t)c.cols * s->nz + s->N1 + s->N2) * sizeof(cplx);
    if (c.threads > 512 || c.shm > 150 * 1024) return false;
    c.grid = (unsigned)((s->nx + c.cols - 1) / c.cols);
    return true;
}
template <bool FWD, int EXPAND>
static int32_t sylv_dft_sym_launch(nep_wep_sylv* s, const SymCfg& c, double sgn, const cplx* src, cplx* dst, hipStream_t st, DftExpand ex,
                                   int batch = 1) {
    static const int xcd_order = getenv("NEP_WEP_DFT_XCD") ? atoi(getenv("NEP_WEP_DFT_XCD")) : 1;
    const double scale = 1.0 / sqrt((double)s->nz);
    const TLay tl{s->ldt, s->lseg};
#define SYMX(C_, K_)                                                                                                          \
    do {                                                                                                                      \
        static thread_local bool attr = false;                                                                                \
        if (!attr) { HIPCHK(hipFuncSetAttribute((const void*)k_dft_cols_sym<FWD, C_, K_, EXPAND>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024)); attr = true; } \
        const dim3 grid_ = EXPAND == 2 ? dim3((unsigned)((ex.L + C_ - 1) / C_), (unsigned)batch) : dim3(c.grid);                \
        hipLaunchKernelGGL((k_dft_cols_sym<FWD, C_, K_, EXPAND>), grid_, dim3(c.threads), c.shm, st, s->nz, s->nx, s->N1, s->N2, \
                           (const int32_t*)s->d_in, (const int32_t*)s->d_in_inv, (const int32_t*)s->d_out, (const cplx*)s->d_w1,  \
                           (const cplx*)s->d_w2, sgn, scale, src, dst, xcd_order, tl, ex);                                     \
    } while (0)
    if (c.cols == 4 && c.kb == 2) SYMX(4, 2); else if (c.cols == 4) SYMX(4, 3); else if (c.kb == 2) SYMX(2, 2); else SYMX(2, 3);
#undef SYMX
    LAUNCHCHK();
    return NEP_OK;
}
template <int MODE>
static int32_t sylv_tri_launch(nep_wep_sylv* s, hipStream_t st, cplx* T, const cplx* T2, cplx* S, int N, int L, int batch = 1,
                               int64_t tstride = 0, int64_t sstride = 0) {
    const int nz = s->nz, nx = s->nx;
    const int seg = (nx + 63) / 64;
    const dim3 g2((unsigned)((nz + 3) / 4), (unsigned)batch);
    const size_t shm = MODE == 1 ? (size_t)4 * nx * sizeof(cplx) : 0;
#define TRIX(S_)                                                                                                              \
    do {                                                                                                                      \
        static thread_local bool attr = false;                                                                                \
        if (MODE == 1 && !attr) { HIPCHK(hipFuncSetAttribute((const void*)k_tridiag_modes<S_, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024)); attr = true; } \
        hipLaunchKernelGGL((k_tridiag_modes<S_, MODE>), g2, dim3(256), shm, st, nz, nx, (const cplx*)s->d_m, (const cplx*)s->d_dinv, s->b, T, T2, S, N, L, tstride, sstride); \
    } while (0)
    if (seg <= 1) TRIX(1); else if (seg <= 2) TRIX(2); else if (seg <= 4) TRIX(4); else if (seg <= 8) TRIX(8); else if (seg <= 16) TRIX(16); else TRIX(32);
#undef TRIX
    LAUNCHCHK();
    return NEP_OK;
}
extern "C" {
int32_t nep_gemv_hd(const nep_cdouble* dA, int64_t lda, int64_t rows, int32_t k, const nep_cdouble* dx,
                    const nep_cdouble* dd, nep_cdouble* dy, nep_stream stream);

int32_t nep_wep_smw_apply(nep_wep_sylv* s, nep_wep_pinv* p, int32_t N, const nep_cdouble* dKsc, double dd1, double dd2,
                          const nep_cdouble* d_sinv, const nep_cdouble* dMinvH, const nep_cdouble* dG, nep_cdouble* dR,
                          nep_stream stream) {
    ARGCHK(s && p && dKsc && d_sinv && dMinvH && dG && dR && N >= 1 && s->nz % N == 0 && s->nx == s->nz + 4 && p->nz == s->nz);
    SymCfg c;
    if (!sylv_sym_cfg(s, c) || (size_t)4 * s->nx * sizeof(cplx) > 150 * 1024) return NEP_ERR_UNSUPPORTED;
    hipStream_t st = as_stream(stream);
    const int nz = s->nz, nx = s->nx, mm = N * (N + 4), L = nz / N;
    const size_t tsz = (size_t)nz * s->ldt;
    // second transposed block + S (nz (N+4)) + f, alpha (mm each) + eb, pb (2 nz each): one pool block, kept with the plan
    const size_t small = (size_t)nz * (N + 4) + 2 * (size_t)mm + 4 * (size_t)nz;
    if (!s->d_T2 || s->smw_N != N) {
        if (s->d_T2) { nep_pool_free(s->d_T2); s->d_T2 = nullptr; }
        int rc = nep_pool_alloc((void**)&s->d_T2, (tsz + small) * sizeof(cplx));
        if (rc) return rc;
        s->smw_N = N;
    }
    cplx* T2 = s->d_T2; cplx* S = T2 + tsz; cplx* f = S + (size_t)nz * (N + 4); cplx* al = f + mm; cplx* eb = al + mm; cplx* pb = eb + 2 * (size_t)nz;
    const DftExpand none{nullptr, nullptr, 0, 0, 0, 0};
    int32_t rc;
    if ((rc = sylv_dft_sym_launch<true, 0>(s, c, -1.0, (const cplx*)dR, s->d_T, st, none))) return rc;              // 1
    if ((rc = sylv_tri_launch<1>(s, st, s->d_T, nullptr, S, N, L))) return rc;                                      // 2
    hipLaunchKernelGGL(k_wep_mode_means, dim3((unsigned)((mm + 3) / 4)), dim3(256), 0, st, nz, (int)N, (const cplx*)dG, (const cplx*)S, f);
    LAUNCHCHK();                                                                                                    // 3
    if ((rc = nep_gemv_hd(dMinvH, mm, mm, mm, (const nep_cdouble*)f, nullptr, (nep_cdouble*)al, stream))) return rc;
    // 4: the boundary pieces eb(alpha) are formed inside the P^{-1} kernel's loader (gnx = -N)
    if ((rc = pinv_apply_impl(p, d_sinv, (const nep_cdouble*)eb, (nep_cdouble*)pb, stream, (const cplx*)al, -(int)N, dd1, dd2))) return rc;
    const DftExpand ex{al, pb, (int)N, L, 0, 0};
    if ((rc = sylv_dft_sym_launch<true, 1>(s, c, -1.0, (const cplx*)dKsc, T2, st, ex))) return rc;                  // 5
    if ((rc = sylv_tri_launch<2>(s, st, s->d_T, T2, nullptr, N, L))) return rc;                                     // 6
    return sylv_dft_sym_launch<false, 0>(s, c, 1.0, (const cplx*)s->d_T, (cplx*)dR, st, none);                      // 7
}

// alpha <- e_kappa (N x (N+4) block as a vector of mm entries)
__global__ void k_unit_vector(int mm, int kappa, cplx* __restrict__ a) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < mm) a[i] = cmake(i == kappa ? 1.0 : 0.0, 0.0);
}
// Y[:, 0] -= pb[0:nz], Y[:, nx-1] -= pb[nz:2nz]
__global__ void k_sub_boundary(int nz, int nx, const cplx* __restrict__ pb, cplx* __restrict__ Y) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nz) { Y[i].x -= pb[i].x; Y[i].y -= pb[i].y; }
    else if (i < 2 * nz) { cplx* y = Y + (int64_t)nz * (nx - 1) + (i - nz); y->x -= pb[i].x; y->y -= pb[i].y; }
}

int32_t nep_wep_region_means(int32_t nz, int32_t nx, int32_t N, const nep_cdouble* dX, nep_cdouble* dOut, nep_stream stream);
int32_t nep_wep_region_expand(int32_t nz, int32_t nx, int32_t N, const nep_cdouble* dAlpha, const nep_cdouble* dKsc, double dd1,
                              double dd2, nep_cdouble* dY, nep_cdouble* dEb, nep_stream stream);

// Sylvester-SMW matrix, all mm = N (N+4) columns in one call (generate_smw_matrix, waveguide_preconditioner.jl:221-313):
//   column kappa of dM (mm x mm, column-major) = region means of Linv(E_kappa),  E_kappa = expansion of the unit vector e_kappa
//   (K_scaled on region kappa, minus P^{-1}(sigma) of the boundary pieces in the first / last grid column).
// The caller adds the identity and inverts (host, mm x mm).  dWork: nz*nx + 4 nz + mm complex of scratch.
int32_t nep_wep_smw_matrix(nep_wep_sylv* s, nep_wep_pinv* p, int32_t N, const nep_cdouble* dKsc, double dd1, double dd2,
                           const nep_cdouble* d_sinv, nep_cdouble* dWork, nep_cdouble* dM, nep_stream stream) {
    ARGCHK(s && p && dKsc && d_sinv && dWork && dM && N >= 1 && s->nz % N == 0 && s->nx == s->nz + 4 && p->nz == s->nz);
    hipStream_t st = as_stream(stream);
    const int nz = s->nz, nx = s->nx, mm = N * (N + 4);
    cplx* Y = (cplx*)dWork; cplx* eb = Y + (size_t)nz * nx; cplx* pb = eb + 2 * (size_t)nz; cplx* alpha = pb + 2 * (size_t)nz;
    for (int kappa = 0; kappa < mm; ++kappa) {
        hipLaunchKernelGGL(k_unit_vector, dim3((mm + 255) / 256), dim3(256), 0, st, mm, kappa, alpha);
        LAUNCHCHK();
        int rc = nep_wep_region_expand(nz, nx, N, (const nep_cdouble*)alpha, dKsc, dd1, dd2, (nep_cdouble*)Y, (nep_cdouble*)eb, stream);
        if (!rc) rc = nep_wep_pinv_apply(p, d_sinv, (const nep_cdouble*)eb, (nep_cdouble*)pb, stream);
        if (rc) return rc;
        hipLaunchKernelGGL(k_sub_boundary, dim3((2 * nz + 255) / 256), dim3(256), 0, st, nz, nx, (const cplx*)pb, Y);
        LAUNCHCHK();
        rc = nep_wep_sylv_solve(s, (nep_cdouble*)Y, stream);
        if (!rc) rc = nep_wep_region_means(nz, nx, N, (const nep_cdouble*)Y, dM + (size_t)kappa * mm, stream);
        if (rc) return rc;
    }
    return NEP_OK;
}

// The same matrix through the pieces of nep_wep_smw_apply: column kappa = G S(Tsolve(F^H E_kappa)) -- no back transform, no
// expanded block, and for the N^2 interior regions (E_kappa = K_scaled on one L x L block, no boundary pieces) only the L columns of
// the region are transformed, `batch` columns of the matrix per launch (the tridiagonal kernel alone has 250 workgroups per
// right-hand side block: one column at a time left the chip three quarters empty).  The 4 N boundary-region columns take the
// general loader (EXPAND = 1) one at a time.  NEP_ERR_UNSUPPORTED: see nep_wep_smw_apply.
int32_t nep_wep_smw_matrix_modes(nep_wep_sylv* s, nep_wep_pinv* p, int32_t N, const nep_cdouble* dKsc, double dd1, double dd2,
                                 const nep_cdouble* d_sinv, const nep_cdouble* dG, nep_cdouble* dM, nep_stream stream) {
    ARGCHK(s && p && dKsc && d_sinv && dG && dM && N >= 1 && s->nz % N == 0 && s->nx == s->nz + 4 && p->nz == s->nz);
    SymCfg c;
    if (!sylv_sym_cfg(s, c) || (size_t)4 * s->nx * sizeof(cplx) > 150 * 1024) return NEP_ERR_UNSUPPORTED;
    hipStream_t st = as_stream(stream);
    const int nz = s->nz, mm = N * (N + 4), L = nz / N;
    const int64_t tsz = (int64_t)nz * s->ldt, ssz = (int64_t)nz * (N + 4);
    int B = getenv("NEP_WEP_SMW_BATCH") ? atoi(getenv("NEP_WEP_SMW_BATCH")) : 16;
    B = std::max(1, std::min(B, (int)std::max<int64_t>(1, ((int64_t)1 << 30) / (tsz * (int64_t)sizeof(cplx)))));
    cplx* work = nullptr;
    int32_t rc = nep_pool_alloc((void**)&work, ((size_t)B * (tsz + ssz) + mm + 4 * (size_t)nz) * sizeof(cplx));
    if (rc) return rc;
    cplx* Tb = work; cplx* Sb = Tb + (size_t)B * tsz; cplx* alpha = Sb + (size_t)B * ssz; cplx* eb = alpha + mm; cplx* pb = eb + 2 * (size_t)nz;
    auto done = [&](int32_t code) { nep_pool_free_on(work, st, true); return code; };
    const dim3 gm((unsigned)((mm + 3) / 4));
    // interior regions, kappa = rz + N rx with 2 <= rx < N + 2: consecutive kappa, B per round
    for (int kap0 = 2 * N; kap0 < (N + 2) * N; kap0 += B) {
        const int nb = std::min(B, (N + 2) * N - kap0);
        hipError_t e = hipMemsetAsync(Tb, 0, (size_t)nb * tsz * sizeof(cplx), st);
        if (e != hipSuccess) { nep_set_error("nep_wep_smw_matrix_modes: %s", hipGetErrorString(e)); return done(NEP_ERR_HIP); }
        const DftExpand ex{nullptr, nullptr, (int)N, L, kap0, tsz};
        if ((rc = sylv_dft_sym_launch<true, 2>(s, c, -1.0, (const cplx*)dKsc, Tb, st, ex, nb))) return done(rc);
        if ((rc = sylv_tri_launch<1>(s, st, Tb, nullptr, Sb, N, L, nb, tsz, ssz))) return done(rc);
        hipLaunchKernelGGL(k_wep_mode_means, dim3(gm.x, (unsigned)nb), dim3(256), 0, st, nz, (int)N, (const cplx*)dG, (const cplx*)Sb,
                           (cplx*)dM + (size_t)kap0 * mm, ssz, (int64_t)mm);
        if (hipGetLastError() != hipSuccess) { nep_set_error("nep_wep_smw_matrix_modes: launch failed"); return done(NEP_ERR_HIP); }
    }
    // the four boundary-region columns of x: unit alpha through the general loader (boundary pieces included)
    for (int kappa = 0; kappa < mm; ++kappa) {
        if (kappa >= 2 * N && kappa < (N + 2) * N) continue;
        hipLaunchKernelGGL(k_unit_vector, dim3((mm + 255) / 256), dim3(256), 0, st, mm, kappa, alpha);
        hipLaunchKernelGGL(k_region_eb, dim3((unsigned)((nz + 255) / 256)), dim3(256), 0, st, nz, (int)N, L, (const cplx*)alpha, dd1, dd2, eb);
        if ((rc = pinv_apply_impl(p, d_sinv, (const nep_cdouble*)eb, (nep_cdouble*)pb, stream, nullptr, 0, 0.0, 0.0))) return done(rc);
        const DftExpand ex{alpha, pb, (int)N, L, 0, 0};
        if ((rc = sylv_dft_sym_launch<true, 1>(s, c, -1.0, (const cplx*)dKsc, Tb, st, ex))) return done(rc);
        if ((rc = sylv_tri_launch<1>(s, st, Tb, nullptr, Sb, N, L))) return done(rc);
        hipLaunchKernelGGL(k_wep_mode_means, gm, dim3(256), 0, st, nz, (int)N, (const cplx*)dG, (const cplx*)Sb, (cplx*)dM + (size_t)kappa * mm,
                           (int64_t)0, (int64_t)0);
        if (hipGetLastError() != hipSuccess) { nep_set_error("nep_wep_smw_matrix_modes: launch failed"); return done(NEP_ERR_HIP); }
    }
    return done(NEP_OK);
}

int32_t nep_wep_region_means(int32_t nz, int32_t nx, int32_t N, const nep_cdouble* dX, nep_cdouble* dOut, nep_stream stream) {
    ARGCHK(dX && dOut && N >= 1 && nz % N == 0 && nx == nz + 4);
    hipLaunchKernelGGL(k_region_means, dim3((unsigned)N, (unsigned)(N + 4)), dim3(256), 0, as_stream(stream), (int)nz, (int)nx, (int)N,
                       (int)(nz / N), (const cplx*)dX, (cplx*)dOut);
    LAUNCHCHK();
    return NEP_OK;
}

int32_t nep_wep_region_expand(int32_t nz, int32_t nx, int32_t N, const nep_cdouble* dAlpha, const nep_cdouble* dKsc, double dd1,
                              double dd2, nep_cdouble* dY, nep_cdouble* dEb, nep_stream stream) {
    ARGCHK(dAlpha && dKsc && dY && dEb && N >= 1 && nz % N == 0 && nx == nz + 4);
    const int64_t total = (int64_t)nz * nx;
    hipLaunchKernelGGL(k_region_expand, dim3((unsigned)std::min<int64_t>((total + 255) / 256, 4096)), dim3(256), 0, as_stream(stream),
                       (int)nz, (int)nx, (int)N, (int)(nz / N), (const cplx*)dAlpha, (const cplx*)dKsc, dd1, dd2, (cplx*)dY, (cplx*)dEb);
    LAUNCHCHK();
    return NEP_OK;
}

}  // extern "C"
