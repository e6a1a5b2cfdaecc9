// libnepmi355: K5 fixed-shift solve  x = A^{-1} b  from a host-computed sparse LU (gfx950).
//
// Pr*A*Pc = L*U comes from the host (SuperLU/UMFPACK-class factorisation, one-off per shift:
// src/LinSolvers.jl:114-116).  A sparse triangular solve is a chain of dependent "levels"
// (~900 levels each in L and U for gun, most of them holding ONE row -- the dense
// trailing block that every sparse LU ends in).  The schedule built by nep_lu_create therefore
// splits the factors at a tail index i0 (T = n - i0 trailing rows/columns):
//
//      L = [L11  0 ]   U = [U11 U12]      S22 = L22*U22  (T x T, dense-ish)
//          [L21 L22]       [ 0  U22]
//
//   head  L11 / U11 : level-scheduled.  Wide levels (many independent rows) run as ONE multi-
//                     workgroup launch each (G lanes per row); runs of consecutive narrow levels
//                     run inside ONE persistent workgroup with a workgroup barrier per level.
//   mid   (blocked) : rows [h0, i0) between head and tail, where levels hold a handful of long rows (the fronts of
//                     the top separators: n = 1e6 waveguide -> 8500 levels, 200 ms per solve level-scheduled).
//                     Cut into blocks of b rows; per block ONE multi-workgroup SpMV with everything outside the
//                     block (r_B = c_B - L[B,<B] x) and ONE dense GEMV with the explicitly inverted b x b diagonal
//                     block (x_B = inv(L_BB) r_B): b dependent levels become 2 launches.  The inverses are built once
//                     per factorisation on the device (one workgroup per column, x in LDS).
//   tail            : t = c2 - L21*y1 (SpMV) ; x2 = S22^{-1} t as ONE dense GEMV.  S22^{-1} is built
//                     once per factorisation ON THE DEVICE by T independent tail solves (one
//                     workgroup per unit vector, x in LDS), turning ~2T dependent steps per solve
//                     into a single bandwidth-bound pass over 16 T^2 bytes.
// All kernels take grid.y = right-hand side index, so Beyn's n x k block solve
// (src/method_beyncontour.jl:91-93) fills the chip.
#include "common.h"
#include "trsv_ml.h"
#include <vector>
#include <algorithm>
#include <chrono>
#include <thread>
#include <cstring>

#define TRSV_WIDE_MIN 48      // a level with at least this many rows gets its own multi-WG launch
#define TRSV_TAIL_SMALL 4     // levels with <= this many rows are "chain" levels (dense tail)
#define TRSV_TAIL_MAX 2048    // cap on the dense tail size (16*T^2 bytes = 67 MB); beyond it the blocked mid region is cheaper to build
#define TRSV_TAIL_MIN 64
#define TRSV_MID_MAX 131072   // cap on the rows of the blocked mid region (2 * 16*b bytes of inverse per row)

struct Seg { int wide; int lev_lo, lev_hi; int slot_lo, slot_hi; int G; };

struct TriFactor {
    int32_t nlev = 0;
    int64_t nrows = 0;             // rows in this (sub)factor
    int64_t nnz = 0;
    int32_t* d_levptr = nullptr;   // nlev+1 positions into level-ordered row slots
    int32_t* d_rowid = nullptr;    // global row index of slot s
    int32_t* d_rowptr = nullptr;   // nrows+1 over slots
    int32_t* d_col = nullptr;      // column indices (global numbering), diagonal excluded
    cplx* d_val = nullptr;
    cplx* d_diag = nullptr;        // per slot (upper only)
    std::vector<Seg> segs;
};

// blocked mid region of one triangular factor: rows [h0, h0 + nblk*b)
struct MidFactor {
    int32_t* d_rp = nullptr;       // CSR over mid rows: entries OUTSIDE the row's diagonal block
    int32_t* d_ci = nullptr;       //   (lower: col < block start; upper: col >= block end)
    cplx* d_vx = nullptr;
    cplx* d_inv = nullptr;         // nblk dense b x b row-major inverses of the diagonal blocks
    int64_t nnz = 0;
    std::vector<int> wpr;          // waves per row of the block's SpMV launch (1 or 4)
};

struct nep_lu {
    int64_t n = 0, i0 = 0, T = 0;
    int64_t h0 = 0;                // heads cover rows [0,h0), mid [h0,i0), tail [i0,n)
    int32_t nblk = 0, bsz = 0;
    MidFactor Lm, Um;
    TriFactor L11, U11;
    // L21 as CSR over tail rows
    int32_t* d_L21p = nullptr; int32_t* d_L21i = nullptr; cplx* d_L21x = nullptr; int64_t nnzL21 = 0;
    cplx* d_Sinv = nullptr;        // T x T row-major
    int32_t* d_perm_r = nullptr;
    int32_t* d_perm_c = nullptr;
    NepScratch work;               // (n + (n - h0)) x nrhs
    int64_t nnzL_in = 0, nnzU_in = 0;
    int32_t levL_full = 0, levU_full = 0;
    int32_t launches = 0;
    // the level sweep between the two permutation kernels as an instantiated hipGraph (one per nrhs in use):
    // ~90 small dependent launches per solve are host-launch-bound when issued eagerly
    hipGraphExec_t graph_exec = nullptr;
    int32_t graph_nrhs = 0;
    void* graph_work = nullptr;
    hipStream_t cap_stream = nullptr;
    int32_t use_graph = 1;
    MLFactor* ml = nullptr;        // elimination-tree block schedule (trsv_ml.hip); when set, the fields above are unused
    int32_t csc = 0;               // layout the caller's factors came in (nep_lu_refactor expects the same)
    hipStream_t last = nullptr;    // stream of the last solve: frees are ordered behind it
    bool used = false;
};

__device__ __forceinline__ cplx cdiv(cplx a, cplx b) {
    // Smith's algorithm (robust against overflow of |b|^2)
    if (fabs(b.x) >= fabs(b.y)) {
        const double r = b.y / b.x, d = b.x + b.y * r;
        return cmake((a.x + a.y * r) / d, (a.y - a.x * r) / d);
    } else {
        const double r = b.x / b.y, d = b.x * r + b.y;
        return cmake((a.x * r + a.y) / d, (a.y * r - a.x) / d);
    }
}

// ---- permutations ------------------------------------------------------------------------------
__global__ void k_perm_in(int64_t n, const int32_t* __restrict__ perm_r, const cplx* __restrict__ B, int64_t ldb,
                          cplx* __restrict__ work, int64_t ldw) {
    const cplx* b = B + (int64_t)blockIdx.y * ldb;
    cplx* x = work + (int64_t)blockIdx.y * ldw;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        x[perm_r ? perm_r[i] : i] = b[i];
}
__global__ void k_perm_out(int64_t n, const int32_t* __restrict__ perm_c, const cplx* __restrict__ work, int64_t ldw,
                           cplx* __restrict__ X, int64_t ldx, double scale, const cplx* __restrict__ add, int64_t lda) {
    const cplx* x = work + (int64_t)blockIdx.y * ldw;
    cplx* xo = X + (int64_t)blockIdx.y * ldx;
    const cplx* ad = add ? add + (int64_t)blockIdx.y * lda : nullptr;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        cplx v = x[perm_c ? perm_c[i] : i];
        if (ad) { v.x += ad[i].x; v.y += ad[i].y; }
        xo[i] = cmake(scale * v.x, scale * v.y);
    }
}

// ---- one wide level: G lanes per row, many workgroups -------------------------------------------
template <int G, bool UPPER>
__global__ __launch_bounds__(256) void k_level_wide(int slot_lo, int slot_hi, const int32_t* __restrict__ rowid,
                                                    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                    const cplx* __restrict__ val, const cplx* __restrict__ diag,
                                                    cplx* work, int64_t ldw) {
    constexpr int RPB = 256 / G;
    cplx* x = work + (int64_t)blockIdx.y * ldw;
    const int sub = threadIdx.x % G;
    const int s = slot_lo + blockIdx.x * RPB + threadIdx.x / G;
    cplx acc = cmake(0.0, 0.0);
    if (s < slot_hi) {
        const int e1 = rowptr[s + 1];
        for (int e = rowptr[s] + sub; e < e1; e += G) cfma(acc, val[e], x[col[e]]);
    }
    acc = group_reduce_sum<G>(acc);
    if (s < slot_hi && sub == 0) {
        const int i = rowid[s];
        cplx v = csub(x[i], acc);
        if (UPPER) v = cdiv(v, diag[s]);
        x[i] = v;
    }
}

// ---- a run of narrow levels inside one persistent workgroup (per right-hand side) ----------------
template <bool UPPER>
__global__ __launch_bounds__(512) void k_levels_narrow(int lev_lo, int lev_hi, const int32_t* __restrict__ levptr,
                                                       const int32_t* __restrict__ rowid,
                                                       const int32_t* __restrict__ rowptr,
                                                       const int32_t* __restrict__ col, const cplx* __restrict__ val,
                                                       const cplx* __restrict__ diag, cplx* work, int64_t ldw) {
    cplx* x = work + (int64_t)blockIdx.y * ldw;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int lev = lev_lo; lev < lev_hi; ++lev) {
        const int s0 = levptr[lev], s1 = levptr[lev + 1];
        for (int s = s0 + wv; s < s1; s += 8) {
            const int e0 = rowptr[s], e1 = rowptr[s + 1];
            cplx acc = cmake(0.0, 0.0);
            for (int e = e0 + lane; e < e1; e += 64) cfma(acc, val[e], x[col[e]]);
            acc = group_reduce_sum<64>(acc);
            if (lane == 0) {
                const int i = rowid[s];
                cplx v = csub(x[i], acc);
                if (UPPER) v = cdiv(v, diag[s]);
                x[i] = v;
            }
        }
        __syncthreads();
    }
}

// ---- tail: x[i0:] = Sinv * tmp   (dense row-major GEMV, wave per row, HBM/L2-bound) --------------
// RB right-hand sides share one pass over the matrix row (PMC: with one launch row per right-hand side the 32-column
// block solve of Beyn re-read Sinv 32 times, 1.86 GB of HBM traffic for a 67 MB matrix)
template <int RB>
__global__ __launch_bounds__(256) void k_tail_gemv(int64_t T, int64_t i0, const cplx* __restrict__ Sinv,
                                                   const cplx* __restrict__ tmp, int64_t ldt, cplx* __restrict__ work,
                                                   int64_t ldw, int nrhs) {
    const int rhs0 = blockIdx.y * RB;
    const int nb = min(RB, nrhs - rhs0);
    const cplx* t = tmp + (int64_t)rhs0 * ldt;
    cplx* x = work + (int64_t)rhs0 * ldw;
    const int lane = threadIdx.x & 63;
    const int64_t r = blockIdx.x * 4LL + (threadIdx.x >> 6);
    if (r >= T) return;
    const cplx* row = Sinv + r * T;
    cplx acc[RB];
#pragma unroll
    for (int q = 0; q < RB; ++q) acc[q] = cmake(0.0, 0.0);
#pragma unroll 2
    for (int64_t c = lane; c < T; c += 64) {
        const cplx m = row[c];
#pragma unroll
        for (int q = 0; q < RB; ++q)
            if (q < nb) cfma(acc[q], m, t[(int64_t)q * ldt + c]);
    }
#pragma unroll
    for (int q = 0; q < RB; ++q) {
        const cplx a = group_reduce_sum<64>(acc[q]);
        if (lane == 0 && q < nb) x[(int64_t)q * ldw + i0 + r] = a;
    }
}

// ---- mid: tmp[B] = x[B] - (off-block part of rows B) * x   (WPR waves per row; WPR=16 -> 1024 threads) ------------
template <int WPR>
__global__ __launch_bounds__(WPR == 16 ? 1024 : 256) void k_mid_spmv(int64_t r0, int nrows, int64_t h0,
                                                                      const int32_t* __restrict__ rp,
                                                                      const int32_t* __restrict__ ci,
                                                                      const cplx* __restrict__ vx,
                                                                      const cplx* __restrict__ work, int64_t ldw,
                                                                      cplx* __restrict__ tmp, int64_t ldt) {
    constexpr int NW = WPR == 16 ? 16 : 4;               // waves per workgroup
    constexpr int RPB = NW / WPR;
    __shared__ cplx part[NW];
    const cplx* x = work + (int64_t)blockIdx.y * ldw;
    cplx* t = tmp + (int64_t)blockIdx.y * ldt;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int lr = blockIdx.x * RPB + wv / WPR;          // uniform per wave
    const bool live = lr < nrows;
    const int64_t row = r0 + lr;
    cplx acc = cmake(0.0, 0.0);
    if (live) {
        const int e1 = rp[row - h0 + 1];
        for (int e = rp[row - h0] + (wv % WPR) * 64 + lane; e < e1; e += 64 * WPR) cfma(acc, vx[e], x[ci[e]]);
    }
    acc = group_reduce_sum<64>(acc);
    if (WPR == 1) {
        if (live && lane == 0) t[row - h0] = csub(x[row], acc);
    } else {
        if (lane == 0) part[wv] = acc;
        __syncthreads();
        if (live && threadIdx.x == 0) {
            cplx a = part[0];
            for (int w = 1; w < NW; ++w) { a.x += part[w].x; a.y += part[w].y; }
            t[row - h0] = csub(x[row], a);
        }
    }
}

static int pick_wpr(double avg_row_nnz) { return avg_row_nnz > 6144.0 ? 16 : (avg_row_nnz > 768.0 ? 4 : 1); }

static void launch_mid_spmv(int wpr, int64_t r0, int nrows, int64_t h0, const int32_t* rp, const int32_t* ci,
                            const cplx* vx, const cplx* work, int64_t ldw, cplx* tmp, int64_t ldt, int nrhs,
                            hipStream_t st) {
    if (wpr == 16)
        hipLaunchKernelGGL((k_mid_spmv<16>), dim3(nrows, nrhs), dim3(1024), 0, st, r0, nrows, h0, rp, ci, vx, work, ldw, tmp, ldt);
    else if (wpr == 4)
        hipLaunchKernelGGL((k_mid_spmv<4>), dim3(nrows, nrhs), dim3(256), 0, st, r0, nrows, h0, rp, ci, vx, work, ldw, tmp, ldt);
    else
        hipLaunchKernelGGL((k_mid_spmv<1>), dim3((nrows + 3) / 4, nrhs), dim3(256), 0, st, r0, nrows, h0, rp, ci, vx, work, ldw, tmp, ldt);
}

// ---- mid: x[B] = inv(D_B) * tmp[B]   (triangular dense GEMV, WPR waves per row, RB right-hand sides per pass) ------
template <bool UPPER, int WPR, int RB>
__global__ __launch_bounds__(256) void k_mid_gemv(int64_t r0, int b, int64_t h0, const cplx* __restrict__ inv,
                                                  const cplx* __restrict__ tmp, int64_t ldt, cplx* __restrict__ work,
                                                  int64_t ldw, int nrhs) {
    __shared__ cplx part[4][RB];
    const int rhs0 = blockIdx.y * RB;
    const int nb = min(RB, nrhs - rhs0);
    const cplx* t = tmp + (int64_t)rhs0 * ldt + (r0 - h0);
    cplx* x = work + (int64_t)rhs0 * ldw;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    constexpr int RPB = 4 / WPR;
    const int r = blockIdx.x * RPB + wv / WPR;
    const bool live = r < b;
    cplx acc[RB];
#pragma unroll
    for (int q = 0; q < RB; ++q) acc[q] = cmake(0.0, 0.0);
    if (live) {
        const cplx* row = inv + (int64_t)r * b;
        const int c0 = UPPER ? r : 0, c1 = UPPER ? b : r + 1;
        for (int c = (c0 & ~63) + (wv % WPR) * 64 + lane; c < c1; c += 64 * WPR)
            if (c >= c0) {
                const cplx m = row[c];
#pragma unroll
                for (int q = 0; q < RB; ++q)
                    if (q < nb) cfma(acc[q], m, t[(int64_t)q * ldt + c]);
            }
    }
#pragma unroll
    for (int q = 0; q < RB; ++q) acc[q] = group_reduce_sum<64>(acc[q]);
    if (WPR == 1) {
        if (live && lane == 0)
            for (int q = 0; q < nb; ++q) x[(int64_t)q * ldw + r0 + r] = acc[q];
    } else {
        if (lane == 0)
            for (int q = 0; q < RB; ++q) part[wv][q] = acc[q];
        __syncthreads();
        if (live && threadIdx.x < nb) {
            const int q = threadIdx.x;
            cplx a = part[0][q];
            for (int w = 1; w < 4; ++w) { a.x += part[w][q].x; a.y += part[w][q].y; }
            x[(int64_t)q * ldw + r0 + r] = a;
        }
    }
}

// ---- setup: column j of inv(D_B) for every diagonal block B (grid = b x nblk), x in LDS --------------------------
// In-block level schedule: blkoff[k] indexes `levptr`, which holds nlev_k + 1 slot positions for block k.
template <bool UPPER>
__global__ __launch_bounds__(256) void k_mid_inverse(int b, int64_t h0, const int32_t* __restrict__ blkoff,
                                                     const int32_t* __restrict__ levptr,
                                                     const int32_t* __restrict__ rowid, const int32_t* __restrict__ rowptr,
                                                     const int32_t* __restrict__ col, const cplx* __restrict__ val,
                                                     const cplx* __restrict__ diag, cplx* __restrict__ inv) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    cplx* x = (cplx*)smem_raw;
    const int j = blockIdx.x, k = blockIdx.y;
    const int64_t s0row = h0 + (int64_t)k * b;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int t = threadIdx.x; t < b; t += 256) x[t] = cmake(t == j ? 1.0 : 0.0, 0.0);
    __syncthreads();
    const int l0 = blkoff[k], nlev = blkoff[k + 1] - l0 - 1;
    for (int lev = UPPER ? 0 : 1; lev < nlev; ++lev) {
        const int s0 = levptr[l0 + lev], s1 = levptr[l0 + lev + 1];
        for (int s = s0 + wv; s < s1; s += 4) {
            const int i = (int)(rowid[s] - s0row);
            if (UPPER ? (i <= j) : (i > j)) {        // the other rows of the column stay zero (uniform per wave)
                cplx acc = cmake(0.0, 0.0);
                for (int e = rowptr[s] + lane; e < rowptr[s + 1]; e += 64) cfma(acc, val[e], x[col[e] - s0row]);
                acc = group_reduce_sum<64>(acc);
                if (lane == 0) x[i] = UPPER ? cdiv(csub(x[i], acc), diag[s]) : csub(x[i], acc);
            }
        }
        __syncthreads();
    }
    cplx* out = inv + (int64_t)k * b * b;
    for (int t = threadIdx.x; t < b; t += 256) out[(int64_t)t * b + j] = x[t];
}

// ---- setup: column j of S22^{-1} = U22^{-1} L22^{-1} e_j, one workgroup per column, x in LDS -------
__global__ __launch_bounds__(512) void k_tail_inverse(int T, int i0,
                                                      const int32_t* __restrict__ Llevptr, int Lnlev,
                                                      const int32_t* __restrict__ Lrowid, const int32_t* __restrict__ Lrowptr,
                                                      const int32_t* __restrict__ Lcol, const cplx* __restrict__ Lval,
                                                      const int32_t* __restrict__ Ulevptr, int Unlev,
                                                      const int32_t* __restrict__ Urowid, const int32_t* __restrict__ Urowptr,
                                                      const int32_t* __restrict__ Ucol, const cplx* __restrict__ Uval,
                                                      const cplx* __restrict__ Udiag, cplx* __restrict__ Sinv) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    cplx* x = (cplx*)smem_raw;  // T entries, local (tail) numbering
    const int j = blockIdx.x;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int t = threadIdx.x; t < T; t += 512) x[t] = cmake(t == j ? 1.0 : 0.0, 0.0);
    __syncthreads();
    for (int lev = 1; lev < Lnlev; ++lev) {      // level 0 of a unit-lower factor needs no work
        const int s0 = Llevptr[lev], s1 = Llevptr[lev + 1];
        for (int s = s0 + wv; s < s1; s += 8) {
            const int i = Lrowid[s] - i0;
            if (i > j) {                          // rows above j stay zero (uniform per wave)
                cplx acc = cmake(0.0, 0.0);
                for (int e = Lrowptr[s] + lane; e < Lrowptr[s + 1]; e += 64) cfma(acc, Lval[e], x[Lcol[e] - i0]);
                acc = group_reduce_sum<64>(acc);
                if (lane == 0) x[i] = csub(x[i], acc);
            }
        }
        __syncthreads();
    }
    for (int lev = 0; lev < Unlev; ++lev) {
        const int s0 = Ulevptr[lev], s1 = Ulevptr[lev + 1];
        for (int s = s0 + wv; s < s1; s += 8) {
            const int i = Urowid[s] - i0;
            cplx acc = cmake(0.0, 0.0);
            for (int e = Urowptr[s] + lane; e < Urowptr[s + 1]; e += 64) cfma(acc, Uval[e], x[Ucol[e] - i0]);
            acc = group_reduce_sum<64>(acc);
            if (lane == 0) x[i] = cdiv(csub(x[i], acc), Udiag[s]);
        }
        __syncthreads();
    }
    for (int t = threadIdx.x; t < T; t += 512) Sinv[(int64_t)t * T + j] = x[t];
}

// ------------------------------------------------------------------------------------------------
static void free_tri(TriFactor& t) {
    if (t.d_levptr) nep_pool_free(t.d_levptr);
    if (t.d_rowid) nep_pool_free(t.d_rowid);
    if (t.d_rowptr) nep_pool_free(t.d_rowptr);
    if (t.d_col) nep_pool_free(t.d_col);
    if (t.d_val) nep_pool_free(t.d_val);
    if (t.d_diag) nep_pool_free(t.d_diag);
    t = TriFactor();
}

// levels of the sub-triangle rows [r_lo, r_hi), considering only dependencies with columns in [r_lo, r_hi)
static int compute_levels(int64_t n, const int32_t* P, const int32_t* I, bool upper, int64_t r_lo, int64_t r_hi,
                          std::vector<int32_t>& level, int32_t& nlev) {
    nlev = 0;
    if (!upper) {
        for (int64_t i = r_lo; i < r_hi; ++i) {
            int32_t lv = 0;
            for (int32_t e = P[i]; e < P[i + 1]; ++e) {
                const int32_t j = I[e];
                if (j >= r_lo && j < i) lv = std::max(lv, level[j] + 1);
            }
            level[i] = lv; nlev = std::max(nlev, lv + 1);
        }
    } else {
        for (int64_t i = r_hi - 1; i >= r_lo; --i) {
            int32_t lv = 0;
            for (int32_t e = P[i]; e < P[i + 1]; ++e) {
                const int32_t j = I[e];
                if (j > i && j < r_hi) lv = std::max(lv, level[j] + 1);
            }
            level[i] = lv; nlev = std::max(nlev, lv + 1);
        }
    }
    return NEP_OK;
}

// builds a level-ordered factor for rows [r_lo,r_hi); keeps entries with column in [c_lo, c_hi) (diag excluded)
static int build_tri(int64_t n, const int32_t* P, const int32_t* I, const nep_cdouble* X, bool upper, int64_t r_lo,
                     int64_t r_hi, int64_t c_lo, int64_t c_hi, const std::vector<int32_t>& level, int32_t nlev,
                     TriFactor& out) {
    const int64_t nr = r_hi - r_lo;
    out.nrows = nr; out.nlev = nlev;
    if (nr == 0) { out.nlev = 0; return NEP_OK; }
    std::vector<int32_t> levptr(nlev + 1, 0);
    for (int64_t i = r_lo; i < r_hi; ++i) levptr[level[i] + 1]++;
    for (int32_t l = 0; l < nlev; ++l) levptr[l + 1] += levptr[l];
    std::vector<int32_t> rowid(nr), pos(levptr.begin(), levptr.end() - 1);
    for (int64_t i = r_lo; i < r_hi; ++i) rowid[pos[level[i]]++] = (int32_t)i;
    std::vector<int32_t> rowptr(nr + 1, 0), col;
    std::vector<nep_cdouble> val, diag(upper ? nr : 0);
    std::vector<double> lev_nnz(nlev, 0.0);
    for (int64_t s = 0; s < nr; ++s) {
        const int32_t i = rowid[s];
        bool have_diag = false;
        for (int32_t e = P[i]; e < P[i + 1]; ++e) {
            const int32_t j = I[e];
            if (j == i) { have_diag = true; if (upper) diag[s] = X[e]; continue; }
            if (j < c_lo || j >= c_hi) continue;
            col.push_back(j); val.push_back(X[e]);
        }
        if (upper && (!have_diag || (diag[s].re == 0.0 && diag[s].im == 0.0))) {
            nep_set_error("U has a zero pivot in row %d (matrix is singular)", i);
            return NEP_ERR_SINGULAR;
        }
        rowptr[s + 1] = (int32_t)col.size();
        lev_nnz[level[i]] += rowptr[s + 1] - rowptr[s];
    }
    out.nnz = (int64_t)col.size();
    // ---- segments: wide levels get their own launch, runs of narrow levels share a persistent WG
    out.segs.clear();
    const int first = upper ? 0 : 1;   // level 0 of unit-lower L: x_i = c_i already
    int l = first;
    while (l < nlev) {
        const int rows = levptr[l + 1] - levptr[l];
        if (rows >= TRSV_WIDE_MIN) {
            Seg sg; sg.wide = 1; sg.lev_lo = l; sg.lev_hi = l + 1; sg.slot_lo = levptr[l]; sg.slot_hi = levptr[l + 1];
            const double avg = lev_nnz[l] / std::max(rows, 1);
            sg.G = avg <= 12 ? 8 : (avg <= 40 ? 16 : 64);
            out.segs.push_back(sg);
            ++l;
        } else {
            int h = l;
            while (h < nlev && levptr[h + 1] - levptr[h] < TRSV_WIDE_MIN) ++h;
            Seg sg; sg.wide = 0; sg.lev_lo = l; sg.lev_hi = h; sg.slot_lo = levptr[l]; sg.slot_hi = levptr[h]; sg.G = 64;
            out.segs.push_back(sg);
            l = h;
        }
    }
    const size_t nnz = col.size();
    { int prc_ = nep_pool_alloc((void**)&out.d_levptr, (size_t)(nlev + 1) * 4); if (prc_) return prc_; }
    { int prc_ = nep_pool_alloc((void**)&out.d_rowid, (size_t)nr * 4); if (prc_) return prc_; }
    { int prc_ = nep_pool_alloc((void**)&out.d_rowptr, (size_t)(nr + 1) * 4); if (prc_) return prc_; }
    { int prc_ = nep_pool_alloc((void**)&out.d_col, (nnz + 1) * 4); if (prc_) return prc_; }
    { int prc_ = nep_pool_alloc((void**)&out.d_val, (nnz + 1) * 16); if (prc_) return prc_; }
    HIPCHK(hipMemcpy(out.d_levptr, levptr.data(), (size_t)(nlev + 1) * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(out.d_rowid, rowid.data(), (size_t)nr * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(out.d_rowptr, rowptr.data(), (size_t)(nr + 1) * 4, hipMemcpyHostToDevice));
    if (nnz) {
        HIPCHK(hipMemcpy(out.d_col, col.data(), nnz * 4, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(out.d_val, val.data(), nnz * 16, hipMemcpyHostToDevice));
    }
    if (upper) {
        { int prc_ = nep_pool_alloc((void**)&out.d_diag, (size_t)nr * 16); if (prc_) return prc_; }
        HIPCHK(hipMemcpy(out.d_diag, diag.data(), (size_t)nr * 16, hipMemcpyHostToDevice));
    }
    return NEP_OK;
}

static void free_mid(MidFactor& m) {
    if (m.d_rp) nep_pool_free(m.d_rp);
    if (m.d_ci) nep_pool_free(m.d_ci);
    if (m.d_vx) nep_pool_free(m.d_vx);
    if (m.d_inv) nep_pool_free(m.d_inv);
    m = MidFactor();
}

// blocked mid region rows [h0, h0 + nblk*b) of one factor: off-block CSR + device-built inverses of the diagonal blocks
static int build_mid(const int32_t* P, const int32_t* I, const nep_cdouble* X, bool upper, int64_t h0, int nblk, int b,
                     MidFactor& out, hipStream_t bst) {
    const int64_t M = (int64_t)nblk * b;
    std::vector<int32_t> rp(M + 1, 0), ci, blkoff(nblk + 1, 0), levptr, rowid(M), rowptr(M + 1, 0), col, lvl(b), cnt;
    std::vector<nep_cdouble> vx, val, diag(upper ? M : 0);
    {   // sizes first (the mid rows are the long ones: avoid vector regrowth)
        size_t noff = 0, nin = 0;
        for (int64_t r = 0; r < M; ++r) {
            const int64_t i = h0 + r, s = h0 + (r / b) * b, e = s + b;
            for (int32_t q = P[i]; q < P[i + 1]; ++q) {
                const int32_t j = I[q];
                if (j == i) continue;
                if (j >= s && j < e) ++nin; else ++noff;
            }
        }
        ci.reserve(noff); vx.reserve(noff); col.reserve(nin); val.reserve(nin);
    }
    out.wpr.assign(nblk, 1);
    for (int k = 0; k < nblk; ++k) {
        const int64_t s = h0 + (int64_t)k * b, e = s + b;
        // ---- off-block part, row order
        for (int64_t i = s; i < e; ++i) {
            for (int32_t q = P[i]; q < P[i + 1]; ++q) {
                const int32_t j = I[q];
                if (upper ? (j >= e) : (j < s)) { ci.push_back(j); vx.push_back(X[q]); }
            }
            rp[i - h0 + 1] = (int32_t)ci.size();
        }
        out.wpr[k] = pick_wpr((double)(rp[e - h0] - rp[s - h0]) / b);
        // ---- in-block levels
        int nlev = 0;
        if (!upper) {
            for (int64_t i = s; i < e; ++i) {
                int lv = 0;
                for (int32_t q = P[i]; q < P[i + 1]; ++q) { const int32_t j = I[q]; if (j >= s && j < i) lv = std::max(lv, lvl[j - s] + 1); }
                lvl[i - s] = lv; nlev = std::max(nlev, lv + 1);
            }
        } else {
            for (int64_t i = e - 1; i >= s; --i) {
                int lv = 0;
                for (int32_t q = P[i]; q < P[i + 1]; ++q) { const int32_t j = I[q]; if (j > i && j < e) lv = std::max(lv, lvl[j - s] + 1); }
                lvl[i - s] = lv; nlev = std::max(nlev, lv + 1);
            }
        }
        cnt.assign(nlev + 1, 0);
        for (int t = 0; t < b; ++t) cnt[lvl[t] + 1]++;
        for (int l = 0; l < nlev; ++l) cnt[l + 1] += cnt[l];
        blkoff[k] = (int32_t)levptr.size();
        const int32_t base = (int32_t)((int64_t)k * b);
        for (int l = 0; l <= nlev; ++l) levptr.push_back(base + cnt[l]);
        {
            std::vector<int32_t> pos(cnt.begin(), cnt.end() - 1);
            for (int t = 0; t < b; ++t) rowid[base + pos[lvl[t]]++] = (int32_t)(s + t);
        }
        for (int t = 0; t < b; ++t) {
            const int32_t slot = base + t, i = rowid[slot];
            bool have_diag = false;
            for (int32_t q = P[i]; q < P[i + 1]; ++q) {
                const int32_t j = I[q];
                if (j == i) { have_diag = true; if (upper) diag[slot] = X[q]; continue; }
                if (j >= s && j < e) { col.push_back(j); val.push_back(X[q]); }
            }
            if (upper && (!have_diag || (diag[slot].re == 0.0 && diag[slot].im == 0.0))) {
                nep_set_error("U has a zero pivot in row %d (matrix is singular)", i);
                return NEP_ERR_SINGULAR;
            }
            rowptr[slot + 1] = (int32_t)col.size();
        }
    }
    blkoff[nblk] = (int32_t)levptr.size();
    out.nnz = (int64_t)ci.size();
    // ---- upload the off-block CSR
    { int prc_ = nep_pool_alloc((void**)&out.d_rp, (size_t)(M + 1) * 4); if (prc_) return prc_; }
    { int prc_ = nep_pool_alloc((void**)&out.d_ci, (ci.size() + 1) * 4); if (prc_) return prc_; }
    { int prc_ = nep_pool_alloc((void**)&out.d_vx, (vx.size() + 1) * 16); if (prc_) return prc_; }
    HIPCHK(hipMemcpy(out.d_rp, rp.data(), (size_t)(M + 1) * 4, hipMemcpyHostToDevice));
    if (!ci.empty()) {
        HIPCHK(hipMemcpy(out.d_ci, ci.data(), ci.size() * 4, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(out.d_vx, vx.data(), vx.size() * 16, hipMemcpyHostToDevice));
    }
    // ---- diagonal-block inverses on the device
    int32_t *d_blkoff = nullptr, *d_levptr = nullptr, *d_rowid = nullptr, *d_rowptr = nullptr, *d_col = nullptr;
    cplx *d_val = nullptr, *d_diag = nullptr;
    int rc = nep_pool_alloc((void**)&out.d_inv, (size_t)M * b * sizeof(cplx));
    if (!rc) rc = nep_pool_alloc((void**)&d_blkoff, (size_t)(nblk + 1) * 4);
    if (!rc) rc = nep_pool_alloc((void**)&d_levptr, levptr.size() * 4);
    if (!rc) rc = nep_pool_alloc((void**)&d_rowid, (size_t)M * 4);
    if (!rc) rc = nep_pool_alloc((void**)&d_rowptr, (size_t)(M + 1) * 4);
    if (!rc) rc = nep_pool_alloc((void**)&d_col, (col.size() + 1) * 4);
    if (!rc) rc = nep_pool_alloc((void**)&d_val, (val.size() + 1) * 16);
    if (!rc && upper) rc = nep_pool_alloc((void**)&d_diag, (size_t)M * 16);
    hipError_t e = hipSuccess;
    if (!rc) {
        e = hipMemcpy(d_blkoff, blkoff.data(), (size_t)(nblk + 1) * 4, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(d_levptr, levptr.data(), levptr.size() * 4, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(d_rowid, rowid.data(), (size_t)M * 4, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(d_rowptr, rowptr.data(), (size_t)(M + 1) * 4, hipMemcpyHostToDevice);
        if (e == hipSuccess && !col.empty()) e = hipMemcpy(d_col, col.data(), col.size() * 4, hipMemcpyHostToDevice);
        if (e == hipSuccess && !val.empty()) e = hipMemcpy(d_val, val.data(), val.size() * 16, hipMemcpyHostToDevice);
        if (e == hipSuccess && upper) e = hipMemcpy(d_diag, diag.data(), (size_t)M * 16, hipMemcpyHostToDevice);
        if (e == hipSuccess) {
            if (upper)
                hipLaunchKernelGGL((k_mid_inverse<true>), dim3(b, nblk), dim3(256), (size_t)b * sizeof(cplx), bst, b, h0,
                                   (const int32_t*)d_blkoff, (const int32_t*)d_levptr, (const int32_t*)d_rowid,
                                   (const int32_t*)d_rowptr, (const int32_t*)d_col, (const cplx*)d_val,
                                   (const cplx*)d_diag, out.d_inv);
            else
                hipLaunchKernelGGL((k_mid_inverse<false>), dim3(b, nblk), dim3(256), (size_t)b * sizeof(cplx), bst, b, h0,
                                   (const int32_t*)d_blkoff, (const int32_t*)d_levptr, (const int32_t*)d_rowid,
                                   (const int32_t*)d_rowptr, (const int32_t*)d_col, (const cplx*)d_val,
                                   (const cplx*)d_diag, out.d_inv);
            e = hipGetLastError();
            if (e == hipSuccess) e = hipStreamSynchronize(bst);     // only this build's work: other threads' builds and solves go on
        }
        if (e != hipSuccess) { nep_set_error("mid inverse build failed: %s", hipGetErrorString(e)); rc = NEP_ERR_HIP; }
    }
    if (d_blkoff) nep_pool_free(d_blkoff);
    if (d_levptr) nep_pool_free(d_levptr);
    if (d_rowid) nep_pool_free(d_rowid);
    if (d_rowptr) nep_pool_free(d_rowptr);
    if (d_col) nep_pool_free(d_col);
    if (d_val) nep_pool_free(d_val);
    if (d_diag) nep_pool_free(d_diag);
    return rc;
}

template <bool UPPER>
static int run_mid(const nep_lu* lu, const MidFactor& m, cplx* work, int64_t ldw, cplx* tmp, int64_t ldt, int nrhs,
                   hipStream_t st, int* launches);

template <bool UPPER>
static int run_head(const TriFactor& f, cplx* work, int64_t ldw, int nrhs, hipStream_t st, int* launches) {
    for (const Seg& sg : f.segs) {
        if (sg.wide) {
            const int rows = sg.slot_hi - sg.slot_lo;
#define WIDE_CASE(G)                                                                                          \
    case G: {                                                                                                 \
        const int rpb = 256 / G;                                                                              \
        hipLaunchKernelGGL((k_level_wide<G, UPPER>), dim3((rows + rpb - 1) / rpb, nrhs), dim3(256), 0, st,    \
                           sg.slot_lo, sg.slot_hi, (const int32_t*)f.d_rowid, (const int32_t*)f.d_rowptr,     \
                           (const int32_t*)f.d_col, (const cplx*)f.d_val, (const cplx*)f.d_diag, work, ldw);  \
        break;                                                                                                \
    }
            switch (sg.G) { WIDE_CASE(8) WIDE_CASE(16) WIDE_CASE(64) }
#undef WIDE_CASE
        } else {
            hipLaunchKernelGGL((k_levels_narrow<UPPER>), dim3(1, nrhs), dim3(512), 0, st, sg.lev_lo, sg.lev_hi,
                               (const int32_t*)f.d_levptr, (const int32_t*)f.d_rowid, (const int32_t*)f.d_rowptr,
                               (const int32_t*)f.d_col, (const cplx*)f.d_val, (const cplx*)f.d_diag, work, ldw);
        }
        LAUNCHCHK();
        if (launches) ++*launches;
    }
    return NEP_OK;
}

template <bool UPPER>
static int run_mid(const nep_lu* lu, const MidFactor& m, cplx* work, int64_t ldw, cplx* tmp, int64_t ldt, int nrhs,
                   hipStream_t st, int* launches) {
    const int b = lu->bsz, nblk = lu->nblk;
    for (int q = 0; q < nblk; ++q) {
        const int k = UPPER ? nblk - 1 - q : q;
        const int64_t r0 = lu->h0 + (int64_t)k * b;
        launch_mid_spmv(m.wpr[k], r0, b, lu->h0, (const int32_t*)m.d_rp, (const int32_t*)m.d_ci, (const cplx*)m.d_vx,
                        (const cplx*)work, ldw, tmp, ldt, nrhs, st);
        LAUNCHCHK();
        const cplx* invk = (const cplx*)(m.d_inv + (int64_t)k * b * b);
#define MID_GEMV(WPR_, RB_)                                                                                          \
    hipLaunchKernelGGL((k_mid_gemv<UPPER, WPR_, RB_>), dim3(WPR_ == 4 ? b : (b + 3) / 4, (nrhs + RB_ - 1) / RB_),     \
                       dim3(256), 0, st, r0, b, lu->h0, invk, (const cplx*)tmp, ldt, work, ldw, nrhs)
        if (b >= 1024) { if (nrhs >= 8) MID_GEMV(4, 8); else if (nrhs >= 2) MID_GEMV(4, 4); else MID_GEMV(4, 1); }
        else           { if (nrhs >= 8) MID_GEMV(1, 8); else if (nrhs >= 2) MID_GEMV(1, 4); else MID_GEMV(1, 1); }
#undef MID_GEMV
        LAUNCHCHK();
        if (launches) *launches += 2;
    }
    return NEP_OK;
}

extern "C" {

int32_t nep_lu_destroy(nep_lu* lu) {
    if (!lu) return NEP_OK;
    if (lu->ml) { ml_destroy(lu->ml); lu->ml = nullptr; }
    // a solve enqueued on lu->last may still read these blocks (Beyn drops the factorisation right after the asynchronous
    // block solve): the pool hands them out again only behind that work
    hipStream_t st = lu->last; const bool fl = lu->used;
    auto rel_tri = [&](TriFactor& t) {
        nep_pool_free_on(t.d_levptr, st, fl); nep_pool_free_on(t.d_rowid, st, fl); nep_pool_free_on(t.d_rowptr, st, fl);
        nep_pool_free_on(t.d_col, st, fl); nep_pool_free_on(t.d_val, st, fl); nep_pool_free_on(t.d_diag, st, fl);
        t = TriFactor();
    };
    auto rel_mid = [&](MidFactor& m) {
        nep_pool_free_on(m.d_rp, st, fl); nep_pool_free_on(m.d_ci, st, fl); nep_pool_free_on(m.d_vx, st, fl);
        nep_pool_free_on(m.d_inv, st, fl);
        m = MidFactor();
    };
    rel_tri(lu->L11); rel_tri(lu->U11); rel_mid(lu->Lm); rel_mid(lu->Um);
    nep_pool_free_on(lu->d_L21p, st, fl); nep_pool_free_on(lu->d_L21i, st, fl); nep_pool_free_on(lu->d_L21x, st, fl);
    nep_pool_free_on(lu->d_Sinv, st, fl); nep_pool_free_on(lu->d_perm_r, st, fl); nep_pool_free_on(lu->d_perm_c, st, fl);
    if (lu->graph_exec) (void)hipGraphExecDestroy(lu->graph_exec);
    if (lu->cap_stream) (void)hipStreamDestroy(lu->cap_stream);
    if (lu->work.dptr) { nep_pool_free_on(lu->work.dptr, st, fl); lu->work.dptr = nullptr; lu->work.cap = 0; }
    delete lu;
    return NEP_OK;
}

static thread_local int g_expected_solves = 50;

static double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
#define TSTAMP(label) do { if (timing) { double t_ = now_ms(); fprintf(stderr, "[nep_lu_create] %-18s %8.3f ms\n", label, t_ - tlast); tlast = t_; } } while (0)

static int lu_build(nep_lu* lu, int64_t n, const int32_t* hLp, const int32_t* hLi, const nep_cdouble* hLx,
                    const int32_t* hUp, const int32_t* hUi, const nep_cdouble* hUx) {
    const bool timing = getenv("NEP_TIMING") != nullptr;
    double tlast = now_ms();
    // ---- validate triangularity and compute the full level structure of L (tail heuristic)
    for (int64_t i = 0; i < n; ++i) {
        for (int32_t e = hLp[i]; e < hLp[i + 1]; ++e) {
            if (hLi[e] < 0 || hLi[e] >= n) { nep_set_error("L: column out of range"); return NEP_ERR_ARG; }
            if (hLi[e] > i) { nep_set_error("L is not lower triangular (row %lld col %d)", (long long)i, hLi[e]); return NEP_ERR_ARG; }
        }
        for (int32_t e = hUp[i]; e < hUp[i + 1]; ++e) {
            if (hUi[e] < 0 || hUi[e] >= n) { nep_set_error("U: column out of range"); return NEP_ERR_ARG; }
            if (hUi[e] < i) { nep_set_error("U is not upper triangular (row %lld col %d)", (long long)i, hUi[e]); return NEP_ERR_ARG; }
        }
    }
    std::vector<int32_t> level(n, 0);
    int32_t nlevL = 0, nlevU = 0;
    compute_levels(n, hLp, hLi, false, 0, n, level, nlevL);
    lu->levL_full = nlevL;
    // tail start.  Seed: rows of the trailing run of "chain" levels (<= TRSV_TAIL_SMALL rows per level).  Then the
    // tail is grown while a cost model of one solve improves: every head level costs a dependent step (a launch
    // for a wide level, a workgroup barrier for a narrow one) while the dense tail costs 16 T^2 bytes of streaming.
    // Measured on gun (n=9956): T=1067 -> 0.76 ms, 2000 -> 0.42 ms, 2500 -> 0.25 ms per solve.
    int64_t i0 = n;
    {
        std::vector<int32_t> cnt(nlevL, 0);
        for (int64_t i = 0; i < n; ++i) cnt[level[i]]++;
        int t0 = nlevL;
        while (t0 > 0 && cnt[t0 - 1] <= TRSV_TAIL_SMALL) --t0;
        int64_t Tseed = 0;
        if (nlevL - t0 >= TRSV_TAIL_MIN) {
            for (int64_t i = 0; i < n; ++i) if (level[i] >= t0) { Tseed = n - i; break; }
        }
        if (Tseed > 0) {
            auto cost_us = [&](int64_t T) -> double {
                const int64_t h = n - T;
                std::vector<int32_t> lv(n, 0);
                double c = 16.0 * (double)T * (double)T / 3.0e6 + 10.0;       // dense GEMV at ~3 TB/s + tail SpMV
                for (int pass = 0; pass < 2; ++pass) {
                    int32_t nl = 0;
                    compute_levels(n, pass ? hUp : hLp, pass ? hUi : hLi, pass == 1, 0, h, lv, nl);
                    std::vector<int32_t> cn(nl, 0);
                    for (int64_t i = 0; i < h; ++i) cn[lv[i]]++;
                    bool in_narrow = false;
                    for (int32_t l = pass ? 0 : 1; l < nl; ++l) {
                        if (cn[l] >= TRSV_WIDE_MIN) { c += 4.5; in_narrow = false; }
                        else { c += 3.0; if (!in_narrow) { c += 5.0; in_narrow = true; } }
                    }
                }
                return c;
            };
            // one-off cost of building S22^{-1} on the device (T workgroups x 2T levels): ~T^2/256 us, amortised over
            // the number of solves the caller expects from this factorisation (nep_lu_set_expected_solves)
            const double nsolve = (double)std::max(1, g_expected_solves);
            auto total_us = [&](int64_t T) { return cost_us(T) + (double)T * (double)T / 256.0 / nsolve; };
            const int64_t Tcap = std::min<int64_t>(TRSV_TAIL_MAX, n / 2);
            int64_t best = std::min(Tseed, Tcap);
            double bestc = total_us(best);
            const double mult[] = {1.25, 1.5, 1.75, 2.0, 2.5, 3.0, 4.0};
            for (double m : mult) {
                int64_t T = std::min<int64_t>((int64_t)(Tseed * m), Tcap);
                if (T <= best) continue;
                const double c = total_us(T);
                if (c < bestc * 0.97) { bestc = c; best = T; }
            }
            i0 = n - best;
        }
        if (const char* e = getenv("NEP_LU_TAIL")) {   // experiment knob: force the tail size
            long v = atol(e);
            if (v >= 0 && v <= n && v <= TRSV_TAIL_MAX) i0 = n - v;
        }
    }
    {
        std::vector<int32_t> lv(n, 0);
        compute_levels(n, hUp, hUi, true, 0, n, lv, nlevU);
        lu->levU_full = nlevU;
    }
    lu->i0 = i0; lu->T = n - i0;
    const int64_t T = lu->T;
    // blocked mid region [h0, i0): nblk blocks of b rows.  A head level costs a dependent launch (~4.5 us), a block two
    // (~8 us + GEMV) plus a one-off inverse build (~b^2/2048 us, amortised over the expected solves).  level[] of the FULL
    // L is also the level of a row inside any leading sub-triangle, so head levels(h) = 1 + max(level[0:h)); U is
    // taken to behave alike (exact for the symmetric strategy).
    int64_t h0 = i0;
    {
        int b = n >= 8192 ? 2048 : (n >= 4096 ? 1024 : (n >= 2048 ? 512 : 256));   // measured: gun 0.35 -> 0.14 ms, n=1e6 203 -> 2.4 ms
        if (const char* e = getenv("NEP_LU_BLOCK")) { int v = atoi(e); if (v >= 64 && v <= 2048 && v % 64 == 0) b = v; }
        const int64_t kmax = std::min<int64_t>(i0 / b, std::min<int64_t>(n / 2, TRSV_MID_MAX) / b);
        if (kmax > 0) {
            std::vector<int32_t> headlev(kmax + 1, 0);
            int32_t run = 0;
            for (int64_t i = 0; i <= i0; ++i) {
                const int64_t d = i0 - i;
                if (d % b == 0 && d / b <= kmax) headlev[d / b] = run;
                if (i < i0) run = std::max(run, level[i] + 1);
            }
            const double nsolve = (double)std::max(1, g_expected_solves);
            const double cblk = 8.0 + 8.0 * (double)b * b / 3.0e6;   // two launches + the triangular GEMV at ~3 TB/s
            int64_t bestk = 0;
            double bestc = 2.0 * headlev[0] * 4.5;
            for (int64_t k = 1; k <= kmax; ++k) {
                const double c = 2.0 * (headlev[k] * 4.5 + k * cblk) + k * ((double)b * b / 2048.0) / nsolve;
                if (c < bestc * 0.98) { bestc = c; bestk = k; }
            }
            if (const char* e = getenv("NEP_LU_MID")) { long v = atol(e); if (v >= 0) bestk = std::min<int64_t>(v / b, kmax); }
            lu->nblk = (int32_t)bestk; lu->bsz = b;
            h0 = i0 - bestk * b;
        }
    }
    lu->h0 = h0;
    int rc;
    TSTAMP("validate+levels");
    // ---- heads and mid blocks: the L side and the U side are independent (host analysis, uploads, inverse kernels),
    // so the U side runs on a second host thread
    int32_t nl = 0;
    int rcU = NEP_OK;
    char errU[512] = {0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    // the inverse-block kernels of this build run on private non-blocking streams and are waited for individually (they
    // used to be followed by hipDeviceSynchronize, which serialised concurrent builds -- Beyn builds several
    // factorisations at once -- against each other and against the solves of the launching thread)
    hipStream_t bst = nullptr;     // per-build stream, destroyed at the end of the build (a thread_local one leaked per builder thread)
    HIPCHK(hipStreamCreateWithFlags(&bst, hipStreamNonBlocking));
    struct StreamGuard { hipStream_t s; ~StreamGuard() { if (s) (void)hipStreamDestroy(s); } } bst_guard{bst};
    std::thread side([&]() {
        (void)hipSetDevice(dev);
        hipStream_t sst = nullptr;
        if (hipStreamCreateWithFlags(&sst, hipStreamNonBlocking) != hipSuccess) sst = nullptr;
        std::vector<int32_t> levU(n, 0);
        int32_t nlU = 0;
        compute_levels(n, hUp, hUi, true, 0, h0, levU, nlU);
        rcU = build_tri(n, hUp, hUi, hUx, true, 0, h0, 0, n, levU, nlU, lu->U11);   // keeps the columns >= h0 (final by then)
        if (rcU == NEP_OK && lu->nblk > 0) rcU = build_mid(hUp, hUi, hUx, true, h0, lu->nblk, lu->bsz, lu->Um, sst);
        if (sst) (void)hipStreamDestroy(sst);
        if (rcU != NEP_OK) { strncpy(errU, nep_last_error(), sizeof(errU) - 1); }
    });
    compute_levels(n, hLp, hLi, false, 0, h0, level, nl);
    rc = build_tri(n, hLp, hLi, hLx, false, 0, h0, 0, h0, level, nl, lu->L11);
    if (rc == NEP_OK && lu->nblk > 0) rc = build_mid(hLp, hLi, hLx, false, h0, lu->nblk, lu->bsz, lu->Lm, bst);
    side.join();
    if (rc) return rc;
    if (rcU) { nep_set_error("%s", errU); return rcU; }
    TSTAMP("heads + mid blocks (L || U)");
    if (T == 0) return NEP_OK;
    // ---- L21 (tail rows, head columns)
    {
        std::vector<int32_t> rp(T + 1, 0), ci;
        std::vector<nep_cdouble> vx;
        for (int64_t r = 0; r < T; ++r) {
            const int64_t i = i0 + r;
            for (int32_t e = hLp[i]; e < hLp[i + 1]; ++e)
                if (hLi[e] < i0) { ci.push_back(hLi[e]); vx.push_back(hLx[e]); }
            rp[r + 1] = (int32_t)ci.size();
        }
        lu->nnzL21 = (int64_t)ci.size();
        { int prc_ = nep_pool_alloc((void**)&lu->d_L21p, (size_t)(T + 1) * 4); if (prc_) return prc_; }
        { int prc_ = nep_pool_alloc((void**)&lu->d_L21i, (ci.size() + 1) * 4); if (prc_) return prc_; }
        { int prc_ = nep_pool_alloc((void**)&lu->d_L21x, (ci.size() + 1) * 16); if (prc_) return prc_; }
        HIPCHK(hipMemcpy(lu->d_L21p, rp.data(), (size_t)(T + 1) * 4, hipMemcpyHostToDevice));
        if (!ci.empty()) {
            HIPCHK(hipMemcpy(lu->d_L21i, ci.data(), ci.size() * 4, hipMemcpyHostToDevice));
            HIPCHK(hipMemcpy(lu->d_L21x, vx.data(), vx.size() * 16, hipMemcpyHostToDevice));
        }
    }
    TSTAMP("L21");
    // ---- S22^{-1} on the device
    TriFactor L22, U22;
    compute_levels(n, hLp, hLi, false, i0, n, level, nl);
    rc = build_tri(n, hLp, hLi, hLx, false, i0, n, i0, n, level, nl, L22);
    if (rc == NEP_OK) {
        compute_levels(n, hUp, hUi, true, i0, n, level, nl);
        rc = build_tri(n, hUp, hUi, hUx, true, i0, n, i0, n, level, nl, U22);
    }
    if (rc == NEP_OK) {
        rc = nep_pool_alloc((void**)&lu->d_Sinv, (size_t)T * T * sizeof(cplx));
    }
    TSTAMP("tail factors");
    if (rc == NEP_OK) {
        hipLaunchKernelGGL(k_tail_inverse, dim3((unsigned)T), dim3(512), (size_t)T * sizeof(cplx), bst, (int)T, (int)i0,
                           (const int32_t*)L22.d_levptr, L22.nlev, (const int32_t*)L22.d_rowid,
                           (const int32_t*)L22.d_rowptr, (const int32_t*)L22.d_col, (const cplx*)L22.d_val,
                           (const int32_t*)U22.d_levptr, U22.nlev, (const int32_t*)U22.d_rowid,
                           (const int32_t*)U22.d_rowptr, (const int32_t*)U22.d_col, (const cplx*)U22.d_val,
                           (const cplx*)U22.d_diag, lu->d_Sinv);
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(bst);
        if (e != hipSuccess) { nep_set_error("tail inverse kernel failed: %s", hipGetErrorString(e)); rc = NEP_ERR_HIP; }
    }
    TSTAMP("tail inverse");
    free_tri(L22);
    free_tri(U22);
    TSTAMP("free tail factors");
    return rc;
}

int32_t nep_lu_set_expected_solves(int32_t nsolves) {
    g_expected_solves = nsolves < 1 ? 1 : nsolves;
    return NEP_OK;
}

static bool use_block_schedule() {
    const char* e = getenv("NEP_LU_SCHED");
    return !(e && (!strcmp(e, "old") || !strcmp(e, "levels")));
}

static int lu_create_levels(int64_t n, const int32_t* hLp, const int32_t* hLi, const nep_cdouble* hLx, const int32_t* hUp,
                            const int32_t* hUi, const nep_cdouble* hUx, const int32_t* h_perm_r, const int32_t* h_perm_c,
                            nep_lu* lu) {
    const bool timing = getenv("NEP_TIMING") != nullptr;
    double tlast = now_ms();
    // a graph pays off from the second solve on; one-shot factorisations (BackslashLinSolver: Beyn builds them on
    // worker threads while the main thread solves) launch eagerly -- stream capture in one thread makes synchronous
    // HIP calls of the other threads fail on this runtime
    lu->use_graph = g_expected_solves >= 3 ? 1 : 0;
    int rc = lu_build(lu, n, hLp, hLi, hLx, hUp, hUi, hUx);
    TSTAMP("lu_build total");
    if (rc != NEP_OK) return rc;
    auto up_perm = [&](const int32_t* hp, int32_t** dp) -> int {
        if (!hp) return NEP_OK;
        std::vector<char> seen(n, 0);
        for (int64_t i = 0; i < n; ++i) {
            if (hp[i] < 0 || hp[i] >= n || seen[hp[i]]) { nep_set_error("invalid permutation"); return NEP_ERR_ARG; }
            seen[hp[i]] = 1;
        }
        { int prc_ = nep_pool_alloc((void**)dp, (size_t)n * 4); if (prc_) return prc_; }
        HIPCHK(hipMemcpy(*dp, hp, (size_t)n * 4, hipMemcpyHostToDevice));
        return NEP_OK;
    };
    rc = up_perm(h_perm_r, &lu->d_perm_r);
    if (rc == NEP_OK) rc = up_perm(h_perm_c, &lu->d_perm_c);
    TSTAMP("perms");
    return rc;
}

// CSC -> CSR for the level schedule (the block schedule reads CSC directly)
static void csc_to_csr_c(int64_t n, const int32_t* cp, const int32_t* ri, const nep_cdouble* vx, std::vector<int32_t>& rp,
                         std::vector<int32_t>& ci, std::vector<nep_cdouble>& rv) {
    const int64_t nnz = cp[n];
    rp.assign(n + 1, 0);
    for (int64_t e = 0; e < nnz; ++e) rp[ri[e] + 1]++;
    for (int64_t i = 0; i < n; ++i) rp[i + 1] += rp[i];
    ci.resize(nnz); rv.resize(nnz);
    std::vector<int32_t> pos(rp.begin(), rp.end() - 1);
    for (int64_t c = 0; c < n; ++c)
        for (int32_t e = cp[c]; e < cp[c + 1]; ++e) { const int32_t q = pos[ri[e]]++; ci[q] = (int32_t)c; rv[q] = vx[e]; }
}

static int32_t lu_create_any(int64_t n, int csc, const int32_t* hLp, const int32_t* hLi, const nep_cdouble* hLx,
                             const int32_t* hUp, const int32_t* hUi, const nep_cdouble* hUx, const int32_t* h_perm_r,
                             const int32_t* h_perm_c, nep_lu** out) {
    ARGCHK(out != nullptr);
    *out = nullptr;
    ARGCHK(n > 0 && n < ((int64_t)1 << 31));
    ARGCHK(hLp && hLi && hLx && hUp && hUi && hUx);
    ARGCHK(hLp[0] == 0 && hUp[0] == 0 && hLp[n] >= 0 && hUp[n] >= 0);
    nep_lu* lu = new nep_lu();
    lu->n = n; lu->csc = csc;
    lu->nnzL_in = hLp[n]; lu->nnzU_in = hUp[n];
    int rc = NEP_ERR_UNSUPPORTED;
    if (use_block_schedule()) {
        if (csc) {      // CSC index sanity before the pattern is transposed
            for (int64_t e = 0; e < hLp[n]; ++e) if (hLi[e] < 0 || hLi[e] >= n) { delete lu; nep_set_error("L: row out of range"); return NEP_ERR_ARG; }
            for (int64_t e = 0; e < hUp[n]; ++e) if (hUi[e] < 0 || hUi[e] >= n) { delete lu; nep_set_error("U: row out of range"); return NEP_ERR_ARG; }
        }
        rc = ml_create(n, csc, hLp, hLi, hLx, hUp, hUi, hUx, h_perm_r, h_perm_c, g_expected_solves, &lu->ml);
    }
    if (rc == NEP_ERR_UNSUPPORTED) {
        if (csc) {
            std::vector<int32_t> Lrp, Lci, Urp, Uci; std::vector<nep_cdouble> Lrv, Urv;
            csc_to_csr_c(n, hLp, hLi, hLx, Lrp, Lci, Lrv);
            csc_to_csr_c(n, hUp, hUi, hUx, Urp, Uci, Urv);
            rc = lu_create_levels(n, Lrp.data(), Lci.data(), Lrv.data(), Urp.data(), Uci.data(), Urv.data(), h_perm_r, h_perm_c, lu);
        } else
            rc = lu_create_levels(n, hLp, hLi, hLx, hUp, hUi, hUx, h_perm_r, h_perm_c, lu);
    }
    if (rc != NEP_OK) { nep_lu_destroy(lu); return rc; }
    *out = lu;
    return NEP_OK;
}

int32_t nep_lu_create(int64_t n, const int32_t* hLp, const int32_t* hLi, const nep_cdouble* hLx, const int32_t* hUp,
                      const int32_t* hUi, const nep_cdouble* hUx, const int32_t* h_perm_r, const int32_t* h_perm_c,
                      nep_lu** out) {
    return lu_create_any(n, 0, hLp, hLi, hLx, hUp, hUi, hUx, h_perm_r, h_perm_c, out);
}

int32_t nep_lu_create_csc(int64_t n, const int32_t* hLp, const int32_t* hLi, const nep_cdouble* hLx, const int32_t* hUp,
                          const int32_t* hUi, const nep_cdouble* hUx, const int32_t* h_perm_r, const int32_t* h_perm_c,
                          nep_lu** out) {
    return lu_create_any(n, 1, hLp, hLi, hLx, hUp, hUi, hUx, h_perm_r, h_perm_c, out);
}

}  // extern "C"
// hooks of csrc/lufac.hip (device-side numeric factorisation)
MLFactor* nep_lu_ml(nep_lu* lu) { return lu ? lu->ml : nullptr; }
nep_lu* nep_lu_wrap_ml(MLFactor* F, int64_t n, int64_t nnzL, int64_t nnzU) {
    nep_lu* lu = new nep_lu();
    lu->n = n; lu->csc = 1; lu->nnzL_in = nnzL; lu->nnzU_in = nnzU; lu->ml = F;
    return lu;
}
extern "C" {

int32_t nep_lu_refactor(nep_lu* lu, const nep_cdouble* hLx, const nep_cdouble* hUx) {
    ARGCHK(lu && hLx && hUx);
    if (!lu->ml) { nep_set_error("nep_lu_refactor needs the block schedule (this handle uses the level schedule)"); return NEP_ERR_UNSUPPORTED; }
    return ml_refactor(lu->ml, hLx, hUx);
}

int32_t nep_lu_set_row_scale(nep_lu* lu, const double* h_rs) {
    ARGCHK(lu != nullptr);
    if (!lu->ml) { nep_set_error("nep_lu_set_row_scale needs the block schedule"); return NEP_ERR_UNSUPPORTED; }
    return ml_set_row_scale(lu->ml, h_rs);
}

int32_t nep_lu_analyze(int64_t n, int32_t csc, const int32_t* hLp, const int32_t* hLi, const int32_t* hUp, const int32_t* hUi,
                       int64_t out[8]) {
    ARGCHK(n > 0 && n < ((int64_t)1 << 31) && hLp && hLi && hUp && hUi && out);
    ARGCHK(hLp[0] == 0 && hUp[0] == 0);
    for (int64_t e = 0; e < hLp[n]; ++e) ARGCHK(hLi[e] >= 0 && hLi[e] < n);
    for (int64_t e = 0; e < hUp[n]; ++e) ARGCHK(hUi[e] >= 0 && hUi[e] < n);
    return ml_analyze(n, csc, hLp, hLi, hUp, hUi, out);
}

int32_t nep_lu_is_block_schedule(const nep_lu* lu, int32_t* out) {
    ARGCHK(lu && out);
    *out = lu->ml ? 1 : 0;
    return NEP_OK;
}

int32_t nep_lu_info(const nep_lu* lu, int64_t info[6]) {
    ARGCHK(lu && info);
    if (lu->ml) { ml_info(lu->ml, info, nullptr); return NEP_OK; }
    info[0] = lu->n; info[1] = lu->nnzL_in; info[2] = lu->nnzU_in;
    // levels actually traversed per solve (head levels; the dense tail counts as one step each way)
    info[3] = lu->L11.nlev + lu->nblk + (lu->T ? 1 : 0); info[4] = lu->U11.nlev + lu->nblk + (lu->T ? 1 : 0);
    // algorithmic bytes of one single-RHS solve: sparse heads + L21 (val 16 + idx 4) + dense tail + vectors
    info[5] = (lu->L11.nnz + lu->U11.nnz + lu->nnzL21 + lu->Lm.nnz + lu->Um.nnz) * 20 + 16 * lu->T * lu->T +
              16 * (int64_t)lu->nblk * lu->bsz * lu->bsz /* two triangular halves */ + 8 * (lu->n + 1) + 16 * lu->n +
              3 * 16 * lu->n;
    return NEP_OK;
}

/* extra introspection used by tests/bench: out[0]=tail size T, out[1]=kernel launches of the last
 * solve, out[2]=full levels(L), out[3]=full levels(U), out[4]=wide segments, out[5]=narrow segments,
 * out[6]=rows of the blocked mid region, out[7]=its block size */
int32_t nep_lu_schedule(const nep_lu* lu, int64_t out[8]) {
    ARGCHK(lu && out);
    if (lu->ml) { ml_info(lu->ml, nullptr, out); return NEP_OK; }
    out[0] = lu->T; out[1] = lu->launches; out[2] = lu->levL_full; out[3] = lu->levU_full;
    int64_t w = 0, nn = 0;
    for (const Seg& s : lu->L11.segs) (s.wide ? w : nn)++;
    for (const Seg& s : lu->U11.segs) (s.wide ? w : nn)++;
    out[4] = w; out[5] = nn;
    out[6] = (int64_t)lu->nblk * lu->bsz; out[7] = lu->bsz;
    return NEP_OK;
}

// enqueues the level sweep (L head, tail, U head) on `st`
static int lu_sweep(nep_lu* lu, int nrhs, cplx* work, cplx* tmp, hipStream_t st, int* launches) {
    const int64_t n = lu->n, T = lu->T, i0 = lu->i0, ldt = n - lu->h0;
    cplx* tmp_tail = tmp + (i0 - lu->h0);
    int rc = run_head<false>(lu->L11, work, n, nrhs, st, launches);
    if (rc) return rc;
    if (lu->nblk > 0) { rc = run_mid<false>(lu, lu->Lm, work, n, tmp, ldt, nrhs, st, launches); if (rc) return rc; }
    if (T > 0) {
        launch_mid_spmv(pick_wpr((double)lu->nnzL21 / (double)T), i0, (int)T, i0, (const int32_t*)lu->d_L21p,
                        (const int32_t*)lu->d_L21i, (const cplx*)lu->d_L21x, (const cplx*)work, n, tmp_tail, ldt, nrhs, st);
        LAUNCHCHK();
#define TAIL_GEMV(RB_)                                                                                              \
    hipLaunchKernelGGL((k_tail_gemv<RB_>), dim3((unsigned)((T + 3) / 4), (nrhs + RB_ - 1) / RB_), dim3(256), 0, st, T, \
                       i0, (const cplx*)lu->d_Sinv, (const cplx*)tmp_tail, ldt, work, n, nrhs)
        if (nrhs >= 8) TAIL_GEMV(8); else if (nrhs >= 2) TAIL_GEMV(4); else TAIL_GEMV(1);
#undef TAIL_GEMV
        LAUNCHCHK();
        if (launches) *launches += 2;
    }
    if (lu->nblk > 0) { rc = run_mid<true>(lu, lu->Um, work, n, tmp, ldt, nrhs, st, launches); if (rc) return rc; }
    return run_head<true>(lu->U11, work, n, nrhs, st, launches);
}

int32_t nep_lu_solve(nep_lu* lu, int32_t nrhs, const nep_cdouble* dB, int64_t ldb, nep_cdouble* dX, int64_t ldx,
                     double scale, nep_stream stream) {
    return nep_lu_solve_add(lu, nrhs, dB, ldb, nullptr, 0, dX, ldx, scale, stream);
}

int32_t nep_lu_solve_add(nep_lu* lu, int32_t nrhs, const nep_cdouble* dB, int64_t ldb, const nep_cdouble* dAdd,
                         int64_t ldadd, nep_cdouble* dX, int64_t ldx, double scale, nep_stream stream) {
    ARGCHK(lu && dB && dX);
    ARGCHK(nrhs >= 1 && nrhs <= 65535 && ldb >= lu->n && ldx >= lu->n);
    ARGCHK(dAdd == nullptr || ldadd >= lu->n);
    hipStream_t st = as_stream(stream);
    if (lu->ml) return ml_solve(lu->ml, nrhs, dB, ldb, dAdd, ldadd, dX, ldx, scale, st);
    lu->last = st; lu->used = true;
    const int64_t n = lu->n, T = lu->T;
    int rc = lu->work.ensure((size_t)(2 * n - lu->h0) * nrhs * sizeof(cplx));
    if (rc) return rc;
    cplx* work = (cplx*)lu->work.dptr;
    cplx* tmp = work + (size_t)n * nrhs;
    int launches = 0;
    const int pg = (int)std::min<int64_t>((n + 255) / 256, 1024);
    hipLaunchKernelGGL(k_perm_in, dim3(pg, nrhs), dim3(256), 0, st, n, (const int32_t*)lu->d_perm_r, (const cplx*)dB, ldb,
                       work, n);
    LAUNCHCHK(); ++launches;
    const int nseg = (int)(lu->L11.segs.size() + lu->U11.segs.size()) + 4 * lu->nblk;
    bool graphed = false;
    if (lu->use_graph && nseg >= 8 && !getenv("NEP_NO_GRAPH")) {
        if (!lu->graph_exec || lu->graph_nrhs != nrhs || lu->graph_work != (void*)work) {
            // (re)capture: same kernels, recorded on a private stream in thread-local capture mode
            if (lu->graph_exec) { (void)hipGraphExecDestroy(lu->graph_exec); lu->graph_exec = nullptr; }
            if (!lu->cap_stream) HIPCHK(hipStreamCreateWithFlags(&lu->cap_stream, hipStreamNonBlocking));
            hipGraph_t g = nullptr;
            hipError_t e = hipStreamBeginCapture(lu->cap_stream, hipStreamCaptureModeThreadLocal);
            if (e == hipSuccess) {
                int rcs = lu_sweep(lu, nrhs, work, tmp, lu->cap_stream, nullptr);
                e = hipStreamEndCapture(lu->cap_stream, &g);
                if (rcs == NEP_OK && e == hipSuccess && g) e = hipGraphInstantiate(&lu->graph_exec, g, nullptr, nullptr, 0);
                else if (e == hipSuccess) e = hipErrorUnknown;
                if (g) (void)hipGraphDestroy(g);
            }
            if (e != hipSuccess || !lu->graph_exec) {
                (void)hipGetLastError();
                lu->graph_exec = nullptr;
                lu->use_graph = 0;              // fall back to eager launches for this factorisation
            } else {
                lu->graph_nrhs = nrhs; lu->graph_work = (void*)work;
            }
        }
        if (lu->graph_exec) {
            HIPCHK(hipGraphLaunch(lu->graph_exec, st));
            launches += nseg + (T > 0 ? 2 : 0);
            graphed = true;
        }
    }
    if (!graphed) {
        rc = lu_sweep(lu, nrhs, work, tmp, st, &launches);
        if (rc) return rc;
    }
    hipLaunchKernelGGL(k_perm_out, dim3(pg, nrhs), dim3(256), 0, st, n, (const int32_t*)lu->d_perm_c, (const cplx*)work, n,
                       (cplx*)dX, ldx, scale, (const cplx*)dAdd, ldadd);
    LAUNCHCHK(); ++launches;
    lu->launches = launches;
    return NEP_OK;
}

}  // extern "C"
