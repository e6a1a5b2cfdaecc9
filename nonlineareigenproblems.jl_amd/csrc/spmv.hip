// libnepmi355: SPMF object (stacked CSR) and the kernels K1 (compute_Mlincomb) and
// K2 (batched residuals / compute_MM SpMM) for gfx950.
//
// K1  z = sum_i A_i (V c_i) is executed as
//   (a) k_vc:   WT[r, i] = sum_j V[r,j] C[j,i]      (tall-skinny, streams V once, HBM-bound)
//   (b) k_spmv: z[r] = sum_{e in row r} val[e] * WT[col(e), term(e)]   (stacked CSR, G lanes/row)
// The reference streams V once PER TERM (src/NEPTypes.jl:1006) and runs one CSC SpMV per term
// (:1007); the DerSPMF formulation (:1154-1157) is the one realised here.
#include "common.h"
#include <cstring>
#include <vector>
#include <algorithm>

struct nep_spmf {
    int64_t n = 0;
    int32_t mt = 0;
    int64_t nnz = 0;
    int32_t valbytes = 8;  // 8: all terms real, 16: complex values
    int32_t lanes = 16;    // lanes per row in k_spmv
    int32_t* d_rowptr = nullptr;
    uint32_t* d_idx = nullptr;
    void* d_vals = nullptr;
    cplx* d_WT = nullptr;     // n x mt row-major workspace
    // SELL-64 copy of the stacked matrix (large n only): lane = row, entries of 64 consecutive rows interleaved
    int32_t* d_sell_ptr = nullptr;   // nslices+1, in units of 64-entry columns
    uint32_t* d_sell_idx = nullptr;
    void* d_sell_val = nullptr;
    int64_t sell_cols = 0;           // padded entries / 64
    NepTiles* tiles = nullptr;       // footprint tiles (spmv_tile.hip): K1 in one launch, no W round trip
    NepScratch coef;          // staged coefficient matrices
    NepScratch part;          // per-block partials
    NepScratch cwpart;        // nep_cw_backward_error's own scratch: it runs on the solve stream while a residual batch
                              // (coef, part) may be in flight on another stream (iar's convergence checks)
    PinnedRing ring;          // pinned staging of host coefficient blocks
    PinnedRing cwring;        // ... of nep_cw_backward_error (may be called from another host thread than the residual batches)
};

// ------------------------------------------------------------------------------------------
// (a) WT[r, i0+i] = sum_j V[r + j*ldv] * C[j + (i0+i)*ldc],  i < MT.
// block = 512 threads = 8 waves.  ROWS = 64: a wave owns 64 rows (1 KiB coalesced per load) and one of 8
// column groups -- the streaming shape for large n.  ROWS = 32: each half-wave owns the same 32 rows and one
// of 16 column groups, which doubles the number of workgroups for small n (gun: 312 blocks on 256 CUs).
template <int MT, int ROWS, bool SHIFT>
__global__ __launch_bounds__(512) void k_vc(const cplx* __restrict__ V, int64_t ldv, int64_t n, int k,
                                            const cplx* __restrict__ C, int64_t ldc, int i0, int mt_total,
                                            cplx* __restrict__ WT, cplx* __restrict__ shift_dst) {
    constexpr int NG = 512 / ROWS;           // column groups
    constexpr int KC = 128;                  // coefficient rows staged in LDS per chunk
    __shared__ cplx sm[NG][MT][ROWS];
    __shared__ cplx cs[MT][KC];
    const int rr0 = threadIdx.x % ROWS;
    const int g = threadIdx.x / ROWS;
    const int64_t row = blockIdx.x * (int64_t)ROWS + rr0;
    const int64_t rowc = row < n ? row : n - 1;
    cplx acc[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) acc[i] = cmake(0.0, 0.0);
    const cplx* vp = V + rowc;
    for (int j0 = 0; j0 < k; j0 += KC) {
        const int kc = min(KC, k - j0);
        if (j0 > 0) __syncthreads();
        for (int t = threadIdx.x; t < kc * MT; t += 512) {
            const int i = t / kc, j = t % kc;
            cs[i][j] = C[j0 + j + (int64_t)(i0 + i) * ldc];
        }
        __syncthreads();
#pragma unroll 4
        for (int j = g; j < kc; j += NG) {
            const cplx v = vp[(int64_t)(j0 + j) * ldv];
            // iar: the block shift of the basis column rides along (dst block j+1 = src block j / (j+1), method_iar.jl:97-98)
            if (SHIFT && row < n) {
                const double sc = 1.0 / (double)(j0 + j + 1);
                shift_dst[row + (int64_t)(j0 + j) * ldv] = cmake(v.x * sc, v.y * sc);
            }
#pragma unroll
            for (int i = 0; i < MT; ++i) cfma(acc[i], v, cs[i][j]);
        }
    }
#pragma unroll
    for (int i = 0; i < MT; ++i) sm[g][i][rr0] = acc[i];
    __syncthreads();
    // ROWS*MT outputs, written contiguously: t -> (row = t / MT, i = t % MT)
    for (int t = threadIdx.x; t < ROWS * MT; t += 512) {
        const int rr = t / MT, i = t % MT;
        cplx s = sm[0][i][rr];
#pragma unroll
        for (int q = 1; q < NG; ++q) s = cadd(s, sm[q][i][rr]);
        const int64_t r = blockIdx.x * (int64_t)ROWS + rr;
        if (r < n) WT[r * mt_total + i0 + i] = s;
    }
}

// (b) stacked-CSR SpMV, G lanes per row
template <int G, typename VT>
__global__ __launch_bounds__(256) void k_spmv(const int32_t* __restrict__ rowptr,
                                              const uint32_t* __restrict__ idx,
                                              const VT* __restrict__ vals, const cplx* __restrict__ WT,
                                              int mt, int64_t n, cplx* __restrict__ z) {
    constexpr int RPB = 256 / G;
    const int sub = threadIdx.x % G;
    const int64_t row = blockIdx.x * (int64_t)RPB + threadIdx.x / G;
    cplx acc = cmake(0.0, 0.0);
    if (row < n) {
        // four entries per lane and trip: index / value loads together, then the four gathers, then the arithmetic (the compiler
        // waits for an entry's gather before it issues the next entry's loads: two dependent round trips per ENTRY otherwise)
        const int e0 = rowptr[row], e1 = rowptr[row + 1];
        for (int e = e0 + sub; e < e1; e += 4 * G) {
            uint32_t id[4]; VT v[4]; cplx wv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int ee = e + u * G < e1 ? e + u * G : e;
                id[u] = idx[ee]; v[u] = vals[ee];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) wv[u] = WT[(int64_t)(id[u] & NEP_COL_MASK) * mt + (id[u] >> NEP_TERM_SHIFT)];
#pragma unroll
            for (int u = 0; u < 4; ++u) if (e + u * G < e1) cfma(acc, v[u], wv[u]);
        }
    }
    acc = group_reduce_sum<G>(acc);
    if (row < n && sub == 0) z[row] = acc;
}

typedef double d2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double ntload(const double* p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ cplx ntload(const cplx* p) {
    const d2v v = __builtin_nontemporal_load((const d2v*)p);
    return cmake(v.x, v.y);
}

// Workgroups are dealt round-robin to the 8 XCDs (private L2 each).  Remapped so that XCD x streams one contiguous range of
// slices: the stencil neighbours of a row (+-1, +-nz rows away) then sit in the same L2 instead of being fetched by
// several XCDs (measured at n = 1e6, k = 8 fused: see DESIGN.md K1).
__device__ __forceinline__ int64_t xcd_block(int swz) {
    const int64_t b = blockIdx.x, nb = gridDim.x;
    if (!swz) return b;
    const int64_t x = b & 7, i = b >> 3, per = nb >> 3, rem = nb & 7;
    return x * per + (x < rem ? x : rem) + i;
}

// (b') SELL-64 SpMV for large n: one lane per row, the entries of 64 consecutive rows are interleaved so that
// every load of a wave is one contiguous 256-byte (idx) / 512-byte (val) segment; no cross-lane reduction.
// FOLD: k == 1 -- the vector is used directly and the per-term coefficient is applied on the fly
// (z = sum_e val[e] * C[term(e)] * v[col(e)]), which saves the k_vc launch and the W round trip.
template <typename VT, bool FOLD>
__global__ __launch_bounds__(256) void k_spmv_sell(const int32_t* __restrict__ sptr, const uint32_t* __restrict__ idx,
                                                   const VT* __restrict__ vals, const cplx* __restrict__ X,
                                                   const cplx* __restrict__ C, int64_t ldc, int mt, int64_t n,
                                                   cplx* __restrict__ z, int swz) {
    const int lane = threadIdx.x & 63;
    const int64_t slice = xcd_block(swz) * 4LL + (threadIdx.x >> 6);
    const int64_t row = slice * 64 + lane;
    if (slice * 64 >= n) return;
    cplx acc = cmake(0.0, 0.0);
    const int64_t e0 = sptr[slice], e1 = sptr[slice + 1];
    const uint32_t* ip = idx + e0 * 64 + lane;
    const VT* vp = vals + e0 * 64 + lane;
    const int cnt = (int)(e1 - e0);
#pragma unroll 4
    for (int e = 0; e < cnt; ++e) {
        const uint32_t id = __builtin_nontemporal_load(ip + (int64_t)e * 64);
        const VT a = ntload(vp + (int64_t)e * 64);
        const int64_t c = id & NEP_COL_MASK;
        const int t = id >> NEP_TERM_SHIFT;
        if (FOLD) {
            cfma(acc, cscale(a, C[(int64_t)t * ldc]), X[c]);
        } else {
            cfma(acc, a, X[c * mt + t]);
        }
    }
    if (row < n) z[row] = acc;
}

// (b'') 2 <= k <= NEP_K1_FUSE_MAX: the coefficient product is folded into the SpMV -- for every entry the k-term inner
// product w = sum_j C[j,term] V[col,j] is formed on the fly (the k loads of a wave are k contiguous 1 KiB segments, and the
// stencil neighbours of a row hit in L1/L2), so neither the k_vc launch nor the n x mt round trip of W through HBM happens:
// the kernel's HBM traffic is the algorithmic minimum (matrix + V + z).  C is staged in LDS as cs[term][j].
__device__ __forceinline__ cplx kdot(const cplx* __restrict__ vp, int64_t ldv, const cplx* cp, int k) {
    cplx w0 = cmake(0.0, 0.0), w1 = cmake(0.0, 0.0);
    int j = 0;
    for (; j + 4 <= k; j += 4) {
        const cplx v0 = vp[(int64_t)j * ldv], v1 = vp[(int64_t)(j + 1) * ldv];
        const cplx v2 = vp[(int64_t)(j + 2) * ldv], v3 = vp[(int64_t)(j + 3) * ldv];
        cfma(w0, cp[j], v0); cfma(w1, cp[j + 1], v1); cfma(w0, cp[j + 2], v2); cfma(w1, cp[j + 3], v3);
    }
    for (; j < k; ++j) cfma(w0, cp[j], vp[(int64_t)j * ldv]);
    return cadd(w0, w1);
}
__device__ __forceinline__ void stage_coef(cplx* cs, const cplx* __restrict__ C, int64_t ldc, int k, int mt) {
    for (int i = threadIdx.x; i < mt * k; i += blockDim.x) cs[i] = C[(i % k) + (int64_t)(i / k) * ldc];
    __syncthreads();
}

// K is a compile-time constant so that the K gathers of an entry (and of the next, unroll 2) are all in flight together:
// with a run-time k the compiler waits after every few loads and the kernel is latency-bound (measured 122 us vs the
// two-launch form's 64 us at n = 1e6, k = 8).
template <typename VT, int K>
__global__ __launch_bounds__(256) void k_spmv_sell_kfused(const int32_t* __restrict__ sptr, const uint32_t* __restrict__ idx,
                                                          const VT* __restrict__ vals, const cplx* __restrict__ V,
                                                          int64_t ldv, const cplx* __restrict__ C, int64_t ldc,
                                                          int mt, int64_t n, cplx* __restrict__ z, int swz) {
    extern __shared__ cplx cs[];
    stage_coef(cs, C, ldc, K, mt);
    const int lane = threadIdx.x & 63;
    const int64_t slice = xcd_block(swz) * 4LL + (threadIdx.x >> 6);
    const int64_t row = slice * 64 + lane;
    if (slice * 64 >= n) return;
    cplx acc = cmake(0.0, 0.0);
    const int64_t e0 = sptr[slice], e1 = sptr[slice + 1];
    const uint32_t* ip = idx + e0 * 64 + lane;
    const VT* vp = vals + e0 * 64 + lane;
    const int cnt = (int)(e1 - e0);
#pragma unroll 2
    for (int e = 0; e < cnt; ++e) {
        const uint32_t id = __builtin_nontemporal_load(ip + (int64_t)e * 64);
        const VT a = ntload(vp + (int64_t)e * 64);
        const cplx* xp = V + (id & NEP_COL_MASK);
        const cplx* cp = cs + (id >> NEP_TERM_SHIFT) * K;
        cplx v[K];
#pragma unroll
        for (int j = 0; j < K; ++j) v[j] = xp[(int64_t)j * ldv];
        cplx w0 = cmake(0.0, 0.0), w1 = cmake(0.0, 0.0);
#pragma unroll
        for (int j = 0; j < K; ++j) {
            if (j & 1) cfma(w1, cp[j], v[j]); else cfma(w0, cp[j], v[j]);
        }
        cfma(acc, a, cadd(w0, w1));
    }
    if (row < n) z[row] = acc;
}

template <int G, typename VT>
__global__ __launch_bounds__(256) void k_spmv_kfused(const int32_t* __restrict__ rowptr, const uint32_t* __restrict__ idx,
                                                     const VT* __restrict__ vals, const cplx* __restrict__ V, int64_t ldv,
                                                     int k, const cplx* __restrict__ C, int64_t ldc, int mt, int64_t n,
                                                     cplx* __restrict__ z) {
    extern __shared__ cplx cs[];
    stage_coef(cs, C, ldc, k, mt);
    constexpr int RPB = 256 / G;
    const int sub = threadIdx.x % G;
    const int64_t row = blockIdx.x * (int64_t)RPB + threadIdx.x / G;
    cplx acc = cmake(0.0, 0.0);
    if (row < n) {
        const int e1 = rowptr[row + 1];
        for (int e = rowptr[row] + sub; e < e1; e += G) {
            const uint32_t id = idx[e];
            const int64_t c = id & NEP_COL_MASK;
            const int t = id >> NEP_TERM_SHIFT;
            cfma(acc, vals[e], kdot(V + c, ldv, cs + t * k, k));
        }
    }
    acc = group_reduce_sum<G>(acc);
    if (row < n && sub == 0) z[row] = acc;
}

// k == 1 fold for the CSR-vector kernel (small n)
template <int G, typename VT>
__global__ __launch_bounds__(256) void k_spmv_fold(const int32_t* __restrict__ rowptr, const uint32_t* __restrict__ idx,
                                                   const VT* __restrict__ vals, const cplx* __restrict__ v,
                                                   const cplx* __restrict__ C, int64_t ldc, int64_t n,
                                                   cplx* __restrict__ z) {
    constexpr int RPB = 256 / G;
    const int sub = threadIdx.x % G;
    const int64_t row = blockIdx.x * (int64_t)RPB + threadIdx.x / G;
    cplx acc = cmake(0.0, 0.0);
    if (row < n) {
        const int e1 = rowptr[row + 1];
        for (int e = rowptr[row] + sub; e < e1; e += G) {
            const uint32_t id = idx[e];
            const int64_t c = id & NEP_COL_MASK;
            const int t = id >> NEP_TERM_SHIFT;
            cfma(acc, cscale(vals[e], C[(int64_t)t * ldc]), v[c]);
        }
    }
    acc = group_reduce_sum<G>(acc);
    if (row < n && sub == 0) z[row] = acc;
}

// ------------------------------------------------------------------------------------------
// K2 / compute_MM SpMM on ROW-major dense blocks.  One wave per row (grid-stride), lanes = columns.
//   acc[s] = sum_e val[e] * coef(term(e), s) * XT[col(e)*ldx + term(e)*xoff + s]
//   F != null: coef = F[t + s*mt] (K2, xoff = 0);  F == null: coef = 1 (MM, xoff = p)
// Outputs (optional): ZT[row*ldz + s] = acc;  per-block partial sums of |acc|^2 and |XT[row,s]|^2.
template <int NCH, typename VT>
__global__ __launch_bounds__(256) void k_spmm_rm(const int32_t* __restrict__ rowptr,
                                                 const uint32_t* __restrict__ idx,
                                                 const VT* __restrict__ vals, int64_t n, int mt, int k,
                                                 const cplx* __restrict__ F, const cplx* __restrict__ XT,
                                                 int64_t ldx, int xoff, cplx* __restrict__ ZT, int64_t ldz,
                                                 double* __restrict__ partial /* [grid][2][k] or null */, int64_t split_row,
                                                 int xcd_rows) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    cplx* Fs = (cplx*)smem_raw;                       // mt*k coefficients (if F)
    double* red = (double*)(Fs + (F ? mt * k : 0));   // [4][2][NCH*64]
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (F) {
        for (int t = threadIdx.x; t < mt * k; t += 256) Fs[t] = F[t];
        __syncthreads();
    }
    double rn[NCH], qn[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) { rn[c] = 0.0; qn[c] = 0.0; }

    // rows -> workgroups, XCD-contiguous (workgroup id % 8 = XCD) for large matrices: XCD x sweeps the rows [x per, (x + 1) per) with
    // its own workgroups, so the rows of XT a stencil row gathers (i, i +- 1, i +- n_z: a megabyte apart at k = 60) can come from ITS
    // L2 (with the plain grid-stride mapping consecutive 4-row chunks go to different XCDs).  xcd_rows = 0: grid-stride mapping.
    // Measured at n = 1e6: no difference in time (0.728 / 0.734 ms at k = 60) -- this kernel is bound by instruction issue (one
    // wave per row, ~60 instructions per entry, 12 % of the lanes busy at k = 8), not by bytes: 0.55 / 0.69 / 0.73 ms at k = 8 / 30 /
    // 60.  A version with the row pointers and entries fetched one and two rows ahead and the gathers issued in groups of eight
    // was SLOWER (0.79 / 0.98 / 1.04 ms: more instructions); removed.
    const int64_t xcd = blockIdx.x & 7;
    const int64_t nslots = xcd_rows ? ((int64_t)gridDim.x + 7 - xcd) >> 3 : gridDim.x;
    const int64_t per = xcd_rows ? (((n + 7) / 8 + 3) & ~3LL) : n;
    const int64_t r_lo = xcd_rows ? xcd * per : 0, r_hi = xcd_rows ? (r_lo + per < n ? r_lo + per : n) : n;
    const int64_t slot = xcd_rows ? (blockIdx.x >> 3) : blockIdx.x;
    for (int64_t row = r_lo + slot * 4LL + w; row < r_hi; row += nslots * 4LL) {
        cplx acc[NCH];
#pragma unroll
        for (int c = 0; c < NCH; ++c) acc[c] = cmake(0.0, 0.0);
        const int e0 = __builtin_amdgcn_readfirstlane(rowptr[row]);
        const int e1 = __builtin_amdgcn_readfirstlane(rowptr[row + 1]);
        for (int base = e0; base < e1; base += 64) {
            const int me = base + lane;
            uint32_t id_l = 0;
            VT a_l;
            if constexpr (sizeof(VT) == 8) a_l = 0.0; else a_l = cmake(0.0, 0.0);
            if (me < e1) { id_l = idx[me]; a_l = vals[me]; }
            const int m = min(64, e1 - base);
            for (int j = 0; j < m; ++j) {
                const uint32_t id = (uint32_t)readlane_i((int)id_l, j);
                const int64_t c = id & NEP_COL_MASK;
                const int t = id >> NEP_TERM_SHIFT;
                const cplx* xrow = XT + c * ldx + (int64_t)t * xoff;
                if constexpr (sizeof(VT) == 8) {
                    const double a = readlane_d(a_l, j);
#pragma unroll
                    for (int ch = 0; ch < NCH; ++ch) {
                        const int s = lane + 64 * ch;
                        if (s < k) {
                            cplx x = xrow[s];
                            if (F) x = cmul(Fs[t + s * mt], x);
                            cfma(acc[ch], a, x);
                        }
                    }
                } else {
                    cplx a; a.x = readlane_d(a_l.x, j); a.y = readlane_d(a_l.y, j);
#pragma unroll
                    for (int ch = 0; ch < NCH; ++ch) {
                        const int s = lane + 64 * ch;
                        if (s < k) {
                            cplx x = xrow[s];
                            if (F) x = cmul(Fs[t + s * mt], x);
                            cfma(acc[ch], a, x);
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            const int s = lane + 64 * ch;
            if (s < k) {
                // split_row >= 0: rows below it only enter the norms, rows from it on are only written (row - split_row)
                const bool wr = split_row < 0 ? ZT != nullptr : row >= split_row;
                const bool nr = split_row < 0 || row < split_row;
                if (wr) ZT[(split_row < 0 ? row : row - split_row) * ldz + s] = acc[ch];
                if (partial) {
                    if (nr) rn[ch] = fma(acc[ch].x, acc[ch].x, fma(acc[ch].y, acc[ch].y, rn[ch]));
                    if (xoff == 0) {
                        const cplx q = XT[row * ldx + s];
                        qn[ch] = fma(q.x, q.x, fma(q.y, q.y, qn[ch]));
                    }
                }
            }
        }
    }
    if (partial) {
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            red[(w * 2 + 0) * (NCH * 64) + ch * 64 + lane] = rn[ch];
            red[(w * 2 + 1) * (NCH * 64) + ch * 64 + lane] = qn[ch];
        }
        __syncthreads();
        for (int t = threadIdx.x; t < 2 * k; t += 256) {
            const int which = t / k, s = t % k;
            double v = 0.0;
            for (int q = 0; q < 4; ++q) v += red[(q * 2 + which) * (NCH * 64) + s];
            partial[((int64_t)blockIdx.x * 2 + which) * k + s] = v;
        }
    }
}

// out[j] = sum_b partial[b*len + j]: one block per output, fixed summation tree -> deterministic
__global__ __launch_bounds__(256) void k_sum_partials_d(int nb, int len, const double* __restrict__ partial,
                                                        double* __restrict__ out) {
    __shared__ double sm[4];
    const int j = blockIdx.x;
    double t = 0.0;
    for (int b = threadIdx.x; b < nb; b += 256) t += partial[(int64_t)b * len + j];
    t = wave_reduce_sum(t);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = t;
    __syncthreads();
    if (threadIdx.x == 0) out[j] = (sm[0] + sm[1]) + (sm[2] + sm[3]);
}

// ------------------------------------------------------------------------------------------
static int xcd_swizzle() {
    static const int v = getenv("NEP_XCD_SWIZZLE") ? atoi(getenv("NEP_XCD_SWIZZLE")) : 1;
    return v;
}
template <typename VT>
static int launch_spmv(const nep_spmf* s, const cplx* WT, cplx* z, hipStream_t st) {
    const VT* vals = (const VT*)s->d_vals;
    const int64_t n = s->n;
    if (s->d_sell_ptr) {
        const int64_t nsl = (n + 63) / 64;
        hipLaunchKernelGGL((k_spmv_sell<VT, false>), dim3((unsigned)((nsl + 3) / 4)), dim3(256), 0, st, s->d_sell_ptr,
                           s->d_sell_idx, (const VT*)s->d_sell_val, WT, (const cplx*)nullptr, (int64_t)0, s->mt, n, z, xcd_swizzle());
        LAUNCHCHK();
        return NEP_OK;
    }
#define SPMV_CASE(G)                                                                               \
    case G: {                                                                                      \
        const int rpb = 256 / G;                                                                   \
        hipLaunchKernelGGL((k_spmv<G, VT>), dim3((unsigned)((n + rpb - 1) / rpb)), dim3(256), 0, st, \
                           s->d_rowptr, s->d_idx, vals, WT, s->mt, n, z);                          \
        break;                                                                                     \
    }
    switch (s->lanes) {
        SPMV_CASE(2) SPMV_CASE(4) SPMV_CASE(8) SPMV_CASE(16) SPMV_CASE(32) SPMV_CASE(64)
        default: nep_set_error("bad lanes %d", s->lanes); return NEP_ERR_ARG;
    }
#undef SPMV_CASE
    LAUNCHCHK();
    return NEP_OK;
}

// k == 1: z = sum_e val[e] * C[term] * v[col]  (one launch, no W)
template <typename VT>
static int launch_spmv_fold(const nep_spmf* s, const cplx* v, const cplx* dC, int64_t ldc, cplx* z, hipStream_t st) {
    const VT* vals = (const VT*)s->d_vals;
    const int64_t n = s->n;
    if (s->d_sell_ptr) {
        const int64_t nsl = (n + 63) / 64;
        hipLaunchKernelGGL((k_spmv_sell<VT, true>), dim3((unsigned)((nsl + 3) / 4)), dim3(256), 0, st, s->d_sell_ptr,
                           s->d_sell_idx, (const VT*)s->d_sell_val, v, dC, ldc, s->mt, n, z, xcd_swizzle());
        LAUNCHCHK();
        return NEP_OK;
    }
#define FOLD_CASE(G)                                                                                    \
    case G: {                                                                                           \
        const int rpb = 256 / G;                                                                        \
        hipLaunchKernelGGL((k_spmv_fold<G, VT>), dim3((unsigned)((n + rpb - 1) / rpb)), dim3(256), 0, st, \
                           s->d_rowptr, s->d_idx, vals, v, dC, ldc, n, z);                              \
        break;                                                                                          \
    }
    switch (s->lanes) {
        FOLD_CASE(2) FOLD_CASE(4) FOLD_CASE(8) FOLD_CASE(16) FOLD_CASE(32) FOLD_CASE(64)
        default: nep_set_error("bad lanes %d", s->lanes); return NEP_ERR_ARG;
    }
#undef FOLD_CASE
    LAUNCHCHK();
    return NEP_OK;
}

static int fuse_max_small() {
    static const int env = getenv("NEP_K1_FUSE_MAX") ? atoi(getenv("NEP_K1_FUSE_MAX")) : -1;
    return env >= 0 ? (env < 16 ? env : 16) : 16;
}
// 2 <= k <= fuse_max(): one launch, no W (see k_spmv_sell_kfused)
static int fuse_max(const nep_spmf* s) {
    static const int env = getenv("NEP_K1_FUSE_MAX") ? atoi(getenv("NEP_K1_FUSE_MAX")) : -1;
    if (env >= 0) return env < 16 ? env : 16;
    // small n (CSR-vector path) is launch-bound: one launch instead of two (gun k = 10: 6.8 -> 4.8 us).  At n = 1e6 the
    // k gathers per entry miss L1 and the kernel is L2-bandwidth-bound (k = 8: 90 us fused vs 64 us for k_vc + SpMV,
    // whose extra W round trip costs less than the 8x gather traffic), so the SELL path keeps the two-launch form.
    return s->d_sell_ptr ? 0 : 16;
}
template <typename VT>
static int launch_spmv_kfused(const nep_spmf* s, int k, const cplx* V, int64_t ldv, const cplx* dC, int64_t ldc, cplx* z,
                              hipStream_t st) {
    const VT* vals = (const VT*)s->d_vals;
    const int64_t n = s->n;
    const size_t shm = (size_t)s->mt * k * sizeof(cplx);
    if (s->d_sell_ptr) {
        const int64_t nsl = (n + 63) / 64;
#define SKF(K)                                                                                                  \
    case K:                                                                                                     \
        hipLaunchKernelGGL((k_spmv_sell_kfused<VT, K>), dim3((unsigned)((nsl + 3) / 4)), dim3(256), shm, st, s->d_sell_ptr, \
                           s->d_sell_idx, (const VT*)s->d_sell_val, V, ldv, dC, ldc, s->mt, n, z, xcd_swizzle()); \
        break;
        switch (k) {
            SKF(2) SKF(3) SKF(4) SKF(5) SKF(6) SKF(7) SKF(8) SKF(9) SKF(10) SKF(11) SKF(12) SKF(13) SKF(14) SKF(15) SKF(16)
            default: nep_set_error("fused K1: k=%d out of range", k); return NEP_ERR_ARG;
        }
#undef SKF
        LAUNCHCHK();
        return NEP_OK;
    }
#define KF_CASE(G)                                                                                        \
    case G: {                                                                                             \
        const int rpb = 256 / G;                                                                          \
        hipLaunchKernelGGL((k_spmv_kfused<G, VT>), dim3((unsigned)((n + rpb - 1) / rpb)), dim3(256), shm, st, \
                           s->d_rowptr, s->d_idx, vals, V, ldv, k, dC, ldc, s->mt, n, z);                 \
        break;                                                                                            \
    }
    switch (s->lanes) {
        KF_CASE(2) KF_CASE(4) KF_CASE(8) KF_CASE(16) KF_CASE(32) KF_CASE(64)
        default: nep_set_error("bad lanes %d", s->lanes); return NEP_ERR_ARG;
    }
#undef KF_CASE
    LAUNCHCHK();
    return NEP_OK;
}

// ------------------------------------------------------------------------------------------
// Componentwise backward error of an approximate solution of M x = b (the refinement criterion of UMFPACK's solve,
// Arioli/Demmel/Duff):  r = b - Mx,  omega = max_i |r_i| / ( sum_t |c_t| sum_j |A_t[i,j]| |x_j| + |b_i| ).
// One pass over the stacked CSR; the maximum lands in *omega_bits (non-negative doubles order like their bit patterns).
__device__ __forceinline__ double absval(double v) { return fabs(v); }
__device__ __forceinline__ double absval(cplx v) { return sqrt(fma(v.x, v.x, v.y * v.y)); }   // (entries of x and A: no overflow concern)

// FUSED: Mx is not given but accumulated in the same pass with the complex coefficients ccf (pure SPMF operators).
template <int G, typename VT, bool FUSED>
__global__ __launch_bounds__(256) void k_cw_resid(const int32_t* __restrict__ rowptr, const uint32_t* __restrict__ idx,
                                                  const VT* __restrict__ vals, const double* __restrict__ cabs,
                                                  const cplx* __restrict__ ccf,
                                                  const cplx* __restrict__ x, const cplx* __restrict__ b,
                                                  const cplx* __restrict__ Mx, int64_t n, cplx* __restrict__ r,
                                                  unsigned long long* omega_bits,
                                                  const cplx* __restrict__ den_extra, double xsign) {
    constexpr int RPB = 256 / G;
    __shared__ double wmax[4];
    const int sub = threadIdx.x % G;
    const int64_t row = blockIdx.x * (int64_t)RPB + threadIdx.x / G;
    double d = 0.0;
    cplx acc = cmake(0.0, 0.0);
    // (the right-hand side entry and the extra denominator term of the row are requested up front, not behind the reduction)
    cplx brow = cmake(0.0, 0.0); double dex = 0.0;
    if (row < n && sub == 0) { brow = b[row]; if (den_extra) dex = den_extra[row].x; }
    if (row < n) {
        // four entries per lane and trip: their index / value loads are issued together, then the four gathers of x, then the
        // arithmetic -- three dependent round trips per trip instead of two per ENTRY (a gun row has ~60 entries: 4 per lane at G = 16)
        const int e0 = rowptr[row], e1 = rowptr[row + 1];
        for (int e = e0 + sub; e < e1; e += 4 * G) {
            uint32_t id[4]; VT v[4]; cplx xv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int ee = e + u * G < e1 ? e + u * G : e;             // (past the end: a valid entry, not accumulated)
                id[u] = idx[ee]; v[u] = vals[ee];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) xv[u] = x[id[u] & NEP_COL_MASK];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (e + u * G < e1) {
                    const int t = id[u] >> NEP_TERM_SHIFT;
                    d += absval(v[u]) * cabs[t] * absval(xv[u]);
                    if (FUSED) cfma(acc, cscale(v[u], ccf[t]), xv[u]);
                }
            }
        }
    }
    d = group_reduce_sum<G>(d);
    if (FUSED) acc = group_reduce_sum<G>(acc);
    double ratio = 0.0;
    if (row < n && sub == 0) {
        const cplx mx = FUSED ? cmake(xsign * acc.x, xsign * acc.y) : Mx[row];      // xsign = -1: the iterate is stored as -x
        const cplx rr = csub(brow, mx);
        r[row] = rr;
        const double num = absval(rr), den = d + absval(brow) + dex;
        ratio = den > 0.0 ? num / den : (num > 0.0 ? 1.0e300 : 0.0);
        if (!(ratio == ratio)) ratio = 1.0e300;     // NaN -> "not converged"
    }
    // workgroup maximum, one atomic per workgroup
    for (int off = 32; off > 0; off >>= 1) ratio = fmax(ratio, __shfl_xor(ratio, off, 64));
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = ratio;
    __syncthreads();
    if (threadIdx.x == 0 && omega_bits) {
        const double m = fmax(fmax(wmax[0], wmax[1]), fmax(wmax[2], wmax[3]));
        // one atomic per workgroup on ONE word serialises at the end of a kernel this short (623 workgroups on gun: ~12 ns each);
        // a maximum only ever grows, so a workgroup whose value does not exceed what it can already see skips the atomic
        // (device-scope relaxed load: served by L2, never a stale L1 line)
        if (m > 0.0) {
            const unsigned long long mine = (unsigned long long)__double_as_longlong(m);
            const unsigned long long seen = __hip_atomic_load(omega_bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (mine > seen) atomicMax(omega_bits, mine);
        }
    }
}

template <typename VT>
static int launch_cw_resid(const nep_spmf* s, const double* cabs, const cplx* ccf, const cplx* x, const cplx* b,
                           const cplx* Mx, cplx* r, unsigned long long* omega_bits, const cplx* den_extra,
                           hipStream_t st, double xsign = 1.0) {
    const int64_t n = s->n;
    const VT* vals = (const VT*)s->d_vals;
#define CW_CASE(G)                                                                                      \
    case G: {                                                                                           \
        const int rpb = 256 / G;                                                                        \
        if (Mx)                                                                                         \
            hipLaunchKernelGGL((k_cw_resid<G, VT, false>), dim3((unsigned)((n + rpb - 1) / rpb)), dim3(256), 0, st, \
                               s->d_rowptr, s->d_idx, vals, cabs, ccf, x, b, Mx, n, r, omega_bits, den_extra, xsign); \
        else                                                                                            \
            hipLaunchKernelGGL((k_cw_resid<G, VT, true>), dim3((unsigned)((n + rpb - 1) / rpb)), dim3(256), 0, st, \
                               s->d_rowptr, s->d_idx, vals, cabs, ccf, x, b, Mx, n, r, omega_bits, den_extra, xsign); \
        break;                                                                                          \
    }
    switch (s->lanes) {
        CW_CASE(2) CW_CASE(4) CW_CASE(8) CW_CASE(16) CW_CASE(32) CW_CASE(64)
        default: nep_set_error("bad lanes %d", s->lanes); return NEP_ERR_ARG;
    }
#undef CW_CASE
    LAUNCHCHK();
    return NEP_OK;
}

template <int ROWS>
static int launch_vc_rows(const nep_spmf* s, int k, const cplx* dC, int64_t ldc, const cplx* V, int64_t ldv, hipStream_t st, cplx* shift_dst) {
    const int64_t n = s->n;
    const dim3 grid((unsigned)((n + ROWS - 1) / ROWS)), block(512);
    int i0 = 0;
    while (i0 < s->mt) {
        const int rem = s->mt - i0;
        const int cnt = rem >= 4 ? 4 : rem;
        switch (cnt) {
            case 4: if (i0 == 0 && shift_dst) hipLaunchKernelGGL((k_vc<4, ROWS, true>), grid, block, 0, st, V, ldv, n, k, dC, ldc, i0, s->mt, s->d_WT, shift_dst); else hipLaunchKernelGGL((k_vc<4, ROWS, false>), grid, block, 0, st, V, ldv, n, k, dC, ldc, i0, s->mt, s->d_WT, (cplx*)nullptr); break;
            case 3: if (i0 == 0 && shift_dst) hipLaunchKernelGGL((k_vc<3, ROWS, true>), grid, block, 0, st, V, ldv, n, k, dC, ldc, i0, s->mt, s->d_WT, shift_dst); else hipLaunchKernelGGL((k_vc<3, ROWS, false>), grid, block, 0, st, V, ldv, n, k, dC, ldc, i0, s->mt, s->d_WT, (cplx*)nullptr); break;
            case 2: if (i0 == 0 && shift_dst) hipLaunchKernelGGL((k_vc<2, ROWS, true>), grid, block, 0, st, V, ldv, n, k, dC, ldc, i0, s->mt, s->d_WT, shift_dst); else hipLaunchKernelGGL((k_vc<2, ROWS, false>), grid, block, 0, st, V, ldv, n, k, dC, ldc, i0, s->mt, s->d_WT, (cplx*)nullptr); break;
            default: if (i0 == 0 && shift_dst) hipLaunchKernelGGL((k_vc<1, ROWS, true>), grid, block, 0, st, V, ldv, n, k, dC, ldc, i0, s->mt, s->d_WT, shift_dst); else hipLaunchKernelGGL((k_vc<1, ROWS, false>), grid, block, 0, st, V, ldv, n, k, dC, ldc, i0, s->mt, s->d_WT, (cplx*)nullptr); break;
        }
        LAUNCHCHK();
        i0 += cnt;
    }
    return NEP_OK;
}
static int launch_vc(const nep_spmf* s, int k, const cplx* dC, int64_t ldc, const cplx* V, int64_t ldv, hipStream_t st, cplx* shift_dst = nullptr) {
    static const int force = getenv("NEP_VC_ROWS") ? atoi(getenv("NEP_VC_ROWS")) : 0;
    if (force == 16) return launch_vc_rows<16>(s, k, dC, ldc, V, ldv, st, shift_dst);
    if (force == 32) return launch_vc_rows<32>(s, k, dC, ldc, V, ldv, st, shift_dst);
    if (force == 64) return launch_vc_rows<64>(s, k, dC, ldc, V, ldv, st, shift_dst);
    if (s->n >= 65536) return launch_vc_rows<64>(s, k, dC, ldc, V, ldv, st, shift_dst);
    return launch_vc_rows<32>(s, k, dC, ldc, V, ldv, st, shift_dst);
}

// The same for k <= 64 with the dependent round trips of a row taken apart (round 3, late): k_spmm_rm waits for the gathered row
// of XT of every ENTRY before it issues the next one (8 round trips per stencil row) behind the row-pointer and entry loads -- a
// wave spends 4.5 us per row whatever k.  Here the entries of the NEXT row are fetched during the current row (its pointers one row
// earlier still), and the gathers of eight entries are issued back to back, unconditionally (lanes >= k read column k - 1, entries
// past the row's end repeat its last entry with coefficient 0), before the first is consumed.
// (One accumulator per term picked by a wave-uniform branch, with the coefficients applied once per row, was tried: the branches cost
// more than the four FP64 operations they save -- 0.48 / 0.50 / 0.52 ms against 0.44 / 0.47 / 0.50.)
template <typename VT, int NCH>
__global__ __launch_bounds__(256) void k_spmm_rm_g(const int32_t* __restrict__ rowptr, const uint32_t* __restrict__ idx,
                                                   const VT* __restrict__ vals, int64_t n, int mt, int k,
                                                   const cplx* __restrict__ F, const cplx* __restrict__ XT,
                                                   int64_t ldx, int xoff, cplx* __restrict__ ZT, int64_t ldz,
                                                   double* __restrict__ partial, int64_t split_row, int xcd_rows) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    cplx* Fs = (cplx*)smem_raw;                       // mt*k coefficients (if F)
    double* red = (double*)(Fs + (F ? mt * k : 0));   // [4][2][NCH*64]
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (F) {
        for (int t = threadIdx.x; t < mt * k; t += 256) Fs[t] = F[t];
        __syncthreads();
    }
    constexpr int GU = NCH == 1 ? 8 : 4;              // entries whose gathers are in flight together (8 loads either way)
    int sl[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) sl[c] = lane + 64 * c < k ? lane + 64 * c : k - 1;
    double rn[NCH], qn[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) { rn[c] = 0.0; qn[c] = 0.0; }
    const int64_t xcd = blockIdx.x & 7;
    const int64_t nslots = xcd_rows ? ((int64_t)gridDim.x + 7 - xcd) >> 3 : gridDim.x;
    const int64_t per = xcd_rows ? (((n + 7) / 8 + 3) & ~3LL) : n;
    const int64_t r_lo = xcd_rows ? xcd * per : 0, r_hi = xcd_rows ? (r_lo + per < n ? r_lo + per : n) : n;
    const int64_t slot = xcd_rows ? (blockIdx.x >> 3) : blockIdx.x;
    const int64_t step = nslots * 4LL;
    int64_t row = r_lo + slot * 4LL + w;
    if (row >= r_hi) row = -1;
    const int64_t last = r_hi - 1;
    const int nnzm1 = max(__builtin_amdgcn_readfirstlane(rowptr[n]) - 1, 0);
    // pipeline registers: entries of the current row (id_c, a_c, extent ce0..ce1), pointers of the next row (e0n, e1n)
    int ce0 = 0, ce1 = 0, e0n = 0, e1n = 0;
    uint32_t id_c = 0; VT a_c;
    if constexpr (sizeof(VT) == 8) a_c = 0.0; else a_c = cmake(0.0, 0.0);
    if (row >= 0) {
        ce0 = __builtin_amdgcn_readfirstlane(rowptr[row]); ce1 = __builtin_amdgcn_readfirstlane(rowptr[row + 1]);
        const int me = min(min(ce0 + lane, ce1 > ce0 ? ce1 - 1 : ce0), nnzm1);        // (an empty LAST row has ce0 = nnz: stay inside the arrays)
        id_c = idx[me]; a_c = vals[me];
        const int64_t r1 = min(row + step, last);
        e0n = __builtin_amdgcn_readfirstlane(rowptr[r1]); e1n = __builtin_amdgcn_readfirstlane(rowptr[r1 + 1]);
    }
    for (; row >= 0 && row < r_hi; row += step) {
        // ---- ahead (unconditional, clamped to the last row of the range): entries of the next row, pointers of the one after
        const int ne0 = e0n, ne1 = e1n;
        const int men = min(min(ne0 + lane, ne1 > ne0 ? ne1 - 1 : ne0), nnzm1);
        const uint32_t id_n = idx[men];
        const VT a_n = vals[men];
        {
            const int64_t r2 = min(row + 2 * step, last);
            e0n = __builtin_amdgcn_readfirstlane(rowptr[r2]); e1n = __builtin_amdgcn_readfirstlane(rowptr[r2 + 1]);
        }
        // (the row of XT for |q|^2 is fetched with the gathers, not after them: one round trip less per row)
        const bool want_q = partial && xoff == 0;
        cplx qv[NCH], acc[NCH];
#pragma unroll
        for (int c = 0; c < NCH; ++c) { qv[c] = cmake(0.0, 0.0); if (want_q) qv[c] = XT[row * ldx + sl[c]]; acc[c] = cmake(0.0, 0.0); }
        auto chunk = [&](uint32_t id_l, VT a_l, int m) __attribute__((always_inline)) {
            // element offsets of all (up to 64) entries at once on the lanes, 32-bit (the host checks n ldx + mt xoff < 2^32): the
            // per-entry scalar work is a readlane and one 64-bit add instead of three scalar multiplies -- this kernel is bound by
            // instruction issue (the scalar unit is shared by the four SIMDs of a CU), not by bytes
            const uint32_t off_l = (id_l & NEP_COL_MASK) * (uint32_t)ldx + (id_l >> NEP_TERM_SHIFT) * (uint32_t)xoff;
            const uint32_t t_l = id_l >> NEP_TERM_SHIFT;
            for (int j0 = 0; j0 < m; j0 += GU) {
                cplx xv[GU][NCH]; int tt[GU]; VT av[GU];
#pragma unroll
                for (int u = 0; u < GU; ++u) {
                    const int j = min(j0 + u, m - 1);
                    const uint32_t o = (uint32_t)readlane_i((int)off_l, j);
                    tt[u] = F ? readlane_i((int)t_l, j) : 0;
                    const cplx* xrow = XT + o;
#pragma unroll
                    for (int c = 0; c < NCH; ++c) xv[u][c] = xrow[sl[c]];
                    if constexpr (sizeof(VT) == 8) { const double a = readlane_d(a_l, j); av[u] = j0 + u < m ? a : 0.0; }
                    else { cplx a; a.x = readlane_d(a_l.x, j); a.y = readlane_d(a_l.y, j); av[u] = j0 + u < m ? a : cmake(0.0, 0.0); }
                }
#pragma unroll
                for (int u = 0; u < GU; ++u)
#pragma unroll
                    for (int c = 0; c < NCH; ++c) {
                        cplx x = xv[u][c];
                        if (F) x = cmul(Fs[tt[u] + sl[c] * mt], x);
                        cfma(acc[c], av[u], x);
                    }
            }
        };
        chunk(id_c, a_c, min(64, ce1 - ce0));          // (straight-line: the only loads in flight are the two prefetches above)
        for (int base = ce0 + 64; base < ce1; base += 64) {      // rows with more than 64 entries: further chunks loaded in place
            const int me = min(base + lane, ce1 - 1);
            chunk(idx[me], vals[me], min(64, ce1 - base));
        }
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int sc = lane + 64 * c;
            if (sc < k) {
                const bool wr = split_row < 0 ? ZT != nullptr : row >= split_row;
                const bool nr = split_row < 0 || row < split_row;
                if (wr) ZT[(split_row < 0 ? row : row - split_row) * ldz + sc] = acc[c];
                if (partial) {
                    if (nr) rn[c] = fma(acc[c].x, acc[c].x, fma(acc[c].y, acc[c].y, rn[c]));
                    if (xoff == 0) qn[c] = fma(qv[c].x, qv[c].x, fma(qv[c].y, qv[c].y, qn[c]));
                }
            }
        }
        ce0 = ne0; ce1 = ne1; id_c = id_n; a_c = a_n;
    }
    if (partial) {
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            red[(w * 2 + 0) * (NCH * 64) + c * 64 + lane] = rn[c];
            red[(w * 2 + 1) * (NCH * 64) + c * 64 + lane] = qn[c];
        }
        __syncthreads();
        for (int t = threadIdx.x; t < 2 * k; t += 256) {
            const int which = t / k, sidx = t % k;
            double v = 0.0;
            for (int q = 0; q < 4; ++q) v += red[(q * 2 + which) * (NCH * 64) + sidx];
            partial[((int64_t)blockIdx.x * 2 + which) * k + sidx] = v;
        }
    }
}

template <typename VT>
static int launch_spmm(const nep_spmf* s, int k, const cplx* dF, const cplx* XT, int64_t ldx, int xoff,
                       cplx* ZT, int64_t ldz, double* partial, int grid, hipStream_t st, int64_t split_row = -1) {
    const int nch = (k + 63) / 64;
    const size_t shm = (dF ? (size_t)s->mt * k * sizeof(cplx) : 0) + (size_t)4 * 2 * nch * 64 * sizeof(double);
    const VT* vals = (const VT*)s->d_vals;
    static const int xcd_env = getenv("NEP_SPMM_XCD") ? atoi(getenv("NEP_SPMM_XCD")) : 1;
    const int xcd_rows = (xcd_env && s->n >= 65536 && grid >= 64) ? 1 : 0;      // small matrices sit in every L2 anyway
#define SPMM_CASE(N)                                                                                   \
    case N:                                                                                            \
        hipLaunchKernelGGL((k_spmm_rm<N, VT>), dim3(grid), dim3(256), shm, st, s->d_rowptr, s->d_idx,  \
                           vals, s->n, s->mt, k, dF, XT, ldx, xoff, ZT, ldz, partial, split_row, xcd_rows); \
        break;
    static const int grouped = getenv("NEP_SPMM_GROUPED") ? atoi(getenv("NEP_SPMM_GROUPED")) : 1;
    if (nch <= 2 && grouped && (uint64_t)s->n * (uint64_t)ldx + (uint64_t)s->mt * (uint64_t)(xoff < 0 ? -xoff : xoff) < (1ull << 32)) {
if (nch == 1) hipLaunchKernelGGL((k_spmm_rm_g<VT, 1>), dim3(grid), dim3(256), shm, st, s->d_rowptr, s->d_idx, vals, s->n, s->mt, k, dF, XT, ldx,
                                         xoff, ZT, ldz, partial, split_row, xcd_rows);
        else hipLaunchKernelGGL((k_spmm_rm_g<VT, 2>), dim3(grid), dim3(256), shm, st, s->d_rowptr, s->d_idx, vals, s->n, s->mt, k, dF, XT, ldx,
                                xoff, ZT, ldz, partial, split_row, xcd_rows);
        LAUNCHCHK();
        return NEP_OK;
    }
    switch (nch) {
        SPMM_CASE(1) SPMM_CASE(2) SPMM_CASE(3) SPMM_CASE(4)
        default: nep_set_error("k=%d too large for one spmm pass", k); return NEP_ERR_ARG;
    }
#undef SPMM_CASE
    LAUNCHCHK();
    return NEP_OK;
}

extern "C" {

// stacked CSR on the host: per row, the entries of all terms sorted by (col, term)
static int stack_terms(int64_t n, int32_t mt, const int32_t* const* h_rowptr, const int32_t* const* h_colind,
                       const void* const* h_vals, const int32_t* h_val_is_complex, std::vector<int32_t>& rowptr,
                       std::vector<uint32_t>& idx, std::vector<double>& vr, std::vector<nep_cdouble>& vc, bool* any_complex_out,
                       int64_t* nnz_out) {
    ARGCHK(n > 0 && n <= (int64_t)NEP_COL_MASK);
    ARGCHK(mt > 0 && mt <= NEP_MAX_TERMS);
    ARGCHK(h_rowptr && h_colind && h_vals && h_val_is_complex);
    bool any_complex = false;
    int64_t nnz = 0;
    for (int i = 0; i < mt; ++i) {
        ARGCHK(h_rowptr[i] && h_rowptr[i][0] == 0);
        nnz += h_rowptr[i][n];
        any_complex |= (h_val_is_complex[i] != 0);
    }
    ARGCHK(nnz < ((int64_t)1 << 31) - 64);
    rowptr.assign(n + 1, 0);
    idx.assign(nnz, 0u);
    vr.assign(any_complex ? 0 : nnz, 0.0);
    vc.assign(any_complex ? nnz : 0, nep_cdouble());
    struct Ent { uint32_t col; uint32_t term; double re, im; };
    std::vector<Ent> tmp;
    int64_t pos = 0;
    rowptr[0] = 0;
    for (int64_t r = 0; r < n; ++r) {
        tmp.clear();
        for (int i = 0; i < mt; ++i) {
            for (int32_t e = h_rowptr[i][r]; e < h_rowptr[i][r + 1]; ++e) {
                const int32_t c = h_colind[i][e];
                if (c < 0 || c >= n) { nep_set_error("column index out of range"); return NEP_ERR_ARG; }
                Ent en; en.col = (uint32_t)c; en.term = (uint32_t)i;
                if (h_val_is_complex[i]) {
                    const nep_cdouble v = ((const nep_cdouble*)h_vals[i])[e]; en.re = v.re; en.im = v.im;
                } else { en.re = ((const double*)h_vals[i])[e]; en.im = 0.0; }
                tmp.push_back(en);
            }
        }
        std::stable_sort(tmp.begin(), tmp.end(), [](const Ent& a, const Ent& b) {
            return a.col != b.col ? a.col < b.col : a.term < b.term; });
        for (const Ent& en : tmp) {
            idx[pos] = (en.term << NEP_TERM_SHIFT) | en.col;
            if (any_complex) { vc[pos].re = en.re; vc[pos].im = en.im; } else vr[pos] = en.re;
            ++pos;
        }
        rowptr[r + 1] = (int32_t)pos;
    }
    *any_complex_out = any_complex; *nnz_out = nnz;
    return NEP_OK;
}

extern "C" int nep_tiles_dryrun(int64_t n, int mt, int valbytes, const int32_t* rowptr, const uint32_t* idx, const void* vals, int k,
                                int64_t info[8], double* maxerr);

// host-only dry run of the footprint tiles of the one-launch K1 kernel (csrc/spmv_tile.hip) for the SPMF given as in
// nep_spmf_create: nothing goes to the device.  info as nep_spmf_tile_info; *maxerr = largest relative difference between
// z = sum_t A_t (V c_t) evaluated through the tiles and directly (deterministic V, C with k columns / rows).
int32_t nep_spmf_tiles_analyze(int64_t n, int32_t mt, const int32_t* const* h_rowptr, const int32_t* const* h_colind,
                               const void* const* h_vals, const int32_t* h_val_is_complex, int32_t k, int64_t info[8],
                               double* maxerr) {
    ARGCHK(info && maxerr && k >= 1);
    std::vector<int32_t> rowptr; std::vector<uint32_t> idx; std::vector<double> vr; std::vector<nep_cdouble> vc;
    bool any_complex = false; int64_t nnz = 0;
    const int rc = stack_terms(n, mt, h_rowptr, h_colind, h_vals, h_val_is_complex, rowptr, idx, vr, vc, &any_complex, &nnz);
    if (rc) return rc;
    return nep_tiles_dryrun(n, mt, any_complex ? 16 : 8, rowptr.data(), idx.data(),
                            any_complex ? (const void*)vc.data() : (const void*)vr.data(), k, info, maxerr);
}

int32_t nep_spmf_create(int64_t n, int32_t mt, const int32_t* const* h_rowptr, const int32_t* const* h_colind,
                        const void* const* h_vals, const int32_t* h_val_is_complex, nep_spmf** out) {
    ARGCHK(out != nullptr);
    *out = nullptr;
    std::vector<int32_t> rowptr; std::vector<uint32_t> idx; std::vector<double> vr; std::vector<nep_cdouble> vc;
    bool any_complex = false; int64_t nnz = 0;
    {
        const int rc = stack_terms(n, mt, h_rowptr, h_colind, h_vals, h_val_is_complex, rowptr, idx, vr, vc, &any_complex, &nnz);
        if (rc) return rc;
    }
    nep_spmf* s = new nep_spmf();
    s->n = n; s->mt = mt; s->nnz = nnz; s->valbytes = any_complex ? 16 : 8;
    // lanes per row: power of two near the mean row length (env NEP_SPMV_LANES overrides)
    {
        const double mean = (double)nnz / (double)n;
        int g = 2;
        while (g < 64 && g < mean) g <<= 1;
        if (const char* e = getenv("NEP_SPMV_LANES")) { int v = atoi(e); if (v >= 2 && v <= 64 && (v & (v - 1)) == 0) g = v; }
        s->lanes = g;
    }
#define CRCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { nep_set_error("%s:%d: %s: %s", __FILE__, __LINE__, #call, hipGetErrorString(e_)); nep_spmf_destroy(s); return NEP_ERR_HIP; } } while (0)
    CRCHK(hipMalloc((void**)&s->d_rowptr, (size_t)(n + 1) * sizeof(int32_t)));
    CRCHK(hipMalloc((void**)&s->d_idx, (size_t)(nnz + 64) * sizeof(uint32_t)));
    CRCHK(hipMalloc(&s->d_vals, (size_t)(nnz + 64) * s->valbytes));
    CRCHK(hipMalloc((void**)&s->d_WT, (size_t)n * mt * sizeof(cplx)));
    CRCHK(hipMemcpy(s->d_rowptr, rowptr.data(), (size_t)(n + 1) * sizeof(int32_t), hipMemcpyHostToDevice));
    CRCHK(hipMemcpy(s->d_idx, idx.data(), (size_t)nnz * sizeof(uint32_t), hipMemcpyHostToDevice));
    if (any_complex) CRCHK(hipMemcpy(s->d_vals, vc.data(), (size_t)nnz * 16, hipMemcpyHostToDevice));
    else CRCHK(hipMemcpy(s->d_vals, vr.data(), (size_t)nnz * 8, hipMemcpyHostToDevice));
    // ---- SELL-64 copy for large n (lane-per-row SpMV); env NEP_SELL=0/1 overrides the size rule
    {
        bool want = n >= 32768;
        if (const char* e = getenv("NEP_SELL")) want = atoi(e) != 0;
        if (want) {
            const int64_t nsl = (n + 63) / 64;
            std::vector<int32_t> sptr(nsl + 1, 0);
            for (int64_t sl = 0; sl < nsl; ++sl) {
                int32_t mx = 0;
                for (int64_t r = sl * 64; r < std::min<int64_t>(n, sl * 64 + 64); ++r) mx = std::max(mx, rowptr[r + 1] - rowptr[r]);
                sptr[sl + 1] = sptr[sl] + mx;
            }
            const int64_t cols = sptr[nsl];
            if (cols * 64 <= 4 * nnz + 4096) {          // refuse pathological padding (> 4x)
                std::vector<uint32_t> sidx((size_t)cols * 64, 0u);
                std::vector<double> sr(any_complex ? 0 : (size_t)cols * 64, 0.0);
                std::vector<nep_cdouble> sc(any_complex ? (size_t)cols * 64 : 0);
                for (int64_t sl = 0; sl < nsl; ++sl)
                    for (int l = 0; l < 64; ++l) {
                        const int64_t r = sl * 64 + l;
                        if (r >= n) continue;
                        for (int32_t e = rowptr[r]; e < rowptr[r + 1]; ++e) {
                            const size_t q = ((size_t)sptr[sl] + (e - rowptr[r])) * 64 + l;
                            sidx[q] = idx[e];
                            if (any_complex) sc[q] = vc[e]; else sr[q] = vr[e];
                        }
                    }
                CRCHK(hipMalloc((void**)&s->d_sell_ptr, (size_t)(nsl + 1) * 4));
                CRCHK(hipMalloc((void**)&s->d_sell_idx, (size_t)cols * 64 * 4 + 256));
                CRCHK(hipMalloc(&s->d_sell_val, (size_t)cols * 64 * s->valbytes + 256));
                CRCHK(hipMemcpy(s->d_sell_ptr, sptr.data(), (size_t)(nsl + 1) * 4, hipMemcpyHostToDevice));
                CRCHK(hipMemcpy(s->d_sell_idx, sidx.data(), (size_t)cols * 64 * 4, hipMemcpyHostToDevice));
                if (any_complex) CRCHK(hipMemcpy(s->d_sell_val, sc.data(), (size_t)cols * 64 * 16, hipMemcpyHostToDevice));
                else CRCHK(hipMemcpy(s->d_sell_val, sr.data(), (size_t)cols * 64 * 8, hipMemcpyHostToDevice));
                s->sell_cols = cols;
            }
        }
    }
#undef CRCHK
    {
        const int rc = nep_tiles_build(n, mt, s->valbytes, rowptr.data(), idx.data(),
                                       any_complex ? (const void*)vc.data() : (const void*)vr.data(), &s->tiles);
        if (rc) { nep_spmf_destroy(s); return rc; }
    }
    *out = s;
    return NEP_OK;
}

// K1 kernel choice (tuning / A-B knob): 0 = automatic, 1 = footprint tiles whenever they exist, 2 = never tiles
static int g_k1_mode = -1;
int32_t nep_k1_set_mode(int32_t mode) { g_k1_mode = mode; return NEP_OK; }
// K2 super-panel kernel (A-B knob, same values as NEP_K2_SP; -1 = the environment decides)
static int g_k2_sp_mode = -1;
int32_t nep_k2_set_sp_mode(int32_t mode) { g_k2_sp_mode = mode; return NEP_OK; }
static bool use_tiles(const nep_spmf* s, int k) {
    if (!s->tiles) return false;
    if (g_k1_mode < 0) g_k1_mode = getenv("NEP_K1_MODE") ? atoi(getenv("NEP_K1_MODE")) : 0;
    if (g_k1_mode == 2) return false;
    if (nep_tiles_shmem(s->tiles, k) > 160 * 1024) return false;
    if (g_k1_mode == 1) return true;
    // measured ranges: at n = 1e6 the tiles win for 2 <= k <= 12 (no W round trip; k = 1 stays with the folded
    // SELL SpMV, large k with k_vc + SELL whose W round trip is small against V); at gun size see NEP_K1_TILE_SMALL_KMIN
    static const int kmin_l = getenv("NEP_K1_TILE_KMIN") ? atoi(getenv("NEP_K1_TILE_KMIN")) : 2;
    static const int kmax_l = getenv("NEP_K1_TILE_KMAX") ? atoi(getenv("NEP_K1_TILE_KMAX")) : 12;
    static const int kmin_s = getenv("NEP_K1_TILE_SMALL_KMIN") ? atoi(getenv("NEP_K1_TILE_SMALL_KMIN")) : (1 << 30);
    if (s->d_sell_ptr) return k >= kmin_l && k <= kmax_l;
    return k >= kmin_s;
}

// K2 on the tiles: large matrices only (at gun size the wave-per-row kernel's gathers are L2 hits and it is launch-bound)
static bool use_tiles_k2(const nep_spmf* s, int k) {
    if (!s->tiles) return false;
    if (g_k1_mode < 0) g_k1_mode = getenv("NEP_K1_MODE") ? atoi(getenv("NEP_K1_MODE")) : 0;
    if (g_k1_mode == 2 || !nep_tiles_resid_ok(s->tiles, k)) return false;
    if (g_k1_mode == 1) return true;
    // measured at n = 1e6: 4.5x faster than the wave-per-row kernel at k = 8, 1.35x at k = 30, slower at k = 60
    // (row-major Q: a column panel of a footprint row is a 64-byte piece of a 16 k-byte row)
    static const int kmax = getenv("NEP_K2_TILE_KMAX") ? atoi(getenv("NEP_K2_TILE_KMAX")) : 20;     // above: k_spmm_rm_g (0.47 ms at k = 30, tiles 0.50-0.63)
    return s->d_sell_ptr != nullptr && k <= kmax;
}

// K2 in super-panels (k_tile_resid_sp, spmv_tile.hip): NEP_K2_SP = 0 never, 1 (default) on large matrices (those with a SELL copy: the
// sizes at which K2 is bound by HBM; at gun size the wave-per-row kernel's gathers are L2 hits), 2 whenever the tiles allow it (tests)
static bool use_sp_k2(const nep_spmf* s, int k, int cm) {
    static const int mode = getenv("NEP_K2_SP") ? atoi(getenv("NEP_K2_SP")) : 1;
    const int m = g_k2_sp_mode >= 0 ? g_k2_sp_mode : mode;
    if (m == 0 || !s->tiles || !nep_tiles_resid_sp_ok(s->tiles, k, cm)) return false;
    return m == 2 || s->d_sell_ptr != nullptr;
}

int32_t nep_spmf_tile_info(const nep_spmf* s, int64_t info[8]) {
    ARGCHK(s && info);
    for (int i = 0; i < 8; ++i) info[i] = 0;
    if (s->tiles) nep_tiles_info(s->tiles, info);
    return NEP_OK;
}

int32_t nep_spmf_destroy(nep_spmf* s) {
    if (!s) return NEP_OK;
    if (s->tiles) nep_tiles_destroy(s->tiles);
    if (s->d_rowptr) (void)hipFree(s->d_rowptr);
    if (s->d_idx) (void)hipFree(s->d_idx);
    if (s->d_vals) (void)hipFree(s->d_vals);
    if (s->d_WT) (void)hipFree(s->d_WT);
    if (s->d_sell_ptr) (void)hipFree(s->d_sell_ptr);
    if (s->d_sell_idx) (void)hipFree(s->d_sell_idx);
    if (s->d_sell_val) (void)hipFree(s->d_sell_val);
    s->coef.release();
    s->part.release();
    s->cwpart.release();
    s->ring.release();
    s->cwring.release();
    delete s;
    return NEP_OK;
}

int32_t nep_spmf_info(const nep_spmf* s, int64_t info[6]) {
    ARGCHK(s && info);
    info[0] = s->n; info[1] = s->mt; info[2] = s->nnz; info[3] = s->valbytes; info[4] = s->lanes;
    info[5] = s->nnz * (s->valbytes + 4) + 4 * (s->n + 1);
    return NEP_OK;
}

int32_t nep_mlincomb(nep_spmf* s, int32_t k, const nep_cdouble* hC, const nep_cdouble* dV, int64_t ldv,
                     nep_cdouble* dz, nep_stream stream) {
    ARGCHK(s && hC && dV && dz);
    ARGCHK(k >= 1 && ldv >= s->n);
    hipStream_t st = as_stream(stream);
    const size_t cbytes = (size_t)k * s->mt * sizeof(cplx);
    int rc = s->coef.ensure(cbytes);
    if (rc) return rc;
    // each call gets its own region of the device staging buffer? no: stream order protects dptr;
    // the pinned ring protects the host side
    rc = s->ring.upload(s->coef.dptr, hC, cbytes, st);
    if (rc) return rc;
    if (use_tiles(s, k))
        return nep_tiles_mlincomb(s->tiles, k, (const cplx*)s->coef.dptr, k, (const cplx*)dV, ldv, (cplx*)dz, nullptr, st);
    if (k == 1) {
        if (s->valbytes == 8) return launch_spmv_fold<double>(s, (const cplx*)dV, (const cplx*)s->coef.dptr, 1, (cplx*)dz, st);
        return launch_spmv_fold<cplx>(s, (const cplx*)dV, (const cplx*)s->coef.dptr, 1, (cplx*)dz, st);
    }
    if (k <= fuse_max(s) && (size_t)s->mt * k * sizeof(cplx) <= 48 * 1024) {
        if (s->valbytes == 8) return launch_spmv_kfused<double>(s, k, (const cplx*)dV, ldv, (const cplx*)s->coef.dptr, k, (cplx*)dz, st);
        return launch_spmv_kfused<cplx>(s, k, (const cplx*)dV, ldv, (const cplx*)s->coef.dptr, k, (cplx*)dz, st);
    }
    rc = launch_vc(s, k, (const cplx*)s->coef.dptr, k, (const cplx*)dV, ldv, st);
    if (rc) return rc;
    if (s->valbytes == 8) return launch_spmv<double>(s, s->d_WT, (cplx*)dz, st);
    return launch_spmv<cplx>(s, s->d_WT, (cplx*)dz, st);
}

int32_t nep_mlincomb_dev(nep_spmf* s, int32_t k, const nep_cdouble* dC, int64_t ldc, const nep_cdouble* dV,
                         int64_t ldv, nep_cdouble* dz, nep_stream stream) {
    ARGCHK(s && dC && dV && dz);
    ARGCHK(k >= 1 && ldv >= s->n && ldc >= k);
    hipStream_t st = as_stream(stream);
    if (use_tiles(s, k))
        return nep_tiles_mlincomb(s->tiles, k, (const cplx*)dC, ldc, (const cplx*)dV, ldv, (cplx*)dz, nullptr, st);
    if (k == 1) {
        if (s->valbytes == 8) return launch_spmv_fold<double>(s, (const cplx*)dV, (const cplx*)dC, ldc, (cplx*)dz, st);
        return launch_spmv_fold<cplx>(s, (const cplx*)dV, (const cplx*)dC, ldc, (cplx*)dz, st);
    }
    if (k <= fuse_max(s) && (size_t)s->mt * k * sizeof(cplx) <= 48 * 1024) {
        if (s->valbytes == 8) return launch_spmv_kfused<double>(s, k, (const cplx*)dV, ldv, (const cplx*)dC, ldc, (cplx*)dz, st);
        return launch_spmv_kfused<cplx>(s, k, (const cplx*)dV, ldv, (const cplx*)dC, ldc, (cplx*)dz, st);
    }
    int rc = launch_vc(s, k, (const cplx*)dC, ldc, (const cplx*)dV, ldv, st);
    if (rc) return rc;
    if (s->valbytes == 8) return launch_spmv<double>(s, s->d_WT, (cplx*)dz, st);
    return launch_spmv<cplx>(s, s->d_WT, (cplx*)dz, st);
}

// nep_mlincomb_dev for iar's step: when the coefficient product runs as its own kernel (k_vc) the block shift of the basis
// column is folded into it (d_shift = destination of block 1; *folded = 1), otherwise the caller issues nep_iar_shift_scale
int nep_mlincomb_dev_shift(nep_spmf* s, int32_t k, const nep_cdouble* dC, int64_t ldc, const nep_cdouble* dV, int64_t ldv,
                           nep_cdouble* dz, nep_cdouble* d_shift, int32_t* folded, hipStream_t st) {
    *folded = 0;
    if (use_tiles(s, k) && !getenv("NEP_NO_SHIFT_FOLD")) {      // one launch: coefficient product, SpMV and the block shift
        *folded = 1;
        return nep_tiles_mlincomb(s->tiles, k, (const cplx*)dC, ldc, (const cplx*)dV, ldv, (cplx*)dz, (cplx*)d_shift, st);
    }
    if (k == 1 || (k <= fuse_max(s) && (size_t)s->mt * k * sizeof(cplx) <= 48 * 1024) || getenv("NEP_NO_SHIFT_FOLD"))
        return nep_mlincomb_dev(s, k, dC, ldc, dV, ldv, dz, (nep_stream)st);
    int rc = launch_vc(s, k, (const cplx*)dC, ldc, (const cplx*)dV, ldv, st, (cplx*)d_shift);
    if (rc) return rc;
    *folded = 1;
    if (s->valbytes == 8) return launch_spmv<double>(s, s->d_WT, (cplx*)dz, st);
    return launch_spmv<cplx>(s, s->d_WT, (cplx*)dz, st);
}

// the second half of K1 on a coefficient product that exists already (iar: formed by the previous step's k_orth_finish_vc)
int nep_spmv_wt(nep_spmf* s, const nep_cdouble* d_WT, nep_cdouble* dz, hipStream_t st) {
    if (s->valbytes == 8) return launch_spmv<double>(s, (const cplx*)d_WT, (cplx*)dz, st);
    return launch_spmv<cplx>(s, (const cplx*)d_WT, (cplx*)dz, st);
}

int32_t nep_cw_backward_error(nep_spmf* s, const double* h_cabs, const nep_cdouble* h_c, const nep_cdouble* dx,
                              const nep_cdouble* db, const nep_cdouble* dMx, const nep_cdouble* d_den_extra,
                              nep_cdouble* dr, double* h_omega, nep_stream stream) {
    ARGCHK(s && h_cabs && dx && db && dr);
    ARGCHK((dMx != nullptr) != (h_c != nullptr));     // exactly one of: M x given, or coefficients to form it
    ARGCHK(s->mt <= NEP_MAX_TERMS);
    hipStream_t st = as_stream(stream);
    const size_t mt = (size_t)s->mt;
    int rc = s->cwpart.ensure(64 + mt * 24);
    if (rc) return rc;
    unsigned long long* bits = (unsigned long long*)s->cwpart.dptr;
    double* cabs = (double*)((char*)s->cwpart.dptr + 64);
    cplx* ccf = (cplx*)((char*)s->cwpart.dptr + 64 + mt * 8);
    double stage[NEP_MAX_TERMS * 3];
    memcpy(stage, h_cabs, mt * 8);
    if (h_c) memcpy(stage + mt, h_c, mt * 16);
    rc = s->cwring.upload(cabs, stage, mt * (h_c ? 24 : 8), st);
    if (rc) return rc;
    if (h_omega) HIPCHK(hipMemsetAsync(bits, 0, 8, st));
    if (s->valbytes == 8)
        rc = launch_cw_resid<double>(s, cabs, ccf, (const cplx*)dx, (const cplx*)db, (const cplx*)dMx, (cplx*)dr, h_omega ? bits : nullptr, (const cplx*)d_den_extra, st);
    else
        rc = launch_cw_resid<cplx>(s, cabs, ccf, (const cplx*)dx, (const cplx*)db, (const cplx*)dMx, (cplx*)dr, h_omega ? bits : nullptr, (const cplx*)d_den_extra, st);
    if (rc) return rc;
    if (h_omega) {
        unsigned long long hb = 0;
        HIPCHK(hipMemcpyAsync(&hb, bits, 8, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        double v;
        memcpy(&v, &hb, 8);
        *h_omega = v;
    }
    return NEP_OK;
}

// Device-only form for callers that keep |f_t|, f_t resident (nep_iar_step): no upload, no memset, no synchronisation.
// d_bits (may be NULL) must hold 0 on entry; afterwards it holds the bit pattern of omega.  xsign = -1: dx stores -x.
int nep_cw_resid_dev(nep_spmf* s, const double* d_cabs, const nep_cdouble* d_ccf, const nep_cdouble* dx, const nep_cdouble* db,
                     nep_cdouble* dr, unsigned long long* d_bits, double xsign, hipStream_t st) {
    if (s->valbytes == 8)
        return launch_cw_resid<double>(s, d_cabs, (const cplx*)d_ccf, (const cplx*)dx, (const cplx*)db, nullptr, (cplx*)dr, d_bits, nullptr, st, xsign);
    return launch_cw_resid<cplx>(s, d_cabs, (const cplx*)d_ccf, (const cplx*)dx, (const cplx*)db, nullptr, (cplx*)dr, d_bits, nullptr, st, xsign);
}

// widest column panel whose mt x kk complex coefficient block fits 48 KiB of LDS (256 for mt <= 12)
static inline int32_t resid_panel_width(int32_t mt) { return std::min(256, std::max(1, 3072 / mt)); }

// shared body: d_out != NULL -> squared norms stay on the device (no synchronisation); else host results
static int resid_panels(nep_spmf* s, int32_t k, const nep_cdouble* hF, const nep_cdouble* dQT, int64_t ldq,
                        double* d_out, double* h_rnorm, double* h_qnorm, hipStream_t st, int64_t split_row = -1,
                        cplx* tail = nullptr, int64_t ldt = 0) {
    // panels of at most 256 Ritz vectors per pass over the matrix (fewer when the mt x kk coefficient block would not
    // fit the 48 KiB LDS budget of k_spmm_rm)
    const int32_t P = resid_panel_width(s->mt);
    for (int32_t j0 = 0; j0 < k; j0 += P) {
        const int32_t kk = std::min(P, k - j0);
        const size_t cbytes = (size_t)kk * s->mt * sizeof(cplx);
        int rc = s->coef.ensure(cbytes);
        if (rc) return rc;
        rc = s->ring.upload(s->coef.dptr, hF + (size_t)j0 * s->mt, cbytes, st);
        if (rc) return rc;
        const bool sp = use_sp_k2(s, kk, 0);
        const bool tiled = sp || use_tiles_k2(s, kk);
        int grid = tiled ? nep_tiles_nblk(s->tiles) : (int)std::min<int64_t>((s->n + 3) / 4, 2048);
        rc = s->part.ensure(((size_t)grid * 2 * kk + 2 * kk) * sizeof(double));
        if (rc) return rc;
        double* partial = (double*)s->part.dptr;
        double* outd = d_out ? d_out + 2 * (size_t)j0 : partial + (size_t)grid * 2 * kk;
        const cplx* Q = (const cplx*)dQT + j0;
        cplx* T = tail ? tail + j0 : nullptr;
        if (sp)
            rc = nep_tiles_resid_sp(s->tiles, kk, (const cplx*)s->coef.dptr, Q, ldq, 0, T, ldt, partial, split_row, st);
        else if (tiled)
            rc = nep_tiles_resid(s->tiles, kk, (const cplx*)s->coef.dptr, Q, ldq, T, ldt, partial, split_row, st);
        else if (s->valbytes == 8)
            rc = launch_spmm<double>(s, kk, (const cplx*)s->coef.dptr, Q, ldq, 0, T, ldt, partial, grid, st, split_row);
        else
            rc = launch_spmm<cplx>(s, kk, (const cplx*)s->coef.dptr, Q, ldq, 0, T, ldt, partial, grid, st, split_row);
        if (rc) return rc;
        hipLaunchKernelGGL(k_sum_partials_d, dim3(2 * kk), dim3(256), 0, st, grid, 2 * kk, partial, outd);
        LAUNCHCHK();
        if (!d_out) {
            std::vector<double> h(2 * kk);
            HIPCHK(hipMemcpyAsync(h.data(), outd, (size_t)2 * kk * sizeof(double), hipMemcpyDeviceToHost, st));
            HIPCHK(hipStreamSynchronize(st));
            for (int j = 0; j < kk; ++j) { h_rnorm[j0 + j] = sqrt(h[j]); h_qnorm[j0 + j] = sqrt(h[kk + j]); }
        }
    }
    return NEP_OK;
}

int32_t nep_resid_batch(nep_spmf* s, int32_t k, const nep_cdouble* hF, const nep_cdouble* dQT, int64_t ldq,
                        double* h_rnorm, double* h_qnorm, nep_stream stream) {
    ARGCHK(s && hF && dQT && h_rnorm && h_qnorm);
    ARGCHK(k >= 1 && ldq >= k);
    return resid_panels(s, k, hF, dQT, ldq, nullptr, h_rnorm, h_qnorm, as_stream(stream));
}

int32_t nep_resid_batch_dev(nep_spmf* s, int32_t k, const nep_cdouble* hF, const nep_cdouble* dQT, int64_t ldq,
                            double* d_out, nep_stream stream) {
    ARGCHK(s && hF && dQT && d_out);
    ARGCHK(k >= 1 && ldq >= k);
    return resid_panels(s, k, hF, dQT, ldq, d_out, nullptr, nullptr, as_stream(stream));
}

// K2 for operators with extra terms on their LAST rows (the waveguide's dense corner block on its 2 nz boundary rows): one pass
// over the matrices gives the squared column norms of the residual over the rows [0, row0) and of Q over all rows (d_out,
// 2k doubles, device), and the residual ROWS [row0, n) themselves (dRT_tail, (n - row0) x k row-major) -- the caller adds its
// extra term to that small block and its norms to d_out.  Neither the n x k residual block nor a second pass over Q touches
// HBM (nep_resid_block + two column-norm kernels moved four times the bytes of Q).
int32_t nep_resid_split_dev(nep_spmf* s, int32_t k, const nep_cdouble* hF, const nep_cdouble* dQT, int64_t ldq, int64_t row0,
                            double* d_out, nep_cdouble* dRT_tail, int64_t ldt, nep_stream stream) {
    ARGCHK(s && hF && dQT && d_out && dRT_tail);
    ARGCHK(k >= 1 && ldq >= k && ldt >= k && row0 >= 0 && row0 <= s->n);
    return resid_panels(s, k, hF, dQT, ldq, d_out, nullptr, nullptr, as_stream(stream), row0, (cplx*)dRT_tail, ldt);
}

// K2 with a COLUMN-major Ritz block (n x k, column s at dQ + s ldq; what K7 writes with y_rowmajor = 0): squared column norms
// of the residual block and of Q into d_out (2k doubles, device), no synchronisation.  Only for matrices with footprint
// tiles (nep_spmf_tile_info) and at most 4 terms -- NEP_ERR_UNSUPPORTED otherwise: the caller then uses the row-major form.
// row0 >= 0 / dR_tail: as nep_resid_split_dev, the tail block column-major ((n - row0) x k, ld ldt).
int32_t nep_resid_batch_cm_dev(nep_spmf* s, int32_t k, const nep_cdouble* hF, const nep_cdouble* dQ, int64_t ldq, int64_t row0,
                               double* d_out, nep_cdouble* dR_tail, int64_t ldt, nep_stream stream) {
    ARGCHK(s && hF && dQ && d_out);
    ARGCHK(k >= 1 && ldq >= s->n && (row0 < 0 || (dR_tail && row0 <= s->n && ldt >= s->n - row0)));
    if (!s->tiles || !nep_tiles_resid_cm_ok(s->tiles, k)) { nep_set_error("column-major K2: no footprint tiles for this matrix / k"); return NEP_ERR_UNSUPPORTED; }
    hipStream_t st = as_stream(stream);
    const size_t cbytes = (size_t)k * s->mt * sizeof(cplx);
    int rc = s->coef.ensure(cbytes);
    if (rc) return rc;
    rc = s->ring.upload(s->coef.dptr, hF, cbytes, st);
    if (rc) return rc;
    const int grid = nep_tiles_nblk(s->tiles);
    rc = s->part.ensure((size_t)grid * 2 * k * sizeof(double));
    if (rc) return rc;
    double* partial = (double*)s->part.dptr;
    // (up to two panels the older kernel -- many short-lived workgroups per CU -- hides a block's start-up better: 69 against 77 us at k = 8)
    static const int sp_cm_kmin = getenv("NEP_K2_SP_CM_KMIN") ? atoi(getenv("NEP_K2_SP_CM_KMIN")) : 9;
    if (use_sp_k2(s, k, 1) && (k >= sp_cm_kmin || g_k2_sp_mode == 2))
        rc = nep_tiles_resid_sp(s->tiles, k, (const cplx*)s->coef.dptr, (const cplx*)dQ, ldq, 1, (cplx*)dR_tail, ldt, partial, row0 < 0 ? -1 : row0, st);
    else
        rc = nep_tiles_resid_cm(s->tiles, k, (const cplx*)s->coef.dptr, (const cplx*)dQ, ldq, (cplx*)dR_tail, ldt, partial, row0 < 0 ? -1 : row0, st);
    if (rc) return rc;
    hipLaunchKernelGGL(k_sum_partials_d, dim3(2 * k), dim3(256), 0, st, grid, 2 * k, partial, d_out);
    LAUNCHCHK();
    return NEP_OK;
}

int32_t nep_resid_block(nep_spmf* s, int32_t k, const nep_cdouble* hF, const nep_cdouble* dQT, int64_t ldq,
                        nep_cdouble* dRT, int64_t ldr, nep_stream stream) {
    ARGCHK(s && hF && dQT && dRT);
    ARGCHK(k >= 1 && k <= 256 && ldq >= k && ldr >= k);
    hipStream_t st = as_stream(stream);
    const size_t cbytes = (size_t)k * s->mt * sizeof(cplx);
    int rc = s->coef.ensure(cbytes);
    if (rc) return rc;
    rc = s->ring.upload(s->coef.dptr, hF, cbytes, st);
    if (rc) return rc;
    int grid = (int)std::min<int64_t>((s->n + 3) / 4, 4096);
    const int32_t P = resid_panel_width(s->mt);
    for (int32_t j0 = 0; j0 < k; j0 += P) {
        const int32_t kk = std::min(P, k - j0);
        const cplx* F = (const cplx*)s->coef.dptr + (size_t)j0 * s->mt;
        const cplx* Q = (const cplx*)dQT + j0;
        cplx* R = (cplx*)dRT + j0;
        if (use_sp_k2(s, kk, 0)) rc = nep_tiles_resid_sp(s->tiles, kk, F, Q, ldq, 0, R, ldr, nullptr, -1, st);
        else if (use_tiles_k2(s, kk)) rc = nep_tiles_resid(s->tiles, kk, F, Q, ldq, R, ldr, nullptr, -1, st);
        else if (s->valbytes == 8) rc = launch_spmm<double>(s, kk, F, Q, ldq, 0, R, ldr, nullptr, grid, st);
        else rc = launch_spmm<cplx>(s, kk, F, Q, ldq, 0, R, ldr, nullptr, grid, st);
        if (rc) return rc;
    }
    return NEP_OK;
}

int32_t nep_spmm_terms(nep_spmf* s, int32_t p, const nep_cdouble* dXT, int64_t ldx, nep_cdouble* dZT,
                       int64_t ldz, nep_stream stream) {
    ARGCHK(s && dXT && dZT);
    ARGCHK(p >= 1 && p <= 256 && ldx >= (int64_t)p * s->mt && ldz >= p);
    hipStream_t st = as_stream(stream);
    int grid = (int)std::min<int64_t>((s->n + 3) / 4, 4096);
    if (s->valbytes == 8)
        return launch_spmm<double>(s, p, nullptr, (const cplx*)dXT, ldx, p, (cplx*)dZT, ldz, nullptr, grid, st);
    return launch_spmm<cplx>(s, p, nullptr, (const cplx*)dXT, ldx, p, (cplx*)dZT, ldz, nullptr, grid, st);
}

}  // extern "C"

// ------------------------------------------------------------------------------------------
// Rectangular CSR operator (low-rank factors of NLEIGS: UU^H is r x n, [L_1 ... L_q] is n x r;
// src/rk_helper/rk_nep.jl:128-152, used at src/method_nleigs.jl:430,464-471,480,510).
//   y = alpha * A * x + beta * z      (z may alias y; z is not read when beta == 0)
// LANES lanes cooperate on a row; rows are few or short here, so the kernel is latency-bound by design.
struct nep_csr {
    int64_t rows = 0, cols = 0, nnz = 0;
    int32_t lanes = 8;
    int32_t* d_rowptr = nullptr;
    int32_t* d_col = nullptr;
    cplx* d_val = nullptr;
};

template <int LANES>
__global__ __launch_bounds__(256) void k_csr_mv(int64_t rows, const int32_t* __restrict__ rowptr,
                                                const int32_t* __restrict__ col, const cplx* __restrict__ val,
                                                cplx alpha, const cplx* __restrict__ x, cplx beta, const cplx* z,
                                                cplx* y) {
    constexpr int RPB = 256 / LANES;
    const int lane = threadIdx.x % LANES;
    const bool use_z = (beta.x != 0.0 || beta.y != 0.0);
    for (int64_t r = blockIdx.x * (int64_t)RPB + threadIdx.x / LANES; r < rows; r += (int64_t)gridDim.x * RPB) {
        cplx acc = cmake(0.0, 0.0);
        const int32_t e0 = rowptr[r], e1 = rowptr[r + 1];
        for (int32_t e = e0 + lane; e < e1; e += LANES) cfma(acc, val[e], x[col[e]]);
        acc = group_reduce_sum<LANES>(acc);
        if (lane == 0) {
            cplx out = cmul(alpha, acc);
            if (use_z) cfma(out, beta, z[r]);
            y[r] = out;
        }
    }
}

extern "C" {

int32_t nep_csr_create(int64_t rows, int64_t cols, const int32_t* h_rowptr, const int32_t* h_colind,
                       const nep_cdouble* h_vals, nep_csr** out) {
    ARGCHK(out && h_rowptr && rows >= 1 && cols >= 1);
    *out = nullptr;
    const int64_t nnz = h_rowptr[rows];
    ARGCHK(h_rowptr[0] == 0 && nnz >= 0 && (nnz == 0 || (h_colind && h_vals)));
    for (int64_t r = 0; r < rows; ++r) ARGCHK(h_rowptr[r + 1] >= h_rowptr[r]);
    for (int64_t e = 0; e < nnz; ++e) ARGCHK(h_colind[e] >= 0 && h_colind[e] < cols);
    nep_csr* a = new nep_csr();
    a->rows = rows; a->cols = cols; a->nnz = nnz;
    const double avg = (double)nnz / (double)rows;
    a->lanes = avg >= 48 ? 64 : (avg >= 6 ? 16 : 4);
#define CSRCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { nep_set_error("%s:%d: %s: %s", __FILE__, __LINE__, #call, hipGetErrorString(e_)); nep_csr_destroy(a); return NEP_ERR_HIP; } } while (0)
    CSRCHK(hipMalloc((void**)&a->d_rowptr, (size_t)(rows + 1) * sizeof(int32_t)));
    CSRCHK(hipMalloc((void**)&a->d_col, (size_t)(nnz + 1) * sizeof(int32_t)));
    CSRCHK(hipMalloc((void**)&a->d_val, (size_t)(nnz + 1) * sizeof(cplx)));
    CSRCHK(hipMemcpy(a->d_rowptr, h_rowptr, (size_t)(rows + 1) * sizeof(int32_t), hipMemcpyHostToDevice));
    if (nnz) {
        CSRCHK(hipMemcpy(a->d_col, h_colind, (size_t)nnz * sizeof(int32_t), hipMemcpyHostToDevice));
        CSRCHK(hipMemcpy(a->d_val, h_vals, (size_t)nnz * sizeof(cplx), hipMemcpyHostToDevice));
    }
#undef CSRCHK
    *out = a;
    return NEP_OK;
}

int32_t nep_csr_destroy(nep_csr* a) {
    if (!a) return NEP_OK;
    if (a->d_rowptr) (void)hipFree(a->d_rowptr);
    if (a->d_col) (void)hipFree(a->d_col);
    if (a->d_val) (void)hipFree(a->d_val);
    delete a;
    return NEP_OK;
}

int32_t nep_csr_mv(const nep_csr* a, nep_cdouble alpha, const nep_cdouble* dx, nep_cdouble beta,
                   const nep_cdouble* dz, nep_cdouble* dy, nep_stream stream) {
    ARGCHK(a && dx && dy);
    ARGCHK((beta.re == 0.0 && beta.im == 0.0) || dz);
    hipStream_t st = as_stream(stream);
    cplx al, be;
    al.x = alpha.re; al.y = alpha.im; be.x = beta.re; be.y = beta.im;
    const int64_t rows = a->rows;
#define CSR_LAUNCH(L)                                                                                         \
    hipLaunchKernelGGL(k_csr_mv<L>, dim3((unsigned)std::min<int64_t>((rows + 256 / L - 1) / (256 / L), 8192)), \
                       dim3(256), 0, st, rows, a->d_rowptr, a->d_col, a->d_val, al, (const cplx*)dx, be,      \
                       (const cplx*)dz, (cplx*)dy)
    if (a->lanes == 64) CSR_LAUNCH(64);
    else if (a->lanes == 16) CSR_LAUNCH(16);
    else CSR_LAUNCH(4);
#undef CSR_LAUNCH
    LAUNCHCHK();
    return NEP_OK;
}

}  // extern "C"
