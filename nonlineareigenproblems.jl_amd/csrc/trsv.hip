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
//   tail            : t = c2 - L21*y1 (SpMV) ; x2 = S22^{-1} t as ONE dense GEMV.  S22^{-1} is built
//                     once per factorisation ON THE DEVICE by T independent tail solves (one
//                     workgroup per unit vector, x in LDS), turning ~2T dependent steps per solve
//                     into a single bandwidth-bound pass over 16 T^2 bytes.
// All kernels take grid.y = right-hand side index, so Beyn's n x k block solve
// (src/method_beyncontour.jl:91-93) fills the chip.
#include "common.h"
#include <vector>
#include <algorithm>
#include <chrono>

#define TRSV_WIDE_MIN 48      // a level with at least this many rows gets its own multi-WG launch
#define TRSV_TAIL_SMALL 4     // levels with <= this many rows are "chain" levels (dense tail)
#define TRSV_TAIL_MAX 4096    // cap on the dense tail size (16*T^2 bytes = 268 MB at 4096)
#define TRSV_TAIL_MIN 64

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

struct nep_lu {
    int64_t n = 0, i0 = 0, T = 0;
    TriFactor L11, U11;
    // L21 as CSR over tail rows
    int32_t* d_L21p = nullptr; int32_t* d_L21i = nullptr; cplx* d_L21x = nullptr; int64_t nnzL21 = 0;
    cplx* d_Sinv = nullptr;        // T x T row-major
    int32_t* d_perm_r = nullptr;
    int32_t* d_perm_c = nullptr;
    NepScratch work;               // (n + T) x nrhs
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
                           cplx* __restrict__ X, int64_t ldx, double scale) {
    const cplx* x = work + (int64_t)blockIdx.y * ldw;
    cplx* xo = X + (int64_t)blockIdx.y * ldx;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const cplx v = x[perm_c ? perm_c[i] : i];
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

// ---- tail: tmp = x[i0:] - L21 * x[:i0]  (wave per tail row) --------------------------------------
__global__ __launch_bounds__(256) void k_tail_spmv(int64_t T, int64_t i0, const int32_t* __restrict__ rp,
                                                   const int32_t* __restrict__ ci, const cplx* __restrict__ vx,
                                                   const cplx* __restrict__ work, int64_t ldw, cplx* __restrict__ tmp,
                                                   int64_t ldt) {
    const cplx* x = work + (int64_t)blockIdx.y * ldw;
    cplx* t = tmp + (int64_t)blockIdx.y * ldt;
    const int lane = threadIdx.x & 63;
    const int64_t r = blockIdx.x * 4LL + (threadIdx.x >> 6);
    if (r >= T) return;
    cplx acc = cmake(0.0, 0.0);
    const int e1 = rp[r + 1];
    for (int e = rp[r] + lane; e < e1; e += 64) cfma(acc, vx[e], x[ci[e]]);
    acc = group_reduce_sum<64>(acc);
    if (lane == 0) t[r] = csub(x[i0 + r], acc);
}

// ---- tail: x[i0:] = Sinv * tmp   (dense row-major GEMV, wave per row, HBM/L2-bound) --------------
__global__ __launch_bounds__(256) void k_tail_gemv(int64_t T, int64_t i0, const cplx* __restrict__ Sinv,
                                                   const cplx* __restrict__ tmp, int64_t ldt, cplx* __restrict__ work,
                                                   int64_t ldw) {
    const cplx* t = tmp + (int64_t)blockIdx.y * ldt;
    cplx* x = work + (int64_t)blockIdx.y * ldw;
    const int lane = threadIdx.x & 63;
    const int64_t r = blockIdx.x * 4LL + (threadIdx.x >> 6);
    if (r >= T) return;
    const cplx* row = Sinv + r * T;
    cplx acc = cmake(0.0, 0.0);
#pragma unroll 4
    for (int64_t c = lane; c < T; c += 64) cfma(acc, row[c], t[c]);
    acc = group_reduce_sum<64>(acc);
    if (lane == 0) x[i0 + r] = acc;
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

extern "C" {

int32_t nep_lu_destroy(nep_lu* lu) {
    if (!lu) return NEP_OK;
    free_tri(lu->L11);
    free_tri(lu->U11);
    if (lu->d_L21p) nep_pool_free(lu->d_L21p);
    if (lu->d_L21i) nep_pool_free(lu->d_L21i);
    if (lu->d_L21x) nep_pool_free(lu->d_L21x);
    if (lu->d_Sinv) nep_pool_free(lu->d_Sinv);
    if (lu->d_perm_r) nep_pool_free(lu->d_perm_r);
    if (lu->d_perm_c) nep_pool_free(lu->d_perm_c);
    if (lu->graph_exec) (void)hipGraphExecDestroy(lu->graph_exec);
    if (lu->cap_stream) (void)hipStreamDestroy(lu->cap_stream);
    lu->work.release();
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
    int rc;
    TSTAMP("validate+levels");
    // ---- heads
    int32_t nl = 0;
    compute_levels(n, hLp, hLi, false, 0, i0, level, nl);
    rc = build_tri(n, hLp, hLi, hLx, false, 0, i0, 0, i0, level, nl, lu->L11);
    if (rc) return rc;
    compute_levels(n, hUp, hUi, true, 0, i0, level, nl);
    rc = build_tri(n, hUp, hUi, hUx, true, 0, i0, 0, n, level, nl, lu->U11);   // keeps U12 entries (tail is final)
    if (rc) return rc;
    TSTAMP("heads build+upload");
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
        hipLaunchKernelGGL(k_tail_inverse, dim3((unsigned)T), dim3(512), (size_t)T * sizeof(cplx), 0, (int)T, (int)i0,
                           (const int32_t*)L22.d_levptr, L22.nlev, (const int32_t*)L22.d_rowid,
                           (const int32_t*)L22.d_rowptr, (const int32_t*)L22.d_col, (const cplx*)L22.d_val,
                           (const int32_t*)U22.d_levptr, U22.nlev, (const int32_t*)U22.d_rowid,
                           (const int32_t*)U22.d_rowptr, (const int32_t*)U22.d_col, (const cplx*)U22.d_val,
                           (const cplx*)U22.d_diag, lu->d_Sinv);
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = hipDeviceSynchronize();
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

int32_t nep_lu_create(int64_t n, const int32_t* hLp, const int32_t* hLi, const nep_cdouble* hLx, const int32_t* hUp,
                      const int32_t* hUi, const nep_cdouble* hUx, const int32_t* h_perm_r, const int32_t* h_perm_c,
                      nep_lu** out) {
    ARGCHK(out != nullptr);
    *out = nullptr;
    ARGCHK(n > 0 && n < ((int64_t)1 << 31));
    ARGCHK(hLp && hLi && hLx && hUp && hUi && hUx);
    const bool timing = getenv("NEP_TIMING") != nullptr;
    double tlast = now_ms();
    nep_lu* lu = new nep_lu();
    lu->n = n;
    lu->nnzL_in = hLp[n]; lu->nnzU_in = hUp[n];
    int rc = lu_build(lu, n, hLp, hLi, hLx, hUp, hUi, hUx);
    TSTAMP("lu_build total");
    if (rc != NEP_OK) { nep_lu_destroy(lu); return rc; }
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
    if (rc != NEP_OK) { nep_lu_destroy(lu); return rc; }
    TSTAMP("perms");
    *out = lu;
    return NEP_OK;
}

int32_t nep_lu_info(const nep_lu* lu, int64_t info[6]) {
    ARGCHK(lu && info);
    info[0] = lu->n; info[1] = lu->nnzL_in; info[2] = lu->nnzU_in;
    // levels actually traversed per solve (head levels; the dense tail counts as one step each way)
    info[3] = lu->L11.nlev + (lu->T ? 1 : 0); info[4] = lu->U11.nlev + (lu->T ? 1 : 0);
    // algorithmic bytes of one single-RHS solve: sparse heads + L21 (val 16 + idx 4) + dense tail + vectors
    info[5] = (lu->L11.nnz + lu->U11.nnz + lu->nnzL21) * 20 + 16 * lu->T * lu->T + 8 * (lu->n + 1) + 16 * lu->n +
              3 * 16 * lu->n;
    return NEP_OK;
}

/* extra introspection used by tests/bench: out[0]=tail size T, out[1]=kernel launches of the last
 * solve, out[2]=full levels(L), out[3]=full levels(U), out[4]=wide segments, out[5]=narrow segments */
int32_t nep_lu_schedule(const nep_lu* lu, int64_t out[6]) {
    ARGCHK(lu && out);
    out[0] = lu->T; out[1] = lu->launches; out[2] = lu->levL_full; out[3] = lu->levU_full;
    int64_t w = 0, nn = 0;
    for (const Seg& s : lu->L11.segs) (s.wide ? w : nn)++;
    for (const Seg& s : lu->U11.segs) (s.wide ? w : nn)++;
    out[4] = w; out[5] = nn;
    return NEP_OK;
}

// enqueues the level sweep (L head, tail, U head) on `st`
static int lu_sweep(nep_lu* lu, int nrhs, cplx* work, cplx* tmp, hipStream_t st, int* launches) {
    const int64_t n = lu->n, T = lu->T, i0 = lu->i0;
    int rc = run_head<false>(lu->L11, work, n, nrhs, st, launches);
    if (rc) return rc;
    if (T > 0) {
        hipLaunchKernelGGL(k_tail_spmv, dim3((unsigned)((T + 3) / 4), nrhs), dim3(256), 0, st, T, i0,
                           (const int32_t*)lu->d_L21p, (const int32_t*)lu->d_L21i, (const cplx*)lu->d_L21x,
                           (const cplx*)work, n, tmp, T);
        LAUNCHCHK();
        hipLaunchKernelGGL(k_tail_gemv, dim3((unsigned)((T + 3) / 4), nrhs), dim3(256), 0, st, T, i0,
                           (const cplx*)lu->d_Sinv, (const cplx*)tmp, T, work, n);
        LAUNCHCHK();
        if (launches) *launches += 2;
    }
    return run_head<true>(lu->U11, work, n, nrhs, st, launches);
}

int32_t nep_lu_solve(nep_lu* lu, int32_t nrhs, const nep_cdouble* dB, int64_t ldb, nep_cdouble* dX, int64_t ldx,
                     double scale, nep_stream stream) {
    ARGCHK(lu && dB && dX);
    ARGCHK(nrhs >= 1 && nrhs <= 65535 && ldb >= lu->n && ldx >= lu->n);
    hipStream_t st = as_stream(stream);
    const int64_t n = lu->n, T = lu->T;
    int rc = lu->work.ensure((size_t)(n + T) * nrhs * sizeof(cplx));
    if (rc) return rc;
    cplx* work = (cplx*)lu->work.dptr;
    cplx* tmp = work + (size_t)n * nrhs;
    int launches = 0;
    const int pg = (int)std::min<int64_t>((n + 255) / 256, 1024);
    hipLaunchKernelGGL(k_perm_in, dim3(pg, nrhs), dim3(256), 0, st, n, (const int32_t*)lu->d_perm_r, (const cplx*)dB, ldb,
                       work, n);
    LAUNCHCHK(); ++launches;
    const int nseg = (int)(lu->L11.segs.size() + lu->U11.segs.size());
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
                       (cplx*)dX, ldx, scale);
    LAUNCHCHK(); ++launches;
    lu->launches = launches;
    return NEP_OK;
}

}  // extern "C"
