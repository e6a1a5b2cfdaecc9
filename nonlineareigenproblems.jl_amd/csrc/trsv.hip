// libnepmi355: K5 fixed-shift solve  x = A^{-1} b  from a host-computed sparse LU (gfx950).
//
// Pr*A*Pc = L*U comes from the host (SuperLU/UMFPACK-class factorisation, one-off per shift:
// src/LinSolvers.jl:114-116).  nep_lu_create uploads the factors in "level order": rows are
// physically permuted so that the rows of one dependency level are contiguous, which removes one
// dependent load (order[] -> rowptr[]) from the critical path of the solve.
//
// The solve is a latency chain of (levels(L) + levels(U)) dependent steps with few rows each
// (SURVEY.md section 7.3-2), so it runs as ONE persistent workgroup per right-hand side: all
// levels are processed inside a single launch with a workgroup barrier between levels instead of
// thousands of kernel launches; a wave takes one row, its 64 lanes stride over the row's
// non-zeros.  Right-hand sides are independent -> grid = nrhs (Beyn's n x k block solve,
// src/method_beyncontour.jl:91-93, fills k CUs).
#include "common.h"
#include <vector>
#include <algorithm>

struct TriFactor {
    int32_t nlev = 0;
    int32_t* d_levptr = nullptr;   // nlev+1, positions into the level-ordered row slots
    int32_t* d_rowid = nullptr;    // n: original row index of slot s
    int32_t* d_rowptr = nullptr;   // n+1 over slots
    int32_t* d_col = nullptr;      // off-diagonal column indices (original numbering)
    cplx* d_val = nullptr;
    cplx* d_diag = nullptr;        // per slot (U only)
    int64_t nnz = 0;
    int32_t first_lev = 0;         // first level that needs work (L: level 0 rows have no deps)
};

struct nep_lu {
    int64_t n = 0;
    TriFactor L, U;
    int32_t* d_perm_r = nullptr;
    int32_t* d_perm_c = nullptr;
    int32_t* d_flag = nullptr;     // singular-pivot flag
    NepScratch work;               // n x nrhs work vectors
    int64_t nnzL_in = 0, nnzU_in = 0;
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

template <bool UPPER>
__device__ __forceinline__ void tri_sweep(const int32_t* __restrict__ levptr, int nlev, int first_lev,
                                          const int32_t* __restrict__ rowid,
                                          const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                          const cplx* __restrict__ val, const cplx* __restrict__ diag,
                                          cplx* x, int lane, int wv, int nw) {
    for (int lev = first_lev; lev < nlev; ++lev) {
        const int s0 = levptr[lev], s1 = levptr[lev + 1];
        for (int s = s0 + wv; s < s1; s += nw) {
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

__global__ __launch_bounds__(512) void k_lu_solve(int64_t n, const int32_t* __restrict__ perm_r,
                                                  const int32_t* __restrict__ perm_c,
                                                  // L
                                                  const int32_t* __restrict__ Llevptr, int Lnlev, int Lfirst,
                                                  const int32_t* __restrict__ Lrowid, const int32_t* __restrict__ Lrowptr,
                                                  const int32_t* __restrict__ Lcol, const cplx* __restrict__ Lval,
                                                  // U
                                                  const int32_t* __restrict__ Ulevptr, int Unlev,
                                                  const int32_t* __restrict__ Urowid, const int32_t* __restrict__ Urowptr,
                                                  const int32_t* __restrict__ Ucol, const cplx* __restrict__ Uval,
                                                  const cplx* __restrict__ Udiag,
                                                  const cplx* B, int64_t ldb, cplx* X, int64_t ldx, cplx* work,
                                                  double scale) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const cplx* b = B + (int64_t)blockIdx.x * ldb;
    cplx* xo = X + (int64_t)blockIdx.x * ldx;
    cplx* x = work + (int64_t)blockIdx.x * n;
    // c = Pr b
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) {
        const int64_t d = perm_r ? perm_r[i] : i;
        x[d] = b[i];
    }
    __syncthreads();
    tri_sweep<false>(Llevptr, Lnlev, Lfirst, Lrowid, Lrowptr, Lcol, Lval, nullptr, x, lane, wv, nw);
    tri_sweep<true>(Ulevptr, Unlev, 0, Urowid, Urowptr, Ucol, Uval, Udiag, x, lane, wv, nw);
    // x = scale * Pc y
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) {
        const int64_t sidx = perm_c ? perm_c[i] : i;
        const cplx v = x[sidx];
        xo[i] = cmake(scale * v.x, scale * v.y);
    }
}

// ------------------------------------------------------------------------------------------
static void free_tri(TriFactor& t) {
    if (t.d_levptr) (void)hipFree(t.d_levptr);
    if (t.d_rowid) (void)hipFree(t.d_rowid);
    if (t.d_rowptr) (void)hipFree(t.d_rowptr);
    if (t.d_col) (void)hipFree(t.d_col);
    if (t.d_val) (void)hipFree(t.d_val);
    if (t.d_diag) (void)hipFree(t.d_diag);
    t = TriFactor();
}

// builds the level-ordered factor on the host and uploads it
static int build_tri(int64_t n, const int32_t* P, const int32_t* I, const nep_cdouble* X, bool upper, TriFactor& out) {
    std::vector<int32_t> level(n, 0);
    int32_t nlev = 0;
    // validate + levels
    if (!upper) {
        for (int64_t i = 0; i < n; ++i) {
            int32_t lv = 0;
            for (int32_t e = P[i]; e < P[i + 1]; ++e) {
                const int32_t j = I[e];
                if (j < 0 || j >= n) { nep_set_error("L: column out of range"); return NEP_ERR_ARG; }
                if (j > i) { nep_set_error("L is not lower triangular (row %lld col %d)", (long long)i, j); return NEP_ERR_ARG; }
                if (j < i) lv = std::max(lv, level[j] + 1);
            }
            level[i] = lv;
            nlev = std::max(nlev, lv + 1);
        }
    } else {
        for (int64_t i = n - 1; i >= 0; --i) {
            int32_t lv = 0;
            for (int32_t e = P[i]; e < P[i + 1]; ++e) {
                const int32_t j = I[e];
                if (j < 0 || j >= n) { nep_set_error("U: column out of range"); return NEP_ERR_ARG; }
                if (j < i) { nep_set_error("U is not upper triangular (row %lld col %d)", (long long)i, j); return NEP_ERR_ARG; }
                if (j > i) lv = std::max(lv, level[j] + 1);
            }
            level[i] = lv;
            nlev = std::max(nlev, lv + 1);
        }
    }
    std::vector<int32_t> levptr(nlev + 1, 0);
    for (int64_t i = 0; i < n; ++i) levptr[level[i] + 1]++;
    for (int32_t l = 0; l < nlev; ++l) levptr[l + 1] += levptr[l];
    std::vector<int32_t> rowid(n), pos(levptr.begin(), levptr.end() - 1);
    for (int64_t i = 0; i < n; ++i) rowid[pos[level[i]]++] = (int32_t)i;
    std::vector<int32_t> rowptr(n + 1, 0);
    std::vector<int32_t> col;
    std::vector<nep_cdouble> val;
    std::vector<nep_cdouble> diag(upper ? n : 0);
    col.reserve(P[n]); val.reserve(P[n]);
    for (int64_t s = 0; s < n; ++s) {
        const int32_t i = rowid[s];
        bool have_diag = false;
        for (int32_t e = P[i]; e < P[i + 1]; ++e) {
            if (I[e] == i) {
                have_diag = true;
                if (upper) diag[s] = X[e];
                continue;
            }
            col.push_back(I[e]); val.push_back(X[e]);
        }
        if (upper && (!have_diag || (diag[s].re == 0.0 && diag[s].im == 0.0))) {
            nep_set_error("U has a zero pivot in row %d (matrix is singular)", i);
            return NEP_ERR_SINGULAR;
        }
        rowptr[s + 1] = (int32_t)col.size();
    }
    out.nlev = nlev;
    out.nnz = (int64_t)col.size();
    // L: rows of level 0 have no dependencies (unit diagonal) -> nothing to do
    out.first_lev = upper ? 0 : 1;
    const size_t nnz = col.size();
    HIPCHK(hipMalloc((void**)&out.d_levptr, (size_t)(nlev + 1) * 4));
    HIPCHK(hipMalloc((void**)&out.d_rowid, (size_t)n * 4));
    HIPCHK(hipMalloc((void**)&out.d_rowptr, (size_t)(n + 1) * 4));
    HIPCHK(hipMalloc((void**)&out.d_col, (nnz + 1) * 4));
    HIPCHK(hipMalloc((void**)&out.d_val, (nnz + 1) * 16));
    HIPCHK(hipMemcpy(out.d_levptr, levptr.data(), (size_t)(nlev + 1) * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(out.d_rowid, rowid.data(), (size_t)n * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(out.d_rowptr, rowptr.data(), (size_t)(n + 1) * 4, hipMemcpyHostToDevice));
    if (nnz) {
        HIPCHK(hipMemcpy(out.d_col, col.data(), nnz * 4, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(out.d_val, val.data(), nnz * 16, hipMemcpyHostToDevice));
    }
    if (upper) {
        HIPCHK(hipMalloc((void**)&out.d_diag, (size_t)n * 16));
        HIPCHK(hipMemcpy(out.d_diag, diag.data(), (size_t)n * 16, hipMemcpyHostToDevice));
    }
    return NEP_OK;
}

extern "C" {

int32_t nep_lu_destroy(nep_lu* lu) {
    if (!lu) return NEP_OK;
    free_tri(lu->L);
    free_tri(lu->U);
    if (lu->d_perm_r) (void)hipFree(lu->d_perm_r);
    if (lu->d_perm_c) (void)hipFree(lu->d_perm_c);
    if (lu->d_flag) (void)hipFree(lu->d_flag);
    lu->work.release();
    delete lu;
    return NEP_OK;
}

int32_t nep_lu_create(int64_t n, const int32_t* hLp, const int32_t* hLi, const nep_cdouble* hLx, const int32_t* hUp,
                      const int32_t* hUi, const nep_cdouble* hUx, const int32_t* h_perm_r, const int32_t* h_perm_c,
                      nep_lu** out) {
    ARGCHK(out != nullptr);
    *out = nullptr;
    ARGCHK(n > 0 && n < ((int64_t)1 << 31));
    ARGCHK(hLp && hLi && hLx && hUp && hUi && hUx);
    nep_lu* lu = new nep_lu();
    lu->n = n;
    lu->nnzL_in = hLp[n]; lu->nnzU_in = hUp[n];
    int rc = build_tri(n, hLp, hLi, hLx, false, lu->L);
    if (rc == NEP_OK) rc = build_tri(n, hUp, hUi, hUx, true, lu->U);
    if (rc != NEP_OK) { nep_lu_destroy(lu); return rc; }
    auto up_perm = [&](const int32_t* hp, int32_t** dp) -> int {
        if (!hp) return NEP_OK;
        std::vector<char> seen(n, 0);
        for (int64_t i = 0; i < n; ++i) {
            if (hp[i] < 0 || hp[i] >= n || seen[hp[i]]) { nep_set_error("invalid permutation"); return NEP_ERR_ARG; }
            seen[hp[i]] = 1;
        }
        HIPCHK(hipMalloc((void**)dp, (size_t)n * 4));
        HIPCHK(hipMemcpy(*dp, hp, (size_t)n * 4, hipMemcpyHostToDevice));
        return NEP_OK;
    };
    rc = up_perm(h_perm_r, &lu->d_perm_r);
    if (rc == NEP_OK) rc = up_perm(h_perm_c, &lu->d_perm_c);
    if (rc != NEP_OK) { nep_lu_destroy(lu); return rc; }
    *out = lu;
    return NEP_OK;
}

int32_t nep_lu_info(const nep_lu* lu, int64_t info[6]) {
    ARGCHK(lu && info);
    info[0] = lu->n; info[1] = lu->nnzL_in; info[2] = lu->nnzU_in; info[3] = lu->L.nlev; info[4] = lu->U.nlev;
    info[5] = (lu->L.nnz + lu->U.nnz) * 20 + 8 * (lu->n + 1) + 16 * lu->n + 3 * 16 * lu->n;
    return NEP_OK;
}

int32_t nep_lu_solve(nep_lu* lu, int32_t nrhs, const nep_cdouble* dB, int64_t ldb, nep_cdouble* dX, int64_t ldx,
                     double scale, nep_stream stream) {
    ARGCHK(lu && dB && dX);
    ARGCHK(nrhs >= 1 && ldb >= lu->n && ldx >= lu->n);
    hipStream_t st = as_stream(stream);
    int rc = lu->work.ensure((size_t)lu->n * nrhs * sizeof(cplx));
    if (rc) return rc;
    hipLaunchKernelGGL(k_lu_solve, dim3(nrhs), dim3(512), 0, st, lu->n, (const int32_t*)lu->d_perm_r,
                       (const int32_t*)lu->d_perm_c, (const int32_t*)lu->L.d_levptr, lu->L.nlev, lu->L.first_lev,
                       (const int32_t*)lu->L.d_rowid, (const int32_t*)lu->L.d_rowptr, (const int32_t*)lu->L.d_col,
                       (const cplx*)lu->L.d_val, (const int32_t*)lu->U.d_levptr, lu->U.nlev,
                       (const int32_t*)lu->U.d_rowid, (const int32_t*)lu->U.d_rowptr, (const int32_t*)lu->U.d_col,
                       (const cplx*)lu->U.d_val, (const cplx*)lu->U.d_diag, (const cplx*)dB, ldb, (cplx*)dX, ldx,
                       (cplx*)lu->work.dptr, scale);
    LAUNCHCHK();
    return NEP_OK;
}

}  // extern "C"
