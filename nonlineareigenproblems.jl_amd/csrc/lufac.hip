// libnepmi355: numeric LU factorisation ON THE DEVICE for a sparsity pattern whose symbolic factorisation is known.
//
// replaces: the numeric phase of `lu(A)` / `factorize` (UMFPACK) behind FactorizeLinSolver, src/LinSolvers.jl:109-122, for the
//           second and later matrices of one sparsity pattern -- the N same-pattern factorisations of
//           src/method_beyncontour.jl:89-94, the shifts of nleigs, repeated solves of one problem.  The first matrix of a
//           pattern is factorised on the host (SuperLU: ordering, pivot sequence, fill pattern); this file re-uses that
//           pivot sequence and fill pattern (static pivoting, as KLU / PARDISO refactorisation do) and computes the VALUES of
//           L and U on the GPU:  L U = Pr A Pc  with the stored patterns of L and U.
//
// Algorithm: right-looking LU in the elimination-tree block partition the solve schedule already has (trsv_ml.hip).  Every
// update  F(i,j) -= L(i,k) U(k,j)  of the factorisation is enumerated ONCE per pattern on the host (22 M products for the gun
// matrix, three int32 each); its destination pivot p = min(i,j) lies either in the block of k ("internal") or in an ancestor
// block on a higher level ("external"):
//   level l:  k_lu_ext   every entry that belongs to a pivot of level l receives the sum of its external products from the
//                        levels below, one thread per entry, products in fixed (ascending k) order -> deterministic
//             k_lu_int   one workgroup per block of the level, pivots in order: divide the column of L by the pivot, then
//                        apply the pivot's internal products (distinct destinations, no atomics)
// Blocks of a level are independent (that is what the partition guarantees), so a level is two launches.  The values land
// in arrays laid out exactly like the host factor's L and U, and the solve schedule is built from them by the same kernels
// as after a host factorisation (ml_create_from_sym).  A health word (zero / non-finite pivot, element growth) is read back
// once after the factorisation kernels; the caller falls back to the host factorisation when it is set.
#include "common.h"
#include "trsv_ml.h"
#include <vector>
#include <algorithm>
#include <chrono>
#include <thread>
#include "devprims.h"
#include <atomic>
// (the ONE-OFF enumeration of a pattern's plan needs an exclusive scan and a radix sort over its products: this library's own,
// csrc/devprims.h -- hipCUB / rocPRIM served here until round 5; tests/test_host_logic.py::test_no_vendor_blas_or_fft_behind_the_abi
// now asserts that NO vendor primitive is compiled in.)
static std::atomic<int> g_plan_threads{0};     // host enumeration threads (0: default 6); set by the host language, not via putenv
extern "C" int32_t nep_lu_set_plan_threads(int32_t n) { g_plan_threads.store(n < 0 ? 0 : (n > 32 ? 32 : n)); return NEP_OK; }
#include <cstring>
#include <math.h>

struct nep_lu;
// trsv.hip
extern "C" int32_t nep_lu_destroy(nep_lu* lu);
MLFactor* nep_lu_ml(nep_lu* lu);
nep_lu* nep_lu_wrap_ml(MLFactor* F, int64_t n, int64_t nnzL, int64_t nnzU);

struct nep_lu_refac {
    bool host_only = false;          // nep_lu_refac_analyze: S is a host-only analysis owned by this object
    MLSym* S = nullptr;
    int64_t n = 0, nnzL = 0, nnzU = 0, nnzA = 0, nprod = 0;
    int nlev = 0, nblk = 0;
    std::vector<int32_t> lev_blk;        // nlev+1
    std::vector<int64_t> ext_seg0;       // nlev+1: first external segment of a level
    std::vector<int32_t> h_blk_se;       // 2 nblk (schedule positions)
    std::vector<int64_t> h_ext_ptr;      // nseg+1 (host copy: launch shapes)
    // device, symbolic
    int32_t* d_amap = nullptr;           // nnzA: entry of A (CSC order of the caller) -> position in F
    int32_t* d_ldiag = nullptr;          // n: position of L(k,k) (unit) in F
    int32_t* d_udiag = nullptr;          // n: position of U(k,k)
    int32_t* d_Lp = nullptr;             // n+1 (CSC of L as given)
    int32_t* d_Li = nullptr;             // nnzL
    int32_t* d_oldof = nullptr;          // n: pivot at schedule position q
    int32_t* d_blk_se = nullptr;         // 2 nblk
    int64_t* d_piv_ptr = nullptr;        // n+1 (schedule order): internal products of the pivot at position q
    int32_t* d_int = nullptr;            // 3 per internal product: gL, gU, gdst
    int64_t* d_ext_ptr = nullptr;        // nseg+1
    int32_t* d_ext_dst = nullptr;        // nseg
    int32_t* d_ext_src = nullptr;        // 2 per external product: gL, gU
    int64_t nint = 0, next_ = 0, nseg = 0;
    // "wide" levels (few, large blocks: the top of the elimination tree): one launch per pivot step, all products of the
    // step's pivots (one per block of the level) spread over the whole device
    std::vector<uint8_t> wide;           // nlev
    std::vector<int64_t> wstep0;         // nlev+1: first step of a level in wide_ptr
    std::vector<int64_t> wide_ptr;       // nsteps+1 (host: launch bounds)
    int32_t* d_wide = nullptr;           // 4 per product: gL, gU, gdst, position of the pivot U(k,k)
    int64_t nwide = 0;
    // wide levels, panels of P consecutive pivots per launch (refac_build_fused): records of P operand pairs per destination,
    // the updates INTO the panel's own later rows / columns deferred to P - 1 launches at the end of the level
    int fuseP = 0;
    std::vector<int64_t> f_step0;        // nlev+1: first panel step of a level
    std::vector<int64_t> f_base;         // per panel step: first record
    std::vector<int32_t> f_cnt;          // per panel step: records
    std::vector<int64_t> fix_base;       // nlev * (P-1): first deferred product of (level, round)
    std::vector<int32_t> fix_cnt;
    int32_t* d_frec = nullptr;           // 2P + 2 per record: gL[P], gU[P], gdst, panel
    int32_t* d_fhdr = nullptr;           // LU_FUSE_HS per panel: positions of the panel's P x P diagonal block, number of pivots
    int32_t* d_ffix = nullptr;           // 4 per deferred product (the k_lu_wide form)
    double t_symbolic_ms = 0.0;
    bool gpu_enumerated = false;         // the product arrays were built by k_lu_enum_* (round 3), not by the host threads
};

// ---- kernels ---------------------------------------------------------------------------------------------------------------
// device health block per matrix: [0] = 1 when a pivot was zero / non-finite, [1] = largest |L| entry, [2] = largest |U| entry,
// [3] = largest |A| entry ([1..3] as bit patterns: non-negative doubles order like integers)
#define NEP_LU_HW 4
__device__ __forceinline__ void health_max(double* slot, double v) {
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
    if ((threadIdx.x & 63) == 0 && v > 0.0) atomicMax((unsigned long long*)slot, (unsigned long long)__double_as_longlong(v));
}
// (all kernels: blockIdx.y = matrix of a batch; the matrices share the plan and sit nF / nnzA / 3 entries apart)
__global__ void k_lu_init(int64_t nF, int64_t n, int64_t nnzA, const int32_t* __restrict__ amap, const cplx* __restrict__ Ax,
                          const int32_t* __restrict__ ldiag, cplx* __restrict__ F, double* __restrict__ health) {
    F += (int64_t)blockIdx.y * nF; Ax += (int64_t)blockIdx.y * nnzA; health += NEP_LU_HW * (int64_t)blockIdx.y;
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    // F and the health words were zeroed by memsets; scatter A, unit diagonal of L, max |A| for the growth factor
    double amax = 0.0;
    if (i < nnzA) { const cplx a = Ax[i]; F[amap[i]] = a; amax = fabs(a.x) + fabs(a.y); }
    if (i < n) F[ldiag[i]] = cmake(1.0, 0.0);
    health_max(&health[3], amax);
}

// the same with the values of matrix b assembled on the fly, A_b = sum_t Cf[b,t] A_t on the union pattern (D[e*mt + t] = value of
// term t at entry e): the B x nnz(A) value block of a contour_beyn batch is neither formed on the host nor uploaded
__global__ void k_lu_init_terms(int64_t nF, int64_t n, int64_t nnzA, const int32_t* __restrict__ amap, const cplx* __restrict__ D,
                                int mt, const cplx* __restrict__ Cf, const int32_t* __restrict__ ldiag, cplx* __restrict__ F,
                                double* __restrict__ health) {
    F += (int64_t)blockIdx.y * nF; Cf += (int64_t)blockIdx.y * mt; health += NEP_LU_HW * (int64_t)blockIdx.y;
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    double amax = 0.0;
    if (i < nnzA) {
        cplx acc = cmake(0.0, 0.0);
        for (int t = 0; t < mt; ++t) cfma(acc, D[i * mt + t], Cf[t]);
        F[amap[i]] = acc;
        amax = fabs(acc.x) + fabs(acc.y);
    }
    if (i < n) F[ldiag[i]] = cmake(1.0, 0.0);
    health_max(&health[3], amax);
}

// largest |U| entry of the finished factor (U occupies F[nnzL, nnzL + nnzU)): with max |A| from k_lu_init this is the element
// growth max|U| / max|A| of the static-pivot factorisation -- the quantity that bounds its backward error (max |L| alone does
// not: growth compounds in U)
__global__ __launch_bounds__(256) void k_lu_umax(int64_t nnzL, int64_t nnzU, const cplx* __restrict__ F, int64_t nF,
                                                 double* __restrict__ health) {
    F += (int64_t)blockIdx.y * nF; health += NEP_LU_HW * (int64_t)blockIdx.y;
    double m = 0.0;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nnzU; i += (int64_t)gridDim.x * blockDim.x) {
        const cplx u = F[nnzL + i];
        const double a = fabs(u.x) + fabs(u.y);
        m = (a == a) ? fmax(m, a) : 1.0e300;           // NaN counts as unbounded growth
    }
    health_max(&health[2], m);
}

// G lanes per destination entry of the level: F[dst] -= sum_products L * U   (sources final: lower levels are done).  Lane g
// takes products g, g + G, ...; the G partial sums are combined by a fixed shuffle tree, so the result does not depend on
// anything but G (segments of the top levels hold thousands of products: one thread per segment left most lanes idle)
template <int G>
__global__ __launch_bounds__(256) void k_lu_ext(int64_t seg0, int64_t seg1, const int64_t* __restrict__ ptr,
                                                const int32_t* __restrict__ dst, const int32_t* __restrict__ src,
                                                cplx* __restrict__ F, int64_t nF) {
    F += (int64_t)blockIdx.y * nF;
    const int sub = threadIdx.x % G;
    const int64_t sidx = seg0 + (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) / G;
    cplx acc = cmake(0.0, 0.0);
    if (sidx < seg1) {
        const int64_t p0 = ptr[sidx], p1 = ptr[sidx + 1];
        for (int64_t p = p0 + sub; p < p1; p += G) cfma(acc, F[src[2 * p]], F[src[2 * p + 1]]);
    }
    acc = group_reduce_sum<G>(acc);
    if (sidx < seg1 && sub == 0) {
        const int32_t d = dst[sidx];
        F[d] = csub(F[d], acc);
    }
}

__device__ __forceinline__ cplx lu_cdiv(cplx a, cplx b);

// wide level, one pivot step: F[dst] -= (F[gL] / pivot) * F[gU], one product per thread (distinct destinations within a pivot;
// the column of L stays unscaled until k_lu_scale at the end of the level)
__global__ __launch_bounds__(256) void k_lu_wide(int64_t t0, int64_t t1, const int4* __restrict__ prod, cplx* __restrict__ F, int64_t nF) {
    F += (int64_t)blockIdx.y * nF;
    const int64_t t = t0 + blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (t >= t1) return;
    const int4 q = prod[t];
    const cplx l = lu_cdiv(F[q.x], F[q.w]), u = F[q.y];
    cplx f = F[q.z];
    f.x -= l.x * u.x - l.y * u.y;
    f.y -= l.x * u.y + l.y * u.x;
    F[q.z] = f;
}

// end of a wide level: divide the columns of L of the level's pivots (schedule positions [q0, q1)), one wave per pivot
__global__ __launch_bounds__(256) void k_lu_scale(int q0, int q1, const int32_t* __restrict__ oldof, const int32_t* __restrict__ Lp,
                                                  const int32_t* __restrict__ Li, const int32_t* __restrict__ udiag,
                                                  cplx* __restrict__ F, double* __restrict__ health, int64_t nF) {
    F += (int64_t)blockIdx.y * nF; health += NEP_LU_HW * (int64_t)blockIdx.y;
    const int q = q0 + blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (q >= q1) return;
    const int k = oldof[q];
    const cplx piv = F[udiag[k]];
    const double ap = fabs(piv.x) + fabs(piv.y);
    if (!(ap > 0.0) || !isfinite(ap)) { if (lane == 0) health[0] = 1.0; return; }
    double maxabs = 0.0;
    for (int e = Lp[k] + lane; e < Lp[k + 1]; e += 64)
        if (Li[e] != k) {
            const cplx v = lu_cdiv(F[e], piv);
            F[e] = v;
            maxabs = fmax(maxabs, fabs(v.x) + fabs(v.y));
        }
    for (int off = 32; off > 0; off >>= 1) maxabs = fmax(maxabs, __shfl_xor(maxabs, off, 64));
    if (lane == 0 && maxabs > 0.0) atomicMax((unsigned long long*)&health[1], (unsigned long long)__double_as_longlong(maxabs));
}

__device__ __forceinline__ cplx lu_cdiv(cplx a, cplx b) {
    if (fabs(b.x) >= fabs(b.y)) {
        const double r = b.y / b.x, d = b.x + b.y * r;
        return cmake((a.x + a.y * r) / d, (a.y - a.x * r) / d);
    }
    const double r = b.x / b.y, d = b.x * r + b.y;
    return cmake((a.x * r + a.y) / d, (a.y * r - a.x) / d);
}

// ---- wide levels, P pivots per launch -------------------------------------------------------------------------------------------
// A wide level is a chain of dependent launches (one per pivot step, ~7 us each, 583 on the gun pattern: 4.1 ms of a 5 ms
// factorisation); the products of a step fill the device for a fraction of that.  P consecutive pivots k_0 .. k_{P-1} of a block
// are applied by ONE launch instead: a destination (i, j) outside the panel's rows and columns receives
//     F(i,j) -= [F(i,k_0) .. F(i,k_{P-1})]  D^{-1}  [F(k_0,j) .. F(k_{P-1},j)]^T ,      D = F(k_q, k_r)  (P x P),
// every thread eliminating the bordered (P+1) x (P+1) matrix of ITS destination from the values as they stand before the launch
// (the P x P block is eliminated redundantly by every thread: 2 P^3 / 3 flops against a launch latency).  For that the panel's
// later rows and columns (of k_1 .. k_{P-1}) must not change during the launch: the products that update them are deferred and
// applied at the end of the level in P - 1 launches of k_lu_wide (round r: the products of every panel's pivot r -- its row and
// column are final after rounds < r; no later panel reads a row or column of an earlier one).  Same operations as the
// step-by-step form, one destination's products summed before the subtraction instead of subtracted one by one.
#define LU_FUSE_MAXP 4
#define LU_FUSE_HS 20                       // header stride: P*P positions (<= 16) + [16] = number of pivots
template <int P>
__global__ __launch_bounds__(256) void k_lu_widep(int64_t t0, int64_t t1, const int32_t* __restrict__ recs,
                                                  const int32_t* __restrict__ hdrs, cplx* __restrict__ F, int64_t nF) {
    F += (int64_t)blockIdx.y * nF;
    const int64_t t = t0 + blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (t >= t1) return;
    constexpr int RS = 2 * P + 2;
    int32_t rc[RS];
    {
        const int2* rp = (const int2*)(recs + RS * t);
#pragma unroll
        for (int w = 0; w < RS / 2; ++w) { const int2 v = rp[w]; rc[2 * w] = v.x; rc[2 * w + 1] = v.y; }
    }
    const int32_t* __restrict__ h = hdrs + LU_FUSE_HS * (int64_t)rc[2 * P + 1];
    const int np = h[16];
    cplx D[P][P], lr[P], ur[P];
#pragma unroll
    for (int q = 0; q < P; ++q) {
#pragma unroll
        for (int c = 0; c < P; ++c) { const int32_t g = h[q * P + c]; D[q][c] = g >= 0 ? F[g] : cmake(0.0, 0.0); }
        lr[q] = rc[q] >= 0 ? F[rc[q]] : cmake(0.0, 0.0);
        ur[q] = rc[P + q] >= 0 ? F[rc[P + q]] : cmake(0.0, 0.0);
    }
    cplx acc = cmake(0.0, 0.0);
#pragma unroll
    for (int q = 0; q < P; ++q) {
        if (q < np) {
            const cplx pv = D[q][q];
#pragma unroll
            for (int r2 = q + 1; r2 < P; ++r2) {
                const cplx m = lu_cdiv(D[r2][q], pv);
#pragma unroll
                for (int c = q + 1; c < P; ++c) { D[r2][c].x -= m.x * D[q][c].x - m.y * D[q][c].y; D[r2][c].y -= m.x * D[q][c].y + m.y * D[q][c].x; }
                ur[r2].x -= m.x * ur[q].x - m.y * ur[q].y; ur[r2].y -= m.x * ur[q].y + m.y * ur[q].x;
            }
            const cplx ml = lu_cdiv(lr[q], pv);
#pragma unroll
            for (int c = q + 1; c < P; ++c) { lr[c].x -= ml.x * D[q][c].x - ml.y * D[q][c].y; lr[c].y -= ml.x * D[q][c].y + ml.y * D[q][c].x; }
            acc.x += ml.x * ur[q].x - ml.y * ur[q].y; acc.y += ml.x * ur[q].y + ml.y * ur[q].x;
        }
    }
    cplx f = F[rc[2 * P]];
    f.x -= acc.x; f.y -= acc.y;
    F[rc[2 * P]] = f;
}

// plan time: position of a panel pivot's U(k,k) -> 4 * panel + index in the panel
__global__ void k_fuse_pivmap(int64_t npanel, const int32_t* __restrict__ hdrs, int P, int32_t* __restrict__ pivmap) {
    const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (t >= npanel * P) return;
    const int64_t hidx = t / P; const int q = (int)(t - hidx * P);
    const int32_t* h = hdrs + LU_FUSE_HS * hidx;
    if (q < h[16]) pivmap[h[q * P + q]] = (int32_t)(hidx * 4 + q);
}
// plan time: the products [t0, t1) of one step (pivot r of its panel, every block of the level) go into the records of their
// destinations (slot[dst]: record of this panel step, handed out on first touch) or, when the destination lies in a row or column
// of a later pivot of the same panel, onto the deferred list of (level, r)
template <int P>
__global__ __launch_bounds__(256) void k_fuse_scatter(int r, int64_t t0, int64_t t1, const int4* __restrict__ prod,
                                                      const int32_t* __restrict__ pivmap, const int32_t* __restrict__ hdrs,
                                                      int32_t* __restrict__ slot, int32_t* __restrict__ cnt, int64_t recbase,
                                                      int32_t* __restrict__ recs, int32_t* __restrict__ fixcnt, int64_t fixbase,
                                                      int4* __restrict__ fix, int32_t* __restrict__ link, int64_t nF) {
    const int64_t t = t0 + blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (t >= t1) return;
    const int4 q = prod[t];
    const int32_t hidx = pivmap[q.w] >> 2;
    const int32_t* __restrict__ h = hdrs + LU_FUSE_HS * (int64_t)hidx;
    bool rowt = false, colt = false;
    for (int q2 = r + 1; q2 < P; ++q2) {
        const int32_t gl = h[q2 * P + r], gu = h[r * P + q2];        // F(k_q2, k_r): the L operand of row k_q2;  F(k_r, k_q2): the U operand
        if (gl >= 0 && q.x == gl) rowt = true;
        if (gu >= 0 && q.y == gu) colt = true;
    }
    if (rowt || colt) {
        fix[fixbase + atomicAdd(fixcnt, 1)] = q;
        // the operand chain of the panel: F(i, k_q2) is updated from F(i, k_r), F(k_q2, j) from F(k_r, j) -- k_fuse_fill follows it
        if (colt && !rowt) link[(int64_t)r * nF + q.z] = q.x;
        if (rowt && !colt) link[(int64_t)r * nF + q.z] = q.y;
        return;
    }
    int32_t sl = slot[q.z];                                          // destinations are distinct within one pivot's products
    if (sl < 0) { sl = atomicAdd(cnt, 1); slot[q.z] = sl; }
    int32_t* o = recs + (2 * P + 2) * (recbase + sl);
    o[r] = q.x; o[P + r] = q.y; o[2 * P] = q.z; o[2 * P + 1] = hidx;
}
// plan time: complete the operands of a panel step's records.  A destination (i, j) that pivot k_q updates needs F(i, k_r), r < q,
// whenever that entry feeds F(i, k_q) inside the panel -- also when k_r itself has no product for (i, j) (U(k_r, j) not stored);
// the same for the U operands.  link[r][position of F(i, k_q)] = position of F(i, k_r) comes from the deferred products.
template <int P>
__global__ __launch_bounds__(256) void k_fuse_fill(const int32_t* __restrict__ cnt, int64_t recbase, int32_t* __restrict__ recs,
                                                   const int32_t* __restrict__ link, int64_t nF) {
    const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (t >= *cnt) return;
    int32_t* o = recs + (2 * P + 2) * (recbase + t);
    for (int side = 0; side < 2; ++side) {
        int32_t* g = o + side * P;
        for (int q = P - 1; q >= 1; --q) {
            if (g[q] < 0) continue;
            for (int r = q - 1; r >= 0; --r)
                if (g[r] < 0) { const int32_t c = link[(int64_t)r * nF + g[q]]; if (c >= 0) g[r] = c; }
        }
    }
}
__global__ __launch_bounds__(256) void k_fuse_reset(int64_t t0, int64_t t1, const int4* __restrict__ prod, int32_t* __restrict__ slot) {
    const int64_t t = t0 + blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (t < t1) slot[prod[t].z] = -1;
}

// one workgroup per block of the level, pivots in schedule order
__global__ __launch_bounds__(512) void k_lu_int(int blk0, const int32_t* __restrict__ blk_se, const int32_t* __restrict__ oldof,
                                                const int32_t* __restrict__ Lp, const int32_t* __restrict__ Li,
                                                const int32_t* __restrict__ udiag, const int64_t* __restrict__ piv_ptr,
                                                const int32_t* __restrict__ tri, cplx* __restrict__ F, double* __restrict__ health,
                                                int64_t nF) {
    F += (int64_t)blockIdx.y * nF; health += NEP_LU_HW * (int64_t)blockIdx.y;
    const int b = blk0 + blockIdx.x;
    const int q0 = blk_se[2 * b], q1 = blk_se[2 * b + 1];
    double minpiv = 1.0e300, maxabs = 0.0;
    for (int q = q0; q < q1; ++q) {
        const int k = oldof[q];
        const cplx piv = F[udiag[k]];
        const double ap = fabs(piv.x) + fabs(piv.y);
        if (threadIdx.x == 0) { if (!(ap > 0.0) || !isfinite(ap)) minpiv = -1.0; else if (minpiv >= 0.0 && ap < minpiv) minpiv = ap; }
        const int e0 = Lp[k], e1 = Lp[k + 1];
        for (int e = e0 + threadIdx.x; e < e1; e += blockDim.x)
            if (Li[e] != k) {
                const cplx v = lu_cdiv(F[e], piv);
                F[e] = v;
                maxabs = fmax(maxabs, fabs(v.x) + fabs(v.y));
            }
        __syncthreads();
        const int64_t t0 = piv_ptr[q], t1 = piv_ptr[q + 1];
        for (int64_t t = t0 + threadIdx.x; t < t1; t += blockDim.x) {
            const int32_t d = tri[3 * t + 2];
            cplx f = F[d];
            const cplx l = F[tri[3 * t]], u = F[tri[3 * t + 1]];
            f.x -= l.x * u.x - l.y * u.y;
            f.y -= l.x * u.y + l.y * u.x;
            F[d] = f;
        }
        __syncthreads();
    }
    // health: [0] = 1 when a pivot was zero / non-finite, [1] = largest |L| entry (bits, non-negative doubles order like ints)
    for (int off = 32; off > 0; off >>= 1) maxabs = fmax(maxabs, __shfl_xor(maxabs, off, 64));
    if ((threadIdx.x & 63) == 0 && maxabs > 0.0)
        atomicMax((unsigned long long*)&health[1], (unsigned long long)__double_as_longlong(maxabs));
    if (threadIdx.x == 0 && minpiv < 0.0) health[0] = 1.0;
}

// ---- plan enumeration on the device (round 3) ---------------------------------------------------------------------------------
// The 22 M products of the gun plan took 0.10 s (six host threads) to 0.18 s (one) to enumerate and 24 ms to upload -- during
// which the second and third iar call of a process still factorised on the host.  On the device: one thread per product
// (pivot k by bisection of the product offsets, then (a, b) -> U entry a of row k and L entry b of column k, destination slot by
// bisection of the union column), the positions of internal and wide products from ONE exclusive scan over the product
// flags, the external products sorted by (destination segment, pivot) with a radix sort.  The arrays are bit-identical to the
// host enumeration's (nep_lu_refac_hash against nep_lu_refac_analyze; NEP_LU_PLAN_GPU=0 keeps the host path).
struct LuEnt { int32_t row; int32_t g; };
struct LuEnumArgs {
    int64_t nprod; int32_t n;
    const int64_t* pbase;        // n+1: first product of pivot k
    const int64_t* lstart;       // n: first entry of union column k below the diagonal
    const int32_t* nL;           // n: entries of L(:,k) below the diagonal
    const int64_t* urp;          // n+1: rows of U (diagonal first)
    const LuEnt* urow; const int64_t* cptr; const LuEnt* cent;
    const int32_t* blk; const int32_t* lvl; const uint8_t* wide;
};
#define LU_KIND_SHIFT 30
__device__ __forceinline__ int32_t lu_find_pivot(const int64_t* __restrict__ pbase, int32_t n, int64_t t) {
    int32_t lo = 0, hi = n;                     // largest k with pbase[k] <= t
    while (hi - lo > 1) { const int32_t mid = (lo + hi) >> 1; if (pbase[mid] <= t) lo = mid; else hi = mid; }
    return lo;
}
// code[t] = destination slot | kind << 30 (0 internal, 1 external, 2 wide); tot[g] += 1 for external products; err: 1 = no slot,
// 2 = update crosses blocks of one level (the host enumeration is re-run for the message)
__global__ void k_lu_enum_classify(LuEnumArgs A, uint32_t* __restrict__ code, int32_t* __restrict__ tot, int32_t* __restrict__ err) {
    const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (t >= A.nprod) return;
    const int32_t k = lu_find_pivot(A.pbase, A.n, t);
    const int64_t local = t - A.pbase[k];
    const int32_t nl = A.nL[k];
    const int64_t a = local / nl; const int32_t b = (int32_t)(local - a * nl);
    const LuEnt ue = A.urow[A.urp[k] + 1 + a];
    const LuEnt le = A.cent[A.lstart[k] + b];
    const int32_t j = ue.row, i = le.row;
    int64_t lo = A.cptr[j], hi = A.cptr[j + 1];
    while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (A.cent[mid].row < i) lo = mid + 1; else hi = mid; }
    if (lo >= A.cptr[j + 1] || A.cent[lo].row != i) { atomicMax(err, 1); code[t] = 3u << LU_KIND_SHIFT; return; }
    const uint32_t gd = (uint32_t)A.cent[lo].g;
    const int32_t p = i < j ? i : j;
    uint32_t kind;
    if (A.blk[p] == A.blk[k]) kind = A.wide[A.lvl[k]] ? 2u : 0u;
    else if (A.lvl[p] <= A.lvl[k]) { atomicMax(err, 2); code[t] = 3u << LU_KIND_SHIFT; return; }
    else { kind = 1u; atomicAdd(&tot[gd], 1); }
    code[t] = gd | (kind << LU_KIND_SHIFT);
}
struct LuFlagIn {                      // scan input: internal products counted in the low word, wide ones in the high word
    const uint32_t* code;
    __device__ __forceinline__ unsigned long long operator()(int64_t i) const {
        const uint32_t kind = code[i] >> LU_KIND_SHIFT;
        return kind == 0u ? 1ull : (kind == 2u ? (1ull << 32) : 0ull);
    }
};
__global__ void k_lu_enum_counts(int32_t n, const int64_t* __restrict__ pbase, const unsigned long long* __restrict__ scan,
                                 int32_t* __restrict__ cnt_int, int32_t* __restrict__ cnt_wide) {
    const int32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const unsigned long long s0 = scan[pbase[k]], s1 = scan[pbase[k + 1]];
    cnt_int[k] = (int32_t)((uint32_t)s1 - (uint32_t)s0);
    cnt_wide[k] = (int32_t)((uint32_t)(s1 >> 32) - (uint32_t)(s0 >> 32));
}
__global__ void k_lu_enum_place(LuEnumArgs A, const uint32_t* __restrict__ code, const unsigned long long* __restrict__ scan,
                                const int64_t* __restrict__ piv_ptr, const int32_t* __restrict__ newpos,
                                const int64_t* __restrict__ wide_off, const int32_t* __restrict__ udiag,
                                const int32_t* __restrict__ segid, int32_t* __restrict__ itri, int32_t* __restrict__ wflat,
                                unsigned long long* __restrict__ keys, unsigned long long* __restrict__ vals) {
    const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (t >= A.nprod) return;
    const int32_t k = lu_find_pivot(A.pbase, A.n, t);
    const int64_t t0 = A.pbase[k];
    const int64_t local = t - t0;
    const int32_t nl = A.nL[k];
    const int64_t a = local / nl; const int32_t b = (int32_t)(local - a * nl);
    const int32_t gU = A.urow[A.urp[k] + 1 + a].g;
    const int32_t gL = A.cent[A.lstart[k] + b].g;
    const uint32_t c = code[t];
    const uint32_t kind = c >> LU_KIND_SHIFT; const int32_t gd = (int32_t)(c & ((1u << LU_KIND_SHIFT) - 1u));
    const unsigned long long s = scan[t], s0 = scan[t0];
    if (kind == 0u) {
        int32_t* o = itri + 3 * (piv_ptr[newpos[k]] + (int64_t)((uint32_t)s - (uint32_t)s0));
        o[0] = gL; o[1] = gU; o[2] = gd;
    } else if (kind == 2u) {
        int32_t* o = wflat + 4 * (wide_off[k] + (int64_t)((uint32_t)(s >> 32) - (uint32_t)(s0 >> 32)));
        o[0] = gL; o[1] = gU; o[2] = gd; o[3] = udiag[k];
    } else if (kind == 1u) {
        const int64_t e = t - (int64_t)(uint32_t)s - (int64_t)(uint32_t)(s >> 32);      // rank among the external products
        keys[e] = ((unsigned long long)(uint32_t)segid[gd] << 32) | (uint32_t)k;
        vals[e] = ((unsigned long long)(uint32_t)gU << 32) | (uint32_t)gL;              // int32 view: [gL, gU]
    }
}

// ---- host: symbolic part -----------------------------------------------------------------------------------------------------
namespace {

double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// dry run (nep_lu_refac_analyze): nothing goes to the device, the arrays are folded into a hash instead -- the plan must not
// depend on the number of enumeration threads
thread_local bool g_refac_dry = false;
thread_local uint64_t g_refac_hash = 0;
thread_local hipStream_t g_refac_stream = nullptr;
template <class T>
int upv(T** d, const std::vector<T>& h) {
    if (g_refac_dry) {
        uint64_t x = g_refac_hash ^ (0x9E3779B97F4A7C15ull * (h.size() + 1));
        const unsigned char* b = (const unsigned char*)h.data();
        for (size_t i = 0; i < h.size() * sizeof(T); ++i) { x ^= b[i]; x *= 0x100000001B3ull; }
        g_refac_hash = x;
        if (getenv("NEP_REFAC_DEBUG")) fprintf(stderr, "[refac dry] array of %zu x %zu bytes -> running hash %016llx\n", h.size(), sizeof(T), (unsigned long long)x);
        *d = nullptr;
        return NEP_OK;
    }
    const size_t cnt = std::max<size_t>(h.size(), 1);
    int rc = nep_pool_alloc((void**)d, cnt * sizeof(T));
    if (rc) return rc;
    // (on the plan's own non-blocking stream when there is one: a copy on the null stream would queue behind everything the
    // caller's stream has enqueued -- an iar call runs up to 100 steps ahead of the device)
    if (!h.empty()) {
        if (g_refac_stream) { HIPCHK(hipMemcpyAsync(*d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, g_refac_stream)); HIPCHK(hipStreamSynchronize(g_refac_stream)); }
        else HIPCHK(hipMemcpy(*d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    }
    return NEP_OK;
}

struct Ent { int32_t row; int32_t g; };

// device side of the plan enumeration (kernels k_lu_enum_*): temporaries from the pool, its own non-blocking stream -- the plan is
// built on a background thread while the caller's stream runs an iar call, and neither may wait for the other
struct LuGpuEnum {
    std::vector<int64_t> pbase, lstart; std::vector<int32_t> nL;
    double t_classify = 0.0;
    hipStream_t st = nullptr;
    std::vector<void*> tmp;
    LuEnumArgs A{};
    uint32_t* d_code = nullptr; unsigned long long* d_scan = nullptr; int32_t* d_tot = nullptr;
    template <class T> int dev(T** d, size_t cnt) {
        int rc = nep_pool_alloc((void**)d, std::max<size_t>(cnt, 1) * sizeof(T));
        if (!rc) tmp.push_back((void*)*d);
        return rc;
    }
    template <class T> int up(const T** d, const T* h, size_t cnt) {
        T* p = nullptr; int rc = dev(&p, cnt); if (rc) return rc;
        if (cnt) HIPCHK(hipMemcpyAsync(p, h, cnt * sizeof(T), hipMemcpyHostToDevice, st));
        *d = p; return NEP_OK;
    }
    void release() {
        if (st) { (void)hipStreamSynchronize(st); }
        for (void* p : tmp) nep_pool_free(p);
        tmp.clear();
        if (st) { (void)hipStreamDestroy(st); st = nullptr; }
    }
    ~LuGpuEnum() { release(); }       // every exit of nep_lu_refac_create returns the stream and the pool temporaries
    LuGpuEnum() = default;
    LuGpuEnum(const LuGpuEnum&) = delete;
    LuGpuEnum& operator=(const LuGpuEnum&) = delete;
    int classify(int64_t n, int64_t nF, const std::vector<int64_t>& urp, const std::vector<Ent>& urow, const std::vector<int64_t>& cptr,
                 const std::vector<Ent>& cent, const int32_t* blk, const int32_t* lvl, const std::vector<uint8_t>& wide,
                 std::vector<int32_t>& cnt_int, std::vector<int32_t>& cnt_wide, std::vector<int32_t>& tot) {
        const int64_t nprod = pbase[n];
        if (nprod <= 0 || nprod >= ((int64_t)1 << 31) - 2) return NEP_ERR_UNSUPPORTED;
        HIPCHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        int rc;
        A.nprod = nprod; A.n = (int32_t)n;
        const LuEnt *d_urow = nullptr, *d_cent = nullptr;
        if ((rc = up(&A.pbase, pbase.data(), n + 1)) || (rc = up(&A.lstart, lstart.data(), n)) || (rc = up(&A.nL, nL.data(), n)) ||
            (rc = up(&A.urp, urp.data(), n + 1)) || (rc = up(&d_urow, (const LuEnt*)urow.data(), urow.size())) ||
            (rc = up(&A.cptr, cptr.data(), n + 1)) || (rc = up(&d_cent, (const LuEnt*)cent.data(), (size_t)cptr[n])) ||
            (rc = up(&A.blk, blk, n)) || (rc = up(&A.lvl, lvl, n)) || (rc = up(&A.wide, wide.data(), wide.size()))) return rc;
        A.urow = d_urow; A.cent = d_cent;
        int32_t *d_err = nullptr, *d_ci = nullptr, *d_cw = nullptr;
        if ((rc = dev(&d_code, (size_t)nprod + 1)) || (rc = dev(&d_scan, (size_t)nprod + 1)) || (rc = dev(&d_tot, (size_t)nF)) ||
            (rc = dev(&d_err, 1)) || (rc = dev(&d_ci, (size_t)n)) || (rc = dev(&d_cw, (size_t)n))) return rc;
        HIPCHK(hipMemsetAsync(d_tot, 0, (size_t)nF * sizeof(int32_t), st));
        HIPCHK(hipMemsetAsync(d_err, 0, sizeof(int32_t), st));
        const uint32_t endcode = 3u << LU_KIND_SHIFT;
        HIPCHK(hipMemcpyAsync(d_code + nprod, &endcode, sizeof(uint32_t), hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(k_lu_enum_classify, dim3((unsigned)((nprod + 255) / 256)), dim3(256), 0, st, A, d_code, d_tot, d_err);
        LAUNCHCHK();
        char* d_tmp = nullptr;
        if ((rc = dev(&d_tmp, nepprim::scan_temp_bytes(nprod + 1)))) return rc;
        if ((rc = nepprim::exclusive_sum_u64(LuFlagIn{(const uint32_t*)d_code}, d_scan, nprod + 1, d_tmp, st))) return rc;
        hipLaunchKernelGGL(k_lu_enum_counts, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (int32_t)n, A.pbase, d_scan, d_ci, d_cw);
        LAUNCHCHK();
        int32_t herr = 0;
        tot.resize((size_t)nF);
        HIPCHK(hipMemcpyAsync(&herr, d_err, sizeof(int32_t), hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync(cnt_int.data(), d_ci, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync(cnt_wide.data(), d_cw, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync(tot.data(), d_tot, (size_t)nF * sizeof(int32_t), hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        if (herr) { std::fill(cnt_int.begin(), cnt_int.end(), 0); std::fill(cnt_wide.begin(), cnt_wide.end(), 0); tot.clear(); return NEP_ERR_UNSUPPORTED; }
        return NEP_OK;
    }
    int place(nep_lu_refac* r, int64_t n, int64_t nF, const std::vector<int64_t>& piv_ptr, const std::vector<int32_t>& newpos,
              const std::vector<int64_t>& wide_off, const std::vector<int32_t>& udiag, const std::vector<int32_t>& segid) {
        int rc;
        const int64_t* d_piv = nullptr; const int32_t* d_newpos = nullptr; const int64_t* d_woff = nullptr;
        const int32_t* d_ud = nullptr; const int32_t* d_seg = nullptr;
        if ((rc = up(&d_piv, piv_ptr.data(), n + 1)) || (rc = up(&d_newpos, newpos.data(), n)) || (rc = up(&d_woff, wide_off.data(), n)) ||
            (rc = up(&d_ud, udiag.data(), n)) || (rc = up(&d_seg, segid.data(), (size_t)nF))) return rc;
        // the plan's own arrays (kept): internal triples, wide quadruples, external pairs
        if ((rc = nep_pool_alloc((void**)&r->d_int, std::max<size_t>((size_t)r->nint * 3, 1) * sizeof(int32_t))) ||
            (rc = nep_pool_alloc((void**)&r->d_wide, std::max<size_t>((size_t)r->nwide * 4, 1) * sizeof(int32_t))) ||
            (rc = nep_pool_alloc((void**)&r->d_ext_src, std::max<size_t>((size_t)r->next_ * 2, 1) * sizeof(int32_t)))) return rc;
        unsigned long long *d_kin = nullptr, *d_vin = nullptr, *d_kout = nullptr;
        if ((rc = dev(&d_kin, (size_t)r->next_)) || (rc = dev(&d_vin, (size_t)r->next_)) || (rc = dev(&d_kout, (size_t)r->next_))) return rc;
        hipLaunchKernelGGL(k_lu_enum_place, dim3((unsigned)((A.nprod + 255) / 256)), dim3(256), 0, st, A, (const uint32_t*)d_code,
                           (const unsigned long long*)d_scan, d_piv, d_newpos, d_woff, d_ud, d_seg, r->d_int, r->d_wide, d_kin, d_vin);
        LAUNCHCHK();
        if (r->next_ > 0) {
            int segbits = 1; while (((int64_t)1 << segbits) < (int64_t)r->nseg + 1 && segbits < 31) ++segbits;
            // stable sort of the (destination segment | slot, source pair) records by key: the sorted VALUES are the plan's array
            char* d_tmp = nullptr;
            if ((rc = dev(&d_tmp, nepprim::sort_temp_bytes(r->next_)))) return rc;
            int in0 = 1;
            if ((rc = nepprim::radix_sort_pairs_u64(d_kin, d_vin, d_kout, (unsigned long long*)r->d_ext_src, r->next_, 32 + segbits, d_tmp, st, &in0)))
                return rc;
            if (in0) HIPCHK(hipMemcpyAsync(r->d_ext_src, d_vin, (size_t)r->next_ * sizeof(unsigned long long), hipMemcpyDeviceToDevice, st));
        }
        HIPCHK(hipStreamSynchronize(st));
        return NEP_OK;
    }
};

}  // namespace

struct U64In { const unsigned long long* p; __device__ __forceinline__ unsigned long long operator()(int64_t i) const { return p[i]; } };

extern "C" {

// the library's own device-wide primitives (csrc/devprims.h) behind two plain entry points: tests and diagnostics
int32_t nep_devprim_exclusive_sum(const uint64_t* d_in, uint64_t* d_out, int64_t n, nep_stream stream) {
    ARGCHK(n >= 0 && (n == 0 || (d_in && d_out)));
    if (n == 0) return NEP_OK;
    void* tmp = nullptr;
    int rc = nep_pool_alloc(&tmp, nepprim::scan_temp_bytes(n)); if (rc) return rc;
    rc = nepprim::exclusive_sum_u64(U64In{(const unsigned long long*)d_in}, (unsigned long long*)d_out, n, tmp, as_stream(stream));
    nep_pool_free_on(tmp, as_stream(stream), true);
    return rc;
}
int32_t nep_devprim_sort_pairs(uint64_t* d_keys, uint64_t* d_vals, int64_t n, int32_t nbits, nep_stream stream) {
    ARGCHK(n >= 0 && nbits >= 0 && nbits <= 64 && (n == 0 || (d_keys && d_vals)));
    if (n <= 1) return NEP_OK;
    hipStream_t st = as_stream(stream);
    void *tmp = nullptr, *k1 = nullptr, *v1 = nullptr;
    int rc = nep_pool_alloc(&tmp, nepprim::sort_temp_bytes(n));
    if (!rc) rc = nep_pool_alloc(&k1, (size_t)n * 8);
    if (!rc) rc = nep_pool_alloc(&v1, (size_t)n * 8);
    int in0 = 1;
    if (!rc) rc = nepprim::radix_sort_pairs_u64((unsigned long long*)d_keys, (unsigned long long*)d_vals, (unsigned long long*)k1,
                                                (unsigned long long*)v1, n, nbits, tmp, st, &in0);
    if (!rc && !in0) {
        if (hipMemcpyAsync(d_keys, k1, (size_t)n * 8, hipMemcpyDeviceToDevice, st) != hipSuccess ||
            hipMemcpyAsync(d_vals, v1, (size_t)n * 8, hipMemcpyDeviceToDevice, st) != hipSuccess) { (void)hipGetLastError(); rc = NEP_ERR_HIP; }
    }
    if (tmp) nep_pool_free_on(tmp, st, true);
    if (k1) nep_pool_free_on(k1, st, true);
    if (v1) nep_pool_free_on(v1, st, true);
    return rc;
}

int32_t nep_lu_refac_destroy(nep_lu_refac* r) {
    if (!r) return NEP_OK;
    if (r->host_only) { if (r->S) ml_sym_free_host(r->S); delete r; return NEP_OK; }
    (void)hipDeviceSynchronize();
    nep_pool_free(r->d_amap); nep_pool_free(r->d_ldiag); nep_pool_free(r->d_udiag); nep_pool_free(r->d_Lp); nep_pool_free(r->d_Li);
    nep_pool_free(r->d_oldof); nep_pool_free(r->d_blk_se); nep_pool_free(r->d_piv_ptr); nep_pool_free(r->d_int);
    nep_pool_free(r->d_ext_ptr); nep_pool_free(r->d_ext_dst); nep_pool_free(r->d_ext_src); nep_pool_free(r->d_wide);
    nep_pool_free(r->d_frec); nep_pool_free(r->d_fhdr); nep_pool_free(r->d_ffix);
    if (r->S) ml_sym_release_ref(r->S);
    delete r;
    return NEP_OK;
}

// ref: a factor of this pattern created from (Lp, Li, Up, Ui, perm_r, perm_c) in CSC (nep_lu_create_csc), block schedule.
// Ap / Ai: CSC pattern of the matrices that will be factorised (caller's numbering); perm_r[i] / perm_c[j] = position of row
// i / column j of A in the factored matrix (SuperLU's convention).  NEP_ERR_UNSUPPORTED: the stored pattern is not closed
// under the elimination (an update has no slot) or the reference factor uses the level schedule.
static int32_t refac_build(nep_lu_refac* r, int64_t n, const int32_t* Lp, const int32_t* Li, const int32_t* Up, const int32_t* Ui,
                           const int32_t* perm_r, const int32_t* perm_c, const int32_t* Ap, const int32_t* Ai, nep_lu_refac** out);
int32_t nep_lu_refac_create(nep_lu* ref, int64_t n, const int32_t* Lp, const int32_t* Li, const int32_t* Up, const int32_t* Ui,
                            const int32_t* perm_r, const int32_t* perm_c, const int32_t* Ap, const int32_t* Ai,
                            nep_lu_refac** out) {
    ARGCHK(out != nullptr);
    *out = nullptr;
    ARGCHK(ref && Lp && Li && Up && Ui && perm_r && perm_c && Ap && Ai && n > 0);
    MLFactor* mf = nep_lu_ml(ref);
    if (!mf) { nep_set_error("device refactorisation needs the block schedule"); return NEP_ERR_UNSUPPORTED; }
    nep_lu_refac* r = new nep_lu_refac();
    r->S = ml_sym_acquire(mf);
    return refac_build(r, n, Lp, Li, Up, Ui, perm_r, perm_c, Ap, Ai, out);
}

// Host-only analysis of a plan (no device is touched): the symbolic partition of the factors and the complete enumeration /
// classification / placement of nep_lu_refac_create, with the arrays hashed instead of uploaded.  out[0] = products, [1]
// internal, [2] external, [3] external destination segments, [4] wide, [5] wide steps, [6] levels, [7] hash of the plan arrays
// (must not depend on NEP_LU_PLAN_THREADS).  For the sanitizer build and the thread-count invariance test.
int32_t nep_lu_refac_analyze(int64_t n, const int32_t* Lp, const int32_t* Li, const int32_t* Up, const int32_t* Ui,
                             const int32_t* perm_r, const int32_t* perm_c, const int32_t* Ap, const int32_t* Ai, int64_t out[8]) {
    ARGCHK(Lp && Li && Up && Ui && perm_r && perm_c && Ap && Ai && out && n > 0 && n < ((int64_t)1 << 31));
    ARGCHK(Lp[0] == 0 && Up[0] == 0 && Ap[0] == 0);
    for (int64_t e = 0; e < Lp[n]; ++e) ARGCHK(Li[e] >= 0 && Li[e] < n);
    for (int64_t e = 0; e < Up[n]; ++e) ARGCHK(Ui[e] >= 0 && Ui[e] < n);
    nep_lu_refac* r = new nep_lu_refac();
    r->host_only = true;
    int rc = ml_sym_build_host(n, Lp, Li, Up, Ui, perm_r, perm_c, &r->S);
    if (rc) { delete r; return rc; }
    nep_lu_refac* res = nullptr;
    g_refac_dry = true; g_refac_hash = 0xCBF29CE484222325ull;
    rc = refac_build(r, n, Lp, Li, Up, Ui, perm_r, perm_c, Ap, Ai, &res);
    g_refac_dry = false;
    if (rc) return rc;                    // (refac_build released r)
    out[0] = res->nprod; out[1] = res->nint; out[2] = res->next_; out[3] = res->nseg; out[4] = res->nwide;
    out[5] = res->wstep0[res->nlev]; out[6] = res->nlev; out[7] = (int64_t)(g_refac_hash >> 1);
    nep_lu_refac_destroy(res);
    return NEP_OK;
}

// the panel form of the wide levels (k_lu_widep), built ON THE DEVICE from the wide products as they sit there (either enumeration
// leaves them in d_wide): per panel step the products of its P steps are merged by destination.  NEP_LU_WIDE_P = 1 keeps the
// step-by-step form.  Record order within a panel step comes from an atomic counter; every record has its own destination, so
// the factor values do not depend on it.
static int32_t refac_build_fused(nep_lu_refac* r, int64_t n, int64_t nF, int nlev, const int32_t* oldof, const int32_t* blk_se,
                                 const int32_t* lev_blk, const std::vector<int64_t>& cptr, const std::vector<Ent>& cent) {
    const int P_env = getenv("NEP_LU_WIDE_P") ? atoi(getenv("NEP_LU_WIDE_P")) : 4;       // read per plan (tests build one per size)
    const int P = std::max(1, std::min(LU_FUSE_MAXP, P_env));
    if (P < 2) return NEP_OK;
    (void)n;
    const double t0 = now_ms();
    auto posof = [&](int32_t i, int32_t j) -> int32_t {              // position of F(i, j) in the union column j, -1: not stored
        const Ent* b = cent.data() + cptr[j]; const Ent* en = cent.data() + cptr[j + 1];
        const Ent* it = std::lower_bound(b, en, i, [](const Ent& a, int32_t v) { return a.row < v; });
        return (it != en && it->row == i) ? it->g : -1;
    };
    // panels (headers) and panel steps
    std::vector<int32_t> hdr;
    r->f_step0.assign(nlev + 1, 0);
    for (int l = 0; l < nlev; ++l) {
        const int64_t steps = r->wstep0[l + 1] - r->wstep0[l];
        r->f_step0[l + 1] = r->f_step0[l] + (r->wide[l] ? (steps + P - 1) / P : 0);
        if (!r->wide[l]) continue;
        for (int b = lev_blk[l]; b < lev_blk[l + 1]; ++b)
            for (int q0 = blk_se[2 * b]; q0 < blk_se[2 * b + 1]; q0 += P) {
                const int np = std::min(P, blk_se[2 * b + 1] - q0);
                const size_t o = hdr.size();
                hdr.resize(o + LU_FUSE_HS, -1);
                for (int a = 0; a < np; ++a)
                    for (int c = 0; c < np; ++c) hdr[o + a * P + c] = posof(oldof[q0 + a], oldof[q0 + c]);
                hdr[o + 16] = np;
            }
    }
    const int64_t npanel = (int64_t)(hdr.size() / LU_FUSE_HS), nps = r->f_step0[nlev];
    r->f_base.assign((size_t)nps, 0); r->f_cnt.assign((size_t)nps, 0);
    r->fix_base.assign((size_t)nlev * (P - 1), 0); r->fix_cnt.assign((size_t)nlev * (P - 1), 0);
    {
        int64_t fb = 0;
        for (int l = 0; l < nlev; ++l) {
            if (!r->wide[l]) continue;
            for (int64_t ps = r->f_step0[l]; ps < r->f_step0[l + 1]; ++ps)
                r->f_base[ps] = r->wide_ptr[r->wstep0[l] + (ps - r->f_step0[l]) * P];      // capacity: the products of its steps
            for (int rr = 0; rr + 1 < P; ++rr) {
                r->fix_base[(size_t)l * (P - 1) + rr] = fb;
                for (int64_t s = r->wstep0[l] + rr; s < r->wstep0[l + 1]; s += P) fb += r->wide_ptr[s + 1] - r->wide_ptr[s];
            }
        }
    }
    const int RS = 2 * P + 2;
    hipStream_t st = g_refac_stream;
    // (scratch of the build: released on every way out, the early returns of HIPCHK / LAUNCHCHK included)
    struct Scratch {
        int32_t *slot = nullptr, *pivmap = nullptr, *cnt = nullptr, *link = nullptr;
        ~Scratch() { nep_pool_free(slot); nep_pool_free(pivmap); nep_pool_free(cnt); nep_pool_free(link); }
    } scr;
    int32_t *&d_slot = scr.slot, *&d_pivmap = scr.pivmap, *&d_cnt = scr.cnt, *&d_link = scr.link;
    const size_t ncnt = (size_t)nps + (size_t)nlev * (P - 1);
    int rc;
    if ((rc = upv(&r->d_fhdr, hdr)) || (rc = nep_pool_alloc((void**)&r->d_frec, (size_t)r->nwide * RS * sizeof(int32_t))) ||
        (rc = nep_pool_alloc((void**)&r->d_ffix, (size_t)r->nwide * 4 * sizeof(int32_t))) ||
        (rc = nep_pool_alloc((void**)&d_slot, (size_t)nF * sizeof(int32_t))) || (rc = nep_pool_alloc((void**)&d_pivmap, (size_t)nF * sizeof(int32_t))) ||
        (rc = nep_pool_alloc((void**)&d_cnt, ncnt * sizeof(int32_t))) ||
        (rc = nep_pool_alloc((void**)&d_link, (size_t)(P - 1) * nF * sizeof(int32_t)))) return rc;
    HIPCHK(hipMemsetAsync(d_link, 0xFF, (size_t)(P - 1) * nF * sizeof(int32_t), st));
    HIPCHK(hipMemsetAsync(r->d_frec, 0xFF, (size_t)r->nwide * RS * sizeof(int32_t), st));
    HIPCHK(hipMemsetAsync(d_slot, 0xFF, (size_t)nF * sizeof(int32_t), st));
    HIPCHK(hipMemsetAsync(d_cnt, 0, ncnt * sizeof(int32_t), st));
    hipLaunchKernelGGL(k_fuse_pivmap, dim3((unsigned)((npanel * P + 255) / 256)), dim3(256), 0, st, npanel, (const int32_t*)r->d_fhdr, P, d_pivmap);
    for (int l = 0; l < nlev; ++l) {
        if (!r->wide[l]) continue;
        for (int64_t ps = r->f_step0[l]; ps < r->f_step0[l + 1]; ++ps) {
            const int64_t s0 = r->wstep0[l] + (ps - r->f_step0[l]) * P;
            for (int pass = 0; pass < 2; ++pass)
                for (int rr = 0; rr < P && s0 + rr < r->wstep0[l + 1]; ++rr) {
                    const int64_t t0p = r->wide_ptr[s0 + rr], t1p = r->wide_ptr[s0 + rr + 1];
                    if (t1p <= t0p) continue;
                    const dim3 g((unsigned)((t1p - t0p + 255) / 256));
                    if (pass) { hipLaunchKernelGGL(k_fuse_reset, g, dim3(256), 0, st, t0p, t1p, (const int4*)r->d_wide, d_slot); continue; }
                    int32_t* fc = d_cnt + nps + (size_t)l * (P - 1) + std::min(rr, P - 2);
                    const int64_t fbase = r->fix_base[(size_t)l * (P - 1) + std::min(rr, P - 2)];
#define FUSE_SCATTER(P_) hipLaunchKernelGGL((k_fuse_scatter<P_>), g, dim3(256), 0, st, rr, t0p, t1p, (const int4*)r->d_wide, (const int32_t*)d_pivmap, \
                                            (const int32_t*)r->d_fhdr, d_slot, d_cnt + ps, r->f_base[ps], r->d_frec, fc, fbase, (int4*)r->d_ffix, d_link, nF)
                    if (P == 2) FUSE_SCATTER(2); else if (P == 3) FUSE_SCATTER(3); else FUSE_SCATTER(4);
#undef FUSE_SCATTER
                }
            {
                const int64_t cap = r->wide_ptr[std::min<int64_t>(s0 + P, r->wstep0[l + 1])] - r->wide_ptr[s0];
                if (cap > 0) {
                    const dim3 g((unsigned)((cap + 255) / 256));
                    if (P == 2) hipLaunchKernelGGL((k_fuse_fill<2>), g, dim3(256), 0, st, (const int32_t*)(d_cnt + ps), r->f_base[ps], r->d_frec, (const int32_t*)d_link, nF);
                    else if (P == 3) hipLaunchKernelGGL((k_fuse_fill<3>), g, dim3(256), 0, st, (const int32_t*)(d_cnt + ps), r->f_base[ps], r->d_frec, (const int32_t*)d_link, nF);
                    else hipLaunchKernelGGL((k_fuse_fill<4>), g, dim3(256), 0, st, (const int32_t*)(d_cnt + ps), r->f_base[ps], r->d_frec, (const int32_t*)d_link, nF);
                }
            }
        }
    }
    LAUNCHCHK();
    std::vector<int32_t> hc(ncnt);
    HIPCHK(hipMemcpyAsync(hc.data(), d_cnt, ncnt * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    int64_t nrec = 0, nfix = 0;
    for (int64_t ps = 0; ps < nps; ++ps) { r->f_cnt[ps] = hc[ps]; nrec += hc[ps]; }
    for (size_t i = 0; i < (size_t)nlev * (P - 1); ++i) { r->fix_cnt[i] = hc[nps + i]; nfix += hc[nps + i]; }
    // right-sized copies (the build arrays hold one slot per product: 4 x the records at P = 4)
    {
        int32_t *d_rec2 = nullptr, *d_fix2 = nullptr;
        if ((rc = nep_pool_alloc((void**)&d_rec2, std::max<size_t>((size_t)nrec * RS, 1) * sizeof(int32_t))) ||
            (rc = nep_pool_alloc((void**)&d_fix2, std::max<size_t>((size_t)nfix * 4, 1) * sizeof(int32_t)))) { nep_pool_free(d_rec2); return rc; }
        int64_t w = 0;
        for (int64_t ps = 0; ps < nps; ++ps) {
            if (r->f_cnt[ps]) HIPCHK(hipMemcpyAsync(d_rec2 + w * RS, r->d_frec + r->f_base[ps] * RS, (size_t)r->f_cnt[ps] * RS * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
            r->f_base[ps] = w; w += r->f_cnt[ps];
        }
        w = 0;
        for (size_t i = 0; i < (size_t)nlev * (P - 1); ++i) {
            if (r->fix_cnt[i]) HIPCHK(hipMemcpyAsync(d_fix2 + w * 4, r->d_ffix + r->fix_base[i] * 4, (size_t)r->fix_cnt[i] * 4 * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
            r->fix_base[i] = w; w += r->fix_cnt[i];
        }
        HIPCHK(hipStreamSynchronize(st));
        nep_pool_free(r->d_frec); nep_pool_free(r->d_ffix);
        r->d_frec = d_rec2; r->d_ffix = d_fix2;
    }
    r->fuseP = P;
    if (getenv("NEP_TIMING"))
        fprintf(stderr, "[lu_refac] wide levels in panels of %d: %lld steps -> %lld launches, %lld products -> %lld records + %lld deferred, %.1f ms\n",
                P, (long long)r->wstep0[nlev], (long long)nps, (long long)r->nwide, (long long)nrec, (long long)nfix, now_ms() - t0);
    return NEP_OK;
}

static int32_t refac_build(nep_lu_refac* r, int64_t n, const int32_t* Lp, const int32_t* Li, const int32_t* Up, const int32_t* Ui,
                           const int32_t* perm_r, const int32_t* perm_c, const int32_t* Ap, const int32_t* Ai, nep_lu_refac** out) {
    const double t0 = now_ms();
    int64_t n2; int nlev, nblk; const int32_t *lvl, *blk, *oldof, *blk_se, *lev_blk;
    ml_sym_partition(r->S, &n2, &nlev, &nblk, &lvl, &blk, &oldof, &blk_se, &lev_blk);
    if (n2 != n) { nep_lu_refac_destroy(r); nep_set_error("refac: size mismatch"); return NEP_ERR_ARG; }
    r->n = n; r->nnzL = Lp[n]; r->nnzU = Up[n]; r->nnzA = Ap[n]; r->nlev = nlev; r->nblk = nblk;
    r->lev_blk.assign(lev_blk, lev_blk + nlev + 1);
    r->h_blk_se.assign(blk_se, blk_se + 2 * nblk);
    const int64_t nnzL = r->nnzL, nnzU = r->nnzU;
    if (nnzL + nnzU >= ((int64_t)1 << 31)) { nep_lu_refac_destroy(r); nep_set_error("refac: factors too large for 32-bit positions"); return NEP_ERR_UNSUPPORTED; }
    // row counts of U first: they give the size of the plan before anything large is built (the waveguide factors -- 126 M
    // entries, ~1e11 products -- are turned away here)
    std::vector<int64_t> urp(n + 1, 0);
    for (int64_t e = 0; e < nnzU; ++e) {
        if (Ui[e] < 0 || Ui[e] >= n) { nep_lu_refac_destroy(r); nep_set_error("refac: U row index out of range"); return NEP_ERR_ARG; }
        urp[Ui[e] + 1]++;
    }
    for (int64_t i = 0; i < n; ++i) urp[i + 1] += urp[i];
    // ---- size of the plan: sum_k |L(:,k)| |U(k,:)| products, 12-16 bytes each on the host and on the device
    {
        double est = 0.0;
        for (int64_t k = 0; k < n; ++k) est += (double)(Lp[k + 1] - Lp[k] - 1) * (double)(urp[k + 1] - urp[k] - 1);
        const char* e = getenv("NEP_LU_DEV_MAXPROD");
        const double lim = e ? atof(e) : 1.5e8;
        if (est > lim) {
            nep_lu_refac_destroy(r);
            nep_set_error("refac: %.3g products exceed the plan limit %.3g (NEP_LU_DEV_MAXPROD)", est, lim);
            return NEP_ERR_UNSUPPORTED;
        }
    }
    // ---- union columns of F, rows sorted: U part (rows <= j, g = nnzL + e) then L part (rows > j, g = e)
    std::vector<int64_t> cptr(n + 1, 0);
    std::vector<Ent> cent((size_t)(nnzL + nnzU));
    std::vector<int32_t> ldiag(n, -1), udiag(n, -1);
    {
        int64_t w = 0;
        std::vector<Ent> tmp;
        for (int64_t j = 0; j < n; ++j) {
            cptr[j] = w;
            tmp.clear();
            for (int32_t e = Up[j]; e < Up[j + 1]; ++e) {
                if (Ui[e] > j) { nep_lu_refac_destroy(r); nep_set_error("refac: U is not upper triangular"); return NEP_ERR_ARG; }
                if (Ui[e] == j) udiag[j] = (int32_t)(nnzL + e);
                tmp.push_back({Ui[e], (int32_t)(nnzL + e)});
            }
            for (int32_t e = Lp[j]; e < Lp[j + 1]; ++e) {
                if (Li[e] < j) { nep_lu_refac_destroy(r); nep_set_error("refac: L is not lower triangular"); return NEP_ERR_ARG; }
                if (Li[e] == j) { ldiag[j] = e; continue; }
                tmp.push_back({Li[e], e});
            }
            std::sort(tmp.begin(), tmp.end(), [](const Ent& a, const Ent& b) { return a.row < b.row; });
            for (const Ent& t : tmp) cent[w++] = t;
            if (udiag[j] < 0 || ldiag[j] < 0) { nep_lu_refac_destroy(r); nep_set_error("refac: missing diagonal entry in column %lld", (long long)j); return NEP_ERR_UNSUPPORTED; }
        }
        cptr[n] = w;
    }
    // rows of U (row k: columns j > k with their positions), sorted by j
    std::vector<Ent> urow((size_t)nnzU);
    {
        std::vector<int64_t> fill(urp.begin(), urp.end() - 1);
        for (int64_t j = 0; j < n; ++j)
            for (int32_t e = Up[j]; e < Up[j + 1]; ++e) urow[fill[Ui[e]]++] = {(int32_t)j, (int32_t)(nnzL + e)};   // j ascending
    }
    const double t_cols = now_ms();
    // ---- A -> F
    std::vector<int32_t> amap((size_t)r->nnzA);
    for (int64_t c = 0; c < n; ++c) {
        const int32_t j = perm_c[c];
        if (j < 0 || j >= n) { nep_lu_refac_destroy(r); nep_set_error("refac: invalid perm_c"); return NEP_ERR_ARG; }
        for (int32_t e = Ap[c]; e < Ap[c + 1]; ++e) {
            const int32_t rr = Ai[e];
            if (rr < 0 || rr >= n) { nep_lu_refac_destroy(r); nep_set_error("refac: row index out of range"); return NEP_ERR_ARG; }
            const int32_t i = perm_r[rr];
            const Ent* b = cent.data() + cptr[j]; const Ent* en = cent.data() + cptr[j + 1];
            const Ent* it = std::lower_bound(b, en, i, [](const Ent& a, int32_t v) { return a.row < v; });
            if (it == en || it->row != i) { nep_lu_refac_destroy(r); nep_set_error("refac: an entry of A has no slot in L + U"); return NEP_ERR_UNSUPPORTED; }
            amap[e] = it->g;
        }
    }
    // ---- levels processed "wide": few blocks (the top of the tree), where a block's pivots form one long sequential chain
    // with tens of thousands of products per pivot -- one workgroup cannot feed that (measured: 189 pivots 4.8 ms)
    r->wide.assign(nlev, 0);
    {
        const char* e = getenv("NEP_LU_WIDE_MAXBLK");
        const int maxblk = e ? atoi(e) : 16;      // gun: level 1 (10 blocks of up to 256 pivots) 2.5 ms in block mode, 1 ms wide
        for (int l = 1; l < nlev; ++l) r->wide[l] = (lev_blk[l + 1] - lev_blk[l]) <= maxblk ? 1 : 0;
    }
    r->wstep0.assign(nlev + 1, 0);
    for (int l = 0; l < nlev; ++l) {
        int mx = 0;
        if (r->wide[l]) for (int b = lev_blk[l]; b < lev_blk[l + 1]; ++b) mx = std::max(mx, blk_se[2 * b + 1] - blk_se[2 * b]);
        r->wstep0[l + 1] = r->wstep0[l] + mx;
    }
    // ---- enumerate the products: TWO passes over the same merges (count, then place).  Both run on worker threads over
    // contiguous ranges of k balanced by their product counts; every product has ONE final position that does not depend on the
    // thread count: internal products in pivot order, wide products by (step, block), external products grouped by
    // destination with ascending k inside a destination (thread ranges ascend, so per-thread base offsets keep that order).
    const double t_enum0 = now_ms();
    const int64_t nF = nnzL + nnzU;
    std::vector<int32_t> newpos(n);
    for (int64_t q = 0; q < n; ++q) newpos[oldof[q]] = (int32_t)q;
    std::vector<int32_t> dstpiv((size_t)nF);                  // pivot min(i, j) of every stored entry
    for (int64_t j = 0; j < n; ++j)
        for (int64_t e = cptr[j]; e < cptr[j + 1]; ++e) dstpiv[cent[e].g] = std::min<int32_t>(cent[e].row, (int32_t)j);
    for (int64_t k = 0; k < n; ++k) dstpiv[ldiag[k]] = (int32_t)k;
    int nthr = g_plan_threads.load() > 0 ? g_plan_threads.load() : 6;       // nep_lu_set_plan_threads; the variable overrides
    if (const char* e = getenv("NEP_LU_PLAN_THREADS")) nthr = std::max(1, std::min(32, atoi(e)));
    std::vector<int64_t> kcut(nthr + 1, n);
    {
        std::vector<double> w(n + 1, 0.0);
        for (int64_t k = 0; k < n; ++k)
            w[k + 1] = w[k] + 1.0 + (double)(Lp[k + 1] - Lp[k] - 1) * (double)(urp[k + 1] - urp[k] - 1);
        kcut[0] = 0;
        for (int t = 1; t < nthr; ++t) kcut[t] = std::lower_bound(w.begin(), w.end(), w[n] * t / nthr) - w.begin();
        for (int t = 1; t <= nthr; ++t) kcut[t] = std::max(kcut[t], kcut[t - 1]);
        kcut[nthr] = n;
    }
    struct Err { int code = 0; int32_t i = 0, j = 0; int64_t k = 0; };
    std::vector<Err> errs(nthr);
    // fn(kind, k, gL, gU, gdst): kind 0 internal, 1 external, 2 wide
    auto visit = [&](int tix, auto&& fn) {
        Err& E = errs[tix];
        std::vector<Ent> Lk;
        for (int64_t k = kcut[tix]; k < kcut[tix + 1] && !E.code; ++k) {
            Lk.clear();
            {
                const Ent* b = cent.data() + cptr[k]; const Ent* en = cent.data() + cptr[k + 1];
                const Ent* it = std::upper_bound(b, en, (int32_t)k, [](int32_t v, const Ent& a) { return v < a.row; });
                for (; it != en; ++it) Lk.push_back(*it);
            }
            if (Lk.empty()) continue;
            const bool kwide = r->wide[lvl[k]] != 0;
            for (int64_t ue = urp[k]; ue < urp[k + 1] && !E.code; ++ue) {
                const int32_t j = urow[ue].row, gU = urow[ue].g;
                if (j <= k) continue;
                // destinations (i, j), i in Lk: merge with the union column j (rows > k)
                const Ent* b = cent.data() + cptr[j]; const Ent* en = cent.data() + cptr[j + 1];
                const Ent* it = std::upper_bound(b, en, (int32_t)k, [](int32_t v, const Ent& a) { return v < a.row; });
                for (const Ent& le : Lk) {
                    while (it != en && it->row < le.row) ++it;
                    if (it == en || it->row != le.row) { E.code = 1; E.i = le.row; E.j = j; E.k = k; break; }
                    const int32_t p = std::min(le.row, j);
                    // (pivots of different blocks of a wide level run in the same launch and may share a destination in an
                    // ancestor block: those products stay "external", summed per destination in fixed order)
                    if (blk[p] == blk[k]) fn(kwide ? 2 : 0, k, le.g, gU, it->g);
                    else if (lvl[p] <= lvl[k]) { E.code = 2; break; }
                    else fn(1, k, le.g, gU, it->g);
                }
            }
        }
    };
    auto run_threads = [&](auto&& body) {
        std::vector<std::thread> th;
        for (int t = 1; t < nthr; ++t) th.emplace_back(body, t);
        body(0);
        for (std::thread& t : th) t.join();
    };
    // pass 1: counts
    std::vector<int32_t> cnt_int(n, 0), cnt_wide(n, 0);
    std::vector<std::vector<int32_t>> cnt_ext(nthr);
    std::vector<int32_t> tot;                                  // external products per destination slot
    // ---- the same on the device (see k_lu_enum_classify): host path when switched off, in the dry run, or when it fails
    static const bool gpu_on = !(getenv("NEP_LU_PLAN_GPU") && atoi(getenv("NEP_LU_PLAN_GPU")) == 0);
    bool gpu = gpu_on && !g_refac_dry && nF < ((int64_t)1 << LU_KIND_SHIFT);
    LuGpuEnum G;
    if (gpu) {
        G.pbase.assign(n + 1, 0); G.lstart.assign(n, 0); G.nL.assign(n, 0);
        for (int64_t k = 0; k < n; ++k) {
            const Ent* b = cent.data() + cptr[k]; const Ent* en = cent.data() + cptr[k + 1];
            const Ent* it = std::upper_bound(b, en, (int32_t)k, [](int32_t v, const Ent& a) { return v < a.row; });
            G.lstart[k] = it - cent.data(); G.nL[k] = (int32_t)(en - it);
            // (row k of U: the diagonal entry first, the urp[k+1] - urp[k] - 1 entries right of it are the U operands)
            G.pbase[k + 1] = G.pbase[k] + (int64_t)G.nL[k] * (urp[k + 1] - urp[k] - 1);
        }
        const double tg0 = now_ms();
        const int rcg = G.classify(n, nF, urp, urow, cptr, cent, blk, lvl, r->wide, cnt_int, cnt_wide, tot);
        G.t_classify = now_ms() - tg0;
        if (rcg) {                                             // (pattern errors are diagnosed by the host enumeration below)
            G.release(); gpu = false;
            // a failure behind the asynchronous copies may have left the count arrays partly written: the host path starts clean
            std::fill(cnt_int.begin(), cnt_int.end(), 0); std::fill(cnt_wide.begin(), cnt_wide.end(), 0); std::fill(tot.begin(), tot.end(), 0);
        }
    }
    if (!gpu) run_threads([&](int tix) {
        std::vector<int32_t>& ce = cnt_ext[tix];
        ce.assign((size_t)nF, 0);
        visit(tix, [&](int kind, int64_t k, int32_t, int32_t, int32_t gd) {
            if (kind == 0) ++cnt_int[k]; else if (kind == 2) ++cnt_wide[k]; else ++ce[gd];
        });
    });
    for (int t = 0; t < nthr; ++t) {
        if (errs[t].code == 1) {
            nep_set_error("refac: the stored pattern is not closed under the elimination (update (%d,%d) from pivot %lld has no slot)", errs[t].i, errs[t].j, (long long)errs[t].k);
            nep_lu_refac_destroy(r);
            return NEP_ERR_UNSUPPORTED;
        }
        if (errs[t].code == 2) { nep_set_error("refac: update crosses blocks of one level"); nep_lu_refac_destroy(r); return NEP_ERR_UNSUPPORTED; }
    }
    // internal: schedule order
    std::vector<int64_t> piv_ptr(n + 1, 0);
    for (int64_t q = 0; q < n; ++q) piv_ptr[q + 1] = piv_ptr[q] + cnt_int[oldof[q]];
    r->nint = piv_ptr[n];
    // wide: by step, blocks of a step in schedule order
    const int64_t nsteps = r->wstep0[nlev];
    r->wide_ptr.assign((size_t)nsteps + 1, 0);
    std::vector<int64_t> wide_off(n, 0);
    {
        auto stepof = [&](int64_t k) { return r->wstep0[lvl[k]] + (newpos[k] - blk_se[2 * blk[k]]); };
        for (int64_t q = 0; q < n; ++q) { const int64_t k = oldof[q]; if (cnt_wide[k]) r->wide_ptr[stepof(k) + 1] += cnt_wide[k]; }
        for (int64_t sidx = 0; sidx < nsteps; ++sidx) r->wide_ptr[sidx + 1] += r->wide_ptr[sidx];
        std::vector<int64_t> cur(r->wide_ptr.begin(), r->wide_ptr.end() - 1);
        for (int64_t q = 0; q < n; ++q) { const int64_t k = oldof[q]; if (cnt_wide[k]) { wide_off[k] = cur[stepof(k)]; cur[stepof(k)] += cnt_wide[k]; } }
    }
    r->nwide = r->wide_ptr[nsteps];
    // external: segments by (destination level, destination), per-thread cursors inside a segment
    std::vector<int64_t> ext_ptr(1, 0);
    std::vector<int32_t> ext_dst;
    r->ext_seg0.assign(nlev + 1, 0);
    {
        if (!gpu) {
            tot.assign((size_t)nF, 0);
            for (int t = 0; t < nthr; ++t) { const int32_t* ce = cnt_ext[t].data(); for (int64_t g = 0; g < nF; ++g) tot[g] += ce[g]; }
        }
        std::vector<int64_t> start((size_t)nF, 0);
        int64_t run = 0;
        for (int l = 0; l < nlev; ++l) {
            r->ext_seg0[l] = (int64_t)ext_dst.size();
            for (int64_t g = 0; g < nF; ++g)
                if (tot[g] > 0 && lvl[dstpiv[g]] == l) { start[g] = run; ext_dst.push_back((int32_t)g); run += tot[g]; ext_ptr.push_back(run); }
        }
        r->ext_seg0[nlev] = (int64_t)ext_dst.size();
        if (run >= ((int64_t)1 << 31)) { nep_lu_refac_destroy(r); nep_set_error("refac: too many external products"); return NEP_ERR_UNSUPPORTED; }
        if (!gpu) for (int64_t g = 0; g < nF; ++g) {
            if (!tot[g]) continue;
            int64_t c = start[g];
            for (int t = 0; t < nthr; ++t) { const int32_t m = cnt_ext[t][g]; cnt_ext[t][g] = (int32_t)c; c += m; }
        }
        r->next_ = run;
    }
    r->nseg = (int64_t)ext_dst.size();
    r->h_ext_ptr = ext_ptr;
    r->nprod = r->nint + r->nwide + r->next_;
    // pass 2: placement
    std::vector<int32_t> itri, ext_src, wflat;
    if (gpu) {
        // destination slot -> segment index, then placement + sort on the device: the three big arrays never exist on the host
        std::vector<int32_t> segid((size_t)nF, -1);
        for (int64_t sg = 0; sg < (int64_t)ext_dst.size(); ++sg) segid[ext_dst[sg]] = (int32_t)sg;
        const double tg1 = now_ms();
        const int rcg = G.place(r, n, nF, piv_ptr, newpos, wide_off, udiag, segid);
        G.release();
        if (getenv("NEP_TIMING")) fprintf(stderr, "[lu_refac] device enumeration: offsets + upload + classify + scan + counts back %.1f ms, placement + sort %.1f ms\n", G.t_classify, now_ms() - tg1);
        if (rcg) { nep_lu_refac_destroy(r); return rcg; }
    } else {
    itri.resize((size_t)r->nint * 3); ext_src.resize((size_t)r->next_ * 2); wflat.resize((size_t)r->nwide * 4);
    run_threads([&](int tix) {
        int32_t* ce = cnt_ext[tix].data();
        int64_t curk = -1, ci = 0, cw = 0;
        visit(tix, [&](int kind, int64_t k, int32_t gL, int32_t gU, int32_t gd) {
            if (k != curk) { curk = k; ci = piv_ptr[newpos[k]]; cw = wide_off[k]; }
            if (kind == 0) { int32_t* o = itri.data() + 3 * ci++; o[0] = gL; o[1] = gU; o[2] = gd; }
            else if (kind == 2) { int32_t* o = wflat.data() + 4 * cw++; o[0] = gL; o[1] = gU; o[2] = gd; o[3] = udiag[k]; }
            else { const int64_t pos = ce[gd]++; ext_src[2 * pos] = gL; ext_src[2 * pos + 1] = gU; }
        });
    });
    }
    { std::vector<std::vector<int32_t>>().swap(cnt_ext); }
    const double t_enum1 = now_ms();
    const double t_group = now_ms();
    // ---- upload
    int rc;
    struct UpStream {
        UpStream(bool on) { if (on && hipStreamCreateWithFlags(&g_refac_stream, hipStreamNonBlocking) != hipSuccess) g_refac_stream = nullptr; }
        ~UpStream() { if (g_refac_stream) { (void)hipStreamDestroy(g_refac_stream); g_refac_stream = nullptr; } }
    } up_stream(!g_refac_dry);
    std::vector<int32_t> vLp(Lp, Lp + n + 1), vLi(Li, Li + nnzL), vold(oldof, oldof + n), vse(blk_se, blk_se + 2 * nblk);
    if ((rc = upv(&r->d_amap, amap)) || (rc = upv(&r->d_ldiag, ldiag)) || (rc = upv(&r->d_udiag, udiag)) || (rc = upv(&r->d_Lp, vLp)) ||
        (rc = upv(&r->d_Li, vLi)) || (rc = upv(&r->d_oldof, vold)) || (rc = upv(&r->d_blk_se, vse)) || (rc = upv(&r->d_piv_ptr, piv_ptr)) ||
        (!gpu && (rc = upv(&r->d_int, itri))) || (rc = upv(&r->d_ext_ptr, ext_ptr)) || (rc = upv(&r->d_ext_dst, ext_dst)) ||
        (!gpu && (rc = upv(&r->d_ext_src, ext_src))) || (!gpu && (rc = upv(&r->d_wide, wflat)))) { nep_lu_refac_destroy(r); return rc; }
    r->gpu_enumerated = gpu;
    if (!g_refac_dry && r->nwide > 0) {
        const int rcf = refac_build_fused(r, n, nF, nlev, oldof, blk_se, lev_blk, cptr, cent);
        if (rcf) { nep_lu_refac_destroy(r); return rcf; }
    }
    r->t_symbolic_ms = now_ms() - t0;
    if (getenv("NEP_TIMING"))
        fprintf(stderr, "[lu_refac] columns %.1f ms, A map + weights %.1f, enumeration %.1f (%s, %d host threads), grouping %.1f, upload %.1f\n",
                t_cols - t0, t_enum0 - t_cols, t_enum1 - t_enum0, gpu ? "device" : "host", nthr, t_group - t_enum1, now_ms() - t_group);
    if (getenv("NEP_TIMING"))
        fprintf(stderr, "[lu_refac] n=%lld products %lld (internal %lld, external %lld in %lld segments, wide %lld in %lld steps), symbolic %.1f ms\n",
                (long long)n, (long long)r->nprod, (long long)r->nint, (long long)r->next_, (long long)r->nseg, (long long)r->nwide,
                (long long)r->wstep0[nlev], r->t_symbolic_ms);
    *out = r;
    return NEP_OK;
}

int32_t nep_lu_factor_dev_batch(nep_lu_refac* r, int32_t B, const nep_cdouble* h_Ax, int32_t expected_solves, double growth_limit,
                                double* h_health, nep_cdouble* h_LUx_out, nep_lu** out, nep_stream stream);

// out[0] = the hash nep_lu_refac_analyze puts in its out[7], computed from the plan arrays AS THEY SIT ON THE DEVICE (downloaded);
// out[1] = 1 when the product arrays were enumerated on the device.  Test: a device-built plan equals the host enumeration bit for bit.
int32_t nep_lu_refac_hash(const nep_lu_refac* r, int64_t out[2]) {
    ARGCHK(r && out && !r->host_only);
    HIPCHK(hipDeviceSynchronize());
    uint64_t x = 0xCBF29CE484222325ull;
    std::vector<unsigned char> buf;
    auto fold = [&](const void* d, size_t cnt, size_t esz) -> int {
        buf.resize(cnt * esz);
        if (cnt) HIPCHK(hipMemcpy(buf.data(), d, cnt * esz, hipMemcpyDeviceToHost));
        x ^= 0x9E3779B97F4A7C15ull * (cnt + 1);
        for (size_t i = 0; i < cnt * esz; ++i) { x ^= buf[i]; x *= 0x100000001B3ull; }
        return NEP_OK;
    };
    int rc;
    if ((rc = fold(r->d_amap, (size_t)r->nnzA, 4)) || (rc = fold(r->d_ldiag, (size_t)r->n, 4)) || (rc = fold(r->d_udiag, (size_t)r->n, 4)) ||
        (rc = fold(r->d_Lp, (size_t)r->n + 1, 4)) || (rc = fold(r->d_Li, (size_t)r->nnzL, 4)) || (rc = fold(r->d_oldof, (size_t)r->n, 4)) ||
        (rc = fold(r->d_blk_se, (size_t)2 * r->nblk, 4)) || (rc = fold(r->d_piv_ptr, (size_t)r->n + 1, 8)) ||
        (rc = fold(r->d_int, (size_t)r->nint * 3, 4)) || (rc = fold(r->d_ext_ptr, (size_t)r->nseg + 1, 8)) ||
        (rc = fold(r->d_ext_dst, (size_t)r->nseg, 4)) || (rc = fold(r->d_ext_src, (size_t)r->next_ * 2, 4)) ||
        (rc = fold(r->d_wide, (size_t)r->nwide * 4, 4))) return rc;
    out[0] = (int64_t)(x >> 1);
    out[1] = r->gpu_enumerated ? 1 : 0;
    return NEP_OK;
}

int32_t nep_lu_refac_info(const nep_lu_refac* r, int64_t out[6]) {
    ARGCHK(r && out);
    out[0] = r->n; out[1] = r->nprod; out[2] = r->nint; out[3] = r->next_; out[4] = r->nseg; out[5] = (int64_t)r->t_symbolic_ms;
    return NEP_OK;
}

// the wide levels of the plan: out[0] = pivots per launch (1: step by step), [1] = pivot steps, [2] = launches per factorisation
// (panel steps + deferred rounds), [3] = destination records, [4] = deferred products
int32_t nep_lu_refac_wide_info(const nep_lu_refac* r, int64_t out[5]) {
    ARGCHK(r && out);
    const int64_t steps = r->wstep0.empty() ? 0 : r->wstep0.back();
    out[0] = r->fuseP >= 2 ? r->fuseP : 1; out[1] = steps; out[2] = steps; out[3] = 0; out[4] = 0;
    if (r->fuseP >= 2) {
        int64_t launches = 0, rec = 0, fix = 0;
        for (int32_t c : r->f_cnt) { launches += c > 0; rec += c; }
        for (int32_t c : r->fix_cnt) { launches += c > 0; fix += c; }
        out[2] = launches; out[3] = rec; out[4] = fix;
    }
    return NEP_OK;
}

// h_Ax: the nnzA values of the new matrix in the CSC order of (Ap, Ai).  h_health[3] (may be NULL): [0] = 1 when a pivot was
// zero or non-finite, [1] = largest |Re| + |Im| over the entries of L (1-ish for a diagonally pivoted factor), [2] = element
// growth max|U| / max|A|; the factor is refused when [1] or [2] exceeds growth_limit.
// h_LUx_out (may be NULL): receives nnzL + nnzU values (L then U, input entry order) -- tests compare them with the host factor.
// NEP_ERR_SINGULAR when a pivot broke down (nothing is returned then).
int32_t nep_lu_factor_dev(nep_lu_refac* r, const nep_cdouble* h_Ax, int32_t expected_solves, double growth_limit,
                          double* h_health, nep_cdouble* h_LUx_out, nep_lu** out, nep_stream stream) {
    ARGCHK(r && h_Ax && out);
    *out = nullptr;
    double hh[3] = {0.0, 0.0, 0.0};
    int32_t rc = nep_lu_factor_dev_batch(r, 1, h_Ax, expected_solves, growth_limit, hh, h_LUx_out, out, stream);
    if (h_health) { h_health[0] = hh[0]; h_health[1] = hh[1]; h_health[2] = hh[2]; }
    if (rc) return rc;
    if (!*out) {
        nep_set_error("device refactorisation: pivot breakdown or growth (max|L| %.3g, max|U|/max|A| %.3g) above %.3g with the stored pivot sequence", hh[1], hh[2], growth_limit);
        return NEP_ERR_SINGULAR;
    }
    return NEP_OK;
}

// B matrices of the plan's pattern in ONE pass (the quadrature nodes of contour_beyn, src/method_beyncontour.jl:89-94): every
// launch of the factorisation carries all B matrices, so the 64 nodes of config C4 cost about as many launches as one.
// h_Ax: B x nnzA values; h_health: B x 3 (required); out[b] = NULL for a matrix whose factorisation was refused (pivot
// breakdown / growth): the caller factorises that one on the host.  h_LUx_out (may be NULL): B x (nnzL + nnzU).
static int32_t lu_factor_batch_impl(nep_lu_refac* r, int32_t B, const nep_cdouble* h_Ax, const nep_cdouble* d_D, int32_t mt,
                                    const nep_cdouble* h_Cf, int32_t expected_solves, double growth_limit, double* h_health,
                                    nep_cdouble* h_LUx_out, nep_lu** out, nep_stream stream);
int32_t nep_lu_factor_dev_batch(nep_lu_refac* r, int32_t B, const nep_cdouble* h_Ax, int32_t expected_solves, double growth_limit,
                                double* h_health, nep_cdouble* h_LUx_out, nep_lu** out, nep_stream stream) {
    ARGCHK(r && h_Ax && out && h_health && B >= 1);
    return lu_factor_batch_impl(r, B, h_Ax, nullptr, 0, nullptr, expected_solves, growth_limit, h_health, h_LUx_out, out, stream);
}
// The same for matrices given as combinations of `mt` terms on the plan's pattern, A_b = sum_t h_Cf[b*mt + t] A_t: d_D (DEVICE,
// nnz(A) x mt, entry-major) holds the term values scattered onto the union pattern -- uploaded once per NEP -- and the values of
// the B matrices are formed inside the scatter kernel (src/method_beyncontour.jl:89-94: M(lam_b) = sum_t f_t(lam_b) A_t)
int32_t nep_lu_factor_dev_batch_terms(nep_lu_refac* r, int32_t B, const nep_cdouble* d_D, int32_t mt, const nep_cdouble* h_Cf,
                                      int32_t expected_solves, double growth_limit, double* h_health, nep_lu** out,
                                      nep_stream stream) {
    ARGCHK(r && d_D && h_Cf && out && h_health && B >= 1 && mt >= 1);
    return lu_factor_batch_impl(r, B, nullptr, d_D, mt, h_Cf, expected_solves, growth_limit, h_health, nullptr, out, stream);
}
static int32_t lu_factor_batch_impl(nep_lu_refac* r, int32_t B, const nep_cdouble* h_Ax, const nep_cdouble* d_D, int32_t mt,
                                    const nep_cdouble* h_Cf, int32_t expected_solves, double growth_limit, double* h_health,
                                    nep_cdouble* h_LUx_out, nep_lu** out, nep_stream stream) {
    for (int b = 0; b < B; ++b) out[b] = nullptr;
    hipStream_t st = as_stream(stream);
    const int64_t nF = r->nnzL + r->nnzU;
    cplx* dF = nullptr; cplx* dA = nullptr; double* dH = nullptr;
    int rc;
    if ((rc = nep_pool_alloc((void**)&dF, (size_t)B * nF * sizeof(cplx) + (size_t)B * NEP_LU_HW * 8 + 64))) return rc;
    // dA: the B x nnz(A) values, or (terms form) the B x mt coefficients
    const size_t a_bytes = h_Ax ? (size_t)B * r->nnzA * sizeof(cplx) : (size_t)B * mt * sizeof(cplx);
    if ((rc = nep_pool_alloc((void**)&dA, a_bytes + 64))) { nep_pool_free(dF); return rc; }
    dH = (double*)((char*)dF + (size_t)B * nF * sizeof(cplx));
    auto fail = [&](int code) { nep_pool_free_on(dF, st, true); nep_pool_free_on(dA, st, true); return code; };
    static thread_local PinnedRing ring;
    {   // in pieces of at most 8 MiB: the ring's pinned slots stay small (a 90 MB slot for 64 nodes costs ~30 ms to pin)
        const char* srcp = h_Ax ? (const char*)h_Ax : (const char*)h_Cf;
        const size_t total = a_bytes, piece = (size_t)8 << 20;
        for (size_t off = 0; off < total; off += piece)
            if ((rc = ring.upload((char*)dA + off, srcp + off, std::min(piece, total - off), st))) return fail(rc);
    }
    HIPCHK(hipMemsetAsync(dF, 0, (size_t)B * nF * sizeof(cplx) + (size_t)B * NEP_LU_HW * 8, st));      // factor values + health words
    const unsigned gy = (unsigned)B;
    {
        const int64_t m = std::max<int64_t>(r->nnzA, r->n);
        if (h_Ax)
            hipLaunchKernelGGL(k_lu_init, dim3((unsigned)((m + 255) / 256), gy), dim3(256), 0, st, nF, r->n, r->nnzA, (const int32_t*)r->d_amap,
                               (const cplx*)dA, (const int32_t*)r->d_ldiag, dF, dH);
        else
            hipLaunchKernelGGL(k_lu_init_terms, dim3((unsigned)((m + 255) / 256), gy), dim3(256), 0, st, nF, r->n, r->nnzA,
                               (const int32_t*)r->d_amap, (const cplx*)d_D, (int)mt, (const cplx*)dA, (const int32_t*)r->d_ldiag, dF, dH);
        LAUNCHCHK();
    }
    for (int l = 0; l < r->nlev; ++l) {
        const int64_t s0 = r->ext_seg0[l], s1 = r->ext_seg0[l + 1];
        if (s1 > s0) {
            // lanes per segment from the level's mean segment length (fixed per plan: part of the summation order)
            const int64_t np_ = r->h_ext_ptr[s1] - r->h_ext_ptr[s0];
            const double avg = (double)np_ / (double)(s1 - s0);
            if (avg > 48.0)
                hipLaunchKernelGGL((k_lu_ext<16>), dim3((unsigned)(((s1 - s0) * 16 + 255) / 256), gy), dim3(256), 0, st, s0, s1,
                                   (const int64_t*)r->d_ext_ptr, (const int32_t*)r->d_ext_dst, (const int32_t*)r->d_ext_src, dF, nF);
            else if (avg > 6.0)
                hipLaunchKernelGGL((k_lu_ext<4>), dim3((unsigned)(((s1 - s0) * 4 + 255) / 256), gy), dim3(256), 0, st, s0, s1,
                                   (const int64_t*)r->d_ext_ptr, (const int32_t*)r->d_ext_dst, (const int32_t*)r->d_ext_src, dF, nF);
            else
                hipLaunchKernelGGL((k_lu_ext<1>), dim3((unsigned)((s1 - s0 + 255) / 256), gy), dim3(256), 0, st, s0, s1,
                                   (const int64_t*)r->d_ext_ptr, (const int32_t*)r->d_ext_dst, (const int32_t*)r->d_ext_src, dF, nF);
            LAUNCHCHK();
        }
        const int b0 = r->lev_blk[l], b1 = r->lev_blk[l + 1];
        if (b1 <= b0) continue;
        if (r->wide[l] && r->fuseP >= 2) {
            const int P = r->fuseP;
            for (int64_t ps = r->f_step0[l]; ps < r->f_step0[l + 1]; ++ps) {
                const int64_t t0 = r->f_base[ps], t1 = t0 + r->f_cnt[ps];
                if (t1 <= t0) continue;
                const dim3 g((unsigned)((t1 - t0 + 255) / 256), gy);
                if (P == 2) hipLaunchKernelGGL((k_lu_widep<2>), g, dim3(256), 0, st, t0, t1, (const int32_t*)r->d_frec, (const int32_t*)r->d_fhdr, dF, nF);
                else if (P == 3) hipLaunchKernelGGL((k_lu_widep<3>), g, dim3(256), 0, st, t0, t1, (const int32_t*)r->d_frec, (const int32_t*)r->d_fhdr, dF, nF);
                else hipLaunchKernelGGL((k_lu_widep<4>), g, dim3(256), 0, st, t0, t1, (const int32_t*)r->d_frec, (const int32_t*)r->d_fhdr, dF, nF);
            }
            for (int rr = 0; rr + 1 < P; ++rr) {
                const int64_t t0 = r->fix_base[(size_t)l * (P - 1) + rr], t1 = t0 + r->fix_cnt[(size_t)l * (P - 1) + rr];
                if (t1 <= t0) continue;
                hipLaunchKernelGGL(k_lu_wide, dim3((unsigned)((t1 - t0 + 255) / 256), gy), dim3(256), 0, st, t0, t1, (const int4*)r->d_ffix, dF, nF);
            }
            LAUNCHCHK();
            const int q0 = r->h_blk_se[2 * b0], q1 = r->h_blk_se[2 * (b1 - 1) + 1];
            hipLaunchKernelGGL(k_lu_scale, dim3((unsigned)((q1 - q0 + 3) / 4), gy), dim3(256), 0, st, q0, q1, (const int32_t*)r->d_oldof,
                               (const int32_t*)r->d_Lp, (const int32_t*)r->d_Li, (const int32_t*)r->d_udiag, dF, dH, nF);
            LAUNCHCHK();
        } else if (r->wide[l]) {
            for (int64_t sidx = r->wstep0[l]; sidx < r->wstep0[l + 1]; ++sidx) {
                const int64_t t0 = r->wide_ptr[sidx], t1 = r->wide_ptr[sidx + 1];
                if (t1 <= t0) continue;
                hipLaunchKernelGGL(k_lu_wide, dim3((unsigned)((t1 - t0 + 255) / 256), gy), dim3(256), 0, st, t0, t1, (const int4*)r->d_wide, dF, nF);
            }
            LAUNCHCHK();
            const int q0 = r->h_blk_se[2 * b0], q1 = r->h_blk_se[2 * (b1 - 1) + 1];
            hipLaunchKernelGGL(k_lu_scale, dim3((unsigned)((q1 - q0 + 3) / 4), gy), dim3(256), 0, st, q0, q1, (const int32_t*)r->d_oldof,
                               (const int32_t*)r->d_Lp, (const int32_t*)r->d_Li, (const int32_t*)r->d_udiag, dF, dH, nF);
            LAUNCHCHK();
        } else {
            hipLaunchKernelGGL(k_lu_int, dim3((unsigned)(b1 - b0), gy), dim3(512), 0, st, b0, (const int32_t*)r->d_blk_se,
                               (const int32_t*)r->d_oldof, (const int32_t*)r->d_Lp, (const int32_t*)r->d_Li, (const int32_t*)r->d_udiag,
                               (const int64_t*)r->d_piv_ptr, (const int32_t*)r->d_int, dF, dH, nF);
            LAUNCHCHK();
        }
    }
    hipLaunchKernelGGL(k_lu_umax, dim3((unsigned)std::min<int64_t>(256, (r->nnzU + 255) / 256), gy), dim3(256), 0, st, r->nnzL, r->nnzU,
                       (const cplx*)dF, nF, dH);
    LAUNCHCHK();
    // health words (and, for tests, the factor values): one read-back behind the factorisation kernels
    std::vector<double> hw((size_t)B * NEP_LU_HW);
    HIPCHK(hipMemcpyAsync(hw.data(), dH, (size_t)B * NEP_LU_HW * 8, hipMemcpyDeviceToHost, st));
    if (h_LUx_out) HIPCHK(hipMemcpyAsync(h_LUx_out, dF, (size_t)B * nF * sizeof(cplx), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    std::vector<int> accepted;
    for (int b = 0; b < B; ++b) {
        // reported: [0] bad pivot, [1] max |L|, [2] element growth max|U| / max|A| of this static-pivot factorisation
        double* hh = h_health + 3 * b;
        const double amax = hw[(size_t)b * NEP_LU_HW + 3], umax = hw[(size_t)b * NEP_LU_HW + 2];
        hh[0] = hw[(size_t)b * NEP_LU_HW]; hh[1] = hw[(size_t)b * NEP_LU_HW + 1];
        hh[2] = amax > 0.0 ? umax / amax : (umax > 0.0 ? 1.0e300 : 0.0);
        // refused (out[b] stays NULL, the caller factorises this one on the host with fresh pivoting): a broken pivot, or
        // growth in L or in U above the limit -- growth in U is the factor that bounds the backward error
        if (hh[0] != 0.0 || !(hh[1] <= growth_limit) || !(hh[2] <= growth_limit)) continue;
        accepted.push_back(b);
    }
    // the solve schedules of all accepted factors in one batched build (value gathers + block inverses with grid.y = factor);
    // NEP_LU_BATCH_BUILD=0: one factor at a time as before
    static const int batch_build = getenv("NEP_LU_BATCH_BUILD") ? atoi(getenv("NEP_LU_BATCH_BUILD")) : 1;
    if (batch_build && accepted.size() > 1) {
        std::vector<const nep_cdouble*> pL(accepted.size()), pU(accepted.size());
        std::vector<MLFactor*> Fs(accepted.size(), nullptr);
        for (size_t a = 0; a < accepted.size(); ++a) {
            const cplx* Fb = dF + (size_t)accepted[a] * nF;
            pL[a] = (const nep_cdouble*)Fb; pU[a] = (const nep_cdouble*)(Fb + r->nnzL);
        }
        rc = ml_create_from_sym_batch(r->S, (int)accepted.size(), pL.data(), pU.data(), st, expected_solves, Fs.data());
        if (rc) return fail(rc);
        for (size_t a = 0; a < accepted.size(); ++a) {
            (void)ml_wait_ready(Fs[a], st);
            out[accepted[a]] = nep_lu_wrap_ml(Fs[a], r->n, r->nnzL, r->nnzU);
        }
    } else {
        for (int b : accepted) {
            MLFactor* F = nullptr;
            const cplx* Fb = dF + (size_t)b * nF;
            rc = ml_create_from_sym(r->S, (const nep_cdouble*)Fb, (const nep_cdouble*)(Fb + r->nnzL), st, expected_solves, &F);
            if (rc) {
                for (int c = 0; c < B; ++c) if (out[c]) { nep_lu_destroy(out[c]); out[c] = nullptr; }
                return fail(rc);
            }
            // the gathers run on a build stream: `st` waits for the factor's ready event so that the frees below, ordered on st,
            // come after them (ml_solve does the same wait before the first solve anyway)
            (void)ml_wait_ready(F, st);
            out[b] = nep_lu_wrap_ml(F, r->n, r->nnzL, r->nnzU);
        }
    }
    nep_pool_free_on(dF, st, true); nep_pool_free_on(dA, st, true);
    return NEP_OK;
}

}  // extern "C"
