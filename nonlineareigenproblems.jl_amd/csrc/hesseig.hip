// libnepmi355: eigen-decomposition of a small upper Hessenberg matrix ON THE DEVICE.
//
// replaces: `D,Z = eigen(H[1:k,1:k])` of the Krylov drivers (src/method_iar.jl:112, src/method_tiar.jl:182) -- LAPACK zgeev on
//           the host in the reference; until round 3 zhseqr + zhsein on host worker threads here (133 ms of CPU per headline
//           call for the 100 decompositions of a run).  With the decomposition on the device no LAPACK thread and no waiter
//           per step exists: H's columns never leave HBM on their way into the Ritz GEMM.
//
// Two kernels (both latency-bound by a serial chain; neither has an HBM or MFMA roofline -- the matrix is <= 160 KB):
//
//   k_hess_qr     TWO wavefronts per matrix (QR half / RQ half, below), the matrix in LDS as a packed upper Hessenberg array
//                 (k <= 128: 16 (k^2/2 + 1.5 k) + 32 k bytes = 86 KB at k = 100 -- half a CU's LDS, so that the streaming kernels
//                 of the recurrence keep a place on the CUs a decomposition runs on).
//                 Explicit single-shift QR iteration on the active window [l, i] (the structure of EISPACK's comqr) with
//                 LAPACK zlahqr's choices: Wilkinson shift from the trailing 2 x 2 block, exceptional shifts at iterations 10
//                 and 20, the Ahues-Tisseur deflation test, 30 max(10, k) iterations per eigenvalue, eigenvalues only (only
//                 the window is transformed).  The explicit form is what maps onto a wavefront: the QR half of a step walks
//                 down the diagonal with lane = column, each lane carrying its column's current upper element in a register
//                 (the next row comes from LDS one step ahead of its use, the finished row goes back fire-and-forget), the
//                 generating pair is read with v_readlane and every lane forms the rotation redundantly -- no LDS round trip
//                 on the serial chain; the RQ half runs with lane = row and no dependency at all between lanes.  The implicit
//                 (bulge-chasing) form needs two dependent LDS round trips per rotation.
//   k_hess_invit  one wavefront per eigenvalue (k workgroups): LAPACK zhsein / zlaein for right eigenvectors -- LU of
//                 H - w I with row interchanges (lane = column, same carried-row scheme, U goes to a scratch block in HBM/L2),
//                 back substitution from the start vector eps3 (1, ..., 1)^T (L is never applied, as in zlaein), growth test
//                 0.1 / sqrt(k) with zlaein's alternative start vectors, eigenvalues closer than eps3 perturbed as in zhsein.
//                 Vectors are returned with unit 2-norm and their largest component real positive (zgeev's convention).
//
// Status travels with the data: w[k] = (0 | 1-based index of the eigenvalue the iteration gave up on, QR sweeps),
// w[k+1] = (eigenvectors whose growth test failed or overflowed, 0); a caller that finds either non-zero falls back to LAPACK.
#include "common.h"
#include <math.h>

namespace {

constexpr double HQ_ULP = 2.220446049250313e-16;        // dlamch('P')
constexpr double HQ_SAFMIN = 2.2250738585072014e-308;   // dlamch('S')
constexpr int HQ_KMAX = 128;                             // two columns per lane; 16 (k^2/2 + 1.5 k) + 32 k bytes of LDS (k = 128: 141 KB)

// packed upper Hessenberg storage: column c holds its rows 0 .. c + 1 (the last column: 0 .. k - 1) at offset c (c + 3) / 2
__host__ __device__ __forceinline__ int hq_off(int c) { return (c * (c + 3)) >> 1; }
__host__ __device__ __forceinline__ int hq_entries(int k) { return hq_off(k - 1) + k; }
__device__ __forceinline__ double cabs1(cplx z) { return fabs(z.x) + fabs(z.y); }
__device__ __forceinline__ cplx readlane_c(cplx v, int lane) { return cmake(readlane_d(v.x, lane), readlane_d(v.y, lane)); }
__device__ __forceinline__ cplx cconj(cplx a) { return cmake(a.x, -a.y); }
__device__ __forceinline__ cplx czero() { return cmake(0.0, 0.0); }

// 1/sqrt(x) and 1/x to working precision from the hardware estimates (two Newton steps each): these sit on the serial chain
// of every rotation / elimination step, where the correctly rounded library sequences cost 3-4 times as many dependent
// instructions
__device__ __forceinline__ double rsqrt_nr(double x) {
    // one third-order step from the hardware estimate y0 (relative error e0 ~ 2^-26): with e = 1 - x y0^2,
    // 1/sqrt(x) = y0 (1 - e)^(-1/2) = y0 (1 + e/2 + 3 e^2/8 + O(e^3)) -- four dependent operations instead of the six of two
    // Newton steps, on the serial chain of every rotation
    const double y = __builtin_amdgcn_rsq(x);
    const double e = fma(-(x * y), y, 1.0);
    const double q = fma(0.375, e, 0.5);
    return fma(y * e, q, y);
}
__device__ __forceinline__ double rcp_nr(double x) {
    double y = __builtin_amdgcn_rcp(x);
    double e = fma(-x, y, 1.0);
    y = fma(y, e, y);
    e = fma(-x, y, 1.0);
    y = fma(y, e, y);
    return y;
}

// Givens rotation in LAPACK zlartg's form: c real, [[c, s], [-conj(s), c]] [f; g] = [r; 0].  The matrix has been scaled to
// max |entry| in [1/2, 1) once at load, so squares neither overflow nor lose anything that matters: an entry whose square
// is below 1e-280 is below 1e-140 of the matrix norm and is treated as zero.
__device__ __forceinline__ void lartg_n(cplx f, cplx g, double& c, cplx& s) {
    const double f2 = fma(f.x, f.x, f.y * f.y), g2 = fma(g.x, g.x, g.y * g.y);
    if (f2 < 1e-280) {                                                // rare: f = 0 -> the rotation swaps the rows (or g = 0 too)
        if (g2 < 1e-280) { c = 1.0; s = czero(); return; }
        const double ig = rsqrt_nr(g2);
        c = 0.0; s = cmake(g.x * ig, -g.y * ig);
        return;
    }
    const double p = rsqrt_nr(f2 * (f2 + g2));                        // g = 0 gives c = 1, s = 0 by itself
    c = f2 * p;
    const cplx fp = cmake(f.x * p, f.y * p);
    s = cmul(cconj(g), fp);
}

// num / den (Smith's form, reciprocals by Newton)
__device__ __forceinline__ cplx cdiv_fast(cplx n, cplx d) {
    if (fabs(d.x) >= fabs(d.y)) {
        const double r = d.y * rcp_nr(d.x);
        const double id = rcp_nr(fma(d.y, r, d.x));
        return cmake(fma(n.y, r, n.x) * id, fma(-n.x, r, n.y) * id);
    }
    const double r = d.x * rcp_nr(d.y);
    const double id = rcp_nr(fma(d.x, r, d.y));
    return cmake(fma(n.x, r, n.y) * id, fma(n.y, r, -n.x) * id);
}

// principal square root of a value of moderate size (the scaled matrix): |z| and sqrt through rsqrt_nr
__device__ __forceinline__ cplx csqrt_fast(cplx z) {
    const double m2 = fma(z.x, z.x, z.y * z.y);
    if (m2 < 1e-290) return czero();
    const double m = m2 * rsqrt_nr(m2);
    const double h = 0.5 * (fabs(z.x) + m);                           // >= m / 2 > 0
    const double ih = rsqrt_nr(h), t = h * ih;
    if (z.x >= 0.0) return cmake(t, 0.5 * z.y * ih);
    return cmake(0.5 * fabs(z.y) * ih, copysign(t, z.y));
}

// zlahqr's test for a negligible subdiagonal entry H(kk, kk-1) (with |re| + |im| where zlahqr, whose subdiagonal is kept
// real, has |re|)
__device__ __forceinline__ bool negligible_sub(const cplx* A, int ld, int kk, int n, double smlnum) {
    const double sub = cabs1(A[hq_off((kk - 1)) + kk]);
    if (sub <= smlnum) return true;
    const cplx d1 = A[hq_off((kk - 1)) + kk - 1], d2 = A[hq_off(kk) + kk];
    double tst = cabs1(d1) + cabs1(d2);
    if (tst == 0.0) {
        if (kk - 2 >= 0) tst += cabs1(A[hq_off((kk - 2)) + kk - 1]);
        if (kk + 1 <= n - 1) tst += cabs1(A[hq_off(kk) + kk + 1]);
    }
    if (sub <= HQ_ULP * tst) {
        const double sup = cabs1(A[hq_off(kk) + kk - 1]);
        const double ab = fmax(sub, sup), ba = fmin(sub, sup);
        const double a2 = cabs1(d2), dd = cabs1(csub(d1, d2));
        const double aa = fmax(a2, dd), bb = fmin(a2, dd);
        const double is = rcp_nr(aa + ab);
        if (ba * (ab * is) <= fmax(smlnum, HQ_ULP * (bb * (aa * is)))) return true;
    }
    return false;
}

// ---- one explicit QR step on the window [l, i] with the shift already subtracted from the window's diagonal -----------------
// QR half: R = G_i ... G_{l+1} H, lane = column (set 0: columns l + lane, set 1: l + 64 + lane).  A lane carries the current
// upper element of its column; the row below comes from LDS one step ahead.  Lanes left of the rotation keep computing on
// values nobody reads (their stores land below the subdiagonal, which the RQ half rewrites or nobody reads): no per-step masks.
// ---- one explicit QR step on the window [l, i], split over two wavefronts ----------------------------------------------------
// Wave 0 (QR half): R = G_i ... G_{l+1} (H - t I), lane = column (set 0: columns l + lane, set 1: l + 64 + lane).  A lane
// carries the current upper element of its column; the row below comes from LDS one step ahead; the finished row and the
// rotation of a step are stored at the START of the next step (unconditional, identical LDS sequence in every iteration:
// the compiler then emits counted waits and no LDS latency sits on the chain generate -> apply -> generate).  Lanes left of
// the rotation keep computing on values nobody reads (their stores land below the subdiagonal, which the RQ half rewrites).
// Wave 1 (RQ half): H <- R G_{l+1}^H ... G_i^H + t I, lane = row, trails wave 0 by two steps: step j needs rotation j and
// the rows <= j of R, i.e. what wave 0 stores at the start of its step j + 2.  Wave 0 publishes its step counter behind
// those stores; the LDS unit executes a wavefront's instructions in order, so a reader that sees the counter sees the data.
// The serial chain (rotation j needs column j after rotation j - 1) is all that is left on wave 0.  Measured split of its
// ~475 cycles per rotation (one column per lane, -DHQ_PROF builds with pieces taken out): rotation generation 130, LDS
// traffic (look-ahead load, three stores, progress word) 140, lane reads + application + loop 205 -- a lone wavefront issues
// one instruction every ~7 cycles, so the count of instructions is what there is to save, not their latency.
struct HqCtl {            // control block in LDS (ints)
    int seq;              // sweep number published by wave 0 (-1: exit)
    int l, i;             // window of that sweep
    int prog;             // wave 0: stores of steps < prog are issued (i + 2: all of them)
    int rq_done;          // wave 1: sweep number whose RQ half is complete
    int pad[3];
    double tx, ty;        // shift of the sweep
};
// the control words are read and written through explicit LDS pointers: a volatile access through a generic pointer becomes
// a FLAT instruction with system scope and a vmcnt(0) wait behind it (measured: 300 of 550 cycles per rotation)
typedef __attribute__((address_space(3))) volatile int lds_vint;
typedef __attribute__((address_space(3))) volatile double lds_vdbl;
__device__ __forceinline__ int ctl_load(const int* p) { return *(lds_vint*)p; }
__device__ __forceinline__ void ctl_store(int* p, int v) { *(lds_vint*)p = v; }
__device__ __forceinline__ double ctl_loadd(const double* p) { return *(lds_vdbl*)p; }
__device__ __forceinline__ void ctl_stored(double* p, double v) { *(lds_vdbl*)p = v; }
#define HQ_CBAR() asm volatile("" ::: "memory")

#define HQ_APPLY(c_, s_, up_, lo_, nu_, nl_)                                                                          \
    cplx nu_ = cscale(c_, up_); cfma(nu_, s_, lo_);                                                                   \
    cplx nl_ = cscale(c_, lo_); cfma(nl_, cmake(-(s_).x, (s_).y), up_)      /* nl = -conj(s) up + c lo */

template <bool TWO>
__device__ __forceinline__ void qr_half(cplx* A, cplx* rot, int* prog, const int ld, const int l, const int i,
                                        const int lane) {
    const int c0 = l + lane;
    if (!TWO) {
        const bool in0 = c0 <= i;
        cplx* col = A + hq_off(in0 ? c0 : i);
        cplx up = col[l], lo = col[l + 1];
        cplx lon = col[l + 2];                                          // (row i + 1 at most: inside the allocation, unused)
        double c; cplx s;
        lartg_n(readlane_c(up, 0), readlane_c(lo, 0), c, s);
        cplx pnu, ps; double pc;
        { HQ_APPLY(c, s, up, lo, nu, nl); pnu = nu; pc = c; ps = s; up = nl; lo = lon; }
        for (int j = l + 2; j <= i; ++j) {
            lon = col[j + 1];
            if (in0 && c0 + 2 >= j) col[j - 2] = pnu;            // rows <= c + 1 exist in the packed column c
            rot[2 * (j - 1)] = cmake(pc, 0.0); rot[2 * (j - 1) + 1] = ps;      // every lane writes the same pair
            HQ_CBAR(); ctl_store(prog, j); HQ_CBAR();
            const int gl = j - 1 - l;
            lartg_n(readlane_c(up, gl), readlane_c(lo, gl), c, s);
            HQ_APPLY(c, s, up, lo, nu, nl);
            pnu = nu; pc = c; ps = s; up = nl; lo = lon;
        }
        if (in0 && c0 + 2 >= i + 1) col[i - 1] = pnu;
        rot[2 * i] = cmake(pc, 0.0); rot[2 * i + 1] = ps;
        if (c0 == i) col[i] = up;
    } else {
        // window wider than a wavefront: every lane of set 0 is inside it
        const int c1 = c0 + 64;
        const bool in1 = c1 <= i;
        cplx* col0 = A + hq_off(c0);
        cplx* col1 = A + hq_off(in1 ? c1 : i);
        cplx up0 = col0[l], up1 = col1[l];
        cplx lo0 = col0[l + 1], lo1 = col1[l + 1];
        cplx lo0n = col0[l + 2], lo1n = col1[l + 2];
        double c; cplx s;
        lartg_n(readlane_c(up0, 0), readlane_c(lo0, 0), c, s);
        cplx pnu0, pnu1, ps; double pc;
        {
            HQ_APPLY(c, s, up0, lo0, nu0, nl0); HQ_APPLY(c, s, up1, lo1, nu1, nl1);
            pnu0 = nu0; pnu1 = nu1; pc = c; ps = s; up0 = nl0; up1 = nl1; lo0 = lo0n; lo1 = lo1n;
        }
        const int jm = l + 64;                                          // last rotation generated inside set 0 (jm <= i)
        for (int j = l + 2; j <= jm; ++j) {
            lo0n = col0[j + 1]; lo1n = col1[j + 1];
            if (c0 + 2 >= j) col0[j - 2] = pnu0;
            if (in1) col1[j - 2] = pnu1;                              // (c1 >= l + 64 >= j - 1 in this loop)
            rot[2 * (j - 1)] = cmake(pc, 0.0); rot[2 * (j - 1) + 1] = ps;
            HQ_CBAR(); ctl_store(prog, j); HQ_CBAR();
            const int gl = j - 1 - l;
            lartg_n(readlane_c(up0, gl), readlane_c(lo0, gl), c, s);
            HQ_APPLY(c, s, up0, lo0, nu0, nl0);
            HQ_APPLY(c, s, up1, lo1, nu1, nl1);
            pnu0 = nu0; pnu1 = nu1; pc = c; ps = s;
            up0 = nl0; up1 = nl1; lo0 = lo0n; lo1 = lo1n;
        }
        if (c0 + 2 >= jm + 1) col0[jm - 1] = pnu0;                       // set 0 is left of every further rotation
        for (int j = jm + 1; j <= i; ++j) {
            lo1n = col1[j + 1];
            if (in1 && c1 + 2 >= j) col1[j - 2] = pnu1;
            rot[2 * (j - 1)] = cmake(pc, 0.0); rot[2 * (j - 1) + 1] = ps;
            HQ_CBAR(); ctl_store(prog, j); HQ_CBAR();
            const int gl = j - 1 - l - 64;
            lartg_n(readlane_c(up1, gl), readlane_c(lo1, gl), c, s);
            HQ_APPLY(c, s, up1, lo1, nu1, nl1);
            pnu1 = nu1; pc = c; ps = s;
            up1 = nl1; lo1 = lo1n;
        }
        if (in1 && c1 + 2 >= i + 1) col1[i - 1] = pnu1;
        rot[2 * i] = cmake(pc, 0.0); rot[2 * i + 1] = ps;
        if (c1 == i) col1[i] = up1;
    }
    HQ_CBAR(); ctl_store(prog, i + 2); HQ_CBAR();
}

// RQ half (wave 1).  Row r joins at step j = r (its entry left of the diagonal is zero); a row that has not joined yet
// computes on zeros and stores zeros below the subdiagonal.  The shift goes back onto the diagonal as the entries are written.
#define HQ_RQ(c_, s_, y_, z_, o_, n_)                                                                                 \
    cplx o_ = cscale(c_, y_); cfma_conj(o_, s_, z_);                        /* c y + conj(s) z */                      \
    cplx n_ = cscale(c_, z_); cfma(n_, cmake(-(s_).x, -(s_).y), y_)         /* -s y + c z */
#define HQ_WAIT(need_)                                                                                                \
    while (pc_ < (need_)) { pc_ = ctl_load(prog); if (pc_ < (need_)) __builtin_amdgcn_s_sleep(1); }                    \
    HQ_CBAR()

template <bool TWO>
__device__ __forceinline__ void rq_half(cplx* A, const cplx* rot, const int* prog, const int ld, const int l, const int i,
                                        const cplx t, const int lane) {
    const int r0 = l + lane;
    int pc_ = 0;
    HQ_WAIT(l + 3);                                                     // rotation l + 1 and the rows l, l + 1 of R are stored
    if (!TWO) {
        const bool in0 = r0 <= i;
        const int r0c = in0 ? r0 : i;
        cplx y = (lane == 0) ? A[hq_off(l) + l] : czero();
        for (int j = l + 1; j <= i; ++j) {
            HQ_WAIT(j + 2);
            cplx z = A[hq_off(j) + r0c];
            const double c = rot[2 * j].x; const cplx s = rot[2 * j + 1];
            if (r0 > j) z = czero();
            HQ_RQ(c, s, y, z, o, ny);
            if (r0 == j - 1) o = cadd(o, t);
            if (in0 && r0 <= j) A[hq_off((j - 1)) + r0] = o;      // rows <= j exist in the packed column j - 1
            y = ny;
        }
        if (in0) { if (r0 == i) y = cadd(y, t); A[hq_off(i) + r0] = y; }
    } else {
        const int r1 = r0 + 64;
        const bool in1 = r1 <= i;
        const int r1c = in1 ? r1 : i;
        cplx y0 = (lane == 0) ? A[hq_off(l) + l] : czero(), y1 = czero();
        const int jm = l + 63;                                          // rows of set 1 join from j = l + 64 on (jm < i)
        for (int j = l + 1; j <= jm; ++j) {
            HQ_WAIT(j + 2);
            cplx z0 = A[hq_off(j) + r0];
            const double c = rot[2 * j].x; const cplx s = rot[2 * j + 1];
            if (r0 > j) z0 = czero();
            HQ_RQ(c, s, y0, z0, o, ny);
            if (r0 == j - 1) o = cadd(o, t);
            if (r0 <= j) A[hq_off((j - 1)) + r0] = o;
            y0 = ny;
        }
        for (int j = jm + 1; j <= i; ++j) {
            HQ_WAIT(j + 2);
            const cplx z0 = A[hq_off(j) + r0];
            cplx z1 = A[hq_off(j) + r1c];
            const double c = rot[2 * j].x; const cplx s = rot[2 * j + 1];
            if (r1 > j) z1 = czero();
            HQ_RQ(c, s, y0, z0, o0, n0);
            HQ_RQ(c, s, y1, z1, o1, n1);
            if (r0 == j - 1) o0 = cadd(o0, t);
            if (r1 == j - 1) o1 = cadd(o1, t);
            A[hq_off((j - 1)) + r0] = o0;
            if (in1 && r1 <= j) A[hq_off((j - 1)) + r1] = o1;
            y0 = n0; y1 = n1;
        }
        A[hq_off(i) + r0] = y0;                                             // (r0 < i: the window is wider than 64)
        if (in1) { if (r1 == i) y1 = cadd(y1, t); A[hq_off(i) + r1] = y1; }
    }
}

// ---- eigenvalues ---------------------------------------------------------------------------------------------------------
// work layout (complex slots): Ht[k*k] row-major copy of H for k_hess_invit, wk[k] perturbed eigenvalues, misc[2]
// (misc[0] = (eps3, smlnum), misc[1] = two 32-bit counters), then the U blocks of the eigenvector kernel
// One workgroup per matrix of a batch: matrix b is the leading k0 + b kstep rows / columns of H (the Arnoldi matrices of
// consecutive steps share their leading blocks), its results go to w_base + b w_stride, its workspace is work + b work_stride.
struct HessWork { cplx* Ht; cplx* wk; cplx* misc; cplx* U; };
__device__ __host__ inline HessWork hq_carve(void* d_work, int k) {
    HessWork W; cplx* p = (cplx*)d_work;
    W.Ht = p; p += (size_t)k * k; W.wk = p; p += k; W.misc = p; p += 2; W.U = p;
    return W;
}

__global__ __launch_bounds__(128) void k_hess_qr(int k0, int kstep, const cplx* __restrict__ H, int64_t ldh, cplx* __restrict__ w_base,
                                                 int64_t w_stride, char* __restrict__ work_base, int64_t work_stride,
                                                 cplx* __restrict__ mirror_base, int64_t mirror_stride) {
    const int k = k0 + (int)blockIdx.x * kstep;
    cplx* __restrict__ w_out = w_base + (size_t)blockIdx.x * w_stride;
    const HessWork Wk = hq_carve(work_base + (size_t)blockIdx.x * work_stride, k);
    cplx* __restrict__ Ht = Wk.Ht; cplx* __restrict__ wk_out = Wk.wk; cplx* __restrict__ misc = Wk.misc;
    cplx* __restrict__ mirror = mirror_base ? mirror_base + (size_t)blockIdx.x * mirror_stride : nullptr;
    extern __shared__ cplx sm[];
    cplx* A = sm;                                // packed upper Hessenberg: A(r, c) = A[hq_off(c) + r], r <= c + 1
    cplx* rot = sm + hq_entries(k);                      // rot[2 j] = (c_j, 0), rot[2 j + 1] = s_j; reused for the eigenvalues at the end
    HqCtl* ctl = (HqCtl*)(rot + 2 * (k + 1));
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ld = k;
    if (threadIdx.x == 0) { ctl->seq = 0; ctl->prog = 0; ctl->rq_done = 0; ctl->l = 0; ctl->i = 0; ctl->tx = 0.0; ctl->ty = 0.0; }
    __syncthreads();
    if (wave == 1) {
        // ---- wave 1: RQ halves, one per published sweep
        int myseq = 0;
        for (;;) {
            int sq;
            while ((sq = ctl_load(&ctl->seq)) == myseq) __builtin_amdgcn_s_sleep(1);
            HQ_CBAR();
            if (sq < 0) break;
            myseq = sq;
            const int lw = __builtin_amdgcn_readfirstlane(ctl_load(&ctl->l)), iw = __builtin_amdgcn_readfirstlane(ctl_load(&ctl->i));
            const cplx t = cmake(ctl_loadd(&ctl->tx), ctl_loadd(&ctl->ty));
            if (iw - lw + 1 > 64) rq_half<true>(A, rot, &ctl->prog, ld, lw, iw, t, lane);
            else rq_half<false>(A, rot, &ctl->prog, ld, lw, iw, t, lane);
            HQ_CBAR();
            ctl_store(&ctl->rq_done, myseq);
            HQ_CBAR();
        }
        __syncthreads();                         // (a) wave 0 has left its loop as well
        __syncthreads();                         // (b) wave 0 has written the results
        return;
    }
    // ---- wave 0: load, scale, iterate (deflation test, shift, QR halves), results
    double amax = 0.0;
    for (int c = 0; c < k; ++c) {
        const int rl = c + 1 < k - 1 ? c + 1 : k - 1;                 // last row stored for column c
        for (int r = lane; r <= rl; r += 64) {
            const cplx v = H[(size_t)c * ldh + r];
            A[hq_off(c) + r] = v;
            amax = fmax(amax, fmax(fabs(v.x), fabs(v.y)));
        }
    }
    // row-major copy for the eigenvector kernel (coalesced reads with lane = column there) and the infinity norm (zlanhs 'I')
    double hn = 0.0;
    for (int r = 0; r < k; ++r)
        for (int c = lane; c < k; c += 64) Ht[(size_t)r * k + c] = (r <= c + 1) ? A[hq_off(c) + r] : czero();
    for (int r = lane; r < k; r += 64) {
        double sum = 0.0;
        for (int c = (r > 0 ? r - 1 : 0); c < k; ++c) { const cplx v = A[hq_off(c) + r]; sum += hypot(v.x, v.y); }
        hn = fmax(hn, sum);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { hn = fmax(hn, shfl_xor_d(hn, off)); amax = fmax(amax, shfl_xor_d(amax, off)); }
    const double smlnum = HQ_SAFMIN * ((double)k / HQ_ULP);
    const double eps3 = hn > 0.0 ? hn * HQ_ULP : smlnum;
    // power-of-two scaling to max |entry| in [1/2, 1): exact, undone on the eigenvalues
    const int sexp = (amax > 0.0 && amax < INFINITY) ? __builtin_amdgcn_frexp_exp(amax) : 0;
    if (sexp != 0)
        for (int e = lane; e < hq_entries(k); e += 64) { const cplx v = A[e]; A[e] = cmake(ldexp(v.x, -sexp), ldexp(v.y, -sexp)); }

    int info = (amax < INFINITY) ? 0 : k, sweeps = 0;                 // Inf / NaN in the input: nothing to iterate on
    int i = k - 1;
    int seq = 0;
#ifdef HQ_PROF
    long long tp_wait = 0, tp_scan = 0, tp_shift = 0, tp_pub = 0, tp_qr = 0, tp_steps = 0; const long long tp_start = clock64();
#define TP(x) const long long x = clock64()
#else
#define TP(x)
#endif
    const int itmax = 30 * (k > 10 ? k : 10);
    while (i >= 0 && info == 0) {
        int l = 0, kdefl = 0;
        bool conv = false;
        for (int its = 0; its <= itmax; ++its) {
            // the RQ half of the previous sweep has to be complete before anything looks at the matrix
            TP(tq0);
            while (ctl_load(&ctl->rq_done) != seq) __builtin_amdgcn_s_sleep(1);
            HQ_CBAR();
            TP(tq1);
            // ---- largest kk in (l, i] whose subdiagonal entry is negligible
            int found = l;
            for (int base = i; base > l; base -= 64) {
                const int kk = base - lane;
                const bool neg = (kk > l) && negligible_sub(A, ld, kk, k, smlnum);
                const unsigned long long m = __ballot(neg);
                if (m) { found = base - (__ffsll((long long)m) - 1); break; }
            }
            l = __builtin_amdgcn_readfirstlane(found);      // wave-uniform by construction; says so to the compiler (scalar loop control)
            if (l > 0 && lane == 0) A[hq_off((l - 1)) + l] = czero();
            if (l >= i) { conv = true; break; }
            ++kdefl; ++sweeps;
            TP(tq2);
            // ---- shift (zlahqr)
            cplx t;
            if (kdefl % 20 == 0) {
                t = A[hq_off(i) + i]; t.x += 0.75 * cabs1(A[hq_off((i - 1)) + i]);
            } else if (kdefl % 10 == 0) {
                t = A[hq_off(l) + l]; t.x += 0.75 * cabs1(A[hq_off(l) + l + 1]);
            } else {
                t = A[hq_off(i) + i];
                const cplx u = cmul(csqrt_fast(A[hq_off(i) + i - 1]), csqrt_fast(A[hq_off((i - 1)) + i]));
                double s = cabs1(u);
                if (s != 0.0) {
                    const cplx x = cscale(0.5, csub(A[hq_off((i - 1)) + i - 1], t));
                    const double sx = cabs1(x);
                    s = fmax(s, sx);
                    const double is = rcp_nr(s);
                    const cplx xs = cscale(is, x), us = cscale(is, u);
                    cplx y = cscale(s, csqrt_fast(cadd(cmul(xs, xs), cmul(us, us))));
                    if (sx > 0.0 && (x.x * y.x + x.y * y.y) < 0.0) y = cmake(-y.x, -y.y);
                    const cplx den = cadd(x, y);
                    if (den.x != 0.0 || den.y != 0.0) t = csub(t, cmul(u, cdiv_fast(u, den)));
                }
            }
            // ---- H - t I on the window's diagonal; publish the sweep; QR half (window bounds as scalars: the loops then run
            //      on scalar counters and branches)
            const int lw = __builtin_amdgcn_readfirstlane(l), iw = __builtin_amdgcn_readfirstlane(i);
            TP(tq3);
            {
                const int d0 = lw + lane, d1 = d0 + 64;
                if (d0 <= iw) A[hq_off(d0) + d0] = csub(A[hq_off(d0) + d0], t);
                if (d1 <= iw) A[hq_off(d1) + d1] = csub(A[hq_off(d1) + d1], t);
            }
            ++seq;
            HQ_CBAR();
            ctl_store(&ctl->prog, 0); ctl_store(&ctl->l, lw); ctl_store(&ctl->i, iw);
            ctl_stored(&ctl->tx, t.x); ctl_stored(&ctl->ty, t.y);
            HQ_CBAR();
            ctl_store(&ctl->seq, seq);
            HQ_CBAR();
            TP(tq4);
            if (iw - lw + 1 > 64) qr_half<true>(A, rot, &ctl->prog, ld, lw, iw, lane);
            else qr_half<false>(A, rot, &ctl->prog, ld, lw, iw, lane);
#ifdef HQ_PROF
            { const long long tq5 = clock64(); tp_wait += tq1 - tq0; tp_scan += tq2 - tq1; tp_shift += tq3 - tq2; tp_pub += tq4 - tq3; tp_qr += tq5 - tq4; tp_steps += iw - lw; }
#endif
        }
        if (!conv) { info = i + 1; break; }
        i = __builtin_amdgcn_readfirstlane(l - 1);
    }
    while (ctl_load(&ctl->rq_done) != seq) __builtin_amdgcn_s_sleep(1);
    HQ_CBAR();
    ctl_store(&ctl->seq, -1);
    HQ_CBAR();
    __syncthreads();                             // (a)
    // ---- zhsein: eigenvalues closer than eps3 to an earlier one are perturbed (independent vectors); wk only feeds the
    //      eigenvector kernel, the eigenvalues returned are the unperturbed ones
    //      (a converged eigenvalue stays on the diagonal: nothing outside the active window is touched afterwards)
    cplx* wk = rot;
    for (int e = lane; e < k; e += 64) {
        cplx we = czero();
        if (info == 0) { we = A[hq_off(e) + e]; we = cmake(ldexp(we.x, sexp), ldexp(we.y, sexp)); }
        wk[e] = we; w_out[e] = we;
        if (mirror) mirror[e] = we;
    }
    if (info == 0) {
        for (int e = 1; e < k; ++e) {
            for (int guard = 0; guard < 4 * k; ++guard) {
                const cplx we = wk[e];
                bool close = false;
                for (int q = lane; q < e; q += 64) close = close || (cabs1(csub(wk[q], we)) < eps3);
                if (!__any(close)) break;
                wk[e] = cmake(we.x + eps3, we.y);             // every lane writes the same value: nothing to communicate
            }
        }
    }
    for (int e = lane; e < k; e += 64) wk_out[e] = wk[e];
    if (lane == 0) {
        w_out[k] = cmake((double)info, (double)sweeps);
        w_out[k + 1] = czero();
        misc[0] = cmake(eps3, smlnum);
        misc[1] = czero();                                            // failure counter, ticket
    }
    if (mirror && lane == 0) mirror[k] = cmake((double)info, (double)sweeps);
#ifdef HQ_PROF
    if (lane == 0) {      // cycle split of the iteration into the U scratch (debug builds only)
        double* dbg = (double*)(misc + 2);
        dbg[0] = (double)(clock64() - tp_start); dbg[1] = (double)tp_wait; dbg[2] = (double)tp_scan; dbg[3] = (double)tp_shift;
        dbg[4] = (double)tp_pub; dbg[5] = (double)tp_qr; dbg[6] = (double)tp_steps; dbg[7] = (double)sweeps;
    }
#endif
    __syncthreads();                             // (b)
}

// ---- eigenvectors --------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_hess_invit(int k0, int kstep, cplx* __restrict__ w_base, int64_t w_stride,
                                                   char* __restrict__ work_base, int64_t work_stride, cplx* __restrict__ Z_base,
                                                   int64_t ldz, int64_t z_stride, cplx* __restrict__ mirror_base,
                                                   int64_t mirror_stride) {
    const int lane = threadIdx.x, e = blockIdx.x;
    const int k = k0 + (int)blockIdx.y * kstep;
    if (e >= k) return;                                               // the grid is as wide as the largest matrix of the batch
    const HessWork Wk = hq_carve(work_base + (size_t)blockIdx.y * work_stride, k);
    const cplx* __restrict__ Ht = Wk.Ht; const cplx* __restrict__ wk = Wk.wk; cplx* __restrict__ misc = Wk.misc;
    cplx* __restrict__ Uall = Wk.U;
    cplx* __restrict__ w_io = w_base + (size_t)blockIdx.y * w_stride;
    cplx* __restrict__ mirror = mirror_base ? mirror_base + (size_t)blockIdx.y * mirror_stride : nullptr;
    const int r0 = lane, r1 = lane + 64;
    cplx* zc = Z_base + (size_t)blockIdx.y * z_stride + (size_t)e * ldz;
    const bool qr_failed = w_io[k].x != 0.0;
    bool failed = false;
    if (qr_failed) {
        if (r0 < k) zc[r0] = czero();
        if (r1 < k) zc[r1] = czero();
    } else if (k == 1) {
        if (lane == 0) zc[0] = cmake(1.0, 0.0);
    } else {
        const double eps3 = misc[0].x;
        const cplx wv = wk[e];
        cplx* U = Uall + (size_t)e * k * k;                          // column-major: U(i, c) = U[c * k + i]
        const bool two = k > 64;
        // ---- LU of B = H - wv I with row interchanges (zlaein), lane = column, current row carried in registers
        const int c0 = lane, c1 = lane + 64;
        cplx up0 = (c0 < k) ? Ht[c0] : czero();
        cplx up1 = (two && c1 < k) ? Ht[c1] : czero();
        if (lane == 0) up0 = csub(up0, wv);
        cplx lo0n = (c0 < k) ? Ht[(size_t)k + c0] : czero();
        cplx lo1n = (two && c1 < k) ? Ht[(size_t)k + c1] : czero();
        for (int i = 0; i < k - 1; ++i) {
            cplx lo0 = lo0n, lo1 = lo1n;
            if (i + 2 < k) {
                lo0n = (c0 >= i + 1 && c0 < k) ? Ht[(size_t)(i + 2) * k + c0] : czero();
                if (two) lo1n = (c1 >= i + 1 && c1 < k) ? Ht[(size_t)(i + 2) * k + c1] : czero();
            }
            if (c0 == i + 1) lo0 = csub(lo0, wv);
            if (c1 == i + 1) lo1 = csub(lo1, wv);
            cplx bii, ei;
            if (i < 64) { bii = readlane_c(up0, i); ei = readlane_c(lo0, i); }
            else { bii = readlane_c(up1, i - 64); ei = readlane_c(lo1, i - 64); }
            const bool swap = cabs1(bii) < cabs1(ei);
            cplx piv = swap ? ei : bii;
            if (!swap && piv.x == 0.0 && piv.y == 0.0) piv = cmake(eps3, 0.0);
            const cplx x = cdiv_fast(swap ? bii : ei, piv);
            const cplx mx = cmake(-x.x, -x.y);
            if (i < 64) {
                cplx row = swap ? lo0 : up0, oth = swap ? up0 : lo0;
                if (c0 == i) row = piv;
                cfma(oth, mx, row);
                if (c0 >= i && c0 < k) U[(size_t)c0 * k + i] = row;
                up0 = oth;
            }
            if (two) {
                cplx row = swap ? lo1 : up1, oth = swap ? up1 : lo1;
                if (c1 == i) row = piv;
                cfma(oth, mx, row);
                if (c1 >= i && c1 < k) U[(size_t)c1 * k + i] = row;
                up1 = oth;
            }
        }
        {
            cplx d = (k - 1 < 64) ? readlane_c(up0, k - 1) : readlane_c(up1, k - 1 - 64);
            if (d.x == 0.0 && d.y == 0.0) d = cmake(eps3, 0.0);
            if (lane == 0) U[(size_t)(k - 1) * k + k - 1] = d;
        }
        __threadfence_block();
        __syncthreads();
        // ---- U x = v by columns (lane = row); v = eps3 (1, ..., 1)^T first, zlaein's alternatives if the growth test fails
        const double rootn = sqrt((double)k), growto = 0.1 / rootn;
        cplx b0 = cmake(eps3, 0.0), b1 = cmake(eps3, 0.0);
        bool ok = false;
        for (int its = 1; its <= k; ++its) {
            cplx u0n = (r0 <= k - 1) ? U[(size_t)(k - 1) * k + r0] : czero();
            cplx u1n = (two && r1 <= k - 1) ? U[(size_t)(k - 1) * k + r1] : czero();
            for (int i = k - 1; i >= 0; --i) {
                const cplx u0 = u0n, u1 = u1n;
                if (i > 0) {
                    u0n = (r0 <= i - 1) ? U[(size_t)(i - 1) * k + r0] : czero();
                    if (two) u1n = (r1 <= i - 1) ? U[(size_t)(i - 1) * k + r1] : czero();
                }
                cplx bi, uii;
                if (i < 64) { bi = readlane_c(b0, i); uii = readlane_c(u0, i); }
                else { bi = readlane_c(b1, i - 64); uii = readlane_c(u1, i - 64); }
                const cplx xi = cdiv_fast(bi, uii);
                const cplx mxi = cmake(-xi.x, -xi.y);
                if (r0 < i) cfma(b0, mxi, u0); else if (r0 == i) b0 = xi;
                if (two) { if (r1 < i) cfma(b1, mxi, u1); else if (r1 == i) b1 = xi; }
            }
            double vn = (r0 < k ? cabs1(b0) : 0.0) + ((two && r1 < k) ? cabs1(b1) : 0.0);
            vn = wave_reduce_sum(vn);
            if (!(vn < INFINITY)) break;                             // overflow / NaN: report, the caller falls back
            if (vn >= growto) { ok = true; break; }
            const double rtemp = eps3 / (rootn + 1.0);
            b0 = cmake(lane == 0 ? eps3 : rtemp, 0.0); b1 = cmake(rtemp, 0.0);
            const int idx = k - its;
            if (r0 == idx) b0.x -= eps3 * rootn;
            if (r1 == idx) b1.x -= eps3 * rootn;
        }
        failed = !ok;
        // ---- unit 2-norm, largest component real positive
        double m0 = (r0 < k) ? fma(b0.x, b0.x, b0.y * b0.y) : 0.0, m1 = (two && r1 < k) ? fma(b1.x, b1.x, b1.y * b1.y) : 0.0;
        double best = fmax(m0, m1);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) best = fmax(best, shfl_xor_d(best, off));
        double nrm2;
        {
            // scaled sum of squares: the entries span up to 1/ulp
            const double ib = best > 0.0 ? 1.0 / best : 0.0;
            nrm2 = wave_reduce_sum(m0 * ib + m1 * ib);
        }
        const unsigned long long who0 = __ballot(m0 == best && r0 < k);
        cplx big;
        if (who0) { const int wl = __ffsll((long long)who0) - 1; big = readlane_c(b0, wl); }
        else { const unsigned long long who1 = __ballot(m1 == best); const int wl = who1 ? __ffsll((long long)who1) - 1 : 0; big = readlane_c(b1, wl); }
        cplx ph = cmake(1.0, 0.0);
        if (ok && best > 0.0) {
            const double ab = sqrt(best);                             // |big|
            const double sc = 1.0 / (ab * sqrt(nrm2));                // 1 / ||v||_2
            ph = cmake(big.x / ab * sc, -big.y / ab * sc);            // conj(big) / |big| / ||v||
        } else {
            b0 = czero(); b1 = czero();
        }
        if (r0 < k) zc[r0] = cmul(b0, ph);
        if (two && r1 < k) zc[r1] = cmul(b1, ph);
    }
    // ---- the last workgroup publishes the number of failed vectors
    if (lane == 0) {
        unsigned* cnt = (unsigned*)&misc[1];
        if (failed) atomicAdd(&cnt[0], 1u);
        __threadfence();
        const unsigned t = atomicAdd(&cnt[1], 1u);
        if (t == (unsigned)k - 1u) {
            __threadfence();
            const unsigned nf = atomicAdd(&cnt[0], 0u);
            w_io[k + 1] = cmake((double)nf, 0.0);
            if (mirror) mirror[k + 1] = cmake((double)nf, 0.0);
        }
    }
}

cplx* mapped(nep_cdouble* h_mirror) {
    if (!h_mirror) return nullptr;
    void* dp = nullptr;
    if (hipHostGetDevicePointer(&dp, h_mirror, 0) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return (cplx*)dp;
}

}  // namespace

extern "C" {

int32_t nep_hess_eig_worksize(int32_t k, int64_t* bytes) {
    ARGCHK(bytes && k >= 1);
    if (k > HQ_KMAX) { nep_set_error("nep_hess_eig: k = %d exceeds the LDS-resident limit %d", k, HQ_KMAX); return NEP_ERR_UNSUPPORTED; }
    *bytes = (int64_t)16 * ((int64_t)k * k * k + (int64_t)k * k + k + 2) + 64;
    return NEP_OK;
}

int32_t nep_hess_eigvals_batch_dev(int32_t nb, int32_t k0, int32_t kstep, const nep_cdouble* dH, int64_t ldh, nep_cdouble* d_w,
                                   int64_t w_stride, void* d_work, int64_t work_stride, nep_cdouble* h_mirror,
                                   int64_t mirror_stride, nep_stream stream) {
    ARGCHK(nb >= 1 && k0 >= 1 && kstep >= 0 && dH && d_w && d_work);
    const int kmax = k0 + (nb - 1) * kstep;
    ARGCHK(ldh >= kmax);
    if (kmax > HQ_KMAX) { nep_set_error("nep_hess_eigvals_dev: k = %d exceeds the LDS-resident limit %d", kmax, HQ_KMAX); return NEP_ERR_UNSUPPORTED; }
    if (nb > 1) {
        int64_t need = 0; int rcw = nep_hess_eig_worksize(kmax, &need); if (rcw) return rcw;
        ARGCHK(w_stride >= kmax + 2 && work_stride >= need && (work_stride % 16) == 0 && (!h_mirror || mirror_stride >= kmax + 2));
    }
    const size_t lds = (size_t)16 * hq_entries(kmax) + (size_t)32 * (kmax + 1) + 48;  // packed matrix, rotations, control block
    if (lds > 65536) { int rc = nep_raise_lds((const void*)k_hess_qr, 163840); if (rc) return rc; }
    cplx* mir = mapped(h_mirror);
    if (h_mirror && !mir) { nep_set_error("nep_hess_eigvals_dev: h_mirror is not mapped pinned host memory"); return NEP_ERR_ARG; }
    hipLaunchKernelGGL(k_hess_qr, dim3((unsigned)nb), dim3(128), lds, as_stream(stream), (int)k0, (int)kstep, (const cplx*)dH, ldh,
                       (cplx*)d_w, w_stride, (char*)d_work, work_stride, mir, mirror_stride);
    LAUNCHCHK();
    return NEP_OK;
}

int32_t nep_hess_eigvecs_batch_dev(int32_t nb, int32_t k0, int32_t kstep, nep_cdouble* d_w, int64_t w_stride, nep_cdouble* dZ,
                                   int64_t ldz, int64_t z_stride, void* d_work, int64_t work_stride, nep_cdouble* h_mirror,
                                   int64_t mirror_stride, nep_stream stream) {
    ARGCHK(nb >= 1 && k0 >= 1 && kstep >= 0 && d_w && dZ && d_work);
    const int kmax = k0 + (nb - 1) * kstep;
    ARGCHK(kmax <= HQ_KMAX && ldz >= kmax && (nb == 1 || z_stride >= (int64_t)kmax * ldz));
    cplx* mir = mapped(h_mirror);
    if (h_mirror && !mir) { nep_set_error("nep_hess_eigvecs_dev: h_mirror is not mapped pinned host memory"); return NEP_ERR_ARG; }
    hipLaunchKernelGGL(k_hess_invit, dim3((unsigned)kmax, (unsigned)nb), dim3(64), 0, as_stream(stream), (int)k0, (int)kstep,
                       (cplx*)d_w, w_stride, (char*)d_work, work_stride, (cplx*)dZ, ldz, z_stride, mir, mirror_stride);
    LAUNCHCHK();
    return NEP_OK;
}

int32_t nep_hess_eigvals_dev(int32_t k, const nep_cdouble* dH, int64_t ldh, nep_cdouble* d_w, void* d_work,
                             nep_cdouble* h_mirror, nep_stream stream) {
    return nep_hess_eigvals_batch_dev(1, k, 0, dH, ldh, d_w, 0, d_work, 0, h_mirror, 0, stream);
}

int32_t nep_hess_eigvecs_dev(int32_t k, nep_cdouble* d_w, nep_cdouble* dZ, int64_t ldz, void* d_work, nep_cdouble* h_mirror,
                             nep_stream stream) {
    return nep_hess_eigvecs_batch_dev(1, k, 0, d_w, 0, dZ, ldz, 0, d_work, 0, h_mirror, 0, stream);
}

}  // extern "C"
