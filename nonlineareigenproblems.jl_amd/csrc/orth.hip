// libnepmi355: K6 Gram-Schmidt orthogonalisation (DGKS / CGS / MGS) for gfx950.
//
// One DGKS pass = two streaming kernels over the basis V (rows x k, column-major):
//   k_orth_dots   : partial[b][j] = sum_{r in row-chunk b} conj(V[r,j]) w[r]      (HBM-bound)
//   k_orth_update : w[r] -= sum_j V[r,j] h[j];  partial norm of the new w          (HBM-bound)
// plus two tiny fixed-order reductions (deterministic, no atomics).  The optional
// `active` array (iar: column j is zero below row (j+1) n, src/method_iar.jl:76,97-98) lets both
// kernels skip the structurally zero part of V, halving the traffic of iar's orthogonalisation.
#include "common.h"
#include <vector>
#include <algorithm>
#include <math.h>
#include <stdlib.h>

// V is streamed once per kernel; for blocks far larger than the 256 MB last-level cache the loads are non-temporal so that the
// stream does not evict what the kernels around a Gram-Schmidt pass re-read (the factors of the fixed-shift solve: 65 MB on gun)
typedef double orth_d2 __attribute__((ext_vector_type(2)));
template <bool NT> __device__ __forceinline__ cplx vload(const cplx* p) {
    if (NT) { const orth_d2 v = __builtin_nontemporal_load((const orth_d2*)p); return cmake(v.x, v.y); }
    return *p;
}

#define DOT_RPT 4
#define DOT_CG 8
#define DOT_RB (256 * DOT_RPT)

// ---- device-side DGKS decision (asynchronous path) ----------------------------------------------------------------------------
// After pass p the criterion ||w|| < ||c|| / sqrt(2) says whether pass p + 1 runs.  History: a one-workgroup kernel of its own
// behind every pass (round 1: 7-8 us + a launch gap, twice per Arnoldi step); then formed by every workgroup of the next pass'
// k_orth_dots from the <= ORTH_NPART update partials (round 2: no extra launch, but 15 us for every gated-off launch); now
// published by workgroup 0 of the pass' own k_orth_update BEFORE the update runs, from ||w||^2 (one more output of k_orth_dots)
// and ||c||^2 by Pythagoras -- the kernels of a gated-off pass read one word and leave.
// state: [1] = passes done, [4 + p] = "pass p ran and wants another one" (p = 1 ..).
#define ORTH_NPART 1024          // k_orth_update launches at most this many workgroups on the asynchronous path (one partial each)
struct OrthDecide {              // what k_orth_finish reads on the asynchronous path (partial == nullptr: not in use)
    const double* partial = nullptr; int np = 0; const cplx* c = nullptr; int k = 0; int method = 0; int* state = nullptr;
    cplx* out_beta = nullptr;
};
__device__ __forceinline__ void orth_pass_norms(const OrthDecide& D, double& nrm, double& p2) {
    // (the first 256 threads of the workgroup, whatever its size: the order of the sum -- and with it the last bit of beta -- is the
    // same in the 256-thread k_orth_finish and the 512-thread k_orth_finish_vc)
    __shared__ double smn[2][4];
    if (threadIdx.x < 256) {
        double acc = 0.0, pj = 0.0;
        for (int b = threadIdx.x; b < D.np; b += 256) acc += D.partial[b];
        for (int j = threadIdx.x; j < D.k; j += 256) pj += D.c[j].x * D.c[j].x + D.c[j].y * D.c[j].y;
        acc = wave_reduce_sum(acc); pj = wave_reduce_sum(pj);
        if ((threadIdx.x & 63) == 0) { smn[0][threadIdx.x >> 6] = acc; smn[1][threadIdx.x >> 6] = pj; }
    }
    __syncthreads();
    double t = 0.0, q2 = 0.0;
    for (int q = 0; q < 4; ++q) { t += smn[0][q]; q2 += smn[1][q]; }
    __syncthreads();
    nrm = sqrt(t); p2 = q2;
}
// DPP: the eight wave sums of a column group by data-parallel-primitive moves (wave_sum_dpp) instead of the shuffle butterfly.  Both are
// fixed-order sums; they round differently.  Blocks below ORTH_DPP_ROWS rows keep the butterfly: the kernel is launch-bound there, and the
// small projected problems of nlar / the inner solvers (n <= 150) are sensitive to the last bit of their Hessenberg entries -- the gun
// twin of test/nlar.jl finds its second eigenvalue with one order and not with the other (scripts/diag/nlar_dbg.py).
#define ORTH_DPP_ROWS 32768
template <bool NT, bool DPP>
__global__ __launch_bounds__(256) void k_orth_dots(const cplx* __restrict__ V, int64_t ldv, int64_t rows,
                                                   int k, const int64_t* __restrict__ active,
                                                   const cplx* __restrict__ w, cplx* __restrict__ partial,
                                                   const int* __restrict__ gate = nullptr,
                                                   double* __restrict__ partial_ww = nullptr) {
    __shared__ cplx sm[DOT_CG][4];
    if (gate && *gate == 0) return;          // device-side DGKS decision (published by the previous pass' update): not needed
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t r0 = (int64_t)blockIdx.x * DOT_RB;
    cplx wr[DOT_RPT];
    int64_t rr[DOT_RPT];
#pragma unroll
    for (int i = 0; i < DOT_RPT; ++i) {
        rr[i] = r0 + threadIdx.x + 256 * i;
        wr[i] = rr[i] < rows ? w[rr[i]] : cmake(0.0, 0.0);
    }
    // ||w||^2 of the slice (asynchronous path): with the projection coefficients it gives the norm AFTER the update by
    // Pythagoras, i.e. the DGKS decision of this pass before its update has run (see k_orth_update)
    if (partial_ww && blockIdx.y == 0) {
        double a = 0.0;
#pragma unroll
        for (int i = 0; i < DOT_RPT; ++i) a += fma(wr[i].x, wr[i].x, wr[i].y * wr[i].y);
        a = wave_reduce_sum(a);
        if (lane == 0) sm[0][wv].x = a;
        __syncthreads();
        if (threadIdx.x == 0) partial_ww[blockIdx.x] = (sm[0][0].x + sm[0][1].x) + (sm[0][2].x + sm[0][3].x);
        __syncthreads();
    }
    // a block keeps its 1024-row slice of w in registers and walks over column groups blockIdx.y, +gridDim.y, ...:
    // with gridDim.y == 1 (large row counts) w is read once per slice instead of once per column group
    // (PMC: 971 MB fetched for 820 MB of algorithmic traffic at iar step 100 with one group per block)
    // (the loads of a column group go out together, unconditionally and with clamped rows, and are masked afterwards: behind the
    // per-column `if (r0 < act)` / per-row `if (row < act)` tests a workgroup waited for one column's four loads, reduced, and only
    // then asked for the next column -- 16 KB in flight per workgroup; at k <= 30 columns (GMRES on the waveguide, the first third
    // of an iar run) the kernel ran at 3.3 TB/s for that reason)
    int64_t rc[DOT_RPT];
#pragma unroll
    for (int i = 0; i < DOT_RPT; ++i) rc[i] = rr[i] < rows ? rr[i] : rows - 1;
    for (int j0 = blockIdx.y * DOT_CG; j0 < k; j0 += gridDim.y * DOT_CG) {
        int64_t actv[DOT_CG];
        bool grp_on = false;
#pragma unroll
        for (int jj = 0; jj < DOT_CG; ++jj) {
            const int j = j0 + jj;
            int64_t a = 0;
            if (j < k) { a = active ? active[j] : rows; if (a > rows) a = rows; }
            actv[jj] = a;
            grp_on = grp_on || r0 < a;
        }
        cplx accs[DOT_CG];
#pragma unroll
        for (int jj = 0; jj < DOT_CG; ++jj) accs[jj] = cmake(0.0, 0.0);
        if (grp_on) {
            // (a column that has no active row on this chunk is skipped by a wave-uniform branch: nothing waits inside it, the
            // loads of the other columns stay in flight together; without the skip the staircase of iar read 998 MB for 830 MB
            // algorithmic at step 100 -- the half-empty column groups along the stairs, profiles/pmc2)
            cplx v[DOT_CG][DOT_RPT];
#pragma unroll
            for (int jj = 0; jj < DOT_CG; ++jj) {
                if (r0 < actv[jj]) {
                    const cplx* vp = V + (int64_t)(j0 + jj) * ldv;
#pragma unroll
                    for (int i = 0; i < DOT_RPT; ++i) v[jj][i] = vload<NT>(vp + rc[i]);
                } else {
#pragma unroll
                    for (int i = 0; i < DOT_RPT; ++i) v[jj][i] = cmake(0.0, 0.0);
                }
            }
#pragma unroll
            for (int jj = 0; jj < DOT_CG; ++jj) {
                cplx acc = cmake(0.0, 0.0);
#pragma unroll
                for (int i = 0; i < DOT_RPT; ++i) {
                    const bool on = rr[i] < actv[jj];
                    cfma_conj(acc, cmake(on ? v[jj][i].x : 0.0, on ? v[jj][i].y : 0.0), wr[i]);
                }
                accs[jj] = DPP ? wave_sum_dpp(acc) : group_reduce_sum<64>(acc);
            }
        }
#pragma unroll
        for (int jj = 0; jj < DOT_CG; ++jj)
            if (lane == 0) sm[jj][wv] = accs[jj];
        __syncthreads();
        if (threadIdx.x < DOT_CG) {
            const int j = j0 + threadIdx.x;
            if (j < k) {
                cplx t = sm[threadIdx.x][0];
                for (int q = 1; q < 4; ++q) t = cadd(t, sm[threadIdx.x][q]);
                partial[(int64_t)blockIdx.x * k + j] = t;
            }
        }
        __syncthreads();
    }
}

// h[j] = sum_b partial[b*k + j]; one block per column, fixed summation tree -> deterministic
__global__ __launch_bounds__(256) void k_orth_reduce_h(int nb, int k, const cplx* __restrict__ partial,
                                                       cplx* __restrict__ h, const int* __restrict__ gate = nullptr,
                                                       cplx* __restrict__ hacc = nullptr, int first = 1,
                                                       int* __restrict__ state_reset = nullptr,
                                                       const OrthDecide dec = OrthDecide(), int pdone = 0,
                                                       const double* __restrict__ partial_ww = nullptr, double* __restrict__ ww = nullptr) {
    __shared__ cplx sm[4];
    // first pass of an asynchronous orthogonalisation: clear the pass state here (nothing reads it before the k_orth_dots of the
    // NEXT pass) instead of a separate memset command in front of every Arnoldi step
    if (state_reset && blockIdx.x == 0 && threadIdx.x < 16) state_reset[threadIdx.x] = 0;
    if (gate && *gate == 0) return;
    (void)dec; (void)pdone;
    const int j = blockIdx.x;
    cplx acc = cmake(0.0, 0.0);
    for (int b = threadIdx.x; b < nb; b += 256) acc = cadd(acc, partial[(int64_t)b * k + j]);
    acc = group_reduce_sum<64>(acc);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        const cplx c = cadd(cadd(sm[0], sm[1]), cadd(sm[2], sm[3]));
        h[j] = c;
        if (hacc) hacc[j] = first ? c : cadd(hacc[j], c);     // accumulated projection coefficients (device path)
    }
    if (partial_ww && blockIdx.x == 0) {                     // ||w||^2 before this pass' update (fixed order)
        __syncthreads();
        double a = 0.0;
        for (int b = threadIdx.x; b < nb; b += 256) a += partial_ww[b];
        a = wave_reduce_sum(a);
        if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6].x = a;
        __syncthreads();
        if (threadIdx.x == 0) ww[0] = (sm[0].x + sm[1].x) + (sm[2].x + sm[3].x);
    }
}

__global__ __launch_bounds__(1024) void k_orth_reduce_n(int nb, const double* __restrict__ partial,
                                                        double* __restrict__ out) {
    __shared__ double sm[16];
    double acc = 0.0;
    for (int b = threadIdx.x; b < nb; b += 1024) acc += partial[b];
    acc = wave_reduce_sum(acc);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int q = 0; q < 16; ++q) t += sm[q];
        out[0] = t;
    }
}

// w[r] -= sum_j V[r,j] h[j];  block = 8 waves x 64 rows, wave q takes columns q, q+8, ...
template <bool NT>
__global__ __launch_bounds__(512) void k_orth_update(const cplx* __restrict__ V, int64_t ldv, int64_t rows,
                                                     int k, const int64_t* __restrict__ active,
                                                     const cplx* __restrict__ h, cplx* __restrict__ w,
                                                     double* __restrict__ partial,
                                                     const int* __restrict__ gate = nullptr,
                                                     const double* __restrict__ ww = nullptr, int* __restrict__ state = nullptr,
                                                     int pdone = 0, int method = 0) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    if (gate && *gate == 0) return;
    cplx* hs = (cplx*)smem_raw;         // k
    cplx* sm = hs + k;                  // [8][64]
    const int lane = threadIdx.x & 63;
    const int q = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int t = threadIdx.x; t < k; t += 512) hs[t] = h[t];
    __syncthreads();
    // Asynchronous path: the DGKS decision of THIS pass (pass `pdone`, 1-based), published by workgroup 0 before the update has
    // run: ||w - V c||^2 = ||w||^2 - ||c||^2 for an orthonormal V, so "||w_new|| < ||c|| / sqrt(2)" reads ||w||^2 < 1.5 ||c||^2
    // (no cancellation near the threshold: the two sides differ by a factor 3 there; where the basis has lost orthogonality
    // the true norm is smaller than the estimate's and the flag in the caller's row reports what the last pass left).  The
    // kernels of the next pass read one word and leave -- the decision used to be re-formed by EVERY workgroup of the next
    // pass' k_orth_dots from the 1024 update partials (15 us per gated-off launch, 1.1 ms per gun run).
    if (state && blockIdx.x == 0 && q == 0) {
        double p2 = 0.0;
        for (int j = lane; j < k; j += 64) p2 += fma(hs[j].x, hs[j].x, hs[j].y * hs[j].y);
        p2 = wave_reduce_sum(p2);
        if (lane == 0) {
            const double w2 = ww[0];
            const int more = (method == 0 && !(w2 >= 1.5 * p2)) ? 1 : 0;
            state[1] = pdone;
            state[4 + pdone] = more;
        }
    }
    // a workgroup walks the 64-row tiles blockIdx.x, + gridDim.x, ... and leaves ONE partial norm (grid = number of tiles on
    // the synchronous path: one tile each, as before; at most ORTH_NPART workgroups on the asynchronous one)
    const int64_t ntiles = (rows + 63) / 64;
    double wg_nn = 0.0;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t r0 = tile * 64LL;
        const int64_t row = r0 + lane;
        const int64_t rowc = row < rows ? row : rows - 1;
        cplx acc = cmake(0.0, 0.0);
        const cplx* vp = V + rowc;
#pragma unroll 4
        for (int j = q; j < k; j += 8) {
            const int64_t act = active ? active[j] : rows;
            if (r0 < act) cfma(acc, vload<NT>(vp + (int64_t)j * ldv), hs[j]);
        }
        sm[q * 64 + lane] = acc;
        __syncthreads();
        if (q == 0) {
            cplx s = sm[lane];
#pragma unroll
            for (int t = 1; t < 8; ++t) s = cadd(s, sm[t * 64 + lane]);
            double nn = 0.0;
            if (row < rows) {
                cplx wn = csub(w[row], s);
                w[row] = wn;
                nn = fma(wn.x, wn.x, wn.y * wn.y);
            }
            wg_nn += wave_reduce_sum(nn);
        }
        __syncthreads();
    }
    if (q == 0 && lane == 0) partial[blockIdx.x] = wg_nn;
}


// Row-per-thread form of k_orth_update for FEW columns: a thread owns one row of a 256-row tile and walks over all k columns with
// eight loads in flight; no shared-memory reduction and no barrier per tile.  The wave-per-column-slice kernel above gives each of
// its 8 waves k / 8 columns of a 64-row tile and meets at two barriers per tile: at k <= 30 that is one or two 1 KB loads per wave
// between barriers (GMRES on the waveguide: 2.9 TB/s; iar steps 1-35: launch- and latency-bound).  Same arguments and outputs
// (one partial norm per workgroup); columns that are structurally zero on the whole tile are skipped as a leading run.
template <bool NT>
__global__ __launch_bounds__(256) void k_orth_update_rows(const cplx* __restrict__ V, int64_t ldv, int64_t rows,
                                                          int k, const int64_t* __restrict__ active,
                                                          const cplx* __restrict__ h, cplx* __restrict__ w,
                                                          double* __restrict__ partial,
                                                          const int* __restrict__ gate = nullptr,
                                                          const double* __restrict__ ww = nullptr, int* __restrict__ state = nullptr,
                                                          int pdone = 0, int method = 0) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    __shared__ double smn[4];
    if (gate && *gate == 0) return;
    cplx* hs = (cplx*)smem_raw;         // k
    const int lane = threadIdx.x & 63;
    const int q = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int t = threadIdx.x; t < k; t += 256) hs[t] = h[t];
    __syncthreads();
    if (state && blockIdx.x == 0 && q == 0) {            // the DGKS decision of this pass: see k_orth_update
        double p2 = 0.0;
        for (int j = lane; j < k; j += 64) p2 += fma(hs[j].x, hs[j].x, hs[j].y * hs[j].y);
        p2 = wave_reduce_sum(p2);
        if (lane == 0) {
            const double w2 = ww[0];
            const int more = (method == 0 && !(w2 >= 1.5 * p2)) ? 1 : 0;
            state[1] = pdone;
            state[4 + pdone] = more;
        }
    }
    const int64_t ntiles = (rows + 255) / 256;
    double nn = 0.0;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t r0 = tile * 256LL;
        const int64_t row = r0 + threadIdx.x;
        const int64_t rowc = row < rows ? row : rows - 1;
        const cplx w0 = w[rowc];
        int jmin = 0;
        if (active) {                                    // leading run of columns with no active row on this tile
            for (int base = 0; base < k; base += 64) {
                const int j = base + lane;
                const bool off = j < k && active[j] <= r0;
                const unsigned long long m = __ballot(off);
                const int run = m == ~0ull ? 64 : __ffsll((long long)~m) - 1;
                jmin = base + run;
                if (run < 64) break;
            }
            if (jmin > k) jmin = k;
            jmin = __builtin_amdgcn_readfirstlane(jmin);
        }
        cplx acc = cmake(0.0, 0.0);
        const cplx* vp = V + rowc;
        int j = jmin;
        for (; j + 8 <= k; j += 8) {
            cplx v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = vload<NT>(vp + (int64_t)(j + u) * ldv);
#pragma unroll
            for (int u = 0; u < 8; ++u) cfma(acc, v[u], hs[j + u]);
        }
        if (j < k) {
            cplx v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = vload<NT>(vp + (int64_t)(j + u < k ? j + u : k - 1) * ldv);
#pragma unroll
            for (int u = 0; u < 8; ++u) if (j + u < k) cfma(acc, v[u], hs[j + u]);
        }
        if (row < rows) {
            const cplx wn = csub(w0, acc);
            w[row] = wn;
            nn += fma(wn.x, wn.x, wn.y * wn.y);
        }
    }
    nn = wave_sum_dpp(nn);
    if (lane == 0) smn[q] = nn;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (smn[0] + smn[1]) + (smn[2] + smn[3]);
}


// w /= beta (beta on the device); records passes / flags behind beta: out[k+1] = (passes, 2*breakdown + more_needed)
// mirror (optional): device-mapped pinned host copy of the caller's row [row, row + nmirror) -- h, beta, flags and whatever the
// caller keeps behind them -- written by block 0, which saves the separate device-to-host copy command of every Arnoldi step
// dec (asynchronous path): the decision of the LAST enqueued pass (npass) is formed here by every workgroup -- when that pass ran;
// otherwise the values published by its k_orth_dots stand
__global__ __launch_bounds__(256) void k_orth_finish(int64_t rows, cplx* __restrict__ w, cplx* __restrict__ out_beta,
                                                     int* __restrict__ state, const cplx* __restrict__ row,
                                                     cplx* __restrict__ mirror, int nmirror,
                                                     const OrthDecide dec = OrthDecide(), int npass = 0) {
    double beta;
    int passes, more, brk;
    if (dec.partial) {
        // asynchronous path: the update partials in dec.partial are those of the last pass that ran (a gated-off update leaves
        // them alone); passes / "another pass wanted" were published by that pass' update
        const OrthDecide& D = dec;
        double nrm, p2;
        orth_pass_norms(D, nrm, p2);
        beta = nrm; passes = D.state[1]; more = D.state[4 + passes];
        // the flag the update published is the Pythagoras ESTIMATE of the criterion; here the true norm of the last pass that ran
        // and its ||c|| exist: the reported "another pass wanted" is the estimate OR the exact test of IterativeSolvers' DGKS loop
        // (a basis that has lost orthogonality makes the true norm smaller than the estimate)
        if (D.method == 0 && nrm < 0.70710678118654752 * sqrt(p2)) more = 1;
        brk = (!(nrm > 0.0) || !isfinite(nrm)) ? 1 : 0;
        if (blockIdx.x == 0 && threadIdx.x == 0) out_beta[0] = cmake(nrm, 0.0);
    } else { beta = out_beta[0].x; passes = state[1]; more = state[0]; brk = state[2]; }
    const double inv = (beta > 0.0 && isfinite(beta)) ? 1.0 / beta : 0.0;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < rows; i += (int64_t)gridDim.x * blockDim.x) {
        cplx v = w[i];
        w[i] = cmake(v.x * inv, v.y * inv);
    }
    if (blockIdx.x == 0) {
        const cplx flags = cmake((double)passes, (double)(2 * brk + more));
        if (threadIdx.x == 0) out_beta[1] = flags;
        if (mirror) {
            // (beta and the flags from this thread's own registers: thread 0 of this workgroup has only just stored them)
            const int iflag = (int)(out_beta + 1 - row);
            for (int i = threadIdx.x; i < nmirror; i += blockDim.x)
                mirror[i] = i == iflag ? flags : (i == iflag - 1 ? cmake(beta, 0.0) : row[i]);
        }
    }
}

// iar: the LAST kernel of step k's orthogonalisation also does the FIRST kernel of step k + 1's compute_Mlincomb.  The vector it
// normalises, v = w / beta (rows = n (k + 1)), is the basis column step k + 1 reshapes into its n x (k + 1) block y
// (method_iar.jl:96-100): the coefficient product WT[r, t] = sum_j y[r, j] C[j, t] (k_vc of csrc/spmv.hip) and the block shift
// of the column (next column, block j + 1 = block j / (j + 1)) read exactly these values.  One workgroup owns ROWS rows r of y and
// ALL k + 1 blocks of them: it forms beta like k_orth_finish (every workgroup sums the <= ORTH_NPART partial norms), scales, writes
// v back, writes the shifted block and accumulates the product -- the same operations in the same order as k_orth_finish followed by
// k_vc<MT, ROWS, true> (NG column groups, the groups summed in order), so WT, the basis and H are bit-identical to the two-kernel form.
// Step k + 1 then starts with the SpMV on WT: one dependent launch and one sweep over the column less per step.
struct OrthNextVc { const cplx* C = nullptr; int64_t ldc = 0; int mt = 0; cplx* WT = nullptr; cplx* shift_dst = nullptr; int64_t n = 0; };
template <int MT, int ROWS>
__global__ __launch_bounds__(512) void k_orth_finish_vc(int64_t n, int kb, cplx* __restrict__ w, cplx* __restrict__ out_beta,
                                                        const cplx* __restrict__ row_, cplx* __restrict__ mirror, int nmirror,
                                                        const OrthDecide dec, const cplx* __restrict__ C, int64_t ldc, int mt_total,
                                                        cplx* __restrict__ WT, cplx* __restrict__ shift_dst) {
    constexpr int NG = 512 / ROWS;
    constexpr int KC = 128;
    __shared__ cplx sm[NG][MT][ROWS];
    __shared__ cplx cs[MT][KC];
    double nrm, p2;
    orth_pass_norms(dec, nrm, p2);
    const double beta = nrm;
    const int passes = dec.state[1];
    int more = dec.state[4 + passes];
    if (dec.method == 0 && nrm < 0.70710678118654752 * sqrt(p2)) more = 1;
    const int brk = (!(nrm > 0.0) || !isfinite(nrm)) ? 1 : 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) out_beta[0] = cmake(nrm, 0.0);
    const double inv = (beta > 0.0 && isfinite(beta)) ? 1.0 / beta : 0.0;
    const int rr0 = threadIdx.x % ROWS;
    const int g = threadIdx.x / ROWS;
    const int64_t row = blockIdx.x * (int64_t)ROWS + rr0;
    const int64_t rowc = row < n ? row : n - 1;
    cplx acc[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) acc[i] = cmake(0.0, 0.0);
    cplx* vp = w + rowc;
    for (int j0 = 0; j0 < kb; j0 += KC) {
        const int kc = min(KC, kb - j0);
        if (j0 > 0) __syncthreads();
        for (int t = threadIdx.x; t < kc * MT; t += 512) {
            const int i = t / kc, j = t % kc;
            cs[i][j] = C[j0 + j + (int64_t)i * ldc];
        }
        __syncthreads();
#pragma unroll 4
        for (int j = g; j < kc; j += NG) {
            const cplx u = vp[(int64_t)(j0 + j) * n];
            const cplx v = cmake(u.x * inv, u.y * inv);
            if (row < n) {
                vp[(int64_t)(j0 + j) * n] = v;
                const double sc = 1.0 / (double)(j0 + j + 1);
                shift_dst[row + (int64_t)(j0 + j) * n] = cmake(v.x * sc, v.y * sc);
            }
#pragma unroll
            for (int i = 0; i < MT; ++i) cfma(acc[i], v, cs[i][j]);
        }
    }
#pragma unroll
    for (int i = 0; i < MT; ++i) sm[g][i][rr0] = acc[i];
    __syncthreads();
    for (int t = threadIdx.x; t < ROWS * MT; t += 512) {
        const int rr = t / MT, i = t % MT;
        cplx s = sm[0][i][rr];
#pragma unroll
        for (int q = 1; q < NG; ++q) s = cadd(s, sm[q][i][rr]);
        const int64_t r = blockIdx.x * (int64_t)ROWS + rr;
        if (r < n) WT[r * mt_total + i] = s;
    }
    if (blockIdx.x == 0) {
        const cplx flags = cmake((double)passes, (double)(2 * brk + more));
        if (threadIdx.x == 0) out_beta[1] = flags;
        if (mirror) {
            const int iflag = (int)(out_beta + 1 - row_);
            for (int i = threadIdx.x; i < nmirror; i += blockDim.x)
                mirror[i] = i == iflag ? flags : (i == iflag - 1 ? cmake(beta, 0.0) : row_[i]);
        }
    }
}
template <int ROWS>
static void launch_finish_vc(const OrthNextVc& nx, int kb, cplx* w, cplx* out_beta, const cplx* row, cplx* mirror, int nmirror,
                             const OrthDecide& D, hipStream_t st) {
    const dim3 grid((unsigned)((nx.n + ROWS - 1) / ROWS)), block(512);
    switch (nx.mt) {
        case 4: hipLaunchKernelGGL((k_orth_finish_vc<4, ROWS>), grid, block, 0, st, nx.n, kb, w, out_beta, row, mirror, nmirror, D, nx.C, nx.ldc, nx.mt, nx.WT, nx.shift_dst); break;
        case 3: hipLaunchKernelGGL((k_orth_finish_vc<3, ROWS>), grid, block, 0, st, nx.n, kb, w, out_beta, row, mirror, nmirror, D, nx.C, nx.ldc, nx.mt, nx.WT, nx.shift_dst); break;
        case 2: hipLaunchKernelGGL((k_orth_finish_vc<2, ROWS>), grid, block, 0, st, nx.n, kb, w, out_beta, row, mirror, nmirror, D, nx.C, nx.ldc, nx.mt, nx.WT, nx.shift_dst); break;
        default: hipLaunchKernelGGL((k_orth_finish_vc<1, ROWS>), grid, block, 0, st, nx.n, kb, w, out_beta, row, mirror, nmirror, D, nx.C, nx.ldc, nx.mt, nx.WT, nx.shift_dst); break;
    }
}

// column groups per launch: enough workgroups to fill 256 CUs a few times over, otherwise as few as possible
static int dots_grid_y(int nchunks, int k) {
    const int ngroups = (k + DOT_CG - 1) / DOT_CG;
    static int target = getenv("NEP_DOTS_TARGET") ? atoi(getenv("NEP_DOTS_TARGET")) : 6144;   // measured 1024: 4.79, 4096: 5.17, 8192: 5.19, 16384: 5.09 TB/s
    int gy = (target + nchunks - 1) / nchunks;
    return gy < 1 ? 1 : (gy > ngroups ? ngroups : gy);
}

#define ORTH_DOTS_LAUNCH(NT_, DPP_, GRID_, ST_, ...)                                                                   \
    do {                                                                                                               \
        if (NT_) { if (DPP_) hipLaunchKernelGGL((k_orth_dots<true, true>), GRID_, dim3(256), 0, ST_, __VA_ARGS__);       \
                   else hipLaunchKernelGGL((k_orth_dots<true, false>), GRID_, dim3(256), 0, ST_, __VA_ARGS__); }          \
        else { if (DPP_) hipLaunchKernelGGL((k_orth_dots<false, true>), GRID_, dim3(256), 0, ST_, __VA_ARGS__);          \
               else hipLaunchKernelGGL((k_orth_dots<false, false>), GRID_, dim3(256), 0, ST_, __VA_ARGS__); }             \
    } while (0)

static thread_local NepScratch g_orth_scratch;

// k at or below which the update runs row-per-thread (k_orth_update_rows); NEP_ORTH_ROWS_K=0 switches that form off.  Measured alone
// the two forms meet at k = 100 (310 against 313 us per pass at iar's last step) and the row form wins below (k = 48: 81 / 88 us,
// k = 8 at 10^6 rows: 62 / 67); inside the gun run, next to the eigenvalue kernels of the other queue, the row form is faster at
// every k (whole call 37.7 ms against 39.3 with the switch at 64 and 39.7 without it), so it is the default for all k
static int orth_rows_k() {
    static const int v = getenv("NEP_ORTH_ROWS_K") ? atoi(getenv("NEP_ORTH_ROWS_K")) : 1 << 30;
    return v;
}

// non-temporal V loads when the streamed block is far larger than the last-level cache (see nep_orth_dev)
static bool orth_use_nt(int64_t rows, int64_t k, bool staircase) {
    static const int nt_env = getenv("NEP_ORTH_NT") ? atoi(getenv("NEP_ORTH_NT")) : -1;
    static const double nt_mb = getenv("NEP_ORTH_NT_MB") ? atof(getenv("NEP_ORTH_NT_MB")) : 192.0;
    // full columns (GMRES / tiar bases; no factorisation to protect next to them): the update's sweep over V follows the
    // projection's at once and finds a block of up to ~1.5x the last-level cache largely still there -- measured at 10^6 rows,
    // one pass: k = 12 103 -> 81 us, k = 16 114 -> 100, k = 20 146 -> 127, k = 24 158 -> 152 with ordinary loads
    static const double nt_full_mb = getenv("NEP_ORTH_NT_FULL_MB") ? atof(getenv("NEP_ORTH_NT_FULL_MB")) : 512.0;
    const double streamed_mb = 16.0e-6 * (double)rows * (double)k * (staircase ? 0.5 : 1.0);
    return nt_env >= 0 ? nt_env != 0 : streamed_mb > (staircase ? nt_mb : nt_full_mb);
}

extern "C" int32_t nep_orth(const nep_cdouble* dV, int64_t ldv, int64_t rows, int32_t k,
                            const int64_t* h_active_rows, nep_cdouble* dw, nep_cdouble* h_h, double* h_beta,
                            int32_t method, int32_t* h_npasses, nep_stream stream) {
    ARGCHK(dV && dw && h_h && h_beta);
    ARGCHK(rows > 0 && k >= 1 && ldv >= rows);
    ARGCHK(method >= 0 && method <= 2);
    hipStream_t st = as_stream(stream);
    const int nchunks = (int)((rows + DOT_RB - 1) / DOT_RB);
    const int nblk = (int)((rows + 63) / 64);
    // scratch layout: [active k int64][partial_h nchunks*k cplx][h k cplx | nrm2 double pad][partial_n nblk dbl]
    size_t off_act = 0;
    size_t off_ph = off_act + (size_t)k * sizeof(int64_t);
    off_ph = (off_ph + 15) & ~(size_t)15;
    size_t off_h = off_ph + (size_t)nchunks * k * sizeof(cplx);
    size_t off_pn = off_h + (size_t)(k + 1) * sizeof(cplx);
    size_t total = off_pn + (size_t)nblk * sizeof(double);
    int rc = g_orth_scratch.ensure(total);
    if (rc) return rc;
    char* base = (char*)g_orth_scratch.dptr;
    int64_t* d_act = h_active_rows ? (int64_t*)(base + off_act) : nullptr;
    cplx* d_ph = (cplx*)(base + off_ph);
    cplx* d_h = (cplx*)(base + off_h);
    double* d_n = (double*)(d_h + k);
    double* d_pn = (double*)(base + off_pn);
    if (h_active_rows)
        HIPCHK(hipMemcpyAsync(d_act, h_active_rows, (size_t)k * sizeof(int64_t), hipMemcpyHostToDevice, st));
    std::vector<nep_cdouble> corr(k + 1);
    for (int j = 0; j < k; ++j) { h_h[j].re = 0.0; h_h[j].im = 0.0; }
    const cplx* V = (const cplx*)dV;
    cplx* w = (cplx*)dw;
    double nrm = 0.0;
    int passes = 0;
    const size_t shm_upd = (size_t)(k + 8 * 64) * sizeof(cplx);

    if (method == 2) {
        // modified Gram-Schmidt: column by column (test/reference-comparison path; not tuned)
        for (int j = 0; j < k; ++j) {
            const int64_t* actj = d_act ? d_act + j : nullptr;
            hipLaunchKernelGGL((k_orth_dots<false, false>), dim3(nchunks, 1), dim3(256), 0, st, V + (int64_t)j * ldv, ldv, rows, 1,
                               actj, (const cplx*)w, d_ph);
            LAUNCHCHK();
            hipLaunchKernelGGL(k_orth_reduce_h, dim3(1), dim3(256), 0, st, nchunks, 1, (const cplx*)d_ph, d_h + j);
            LAUNCHCHK();
            hipLaunchKernelGGL(k_orth_update<false>, dim3(nblk), dim3(512), (1 + 8 * 64) * sizeof(cplx), st,
                               V + (int64_t)j * ldv, ldv, rows, 1, actj, (const cplx*)(d_h + j), w, d_pn);
            LAUNCHCHK();
        }
        hipLaunchKernelGGL(k_orth_reduce_n, dim3(1), dim3(1024), 0, st, nblk, (const double*)d_pn, d_n);
        LAUNCHCHK();
        HIPCHK(hipMemcpyAsync(corr.data(), d_h, (size_t)(k + 1) * sizeof(cplx), hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        for (int j = 0; j < k; ++j) h_h[j] = corr[j];
        nrm = sqrt(corr[k].re);
        passes = 1;
    } else {
        const double eta = 1.0 / sqrt(2.0);
        const bool nt = orth_use_nt(rows, k, d_act != nullptr);
        while (true) {
            ORTH_DOTS_LAUNCH(nt, rows >= ORTH_DPP_ROWS, dim3(nchunks, dots_grid_y(nchunks, k)), st, V, ldv,
                             rows, (int)k, (const int64_t*)d_act, (const cplx*)w, d_ph);
            LAUNCHCHK();
            hipLaunchKernelGGL(k_orth_reduce_h, dim3(k), dim3(256), 0, st, nchunks, (int)k, (const cplx*)d_ph, d_h);
            LAUNCHCHK();
            if (nt)
                hipLaunchKernelGGL(k_orth_update<true>, dim3(nblk), dim3(512), shm_upd, st, V, ldv, rows, (int)k,
                                   (const int64_t*)d_act, (const cplx*)d_h, w, d_pn);
            else
                hipLaunchKernelGGL(k_orth_update<false>, dim3(nblk), dim3(512), shm_upd, st, V, ldv, rows, (int)k,
                                   (const int64_t*)d_act, (const cplx*)d_h, w, d_pn);
            LAUNCHCHK();
            hipLaunchKernelGGL(k_orth_reduce_n, dim3(1), dim3(1024), 0, st, nblk, (const double*)d_pn, d_n);
            LAUNCHCHK();
            HIPCHK(hipMemcpyAsync(corr.data(), d_h, (size_t)(k + 1) * sizeof(cplx), hipMemcpyDeviceToHost, st));
            HIPCHK(hipStreamSynchronize(st));
            ++passes;
            double proj2 = 0.0;
            for (int j = 0; j < k; ++j) {
                h_h[j].re += corr[j].re; h_h[j].im += corr[j].im;
                proj2 += corr[j].re * corr[j].re + corr[j].im * corr[j].im;
            }
            nrm = sqrt(corr[k].re);
            if (method == 1) break;                       // classical GS: single pass
            if (!(nrm < eta * sqrt(proj2))) break;        // DGKS criterion
            if (passes >= 8) break;                       // safety net (never met in practice)
        }
    }
    if (h_npasses) *h_npasses = passes;
    *h_beta = nrm;
    if (!(nrm > 0.0) || !isfinite(nrm)) {
        nep_set_error("orthogonalisation breakdown: ||w|| = %g", nrm);
        return NEP_ERR_BREAKDOWN;
    }
    nep_cdouble inv; inv.re = 1.0 / nrm; inv.im = 0.0;
    return nep_scal(rows, inv, dw, stream);
}

// Fully asynchronous DGKS/CGS: no host synchronisation.  The re-orthogonalisation passes are always enqueued and
// switch themselves off through a device flag (criterion ||w|| < ||c||/sqrt(2) evaluated inside the next pass' k_orth_dots / k_orth_finish), at most
// orth_dev_passes() passes.  d_out (k+2 complex, device): h[0..k), (beta,0), (passes, 2*breakdown + more_needed).
// "Twice is enough" (Kahan/Parlett): after the second pass w is orthogonal to machine precision unless it lies
// numerically inside span(V), which the breakdown flag reports.  The default therefore enqueues 2 passes (every enqueued
// pass costs three launches even when its gate is closed: 3 -> 2 passes took 1.4 ms off the 100 steps of the gun run);
// NEP_ORTH_DEV_PASSES=3.. restores the longer chain; the `another_pass_wanted` flag in d_out tells if the criterion still
// held after the last enqueued pass.
static int orth_dev_passes() {
    static int np = 0;
    if (!np) { const char* e = getenv("NEP_ORTH_DEV_PASSES"); np = e ? atoi(e) : 2; if (np < 1 || np > 8) np = 2; }
    return np;
}
extern "C" int32_t nep_orth_dev(const nep_cdouble* dV, int64_t ldv, int64_t rows, int32_t k,
                                const int64_t* d_active_rows, nep_cdouble* dw, nep_cdouble* d_out, int32_t method,
                                nep_stream stream) {
    return nep_orth_dev_mirror(dV, ldv, rows, k, d_active_rows, dw, d_out, method, nullptr, 0, stream);
}
// d_mirror / nmirror: see k_orth_finish (internal: nep_iar_step)
extern "C" int32_t nep_orth_dev_mirror(const nep_cdouble* dV, int64_t ldv, int64_t rows, int32_t k,
                                       const int64_t* d_active_rows, nep_cdouble* dw, nep_cdouble* d_out, int32_t method,
                                       nep_cdouble* d_mirror, int32_t nmirror, nep_stream stream) {
    return nep_orth_dev_mirror_ev(dV, ldv, rows, k, d_active_rows, dw, d_out, method, d_mirror, nmirror, nullptr, stream);
}
// before_write (may be NULL): an event the stream waits for before the first kernel that WRITES w (the first update): work on
// another stream that still reads w -- iar's recorded residual of the kept iterate -- runs next to the projections
static int orth_dev_impl(const nep_cdouble* dV, int64_t ldv, int64_t rows, int32_t k,
                         const int64_t* d_active_rows, nep_cdouble* dw, nep_cdouble* d_out, int32_t method,
                         nep_cdouble* d_mirror, int32_t nmirror, void* before_write, const OrthNextVc* next, nep_stream stream);
extern "C" int32_t nep_orth_dev_mirror_ev(const nep_cdouble* dV, int64_t ldv, int64_t rows, int32_t k,
                                          const int64_t* d_active_rows, nep_cdouble* dw, nep_cdouble* d_out, int32_t method,
                                          nep_cdouble* d_mirror, int32_t nmirror, void* before_write, nep_stream stream) {
    return orth_dev_impl(dV, ldv, rows, k, d_active_rows, dw, d_out, method, d_mirror, nmirror, before_write, nullptr, stream);
}
// iar's form: rows = n (k + 1); the last kernel also forms step k + 1's coefficient product d_WT (n x mt, row-major) from the
// normalised vector and the coefficient table dC (ldc), and writes the block shift of the vector to d_shift (see k_orth_finish_vc)
extern "C" int32_t nep_orth_dev_iar_next(const nep_cdouble* dV, int64_t ldv, int64_t n, int32_t k, const int64_t* d_active_rows,
                                         nep_cdouble* dw, nep_cdouble* d_out, int32_t method, nep_cdouble* d_mirror, int32_t nmirror,
                                         void* before_write, const nep_cdouble* dC, int64_t ldc, int32_t mt, nep_cdouble* d_WT,
                                         nep_cdouble* d_shift, nep_stream stream) {
    ARGCHK(dC && d_WT && d_shift && mt >= 1 && mt <= 4 && ldc >= k + 1 && n > 0);
    OrthNextVc nx; nx.C = (const cplx*)dC; nx.ldc = ldc; nx.mt = mt; nx.WT = (cplx*)d_WT; nx.shift_dst = (cplx*)d_shift; nx.n = n;
    return orth_dev_impl(dV, ldv, n * (int64_t)(k + 1), k, d_active_rows, dw, d_out, method, d_mirror, nmirror, before_write, &nx, stream);
}
static int orth_dev_impl(const nep_cdouble* dV, int64_t ldv, int64_t rows, int32_t k,
                         const int64_t* d_active_rows, nep_cdouble* dw, nep_cdouble* d_out, int32_t method,
                         nep_cdouble* d_mirror, int32_t nmirror, void* before_write, const OrthNextVc* next, nep_stream stream) {
    ARGCHK(dV && dw && d_out);
    ARGCHK(rows > 0 && k >= 1 && ldv >= rows);
    ARGCHK(method == 0 || method == 1);
    hipStream_t st = as_stream(stream);
    const int nchunks = (int)((rows + DOT_RB - 1) / DOT_RB);
    const int nblk = (int)((rows + 63) / 64);
    static const int npart_max = getenv("NEP_ORTH_NPART") ? std::max(64, atoi(getenv("NEP_ORTH_NPART"))) : ORTH_NPART;
    const int npart = std::min(nblk, npart_max);
    const int npass = method == 1 ? 1 : orth_dev_passes();
    // (a variant that formed the second pass' projections inside the first update -- one sweep over V instead of two in the 24 %
    // of the gun steps that re-orthogonalise -- was built in round 3 and measured SLOWER on the headline run, 45.0 against 42.2 ms
    // per call: the tile's V values held in registers cost every first update more than the saved k_orth_dots; removed in round 4)
    // scratch: [state 16 int][partial_h nchunks*k cplx][c k cplx][c2 k cplx][partial_n npart dbl][partial_c2 npart*k cplx]
    size_t off_ph = 64;
    size_t off_c = off_ph + (size_t)nchunks * k * sizeof(cplx);
    size_t off_c2 = off_c + (size_t)k * sizeof(cplx);
    size_t off_pn = off_c2 + (size_t)k * sizeof(cplx);
    size_t off_pc2 = (off_pn + (size_t)npart * sizeof(double) + 15) & ~(size_t)15;
    size_t off_pww = off_pc2;
    size_t total = off_pww + ((size_t)nchunks + 2) * sizeof(double);
    int rc = g_orth_scratch.ensure(total);
    if (rc) return rc;
    char* base = (char*)g_orth_scratch.dptr;
    int* d_state = (int*)base;
    cplx* d_ph = (cplx*)(base + off_ph);
    cplx* d_c = (cplx*)(base + off_c);
    (void)off_c2;
    double* d_pn = (double*)(base + off_pn);

    double* d_pww = (double*)(base + off_pww);
    double* d_ww = d_pww + nchunks;
    const cplx* V = (const cplx*)dV;
    cplx* w = (cplx*)dw;
    cplx* out = (cplx*)d_out;
    const size_t shm_upd = (size_t)(k + 8 * 64) * sizeof(cplx);
    // non-temporal V loads when the block that is streamed (iar: the non-zero staircase, about half of rows x k) is far larger
    // than the last-level cache: NEP_ORTH_NT = 0 never, 1 always, unset: above NEP_ORTH_NT_MB (default 192) megabytes
    const bool nt = orth_use_nt(rows, k, d_active_rows != nullptr);
    OrthDecide D;
    const bool rows_form = k <= orth_rows_k();
    const int nb_rows = (int)std::min<int64_t>((rows + 255) / 256, npart);
    D.partial = d_pn; D.np = rows_form ? nb_rows : npart; D.c = d_c; D.k = (int)k; D.method = (int)method; D.state = d_state; D.out_beta = out + k;
    {
    // per pass three launches (dots, coefficient reduction, update); the decision of pass p is published by its update kernel
    // BEFORE that update runs (Pythagoras, see k_orth_update), so a gated-off pass costs three launches that read one word
    for (int p = 0; p < npass; ++p) {
        const int* gate = p == 0 ? nullptr : d_state + 4 + p;       // "pass p ran and wants pass p + 1"
        ORTH_DOTS_LAUNCH(nt, rows >= ORTH_DPP_ROWS, dim3(nchunks, dots_grid_y(nchunks, k)), st, V, ldv, rows, (int)k,
                         d_active_rows, (const cplx*)w, d_ph, gate, d_pww);
        LAUNCHCHK();
        hipLaunchKernelGGL(k_orth_reduce_h, dim3(k), dim3(256), 0, st, nchunks, (int)k, (const cplx*)d_ph, d_c, gate, out,
                           p == 0 ? 1 : 0, p == 0 ? d_state : (int*)nullptr, OrthDecide(), 0, (const double*)d_pww, d_ww);
        LAUNCHCHK();
        if (p == 0 && before_write) HIPCHK(hipStreamWaitEvent(st, (hipEvent_t)before_write, 0));
        if (rows_form) {
            const int nb = nb_rows;
            if (nt)
                hipLaunchKernelGGL(k_orth_update_rows<true>, dim3(nb), dim3(256), (size_t)k * sizeof(cplx), st, V, ldv, rows, (int)k,
                                   d_active_rows, (const cplx*)d_c, w, d_pn, gate, (const double*)d_ww, d_state, p + 1, (int)method);
            else
                hipLaunchKernelGGL(k_orth_update_rows<false>, dim3(nb), dim3(256), (size_t)k * sizeof(cplx), st, V, ldv, rows, (int)k,
                                   d_active_rows, (const cplx*)d_c, w, d_pn, gate, (const double*)d_ww, d_state, p + 1, (int)method);
        } else if (nt)
            hipLaunchKernelGGL(k_orth_update<true>, dim3(npart), dim3(512), shm_upd, st, V, ldv, rows, (int)k, d_active_rows,
                               (const cplx*)d_c, w, d_pn, gate, (const double*)d_ww, d_state, p + 1, (int)method);
        else
            hipLaunchKernelGGL(k_orth_update<false>, dim3(npart), dim3(512), shm_upd, st, V, ldv, rows, (int)k, d_active_rows,
                               (const cplx*)d_c, w, d_pn, gate, (const double*)d_ww, d_state, p + 1, (int)method);
        LAUNCHCHK();
    }
    }
    if (next) {
        if (next->n >= 65536) launch_finish_vc<64>(*next, (int)k + 1, w, out + k, (const cplx*)out, (cplx*)d_mirror, (int)nmirror, D, st);
        else launch_finish_vc<32>(*next, (int)k + 1, w, out + k, (const cplx*)out, (cplx*)d_mirror, (int)nmirror, D, st);
        LAUNCHCHK();
        return NEP_OK;
    }
    const int g = (int)std::min<int64_t>((rows + 255) / 256, 2048);
    hipLaunchKernelGGL(k_orth_finish, dim3(g), dim3(256), 0, st, rows, w, out + k, d_state, (const cplx*)out,
                       (cplx*)d_mirror, (int)nmirror, D, npass);
    LAUNCHCHK();
    return NEP_OK;
}

// first column of a thin QR: beta = ||w||, w /= beta; out[0] = (beta, 0), out[1] = (1 pass, 2 * breakdown).  One workgroup.
__global__ __launch_bounds__(1024) void k_qr_first(int64_t rows, cplx* __restrict__ w, cplx* __restrict__ out) {
    __shared__ double sm[16];
    double a = 0.0;
    for (int64_t i = threadIdx.x; i < rows; i += 1024) a += fma(w[i].x, w[i].x, w[i].y * w[i].y);
    a = wave_reduce_sum(a);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = a;
    __syncthreads();
    double t = 0.0;
    for (int q = 0; q < 16; ++q) t += sm[q];
    const double nrm = sqrt(t);
    const int brk = (!(nrm > 0.0) || !isfinite(nrm)) ? 1 : 0;
    const double inv = brk ? 0.0 : 1.0 / nrm;
    for (int64_t i = threadIdx.x; i < rows; i += 1024) w[i] = cmake(w[i].x * inv, w[i].y * inv);
    if (threadIdx.x == 0) { out[0] = cmake(nrm, 0.0); out[1] = cmake(1.0, (double)(2 * brk)); }
}
// Thin QR of a tall block by column-wise DGKS, all on the device and without a host synchronisation: column j of dQ (rows x k,
// column-major, ld ldq) is orthogonalised against the columns before it and normalised in place (nep_orth_dev); row j of d_out
// (k rows of k + 2 complex) receives h[0..j) = R[0..j, j], (beta, 0) = R[j, j], and the (passes, flags) word of nep_orth_dev.
// A column inside the span of its predecessors leaves a tiny R[j, j] and a normalised noise vector -- the caller reads the rank off
// R.  (Beyn's method: the SVD of the n x k moment block becomes svd(R), method_beyncontour.jl:114-128.)
extern "C" int32_t nep_orth_qr_dev(nep_cdouble* dQ, int64_t ldq, int64_t rows, int32_t k, nep_cdouble* d_out, nep_stream stream) {
    ARGCHK(dQ && d_out && rows > 0 && k >= 1 && ldq >= rows);
    hipLaunchKernelGGL(k_qr_first, dim3(1), dim3(1024), 0, as_stream(stream), rows, (cplx*)dQ, (cplx*)d_out);
    LAUNCHCHK();
    for (int32_t j = 1; j < k; ++j) {
        const int rc = nep_orth_dev(dQ, ldq, rows, j, nullptr, dQ + (int64_t)j * ldq, d_out + (int64_t)j * (k + 2), 0, stream);
        if (rc) return rc;
    }
    return NEP_OK;
}

// h = V^H w  (rows x k block, no update of w): the projection products W^H (A_i v) of Proj_SPMF_NEP
// (src/NEPTypes.jl:724-790) and Gram matrices.  Synchronous (k host results).
extern "C" int32_t nep_gemv_h(const nep_cdouble* dV, int64_t ldv, int64_t rows, int32_t k, const nep_cdouble* dw,
                              nep_cdouble* h_h, nep_stream stream) {
    ARGCHK(dV && dw && h_h);
    ARGCHK(rows > 0 && k >= 1 && ldv >= rows);
    hipStream_t st = as_stream(stream);
    const int nchunks = (int)((rows + DOT_RB - 1) / DOT_RB);
    size_t off_h = (size_t)nchunks * k * sizeof(cplx);
    int rc = g_orth_scratch.ensure(off_h + (size_t)k * sizeof(cplx));
    if (rc) return rc;
    cplx* d_ph = (cplx*)g_orth_scratch.dptr;
    cplx* d_h = (cplx*)((char*)g_orth_scratch.dptr + off_h);
    ORTH_DOTS_LAUNCH(false, rows >= ORTH_DPP_ROWS, dim3(nchunks, dots_grid_y(nchunks, k)), st, (const cplx*)dV, ldv, rows,
                     (int)k, (const int64_t*)nullptr, (const cplx*)dw, d_ph);
    LAUNCHCHK();
    hipLaunchKernelGGL(k_orth_reduce_h, dim3(k), dim3(256), 0, st, nchunks, (int)k, (const cplx*)d_ph, d_h);
    LAUNCHCHK();
    std::vector<nep_cdouble> tmp(k);
    HIPCHK(hipMemcpyAsync(tmp.data(), d_h, (size_t)k * sizeof(cplx), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    for (int j = 0; j < k; ++j) h_h[j] = tmp[j];
    return NEP_OK;
}
