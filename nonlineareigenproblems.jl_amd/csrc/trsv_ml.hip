// libnepmi355: K5 fixed-shift solve, elimination-tree block schedule (gfx950).
//
// Replaces the dependent level sweep of a sparse triangular solve (gun: 515 levels per factor; src/LinSolvers.jl:125-137,
// `Afact \ x`) by a handful of bandwidth-bound launches.  Pr*A*Pc = L*U comes from the host (one-off per shift,
// src/LinSolvers.jl:114-116).  Symbolic part, once per sparsity pattern (cached, see MLCache):
//
//   1. elimination tree of the symmetrised pattern struct(L) + struct(U)^T (Liu's algorithm): every dependency of the
//      forward and of the backward substitution points from a node to one of its ancestors;
//   2. multilevel partition of the tree: level 0 = the maximal subtrees with at most `bmax` nodes, level 1 = the maximal
//      subtrees of what is left, ... (gun, n = 9956, bmax = 256: 51 + 10 + 2 + 1 blocks in 4 levels).  Blocks of one
//      level are independent of each other and depend on earlier levels only (L) / later levels only (U);
//   3. a symmetric permutation that makes every block a contiguous row range, levels in order.
//
// Numeric part, per factorisation: the diagonal blocks L_BB, U_BB (at most bmax x bmax, sparse) are inverted explicitly on
// the device (k_ml_inverse: one workgroup per column, x in LDS, in-block level schedule) and stored as packed dense
// triangular rows; everything outside the diagonal blocks stays sparse ("coupling" CSR).  One solve is then, per level,
//
//      L:  y_B = inv(L_BB) (b_B - L[B, earlier] y)          U:  x_B = inv(U_BB) (y_B - U[B, later] x)
//
// i.e. ONE launch per level and factor (k_ml_level; the coupling product is formed redundantly per row chunk in LDS) or
// two when the coupling rows are long (k_ml_coupling + k_ml_level<MODE 1>).  The input permutation is folded into the
// first launch and the output permutation / scaling / refinement update into the last one.  Dependent steps per solve:
// gun 23 launches (0.139 ms) -> 2 * nlev (+ split levels).
#include "common.h"
#include "trsv_ml.h"
#include <vector>
#include <algorithm>
#include <mutex>
#include <list>
#include <chrono>
#include <cstring>

#define ML_BLK_SEG 64          // rows of a diagonal block per workgroup of k_ml_level_blk (a multiple of every chunk size)
#define ML_BMAX 256            // largest diagonal block (rows): LDS staging of the fused level kernel is sized for it

struct MLChunk { int32_t a, b, s, e; int64_t ipa; };   // rows [a,b) of the block with rows [s,e); ipa = offset of row a's packed inverse row

struct MLFacSym {
    // device, symbolic
    int32_t* d_cp = nullptr;       // n+1  coupling CSR (new row order; columns in new numbering)
    int32_t* d_ci = nullptr;
    int64_t* d_ip = nullptr;       // n+1  offsets of the packed inverse rows
    int32_t* d_bp = nullptr;       // n+1  in-block CSR over slots (rows of a block sorted by in-block level)
    int32_t* d_bi = nullptr;       //      local column (col - block start)
    int32_t* d_slotrow = nullptr;  // n    slot -> local row
    int32_t* d_lvp = nullptr;      //      in-block level pointers (slot positions), block k: [lvo[k], lvo[k+1])
    int32_t* d_lvo = nullptr;      // nblk+1
    int32_t* d_rowlev = nullptr;   // n    in-block level of every row (k_ml_inverse starts column j at the level of row j)
    // host
    std::vector<int32_t> map;      // input entry -> slot*4 + kind   (kind 0 skip, 1 coupling, 2 in-block, 3 diagonal)
    int32_t* d_map = nullptr;      // the same on the device (uploaded on first use by the device-side numeric path)
    int64_t ncoup = 0, nin = 0, ninv = 0;
    MLChunk* d_chunks = nullptr;   // row chunks of the level kernels (chunk size depends on the level's mode)
    int32_t* d_segs = nullptr;     // k_ml_level_blk: the chunks (index inside their level) that start a 64-row segment of a block
    std::vector<int32_t> lev_seg;  // nlev+1
    std::vector<int32_t> lev_chunk;    // nlev+1
    std::vector<int> lev_ch;       // per level: rows per chunk (4, 16 or 32)
    std::vector<uint8_t> split;    // per level: coupling product as its own launch
    std::vector<int> cpl_lanes;    // per level: lanes per row of that launch (8, 64, 256)
    std::vector<int64_t> lev_coup; // per level: coupling non-zeros
};

struct MLSym {
    int64_t n = 0;
    int nlev = 0, nblk = 0;
    std::vector<int32_t> lev_row, lev_blk;       // nlev+1 each: row / block range of a level
    int32_t* d_blk_se = nullptr;   // 2*nblk
    int32_t* d_rowblk = nullptr;   // n
    int32_t* d_pin = nullptr;      // n: new row q reads b[pin[q]]
    int32_t* d_pout = nullptr;     // n: new row q writes X[pout[q]]
    MLFacSym L, U;
    uint64_t key0 = 0, key1 = 0;
    int64_t nnzL = 0, nnzU = 0;
    int refs = 0;
    int csc = 0;
    int max_block = 0;
    // partition in the factor's own (input) numbering, for the device-side numeric factorisation (csrc/lufac.hip)
    std::vector<int32_t> h_lvl, h_blk, h_oldof, h_blk_se;
    double t_build_ms = 0.0;
    // dense apex build: K range of every 64-row tile of a level's block-diagonal factor (2 ints per tile), levels >= apex_kr_la
    int32_t* d_apex_kr = nullptr; int apex_kr_la = -1; std::vector<int32_t> apex_kr_off;
    // ... and its 9 T^2 workspace (253 MB on gun), kept with the pattern: one build at a time uses it, ordered by an event (a block
    // taken from the pool per factorisation and freed behind its stream made the pool's footprint depend on the timing of the calls)
    cplx* d_apex_work = nullptr; int64_t apex_work_T = 0; hipEvent_t apex_work_ev = nullptr; bool apex_work_used = false;
};

struct MLFactor {
    MLSym* sym = nullptr;
    cplx* d_vals = nullptr;        // [cxL | bxL | cxU | bxU | diagU]
    cplx* d_ixL = nullptr;
    cplx* d_ixU = nullptr;
    cplx* d_Sinv = nullptr;        // apex: dense row-major inverse of the rows of levels >= apex_la (0 = no apex)
    int apex_la = 0;
    // the apex is built BEHIND the `ready` event: solves that arrive before it is finished walk the apex levels like any
    // other level (22 us more per gun solve) instead of waiting 3.3 ms for it; apex_live flips when apex_ev has completed
    hipEvent_t apex_ev = nullptr; bool apex_live = false;
    int solves_since_numeric = 0;      // the switch to the apex happens at a FIXED solve of a factor (NEP_ML_APEX_AT), not when a query says so
    double* d_rscale = nullptr;    // optional row scaling (UMFPACK's Rs): b is multiplied by it on the way in
    NepScratch work;               // bw | y | x | tmp, each n*nrhs
    hipEvent_t ready = nullptr;    // numeric build complete (recorded on the build stream)
    hipStream_t synced = nullptr;  // stream that has already waited for `ready`
    bool synced_valid = false;
    hipStream_t last = nullptr;    // stream of the last solve (frees are ordered behind it)
    bool used = false;
    hipGraphExec_t graph = nullptr;
    int graph_nrhs = 0; void* graph_work = nullptr;
    hipStream_t cap_stream = nullptr;
    int use_graph = 1;
    int launches = 0;
    void* pinned = nullptr; size_t pinned_cap = 0;
    // single-launch form (k_ml_fused): phase counters on the device, launches issued so far, device-mapped error flag
    unsigned long long* d_fctr = nullptr; unsigned long long fuse_epoch = 0; int* h_ferr = nullptr; int* d_ferr = nullptr; int fuse_off = 0;
};

static double ml_now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

__device__ __forceinline__ cplx ml_cdiv(cplx a, cplx b) {
    if (fabs(b.x) >= fabs(b.y)) {
        const double r = b.y / b.x, d = b.x + b.y * r;
        return cmake((a.x + a.y * r) / d, (a.y - a.x * r) / d);
    } else {
        const double r = b.x / b.y, d = b.x * r + b.y;
        return cmake((a.x * r + a.y) / d, (a.y * r - a.x) / d);
    }
}

// ---- numeric set-up: column j of inv(L_BB) / inv(U_BB) for every diagonal block --------------------------------------
// grid.x = n (one workgroup per column of the block-diagonal inverse), 16 lanes per row, x in LDS
// threads per column of k_ml_inverse: ONE wave (4 slots of 16 lanes per batch).  A dense block has one row per in-block level, so
// 256 threads mostly waited at the barrier of every level (9956 columns x ~100 levels); with one wave per column the barrier is free
// and four times as many columns are resident (measured: 0.55 + 0.61 ms -> see DESIGN.md section 3 K5 set-up)
#define ML_INV_NT 64
struct MLBatchTab { cplx* const* vals = nullptr; cplx* const* ix = nullptr; int64_t off_bx = 0, off_diag = 0; };
template <bool UPPER>
__global__ __launch_bounds__(ML_INV_NT) void k_ml_inverse(const int32_t* __restrict__ rowblk, const int32_t* __restrict__ blk_se,
                                                    const int32_t* __restrict__ lvo, const int32_t* __restrict__ lvp,
                                                    const int32_t* __restrict__ slotrow, const int32_t* __restrict__ bp,
                                                    const int32_t* __restrict__ bi, const cplx* bx,
                                                    const cplx* diag, const int64_t* __restrict__ ip,
                                                    cplx* ix, const int32_t* __restrict__ rowlev,
                                                    const MLBatchTab tab = MLBatchTab()) {
    // batch of factors of one pattern (ml_create_from_sym_batch: the nodes of contour_beyn): grid.y = factor; its value block and its
    // packed-inverse block come from the pointer table, bx / diag are then OFFSETS into the value block
    if (tab.vals) {
        cplx* vb = tab.vals[blockIdx.y];
        bx = vb + tab.off_bx; diag = UPPER ? vb + tab.off_diag : nullptr;
        ix = tab.ix[blockIdx.y];
    }
    __shared__ cplx x[ML_BMAX];
    // round 3: the block's level pointers, slot rows and row pointers sit in LDS (one coalesced load each at the start), and the
    // first entry of every lane for the NEXT batch of 16 slots is fetched while the current batch is reduced -- a dense 256-row block
    // has 256 in-block levels, and each used to cost two dependent global round trips (row pointers -> entries) before its barrier
    __shared__ int32_t s_lvp[ML_BMAX + 2], s_row[ML_BMAX + 1], s_bp[ML_BMAX + 2];
    const int q = blockIdx.x;
    const int k = rowblk[q];
    const int s = blk_se[2 * k], e = blk_se[2 * k + 1];
    const int j = q - s, bsz = e - s;
    const int l0 = lvo[k], nlev = lvo[k + 1] - l0 - 1;
    for (int t = threadIdx.x; t < bsz; t += ML_INV_NT) x[t] = cmake(t == j ? 1.0 : 0.0, 0.0);
    for (int t = threadIdx.x; t <= nlev; t += ML_INV_NT) s_lvp[t] = lvp[l0 + t];
    const int slot0 = lvp[l0], nslot = lvp[l0 + nlev] - slot0;          // the block's slots are contiguous, level after level
    for (int t = threadIdx.x; t < nslot; t += ML_INV_NT) s_row[t] = slotrow[slot0 + t];
    for (int t = threadIdx.x; t <= nslot; t += ML_INV_NT) s_bp[t] = bp[slot0 + t];
    __syncthreads();
    const int sub = threadIdx.x & 15, grp = threadIdx.x >> 4;
    // column j of the inverse is zero in every row that does not depend on row j, i.e. in all rows of lower in-block levels
    // (and, for the unit-lower factor, of row j's own level): the substitution starts at the level of row j -- half of the
    // levels of a dense block on average -- and leaves bit-identical values (the skipped rows computed 0 - 0)
    const int lstart = UPPER ? rowlev[q] : rowlev[q] + 1;
    // batch = ML_INV_NT / 16 consecutive slots of one level.  The lane's first entry of a batch is fetched FOUR batches ahead (four
    // rotating register sets, the loop body replicated four times): a level of a dense block is one short batch, shorter than a
    // global round trip, so a look-ahead of one batch still left one round trip per level (0.36 + 0.41 ms with it, this form: see
    // DESIGN.md section 3 K5 set-up)
    // A level with a single row (every level of a dense block) is taken by all 64 lanes of the wave instead of 16: such rows have up
    // to 255 entries, and 16 trips of 16 lanes were 4 grouped round trips per level.
    struct St { int i, p, e1, col, last, valid, wide; cplx v; };
    int plev = lstart, psl0 = plev < nlev ? s_lvp[plev] - slot0 : 0;       // prefetch cursor
    auto issue = [&](St& S) __attribute__((always_inline)) {
        S.i = -1; S.p = 0; S.e1 = 0; S.col = 0; S.last = 0; S.wide = 0; S.valid = plev < nlev; S.v = cmake(0.0, 0.0);
        if (!S.valid) return;
        const int lend = s_lvp[plev + 1] - slot0;
        S.wide = (ML_INV_NT == 64 && lend - psl0 == 1) ? 1 : 0;          // (psl0 is the level's first slot whenever one slot remains and ...)
        const int sl = S.wide ? psl0 : psl0 + grp;
        if (sl < lend) {
            const int i = s_row[sl];
            if (UPPER ? (i <= j) : (i > j)) {                   // the other rows of the column stay zero
                S.i = i; S.p = s_bp[sl] + (S.wide ? (int)threadIdx.x : sub); S.e1 = s_bp[sl + 1];
                if (S.p < S.e1) { S.v = bx[S.p]; S.col = bi[S.p]; }
            }
        }
        psl0 += ML_INV_NT / 16;
        if (psl0 >= lend) { S.last = 1; ++plev; psl0 = plev < nlev ? s_lvp[plev] - slot0 : 0; }
    };
    auto consume = [&](const St& S) __attribute__((always_inline)) -> bool {
        if (!S.valid) return false;
        cplx acc = cmake(0.0, 0.0);
        if (S.i >= 0) {
            const int st = S.wide ? 64 : 16;
            if (S.p < S.e1) cfma(acc, S.v, x[S.col]);
            for (int p = S.p + st; p < S.e1; p += 4 * st) {      // long rows: four trips' loads together
                cplx v[4]; int col[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { const int pp = p + st * u < S.e1 ? p + st * u : p; v[u] = bx[pp]; col[u] = bi[pp]; }
#pragma unroll
                for (int u = 0; u < 4; ++u) if (p + st * u < S.e1) cfma(acc, v[u], x[col[u]]);
            }
        }
        acc = group_reduce_sum<16>(acc);
        if (S.wide) {                                           // the four 16-lane groups worked on ONE row: add their sums
            acc.x += __shfl_xor(acc.x, 16, 64); acc.y += __shfl_xor(acc.y, 16, 64);
            acc.x += __shfl_xor(acc.x, 32, 64); acc.y += __shfl_xor(acc.y, 32, 64);
        }
        if (S.i >= 0 && (S.wide ? threadIdx.x == 0 : sub == 0)) x[S.i] = UPPER ? ml_cdiv(csub(x[S.i], acc), diag[s + S.i]) : csub(x[S.i], acc);
        if (S.last) __syncthreads();
        return true;
    };
    St A, B, C, D;
    issue(A); issue(B); issue(C); issue(D);
    while (true) {
        if (!consume(A)) break; issue(A);
        if (!consume(B)) break; issue(B);
        if (!consume(C)) break; issue(C);
        if (!consume(D)) break; issue(D);
    }
    __syncthreads();
    if (UPPER) { for (int t = threadIdx.x; t <= j; t += ML_INV_NT) ix[ip[s + t] + (j - t)] = x[t]; }
    else       { for (int t = j + threadIdx.x; t < bsz; t += ML_INV_NT) ix[ip[s + t] + j] = x[t]; }
}

// device-side numeric path: values of a factor (input entry order) -> the schedule's value array
__global__ void k_ml_gather(int64_t nnz, const int32_t* __restrict__ map, const cplx* src, cplx* vals,
                            int64_t o1, int64_t o2, int64_t o3, cplx* const* batch_vals = nullptr, const cplx* const* batch_src = nullptr) {
    if (batch_vals) { vals = batch_vals[blockIdx.y]; src = batch_src[blockIdx.y]; }
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < nnz; e += (int64_t)gridDim.x * blockDim.x) {
        const int32_t v = map[e];
        const int kind = v & 3;
        if (kind == 1) vals[o1 + (v >> 2)] = src[e];
        else if (kind == 2) vals[o2 + (v >> 2)] = src[e];
        else if (kind == 3) vals[o3 + (v >> 2)] = src[e];
    }
}

// ---- solve kernels ------------------------------------------------------------------------------------------------
struct MLArgs {
    const MLChunk* chunks; int nchunks;
    const int32_t* segs; int nsegs; int nside;                     // k_ml_level_blk: segment-starting chunks, side-job workgroups
    const int32_t* cp; const int32_t* ci; const cplx* cx;          // coupling CSR
    const cplx* ix;                                                // packed inverse rows (offsets in the chunk records)
    int has_coupling;
    const cplx* src; int64_t ldsrc; const int32_t* gat; const double* rs;   // right-hand side of the level: [rs]*src[gat[c]]
    int ident_row0;                                                // >= 0: the right-hand sides are unit vectors, rhs j = e_(ident_row0 + j)
    int col_lo;                                                    // coupling entries with column < col_lo are skipped (apex build)
    const cplx* xin; int64_t ldxin;                                // solved rows of the other levels
    cplx* xout; int64_t ldxout;
    const cplx* tmp; int64_t ldtmp;                                // MODE 1: r precomputed by k_ml_coupling
    cplx* outX; int64_t ldX; const int32_t* pout; double scale; const cplx* add; int64_t ldadd;   // final output (optional)
    // side job of extra workgroups, rows [side_lo, side_hi): LOWER copies the right-hand side of the other levels into
    // side_dst (new order); UPPER scatters the finished rows of the other levels (side_src) to outX
    int64_t side_lo, side_hi; cplx* side_dst; int64_t ldside; const cplx* side_src; int64_t ldsidesrc;
    int nrhs;
};

// G2 lanes per row, CH = 256 / G2 rows per chunk: all rows of a chunk are processed concurrently
template <bool UPPER, int RB, int MODE, int G2>
__device__ __forceinline__ void ml_level_body(const MLArgs& A, const int bx, const int by) {
    const int rhs0 = by * RB;
    const int nb = min(RB, A.nrhs - rhs0);
    if (bx >= A.nchunks) {                            // ---- side job
        const int64_t q = A.side_lo + ((int64_t)bx - A.nchunks) * 256 + threadIdx.x;
        if (q < A.side_hi) {
            if (!UPPER) {
                const int64_t g = A.gat ? A.gat[q] : q;
                const double sc = A.rs ? A.rs[g] : 1.0;
                for (int r = 0; r < nb; ++r) {
                    const cplx v = A.src[(int64_t)(rhs0 + r) * A.ldsrc + g];
                    A.side_dst[(int64_t)(rhs0 + r) * A.ldside + q] = cmake(sc * v.x, sc * v.y);
                }
            } else {
                const int64_t g = A.pout ? A.pout[q] : q;
                for (int r = 0; r < nb; ++r) {
                    cplx v = A.side_src[(int64_t)(rhs0 + r) * A.ldsidesrc + q];
                    if (A.add) { const cplx ad = A.add[(int64_t)(rhs0 + r) * A.ldadd + g]; v.x += ad.x; v.y += ad.y; }
                    A.outX[(int64_t)(rhs0 + r) * A.ldX + g] = cmake(A.scale * v.x, A.scale * v.y);
                }
            }
        }
        return;
    }
    __shared__ cplx rbuf[MODE == 0 ? RB * ML_BMAX : 1];
    const MLChunk ch = A.chunks[bx];
    const int base = UPPER ? ch.a : ch.s;                           // first row whose r is needed
    if (MODE == 0) {
        // r_c = rhs_c - C[c,:] xin for the rows the chunk's dot products read: one thread per row (at most ML_BMAX rows;
        // levels whose coupling rows are long run the product as its own launch instead, MODE 1)
        const int cnt = UPPER ? ch.e - ch.a : ch.b - ch.s;
        const int t = threadIdx.x;
        if (t < cnt) {
            const int c = base + t;
            cplx acc[RB];
#pragma unroll
            for (int r = 0; r < RB; ++r) acc[r] = cmake(0.0, 0.0);
            if (A.has_coupling) {
                const int e1 = A.cp[c + 1];
                if (RB == 1) {
                    for (int p = A.cp[c]; p < e1; p += 4) {            // four entries per trip: loads first, then the gathers
                        cplx v[4], xv[4]; int64_t col[4]; bool on[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int pp = p + u < e1 ? p + u : p;
                            v[u] = A.cx[pp]; col[u] = A.ci[pp];
                            on[u] = p + u < e1 && col[u] >= A.col_lo;
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u) xv[u] = A.xin[(int64_t)rhs0 * A.ldxin + (on[u] ? col[u] : (int64_t)A.col_lo)];
#pragma unroll
                        for (int u = 0; u < 4; ++u) if (on[u]) cfma(acc[0], v[u], xv[u]);
                    }
                } else {
                    for (int p = A.cp[c]; p < e1; ++p) {
                        const cplx v = A.cx[p];
                        const int64_t col = A.ci[p];
                        if (col < A.col_lo) continue;
#pragma unroll
                        for (int r = 0; r < RB; ++r)
                            if (r < nb) cfma(acc[r], v, A.xin[(int64_t)(rhs0 + r) * A.ldxin + col]);
                    }
                }
            }
            const int64_t g = A.gat ? A.gat[c] : c;
            const double sc = A.rs ? A.rs[g] : 1.0;
#pragma unroll
            for (int r = 0; r < RB; ++r)
                if (r < nb) {
                    const cplx v = A.ident_row0 >= 0 ? cmake(c - A.ident_row0 == rhs0 + r ? 1.0 : 0.0, 0.0)
                                                     : A.src[(int64_t)(rhs0 + r) * A.ldsrc + g];
                    rbuf[r * ML_BMAX + t] = cmake(sc * v.x - acc[r].x, sc * v.y - acc[r].y);
                }
        }
        __syncthreads();
    }
    const int sub = threadIdx.x % G2;
    const int rho = ch.a + threadIdx.x / G2;
    const bool live = rho < ch.b;
    const int d = rho - ch.a;
    // packed rows: LOWER row r holds columns [s, r], UPPER row r holds [r, e)
    const int len = !live ? 0 : (UPPER ? ch.e - rho : rho - ch.s + 1);
    const int c0 = UPPER ? rho : ch.s;
    const int64_t off = UPPER ? ch.ipa + (int64_t)d * (ch.e - ch.a) - (int64_t)d * (d - 1) / 2
                              : ch.ipa + (int64_t)d * (ch.a - ch.s + 1) + (int64_t)d * (d - 1) / 2;
    const cplx* row = A.ix + off;
    cplx acc[RB];
#pragma unroll
    for (int r = 0; r < RB; ++r) acc[r] = cmake(0.0, 0.0);
    if (RB == 1) {
        // (fetching the first trip ahead of the right-hand-side gather and its barrier was measured: 8.1 -> 10.2 us, removed)
        // single right-hand side: the loads of four trips are issued together (a row of a packed inverse has up to 256 entries,
        // 4 per lane at G2 = 64 -- one round trip instead of four)
        for (int t = sub; t < len; t += 4 * G2) {
            cplx m[4], rv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int tt = t + u * G2 < len ? t + u * G2 : t;
                m[u] = row[tt];
                rv[u] = MODE == 0 ? rbuf[(c0 - base) + tt] : A.tmp[(int64_t)rhs0 * A.ldtmp + c0 + tt];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) if (t + u * G2 < len) cfma(acc[0], m[u], rv[u]);
        }
    } else {
        for (int t = sub; t < len; t += G2) {
            const cplx m = row[t];
#pragma unroll
            for (int r = 0; r < RB; ++r)
                if (r < nb) {
                    const cplx rv = MODE == 0 ? rbuf[r * ML_BMAX + (c0 - base) + t]
                                              : A.tmp[(int64_t)(rhs0 + r) * A.ldtmp + c0 + t];
                    cfma(acc[r], m, rv);
                }
        }
    }
#pragma unroll
    for (int r = 0; r < RB; ++r) acc[r] = group_reduce_sum<G2>(acc[r]);
    if (live && sub == 0) {
        for (int r = 0; r < nb; ++r) {
            A.xout[(int64_t)(rhs0 + r) * A.ldxout + rho] = acc[r];
            if (UPPER && A.outX) {
                const int64_t g = A.pout ? A.pout[rho] : rho;
                cplx v = acc[r];
                if (A.add) { const cplx ad = A.add[(int64_t)(rhs0 + r) * A.ldadd + g]; v.x += ad.x; v.y += ad.y; }
                A.outX[(int64_t)(rhs0 + r) * A.ldX + g] = cmake(A.scale * v.x, A.scale * v.y);
            }
        }
    }
}

template <bool UPPER, int RB, int MODE, int G2>
__global__ __launch_bounds__(256) void k_ml_level(const MLArgs A) { ml_level_body<UPPER, RB, MODE, G2>(A, (int)blockIdx.x, (int)blockIdx.y); }

// Blocks of right-hand sides (contour_beyn: 32 per node): one workgroup per 64-row SEGMENT of a diagonal block and group of RB
// right-hand sides.  In the chunk form above every 4-row chunk stages the right-hand-side rows its dot products read -- up to the
// whole block, RB columns, gathered through the input permutation: 64 chunks of a 256-row block load the same 32 KB
// (k_ml_level<false, 8, 0> on the gun factor: 301 us for level 0, 10 000 workgroups, 25 x the bytes and the flops of the product).
// Here a segment stages the columns its rows read once ([s, end of segment) for LOWER, [start of segment, e) for UPPER) and walks
// its packed inverse rows 16 at a time, 16 lanes per row; same sums per row up to the order of the lane partials.  (A whole block
// per workgroup was one long chain: 131 us on gun's level 0, 30-90 us on a level of six blocks.)  The segments of a level come
// from a list built with the chunks (MLFacSym::d_segs); side-job workgroups follow them in the grid.
template <bool UPPER, int RB, int MODE>
__global__ __launch_bounds__(256) void k_ml_level_blk(const MLArgs A) {
    // 1-D grid over (segment or side-job workgroup) x (group of RB right-hand sides), XCD-aware: workgroups are dealt to the 8
    // XCDs round robin by their linear id, so the groups of right-hand sides of ONE segment take ids 8 apart: they land on one
    // XCD, one after the other -- the segment's packed inverse rows come from HBM once and from that XCD's L2 for the other groups
    // (a 2-D grid put every group on another XCD -- 8 x the reads -- or, with gridDim.x a multiple of 8, all of them on one)
    const int ngy = (A.nrhs + RB - 1) / RB;
    const int xcd = (int)(blockIdx.x & 7), slot = (int)(blockIdx.x >> 3);
    const int by = slot % ngy, item = (slot / ngy) * 8 + xcd;
    if (item >= A.nsegs + A.nside) return;
    if (item >= A.nsegs) { ml_level_body<UPPER, RB, MODE, 64>(A, A.nchunks + item - A.nsegs, by); return; }      // side job
    const int bx = A.segs[item];
    const MLChunk ch = A.chunks[bx];
    // (a block's rows in segments of ML_BLK_SEG; a segment stages the columns its rows read -- [s, end of segment) for LOWER,
    // [start of segment, e) for UPPER)
    const int d0 = ch.a - ch.s;
    const int rhs0 = by * RB;
    const int nb = min(RB, A.nrhs - rhs0);
    const int s = ch.s, e = ch.e, nrow = e - s;
    const int d1 = min(d0 + ML_BLK_SEG, nrow);
    const int64_t ipa_s = ch.ipa - (UPPER ? (int64_t)d0 * nrow - (int64_t)d0 * (d0 - 1) / 2 : (int64_t)d0 * (d0 + 1) / 2);
    __shared__ cplx rbuf[RB * ML_BMAX];
    for (int t = (UPPER ? d0 : 0) + threadIdx.x; t < (UPPER ? nrow : d1); t += 256) {
        const int c = s + t;
        if (MODE == 0) {
            cplx acc[RB];
#pragma unroll
            for (int r = 0; r < RB; ++r) acc[r] = cmake(0.0, 0.0);
            if (A.has_coupling) {
                const int e1 = A.cp[c + 1];
                for (int p = A.cp[c]; p < e1; ++p) {
                    const cplx v = A.cx[p];
                    const int64_t col = A.ci[p];
                    if (col < A.col_lo) continue;
#pragma unroll
                    for (int r = 0; r < RB; ++r)
                        if (r < nb) cfma(acc[r], v, A.xin[(int64_t)(rhs0 + r) * A.ldxin + col]);
                }
            }
            const int64_t g = A.gat ? A.gat[c] : c;
            const double sc = A.rs ? A.rs[g] : 1.0;
#pragma unroll
            for (int r = 0; r < RB; ++r)
                if (r < nb) {
                    const cplx v = A.src[(int64_t)(rhs0 + r) * A.ldsrc + g];
                    rbuf[r * ML_BMAX + t] = cmake(sc * v.x - acc[r].x, sc * v.y - acc[r].y);
                }
        } else {
#pragma unroll
            for (int r = 0; r < RB; ++r)
                if (r < nb) rbuf[r * ML_BMAX + t] = A.tmp[(int64_t)(rhs0 + r) * A.ldtmp + c];
        }
    }
    __syncthreads();
    const int sub = threadIdx.x & 15, rloc = threadIdx.x >> 4;
    for (int row0 = d0; row0 < d1; row0 += 16) {
        const int d = row0 + rloc;
        const bool live = d < d1;
        const int rho = s + d;
        // packed rows: LOWER row d holds columns [s, s + d], UPPER row d holds [s + d, e)
        const int len = !live ? 0 : (UPPER ? nrow - d : d + 1);
        const int c0 = UPPER ? d : 0;
        const int64_t off = ipa_s + (UPPER ? (int64_t)d * nrow - (int64_t)d * (d - 1) / 2 : (int64_t)d * (d + 1) / 2);
        const cplx* __restrict__ row = A.ix + off;
        cplx acc[RB];
#pragma unroll
        for (int r = 0; r < RB; ++r) acc[r] = cmake(0.0, 0.0);
        // a row has at most 256 entries, 16 per lane: all of a lane's loads go out before the first product (one round trip
        // per 16 rows; a load per trip left the single workgroup of a block waiting 16 times as often)
        for (int t0 = 0; t0 < len; t0 += 128) {
            cplx m[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int t = t0 + sub + 16 * u; m[u] = t < len ? row[t] : cmake(0.0, 0.0); }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int t = t0 + sub + 16 * u;
                if (t < len) {
#pragma unroll
                    for (int r = 0; r < RB; ++r) cfma(acc[r], m[u], rbuf[r * ML_BMAX + c0 + t]);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < RB; ++r) acc[r] = group_reduce_sum<16>(acc[r]);
        if (live && sub == 0) {
#pragma unroll
            for (int r = 0; r < RB; ++r) {        // (static indices: a run-time bound put acc[] into scratch memory)
                if (r >= nb) break;
                A.xout[(int64_t)(rhs0 + r) * A.ldxout + rho] = acc[r];
                if (UPPER && A.outX) {
                    const int64_t g = A.pout ? A.pout[rho] : rho;
                    cplx v = acc[r];
                    if (A.add) { const cplx ad = A.add[(int64_t)(rhs0 + r) * A.ldadd + g]; v.x += ad.x; v.y += ad.y; }
                    A.outX[(int64_t)(rhs0 + r) * A.ldX + g] = cmake(A.scale * v.x, A.scale * v.y);
                }
            }
        }
    }
}

// tmp[q] = src[q] - sum_{col_lo <= col < col_hi} C[q,col] xin[col]   for the rows [r0, r1) of one level; G lanes per row
// (G = 256: one workgroup per row); ident_row0 >= 0: src is the identity block (rhs j = e_(ident_row0 + j))
struct MLCplArgs {
    int r0, r1; const int32_t* cp; const int32_t* ci; const cplx* cx; const cplx* src; int64_t ldsrc; const cplx* xin; int64_t ldxin;
    cplx* tmp; int64_t ldtmp; int nrhs, col_lo, col_hi, ident_row0;
};
template <int G, int RB>
__device__ __forceinline__ void ml_coupling_body(const MLCplArgs& C, const int bx, const int by) {
    const int r0 = C.r0, r1 = C.r1, nrhs = C.nrhs, col_lo = C.col_lo, col_hi = C.col_hi, ident_row0 = C.ident_row0;
    const int32_t* __restrict__ cp = C.cp; const int32_t* __restrict__ ci = C.ci; const cplx* __restrict__ cx = C.cx;
    const cplx* __restrict__ src = C.src; const cplx* __restrict__ xin = C.xin; cplx* __restrict__ tmp = C.tmp;
    const int64_t ldsrc = C.ldsrc, ldxin = C.ldxin, ldtmp = C.ldtmp;
    constexpr int GG = G == 256 ? 64 : G;
    constexpr int RPB = G == 256 ? 1 : 256 / G;
    const int rhs0 = by * RB;
    const int nb = min(RB, nrhs - rhs0);
    const int sub = G == 256 ? threadIdx.x : (threadIdx.x % GG);
    const int q = r0 + bx * RPB + (G == 256 ? 0 : threadIdx.x / GG);
    cplx acc[RB];
#pragma unroll
    for (int r = 0; r < RB; ++r) acc[r] = cmake(0.0, 0.0);
    if (q < r1) {
        const int e0 = cp[q], e1 = cp[q + 1];
        if (RB == 1) {
            // single right-hand side: four entries per lane and trip -- index and value loads together, then the four gathers of
            // xin, then the arithmetic: three dependent round trips per trip instead of two per entry (the compiler waits for the
            // gather of an entry before it issues the loads of the next)
            for (int p = e0 + sub; p < e1; p += 4 * G) {
                int col[4]; cplx v[4], xv[4]; bool on[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int pp = p + u * G < e1 ? p + u * G : p;
                    col[u] = ci[pp]; v[u] = cx[pp];
                    on[u] = p + u * G < e1 && col[u] >= col_lo && col[u] < col_hi;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) xv[u] = xin[(int64_t)rhs0 * ldxin + (on[u] ? col[u] : col_lo)];
#pragma unroll
                for (int u = 0; u < 4; ++u) if (on[u]) cfma(acc[0], v[u], xv[u]);
            }
        } else {
            // (several right-hand sides, i.e. the apex build: grouping two entries' 2 + 16 loads per trip was measured -- coupling
            // 225 -> 256 us, level kernels 368 -> 402 / 504 -> 469 us: no net gain, not in)
            for (int p = e0 + sub; p < e1; p += G) {
                const int col = ci[p];
                if (col < col_lo || col >= col_hi) continue;
                const cplx v = cx[p];
#pragma unroll
                for (int r = 0; r < RB; ++r)
                    if (r < nb) cfma(acc[r], v, xin[(int64_t)(rhs0 + r) * ldxin + col]);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < RB; ++r) acc[r] = group_reduce_sum<GG>(acc[r]);
    if (G == 256) {
        __shared__ cplx part[4][RB];
        const int wv = threadIdx.x >> 6;
        if ((threadIdx.x & 63) == 0)
            for (int r = 0; r < RB; ++r) part[wv][r] = acc[r];
        __syncthreads();
        if (threadIdx.x < nb && q < r1) {
            const int r = threadIdx.x;
            cplx a = part[0][r];
            for (int w = 1; w < 4; ++w) { a.x += part[w][r].x; a.y += part[w][r].y; }
            const cplx v = ident_row0 >= 0 ? cmake(q - ident_row0 == rhs0 + r ? 1.0 : 0.0, 0.0) : src[(int64_t)(rhs0 + r) * ldsrc + q];
            tmp[(int64_t)(rhs0 + r) * ldtmp + q] = csub(v, a);
        }
    } else if (q < r1 && sub == 0) {
        for (int r = 0; r < nb; ++r) {
            const cplx v = ident_row0 >= 0 ? cmake(q - ident_row0 == rhs0 + r ? 1.0 : 0.0, 0.0) : src[(int64_t)(rhs0 + r) * ldsrc + q];
            tmp[(int64_t)(rhs0 + r) * ldtmp + q] = csub(v, acc[r]);
        }
    }
}

template <int G, int RB>
__global__ __launch_bounds__(256) void k_ml_coupling(const MLCplArgs C) { ml_coupling_body<G, RB>(C, (int)blockIdx.x, (int)blockIdx.y); }

// ---- last launch of a single-vector solve: U level 0 with its coupling product inside ------------------------------------------
//   x_B = inv(U_BB) (y_B - U[B, T] x_T)   for every level-0 block B, then the caller's X through the output permutation.
// The two-launch form (k_ml_coupling over the level's rows, then k_ml_level<MODE 1> in 4-row chunks) pays a kernel boundary
// and a round trip of the coupled right-hand side; forming the product inside the chunked level kernel would repeat it for
// every chunk of a block.  Here TWO workgroups of 1024 threads own a block: each forms r_B = y_B - U[B, T] x_T once in LDS
// (4 lanes per row; twice per block in total), then applies its half of the packed inverse rows.  The rows of a triangular
// inverse have lengths 1 .. b: rows l and b-1-l are PAIRED (b + 1 entries together) and every pair is walked by 16 lanes,
// so all lanes carry the same number of loads; the workgroup with index parity h takes the pairs p = h, h + 2, ...
struct MLU0Args {
    const int32_t* blk_se; int blk0, nblk;
    const int32_t* cp; const int32_t* ci; const cplx* cx;        // U coupling CSR (rows of the level, columns in the apex / later levels)
    const cplx* ix; const int64_t* ip;                           // packed inverse rows of U_BB: row q holds columns [q, e)
    const cplx* y; const cplx* x;                                // y (all rows), x of the later levels (new order)
    cplx* outX; const int32_t* pout; double scale; const cplx* add;
    int64_t side_lo, side_hi;                                    // rows of the later levels: scattered to outX by extra workgroups
};
__global__ __launch_bounds__(1024) void k_ml_u0_fused(const MLU0Args A) {
    __shared__ cplx r[ML_BMAX];
    const int tid = threadIdx.x;
    if ((int)blockIdx.x >= 2 * A.nblk) {                          // ---- side job: x of the later levels -> caller's X
        const int64_t q = A.side_lo + ((int64_t)blockIdx.x - 2 * A.nblk) * 1024 + tid;
        if (q < A.side_hi) {
            const int64_t g = A.pout ? A.pout[q] : q;
            cplx v = A.x[q];
            if (A.add) { const cplx ad = A.add[g]; v.x += ad.x; v.y += ad.y; }
            A.outX[g] = cmake(A.scale * v.x, A.scale * v.y);
        }
        return;
    }
    const int k = A.blk0 + ((int)blockIdx.x >> 1), half = (int)blockIdx.x & 1;
    const int s = A.blk_se[2 * k], e = A.blk_se[2 * k + 1], b = e - s;
    {   // r_l = y[s + l] - sum_p U[s + l, col_p] x[col_p]: 4 lanes per row
        const int l = tid >> 2, sub = tid & 3;
        cplx acc = cmake(0.0, 0.0);
        if (l < b) {
            const int q = s + l;
            const int e1 = A.cp[q + 1];
#pragma unroll 4
            for (int p = A.cp[q] + sub; p < e1; p += 4) cfma(acc, A.cx[p], A.x[A.ci[p]]);
        }
        acc = group_reduce_sum<4>(acc);
        if (l < b && sub == 0) r[l] = csub(A.y[s + l], acc);
    }
    __syncthreads();
    // pairs of rows (l, b-1-l), l < ceil(b/2): 16 lanes per pair, 64 pairs per pass
    const int npair = (b + 1) >> 1;
    const int sub = tid & 15;
    for (int p = half + 2 * (tid >> 4); p < npair; p += 128) {
        const int la = p, lb = b - 1 - p;                         // la <= lb
        const int lenA = b - la, lenB = (lb > la) ? b - lb : 0;   // row l holds columns [l, b)
        const cplx* rowA = A.ix + A.ip[s + la];
        const cplx* rowB = A.ix + A.ip[s + lb];
        cplx accA = cmake(0.0, 0.0), accB = cmake(0.0, 0.0);
        const int tot = lenA + lenB;
#pragma unroll 4
        for (int pos = sub; pos < tot; pos += 16) {
            if (pos < lenA) cfma(accA, rowA[pos], r[la + pos]);
            else { const int t = pos - lenA; cfma(accB, rowB[t], r[lb + t]); }
        }
        accA = group_reduce_sum<16>(accA);
        accB = group_reduce_sum<16>(accB);
        if (sub == 0) {
            const int64_t ga = A.pout ? A.pout[s + la] : s + la;
            cplx v = accA;
            if (A.add) { const cplx ad = A.add[ga]; v.x += ad.x; v.y += ad.y; }
            A.outX[ga] = cmake(A.scale * v.x, A.scale * v.y);
            if (lb > la) {
                const int64_t gb = A.pout ? A.pout[s + lb] : s + lb;
                cplx w = accB;
                if (A.add) { const cplx ad = A.add[gb]; w.x += ad.x; w.y += ad.y; }
                A.outX[gb] = cmake(A.scale * w.x, A.scale * w.y);
            }
        }
    }
}

// ---- apex: the last levels as ONE dense inverse S^{-1} = inv(U_TT) inv(L_TT) (T x T, row-major) ---------------------------
// out (row-major T x T) = transpose of the column-major block `in` (ld = T)
__global__ __launch_bounds__(256) void k_apex_transpose(int T, const cplx* __restrict__ in, cplx* __restrict__ out) {
    __shared__ cplx tile[16][17];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int r = blockIdx.x * 16 + tx, c = blockIdx.y * 16 + ty;        // in[(c)*T + r]: r fastest
    if (r < T && c < T) tile[ty][tx] = in[(int64_t)c * T + r];
    __syncthreads();
    const int c2 = blockIdx.y * 16 + tx, r2 = blockIdx.x * 16 + ty;
    if (r2 < T && c2 < T) out[(int64_t)r2 * T + c2] = tile[tx][ty];
}
// x[R0 + r] = sum_c Sinv[r, c] t[R0 + c]: wave per row, RB right-hand sides share one pass over the row
struct MLApexArgs { int T, R0; const cplx* Sinv; const cplx* t; int64_t ldt; cplx* x; int64_t ldx; int nrhs; };
template <int RB>
__device__ __forceinline__ void apex_gemv_body(const MLApexArgs& P, const int bx, const int by) {
    const int T = P.T, R0 = P.R0, nrhs = P.nrhs;
    const cplx* __restrict__ Sinv = P.Sinv; const cplx* __restrict__ t = P.t; cplx* __restrict__ x = P.x;
    const int64_t ldt = P.ldt, ldx = P.ldx;
    const int rhs0 = by * RB;
    const int nb = min(RB, nrhs - rhs0);
    const int lane = threadIdx.x & 63;
    const int r = bx * 4 + (threadIdx.x >> 6);
    if (r >= T) return;
    const cplx* row = Sinv + (int64_t)r * T;
    cplx acc[RB];
#pragma unroll
    for (int q = 0; q < RB; ++q) acc[q] = cmake(0.0, 0.0);
#pragma unroll 4
    for (int c = lane; c < T; c += 64) {
        const cplx m = row[c];
#pragma unroll
        for (int q = 0; q < RB; ++q)
            if (q < nb) cfma(acc[q], m, t[(int64_t)(rhs0 + q) * ldt + R0 + c]);
    }
#pragma unroll
    for (int q = 0; q < RB; ++q) {
        const cplx a = group_reduce_sum<64>(acc[q]);
        if (lane == 0 && q < nb) x[(int64_t)(rhs0 + q) * ldx + R0 + r] = a;
    }
}

template <int RB>
__global__ __launch_bounds__(256) void k_apex_gemv(const MLApexArgs P) { apex_gemv_body<RB>(P, (int)blockIdx.x, (int)blockIdx.y); }

// single right-hand side: the whole vector t (T <= 2048 entries) is staged in LDS once per workgroup, every wave owns two rows
// and keeps eight 16-byte loads of S^{-1} in flight per lane (the wave-per-row form above had four: 28 MB in 9.4 us)
#define ML_APEX_TMAX 2048
__global__ __launch_bounds__(256) void k_apex_gemv1(const MLApexArgs P) {
    __shared__ cplx ts[ML_APEX_TMAX];
    const int T = P.T, R0 = P.R0;
    for (int c = threadIdx.x; c < T; c += 256) ts[c] = P.t[R0 + c];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int r0 = ((int)blockIdx.x * 4 + (threadIdx.x >> 6)) * 2;
    if (r0 >= T) return;
    const bool two = r0 + 1 < T;
    const cplx* __restrict__ rowA = P.Sinv + (int64_t)r0 * T;
    const cplx* __restrict__ rowB = P.Sinv + (int64_t)(two ? r0 + 1 : r0) * T;
    cplx a0 = cmake(0.0, 0.0), a1 = cmake(0.0, 0.0), b0 = cmake(0.0, 0.0), b1 = cmake(0.0, 0.0);
    int c = lane;
    for (; c + 192 < T; c += 256) {
        const cplx m0 = rowA[c], m1 = rowA[c + 64], m2 = rowA[c + 128], m3 = rowA[c + 192];
        const cplx n0 = rowB[c], n1 = rowB[c + 64], n2 = rowB[c + 128], n3 = rowB[c + 192];
        const cplx t0 = ts[c], t1 = ts[c + 64], t2 = ts[c + 128], t3 = ts[c + 192];
        cfma(a0, m0, t0); cfma(a1, m1, t1); cfma(a0, m2, t2); cfma(a1, m3, t3);
        cfma(b0, n0, t0); cfma(b1, n1, t1); cfma(b0, n2, t2); cfma(b1, n3, t3);
    }
    for (; c < T; c += 64) { const cplx tt = ts[c]; cfma(a0, rowA[c], tt); cfma(b0, rowB[c], tt); }
    const cplx sa = group_reduce_sum<64>(cadd(a0, a1)), sb = group_reduce_sum<64>(cadd(b0, b1));
    if (lane == 0) { P.x[R0 + r0] = sa; if (two) P.x[R0 + r0 + 1] = sb; }
}

// =====================================================================================================================
// host side
// =====================================================================================================================
namespace {

struct PinnedPool {           // pinned staging buffers for the value upload (hipHostMalloc costs ~1 ms per call)
    struct Buf { void* p; size_t cap; hipEvent_t ev; bool busy; };
    std::mutex mu;
    std::vector<Buf> bufs;
    int acquire(size_t bytes, int* idx) {
        std::lock_guard<std::mutex> lk(mu);
        for (size_t i = 0; i < bufs.size(); ++i) {
            Buf& b = bufs[i];
            if (!b.busy && b.cap >= bytes && b.cap <= 2 * bytes + (1 << 20)) {
                if (b.ev && hipEventQuery(b.ev) != hipSuccess) continue;
                b.busy = true; *idx = (int)i; return NEP_OK;
            }
        }
        Buf nb; nb.p = nullptr; nb.cap = bytes + bytes / 8 + 4096; nb.ev = nullptr; nb.busy = true;
        HIPCHK(hipHostMalloc(&nb.p, nb.cap, hipHostMallocDefault));
        HIPCHK(hipEventCreateWithFlags(&nb.ev, hipEventDisableTiming));
        if (bufs.size() >= 24) {                  // bounded: drop an idle one
            for (size_t i = 0; i < bufs.size(); ++i)
                if (!bufs[i].busy && (!bufs[i].ev || hipEventQuery(bufs[i].ev) == hipSuccess)) {
                    (void)hipHostFree(bufs[i].p); (void)hipEventDestroy(bufs[i].ev);
                    bufs[i] = nb; *idx = (int)i; return NEP_OK;
                }
        }
        bufs.push_back(nb); *idx = (int)bufs.size() - 1;
        return NEP_OK;
    }
    void* ptr(int idx) { std::lock_guard<std::mutex> lk(mu); return bufs[idx].p; }
    void release(int idx, hipStream_t st) {       // reusable once the copies enqueued on st have completed
        std::lock_guard<std::mutex> lk(mu);
        (void)hipEventRecord(bufs[idx].ev, st);
        bufs[idx].busy = false;
    }
};
PinnedPool g_pinned;

struct BuildStreams {          // a fixed set of non-blocking streams for the numeric builds (never destroyed: process-wide)
    std::mutex mu; hipStream_t st[4] = {nullptr, nullptr, nullptr, nullptr}; unsigned next = 0;
    hipStream_t get() {
        std::lock_guard<std::mutex> lk(mu);
        const unsigned i = next++ % 4;
        if (!st[i] && hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking) != hipSuccess) st[i] = nullptr;
        return st[i];
    }
};
BuildStreams g_bstreams;

std::mutex g_cache_mu;
std::list<MLSym*> g_cache;       // most recently used first
const size_t ML_CACHE_MAX = 8;

void hash_words(uint64_t& h0, uint64_t& h1, const void* p, size_t bytes) {
    const uint8_t* b = (const uint8_t*)p;
    size_t i = 0;
    for (; i + 8 <= bytes; i += 8) {
        uint64_t w; memcpy(&w, b + i, 8);
        h0 = (h0 ^ w) * 0x9E3779B97F4A7C15ull; h0 ^= h0 >> 29;
        h1 = (h1 + w) * 0xC2B2AE3D27D4EB4Full; h1 ^= h1 >> 31;
    }
    uint64_t w = 0;
    if (i < bytes) memcpy(&w, b + i, bytes - i);
    h0 = (h0 ^ w ^ bytes) * 0x9E3779B97F4A7C15ull;
    h1 = (h1 + w + bytes) * 0xC2B2AE3D27D4EB4Full;
}

thread_local bool g_ml_dry = false;     // ml_analyze: host analysis only, nothing is uploaded

template <class T>
int up(T** d, const std::vector<T>& h, size_t min_count = 1) {
    if (g_ml_dry) { *d = nullptr; return NEP_OK; }
    const size_t cnt = std::max(h.size(), min_count);
    int rc = nep_pool_alloc((void**)d, cnt * sizeof(T));
    if (rc) return rc;
    if (!h.empty()) HIPCHK(hipMemcpy(*d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    return NEP_OK;
}

void free_fac(MLFacSym& f) {
    nep_pool_free(f.d_cp); nep_pool_free(f.d_ci); nep_pool_free(f.d_ip); nep_pool_free(f.d_bp); nep_pool_free(f.d_bi);
    nep_pool_free(f.d_slotrow); nep_pool_free(f.d_lvp); nep_pool_free(f.d_lvo); nep_pool_free(f.d_rowlev); nep_pool_free(f.d_chunks); nep_pool_free(f.d_segs);
    if (f.d_map) nep_pool_free(f.d_map);
}
void free_sym(MLSym* s) {
    if (!s) return;
    free_fac(s->L); free_fac(s->U);
    nep_pool_free(s->d_blk_se); nep_pool_free(s->d_rowblk);
    nep_pool_free(s->d_pin); nep_pool_free(s->d_pout);
    if (s->d_apex_kr) nep_pool_free(s->d_apex_kr);
    if (s->apex_work_ev) { (void)hipEventSynchronize(s->apex_work_ev); (void)hipEventDestroy(s->apex_work_ev); }
    if (s->d_apex_work) nep_pool_free(s->d_apex_work);
    delete s;
}

// CSR pattern (rowptr, colidx) + source index of every entry in the caller's arrays, from a CSC pattern
void transpose_pattern(int64_t n, const int32_t* cp, const int32_t* ri, std::vector<int32_t>& rp, std::vector<int32_t>& ci,
                       std::vector<int32_t>* srcidx) {
    const int64_t nnz = cp[n];
    rp.assign(n + 1, 0);
    for (int64_t e = 0; e < nnz; ++e) rp[ri[e] + 1]++;
    for (int64_t i = 0; i < n; ++i) rp[i + 1] += rp[i];
    ci.resize(nnz);
    if (srcidx) srcidx->resize(nnz);
    std::vector<int32_t> pos(rp.begin(), rp.end() - 1);
    for (int64_t c = 0; c < n; ++c)
        for (int32_t e = cp[c]; e < cp[c + 1]; ++e) {
            const int32_t q = pos[ri[e]]++;
            ci[q] = (int32_t)c;
            if (srcidx) (*srcidx)[q] = e;
        }
}

int bmax_env() { const char* e = getenv("NEP_ML_BMAX"); const int v = e ? atoi(e) : -1; return (v >= 8 && v <= ML_BMAX) ? v : -1; }
int bmax_of_level(int lev, int64_t n, int forced) {
    if (forced > 0) return forced;
    if (n > 200000 && lev == 0) return 64;          // most rows sit in level 0: 16*b/2 bytes of inverse per row
    if (n > 200000 && lev == 1) return 128;
    return ML_BMAX;
}

// one triangular factor in the new order.  rp/ci: CSR pattern in the ORIGINAL factor numbering, src: index of each CSR
// entry in the caller's value array (nullptr = identity)
int build_factor(const MLSym& S, bool upper, const int32_t* rp, const int32_t* ci, const int32_t* src,
                 const std::vector<int32_t>& newpos, const std::vector<int32_t>& oldof, const std::vector<int32_t>& rowblk,
                 const std::vector<int32_t>& blk_se, const std::vector<int32_t>& rowlev, MLFacSym& F) {
    const int64_t n = S.n;
    const int64_t nnz = rp[n];
    F.map.assign(nnz, 0);
    std::vector<int32_t> cp(n + 1, 0), bcnt(n + 1, 0), lvl_in(n, 0);
    // pass 1: classify, count, in-block levels
    auto classify = [&](int64_t q, int32_t c, int& kind) -> int {
        if (c == q) { kind = upper ? 3 : 0; return NEP_OK; }
        if (rowblk[c] == rowblk[q]) {
            if (upper ? (c < q) : (c > q)) return NEP_ERR_ARG;
            kind = 2; return NEP_OK;
        }
        if (upper ? (rowlev[c] <= rowlev[q]) : (rowlev[c] >= rowlev[q])) return NEP_ERR_ARG;
        kind = 1; return NEP_OK;
    };
    int bad = 0;
    auto row_pass1 = [&](int64_t q) {
        const int32_t i = oldof[q];
        int lv = 0, ncp = 0, nbk = 0;
        for (int32_t e = rp[i]; e < rp[i + 1]; ++e) {
            const int32_t c = newpos[ci[e]];
            int kind = 0;
            if (classify(q, c, kind)) { bad = 1; continue; }
            if (kind == 1) ++ncp;
            else if (kind == 2) { ++nbk; lv = std::max(lv, lvl_in[c] + 1); }
        }
        lvl_in[q] = lv; cp[q + 1] = ncp; bcnt[q + 1] = nbk;
    };
    if (!upper) for (int64_t q = 0; q < n; ++q) row_pass1(q);
    else for (int64_t q = n - 1; q >= 0; --q) row_pass1(q);
    if (bad) return NEP_ERR_ARG;
    for (int64_t q = 0; q < n; ++q) cp[q + 1] += cp[q];
    F.ncoup = cp[n];
    // slots: rows of each block sorted by in-block level
    std::vector<int32_t> slot_of(n), slotrow(n), lvo(S.nblk + 1, 0), lvp;
    lvp.reserve(S.nblk * 8);
    {
        std::vector<int32_t> cnt;
        for (int k = 0; k < S.nblk; ++k) {
            const int32_t s = blk_se[2 * k], e = blk_se[2 * k + 1];
            int nl = 0;
            for (int32_t q = s; q < e; ++q) nl = std::max(nl, lvl_in[q] + 1);
            cnt.assign(nl + 1, 0);
            for (int32_t q = s; q < e; ++q) cnt[lvl_in[q] + 1]++;
            for (int l = 0; l < nl; ++l) cnt[l + 1] += cnt[l];
            lvo[k] = (int32_t)lvp.size();
            for (int l = 0; l <= nl; ++l) lvp.push_back(s + cnt[l]);
            for (int32_t q = s; q < e; ++q) { const int32_t sl = s + cnt[lvl_in[q]]++; slot_of[q] = sl; slotrow[sl] = q - s; }
        }
        lvo[S.nblk] = (int32_t)lvp.size();
    }
    std::vector<int32_t> bp(n + 1, 0);
    for (int64_t q = 0; q < n; ++q) bp[slot_of[q] + 1] = bcnt[q + 1];
    for (int64_t q = 0; q < n; ++q) bp[q + 1] += bp[q];
    F.nin = bp[n];
    std::vector<int64_t> ip(n + 1, 0);
    for (int64_t q = 0; q < n; ++q) {
        const int k = rowblk[q];
        ip[q + 1] = ip[q] + (upper ? blk_se[2 * k + 1] - q : q - blk_se[2 * k] + 1);
    }
    F.ninv = ip[n];
    // pass 2: fill
    std::vector<int32_t> cci(std::max<int64_t>(F.ncoup, 1)), bi(std::max<int64_t>(F.nin, 1));
    for (int64_t q = 0; q < n; ++q) {
        const int32_t i = oldof[q];
        int32_t pc = cp[q], pb = bp[slot_of[q]];
        const int32_t s = blk_se[2 * rowblk[q]];
        for (int32_t e = rp[i]; e < rp[i + 1]; ++e) {
            const int32_t c = newpos[ci[e]];
            const int32_t se = src ? src[e] : e;
            if (c == q) { F.map[se] = upper ? (int32_t)(q * 4 + 3) : 0; continue; }
            if (rowblk[c] == rowblk[q]) { bi[pb] = c - s; F.map[se] = pb * 4 + 2; ++pb; }
            else { cci[pc] = c; F.map[se] = pc * 4 + 1; ++pc; }
        }
    }
    if (F.ncoup >= ((int64_t)1 << 29) || F.nin >= ((int64_t)1 << 29)) { nep_set_error("factor too large for the 32-bit slot map"); return NEP_ERR_ARG; }
    // per level: coupling non-zeros; whether the coupling product runs inside the level kernel (one thread per row, formed
    // redundantly by every chunk of a block) or as its own launch; rows per chunk
    F.split.assign(S.nlev, 0); F.cpl_lanes.assign(S.nlev, 8); F.lev_coup.assign(S.nlev, 0); F.lev_ch.assign(S.nlev, 4);
    F.lev_chunk.assign(S.nlev + 1, 0);
    F.lev_seg.assign(S.nlev + 1, 0);
    std::vector<MLChunk> chunks;
    std::vector<int32_t> segs;
    const char* fs = getenv("NEP_ML_SPLIT");       // experiment knobs: 0 = always fused, 1 = always split; rows per chunk
    int ch_env = 0;
    if (const char* e = getenv("NEP_ML_CHUNK")) { const int v = atoi(e); if (v == 4 || v == 16 || v == 32) ch_env = v; }
    for (int l = 0; l < S.nlev; ++l) {
        const int32_t r0 = S.lev_row[l], r1 = S.lev_row[l + 1];
        const int64_t nz = cp[r1] - cp[r0];
        F.lev_coup[l] = nz;
        const double avg = nz / (double)std::max(1, r1 - r0);
        int32_t maxrow = 0;
        for (int32_t q = r0; q < r1; ++q) maxrow = std::max(maxrow, cp[q + 1] - cp[q]);
        F.cpl_lanes[l] = avg > 2048.0 ? 256 : (avg > 24.0 ? 64 : 8);
        // fused: 16 rows per chunk (the chunk recomputes r for up to a whole block); split or no coupling: 4 rows per chunk
        int64_t redundant = 0;
        for (int k = S.lev_blk[l]; k < S.lev_blk[l + 1]; ++k) {
            const int32_t s = blk_se[2 * k], e = blk_se[2 * k + 1];
            const int64_t nch = (e - s + 15) / 16;
            redundant += (int64_t)(cp[e] - cp[s]) * (nch + 1) / 2;
        }
        // measured on gun (U level 0: 10 non-zeros per row on average, 60 at most): fused 37 us (the thread with the longest
        // row walks it alone, one dependent gather per entry) against 4.5 + 4.5 us for the two launches
        bool split = nz > 0 && (maxrow > 8 || redundant > 1500000);
        if (fs && nz > 0) split = atoi(fs) != 0;
        F.split[l] = split ? 1 : 0;
        int CH = (nz == 0 || split) ? 4 : 16;
        if (ch_env) CH = ch_env;
        F.lev_ch[l] = CH;
        F.lev_chunk[l] = (int32_t)chunks.size();
        F.lev_seg[l] = (int32_t)segs.size();
        for (int k = S.lev_blk[l]; k < S.lev_blk[l + 1]; ++k) {
            const int32_t s = blk_se[2 * k], e = blk_se[2 * k + 1];
            for (int32_t a = s; a < e; a += CH) {
                if ((a - s) % ML_BLK_SEG == 0) segs.push_back((int32_t)chunks.size() - F.lev_chunk[l]);
                chunks.push_back(MLChunk{a, std::min(a + CH, e), s, e, ip[a]});
            }
        }
    }
    F.lev_chunk[S.nlev] = (int32_t)chunks.size();
    F.lev_seg[S.nlev] = (int32_t)segs.size();
    int rc;
    if ((rc = up(&F.d_cp, cp))) return rc;
    if ((rc = up(&F.d_ci, cci))) return rc;
    if ((rc = up(&F.d_ip, ip))) return rc;
    if ((rc = up(&F.d_bp, bp))) return rc;
    if ((rc = up(&F.d_bi, bi))) return rc;
    if ((rc = up(&F.d_slotrow, slotrow))) return rc;
    if ((rc = up(&F.d_lvp, lvp))) return rc;
    if ((rc = up(&F.d_lvo, lvo))) return rc;
    {
        std::vector<int32_t> rl(lvl_in.begin(), lvl_in.end());
        if ((rc = up(&F.d_rowlev, rl))) return rc;
    }
    if ((rc = up(&F.d_chunks, chunks))) return rc;
    if ((rc = up(&F.d_segs, segs))) return rc;
    return NEP_OK;
}

// symbolic analysis.  Lrp/Lci and Urp/Uci: CSR patterns of L and U; UTp/UTi: CSR pattern of U^T (= CSC of U);
// Lsrc/Usrc: source index per CSR entry (nullptr = identity)
int build_symbolic(MLSym* S, int64_t n, const int32_t* Lrp, const int32_t* Lci, const int32_t* Lsrc, const int32_t* Urp,
                   const int32_t* Uci, const int32_t* Usrc, const int32_t* UTp, const int32_t* UTi,
                   const int32_t* perm_r, const int32_t* perm_c) {
    S->n = n;
    // ---- validate triangularity
    for (int64_t i = 0; i < n; ++i) {
        for (int32_t e = Lrp[i]; e < Lrp[i + 1]; ++e) {
            if (Lci[e] < 0 || Lci[e] >= n) { nep_set_error("L: column out of range"); return NEP_ERR_ARG; }
            if (Lci[e] > i) { nep_set_error("L is not lower triangular (row %lld col %d)", (long long)i, Lci[e]); return NEP_ERR_ARG; }
        }
        for (int32_t e = Urp[i]; e < Urp[i + 1]; ++e) {
            if (Uci[e] < 0 || Uci[e] >= n) { nep_set_error("U: column out of range"); return NEP_ERR_ARG; }
            if (Uci[e] < i) { nep_set_error("U is not upper triangular (row %lld col %d)", (long long)i, Uci[e]); return NEP_ERR_ARG; }
        }
    }
    // ---- elimination tree of struct(L) + struct(U)^T (Liu, path compression)
    std::vector<int32_t> parent(n, -1), anc(n, -1);
    for (int64_t i = 0; i < n; ++i) {
        for (int pass = 0; pass < 2; ++pass) {
            const int32_t* P = pass ? UTp : Lrp; const int32_t* I = pass ? UTi : Lci;
            for (int32_t e = P[i]; e < P[i + 1]; ++e) {
                int32_t k = I[e];
                while (k >= 0 && k < i) {
                    const int32_t nx = anc[k];
                    anc[k] = (int32_t)i;
                    if (nx < 0) { parent[k] = (int32_t)i; break; }
                    k = nx;
                }
            }
        }
    }
    // ---- multilevel partition in ONE ascending pass: lvl[j] = level, rsz[j] = nodes of j's subtree in j's level
    std::vector<int32_t> lvl(n, 0), rsz(n, 1), pmax(n, -1), psum(n, 0);
    int nlev = 0;
    const int forced_bmax = bmax_env();
    for (int64_t j = 0; j < n; ++j) {
        const int M = std::max(pmax[j], 0);
        const int s = pmax[j] >= 0 ? psum[j] : 0;
        if (s + 1 <= bmax_of_level(M, n, forced_bmax)) { lvl[j] = M; rsz[j] = s + 1; }
        else { lvl[j] = M + 1; rsz[j] = 1; }
        nlev = std::max(nlev, lvl[j] + 1);
        const int32_t p = parent[j];
        if (p >= 0) {
            if (lvl[j] > pmax[p]) { pmax[p] = lvl[j]; psum[p] = rsz[j]; }
            else if (lvl[j] == pmax[p]) psum[p] += rsz[j];
        }
    }
    std::vector<int32_t> bid(n);
    for (int64_t j = n - 1; j >= 0; --j) {
        const int32_t p = parent[j];
        bid[j] = (p >= 0 && lvl[p] == lvl[j]) ? bid[p] : (int32_t)j;
    }
    // ---- new order: levels in sequence, blocks of a level by root index, rows of a block by original index
    S->nlev = nlev;
    std::vector<int32_t> lev_nblk(nlev + 1, 0);
    for (int64_t j = 0; j < n; ++j) if (bid[j] == j) lev_nblk[lvl[j] + 1]++;
    for (int l = 0; l < nlev; ++l) lev_nblk[l + 1] += lev_nblk[l];
    const int nblk = lev_nblk[nlev];
    S->nblk = nblk;
    std::vector<int32_t> blkid(n, -1), blk_root(nblk), pos(lev_nblk.begin(), lev_nblk.end() - 1);
    for (int64_t j = 0; j < n; ++j) if (bid[j] == j) { const int k = pos[lvl[j]]++; blkid[j] = k; blk_root[k] = (int32_t)j; }
    std::vector<int32_t> blk_se(2 * (size_t)nblk), fill(nblk, 0);
    S->lev_row.assign(nlev + 1, 0);
    {
        int32_t cur = 0;
        for (int l = 0; l < nlev; ++l) {
            S->lev_row[l] = cur;
            for (int k = lev_nblk[l]; k < lev_nblk[l + 1]; ++k) { blk_se[2 * k] = cur; cur += rsz[blk_root[k]]; blk_se[2 * k + 1] = cur; }
        }
        S->lev_row[nlev] = cur;
        if (cur != n) { nep_set_error("internal: block sizes do not add up"); return NEP_ERR_ARG; }
    }
    std::vector<int32_t> newpos(n), oldof(n), rowblk(n), rowlev(n);
    for (int64_t j = 0; j < n; ++j) {
        const int k = blkid[bid[j]];
        const int32_t q = blk_se[2 * k] + fill[k]++;
        newpos[j] = q; oldof[q] = (int32_t)j; rowblk[q] = k; rowlev[q] = lvl[j];
    }
    S->lev_blk = lev_nblk;
    S->h_lvl.assign(lvl.begin(), lvl.end()); S->h_oldof = oldof; S->h_blk_se = blk_se;
    S->h_blk.resize(n);
    for (int64_t j = 0; j < n; ++j) S->h_blk[j] = blkid[bid[j]];
    for (int k = 0; k < nblk; ++k) S->max_block = std::max(S->max_block, blk_se[2 * k + 1] - blk_se[2 * k]);
    // ---- permutations folded into the first and last launch
    std::vector<int32_t> pin(n), pout(n);
    {
        std::vector<int32_t> ipr, ipc;
        auto invert = [&](const int32_t* p, std::vector<int32_t>& ip) -> int {
            ip.assign(n, -1);
            for (int64_t i = 0; i < n; ++i) {
                if (p[i] < 0 || p[i] >= n || ip[p[i]] >= 0) { nep_set_error("invalid permutation"); return NEP_ERR_ARG; }
                ip[p[i]] = (int32_t)i;
            }
            return NEP_OK;
        };
        if (perm_r) { int rc = invert(perm_r, ipr); if (rc) return rc; }
        if (perm_c) { int rc = invert(perm_c, ipc); if (rc) return rc; }
        for (int64_t q = 0; q < n; ++q) {
            pin[q] = perm_r ? ipr[oldof[q]] : oldof[q];        // (Pr b)[perm_r[i]] = b[i]
            pout[q] = perm_c ? ipc[oldof[q]] : oldof[q];       // x[i] = y[perm_c[i]]
        }
    }
    int rc;
    if ((rc = up(&S->d_blk_se, blk_se))) return rc;
    if ((rc = up(&S->d_rowblk, rowblk))) return rc;
    if ((rc = up(&S->d_pin, pin))) return rc;
    if ((rc = up(&S->d_pout, pout))) return rc;
    rc = build_factor(*S, false, Lrp, Lci, Lsrc, newpos, oldof, rowblk, blk_se, rowlev, S->L);
    if (rc == NEP_ERR_ARG) { nep_set_error("block schedule: L has a dependency outside the elimination tree"); return NEP_ERR_UNSUPPORTED; }
    if (rc) return rc;
    rc = build_factor(*S, true, Urp, Uci, Usrc, newpos, oldof, rowblk, blk_se, rowlev, S->U);
    if (rc == NEP_ERR_ARG) { nep_set_error("block schedule: U has a dependency outside the elimination tree"); return NEP_ERR_UNSUPPORTED; }
    return rc;
}

}  // namespace

static int ml_build_apex(MLFactor* F, hipStream_t bst);
static int ml_build_apex_dense(MLFactor* F, hipStream_t bst);
static int choose_apex(const MLSym* S, int expected_solves);
// end of a numeric build on bst: `ready` = block inverses done (solves may start), then the apex behind it
static int ml_finish_numeric(MLFactor* F, hipStream_t bst) {
    static const int apex_sync = getenv("NEP_ML_APEX_SYNC") ? atoi(getenv("NEP_ML_APEX_SYNC")) : 0;
    F->apex_live = false;
    F->solves_since_numeric = 0;
    F->synced_valid = false;
    if (F->graph) { (void)hipGraphExecDestroy(F->graph); F->graph = nullptr; }
    // NEP_ML_APEX_DENSE: 0 = the sparse build (T unit vectors through the level kernels, 3.4 ms on gun) behind `ready`; 1 = the dense build
    // finished BEFORE the first solve (apex from solve 1); 2 (default) = the dense build behind `ready`, the switch at solve NEP_ML_APEX_AT:
    // measured on the headline call 37.0 / 35.0-35.8 / 34.5 ms (the 1.07 ms build next to the first five solves disturbs them far less than
    // the sparse one did, and nothing waits for it)
    static const int apex_dense = getenv("NEP_ML_APEX_DENSE") ? atoi(getenv("NEP_ML_APEX_DENSE")) : 2;
    if (F->apex_la > 0 && apex_dense == 2) {    // dense build behind `ready`, switch at solve NEP_ML_APEX_AT like the sparse build
        HIPCHK(hipEventRecord(F->ready, bst));
        int rc = ml_build_apex_dense(F, bst);
        if (rc) return rc;
        if (!F->apex_ev) HIPCHK(hipEventCreateWithFlags(&F->apex_ev, hipEventDisableTiming));
        HIPCHK(hipEventRecord(F->apex_ev, bst));
        return NEP_OK;
    }
    if (F->apex_la > 0 && apex_dense) {         // dense build: short enough to finish before the first solve (no switch point)
        int rc = ml_build_apex_dense(F, bst);
        if (rc) return rc;
        F->apex_live = true;
        HIPCHK(hipEventRecord(F->ready, bst));
        return NEP_OK;
    }
    if (F->apex_la > 0 && apex_sync) {          // old behaviour: nothing may start before the apex exists
        int rc = ml_build_apex(F, bst);
        if (rc) return rc;
        F->apex_live = true;
        HIPCHK(hipEventRecord(F->ready, bst));
        return NEP_OK;
    }
    HIPCHK(hipEventRecord(F->ready, bst));
    if (F->apex_la > 0) {
        int rc = ml_build_apex(F, bst);
        if (rc) return rc;
        if (!F->apex_ev) HIPCHK(hipEventCreateWithFlags(&F->apex_ev, hipEventDisableTiming));
        HIPCHK(hipEventRecord(F->apex_ev, bst));
    }
    return NEP_OK;
}
static inline int eff_apex(const MLFactor* F) { return F->apex_live ? F->apex_la : 0; }

// ---- numeric part: gather the values into the schedule's order, upload, invert the diagonal blocks -----------------------
static int ml_numeric(MLFactor* F, const nep_cdouble* Lx, const nep_cdouble* Ux) {
    MLSym* S = F->sym;
    const int64_t n = S->n;
    const int64_t oL = 0, oLb = oL + S->L.ncoup, oU = oLb + S->L.nin, oUb = oU + S->U.ncoup, oD = oUb + S->U.nin;
    const int64_t ntot = oD + n;
    int pidx = -1;
    int rc = g_pinned.acquire((size_t)ntot * sizeof(cplx), &pidx);
    if (rc) return rc;
    nep_cdouble* h = (nep_cdouble*)g_pinned.ptr(pidx);
    for (int64_t q = 0; q < n; ++q) { h[oD + q].re = 0.0; h[oD + q].im = 0.0; }
    {
        const int32_t* m = S->L.map.data();
        for (int64_t e = 0; e < S->nnzL; ++e) {
            const int32_t v = m[e];
            const int kind = v & 3;
            if (kind == 1) h[oL + (v >> 2)] = Lx[e];
            else if (kind == 2) h[oLb + (v >> 2)] = Lx[e];
        }
        m = S->U.map.data();
        for (int64_t e = 0; e < S->nnzU; ++e) {
            const int32_t v = m[e];
            const int kind = v & 3;
            if (kind == 1) h[oU + (v >> 2)] = Ux[e];
            else if (kind == 2) h[oUb + (v >> 2)] = Ux[e];
            else if (kind == 3) h[oD + (v >> 2)] = Ux[e];
        }
    }
    for (int64_t q = 0; q < n; ++q)
        if (h[oD + q].re == 0.0 && h[oD + q].im == 0.0) {
            g_pinned.release(pidx, nullptr);
            nep_set_error("U has a zero pivot (matrix is singular)");
            return NEP_ERR_SINGULAR;
        }
    hipStream_t bst = g_bstreams.get();
    if (!F->d_vals) {
        if ((rc = nep_pool_alloc((void**)&F->d_vals, (size_t)ntot * sizeof(cplx)))) { g_pinned.release(pidx, nullptr); return rc; }
        if ((rc = nep_pool_alloc((void**)&F->d_ixL, (size_t)std::max<int64_t>(S->L.ninv, 1) * sizeof(cplx)))) { g_pinned.release(pidx, nullptr); return rc; }
        if ((rc = nep_pool_alloc((void**)&F->d_ixU, (size_t)std::max<int64_t>(S->U.ninv, 1) * sizeof(cplx)))) { g_pinned.release(pidx, nullptr); return rc; }
        HIPCHK(hipEventCreateWithFlags(&F->ready, hipEventDisableTiming));
    } else if (F->used && F->last != bst) {
        // refactorisation: the solves in flight read the old values
        hipEvent_t ev; HIPCHK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        HIPCHK(hipEventRecord(ev, F->last));
        HIPCHK(hipStreamWaitEvent(bst, ev, 0));
        (void)hipEventDestroy(ev);
    }
    HIPCHK(hipMemcpyAsync(F->d_vals, h, (size_t)ntot * sizeof(cplx), hipMemcpyHostToDevice, bst));
    g_pinned.release(pidx, bst);
    hipLaunchKernelGGL((k_ml_inverse<false>), dim3((unsigned)n), dim3(ML_INV_NT), 0, bst, (const int32_t*)S->d_rowblk,
                       (const int32_t*)S->d_blk_se, (const int32_t*)S->L.d_lvo, (const int32_t*)S->L.d_lvp,
                       (const int32_t*)S->L.d_slotrow, (const int32_t*)S->L.d_bp, (const int32_t*)S->L.d_bi,
                       (const cplx*)(F->d_vals + oLb), (const cplx*)nullptr, (const int64_t*)S->L.d_ip, F->d_ixL, (const int32_t*)S->L.d_rowlev);
    LAUNCHCHK();
    hipLaunchKernelGGL((k_ml_inverse<true>), dim3((unsigned)n), dim3(ML_INV_NT), 0, bst, (const int32_t*)S->d_rowblk,
                       (const int32_t*)S->d_blk_se, (const int32_t*)S->U.d_lvo, (const int32_t*)S->U.d_lvp,
                       (const int32_t*)S->U.d_slotrow, (const int32_t*)S->U.d_bp, (const int32_t*)S->U.d_bi,
                       (const cplx*)(F->d_vals + oUb), (const cplx*)(F->d_vals + oD), (const int64_t*)S->U.d_ip, F->d_ixU, (const int32_t*)S->U.d_rowlev);
    LAUNCHCHK();
    if ((rc = ml_finish_numeric(F, bst))) return rc;
    return NEP_OK;
}

static void sym_release(MLSym* s) {
    std::lock_guard<std::mutex> lk(g_cache_mu);
    if (--s->refs > 0) return;
    // unreferenced patterns stay cached (LRU) up to ML_CACHE_MAX entries
    size_t idle = 0;
    for (MLSym* t : g_cache) if (t->refs == 0) ++idle;
    while (idle > ML_CACHE_MAX) {
        for (auto it = g_cache.rbegin(); it != g_cache.rend(); ++it)
            if ((*it)->refs == 0) { MLSym* dead = *it; g_cache.erase(std::next(it).base()); free_sym(dead); --idle; break; }
    }
}

int ml_create(int64_t n, int csc, const int32_t* Lp, const int32_t* Li, const nep_cdouble* Lx, const int32_t* Up,
              const int32_t* Ui, const nep_cdouble* Ux, const int32_t* perm_r, const int32_t* perm_c, int expected_solves,
              MLFactor** out) {
    *out = nullptr;
    const bool timing = getenv("NEP_TIMING") != nullptr;
    const double t0 = ml_now_ms();
    uint64_t h0 = 0x243F6A8885A308D3ull ^ (uint64_t)n, h1 = 0x13198A2E03707344ull + (uint64_t)csc;
    hash_words(h0, h1, Lp, (size_t)(n + 1) * 4); hash_words(h0, h1, Li, (size_t)Lp[n] * 4);
    hash_words(h0, h1, Up, (size_t)(n + 1) * 4); hash_words(h0, h1, Ui, (size_t)Up[n] * 4);
    if (perm_r) hash_words(h0, h1, perm_r, (size_t)n * 4);
    if (perm_c) hash_words(h0, h1, perm_c, (size_t)n * 4);
    h1 += (perm_r ? 1 : 0) + (perm_c ? 2 : 0);
    for (const char* kn : {"NEP_ML_BMAX", "NEP_ML_SPLIT", "NEP_ML_CHUNK"}) {     // experiment knobs change the schedule
        const char* e = getenv(kn);
        h0 = (h0 ^ (uint64_t)(e ? atoi(e) + 7 : 3)) * 0x9E3779B97F4A7C15ull;
    }
    const double t1 = ml_now_ms();
    MLSym* S = nullptr;
    bool hit = false;
    {
        std::lock_guard<std::mutex> lk(g_cache_mu);      // held during the symbolic build: concurrent creators of one
        const bool nocache = getenv("NEP_ML_NOCACHE") != nullptr;   // pattern (Beyn) wait instead of duplicating it
        if (!nocache)
            for (auto it = g_cache.begin(); it != g_cache.end(); ++it)
                if ((*it)->key0 == h0 && (*it)->key1 == h1 && (*it)->n == n && (*it)->nnzL == Lp[n] && (*it)->nnzU == Up[n]) {
                    S = *it; g_cache.erase(it); g_cache.push_front(S); hit = true; break;
                }
        if (!S) {
            S = new MLSym();
            S->key0 = h0; S->key1 = h1; S->nnzL = Lp[n]; S->nnzU = Up[n]; S->csc = csc;
            int rc;
            if (csc) {
                std::vector<int32_t> Lrp, Lci, Lsrc, Urp, Uci, Usrc;
                transpose_pattern(n, Lp, Li, Lrp, Lci, &Lsrc);
                transpose_pattern(n, Up, Ui, Urp, Uci, &Usrc);
                rc = build_symbolic(S, n, Lrp.data(), Lci.data(), Lsrc.data(), Urp.data(), Uci.data(), Usrc.data(), Up, Ui,
                                    perm_r, perm_c);
            } else {
                std::vector<int32_t> UTp, UTi;
                transpose_pattern(n, Up, Ui, UTp, UTi, nullptr);
                rc = build_symbolic(S, n, Lp, Li, nullptr, Up, Ui, nullptr, UTp.data(), UTi.data(), perm_r, perm_c);
            }
            if (rc) { free_sym(S); return rc; }
            S->t_build_ms = ml_now_ms() - t1;
            g_cache.push_front(S);
        }
        S->refs++;
    }
    const double t2 = ml_now_ms();
    MLFactor* F = new MLFactor();
    F->sym = S;
    F->use_graph = expected_solves >= 3 ? 1 : 0;
    F->apex_la = choose_apex(S, expected_solves);
    int rc = ml_numeric(F, Lx, Ux);
    if (rc) { ml_destroy(F); return rc; }
    if (timing) {
        fprintf(stderr, "[ml_create] n=%lld levels=%d blocks=%d hash %.3f ms, symbolic %s %.3f ms, numeric (host) %.3f ms\n",
                (long long)n, S->nlev, S->nblk, t1 - t0, hit ? "hit" : "built", t2 - t1, ml_now_ms() - t2);
        if (!hit)
            for (int l = 0; l < S->nlev; ++l)
                fprintf(stderr, "[ml_create]   level %d: rows %d blocks %d | L coupling %lld %s ch %d | U coupling %lld %s ch %d\n", l,
                        S->lev_row[l + 1] - S->lev_row[l], S->lev_blk[l + 1] - S->lev_blk[l], (long long)S->L.lev_coup[l],
                        S->L.split[l] ? "split" : "fused", S->L.lev_ch[l], (long long)S->U.lev_coup[l],
                        S->U.split[l] ? "split" : "fused", S->U.lev_ch[l]);
    }
    *out = F;
    return NEP_OK;
}

int ml_refactor(MLFactor* F, const nep_cdouble* Lx, const nep_cdouble* Ux) { return ml_numeric(F, Lx, Ux); }

// numeric part with the factor values already on the device (d_Lx / d_Ux in the input entry order, produced on `producer`)
static int ml_numeric_dev(MLFactor* F, const cplx* d_Lx, const cplx* d_Ux, hipStream_t producer) {
    MLSym* S = F->sym;
    const int64_t n = S->n;
    const int64_t oL = 0, oLb = oL + S->L.ncoup, oU = oLb + S->L.nin, oUb = oU + S->U.ncoup, oD = oUb + S->U.nin;
    const int64_t ntot = oD + n;
    int rc;
    {
        std::lock_guard<std::mutex> lk(g_cache_mu);
        if (!S->L.d_map && (rc = up(&S->L.d_map, S->L.map))) return rc;
        if (!S->U.d_map && (rc = up(&S->U.d_map, S->U.map))) return rc;
    }
    hipStream_t bst = g_bstreams.get();
    if (!F->d_vals) {
        if ((rc = nep_pool_alloc((void**)&F->d_vals, (size_t)ntot * sizeof(cplx)))) return rc;
        if ((rc = nep_pool_alloc((void**)&F->d_ixL, (size_t)std::max<int64_t>(S->L.ninv, 1) * sizeof(cplx)))) return rc;
        if ((rc = nep_pool_alloc((void**)&F->d_ixU, (size_t)std::max<int64_t>(S->U.ninv, 1) * sizeof(cplx)))) return rc;
        HIPCHK(hipEventCreateWithFlags(&F->ready, hipEventDisableTiming));
    }
    {
        hipEvent_t ev; HIPCHK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        HIPCHK(hipEventRecord(ev, producer)); HIPCHK(hipStreamWaitEvent(bst, ev, 0)); (void)hipEventDestroy(ev);
    }
    const int gl = (int)std::min<int64_t>((S->nnzL + 255) / 256, 4096), gu = (int)std::min<int64_t>((S->nnzU + 255) / 256, 4096);
    hipLaunchKernelGGL(k_ml_gather, dim3(gl), dim3(256), 0, bst, S->nnzL, (const int32_t*)S->L.d_map, d_Lx, F->d_vals, oL, oLb, oD);
    LAUNCHCHK();
    hipLaunchKernelGGL(k_ml_gather, dim3(gu), dim3(256), 0, bst, S->nnzU, (const int32_t*)S->U.d_map, d_Ux, F->d_vals, oU, oUb, oD);
    LAUNCHCHK();
    // the two block-inverse builds are independent (and each a chain of dependent in-block levels that leaves most of the GPU idle):
    // the U side runs on a second build stream next to the L side (NEP_ML_INV_2STREAM=0: one after the other)
    static const int two = getenv("NEP_ML_INV_2STREAM") ? atoi(getenv("NEP_ML_INV_2STREAM")) : 1;
    hipStream_t bst2 = two ? g_bstreams.get() : bst;
    if (bst2 == bst || !bst2) bst2 = bst;
    if (bst2 != bst) {
        hipEvent_t ev; HIPCHK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        HIPCHK(hipEventRecord(ev, bst)); HIPCHK(hipStreamWaitEvent(bst2, ev, 0)); (void)hipEventDestroy(ev);
    }
    hipLaunchKernelGGL((k_ml_inverse<false>), dim3((unsigned)n), dim3(ML_INV_NT), 0, bst, (const int32_t*)S->d_rowblk,
                       (const int32_t*)S->d_blk_se, (const int32_t*)S->L.d_lvo, (const int32_t*)S->L.d_lvp,
                       (const int32_t*)S->L.d_slotrow, (const int32_t*)S->L.d_bp, (const int32_t*)S->L.d_bi,
                       (const cplx*)(F->d_vals + oLb), (const cplx*)nullptr, (const int64_t*)S->L.d_ip, F->d_ixL, (const int32_t*)S->L.d_rowlev);
    LAUNCHCHK();
    hipLaunchKernelGGL((k_ml_inverse<true>), dim3((unsigned)n), dim3(ML_INV_NT), 0, bst2, (const int32_t*)S->d_rowblk,
                       (const int32_t*)S->d_blk_se, (const int32_t*)S->U.d_lvo, (const int32_t*)S->U.d_lvp,
                       (const int32_t*)S->U.d_slotrow, (const int32_t*)S->U.d_bp, (const int32_t*)S->U.d_bi,
                       (const cplx*)(F->d_vals + oUb), (const cplx*)(F->d_vals + oD), (const int64_t*)S->U.d_ip, F->d_ixU, (const int32_t*)S->U.d_rowlev);
    LAUNCHCHK();
    if (bst2 != bst) {
        hipEvent_t ev; HIPCHK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        HIPCHK(hipEventRecord(ev, bst2)); HIPCHK(hipStreamWaitEvent(bst, ev, 0)); (void)hipEventDestroy(ev);
    }
    if ((rc = ml_finish_numeric(F, bst))) return rc;
    return NEP_OK;
}

// ---- interface of the device-side numeric factorisation (csrc/lufac.hip) -------------------------------------------------
MLSym* ml_sym_acquire(MLFactor* F) {
    std::lock_guard<std::mutex> lk(g_cache_mu);
    F->sym->refs++;
    return F->sym;
}
void ml_sym_release_ref(MLSym* S) { if (S) sym_release(S); }
void ml_sym_partition(const MLSym* S, int64_t* n, int* nlev, int* nblk, const int32_t** lvl, const int32_t** blk,
                      const int32_t** oldof, const int32_t** blk_se, const int32_t** lev_blk) {
    *n = S->n; *nlev = S->nlev; *nblk = S->nblk; *lvl = S->h_lvl.data(); *blk = S->h_blk.data(); *oldof = S->h_oldof.data();
    *blk_se = S->h_blk_se.data(); *lev_blk = S->lev_blk.data();
}
int ml_wait_ready(MLFactor* F, hipStream_t st) {
    HIPCHK(hipStreamWaitEvent(st, F->ready, 0));
    F->synced = st; F->synced_valid = true;
    return NEP_OK;
}
// B factors of ONE pattern in one go (the quadrature nodes of contour_beyn, lufac.hip): the value gathers and the block-inverse builds of
// all of them in four launches with grid.y = factor.  One factor at a time, k_ml_inverse is a chain of dependent in-block levels that
// leaves most of the chip idle for 0.3 ms -- 64 nodes were 128 such launches over four build streams, 39 ms of the 100 ms of config C4.
// d_Lx[b] / d_Ux[b]: device value arrays (input entry order) of factor b.  out[b] receives the factors (all or none).
int ml_create_from_sym_batch(MLSym* S, int B, const nep_cdouble* const* d_Lx, const nep_cdouble* const* d_Ux, hipStream_t producer,
                             int expected_solves, MLFactor** out) {
    for (int b = 0; b < B; ++b) out[b] = nullptr;
    if (B <= 0) return NEP_OK;
    const int64_t n = S->n;
    const int64_t oL = 0, oLb = oL + S->L.ncoup, oU = oLb + S->L.nin, oUb = oU + S->U.ncoup, oD = oUb + S->U.nin;
    const int64_t ntot = oD + n;
    int rc;
    {
        std::lock_guard<std::mutex> lk(g_cache_mu);
        if (!S->L.d_map && (rc = up(&S->L.d_map, S->L.map))) return rc;
        if (!S->U.d_map && (rc = up(&S->U.d_map, S->U.map))) return rc;
    }
    auto undo = [&](int code) { for (int b = 0; b < B; ++b) if (out[b]) { ml_destroy(out[b]); out[b] = nullptr; } return code; };
    std::vector<void*> tab((size_t)5 * B);          // [vals | ixL | ixU | srcL | srcU] x B
    for (int b = 0; b < B; ++b) {
        { std::lock_guard<std::mutex> lk(g_cache_mu); S->refs++; }
        MLFactor* F = new MLFactor();
        F->sym = S; F->use_graph = expected_solves >= 3 ? 1 : 0; F->apex_la = choose_apex(S, expected_solves);
        out[b] = F;
        if ((rc = nep_pool_alloc((void**)&F->d_vals, (size_t)ntot * sizeof(cplx)))) return undo(rc);
        if ((rc = nep_pool_alloc((void**)&F->d_ixL, (size_t)std::max<int64_t>(S->L.ninv, 1) * sizeof(cplx)))) return undo(rc);
        if ((rc = nep_pool_alloc((void**)&F->d_ixU, (size_t)std::max<int64_t>(S->U.ninv, 1) * sizeof(cplx)))) return undo(rc);
        if (hipEventCreateWithFlags(&F->ready, hipEventDisableTiming) != hipSuccess) return undo(NEP_ERR_HIP);
        tab[b] = F->d_vals; tab[(size_t)B + b] = F->d_ixL; tab[(size_t)2 * B + b] = F->d_ixU;
        tab[(size_t)3 * B + b] = (void*)d_Lx[b]; tab[(size_t)4 * B + b] = (void*)d_Ux[b];
    }
    void** d_tab = nullptr;
    if ((rc = nep_pool_alloc((void**)&d_tab, tab.size() * sizeof(void*)))) return undo(rc);
    hipStream_t bst = g_bstreams.get();
    {
        hipEvent_t ev;
        if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) { nep_pool_free(d_tab); return undo(NEP_ERR_HIP); }
        (void)hipEventRecord(ev, producer); (void)hipStreamWaitEvent(bst, ev, 0); (void)hipEventDestroy(ev);
    }
    {   // through a pinned staging slot: the table may go out of scope right after the call
        static thread_local PinnedRing ring;
        if ((rc = ring.upload(d_tab, tab.data(), tab.size() * sizeof(void*), bst))) { nep_pool_free(d_tab); return undo(rc); }
    }
    cplx* const* t_vals = (cplx* const*)d_tab; cplx* const* t_ixL = (cplx* const*)(d_tab + B); cplx* const* t_ixU = (cplx* const*)(d_tab + 2 * (size_t)B);
    const cplx* const* t_sL = (const cplx* const*)(d_tab + 3 * (size_t)B); const cplx* const* t_sU = (const cplx* const*)(d_tab + 4 * (size_t)B);
    const int gl = (int)std::min<int64_t>((S->nnzL + 255) / 256, 4096), gu = (int)std::min<int64_t>((S->nnzU + 255) / 256, 4096);
    hipLaunchKernelGGL(k_ml_gather, dim3(gl, B), dim3(256), 0, bst, S->nnzL, (const int32_t*)S->L.d_map, (const cplx*)nullptr, (cplx*)nullptr, oL, oLb, oD, t_vals, t_sL);
    hipLaunchKernelGGL(k_ml_gather, dim3(gu, B), dim3(256), 0, bst, S->nnzU, (const int32_t*)S->U.d_map, (const cplx*)nullptr, (cplx*)nullptr, oU, oUb, oD, t_vals, t_sU);
    MLBatchTab tl; tl.vals = t_vals; tl.ix = t_ixL; tl.off_bx = oLb; tl.off_diag = 0;
    MLBatchTab tu; tu.vals = t_vals; tu.ix = t_ixU; tu.off_bx = oUb; tu.off_diag = oD;
    hipLaunchKernelGGL((k_ml_inverse<false>), dim3((unsigned)n, (unsigned)B), dim3(ML_INV_NT), 0, bst, (const int32_t*)S->d_rowblk,
                       (const int32_t*)S->d_blk_se, (const int32_t*)S->L.d_lvo, (const int32_t*)S->L.d_lvp,
                       (const int32_t*)S->L.d_slotrow, (const int32_t*)S->L.d_bp, (const int32_t*)S->L.d_bi,
                       (const cplx*)nullptr, (const cplx*)nullptr, (const int64_t*)S->L.d_ip, (cplx*)nullptr, (const int32_t*)S->L.d_rowlev, tl);
    hipLaunchKernelGGL((k_ml_inverse<true>), dim3((unsigned)n, (unsigned)B), dim3(ML_INV_NT), 0, bst, (const int32_t*)S->d_rowblk,
                       (const int32_t*)S->d_blk_se, (const int32_t*)S->U.d_lvo, (const int32_t*)S->U.d_lvp,
                       (const int32_t*)S->U.d_slotrow, (const int32_t*)S->U.d_bp, (const int32_t*)S->U.d_bi,
                       (const cplx*)nullptr, (const cplx*)nullptr, (const int64_t*)S->U.d_ip, (cplx*)nullptr, (const int32_t*)S->U.d_rowlev, tu);
    if (hipGetLastError() != hipSuccess) { nep_pool_free_on(d_tab, bst, true); nep_set_error("batched numeric build failed"); return undo(NEP_ERR_HIP); }
    nep_pool_free_on(d_tab, bst, true);
    for (int b = 0; b < B; ++b)
        if ((rc = ml_finish_numeric(out[b], bst))) return undo(rc);
    return NEP_OK;
}

int ml_create_from_sym(MLSym* S, const nep_cdouble* d_Lx, const nep_cdouble* d_Ux, hipStream_t producer, int expected_solves,
                       MLFactor** out) {
    *out = nullptr;
    { std::lock_guard<std::mutex> lk(g_cache_mu); S->refs++; }
    MLFactor* F = new MLFactor();
    F->sym = S;
    F->use_graph = expected_solves >= 3 ? 1 : 0;
    F->apex_la = choose_apex(S, expected_solves);
    int rc = ml_numeric_dev(F, (const cplx*)d_Lx, (const cplx*)d_Ux, producer);
    if (rc) { ml_destroy(F); return rc; }
    *out = F;
    return NEP_OK;
}

// host-only analysis (no device): out[0]=levels, out[1]=blocks, out[2]=largest block, out[3]/[4]=coupling non-zeros and
// packed inverse entries of L, out[5]/[6] of U, out[7]=levels with a separate coupling launch (L+U)
int ml_analyze(int64_t n, int csc, const int32_t* Lp, const int32_t* Li, const int32_t* Up, const int32_t* Ui, int64_t out[8]) {
    MLSym* S = new MLSym();
    g_ml_dry = true;
    int rc;
    if (csc) {
        std::vector<int32_t> Lrp, Lci, Lsrc, Urp, Uci, Usrc;
        transpose_pattern(n, Lp, Li, Lrp, Lci, &Lsrc);
        transpose_pattern(n, Up, Ui, Urp, Uci, &Usrc);
        rc = build_symbolic(S, n, Lrp.data(), Lci.data(), Lsrc.data(), Urp.data(), Uci.data(), Usrc.data(), Up, Ui, nullptr, nullptr);
    } else {
        std::vector<int32_t> UTp, UTi;
        transpose_pattern(n, Up, Ui, UTp, UTi, nullptr);
        rc = build_symbolic(S, n, Lp, Li, nullptr, Up, Ui, nullptr, UTp.data(), UTi.data(), nullptr, nullptr);
    }
    g_ml_dry = false;
    if (rc == NEP_OK) {
        out[0] = S->nlev; out[1] = S->nblk; out[2] = S->max_block;
        out[3] = S->L.ncoup; out[4] = S->L.ninv; out[5] = S->U.ncoup; out[6] = S->U.ninv;
        int64_t sp = 0;
        for (int l = 0; l < S->nlev; ++l) sp += S->L.split[l] + S->U.split[l];
        out[7] = sp;
    }
    delete S;
    return rc;
}

// host-only symbolic analysis of CSC factors (no device): the partition a plan of csrc/lufac.hip is built on
// (nep_lu_refac_analyze).  The caller owns the result and frees it with ml_sym_free_host.
int ml_sym_build_host(int64_t n, const int32_t* Lp, const int32_t* Li, const int32_t* Up, const int32_t* Ui, const int32_t* perm_r,
                      const int32_t* perm_c, MLSym** out) {
    *out = nullptr;
    MLSym* S = new MLSym();
    g_ml_dry = true;
    std::vector<int32_t> Lrp, Lci, Lsrc, Urp, Uci, Usrc;
    transpose_pattern(n, Lp, Li, Lrp, Lci, &Lsrc);
    transpose_pattern(n, Up, Ui, Urp, Uci, &Usrc);
    const int rc = build_symbolic(S, n, Lrp.data(), Lci.data(), Lsrc.data(), Urp.data(), Uci.data(), Usrc.data(), Up, Ui, perm_r, perm_c);
    g_ml_dry = false;
    if (rc) { delete S; return rc; }
    *out = S;
    return NEP_OK;
}
void ml_sym_free_host(MLSym* S) { delete S; }

int ml_set_row_scale(MLFactor* F, const double* h_rs) {
    const int64_t n = F->sym->n;
    if (!h_rs) { if (F->d_rscale) nep_pool_free(F->d_rscale); F->d_rscale = nullptr; return NEP_OK; }
    if (!F->d_rscale) { int rc = nep_pool_alloc((void**)&F->d_rscale, (size_t)n * sizeof(double)); if (rc) return rc; }
    HIPCHK(hipMemcpy(F->d_rscale, h_rs, (size_t)n * sizeof(double), hipMemcpyHostToDevice));
    return NEP_OK;
}

void ml_destroy(MLFactor* F) {
    if (!F) return;
    // the solves enqueued on F->last may still read these blocks: the pool hands them out again only behind that work
    hipStream_t st = F->used ? F->last : nullptr;
    if (F->ready) { (void)hipEventSynchronize(F->ready); (void)hipEventDestroy(F->ready); }   // numeric build done (cheap: long past)
    if (F->apex_ev) { (void)hipEventSynchronize(F->apex_ev); (void)hipEventDestroy(F->apex_ev); }
    nep_pool_free_on(F->d_vals, st, F->used); nep_pool_free_on(F->d_ixL, st, F->used); nep_pool_free_on(F->d_ixU, st, F->used);
    nep_pool_free_on(F->d_rscale, st, F->used); nep_pool_free_on(F->d_Sinv, st, F->used);
    if (F->work.dptr) { nep_pool_free_on(F->work.dptr, st, F->used); F->work.dptr = nullptr; F->work.cap = 0; }
    if (F->graph) (void)hipGraphExecDestroy(F->graph);
    if (F->cap_stream) (void)hipStreamDestroy(F->cap_stream);
    if (F->d_fctr) nep_pool_free_on(F->d_fctr, st, F->used);
    if (F->h_ferr) { if (st) (void)hipStreamSynchronize(st); (void)hipHostFree(F->h_ferr); }
    if (F->sym) sym_release(F->sym);
    delete F;
}

void ml_info(const MLFactor* F, int64_t info[6], int64_t sched[8]) {
    const MLSym* S = F->sym;
    if (info) {
        info[0] = S->n; info[1] = S->nnzL; info[2] = S->nnzU;
        const int top = F->apex_la > 0 ? F->apex_la : S->nlev;
        info[3] = top + (F->apex_la > 0 ? 1 : 0); info[4] = info[3];
        // bytes one single-RHS solve moves: coupling (16 + 4 per non-zero), packed inverses, index arrays, vectors
        // (with an apex, the inverses / in-apex coupling of its levels are replaced by the dense T x T block: upper bound)
        const int64_t T = F->apex_la > 0 ? S->n - S->lev_row[F->apex_la] : 0;
        info[5] = (S->L.ncoup + S->U.ncoup) * 20 + (S->L.ninv + S->U.ninv) * 16 + 16 * T * T + 2 * (8 + 4) * (S->n + 1) + 6 * 16 * S->n;
    }
    if (sched) {
        sched[0] = F->apex_la > 0 ? S->n - S->lev_row[F->apex_la] : 0; sched[1] = F->launches; sched[2] = S->nlev; sched[3] = S->nlev;
        int64_t sp = 0;
        for (int l = 0; l < S->nlev; ++l) sp += S->L.split[l] + S->U.split[l];
        sched[4] = sp; sched[5] = S->nblk; sched[6] = S->n; sched[7] = S->max_block;
    }
}

// ---- launches ---------------------------------------------------------------------------------------------------------
// ---- single-launch form (one right-hand side): the launch helpers below append a phase record instead of launching when a
// recorder is active, and ml_solve issues ONE k_ml_fused launch that runs the phases back to back (see k_ml_fused)
enum { PH_LEVEL = 0, PH_COUPLING = 1, PH_APEX = 2 };
struct MLFusedPhase {
    int type, sel, nwg, pad;                 // PH_LEVEL: sel = UPPER*6 + MODE*3 + {0: G2 64, 1: G2 16, 2: G2 8}; PH_COUPLING: sel = lanes
    union U { MLArgs lv; MLCplArgs cp; MLApexArgs ax; __host__ __device__ U() {} } u;
};
#define ML_FUSE_MAXP 8
struct MLFusedArgs {
    int nph, total, mode, pad0;
    unsigned long long* ctr;                 // [0] ticket counter, [1 + p] workgroups of phase p that have finished (monotonic over solves)
    unsigned long long epoch;                // fused launches issued on this factor before this one
    int* err;                                // device-mapped host flag: set when a wait ran into its bound
    MLFusedPhase ph[ML_FUSE_MAXP];
};
struct MLRecorder { MLFusedArgs a; bool overflow; };
static thread_local MLRecorder* g_rec = nullptr;
static bool rec_push(MLFusedPhase** out, int type, int sel, int nwg) {
    MLRecorder* r = g_rec;
    if (r->a.nph >= ML_FUSE_MAXP) { r->overflow = true; return false; }
    MLFusedPhase* p = &r->a.ph[r->a.nph++];
    p->type = type; p->sel = sel; p->nwg = nwg; p->pad = 0;
    r->a.total += nwg;
    *out = p;
    return true;
}

// All phases of a single-vector solve as ONE launch.  Workgroups take a ticket (start order); the ticket decides phase and
// block index, so a workgroup only ever waits for workgroups that started before it -- no co-residency assumption, no
// deadlock.  A phase starts when every workgroup of the previous one has published its results: results are written, the
// wave executes an agent-scope release fence (L2 write-back: the 8 XCDs do not share an L2), one thread bumps the phase
// counter; the consumer polls the counter, then every wave executes an acquire fence (L1/L2 invalidate) before its first
// read.  The counters are monotonic over solves (target = (epoch + 1) * workgroups of the phase), so nothing is reset
// between launches.  Every wait is bounded: a bound hit sets *err (device-mapped host memory) and the workgroup leaves.
__global__ __launch_bounds__(256) void k_ml_fused(const MLFusedArgs F) {
    __shared__ int s_ticket;
    int t;
    if (F.mode & 1) {                                                  // experiment: trust in-order dispatch, no ticket atomics
        t = (int)blockIdx.x;
    } else {
        if (threadIdx.x == 0)
            s_ticket = (int)(atomicAdd(&F.ctr[0], 1ULL) - F.epoch * (unsigned long long)F.total);
        __syncthreads();
        t = __builtin_amdgcn_readfirstlane(s_ticket);                  // uniform: the phase record is read with scalar loads
    }
    int p = 0;
    while (p < F.nph - 1 && t >= F.ph[p].nwg) { t -= F.ph[p].nwg; ++p; }
    if (p > 0) {
        if (threadIdx.x == 0) {
            const unsigned long long target = (F.epoch + 1ULL) * (unsigned long long)F.ph[p - 1].nwg;
            int it = 0;
            while (__hip_atomic_load(&F.ctr[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                if (++it > (1 << 21)) { *F.err = 1; break; }
                if (F.mode & 2) __builtin_amdgcn_s_sleep(64); else __builtin_amdgcn_s_sleep(2);
            }
        }
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    const MLFusedPhase& ph = F.ph[p];
    if (ph.type == PH_LEVEL) {
        switch (ph.sel) {
            case 0:  ml_level_body<false, 1, 0, 64>(ph.u.lv, t, 0); break;
            case 1:  ml_level_body<false, 1, 0, 16>(ph.u.lv, t, 0); break;
            case 2:  ml_level_body<false, 1, 0, 8>(ph.u.lv, t, 0); break;
            case 3:  ml_level_body<false, 1, 1, 64>(ph.u.lv, t, 0); break;
            case 4:  ml_level_body<false, 1, 1, 16>(ph.u.lv, t, 0); break;
            case 5:  ml_level_body<false, 1, 1, 8>(ph.u.lv, t, 0); break;
            case 6:  ml_level_body<true, 1, 0, 64>(ph.u.lv, t, 0); break;
            case 7:  ml_level_body<true, 1, 0, 16>(ph.u.lv, t, 0); break;
            case 8:  ml_level_body<true, 1, 0, 8>(ph.u.lv, t, 0); break;
            case 9:  ml_level_body<true, 1, 1, 64>(ph.u.lv, t, 0); break;
            case 10: ml_level_body<true, 1, 1, 16>(ph.u.lv, t, 0); break;
            default: ml_level_body<true, 1, 1, 8>(ph.u.lv, t, 0); break;
        }
    } else if (ph.type == PH_COUPLING) {
        if (ph.sel == 256) ml_coupling_body<256, 1>(ph.u.cp, t, 0);
        else if (ph.sel == 64) ml_coupling_body<64, 1>(ph.u.cp, t, 0);
        else ml_coupling_body<8, 1>(ph.u.cp, t, 0);
    } else {
        apex_gemv_body<1>(ph.u.ax, t, 0);
    }
    if (p < F.nph - 1) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_fetch_add(&F.ctr[1 + p], 1ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

template <bool UPPER, int MODE, int G2>
static void launch_level_g(const MLArgs& a, int nside_wg, int nrhs, hipStream_t st) {
    const unsigned gx = (unsigned)(a.nchunks + nside_wg);
    const dim3 b(256);
    if (g_rec) {
        MLFusedPhase* p;
        if (rec_push(&p, PH_LEVEL, (UPPER ? 6 : 0) + MODE * 3 + (G2 == 64 ? 0 : (G2 == 16 ? 1 : 2)), (int)gx)) p->u.lv = a;
        return;
    }
    // blocks of right-hand sides: one workgroup per 64-row segment of a diagonal block and 4 right-hand sides (k_ml_level_blk;
    // C4 77 -> 62 ms; 8 right-hand sides per workgroup: 83 ms -- registers).  With whole blocks per workgroup the levels of few
    // blocks were better off in the chunk form (a lone workgroup per block took 30-90 us: 70 ms with every level in block form,
    // 65.5 ms with levels of >= 4000 rows only); in segments every level gains.
    // (read per multi-right-hand-side launch, not cached: tests switch forms inside one process; single-vector solves never get here)
    int blk_rhs = 0, blk_min = 0;
    if (nrhs >= 8) {
        const char* e1 = getenv("NEP_ML_BLK_RHS"); const char* e2 = getenv("NEP_ML_BLK_RHS_MIN");
        blk_rhs = e1 ? atoi(e1) : 4;                // 0: chunk form everywhere
        blk_min = e2 ? atoi(e2) : 0;                // rows of the level below which the chunk form is kept
    }
    const bool blk_ok = nrhs >= 8 && a.ident_row0 < 0 && (int64_t)a.nchunks * (256 / G2) >= blk_min;
    if (blk_ok && blk_rhs) {
        MLArgs a2 = a; a2.nside = nside_wg;
        const unsigned items8 = (unsigned)((a.nsegs + nside_wg + 7) / 8);
        if (blk_rhs == 4) hipLaunchKernelGGL((k_ml_level_blk<UPPER, 4, MODE>), dim3(items8 * 8 * (unsigned)((nrhs + 3) / 4)), b, 0, st, a2);
        else hipLaunchKernelGGL((k_ml_level_blk<UPPER, 8, MODE>), dim3(items8 * 8 * (unsigned)((nrhs + 7) / 8)), b, 0, st, a2);
    }
    else if (nrhs >= 8) hipLaunchKernelGGL((k_ml_level<UPPER, 8, MODE, G2>), dim3(gx, (nrhs + 7) / 8), b, 0, st, a);
    else if (nrhs >= 2) hipLaunchKernelGGL((k_ml_level<UPPER, 4, MODE, G2>), dim3(gx, (nrhs + 3) / 4), b, 0, st, a);
    else hipLaunchKernelGGL((k_ml_level<UPPER, 1, MODE, G2>), dim3(gx, 1), b, 0, st, a);
}
template <bool UPPER, int MODE>
static void launch_level(const MLArgs& a, int ch, int nside_wg, int nrhs, hipStream_t st) {
    if (ch == 4) launch_level_g<UPPER, MODE, 64>(a, nside_wg, nrhs, st);
    else if (ch == 16) launch_level_g<UPPER, MODE, 16>(a, nside_wg, nrhs, st);
    else launch_level_g<UPPER, MODE, 8>(a, nside_wg, nrhs, st);
}

static void launch_coupling(int lanes, int r0, int r1, const int32_t* cp, const int32_t* ci, const cplx* cx, const cplx* src,
                            int64_t ldsrc, const cplx* xin, int64_t ldxin, cplx* tmp, int64_t ldtmp, int nrhs, hipStream_t st,
                            int col_lo = 0, int col_hi = 0x7fffffff, int ident_row0 = -1) {
    const int rows = r1 - r0;
    MLCplArgs C;
    C.r0 = r0; C.r1 = r1; C.cp = cp; C.ci = ci; C.cx = cx; C.src = src; C.ldsrc = ldsrc; C.xin = xin; C.ldxin = ldxin; C.tmp = tmp;
    C.ldtmp = ldtmp; C.nrhs = nrhs; C.col_lo = col_lo; C.col_hi = col_hi; C.ident_row0 = ident_row0;
    if (lanes != 256 && lanes != 64) lanes = 8;
    const unsigned gx = (unsigned)(lanes == 256 ? rows : (rows + 256 / lanes - 1) / (256 / lanes));
    if (g_rec) {
        MLFusedPhase* p;
        if (rec_push(&p, PH_COUPLING, lanes, (int)gx)) p->u.cp = C;
        return;
    }
#define CPL(G_, RB_) hipLaunchKernelGGL((k_ml_coupling<G_, RB_>), dim3(gx, (nrhs + RB_ - 1) / RB_), dim3(256), 0, st, C)
#define CPLG(G_) do { if (nrhs >= 8) CPL(G_, 8); else if (nrhs >= 2) CPL(G_, 4); else CPL(G_, 1); } while (0)
    if (lanes == 256) CPLG(256); else if (lanes == 64) CPLG(64); else CPLG(8);
#undef CPLG
#undef CPL
}

struct MLSolveCtx {
    MLFactor* F; int nrhs; cplx *bw, *y, *x, *tmp; int64_t ld;     // work vectors (ld = n; apex build: T, pointers offset by -R0)
    const cplx* dB; int64_t ldb; cplx* dX; int64_t ldx; const cplx* dAdd; int64_t ldadd; double scale;
    int ident_row0 = -1;      // >= 0: right-hand sides are unit vectors (apex build)
    int col_lo = 0;           // coupling columns below it are skipped (apex build)
};

// L level l.  first = the level reads the caller's B through the input permutation (and copies the rest of B to bw)
static int run_L(const MLSolveCtx& c, int l, bool first, hipStream_t st, int* launches) {
    const MLSym* S = c.F->sym;
    const MLFacSym& f = S->L;
    const int64_t n = S->n;
    const cplx* cx = c.F->d_vals;
    const cplx* src = first ? c.dB : c.bw;
    const int64_t ldsrc = first ? c.ldb : c.ld;
    const bool split = f.split[l] != 0;          // level 0 has no coupling, so `first` is never split
    if (split) {
        launch_coupling(f.cpl_lanes[l], S->lev_row[l], S->lev_row[l + 1], f.d_cp, f.d_ci, cx, src, ldsrc, c.y, c.ld, c.tmp, c.ld,
                        c.nrhs, st, c.col_lo, 0x7fffffff, c.ident_row0);
        LAUNCHCHK(); if (launches) ++*launches;
    }
    MLArgs a; memset(&a, 0, sizeof(a));
    a.chunks = f.d_chunks + f.lev_chunk[l]; a.nchunks = f.lev_chunk[l + 1] - f.lev_chunk[l];
    a.segs = f.d_segs + f.lev_seg[l]; a.nsegs = f.lev_seg[l + 1] - f.lev_seg[l];
    a.cp = f.d_cp; a.ci = f.d_ci; a.cx = cx; a.ix = c.F->d_ixL; a.has_coupling = f.lev_coup[l] > 0 ? 1 : 0;
    a.src = src; a.ldsrc = ldsrc; a.gat = first ? S->d_pin : nullptr; a.rs = first ? c.F->d_rscale : nullptr;
    a.ident_row0 = c.ident_row0; a.col_lo = c.col_lo;
    a.xin = c.y; a.ldxin = c.ld; a.xout = c.y; a.ldxout = c.ld; a.tmp = c.tmp; a.ldtmp = c.ld;
    a.nrhs = c.nrhs;
    int nside = 0;
    if (first && S->nlev > 1) {
        a.side_lo = S->lev_row[1]; a.side_hi = n; a.side_dst = c.bw; a.ldside = c.ld;
        nside = (int)((a.side_hi - a.side_lo + 255) / 256);
    }
    if (split) launch_level<false, 1>(a, f.lev_ch[l], nside, c.nrhs, st); else launch_level<false, 0>(a, f.lev_ch[l], nside, c.nrhs, st);
    LAUNCHCHK(); if (launches) ++*launches;
    return NEP_OK;
}

static int run_U_coupling(const MLSolveCtx& c, int l, hipStream_t st, int* launches) {
    const MLSym* S = c.F->sym;
    const MLFacSym& f = S->U;
    if (!f.split[l]) return NEP_OK;
    const cplx* cx = c.F->d_vals + (S->L.ncoup + S->L.nin);
    launch_coupling(f.cpl_lanes[l], S->lev_row[l], S->lev_row[l + 1], f.d_cp, f.d_ci, cx, c.y, c.ld, c.x, c.ld, c.tmp, c.ld, c.nrhs, st);
    LAUNCHCHK(); if (launches) ++*launches;
    return NEP_OK;
}
// U level l; `final` = the launch that also produces the caller's X
static int run_U_level(const MLSolveCtx& c, int l, bool final, hipStream_t st, int* launches) {
    const MLSym* S = c.F->sym;
    const MLFacSym& f = S->U;
    const int64_t n = S->n;
    const cplx* cx = c.F->d_vals + (S->L.ncoup + S->L.nin);
    MLArgs a; memset(&a, 0, sizeof(a));
    a.chunks = f.d_chunks + f.lev_chunk[l]; a.nchunks = f.lev_chunk[l + 1] - f.lev_chunk[l];
    a.segs = f.d_segs + f.lev_seg[l]; a.nsegs = f.lev_seg[l + 1] - f.lev_seg[l];
    a.cp = f.d_cp; a.ci = f.d_ci; a.cx = cx; a.ix = c.F->d_ixU; a.has_coupling = f.lev_coup[l] > 0 ? 1 : 0;
    a.src = c.y; a.ldsrc = c.ld; a.xin = c.x; a.ldxin = c.ld; a.xout = c.x; a.ldxout = c.ld; a.tmp = c.tmp; a.ldtmp = c.ld;
    a.ident_row0 = -1; a.col_lo = 0;
    a.nrhs = c.nrhs;
    int nside = 0;
    if (final) {
        a.outX = c.dX; a.ldX = c.ldx; a.pout = S->d_pout; a.scale = c.scale; a.add = c.dAdd; a.ldadd = c.ldadd;
        if (S->nlev > 1) {
            a.side_lo = S->lev_row[1]; a.side_hi = n; a.side_src = c.x; a.ldsidesrc = c.ld;
            nside = (int)((a.side_hi - a.side_lo + 255) / 256);
        }
    }
    if (f.split[l]) launch_level<true, 1>(a, f.lev_ch[l], nside, c.nrhs, st); else launch_level<true, 0>(a, f.lev_ch[l], nside, c.nrhs, st);
    LAUNCHCHK(); if (launches) ++*launches;
    return NEP_OK;
}

// apex of one solve: t = b_T - L[T, <R0] y (coupling restricted to the columns below the apex), x_T = S^{-1} t
static int run_apex(const MLSolveCtx& c, hipStream_t st, int* launches) {
    MLFactor* F = c.F;
    const MLSym* S = F->sym;
    const int la = F->apex_la;
    const int R0 = S->lev_row[la], T = (int)(S->n - R0);
    const MLFacSym& f = S->L;
    int lanes = 8;
    { int64_t nz = 0; for (int l = la; l < S->nlev; ++l) nz += f.lev_coup[l]; const double avg = nz / (double)T; lanes = avg > 2048.0 ? 256 : (avg > 24.0 ? 64 : 8); }
    launch_coupling(lanes, R0, (int)S->n, f.d_cp, f.d_ci, F->d_vals, c.bw, c.ld, c.y, c.ld, c.tmp, c.ld, c.nrhs, st, 0, R0, -1);
    LAUNCHCHK();
    MLApexArgs P;
    P.T = T; P.R0 = R0; P.Sinv = (const cplx*)F->d_Sinv; P.t = (const cplx*)c.tmp; P.ldt = c.ld; P.x = c.x; P.ldx = c.ld; P.nrhs = c.nrhs;
    if (g_rec) {
        MLFusedPhase* p;
        if (rec_push(&p, PH_APEX, 0, (T + 3) / 4)) p->u.ax = P;
        if (launches) *launches += 2;
        return NEP_OK;
    }
#define APEX_GEMV(RB_) hipLaunchKernelGGL((k_apex_gemv<RB_>), dim3((unsigned)((T + 3) / 4), (c.nrhs + RB_ - 1) / RB_), dim3(256), 0, st, P)
    const int gemv1 = getenv("NEP_ML_GEMV1") ? atoi(getenv("NEP_ML_GEMV1")) : 1;
    if (c.nrhs >= 8) APEX_GEMV(8); else if (c.nrhs >= 2) APEX_GEMV(4);
    else if (gemv1 && T <= ML_APEX_TMAX) hipLaunchKernelGGL(k_apex_gemv1, dim3((unsigned)((T + 7) / 8)), dim3(256), 0, st, P);
    else APEX_GEMV(1);
#undef APEX_GEMV
    LAUNCHCHK();
    if (launches) *launches += 2;
    return NEP_OK;
}

// everything between the first (L level 0) and the last (U level 0) launch: fixed buffers only -> one hipGraph
// the last launch of a single-vector solve takes U level 0's coupling product inside (k_ml_u0_fused) when that level runs the
// product as its own launch today
static bool u0_fused(const MLFactor* F, int nrhs) {
    // MEASURED: no gain on gun (4 launches 38.4 us against 36.7 us for 5 launches with the same apex kernel): the fused kernel
    // strings the dependent-load chains of the two kernels it replaces together (block range -> row pointers -> entries ->
    // gathers -> LDS -> inverse row offsets -> rows -> permutation -> store), and a solve at this size is bound by those
    // chains, not by its launches.  Kept opt-in (NEP_ML_U0FUSE=1) and tested.
    const char* e = getenv("NEP_ML_U0FUSE");            // (read per solve: tests switch it)
    const int on = e ? atoi(e) : 0;
    const MLSym* S = F->sym;
    if (!on || nrhs != 1 || g_rec || S->nlev < 2 || !S->U.split[0] || S->max_block > ML_BMAX) return false;
    // two 1024-thread workgroups per block pay off for blocks of ~100 rows and more (gun: 51 blocks of 169 rows on average);
    // the 64-row blocks of a million-row factor stay with the chunked kernels
    const int nb0 = S->lev_blk[1] - S->lev_blk[0];
    return nb0 > 0 && S->lev_row[1] / nb0 >= 96;
}
static int run_U0_fused(const MLSolveCtx& c, hipStream_t st, int* launches) {
    const MLSym* S = c.F->sym;
    const MLFacSym& f = S->U;
    MLU0Args a;
    a.blk_se = S->d_blk_se; a.blk0 = S->lev_blk[0]; a.nblk = S->lev_blk[1] - S->lev_blk[0];
    a.cp = f.d_cp; a.ci = f.d_ci; a.cx = c.F->d_vals + (S->L.ncoup + S->L.nin);
    a.ix = c.F->d_ixU; a.ip = f.d_ip; a.y = c.y; a.x = c.x;
    a.outX = c.dX; a.pout = S->d_pout; a.scale = c.scale; a.add = c.dAdd;
    a.side_lo = S->lev_row[1]; a.side_hi = S->n;
    const unsigned gx = (unsigned)(2 * a.nblk + (a.side_hi - a.side_lo + 1023) / 1024);
    hipLaunchKernelGGL(k_ml_u0_fused, dim3(gx), dim3(1024), 0, st, a);
    LAUNCHCHK(); if (launches) ++*launches;
    return NEP_OK;
}

static int ml_middle(const MLSolveCtx& c, hipStream_t st, int* launches) {
    const MLSym* S = c.F->sym;
    const int top = eff_apex(c.F) > 0 ? eff_apex(c.F) : S->nlev;      // levels [top, nlev) are handled by the apex
    int rc;
    for (int l = 1; l < top; ++l) if ((rc = run_L(c, l, false, st, launches))) return rc;
    if (eff_apex(c.F) > 0 && (rc = run_apex(c, st, launches))) return rc;
    for (int l = top - 1; l >= 1; --l) {
        if ((rc = run_U_coupling(c, l, st, launches))) return rc;
        if ((rc = run_U_level(c, l, false, st, launches))) return rc;
    }
    if (u0_fused(c.F, c.nrhs)) return NEP_OK;
    return run_U_coupling(c, 0, st, launches);
}
static int ml_middle_count(const MLFactor* F, int nrhs) {
    const MLSym* S = F->sym;
    const int top = eff_apex(F) > 0 ? eff_apex(F) : S->nlev;
    int nmid = 0;
    for (int l = 1; l < top; ++l) nmid += 2 + S->L.split[l] + S->U.split[l];
    if (eff_apex(F) > 0) nmid += 2;
    return nmid + (u0_fused(F, nrhs) ? 0 : S->U.split[0]);
}

// numeric build of the apex: S^{-1} e_j for all T unit vectors = the block solve of levels >= la restricted to the apex
// rows, T right-hand sides at once (work vectors T x T, addressed with the global row index through pointers offset by -R0)
static int ml_build_apex(MLFactor* F, hipStream_t bst) {
    MLSym* S = F->sym;
    const int la = F->apex_la;
    const int R0 = S->lev_row[la];
    const int64_t T = S->n - R0;
    int rc;
    if (!F->d_Sinv && (rc = nep_pool_alloc((void**)&F->d_Sinv, (size_t)T * T * sizeof(cplx)))) return rc;
    cplx* wk = nullptr;
    if ((rc = nep_pool_alloc((void**)&wk, (size_t)3 * T * T * sizeof(cplx)))) return rc;
    MLSolveCtx c;
    c.F = F; c.nrhs = (int)T; c.ld = T;
    c.y = wk - R0; c.x = wk + (size_t)T * T - R0; c.tmp = wk + (size_t)2 * T * T - R0; c.bw = nullptr;
    c.dB = nullptr; c.ldb = 0; c.dX = nullptr; c.ldx = 0; c.dAdd = nullptr; c.ldadd = 0; c.scale = 1.0;
    c.ident_row0 = R0; c.col_lo = R0;
    for (int l = la; l < S->nlev; ++l) if ((rc = run_L(c, l, false, bst, nullptr))) { nep_pool_free_on(wk, bst, true); return rc; }
    for (int l = S->nlev - 1; l >= la; --l) {
        if ((rc = run_U_coupling(c, l, bst, nullptr)) || (rc = run_U_level(c, l, false, bst, nullptr))) { nep_pool_free_on(wk, bst, true); return rc; }
    }
    hipLaunchKernelGGL(k_apex_transpose, dim3((unsigned)((T + 15) / 16), (unsigned)((T + 15) / 16)), dim3(256), 0, bst, (int)T,
                       (const cplx*)(wk + (size_t)T * T), F->d_Sinv);
    hipError_t e = hipGetLastError();
    nep_pool_free_on(wk, bst, true);
    if (e != hipSuccess) { nep_set_error("apex build failed: %s", hipGetErrorString(e)); return NEP_ERR_HIP; }
    return NEP_OK;
}

// ---- the apex inverse as DENSE block algebra (round 4) ---------------------------------------------------------------------------
// ml_build_apex above runs the sparse level kernels on T unit vectors, eight at a time: 3.4 ms of device-filling launches on gun
// (T = 1326) next to the first solves of the factor, whose few-microsecond kernels then take 25-80 us each -- measured on the headline
// call: 37.0 ms, 37.4 with the build finished before the first solve, 41.5 without an apex; the build costs ~3 ms either way.
// Here the apex rows of the factors are scattered into dense T x T blocks (coupling entries between apex levels; the explicit
// inverses of the diagonal blocks), and S^{-1} = inv(U_TT) inv(L_TT) is formed level by level with the library's own GEMM:
//   X[I_l, :] = DLinv_l (E_l - Lc[I_l, <a_l] X[<a_l, :]),     then  W[I_l, :] = DUinv_l (X[I_l, :] - Uc[I_l, >=b_l] W[>=b_l, :])
// one coupling product per level and one product per diagonal block.  The arithmetic differs from the substitution by rounding only.
__global__ __launch_bounds__(256) void k_apex_densify(int R0, int T, const int32_t* __restrict__ rowblk, const int32_t* __restrict__ blk_se,
                                                      const int32_t* __restrict__ cpL, const int32_t* __restrict__ ciL, const cplx* __restrict__ cxL,
                                                      const int32_t* __restrict__ cpU, const int32_t* __restrict__ ciU, const cplx* __restrict__ cxU,
                                                      const int64_t* __restrict__ ipL, const cplx* __restrict__ ixL,
                                                      const int64_t* __restrict__ ipU, const cplx* __restrict__ ixU,
                                                      cplx* __restrict__ Lc, cplx* __restrict__ Uc, cplx* __restrict__ DL, cplx* __restrict__ DU) {
    const int r = R0 + (int)blockIdx.x;                     // one workgroup per apex row (new numbering)
    const int64_t lr = r - R0;
    for (int p = cpL[r] + threadIdx.x; p < cpL[r + 1]; p += 256) { const int c = ciL[p]; if (c >= R0) Lc[lr + (int64_t)(c - R0) * T] = cxL[p]; }
    for (int p = cpU[r] + threadIdx.x; p < cpU[r + 1]; p += 256) { const int c = ciU[p]; if (c >= R0) Uc[lr + (int64_t)(c - R0) * T] = cxU[p]; }
    const int b = rowblk[r], s = blk_se[2 * b], e = blk_se[2 * b + 1];
    for (int t = threadIdx.x; t <= r - s; t += 256) DL[lr + (int64_t)(s + t - R0) * T] = ixL[ipL[r] + t];      // columns [s, r]
    for (int t = threadIdx.x; t < e - r; t += 256) DU[lr + (int64_t)(r + t - R0) * T] = ixU[ipU[r] + t];       // columns [r, e)
}
// Y[a + i, a + i] = 1 for the rows [a, b) of a level (the block is cleared by the caller)
__global__ void k_apex_ident(int a, int b, int T, cplx* __restrict__ Y) {
    const int i = a + blockIdx.x * blockDim.x + threadIdx.x;
    if (i < b) Y[i + (int64_t)i * T] = cmake(1.0, 0.0);
}
extern "C" int32_t nep_zgemm_ex(int32_t m, int32_t n, int32_t k, nep_cdouble alpha, const nep_cdouble* dA, int64_t lda,
                                const nep_cdouble* dB, int64_t ldb, nep_cdouble beta, nep_cdouble* dC, int64_t ldc,
                                const int32_t* d_krange, int32_t ksplit, nep_cdouble* dWork, nep_stream stream);

static int ml_build_apex_dense(MLFactor* F, hipStream_t bst) {
    MLSym* S = F->sym;
    const int la = F->apex_la;
    const int R0 = S->lev_row[la];
    const int T = (int)(S->n - R0);
    const int64_t TT = (int64_t)T * T;
    int rc;
    if (!F->d_Sinv && (rc = nep_pool_alloc((void**)&F->d_Sinv, (size_t)TT * sizeof(cplx)))) return rc;
    // K range per 64-row tile of every level's block-diagonal factor (once per pattern and apex level)
    static std::mutex kr_mu;                                  // (factors of one pattern may be built from several host threads)
    std::unique_lock<std::mutex> kr_lock(kr_mu);
    if (!S->d_apex_kr || S->apex_kr_la != la) {
        if (S->d_apex_kr) { nep_pool_free(S->d_apex_kr); S->d_apex_kr = nullptr; }
        std::vector<int32_t> kr; S->apex_kr_off.assign(S->nlev + 1, 0);
        for (int l = la; l < S->nlev; ++l) {
            S->apex_kr_off[l] = (int32_t)kr.size();
            const int a = S->lev_row[l], b = S->lev_row[l + 1];
            int k = S->lev_blk[l];
            for (int r0 = a; r0 < b; r0 += 64) {
                const int r1 = std::min(b, r0 + 64) - 1;
                while (S->h_blk_se[2 * k + 1] <= r0) ++k;
                int k1 = k;
                while (S->h_blk_se[2 * k1 + 1] <= r1) ++k1;
                kr.push_back(S->h_blk_se[2 * k] - a); kr.push_back(S->h_blk_se[2 * k1 + 1] - a);
            }
        }
        if ((rc = nep_pool_alloc((void**)&S->d_apex_kr, std::max<size_t>(kr.size(), 2) * sizeof(int32_t)))) return rc;
        HIPCHK(hipMemcpy(S->d_apex_kr, kr.data(), kr.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        S->apex_kr_la = la;
    }
    // the workspace belongs to the pattern; the lock is held until this build is enqueued, the event orders builds on different streams
    if (!S->d_apex_work || S->apex_work_T != T) {
        if (S->d_apex_work) { if (S->apex_work_ev) (void)hipEventSynchronize(S->apex_work_ev); nep_pool_free(S->d_apex_work); S->d_apex_work = nullptr; }
        if ((rc = nep_pool_alloc((void**)&S->d_apex_work, (size_t)9 * TT * sizeof(cplx)))) return rc;
        S->apex_work_T = T; S->apex_work_used = false;
    }
    if (!S->apex_work_ev) HIPCHK(hipEventCreateWithFlags(&S->apex_work_ev, hipEventDisableTiming));
    if (S->apex_work_used) HIPCHK(hipStreamWaitEvent(bst, S->apex_work_ev, 0));
    cplx* wk = S->d_apex_work;
    cplx *Lc = wk, *Uc = wk + TT, *DL = wk + 2 * TT, *DU = wk + 3 * TT, *X = wk + 4 * TT, *Y = wk + 5 * TT, *W = wk + 6 * TT, *P = wk + 7 * TT;
    // split-K factor of a coupling product: enough workgroups for the chip, partial slices within the 2 T^2 workspace
    auto ksplit_of = [&](int m, int n, int k) {
        const int64_t tiles = (int64_t)((m + 63) / 64) * ((n + 63) / 64);
        int ks = (int)std::min<int64_t>(8, (640 + tiles - 1) / tiles);
        while (ks > 1 && (int64_t)ks * m * n > 2 * TT) --ks;
        if (k < 128 * ks) ks = std::max(1, k / 128);
        return ks;
    };
    auto fail = [&](int code) { (void)hipEventRecord(S->apex_work_ev, bst); S->apex_work_used = true; return code; };
    if (hipMemsetAsync(wk, 0, (size_t)6 * TT * sizeof(cplx), bst) != hipSuccess) return fail(NEP_ERR_HIP);
    const cplx* cxL = F->d_vals;
    const cplx* cxU = F->d_vals + (S->L.ncoup + S->L.nin);
    hipLaunchKernelGGL(k_apex_densify, dim3((unsigned)T), dim3(256), 0, bst, R0, T, (const int32_t*)S->d_rowblk, (const int32_t*)S->d_blk_se,
                       (const int32_t*)S->L.d_cp, (const int32_t*)S->L.d_ci, cxL, (const int32_t*)S->U.d_cp, (const int32_t*)S->U.d_ci, cxU,
                       (const int64_t*)S->L.d_ip, (const cplx*)F->d_ixL, (const int64_t*)S->U.d_ip, (const cplx*)F->d_ixU, Lc, Uc, DL, DU);
    if (hipGetLastError() != hipSuccess) return fail(NEP_ERR_HIP);
    const nep_cdouble one{1.0, 0.0}, mone{-1.0, 0.0}, zero{0.0, 0.0};
    nep_stream ns = (nep_stream)bst;
#define ZGS(m_, n_, k_, al_, A_, B_, be_, C_) do { if ((rc = nep_zgemm_ex(m_, n_, k_, al_, (const nep_cdouble*)(A_), (int64_t)T, (const nep_cdouble*)(B_), (int64_t)T, be_, \
        (nep_cdouble*)(C_), (int64_t)T, nullptr, ksplit_of(m_, n_, k_), (nep_cdouble*)P, ns))) return fail(rc); } while (0)
#define ZGD(l_, m_, n_, A_, B_, C_) do { if ((rc = nep_zgemm_ex(m_, n_, m_, one, (const nep_cdouble*)(A_), (int64_t)T, (const nep_cdouble*)(B_), (int64_t)T, zero, \
        (nep_cdouble*)(C_), (int64_t)T, (const int32_t*)S->d_apex_kr + S->apex_kr_off[l_], 1, nullptr, ns))) return fail(rc); } while (0)
    // ---- X = inv(L_TT): rows of level l only have columns < b_l
    for (int l = la; l < S->nlev; ++l) {
        const int a = S->lev_row[l] - R0, b = S->lev_row[l + 1] - R0, ml = b - a;
        if (ml <= 0) continue;
        hipLaunchKernelGGL(k_apex_ident, dim3((unsigned)((ml + 255) / 256)), dim3(256), 0, bst, a, b, T, Y);
        if (a > 0) ZGS(ml, a, a, mone, Lc + a, X, one, Y + a);
        ZGD(l, ml, b, DL + a + (int64_t)a * T, Y + a, X + a);            // all diagonal blocks of the level in one product
    }
    // ---- W = inv(U_TT) X, from the last level up
    for (int l = S->nlev - 1; l >= la; --l) {
        const int a = S->lev_row[l] - R0, b = S->lev_row[l + 1] - R0, ml = b - a;
        if (ml <= 0) continue;
        if (b < T) ZGS(ml, T, T - b, mone, Uc + a + (int64_t)b * T, W + b, one, X + a);
        ZGD(l, ml, T, DU + a + (int64_t)a * T, X + a, W + a);
    }
#undef ZGS
#undef ZGD
    hipLaunchKernelGGL(k_apex_transpose, dim3((unsigned)((T + 15) / 16), (unsigned)((T + 15) / 16)), dim3(256), 0, bst, T, (const cplx*)W, F->d_Sinv);
    hipError_t e = hipGetLastError();
    (void)hipEventRecord(S->apex_work_ev, bst); S->apex_work_used = true;
    if (e != hipSuccess) { nep_set_error("dense apex build failed: %s", hipGetErrorString(e)); return NEP_ERR_HIP; }
    return NEP_OK;
}

// which levels to merge into the dense apex: launches saved (about 4.5 us each) against 16 T^2 bytes of extra streaming
static int choose_apex(const MLSym* S, int expected_solves) {
    if (const char* e = getenv("NEP_ML_APEX")) { const int v = atoi(e); return (v >= 1 && v < S->nlev && S->n - S->lev_row[v] <= 4096) ? v : 0; }
    if (expected_solves < 8 || S->nlev < 2) return 0;
    int best = 0; double bestgain = 2.0;              // at least 2 us per solve
    for (int la = 1; la < S->nlev; ++la) {
        const double T = (double)(S->n - S->lev_row[la]);
        if (T > 2048.0) continue;
        int saved = -2;
        double bytes_now = 0.0;
        for (int l = la; l < S->nlev; ++l) {
            saved += 2 + S->L.split[l] + S->U.split[l];
            bytes_now += 20.0 * (S->L.lev_coup[l] + S->U.lev_coup[l]);
        }
        const double gain = 4.5 * saved - (16.0 * T * T - bytes_now) / 3.5e6;      // us
        if (gain > bestgain) { bestgain = gain; best = la; }
    }
    return best;
}

int ml_solve(MLFactor* F, int nrhs, const nep_cdouble* dB, int64_t ldb, const nep_cdouble* dAdd, int64_t ldadd,
             nep_cdouble* dX, int64_t ldx, double scale, hipStream_t st) {
    MLSym* S = F->sym;
    const int64_t n = S->n;
    // the previous users of these buffers ran on F->last; a different stream must queue behind them
    if (F->used && F->last != st) {
        hipEvent_t ev; HIPCHK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        HIPCHK(hipEventRecord(ev, F->last)); HIPCHK(hipStreamWaitEvent(st, ev, 0)); (void)hipEventDestroy(ev);
    }
    if (!F->synced_valid || F->synced != st) { HIPCHK(hipStreamWaitEvent(st, F->ready, 0)); F->synced = st; F->synced_valid = true; }
    // Which solve of a factor first uses the dense apex is FIXED (the two paths round differently: a switch point that depended on
    // host / device timing made iar iterates differ from run to run in the last bits): solve number NEP_ML_APEX_AT
    // (default 6) waits for the build on the stream -- the host never blocks -- and every later solve takes the apex.  Measured per
    // iar call: 38.5 ms at 4 ... 8, 38.9-39.8 at 1, 39.5-40.2 at 12, 41.4 at 40: a short wait while the build has the device to
    // itself beats both more solves through the apex levels and an immediate wait.  NEP_ML_APEX_AT=0: as soon as a query finds the
    // build finished (round 2).
    static const int apex_at = getenv("NEP_ML_APEX_AT") ? atoi(getenv("NEP_ML_APEX_AT")) : 6;
    if (F->apex_la > 0 && !F->apex_live && F->apex_ev) {
        bool take = false;
        if (apex_at > 0) take = F->solves_since_numeric >= apex_at;
        else if (hipEventQuery(F->apex_ev) == hipSuccess) take = true;
        else (void)hipGetLastError();                               // hipErrorNotReady of the query
        if (take) {
            HIPCHK(hipStreamWaitEvent(st, F->apex_ev, 0));
            F->apex_live = true;
            if (F->graph) { (void)hipGraphExecDestroy(F->graph); F->graph = nullptr; }
        }
    }
    ++F->solves_since_numeric;
    int rc;
    const size_t need = (size_t)4 * n * nrhs * sizeof(cplx);
    if (F->work.cap < need) {          // the old block may still be in use by solves in flight on F->last
        if (F->work.dptr) { nep_pool_free_on(F->work.dptr, F->last, F->used); F->work.dptr = nullptr; F->work.cap = 0; }
        if ((rc = nep_pool_alloc(&F->work.dptr, need))) return rc;
        F->work.cap = need;
    }
    MLSolveCtx c;
    c.F = F; c.nrhs = nrhs; c.ld = n;
    c.bw = (cplx*)F->work.dptr; c.y = c.bw + (size_t)n * nrhs; c.x = c.y + (size_t)n * nrhs; c.tmp = c.x + (size_t)n * nrhs;
    c.dB = (const cplx*)dB; c.ldb = ldb; c.dX = (cplx*)dX; c.ldx = ldx; c.dAdd = (const cplx*)dAdd; c.ldadd = ldadd; c.scale = scale;
    int launches = 0;
    if (F->h_ferr && *F->h_ferr) {
        nep_set_error("block-schedule solve: a phase wait of the single-launch kernel hit its bound (earlier solve incomplete)");
        return NEP_ERR_HIP;
    }
    // MEASURED NEGATIVE (kept opt-in, NEP_ML_FUSE=1, so that it can be reproduced): on the gun factors the one-launch form
    // takes 770 us per solve against 38 us for the five launches (470 us without the ticket atomics, NEP_ML_FUSE_MODE=1).
    // The 8 XCDs have no common L2, so every agent-scope atomic on the phase counters is performed memory-side; ~3000
    // same-address increments per solve serialise at ~150 ns each, and the release/acquire fences are the same L2
    // write-back/invalidate a kernel boundary performs.  A kernel boundary (~2.5 us on this part) IS the cheap grid barrier.
    const char* fuse_env = nrhs == 1 ? getenv("NEP_ML_FUSE") : nullptr;
    if (nrhs == 1 && !F->fuse_off && fuse_env && atoi(fuse_env)) {
        // record the launches of the multi-launch path as phases and issue them as one kernel
        MLRecorder rec; rec.a.nph = 0; rec.a.total = 0; rec.overflow = false;
        g_rec = &rec;
        rc = run_L(c, 0, true, st, nullptr);
        if (!rc) rc = ml_middle(c, st, nullptr);
        if (!rc) rc = run_U_level(c, 0, true, st, nullptr);
        g_rec = nullptr;
        if (rc) return rc;
        if (!rec.overflow && rec.a.nph >= 2) {
            if (!F->d_fctr) {
                if ((rc = nep_pool_alloc((void**)&F->d_fctr, 16 * sizeof(unsigned long long)))) return rc;
                HIPCHK(hipMemsetAsync(F->d_fctr, 0, 16 * sizeof(unsigned long long), st));
                HIPCHK(hipHostMalloc((void**)&F->h_ferr, 64, hipHostMallocMapped));
                HIPCHK(hipHostGetDevicePointer((void**)&F->d_ferr, F->h_ferr, 0));
                *F->h_ferr = 0; F->fuse_epoch = 0;
            }
            rec.a.mode = getenv("NEP_ML_FUSE_MODE") ? atoi(getenv("NEP_ML_FUSE_MODE")) : 0; rec.a.pad0 = 0;
            rec.a.ctr = F->d_fctr; rec.a.epoch = F->fuse_epoch++; rec.a.err = F->d_ferr;
            hipLaunchKernelGGL(k_ml_fused, dim3((unsigned)rec.a.total), dim3(256), 0, st, rec.a);
            LAUNCHCHK();
            F->launches = 1;
            F->last = st; F->used = true;
            return NEP_OK;
        }
        F->fuse_off = 1;          // too many phases for one kernel-argument block: multi-launch path from now on
    }
    if ((rc = run_L(c, 0, true, st, &launches))) return rc;
    const int nmid = ml_middle_count(F, nrhs);
    bool graphed = false;
    if (F->use_graph && nmid >= 4 && (F->apex_la == 0 || F->apex_live) && !getenv("NEP_NO_GRAPH")) {   // no capture for the few solves before the apex
        if (!F->graph || F->graph_nrhs != nrhs || F->graph_work != F->work.dptr) {
            if (F->graph) { (void)hipGraphExecDestroy(F->graph); F->graph = nullptr; }
            if (!F->cap_stream) HIPCHK(hipStreamCreateWithFlags(&F->cap_stream, hipStreamNonBlocking));
            hipGraph_t g = nullptr;
            hipError_t e = hipStreamBeginCapture(F->cap_stream, hipStreamCaptureModeThreadLocal);
            if (e == hipSuccess) {
                const int rcs = ml_middle(c, F->cap_stream, nullptr);
                e = hipStreamEndCapture(F->cap_stream, &g);
                if (rcs == NEP_OK && e == hipSuccess && g) e = hipGraphInstantiate(&F->graph, g, nullptr, nullptr, 0);
                else if (e == hipSuccess) e = hipErrorUnknown;
                if (g) (void)hipGraphDestroy(g);
            }
            if (e != hipSuccess || !F->graph) { (void)hipGetLastError(); F->graph = nullptr; F->use_graph = 0; }
            else { F->graph_nrhs = nrhs; F->graph_work = F->work.dptr; }
        }
        if (F->graph) { HIPCHK(hipGraphLaunch(F->graph, st)); launches += nmid; graphed = true; }
    }
    if (!graphed && (rc = ml_middle(c, st, &launches))) return rc;
    if (u0_fused(F, nrhs)) { if ((rc = run_U0_fused(c, st, &launches))) return rc; }
    else if ((rc = run_U_level(c, 0, true, st, &launches))) return rc;
    F->launches = launches;
    F->last = st; F->used = true;
    return NEP_OK;
}
