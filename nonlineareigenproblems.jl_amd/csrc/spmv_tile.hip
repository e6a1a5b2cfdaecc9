// libnepmi355: K1 (compute_Mlincomb, src/NEPTypes.jl:972-1011 / 1130-1160) as ONE launch on footprint tiles (gfx950).
//
//   z = sum_t A_t (V c_t)                     V n x k (column-major), c_t = C[:, t]
//
// The two-launch form (k_vc: W = V C, n x m_t through HBM; then the SpMV over the stacked matrix gathers W) pays a round trip
// of W and a kernel boundary; folding the coefficient product into the SpMV per ENTRY instead (k_spmv_sell_kfused) turns every
// matrix entry into k gathers that go to L2.  Here the matrix is cut into blocks of rows whose COLUMN FOOTPRINT (the distinct
// columns its entries touch) fits in LDS:
//
//   phase 1  for every footprint column c of the block:  W[t][c] = sum_j V[c, j] C[j, t]   -> LDS     (V read coalesced along c)
//   phase 2  for every row r of the block:               z[r] = sum_e val[e] W[term(e)][local(e)]      (entries from HBM, W from LDS)
//
// For matrices of a 2-D grid (the waveguide's 5-point stencils with row = x nz + z, the gun stand-in's 9-point stencil) the
// blocks are xp x zp PATCHES of the grid -- the dominant off-diagonal stride s is detected from the pattern -- so the
// footprint is the patch plus a one-point halo ((xp+2)(zp+2) columns for xp zp rows: 1.3x at 8 x 64) instead of the 2 s + R
// columns a block of R consecutive rows touches.  The halo columns are read by the neighbouring blocks as well; blocks are
// dealt to the XCDs in contiguous ranges, so those re-reads are L2 hits and the HBM traffic of the launch is the algorithmic
// minimum: matrix + V + z.  Entries carry a 16-bit (term, local column) index instead of the 32-bit (term, column) of the
// stacked CSR: 10 bytes per non-zero of a real matrix instead of 12.
//
// Nothing here depends on the grid guess being right: the footprint of every block is computed from its entries, blocks whose
// footprint does not fit are split, and a matrix with a row that does not fit at all simply gets no tiles (the caller keeps
// the two-launch path).
#include "common.h"
#include <vector>
#include <algorithm>
#include <cmath>

struct TileDesc {
    int32_t r0, stride, zp, nrows;     // local row l -> global row r0 + (l / zp) * stride + l % zp,  l < nrows
    int32_t fp_off, fp_cnt;            // footprint slots [fp_off, fp_off + fp_cnt)
    int32_t ent_off64;                 // first entry / 64
    int32_t wrb;                       // width (entries per row, padded) | rb << 16 (rows padded to a multiple of 16)
};

struct NepTiles {
    int64_t n = 0; int mt = 0; int valbytes = 8;
    int nblk = 0, fcap = 0, lbits = 13, stride = 0, xp = 0, zp = 0;
    int wmax = 0, rbmax = 0;        // widest row / largest padded row count over the blocks
    // term-slotted entries (matrices whose per-term row lengths add up to at most 8: the waveguide's 5 + 2 + 1): entry slot j of EVERY
    // row holds an entry of term (slot_terms >> 4 j) & 15 or a zero -- the term of an entry is then uniform over a wavefront
    int slotted = 0; uint32_t slot_terms = 0;
    int64_t nent = 0, nfp = 0;
    TileDesc* d_desc = nullptr;
    uint32_t* d_fp = nullptr;
    uint16_t* d_eidx = nullptr;
    void* d_eval = nullptr;
};

#define TILE_OWN 0x80000000u

__device__ __forceinline__ int tile_block(int swz) {
    const int b = blockIdx.x, nb = gridDim.x;
    if (!swz) return b;
    const int x = b & 7, i = b >> 3, per = nb >> 3, rem = nb & 7;
    return x * per + (x < rem ? x : rem) + i;
}

typedef double d2t __attribute__((ext_vector_type(2)));
template <bool NT> __device__ __forceinline__ double tload(const double* p) { return NT ? __builtin_nontemporal_load(p) : *p; }
template <bool NT> __device__ __forceinline__ cplx tload(const cplx* p) {
    if (NT) { const d2t v = __builtin_nontemporal_load((const d2t*)p); return cmake(v.x, v.y); }
    return *p;
}

// MTC = min(mt, 4) accumulators per footprint column; mt in 5..8 takes two passes over V (L1/L2 hits).
// blockDim.x = 256 (large matrices: several workgroups per CU overlap their phases) or 512 / 1024 (small matrices with many
// columns: the k loads of a footprint column are split over NG thread groups so that every load of a block is in flight at
// once -- at gun size the kernel is bound by dependent-load round trips, not by bytes).
// PF: the entries of the thread's rows (phase 2) are fetched into registers BEFORE the coefficient phase and its barrier
// (blocks of at most 2 blockDim rows x 8 entries: the waveguide stencils), so phase 2 only reads LDS.
template <typename VT, int MTC, bool NT, bool PF>
__global__ __launch_bounds__(PF ? 256 : 1024) void k_tile_mlincomb(const TileDesc* __restrict__ desc, const uint32_t* __restrict__ fp,
                                                        const uint16_t* __restrict__ eidx, const VT* __restrict__ eval,
                                                        const cplx* __restrict__ V, int64_t ldv, int k,
                                                        const cplx* __restrict__ C, int64_t ldc, int mt, int fcap, int lbits,
                                                        cplx* __restrict__ z, cplx* __restrict__ shift_dst, int swz, int split) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cplx* W = (cplx*)smem;                       // [mt][fcap]
    cplx* cs = W + (size_t)mt * fcap;            // [mt][k]
    cplx* part = cs + (size_t)mt * k;            // [NG][MTC][Fpad], NG * Fpad <= blockDim (split launches only)
    const int tid = threadIdx.x, nthr = blockDim.x;
    const TileDesc d = desc[tile_block(swz)];
    const int width = d.wrb & 0xffff, rb = d.wrb >> 16;
    const uint16_t* __restrict__ ib = eidx + (int64_t)d.ent_off64 * 64;
    const VT* __restrict__ vb = eval + (int64_t)d.ent_off64 * 64;
    // ---- phase-2 operands fetched ahead (independent of everything below)
    uint16_t pidx[2][8]; VT pval[2][8];
    const bool pf = PF && width <= 8 && rb <= 2 * nthr;
    if (PF && pf) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int l = tid + q * nthr;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const bool on = l < rb && j < width;
                const int64_t e = on ? (int64_t)j * rb + l : 0;
                pidx[q][j] = NT ? __builtin_nontemporal_load(ib + e) : ib[e];
                pval[q][j] = tload<NT>(vb + e);
                if (!on) { pidx[q][j] = 0; if constexpr (sizeof(VT) == 8) pval[q][j] = 0.0; else pval[q][j] = cmake(0.0, 0.0); }
            }
        }
    }
    for (int i = tid; i < mt * k; i += nthr) cs[i] = C[(i % k) + (int64_t)(i / k) * ldc];
    __syncthreads();
    const int F = d.fp_cnt;
    const uint32_t* __restrict__ fpb = fp + d.fp_off;
    const int Fpad = (F + 15) & ~15;
    int NG = 1;
    if (split) { NG = nthr / Fpad; if (NG > k / 2) NG = k / 2; if (NG < 1) NG = 1; }
    if (NG == 1) {
        // (the thread's footprint list entries are fetched up front: list entry -> V round trips of one trip were waited for before the
        // next trip's list entry was loaded)
        uint32_t raws[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) raws[u] = tid + u * nthr < F ? fpb[tid + u * nthr] : 0u;
        int trip = 0;
        for (int f = tid; f < F; f += nthr, ++trip) {
            const uint32_t raw = trip < 4 ? (trip == 0 ? raws[0] : trip == 1 ? raws[1] : trip == 2 ? raws[2] : raws[3]) : fpb[f];
            const int64_t c = raw & NEP_COL_MASK;
            const bool own = (raw & TILE_OWN) != 0 && shift_dst != nullptr;
            const cplx* vp = V + c;
            for (int i0 = 0; i0 < mt; i0 += MTC) {
                cplx acc[MTC];
#pragma unroll
                for (int i = 0; i < MTC; ++i) acc[i] = cmake(0.0, 0.0);
                const cplx* cp = cs + (size_t)i0 * k;
#pragma unroll 8
                for (int j = 0; j < k; ++j) {
                    const cplx v = vp[(int64_t)j * ldv];
                    if (own && i0 == 0) {               // iar: block j+1 of the next basis column = block j / (j+1) (method_iar.jl:97-98)
                        const double sc = 1.0 / (double)(j + 1);
                        shift_dst[c + (int64_t)j * ldv] = cmake(v.x * sc, v.y * sc);
                    }
#pragma unroll
                    for (int i = 0; i < MTC; ++i)
                        if (i0 + i < mt) cfma(acc[i], v, cp[i * k + j]);
                }
#pragma unroll
                for (int i = 0; i < MTC; ++i)
                    if (i0 + i < mt) W[(size_t)(i0 + i) * fcap + f] = acc[i];
            }
        }
    } else {
        const int g = tid / Fpad, f = tid - g * Fpad;
        const bool live = g < NG && f < F;
        const uint32_t raw = live ? fpb[f] : 0u;
        const int64_t c = raw & NEP_COL_MASK;
        const bool own = live && (raw & TILE_OWN) != 0 && shift_dst != nullptr;
        const cplx* vp = V + c;
        for (int i0 = 0; i0 < mt; i0 += MTC) {
            cplx acc[MTC];
#pragma unroll
            for (int i = 0; i < MTC; ++i) acc[i] = cmake(0.0, 0.0);
            const cplx* cp = cs + (size_t)i0 * k;
            if (live) {
#pragma unroll 8
                for (int j = g; j < k; j += NG) {
                    const cplx v = vp[(int64_t)j * ldv];
                    if (own && i0 == 0) {
                        const double sc = 1.0 / (double)(j + 1);
                        shift_dst[c + (int64_t)j * ldv] = cmake(v.x * sc, v.y * sc);
                    }
#pragma unroll
                    for (int i = 0; i < MTC; ++i)
                        if (i0 + i < mt) cfma(acc[i], v, cp[i * k + j]);
                }
            }
            if (i0 > 0) __syncthreads();
            if (g < NG) {
#pragma unroll
                for (int i = 0; i < MTC; ++i) part[(size_t)(g * MTC + i) * Fpad + f] = acc[i];
            }
            __syncthreads();
            for (int t = tid; t < MTC * Fpad; t += nthr) {
                const int i = t / Fpad, ff = t - i * Fpad;
                if (ff < F && i0 + i < mt) {
                    cplx s = part[(size_t)i * Fpad + ff];
                    for (int q = 1; q < NG; ++q) s = cadd(s, part[(size_t)(q * MTC + i) * Fpad + ff]);
                    W[(size_t)(i0 + i) * fcap + ff] = s;
                }
            }
        }
    }
    __syncthreads();
    // ---- phase 2: rows of the block, G lanes per row when the block has few rows
    const uint32_t lmask = (1u << lbits) - 1u;
    if (PF && pf) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int l = tid + q * nthr;
            cplx acc = cmake(0.0, 0.0);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const uint32_t id = pidx[q][j];
                cfma(acc, pval[q][j], W[(size_t)(id >> lbits) * fcap + (id & lmask)]);
            }
            if (l < d.nrows) {
                const int i = l / d.zp, jz = l - i * d.zp;
                z[(int64_t)d.r0 + (int64_t)i * d.stride + jz] = acc;
            }
        }
        return;
    }
    int G = 1;
    while (G < 16 && rb * G * 2 <= nthr) G <<= 1;
    const int rpp = nthr / G;
    const int g = tid % G;
    for (int l0 = 0; l0 < rb; l0 += rpp) {
        const int l = l0 + tid / G;
        cplx acc = cmake(0.0, 0.0);
        if (l < rb) {
#pragma unroll 4
            for (int j = g; j < width; j += G) {
                const int64_t e = (int64_t)j * rb + l;
                const uint32_t id = NT ? __builtin_nontemporal_load(ib + e) : ib[e];
                const VT a = tload<NT>(vb + e);
                cfma(acc, a, W[(size_t)(id >> lbits) * fcap + (id & lmask)]);
            }
        }
        for (int off = G >> 1; off > 0; off >>= 1) { acc.x += shfl_xor_d(acc.x, off); acc.y += shfl_xor_d(acc.y, off); }
        if (g == 0 && l < d.nrows) {
            const int i = l / d.zp, jz = l - i * d.zp;
            z[(int64_t)d.r0 + (int64_t)i * d.stride + jz] = acc;
        }
    }
}

// ---- K2 on the same tiles: residuals of all k Ritz pairs of a convergence check (src/errmeasure.jl:128-130, 186-190) ----------
//   R[r, s] = sum_e val[e] F[term(e), s] Q[col(e), s]      Q row-major (n x k, row r holds the k Ritz vectors' r-th entries)
// The wave-per-row kernel (k_spmm_rm) gathers one Q row per matrix entry through L2: 8 entries per row at waveguide scale =
// 8x the bytes of Q through L2 and 0.18 of the HBM roofline at k = 60.  Here the Q rows of a block's column footprint are
// staged in LDS once per column panel (PS columns: F x PS x 16 B), one thread per row walks the row's entries with the
// panel's PS columns in registers, one accumulator per term, and the coefficients are applied once per row at the end.
// HBM traffic of a launch: matrix + Q (the halo rows of a patch are L2 hits, see the header of this file).
// Column norms: per-wave shuffle sums -> LDS -> one partial per block and column, summed in block order by k_sum_partials_d:
// deterministic, no atomics.
template <typename VT, int MT, int PS, bool NT>
__global__ __launch_bounds__(256) void k_tile_resid(const TileDesc* __restrict__ desc, const uint32_t* __restrict__ fp,
                                                    const uint16_t* __restrict__ eidx, const VT* __restrict__ eval,
                                                    const cplx* __restrict__ QT, int64_t ldq, int k, const cplx* __restrict__ F,
                                                    int mt, int fcap, int lbits, cplx* __restrict__ ZT, int64_t ldz,
                                                    double* __restrict__ partial, int swz, int64_t split_row) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cplx* Qt = (cplx*)smem;                                 // [fcap][PS]
    cplx* Fs = Qt + (size_t)fcap * PS;                      // [k][mt]  (F[t + s * mt])
    double* wsum = (double*)(Fs + (size_t)mt * k);          // [2][4 waves][kpad]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int kpad = (k + PS - 1) / PS * PS;
    const int blk = tile_block(swz);
    const TileDesc d = desc[blk];
    for (int i = tid; i < mt * k; i += 256) Fs[i] = F[i];
    const int Fn = d.fp_cnt;
    const uint32_t* __restrict__ fpb = fp + d.fp_off;
    const int width = d.wrb & 0xffff, rb = d.wrb >> 16;
    const uint32_t lmask = (1u << lbits) - 1u;
    const uint16_t* __restrict__ ib = eidx + (int64_t)d.ent_off64 * 64;
    const VT* __restrict__ vb = eval + (int64_t)d.ent_off64 * 64;
    const int s1 = tid % PS;                                // phase 1: this thread's column within the panel
    for (int p0 = 0; p0 < k; p0 += PS) {
        const int pw = min(PS, k - p0);
        if (p0 > 0) __syncthreads();                        // the previous panel's readers are done with Qt
        double qn = 0.0;
        for (int i = tid; i < Fn * PS; i += 256) {
            const int f = i / PS;
            const uint32_t raw = fpb[f];
            cplx q = cmake(0.0, 0.0);
            if (s1 < pw) q = QT[(int64_t)(raw & NEP_COL_MASK) * ldq + p0 + s1];
            Qt[i] = q;
            if (raw & TILE_OWN) qn = fma(q.x, q.x, fma(q.y, q.y, qn));
        }
        __syncthreads();
        double rn[PS];
#pragma unroll
        for (int s = 0; s < PS; ++s) rn[s] = 0.0;
        for (int l = tid; l < rb; l += 256) {
            cplx acc[MT][PS];
#pragma unroll
            for (int t = 0; t < MT; ++t)
#pragma unroll
                for (int s = 0; s < PS; ++s) acc[t][s] = cmake(0.0, 0.0);
#pragma unroll 2
            for (int j = 0; j < width; ++j) {
                const int64_t e = (int64_t)j * rb + l;
                const uint32_t id = NT ? __builtin_nontemporal_load(ib + e) : ib[e];
                const VT a = tload<NT>(vb + e);
                const int t = id >> lbits;
                const cplx* qp = Qt + (size_t)(id & lmask) * PS;
#pragma unroll
                for (int tt = 0; tt < MT; ++tt) {
                    VT am;
                    if constexpr (sizeof(VT) == 8) am = (t == tt) ? a : 0.0; else am = (t == tt) ? a : cmake(0.0, 0.0);
#pragma unroll
                    for (int s = 0; s < PS; ++s) cfma(acc[tt][s], am, qp[s]);
                }
            }
            if (l < d.nrows) {
                const int i = l / d.zp, jz = l - i * d.zp;
                const int64_t row = (int64_t)d.r0 + (int64_t)i * d.stride + jz;
#pragma unroll
                for (int s = 0; s < PS; ++s) {
                    if (s < pw) {
                        cplx r = cmake(0.0, 0.0);
#pragma unroll
                        for (int tt = 0; tt < MT; ++tt)
                            if (tt < mt) cfma(r, Fs[(size_t)(p0 + s) * mt + tt], acc[tt][s]);
                        // split_row >= 0: rows below it only enter the norms, rows from it on are only written (row - split_row)
                        if (split_row < 0) { if (ZT) ZT[row * ldz + p0 + s] = r; rn[s] = fma(r.x, r.x, fma(r.y, r.y, rn[s])); }
                        else if (row < split_row) rn[s] = fma(r.x, r.x, fma(r.y, r.y, rn[s]));
                        else ZT[(row - split_row) * ldz + p0 + s] = r;
                    }
                }
            }
        }
        if (partial) {
            // |r|^2: all 64 lanes hold their rows' sums for the panel's PS columns; |q|^2: lanes with the same tid % PS
#pragma unroll
            for (int s = 0; s < PS; ++s) {
                const double v = wave_reduce_sum(rn[s]);
                if (lane == 0) wsum[(size_t)(0 * 4 + wv) * kpad + p0 + s] = v;
            }
            for (int off = 32; off >= PS; off >>= 1) qn += shfl_xor_d(qn, off);
            if (lane < PS) wsum[(size_t)(1 * 4 + wv) * kpad + p0 + lane] = qn;
        }
    }
    if (partial) {
        __syncthreads();
        for (int t = tid; t < 2 * k; t += 256) {
            const int which = t / k, s = t - which * k;
            const double* w = wsum + (size_t)which * 4 * kpad + s;
            partial[((int64_t)blk * 2 + which) * k + s] = (w[0] + w[kpad]) + (w[2 * kpad] + w[3 * kpad]);
        }
    }
}

// ---- K2 on the tiles, COLUMN-major Q (n x k, column s at Q + s ldq) ------------------------------------------------------------
// With a row-major Q a column panel of a footprint row is a 64-byte piece of a 16 k-byte row: k_tile_resid moves 2.9x the
// algorithmic bytes (PMC, profiles/pmc2/r3_tiles_traffic.json) and walks its panels one after the other inside a block, two
// barriers each.  Column-major, the panel loads are what K1's coefficient phase does -- consecutive lanes read consecutive
// rows of ONE column, 1 KiB per wave load -- and the panels become the SECOND GRID DIMENSION: workgroup (b, p) handles block b
// and the PS columns of panel p, start to end without a loop (footprint x PS tile into LDS, one barrier, thread per row).  The
// block's entries are re-read by the k / PS workgroups that share it (L2 hits: 41 KB per block); every byte of Q crosses HBM
// once.  Measured at n = 1 003 995: k = 8 0.081 ms (0.35 of the HBM roofline; wave-per-row kernel 0.57 ms, row-major tiles
// 0.124 ms), k = 60 0.51 ms (0.26; 0.73 / 0.95 ms).  Two other forms were built, measured and removed: the panel loop inside
// the workgroup with the entries in registers (512 threads, one workgroup per CU: 0.88 ms at k = 60), and a software-pipelined
// one (next panel's loads in flight during the reduction, two LDS tiles: 0.62-0.74 ms); fetching the thread's entries into registers
// ahead of the gather and its barrier (what helps K1) changed nothing either (0.52-0.54 ms).  What all forms share is phase 2's LDS
// traffic: rows x 8 entries x k columns x 16 B = 7.7 GB of 16-byte reads at k = 60, two- to three-way bank conflicts at the 32-byte
// stride of a 2-column tile -- about 0.3 ms of LDS pipe per CU however the panels are arranged; the short-lived workgroups of this
// form overlap it best with the gathers.  Launch order (round 3, late): one-dimensional, the panels of a block back to back on ONE
// XCD, so that the block's entries and footprint list are L2 hits for all but the first panel: 0.531 -> 0.500 ms at k = 60,
// 0.082 -> 0.076 at k = 8 (NEP_K2_CM_ORDER=0: panels as grid.y, a block's workgroups rotate over the XCDs).
template <typename VT, int MT, int PS, bool NT>
__global__ __launch_bounds__(256) void k_tile_resid_cm(const TileDesc* __restrict__ desc, const uint32_t* __restrict__ fp,
                                                       const uint16_t* __restrict__ eidx, const VT* __restrict__ eval,
                                                       const cplx* __restrict__ Q, int64_t ldq, int k, const cplx* __restrict__ F,
                                                       int mt, int fcap, int lbits, cplx* __restrict__ R, int64_t ldr,
                                                       double* __restrict__ partial, int swz, int64_t split_row, int nblk, int npan) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cplx* Qt = (cplx*)smem;                                 // [fcap][PS]
    double* wsum = (double*)(Qt + (size_t)fcap * PS);       // [2][4 waves][PS]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    int blk, p0;
    if (gridDim.y == 1 && npan > 0) {
        // one-dimensional launch, panel fastest INSIDE an XCD (workgroup id % 8 = XCD): the npan workgroups that share a block's
        // entries and footprint list run back to back on the same L2.  (With the panels as grid.y a block's workgroups sit nblk ids
        // apart and rotate over the XCDs unless nblk % 8 == 0: every panel fetched the entries again.)
        const int x = blockIdx.x & 7, i = blockIdx.x >> 3;
        const int per = nblk >> 3, rem = nblk & 7;
        const int tl = i / npan;
        if (tl >= per + (x < rem ? 1 : 0)) return;
        blk = x * per + (x < rem ? x : rem) + tl;
        p0 = (i - tl * npan) * PS;
    } else {
        blk = tile_block(swz);
        p0 = (int)blockIdx.y * PS;
    }
    const int pw = min(PS, k - p0);
    const TileDesc d = desc[blk];
    const int Fn = d.fp_cnt;
    const uint32_t* __restrict__ fpb = fp + d.fp_off;
    const int width = d.wrb & 0xffff, rb = d.wrb >> 16;
    const uint32_t lmask = (1u << lbits) - 1u;
    const uint16_t* __restrict__ ib = eidx + (int64_t)d.ent_off64 * 64;
    const VT* __restrict__ vb = eval + (int64_t)d.ent_off64 * 64;
    // coefficients of this panel: F[t + s mt], s in the panel (uniform loads)
    cplx fc[MT][PS];
#pragma unroll
    for (int s = 0; s < PS; ++s)
#pragma unroll
        for (int t = 0; t < MT; ++t) fc[t][s] = (s < pw && t < mt) ? F[(size_t)(p0 + s) * mt + t] : cmake(0.0, 0.0);
    double qn[PS];
#pragma unroll
    for (int s = 0; s < PS; ++s) qn[s] = 0.0;
    // (the footprint list entries of four trips are loaded together, then their Q values: a trip's list entry -> Q round trips were
    // waited for before the next trip started -- three serial pairs of round trips for a 660-row footprint)
    for (int f0 = tid; f0 < Fn; f0 += 4 * 256) {
        uint32_t raw[4]; cplx qv[4][PS];
#pragma unroll
        for (int u = 0; u < 4; ++u) raw[u] = fpb[f0 + 256 * u < Fn ? f0 + 256 * u : f0];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const cplx* qp = Q + (int64_t)(raw[u] & NEP_COL_MASK) + (int64_t)p0 * ldq;
#pragma unroll
            for (int s = 0; s < PS; ++s) qv[u][s] = qp[(int64_t)(s < pw ? s : 0) * ldq];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int f = f0 + 256 * u;
            if (f < Fn) {
                const bool own = (raw[u] & TILE_OWN) != 0;
#pragma unroll
                for (int s = 0; s < PS; ++s) {
                    const cplx q = s < pw ? qv[u][s] : cmake(0.0, 0.0);
                    Qt[(size_t)f * PS + s] = q;
                    if (own) qn[s] = fma(q.x, q.x, fma(q.y, q.y, qn[s]));
                }
            }
        }
    }
    __syncthreads();
    double rn[PS];
#pragma unroll
    for (int s = 0; s < PS; ++s) rn[s] = 0.0;
    for (int l = tid; l < rb; l += 256) {
        cplx acc[MT][PS];
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int s = 0; s < PS; ++s) acc[t][s] = cmake(0.0, 0.0);
#pragma unroll 4
        for (int j = 0; j < width; ++j) {
            const int64_t e = (int64_t)j * rb + l;
            const uint32_t id = ib[e];                      // (re-read by the other panels of the block: L2 hits, see the launch order)
            const VT a = vb[e];
            const int t = id >> lbits;
            const cplx* tp = Qt + (size_t)(id & lmask) * PS;
#pragma unroll
            for (int tt = 0; tt < MT; ++tt) {
                VT am;
                if constexpr (sizeof(VT) == 8) am = (t == tt) ? a : 0.0; else am = (t == tt) ? a : cmake(0.0, 0.0);
#pragma unroll
                for (int s = 0; s < PS; ++s) cfma(acc[tt][s], am, tp[s]);
            }
        }
        if (l < d.nrows) {
            const int i = l / d.zp, jz = l - i * d.zp;
            const int64_t row = (int64_t)d.r0 + (int64_t)i * d.stride + jz;
#pragma unroll
            for (int s = 0; s < PS; ++s) {
                if (s < pw) {
                    cplx r = cmake(0.0, 0.0);
#pragma unroll
                    for (int tt = 0; tt < MT; ++tt) cfma(r, fc[tt][s], acc[tt][s]);
                    if (split_row < 0) { if (R) R[row + (int64_t)(p0 + s) * ldr] = r; rn[s] = fma(r.x, r.x, fma(r.y, r.y, rn[s])); }
                    else if (row < split_row) rn[s] = fma(r.x, r.x, fma(r.y, r.y, rn[s]));
                    else R[(row - split_row) + (int64_t)(p0 + s) * ldr] = r;
                }
            }
        }
    }
    if (partial) {
#pragma unroll
        for (int s = 0; s < PS; ++s) {
            const double v = wave_reduce_sum(rn[s]);
            const double u = wave_reduce_sum(qn[s]);
            if (lane == 0) { wsum[(0 * 4 + wv) * PS + s] = v; wsum[(1 * 4 + wv) * PS + s] = u; }
        }
        __syncthreads();
        if (tid < 2 * pw) {
            const int which = tid / pw, s = tid - which * pw;
            const double* w = wsum + (size_t)which * 4 * PS + s;
            partial[((int64_t)blk * 2 + which) * k + p0 + s] = (w[0] + w[PS]) + (w[2 * PS] + w[3 * PS]);
        }
    }
}


// ---- K2 in SUPER-PANELS (round 5) ------------------------------------------------------------------------------------------------
// One workgroup per block, one thread per row, ALL k columns in one residency: the row's entries are read ONCE into registers
// (8 entries x 10 bytes; k_tile_resid_cm re-reads them for each of its k / 2 panels) and the Ritz block is walked in panels of 4
// columns through TWO footprint tiles in LDS.  The tiles are filled by the CU's LDS-DMA path (gfx950 global_load_lds_dwordx4: 16
// bytes per lane straight from HBM into LDS, lane l of a wave to slot base + 16 l; no registers, no ds_write): while the workgroup
// reduces panel p against its entries, the loads of panel p + 1 are in flight into the other tile -- the ~55 KB a CU keeps outstanding
// is what streaming at the chip's rate needs (Little: 31 GB/s per CU x ~2 us).  A tile is stored COLUMN by column (Qt[c][f]: the 64 rows
// of a wave read 64 consecutive slots of one column -- a conflict-free 16-byte read; the [f][c] tiles of the older kernels had a 2-3 way
// conflict at their 32-byte stride, 0.3 ms of LDS pipe at k = 60).  Either layout of the Ritz block feeds the same tile: column-major (CM)
// or row-major (the package's own drivers).  HBM traffic of a launch: entries once + halo x 16 n k.  Norm partials and the optional tail
// block as in k_tile_resid_cm; deterministic (fixed-order sums, no atomics).
#define SP_PSW 4
#define SP_IMAX 6            // LDS-DMA instructions per wave and panel: 4 columns x (fpad / 64) chunks over the waves of the workgroup
#define SP_FPAD(NTHR_) ((NTHR_) == 512 ? 640 : ((NTHR_) == 768 ? 896 : 1152))     // footprint slots per tile column, by workgroup size
// (inline assembly, not __builtin_amdgcn_global_load_lds: the compiler orders every LDS read behind ALL outstanding LDS-DMA of the
// wave -- an `s_waitcnt vmcnt(0)` in front of the first ds_read of the panel being reduced, i.e. no overlap at all; here it does not
// know, and the kernel waits itself: vmcnt(0) + barrier before a tile is read)
__device__ __forceinline__ void sp_dma16(const cplx* src, cplx* lds_dst) {
    const uint32_t a = (uint32_t)(size_t)((__attribute__((address_space(3))) char*)lds_dst);
    asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(src), "s"(a) : "memory", "m0");
}
__device__ __forceinline__ void sp_dma4(const uint32_t* src, uint32_t* lds_dst) {            // 4 bytes per lane, lane l to lds_dst + l
    const uint32_t a = (uint32_t)(size_t)((__attribute__((address_space(3))) char*)lds_dst);
    asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dword %0, off" : : "v"(src), "s"(a) : "memory", "m0");
}
// eight per-lane doubles -> their eight wave sums in 3 halving exchanges + 3 plain ones (10 exchanges instead of 48): after the
// exchange with lane ^ 32 a lane keeps v[0..4) or v[4..8) by its bit 5, after ^ 16 two of them by bit 4, after ^ 8 one by bit 3; the
// 8 lanes that share bits 5..3 then add up.  Returns the sum of v[i] on every lane whose bits 5..3 spell i.  Fixed order.
__device__ __forceinline__ double sp_wave_reduce8(const double v[8], int lane) {
    double a[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const double keep = (lane & 32) ? v[4 + i] : v[i], give = (lane & 32) ? v[i] : v[4 + i];
        a[i] = keep + shfl_xor_d(give, 32);
    }
    double b[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const double keep = (lane & 16) ? a[2 + i] : a[i], give = (lane & 16) ? a[i] : a[2 + i];
        b[i] = keep + shfl_xor_d(give, 16);
    }
    const double keep = (lane & 8) ? b[1] : b[0], give = (lane & 8) ? b[0] : b[1];
    double c = keep + shfl_xor_d(give, 8);
    c += shfl_xor_d(c, 4); c += shfl_xor_d(c, 2); c += shfl_xor_d(c, 1);
    return c;
}
// The tiles' entries are TERM-SLOTTED (NepTiles::slotted): slot j of every row belongs to term (slot_terms >> 4 j) & 15, so the
// term of an entry is a scalar.  One accumulator per column runs over the slots, and where the term changes (a uniform branch) it is
// flushed into the residual with the term's coefficient -- 2 FMA per (entry, column) instead of the 2 x m_t + selects of per-term
// accumulators picked by a per-lane mask (measured: the masked form ran VALU-bound at 0.55 ms for k = 60, slower than the kernel
// it was to replace).
// Where a footprint value lives in a tile, by layout of the Ritz block:
//   column-major Q  tile[c][f]              -- a DMA instruction = 64 consecutive slots of ONE panel column (64 consecutive rows of the
//                                              block's column: 1 KB contiguous); a wave's tile read = 64 consecutive 16-byte slots
//   row-major Q     tile[f][c ^ g(f)], g(f) from bits 1..3 of f (see SpMap::g)
//                                           -- a DMA instruction = 16 consecutive slots x the panel's 4 columns (16 rows of the block,
//                                              64 contiguous bytes each: a quarter of the requests of a one-column gather); the
//                                              exchange of columns inside a slot's 64 bytes makes 8 consecutive slots of one column
//                                              fall into 8 different 16-byte bank groups: a wave's tile read is conflict-free too
template <bool CM, int NTHR> struct SpMap {
    static constexpr int fpad = SP_FPAD(NTHR);
    static constexpr int NI = SP_PSW * (fpad >> 6);                                    // DMA instructions per panel (either layout)
    __device__ __forceinline__ static int slot(int i, int lane) { return CM ? (i >> 2) * 64 + lane : i * 16 + (lane >> 2); }
    __device__ __forceinline__ static int g(int f) { return (((f >> 1) ^ (f >> 3)) & 1) | (((f >> 2) & 1) << 1); }
    __device__ __forceinline__ static int col(int i, int lane, int f) { return CM ? (i & 3) : ((lane & 3) ^ g(f)); }
    __device__ __forceinline__ static size_t dst(int i) { return CM ? (size_t)(i & 3) * fpad + (size_t)(i >> 2) * 64 : (size_t)i * 64; }
    __device__ __forceinline__ static bool lists(int i, int lane) { return CM ? (i & 3) == 0 : (lane & 3) == 0; }
    __device__ __forceinline__ static uint32_t entry_off(uint32_t loc) { return CM ? loc * 16u : loc * 64u + (uint32_t)g((int)loc) * 16u; }   // bytes, < 2^16
    __device__ __forceinline__ static size_t elem(int f, int c) { return CM ? (size_t)c * fpad + f : (size_t)f * 4 + (c ^ g(f)); }
};
// one panel (4 columns in `tile`) against the thread's row: residual entries, their squares, the optional store, |q|^2 of the block's
// own columns, and the 8 wave sums into ws[wv][0..8)
template <typename VT, bool CM, int NTHR, int FM>
__device__ __forceinline__ void sp_panel(const cplx* __restrict__ tile, const uint32_t (&pid2)[4], const VT (&pv)[8], int p0, int k,
                                         const cplx* __restrict__ F, int mt, uint32_t slot_terms, bool rowon, int64_t row,
                                         int64_t split_row, cplx* __restrict__ R, int64_t ldr, bool want_norms,
                                         const uint32_t* __restrict__ fpl, int Fn, int tid, int lane, int wv, double* __restrict__ ws) {
    constexpr int PSW = SP_PSW;
    constexpr int fpad = SP_FPAD(NTHR);
    // two columns at a time, each finished (residual entry, its square, the optional store) before the next pair starts: the
    // accumulators of four columns with their tile data in flight pushed loop invariants into scratch
    double red[2 * PSW];
#pragma unroll
    for (int h = 0; h < PSW / 2; ++h) {
        cplx acc[2], r[2];
        const cplx* Fc[2];                                 // coefficient column of each panel column (uniform: scalar loads)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            acc[s] = cmake(0.0, 0.0); r[s] = cmake(0.0, 0.0);
            Fc[s] = F + (p0 + 2 * h + s < k ? p0 + 2 * h + s : k - 1) * mt;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint32_t o16 = (j & 1) ? (pid2[j >> 1] >> 16) : (pid2[j >> 1] & 0xffffu);      // SpMap::entry_off of the entry's footprint slot
            const bool first = j == 0 || ((FM >> (j - 1)) & 1);          // first slot of its term: the accumulators start over
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const cplx q = CM ? *(const cplx*)((const char*)(tile + (size_t)(2 * h + s) * fpad) + o16)
                                  : *(const cplx*)((const char*)tile + (o16 ^ (uint32_t)((2 * h + s) * 16)));
                if (first) acc[s] = cscale(pv[j], q); else cfma(acc[s], pv[j], q);
            }
            if ((FM >> j) & 1) {                                          // last slot of its term (compile time)
                const int tj = (int)((slot_terms >> (4 * j)) & 15u);
#pragma unroll
                for (int s = 0; s < 2; ++s) cfma(r[s], Fc[s][tj], acc[s]);
            }
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int col = p0 + 2 * h + s;
            double r2 = 0.0;
            if (rowon && col < k) {
                r2 = fma(r[s].x, r[s].x, r[s].y * r[s].y);
                if (split_row < 0) { if (R) R[CM ? row + (int64_t)col * ldr : row * ldr + col] = r[s]; }
                else if (row >= split_row) { R[CM ? (row - split_row) + (int64_t)col * ldr : (row - split_row) * ldr + col] = r[s]; r2 = 0.0; }
            }
            red[2 * h + s] = r2; red[PSW + 2 * h + s] = 0.0;
        }
        // (a scheduling fence: the compiler would otherwise issue the tile reads of the whole panel up front and SPILL loop
        // invariants to make room -- and a spill reload is a scratch load whose vmcnt(0) drains the DMA queue, see k_tile_resid_sp)
        __builtin_amdgcn_sched_barrier(0);
    }
    if (want_norms) {
        // |q|^2 over the columns this block owns
        for (int f = tid; f < Fn; f += NTHR) {
            if (fpl[f] & TILE_OWN) {
#pragma unroll
                for (int s = 0; s < PSW; ++s) { const cplx q = tile[SpMap<CM, NTHR>::elem(f, s)]; red[PSW + s] = fma(q.x, q.x, fma(q.y, q.y, red[PSW + s])); }
            }
        }
        const double sum = sp_wave_reduce8(red, lane);
        if ((lane & 7) == 0) ws[wv * 2 * PSW + (lane >> 3)] = sum;          // the lanes 0, 8, .., 56 hold the 8 sums
    }
}
// the same in two steps for the persistent kernel: the loads (results untouched, so that nothing waits for them before the DMA that is
// issued behind them) and, at the block switch, the packing
template <typename VT>
__device__ __forceinline__ void sp_entries_load(const uint16_t* __restrict__ ib, const VT* __restrict__ vb, int rb, int width, int tid,
                                                uint32_t (&nid)[8], VT (&pv)[8]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const bool on = tid < rb && j < width;
        const int64_t e = on ? (int64_t)j * rb + tid : 0;
        nid[j] = __builtin_nontemporal_load(ib + e);
        pv[j] = tload<true>(vb + e);
    }
}
template <typename VT, bool CM, int NTHR>
__device__ __forceinline__ void sp_entries_pack(const uint32_t (&nid)[8], const VT (&npv)[8], int rb, int width, int tid, uint32_t lmask,
                                                uint32_t (&pid2)[4], VT (&pv)[8]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) pid2[j] = 0u;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const bool on = tid < rb && j < width;
        uint32_t o = SpMap<CM, NTHR>::entry_off((uint32_t)nid[j] & lmask);
        pv[j] = npv[j];
        if (!on) { o = 0; if constexpr (sizeof(VT) == 8) pv[j] = 0.0; else pv[j] = cmake(0.0, 0.0); }
        pid2[j >> 1] |= o << (16 * (j & 1));
    }
}
// the row's entries of a block (local footprint BYTE offsets, two per register; the term is the slot's) -- once per block
template <typename VT, bool CM, int NTHR>
__device__ __forceinline__ void sp_entries(const uint16_t* __restrict__ ib, const VT* __restrict__ vb, int rb, int width, int tid,
                                           uint32_t lmask, uint32_t (&pid2)[4], VT (&pv)[8]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) pid2[j] = 0u;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const bool on = tid < rb && j < width;
        const int64_t e = on ? (int64_t)j * rb + tid : 0;
        uint32_t o = SpMap<CM, NTHR>::entry_off((uint32_t)__builtin_nontemporal_load(ib + e) & lmask);
        pv[j] = tload<true>(vb + e);
        if (!on) { o = 0; if constexpr (sizeof(VT) == 8) pv[j] = 0.0; else pv[j] = cmake(0.0, 0.0); }
        pid2[j >> 1] |= o << (16 * (j & 1));
    }
}
// FM: the slots after which the accumulators are flushed (bit j: slot j is the last of its term), known at compile time for the
// common slot layouts (0xD0: the waveguide's 5 + 2 + 1; 0x80: one term; 0xFF: after every slot -- right for ANY slotting) so that
// the panel loop has no branch in it.
template <typename VT, bool CM, int NTHR, int FM>
__global__ __launch_bounds__(NTHR) void k_tile_resid_sp(const TileDesc* __restrict__ desc, const uint32_t* __restrict__ fp,
                                                        const uint16_t* __restrict__ eidx, const VT* __restrict__ eval,
                                                        const cplx* __restrict__ Q, int64_t ldq, int k, const cplx* __restrict__ F,
                                                        int mt, int lbits, uint32_t slot_terms, cplx* __restrict__ R,
                                                        int64_t ldr, double* __restrict__ partial, int swz, int64_t split_row) {
    constexpr int NW = NTHR / 64;
    constexpr int PSW = SP_PSW;
    constexpr int fpad = SP_FPAD(NTHR);       // compile-time tile pitch: the 4 columns of an entry are ONE address + immediate offsets
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cplx* Qt = (cplx*)smem;                                   // [2][PSW][fpad], fpad a multiple of 64
    double* wsum = (double*)(Qt + (size_t)2 * PSW * fpad);    // [2 parities][NW][2 PSW]
    uint32_t* fpl = (uint32_t*)(wsum + 2 * NW * 2 * PSW);     // [fpad]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int blk = tile_block(swz);
    const TileDesc d = desc[blk];
    const int Fn = d.fp_cnt;
    const uint32_t* __restrict__ fpb = fp + d.fp_off;
    const int width = d.wrb & 0xffff, rb = d.wrb >> 16;
    const uint32_t lmask = (1u << lbits) - 1u;
    const uint16_t* __restrict__ ib = eidx + (int64_t)d.ent_off64 * 64;
    const VT* __restrict__ vb = eval + (int64_t)d.ent_off64 * 64;
    // ---- DMA instruction i = wv + t NW of a panel is (chunk i / 4 of 64 footprint slots, panel column i % 4); the lane's footprint
    // column comes from the list in LDS every time (NOT from registers kept across the panels: the compiler spilled those, and a
    // scratch reload is a VMEM load -- its `s_waitcnt vmcnt(0)` waits for every DMA in flight, one after the other, 5 us per panel);
    // slots past the footprint repeat its last one (their tile slots are padding)
    constexpr int NI = PSW * (fpad >> 6);
    const int np = (k + PSW - 1) / PSW;
    auto issue = [&](int p, int buf) {
#pragma unroll
        for (int t = 0; t < SP_IMAX; ++t) {
            const int i = wv + t * NW;
            if (i < NI) {                                      // wave-uniform
                const int f = SpMap<CM, NTHR>::slot(i, lane);
                const int c = SpMap<CM, NTHR>::col(i, lane, f);
                const int cc = p * PSW + c < k ? p * PSW + c : k - 1;
                const int64_t col = (int64_t)(fpl[f] & NEP_COL_MASK);
                sp_dma16(Q + (CM ? col + (int64_t)cc * ldq : col * ldq + cc), Qt + ((size_t)(buf * PSW) * fpad + SpMap<CM, NTHR>::dst(i)));
            }
        }
    };
    // the first TWO panels go out at once, straight from the footprint list in global memory (both tiles are free, and the list in
    // LDS needs a barrier before anybody reads it): a block starts with one memory latency for both instead of one after the other
    {
        uint32_t raw[SP_IMAX];
#pragma unroll
        for (int t = 0; t < SP_IMAX; ++t) {
            const int i = wv + t * NW;
            const int f = SpMap<CM, NTHR>::slot(i < NI ? i : 0, lane);
            raw[t] = fpb[f < Fn ? f : Fn - 1];
        }
#pragma unroll
        for (int t = 0; t < SP_IMAX; ++t) {
            const int i = wv + t * NW;
            if (i < NI) {
                const int f = SpMap<CM, NTHR>::slot(i, lane);
                const int c = SpMap<CM, NTHR>::col(i, lane, f);
                if (SpMap<CM, NTHR>::lists(i, lane)) fpl[f] = raw[t];          // every footprint slot exactly once
                const int64_t col = (int64_t)(raw[t] & NEP_COL_MASK);
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    if (p < np) {
                        const int cc = p * PSW + c < k ? p * PSW + c : k - 1;
                        sp_dma16(Q + (CM ? col + (int64_t)cc * ldq : col * ldq + cc), Qt + ((size_t)(p * PSW) * fpad + SpMap<CM, NTHR>::dst(i)));
                    }
                }
            }
        }
    }
    uint32_t pid2[4]; VT pv[8];
    sp_entries<VT, CM, NTHR>(ib, vb, rb, width, tid, lmask, pid2, pv);
    const int li = tid / d.zp, ljz = tid - li * d.zp;
    const int64_t row = (int64_t)d.r0 + (int64_t)li * d.stride + ljz;
    const bool rowon = tid < d.nrows;
    for (int p = 0; p < np; ++p) {
        const int p0 = p * PSW;
        __builtin_amdgcn_s_waitcnt(0x0f70);        // vmcnt(0): this wave's share of panel p has landed in LDS
        __syncthreads();                           // ... everybody's; and every wave is done with the other tile and with wsum of panel p - 1
        if (partial && p > 0 && tid < 2 * PSW) {
            const int which = tid / PSW, s = tid - which * PSW;
            const double* w = wsum + (size_t)((p - 1) & 1) * NW * 2 * PSW + which * PSW + s;
            double a = 0.0;
#pragma unroll
            for (int q = 0; q < NW; ++q) a += w[q * 2 * PSW];
            partial[((int64_t)blk * 2 + which) * k + p0 - PSW + s] = a;        // (p0 - PSW + s < k: only the LAST panel can be short)
        }
        if (p >= 1 && p + 1 < np) issue(p + 1, (p + 1) & 1);       // (panel 1 went out with panel 0)
        sp_panel<VT, CM, NTHR, FM>(Qt + (size_t)((p & 1) * PSW) * fpad, pid2, pv, p0, k, F, mt, slot_terms, rowon, row, split_row, R, ldr,
                                   partial != nullptr, fpl, Fn, tid, lane, wv, wsum + (size_t)(p & 1) * NW * 2 * PSW);
    }
    __syncthreads();
    if (partial) {
        const int p0 = (np - 1) * PSW, pw = k - p0;
        if (tid < 2 * PSW) {
            const int which = tid / PSW, s = tid - which * PSW;
            if (s < pw) {
                const double* w = wsum + (size_t)((np - 1) & 1) * NW * 2 * PSW + which * PSW + s;
                double a = 0.0;
#pragma unroll
                for (int q = 0; q < NW; ++q) a += w[q * 2 * PSW];
                partial[((int64_t)blk * 2 + which) * k + p0 + s] = a;
            }
        }
    }
}

// ---- the same kernel with a RING OF FOUR HALF-TILES (round 6; column-major Ritz blocks) -----------------------------------------
// k_tile_resid_sp keeps ONE panel (4 columns, fpad x 64 bytes = 57 KB at 768 threads) in flight while it reduces the other: per panel it
// pays max(arithmetic, memory latency), and the arithmetic of a panel (8 entries x 4 columns per row) is well under the ~2.5 us a
// loaded HBM takes to answer -- the kernel ran latency-bound at 0.44 of the bandwidth with traffic already at 1.08 x algorithmic.
// (That was the hypothesis; the measurement refuted it -- see the launch site.  Kept as an opt-in variant.)
// Here the two tile buffers are cut into four half-tiles of 2 columns; three half-panels (86 KB per CU) are in flight while the
// fourth is reduced.  A wave waits for the OLDEST of its outstanding LDS-DMA groups only: loads return in order, so
// `s_waitcnt vmcnt(N)` with N = the DMA instructions this wave issued after that group is exact (the count is a compile-time
// constant per wave class: waves below NIh % NW issue one instruction more per half-panel).  Same LDS, same entries in registers,
// same fixed-order sums -> bitwise the results of k_tile_resid_sp.
__device__ __forceinline__ double sp_wave_reduce4(const double v[4], int lane) {     // sums of v[i] over the wave on the lanes whose bits 5..4 spell i
    double a[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const double keep = (lane & 32) ? v[2 + i] : v[i], give = (lane & 32) ? v[i] : v[2 + i];
        a[i] = keep + shfl_xor_d(give, 32);
    }
    const double keep = (lane & 16) ? a[1] : a[0], give = (lane & 16) ? a[0] : a[1];
    double c = keep + shfl_xor_d(give, 16);
    c += shfl_xor_d(c, 8); c += shfl_xor_d(c, 4); c += shfl_xor_d(c, 2); c += shfl_xor_d(c, 1);
    return c;
}
template <int N> __device__ __forceinline__ void sp_wait_vm() { __builtin_amdgcn_s_waitcnt(0x0f70 | (N & 15) | ((N >> 4) << 14)); }
template <typename VT, int NTHR, int FM, int RING>
__global__ __launch_bounds__(NTHR, (RING == 2 ? 2 * (NTHR / 64) / 4 : 1)) void k_tile_resid_sp4(const TileDesc* __restrict__ desc, const uint32_t* __restrict__ fp,
                                                         const uint16_t* __restrict__ eidx, const VT* __restrict__ eval,
                                                         const cplx* __restrict__ Q, int64_t ldq, int k, const cplx* __restrict__ F,
                                                         int mt, int lbits, uint32_t slot_terms, cplx* __restrict__ R,
                                                         int64_t ldr, double* __restrict__ partial, int swz, int64_t split_row) {
    constexpr int NW = NTHR / 64;
    constexpr int fpad = SP_FPAD(NTHR);
    constexpr int CH = fpad >> 6;                             // 64-slot chunks of a tile column
    constexpr int NIh = 2 * CH;                               // DMA instructions of a half-panel
    constexpr int IPW = (NIh + NW - 1) / NW;                  // ... per wave (waves >= NIh % NW issue one less when NIh % NW != 0)
    constexpr int REMW = NIh % NW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cplx* Qt = (cplx*)smem;                                   // [RING half-tiles][2][fpad]
    double* wsum = (double*)(Qt + (size_t)2 * RING * fpad);   // [2 parities][NW][4]
    uint32_t* fpl = (uint32_t*)(wsum + 2 * NW * 4);           // [fpad]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int blk = tile_block(swz);
    const TileDesc d = desc[blk];
    const int Fn = d.fp_cnt;
    const uint32_t* __restrict__ fpb = fp + d.fp_off;
    const int width = d.wrb & 0xffff, rb = d.wrb >> 16;
    const uint32_t lmask = (1u << lbits) - 1u;
    const uint16_t* __restrict__ ib = eidx + (int64_t)d.ent_off64 * 64;
    const VT* __restrict__ vb = eval + (int64_t)d.ent_off64 * 64;
    const int nh = (k + 1) >> 1;                              // half-panels
    // instruction i of a half-panel: chunk i >> 1 of 64 footprint slots, column i & 1
    auto issue = [&](int hp, int slot) {
#pragma unroll
        for (int t = 0; t < IPW; ++t) {
            const int i = wv + t * NW;
            if (i < NIh) {
                const int f = (i >> 1) * 64 + lane;
                const int c = 2 * hp + (i & 1);
                const int64_t col = (int64_t)(fpl[f] & NEP_COL_MASK);
                sp_dma16(Q + col + (int64_t)(c < k ? c : k - 1) * ldq, Qt + ((size_t)(2 * slot + (i & 1)) * fpad + (size_t)(i >> 1) * 64));
            }
        }
    };
    // the first RING - 1 half-panels go out at once, straight from the footprint list in global memory
    {
        uint32_t raw[IPW];
#pragma unroll
        for (int t = 0; t < IPW; ++t) {
            const int i = wv + t * NW;
            const int f = ((i < NIh ? i : 0) >> 1) * 64 + lane;
            raw[t] = fpb[f < Fn ? f : Fn - 1];
        }
        // all of the list first: the compiler counts only its own loads, so a wait for raw[t] placed between two DMA instructions
        // would also wait for the DMA issued before it (the half-panels would go out one memory latency apart)
        __builtin_amdgcn_s_waitcnt(0x0f70);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int hp = 0; hp < RING - 1; ++hp) {
            if (hp < nh) {
#pragma unroll
                for (int t = 0; t < IPW; ++t) {
                    const int i = wv + t * NW;
                    if (i < NIh) {
                        const int f = (i >> 1) * 64 + lane;
                        if (hp == 0 && (i & 1) == 0) fpl[f] = raw[t];              // every footprint slot exactly once
                        const int c = 2 * hp + (i & 1);
                        const int64_t col = (int64_t)(raw[t] & NEP_COL_MASK);
                        sp_dma16(Q + col + (int64_t)(c < k ? c : k - 1) * ldq, Qt + ((size_t)(2 * hp + (i & 1)) * fpad + (size_t)(i >> 1) * 64));
                    }
                }
            }
        }
    }
    uint32_t pid2[4]; VT pv[8];
    sp_entries<VT, true, NTHR>(ib, vb, rb, width, tid, lmask, pid2, pv);
    const int li = tid / d.zp, ljz = tid - li * d.zp;
    const int64_t row = (int64_t)d.r0 + (int64_t)li * d.stride + ljz;
    const bool rowon = tid < d.nrows;
    const bool big = REMW == 0 || wv < REMW;                  // this wave issues IPW instructions per half-panel (else IPW - 1)
    for (int hp = 0; hp < nh; ++hp) {
        // half-panels hp + 1 .. hp + RING - 2 (when they exist) were issued after hp and may stay in flight
        const int after = nh - 1 - hp < RING - 2 ? nh - 1 - hp : RING - 2;
        if (RING >= 4 && after == 2) { if (big) sp_wait_vm<2 * IPW>(); else sp_wait_vm<2 * (IPW - 1)>(); }
        else if (RING >= 3 && after == 1) { if (big) sp_wait_vm<IPW>(); else sp_wait_vm<IPW - 1>(); }
        else sp_wait_vm<0>();
        __syncthreads();                           // everybody's share has landed; every wave is done with half-tile hp - 1 and with wsum of hp - 1's parity
        if (partial && hp > 0 && tid < 4) {
            const int which = tid >> 1, s = tid & 1, col = 2 * (hp - 1) + s;
            if (col < k) {
                const double* w = wsum + (size_t)((hp - 1) & 1) * NW * 4 + which * 2 + s;
                double a = 0.0;
#pragma unroll
                for (int q = 0; q < NW; ++q) a += w[q * 4];
                partial[((int64_t)blk * 2 + which) * k + col] = a;
            }
        }
        if (hp + RING - 1 < nh) issue(hp + RING - 1, (hp + RING - 1) % RING);      // into the half-tile that hp - 1 has just left
        // ---- reduce half-panel hp (columns 2 hp, 2 hp + 1) against the row's entries: the arithmetic of sp_panel for one pair
        const cplx* tile = Qt + (size_t)(2 * (hp % RING)) * fpad;
        cplx acc[2], r[2];
        const cplx* Fc[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            acc[s] = cmake(0.0, 0.0); r[s] = cmake(0.0, 0.0);
            Fc[s] = F + (2 * hp + s < k ? 2 * hp + s : k - 1) * mt;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint32_t o16 = (j & 1) ? (pid2[j >> 1] >> 16) : (pid2[j >> 1] & 0xffffu);
            const bool first = j == 0 || ((FM >> (j - 1)) & 1);
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const cplx q = *(const cplx*)((const char*)(tile + (size_t)s * fpad) + o16);
                if (first) acc[s] = cscale(pv[j], q); else cfma(acc[s], pv[j], q);
            }
            if ((FM >> j) & 1) {
                const int tj = (int)((slot_terms >> (4 * j)) & 15u);
#pragma unroll
                for (int s = 0; s < 2; ++s) cfma(r[s], Fc[s][tj], acc[s]);
            }
        }
        double red[4];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int col = 2 * hp + s;
            double r2 = 0.0;
            if (rowon && col < k) {
                r2 = fma(r[s].x, r[s].x, r[s].y * r[s].y);
                if (split_row < 0) { if (R) R[row + (int64_t)col * ldr] = r[s]; }
                else if (row >= split_row) { R[(row - split_row) + (int64_t)col * ldr] = r[s]; r2 = 0.0; }
            }
            red[s] = r2; red[2 + s] = 0.0;
        }
        if (partial) {
            for (int f = tid; f < Fn; f += NTHR) {
                if (fpl[f] & TILE_OWN) {
#pragma unroll
                    for (int s = 0; s < 2; ++s) { const cplx q = tile[(size_t)s * fpad + f]; red[2 + s] = fma(q.x, q.x, fma(q.y, q.y, red[2 + s])); }
                }
            }
            const double sum = sp_wave_reduce4(red, lane);
            if ((lane & 15) == 0) (wsum + (size_t)(hp & 1) * NW * 4)[wv * 4 + (lane >> 4)] = sum;      // lanes 0, 16, 32, 48 hold the 4 sums
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
    if (partial && tid < 4) {
        const int which = tid >> 1, s = tid & 1, col = 2 * (nh - 1) + s;
        if (col < k) {
            const double* w = wsum + (size_t)((nh - 1) & 1) * NW * 4 + which * 2 + s;
            double a = 0.0;
#pragma unroll
            for (int q = 0; q < NW; ++q) a += w[q * 4];
            partial[((int64_t)blk * 2 + which) * k + col] = a;
        }
    }
}

// PERSISTENT form: one workgroup per CU walks the blocks of its XCD's range, and the (block, panel) items form ONE pipeline -- while
// the last panel of a block is reduced, the first panel of the next block is already on its way into the other tile.  A block's own
// start-up chain (descriptor -> footprint list -> first tile: three dependent memory latencies, ~4 us of the ~12 us a block takes at
// k = 8) then overlaps with the previous block's arithmetic: the LDS tiles leave room for one workgroup per CU only, so nothing else
// on the CU would hide it.  What travels ahead of the pipeline: the next block's footprint list (by LDS-DMA into the OTHER list
// buffer during the second-to-last panel) and the next block's entries (loaded during the last panel into a second register set,
// taken over at its end: ordinary loads issued BEFORE the item's DMA and consumed at the item's end -- the compiler's vmcnt wait for
// them covers the younger DMA as well, which is exactly the wait the next item starts with).  np >= 2.
template <typename VT, bool CM, int NTHR, int FM>
__global__ __launch_bounds__(NTHR) void k_tile_resid_spp(const TileDesc* __restrict__ desc, const uint32_t* __restrict__ fp,
                                                         const uint16_t* __restrict__ eidx, const VT* __restrict__ eval,
                                                         const cplx* __restrict__ Q, int64_t ldq, int k, const cplx* __restrict__ F,
                                                         int mt, int lbits, uint32_t slot_terms, cplx* __restrict__ R,
                                                         int64_t ldr, double* __restrict__ partial, int64_t split_row, int nblk) {
    constexpr int NW = NTHR / 64;
    constexpr int PSW = SP_PSW;
    constexpr int fpad = SP_FPAD(NTHR);
    constexpr int NI = PSW * (fpad >> 6);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cplx* Qt = (cplx*)smem;                                   // [2][PSW][fpad]
    double* wsum = (double*)(Qt + (size_t)2 * PSW * fpad);    // [2 parities][NW][2 PSW]
    uint32_t* fplb = (uint32_t*)(wsum + 2 * NW * 2 * PSW);    // [2][fpad]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t lmask = (1u << lbits) - 1u;
    // the blocks of this workgroup: XCD x = id % 8 owns a contiguous range of blocks, its gridDim / 8 workgroups take every
    // (gridDim / 8)-th of them (neighbouring blocks -- shared halo columns -- run at the same time on the same L2)
    const int x = blockIdx.x & 7, wi = blockIdx.x >> 3, g8 = gridDim.x >> 3;
    const int per = nblk >> 3, rem = nblk & 7;
    const int lo = x * per + (x < rem ? x : rem), cnt = per + (x < rem ? 1 : 0);
    if (wi >= cnt) return;
    const int M = (cnt - wi + g8 - 1) / g8;
    const int np = (k + PSW - 1) / PSW;
    auto dma_panel = [&](const uint32_t* fl, int p, int buf) {
#pragma unroll
        for (int t = 0; t < SP_IMAX; ++t) {
            const int i = wv + t * NW;
            if (i < NI) {                                      // wave-uniform
                const int f = SpMap<CM, NTHR>::slot(i, lane);
                const int c = SpMap<CM, NTHR>::col(i, lane, f);
                const int cc = p * PSW + c < k ? p * PSW + c : k - 1;
                const int64_t col = (int64_t)(fl[f] & NEP_COL_MASK);
                sp_dma16(Q + (CM ? col + (int64_t)cc * ldq : col * ldq + cc), Qt + ((size_t)(buf * PSW) * fpad + SpMap<CM, NTHR>::dst(i)));
            }
        }
    };
    // ---- block 0: footprint list straight from global memory, both first panels at once (as k_tile_resid_sp)
    int cblk = lo + wi;
    TileDesc d = desc[cblk];
    {
        const uint32_t* __restrict__ fpb = fp + d.fp_off;
        uint32_t raw[SP_IMAX];
#pragma unroll
        for (int t = 0; t < SP_IMAX; ++t) {
            const int i = wv + t * NW;
            const int f = SpMap<CM, NTHR>::slot(i < NI ? i : 0, lane);
            raw[t] = fpb[f < d.fp_cnt ? f : d.fp_cnt - 1];
        }
#pragma unroll
        for (int t = 0; t < SP_IMAX; ++t) {
            const int i = wv + t * NW;
            if (i < NI) {
                const int f = SpMap<CM, NTHR>::slot(i, lane);
                const int c = SpMap<CM, NTHR>::col(i, lane, f);
                if (SpMap<CM, NTHR>::lists(i, lane)) fplb[f] = raw[t];
                const int64_t col = (int64_t)(raw[t] & NEP_COL_MASK);
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    const int cc = p * PSW + c < k ? p * PSW + c : k - 1;
                    sp_dma16(Q + (CM ? col + (int64_t)cc * ldq : col * ldq + cc), Qt + ((size_t)(p * PSW) * fpad + SpMap<CM, NTHR>::dst(i)));
                }
            }
        }
    }
    uint32_t pid2[4]; VT pv[8];
    sp_entries<VT, CM, NTHR>(eidx + (int64_t)d.ent_off64 * 64, eval + (int64_t)d.ent_off64 * 64, d.wrb >> 16, d.wrb & 0xffff, tid, lmask, pid2, pv);
    int li = tid / d.zp;
    int64_t row = (int64_t)d.r0 + (int64_t)li * d.stride + (tid - li * d.zp);
    bool rowon = tid < d.nrows;
    int Fn = d.fp_cnt;
    int q = 0, pblk = 0, pp0 = 0;                 // item counter; block and first column of the previous item (its norm partials)
    for (int m = 0; m < M; ++m) {
        const bool have_next = m + 1 < M;
        const int nblk_id = cblk + g8;
        uint32_t nid[8]; VT npv[8];
        TileDesc dn = d;
        if (have_next) dn = desc[nblk_id];
        for (int p = 0; p < np; ++p, ++q) {
            const int p0 = p * PSW;
            __builtin_amdgcn_s_waitcnt(0x0f70);        // vmcnt(0): this wave's share of the item's tile has landed in LDS
            __syncthreads();                           // ... everybody's; every wave is done with the other tile, the other list, wsum of item q - 1
            if (partial && q > 0 && tid < 2 * PSW) {
                const int which = tid / PSW, s_ = tid - which * PSW;
                if (pp0 + s_ < k) {
                    const double* w = wsum + (size_t)((q - 1) & 1) * NW * 2 * PSW + which * PSW + s_;
                    double a = 0.0;
#pragma unroll
                    for (int u = 0; u < NW; ++u) a += w[u * 2 * PSW];
                    partial[((int64_t)pblk * 2 + which) * k + pp0 + s_] = a;
                }
            }
            // what the NEXT block needs ahead of its first item, issued before this item's DMA (see the kernel comment)
            if (have_next && p == np - 2) {            // the list itself by LDS-DMA (4 bytes per lane) into the other list buffer: no registers
                const uint32_t* __restrict__ fpn = fp + dn.fp_off;
                for (int ch = wv; ch < (fpad >> 6); ch += NW) {
                    const int f = ch * 64 + lane;
                    sp_dma4(fpn + (f < dn.fp_cnt ? f : dn.fp_cnt - 1), fplb + (size_t)((m + 1) & 1) * fpad + ch * 64);
                }
            }
            if (have_next && p == np - 1)
                sp_entries_load<VT>(eidx + (int64_t)dn.ent_off64 * 64, eval + (int64_t)dn.ent_off64 * 64, dn.wrb >> 16, dn.wrb & 0xffff, tid, nid, npv);
            // the tile of item q + 1 (items 0 and 1 went out together above)
            if (q >= 1) {
                if (p + 1 < np) dma_panel(fplb + (size_t)(m & 1) * fpad, p + 1, (q + 1) & 1);
                else if (have_next) dma_panel(fplb + (size_t)((m + 1) & 1) * fpad, 0, (q + 1) & 1);
            }
            sp_panel<VT, CM, NTHR, FM>(Qt + (size_t)((q & 1) * PSW) * fpad, pid2, pv, p0, k, F, mt, slot_terms, rowon, row, split_row, R, ldr,
                                       partial != nullptr, fplb + (size_t)(m & 1) * fpad, Fn, tid, lane, wv, wsum + (size_t)(q & 1) * NW * 2 * PSW);
            pblk = cblk; pp0 = p0;
        }
        if (have_next) {                               // the next block becomes the current one
            sp_entries_pack<VT, CM, NTHR>(nid, npv, dn.wrb >> 16, dn.wrb & 0xffff, tid, lmask, pid2, pv);
            d = dn; cblk = nblk_id;
            li = tid / d.zp;
            row = (int64_t)d.r0 + (int64_t)li * d.stride + (tid - li * d.zp);
            rowon = tid < d.nrows;
            Fn = d.fp_cnt;
        }
    }
    __syncthreads();
    if (partial && tid < 2 * PSW) {
        const int which = tid / PSW, s_ = tid - which * PSW;
        if (pp0 + s_ < k) {
            const double* w = wsum + (size_t)((q - 1) & 1) * NW * 2 * PSW + which * PSW + s_;
            double a = 0.0;
#pragma unroll
            for (int u = 0; u < NW; ++u) a += w[u * 2 * PSW];
            partial[((int64_t)pblk * 2 + which) * k + pp0 + s_] = a;
        }
    }
}

// ---- host: tiles from the stacked CSR -----------------------------------------------------------------------------------------
namespace {

struct TileBuilder {
    int64_t n; int mt; int valbytes; int lbits; int fcap_max;
    const int32_t* rowptr; const uint32_t* idx; const void* vals;
    std::vector<TileDesc> desc;
    std::vector<uint32_t> fp;
    std::vector<uint16_t> eidx;
    std::vector<double> evr; std::vector<cplx> evc;
    std::vector<int32_t> mark, loc;
    std::vector<int64_t> rows;      // scratch: global rows of the block in local order
    std::vector<uint32_t> cols;
    int fcap_seen = 0, wmax = 0, rbmax = 0;
    bool failed = false;
    int slot_w = 0; int slot_off[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};      // slot_w > 0: term t owns the entry slots [slot_off[t], slot_off[t + 1])

    // rows of the rectangle [x0,x1) x [z0,z1) of the grid with line length s (s = 0: the 1-D range [z0, z1))
    int footprint(int64_t r0, int stride, int nx, int nz) {
        const int32_t stamp = split_salt;
        cols.clear();
        for (int i = 0; i < nx; ++i)
            for (int j = 0; j < nz; ++j) {
                const int64_t r = r0 + (int64_t)i * stride + j;
                for (int32_t e = rowptr[r]; e < rowptr[r + 1]; ++e) {
                    const uint32_t c = idx[e] & NEP_COL_MASK;
                    if (mark[c] != stamp) { mark[c] = stamp; cols.push_back(c); }
                }
                if (mark[r] != stamp) { mark[r] = stamp; cols.push_back((uint32_t)r); }     // owned rows are always in the footprint
            }
        return (int)cols.size();
    }
    int32_t split_salt = 0;

    void block(int64_t r0, int stride, int nx, int nz) {
        if (failed) return;
        if (++split_salt == 0x7fffffff) { std::fill(mark.begin(), mark.end(), 0); split_salt = 1; }   // a fresh stamp per attempt
        const int F = footprint(r0, stride, nx, nz);
        if (F > fcap_max || nx * nz > 4096) {
            if (nx > 1) { const int h = nx / 2; block(r0, stride, h, nz); block(r0 + (int64_t)h * stride, stride, nx - h, nz); }
            else if (nz > 1) { const int h = nz / 2; block(r0, stride, 1, h); block(r0 + h, stride, 1, nz - h); }
            else failed = true;
            return;
        }
        std::sort(cols.begin(), cols.end());
        for (int f = 0; f < F; ++f) loc[cols[f]] = f;
        TileDesc d;
        d.r0 = (int32_t)r0; d.stride = stride; d.zp = nz; d.nrows = nx * nz;
        d.fp_off = (int32_t)fp.size(); d.fp_cnt = F;
        const int nrows = nx * nz;
        const int rb = (nrows + 15) / 16 * 16;
        int width = slot_w;
        if (!slot_w)
            for (int i = 0; i < nx; ++i)
                for (int j = 0; j < nz; ++j) {
                    const int64_t r = r0 + (int64_t)i * stride + j;
                    width = std::max(width, rowptr[r + 1] - rowptr[r]);
                }
        // footprint slots; the owned rows carry TILE_OWN
        const size_t fp0 = fp.size();
        for (int f = 0; f < F; ++f) fp.push_back(cols[f]);
        for (int i = 0; i < nx; ++i)
            for (int j = 0; j < nz; ++j) fp[fp0 + loc[r0 + (int64_t)i * stride + j]] |= TILE_OWN;
        // entries: (j, l) at ent0 + j * rb + l, zero padded
        const size_t ent0 = (eidx.size() + 63) / 64 * 64;
        const size_t cnt = (size_t)width * rb;
        eidx.resize(ent0 + cnt, 0);
        if (valbytes == 8) evr.resize(ent0 + cnt, 0.0); else evc.resize(ent0 + cnt, cmake_h(0.0, 0.0));
        for (int i = 0; i < nx; ++i)
            for (int j = 0; j < nz; ++j) {
                const int l = i * nz + j;
                const int64_t r = r0 + (int64_t)i * stride + j;
                int used[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                for (int32_t e = rowptr[r]; e < rowptr[r + 1]; ++e) {
                    const uint32_t c = idx[e] & NEP_COL_MASK, t = idx[e] >> NEP_TERM_SHIFT;
                    const int slot = slot_w ? slot_off[t] + used[t]++ : (int)(e - rowptr[r]);
                    const size_t q = ent0 + (size_t)slot * rb + l;
                    eidx[q] = (uint16_t)((t << lbits) | (uint32_t)loc[c]);
                    if (valbytes == 8) evr[q] = ((const double*)vals)[e]; else evc[q] = ((const cplx*)vals)[e];
                }
            }
        d.ent_off64 = (int32_t)(ent0 / 64);
        d.wrb = width | (rb << 16);
        desc.push_back(d);
        fcap_seen = std::max(fcap_seen, F);
        wmax = std::max(wmax, width); rbmax = std::max(rbmax, rb);
    }
    static cplx cmake_h(double a, double b) { cplx r; r.x = a; r.y = b; return r; }
};

int env_int(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }

}  // namespace

extern "C" {

void nep_tiles_destroy(NepTiles* t) {
    if (!t) return;
    if (t->d_desc) (void)hipFree(t->d_desc);
    if (t->d_fp) (void)hipFree(t->d_fp);
    if (t->d_eidx) (void)hipFree(t->d_eidx);
    if (t->d_eval) (void)hipFree(t->d_eval);
    delete t;
}

// host part of nep_tiles_build: false when the matrix does not qualify
static bool tiles_build_host(TileBuilder& B, int64_t n, int mt, int valbytes, const int32_t* rowptr, const uint32_t* idx,
                             const void* vals, int* stride_out, int* xp_out, int* zp_out) {
    if (env_int("NEP_K1_TILE", 1) == 0 || mt > 8 || n < 64) return false;
    int tb = 0; while ((1 << tb) < mt) ++tb;
    B.n = n; B.mt = mt; B.valbytes = valbytes; B.rowptr = rowptr; B.idx = idx; B.vals = vals;
    B.lbits = 16 - tb;
    const int lds_kb = env_int("NEP_K1_TILE_LDS_KB", 64);
    B.fcap_max = std::min(1 << B.lbits, lds_kb * 1024 / (16 * mt));
    B.mark.assign(n, 0); B.loc.assign(n, 0);
    // dominant off-diagonal stride: the most frequent column offset > 1 (the line length of a 2-D grid numbering)
    int stride = 0;
    {
        const int64_t dmax = std::min<int64_t>(n, (int64_t)1 << 22);
        std::vector<int64_t> cnt(dmax, 0);
        for (int64_t r = 0; r < n; ++r)
            for (int32_t e = rowptr[r]; e < rowptr[r + 1]; ++e) {
                const int64_t dd = (int64_t)(idx[e] & NEP_COL_MASK) - r;
                if (dd > 1 && dd < dmax) ++cnt[dd];
            }
        int64_t best = 0;
        for (int64_t dd = 2; dd < dmax; ++dd) if (cnt[dd] > best) { best = cnt[dd]; stride = (int)dd; }
        if (best < n / 4) stride = 0;
        if (const char* e = getenv("NEP_K1_TILE_STRIDE")) stride = atoi(e);
    }
    // term-slotted entries when the longest row of every term adds up to at most 8 slots (see NepTiles::slotted)
    {
        int cmax[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int64_t r = 0; r < n; ++r) {
            int c[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int32_t e = rowptr[r]; e < rowptr[r + 1]; ++e) ++c[idx[e] >> NEP_TERM_SHIFT];
            for (int t = 0; t < mt; ++t) cmax[t] = std::max(cmax[t], c[t]);
        }
        int tot = 0;
        for (int t = 0; t < mt; ++t) tot += cmax[t];
        if (tot >= 1 && tot <= 8 && env_int("NEP_TILE_SLOTTED", 1)) {
            B.slot_w = tot;
            for (int t = 0; t < mt; ++t) B.slot_off[t + 1] = B.slot_off[t] + cmax[t];
            for (int t = mt; t < 8; ++t) B.slot_off[t + 1] = B.slot_off[t];
        }
    }
    const bool small = n < 32768;
    int zp = env_int("NEP_K1_TILE_ZP", small ? 16 : 64), xp = env_int("NEP_K1_TILE_XP", small ? 4 : 8);
    if (zp < 1) zp = 1;
    if (xp < 1) xp = 1;
    // Patch height for large grids (round 3, late): the launch runs in ROUNDS of (CUs x resident workgroups per CU) blocks, and a
    // last round that is barely filled costs as much as a full one -- 8 x 64 patches on the 1003 x 999 waveguide grid: 2016 blocks
    // = 1.6 rounds of 5 per CU (0.58 of the HBM roofline at k = 8), 10 x 64: 2.1 rounds of 3 (0.47), 11 x 64 / 12 x 64: 1.9 / 1.75
    // rounds of 3 (0.60-0.62).  Pick the height that minimises ceil(rounds) x resident blocks x footprint (a 5-point halo assumed).
    if (!small && stride > 0 && !getenv("NEP_K1_TILE_XP")) {
        const int ncu = env_int("NEP_K1_TILE_NCU", 256);
        const int zq = std::min(zp, stride);
        const int64_t X = n / stride;
        const int mtc = std::min(mt, 4);
        double best = 1.0e300; int best_xp = xp;
        for (int cand = 8; cand <= 14; ++cand) {
            const int64_t nb = ((X + cand - 1) / cand) * ((stride + zq - 1) / zq);
            const int64_t fpn = (int64_t)(cand + 2) * (zq + 2);
            const double lds = (double)fpn * mtc * 16.0;
            if (lds > lds_kb * 1024.0 || fpn > B.fcap_max) continue;
            const int bpc = std::min(8, (int)(152.0 * 1024.0 / lds));
            if (bpc < 3) continue;                            // two workgroups per CU do not hide their own dependent loads
            const int64_t rounds = (nb + (int64_t)ncu * bpc - 1) / ((int64_t)ncu * bpc);
            const double score = (double)rounds * bpc * (double)fpn;
            if (score < best - 1e-9) { best = score; best_xp = cand; }
        }
        xp = best_xp;
    }
    if (stride > 0) {
        zp = std::min(zp, stride);
        const int64_t X = n / stride;
        for (int64_t x0 = 0; x0 < X; x0 += xp)
            for (int z0 = 0; z0 < stride; z0 += zp)
                B.block(x0 * stride + z0, stride, (int)std::min<int64_t>(xp, X - x0), std::min(zp, stride - z0));
        const int R = xp * zp;
        for (int64_t a = X * stride; a < n; a += R) B.block(a, 0, 1, (int)std::min<int64_t>(R, n - a));
    } else {
        const int R = xp * zp;
        for (int64_t a = 0; a < n; a += R) B.block(a, 0, 1, (int)std::min<int64_t>(R, n - a));
    }
    *stride_out = stride; *xp_out = xp; *zp_out = zp;
    return !(B.failed || B.desc.empty());
}

// host-only dry run (tests without a GPU, sanitizer build): builds the tiles, then evaluates z = sum_t A_t (V c_t) for a
// deterministic V (n x k) and C (k x mt) twice -- through the tiles exactly as k_tile_mlincomb walks them (footprint ->
// W -> entries -> row map) and directly from the stacked CSR -- and returns the largest difference relative to max |z|.
// info as nep_tiles_info; info[0] = 0 when the matrix gets no tiles (then *maxerr = 0).
int nep_tiles_dryrun(int64_t n, int mt, int valbytes, const int32_t* rowptr, const uint32_t* idx, const void* vals, int k,
                     int64_t info[8], double* maxerr) {
    for (int i = 0; i < 8; ++i) info[i] = 0;
    *maxerr = 0.0;
    TileBuilder B;
    int stride = 0, xp = 0, zp = 0;
    if (!tiles_build_host(B, n, mt, valbytes, rowptr, idx, vals, &stride, &xp, &zp)) return NEP_OK;
    info[0] = (int64_t)B.desc.size(); info[1] = B.fcap_seen; info[2] = stride; info[3] = xp; info[4] = zp;
    info[5] = (int64_t)B.eidx.size(); info[6] = (int64_t)B.fp.size();
    info[7] = info[5] * (2 + valbytes) + info[6] * 4 + info[0] * (int64_t)sizeof(TileDesc);
    auto hv = [&](int64_t r, int j) { cplx v; v.x = sin(0.37 * (double)(r % 1009) + 1.3 * j) + 0.1; v.y = cos(0.11 * (double)(r % 2003) - 0.7 * j); return v; };
    auto hc = [&](int j, int t) { cplx v; v.x = 1.0 / (1.0 + j + 2 * t); v.y = 0.25 * (t - j % 3); return v; };
    auto mul = [](cplx a, cplx b) { cplx r; r.x = a.x * b.x - a.y * b.y; r.y = a.x * b.y + a.y * b.x; return r; };
    std::vector<cplx> Wd((size_t)n * mt), zd(n), zt(n), Wl;
    std::vector<int> hits(n, 0);
    for (int64_t r = 0; r < n; ++r)
        for (int t = 0; t < mt; ++t) {
            cplx a; a.x = 0; a.y = 0;
            for (int j = 0; j < k; ++j) { const cplx p = mul(hv(r, j), hc(j, t)); a.x += p.x; a.y += p.y; }
            Wd[(size_t)r * mt + t] = a;
        }
    double zmax = 0.0;
    for (int64_t r = 0; r < n; ++r) {
        cplx a; a.x = 0; a.y = 0;
        for (int32_t e = rowptr[r]; e < rowptr[r + 1]; ++e) {
            cplx v; if (valbytes == 8) { v.x = ((const double*)vals)[e]; v.y = 0; } else v = ((const cplx*)vals)[e];
            const cplx p = mul(v, Wd[(size_t)(idx[e] & NEP_COL_MASK) * mt + (idx[e] >> NEP_TERM_SHIFT)]);
            a.x += p.x; a.y += p.y;
        }
        zd[r] = a; zmax = std::max(zmax, std::max(fabs(a.x), fabs(a.y)));
    }
    const uint32_t lmask = (1u << B.lbits) - 1u;
    for (const TileDesc& d : B.desc) {
        Wl.assign((size_t)mt * d.fp_cnt, cplx());
        int owned = 0;
        for (int f = 0; f < d.fp_cnt; ++f) {
            const uint32_t raw = B.fp[d.fp_off + f];
            if (raw & TILE_OWN) ++owned;
            for (int t = 0; t < mt; ++t) Wl[(size_t)t * d.fp_cnt + f] = Wd[(size_t)(raw & NEP_COL_MASK) * mt + t];
        }
        if (owned != d.nrows) { *maxerr = 1e300; return NEP_OK; }
        const int width = d.wrb & 0xffff, rb = d.wrb >> 16;
        for (int l = 0; l < d.nrows; ++l) {
            cplx a; a.x = 0; a.y = 0;
            for (int j = 0; j < width; ++j) {
                const size_t e = (size_t)d.ent_off64 * 64 + (size_t)j * rb + l;
                cplx v; if (valbytes == 8) { v.x = B.evr[e]; v.y = 0; } else v = B.evc[e];
                const uint32_t id = B.eidx[e];
                const cplx p = mul(v, Wl[(size_t)(id >> B.lbits) * d.fp_cnt + (id & lmask)]);
                a.x += p.x; a.y += p.y;
            }
            const int i = l / d.zp, jz = l - i * d.zp;
            const int64_t r = (int64_t)d.r0 + (int64_t)i * d.stride + jz;
            zt[r] = a; ++hits[r];
        }
    }
    double err = 0.0;
    for (int64_t r = 0; r < n; ++r) {
        if (hits[r] != 1) { *maxerr = 1e300; return NEP_OK; }        // every row belongs to exactly one block
        err = std::max(err, std::max(fabs(zt[r].x - zd[r].x), fabs(zt[r].y - zd[r].y)));
    }
    *maxerr = zmax > 0 ? err / zmax : err;
    return NEP_OK;
}

// host arrays of the stacked CSR (idx = term << 25 | column).  *out stays NULL when the matrix does not qualify (more than 8
// terms, a row whose footprint exceeds the LDS budget, NEP_K1_TILE=0): not an error, the caller keeps its other kernels.
int nep_tiles_build(int64_t n, int mt, int valbytes, const int32_t* rowptr, const uint32_t* idx, const void* vals, NepTiles** out) {
    *out = nullptr;
    TileBuilder B;
    int stride = 0, xp = 0, zp = 0;
    if (!tiles_build_host(B, n, mt, valbytes, rowptr, idx, vals, &stride, &xp, &zp)) return NEP_OK;
    NepTiles* t = new NepTiles();
    t->n = n; t->mt = mt; t->valbytes = valbytes; t->nblk = (int)B.desc.size(); t->fcap = (B.fcap_seen + 15) / 16 * 16;
    t->lbits = B.lbits; t->stride = stride; t->xp = xp; t->zp = zp; t->wmax = B.wmax; t->rbmax = B.rbmax;
    t->nent = (int64_t)B.eidx.size(); t->nfp = (int64_t)B.fp.size();
    if (B.slot_w) {
        t->slotted = 1; t->slot_terms = 0;
        int last = 0;
        for (int j = 0; j < 8; ++j) {
            int term = last;
            for (int q = 0; q < mt; ++q) if (j >= B.slot_off[q] && j < B.slot_off[q + 1]) term = q;
            t->slot_terms |= (uint32_t)term << (4 * j); last = term;          // slots past the width repeat the last term (their values are 0)
        }
    }
#define TCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { nep_set_error("%s:%d: %s: %s", __FILE__, __LINE__, #call, hipGetErrorString(e_)); nep_tiles_destroy(t); return NEP_ERR_HIP; } } while (0)
    TCHK(hipMalloc((void**)&t->d_desc, B.desc.size() * sizeof(TileDesc)));
    TCHK(hipMalloc((void**)&t->d_fp, B.fp.size() * 4 + 256));
    TCHK(hipMalloc((void**)&t->d_eidx, B.eidx.size() * 2 + 256));
    TCHK(hipMalloc(&t->d_eval, B.eidx.size() * (size_t)valbytes + 256));
    TCHK(hipMemcpy(t->d_desc, B.desc.data(), B.desc.size() * sizeof(TileDesc), hipMemcpyHostToDevice));
    TCHK(hipMemcpy(t->d_fp, B.fp.data(), B.fp.size() * 4, hipMemcpyHostToDevice));
    TCHK(hipMemcpy(t->d_eidx, B.eidx.data(), B.eidx.size() * 2, hipMemcpyHostToDevice));
    if (valbytes == 8) TCHK(hipMemcpy(t->d_eval, B.evr.data(), B.eidx.size() * 8, hipMemcpyHostToDevice));
    else TCHK(hipMemcpy(t->d_eval, B.evc.data(), B.eidx.size() * 16, hipMemcpyHostToDevice));
#undef TCHK
    *out = t;
    return NEP_OK;
}

// info: blocks, largest footprint, stride, xp, zp, entries (padded), footprint slots, bytes one launch streams (entries +
// footprints + descriptors)
void nep_tiles_info(const NepTiles* t, int64_t info[8]) {
    info[0] = t->nblk; info[1] = t->fcap; info[2] = t->stride; info[3] = t->xp; info[4] = t->zp; info[5] = t->nent; info[6] = t->nfp;
    info[7] = t->nent * (2 + t->valbytes) + t->nfp * 4 + (int64_t)t->nblk * (int64_t)sizeof(TileDesc);
}

// threads per workgroup / split of the k columns over thread groups, by matrix size and k (see the kernel's header)
static void tiles_launch_shape(const NepTiles* t, int k, int* nthr, int* split) {
    *nthr = 256; *split = 0;
    if (t->n < 32768) {
        static const int force = env_int("NEP_K1_TILE_THREADS", 0);
        *nthr = force ? force : (k >= 48 ? 1024 : (k >= 16 ? 512 : 256));
        *split = (k >= 4 && ((t->fcap + 15) & ~15) * 2 <= *nthr) ? 1 : 0;
    }
}
size_t nep_tiles_shmem(const NepTiles* t, int k) {
    int nthr, split; tiles_launch_shape(t, k, &nthr, &split);
    return ((size_t)t->mt * t->fcap + (size_t)t->mt * k + (split ? (size_t)nthr * std::min(t->mt, 4) : 0)) * sizeof(cplx);
}

int nep_tiles_mlincomb(const NepTiles* t, int k, const cplx* dC, int64_t ldc, const cplx* dV, int64_t ldv, cplx* dz,
                       cplx* d_shift, hipStream_t st) {
    const size_t shm = nep_tiles_shmem(t, k);
    if (shm > 160 * 1024) { nep_set_error("tiled K1: k = %d needs %zu bytes of LDS", k, shm); return NEP_ERR_ARG; }
    static const int swz = env_int("NEP_XCD_SWIZZLE", 1);
    static const int pf_on = env_int("NEP_K1_TILE_PF", 1);
    int nthr, split; tiles_launch_shape(t, k, &nthr, &split);
    const bool nt = t->n >= 32768;          // entries streamed once (HBM) vs re-read from L2 by every call (small matrices)
    const bool pf = nt && pf_on && t->rbmax <= 512;      // the register prefetch covers 2 x 256 rows per block (k_tile_mlincomb: rb <= 2 nthr)
    const int mtc = std::min(t->mt, 4);
#define TL(VT, M, NTF, PFF)                                                                                                    \
    do {                                                                                                                       \
        if (shm > 64 * 1024) {                                                                                                 \
            { const int rc_ = nep_raise_lds((const void*)k_tile_mlincomb<VT, M, NTF, PFF>, 160 * 1024); if (rc_) return rc_; }                                  \
        }                                                                                                                      \
        hipLaunchKernelGGL((k_tile_mlincomb<VT, M, NTF, PFF>), dim3((unsigned)t->nblk), dim3(nthr), shm, st, (const TileDesc*)t->d_desc, \
                           (const uint32_t*)t->d_fp, (const uint16_t*)t->d_eidx, (const VT*)t->d_eval, dV, ldv, k, dC, ldc, t->mt,  \
                           t->fcap, t->lbits, dz, d_shift, swz, split);                                                        \
    } while (0)
#define TL_M(VT, NTF, PFF) do { switch (mtc) { case 1: TL(VT, 1, NTF, PFF); break; case 2: TL(VT, 2, NTF, PFF); break; case 3: TL(VT, 3, NTF, PFF); break; default: TL(VT, 4, NTF, PFF); break; } } while (0)
    if (t->valbytes == 8) { if (nt) { if (pf) TL_M(double, true, true); else TL_M(double, true, false); } else TL_M(double, false, false); }
    else { if (nt) { if (pf) TL_M(cplx, true, true); else TL_M(cplx, true, false); } else TL_M(cplx, false, false); }
#undef TL_M
#undef TL
    LAUNCHCHK();
    return NEP_OK;
}

size_t nep_tiles_resid_shmem(const NepTiles* t, int k, int ps) {
    const int kpad = (k + ps - 1) / ps * ps;
    return ((size_t)t->fcap * ps + (size_t)t->mt * k) * sizeof(cplx) + (size_t)2 * 4 * kpad * sizeof(double);
}
// panel width: 8 columns when the footprint tile then still leaves room for two workgroups per CU, else 4
int nep_tiles_resid_ps(const NepTiles* t, int k) {
    static const int force = env_int("NEP_K2_TILE_PS", 0);
    if (force == 4 || force == 8) return force;
    return nep_tiles_resid_shmem(t, k, 8) <= 72 * 1024 ? 8 : 4;
}
int nep_tiles_nblk(const NepTiles* t) { return t->nblk; }
bool nep_tiles_resid_ok(const NepTiles* t, int k) {
    return t->mt <= 4 && nep_tiles_resid_shmem(t, k, nep_tiles_resid_ps(t, k)) <= 160 * 1024;
}

// partial: [nblk][2][k] doubles (|r|^2 then |q|^2 per column), or NULL; ZT (n x k row-major, ld ldz) or NULL
// column-major Q (n x k, ld ldq >= n) and, when given, column-major R: see k_tile_resid_cm
int nep_tiles_resid_cm(const NepTiles* t, int k, const cplx* dF, const cplx* Q, int64_t ldq, cplx* R, int64_t ldr, double* partial,
                       int64_t split_row, hipStream_t st) {
    if (t->mt > 4) { nep_set_error("tiled K2 (column-major): mt = %d not supported", t->mt); return NEP_ERR_ARG; }
    // columns per panel: 2 (measured at n = 1e6, k = 60: 0.51 ms with 2, 0.64 with 4, 1.24 with 8 -- the smaller tile leaves
    // room for more workgroups per CU, and the kernel is bound by the dependent loads of each workgroup, not by bytes)
    static const int ps_env = env_int("NEP_K2_CM_PS", 2);
    const int ps = ps_env == 8 ? 8 : (ps_env == 4 ? 4 : 2);
    const size_t shm = (size_t)t->fcap * ps * sizeof(cplx) + (size_t)2 * 4 * ps * sizeof(double);
    if (shm > 160 * 1024) { nep_set_error("tiled K2 (column-major): footprint too large"); return NEP_ERR_ARG; }
    static const int swz = env_int("NEP_XCD_SWIZZLE", 1);
    const bool nt = t->n >= 32768;
    const int npan = (k + ps - 1) / ps;
    static const int order = env_int("NEP_K2_CM_ORDER", 1);      // 1: panels of a block back to back on one XCD (1-D launch); 0: panels as grid.y
    const dim3 grid = order ? dim3((unsigned)(8 * ((t->nblk + 7) / 8) * npan)) : dim3((unsigned)t->nblk, (unsigned)npan);
    const int npan_arg = order ? npan : 0;
#define RC(VT, M, P, NTF)                                                                                                      \
    do {                                                                                                                       \
        if (shm > 64 * 1024) {                                                                                                 \
            { const int rc_ = nep_raise_lds((const void*)k_tile_resid_cm<VT, M, P, NTF>, 160 * 1024); if (rc_) return rc_; }                                  \
        }                                                                                                                      \
        hipLaunchKernelGGL((k_tile_resid_cm<VT, M, P, NTF>), grid, dim3(256), shm, st, (const TileDesc*)t->d_desc,             \
                           (const uint32_t*)t->d_fp, (const uint16_t*)t->d_eidx, (const VT*)t->d_eval, Q, ldq, k, dF, t->mt,   \
                           t->fcap, t->lbits, R, ldr, partial, swz, split_row, t->nblk, npan_arg);                             \
    } while (0)
#define RC_M(VT, P, NTF) do { switch (t->mt) { case 1: RC(VT, 1, P, NTF); break; case 2: RC(VT, 2, P, NTF); break; case 3: RC(VT, 3, P, NTF); break; default: RC(VT, 4, P, NTF); break; } } while (0)
#define RC_P(VT, NTF) do { if (ps == 8) RC_M(VT, 8, NTF); else if (ps == 2) RC_M(VT, 2, NTF); else RC_M(VT, 4, NTF); } while (0)
    if (t->valbytes == 8) { if (nt) RC_P(double, true); else RC_P(double, false); }
    else { if (nt) RC_P(cplx, true); else RC_P(cplx, false); }
#undef RC_P
#undef RC_M
#undef RC
    LAUNCHCHK();
    return NEP_OK;
}
// K2 in super-panels (k_tile_resid_sp): every row of a block on its own thread, term-slotted entries (at most 8 per row), two
// footprint tiles of 4 columns in LDS.  cm: column-major Q (ld ldq >= n) and R, else row-major (ld >= k).
static int sp_threads(const NepTiles* t) { return t->rbmax <= 512 ? 512 : (t->rbmax <= 768 ? 768 : 1024); }
static size_t sp_shmem(const NepTiles* t, int nthr) {
    const int fpad = SP_FPAD(nthr);
    return (size_t)2 * SP_PSW * fpad * sizeof(cplx) + (size_t)2 * (nthr / 64) * 2 * SP_PSW * sizeof(double) + (size_t)fpad * 4;
}
bool nep_tiles_resid_sp_ok(const NepTiles* t, int k, int cm) {
    (void)k;
    if (!t || !t->slotted || t->mt > 8 || t->wmax > 8 || t->rbmax > 1024) return false;
    // the row's entries carry their tile BYTE offset in 16 bits (sp_entries, SpMap::entry_off): 16 f column-major, 64 f + 16 g(f)
    // row-major -- a footprint slot f >= 1024 of the row-major tile does not fit (fpad is 1152 with 1024-thread workgroups, a
    // 14 x 64 patch has a 16 x 66 = 1056-slot footprint); such matrices take the older row-major kernels
    if (!cm && t->fcap > 1024) return false;
    const int nthr = sp_threads(t);
    const int fpad = SP_FPAD(nthr);
    if (t->fcap > fpad || SP_PSW * (fpad / 64) > SP_IMAX * (nthr / 64)) return false;          // tile pitch; DMA instructions per wave
    return sp_shmem(t, nthr) + (size_t)fpad * 4 <= 160 * 1024;
}
int nep_tiles_resid_sp(const NepTiles* t, int k, const cplx* dF, const cplx* Q, int64_t ldq, int cm, cplx* R, int64_t ldr,
                       double* partial, int64_t split_row, hipStream_t st) {
    if (!nep_tiles_resid_sp_ok(t, k, cm)) { nep_set_error("super-panel K2: not available for this matrix / k = %d / layout %d", k, cm); return NEP_ERR_ARG; }
    const int nthr = sp_threads(t);
    const size_t shm = sp_shmem(t, nthr) + (size_t)SP_FPAD(nthr) * 4;          // (+ the second footprint list of the persistent form)
    static const int swz = env_int("NEP_XCD_SWIZZLE", 1);
    // persistent form (k_tile_resid_spp) from two panels on: one workgroup per CU (the tiles leave room for one), a multiple of 8
    static const int persist = env_int("NEP_K2_SP_PERSIST", 0);
    static int ncu = 0;
    if (!ncu) {
        int dev = 0; hipDeviceProp_t pr;
        ncu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount >= 8) ? pr.multiProcessorCount : 256;
        ncu = env_int("NEP_K2_SP_GRID", ncu) / 8 * 8;
        if (ncu < 8) ncu = 8;
    }
    // measured at n = 1e6 (kernel time, rocprofv3): k = 8 persistent 58.6 us / one block per workgroup 64.6 us (the old kernel: 59.4);
    // k = 60 320 / 297 us (450): with many panels per block the start-up is a small part, and the hardware's own dispatch of one
    // workgroup per free CU balances the tail better than the static walk.  With the second register set of the next block's
    // entries the persistent form sits at the 168-VGPR budget of a 768-thread workgroup and spills since the row-major tile layout
    // added its address arithmetic (84.7 against 70 us per launch at k = 8): OPT-IN.  NEP_K2_SP_PERSIST = 0 (default) never,
    // 1 for 4 < k <= 12, 2 always (k > 4)
    const bool pers = k > SP_PSW && (persist == 2 || (persist == 1 && k <= 3 * SP_PSW));
    // column-major blocks: NEP_K2_SP_RING=4 selects the ring of four half-tiles (k_tile_resid_sp4).  MEASURED (round 6, n = 1e6,
    // k = 60): 0.361 ms against 0.311 ms of the two-tile kernel -- three half-panels in flight instead of one panel did not help,
    // twice as many barriers / wave reductions per block cost 0.3 us each: the kernel is NOT waiting for HBM latency (see DESIGN
    // section 7, round 6: the per-panel arithmetic + LDS time equals the panel's HBM time).  Opt-in, tested.
    // (also tried: RING = 2 with half the LDS so that TWO 768-thread workgroups share a CU -- 24 waves leave 80 VGPRs per lane, the
    // kernel needs 111: 36 spills, their scratch reloads drain the DMA queue inside the panel loop, 0.73 ms.  Not kept.)
    const bool ring4 = cm && !pers && env_int("NEP_K2_SP_RING", 2) == 4;
    const int pgrid = std::min(ncu, (t->nblk + 7) / 8 * 8);
    // flush mask of the slot layout: bit j = slot j is the last of its term
    int fm = 0x80;
    for (int j = 0; j < 7; ++j) if (((t->slot_terms >> (4 * j)) & 15u) != ((t->slot_terms >> (4 * (j + 1))) & 15u)) fm |= 1 << j;
#define SPL(VT, C, NT_, FM_)                                                                                                   \
    do {                                                                                                                       \
        if (pers) {                                                                                                            \
            if (shm > 64 * 1024) { const int rc_ = nep_raise_lds((const void*)k_tile_resid_spp<VT, C, NT_, FM_>, 160 * 1024); if (rc_) return rc_; } \
            hipLaunchKernelGGL((k_tile_resid_spp<VT, C, NT_, FM_>), dim3((unsigned)pgrid), dim3(NT_), shm, st, (const TileDesc*)t->d_desc, \
                               (const uint32_t*)t->d_fp, (const uint16_t*)t->d_eidx, (const VT*)t->d_eval, Q, ldq, k, dF, t->mt,    \
                               t->lbits, t->slot_terms, R, ldr, partial, split_row, t->nblk);                                   \
        } else {                                                                                                               \
            if (shm > 64 * 1024) { const int rc_ = nep_raise_lds((const void*)k_tile_resid_sp<VT, C, NT_, FM_>, 160 * 1024); if (rc_) return rc_; } \
            hipLaunchKernelGGL((k_tile_resid_sp<VT, C, NT_, FM_>), dim3((unsigned)t->nblk), dim3(NT_), shm, st, (const TileDesc*)t->d_desc, \
                               (const uint32_t*)t->d_fp, (const uint16_t*)t->d_eidx, (const VT*)t->d_eval, Q, ldq, k, dF, t->mt,    \
                               t->lbits, t->slot_terms, R, ldr, partial, swz, split_row);                                       \
        }                                                                                                                      \
    } while (0)
#define SPL4R(VT, NT_, FM_, RG_, SHM_)                                                                                           \
    do {                                                                                                                       \
        if ((SHM_) > 64 * 1024) { const int rc_ = nep_raise_lds((const void*)k_tile_resid_sp4<VT, NT_, FM_, RG_>, 160 * 1024); if (rc_) return rc_; } \
        hipLaunchKernelGGL((k_tile_resid_sp4<VT, NT_, FM_, RG_>), dim3((unsigned)t->nblk), dim3(NT_), (SHM_), st, (const TileDesc*)t->d_desc, \
                           (const uint32_t*)t->d_fp, (const uint16_t*)t->d_eidx, (const VT*)t->d_eval, Q, ldq, k, dF, t->mt,        \
                           t->lbits, t->slot_terms, R, ldr, partial, swz, split_row);                                           \
    } while (0)
#define SPL4(VT, NT_, FM_) SPL4R(VT, NT_, FM_, 4, shm)
#define SPL_F(VT, C, NT_) do { if (C && ring4) { if (fm == 0xD0) SPL4(VT, NT_, 0xD0); else if (fm == 0x80) SPL4(VT, NT_, 0x80); else SPL4(VT, NT_, 0xFF); } \
                               else if (fm == 0xD0) SPL(VT, C, NT_, 0xD0); else if (fm == 0x80) SPL(VT, C, NT_, 0x80); else SPL(VT, C, NT_, 0xFF); } while (0)
#define SPL_C(VT, NT_) do { if (cm) SPL_F(VT, true, NT_); else SPL_F(VT, false, NT_); } while (0)
#define SPL_T(VT) do { if (nthr == 512) SPL_C(VT, 512); else if (nthr == 768) SPL_C(VT, 768); else SPL_C(VT, 1024); } while (0)
    if (t->valbytes == 8) SPL_T(double); else SPL_T(cplx);
#undef SPL_T
#undef SPL_C
#undef SPL_F
#undef SPL4
#undef SPL4R
#undef SPL
    LAUNCHCHK();
    return NEP_OK;
}

bool nep_tiles_resid_cm_ok(const NepTiles* t, int k) { return t->mt <= 4 && (size_t)t->fcap * 8 * sizeof(cplx) <= 150 * 1024; }

int nep_tiles_resid(const NepTiles* t, int k, const cplx* dF, const cplx* QT, int64_t ldq, cplx* ZT, int64_t ldz, double* partial,
                    int64_t split_row, hipStream_t st) {
    const int ps = nep_tiles_resid_ps(t, k);
    const size_t shm = nep_tiles_resid_shmem(t, k, ps);
    if (t->mt > 4 || shm > 160 * 1024) { nep_set_error("tiled K2: mt = %d, k = %d not supported", t->mt, k); return NEP_ERR_ARG; }
    static const int swz = env_int("NEP_XCD_SWIZZLE", 1);
    const bool nt = t->n >= 32768;
#define RL(VT, M, P, NTF)                                                                                                      \
    do {                                                                                                                       \
        if (shm > 64 * 1024) {                                                                                                 \
            { const int rc_ = nep_raise_lds((const void*)k_tile_resid<VT, M, P, NTF>, 160 * 1024); if (rc_) return rc_; }                                  \
        }                                                                                                                      \
        hipLaunchKernelGGL((k_tile_resid<VT, M, P, NTF>), dim3((unsigned)t->nblk), dim3(256), shm, st, (const TileDesc*)t->d_desc,  \
                           (const uint32_t*)t->d_fp, (const uint16_t*)t->d_eidx, (const VT*)t->d_eval, QT, ldq, k, dF, t->mt,       \
                           t->fcap, t->lbits, ZT, ldz, partial, swz, split_row);                                               \
    } while (0)
#define RL_M(VT, P, NTF) do { switch (t->mt) { case 1: RL(VT, 1, P, NTF); break; case 2: RL(VT, 2, P, NTF); break; case 3: RL(VT, 3, P, NTF); break; default: RL(VT, 4, P, NTF); break; } } while (0)
#define RL_P(VT, NTF) do { if (ps == 8) RL_M(VT, 8, NTF); else RL_M(VT, 4, NTF); } while (0)
    if (t->valbytes == 8) { if (nt) RL_P(double, true); else RL_P(double, false); }
    else { if (nt) RL_P(cplx, true); else RL_P(cplx, false); }
#undef RL_P
#undef RL_M
#undef RL
    LAUNCHCHK();
    return NEP_OK;
}

}  // extern "C"
