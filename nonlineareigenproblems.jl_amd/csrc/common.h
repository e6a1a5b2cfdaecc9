// Shared definitions for libnepmi355 (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdarg.h>
#include "nepmi355.h"

typedef double2 cplx;  // x = re, y = im  (same bytes as nep_cdouble / Julia ComplexF64)

void nep_set_error(const char* fmt, ...);

#define HIPCHK(call)                                                                       \
    do {                                                                                   \
        hipError_t e_ = (call);                                                            \
        if (e_ != hipSuccess) {                                                            \
            nep_set_error("%s:%d: %s failed: %s", __FILE__, __LINE__, #call,               \
                          hipGetErrorString(e_));                                          \
            return NEP_ERR_HIP;                                                            \
        }                                                                                  \
    } while (0)

#define ARGCHK(cond)                                                                       \
    do {                                                                                   \
        if (!(cond)) {                                                                     \
            nep_set_error("%s:%d: invalid argument: %s", __FILE__, __LINE__, #cond);       \
            return NEP_ERR_ARG;                                                            \
        }                                                                                  \
    } while (0)

#define LAUNCHCHK() HIPCHK(hipGetLastError())

// ---- complex helpers ---------------------------------------------------------------------
__device__ __forceinline__ cplx cmake(double re, double im) { cplx r; r.x = re; r.y = im; return r; }
__device__ __forceinline__ cplx cadd(cplx a, cplx b) { return cmake(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ cplx csub(cplx a, cplx b) { return cmake(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ cplx cmul(cplx a, cplx b) {
    return cmake(fma(a.x, b.x, -(a.y * b.y)), fma(a.x, b.y, a.y * b.x));
}
// acc += a*b
__device__ __forceinline__ void cfma(cplx& acc, cplx a, cplx b) {
    acc.x = fma(a.x, b.x, acc.x); acc.x = fma(-a.y, b.y, acc.x);
    acc.y = fma(a.x, b.y, acc.y); acc.y = fma(a.y, b.x, acc.y);
}
// acc += conj(a)*b
__device__ __forceinline__ void cfma_conj(cplx& acc, cplx a, cplx b) {
    acc.x = fma(a.x, b.x, acc.x); acc.x = fma(a.y, b.y, acc.x);
    acc.y = fma(a.x, b.y, acc.y); acc.y = fma(-a.y, b.x, acc.y);
}
// acc += s*b  (real s)
__device__ __forceinline__ void cfma(cplx& acc, double s, cplx b) {
    acc.x = fma(s, b.x, acc.x); acc.y = fma(s, b.y, acc.y);
}
__device__ __forceinline__ cplx cscale(double s, cplx b) { return cmake(s * b.x, s * b.y); }
__device__ __forceinline__ cplx cscale(cplx s, cplx b) { return cmul(s, b); }

// ---- wave-level helpers (wave = 64) -------------------------------------------------------
__device__ __forceinline__ double shfl_xor_d(double v, int mask) { return __shfl_xor(v, mask, 64); }

template <int W>
__device__ __forceinline__ cplx group_reduce_sum(cplx v) {  // sum over aligned groups of W lanes
#pragma unroll
    for (int off = W / 2; off > 0; off >>= 1) {
        v.x += shfl_xor_d(v.x, off);
        v.y += shfl_xor_d(v.y, off);
    }
    return v;
}
template <int W>
__device__ __forceinline__ double group_reduce_sum(double v) {
#pragma unroll
    for (int off = W / 2; off > 0; off >>= 1) v += shfl_xor_d(v, off);
    return v;
}
__device__ __forceinline__ double wave_reduce_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += shfl_xor_d(v, off);
    return v;
}
// Sum over the 64 lanes by data-parallel-primitive moves (no LDS crossbar: __shfl_xor is a ds_bpermute round trip of ~100 cycles per
// 32-bit half and step): quad permutes (xor 1, xor 2), row rotations by 4 and 8 (every lane of a 16-lane row then holds the row's
// sum), and the four row sums through v_readlane.  The result is uniform.  Fixed order -> deterministic.
template <int CTRL>
__device__ __forceinline__ double dpp_move_d(double v) {
    union { double d; int i[2]; } u, r; u.d = v;
    r.i[0] = __builtin_amdgcn_update_dpp(0, u.i[0], CTRL, 0xf, 0xf, false);
    r.i[1] = __builtin_amdgcn_update_dpp(0, u.i[1], CTRL, 0xf, 0xf, false);
    return r.d;
}
__device__ __forceinline__ double wave_sum_dpp(double v) {
    v += dpp_move_d<0xB1>(v);          // quad_perm [1,0,3,2]
    v += dpp_move_d<0x4E>(v);          // quad_perm [2,3,0,1]
    v += dpp_move_d<0x124>(v);         // row_ror:4
    v += dpp_move_d<0x128>(v);         // row_ror:8
    union { double d; int i[2]; } u, a, b, c, d; u.d = v;
    a.i[0] = __builtin_amdgcn_readlane(u.i[0], 0);  a.i[1] = __builtin_amdgcn_readlane(u.i[1], 0);
    b.i[0] = __builtin_amdgcn_readlane(u.i[0], 16); b.i[1] = __builtin_amdgcn_readlane(u.i[1], 16);
    c.i[0] = __builtin_amdgcn_readlane(u.i[0], 32); c.i[1] = __builtin_amdgcn_readlane(u.i[1], 32);
    d.i[0] = __builtin_amdgcn_readlane(u.i[0], 48); d.i[1] = __builtin_amdgcn_readlane(u.i[1], 48);
    return (a.d + b.d) + (c.d + d.d);
}
__device__ __forceinline__ cplx wave_sum_dpp(cplx v) { return cmake(wave_sum_dpp(v.x), wave_sum_dpp(v.y)); }
__device__ __forceinline__ int readlane_i(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }
__device__ __forceinline__ double readlane_d(double v, int lane) {
    union { double d; int i[2]; } u; u.d = v;
    u.i[0] = __builtin_amdgcn_readlane(u.i[0], lane);
    u.i[1] = __builtin_amdgcn_readlane(u.i[1], lane);
    return u.d;
}

static inline hipStream_t as_stream(nep_stream s) { return (hipStream_t)s; }

// stacked-CSR index packing: high 7 bits = term (<= 128 terms: the particle example of test/nleigs has 83),
// low 25 bits = column (n <= 33.5 M)
#define NEP_TERM_SHIFT 25
#define NEP_COL_MASK ((1u << NEP_TERM_SHIFT) - 1u)
#define NEP_MAX_TERMS 128

// Pinned host staging ring for small host->device parameter blocks (coefficient matrices, MFMA
// B fragments).  hipMemcpyAsync from PAGEABLE memory may read the source after the call returns,
// so caller-owned temporaries must never be handed to it directly: the block is first copied into
// a pinned slot, the async copy is issued from there, and an event guards the slot's reuse.
struct PinnedRing {
    static const int NSLOT = 8;
    void* slot[NSLOT] = {nullptr};
    size_t cap[NSLOT] = {0};
    hipEvent_t ev[NSLOT] = {nullptr};
    bool used[NSLOT] = {false};
    int next = 0;
    // copies `bytes` from hsrc to ddst on `st`; hsrc may be freed by the caller right after return
    int upload(void* ddst, const void* hsrc, size_t bytes, hipStream_t st);
    void release();
    PinnedRing() = default;
    PinnedRing(const PinnedRing&) = delete;
    PinnedRing& operator=(const PinnedRing&) = delete;
    // thread_local rings of short-lived host threads (iar's checker thread, Beyn's builders) are returned when the thread ends
    ~PinnedRing() { release(); }
};

// Caching device allocator for objects that are created and destroyed repeatedly (one LU per quadrature node in
// Beyn, one per iar call): hipMalloc/hipFree of 100 MB blocks cost 10-100 ms each and fluctuate; freed blocks are
// kept (up to a cap) and handed out again.  Reuse is stream-ordered: the library issues all its work on the
// caller's stream, so a block is never reused before the kernels that read it have been enqueued ahead of its
// next writer.
int nep_pool_alloc(void** p, size_t bytes);
void nep_pool_free(void* p);
// free a block that work already enqueued on `st` may still read or write: an event recorded on st travels with the
// block and the next nep_pool_alloc that picks it waits for it (blocks whose event has completed are preferred).
// in_flight = false degrades to nep_pool_free.
void nep_pool_free_on(void* p, hipStream_t st, bool in_flight);

// ---- internal cross-file entry points (exported with C linkage, NOT part of include/nepmi355.h) -------------------------
struct nep_spmf;
extern "C" {
// spmv.hip: UMFPACK's componentwise residual with the coefficients resident on the device (no upload, no memset, no sync);
// d_bits (may be NULL, must hold 0) receives the bit pattern of omega; xsign = -1: dx stores -x
int nep_cw_resid_dev(nep_spmf* s, const double* d_cabs, const nep_cdouble* d_ccf, const nep_cdouble* dx, const nep_cdouble* db,
                     nep_cdouble* dr, unsigned long long* d_bits, double xsign, hipStream_t st);
// spmv.hip: nep_mlincomb_dev with iar's block shift folded into k_vc when that kernel runs (*folded = 1)
int nep_mlincomb_dev_shift(nep_spmf* s, int32_t k, const nep_cdouble* dC, int64_t ldc, const nep_cdouble* dV, int64_t ldv,
                           nep_cdouble* dz, nep_cdouble* d_shift, int32_t* folded, hipStream_t st);
// orth.hip: nep_orth_dev whose last kernel also writes the caller's row (h, beta, flags, ...) to device-mapped host memory
int32_t nep_orth_dev_mirror(const nep_cdouble* dV, int64_t ldv, int64_t rows, int32_t k, const int64_t* d_active_rows,
                            nep_cdouble* dw, nep_cdouble* d_out, int32_t method, nep_cdouble* d_mirror, int32_t nmirror,
                            nep_stream stream);
// ... and with an event (hipEvent_t or NULL) the stream waits for before the first kernel that writes w
int32_t nep_orth_dev_mirror_ev(const nep_cdouble* dV, int64_t ldv, int64_t rows, int32_t k, const int64_t* d_active_rows,
                               nep_cdouble* dw, nep_cdouble* d_out, int32_t method, nep_cdouble* d_mirror, int32_t nmirror,
                               void* before_write, nep_stream stream);
// orth.hip: iar's form (rows = n (k + 1)): the last kernel also forms step k + 1's coefficient product d_WT (n x mt) from the
// normalised vector and writes its block shift to d_shift (k_orth_finish_vc); spmv.hip: the SpMV on such a product
int32_t nep_orth_dev_iar_next(const nep_cdouble* dV, int64_t ldv, int64_t n, int32_t k, const int64_t* d_active_rows,
                              nep_cdouble* dw, nep_cdouble* d_out, int32_t method, nep_cdouble* d_mirror, int32_t nmirror,
                              void* before_write, const nep_cdouble* dC, int64_t ldc, int32_t mt, nep_cdouble* d_WT,
                              nep_cdouble* d_shift, nep_stream stream);
int nep_spmv_wt(nep_spmf* s, const nep_cdouble* d_WT, nep_cdouble* dz, hipStream_t st);
}

// spmv_tile.hip: K1 in one launch on footprint tiles (built from the host arrays of the stacked CSR; *out stays NULL when the
// matrix does not qualify)
struct NepTiles;
extern "C" {
int nep_tiles_build(int64_t n, int mt, int valbytes, const int32_t* rowptr, const uint32_t* idx, const void* vals, NepTiles** out);
void nep_tiles_destroy(NepTiles* t);
void nep_tiles_info(const NepTiles* t, int64_t info[8]);
size_t nep_tiles_shmem(const NepTiles* t, int k);
int nep_tiles_mlincomb(const NepTiles* t, int k, const cplx* dC, int64_t ldc, const cplx* dV, int64_t ldv, cplx* dz, cplx* d_shift,
                       hipStream_t st);
// K2 on the tiles: R = residual block of k Ritz pairs (row-major Q), column norms as [nblk][2][k] partials and / or R itself
int nep_tiles_nblk(const NepTiles* t);
bool nep_tiles_resid_ok(const NepTiles* t, int k);
bool nep_tiles_resid_cm_ok(const NepTiles* t, int k);
int nep_tiles_resid_cm(const NepTiles* t, int k, const cplx* dF, const cplx* Q, int64_t ldq, cplx* R, int64_t ldr, double* partial,
                       int64_t split_row, hipStream_t st);
// split_row >= 0: rows below it enter the norms only, rows from it on are written to ZT (row - split_row) only
int nep_tiles_resid(const NepTiles* t, int k, const cplx* dF, const cplx* QT, int64_t ldq, cplx* ZT, int64_t ldz, double* partial,
                    int64_t split_row, hipStream_t st);
// K2 in super-panels of 8 columns, entries in registers (k_tile_resid_sp): either layout of Q / R (cm != 0: column-major)
bool nep_tiles_resid_sp_ok(const NepTiles* t, int k, int cm);
int nep_tiles_resid_sp(const NepTiles* t, int k, const cplx* dF, const cplx* Q, int64_t ldq, int cm, cplx* R, int64_t ldr,
                       double* partial, int64_t split_row, hipStream_t st);
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (host thread, device, kernel): the attribute belongs to the function
// object of the CURRENT device, so a process that drives several GPUs (nep_set_device) must raise it on each of them -- a plain
// `static bool raised` would launch with more than 64 KB of LDS on a device where the limit was never raised
extern "C" int nep_raise_lds(const void* kernel, int bytes);

// small per-library scratch (device) helpers, defined in util.hip
struct NepScratch {
    void* dptr = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes);
    void release();
    NepScratch() = default;
    NepScratch(const NepScratch&) = delete;
    NepScratch& operator=(const NepScratch&) = delete;
    // A thread_local scratch of a short-lived host thread goes back to the pool when the thread ends (iar starts one checker
    // thread per call: without this every call left its 4 MiB blocks behind -- 9 MB per call, scripts/diag/leak_check.py).
    // The owner must not exit with kernels on the block still pending (iar's checker drains its stream before it returns).
    ~NepScratch() { release(); }
};
