/*
 * nepmi355.h -- C ABI of libnepmi355.so, the MI355X (gfx950) inner linear-algebra backend
 * for NEP-PACK-style nonlinear eigensolvers.
 *
 * The reference (nep-pack/NonlinearEigenproblems.jl, 100 % Julia) has no FFI; its plug-in
 * seams are four small abstract types.  Every entry point below names
 * the reference interface it replaces (file:line under /root/reference).  A Julia binding
 * reaches these with `ccall((:nep_xxx, "libnepmi355"), Cint, (...), ...)` -- see INTEGRATION.md.
 *
 * Conventions
 *   - plain C types only; no C++/torch types cross this boundary.
 *   - every function returns int32 status: 0 = ok, <0 = error (nep_last_error() has text);
 *     nothing throws.
 *   - `d`-prefixed pointers are DEVICE pointers (from nep_dev_alloc, or any HIP allocation of
 *     the same process, e.g. a torch tensor's data_ptr()); `h`-prefixed pointers are HOST.
 *   - dense blocks are complex128 (re,im interleaved), COLUMN-MAJOR with explicit leading
 *     dimension, i.e. exactly Julia's Matrix{ComplexF64} layout -- unless a parameter says
 *     row-major.
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).  Functions whose
 *     results are written to host memory synchronise that stream before returning; all others
 *     are asynchronous with respect to the host.
 *   - a handle is used by one host thread at a time; distinct handles may be used concurrently.
 */
#ifndef NEPMI355_H
#define NEPMI355_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { double re, im; } nep_cdouble;
typedef struct nep_spmf nep_spmf; /* device-resident SPMF: stacked CSR of A_1..A_mt */
typedef struct nep_lu nep_lu;     /* device-resident sparse LU factors + solve schedule */
typedef struct nep_comm nep_comm; /* one rank of the multi-GPU exchange (RCCL communicator) */
typedef void* nep_stream;

#define NEP_OK 0
#define NEP_ERR_HIP -1      /* a HIP runtime call failed (no device, OOM, launch failure) */
#define NEP_ERR_ARG -2      /* invalid argument */
#define NEP_ERR_SINGULAR -3 /* zero pivot met in the triangular solve (SingularException analogue) */
#define NEP_ERR_BREAKDOWN -4 /* orthogonalisation breakdown: ||w|| == 0 */
#define NEP_ERR_UNSUPPORTED -5 /* the request does not fit this code path (internal: callers fall back) */
#define NEP_ERR_RETRY -6       /* nep_iar_run: the recorded refinement / DGKS / eig status of a step asks for more than the
                                * enqueued work did -- re-run through the step-synchronous route (nep_iar_result.retry_reason) */
#define NEP_ERR_NOCONV -7      /* nep_iar_run: fewer than `neigs` pairs converged within maxit steps; the best pairs ARE returned
                                * (the reference throws NoConvergenceException(lambda, Q, err, msg), src/method_iar.jl:163-175) */

/* ---- library / device ---------------------------------------------------------------- */
int32_t nep_version(void);
/* digest (sha256 prefix) of the sources the library was built from; "unknown" for builds outside build.py */
const char* nep_src_digest(void);
const char* nep_last_error(void);
int32_t nep_device_count(int32_t* n);
int32_t nep_set_device(int32_t dev);
int32_t nep_device_name(char* buf, int32_t buflen);

/* ---- raw device memory (for hosts without their own HIP allocator, e.g. Julia) ------- */
int32_t nep_dev_alloc(void** dptr, size_t bytes);
int32_t nep_dev_free(void* dptr);
int32_t nep_dev_memset(void* dptr, int32_t value, size_t bytes, nep_stream stream);
int32_t nep_upload(void* ddst, const void* hsrc, size_t bytes, nep_stream stream);   /* sync */
int32_t nep_download(void* hdst, const void* dsrc, size_t bytes, nep_stream stream); /* sync */
int32_t nep_dev_copy(void* ddst, const void* dsrc, size_t bytes, nep_stream stream);
int32_t nep_stream_sync(nep_stream stream);
/* *out = 1 when kernels of the two streams execute one after the other (the streams share a hardware queue of the runtime's
 * pool), 0 when they overlap.  Measured (three ~0.4 ms probes; both streams are synchronised): a host that puts long
 * one-wavefront kernels (nep_hess_eig*_dev) on a side stream picks one that serialises with neither its main stream nor the
 * stream of its convergence checks. */
int32_t nep_stream_pair_serializes(nep_stream a, nep_stream b, int32_t* out);

/* ---- SPMF object: M(lambda) = sum_i f_i(lambda) A_i ------------------------------------
 * replaces: struct SPMF_NEP  src/NEPTypes.jl:162-170 (+ get_Av :103, DEP :427-443,
 *           PEP src/types_poly.jl:31-34, SPMFSumNEP src/NEPTypes.jl:845-847 -- all are
 *           "a list of matrices" to the device).
 * Input: mt matrices of size n x n in CSR (0-based, int32): rowptr[i] (n+1), colind[i], vals[i]
 * (double if val_is_complex[i]==0 else nep_cdouble).  Julia's CSC of A is the CSR of A^T; the
 * binding passes transpose(A_i) in CSC form, or uses nep_csc_to_csr below.
 * The library builds ONE stacked CSR over all terms (entries sorted by (col,term) per row). */
int32_t nep_spmf_create(int64_t n, int32_t mt, const int32_t* const* h_rowptr,
                        const int32_t* const* h_colind, const void* const* h_vals,
                        const int32_t* h_val_is_complex, nep_spmf** out);
int32_t nep_spmf_destroy(nep_spmf* s);
/* info[0]=n info[1]=mt info[2]=total nnz info[3]=value bytes (8|16) info[4]=lanes per row
 * chosen for the SpMV info[5]=algorithmic matrix bytes of one pass over the stacked CSR */
int32_t nep_spmf_info(const nep_spmf* s, int64_t info[6]);
/* footprint tiles of the one-launch compute_Mlincomb kernel (csrc/spmv_tile.hip; all zero when the matrix has none):
 * info[0]=blocks info[1]=largest column footprint info[2]=detected grid stride (0: consecutive rows) info[3],info[4]=patch
 * shape info[5]=padded entries info[6]=footprint slots info[7]=bytes one launch streams for the matrix side */
int32_t nep_spmf_tile_info(const nep_spmf* s, int64_t info[8]);
/* host-only dry run of those tiles for an SPMF given as in nep_spmf_create (nothing goes to the device; tests without a GPU,
 * sanitizer build): info as above, *maxerr = largest relative difference between z = sum_t A_t (V c_t) evaluated through the
 * tiles (footprint -> W -> entries -> row map, as the kernel walks them) and directly, deterministic V (n x k), C (k x mt) */
int32_t nep_spmf_tiles_analyze(int64_t n, int32_t mt, const int32_t* const* h_rowptr, const int32_t* const* h_colind,
                               const void* const* h_vals, const int32_t* h_val_is_complex, int32_t k, int64_t info[8],
                               double* maxerr);
/* tuning / A-B knob for compute_Mlincomb: 0 = automatic choice per (n, k), 1 = the tiled one-launch kernel whenever the
 * matrix has tiles, 2 = never (k_vc + SpMV / folded SpMV).  Process-wide. */
int32_t nep_k1_set_mode(int32_t mode);
/* K2 kernel choice (A/B knob): the super-panel residual kernel (csrc/spmv_tile.hip k_tile_resid_sp) 0 = never, 1 = on large matrices
 * (default), 2 = whenever the footprint tiles allow it, -1 = the environment variable NEP_K2_SP decides.  Reference: the residual of
 * all Ritz pairs, src/errmeasure.jl:128-130,186-190. */
int32_t nep_k2_set_sp_mode(int32_t mode);
int32_t nep_csc_to_csr(int64_t n, const int64_t* colptr, const int64_t* rowval, const void* nzval,
                       int32_t val_is_complex, int32_t one_based, int32_t* rowptr, int32_t* colind,
                       void* vals);

/* ---- rectangular CSR operator (complex values, 0-based int32 indices) ----------------------
 * replaces: the low-rank factors held by RKNEP, src/rk_helper/rk_nep.jl:19-32,128-152 -- UU (hcat of the U_i,
 *           applied as UU' at src/method_nleigs.jl:430,480,510) and the rows LL / iLr of [L_1 ... L_q]
 *           (scalar loop at :464-471).  nep_csr_mv:  y = alpha*A*x + beta*z   (x: cols entries, y, z: rows
 *           entries; z may alias y and is not read when beta == 0). */
typedef struct nep_csr nep_csr;
int32_t nep_csr_create(int64_t rows, int64_t cols, const int32_t* h_rowptr, const int32_t* h_colind,
                       const nep_cdouble* h_vals, nep_csr** out);
int32_t nep_csr_destroy(nep_csr* a);
int32_t nep_csr_mv(const nep_csr* a, nep_cdouble alpha, const nep_cdouble* dx, nep_cdouble beta,
                   const nep_cdouble* dz, nep_cdouble* dy, nep_stream stream);

/* K1  z = sum_i A_i * (V * C[:,i])          V: n x k (ldv), C: k x mt (host, column-major)
 * replaces: compute_Mlincomb!(::SPMF_NEP,...) src/NEPTypes.jl:972-1011 and
 *           compute_Mlincomb(::DerSPMF,...)   src/NEPTypes.jl:1130-1160 (VafD=V*(a.*fD); z+=Av[j]*VafD[:,j]),
 *           PEP :1016-1045, DEP :940-970, SumNEP :889-890 (all are this with other C).
 * The host forms C[j,i] = a_j * f_i^(j-1)(lambda) (NEPCore.jl:218-228 identity). dV is not modified. */
int32_t nep_mlincomb(nep_spmf* s, int32_t k, const nep_cdouble* hC, const nep_cdouble* dV,
                     int64_t ldv, nep_cdouble* dz, nep_stream stream);

/* same with the coefficient block already resident on the device (column stride ldc >= k): a
 * driver uploads the derivative table once (DerSPMF, src/NEPTypes.jl:1108-1128) and every iteration
 * uses its first k rows -- no host-to-device copy on the critical path. */
int32_t nep_mlincomb_dev(nep_spmf* s, int32_t k, const nep_cdouble* dC, int64_t ldc, const nep_cdouble* dV,
                         int64_t ldv, nep_cdouble* dz, nep_stream stream);

/* K2  residual batch: r_s = sum_i F[i,s] A_i q_s, s=1..k; returns ||r_s||_2 and ||q_s||_2.
 * replaces: k calls of estimate_error -> compute_Mlincomb(nep,lambda_s,q_s)
 *           src/errmeasure.jl:128-130,186-190; call sites src/method_iar.jl:134-135,
 *           src/method_tiar.jl:211-212, src/method_beyncontour.jl:142-144.
 * dQT: the k vectors stored ROW-major (row r holds q_1[r]..q_k[r]; row stride ldq >= k), which is
 * what nep_gemm_ts(..., y_rowmajor=1) produces.  hF: mt x k column-major host.  Synchronous. */
int32_t nep_resid_batch(nep_spmf* s, int32_t k, const nep_cdouble* hF, const nep_cdouble* dQT,
                        int64_t ldq, double* h_rnorm, double* h_qnorm, nep_stream stream);
/* asynchronous variant: no host synchronisation, SQUARED norms stay on the device.  d_out (2k doubles), per panel of
 * kk <= P columns starting at column j0: d_out[2*j0 + j] = ||M(lam_j) q_j||^2, d_out[2*j0 + kk + j] = ||q_j||^2, where the
 * panel width is P = min(256, max(1, 3072 / mt)) (the mt x P coefficient block has to fit 48 KiB of LDS). */
int32_t nep_resid_batch_dev(nep_spmf* s, int32_t k, const nep_cdouble* hF, const nep_cdouble* dQT, int64_t ldq,
                            double* d_out, nep_stream stream);

/* K2 for operators with an extra term on their last rows (WEP: dense corner block on the 2 nz boundary rows,
 * src/gallery_extra/waveguide/Waveguide.jl:351-374): one pass gives d_out (device, 2k doubles) = squared column norms of the
 * SPMF residual over rows [0, row0) and of Q over all rows, and dRT_tail ((n - row0) x k row-major, ld ldt) = the residual
 * rows [row0, n); the caller adds its term to the tail and the tail's norms to d_out. */
int32_t nep_resid_split_dev(nep_spmf* s, int32_t k, const nep_cdouble* hF, const nep_cdouble* dQT, int64_t ldq, int64_t row0,
                            double* d_out, nep_cdouble* dRT_tail, int64_t ldt, nep_stream stream);
/* K2 with a COLUMN-major Ritz block (n x k, column s at dQ + s ldq, what nep_gemm_ts writes with y_rowmajor = 0): at
 * waveguide scale the panel loads of the tiled kernel are then contiguous per column and every byte of Q crosses HBM once
 * (csrc/spmv_tile.hip k_tile_resid_cm).  d_out: 2k squared norms (device).  row0 < 0: whole residual in the norms; row0 >= 0:
 * rows [0, row0) in the norms, rows [row0, n) written to dR_tail ((n - row0) x k column-major, ld ldt) as in
 * nep_resid_split_dev.  NEP_ERR_UNSUPPORTED when the matrix has no footprint tiles or more than 4 terms. */
int32_t nep_resid_batch_cm_dev(nep_spmf* s, int32_t k, const nep_cdouble* hF, const nep_cdouble* dQ, int64_t ldq, int64_t row0,
                               double* d_out, nep_cdouble* dR_tail, int64_t ldt, nep_stream stream);

/* same residuals, but the block R^T (row-major, row stride ldr >= k) is written instead of its norms --
 * for NEPs with an extra non-SPMF term (the WEP corner, src/gallery_extra/waveguide/Waveguide.jl:351-374)
 * whose contribution is added before the norms are taken.  Asynchronous. */
int32_t nep_resid_block(nep_spmf* s, int32_t k, const nep_cdouble* hF, const nep_cdouble* dQT, int64_t ldq,
                        nep_cdouble* dRT, int64_t ldr, nep_stream stream);

/* compute_MM building block: ZT = sum_i A_i * XT[:, i*p:(i+1)*p]  (row-major in and out)
 * replaces: compute_MM(::SPMF_NEP,S,V) src/NEPTypes.jl:276-319 (Z += AA[i]*(V*f_i(S))) after
 *           XT = (V*[F_1..F_mt])^T has been formed with nep_gemm_ts. */
int32_t nep_spmm_terms(nep_spmf* s, int32_t p, const nep_cdouble* dXT, int64_t ldx,
                       nep_cdouble* dZT, int64_t ldz, nep_stream stream);

/* ---- K6 Gram-Schmidt -------------------------------------------------------------------
 * replaces: IterativeSolvers.orthogonalize_and_normalize!(V,w,h,method) (third-party,
 *           IterativeSolvers 0.9.2) at src/method_iar.jl:107, src/method_tiar.jl:128,
 *           src/method_nleigs.jl:293.
 * h = V^H w; w -= V h; [DGKS: while ||w|| < ||corr||/sqrt(2): corr=V^H w; w-=V corr; h+=corr];
 * beta=||w||; w/=beta.   dV: rows x k (ldv).  h_active_rows (nullable, k entries): number of
 * leading rows of column j that can be non-zero (iar's block-triangular basis); rows beyond are
 * skipped.  method: 0 = DGKS, 1 = classical GS (one pass), 2 = modified GS.  Synchronous. */
int32_t nep_orth(const nep_cdouble* dV, int64_t ldv, int64_t rows, int32_t k,
                 const int64_t* h_active_rows, nep_cdouble* dw, nep_cdouble* h_h, double* h_beta,
                 int32_t method, int32_t* h_npasses, nep_stream stream);
/* asynchronous DGKS (method 0) / CGS (method 1): no host synchronisation.  The re-orthogonalisation passes are
 * always enqueued and gate themselves on the device with the same criterion (at most 2 passes, NEP_ORTH_DEV_PASSES); w is normalised on the
 * device.  d_active_rows: DEVICE array (or NULL).  d_out (k+2 complex, device): h[0..k), (beta, 0),
 * (passes, 2*breakdown + another_pass_wanted). */
int32_t nep_orth_dev(const nep_cdouble* dV, int64_t ldv, int64_t rows, int32_t k, const int64_t* d_active_rows,
                     nep_cdouble* dw, nep_cdouble* d_out, int32_t method, nep_stream stream);
/* h = V^H w for a rows x k block (w untouched, k host results, synchronous): the products W^H (A_i v) behind
 * set_projectmatrices! / expand_projectmatrices! of Proj_SPMF_NEP (src/NEPTypes.jl:724-790) and Gram matrices. */
int32_t nep_gemv_h(const nep_cdouble* dV, int64_t ldv, int64_t rows, int32_t k, const nep_cdouble* dw,
                   nep_cdouble* h_h, nep_stream stream);
/* K9  C = W^H Y (k x p, host, column-major) for two ROW-major device blocks WT (rows x k), YT (rows x p), k, p <= 256:
 * all of B_i = W^H (A_i V) at once (YT from nep_resid_block with F = e_i 1^T), FP64 MFMA with the row index as contraction
 * index, per-workgroup partial tiles summed in a fixed order.  Synchronous. */
int32_t nep_gemm_h_rm(const nep_cdouble* dWT, int64_t ldw, const nep_cdouble* dYT, int64_t ldy, int64_t rows,
                      int32_t k, int32_t p, nep_cdouble* h_C, nep_stream stream);

/* ---- eigen-decomposition of the small Hessenberg matrix of a Krylov driver, on the device ---------------------------------
 * replaces: `D,Z = eigen(H[1:k,1:k])`  src/method_iar.jl:112, src/method_tiar.jl:182 (LAPACK zgeev on the host there).
 * dH: k x k upper Hessenberg, column-major with leading dimension ldh (entries below the first subdiagonal are not read --
 * the rows of nep_iar_step's device H block, leading dimension m + 4, are exactly this layout).
 * nep_hess_eigvals_dev: eigenvalues by the shifted QR iteration (two wavefronts, matrix packed in LDS; k <= 128, else
 *   NEP_ERR_UNSUPPORTED) -> d_w[0..k); d_w[k] = (0 | 1-based index of the eigenvalue the iteration gave up on, sweeps).
 * nep_hess_eigvecs_dev: right eigenvectors by inverse iteration (LAPACK zhsein's scheme, one wavefront per eigenvalue) ->
 *   dZ (k x k column-major, ldz), unit 2-norm, largest component real positive; d_w[k+1] = (vectors that failed, 0).
 *   Must follow nep_hess_eigvals_dev on the same stream with the same d_w / d_work.
 * d_w: k + 2 complex (device).  d_work: nep_hess_eig_worksize bytes (device).  h_mirror: NULL, or mapped pinned host memory
 * of k + 2 complex that receives a copy of d_w (eigenvalues + status as soon as the first kernel ends, the second status
 * word when the last eigenvector is done) -- a host that records an event behind each call reads them without a copy command.
 * A caller that finds a non-zero status falls back to LAPACK.  Asynchronous. */
int32_t nep_hess_eig_worksize(int32_t k, int64_t* bytes);
int32_t nep_hess_eigvals_dev(int32_t k, const nep_cdouble* dH, int64_t ldh, nep_cdouble* d_w, void* d_work,
                             nep_cdouble* h_mirror, nep_stream stream);
int32_t nep_hess_eigvecs_dev(int32_t k, nep_cdouble* d_w, nep_cdouble* dZ, int64_t ldz, void* d_work, nep_cdouble* h_mirror,
                             nep_stream stream);
/* the same for a batch of nb leading blocks of ONE Hessenberg matrix, sizes k0, k0 + kstep, ... (the Arnoldi matrices of
 * consecutive steps), one workgroup per block in a single launch -- the decompositions of a driver's steps then overlap each
 * other on one stream instead of queueing behind each other (a decomposition is a serial chain of ~50 k^2 rotations: 3 ms at
 * k = 100, while an Arnoldi step takes 0.3 ms).  Block b: results at d_w + b w_stride (w_stride >= kmax + 2), eigenvectors at
 * dZ + b z_stride with leading dimension ldz >= kmax, workspace d_work + b work_stride bytes (work_stride >=
 * nep_hess_eig_worksize(kmax), multiple of 16), mirror at h_mirror + b mirror_stride. */
int32_t nep_hess_eigvals_batch_dev(int32_t nb, int32_t k0, int32_t kstep, const nep_cdouble* dH, int64_t ldh, nep_cdouble* d_w,
                                   int64_t w_stride, void* d_work, int64_t work_stride, nep_cdouble* h_mirror,
                                   int64_t mirror_stride, nep_stream stream);
int32_t nep_hess_eigvecs_batch_dev(int32_t nb, int32_t k0, int32_t kstep, nep_cdouble* d_w, int64_t w_stride, nep_cdouble* dZ,
                                   int64_t ldz, int64_t z_stride, void* d_work, int64_t work_stride, nep_cdouble* h_mirror,
                                   int64_t mirror_stride, nep_stream stream);

/* ---- K7 tall-skinny GEMM on the FP64 matrix cores --------------------------------------
 * Y = Z * B,  Z: rows x k (ldz, device), B: k x p (host, column-major, ldb), Y: rows x p.
 * y_rowmajor=0: Y column-major (ldy >= rows); =1: Y row-major (ldy >= p).
 * replaces: Q=VV*Z src/method_iar.jl:115; Z[:,1:k]*transpose(a[1:k,k,1:k]) src/method_tiar.jl:119,
 *           :188-189; V*(H*S) src/method_nleigs.jl:318; V*(a.*fD) src/NEPTypes.jl:1154. */
int32_t nep_gemm_ts(const nep_cdouble* dZ, int64_t ldz, int64_t rows, int32_t k,
                    const nep_cdouble* hB, int64_t ldb, int32_t p, nep_cdouble* dY, int64_t ldy,
                    int32_t y_rowmajor, nep_stream stream);

/* same with B resident on the device: B[c,j] = dB[j*ldb + c] (b_rowmajor=0) or dB[c*ldb + j] (=1).
 * Used for the WEP corner term R*diag(s)*R^H (dense nz x nz scaled-DFT blocks, Waveguide.jl:53-65,351-374). */
int32_t nep_gemm_ts_dev(const nep_cdouble* dZ, int64_t ldz, int64_t rows, int32_t k,
                        const nep_cdouble* dB, int64_t ldb, int32_t b_rowmajor, int32_t p,
                        nep_cdouble* dY, int64_t ldy, int32_t y_rowmajor, nep_stream stream);

/* plain dense complex GEMM, column-major:  C = alpha op(A) op(B) + beta C,  op = 0 none | 1 transpose | 2 conjugate
 * transpose; A: m x k after op, B: k x n after op.  The library's own LDS-tiled kernel (csrc/gemm.hip k_gemm_general; no
 * vendor BLAS behind the ABI); C is not read when beta == 0.
 * replaces: the FFTW transforms of the waveguide Sylvester solver, src/gallery_extra/waveguide/waveguide_preconditioner.jl:
 *           120-219 (V!, Vh!, W, Wh as dense DFT / sine-transform matrices), and the small dense products of the
 *           Sylvester-SMW preconditioner (:221-421). */
/* Thin QR of a tall device block by column-wise DGKS (K6), no host synchronisation: dQ (rows x k, column-major, ld ldq) is
 * orthonormalised in place; row j of d_out (k rows of k + 2 complex, device) = R[0..j, j], (R[j, j], 0), (passes, 2 * breakdown +
 * another_pass_wanted).  A column inside the span of its predecessors leaves a tiny R[j, j] (the rank shows in R).
 * replaces: `V, S, W = svd(A0)` of the n x k moment block in Beyn's method, src/method_beyncontour.jl:114-121, which becomes
 *           svd(R) of the k x k factor (V = Q U_R). */
int32_t nep_orth_qr_dev(nep_cdouble* dQ, int64_t ldq, int64_t rows, int32_t k, nep_cdouble* d_out, nep_stream stream);
/* Dense complex inverse on the device (in-place Gauss-Jordan with partial pivoting, deterministic):
 *   dOut (n x n, column-major, ld ldo) = inv(M + add_identity * I)^H,   M n x n column-major (ld ldm), both on the device.
 * dWork: 2 n + 2 complex of device scratch.  *h_info = 0, or 1 + the step whose pivot was zero / not finite (the call synchronises
 * the stream once at its end to read it).
 * replaces: `inv(M)` of the mm x mm Sylvester-SMW matrix, src/gallery_extra/waveguide/waveguide_preconditioner.jl:221-313 (the
 *           adjoint of the inverse is what nep_gemv_hd applies). */
int32_t nep_zinv_h_dev(int32_t n, const nep_cdouble* dM, int64_t ldm, double add_identity, nep_cdouble* dOut, int64_t ldo,
                       nep_cdouble* dWork, int32_t* h_info, nep_stream stream);
/* nep_zgemm with the reduction split over `ksplit` workgroups per tile, slices summed in order (deterministic); dWork: ksplit * m * n
 * complex (device).  For products of a few tiles with a long K: the k x k block V0^H A1 of src/method_beyncontour.jl:123. */
int32_t nep_zgemm_sk(int32_t transa, int32_t transb, int32_t m, int32_t n, int32_t k, nep_cdouble alpha,
                     const nep_cdouble* dA, int64_t lda, const nep_cdouble* dB, int64_t ldb, nep_cdouble beta,
                     nep_cdouble* dC, int64_t ldc, int32_t ksplit, nep_cdouble* dWork, nep_stream stream);
int32_t nep_zgemm(int32_t transa, int32_t transb, int32_t m, int32_t n, int32_t k, nep_cdouble alpha,
                  const nep_cdouble* dA, int64_t lda, const nep_cdouble* dB, int64_t ldb, nep_cdouble beta,
                  nep_cdouble* dC, int64_t ldc, nep_stream stream);

/* real GEMM (same kernel, float64), op = 0 none | 1 transpose.  A complex column-major m x n block is a real 2m x n block, so
 * products X * W with a real W (the sine transform W of waveguide_preconditioner.jl:176-199) run at half the flops. */
int32_t nep_dgemm(int32_t transa, int32_t transb, int32_t m, int32_t n, int32_t k, double alpha,
                  const double* dA, int64_t lda, const double* dB, int64_t ldb, double beta,
                  double* dC, int64_t ldc, nep_stream stream);

/* Refinement criterion of the fixed-shift solve (replaces UMFPACK's internal iterative refinement behind
 * `Afact \ x`, src/LinSolvers.jl:114-122 with control[8] = umfpack_refinements): writes r = b - M(lam) x and, if
 * h_omega != NULL, returns the componentwise backward error
 *     max_i |r_i| / (sum_t h_cabs[t] * (|A_t| |x|)_i + |b_i|)          (Arioli/Demmel/Duff), h_cabs[t] = |f_t(lam)|
 * after a stream synchronisation; with h_omega == NULL nothing is read back (blind refinement step).  M(lam) x is
 * either given (dMx, e.g. from a NEP-specific compute_Mlincomb with non-SPMF terms; h_c == NULL) or formed in the same
 * pass over the matrices from the host coefficients h_c[t] = f_t(lam) (dMx == NULL).  d_den_extra (device, n complex,
 * real parts used, may be NULL) is added to the denominator: (|P||x|)_i of operator parts outside the SPMF terms (the
 * dense corner block of the waveguide problem). */
int32_t nep_cw_backward_error(nep_spmf* s, const double* h_cabs, const nep_cdouble* h_c, const nep_cdouble* dx,
                              const nep_cdouble* db, const nep_cdouble* dMx, const nep_cdouble* d_den_extra,
                              nep_cdouble* dr, double* h_omega, nep_stream stream);
/* out[i] = (|x[i]|, 0): feeds the non-SPMF part of the denominator above */
int32_t nep_absvec(int64_t len, const nep_cdouble* dx, nep_cdouble* dout, nep_stream stream);

/* ---- K5 fixed-shift solve with a host-computed sparse LU -------------------------------
 * replaces: FactorizeLinSolver / lin_solve src/LinSolvers.jl:109-137 (Afact \ x) and
 *           BackslashLinSolver :147-159; the factorisation (UMFPACK in the reference) stays on
 *           the host, one-off per shift (BASELINE.json north_star).
 * Factors satisfy Pr*A*Pc = L*U with L unit lower (diagonal stored or not), U upper.  L and U are
 * given in CSR (0-based int32, complex128 values).  perm_r/perm_c follow SciPy/SuperLU:
 * (Pr b)[perm_r[i]] = b[i],  x[i] = y[perm_c[i]].  NULL perms mean identity. */
int32_t nep_lu_create(int64_t n, const int32_t* hLp, const int32_t* hLi, const nep_cdouble* hLx,
                      const int32_t* hUp, const int32_t* hUi, const nep_cdouble* hUx,
                      const int32_t* h_perm_r, const int32_t* h_perm_c, nep_lu** out);
/* The same with L and U in CSC (compressed columns: colptr, rowidx, values) -- the layout UMFPACK (Julia's `F.L`, `F.U`,
 * SparseMatrixCSC) and SuperLU (SciPy's `lu.L`, `lu.U`) hand out, so the host never converts.  Row indices inside a
 * column need not be sorted. */
int32_t nep_lu_create_csc(int64_t n, const int32_t* hLp, const int32_t* hLi, const nep_cdouble* hLx,
                          const int32_t* hUp, const int32_t* hUi, const nep_cdouble* hUx,
                          const int32_t* h_perm_r, const int32_t* h_perm_c, nep_lu** out);
/* Same-pattern refactorisation (src/method_beyncontour.jl:89-94 factors N matrices M(sigma + g(t_i)) of one sparsity
 * pattern; src/LinSolverCreators.jl:62-122 recycles by lambda): new values for the factors `lu` was created with, in the
 * same entry order (and the same CSR/CSC layout).  Only the numeric part runs: value upload + the device kernels that
 * invert the diagonal blocks.  The symbolic analysis (elimination tree, block partition, index arrays) is also shared
 * automatically between nep_lu_create[_csc] calls whose L/U patterns and permutations coincide (pattern-hash cache). */
int32_t nep_lu_refactor(nep_lu* lu, const nep_cdouble* hLx, const nep_cdouble* hUx);
/* Row scaling of the factorised matrix: the factors are those of Pr*diag(rs)*A*Pc (Julia's UMFPACK wrapper:
 * F.L*F.U == (F.Rs .* A)[F.p, F.q], i.e. rs = F.Rs), so every right-hand side is multiplied by rs on the way in (inside the
 * first kernel of the solve).  h_rs: n doubles, NULL removes it. */
int32_t nep_lu_set_row_scale(nep_lu* lu, const double* h_rs);
int32_t nep_lu_destroy(nep_lu* lu);
/* hint for the NEXT nep_lu_create of the calling thread: how many solves the factorisation will serve (default 50).
 * It sizes the dense tail block whose inverse is built at creation time (one-off ~T^2/256 us vs a shorter
 * dependency chain per solve): FactorizeLinSolver (iar/tiar: maxit solves) passes a large number,
 * BackslashLinSolver (one block solve per factorisation, src/LinSolvers.jl:157-159) passes 1. */
int32_t nep_lu_set_expected_solves(int32_t nsolves);
/* host threads of the plan enumeration of nep_lu_refac_create when it runs on the host (0 = default 6; the variable
 * NEP_LU_PLAN_THREADS overrides): a setter instead of an environment write from a running multi-threaded host */
int32_t nep_lu_set_plan_threads(int32_t n);
/* info[0]=n info[1]=nnz(L) info[2]=nnz(U) info[3]=dependent steps(L) info[4]=dependent steps(U)
 * info[5]=bytes one solve with one right-hand side moves under the schedule in use (the ALGORITHMIC bytes of SURVEY.md
 * section 8d are (nnz(L)+nnz(U))*20 + 8(n+1) + 48n, computable from info[0..2]) */
int32_t nep_lu_info(const nep_lu* lu, int64_t info[6]);
/* schedule introspection.  Block schedule (default, trsv_ml.hip): out[0]=0, out[1]=kernel launches of the last solve,
 * out[2]=out[3]=levels of the block partition, out[4]=levels whose coupling product is a separate launch (L+U),
 * out[5]=number of diagonal blocks, out[6]=rows covered by inverted blocks (= n), out[7]=largest block size.
 * Level schedule (NEP_LU_SCHED=old, and the fallback): out[0]=dense tail size T, out[1]=launches, out[2]=levels(L),
 * out[3]=levels(U) of the plain level schedule, out[4]=wide, out[5]=narrow segments, out[6]=rows of the blocked mid
 * region, out[7]=its block size.  nep_lu_is_block_schedule tells which one a handle uses. */
int32_t nep_lu_schedule(const nep_lu* lu, int64_t out[8]);
int32_t nep_lu_is_block_schedule(const nep_lu* lu, int32_t* out);
/* host-only symbolic analysis of the block schedule for a pair of factor patterns (csc = 0: CSR, 1: CSC); touches no
 * device.  out[0]=levels, out[1]=diagonal blocks, out[2]=largest block, out[3]/out[4]=coupling non-zeros / packed inverse
 * entries of L, out[5]/out[6] the same for U, out[7]=levels with a separate coupling launch.  NEP_ERR_UNSUPPORTED when the
 * patterns have a dependency outside the elimination tree (the level schedule is used then). */
int32_t nep_lu_analyze(int64_t n, int32_t csc, const int32_t* hLp, const int32_t* hLi, const int32_t* hUp,
                       const int32_t* hUi, int64_t out[8]);
/* X = A^{-1} B for nrhs right-hand sides; dB, dX: n x nrhs column-major; dX may alias dB.
 * scale is applied to the result (iar/tiar use -1: y = -lin_solve(...), src/method_iar.jl:103). */
int32_t nep_lu_solve(nep_lu* lu, int32_t nrhs, const nep_cdouble* dB, int64_t ldb, nep_cdouble* dX,
                     int64_t ldx, double scale, nep_stream stream);

/* X = scale * (Add + A^{-1} B): the update step of the iterative refinement (x + A^{-1} r, UMFPACK's solve behind
 * src/LinSolvers.jl:114-122) fused into the output permutation; dAdd may alias dX, NULL means zero. */
int32_t nep_lu_solve_add(nep_lu* lu, int32_t nrhs, const nep_cdouble* dB, int64_t ldb, const nep_cdouble* dAdd,
                         int64_t ldadd, nep_cdouble* dX, int64_t ldx, double scale, nep_stream stream);

/* ---- small BLAS-1 style helpers used by the drivers ------------------------------------ */
/* iar basis step (src/method_iar.jl:100-101,105): dst[(j+1)*n + r] = src[j*n + r]/(j+1), j=0..k-1 */
int32_t nep_iar_shift_scale(int64_t n, int32_t k, const nep_cdouble* dsrc, nep_cdouble* ddst,
                            nep_stream stream);
/* NLEIGS continuation-vector helpers (src/method_nleigs.jl:399-518, backslash):
 *   nep_rk_bw:       Bw[0:n] = 0; Bw[i n + r] = wc[(i-1) n + r] + c[i-1]*wc[i n + r], i = 1..N     (:418-435)
 *   nep_block_recur: x_i = a[i-1]*y_i + b[i-1]*x_{i-1}, i = 1..N, blocks of n entries, x_0 given   (:445-487, :496-515)
 * h_c, h_a, h_b are host arrays of N complex coefficients. */
int32_t nep_rk_bw(int64_t n, int32_t N, const nep_cdouble* dwc, const nep_cdouble* h_c, nep_cdouble* dBw,
                  nep_stream stream);
int32_t nep_block_recur(int64_t n, int32_t N, const nep_cdouble* h_a, const nep_cdouble* h_b, const nep_cdouble* dy,
                        nep_cdouble* dx, nep_stream stream);
/* HOST function (no device work): the boundary points of src/rk_helper/discretizepolygon.jl for a polygon of nz >= 3 vertices --
 * npts points at equal arc length, walked point by point exactly as the reference does (alph += remL / d per point; the Leja-Bagby
 * selection of nleigs takes an argmax over these candidates, so the walk is reproduced operation by operation, without contraction
 * into fused multiply-adds: bit-identical to the interpreted walk it replaces -- 10 000 iterations, 8 ms of a nleigs call).
 * h_z: nz vertices; h_out: npts points (the vertices the reference appends behind them are the caller's to add). */
int32_t nep_discretize_polygon(int32_t nz, const nep_cdouble* h_z, int32_t npts, nep_cdouble* h_out);
/* y += alpha * x   (len complex entries); K8 quadrature accumulation
 * src/method_contour_common.jl:88-90 (S[:,:,j] += temp*G[i,j]) */
int32_t nep_axpy(int64_t len, nep_cdouble alpha, const nep_cdouble* dx, nep_cdouble* dy,
                 nep_stream stream);
/* x *= alpha */
int32_t nep_scal(int64_t len, nep_cdouble alpha, nep_cdouble* dx, nep_stream stream);
/* ||x||_2 of len complex entries (synchronous) */
int32_t nep_nrm2(int64_t len, const nep_cdouble* dx, double* h_out, nep_stream stream);
/* column norms of a column-major rows x k block (synchronous) */
int32_t nep_colnorms(int64_t rows, int32_t k, const nep_cdouble* dX, int64_t ldx, double* h_out,
                     nep_stream stream);
/* dot products d_j = x_j^H y_j of the columns of two rows x k blocks (synchronous) */
int32_t nep_coldots(int64_t rows, int32_t k, const nep_cdouble* dX, int64_t ldx,
                    const nep_cdouble* dY, int64_t ldy, nep_cdouble* h_out, nep_stream stream);
/* same without conjugation, d_j = x_j^T y_j: the bilinear sums `mat_sum` of the infinite Lanczos three-term recurrence
 * for symmetric NEPs (src/method_ilan.jl:299-308) */
int32_t nep_coldotsu(int64_t rows, int32_t k, const nep_cdouble* dX, int64_t ldx,
                     const nep_cdouble* dY, int64_t ldy, nep_cdouble* h_out, nep_stream stream);
/* y = d .* (A^H x), result on the device (no synchronisation); A: rows x k column-major, d: k entries or NULL.
 * replaces: R(nep, Rinv(nep, v) ./ coeffs) -- the boundary-operator inverse of the waveguide problem,
 *           src/gallery_extra/waveguide/Waveguide.jl:159-170,270-294 (dense scaled-DFT matrix instead of FFTs), and
 *           `alpha = M\b` of the Sylvester-SMW preconditioner, waveguide_preconditioner.jl:378 (M^{-1} precomputed). */
int32_t nep_gemv_hd(const nep_cdouble* dA, int64_t lda, int64_t rows, int32_t k, const nep_cdouble* dx,
                    const nep_cdouble* dd, nep_cdouble* dy, nep_stream stream);
/* out[r] = sum_j A[r,j]*B[r,j] (column-major blocks, no conjugation) */
int32_t nep_rowdot(int64_t rows, int32_t k, const nep_cdouble* dA, int64_t lda, const nep_cdouble* dB, int64_t ldb,
                   nep_cdouble* dout, nep_stream stream);
/* A[r,j] *= B[r,j] (column-major blocks) */
int32_t nep_hadamard(int64_t rows, int32_t k, nep_cdouble* dA, int64_t lda, const nep_cdouble* dB, int64_t ldb,
                     nep_stream stream);
/* column 2-norms of a ROW-major rows x k block (row stride ld); synchronous */
int32_t nep_rowmajor_colnorms(int64_t rows, int32_t k, const nep_cdouble* dXT, int64_t ld, double* h_out,
                              nep_stream stream);
/* out-of-place transpose: row-major (rows x k, ld lds) -> column-major (ldd), selected columns
 * cols[0..ncols) (host int32, NULL = all) */
int32_t nep_rowmajor_to_colmajor(int64_t rows, int32_t k, const nep_cdouble* dsrc, int64_t lds,
                                 const int32_t* h_cols, int32_t ncols, nep_cdouble* ddst,
                                 int64_t ldd, nep_stream stream);

/* ---- waveguide preconditioner: Sylvester solve and region operators --------------------------------------
 * replaces: solve_wg_sylvester_fft! src/gallery_extra/waveguide/waveguide_preconditioner.jl:120-219 (FFT along z, sine
 *           transform along x) and the region means / expansions of solve_smw :263-304,:382-412.
 * nep_wep_sylv: X <- solution of A(sigma) X + X B = X in place (nz x nx, column-major): prime-factor DFT along z (dense small
 * DFTs out of LDS), one tridiagonal solve along x per z-mode (d_i I + B, B = tridiag(1,-2,1)*b) by in-wave scans, inverse
 * DFT.  h_d: the nz eigenvalues d_i of the z-operator in the basis F^H X (including sigma^2 + k_bar).  nx <= 2048. */
typedef struct nep_wep_sylv nep_wep_sylv;
int32_t nep_wep_sylv_create(int32_t nz, int32_t nx, const nep_cdouble* h_d, double b, nep_wep_sylv** out);
int32_t nep_wep_sylv_destroy(nep_wep_sylv* s);
int32_t nep_wep_sylv_info(const nep_wep_sylv* s, int32_t out[4]);   /* N1, N2 (nz = N1 N2 coprime), columns per workgroup, x per lane */
int32_t nep_wep_sylv_solve(nep_wep_sylv* s, nep_cdouble* dX, nep_stream stream);
/* boundary operator of the waveguide: dOut (2 nz) = blkdiag(R, R) diag(d_sinv) blkdiag(R, R)^H dX with R x = reverse(bb .* fft(x))
 * (Waveguide.jl:53-65) and d_sinv = 1 / (nz s_j(lam)) (P_inv_m / P_inv_p, Waveguide.jl:159-170): two prime-factor DFTs per half
 * in ONE launch (replaces four dense nz x nz GEMVs).  dOut may alias dX. */
typedef struct nep_wep_pinv nep_wep_pinv;
int32_t nep_wep_pinv_create(int32_t nz, const nep_cdouble* h_bb, nep_wep_pinv** out);
int32_t nep_wep_pinv_destroy(nep_wep_pinv* p);
int32_t nep_wep_pinv_apply(nep_wep_pinv* p, const nep_cdouble* d_sinv, const nep_cdouble* dX, nep_cdouble* dOut, nep_stream stream);
/* Schur complement of the waveguide applied matrix-free (SchurMatVec, /root/reference/src/gallery_extra/waveguide/Waveguide.jl:394-425):
 * dOut (nz nx) = vec(A(lam) X + X B + K .* X) - C1 P(lam)^{-1} C2T v for X = reshape(dV, nz, nx).  The interior operator is a
 * five-point stencil: weights cp / cm on the z + 1 / z - 1 neighbours (periodic), cx on the x - 1 / x + 1 neighbours (Dirichlet), the
 * diagonal dD0 (nz nx entries: K + lam^2 - 2/hz^2 - 2/hx^2); C2T v = d1 X[:, 0] + d2 X[:, 1] (and the mirror image at the other
 * end), C1 = c1s times the first / last column (generate_fd_boundary_mat).  dP: 2 nz entries of work.  dV and dOut must differ. */
int32_t nep_wep_schur_matvec(nep_wep_pinv* p, const nep_cdouble* d_sinv, int32_t nx, const nep_cdouble* dV, const nep_cdouble* dD0,
                             nep_cdouble cp, nep_cdouble cm, double cx, double d1, double d2, double c1s, nep_cdouble* dP,
                             nep_cdouble* dOut, nep_stream stream);
/* Sylvester-SMW matrix (generate_smw_matrix, waveguide_preconditioner.jl:221-313): column kappa of dM (mm x mm column-major,
 * mm = N (N+4)) = region means of Linv(E_kappa); the caller adds the identity and inverts.  All mm columns in one call (1517
 * Sylvester solves at N = 37).  dWork: nz*nx + 4 nz + mm complex. */
int32_t nep_wep_smw_matrix(nep_wep_sylv* s, nep_wep_pinv* p, int32_t N, const nep_cdouble* dKsc, double dd1, double dd2,
                           const nep_cdouble* d_sinv, nep_cdouble* dWork, nep_cdouble* dM, nep_stream stream);
/* The same matrix in mode space (column kappa = G S(Tsolve(F^H E_kappa)), dG as in nep_wep_smw_apply): no back transform, the
 * interior-region columns batched and restricted to the L grid columns of their region.  NEP_ERR_UNSUPPORTED like nep_wep_smw_apply. */
int32_t nep_wep_smw_matrix_modes(nep_wep_sylv* s, nep_wep_pinv* p, int32_t N, const nep_cdouble* dKsc, double dd1, double dd2,
                                 const nep_cdouble* d_sinv, const nep_cdouble* dG, nep_cdouble* dM, nep_stream stream);
/* One application of the Sylvester-SMW preconditioner, in place on dR (nz nx) (solve_smw, waveguide_preconditioner.jl:323-421):
 * dR <- Linv dR - Linv(sum_k alpha_k E_k), alpha = M^{-1} f(Linv dR).  Three transforms instead of four: the region means f are
 * taken in mode space (f = G S, S = x-region sums of the tridiagonal solutions) and the second solve is subtracted in mode space;
 * the expansion sum_k alpha_k E_k is formed inside the transform's loader from dKsc.  dMinvH: (M^{-1})^H, mm x mm column-major;
 * dG (N x nz, row rz at dG + rz nz): G[rz, i] = mean over the z of region rz of exp(-2 pi i z i / nz) / sqrt(nz).
 * NEP_ERR_UNSUPPORTED when nz has no odd coprime factorisation that fits the symmetric-half DFT kernel (use the pieces above). */
int32_t nep_wep_smw_apply(nep_wep_sylv* s, nep_wep_pinv* p, int32_t N, const nep_cdouble* dKsc, double dd1, double dd2,
                          const nep_cdouble* d_sinv, const nep_cdouble* dMinvH, const nep_cdouble* dG, nep_cdouble* dR,
                          nep_stream stream);
/* dOut (N x (N+4), column-major) = means of X over the N x (N+4) regions (interior regions L x L with L = nz/N, the four
 * boundary columns of X are regions of their own); needs nx = nz + 4 */
int32_t nep_wep_region_means(int32_t nz, int32_t nx, int32_t N, const nep_cdouble* dX, nep_cdouble* dOut, nep_stream stream);
/* dY[z, x] = alpha[region(z), region(x)] * Ksc[z, x];  dEb (nz x 2): dd1/dd2 combinations of the boundary region columns */
int32_t nep_wep_region_expand(int32_t nz, int32_t nx, int32_t N, const nep_cdouble* dAlpha, const nep_cdouble* dKsc, double dd1,
                              double dd2, nep_cdouble* dY, nep_cdouble* dEb, nep_stream stream);

/* ---- numeric LU factorisation on the device for a known pattern ----------------------------------------------
 * replaces: the numeric phase of `lu(A)` / `factorize` behind FactorizeLinSolver (src/LinSolvers.jl:109-122) for the second
 *           and later matrices of one sparsity pattern: the N same-pattern factorisations of contour_beyn
 *           (src/method_beyncontour.jl:89-94), the shifts of nleigs, repeated solves of one problem.  The first matrix of a
 *           pattern is factorised on the host (ordering, pivot sequence, fill pattern); nep_lu_refac_create enumerates the
 *           updates of the right-looking factorisation for that pivot sequence once, nep_lu_factor_dev computes the values of
 *           L and U on the GPU (static pivoting, as KLU / PARDISO refactorisation) and returns an ordinary nep_lu handle.
 * ref: a handle created by nep_lu_create_csc from (Lp, Li, Up, Ui, perm_r, perm_c); Ap / Ai: CSC pattern of the matrices to
 * come (caller's numbering), perm_r[i] / perm_c[j]: position of row i / column j of A in the factored matrix.
 * NEP_ERR_UNSUPPORTED: level-schedule handle, or the stored pattern is not closed under the elimination.
 * nep_lu_factor_dev: h_Ax = the nnz(A) values in the order of (Ap, Ai); growth_limit: largest accepted value of BOTH max |L|
 * (|Re|+|Im|; a diagonally pivoted factor stays near 1) and the element growth max|U| / max|A| (the stored pivot sequence may
 * not suit the new values; growth in U is what bounds the backward error of a static-pivot factorisation);
 * h_health[3] (may be NULL): [0] pivot breakdown flag, [1] max |L|, [2] max|U| / max|A|; h_LUx_out (may be NULL): the nnz(L) + nnz(U)
 * computed values in the input entry order (tests).  NEP_ERR_SINGULAR: breakdown / growth -- factorise on the host instead. */
typedef struct nep_lu_refac nep_lu_refac;
int32_t nep_lu_refac_create(nep_lu* ref, int64_t n, const int32_t* Lp, const int32_t* Li, const int32_t* Up, const int32_t* Ui,
                            const int32_t* perm_r, const int32_t* perm_c, const int32_t* Ap, const int32_t* Ai,
                            nep_lu_refac** out);
int32_t nep_lu_refac_destroy(nep_lu_refac* r);
/* Host-only analysis of a plan (no device is touched): symbolic partition + the complete enumeration of nep_lu_refac_create with
 * the plan arrays hashed instead of uploaded.  out[0] = products, [1] internal, [2] external, [3] external destination segments,
 * [4] wide, [5] wide pivot steps, [6] levels, [7] hash of the plan arrays (independent of NEP_LU_PLAN_THREADS).  Sanitizer
 * build (tests/sanitize) and thread-count invariance test. */
int32_t nep_lu_refac_analyze(int64_t n, const int32_t* Lp, const int32_t* Li, const int32_t* Up, const int32_t* Ui,
                             const int32_t* perm_r, const int32_t* perm_c, const int32_t* Ap, const int32_t* Ai, int64_t out[8]);
/* out[0..5] = n, products, internal products, external products, external destination segments, symbolic time in ms */
int32_t nep_lu_refac_info(const nep_lu_refac* r, int64_t out[6]);
/* The "wide" levels (top of the elimination tree: few blocks, one long chain of pivots each) are factorised in panels of P
 * consecutive pivots per launch (round 4; NEP_LU_WIDE_P = 1..4, default 4; 1 = one launch per pivot step as before):
 * out[0] = P, [1] = pivot steps of the wide levels, [2] = launches they take per factorisation, [3] = destination records,
 * [4] = deferred products (updates into a panel's own later rows / columns, applied at the end of the level). */
int32_t nep_lu_refac_wide_info(const nep_lu_refac* r, int64_t out[5]);
/* out[0] = hash of the plan arrays as they sit on the device (the quantity nep_lu_refac_analyze returns in out[7] for the host
 * enumeration of the same inputs), out[1] = 1 when the products were enumerated on the device (default; NEP_LU_PLAN_GPU=0: on
 * host threads).  Round 3: the enumeration of nep_lu_refac_create runs on the GPU (0.10-0.18 s -> a few ms for the gun pattern). */
int32_t nep_lu_refac_hash(const nep_lu_refac* r, int64_t out[2]);
int32_t nep_lu_factor_dev(nep_lu_refac* r, const nep_cdouble* h_Ax, int32_t expected_solves, double growth_limit,
                          double* h_health, nep_cdouble* h_LUx_out, nep_lu** out, nep_stream stream);
/* B matrices of the plan's pattern in one pass (every launch carries all of them): h_Ax B x nnz(A), h_health B x 3 (required),
 * out[b] = NULL for a matrix whose factorisation was refused -- factorise that one on the host. */
int32_t nep_lu_factor_dev_batch(nep_lu_refac* r, int32_t B, const nep_cdouble* h_Ax, int32_t expected_solves, double growth_limit,
                                double* h_health, nep_cdouble* h_LUx_out, nep_lu** out, nep_stream stream);
/* The same for matrices that are combinations of `mt` terms on the plan's pattern, A_b = sum_t h_Cf[b*mt + t] A_t
 * (M(lam_b) = sum_t f_t(lam_b) A_t at the quadrature nodes of src/method_beyncontour.jl:89-94): d_D is a DEVICE array, nnz(A) x mt
 * entry-major, with the values of every term scattered onto the union pattern (uploaded once per NEP); the B x nnz(A) value
 * block is formed inside the scatter kernel instead of on the host. */
int32_t nep_lu_factor_dev_batch_terms(nep_lu_refac* r, int32_t B, const nep_cdouble* d_D, int32_t mt, const nep_cdouble* h_Cf,
                                      int32_t expected_solves, double growth_limit, double* h_health, nep_lu** out,
                                      nep_stream stream);

/* the library's own device-wide primitives (csrc/devprims.h: reduce-then-scan exclusive prefix sum; stable LSD radix sort, 8-bit
 * digits, wave-ballot ranking) that the one-off plan enumeration of nep_lu_refac_create runs on -- no hipCUB / rocPRIM behind the
 * ABI.  d_out[i] = sum_{j<i} d_in[j]; (d_keys, d_vals) sorted in place by the key bits [0, nbits), equal keys keep their order.
 * Asynchronous.  Entry points for tests and diagnostics. */
int32_t nep_devprim_exclusive_sum(const uint64_t* d_in, uint64_t* d_out, int64_t n, nep_stream stream);
int32_t nep_devprim_sort_pairs(uint64_t* d_keys, uint64_t* d_vals, int64_t n, int32_t nbits, nep_stream stream);

/* ---- one infinite-Arnoldi step as one call ------------------------------------------------------
 * replaces: the loop body of iar between two eigenvalue checks, src/method_iar.jl:94-109
 *           (compute_Mlincomb! at sigma through the DerSPMF table :1130-1160, lin_solve, the 1/j shift of the block
 *           structure, orthogonalize_and_normalize!).  Sequences nep_mlincomb_dev, nep_lu_solve[_add] (+ refine_steps blind
 *           refinement steps on nep_cw_backward_error's residual), nep_iar_shift_scale and nep_orth_dev, then copies row k-1 of
 *           the device H block ((m+4) complex per row: h[0..k), beta, flags, then 4 doubles = UMFPACK's componentwise backward
 *           error omega of the refinement iterates x_0..x_refine_steps when h_cabs/h_cf were given) to the caller's PINNED host
 *           block behind an event.  Nothing waits for the device; nep_iar_wait(k) blocks until H's column k has arrived; the
 *           host replays UMFPACK's stopping rule on the recorded omegas instead of reading them back inside the step.
 *           dH must be zero-filled by the caller (m rows of m+4).
 * dV: the basis, column j at dV + j*ldv (ldv >= n(m+1), zero-initialised, column 0 = start vector); dCtab: the DerSPMF
 * coefficient table (mt columns of ldc >= m entries, row j-1 = alpha_j/j f^(j)(sigma)); d_active: device int64[m+1], active
 * rows per basis column; dwork3n: 3n complex of scratch; h_cabs / h_cf: |f_t(sigma)| and f_t(sigma) for the refinement
 * (may be NULL when refine_steps is always 0); orth_method: 0 DGKS, 1 CGS. */
typedef struct nep_iar nep_iar;
int32_t nep_iar_create(nep_spmf* spmf, nep_lu* lu, int64_t n, int32_t m, nep_cdouble* dV, int64_t ldv,
                       const nep_cdouble* dCtab, int64_t ldc, const int64_t* d_active, nep_cdouble* dwork3n,
                       const double* h_cabs, const nep_cdouble* h_cf, int32_t mt, nep_cdouble* dH, nep_cdouble* h_pinnedH,
                       int32_t orth_method, nep_iar** out);
int32_t nep_iar_destroy(nep_iar* s);
/* refine_steps | 0x100: the backward error of the KEPT iterate is recorded only in steps whose index is a multiple of 8 (its slot
 * stays 0 otherwise) -- for callers whose refinement count has settled (the check costs one pass over the matrices per step) */
int32_t nep_iar_step(nep_iar* s, int32_t k, int32_t refine_steps, nep_stream stream);
/* steps k0 .. k0+count-1 (same refine_steps) in one foreign call */
int32_t nep_iar_steps(nep_iar* s, int32_t k0, int32_t count, int32_t refine_steps, nep_stream stream);
int32_t nep_iar_wait(nep_iar* s, int32_t k);
/* device-side counterpart of nep_iar_wait: work enqueued on `stream` after this call starts only when column k of H is
 * complete (the eigen-decomposition of step k on a second stream: nep_hess_eigvals_dev on row block dH of nep_iar_create) */
int32_t nep_iar_stream_wait(nep_iar* s, int32_t k, nep_stream stream);

/* ---- the whole infinite-Arnoldi run as one call ---------------------------------------------
 * replaces: the body of `iar(::Type{T}, nep::NEP; orthmethod, maxit, linsolvercreator, tol, neigs, errmeasure, sigma, gamma, v, logger,
 *           check_error_every, ...)`, src/method_iar.jl:46-182, after `create_linsolver` (:84): recurrence (:94-109, nep_iar_step),
 *           `eigen(H[1:k,1:k])` (:112, nep_hess_eig*_batch_dev), `Q = VV*Z` (:115, nep_gemm_ts_dev), `estimate_error` for every Ritz
 *           pair (:133-135, nep_resid_batch_dev; StandardSPMFErrmeasure src/errmeasure.jl:174-190 / ResidualErrmeasure :114,128-130),
 *           convergence count, sort and extraction (:137-160), the NoConvergenceException (:163-175).  A Julia method
 *           `iar(::Type{T}, nep::DeviceSPMF; ...)` is one ccall of this (julia/NEPMI355X.jl); the Python host calls it too.
 * lu: the factors of M(sigma) (nep_lu_create_csc / nep_lu_factor_dev); h_v0: the start vector (n, not normalised);
 * h_Ctab: m x mt column-major, row j-1 = gamma^j / j * f_t^(j)(sigma) (the DerSPMF table src/NEPTypes.jl:1108-1128 with iar's
 * alpha_j / j scaling :83,101); h_cabs / h_cf: |f_t(sigma)|, f_t(sigma) (required when umfpack_refinements > 0);
 * h_fro: ||A_t||_F (errmeasure 1); fv(ctx, nlam, lam, F): the host evaluates F[t + s mt] = f_t(lam[s]) (the scalar functions of
 * the SPMF are closures of the host language), returns 0 -- called on the calling thread only.
 * Results: h_lam (capacity maxit), dQ (device, n x maxit column-major, ld n; may be NULL), h_Q (host, same shape; may be NULL:
 * pinned memory gets the link's rate), h_err (maxit x maxit column-major, err[k-1 + (s-1) maxit] = sorted error s of step k, NaN
 * where unset; may be NULL), dV_basis (device, maxit+1 columns of leading dimension n (maxit+1): the Krylov basis the reference
 * returns as V[:,1:k]; NULL: library-owned and released at return).  NEP_OK: res->nret converged pairs (all pairs below tol when neigs = INFINITY); NEP_ERR_NOCONV: the
 * res->nret best pairs; NEP_ERR_RETRY / NEP_ERR_UNSUPPORTED (maxit > 128): nothing returned, use the per-step entry points. */
typedef struct nep_iar_opts {
    int32_t maxit;               /* m */
    int32_t check_error_every;
    int32_t orth_method;         /* 0 DGKS, 1 classical Gram-Schmidt */
    int32_t umfpack_refinements; /* <= 0: plain solves; else UMFPACK's refinement rule with this bound (control[8]) */
    int32_t errmeasure;          /* 0: ||M(lam) v|| / ||v||;  1: that / sum_t ||A_t||_F |f_t(lam)| */
    int32_t refine_hint;         /* sweeps a previous run at THIS shift settled on (nep_iar_result.refine_plan), -1: none */
    double tol;
    double neigs;                /* INFINITY: run to maxit, return every pair below tol */
    nep_cdouble sigma, gamma;
} nep_iar_opts;
typedef struct nep_iar_result {
    int32_t k;                   /* step whose check produced the returned pairs */
    int32_t nconv;               /* pairs below tol at that step */
    int32_t nret;                /* pairs written to h_lam / dQ / h_Q */
    int32_t refine_plan;         /* refinement sweeps the run settled on (-1: none) -- the next run's refine_hint */
    int32_t refine_hint_off;     /* 1: a miss withdrew the hint for this NEP */
    int32_t retry_reason;        /* NEP_ERR_RETRY: 1 refinement record, 2 DGKS pass, 3 device eigen-decomposition */
} nep_iar_result;
typedef int32_t (*nep_fv_eval)(void* ctx, int32_t nlam, const nep_cdouble* lam, nep_cdouble* F);
/* UMFPACK's refinement stopping rule (umfpack_solve behind `Afact \ x` with control[8] = umfpack_refinements, src/LinSolvers.jl:
 * 114-122) replayed on the omegas w4[0..plan] that nep_iar_step recorded for a solve that took `plan` sweeps without reading them
 * back (final_recorded = 0: omega of the kept iterate x_plan was not evaluated).  out[0] = 1: what the step kept is what the checked
 * loop keeps (or at least as good); 0: a miss -- re-run with checked solves.  out[1] = sweeps to plan from now on (-1: unchanged),
 * out[2] = hint for later solvers at this shift (-1: none), out[3] = 1: the hint is withdrawn for good. */
int32_t nep_refine_review(int32_t umfpack_refinements, int32_t plan, int32_t final_recorded, const double* w4, int32_t hint_in,
                          int32_t out[4]);
int32_t nep_iar_run(nep_spmf* spmf, nep_lu* lu, int64_t n, const nep_iar_opts* opts, const nep_cdouble* h_v0,
                    const nep_cdouble* h_Ctab, int32_t mt, const double* h_cabs, const nep_cdouble* h_cf, const double* h_fro,
                    nep_fv_eval fv, void* ctx, nep_cdouble* h_lam, nep_cdouble* dQ, nep_cdouble* h_Q, double* h_err,
                    nep_cdouble* dV_basis, nep_iar_result* res, nep_stream stream);

/* ---- multi-GPU exchange of the contour integrators ----------------------------------------
 * replaces: the reduction inside `integrate_interval(::Type{<:MatrixIntegrator}, ...)` src/method_contour_common.jl:46,61-94
 *           when the N quadrature nodes of contour_beyn / contour_block_SS (src/method_beyncontour.jl:89-104,
 *           src/method_block_SS.jl:81-86,129-135) are sharded over the GPUs of one node: rank r owns the nodes
 *           i = r (mod P) and accumulates their moments locally; the only exchange is one all-gather of the partial
 *           moment block followed by a sum in fixed rank order (bit-identical result on every rank).
 * One process per GPU.  Rank 0 obtains a 128-byte unique id and hands it to the other ranks out of band (MPI.jl bcast,
 * torch.distributed, a shared file); every rank then creates its communicator on ITS current device (collective call).
 * RCCL (librccl.so) is loaded on first use. */
int32_t nep_comm_unique_id(void* h_out128);
int32_t nep_comm_create(int32_t rank, int32_t world, const void* h_unique_id128, nep_comm** out);
int32_t nep_comm_destroy(nep_comm* c);
int32_t nep_comm_info(const nep_comm* c, int32_t out[2]);     /* out[0] = rank, out[1] = world */
/* d_total[i] = sum_{r=0}^{world-1} partial_r[i], i < len (complex128), on every rank; ncclAllGather over xGMI into a
 * library-owned world x len block + one summation kernel, all on `stream` (asynchronous).  d_total may alias
 * d_partial. */
int32_t nep_allgather_sum(nep_comm* c, const nep_cdouble* d_partial, int64_t len, nep_cdouble* d_total, nep_stream stream);
/* the reduction half of nep_allgather_sum on a caller-provided gather buffer (world x len complex128, rank r's block at
 * r * len): d_total[i] = sum_r d_parts[r * len + i] in rank order r = 0 .. world-1 (bit-identical wherever it runs);
 * d_total may alias block 0.  For tests and for hosts that move the blocks themselves. */
int32_t nep_sum_ranks(const nep_cdouble* d_parts, int64_t len, int32_t world, nep_cdouble* d_total, nep_stream stream);

#ifdef __cplusplus
}
#endif
#endif /* NEPMI355_H */
