// K5 elimination-tree block schedule (trsv_ml.hip): internal interface used by the C ABI in trsv.hip
#pragma once
#include "common.h"

struct MLFactor;

// csc = 0: L and U in CSR; 1: in CSC.  NEP_ERR_UNSUPPORTED: the factors do not fit the block schedule (caller falls back
// to the level schedule).
int ml_create(int64_t n, int csc, const int32_t* Lp, const int32_t* Li, const nep_cdouble* Lx, const int32_t* Up,
              const int32_t* Ui, const nep_cdouble* Ux, const int32_t* perm_r, const int32_t* perm_c, int expected_solves,
              MLFactor** out);
int ml_analyze(int64_t n, int csc, const int32_t* Lp, const int32_t* Li, const int32_t* Up, const int32_t* Ui, int64_t out[8]);
int ml_refactor(MLFactor* F, const nep_cdouble* Lx, const nep_cdouble* Ux);
int ml_set_row_scale(MLFactor* F, const double* h_rs);
void ml_destroy(MLFactor* F);
void ml_info(const MLFactor* F, int64_t info[6], int64_t sched[8]);
int ml_solve(MLFactor* F, int nrhs, const nep_cdouble* dB, int64_t ldb, const nep_cdouble* dAdd, int64_t ldadd,
             nep_cdouble* dX, int64_t ldx, double scale, hipStream_t st);

// ---- hooks of the device-side numeric factorisation (csrc/lufac.hip): the symbolic part (partition, schedule) of an existing
// factor is shared; the new factor's values arrive in device arrays laid out like the input L / U of that factor
struct MLSym;
MLSym* ml_sym_acquire(MLFactor* F);                 // takes a reference
void ml_sym_release_ref(MLSym* S);
int ml_sym_build_host(int64_t n, const int32_t* Lp, const int32_t* Li, const int32_t* Up, const int32_t* Ui, const int32_t* perm_r,
                      const int32_t* perm_c, MLSym** out);      // host-only analysis of CSC factors (no device)
void ml_sym_free_host(MLSym* S);
// partition in the factor's input numbering: level and block of every pivot, the pivots in schedule order (oldof), the
// block boundaries in that order (blk_se[2k], blk_se[2k+1]) and the block range of every level (lev_blk[l], lev_blk[l+1])
void ml_sym_partition(const MLSym* S, int64_t* n, int* nlev, int* nblk, const int32_t** lvl, const int32_t** blk,
                      const int32_t** oldof, const int32_t** blk_se, const int32_t** lev_blk);
int ml_create_from_sym(MLSym* S, const nep_cdouble* d_Lx, const nep_cdouble* d_Ux, hipStream_t producer, int expected_solves,
                       MLFactor** out);
int ml_create_from_sym_batch(MLSym* S, int B, const nep_cdouble* const* d_Lx, const nep_cdouble* const* d_Ux, hipStream_t producer,
                             int expected_solves, MLFactor** out);
int ml_wait_ready(MLFactor* F, hipStream_t st);   // st waits for the numeric build of F
