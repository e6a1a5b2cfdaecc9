/* CPU restatement (plain C, single thread) of the reference's compute_Mlincomb for an SPMF at a
 * point where the derivative coefficients are known -- TEST/BASELINE INFRASTRUCTURE ONLY.
 *
 * Follows the operation structure of src/NEPTypes.jl:1000-1010 (compute_Mlincomb!(::SPMF_NEP)):
 *   for every term i:   VFi1 = V * Fi1          (dense gemv, streams V once PER TERM, :1006)
 *                       z   += A[i] * VFi1      (SparseArrays CSC mat-vec = column scatter, :1007)
 * Matrices are CSC with 0-based int32 indices and real (double) values, V is column-major
 * complex128, C[:,i] is the coefficient vector Fi1 of term i.  Used by bench.py (cpu_baseline) and
 * tests/test_oracle_c.py; never by the product.
 */
#include <stdint.h>
#include <string.h>

typedef struct { double re, im; } cd;

void ref_mlincomb_csc(int64_t n, int32_t mt, const int32_t* const* colptr, const int32_t* const* rowval,
                      const double* const* nzval, int32_t k, const cd* C /* k x mt col-major */,
                      const cd* V, int64_t ldv, cd* z, cd* work /* n */) {
    memset(z, 0, (size_t)n * sizeof(cd));
    for (int32_t i = 0; i < mt; ++i) {
        /* VFi1 = V * C[:,i]  (gemv, column sweep like BLAS zgemv 'N') */
        memset(work, 0, (size_t)n * sizeof(cd));
        for (int32_t j = 0; j < k; ++j) {
            const cd c = C[j + (int64_t)i * k];
            if (c.re == 0.0 && c.im == 0.0) continue;
            const cd* v = V + (int64_t)j * ldv;
            for (int64_t r = 0; r < n; ++r) {
                work[r].re += v[r].re * c.re - v[r].im * c.im;
                work[r].im += v[r].re * c.im + v[r].im * c.re;
            }
        }
        /* z += A_i * VFi1  (CSC scatter form, as Julia's SparseArrays mul!) */
        const int32_t* cp = colptr[i]; const int32_t* rv = rowval[i]; const double* nz = nzval[i];
        for (int64_t col = 0; col < n; ++col) {
            const cd x = work[col];
            for (int32_t e = cp[col]; e < cp[col + 1]; ++e) {
                z[rv[e]].re += nz[e] * x.re;
                z[rv[e]].im += nz[e] * x.im;
            }
        }
    }
}

/* DGKS pass pieces (IterativeSolvers 0.9.2 orthogonalize_and_normalize!, BLAS-2 structure):
 * h = V^H w ; w -= V h ; returns ||w||^2.  V: rows x k column-major. */
double ref_gs_pass(int64_t rows, int32_t k, const cd* V, int64_t ldv, cd* w, cd* h) {
    for (int32_t j = 0; j < k; ++j) {
        const cd* v = V + (int64_t)j * ldv;
        double sr = 0.0, si = 0.0;
        for (int64_t r = 0; r < rows; ++r) {
            sr += v[r].re * w[r].re + v[r].im * w[r].im;
            si += v[r].re * w[r].im - v[r].im * w[r].re;
        }
        h[j].re = sr; h[j].im = si;
    }
    for (int32_t j = 0; j < k; ++j) {
        const cd* v = V + (int64_t)j * ldv;
        const cd c = h[j];
        for (int64_t r = 0; r < rows; ++r) {
            w[r].re -= v[r].re * c.re - v[r].im * c.im;
            w[r].im -= v[r].re * c.im + v[r].im * c.re;
        }
    }
    double nn = 0.0;
    for (int64_t r = 0; r < rows; ++r) nn += w[r].re * w[r].re + w[r].im * w[r].im;
    return nn;
}
