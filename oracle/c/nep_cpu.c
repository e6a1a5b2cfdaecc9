/* CPU restatement (plain C, single thread) of the reference's compute_Mlincomb for an SPMF at a
 * point where the derivative coefficients are known -- TEST/BASELINE INFRASTRUCTURE ONLY.
 *
 * Follows the operation structure of src/NEPTypes.jl:1000-1010 (compute_Mlincomb!(::SPMF_NEP)):
 *   for every term i:   VFi1 = V * Fi1          (dense gemv, streams V once PER TERM, :1006)
 *                       z   += A[i] * VFi1      (SparseArrays CSC mat-vec = column scatter, :1007)
 * Matrices are CSC with 0-based int32 indices and real (double) values, V is column-major
 * complex128, C[:,i] is the coefficient vector Fi1 of term i.  Used by bench.py (cpu_baseline) and
 * tests/test_oracle_kat.py::test_c_port_of_compute_Mlincomb; never by the product.
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

/* All-cores variant (OpenMP) of the same operation, SURVEY.md section 8d "CPU baseline": row-parallel formulation --
 * every thread owns a contiguous row range, forms W[r, i] = sum_j V[r, j] C[j, i] for its rows and then the CSR products
 * z[r] = sum_i sum_e A_i[r, col_e] W[col_e, i].  The matrices come in CSR (the transposed layout of the routine above)
 * because a CSC scatter has write conflicts between threads.  W: n x mt column-major scratch. */
#ifdef _OPENMP
#include <omp.h>
#endif
void ref_mlincomb_csr_omp(int64_t n, int32_t mt, const int32_t* const* rowptr, const int32_t* const* colind,
                          const double* const* nzval, int32_t k, const cd* C /* k x mt col-major */, const cd* V,
                          int64_t ldv, cd* z, cd* W /* n x mt */) {
#pragma omp parallel
    {
        int nt = 1, tid = 0;
#ifdef _OPENMP
        nt = omp_get_num_threads(); tid = omp_get_thread_num();
#endif
        const int64_t r0 = n * tid / nt, r1 = n * (tid + 1) / nt;     /* this thread's rows: column sweeps like zgemv 'N' */
        for (int32_t i = 0; i < mt; ++i) {
            cd* w = W + (int64_t)i * n;
            for (int64_t r = r0; r < r1; ++r) { w[r].re = 0.0; w[r].im = 0.0; }
            for (int32_t j = 0; j < k; ++j) {
                const cd c = C[j + (int64_t)i * k];
                if (c.re == 0.0 && c.im == 0.0) continue;
                const cd* v = V + (int64_t)j * ldv;
                for (int64_t r = r0; r < r1; ++r) {
                    w[r].re += v[r].re * c.re - v[r].im * c.im;
                    w[r].im += v[r].re * c.im + v[r].im * c.re;
                }
            }
        }
#pragma omp barrier
#pragma omp for schedule(static)
        for (int64_t r = 0; r < n; ++r) {
            double sr = 0.0, si = 0.0;
            for (int32_t i = 0; i < mt; ++i) {
                const int32_t* rp = rowptr[i]; const int32_t* ci = colind[i]; const double* nz = nzval[i];
                const cd* w = W + (int64_t)i * n;
                for (int32_t e = rp[r]; e < rp[r + 1]; ++e) { sr += nz[e] * w[ci[e]].re; si += nz[e] * w[ci[e]].im; }
            }
            z[r].re = sr; z[r].im = si;
        }
    }
}
int32_t ref_omp_threads(void) {
#ifdef _OPENMP
    return (int32_t)omp_get_max_threads();
#else
    return 1;
#endif
}
