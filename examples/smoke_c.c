/* Plain-C smoke program for libnepmi355.so: drives the hot path through the C ABI with no Python in the process --
 *   nep_spmf_create -> nep_mlincomb (K1) -> nep_lu_create_csc + nep_lu_solve (K5) -> ... -> nep_iar_run (the whole iar call, with a C callback)
 *   nep_spmf_create -> nep_mlincomb (K1) -> nep_lu_create_csc + nep_lu_solve (K5) -> nep_lu_refac_create + nep_lu_factor_dev
 *   (numeric LU on the device) -> nep_orth (K6) -> nep_gemm_ts (K7)
 * and checks every result against a few lines of host arithmetic.  Built by __graft_entry__.build():
 *   gcc -O2 -I include examples/smoke_c.c -o examples/smoke_c -L nonlineareigenproblems.jl_amd -lnepmi355 -lm
 * Exit code 0 = all checks passed (needs a GPU); 77 = no HIP device visible. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "nepmi355.h"

#define CHECK(call) do { int32_t rc_ = (call); if (rc_ != NEP_OK) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, nep_last_error()); return 1; } } while (0)

static double rnd(unsigned* s) { *s = *s * 1664525u + 1013904223u; return ((*s >> 8) & 0xffff) / 65536.0 - 0.5; }
static nep_cdouble cmul(nep_cdouble a, nep_cdouble b) { nep_cdouble r = {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; return r; }

/* f_t(lambda) of the SPMF M(lambda) = A0 - lambda A1 for nep_iar_run: F[t + s mt], t = 0: 1, t = 1: -lambda */
static int32_t fv_eval(void* ctx, int32_t nlam, const nep_cdouble* lam, nep_cdouble* F) {
    int* calls = (int*)ctx;
    ++*calls;
    for (int s = 0; s < nlam; ++s) {
        F[0 + 2 * s].re = 1.0; F[0 + 2 * s].im = 0.0;
        F[1 + 2 * s].re = -lam[s].re; F[1 + 2 * s].im = -lam[s].im;
    }
    return 0;
}

int main(void) {
    int32_t ndev = 0;
    CHECK(nep_device_count(&ndev));
    if (ndev < 1) { fprintf(stderr, "no HIP device\n"); return 77; }
    char name[128];
    CHECK(nep_device_name(name, sizeof name));
    const int n = 500, k = 3, mt = 2;
    unsigned seed = 12345u;

    /* ---- SPMF with two real terms: A0 = tridiag(-1, 2, -1), A1 = diag(1..n)/n  (CSR) */
    int32_t* rp0 = malloc((n + 1) * sizeof *rp0); int32_t* ci0 = malloc(3 * n * sizeof *ci0); double* v0 = malloc(3 * n * sizeof *v0);
    int32_t* rp1 = malloc((n + 1) * sizeof *rp1); int32_t* ci1 = malloc(n * sizeof *ci1); double* v1 = malloc(n * sizeof *v1);
    int nz = 0;
    for (int i = 0; i < n; ++i) {
        rp0[i] = nz;
        if (i > 0) { ci0[nz] = i - 1; v0[nz++] = -1.0; }
        ci0[nz] = i; v0[nz++] = 2.0;
        if (i < n - 1) { ci0[nz] = i + 1; v0[nz++] = -1.0; }
        rp1[i] = i; ci1[i] = i; v1[i] = (i + 1.0) / n;
    }
    rp0[n] = nz; rp1[n] = n;
    const int32_t* rps[2] = {rp0, rp1}; const int32_t* cis[2] = {ci0, ci1}; const void* vs[2] = {v0, v1};
    const int32_t iscomplex[2] = {0, 0};
    nep_spmf* spmf = NULL;
    CHECK(nep_spmf_create(n, mt, rps, cis, vs, iscomplex, &spmf));

    /* ---- K1: z = sum_i A_i (V C[:,i]) */
    nep_cdouble* V = malloc((size_t)n * k * sizeof *V); nep_cdouble Cm[3 * 2]; nep_cdouble* z = malloc(n * sizeof *z);
    for (int i = 0; i < n * k; ++i) { V[i].re = rnd(&seed); V[i].im = rnd(&seed); }
    for (int i = 0; i < k * mt; ++i) { Cm[i].re = rnd(&seed); Cm[i].im = rnd(&seed); }
    void *dV = NULL, *dz = NULL;
    CHECK(nep_dev_alloc(&dV, (size_t)n * k * sizeof *V)); CHECK(nep_dev_alloc(&dz, n * sizeof *z));
    CHECK(nep_upload(dV, V, (size_t)n * k * sizeof *V, NULL));
    CHECK(nep_mlincomb(spmf, k, Cm, dV, n, dz, NULL));
    CHECK(nep_download(z, dz, n * sizeof *z, NULL));
    double err = 0.0, nrm = 0.0;
    for (int r = 0; r < n; ++r) {
        nep_cdouble acc = {0.0, 0.0};
        for (int t = 0; t < mt; ++t) {
            const int32_t* rp = rps[t]; const int32_t* ci = cis[t]; const double* vv = (const double*)vs[t];
            for (int e = rp[r]; e < rp[r + 1]; ++e) {
                nep_cdouble w = {0.0, 0.0};
                for (int j = 0; j < k; ++j) { nep_cdouble p = cmul(V[ci[e] + (size_t)j * n], Cm[j + t * k]); w.re += p.re; w.im += p.im; }
                acc.re += vv[e] * w.re; acc.im += vv[e] * w.im;
            }
        }
        err += (acc.re - z[r].re) * (acc.re - z[r].re) + (acc.im - z[r].im) * (acc.im - z[r].im);
        nrm += acc.re * acc.re + acc.im * acc.im;
    }
    printf("K1 nep_mlincomb          rel err %.2e\n", sqrt(err / nrm));
    if (sqrt(err / nrm) > 1e-13) return 2;

    /* ---- K5: A = L U, L unit lower bidiagonal, U upper bidiagonal (CSC); solve A x = b */
    int32_t* Lp = malloc((n + 1) * sizeof *Lp); int32_t* Li = malloc(2 * n * sizeof *Li); nep_cdouble* Lx = malloc(2 * n * sizeof *Lx);
    int32_t* Up = malloc((n + 1) * sizeof *Up); int32_t* Ui = malloc(2 * n * sizeof *Ui); nep_cdouble* Ux = malloc(2 * n * sizeof *Ux);
    int ln = 0, un = 0;
    for (int j = 0; j < n; ++j) {
        Lp[j] = ln; Li[ln] = j; Lx[ln].re = 1.0; Lx[ln++].im = 0.0;
        if (j < n - 1) { Li[ln] = j + 1; Lx[ln].re = 0.3 * rnd(&seed); Lx[ln++].im = 0.3 * rnd(&seed); }
        Up[j] = un;
        if (j > 0) { Ui[un] = j - 1; Ux[un].re = 0.4 * rnd(&seed); Ux[un++].im = 0.4 * rnd(&seed); }
        Ui[un] = j; Ux[un].re = 2.0 + rnd(&seed); Ux[un++].im = rnd(&seed);
    }
    Lp[n] = ln; Up[n] = un;
    nep_lu* lu = NULL;
    CHECK(nep_lu_create_csc(n, Lp, Li, Lx, Up, Ui, Ux, NULL, NULL, &lu));
    nep_cdouble* b = malloc(n * sizeof *b); nep_cdouble* x = malloc(n * sizeof *x); nep_cdouble* y = malloc(n * sizeof *y);
    for (int i = 0; i < n; ++i) { b[i].re = rnd(&seed); b[i].im = rnd(&seed); }
    void *db = NULL, *dx = NULL;
    CHECK(nep_dev_alloc(&db, n * sizeof *b)); CHECK(nep_dev_alloc(&dx, n * sizeof *x));
    CHECK(nep_upload(db, b, n * sizeof *b, NULL));
    CHECK(nep_lu_solve(lu, 1, db, n, dx, n, 1.0, NULL));
    CHECK(nep_download(x, dx, n * sizeof *x, NULL));
    /* y = U x (column sweep), then r = L y - b */
    memset(y, 0, n * sizeof *y);
    for (int j = 0; j < n; ++j) for (int e = Up[j]; e < Up[j + 1]; ++e) { nep_cdouble p = cmul(Ux[e], x[j]); y[Ui[e]].re += p.re; y[Ui[e]].im += p.im; }
    nep_cdouble* r = calloc(n, sizeof *r);
    for (int j = 0; j < n; ++j) for (int e = Lp[j]; e < Lp[j + 1]; ++e) { nep_cdouble p = cmul(Lx[e], y[j]); r[Li[e]].re += p.re; r[Li[e]].im += p.im; }
    err = nrm = 0.0;
    for (int i = 0; i < n; ++i) { err += (r[i].re - b[i].re) * (r[i].re - b[i].re) + (r[i].im - b[i].im) * (r[i].im - b[i].im); nrm += b[i].re * b[i].re + b[i].im * b[i].im; }
    printf("K5 nep_lu_solve          ||LUx - b||/||b|| %.2e\n", sqrt(err / nrm));
    if (sqrt(err / nrm) > 1e-10) return 3;

    /* ---- numeric LU on the device for a known pattern: the plan from the factors above (identity permutations: A = L U is
     * tridiagonal), then a NEW matrix A2 = L2 U2 of the same pattern is factorised on the GPU from its values alone and solved */
    {
        int32_t* perm = malloc(n * sizeof *perm); int32_t* Ap = malloc((n + 1) * sizeof *Ap); int32_t* Ai = malloc(3 * n * sizeof *Ai);
        nep_cdouble* A2 = malloc(3 * n * sizeof *A2);
        nep_cdouble* l2 = malloc(n * sizeof *l2); nep_cdouble* u2 = malloc(n * sizeof *u2); nep_cdouble* s2 = malloc(n * sizeof *s2);
        for (int j = 0; j < n; ++j) {
            perm[j] = j;
            l2[j].re = 0.3 * rnd(&seed); l2[j].im = 0.3 * rnd(&seed);        /* L2[j+1][j] */
            s2[j].re = 0.4 * rnd(&seed); s2[j].im = 0.4 * rnd(&seed);        /* U2[j-1][j] */
            u2[j].re = 2.0 + rnd(&seed); u2[j].im = rnd(&seed);              /* U2[j][j]   */
        }
        int an = 0;
        for (int j = 0; j < n; ++j) {
            Ap[j] = an;
            if (j > 0) { Ai[an] = j - 1; A2[an++] = s2[j]; }
            Ai[an] = j; A2[an] = u2[j];
            if (j > 0) { nep_cdouble q = cmul(l2[j - 1], s2[j]); A2[an].re += q.re; A2[an].im += q.im; }
            ++an;
            if (j < n - 1) { Ai[an] = j + 1; A2[an++] = cmul(l2[j], u2[j]); }
        }
        Ap[n] = an;
        nep_lu_refac* plan = NULL;
        CHECK(nep_lu_refac_create(lu, n, Lp, Li, Up, Ui, perm, perm, Ap, Ai, &plan));
        double health[3] = {0.0, 0.0, 0.0};
        nep_cdouble* LU2 = malloc((size_t)(ln + un) * sizeof *LU2);
        nep_lu* lu2 = NULL;
        CHECK(nep_lu_factor_dev(plan, A2, 10, 1e6, health, LU2, &lu2, NULL));
        /* the factor values against the L2, U2 the matrix was built from (input entry order: L then U) */
        double ferr = 0.0;
        for (int j = 0; j < n; ++j) {
            for (int e = Lp[j]; e < Lp[j + 1]; ++e) {
                const nep_cdouble want = (Li[e] == j) ? (nep_cdouble){1.0, 0.0} : l2[j];
                ferr = fmax(ferr, hypot(LU2[e].re - want.re, LU2[e].im - want.im));
            }
            for (int e = Up[j]; e < Up[j + 1]; ++e) {
                const nep_cdouble want = (Ui[e] == j) ? u2[j] : s2[j];
                ferr = fmax(ferr, hypot(LU2[ln + e].re - want.re, LU2[ln + e].im - want.im));
            }
        }
        CHECK(nep_lu_solve(lu2, 1, db, n, dx, n, 1.0, NULL));
        CHECK(nep_download(x, dx, n * sizeof *x, NULL));
        err = nrm = 0.0;
        for (int i = 0; i < n; ++i) {                                         /* r = A2 x - b, column sweep over the CSC of A2 */
            r[i].re = -b[i].re; r[i].im = -b[i].im;
        }
        for (int j = 0; j < n; ++j) for (int e = Ap[j]; e < Ap[j + 1]; ++e) { nep_cdouble q = cmul(A2[e], x[j]); r[Ai[e]].re += q.re; r[Ai[e]].im += q.im; }
        for (int i = 0; i < n; ++i) { err += r[i].re * r[i].re + r[i].im * r[i].im; nrm += b[i].re * b[i].re + b[i].im * b[i].im; }
        printf("   nep_lu_factor_dev     max |LU - L2 U2 factors| %.2e, ||A2 x - b||/||b|| %.2e, growth %.2f\n", ferr, sqrt(err / nrm), health[1]);
        if (ferr > 1e-12 || sqrt(err / nrm) > 1e-10 || health[0] != 0.0) return 6;
        CHECK(nep_lu_destroy(lu2)); CHECK(nep_lu_refac_destroy(plan));
        free(perm); free(Ap); free(Ai); free(A2); free(l2); free(u2); free(s2); free(LU2);
    }

    /* ---- K6: orthogonalise w against the (orthonormalised) columns of V */
    nep_cdouble h[3]; double beta = 0.0; int32_t npass = 0;
    for (int j = 0; j < k; ++j) {        /* build an orthonormal basis column by column with the kernel itself */
        void* wj = (char*)dV + (size_t)j * n * sizeof *V;
        if (j == 0) {
            double s = 0.0; for (int i = 0; i < n; ++i) s += V[i].re * V[i].re + V[i].im * V[i].im;
            nep_cdouble a = {1.0 / sqrt(s), 0.0};
            CHECK(nep_scal(n, a, wj, NULL));
        } else CHECK(nep_orth(dV, n, n, j, NULL, wj, h, &beta, 0, &npass, NULL));
    }
    CHECK(nep_download(V, dV, (size_t)n * k * sizeof *V, NULL));
    double worst = 0.0;
    for (int a = 0; a < k; ++a) for (int c = 0; c < k; ++c) {
        double sr = 0.0, si = 0.0;
        for (int i = 0; i < n; ++i) { sr += V[i + (size_t)a * n].re * V[i + (size_t)c * n].re + V[i + (size_t)a * n].im * V[i + (size_t)c * n].im;
                                      si += V[i + (size_t)a * n].re * V[i + (size_t)c * n].im - V[i + (size_t)a * n].im * V[i + (size_t)c * n].re; }
        const double d = hypot(sr - (a == c ? 1.0 : 0.0), si);
        if (d > worst) worst = d;
    }
    printf("K6 nep_orth (DGKS)       ||V^H V - I||_max %.2e (last pass count %d)\n", worst, npass);
    if (worst > 1e-13) return 4;

    /* ---- K7: Y = V B (tall-skinny GEMM on the FP64 matrix cores) */
    nep_cdouble B[3 * 2]; for (int i = 0; i < 6; ++i) { B[i].re = rnd(&seed); B[i].im = rnd(&seed); }
    void* dY = NULL; CHECK(nep_dev_alloc(&dY, (size_t)n * 2 * sizeof *V));
    CHECK(nep_gemm_ts(dV, n, n, k, B, k, 2, dY, n, 0, NULL));
    nep_cdouble* Y = malloc((size_t)n * 2 * sizeof *Y);
    CHECK(nep_download(Y, dY, (size_t)n * 2 * sizeof *Y, NULL));
    err = nrm = 0.0;
    for (int p = 0; p < 2; ++p) for (int i = 0; i < n; ++i) {
        nep_cdouble acc = {0.0, 0.0};
        for (int j = 0; j < k; ++j) { nep_cdouble q = cmul(V[i + (size_t)j * n], B[j + p * k]); acc.re += q.re; acc.im += q.im; }
        err += (acc.re - Y[i + (size_t)p * n].re) * (acc.re - Y[i + (size_t)p * n].re) + (acc.im - Y[i + (size_t)p * n].im) * (acc.im - Y[i + (size_t)p * n].im);
        nrm += acc.re * acc.re + acc.im * acc.im;
    }
    printf("K7 nep_gemm_ts           rel err %.2e\n", sqrt(err / nrm));
    if (sqrt(err / nrm) > 1e-13) return 5;

    /* ---- the whole infinite Arnoldi method as ONE call (nep_iar_run; what `iar(nep::DeviceSPMF; ...)` of julia/NEPMI355X.jl is):
     * M(lambda) = A0 - lambda A1 on the SPMF above (f_0 = 1, f_1 = -lambda), sigma = 0: the factors of M(0) = A0 = tridiag(-1, 2, -1)
     * by the Thomas recurrence, the derivative table (only f_1' = -1 is non-zero), the four eigenvalues next to 0 */
    {
        const int m = 40;
        int32_t* Lp2 = malloc((n + 1) * sizeof *Lp2); int32_t* Li2 = malloc(2 * n * sizeof *Li2); nep_cdouble* Lx2 = calloc(2 * n, sizeof *Lx2);
        int32_t* Up2 = malloc((n + 1) * sizeof *Up2); int32_t* Ui2 = malloc(2 * n * sizeof *Ui2); nep_cdouble* Ux2 = calloc(2 * n, sizeof *Ux2);
        double* dg = malloc(n * sizeof *dg);
        dg[0] = 2.0; for (int i = 1; i < n; ++i) dg[i] = 2.0 - 1.0 / dg[i - 1];
        int l2n = 0, u2n = 0;
        for (int j = 0; j < n; ++j) {
            Lp2[j] = l2n; Li2[l2n] = j; Lx2[l2n++].re = 1.0;
            if (j < n - 1) { Li2[l2n] = j + 1; Lx2[l2n++].re = -1.0 / dg[j]; }
            Up2[j] = u2n;
            if (j > 0) { Ui2[u2n] = j - 1; Ux2[u2n++].re = -1.0; }
            Ui2[u2n] = j; Ux2[u2n++].re = dg[j];
        }
        Lp2[n] = l2n; Up2[n] = u2n;
        nep_lu* lu0 = NULL;
        CHECK(nep_lu_create_csc(n, Lp2, Li2, Lx2, Up2, Ui2, Ux2, NULL, NULL, &lu0));
        nep_cdouble* Ctab = calloc((size_t)m * mt, sizeof *Ctab);           /* m x mt column-major: row j-1 = gamma^j / j f_t^(j)(0) */
        Ctab[0 + (size_t)1 * m].re = -1.0;
        const double cabs[2] = {1.0, 0.0}; const nep_cdouble cf[2] = {{1.0, 0.0}, {0.0, 0.0}};
        double fro[2] = {0.0, 0.0};
        for (int e = 0; e < rp0[n]; ++e) fro[0] += v0[e] * v0[e];
        for (int e = 0; e < n; ++e) fro[1] += v1[e] * v1[e];
        fro[0] = sqrt(fro[0]); fro[1] = sqrt(fro[1]);
        nep_cdouble* vs0 = malloc(n * sizeof *vs0); for (int i = 0; i < n; ++i) { vs0[i].re = 1.0; vs0[i].im = 0.0; }
        nep_iar_opts o; memset(&o, 0, sizeof o);
        o.maxit = m; o.check_error_every = 1; o.orth_method = 0; o.umfpack_refinements = 10; o.errmeasure = 1; o.refine_hint = -1;
        o.tol = 1e-10; o.neigs = 4.0; o.sigma.re = 0.0; o.gamma.re = 1.0;
        nep_iar_result res;
        nep_cdouble* lam = calloc(m, sizeof *lam); nep_cdouble* Qh = malloc((size_t)n * m * sizeof *Qh); double* errs = malloc((size_t)m * m * sizeof *errs);
        int calls = 0;
        CHECK(nep_iar_run(spmf, lu0, n, &o, vs0, Ctab, mt, cabs, cf, fro, fv_eval, &calls, lam, NULL, Qh, errs, NULL, &res, NULL));
        double worst_r = 0.0;
        for (int s = 0; s < res.nret; ++s) {                                /* ||(A0 - lambda A1) q|| / ||q||, host arithmetic */
            double rr = 0.0, qq = 0.0;
            for (int i = 0; i < n; ++i) {
                nep_cdouble acc = {0.0, 0.0};
                for (int e = rp0[i]; e < rp0[i + 1]; ++e) { acc.re += v0[e] * Qh[ci0[e] + (size_t)s * n].re; acc.im += v0[e] * Qh[ci0[e] + (size_t)s * n].im; }
                nep_cdouble t1 = {v1[i] * Qh[i + (size_t)s * n].re, v1[i] * Qh[i + (size_t)s * n].im};
                nep_cdouble lt = cmul(lam[s], t1);
                acc.re -= lt.re; acc.im -= lt.im;
                rr += acc.re * acc.re + acc.im * acc.im;
                qq += Qh[i + (size_t)s * n].re * Qh[i + (size_t)s * n].re + Qh[i + (size_t)s * n].im * Qh[i + (size_t)s * n].im;
            }
            worst_r = fmax(worst_r, sqrt(rr / qq));
        }
        printf("   nep_iar_run           %d pairs after %d steps (%d callbacks), lambda_1 = %.6e, max ||M(lambda) q||/||q|| %.2e\n",
               res.nret, res.k, calls, lam[0].re, worst_r);
        if (res.nret != 4 || res.nconv < 4 || res.k > m || calls < res.k || worst_r > 1e-8 || lam[0].re <= 0.0) return 7;
        CHECK(nep_lu_destroy(lu0));
        free(Lp2); free(Li2); free(Lx2); free(Up2); free(Ui2); free(Ux2); free(dg); free(Ctab); free(vs0); free(lam); free(Qh); free(errs);
    }

    CHECK(nep_lu_destroy(lu)); CHECK(nep_spmf_destroy(spmf));
    CHECK(nep_dev_free(dV)); CHECK(nep_dev_free(dz)); CHECK(nep_dev_free(db)); CHECK(nep_dev_free(dx)); CHECK(nep_dev_free(dY));
    printf("smoke_c ok on %s (libnepmi355 version %d)\n", name, nep_version());
    return 0;
}
