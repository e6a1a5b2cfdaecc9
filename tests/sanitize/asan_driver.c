/* AddressSanitizer / UBSan driver for the host-side analysis of the K5 block schedule (nep_lu_analyze: elimination tree,
 * multilevel partition, permutation, per-factor CSR splitting, chunk tables -- csrc/trsv_ml.hip; no device is touched)
 * and for the plan builder of the device-side numeric LU (nep_lu_refac_analyze -- csrc/lufac.hip: the enumeration runs on
 * worker threads and must give the same plan for every thread count).
 * Random lower/upper patterns of several shapes, CSR and CSC, from 4 threads at once.  Built and run by
 * tests/test_host_logic.py::test_asan_host_analysis (make -C tests/sanitize). */
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "nepmi355.h"

static unsigned lcg(unsigned* s) { *s = *s * 1664525u + 1013904223u; return *s >> 8; }

/* random pattern of a unit-lower L (with diagonal) and an upper U (with diagonal) in CSR; band + random fill */
static void make(int n, int band, int extra, unsigned seed, int32_t** Lp, int32_t** Li, int32_t** Up, int32_t** Ui) {
    *Lp = malloc((n + 1) * sizeof(int32_t)); *Up = malloc((n + 1) * sizeof(int32_t));
    *Li = malloc((size_t)n * (band + extra + 1) * sizeof(int32_t)); *Ui = malloc((size_t)n * (band + extra + 1) * sizeof(int32_t));
    int ln = 0, un = 0;
    for (int i = 0; i < n; ++i) {
        (*Lp)[i] = ln; (*Up)[i] = un;
        for (int b = band; b >= 1; --b) if (i - b >= 0 && (lcg(&seed) & 3)) (*Li)[ln++] = i - b;
        (*Li)[ln++] = i;
        (*Ui)[un++] = i;
        for (int b = 1; b <= band; ++b) if (i + b < n && (lcg(&seed) & 3)) (*Ui)[un++] = i + b;
        for (int e = 0; e < extra; ++e) { int j = (int)(lcg(&seed) % (unsigned)n); if (j > i + band) (*Ui)[un++] = j; }
    }
    (*Lp)[n] = ln; (*Up)[n] = un;
}
/* CSR -> CSC of the pattern */
static void transpose(int n, const int32_t* P, const int32_t* I, int32_t** TP, int32_t** TI) {
    *TP = calloc(n + 1, sizeof(int32_t)); *TI = malloc((size_t)P[n] * sizeof(int32_t) + 4);
    for (int e = 0; e < P[n]; ++e) (*TP)[I[e] + 1]++;
    for (int i = 0; i < n; ++i) (*TP)[i + 1] += (*TP)[i];
    int32_t* pos = malloc(n * sizeof(int32_t));
    for (int i = 0; i < n; ++i) pos[i] = (*TP)[i];
    for (int i = 0; i < n; ++i) for (int e = P[i]; e < P[i + 1]; ++e) (*TI)[pos[I[e]]++] = i;
    free(pos);
}

static void* worker(void* arg) {
    const unsigned tid = (unsigned)(uintptr_t)arg;
    const int sizes[] = {1, 2, 17, 300, 2500};
    long checks = 0;
    for (int s = 0; s < 5; ++s) for (int band = 0; band <= 6; band += 3) for (int extra = 0; extra <= 2; extra += 2) {
        const int n = sizes[s];
        int32_t *Lp, *Li, *Up, *Ui, *LTp, *LTi, *UTp, *UTi;
        make(n, band, extra, 1000u * tid + 10u * s + band + extra, &Lp, &Li, &Up, &Ui);
        int64_t a[8], b[8];
        const int rc = nep_lu_analyze(n, 0, Lp, Li, Up, Ui, a);
        transpose(n, Lp, Li, &LTp, &LTi); transpose(n, Up, Ui, &UTp, &UTi);
        const int rc2 = nep_lu_analyze(n, 1, LTp, LTi, UTp, UTi, b);
        if (rc != rc2 || (rc == 0 && (a[0] != b[0] || a[1] != b[1] || a[3] != b[3] || a[5] != b[5]))) {
            fprintf(stderr, "CSR/CSC mismatch n=%d band=%d extra=%d rc=%d/%d\n", n, band, extra, rc, rc2); exit(2);
        }
        if (rc == 0 && (a[0] < 1 || a[1] < 1 || a[2] > 256 || a[4] < n || a[6] < n)) { fprintf(stderr, "implausible analysis\n"); exit(3); }
        ++checks;
        free(Lp); free(Li); free(Up); free(Ui); free(LTp); free(LTi); free(UTp); free(UTi);
    }
    /* malformed input must be rejected, not read out of bounds */
    int32_t Lp[3] = {0, 1, 2}, Li[2] = {0, 7}, Up[3] = {0, 1, 2}, Ui[2] = {0, 1};
    int64_t o[8];
    if (nep_lu_analyze(2, 0, Lp, Li, Up, Ui, o) == 0) { fprintf(stderr, "out-of-range column accepted\n"); exit(4); }
    return (void*)checks;
}

/* ---- plan builder of the device-side numeric LU (nep_lu_refac_analyze: symbolic partition + two-pass product enumeration on
 * worker threads, host only).  Input: a random structurally symmetric pattern A with full diagonal and its filled pattern
 * (symbolic elimination without pivoting, done here on a dense boolean matrix), i.e. a pattern closed under the elimination. */
static long plan_checks(unsigned tid) {
    const int sizes[] = {3, 40, 300, 700};
    long checks = 0;
    for (int s = 0; s < 4; ++s) {
        const int n = sizes[s];
        unsigned seed = 77u + 13u * tid + (unsigned)s;
        unsigned char* M = calloc((size_t)n * n, 1);           /* M[i*n + j] != 0: entry (i, j); bit 1: original entry of A */
        for (int i = 0; i < n; ++i) {
            M[(size_t)i * n + i] = 3;
            for (int b = 1; b <= 3; ++b) if (i + b < n && (lcg(&seed) & 1)) { M[(size_t)i * n + i + b] = 3; M[(size_t)(i + b) * n + i] = 3; }
            if ((lcg(&seed) & 7) == 0) { int j = (int)(lcg(&seed) % (unsigned)n); M[(size_t)i * n + j] = 3; M[(size_t)j * n + i] = 3; }
        }
        for (int k = 0; k < n; ++k)                            /* fill */
            for (int i = k + 1; i < n; ++i) if (M[(size_t)i * n + k])
                for (int j = k + 1; j < n; ++j) if (M[(size_t)k * n + j] && !M[(size_t)i * n + j]) M[(size_t)i * n + j] = 1;
        /* CSC of L (rows >= j), U (rows <= j), A (original entries) */
        int32_t *Lp = calloc(n + 1, 4), *Up = calloc(n + 1, 4), *Ap = calloc(n + 1, 4);
        long nl = 0, nu = 0, na = 0;
        for (int j = 0; j < n; ++j) for (int i = 0; i < n; ++i) if (M[(size_t)i * n + j]) { if (i >= j) ++nl; if (i <= j) ++nu; if (M[(size_t)i * n + j] & 2) ++na; }
        int32_t *Li = malloc(nl * 4 + 4), *Ui = malloc(nu * 4 + 4), *Ai = malloc(na * 4 + 4), *perm = malloc(n * 4);
        nl = nu = na = 0;
        long long expect = 0;
        for (int j = 0; j < n; ++j) {
            perm[j] = j;
            for (int i = 0; i < n; ++i) if (M[(size_t)i * n + j]) {
                if (i <= j) Ui[nu++] = i;
                if (i >= j) Li[nl++] = i;
                if (M[(size_t)i * n + j] & 2) Ai[na++] = i;
            }
            Lp[j + 1] = (int32_t)nl; Up[j + 1] = (int32_t)nu; Ap[j + 1] = (int32_t)na;
        }
        for (int k = 0; k < n; ++k) {
            long lc = 0, uc = 0;
            for (int i = k + 1; i < n; ++i) { if (M[(size_t)i * n + k]) ++lc; if (M[(size_t)k * n + i]) ++uc; }
            expect += lc * uc;
        }
        int64_t o1[8], o4[8];
        setenv("NEP_LU_PLAN_THREADS", "1", 1);
        const int rc1 = nep_lu_refac_analyze(n, Lp, Li, Up, Ui, perm, perm, Ap, Ai, o1);
        setenv("NEP_LU_PLAN_THREADS", "4", 1);
        const int rc4 = nep_lu_refac_analyze(n, Lp, Li, Up, Ui, perm, perm, Ap, Ai, o4);
        if (rc1 || rc4) { fprintf(stderr, "refac_analyze failed n=%d rc=%d/%d: %s\n", n, rc1, rc4, nep_last_error()); exit(5); }
        for (int q = 0; q < 8; ++q) if (o1[q] != o4[q]) { fprintf(stderr, "plan depends on the thread count (n=%d, field %d)\n", n, q); exit(6); }
        if (o1[0] != expect || o1[0] != o1[1] + o1[2] + o1[4]) { fprintf(stderr, "product count %lld, expected %lld\n", (long long)o1[0], expect); exit(7); }
        /* an entry of A without a slot in L + U must be refused */
        if (n >= 40) {
            int32_t keep = Li[Lp[1] - 1];
            Li[Lp[1] - 1] = Li[Lp[1] - 1] == n - 1 ? n - 2 : n - 1;      /* move the last row index of column 0: pattern no longer closed / sorted */
            int64_t ob[8];
            (void)nep_lu_refac_analyze(n, Lp, Li, Up, Ui, perm, perm, Ap, Ai, ob);   /* any return code; must not crash or leak */
            Li[Lp[1] - 1] = keep;
        }
        ++checks;
        free(M); free(Lp); free(Up); free(Ap); free(Li); free(Ui); free(Ai); free(perm);
    }
    return checks;
}

int main(void) {
    pthread_t th[4];
    for (uintptr_t t = 0; t < 4; ++t) pthread_create(&th[t], NULL, worker, (void*)t);
    long total = 0;
    for (int t = 0; t < 4; ++t) { void* r; pthread_join(th[t], &r); total += (long)r; }
    const long plans = plan_checks(0) + plan_checks(1);       /* (setenv: not from concurrent threads) */
    printf("asan_driver ok: %ld analyses, %ld device-LU plans\n", total, plans);
    return 0;
}
