/* AddressSanitizer / UBSan driver for the host-side analysis of the K5 block schedule (nep_lu_analyze: elimination tree,
 * multilevel partition, permutation, per-factor CSR splitting, chunk tables -- csrc/trsv_ml.hip; no device is touched).
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

int main(void) {
    pthread_t th[4];
    for (uintptr_t t = 0; t < 4; ++t) pthread_create(&th[t], NULL, worker, (void*)t);
    long total = 0;
    for (int t = 0; t < 4; ++t) { void* r; pthread_join(th[t], &r); total += (long)r; }
    printf("asan_driver ok: %ld analyses\n", total);
    return 0;
}
