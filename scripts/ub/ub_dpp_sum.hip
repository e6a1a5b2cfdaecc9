// micro-check: wave_sum_dpp (common.h) against the shuffle butterfly on random data, eight sums back to back as in k_orth_dots
#include "../../nonlineareigenproblems.jl_amd/csrc/common.h"
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
__global__ void k(const double* x, double* o1, double* o2, int n) {
    const int w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    double a[8], r1[8], r2[8];
    for (int j = 0; j < 8; ++j) a[j] = x[((size_t)w * 8 + j) * 64 + lane];
    for (int j = 0; j < 8; ++j) r1[j] = wave_sum_dpp(a[j]);
    for (int j = 0; j < 8; ++j) r2[j] = wave_reduce_sum(a[j]);
    if (lane == 0) for (int j = 0; j < 8; ++j) { o1[w * 8 + j] = r1[j]; o2[w * 8 + j] = r2[j]; }
}
int main() {
    const int waves = 4096, n = waves * 8 * 64;
    std::vector<double> h(n);
    srand(1);
    for (int i = 0; i < n; ++i) { double s = ldexp(1.0, (rand() % 80) - 40); h[i] = s * ((rand() / (double)RAND_MAX) - 0.5); if (rand() % 17 == 0) h[i] = 0.0; }
    double *dx, *d1, *d2;
    hipMalloc(&dx, n * 8); hipMalloc(&d1, waves * 8 * 8); hipMalloc(&d2, waves * 8 * 8);
    hipMemcpy(dx, h.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(waves / 4), dim3(256), 0, 0, dx, d1, d2, n);
    std::vector<double> r1(waves * 8), r2(waves * 8);
    hipMemcpy(r1.data(), d1, waves * 64, hipMemcpyDeviceToHost); hipMemcpy(r2.data(), d2, waves * 64, hipMemcpyDeviceToHost);
    double worst = 0; int bad = 0;
    for (int i = 0; i < waves * 8; ++i) {
        long double ex = 0, ab = 0; for (int l = 0; l < 64; ++l) { ex += h[(size_t)i * 64 + l]; ab += fabsl(h[(size_t)i * 64 + l]); }
        double e1 = fabs((double)(r1[i] - ex)) / (double)ab, e2 = fabs((double)(r2[i] - ex)) / (double)ab;
        if (e1 > worst) worst = e1;
        if (e1 > 1e-14) { if (bad < 5) printf("bad %d dpp %.17g shfl %.17g exact %.17Lg\n", i, r1[i], r2[i], ex); ++bad; }
        (void)e2;
    }
    printf("dpp sum: worst rel err %.3e (vs sum|x|), bad %d of %d\n", worst, bad, waves * 8);
    return 0;
}
