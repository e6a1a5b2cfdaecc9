// Issue-rate microbenchmark of v_mfma_f64_16x16x4_f64 on gfx950, second version: the accumulators are pinned in VGPRs through
// inline asm (the first version, ub_mfma_f64.hip, let the compiler shuttle all 64 accumulator registers between the VGPR and
// AGPR files every iteration -- 128 v_accvgpr moves per 8 MFMAs -- and measured that, not the matrix pipe).
//   mode 0: operands in registers          mode 1: B operand of every MFMA from LDS (one ds_read_b64 per MFMA)
//   mode 2: one ds_read_b64 per TWO MFMAs  (two accumulator sets share the fragment)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4 __attribute__((ext_vector_type(4)));
#define MFMA(acc, a, b) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
template <int MODE>
__global__ __launch_bounds__(512) void k(double* out, int iters) {
    __shared__ double lds[8 * 2 * 64 * 4];
    for (int i = threadIdx.x; i < 8 * 2 * 64 * 4; i += blockDim.x) lds[i] = 1e-3 * i;
    __syncthreads();
    d4 acc[8], acc2[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { acc[i] = (d4){0, 0, 0, 0}; acc2[i] = (d4){0, 0, 0, 0}; }
    double a = threadIdx.x * 1e-3, a2 = a + 1.0, b = threadIdx.x * 2e-3 + 1;
    const int lane = threadIdx.x & 63;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 8; ++i) MFMA(acc[i], a, b);
#pragma unroll
            for (int i = 0; i < 8; ++i) MFMA(acc2[i], a2, b);
        } else if (MODE == 1) {
            const double* bk = lds + (it & 3) * 1024 + lane;
#pragma unroll
            for (int i = 0; i < 8; ++i) { const double bb = bk[i * 128]; MFMA(acc[i], a, bb); }
#pragma unroll
            for (int i = 0; i < 8; ++i) { const double bb = bk[i * 128 + 64]; MFMA(acc2[i], a2, bb); }
        } else {
            const double* bk = lds + (it & 3) * 1024 + lane;
#pragma unroll
            for (int i = 0; i < 8; ++i) { const double bb = bk[i * 128]; MFMA(acc[i], a, bb); MFMA(acc2[i], a2, bb); }
        }
    }
    double s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + acc2[i][0] + acc2[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE>
static void run(double* d, int wpb, int blocks) {
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64 * wpb), 0, 0, d, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64 * wpb), 0, 0, d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double nm = (double)blocks * wpb * iters * 16;
    printf("mode %d waves/block %d blocks %d: %.3f ms  %.1f TFLOP/s  (%.1f cycles@2.4GHz per MFMA per SIMD)\n", MODE, wpb, blocks, ms,
           nm * 2048.0 / ms / 1e9, ms * 1e-3 * 2.4e9 / (nm / 1024.0));
}
int main() {
    double* d; hipMalloc(&d, 256 * 4096 * 8 * 8);
    for (int wpb : {4, 8}) { run<0>(d, wpb, 256); run<1>(d, wpb, 256); run<2>(d, wpb, 256); }
    run<0>(d, 4, 512); run<1>(d, 4, 512);
    return 0;
}
