// what ONE dependent dispatch costs: N empty kernels back to back in a stream, the same chain replayed as a hipGraph, and the chain
// with a small real kernel (one wave reads a word and leaves) -- per-kernel time from HIP events
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__global__ void k_empty() {}
__global__ void k_word(const int* p, int* q) { if (threadIdx.x == 0 && *p == 12345) *q = 1; }
__global__ void k_grid(const int* p, int* q) { if (blockIdx.x == 0 && threadIdx.x == 0 && *p == 12345) *q = 1; }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main() {
    const int N = 2000;
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    int *p, *q; CK(hipMalloc(&p, 4)); CK(hipMalloc(&q, 4)); CK(hipMemset(p, 0, 4)); CK(hipMemset(q, 0, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int variant = 0; variant < 3; ++variant) {
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipStreamSynchronize(st));
            CK(hipEventRecord(e0, st));
            for (int i = 0; i < N; ++i) {
                if (variant == 0) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st);
                else if (variant == 1) hipLaunchKernelGGL(k_word, dim3(1), dim3(64), 0, st, p, q);
                else hipLaunchKernelGGL(k_grid, dim3(1024), dim3(256), 0, st, p, q);
            }
            CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep == 2) printf("stream   variant %d: %.2f us per kernel\n", variant, ms * 1e3 / N);
        }
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
        for (int i = 0; i < N; ++i) {
            if (variant == 0) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st);
            else if (variant == 1) hipLaunchKernelGGL(k_word, dim3(1), dim3(64), 0, st, p, q);
            else hipLaunchKernelGGL(k_grid, dim3(1024), dim3(256), 0, st, p, q);
        }
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipStreamSynchronize(st));
            CK(hipEventRecord(e0, st));
            CK(hipGraphLaunch(ge, st));
            CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep == 2) printf("hipGraph variant %d: %.2f us per kernel\n", variant, ms * 1e3 / N);
        }
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    return 0;
}
