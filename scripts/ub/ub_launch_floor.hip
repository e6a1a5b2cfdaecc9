// micro-benchmark: duration of a chain of tiny dependent kernels (HIP events around a graph replay / eager loop)
// variants: empty kernel; 1, 2, 3 dependent global loads; small vs 256-byte kernarg
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
struct Big { const int* idx; const double* v; double* out; int n; long pad[28]; };
__global__ void k_empty(int n) {}
__global__ void k_chain(const int* idx, const double* v, double* out, int depth, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int j = i;
    for (int d = 0; d < depth; ++d) j = idx[j];
    out[i] = v[j];
}
__global__ void k_big(const Big a) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    a.out[i] = a.v[a.idx[i]];
}
int main() {
    const int n = 1 << 16;
    std::vector<int> h(n); for (int i = 0; i < n; ++i) h[i] = (i * 7919 + 13) % n;
    int* idx; double *v, *out;
    CK(hipMalloc(&idx, n * 4)); CK(hipMalloc(&v, n * 8)); CK(hipMalloc(&out, n * 8));
    CK(hipMemcpy(idx, h.data(), n * 4, hipMemcpyHostToDevice)); CK(hipMemset(v, 0, n * 8));
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int NK = 20, REP = 200;
    for (int wg : {1, 64, 1024}) {
      for (int variant = 0; variant < 6; ++variant) {
        auto launch = [&](hipStream_t s) {
            for (int k = 0; k < NK; ++k) {
                if (variant == 0) hipLaunchKernelGGL(k_empty, dim3(wg), dim3(256), 0, s, n);
                else if (variant <= 4) hipLaunchKernelGGL(k_chain, dim3(wg), dim3(256), 0, s, (const int*)idx, (const double*)v, out, variant - 1, wg * 256 < n ? wg * 256 : n);
                else { Big b; b.idx = idx; b.v = v; b.out = out; b.n = wg * 256 < n ? wg * 256 : n; hipLaunchKernelGGL(k_big, dim3(wg), dim3(256), 0, s, b); }
            }
        };
        // graph
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal)); launch(st); CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int r = 0; r < 5; ++r) CK(hipGraphLaunch(ge, st));
        CK(hipStreamSynchronize(st));
        CK(hipEventRecord(e0, st));
        for (int r = 0; r < REP; ++r) CK(hipGraphLaunch(ge, st));
        CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
        float msg; CK(hipEventElapsedTime(&msg, e0, e1));
        // eager
        for (int r = 0; r < 5; ++r) launch(st);
        CK(hipStreamSynchronize(st));
        CK(hipEventRecord(e0, st));
        for (int r = 0; r < REP; ++r) launch(st);
        CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
        float mse; CK(hipEventElapsedTime(&mse, e0, e1));
        const char* nm[] = {"empty", "0 dep loads (1 load)", "1 dep load", "2 dep loads", "3 dep loads", "256B kernarg, 1 dep load"};
        printf("wg=%4d %-28s graph %.2f us/kernel   eager %.2f us/kernel\n", wg, nm[variant], msg * 1e3 / (REP * NK), mse * 1e3 / (REP * NK));
        hipGraphExecDestroy(ge); hipGraphDestroy(g);
      }
    }
    return 0;
}
