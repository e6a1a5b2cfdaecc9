#include <hip/hip_runtime.h>
typedef double d2t __attribute__((ext_vector_type(2)));
__global__ void k(const d2t* __restrict__ g, d2t* out, int n) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    d2t* L = (d2t*)smem;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // wave wv loads 64 consecutive 16-byte items, gathered: lane reads g[(lane*7+wv) % n]
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(g + ((lane * 7 + wv) % n)),
                                     (void __attribute__((address_space(3)))*)(L + wv * 64), 16, 0, 0);
    __builtin_amdgcn_s_waitcnt(0x0f70);
    __syncthreads();
    out[threadIdx.x] = L[threadIdx.x];
}
int main() {
    const int n = 1000; d2t* g; d2t* o;
    hipMalloc(&g, n * 16); hipMalloc(&o, 256 * 16);
    d2t h[1000]; for (int i = 0; i < n; ++i) { h[i].x = i; h[i].y = -i; }
    hipMemcpy(g, h, n * 16, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 256 * 16, 0, g, o, n);
    d2t r[256]; hipMemcpy(r, o, 256 * 16, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int t = 0; t < 256; ++t) { int lane = t & 63, wv = t >> 6; int e = (lane * 7 + wv) % n; if (r[t].x != e || r[t].y != -e) ++bad; }
    printf("bad %d (r[5] = %g %g)\n", bad, r[5].x, r[5].y);
    return bad != 0;
}
