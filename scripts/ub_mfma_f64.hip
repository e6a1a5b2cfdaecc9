#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4 __attribute__((ext_vector_type(4)));
template<int NACC>
__global__ __launch_bounds__(256) void k(double* out, int iters){
  d4 acc[NACC];
  for(int i=0;i<NACC;i++) acc[i]=(d4){0,0,0,0};
  double a = threadIdx.x*1e-3, b = threadIdx.x*2e-3+1;
  for(int it=0; it<iters; ++it){
#pragma unroll
    for(int i=0;i<NACC;i++) acc[i]=__builtin_amdgcn_mfma_f64_16x16x4f64(a,b,acc[i],0,0,0);
  }
  double s=0; for(int i=0;i<NACC;i++) s+=acc[i][0]+acc[i][1]+acc[i][2]+acc[i][3];
  out[blockIdx.x*blockDim.x+threadIdx.x]=s;
}
int main(){
  double* d; hipMalloc(&d, 256*2048*8*8);
  hipEvent_t e0,e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for(int wpb : {4}) for (int blocks : {256, 512, 1024}) {
    int iters=4000; const int NACC=8;
    hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(64*wpb), 0, 0, d, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(64*wpb), 0, 0, d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms,e0,e1);
    double flops = (double)blocks*wpb*iters*NACC*2048.0;
    printf("waves/block %d blocks %d: %.3f ms  %.1f TFLOP/s  (%.1f cycles@2.4GHz per MFMA per SIMD)\n", wpb, blocks, ms, flops/ms/1e9, ms*1e-3*2.4e9/((double)blocks*wpb*iters*NACC/1024.0));
  }
  return 0;
}
