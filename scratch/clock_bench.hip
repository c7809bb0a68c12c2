#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned long long* out, int iters){
    unsigned long long t0=__builtin_amdgcn_s_memtime(), r0=__builtin_amdgcn_s_memrealtime();
    float s=threadIdx.x;
    for(int i=0;i<iters;++i){ s=s*1.0001f+0.5f; asm volatile("" : "+v"(s)); }
    unsigned long long t1=__builtin_amdgcn_s_memtime(), r1=__builtin_amdgcn_s_memrealtime();
    if(threadIdx.x==0){ out[blockIdx.x*3]=t1-t0; out[blockIdx.x*3+1]=r1-r0; out[blockIdx.x*3+2]=(unsigned long long)s; }
}
int main(){
    unsigned long long* d; hipMalloc(&d, 8*3*4096); unsigned long long h[3*4096];
    for (int blocks : {1, 157, 1024, 4096}) for (int iters : {2000, 200000}) {
        for (int rep=0; rep<3; ++rep) hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, d, iters);
        hipDeviceSynchronize(); hipMemcpy(h, d, 8*3*blocks, hipMemcpyDeviceToHost);
        double c=h[0], r=h[1];
        printf("blocks %5d iters %7d: shader cycles %.0f, realtime ticks %.0f -> clock %.0f MHz, %.2f cycles/iter\n", blocks, iters, c, r, c/r*100.0, c/iters);
    }
    return 0;
}
