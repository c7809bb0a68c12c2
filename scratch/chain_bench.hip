// micro-benchmarks behind the ward_update_exact_kernel design (run on the GPU box; not part of the product)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#pragma clang fp contract(off)
#define CHECK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)

__global__ __launch_bounds__(64) void k_chain(float* out, float seed, int n){
    float s = seed, p = seed*1e-3f;
    for (int i=0;i<n;i+=8){
        s=s+p; s=s+p; s=s+p; s=s+p; s=s+p; s=s+p; s=s+p; s=s+p;
        asm volatile("" : "+v"(s));
    }
    out[blockIdx.x*64+threadIdx.x]=s;
}
__global__ __launch_bounds__(64) void k_chain3(float* out, const float* in, int n){
    float s = 0; float x = in[threadIdx.x], c = in[64+threadIdx.x];
    for (int i=0;i<n;i+=4){
        float d0=x-c, d1=x-(c+1.f), d2=x-(c+2.f), d3=x-(c+3.f);
        float p0=d0*d0,p1=d1*d1,p2=d2*d2,p3=d3*d3;
        s=s+p0; s=s+p1; s=s+p2; s=s+p3;
        x = x + 1.0f;
        asm volatile("" : "+v"(s), "+v"(x));
    }
    out[blockIdx.x*64+threadIdx.x]=s;
}
// stream a column set like the update kernel: float4 per lane per group, ring depth R x G, no dependent math
template<int G,int R>
__global__ __launch_bounds__(64) void k_stream(const float4* __restrict__ ct, long S, int ngroups, float* out){
    const float4* col = ct + (long)blockIdx.x*64 + threadIdx.x;
    float acc=0;
    float4 r[R][G];
#pragma unroll
    for(int b=0;b<R-1;++b)
#pragma unroll
        for(int u=0;u<G;++u) r[b][u]=col[(long)(b*G+u)*S];
    for(int g0=0; g0<ngroups; g0+=R*G){
#pragma unroll
        for(int b=0;b<R;++b){
            const int lb=(b+R-1)%R;
#pragma unroll
            for(int u=0;u<G;++u) r[lb][u]=col[(long)(g0+(b+R-1)*G+u)*S];
#pragma unroll
            for(int u=0;u<G;++u) acc += r[b][u].x + r[b][u].y + r[b][u].z + r[b][u].w;
        }
    }
    out[blockIdx.x*64+threadIdx.x]=acc;
}
int main(){
    const long S=10048; const int ngroups=512+64; const int nb=157;
    float4* ct; float* out; float* in;
    CHECK(hipMalloc(&ct, sizeof(float4)*S*(ngroups+64)));
    CHECK(hipMemset(ct, 0, sizeof(float4)*S*(ngroups+64)));
    CHECK(hipMalloc(&out, 4*64*4096)); CHECK(hipMalloc(&in, 4*128)); CHECK(hipMemset(in,0,4*128));
    hipEvent_t a,b; hipEventCreate(&a); hipEventCreate(&b);
    auto timeit=[&](const char* name, auto launch, int reps){
        launch(); hipDeviceSynchronize();
        hipEventRecord(a); for(int i=0;i<reps;++i) launch(); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms,a,b); printf("%-40s %8.2f us/launch\n", name, ms*1e3/reps);
    };
    for (int blocks : {1, 157, 1024}) {
        printf("-- %d blocks of one wave\n", blocks);
        timeit("chain 2048 dependent adds", [&]{ hipLaunchKernelGGL(k_chain, dim3(blocks), dim3(64), 0, 0, out, 1.0f, 2048); }, 200);
        timeit("chain 2048 k: sub,mul,add", [&]{ hipLaunchKernelGGL(k_chain3, dim3(blocks), dim3(64), 0, 0, out, in, 2048); }, 200);
    }
    for (int blocks : {18, 157}) {
        printf("-- stream %d waves x 64 slots, 512 groups of 4 k (%.1f MB)\n", blocks, blocks*64*512*16/1e6);
        timeit("stream G=8 R=4", [&]{ hipLaunchKernelGGL((k_stream<8,4>), dim3(blocks), dim3(64), 0, 0, ct, S, 512, out); }, 200);
        timeit("stream G=16 R=3", [&]{ hipLaunchKernelGGL((k_stream<16,3>), dim3(blocks), dim3(64), 0, 0, ct, S, 528, out); }, 200);
        timeit("stream G=8 R=2", [&]{ hipLaunchKernelGGL((k_stream<8,2>), dim3(blocks), dim3(64), 0, 0, ct, S, 512, out); }, 200);
        timeit("stream G=4 R=8", [&]{ hipLaunchKernelGGL((k_stream<4,8>), dim3(blocks), dim3(64), 0, 0, ct, S, 512, out); }, 200);
    }
    return 0;
}
