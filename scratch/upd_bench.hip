// ablation of ward_update_exact_kernel's pipeline (scratch; not part of the product)
#include <hip/hip_runtime.h>
#include <cstdio>
#pragma clang fp contract(off)
#ifndef UPD_P
#define UPD_P 6
#endif
#ifndef UPD_GP
#define UPD_GP 4
#endif
#define NT (64*(UPD_P+1))
#define UPD_SG (UPD_P*UPD_GP)
template<int MODE> // 0 full, 1 no chain adds, 2 no produce math (just loads+barriers), 3 no loads, 4 barriers only
__global__ __launch_bounds__(NT) void upd(const float4* __restrict__ CT, const float4* __restrict__ cnew, long S, int dqp, float* out){
    extern __shared__ __attribute__((aligned(16))) float4 dyn[];
    float4 (*ring)[UPD_SG][64] = reinterpret_cast<float4 (*)[UPD_SG][64]>(dyn);
    float4* cn4 = dyn + 2*UPD_SG*64;
    const int lane=threadIdx.x&63, wave=threadIdx.x>>6;
    for (int g=threadIdx.x; g<dqp+2*UPD_SG; g+=NT) cn4[g]=cnew[g];
    __syncthreads();
    const float4* col = CT + (long)blockIdx.x*64 + lane;
    const int pj=wave-1; float s=0;
    float4 va[UPD_GP], vb[UPD_GP], vc[UPD_GP];
    auto load=[&](float4 (&v)[UPD_GP], int stage){ const int g0=stage*UPD_SG+pj*UPD_GP;
        if (MODE>=3) { for(int u=0;u<UPD_GP;++u) v[u]=make_float4(1,2,3,4); return; }
#pragma unroll
        for(int u=0;u<UPD_GP;++u) v[u]=col[(long)(g0+u)*S]; };
    auto produce=[&](const float4 (&v)[UPD_GP], int stage, int buf){ const int g0=stage*UPD_SG+pj*UPD_GP;
        if (MODE==4) return;
#pragma unroll
        for(int u=0;u<UPD_GP;++u){
            if (MODE==2) { ring[buf][pj*UPD_GP+u][lane]=v[u]; continue; }
            const float4 cv=cn4[g0+u];
            const float d0=v[u].x-cv.x,d1=v[u].y-cv.y,d2=v[u].z-cv.z,d3=v[u].w-cv.w;
            ring[buf][pj*UPD_GP+u][lane]=make_float4(d0*d0,d1*d1,d2*d2,d3*d3);
        } };
    auto consume=[&](int buf){
        if (MODE==1 || MODE==4) return;
#pragma unroll
        for(int g=0;g<UPD_SG;++g){ const float4 p=ring[buf][g][lane]; s=s+p.x; s=s+p.y; s=s+p.z; s=s+p.w; } };
    const int nstage=dqp/UPD_SG;
    if (wave>0){ load(va,0); load(vb,1); }
    for (int i=0;i<nstage;i+=3){
        if (wave>0){ load(vc,i+2); produce(va,i,i&1);} __syncthreads(); if(wave==0) consume(i&1);
        if (wave>0){ load(va,i+3); produce(vb,i+1,(i+1)&1);} __syncthreads(); if(wave==0) consume((i+1)&1);
        if (wave>0){ load(vb,i+4); produce(vc,i+2,i&1);} __syncthreads(); if(wave==0) consume(i&1);
    }
    if (wave==0) out[blockIdx.x*64+lane]=s;
}
int main(){
    const long S=10048; const int dqp=((512+3*UPD_SG-1)/(3*UPD_SG))*(3*UPD_SG); float4 *ct,*cn; float* out;
    hipMalloc(&ct, sizeof(float4)*S*(dqp+256)); hipMemset(ct,0,sizeof(float4)*S*(dqp+256));
    hipMalloc(&cn, sizeof(float4)*(dqp+256)); hipMemset(cn,0,sizeof(float4)*(dqp+256)); hipMalloc(&out,4*64*1024);
    hipEvent_t a,b; hipEventCreate(&a); hipEventCreate(&b);
    auto timeit=[&](const char* name, auto launch){ launch(); hipDeviceSynchronize(); hipEventRecord(a); for(int i=0;i<200;++i) launch(); hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms,a,b); printf("%-34s %7.2f us\n",name,ms*5); };
    const size_t lds=(dqp+2*UPD_SG)*16 + 2*UPD_SG*64*16; printf("P=%d GP=%d SG=%d dqp=%d lds=%zu\n",UPD_P,UPD_GP,UPD_SG,dqp,lds);
    hipFuncSetAttribute((const void*)upd<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);hipFuncSetAttribute((const void*)upd<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);hipFuncSetAttribute((const void*)upd<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);hipFuncSetAttribute((const void*)upd<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);hipFuncSetAttribute((const void*)upd<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int nb : {157}) { printf("-- %d workgroups\n", nb);
    timeit("full", [&]{ hipLaunchKernelGGL(upd<0>, dim3(nb), dim3(NT), lds, 0, ct, cn, S, dqp, out); });
    timeit("no chain adds", [&]{ hipLaunchKernelGGL(upd<1>, dim3(nb), dim3(NT), lds, 0, ct, cn, S, dqp, out); });
    timeit("no produce math", [&]{ hipLaunchKernelGGL(upd<2>, dim3(nb), dim3(NT), lds, 0, ct, cn, S, dqp, out); });
    timeit("no global loads", [&]{ hipLaunchKernelGGL(upd<3>, dim3(nb), dim3(NT), lds, 0, ct, cn, S, dqp, out); });
    timeit("barriers only", [&]{ hipLaunchKernelGGL(upd<4>, dim3(nb), dim3(NT), lds, 0, ct, cn, S, dqp, out); }); }
    return 0;
}
