// dma_bench.hip -- the update kernel's fetch path alone: 256 persistent workgroups of 4 loader waves stream 64-slot blocks
// (512 k-groups x 1 KB) into a 3-stage LDS ring with global_load_lds_dwordx4, one barrier per stage of 32 k-groups, no compute.
// LAYOUT 0: CT4[g][slot][4] as the engine keeps it (a block's k-groups are S*16 B apart); LAYOUT 1: blocked (a block's 512 KB
// contiguous).  Reports the aggregate rate for N slots.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define SG 32
#define KG 512
#define R 3
__device__ __forceinline__ void glds16(const void *g, unsigned l)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(g), "s"(l) : "memory");
}
template <int LAYOUT, int XW>
__global__ __launch_bounds__(256 + 64 * XW) void k(const float *CT, long S, int nblk, int *ctr, float *sink)
{
    extern __shared__ float4 lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const void *)lds;
    __shared__ int cur;
    float acc = 0.f;
    for (;;) {
        __syncthreads();
        if (threadIdx.x == 0) cur = atomicAdd(ctr, 1);
        __syncthreads();
        const int b = cur;
        if (b >= nblk) break;
        const long slot = (long)b * 64 + lane;
        auto issue = [&](int st) {
            if (wave >= 4) return;
#pragma unroll
            for (int q = 0; q < SG / 4; ++q) {
                const int g = st * SG + wave * (SG / 4) + q;
                const char *src = LAYOUT == 0 ? (const char *)CT + ((long)g * S + slot) * 16 : (const char *)CT + (((long)b * KG + g) * 64 + lane) * 16;
                glds16(src, base + (unsigned)(((st % R) * SG + wave * (SG / 4) + q) * 1024));
            }
        };
        for (int i = 0; i < R - 1; ++i) issue(i);
        for (int i = 0; i < KG / SG; ++i) {
            if (wave < 4) {
                if (i + R - 2 < KG / SG) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((R - 2) * (SG / 4)) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();
            if (wave >= 4) acc += lds[(i % R) * SG * 64 + (wave - 4) * 64 + lane].x; // the chain waves' stand-in: one read per stage
            if (i + R - 1 < KG / SG) issue(i + R - 1);
        }
    }
    if (acc == 12345.f) sink[threadIdx.x] = acc;
}
template <int LAYOUT, int XW>
static void run(const float *CT, long S, int *ctr, float *sink)
{
    const int nblk = (int)(S / 64);
    const size_t ldsb = (size_t)R * SG * 1024;
    (void)hipFuncSetAttribute((const void *)k<LAYOUT, XW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        (void)hipMemset(ctr, 0, 4);
        (void)hipEventRecord(a);
        hipLaunchKernelGGL((k<LAYOUT, XW>), dim3(256), dim3(256 + 64 * XW), ldsb, 0, CT, S, nblk, ctr, sink);
        (void)hipEventRecord(b);
        (void)hipEventSynchronize(b);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, a, b);
        best = ms < best ? ms : best;
    }
    printf("layout %s, %d extra waves, S=%ld: %.1f us per pass, %.2f TB/s\n", LAYOUT ? "blocked" : "CT4[g][slot]", XW, S, best * 1e3, (double)S * KG * 16 / (best * 1e-3) / 1e12);
}
int main(int argc, char **argv)
{
    const long S = argc > 1 ? atol(argv[1]) : 100032;
    float *CT, *sink;
    int *ctr;
    (void)hipMalloc(&CT, (size_t)S * KG * 16);
    (void)hipMemset(CT, 0, (size_t)S * KG * 16);
    (void)hipMalloc(&sink, 4096);
    (void)hipMalloc(&ctr, 4);
    run<0, 0>(CT, S, ctr, sink);
    run<1, 0>(CT, S, ctr, sink);
    run<0, 8>(CT, S, ctr, sink);
    run<1, 8>(CT, S, ctr, sink);
    return 0;
}
