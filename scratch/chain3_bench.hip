// chain3_bench.hip -- the packed chain-wave loop of ward_update_batch2_kernel in isolation (operands resident in LDS, no DMA):
// how does time per k-group scale with the number of chain waves per SIMD (8 / 12 / 16 waves per workgroup, one workgroup per
// CU), with and without one raw barrier per 16 / 32 k-groups?   build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off
// -fno-slp-vectorize scratch/chain3_bench.hip -o scratch/chain3_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#pragma clang fp contract(off)
typedef float f2 __attribute__((ext_vector_type(2)));
#define SG 32          /* k-groups per stage */
#define NSTAGE 16      /* D = 2048 */
#define RING 3

template <int WAVES, int BAR, int MODE>
__global__ __launch_bounds__(WAVES * 64) void k(float *out, const float *in, int reps)
{
    extern __shared__ float4 lds[]; // ring [RING][SG*64 x | 8 pairs * 2*SG c]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < RING * (SG * 64 + 16 * SG); i += WAVES * 64) lds[i] = make_float4(in[i & 255], in[(i + 1) & 255], 0.5f, 0.25f);
    __syncthreads();
    f2 sP = {0.f, 0.f};
    float4 xv[4] = {}, c0[4] = {}, c1[4] = {};
    const int pair = wave & 7;
    for (int r = 0; r < reps; ++r)
        for (int st = 0; st < NSTAGE; ++st) {
            if (BAR) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
            const float4 *sb = lds + (st % RING) * (SG * 64 + 16 * SG);
            const float4 *xr = sb + lane, *ca = sb + SG * 64 + pair * 2 * SG;
#pragma unroll
            for (int q = 0; q < SG / 4; ++q) {
#pragma unroll
                for (int g = 0; g < 4; ++g) { // MODE 0: x and c from LDS; 1: only x from LDS; 2: nothing from LDS (pure VALU)
                    if (MODE <= 1 || (st == 0 && q == 0 && r == 0)) xv[g] = xr[(q * 4 + g) * 64];
                    if (MODE == 0 || (st == 0 && q == 0 && r == 0)) {
                        c0[g] = ca[(q * 4 + g) * 2];
                        c1[g] = ca[(q * 4 + g) * 2 + 1];
                    }
                    asm volatile("" : "+v"(xv[g].x), "+v"(c0[g].x), "+v"(c1[g].x));
                }
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f2 k0 = {c0[g].x, c0[g].y}, k1 = {c0[g].z, c0[g].w}, k2 = {c1[g].x, c1[g].y}, k3 = {c1[g].z, c1[g].w};
                    const f2 x0 = {xv[g].x, xv[g].x}, x1 = {xv[g].y, xv[g].y}, x2 = {xv[g].z, xv[g].z}, x3 = {xv[g].w, xv[g].w};
                    const f2 d0 = x0 - k0, d1 = x1 - k1, d2 = x2 - k2, d3 = x3 - k3;
                    const f2 q0 = d0 * d0, q1 = d1 * d1, q2 = d2 * d2, q3 = d3 * d3;
                    sP = sP + q0;
                    sP = sP + q1;
                    sP = sP + q2;
                    sP = sP + q3;
                }
            }
        }
    out[(blockIdx.x * WAVES * 64 + threadIdx.x) * 2] = sP.x;
    out[(blockIdx.x * WAVES * 64 + threadIdx.x) * 2 + 1] = sP.y;
}

template <int WAVES, int BAR, int MODE>
static void run(float *out, const float *in)
{
    const size_t lds = RING * (SG * 64 + 16 * SG) * 16;
    (void)hipFuncSetAttribute((const void *)k<WAVES, BAR, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    const int reps = 20;
    hipLaunchKernelGGL((k<WAVES, BAR, MODE>), dim3(256), dim3(WAVES * 64), lds, 0, out, in, 2);
    hipEventRecord(a);
    hipLaunchKernelGGL((k<WAVES, BAR, MODE>), dim3(256), dim3(WAVES * 64), lds, 0, out, in, reps);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    const double us_per_block = ms * 1e3 / reps;                       // one pass = one 64-slot block's 512 k-groups for every wave
    const double per_pair_block = us_per_block / (WAVES / 8.0);        // time per (8 pairs x 64 slots) unit of work
    printf("mode %d waves %2d (%d per SIMD) barrier %d: %.2f us per pass, %.2f us per 64-slot x 16-chain unit\n", MODE, WAVES, WAVES / 4, BAR, us_per_block, per_pair_block);
}

int main()
{
    float *out, *in;
    hipMalloc(&out, 256 * 1024 * 2 * 4);
    hipMalloc(&in, 1024);
    std::vector<float> h(256);
    for (int i = 0; i < 256; ++i) h[i] = 0.001f * i;
    hipMemcpy(in, h.data(), 1024, hipMemcpyHostToDevice);
    run<8, 1, 0>(out, in);
    run<8, 1, 1>(out, in);
    run<8, 1, 2>(out, in);
    run<16, 1, 0>(out, in);
    run<16, 1, 1>(out, in);
    run<16, 1, 2>(out, in);
    return 0;
}
