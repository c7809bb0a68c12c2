// chain4_bench.hip -- chains per wave: how many (pairs of) in-order sums should one chain wave run per x read?  Operands in LDS,
// one barrier per 32 k-groups, 16 chains per 64-slot block in every variant:  PAIRS pairs per wave x (8 / PAIRS) chain waves.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#pragma clang fp contract(off)
typedef float f2 __attribute__((ext_vector_type(2)));
#define SG 32
#define NSTAGE 16
#define RING 3
template <int PAIRS>
__global__ __launch_bounds__((8 / PAIRS) * 64) void k(float *out, const float *in, int reps)
{
    extern __shared__ float4 lds[];
    constexpr int WAVES = 8 / PAIRS;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < RING * (SG * 64 + 16 * SG); i += WAVES * 64) lds[i] = make_float4(in[i & 255], in[(i + 1) & 255], 0.5f, 0.25f);
    __syncthreads();
    f2 sP[PAIRS];
#pragma unroll
    for (int p = 0; p < PAIRS; ++p) sP[p] = f2{0.f, 0.f};
    for (int r = 0; r < reps; ++r)
        for (int st = 0; st < NSTAGE; ++st) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            const float4 *sb = lds + (st % RING) * (SG * 64 + 16 * SG);
            const float4 *xr = sb + lane, *ca = sb + SG * 64 + (wave * PAIRS) * 2 * SG;
#pragma unroll
            for (int q = 0; q < SG / 2; ++q) { // two k-groups at a time
                float4 xv[2], c0[PAIRS][2], c1[PAIRS][2];
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    xv[g] = xr[(q * 2 + g) * 64];
#pragma unroll
                    for (int p = 0; p < PAIRS; ++p) {
                        c0[p][g] = ca[p * 2 * SG + (q * 2 + g) * 2];
                        c1[p][g] = ca[p * 2 * SG + (q * 2 + g) * 2 + 1];
                    }
                }
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    const f2 x0 = {xv[g].x, xv[g].x}, x1 = {xv[g].y, xv[g].y}, x2 = {xv[g].z, xv[g].z}, x3 = {xv[g].w, xv[g].w};
#pragma unroll
                    for (int p = 0; p < PAIRS; ++p) {
                        const f2 k0 = {c0[p][g].x, c0[p][g].y}, k1 = {c0[p][g].z, c0[p][g].w}, k2 = {c1[p][g].x, c1[p][g].y}, k3 = {c1[p][g].z, c1[p][g].w};
                        const f2 d0 = x0 - k0, d1 = x1 - k1, d2 = x2 - k2, d3 = x3 - k3;
                        const f2 q0 = d0 * d0, q1 = d1 * d1, q2 = d2 * d2, q3 = d3 * d3;
                        sP[p] = sP[p] + q0;
                        sP[p] = sP[p] + q1;
                        sP[p] = sP[p] + q2;
                        sP[p] = sP[p] + q3;
                    }
                }
            }
        }
#pragma unroll
    for (int p = 0; p < PAIRS; ++p) {
        out[((blockIdx.x * WAVES * 64 + threadIdx.x) * PAIRS + p) * 2] = sP[p].x;
        out[((blockIdx.x * WAVES * 64 + threadIdx.x) * PAIRS + p) * 2 + 1] = sP[p].y;
    }
}
template <int PAIRS>
static void run(float *out, const float *in)
{
    const size_t lds = RING * (SG * 64 + 16 * SG) * 16;
    (void)hipFuncSetAttribute((const void *)k<PAIRS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    const int reps = 20;
    hipLaunchKernelGGL((k<PAIRS>), dim3(256), dim3((8 / PAIRS) * 64), lds, 0, out, in, 2);
    (void)hipEventRecord(a);
    hipLaunchKernelGGL((k<PAIRS>), dim3(256), dim3((8 / PAIRS) * 64), lds, 0, out, in, reps);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    printf("pairs per wave %d, chain waves %d: %.2f us per 64-slot x 16-chain block\n", PAIRS, 8 / PAIRS, ms * 1e3 / reps);
}
int main()
{
    float *out, *in;
    (void)hipMalloc(&out, 256 * 1024 * 8 * 4);
    (void)hipMalloc(&in, 1024);
    std::vector<float> h(256);
    for (int i = 0; i < 256; ++i) h[i] = 0.001f * i;
    (void)hipMemcpy(in, h.data(), 1024, hipMemcpyHostToDevice);
    run<1>(out, in);
    run<2>(out, in);
    run<4>(out, in);
    run<8>(out, in);
    return 0;
}
