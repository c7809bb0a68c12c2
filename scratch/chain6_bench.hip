// chain6_bench.hip -- chain5's 1 slot x 4 pairs loop in the real kernel's setting: ring of RINGS stages (139 KB at 4), XW extra waves that only
// take part in the barriers (the loader waves), 168-VGPR budget.  Derived from chain5_bench.hip -- register tiling of the update kernel's chain waves: one lane runs SL slots x PAIRS chain pairs, so one x
// read serves 2*PAIRS chains and one centroid read serves SL slots (the distance-tile kernel's 8x8 idea applied to the update).
// 8 chain waves in every variant: 8/PAIRS chain groups x PAIRS slot groups of 64*SL slots; operands in LDS, one barrier per stage
// of SGK k-groups.  Reported per 64-slot x 16-chain unit (a pass covers PAIRS*SL of them).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#pragma clang fp contract(off)
typedef float f2 __attribute__((ext_vector_type(2)));
#define KG 512
template <int SL, int PAIRS, int SGK, int RINGS, int XW>
__global__ __launch_bounds__(512 + 64 * XW) void k(float *out, const float *in, int reps)
{
    extern __shared__ float4 lds[];
    constexpr int SLOTS = PAIRS * SL * 64;      // slots per stage
    constexpr int STAGE = SGK * SLOTS + 16 * SGK; // float4: x [kg][slot], then centroids [chain pair 0..7][kg][2]
    constexpr int RING = RINGS;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cg = wave % (8 / PAIRS), sg = wave / (8 / PAIRS); // chain group, slot group
    for (int i = threadIdx.x; i < RING * STAGE; i += 512 + 64 * XW) lds[i] = make_float4(in[i & 255], in[(i + 1) & 255], 0.5f, 0.25f);
    __syncthreads();
    f2 s[SL][PAIRS];
#pragma unroll
    for (int a = 0; a < SL; ++a)
#pragma unroll
        for (int p = 0; p < PAIRS; ++p) s[a][p] = f2{0.f, 0.f};
    if (wave >= 8) { // the loader waves' part: one barrier per stage
        for (int r = 0; r < reps; ++r)
            for (int st = 0; st < KG / SGK; ++st) __builtin_amdgcn_s_barrier();
        return;
    }
    for (int r = 0; r < reps; ++r)
        for (int st = 0; st < KG / SGK; ++st) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            const float4 *sb = lds + (st % RING) * STAGE;
            const float4 *xr = sb + sg * SL * 64 + lane, *ca = sb + SGK * SLOTS + (cg * PAIRS) * 2 * SGK;
#pragma unroll
            for (int g = 0; g < SGK; ++g) {
                float4 xv[SL], c0[PAIRS], c1[PAIRS];
#pragma unroll
                for (int a = 0; a < SL; ++a) xv[a] = xr[g * SLOTS + a * 64];
#pragma unroll
                for (int p = 0; p < PAIRS; ++p) {
                    c0[p] = ca[p * 2 * SGK + g * 2];
                    c1[p] = ca[p * 2 * SGK + g * 2 + 1];
                }
#pragma unroll
                for (int a = 0; a < SL; ++a) {
                    const f2 x0 = {xv[a].x, xv[a].x}, x1 = {xv[a].y, xv[a].y}, x2 = {xv[a].z, xv[a].z}, x3 = {xv[a].w, xv[a].w};
#pragma unroll
                    for (int p = 0; p < PAIRS; ++p) {
                        const f2 k0 = {c0[p].x, c0[p].y}, k1 = {c0[p].z, c0[p].w}, k2 = {c1[p].x, c1[p].y}, k3 = {c1[p].z, c1[p].w};
                        const f2 d0 = x0 - k0, d1 = x1 - k1, d2 = x2 - k2, d3 = x3 - k3;
                        const f2 q0 = d0 * d0, q1 = d1 * d1, q2 = d2 * d2, q3 = d3 * d3;
                        s[a][p] = s[a][p] + q0;
                        s[a][p] = s[a][p] + q1;
                        s[a][p] = s[a][p] + q2;
                        s[a][p] = s[a][p] + q3;
                    }
                }
            }
        }
#pragma unroll
    for (int a = 0; a < SL; ++a)
#pragma unroll
        for (int p = 0; p < PAIRS; ++p) {
            out[(((blockIdx.x * 512 + threadIdx.x) * SL + a) * PAIRS + p) * 2] = s[a][p].x;
            out[(((blockIdx.x * 512 + threadIdx.x) * SL + a) * PAIRS + p) * 2 + 1] = s[a][p].y;
        }
}
template <int SL, int PAIRS, int SGK, int RINGS, int XW>
static void run(float *out, const float *in)
{
    const size_t lds = RINGS * (size_t)(SGK * PAIRS * SL * 64 + 16 * SGK) * 16;
    if (hipFuncSetAttribute((const void *)k<SL, PAIRS, SGK, RINGS, XW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) { printf("lds %zu refused\n", lds); return; }
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    const int reps = 20;
    hipLaunchKernelGGL((k<SL, PAIRS, SGK, RINGS, XW>), dim3(256), dim3(512 + 64 * XW), lds, 0, out, in, 2);
    (void)hipEventRecord(a);
    hipLaunchKernelGGL((k<SL, PAIRS, SGK, RINGS, XW>), dim3(256), dim3(512 + 64 * XW), lds, 0, out, in, reps);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    printf("ring %d, extra waves %d, slots/lane %d, pairs/lane %d, %2d k-groups per stage (LDS %3zu KB): %.2f us per pass, %.2f us per 64-slot x 16-chain unit\n", RINGS, XW, SL, PAIRS, SGK,
           lds >> 10, ms * 1e3 / reps, ms * 1e3 / reps / (PAIRS * SL));
}
int main()
{
    float *out, *in;
    (void)hipMalloc(&out, (size_t)256 * 512 * 64 * 4);
    (void)hipMalloc(&in, 1024);
    std::vector<float> h(256);
    for (int i = 0; i < 256; ++i) h[i] = 0.001f * i;
    (void)hipMemcpy(in, h.data(), 1024, hipMemcpyHostToDevice);
    run<1, 4, 8, 2, 0>(out, in);
    run<1, 4, 8, 4, 0>(out, in);
    run<1, 4, 8, 2, 4>(out, in);
    run<1, 4, 8, 4, 4>(out, in);
    run<1, 1, 32, 2, 0>(out, in);
    run<1, 1, 32, 3, 4>(out, in);
    return 0;
}
