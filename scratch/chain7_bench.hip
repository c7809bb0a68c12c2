// chain7_bench.hip -- centroids through DPP instead of LDS broadcast reads.  The 64-slot block's chain wave needs three ds_read_b128
// per k-group (x + two broadcast centroid quads) for 12 v_pk, which keeps the LDS array as busy as the vector ALU.  Here the
// wave's 256 centroid floats of a stage ([k][chain A/B], 32 k-groups) are read ONCE per stage with four non-broadcast
// ds_read_b128 (every 16-lane row holds the whole table, 16 floats per lane) and each difference is
//   v_sub_f32_dpp d, c_reg, x  row_newbcast:n        (d = c[e] - x: the sign is squared away, the rounding is that of x - c)
// then {dA, dB} are squared and added with the packed ops as before: 16 VALU per k-group instead of 12, one LDS read instead of 3.
// MODE 0 = the kernel's loop (broadcast reads), MODE 1 = DPP.  8 chain waves, ring of 3 stages, one barrier per stage.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#pragma clang fp contract(off)
typedef float f2 __attribute__((ext_vector_type(2)));
#define SG 32
#define KG 512
#define RING 3
#define STAGE (SG * 64 + 16 * SG)
template <int N>
__device__ __forceinline__ float sub_bcast(float c, float x)
{
    float d;
    asm("v_sub_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "=v"(d) : "v"(c), "v"(x), "i"(N));
    return d;
}
template <int E>
__device__ __forceinline__ float dsub(const float4 (&ct)[4], float x)
{
    // element E of the stage table lives in register (E / 64, E % 4) of lane (E % 64) / 4 of every row
    constexpr int q = E / 64, n = (E % 64) / 4, c4 = E % 4;
    const float c = c4 == 0 ? ct[q].x : c4 == 1 ? ct[q].y : c4 == 2 ? ct[q].z : ct[q].w;
    return sub_bcast<n>(c, x);
}
template <int G>
__device__ __forceinline__ void kgroup_dpp(const float4 (&ct)[4], const float4 xv, f2 &s)
{
    constexpr int E = 8 * G; // k_local = 4G + kk, element = 2 k_local + ab
    const f2 d0 = {dsub<E + 0>(ct, xv.x), dsub<E + 1>(ct, xv.x)}, d1 = {dsub<E + 2>(ct, xv.y), dsub<E + 3>(ct, xv.y)};
    const f2 d2 = {dsub<E + 4>(ct, xv.z), dsub<E + 5>(ct, xv.z)}, d3 = {dsub<E + 6>(ct, xv.w), dsub<E + 7>(ct, xv.w)};
    const f2 q0 = d0 * d0, q1 = d1 * d1, q2 = d2 * d2, q3 = d3 * d3;
    s = s + q0;
    s = s + q1;
    s = s + q2;
    s = s + q3;
}
template <int G0, int N>
struct kg_loop {
    static __device__ __forceinline__ void run(const float4 (&ct)[4], const float4 *xr, f2 &s)
    {
        kgroup_dpp<G0>(ct, xr[G0 * 64], s);
        kg_loop<G0 + 1, N>::run(ct, xr, s);
    }
};
template <int N>
struct kg_loop<N, N> {
    static __device__ __forceinline__ void run(const float4 (&)[4], const float4 *, f2 &) {}
};
template <int MODE>
__global__ __launch_bounds__(768) void k(float *out, const float *in, int reps)
{
    extern __shared__ float4 lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < RING * STAGE; i += 768) lds[i] = make_float4(in[i & 255], in[(i + 1) & 255], 0.5f, 0.25f);
    __syncthreads();
    if (wave >= 8) { // the loader waves' part: one barrier per stage
        for (int r = 0; r < reps; ++r)
            for (int st = 0; st < KG / SG; ++st) __builtin_amdgcn_s_barrier();
        return;
    }
    f2 s = {0.f, 0.f};
    for (int r = 0; r < reps; ++r)
        for (int st = 0; st < KG / SG; ++st) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            const float4 *sb = lds + (st % RING) * STAGE;
            const float4 *xr = sb + lane, *ca = sb + SG * 64 + wave * 2 * SG; // this wave's pair: [k-group][2] quads = 64 float4
            if (MODE == 0) {
#pragma unroll
                for (int g = 0; g < SG; ++g) {
                    const float4 xv = xr[g * 64], c0 = ca[2 * g], c1 = ca[2 * g + 1];
                    const f2 x0 = {xv.x, xv.x}, x1 = {xv.y, xv.y}, x2 = {xv.z, xv.z}, x3 = {xv.w, xv.w};
                    const f2 k0 = {c0.x, c0.y}, k1 = {c0.z, c0.w}, k2 = {c1.x, c1.y}, k3 = {c1.z, c1.w};
                    const f2 d0 = x0 - k0, d1 = x1 - k1, d2 = x2 - k2, d3 = x3 - k3;
                    const f2 q0 = d0 * d0, q1 = d1 * d1, q2 = d2 * d2, q3 = d3 * d3;
                    s = s + q0;
                    s = s + q1;
                    s = s + q2;
                    s = s + q3;
                }
            } else {
                float4 ct[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) ct[q] = ca[q * 16 + (lane & 15)]; // floats 64q + 4n + {0..3} in lane n of every row
                kg_loop<0, SG>::run(ct, xr, s);
            }
        }
    out[(blockIdx.x * 512 + threadIdx.x) * 2] = s.x;
    out[(blockIdx.x * 512 + threadIdx.x) * 2 + 1] = s.y;
}
template <int MODE>
static void run(float *out, const float *in, std::vector<float> *res)
{
    const size_t lds = (size_t)RING * STAGE * 16;
    (void)hipFuncSetAttribute((const void *)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    const int reps = 20;
    hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(768), lds, 0, out, in, 1);
    (void)hipDeviceSynchronize();
    res->resize(1024);
    (void)hipMemcpy(res->data(), out, 4096, hipMemcpyDeviceToHost);
    (void)hipEventRecord(a);
    hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(768), lds, 0, out, in, reps);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    printf("%s: %.2f us per 64-slot x 16-chain block\n", MODE ? "centroids by DPP row_newbcast (4 reads per stage)" : "centroids by broadcast ds_read_b128      ", ms * 1e3 / reps);
}
int main()
{
    float *out, *in;
    (void)hipMalloc(&out, (size_t)256 * 768 * 8);
    (void)hipMalloc(&in, 1024);
    std::vector<float> h(256), r0, r1;
    for (int i = 0; i < 256; ++i) h[i] = 0.001f * i + 0.37f * (i % 7);
    (void)hipMemcpy(in, h.data(), 1024, hipMemcpyHostToDevice);
    run<0>(out, in, &r0);
    run<1>(out, in, &r1);
    int bad = 0;
    for (int i = 0; i < 1024; ++i) bad += r0[i] != r1[i];
    printf("results %s (%d of 1024 sums differ)\n", bad ? "DIFFER" : "bit-identical", bad);
    return 0;
}
