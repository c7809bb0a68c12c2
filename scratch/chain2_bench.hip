// chain-wave inner loop variants for the batched update kernel: x and c come from LDS, s += (x-c)^2 strictly in order
#include <hip/hip_runtime.h>
#include <cstdio>
#pragma clang fp contract(off)
typedef float f2 __attribute__((ext_vector_type(2)));
#define NG 512
template <int V>
__global__ __launch_bounds__(64) void k(float *out, const float *in)
{
    __shared__ float4 xs[24][64];
    __shared__ float4 cs[NG + 32];
    for (int g = 0; g < 24; ++g) xs[g][threadIdx.x] = make_float4(in[threadIdx.x], 1.f, 2.f, 3.f);
    for (int g = threadIdx.x; g < NG + 32; g += 64) cs[g] = make_float4(in[g & 63], 0.5f, 0.25f, 0.125f);
    __syncthreads();
    float s = 0;
    const int lane = threadIdx.x;
    if (V == 0) { // natural, packed
        for (int st = 0; st < NG; st += 12) {
            float4 xv[12], cv[12];
#pragma unroll
            for (int g = 0; g < 12; ++g) { xv[g] = xs[(st + g) % 24][lane]; cv[g] = cs[st + g]; }
#pragma unroll
            for (int g = 0; g < 12; ++g) {
                const f2 xa = {xv[g].x, xv[g].y}, xb = {xv[g].z, xv[g].w}, ca = {cv[g].x, cv[g].y}, cb = {cv[g].z, cv[g].w};
                const f2 da = xa - ca, db = xb - cb, qa = da * da, qb = db * db;
                s = s + qa.x; s = s + qa.y; s = s + qb.x; s = s + qb.y;
            }
        }
    } else if (V == 1) { // natural, scalar
        for (int st = 0; st < NG; st += 12) {
            float4 xv[12], cv[12];
#pragma unroll
            for (int g = 0; g < 12; ++g) { xv[g] = xs[(st + g) % 24][lane]; cv[g] = cs[st + g]; }
#pragma unroll
            for (int g = 0; g < 12; ++g) {
                float d0 = xv[g].x - cv[g].x, d1 = xv[g].y - cv[g].y, d2 = xv[g].z - cv[g].z, d3 = xv[g].w - cv[g].w;
                asm volatile("" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));
                float q0 = d0 * d0, q1 = d1 * d1, q2 = d2 * d2, q3 = d3 * d3;
                asm volatile("" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3));
                s = s + q0; s = s + q1; s = s + q2; s = s + q3;
            }
        }
    } else if (V == 2) { // interleaved, packed: squares of group g+1 between the adds of group g
        f2 qa = {0, 0}, qb = {0, 0};
        for (int st = 0; st < NG; st += 12) {
            float4 xv[12], cv[12];
#pragma unroll
            for (int g = 0; g < 12; ++g) { xv[g] = xs[(st + g) % 24][lane]; cv[g] = cs[st + g]; }
#pragma unroll
            for (int g = 0; g < 12; ++g) {
                const f2 xa = {xv[g].x, xv[g].y}, xb = {xv[g].z, xv[g].w}, ca = {cv[g].x, cv[g].y}, cb = {cv[g].z, cv[g].w};
                s = s + qa.x;
                const f2 da = xa - ca;
                __builtin_amdgcn_sched_barrier(0);
                s = s + qa.y;
                const f2 db = xb - cb;
                __builtin_amdgcn_sched_barrier(0);
                s = s + qb.x;
                const f2 na = da * da;
                __builtin_amdgcn_sched_barrier(0);
                s = s + qb.y;
                const f2 nb = db * db;
                __builtin_amdgcn_sched_barrier(0);
                qa = na; qb = nb;
            }
        }
        s = s + qa.x; s = s + qa.y; s = s + qb.x; s = s + qb.y;
    } else if (V == 3) { // interleaved, scalar
        float q0 = 0, q1 = 0, q2 = 0, q3 = 0;
        for (int st = 0; st < NG; st += 12) {
            float4 xv[12], cv[12];
#pragma unroll
            for (int g = 0; g < 12; ++g) { xv[g] = xs[(st + g) % 24][lane]; cv[g] = cs[st + g]; }
#pragma unroll
            for (int g = 0; g < 12; ++g) {
                s = s + q0; float d0 = xv[g].x - cv[g].x; float d1 = xv[g].y - cv[g].y; __builtin_amdgcn_sched_barrier(0);
                s = s + q1; float d2 = xv[g].z - cv[g].z; float d3 = xv[g].w - cv[g].w; __builtin_amdgcn_sched_barrier(0);
                s = s + q2; float n0 = d0 * d0; float n1 = d1 * d1; __builtin_amdgcn_sched_barrier(0);
                s = s + q3; float n2 = d2 * d2; float n3 = d3 * d3; __builtin_amdgcn_sched_barrier(0);
                q0 = n0; q1 = n1; q2 = n2; q3 = n3;
            }
        }
        s = s + q0; s = s + q1; s = s + q2; s = s + q3;
    } else if (V == 4) { // adds only (q from LDS directly): the floor
        for (int st = 0; st < NG; st += 12) {
            float4 xv[12];
#pragma unroll
            for (int g = 0; g < 12; ++g) xv[g] = xs[(st + g) % 24][lane];
#pragma unroll
            for (int g = 0; g < 12; ++g) { s = s + xv[g].x; s = s + xv[g].y; s = s + xv[g].z; s = s + xv[g].w; }
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = s;
}
int main()
{
    float *out, *in;
    hipMalloc(&out, 4 * 64 * 4096);
    hipMalloc(&in, 4 * 128);
    hipMemset(in, 0, 4 * 128);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    auto timeit = [&](const char *name, auto launch) {
        launch();
        hipDeviceSynchronize();
        hipEventRecord(a);
        for (int i = 0; i < 200; ++i) launch();
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        printf("%-44s %8.2f us/launch\n", name, ms * 1e3 / 200);
    };
    for (int blocks : {1, 157}) {
        printf("-- %d blocks of one wave, 512 groups (2048 k)\n", blocks);
        timeit("V0 natural packed", [&] { hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(64), 0, 0, out, in); });
        timeit("V1 natural scalar", [&] { hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(64), 0, 0, out, in); });
        timeit("V2 interleaved packed", [&] { hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(64), 0, 0, out, in); });
        timeit("V3 interleaved scalar", [&] { hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(64), 0, 0, out, in); });
        timeit("V4 adds only", [&] { hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(64), 0, 0, out, in); });
    }
    return 0;
}
