// pk_rate_bench.hip -- issue rate of packed fp32 ops on gfx950: cycles per wave-instruction on one SIMD with W waves per SIMD, for
// independent v_pk_add_f32 / v_pk_mul_f32, the same with the op_sel broadcast + neg modifiers the update kernel uses, a dependent
// v_pk_add chain, and the kernel's (sub, mul, add) pattern with 1 or 4 running sums.  Operands in registers only.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
#pragma clang fp contract(off)
#define ITER 4096
template <int MODE>
__global__ void k(float *out, const float *in, int iters)
{
    f2 a[8], c = {in[1], in[2]};
    float x = in[threadIdx.x & 15];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = f2{in[i], in[i + 1]};
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) { // independent pk_add
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
        } else if (MODE == 1) { // independent pk_add with broadcast + neg (x - c)
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_pk_add_f32 %0, %0, %1 op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]" : "+v"(a[i]) : "v"(c));
        } else if (MODE == 2) { // independent pk_mul
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
        } else if (MODE == 3) { // dependent pk_add chain
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[0]) : "v"(c));
        } else if (MODE == 4) { // scalar v_add_f32 independent
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i].x) : "v"(c.x));
        } else if (MODE == 5) { // the kernel's pattern, ONE running sum: 8 x (sub, mul, add)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                f2 d, q;
                asm volatile("v_pk_add_f32 %0, %1, %2 op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a[1 + (i & 3)]), "v"(c));
                asm volatile("v_pk_mul_f32 %0, %1, %1" : "=v"(q) : "v"(d));
                asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[0]) : "v"(q));
            }
        } else if (MODE == 6) { // the same with FOUR running sums
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                f2 d, q;
                asm volatile("v_pk_add_f32 %0, %1, %2 op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a[4 + (i & 3)]), "v"(c));
                asm volatile("v_pk_mul_f32 %0, %1, %1" : "=v"(q) : "v"(d));
                asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i & 3]) : "v"(q));
            }
        } else if (MODE == 8) { // independent v_pk_fma_f32
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(c));
        } else if (MODE == 9) { // the kernel's pattern written with v_pk_fma_f32 only (x*1 - c, d*d + 0, q*1 + s: the same roundings)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                f2 d, q;
                asm volatile("v_pk_fma_f32 %0, %1, 1.0, %2 op_sel_hi:[0,0,1] neg_lo:[0,0,1] neg_hi:[0,0,1]" : "=v"(d) : "v"(a[1 + (i & 3)]), "v"(c));
                asm volatile("v_pk_fma_f32 %0, %1, %1, 0" : "=v"(q) : "v"(d));
                asm volatile("v_pk_fma_f32 %0, %1, 1.0, %0 op_sel_hi:[1,0,1]" : "+v"(a[0]) : "v"(q));
            }
        } else if (MODE == 10) { // scalar v_fma_f32 independent
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i].x) : "v"(c.x));
        } else if (MODE == 7) { // scalar version of the pattern for two chains: 6 scalar ops per k
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float d0, d1, q0, q1;
                asm volatile("v_sub_f32 %0, %1, %2" : "=v"(d0) : "v"(a[1 + (i & 3)].x), "v"(c.x));
                asm volatile("v_sub_f32 %0, %1, %2" : "=v"(d1) : "v"(a[1 + (i & 3)].x), "v"(c.y));
                asm volatile("v_mul_f32 %0, %1, %1" : "=v"(q0) : "v"(d0));
                asm volatile("v_mul_f32 %0, %1, %1" : "=v"(q1) : "v"(d1));
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[0].x) : "v"(q0));
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[0].y) : "v"(q1));
            }
        }
    }
    float r = x;
#pragma unroll
    for (int i = 0; i < 8; ++i) r += a[i].x + a[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int MODE>
static void run(const char *name, int instr_per_iter, float *out, const float *in)
{
    for (int waves = 4; waves <= 16; waves *= 2) { // waves per CU (1, 2, 4 per SIMD)
        hipEvent_t a, b;
        (void)hipEventCreate(&a);
        (void)hipEventCreate(&b);
        hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(waves * 64), 0, 0, out, in, 16);
        (void)hipEventRecord(a);
        hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(waves * 64), 0, 0, out, in, ITER);
        (void)hipEventRecord(b);
        (void)hipEventSynchronize(b);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, a, b);
        const double clk = ms * 1e-3 * 2.4e9; // at the nominal 2.4 GHz
        printf("%-58s %d waves/SIMD: %.2f clk per wave-instruction per SIMD\n", name, waves / 4, clk / ((double)ITER * instr_per_iter * (waves / 4)));
    }
}
int main()
{
    float *out, *in;
    (void)hipMalloc(&out, 256 * 1024 * 4);
    (void)hipMalloc(&in, 1024);
    float h[256];
    for (int i = 0; i < 256; ++i) h[i] = 1.0f + 0.001f * i;
    (void)hipMemcpy(in, h, 1024, hipMemcpyHostToDevice);
    run<0>("v_pk_add_f32, 8 independent", 8, out, in);
    run<1>("v_pk_add_f32 op_sel_hi broadcast + neg, 8 independent", 8, out, in);
    run<2>("v_pk_mul_f32, 8 independent", 8, out, in);
    run<3>("v_pk_add_f32, one dependent chain", 8, out, in);
    run<4>("v_add_f32, 8 independent", 8, out, in);
    run<5>("(pk sub, pk mul, pk add) one running sum", 24, out, in);
    run<6>("(pk sub, pk mul, pk add) four running sums", 24, out, in);
    run<7>("scalar (2 sub, 2 mul, 2 add) one running pair", 48, out, in);
    run<8>("v_pk_fma_f32, 8 independent", 8, out, in);
    run<9>("(pk fma x*1-c, pk fma d*d+0, pk fma q*1+s) one running sum", 24, out, in);
    run<10>("v_fma_f32, 8 independent", 8, out, in);
    return 0;
}
