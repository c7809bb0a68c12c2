// rowscan_bench.hip -- what a merge-loop row re-scan costs on one workgroup, on an otherwise idle GPU.
// A 100k x 100k float matrix (40 GB, like the benchmark's), one packed side word per column (L2-resident), G workgroups of T threads,
// each scanning R rows (random row numbers) one after the other with the predicate of scan_row_m.  Variants:
//   0 = the engine's loop (4 x 16-byte loads of both streams in flight per lane, wait, visit)
//   1 = the same with the next group's loads requested before the current group is visited (software pipeline, 2 x 4 in flight)
//   2 = 8 x 16-byte loads in flight, no pipeline
//   3 = values only (no side stream), 4 in flight: the floor of the value stream
// Prints microseconds per row.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define MAXF 3.40282346638528859811704183484516925e+38f
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ void visit(float v, uint32_t k, int szb, uint32_t smask, int my_size, int max_size, int my_id, float &bv, int &bi)
{
    const int m = (int)(k & smask), c = (int)(k >> szb);
    if (m > 0 && m + my_size <= max_size && c < my_id && (v < bv || (v == bv && c < bi))) {
        bv = v;
        bi = c;
    }
}

template <int VAR>
__global__ __launch_bounds__(1024) void k(const float *__restrict__ D, int64_t ld, const uint32_t *__restrict__ mpk, const int *__restrict__ rows, int R,
                                          int64_t len, int max_size, float *out, int *outi)
{
    const int szb = 32 - __clz(max_size);
    const uint32_t smask = (1u << szb) - 1u;
    __shared__ float sv[16];
    __shared__ int si[16];
    for (int it = 0; it < R; ++it) {
        const int r = rows[blockIdx.x * R + it];
        const float *row = D + (int64_t)r * ld;
        const int my_id = 1 << 30, my_size = 1;
        float bv = MAXF;
        int bi = -1;
        const int64_t nvec = len >> 2;
        if (VAR == 0 || VAR == 3) {
            for (int64_t q0 = threadIdx.x; q0 < nvec; q0 += 4 * (int64_t)blockDim.x) {
                float4 v[4];
                uint4 kk[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int64_t q = q0 + (int64_t)j * blockDim.x;
                    const bool has = q < nvec;
                    v[j] = has ? reinterpret_cast<const float4 *>(row)[q] : make_float4(MAXF, MAXF, MAXF, MAXF);
                    kk[j] = (VAR == 0 && has) ? reinterpret_cast<const uint4 *>(mpk)[q] : make_uint4(1, 1, 1, 1);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    visit(v[j].x, kk[j].x, szb, smask, my_size, max_size, my_id, bv, bi);
                    visit(v[j].y, kk[j].y, szb, smask, my_size, max_size, my_id, bv, bi);
                    visit(v[j].z, kk[j].z, szb, smask, my_size, max_size, my_id, bv, bi);
                    visit(v[j].w, kk[j].w, szb, smask, my_size, max_size, my_id, bv, bi);
                }
            }
        } else if (VAR == 2) {
            for (int64_t q0 = threadIdx.x; q0 < nvec; q0 += 8 * (int64_t)blockDim.x) {
                float4 v[8];
                uint4 kk[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int64_t q = q0 + (int64_t)j * blockDim.x;
                    const bool has = q < nvec;
                    v[j] = has ? reinterpret_cast<const float4 *>(row)[q] : make_float4(MAXF, MAXF, MAXF, MAXF);
                    kk[j] = has ? reinterpret_cast<const uint4 *>(mpk)[q] : make_uint4(0, 0, 0, 0);
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    visit(v[j].x, kk[j].x, szb, smask, my_size, max_size, my_id, bv, bi);
                    visit(v[j].y, kk[j].y, szb, smask, my_size, max_size, my_id, bv, bi);
                    visit(v[j].z, kk[j].z, szb, smask, my_size, max_size, my_id, bv, bi);
                    visit(v[j].w, kk[j].w, szb, smask, my_size, max_size, my_id, bv, bi);
                }
            }
        } else if (VAR == 1) {
            float4 v[4], nv[4];
            uint4 kk[4], nk[4];
            auto load = [&](int64_t q0, float4 *vv, uint4 *kq) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int64_t q = q0 + (int64_t)j * blockDim.x;
                    const bool has = q < nvec;
                    vv[j] = has ? reinterpret_cast<const float4 *>(row)[q] : make_float4(MAXF, MAXF, MAXF, MAXF);
                    kq[j] = has ? reinterpret_cast<const uint4 *>(mpk)[q] : make_uint4(0, 0, 0, 0);
                }
            };
            load(threadIdx.x, v, kk);
            for (int64_t q0 = threadIdx.x; q0 < nvec; q0 += 4 * (int64_t)blockDim.x) {
                load(q0 + 4 * (int64_t)blockDim.x, nv, nk);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    visit(v[j].x, kk[j].x, szb, smask, my_size, max_size, my_id, bv, bi);
                    visit(v[j].y, kk[j].y, szb, smask, my_size, max_size, my_id, bv, bi);
                    visit(v[j].z, kk[j].z, szb, smask, my_size, max_size, my_id, bv, bi);
                    visit(v[j].w, kk[j].w, szb, smask, my_size, max_size, my_id, bv, bi);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    v[j] = nv[j];
                    kk[j] = nk[j];
                }
            }
        }
        // workgroup reduce (as block_argmin)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float ov = __shfl_down(bv, off, 64);
            const int oi = __shfl_down(bi, off, 64);
            if (ov < bv || (ov == bv && oi >= 0 && (bi < 0 || oi < bi))) {
                bv = ov;
                bi = oi;
            }
        }
        const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
        if (lane == 0) {
            sv[wid] = bv;
            si[wid] = bi;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            float b = sv[0];
            int bj = si[0];
            for (int w = 1; w < nw; ++w)
                if (sv[w] < b || (sv[w] == b && si[w] >= 0 && (bj < 0 || si[w] < bj))) {
                    b = sv[w];
                    bj = si[w];
                }
            out[blockIdx.x * R + it] = b;
            outi[blockIdx.x * R + it] = bj;
        }
        __syncthreads();
    }
}

__global__ void fill(float *D, int64_t total)
{
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x)
        D[i] = 1.0f + (float)((i * 2654435761ull) & 0xffff) * 1e-3f;
}

template <int VAR>
static void run(const char *name, int G, int T, int R, const float *D, int64_t ld, const uint32_t *mpk, const int *rows, int64_t len, float *out, int *outi)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    hipLaunchKernelGGL((k<VAR>), dim3(G), dim3(T), 0, 0, D, ld, mpk, rows, 1, len, 50, out, outi);
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((k<VAR>), dim3(G), dim3(T), 0, 0, D, ld, mpk, rows + G, R, len, 50, out, outi);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    printf("%-46s G=%3d T=%4d len=%6lld: %7.1f us per row (%.1f GB/s per workgroup)\n", name, G, T, (long long)len, ms * 1e3 / R,
           (double)len * (VAR == 3 ? 4 : 8) * R / (ms * 1e-3) / 1e9);
}

int main(int argc, char **argv)
{
    const int64_t n = argc > 1 ? atoll(argv[1]) : 100000, ld = (n + 63) / 64 * 64;
    float *D;
    uint32_t *mpk;
    int *rows, *outi;
    float *out;
    CK(hipMalloc(&D, (size_t)(n * ld) * 4));
    CK(hipMalloc(&mpk, (size_t)ld * 4));
    const int G = 256, R = 32;
    CK(hipMalloc(&rows, (size_t)(G * (R + 1)) * 4));
    CK(hipMalloc(&out, (size_t)(G * R) * 4));
    CK(hipMalloc(&outi, (size_t)(G * R) * 4));
    hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, D, n * ld);
    std::vector<uint32_t> hk((size_t)ld);
    for (int64_t i = 0; i < ld; ++i) hk[(size_t)i] = (uint32_t)((i << 6) | ((i % 3) ? 1 : 0)); // a third of the columns dead
    CK(hipMemcpy(mpk, hk.data(), (size_t)ld * 4, hipMemcpyHostToDevice));
    std::vector<int> hr((size_t)(G * (R + 1)));
    uint64_t s = 12345;
    for (auto &x : hr) {
        s = s * 6364136223846793005ull + 1442695040888963407ull;
        x = (int)((s >> 33) % (uint64_t)n);
    }
    CK(hipMemcpy(rows, hr.data(), hr.size() * 4, hipMemcpyHostToDevice));
    CK(hipDeviceSynchronize());
    for (int g : {1, 48, 192}) {
        run<0>("engine loop (4+4 x 16 B in flight)", g, 768, R, D, ld, mpk, rows, n, out, outi);
        run<1>("software pipelined (2 x (4+4))", g, 768, R, D, ld, mpk, rows, n, out, outi);
        run<2>("8+8 in flight", g, 768, R, D, ld, mpk, rows, n, out, outi);
        run<3>("values only, 4 in flight", g, 768, R, D, ld, mpk, rows, n, out, outi);
        run<0>("engine loop, 1024 threads", g, 1024, R, D, ld, mpk, rows, n, out, outi);
        run<2>("8+8 in flight, 1024 threads", g, 1024, R, D, ld, mpk, rows, n, out, outi);
        run<0>("engine loop, 256 threads", g, 256, R, D, ld, mpk, rows, n, out, outi);
    }
    run<0>("engine loop, half row", 48, 768, R, D, ld, mpk, rows, n / 2, out, outi);
    run<0>("engine loop, quarter row", 192, 768, R, D, ld, mpk, rows, n / 4, out, outi);
    run<0>("engine loop, 1/8 row, 256 thr", 192, 256, R, D, ld, mpk, rows, n / 8, out, outi);
    return 0;
}
