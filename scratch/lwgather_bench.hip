// lwgather_bench.hip -- what the reads of a Lance-Williams row update cost on the recycled triangular matrix (40 GB at N=100k):
// for P parents p (random rows) and L live clusters x (a random subset of [0,N)), one thread per x reads the pair's entry from the row
// of the younger of the two: x younger than p -> D[x*ld + p] (one 4-byte read per row: scattered), else D[p*ld + x] (contiguous over x).
// All P loads of a thread are in flight together.  Prints microseconds per launch.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
template <int P>
__global__ __launch_bounds__(256) void k(const float *__restrict__ D, int64_t ld, const int *__restrict__ par, const int *__restrict__ live, int L, float *out, int it)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= L) return;
    const int x = live[i];
    float v[P];
#pragma unroll
    for (int j = 0; j < P; ++j) {
        const int p = par[it * P + j];
        v[j] = x > p ? D[(int64_t)x * ld + p] : D[(int64_t)p * ld + x];
    }
    float s = 0;
#pragma unroll
    for (int j = 0; j < P; ++j) s += v[j];
    out[i] = s;
}
__global__ void fill(float *D, int64_t total)
{
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) D[i] = (float)(i & 1023);
}
int main(int argc, char **argv)
{
    const int64_t n = argc > 1 ? atoll(argv[1]) : 100000, ld = (n + 63) / 64 * 64;
    float *D, *out;
    int *par, *live;
    CK(hipMalloc(&D, (size_t)(n * ld) * 4));
    CK(hipMalloc(&out, (size_t)n * 4));
    const int IT = 64;
    CK(hipMalloc(&par, IT * 32 * 4));
    CK(hipMalloc(&live, (size_t)n * 4));
    hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, D, n * ld);
    std::vector<int> hp(IT * 32), hl((size_t)n);
    uint64_t s = 99;
    auto rnd = [&]() { s = s * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(s >> 33); };
    for (auto &x : hp) x = (int)(rnd() % (uint64_t)n);
    for (int64_t i = 0; i < n; ++i) hl[(size_t)i] = (int)i;
    for (int64_t i = n - 1; i > 0; --i) std::swap(hl[(size_t)i], hl[(size_t)(rnd() % (uint64_t)(i + 1))]);
    CK(hipMemcpy(par, hp.data(), hp.size() * 4, hipMemcpyHostToDevice));
    for (int L : {100000, 55000, 20000}) {
        std::vector<int> sub(hl.begin(), hl.begin() + L);
        std::sort(sub.begin(), sub.end()); // live clusters in slot order ~ id order
        CK(hipMemcpy(live, sub.data(), (size_t)L * 4, hipMemcpyHostToDevice));
        hipEvent_t a, b;
        CK(hipEventCreate(&a));
        CK(hipEventCreate(&b));
        hipLaunchKernelGGL((k<32>), dim3((L + 255) / 256), dim3(256), 0, 0, D, ld, par, live, L, out, 0);
        CK(hipEventRecord(a));
        for (int it = 1; it < IT; ++it) hipLaunchKernelGGL((k<32>), dim3((L + 255) / 256), dim3(256), 0, 0, D, ld, par, live, L, out, it);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, a, b));
        printf("32 parents x %6d live clusters: %7.1f us per launch (%.2f G entries/s)\n", L, ms * 1e3 / (IT - 1), 32.0 * L * (IT - 1) / (ms * 1e-3) / 1e9);
    }
    return 0;
}
