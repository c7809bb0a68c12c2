// distance_mfma.hip -- K6: the pairwise Ward (half squared Euclidean) matrix in GEMM form on the matrix cores,
//     D~[i][j] = 0.5*(|e_i|^2 + |e_j|^2) - e_i . e_j          (north_star: "-2 E E^T on MFMA plus broadcast row-norms")
// for the FAST clustering mode (ICL_UPDATE_LW) and as a stand-alone entry point (icl_distance_mfma_dev).
//
// It replaces the arithmetic of ComputeInitialDistanceMatrix (/root/reference/internal/clustering/clustering.go:61-73)
// only approximately: the GEMM form sums in a different order and cancels, so its low bits differ from the reference's
// sequential sum (clustering.go:137-141,152-155).  The bit-exact tile is ward_dist_exact_kernel (ward.hip); cluster
// ids produced from D~ are reported, never asserted, against the reference.
//
// Precision: fp32 operands are split e = hi + lo with hi = bf16(e), lo = bf16(e - hi) and the dot product is
// hi.hi + hi.lo + lo.hi (the lo.lo term is below fp32 resolution): the three products are ONE bf16 GEMM over the
// concatenated K axis  A' = [hi | hi | lo],  B' = [hi | lo | hi]  (K' = 3*D), fp32 accumulate -> ~2^-16 relative to
// |e_i||e_j| per term.  Tiles, LDS image, LDS-DMA staging and the MFMA sweep are the convolution kernel's
// (mfma_tile.h): 128x128 outputs per workgroup, lower-triangle tiles only.
#include "icl_common.h"
#include "mfma_tile.h"

#include <algorithm>

// E [n][d] fp32 -> A' / B' [n][3*dp] bf16 (dp = d rounded up to 64, zero padded) and norms[n] = sum e^2 (fp32).
__global__ __launch_bounds__(256) void dist_split_kernel(const float *__restrict__ E, int64_t n, int d, int dp,
                                                        uint16_t *__restrict__ A, uint16_t *__restrict__ B, float *__restrict__ norms)
{
    __shared__ float red[4];
    const int64_t r = blockIdx.x;
    if (r >= n) return;
    float acc = 0.0f;
    for (int k = threadIdx.x; k < dp; k += 256) {
        const float e = k < d ? E[r * d + k] : 0.0f;
        const uint16_t hi = BF16::from_f(e);
        const uint16_t lo = BF16::from_f(e - BF16::to_f(hi));
        acc += e * e;
        uint16_t *a = A + r * 3 * dp, *b = B + r * 3 * dp;
        a[k] = hi; a[dp + k] = hi; a[2 * dp + k] = lo;
        b[k] = hi; b[dp + k] = lo; b[2 * dp + k] = hi;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) norms[r] = red[0] + red[1] + red[2] + red[3];
}

struct dist_args {
    const uint16_t *A, *B; // [n][K]
    const float *norms;
    const void *zero;
    float *out;
    const int64_t *rowoff; // packed lower triangle (MODE 0) or nullptr
    int64_t n, ld;
    int K;
};

__device__ __forceinline__ void tri_decode32(int64_t b, int &ti, int &tj)
{
    int64_t t = (int64_t)((sqrt(8.0 * (double)b + 1.0) - 1.0) * 0.5);
    while ((t + 1) * (t + 2) / 2 <= b) ++t;
    while (t * (t + 1) / 2 > b) --t;
    ti = (int)t;
    tj = (int)(b - t * (t + 1) / 2);
}

// MODE 0: packed lower triangle j < i at out[rowoff[i] + j];  MODE 1: dense rows, j <= i at out[i*ld + j].
template <int MODE>
__global__ __launch_bounds__(256) void dist_mfma_kernel(const dist_args p)
{
    constexpr int BN = 128;
    constexpr int STAGE = (BN + CV_BM) * CV_ROWB;
    constexpr int EP_LD = BN + 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid & 1, wn = wid >> 1;
    int ti, tj;
    tri_decode32(blockIdx.x, ti, tj);
    const int64_t i0 = (int64_t)ti * CV_BM, j0 = (int64_t)tj * BN;
    // staging roles: one LDS-DMA piece = 8 rows x 128 B; "x" rows = i (lane of the MFMA result), "w" rows = j
    const int prow = lane >> 3, ps = lane & 7;
    const uint16_t *xsrc[4], *wsrc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int row = wid * 32 + q * 8 + prow;
        const int64_t ri = i0 + row, rj = j0 + row;
        xsrc[q] = ri < p.n ? p.A + ri * p.K + lds_swz(row, ps) * 8 : nullptr;
        wsrc[q] = rj < p.n ? p.B + rj * p.K + lds_swz(row, ps) * 8 : nullptr;
    }
    const unsigned smem_base = __builtin_amdgcn_readfirstlane(lds_addr_of(smem));
    const unsigned wave_off = __builtin_amdgcn_readfirstlane(wid * 32 * CV_ROWB);
    int k0 = 0;
    auto stage = [&](int buf) {
        const unsigned wdst = smem_base + buf * STAGE + wave_off, xdst = wdst + BN * CV_ROWB;
#pragma unroll
        for (int q = 0; q < 4; ++q) glds16_asm(wsrc[q] ? (const void *)(wsrc[q] + k0) : p.zero, wdst + q * 8 * CV_ROWB);
#pragma unroll
        for (int q = 0; q < 4; ++q) glds16_asm(xsrc[q] ? (const void *)(xsrc[q] + k0) : p.zero, xdst + q * 8 * CV_ROWB);
        k0 += 64;
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;
    const int nk = p.K / 64;
    stage(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int fr = lane & 31, fh = lane >> 5;
    for (int ks = 0; ks < nk; ++ks) {
        const int cur = ks & 1;
        if (ks + 1 < nk) stage(cur ^ 1);
        const unsigned char *wsm = smem + cur * STAGE;
        conv_mma_kstep<BF16, BN>(wsm, wsm + BN * CV_ROWB, wm, wn, fr, fh, acc);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    // accumulators (lane = i, register quads = 4 consecutive j) -> fp32 LDS tile [i][j] -> coalesced rows
    float *ep = reinterpret_cast<float *>(smem);
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int ml = wm * 64 + b * 32 + fr;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int nl = wn * 64 + a * 32 + 8 * g + 4 * fh;
                *reinterpret_cast<float4 *>(ep + ml * EP_LD + nl) =
                    make_float4(acc[a][b][4 * g + 0], acc[a][b][4 * g + 1], acc[a][b][4 * g + 2], acc[a][b][4 * g + 3]);
            }
    }
    __syncthreads();
    const int nl = (tid & 31) * 4; // 32 lanes x 4 floats = one 128-wide tile row; 8 rows per pass
    float nj[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) nj[q] = (j0 + nl + q) < p.n ? p.norms[j0 + nl + q] : 0.0f;
    for (int ml = tid >> 5; ml < CV_BM; ml += 8) {
        const int64_t i = i0 + ml;
        if (i >= p.n) break;
        const float ni = p.norms[i];
        const float4 t = *reinterpret_cast<const float4 *>(ep + ml * EP_LD + nl);
        const float v[4] = {0.5f * (ni + nj[0]) - t.x, 0.5f * (ni + nj[1]) - t.y, 0.5f * (ni + nj[2]) - t.z, 0.5f * (ni + nj[3]) - t.w};
        const int64_t jb = j0 + nl;
        const int64_t lim = MODE == 0 ? i : i + 1; // columns j < lim are written
        float *row = MODE == 0 ? p.out + p.rowoff[i] : p.out + i * p.ld;
        if (jb + 3 < i && ((MODE == 0) || ((p.ld & 3) == 0))) { // strictly below the diagonal: one 16-byte store
            *reinterpret_cast<float4 *>(row + jb) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (jb + q < lim) row[jb + q] = (MODE == 1 && jb + q == i) ? 0.0f : v[q];
        }
    }
}

static size_t dist_lds_bytes()
{
    const size_t stages = 2 * (size_t)(128 + CV_BM) * CV_ROWB, ep = (size_t)CV_BM * (128 + 4) * 4;
    return std::max(stages, ep);
}

// Shared by icl_distance_mfma_dev and the LW clustering mode (ward.hip).  Scratch (A', B', norms, zero page) is
// allocated per call and released after the stream drains.
int icl_dist_mfma_launch(icl_ctx *ctx, const float *d_E, int64_t n, int d, float *d_out, const int64_t *d_rowoff, int64_t ld)
{
    if (n <= 0) return ICL_OK;
    const int dp = (d + 63) / 64 * 64;
    const int K = 3 * std::max(dp, 64);
    const int dpe = K / 3;
    uint16_t *A = nullptr, *B = nullptr;
    float *norms = nullptr;
    void *zero = nullptr;
    ICL_HIP(ctx, hipMalloc((void **)&A, (size_t)n * K * 2));
    ICL_HIP(ctx, hipMalloc((void **)&B, (size_t)n * K * 2));
    ICL_HIP(ctx, hipMalloc((void **)&norms, (size_t)n * 4));
    ICL_HIP(ctx, hipMalloc(&zero, 256));
    ICL_HIP(ctx, hipMemsetAsync(zero, 0, 256, ctx->stream));
    hipLaunchKernelGGL(dist_split_kernel, dim3((unsigned)n), dim3(256), 0, ctx->stream, d_E, n, d, dpe, A, B, norms);
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void *)dist_mfma_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dist_lds_bytes());
        (void)hipFuncSetAttribute((const void *)dist_mfma_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dist_lds_bytes());
        attr_done = true;
    }
    dist_args a{A, B, norms, zero, d_out, d_rowoff, n, ld, K};
    const int64_t nt = icl_ceil_div(n, CV_BM), nblocks = nt * (nt + 1) / 2;
    if (nblocks > 0x7fffffffLL) return icl_fail(ctx, ICL_ERR_UNSUPPORTED, "distance tile grid too large");
    {
        const double pairs = (double)n * (double)(n + 1) * 0.5;
        icl_prof_scope ps(ctx, ICL_K_DIST_MFMA, 2.0 * pairs * d, 4.0 * pairs + 4.0 * (double)n * d);
        if (d_rowoff)
            hipLaunchKernelGGL(dist_mfma_kernel<0>, dim3((unsigned)nblocks), dim3(256), dist_lds_bytes(), ctx->stream, a);
        else
            hipLaunchKernelGGL(dist_mfma_kernel<1>, dim3((unsigned)nblocks), dim3(256), dist_lds_bytes(), ctx->stream, a);
    }
    ICL_HIP(ctx, hipGetLastError());
    ICL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    (void)hipFree(A);
    (void)hipFree(B);
    (void)hipFree(norms);
    (void)hipFree(zero);
    icl_prof_collect(ctx);
    return ICL_OK;
}

extern "C" int icl_distance_mfma_dev(icl_ctx *ctx, const float *d_E, int64_t n, int32_t d, float *d_D, int64_t ld)
{
    if (!ctx || n < 0 || d < 0 || ld < n || (n && (!d_E || !d_D))) return icl_fail(ctx, ICL_ERR_ARG, "icl_distance_mfma_dev: bad argument");
    std::lock_guard<std::mutex> lk(ctx->mu);
    icl_device_guard g(ctx->device);
    return icl_dist_mfma_launch(ctx, d_E, n, d, d_D, nullptr, ld);
}
