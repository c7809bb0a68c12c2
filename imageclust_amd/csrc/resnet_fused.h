// resnet_fused.h -- cross-layer fusions of the ResNet50-v1 forward pass (included by resnet.hip).
//
// The reference's forward pass (internal/embeddings/embeddings.go:141, Net.Forward inside OpenCV-DNN) fuses layers
// internally; the unfused graph of this engine is HBM-bound in its first quarter: the 112x112 stem output and
// the 56x56 stage-1 tensors are each written once and read back once per layer.  Two kernels remove those round trips:
//
//   stem_pool_kernel  conv0 7x7/2 + BN + ReLU + maxpool 3x3/2 in one launch: the 411 MB stem output never reaches HBM.
//   bneck56_kernel    one whole stage-1 bottleneck (1x1 -> 3x3 -> 1x1 (+ residual | + downsample branch) + ReLU) per
//                     launch: reads the block input once, writes the block output once (1 645 -> 822 MB per block).
#pragma once
#include "mfma_tile.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------------------------
// Stem + maxpool.  Work unit = (image, strip): a strip is 16 conv columns (14s-1 .. 14s+14, 15 of them used) =
// 7 pooled columns, so 8 strips cover the 56 pooled columns (1.14x recompute of the stem along x, none along y).
// A persistent workgroup walks a unit's 14 tiles of 8 conv rows top to bottom; tile t yields the pooled rows
// 4t .. 4t+3, whose windows need the conv rows 8t-1 .. 8t+7: the last conv row of a tile is kept in LDS for
// the next one.  Pooling takes the maximum of the ROUNDED conv outputs (rounding is monotonic: equal to rounding the
// maximum), so the result is bit-identical to conv0 -> store -> maxpool_kernel in both precisions.
// ------------------------------------------------------------------------------------------------------------
#define STEM_K 192
#define STEM_ROWK 24         /* k slots per filter row (21 used) */
#define STEM_PW 120          /* patch row stride in bytes: 1 (alignment) + 37 pixels * 3 = 112, + slack for the padded slots, 4-aligned */
#define STEM_PH 21
#define SP_STRIPS 8
#define SP_TILES 14

template <typename T>
static constexpr size_t stem_pool_lds_bytes()
{
    // [weights][2 activation buffers == pool tile][input patch][fp32 epilogue half tile][2 carried conv rows]
    return (((size_t)(STEM_K / T::BK) * 64 * CV_ROWB + 2 * (size_t)CV_BM * CV_ROWB + STEM_PH * STEM_PW + 16 + 15) & ~(size_t)15) + (size_t)64 * (64 + 4) * 4 +
           2 * 16 * 64 * sizeof(typename T::elem);
}

template <typename T>
__global__ __launch_bounds__(256) void stem_pool_kernel(const uint8_t *__restrict__ img, const conv_args p, int nunits)
{
    typedef typename T::elem elem;
    constexpr int BN = 64;
    constexpr int NKS = STEM_K / T::BK;            // k-steps: 3 (bf16) or 6 (f32)
    constexpr int WST = BN * CV_ROWB;              // bytes of one weight k-step image
    constexpr int XST = CV_BM * CV_ROWB;
    constexpr int PATCH = STEM_PH * STEM_PW + 16;  // + slack: chunk reads run past a row's last pixel
    constexpr int PROW = 64 * (int)sizeof(elem);   // one conv pixel's 64 channels in the pool tile
    static_assert(CV_BM * PROW <= 2 * XST, "the pool tile lives in the activation buffers");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *ptile = smem + NKS * WST; // 8 x 16 conv pixels after BN + ReLU, rounded (aliases the activation buffers)
    unsigned char *patch = smem + NKS * WST + 2 * XST;
    unsigned char *ep_smem = smem + ((NKS * WST + 2 * XST + PATCH + 15) & ~15);
    unsigned char *carry = ep_smem + 64 * (BN + 4) * 4; // [2][16 columns][64 channels]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid & 1, wn = wid >> 1;
    const elem *Wg = (const elem *)p.Wt;
    {   // all weights (64 x 192) by LDS-DMA, once per workgroup
        const int prow = lane >> 3, ps = lane & 7;
#pragma unroll
        for (int j = 0; j < NKS; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int row = wid * 16 + i * 8 + prow;
                __builtin_amdgcn_global_load_lds((gptr_t)(Wg + (int64_t)row * STEM_K + j * T::BK + lds_swz(row, ps) * T::KE),
                                                 (lptr_t)(smem + j * WST + (wid * 16 + i * 8) * CV_ROWB), 16, 0, 0);
            }
    }
    // input patch of tile (unit = image * 8 + strip, t): rows iy = 16t-3 .. +20, byte columns 3*(28s-5) .. ; zero outside the
    // image.  The first byte column, 84s - 15, is 1 mod 4 for every strip and rows / images are 672 / 150528 bytes apart, so the
    // patch is fetched as ALIGNED dwords from byte column 84s - 16 on and the chunk addresses below carry the 1-byte offset.
    constexpr int RW = STEM_PW / 4; // dwords per patch row
    static_assert(STEM_PW % 4 == 0, "patch rows are whole dwords");
    constexpr int NDW = STEM_PH * RW + 4;
    constexpr int NV = (NDW + 255) / 256;
    uint32_t pv[NV];
    auto patch_fetch = [&](int unit, int t) {
        const uint8_t *ib = img + (int64_t)(unit >> 3) * (int64_t)ICL_IMG_BYTES;
        const int iy0 = t * 16 - 3, bx0 = 84 * (unit & 7) - 16;
#pragma unroll
        for (int q = 0; q < NV; ++q) {
            const int i = tid + q * 256;
            const int pr = i / RW, pc = i - pr * RW;
            const int iy = iy0 + pr, bx = bx0 + pc * 4;
            const bool ok = i < NDW && pr < STEM_PH && (unsigned)iy < 224u && (unsigned)bx < 672u;
            pv[q] = ok ? *reinterpret_cast<const uint32_t *>(ib + iy * 672 + bx) : 0u;
        }
    };
    auto patch_store = [&]() {
        uint32_t *p32 = reinterpret_cast<uint32_t *>(patch);
#pragma unroll
        for (int q = 0; q < NV; ++q) {
            const int i = tid + q * 256;
            if (i < NDW) p32[i] = pv[q];
        }
    };
    // chunk roles: lane cuts the chunk (row, logical slot ls) for 4 tile rows; ls is fixed per lane
    const int ls = tid & 7;
    const float sc255 = (float)(1.0 / 255.0);
    auto gather = [&](int j, int buf) {
        const int k0 = j * T::BK + ls * T::KE;     // first k of the chunk; never straddles a filter row (24 % KE == 0)
        const int kh = k0 / STEM_ROWK, r0 = k0 - kh * STEM_ROWK;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = (tid >> 3) + 32 * i;
            const int oyl = row >> 4, oxl = row & 15;
            elem v[T::KE];
            if (kh < 7) {
                const int addr = (oyl * 2 + kh) * STEM_PW + oxl * 6 + r0 + 1; // + 1: the patch starts 1 byte left of the tile
                const uint32_t *w32 = reinterpret_cast<const uint32_t *>(patch + (addr & ~3));
                const uint32_t d0 = w32[0], d1 = w32[1], d2 = w32[2];
                const int sh = addr & 3;
                const uint32_t lo = __builtin_amdgcn_alignbyte(d1, d0, sh), hi = __builtin_amdgcn_alignbyte(d2, d1, sh);
#pragma unroll
                for (int e = 0; e < T::KE; ++e) {
                    const uint32_t byte = ((e < 4 ? lo : hi) >> (8 * (e & 3))) & 0xffu;
                    v[e] = T::from_f((float)byte * sc255);
                }
            } else {
#pragma unroll
                for (int e = 0; e < T::KE; ++e) v[e] = T::from_f(0.0f);
            }
            *reinterpret_cast<uint4 *>(smem + NKS * WST + buf * XST + row * CV_ROWB + (lds_swz(row, ls) << 4)) = *reinterpret_cast<const uint4 *>(v);
        }
    };
    const int fr = lane & 31, fh = lane >> 5;
    // epilogue roles (conv_epilogue's): one 16-byte channel chunk per lane
    constexpr int CPR = BN / T::KE, RPP = 256 / CPR, NPASS = 64 / RPP, EP_LD = BN + 4;
    const int nl = (tid % CPR) * T::KE;
    float sc[T::KE], sh[T::KE];
#pragma unroll
    for (int q = 0; q < T::KE; q += 4) {
        const float4 a4 = *reinterpret_cast<const float4 *>(p.scale + nl + q);
        const float4 b4 = *reinterpret_cast<const float4 *>(p.shift + nl + q);
        sc[q] = a4.x; sc[q + 1] = a4.y; sc[q + 2] = a4.z; sc[q + 3] = a4.w;
        sh[q] = b4.x; sh[q + 1] = b4.y; sh[q + 2] = b4.z; sh[q + 3] = b4.w;
    }
    float *ep = reinterpret_cast<float *>(ep_smem);
    elem *Yg = (elem *)p.Y; // pooled output [B][56][56][64]

    int unit = blockIdx.x, t = 0;
    if (unit >= nunits) return;
    patch_fetch(unit, t);
    patch_store();
    __syncthreads();
    for (;;) {
        int nt = t + 1, nu = unit;
        if (nt == SP_TILES) {
            nt = 0;
            nu = unit + (int)gridDim.x;
        }
        const bool has_next = nu < nunits; // workgroup-uniform
        f32x16 acc[1][2];
#pragma unroll
        for (int bb = 0; bb < 2; ++bb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[0][bb][r] = 0.0f;
        gather(0, 0);
        if (has_next) patch_fetch(nu, nt); // in flight while this tile is computed
        __syncthreads();                   // (the first time: also drains the weight DMA)
#pragma unroll
        for (int j = 0; j < NKS; ++j) {
            if (j + 1 < NKS) gather(j + 1, (j + 1) & 1);
            conv_mma_kstep<T, BN>(smem + j * WST, smem + NKS * WST + (j & 1) * XST, wm, wn, fr, fh, acc);
            __syncthreads();
        }
        if (has_next) patch_store(); // every gather of this tile has read the patch
        // ---- epilogue: BN + ReLU, rounded, into the pool tile (row r = conv pixel (r >> 4, r & 15) of the tile); the tile's
        // last conv row also goes to the carry buffer of the next tile
        unsigned char *carry_cur = carry + (t & 1) * 16 * PROW, *carry_nxt = carry + ((t + 1) & 1) * 16 * PROW;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            if (half) __syncthreads();
            if (wm == half) {
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const int ml = b * 32 + fr;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int nn = wn * 32 + 8 * g + 4 * fh;
                        *reinterpret_cast<float4 *>(ep + ml * EP_LD + nn) =
                            make_float4(acc[0][b][4 * g + 0], acc[0][b][4 * g + 1], acc[0][b][4 * g + 2], acc[0][b][4 * g + 3]);
                    }
                }
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < NPASS; ++i) {
                const int ml = tid / CPR + i * RPP;
                const int r = half * 64 + ml;
                float v[T::KE];
#pragma unroll
                for (int q = 0; q < T::KE; q += 4) {
                    const float4 tt = *reinterpret_cast<const float4 *>(ep + ml * EP_LD + nl + q);
                    v[q] = tt.x * sc[q] + sh[q];
                    v[q + 1] = tt.y * sc[q + 1] + sh[q + 1];
                    v[q + 2] = tt.z * sc[q + 2] + sh[q + 2];
                    v[q + 3] = tt.w * sc[q + 3] + sh[q + 3];
                }
                if (p.relu) {
#pragma unroll
                    for (int q = 0; q < T::KE; ++q) v[q] = fmaxf(v[q], 0.0f);
                }
                uint4 ov;
                elem *oe = reinterpret_cast<elem *>(&ov);
#pragma unroll
                for (int q = 0; q < T::KE; ++q) oe[q] = T::from_f(v[q]);
                *reinterpret_cast<uint4 *>(ptile + r * PROW + nl * (int)sizeof(elem)) = ov;
                if ((r >> 4) == 7) *reinterpret_cast<uint4 *>(carry_nxt + (r & 15) * PROW + nl * (int)sizeof(elem)) = ov;
            }
        }
        __syncthreads(); // the pool tile is complete
        // ---- maxpool 3x3/2 p1 (padding never wins): pooled rows 4t .. 4t+3, pooled columns 7s .. 7s+6
        {
            const int s = unit & 7;
            const int64_t b = unit >> 3;
            for (int it = tid; it < 28 * CPR; it += 256) {
                const int ch = it % CPR, pix = it / CPR, pr = pix / 7, pc = pix - 7 * pr;
                float best[T::KE];
#pragma unroll
                for (int e = 0; e < T::KE; ++e) best[e] = -INFINITY;
#pragma unroll
                for (int kh = 0; kh < 3; ++kh) {
                    const int rr = 2 * pr + kh; // 0: the carried conv row 8t-1; 1 .. 8: this tile's rows
                    if (rr == 0 && t == 0) continue;
                    const unsigned char *rowp = rr == 0 ? carry_cur : ptile + (rr - 1) * 16 * PROW;
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw) {
                        const int ci = 2 * pc + kw; // conv column 14s - 1 + ci
                        if (s == 0 && ci == 0) continue;
                        const uint4 raw = *reinterpret_cast<const uint4 *>(rowp + ci * PROW + ch * 16);
                        const elem *pvv = reinterpret_cast<const elem *>(&raw);
#pragma unroll
                        for (int e = 0; e < T::KE; ++e) {
                            const float f = T::to_f(pvv[e]);
                            if (f > best[e]) best[e] = f;
                        }
                    }
                }
                elem o[T::KE];
#pragma unroll
                for (int e = 0; e < T::KE; ++e) o[e] = T::from_f(best[e]);
                *reinterpret_cast<uint4 *>(Yg + ((b * 56 + 4 * t + pr) * 56 + 7 * s + pc) * 64 + ch * T::KE) = *reinterpret_cast<const uint4 *>(o);
            }
        }
        if (!has_next) break;
        unit = nu;
        t = nt;
        __syncthreads(); // the next patch is in LDS; the pool tile (activation buffer 0) and the epilogue tile are free again
    }
}

// ------------------------------------------------------------------------------------------------------------
// One stage-1 bottleneck per launch (bf16):  y = relu(bn3(conv3(relu(bn2(conv2_3x3(relu(bn1(conv1(x))))))) + x)           [identity]
//                                            y = relu(bn3(conv3(t2)) + bn_ds(conv_ds(x)))                               [DS: block 0]
// mid = 64 channels, Cout = 256, Cin = 256 (identity) or 64 (DS); any H, W.  Every BatchNorm scale is folded into the bf16
// weights at load time (W' = bf16(W * scale), as an inference engine does; the downsample fusion of round 1 already did), the
// shift is the accumulator's initial value: the epilogues are ReLU + convert only.
//
//   * One 512-thread workgroup per CU, two waves per SIMD.  Every WEIGHT lives in registers for the whole launch -- no weight
//     byte crosses L2 -> LDS after the prologue -- as A operands of v_mfma_f32_16x16x32_bf16; activations are the B operand, read
//     from LDS.  The waves form a two-stage pipeline inside the workgroup:
//       front waves 0..3 (W1, W2: rows 16w .. 16w+15): conv1 and conv2 of step j        -> t2[j & 1] in LDS
//       back  waves 4..7 (W3: rows 64b .. 64b+63):     conv3 + residual / downsample + ReLU + stores of step j-1
//     so on each SIMD an MFMA-heavy wave runs beside an epilogue-heavy one (alone on its SIMD a wave is bound by instruction issue:
//     ~10 vector / scalar instructions per 16-cycle MFMA; measured 13.7 us per step for the one-wave-per-SIMD form of this kernel).
//   * A workgroup owns a strip of 14 output columns of a run of images and walks it top to bottom in steps of 8 rows.  The
//     images of a run are stacked with ONE zero row between them (a "virtual" row index v: image v / (H+1), row v % (H+1),
//     row H = padding), so the walk never restarts: step j brings the x tile of virtual rows S+8j .. S+8j+7 (16 columns:
//     the strip + one halo column each side) in by LDS-DMA, conv1 turns it into 8 new rows of t1 behind the two rows kept
//     from the previous step (no halo recompute along y, 16/14 along x), conv2 produces t2 for the virtual rows
//     S+8j-1 .. S+8j+6 from the 10 t1 rows, conv3 (+ residual from L2 / + downsample operand straight from global memory into
//     the B fragments) writes y.  t1 and t2 never leave the CU; HBM sees x once and y once.
//   * The x tile of step j+1 is requested as soon as conv1 of step j has consumed the buffer and lands behind conv2.
//   * Three workgroup barriers per step (D: t2 / x tile complete, E: conv1 has read the x tile and the kept t1 rows are moved,
//     C: t1 written); the back waves join them between the halves of their work.
//   k order per output element: channels ascending for the 1x1s, (row, kw, channel) for the 3x3, [t2 | x] for DS: fixed,
//   so results do not depend on the batch or on the strip decomposition.
// ------------------------------------------------------------------------------------------------------------
struct bneck_args {
    const uint16_t *X;  // [B][H][W][CIN]
    uint16_t *Y;        // [B][H][W][256]
    const uint16_t *W1; // [64][CIN]        * scale1
    const uint16_t *W2; // [64][3][3][64]   * scale2
    const uint16_t *W3; // identity: [256][64] * scale3; DS: [256][64 + 64] = [W3*scale3 | Wds*scale_ds]
    const float *sh1, *sh2, *sh3; // folded BN shifts (DS: sh3 = shift3 + shift_ds)
    const void *zero;   // >= 16 zero bytes
    int B, H, W;
    int nstrips, ngroups; // grid = nstrips * ngroups; group g owns the images [g*B/ngroups, (g+1)*B/ngroups)
};

#define BN56_COLS 14                         /* output columns per strip */
#define BN56_T1_BYTES (11 * 16 * 128)        /* 10 rows of t1 (2 kept + 8 new) + 1 row of slack for the garbage columns' taps */
#define BN56_T2_BYTES (128 * 128)

template <bool DS>
static constexpr size_t bneck56_lds_bytes()
{
    return (size_t)(DS ? 1 : 4) * 128 * 128 + BN56_T1_BYTES + 2 * BN56_T2_BYTES;
}

__device__ __forceinline__ f32x4 mfma16(const uint4 &a, const uint4 &b, const f32x4 &c)
{
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
__device__ __forceinline__ uint32_t pack_bf16x2(float a, float b)
{
    return (uint32_t)BF16::from_f(a) | ((uint32_t)BF16::from_f(b) << 16);
}
// workgroup barrier that waits for this wave's LDS traffic only (the front waves' LDS-DMA stays in flight across it; their own
// counted wait covers it before the barrier that publishes the tile)
__device__ __forceinline__ void bn56_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// (image, row) of a virtual row cursor, advanced without divisions (VH = H + 1 rows per image, row H = padding)
struct bn56_row {
    int b, r;
    __device__ __forceinline__ void step(int VH)
    {
        if (++r == VH) {
            r = 0;
            ++b;
        }
    }
};

template <bool DS>
__global__ __launch_bounds__(512, 2) void bneck56_kernel(const bneck_args p)
{
    constexpr int CIN = DS ? 64 : 256;
    constexpr int NCH = CIN / 64;       // 64-channel chunks of x
    constexpr int K3 = DS ? 128 : 64;   // conv3's K: [t2 | x] or t2
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *XT = smem;                       // [NCH][128 pixels][128 B], swizzled
    unsigned char *T1 = smem + NCH * 16384;         // [11 rows][16 columns][128 B], swizzled
    unsigned char *T2 = T1 + BN56_T1_BYTES;         // [2][128 pixels][128 B], swizzled
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int w = wid & 3;
    const int r16 = lane & 15, q = lane >> 4;
    const int H = p.H, Wd = p.W, VH = H + 1;
    const int strip = (int)blockIdx.x % p.nstrips, grp = (int)blockIdx.x / p.nstrips;
    const int b0 = (int)(((int64_t)grp * p.B) / p.ngroups), b1 = (int)(((int64_t)(grp + 1) * p.B) / p.ngroups);
    if (b0 >= b1) return;
    const int c0 = strip * BN56_COLS;
    const int nsteps = ((b1 - b0) * VH + 7) / 8;
    // B-fragment reads: pixel px = 16 * row + column; (px >> 1) & 7 depends on the column only, so a lane's swizzled byte
    // offset inside a pixel row is fixed and every (row, chunk) is an immediate offset
    const int swc = (r16 >> 1) & 7;
    unsigned xoff[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) xoff[ks] = (unsigned)(r16 * 128 + ((((4 * ks + q) ^ swc) & 7) << 4));

    if (wid < 4) {
        // =========================================== front waves: conv1, conv2 ===========================================
        uint4 w1r[NCH * 2], w2r[18];
#pragma unroll
        for (int kk = 0; kk < NCH * 2; ++kk) w1r[kk] = *reinterpret_cast<const uint4 *>(p.W1 + (size_t)(16 * w + r16) * CIN + 32 * kk + 8 * q);
#pragma unroll
        for (int kk = 0; kk < 18; ++kk) w2r[kk] = *reinterpret_cast<const uint4 *>(p.W2 + (size_t)(16 * w + r16) * 576 + 32 * kk + 8 * q);
        const float4 h1 = *reinterpret_cast<const float4 *>(p.sh1 + 16 * w + 4 * q), h2 = *reinterpret_cast<const float4 *>(p.sh2 + 16 * w + 4 * q);
        unsigned t1off[3][2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int c = r16 + kw;
                t1off[kw][ks] = (unsigned)(c * 128 + ((((4 * ks + q) ^ (c >> 1)) & 7) << 4));
            }
        // t1 / t2 stores: this lane's 4 channels 16w + 4q .. of pixel (row, r16): 8 bytes in slot 2w + (q >> 1)
        const unsigned toff = (unsigned)(r16 * 128 + ((((2 * w + (q >> 1)) ^ swc) & 7) << 4) + 8 * (q & 1));
        const bool col_in = (unsigned)(c0 - 1 + r16) < (unsigned)Wd; // t1 column of this lane inside the image
        // LDS-DMA roles: wave w fills the tile rows 2w, 2w+1 (pieces of 8 pixels x 128 B): piece i of a row = columns 8i ..
        const int dcol = lane >> 3, dps = lane & 7;
        const unsigned xt_wave = __builtin_amdgcn_readfirstlane(lds_addr_of(smem) + (unsigned)w * 4096u);
        int dcoff[2]; // element offset of this lane's source chunk inside an image row, or -1: column outside the image
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c = 8 * i + dcol, col = c0 - 1 + c;
            dcoff[i] = (unsigned)col < (unsigned)Wd ? col * CIN + ((dps ^ (c >> 1)) & 7) * 8 : -1;
        }
        auto stage_x = [&](bn56_row t) { // the tile rows 2w, 2w+1 of the x tile whose row 2w is the virtual row t
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                const bool rok = t.b < b1 && t.r < H;
                const uint16_t *rowp = p.X + ((size_t)t.b * H + t.r) * (size_t)Wd * CIN;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int ch = 0; ch < NCH; ++ch) {
                        const void *src = (rok && dcoff[i] >= 0) ? (const void *)(rowp + dcoff[i] + 64 * ch) : p.zero;
                        glds16_asm(src, xt_wave + (unsigned)(ch * 16384 + (2 * rr + i) * 1024));
                    }
                t.step(VH);
            }
        };
        bn56_row cur{b0, 0}; // virtual row S + 8j: the first new t1 row of step j
        bn56_row dma = cur;  // ... + 2w: this wave's first DMA row
        for (int i = 0; i < 2 * w; ++i) dma.step(VH);
        if (tid < 256) reinterpret_cast<uint4 *>(T1)[tid] = make_uint4(0, 0, 0, 0); // the two kept t1 rows of step 0: padding
        stage_x(dma);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        bn56_barrier(); // D
        for (int j = 0; j < nsteps; ++j) {
            // keep the last two t1 rows of the previous step (rows 8, 9 -> 0, 1: same swizzle, the pixel index moves by 128)
            if (j > 0) reinterpret_cast<uint4 *>(T1)[tid] = reinterpret_cast<const uint4 *>(T1 + 16384)[tid];
            // ---- conv1: t1[rows 2..9] = relu(W1' . x + shift1)
            f32x4 a1[8];
#pragma unroll
            for (int m = 0; m < 8; ++m) a1[m] = f32x4{h1.x, h1.y, h1.z, h1.w};
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int m = 0; m < 8; ++m) {
                        const uint4 bf = *reinterpret_cast<const uint4 *>(XT + ch * 16384 + m * 2048 + xoff[ks]);
                        a1[m] = mfma16(w1r[ch * 2 + ks], bf, a1[m]);
                    }
            bn56_barrier(); // E: every front wave has read the x tile; the kept rows are in place
            for (int i = 0; i < 8; ++i) dma.step(VH);
            if (j + 1 < nsteps) stage_x(dma);
            {
                bn56_row t = cur;
#pragma unroll
                for (int m = 0; m < 8; ++m) {
                    const bool ok = t.b < b1 && t.r < H && col_in; // conv2 pads t1 with zeros, not with relu(shift1)
                    uint2 o;
                    o.x = ok ? pack_bf16x2(fmaxf(a1[m][0], 0.f), fmaxf(a1[m][1], 0.f)) : 0u;
                    o.y = ok ? pack_bf16x2(fmaxf(a1[m][2], 0.f), fmaxf(a1[m][3], 0.f)) : 0u;
                    *reinterpret_cast<uint2 *>(T1 + (2 + m) * 2048 + toff) = o;
                    t.step(VH);
                }
                cur = t;
            }
            bn56_barrier(); // C: t1 rows 2..9 are written
            // ---- conv2: t2 = relu(W2' * t1 + shift2): every t1 fragment (row R, kw, k half) is read once and feeds the taps kh = R - m
            f32x4 a2[8];
#pragma unroll
            for (int m = 0; m < 8; ++m) a2[m] = f32x4{h2.x, h2.y, h2.z, h2.w};
#pragma unroll
            for (int R = 0; R < 10; ++R)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw)
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        const uint4 bf = *reinterpret_cast<const uint4 *>(T1 + R * 2048 + t1off[kw][ks]);
#pragma unroll
                        for (int kh = 0; kh < 3; ++kh) {
                            const int m = R - kh;
                            if (m >= 0 && m < 8) a2[m] = mfma16(w2r[(kh * 3 + kw) * 2 + ks], bf, a2[m]);
                        }
                    }
            unsigned char *T2w = T2 + (j & 1) * BN56_T2_BYTES;
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                uint2 o;
                o.x = pack_bf16x2(fmaxf(a2[m][0], 0.f), fmaxf(a2[m][1], 0.f));
                o.y = pack_bf16x2(fmaxf(a2[m][2], 0.f), fmaxf(a2[m][3], 0.f));
                *reinterpret_cast<uint2 *>(T2w + m * 2048 + toff) = o;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // this wave's pieces of the next x tile have landed
            bn56_barrier(); // D: t2[j & 1] and the next x tile are complete; every front wave has read t1
        }
        // the back waves still work on the last step: join its two barriers (E, C)
        bn56_barrier();
        bn56_barrier();
    } else {
        // =========================================== back waves: conv3 + residual / downsample + ReLU + stores ===========================================
        uint4 w3r[4 * (K3 / 32)];
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int ks = 0; ks < K3 / 32; ++ks)
                w3r[g * (K3 / 32) + ks] = *reinterpret_cast<const uint4 *>(p.W3 + (size_t)(64 * w + 16 * g + r16) * K3 + 32 * ks + 8 * q);
        float4 h3[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) h3[g] = *reinterpret_cast<const float4 *>(p.sh3 + 64 * w + 16 * g + 4 * q);
        const bool col_out = r16 < BN56_COLS && c0 + r16 < Wd; // output column of this lane exists
        // operand of an output pixel that comes from global memory: the residual chunks (identity: 4 x 8 bytes) or the downsample
        // branch's B fragments (DS: 2 x 16 bytes)
        struct side_t {
            uint4 v[2];
        };
        bn56_row out{b0 - 1, H}; // virtual row S - 1 (the padding row above the run): the first output row of step 0
        auto load_side = [&](const bn56_row &t0, side_t (&sd)[4], bool active) { // the 4 rows from t0 on
            bn56_row t = t0;
#pragma unroll
            for (int mm = 0; mm < 4; ++mm) {
                const bool ok = active && t.b >= b0 && t.b < b1 && t.r < H && col_out;
                const size_t pix = ((size_t)t.b * H + t.r) * (size_t)Wd + c0 + r16;
                if constexpr (DS) {
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks)
                        sd[mm].v[ks] = *reinterpret_cast<const uint4 *>(ok ? p.X + pix * 64 + 32 * ks + 8 * q : (const uint16_t *)p.zero);
                } else {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const uint2 rv = *reinterpret_cast<const uint2 *>(ok ? p.X + pix * 256 + 64 * w + 16 * g + 4 * q : (const uint16_t *)p.zero);
                        if (g & 1) { sd[mm].v[g >> 1].z = rv.x; sd[mm].v[g >> 1].w = rv.y; }
                        else { sd[mm].v[g >> 1].x = rv.x; sd[mm].v[g >> 1].y = rv.y; }
                    }
                }
                t.step(VH);
            }
        };
        auto half = [&](const unsigned char *T2r, int hh, const bn56_row &t0, const side_t (&sd)[4]) { // output rows t0 .. t0+3 = tile rows 4hh ..
            f32x4 a3[4][4];
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int mm = 0; mm < 4; ++mm) {
                    if constexpr (DS) {
                        a3[g][mm] = f32x4{h3[g].x, h3[g].y, h3[g].z, h3[g].w};
                    } else { // shift + residual: the accumulator's initial value
                        const uint32_t lo = (g & 1) ? sd[mm].v[g >> 1].z : sd[mm].v[g >> 1].x, hi = (g & 1) ? sd[mm].v[g >> 1].w : sd[mm].v[g >> 1].y;
                        a3[g][mm] = f32x4{h3[g].x + __uint_as_float(lo << 16), h3[g].y + __uint_as_float(lo & 0xffff0000u),
                                          h3[g].z + __uint_as_float(hi << 16), h3[g].w + __uint_as_float(hi & 0xffff0000u)};
                    }
                }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int mm = 0; mm < 4; ++mm) {
                    const uint4 bf = *reinterpret_cast<const uint4 *>(T2r + (4 * hh + mm) * 2048 + xoff[ks]);
#pragma unroll
                    for (int g = 0; g < 4; ++g) a3[g][mm] = mfma16(w3r[g * (K3 / 32) + ks], bf, a3[g][mm]);
                }
            if constexpr (DS) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int mm = 0; mm < 4; ++mm)
#pragma unroll
                        for (int g = 0; g < 4; ++g) a3[g][mm] = mfma16(w3r[g * (K3 / 32) + 2 + ks], sd[mm].v[ks], a3[g][mm]);
            }
            bn56_row t = t0;
#pragma unroll
            for (int mm = 0; mm < 4; ++mm) {
                if (t.b >= b0 && t.b < b1 && t.r < H && col_out) {
                    uint16_t *dst = p.Y + (((size_t)t.b * H + t.r) * (size_t)Wd + c0 + r16) * 256 + 64 * w + 4 * q;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        uint2 o;
                        o.x = pack_bf16x2(fmaxf(a3[g][mm][0], 0.f), fmaxf(a3[g][mm][1], 0.f));
                        o.y = pack_bf16x2(fmaxf(a3[g][mm][2], 0.f), fmaxf(a3[g][mm][3], 0.f));
                        *reinterpret_cast<uint2 *>(dst + 16 * g) = o;
                    }
                }
                t.step(VH);
            }
        };
        side_t s0[4], s1[4];
        bn56_barrier(); // D (prologue)
        // iteration 0: the front waves compute step 0; request the first half's side operands of step 0 meanwhile
        bn56_barrier(); // E
        load_side(out, s0, true);
        bn56_barrier(); // C
        bn56_barrier(); // D: t2[0] is complete
        for (int j = 1; j <= nsteps; ++j) { // outputs of step j - 1
            const unsigned char *T2r = T2 + ((j - 1) & 1) * BN56_T2_BYTES;
            bn56_row mid = out;
#pragma unroll
            for (int i = 0; i < 4; ++i) mid.step(VH);
            load_side(mid, s1, true);
            half(T2r, 0, out, s0);
            bn56_barrier(); // E
            bn56_row nxt = mid;
#pragma unroll
            for (int i = 0; i < 4; ++i) nxt.step(VH);
            load_side(nxt, s0, j < nsteps); // the next step's first half
            bn56_barrier(); // C
            half(T2r, 1, mid, s1);
            out = nxt;
            if (j < nsteps) bn56_barrier(); // D (the front waves' last D belongs to step nsteps - 1; their two trailing barriers pair with this iteration's E, C)
        }
    }
}
