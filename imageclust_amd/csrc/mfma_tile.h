// mfma_tile.h -- the shared MFMA tile machinery of libimageclust_hip.so: element traits (bf16 / f32 MFMA step), the
// swizzled 128-byte-row LDS image, inline-asm LDS-DMA staging, the per-k-step MFMA sweep of a 128 x BN tile and the
// XCD-aware tile order.  Used by the convolution kernels (resnet.hip) and the MFMA distance tile (distance_mfma.hip).
#pragma once
#include "icl_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

// ------------------------------------------------------------------------------------------------------------
// element traits
// ------------------------------------------------------------------------------------------------------------
struct BF16 {
    typedef uint16_t elem;
    static constexpr int KE = 8;  // elements per 16-byte chunk
    static constexpr int BK = 64; // elements per 128-byte LDS row
    __device__ static __forceinline__ float to_f(elem v) { return __uint_as_float((uint32_t)v << 16); }
    __device__ static __forceinline__ elem from_f(float f)
    {
        // plain cast: hipcc emits v_cvt_pk_bf16_f32 (round to nearest even, NaN stays NaN;
        // MI355X_MICROARCH.md "Correctness boundaries")
        const __bf16 h = (__bf16)f;
        return __builtin_bit_cast(elem, h);
    }
    __device__ static __forceinline__ void mma(const uint4 &w, const uint4 &x, f32x16 &acc)
    {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, w), __builtin_bit_cast(bf16x8_t, x), acc, 0, 0, 0);
    }
};
struct F32 {
    typedef float elem;
    static constexpr int KE = 4;
    static constexpr int BK = 32;
    __device__ static __forceinline__ float to_f(elem v) { return v; }
    __device__ static __forceinline__ elem from_f(float f) { return f; }
    __device__ static __forceinline__ void mma(const uint4 &w, const uint4 &x, f32x16 &acc)
    {
        // lane half h holds k = 4h..4h+3 of this 8-deep slice; MFMA q pairs k=q (h=0) with k=4+q (h=1) on both operands
        const float4 wf = __builtin_bit_cast(float4, w), xf = __builtin_bit_cast(float4, x);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wf.x, xf.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wf.y, xf.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wf.z, xf.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wf.w, xf.w, acc, 0, 0, 0);
    }
};

#define CV_BM 128
#define CV_ROWB 128 /* bytes per LDS row = one k-step of one tile row: 64 bf16 or 32 f32 */

// LDS image of a [rows][128 B] tile: logical 16-byte slot s of row r lives at r*128 + ((s ^ ((r>>1)&7)) * 16).
// ds_read_b128 serves fixed 16-lane groups (rows {0-3,12-15,20-27} / {4-11,16-19,28-31} of a 32-row fragment, all
// reading the same logical slot): the XOR spreads them over the 16 distinct 16-byte positions of two 256-byte bank
// rows -> conflict-free (MI355X_MICROARCH.md "LDS").  The tile is filled by LDS-DMA, whose destination is
// lane-linear (base + lane*16), so the swizzle is applied to each lane's SOURCE address (which k-chunk it fetches)
// and again on the read: the same involution on both sides (cdna_hip_programming.md rule 21).
__device__ __forceinline__ int lds_swz(int row, int slot) { return (slot ^ (row >> 1)) & 7; }

typedef const void __attribute__((address_space(1))) *gptr_t;
typedef void __attribute__((address_space(3))) *lptr_t;

// LDS-DMA piece issued from inline asm: 64 lanes x 16 B -> 1 KiB at the wave-uniform LDS byte address `lds_dst`
// (M0 is written and restored inside the statement).  hipcc does not track it: unlike the builtin it puts no
// vmcnt(0) in front of the next ds_read, so the transfer really overlaps the MFMAs of the current k-step; the
// kernel waits for it itself (s_waitcnt vmcnt(0) + barrier) before the stage is read
// (cdna_hip_programming.md 5.7 "LDS-DMA recipe").
__device__ __forceinline__ void glds16_asm(const void *gsrc, unsigned lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}
__device__ __forceinline__ unsigned lds_addr_of(const void *p)
{
    return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const void *)p;
}

// XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs (private L2s).  Give each XCD a contiguous
// run of tiles with the Cout tile index fastest, so the workgroups that share an activation row-panel (and the
// whole weight matrix) sit behind one L2.  Bijective for any grid size.
__device__ __forceinline__ int xcd_remap(int bid, int nwg)
{
    const int xcd = bid & 7, j = bid >> 3, q = nwg >> 3, r = nwg & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
}

// One k-step (128 bytes of K per tile row) of MFMAs from a staged LDS image: wsm = BN weight rows, xsm = 128 pixel rows.
template <typename T, int BN>
__device__ __forceinline__ void conv_mma_kstep(const unsigned char *wsm, const unsigned char *xsm, int wm, int wn, int fr, int fh,
                                               f32x16 (&acc)[BN / 64][2])
{
    constexpr int NT = BN / 64;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        uint4 wf[NT], xf[2];
#pragma unroll
        for (int a = 0; a < NT; ++a) {
            const int row = wn * (BN / 2) + a * 32 + fr;
            wf[a] = *reinterpret_cast<const uint4 *>(wsm + row * CV_ROWB + (lds_swz(row, 2 * s + fh) << 4));
        }
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int row = wm * 64 + b * 32 + fr;
            xf[b] = *reinterpret_cast<const uint4 *>(xsm + row * CV_ROWB + (lds_swz(row, 2 * s + fh) << 4));
        }
#pragma unroll
        for (int a = 0; a < NT; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) T::mma(wf[a], xf[b], acc[a][b]);
    }
}

