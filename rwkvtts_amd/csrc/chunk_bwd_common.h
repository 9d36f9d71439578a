// rwkvtts_amd/csrc/chunk_bwd_common.h -- helpers of the chunked WKV7 per-chunk gradient kernel (wkv7_chunk_bwd9.hip).
#pragma once
#include "chunk_common.h"

namespace rwkv7 {
namespace {
constexpr int LDK = kN + kPad;  // planes with 64 contiguous elements per row
constexpr int LDC = kC + kPad;  // planes with 32 contiguous elements per row

// keep D[m][n] where m >= n (STRICT: m > n)
template <bool STRICT>
__device__ __forceinline__ void mask_upper_T(f32x16 &acc, int lane) {
    const int n = lane & 31;
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int m = d_row(r, lane);
        const bool keep = STRICT ? (m > n) : (m >= n);
        acc[r] = keep ? acc[r] : 0.f;
    }
}

// X split, Y exact (single plane)
template <int K, int CH = kMmaChains>
__device__ __forceinline__ void mma_xs_ye(f32x16 &acc, const uint16_t *Xh, const uint16_t *Xl, int ldx, const uint16_t *Y,
                                          int ldy, int lane) {
    constexpr int NK = K / 16;
    const int xo = (lane & 31) * ldx + (lane >> 5) * 8, yo = (lane & 31) * ldy + (lane >> 5) * 8;
    bf16x8 xh[NK], xl[NK], y[NK];
#pragma unroll
    for (int i = 0; i < NK; i++) {
        xh[i] = *reinterpret_cast<const bf16x8 *>(Xh + xo + 16 * i);
        y[i] = *reinterpret_cast<const bf16x8 *>(Y + yo + 16 * i);
        xl[i] = *reinterpret_cast<const bf16x8 *>(Xl + xo + 16 * i);
    }
    __builtin_amdgcn_sched_barrier(0);  // fragment loads stay above, MFMAs below
    f32x16 acc_b = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NK; i++) {
        if (CH == 2 && (i & 1)) acc_b = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh[i], y[i], acc_b, 0, 0, 0);
        else acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh[i], y[i], acc, 0, 0, 0);
        if (CH == 2 && ((i + 1) & 1)) acc_b = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xl[i], y[i], acc_b, 0, 0, 0);
        else acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xl[i], y[i], acc, 0, 0, 0);
    }
    if (CH == 2) acc += acc_b;
}

struct Raw8 {
    uint4 r;
};
__device__ __forceinline__ Raw8 ld8(const bf16_t *p) {
    Raw8 o;
    o.r = *reinterpret_cast<const uint4 *>(p);
    return o;
}
__device__ __forceinline__ void cvt8(const Raw8 &x, float (&f)[8]) {
    f[0] = __uint_as_float(x.r.x << 16); f[1] = __uint_as_float(x.r.x & 0xffff0000u);
    f[2] = __uint_as_float(x.r.y << 16); f[3] = __uint_as_float(x.r.y & 0xffff0000u);
    f[4] = __uint_as_float(x.r.z << 16); f[5] = __uint_as_float(x.r.z & 0xffff0000u);
    f[6] = __uint_as_float(x.r.w << 16); f[7] = __uint_as_float(x.r.w & 0xffff0000u);
}

// 8 fp32 -> hi/lo bf16, stored as one 16-byte row segment of a time-major plane pair
__device__ __forceinline__ void put_row8(uint16_t *Ph, uint16_t *Pl, int off, const float (&x)[8], uint32_t (&hi)[4],
                                         uint32_t (&lo)[4]) {
#pragma unroll
    for (int j = 0; j < 4; j++) split_pk(x[2 * j], x[2 * j + 1], hi[j], lo[j]);
    *reinterpret_cast<uint4 *>(Ph + off) = make_uint4(hi[0], hi[1], hi[2], hi[3]);
    *reinterpret_cast<uint4 *>(Pl + off) = make_uint4(lo[0], lo[1], lo[2], lo[3]);
}
// G[j] = sum_{s <= pt} lw_s[pk + j]
__device__ __forceinline__ void chunk_cumsum(const float (&lw)[8], float (&G)[8]) {
#pragma unroll
    for (int j = 0; j < 8; j++) G[j] = scan32(lw[j]);
}

// T (or T^T) of the chunk, fp32 [32][32] in global memory -> bf16 hi/lo planes [32][LDC]; thread tid holds T[tid>>3][4(tid&7)..]
template <bool TRANSPOSE>
__device__ __forceinline__ void put_tm(const float4 x, uint16_t *Th, uint16_t *Tl, int tid) {
    const int tr = tid >> 3, tc = (tid & 7) * 4;
    if (!TRANSPOSE) {
        uint32_t h0, l0, h1, l1;
        split_pk(x.x, x.y, h0, l0);
        split_pk(x.z, x.w, h1, l1);
        *reinterpret_cast<uint2 *>(Th + tr * LDC + tc) = make_uint2(h0, h1);
        *reinterpret_cast<uint2 *>(Tl + tr * LDC + tc) = make_uint2(l0, l1);
    } else {
        const float xs[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
        for (int j = 0; j < 4; j++) split2(xs[j], Th[(tc + j) * LDC + tr], Tl[(tc + j) * LDC + tr]);
    }
}

}  // namespace
}  // namespace rwkv7
