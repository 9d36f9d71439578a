// rwkvtts_amd/csrc/chunk_common.h -- building blocks of the chunked (MFMA) WKV7 kernels for gfx950.
//
// Formulation (per head, H = S^T in R^{K x V}, chunk of C = 32 steps, cumulative decay g_t = prod_{s<=t} w~_s):
//   q~ = q g_t, a~ = a g_{t-1}, k^ = k / g_t, b^ = b / g_t
//   A_ab[t,s] = a~_t.b^_s (s<t)  A_ak[t,s] = a~_t.k^_s (s<t)  A_qb[t,s] = q~_t.b^_s (s<=t)  A_qk[t,s] = q~_t.k^_s (s<=t)
//   U = (I - A_ab)^-1 (A~ H0 + A_ak V)        (u_t = sa_t of the scalar kernel)
//   Y = Q~ H0 + A_qb U + A_qk V               H_C = g_C * (H0 + B^^T U + K^^T V)
// validated against the scalar oracle on CPU by tools/chunked_proto.py (fp32 rel. err 3e-7; with the 2-way bf16
// operand split used here 5e-6 -- three orders below the bf16 rounding of the outputs).
//
// Every matrix lives in LDS as bf16 "planes" [rows][K + 8] (K contiguous; +8 elements of padding make the 16-byte
// fragment reads of 32 consecutive rows hit 32 distinct bank quads).  fp32 values are split x = hi + lo with
// hi = bf16(x), lo = bf16(x - hi); a product X Y^T is accumulated as Xh Yh + Xh Yl + Xl Yh on
// v_mfma_f32_32x32x16_bf16 (fp32 accumulate), i.e. ~16 mantissa bits per operand.  One primitive does all products:
//   D[m][n] = sum_k X[m][k] * Y[n][k]        X, Y row-major planes; D in the MFMA C/D layout
//   (lane: n = lane & 31, register r: m = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5))
// and a D tile is written back TRANSPOSED, OUT[n][m], so that it can be the X or Y operand of the next product.
#pragma once
#include "wkv7_common.h"

namespace rwkv7 {

using bf16x8 = __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16;
using f32x16 = __attribute__((__vector_size__(16 * sizeof(float)))) float;

constexpr int kC = 32;        // chunk length
constexpr int kPad = 8;       // plane row padding (elements)

__device__ __forceinline__ uint16_t bf_hi(float x) { return f2bf(x); }
__device__ __forceinline__ void split2(float x, uint16_t &hi, uint16_t &lo) {
    hi = f2bf(x);
    lo = f2bf(x - bf2f(hi));
}

// acc += X[m0 + (0..31)][0..K) . Y[n0 + (0..31)][0..K)^T     (single planes)
template <int K>
__device__ __forceinline__ void mma_tile(f32x16 &acc, const uint16_t *X, int ldx, const uint16_t *Y, int ldy, int lane) {
    const uint16_t *xp = X + (lane & 31) * ldx + (lane >> 5) * 8;
    const uint16_t *yp = Y + (lane & 31) * ldy + (lane >> 5) * 8;
#pragma unroll
    for (int k0 = 0; k0 < K; k0 += 16) {
        const bf16x8 fa = *reinterpret_cast<const bf16x8 *>(xp + k0);
        const bf16x8 fb = *reinterpret_cast<const bf16x8 *>(yp + k0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc, 0, 0, 0);
    }
}
// both operands split: Xh Yh + Xh Yl + Xl Yh
template <int K>
__device__ __forceinline__ void mma_tile3(f32x16 &acc, const uint16_t *Xh, const uint16_t *Xl, int ldx,
                                          const uint16_t *Yh, const uint16_t *Yl, int ldy, int lane) {
    mma_tile<K>(acc, Xh, ldx, Yh, ldy, lane);
    mma_tile<K>(acc, Xh, ldx, Yl, ldy, lane);
    mma_tile<K>(acc, Xl, ldx, Yh, ldy, lane);
}
// Y exact in bf16 (raw v / dy): Xh Y + Xl Y
template <int K>
__device__ __forceinline__ void mma_tile2x(f32x16 &acc, const uint16_t *Xh, const uint16_t *Xl, int ldx,
                                           const uint16_t *Y, int ldy, int lane) {
    mma_tile<K>(acc, Xh, ldx, Y, ldy, lane);
    mma_tile<K>(acc, Xl, ldx, Y, ldy, lane);
}
// X exact in bf16: X Yh + X Yl
template <int K>
__device__ __forceinline__ void mma_tile2y(f32x16 &acc, const uint16_t *X, int ldx, const uint16_t *Yh,
                                           const uint16_t *Yl, int ldy, int lane) {
    mma_tile<K>(acc, X, ldx, Yh, ldy, lane);
    mma_tile<K>(acc, X, ldx, Yl, ldy, lane);
}

// row index of accumulator register r for this lane
__device__ __forceinline__ int d_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// write the D tile transposed into hi/lo planes: OUT[n0 + n][m0 + m] = D[m][n]
__device__ __forceinline__ void store_T_split(const f32x16 &acc, uint16_t *Oh, uint16_t *Ol, int ld, int lane) {
    const int n = lane & 31, h = lane >> 5;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        uint16_t hi[4], lo[4];
#pragma unroll
        for (int i = 0; i < 4; i++) split2(acc[4 * j + i], hi[i], lo[i]);
        const int off = n * ld + 8 * j + 4 * h;
        *reinterpret_cast<uint2 *>(Oh + off) = make_uint2((uint32_t)hi[0] | ((uint32_t)hi[1] << 16), (uint32_t)hi[2] | ((uint32_t)hi[3] << 16));
        *reinterpret_cast<uint2 *>(Ol + off) = make_uint2((uint32_t)lo[0] | ((uint32_t)lo[1] << 16), (uint32_t)lo[2] | ((uint32_t)lo[3] << 16));
    }
}

// masks on D[m][n]: keep m < n (strict) or m <= n
template <bool STRICT>
__device__ __forceinline__ void mask_lower_T(f32x16 &acc, int lane) {
    // D[m][n] holds A[t = n][s = m]: keep s < t (STRICT) or s <= t
    const int n = lane & 31;
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int m = d_row(r, lane);
        const bool keep = STRICT ? (m < n) : (m <= n);
        acc[r] = keep ? acc[r] : 0.f;
    }
}

__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int r = 0; r < 16; r++) z[r] = 0.f;
    return z;
}

}  // namespace rwkv7
