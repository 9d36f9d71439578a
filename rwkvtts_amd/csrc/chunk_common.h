// rwkvtts_amd/csrc/chunk_common.h -- building blocks of the chunked (MFMA) WKV7 kernels for gfx950.
//
// Formulation (per head, H = S^T in R^{K x V}, chunk of C = 32 steps, cumulative decay g_t = prod_{s<=t} w~_s):
//   q~ = q g_t, a~ = a g_{t-1}, k^ = k / g_t, b^ = b / g_t
//   A_ab[t,s] = a~_t.b^_s (s<t)  A_ak[t,s] = a~_t.k^_s (s<t)  A_qb[t,s] = q~_t.b^_s (s<=t)  A_qk[t,s] = q~_t.k^_s (s<=t)
//   U = (I - A_ab)^-1 (A~ H0 + A_ak V)        (u_t = sa_t of the scalar kernel)
//   Y = Q~ H0 + A_qb U + A_qk V               H_C = g_C * (H0 + B^^T U + K^^T V)
// validated against the scalar oracle on CPU by tests/chunked_proto.py (fp32 rel. err 3e-7; with the 2-way bf16
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
// Number of independent accumulator chains inside one tile product.  A lone wave per SIMD cannot hide the latency of
// back-to-back dependent MFMAs; two chains (even / odd MFMAs, summed at the end) do, at the price of 16 more registers.
// Per translation unit (define WKV7C_MMA_CHAINS before including) or per call (last template argument).
#ifndef WKV7C_MMA_CHAINS
#define WKV7C_MMA_CHAINS 1
#endif
constexpr int kMmaChains = WKV7C_MMA_CHAINS;

typedef __bf16 bf2_t __attribute__((ext_vector_type(2)));
typedef float f2_t __attribute__((ext_vector_type(2)));
// two fp32 -> packed bf16 pair (low half = a), round-to-nearest-even: one v_cvt_pk_bf16_f32
__device__ __forceinline__ uint32_t cvt_pk(float a, float b) {
    const f2_t v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf2_t));
}
// x = hi + lo with hi = bf16(x), lo = bf16(x - hi), for a pair: 2 cvt_pk + shift + and + 2 sub
__device__ __forceinline__ void split_pk(float a, float b, uint32_t &hi, uint32_t &lo) {
    hi = cvt_pk(a, b);
    lo = cvt_pk(a - __uint_as_float(hi << 16), b - __uint_as_float(hi & 0xffff0000u));
}
__device__ __forceinline__ void split2(float x, uint16_t &hi, uint16_t &lo) {
    uint32_t h, l;
    split_pk(x, 0.f, h, l);
    hi = (uint16_t)h;
    lo = (uint16_t)l;
}

// acc += X[m0 + (0..31)][0..K) . Y[n0 + (0..31)][0..K)^T     (single planes)
template <int K, int CH = kMmaChains>
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
// both operands split: Xh Yh + Xh Yl + Xl Yh.  All fragments are fetched first, then the MFMAs issue back to back
// (hipcc otherwise interleaves each ds_read pair with its dependent MFMA and the lone wave eats the LDS latency
// K/16 * 3 times per product).
template <int K, int CH = kMmaChains>
__device__ __forceinline__ void mma_tile3(f32x16 &acc, const uint16_t *Xh, const uint16_t *Xl, int ldx,
                                          const uint16_t *Yh, const uint16_t *Yl, int ldy, int lane) {
    constexpr int NK = K / 16;
    const int xo = (lane & 31) * ldx + (lane >> 5) * 8, yo = (lane & 31) * ldy + (lane >> 5) * 8;
    bf16x8 xh[NK], xl[NK], yh[NK], yl[NK];
#pragma unroll
    for (int i = 0; i < NK; i++) {
        xh[i] = *reinterpret_cast<const bf16x8 *>(Xh + xo + 16 * i);
        yh[i] = *reinterpret_cast<const bf16x8 *>(Yh + yo + 16 * i);
        xl[i] = *reinterpret_cast<const bf16x8 *>(Xl + xo + 16 * i);
        yl[i] = *reinterpret_cast<const bf16x8 *>(Yl + yo + 16 * i);
    }
    __builtin_amdgcn_sched_barrier(0);  // fragment loads stay above, MFMAs below
    f32x16 acc_b = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};  // CH == 2: second, independent MFMA chain
#pragma unroll
    for (int i = 0; i < NK; i++) {
        if (CH == 2 && ((i + 0) & 1)) acc_b = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh[i], yh[i], acc_b, 0, 0, 0); else acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh[i], yh[i], acc, 0, 0, 0);
        if (CH == 2 && ((i + 1) & 1)) acc_b = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh[i], yl[i], acc_b, 0, 0, 0); else acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh[i], yl[i], acc, 0, 0, 0);
        if (CH == 2 && ((i + 0) & 1)) acc_b = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xl[i], yh[i], acc_b, 0, 0, 0); else acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xl[i], yh[i], acc, 0, 0, 0);
    }
    if (CH == 2) acc += acc_b;
}
// Y exact in bf16 (raw v / dy): Xh Y + Xl Y
template <int K, int CH = kMmaChains>
__device__ __forceinline__ void mma_tile2x(f32x16 &acc, const uint16_t *Xh, const uint16_t *Xl, int ldx,
                                           const uint16_t *Y, int ldy, int lane) {
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
    f32x16 acc_b = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};  // CH == 2: second, independent MFMA chain
#pragma unroll
    for (int i = 0; i < NK; i++) {
        if (CH == 2 && ((i + 0) & 1)) acc_b = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh[i], y[i], acc_b, 0, 0, 0); else acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh[i], y[i], acc, 0, 0, 0);
        if (CH == 2 && ((i + 1) & 1)) acc_b = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xl[i], y[i], acc_b, 0, 0, 0); else acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xl[i], y[i], acc, 0, 0, 0);
    }
    if (CH == 2) acc += acc_b;
}
// X exact in bf16: X Yh + X Yl
template <int K, int CH = kMmaChains>
__device__ __forceinline__ void mma_tile2y(f32x16 &acc, const uint16_t *X, int ldx, const uint16_t *Yh,
                                           const uint16_t *Yl, int ldy, int lane) {
    constexpr int NK = K / 16;
    const int xo = (lane & 31) * ldx + (lane >> 5) * 8, yo = (lane & 31) * ldy + (lane >> 5) * 8;
    bf16x8 x[NK], yh[NK], yl[NK];
#pragma unroll
    for (int i = 0; i < NK; i++) {
        x[i] = *reinterpret_cast<const bf16x8 *>(X + xo + 16 * i);
        yh[i] = *reinterpret_cast<const bf16x8 *>(Yh + yo + 16 * i);
        yl[i] = *reinterpret_cast<const bf16x8 *>(Yl + yo + 16 * i);
    }
    __builtin_amdgcn_sched_barrier(0);  // fragment loads stay above, MFMAs below
    f32x16 acc_b = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};  // CH == 2: second, independent MFMA chain
#pragma unroll
    for (int i = 0; i < NK; i++) {
        if (CH == 2 && ((i + 0) & 1)) acc_b = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[i], yh[i], acc_b, 0, 0, 0); else acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[i], yh[i], acc, 0, 0, 0);
        if (CH == 2 && ((i + 1) & 1)) acc_b = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[i], yl[i], acc_b, 0, 0, 0); else acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[i], yl[i], acc, 0, 0, 0);
    }
    if (CH == 2) acc += acc_b;
}

// ---- operands whose contraction index is the ROW index of the stored plane (k-major, P[k][n]) ---------------------------
// gfx950's LDS transpose read ds_read_b64_tr_b16 works on groups of 16 lanes: lane i passes the address of a row R_i of four
// 16-bit elements and receives { R_{4j + (i >> 2)}[i & 3] : j = 0..3 } (probed on hardware, tools/tr16_probe.py).  Pointing
// lane r of a group at P[k0 + (r >> 2)][n0 + 4 (r & 3) ..] therefore hands lane i the four values P[k0 .. k0+3][n0 + i]:
// two such reads give the 8 consecutive k of one MFMA B fragment for column n -- no transposed copy of the plane needed.
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8e_t __attribute__((ext_vector_type(8)));
__device__ __forceinline__ bf16x8 frag_tr(const uint16_t *P, int ld, int k0, int n_base, int lane) {
    const int i16 = lane & 15;
    const uint16_t *p = P + (k0 + 8 * (lane >> 5) + (i16 >> 2)) * ld + n_base + 16 * ((lane >> 4) & 1) + 4 * (i16 & 3);
    using lds_ptr = bf16x4_t __attribute__((address_space(3))) *;
    const bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_ptr)(p));
    const bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_ptr)(p + 4 * ld));
    const bf16x8e_t v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}
// acc += X[m][0..K) . Y where Y is k-major: Y[k][n_base + n];  X split, Y split
template <int K, int CH = kMmaChains>
__device__ __forceinline__ void mma_tile3_yK(f32x16 &acc, const uint16_t *Xh, const uint16_t *Xl, int ldx, const uint16_t *Yh,
                                             const uint16_t *Yl, int ldy, int n_base, int lane) {
    constexpr int NK = K / 16;
    const int xo = (lane & 31) * ldx + (lane >> 5) * 8;
    bf16x8 xh[NK], xl[NK], yh[NK], yl[NK];
#pragma unroll
    for (int i = 0; i < NK; i++) {
        xh[i] = *reinterpret_cast<const bf16x8 *>(Xh + xo + 16 * i);
        xl[i] = *reinterpret_cast<const bf16x8 *>(Xl + xo + 16 * i);
        yh[i] = frag_tr(Yh, ldy, 16 * i, n_base, lane);
        yl[i] = frag_tr(Yl, ldy, 16 * i, n_base, lane);
    }
    __builtin_amdgcn_sched_barrier(0);  // fragment loads stay above, MFMAs below
    f32x16 acc_b = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};  // CH == 2: second, independent MFMA chain
#pragma unroll
    for (int i = 0; i < NK; i++) {
        if (CH == 2 && ((i + 0) & 1)) acc_b = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh[i], yh[i], acc_b, 0, 0, 0); else acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh[i], yh[i], acc, 0, 0, 0);
        if (CH == 2 && ((i + 1) & 1)) acc_b = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh[i], yl[i], acc_b, 0, 0, 0); else acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh[i], yl[i], acc, 0, 0, 0);
        if (CH == 2 && ((i + 0) & 1)) acc_b = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xl[i], yh[i], acc_b, 0, 0, 0); else acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xl[i], yh[i], acc, 0, 0, 0);
    }
    if (CH == 2) acc += acc_b;
}
// X split, Y exact (single plane), k-major
template <int K, int CH = kMmaChains>
__device__ __forceinline__ void mma_xs_yeK(f32x16 &acc, const uint16_t *Xh, const uint16_t *Xl, int ldx, const uint16_t *Y,
                                           int ldy, int n_base, int lane) {
    constexpr int NK = K / 16;
    const int xo = (lane & 31) * ldx + (lane >> 5) * 8;
    bf16x8 xh[NK], xl[NK], y[NK];
#pragma unroll
    for (int i = 0; i < NK; i++) {
        xh[i] = *reinterpret_cast<const bf16x8 *>(Xh + xo + 16 * i);
        xl[i] = *reinterpret_cast<const bf16x8 *>(Xl + xo + 16 * i);
        y[i] = frag_tr(Y, ldy, 16 * i, n_base, lane);
    }
    __builtin_amdgcn_sched_barrier(0);  // fragment loads stay above, MFMAs below
    f32x16 acc_b = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};  // CH == 2: second, independent MFMA chain
#pragma unroll
    for (int i = 0; i < NK; i++) {
        if (CH == 2 && ((i + 0) & 1)) acc_b = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh[i], y[i], acc_b, 0, 0, 0); else acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh[i], y[i], acc, 0, 0, 0);
        if (CH == 2 && ((i + 1) & 1)) acc_b = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xl[i], y[i], acc_b, 0, 0, 0); else acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xl[i], y[i], acc, 0, 0, 0);
    }
    if (CH == 2) acc += acc_b;
}

// General tile product: acc[m][n] += sum_k X[m][k] Y[n][k] over K, where each operand is either row-major (free index = row,
// k contiguous; `base` = first row) or k-major (k = row, free index contiguous; `base` = first column; fetched with
// frag_tr), and either exact bf16 (one plane) or split hi/lo.  Terms: Xh Yh (+ Xl Yh) (+ Xh Yl).
template <int K, bool XKM, bool XSPLIT, bool YKM, bool YSPLIT, int CH = kMmaChains>
__device__ __forceinline__ void mma_gen(f32x16 &acc, const uint16_t *Xh, const uint16_t *Xl, int ldx, int xbase,
                                        const uint16_t *Yh, const uint16_t *Yl, int ldy, int ybase, int lane) {
    constexpr int NK = K / 16;
    bf16x8 xh[NK], xl[NK], yh[NK], yl[NK];
    const int xo = (xbase + (lane & 31)) * ldx + (lane >> 5) * 8, yo = (ybase + (lane & 31)) * ldy + (lane >> 5) * 8;
#pragma unroll
    for (int i = 0; i < NK; i++) {
        xh[i] = XKM ? frag_tr(Xh, ldx, 16 * i, xbase, lane) : *reinterpret_cast<const bf16x8 *>(Xh + xo + 16 * i);
        yh[i] = YKM ? frag_tr(Yh, ldy, 16 * i, ybase, lane) : *reinterpret_cast<const bf16x8 *>(Yh + yo + 16 * i);
        if (XSPLIT) xl[i] = XKM ? frag_tr(Xl, ldx, 16 * i, xbase, lane) : *reinterpret_cast<const bf16x8 *>(Xl + xo + 16 * i);
        if (YSPLIT) yl[i] = YKM ? frag_tr(Yl, ldy, 16 * i, ybase, lane) : *reinterpret_cast<const bf16x8 *>(Yl + yo + 16 * i);
    }
    __builtin_amdgcn_sched_barrier(0);  // fragment loads stay above, MFMAs below
    f32x16 acc_b = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};  // CH == 2: second, independent MFMA chain
#pragma unroll
    for (int i = 0; i < NK; i++) {
        if (CH == 2 && ((i + 0) & 1)) acc_b = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh[i], yh[i], acc_b, 0, 0, 0); else acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh[i], yh[i], acc, 0, 0, 0);
        if (YSPLIT) { if (CH == 2 && ((i + 1) & 1)) acc_b = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh[i], yl[i], acc_b, 0, 0, 0); else acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh[i], yl[i], acc, 0, 0, 0); }
        if (XSPLIT) { if (CH == 2 && ((i + 0) & 1)) acc_b = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xl[i], yh[i], acc_b, 0, 0, 0); else acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xl[i], yh[i], acc, 0, 0, 0); }
    }
    if (CH == 2) acc += acc_b;
}

// row index of accumulator register r for this lane
__device__ __forceinline__ int d_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// write the D tile transposed into hi/lo planes: OUT[n0 + n][m0 + m] = D[m][n]
__device__ __forceinline__ void store_T_split(const f32x16 &acc, uint16_t *Oh, uint16_t *Ol, int ld, int lane) {
    const int n = lane & 31, h = lane >> 5;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        uint32_t h0, l0, h1, l1;
        split_pk(acc[4 * j + 0], acc[4 * j + 1], h0, l0);
        split_pk(acc[4 * j + 2], acc[4 * j + 3], h1, l1);
        const int off = n * ld + 8 * j + 4 * h;
        *reinterpret_cast<uint2 *>(Oh + off) = make_uint2(h0, h1);
        *reinterpret_cast<uint2 *>(Ol + off) = make_uint2(l0, l1);
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

// Workgroup barrier for LDS hand-offs only: waits for this wave's LDS traffic (lgkmcnt) and joins the barrier,
// but does NOT drain the vector-memory queue.  __syncthreads() carries a workgroup fence that also waits for
// outstanding global stores (gfx950: one vmcnt for loads and stores), which put ~1-2 us of HBM write latency on
// every barrier that followed the sa/y/hs stores of a chunk.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ---- scans along the time axis.  In the chunk kernels thread (pt = lane & 31, ...) holds step pt, so the 32 steps of a
// chunk are the 32 consecutive lanes of a half-wave: prefix sums are DPP row shifts (no LDS, no barrier).
template <int CTRL, int ROW_MASK, bool BOUND>
__device__ __forceinline__ float dpp0(float x) {  // lanes without a source (or outside ROW_MASK) get 0
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, ROW_MASK, 0xf, BOUND));
}
// inclusive prefix sum over lanes [0,32) and [32,64) separately
__device__ __forceinline__ float scan32(float x) {
    x += dpp0<0x111, 0xf, true>(x);   // row_shr:1
    x += dpp0<0x112, 0xf, true>(x);   // row_shr:2
    x += dpp0<0x114, 0xf, true>(x);   // row_shr:4
    x += dpp0<0x118, 0xf, true>(x);   // row_shr:8
    x += dpp0<0x142, 0xa, false>(x);  // row_bcast:15 into rows 1 and 3: lane 15 / 47 carries the first 16 steps
    return x;
}
// value of the previous lane (previous time step) inside the half-wave; `first` for step 0.  One DPP move (wave_shr:1) + one select:
// gamma_{t-1} from gamma_t without a second exponential (v_exp_f32 is a quarter-rate instruction)
__device__ __forceinline__ float prev32(float x, float first, int lane) {
    const float y = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x138, 0xf, 0xf, true));   // wave_shr:1
    return (lane & 31) == 0 ? first : y;
}
// value of lane 31 (63) for every lane of the half-wave
__device__ __forceinline__ float last32(float x, int lane) {
    return __int_as_float(__builtin_amdgcn_ds_bpermute(((lane & 32) | 31) << 2, __float_as_int(x)));
}
// value of the next lane (next time step); 0 for the last step of the chunk
__device__ __forceinline__ float next32(float x, int lane) {
    const float y = __int_as_float(__builtin_amdgcn_ds_bpermute(((lane + 1) & 63) << 2, __float_as_int(x)));
    return (lane & 31) == 31 ? 0.f : y;
}

// ---- state checkpoints between kernels ("q15" records) ------------------------------------------------------------------------
// The two 64x64 states that cross kernel boundaries once per chunk -- H at the start of a chunk (forward -> backward) and the
// adjoint E (state recurrence -> per-chunk gradient kernel) -- are stored as int16 mantissas with one fp32 scale per MFMA lane:
// the producer holds a 32(k) x 32(v) accumulator tile per wave (lane = value column + 32 * bit 2 of k, 16 registers = 16 key
// rows), takes the maximum of its own 16 values (no cross-lane work), and writes 16 int16 (two 16-byte stores) + one scale:
//     record[b,h,c] = mant[vh][kt][lane][16] int16 (8 KB)  +  scale[vh][kt][lane] fp32 (1 KB)         x ~ mant * scale
// vh = half of the value columns (the producers split a head over two workgroups), kt = key tile, register r of lane l holds
// key k = 32 kt + (r & 3) + 8 (r >> 2) + 4 (l >> 5), value v = 32 vh + (l & 31).  9 KB per record instead of 16 KB of fp32 (and 3
// stores per lane instead of 16 scattered 4-byte stores on the recurrence's critical waves).  The recurrences themselves never
// see the rounded copy (they carry fp32 accumulators / hi + lo planes); the per-chunk kernel reads each record once.
// Why not bf16: a float64 simulation of the per-chunk gradients with rounded checkpoints (tools/diag_checkpoint_precision.py)
// gives up to 5.8 bf16 ulp of extra error on the gradients (20 on dw, where rowsum(E * H_C) cancels against the in-chunk sums)
// for bf16 records, 0.6 (2.6) ulp for fp16, 0.05 (0.17) ulp for this format.
constexpr int kQMant = kN * kN;            // uint16 units
constexpr int kQRec = kQMant + 2 * 256;    // + 256 floats
typedef short short2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void q15_encode_tile(const f32x16 &acc, uint16_t *rec, int vh, int kt, int lane) {
    float m = 0.f;
#pragma unroll
    for (int r = 0; r < 16; r++) m = fmaxf(m, fabsf(acc[r]));
    const float inv = m > 0.f ? 1.f / m : 0.f;
    uint32_t w[8];
#pragma unroll
    for (int j = 0; j < 8; j++)   // v_cvt_pknorm_i16_f32: round(x * 32767) for x in [-1, 1], two per instruction
        w[j] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pknorm_i16(acc[2 * j] * inv, acc[2 * j + 1] * inv));
    const int slot = (vh * 2 + kt) * 64 + lane;
    uint4 *q = reinterpret_cast<uint4 *>(rec + slot * 16);
    q[0] = make_uint4(w[0], w[1], w[2], w[3]);
    q[1] = make_uint4(w[4], w[5], w[6], w[7]);
    reinterpret_cast<float *>(rec + kQMant)[slot] = m * (1.f / 32767.f);
}
// Reader side, thread = (value row v, keys k8 .. k8 + 7, k8 % 8 == 0): the 8 values are registers r0 .. r0 + 3 of lane (v & 31)
// (keys k8 .. k8 + 3) and of lane (v & 31) + 32 (keys k8 + 4 .. k8 + 7), r0 = 4 ((k8 >> 3) & 3), tile kt = k8 >> 5.
__device__ __forceinline__ int q15_slot(int v, int k8) { return ((v >> 5) * 2 + (k8 >> 5)) * 64 + (v & 31); }
__device__ __forceinline__ void q15_load8(const uint16_t *rec, int v, int k8, uint2 &lo4, uint2 &hi4) {
    const uint16_t *p = rec + q15_slot(v, k8) * 16 + 4 * ((k8 >> 3) & 3);
    lo4 = *reinterpret_cast<const uint2 *>(p);
    hi4 = *reinterpret_cast<const uint2 *>(p + 32 * 16);
}
__device__ __forceinline__ void q15_decode8(const uint2 lo4, const uint2 hi4, float s_lo, float s_hi, float (&x)[8]) {
    x[0] = (float)(int)(int16_t)(lo4.x & 0xffffu) * s_lo; x[1] = (float)((int)lo4.x >> 16) * s_lo;
    x[2] = (float)(int)(int16_t)(lo4.y & 0xffffu) * s_lo; x[3] = (float)((int)lo4.y >> 16) * s_lo;
    x[4] = (float)(int)(int16_t)(hi4.x & 0xffffu) * s_hi; x[5] = (float)((int)hi4.x >> 16) * s_hi;
    x[6] = (float)(int)(int16_t)(hi4.y & 0xffffu) * s_hi; x[7] = (float)((int)hi4.y >> 16) * s_hi;
}

__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int r = 0; r < 16; r++) z[r] = 0.f;
    return z;
}

}  // namespace rwkv7
