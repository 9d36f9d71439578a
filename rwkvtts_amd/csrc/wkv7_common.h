// rwkvtts_amd/csrc/wkv7_common.h -- shared device helpers for the gfx950 WKV7 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rwkv7 {

constexpr int kN = 64;      // head size (reference: -D_C_=64 / -D_N_=64)
constexpr int kChunk = 16;  // state checkpoint interval (reference: _CHUNK_LEN_)
constexpr int kTB = 16;     // time steps staged through LDS per barrier pair

struct bf16_t {
    uint16_t x;
};

__device__ __forceinline__ float bf2f(uint16_t u) { return __uint_as_float(((uint32_t)u) << 16); }

// round-to-nearest-even, NaN kept quiet (== __float2bfloat16_rn, wkv7_cuda.cu:6)
__device__ __forceinline__ uint16_t f2bf(float f) {
    uint32_t x = __float_as_uint(f);
    if ((x & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((x >> 16) | 0x0040u);
    x += 0x7fffu + ((x >> 16) & 1u);
    return (uint16_t)(x >> 16);
}

// ---- 4-element vector I/O, templated on the tensor element type -------------------------------
template <typename T>
struct Raw4;
template <>
struct Raw4<bf16_t> {
    uint2 r;
};
template <>
struct Raw4<float> {
    float4 r;
};

template <typename T>
__device__ __forceinline__ Raw4<T> ld4(const T *p, bool ok);
template <>
__device__ __forceinline__ Raw4<bf16_t> ld4<bf16_t>(const bf16_t *p, bool ok) {
    Raw4<bf16_t> o;
    o.r = ok ? *reinterpret_cast<const uint2 *>(p) : make_uint2(0u, 0u);
    return o;
}
template <>
__device__ __forceinline__ Raw4<float> ld4<float>(const float *p, bool ok) {
    Raw4<float> o;
    o.r = ok ? *reinterpret_cast<const float4 *>(p) : make_float4(0.f, 0.f, 0.f, 0.f);
    return o;
}

__device__ __forceinline__ float4 cvt4(const Raw4<bf16_t> &x) {
    return make_float4(__uint_as_float(x.r.x << 16), __uint_as_float(x.r.x & 0xffff0000u),
                       __uint_as_float(x.r.y << 16), __uint_as_float(x.r.y & 0xffff0000u));
}
__device__ __forceinline__ float4 cvt4(const Raw4<float> &x) { return x.r; }

__device__ __forceinline__ void st4(bf16_t *p, float4 v) {
    uint2 o;
    o.x = (uint32_t)f2bf(v.x) | ((uint32_t)f2bf(v.y) << 16);
    o.y = (uint32_t)f2bf(v.z) | ((uint32_t)f2bf(v.w) << 16);
    *reinterpret_cast<uint2 *>(p) = o;
}
__device__ __forceinline__ void st4(float *p, float4 v) { *reinterpret_cast<float4 *>(p) = v; }

// ---- 2-element vector I/O ------------------------------------------------------------------------
template <typename T>
struct Raw2;
template <>
struct Raw2<bf16_t> {
    uint32_t r;
};
template <>
struct Raw2<float> {
    float2 r;
};
template <typename T>
__device__ __forceinline__ Raw2<T> ld2(const T *p, bool ok);
template <>
__device__ __forceinline__ Raw2<bf16_t> ld2<bf16_t>(const bf16_t *p, bool ok) {
    Raw2<bf16_t> o;
    o.r = ok ? *reinterpret_cast<const uint32_t *>(p) : 0u;
    return o;
}
template <>
__device__ __forceinline__ Raw2<float> ld2<float>(const float *p, bool ok) {
    Raw2<float> o;
    o.r = ok ? *reinterpret_cast<const float2 *>(p) : make_float2(0.f, 0.f);
    return o;
}
__device__ __forceinline__ float2 cvt2(const Raw2<bf16_t> &x) {
    return make_float2(__uint_as_float(x.r << 16), __uint_as_float(x.r & 0xffff0000u));
}
__device__ __forceinline__ float2 cvt2(const Raw2<float> &x) { return x.r; }
__device__ __forceinline__ void st2(bf16_t *p, float2 v) {
    *reinterpret_cast<uint32_t *>(p) = (uint32_t)f2bf(v.x) | ((uint32_t)f2bf(v.y) << 16);
}
__device__ __forceinline__ void st2(float *p, float2 v) { *reinterpret_cast<float2 *>(p) = v; }

// ---- cross-lane sums on the DPP path (no LDS traffic) --------------------------------------------
// dpp_ctrl encodings (LLVM AMDGPU DppCtrl): quad_perm = p0|p1<<2|p2<<4|p3<<6, row_mirror = 0x140,
// row_half_mirror = 0x141.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, true));
}
// sum over the 8 consecutive lanes {8g .. 8g+7}; every lane of the group gets the total
__device__ __forceinline__ float sum8(float x) {
    x += dpp_mov<0xB1>(x);   // quad_perm [1,0,3,2]
    x += dpp_mov<0x4E>(x);   // quad_perm [2,3,0,1]
    x += dpp_mov<0x141>(x);  // row_half_mirror
    return x;
}
// sum over the 16 lanes of a DPP row; every lane of the row gets the total
__device__ __forceinline__ float sum16(float x) {
    x = sum8(x);
    x += dpp_mov<0x140>(x);  // row_mirror
    return x;
}

// ---- transposing pair reductions across DPP rows (gfx950 v_permlane{16,32}_swap) ----------------------
// swap32_sum(lo, hi): lanes 0-31 return lo[l] + lo[l+32], lanes 32-63 return hi[l-32] + hi[l].
// v_permlane32_swap exchanges the upper half of its first operand with the lower half of its second.
__device__ __forceinline__ float swap32_sum(float lo, float hi) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(lo), __float_as_uint(hi), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// swap16_sum(lo, hi): even 16-lane rows return lo[l] + lo[l+16], odd rows return hi[l-16] + hi[l].
// v_permlane16_swap exchanges the odd rows of its first operand with the even rows of its second.
__device__ __forceinline__ float swap16_sum(float lo, float hi) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(lo), __float_as_uint(hi), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

__device__ __forceinline__ float fast_exp(float x) { return __expf(x); }

}  // namespace rwkv7
