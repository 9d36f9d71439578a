// rwkvtts_amd/csrc/lora_down.hip -- the low-rank branches' DOWN projections with the token-shift lerp as the GEMM's A prologue, bf16, gfx950.
//
// Reference arithmetic (model/llm/rwkv_s2s_single_ffn.py:160-190): x_i = xm + (shift(xm) - xm) mu_i (bf16), h_i = act_i(x_i W1_i^T) for the
// w / a / v / g branches (Linear(D, r_i), r_i = 64 / 64 / 32 / 128 at 0.4B).  Round 4 took these projections THROUGH the lerp
// (csrc/mix_lora.hip: one library GEMM [M, D] x [D, 2 R] + a combine kernel on [M, 2 R]); the library runs that N = 576 shape at 0.63 PF/s
// (61 us per layer; it takes the same time for N = 1024) and the combine kernel costs 20 us more: 81 us for 19 GFLOP and 67 MB.  Here
// (round 6) the same outputs come from ONE kernel that streams x once:
//   * a workgroup owns 128 rows and ALL R output columns; eight waves = 4 row groups (one 32-row MFMA tile each) x 2 column halves
//     (contiguous 32-column tiles, cut on the host so that a half holds at most two branches; waves w and w + 4 share a SIMD);
//   * x goes through LDS in stages of 64 columns ([129][64 + 8] bf16, the extra row is the previous row of the tile's first one), three
//     buffers (W1 stages likewise, both requested TWO stages ahead, two loads per k step between the products); a wave reads its rows' fragments AND the fragments one row up, forms bf16(xc + (xp - xc) mu_i) in the MFMA operand layout
//     (the reference's rounding point; the through-the-lerp form skipped it; v_pk_add_f32 / v_pk_fma_f32) and multiplies with W1
//     fragments from LDS: W1 is re-packed once per call into fragment order ([tile][k step][lane] 16 bytes, rwkv7_lora_down_pack_bf16),
//     a stage's fragments are staged lane-linear (two buffers), so a fragment read is one conflict-free 1 KB ds_read_b128;
//   * epilogue: bf16 rounding, the branch's activation on the rounded value (what Linear + activation produce), through a wave-private
//     LDS tile so that the stores into the branch's own [M][r_i] tensor are 64-byte row pieces.
// Measured (tools/lora_down_ab.py under rocprofv3, B = 8, T = 4096, D = 1024, same process): 43.4 us + 4.7 us for the pack against
// 55.6 + 18.4 us for the library GEMM + combine kernel (+ 6.0 us wcat_fwd, which now runs in the backward); in the training step
// -0.44 ms (tools/ab_step.py, 118.49 -> 118.05 ms, same box).  What bounds it (switches in the lab cuts, since removed): with the
// loads off 24 us of fragment reads (72 KB per k step and workgroup) + lerp VALU (~50 instructions per wave and k step) + MFMA that
// overlap only across the two waves of a SIMD; ~10 us of staging skeleton (16 stages: 45 KB of register -> LDS stores and a barrier
// each); the loads add ~8 us that hide only in part.  HBM: 67 MB in, 19 MB out.  The backward is unchanged (fused.py:
// rwkv7_mix_lora_combine_bwd / wcat_* on the through-the-lerp form, which is the exact gradient of this forward).
#include "chunk_common.h"
#include "launch_attr.h"
#include <type_traits>

namespace rwkv7 {

struct LoraDownDesc {
    int nb;               // branches
    int r[4], off[4];     // rank and first output column of each branch inside R
    int act[4];           // 0 none, 1 tanh, 2 sigmoid
    const void *w1[4];    // pack: Linear(D, r_i).weight [r_i][D]
    const void *mu[4];    // lerp coefficient [D]
    void *out[4];         // [M][r_i]
    int tile0[5];         // column group g owns the 32-column tiles tile0[g] .. tile0[g + 1] - 1
};

namespace {
constexpr int kRows = 128;          // rows per workgroup
constexpr int kKt = 64;             // columns per stage
constexpr int kLd = kKt + 8;        // LDS row stride in elements: 36 dwords: 16 consecutive rows x 16 bytes fall into distinct bank quads twice over
constexpr int kTileRows = kRows + 1;
constexpr int kXBuf = kTileRows * kLd;   // elements per x buffer
constexpr int kMaxTiles = 10;       // R <= 320 (two W1 stage buffers of R x 128 bytes beside three x buffers)
constexpr int kWPer = (kMaxTiles * 256 + 511) / 512;   // W1 chunks per thread and stage

__device__ __forceinline__ int ld_branch_of(const LoraDownDesc &d, int col) {
    int i = 0;
#pragma unroll
    for (int j = 1; j < 4; j++)
        if (j < d.nb && col >= d.off[j]) i = j;
    return i;
}
// tanh without the library call (its range reduction is ~60 instructions per value, 7 us of this kernel): odd polynomial below 0.25
// (x^9 term < 1e-7 relative), 1 - 2 / (e^2x + 1) above; both well inside half a bf16 ulp of the rounded result
__device__ __forceinline__ float ld_tanh(float x) {
    const float x2 = x * x;
    const float small = x * (1.f + x2 * (-0.33333334f + x2 * (0.13333334f + x2 * -0.053968254f)));
    const float big = 1.f - 2.f / (__expf(2.f * x) + 1.f);
    return fabsf(x) < 0.25f ? small : big;
}
__device__ __forceinline__ float ld_act(int a, float x) {
    if (a == 1) return ld_tanh(x);
    if (a == 2) return 1.f / (1.f + __expf(-x));
    return x;
}

// packed[(tile * (D / 16) + ks) * 64 + lane] = W1cat[32 tile + (lane & 31)][16 ks + 8 (lane >> 5) .. + 8]: one wave per (tile, k step)
__global__ __launch_bounds__(256) void lora_pack_kernel(LoraDownDesc d, int R, int D, uint4 *__restrict__ packed) {
    const int nks = D / 16, lane = threadIdx.x & 63;
    const int item = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= (R / 32) * nks) return;
    const int tile = item / nks, ks = item - tile * nks;
    const int row = tile * 32 + (lane & 31), b = ld_branch_of(d, row);
    const uint16_t *w = reinterpret_cast<const uint16_t *>(d.w1[b]) + (long)(row - d.off[b]) * D + ks * 16 + 8 * (lane >> 5);
    packed[(long)item * 64 + lane] = *reinterpret_cast<const uint4 *>(w);
}

typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
// a bf16 pair -> two fp32 (packed: the lerp below runs on v_pk_add_f32 / v_pk_fma_f32)
__device__ __forceinline__ f2_t up2(uint32_t u) {
    const f2_t v = {__uint_as_float(u << 16), __uint_as_float(u & 0xffff0000u)};
    return v;
}
__device__ __forceinline__ uint32_t pk2(f2_t v) { return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf2_t)); }

// One wave's work: 32 rows x NT column tiles, the first N0 of branch b0 and the rest of branch b1 (N0 == NT: one branch).  Straight-line
// per (NT, N0).  Everything a k step needs comes from LDS: x in stages of 64 columns ([129][64 + 8], THREE buffers: the rows of stage
// s + 2 are requested at the start of stage s and stored at the end of stage s + 1 -- two stages of HBM latency), the W1 fragments of a
// stage (R / 32 tiles x 4 k steps x 1 KB, lane-linear as packed, TWO buffers: requested at the start of stage s for s + 1).  vmcnt
// retires in order, so inside a stage the W1 requests go out BEFORE the x requests: the stage-end wait for W1 leaves the younger x loads
// in flight.
// What the earlier cuts measured (tools/lora_down_exp.py, switches removed since): W1 fragments straight into registers through a ring two
// k steps deep, x one stage ahead: 56 us, every stage behind a full memory round trip; all through LDS with eight waves = 2 row halves
// x 4 column groups: 61 us, of which 47 with the loads switched off -- each of the four column groups unpacked and subtracted the same
// rows (35 VALU instructions per bf16 pair of x where the lerps themselves need 10): the split below (4 row groups x 2 column halves)
// halves that.
// GENERIC: per-row multipliers (mask, sequence starts anywhere).  Otherwise: no mask and T % 128 == 0 -- a sequence can only start on a
// tile's first row, whose previous row is the extra LDS row: it is stored as zeros and nothing is multiplied.
template <bool GENERIC, int NT, int N0>
__device__ __forceinline__ void lora_down_body(const LoraDownDesc &d, long M, int T, int D, int NTALL, const uint16_t *__restrict__ x,
                                               const uint16_t *__restrict__ mask, const uint4 *__restrict__ packed, uint16_t *sm, int t0,
                                               int rowgrp) {
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, rl = lane & 31;
    uint16_t *smw = sm + 3 * kXBuf;                                  // [2][NTALL][4][64] uint4
    uint16_t *smu = smw + 2 * NTALL * 4 * 64 * 8;                    // [nb][D]
    const long row0 = (long)blockIdx.x * kRows;
    const int nks = D / 16, nstage = D / kKt;
    const int b0 = ld_branch_of(d, 32 * t0), b1 = ld_branch_of(d, 32 * (t0 + (N0 < NT ? N0 : 0)));

    for (int i = tid; i < d.nb * (D / 8); i += 512) {
        const int b = i / (D / 8), c = i - b * (D / 8);
        reinterpret_cast<uint4 *>(smu)[i] = reinterpret_cast<const uint4 *>(d.mu[b])[c];
    }
    // per-lane row multipliers (GENERIC)
    float mc = 1.f, mp = 1.f;
    if (GENERIC) {
        const long g = row0 + rowgrp * 32 + rl;
        const long gp = g > 0 ? g - 1 : 0;
        const bool first = (g % T) == 0;
        mc = mask ? bf2f(mask[g]) : 1.f;
        mp = first ? 0.f : (mask ? bf2f(mask[gp]) : 1.f);
    }
    // x stage: chunk q = tid + 512 j (j = 0, 1): tile row 1 + (q >> 3), 16-byte chunk q & 7; the extra row rides on lanes 0..7
    const bool zero_halo = !GENERIC && (row0 % T) == 0;
    const long halo_row = row0 > 0 ? row0 - 1 : 0;
    // the registers in flight are plain named uint4 variables (a struct that is written field by field inside a lambda stays in scratch
    // memory: 96 bytes per lane in the first cut)
    auto xsrc_of = [&](int s) { return x + (row0 + (tid >> 3)) * D + (s < nstage ? s : nstage - 1) * kKt + (tid & 7) * 8; };   // clamped
    auto hsrc_of = [&](int s) { return x + halo_row * D + (s < nstage ? s : nstage - 1) * kKt + (tid & 7) * 8; };
    auto ld16 = [&](const uint16_t *p) { return *reinterpret_cast<const uint4 *>(p); };
    auto halo = [&](uint4 v) { return zero_halo ? make_uint4(0u, 0u, 0u, 0u) : v; };
    auto stash_x = [&](uint4 c0, uint4 c1, uint4 hl, int buf) {
        uint16_t *t = sm + buf * kXBuf + (1 + (tid >> 3)) * kLd + (tid & 7) * 8;
        *reinterpret_cast<uint4 *>(t) = c0;
        *reinterpret_cast<uint4 *>(t + 64 * kLd) = c1;
        if (tid < 8) *reinterpret_cast<uint4 *>(sm + buf * kXBuf + tid * 8) = hl;
    };
    // W1 stage: NTALL tiles x 256 chunks of 16 bytes (4 k steps x 64 lanes); chunk q = tid + 512 j, j < kWPer (q clamped, store guarded)
    static_assert(kWPer == 5, "five W1 chunks per thread and stage");
    const int wchunks = NTALL * 256;
    auto wsrc = [&](int s, int j) {
        int q = tid + 512 * j;
        q = q < wchunks ? q : wchunks - 1;
        return packed + ((long)(q >> 8) * nks + 4 * (s < nstage ? s : nstage - 1)) * 64 + (q & 255);
    };
    auto stash_w = [&](uint4 c0, uint4 c1, uint4 c2, uint4 c3, uint4 c4, int buf) {
        uint4 *w = reinterpret_cast<uint4 *>(smw) + buf * NTALL * 256 + tid;
        if (tid < wchunks) w[0] = c0;
        if (tid + 512 < wchunks) w[512] = c1;
        if (tid + 1024 < wchunks) w[1024] = c2;
        if (tid + 1536 < wchunks) w[1536] = c3;
        if (tid + 2048 < wchunks) w[2048] = c4;
    };

    f32x16 acc[NT];
#pragma unroll
    for (int i = 0; i < NT; i++) acc[i] = zero16();

    // one stage; the NEXT stages' loads are issued two per k step between the products (all eight at the top of the stage kept every wave
    // at the texture addresser for ~1k cycles with nothing else to run: +10 us): W1 of stage s + 1 first, then x of stage s + 2
    uint4 xa0, xa1, xah, xb0, xb1, xbh;                       // x of the two stages in flight
    uint4 wa0, wa1, wa2, wa3, wa4, wb0, wb1, wb2, wb3, wb4;   // W1 likewise (first cut: W1 one stage ahead only, requested 1-3 k steps before its stage-end wait)
    auto compute = [&](int s, auto which) {
        constexpr bool TO_B = decltype(which)::value == 1;
        const uint16_t *tile = sm + (s % 3) * kXBuf;
        const uint4 *wt = reinterpret_cast<const uint4 *>(smw) + ((s & 1) * NTALL + t0) * 256 + lane;
        const int sw = s + 2, sx = s + 2;   // (clamped inside wsrc / xsrc_of: unconditional loads)
        const uint16_t *xsrc = xsrc_of(sx);
#pragma unroll
        for (int ks = 0; ks < kKt / 16; ks++) {
            const int kg = s * (kKt / 16) + ks;
            if (ks == 0) {
                const uint4 u0 = *wsrc(sw, 0), u1 = *wsrc(sw, 1);
                if (TO_B) { wb0 = u0; wb1 = u1; } else { wa0 = u0; wa1 = u1; }
            }
            if (ks == 1) {
                const uint4 u2 = *wsrc(sw, 2), u3 = *wsrc(sw, 3);
                if (TO_B) { wb2 = u2; wb3 = u3; } else { wa2 = u2; wa3 = u3; }
            }
            if (ks == 2) {
                const uint4 u4 = *wsrc(sw, 4);
                if (TO_B) wb4 = u4; else wa4 = u4;
                const uint4 v = ld16(xsrc);
                if (TO_B) xb0 = v; else xa0 = v;
            }
            if (ks == 3) {
                const uint4 v1 = ld16(xsrc + 64L * D), vh = halo(ld16(hsrc_of(sx)));
                if (TO_B) { xb1 = v1; xbh = vh; } else { xa1 = v1; xah = vh; }
            }
            __builtin_amdgcn_sched_barrier(0);
            f2_t xc[4], dx[4];
            const uint16_t *rp = tile + (rowgrp * 32 + rl) * kLd + ks * 16 + 8 * h;   // the row above; + kLd: the row itself
            const u32x4_t vc = *reinterpret_cast<const u32x4_t *>(rp + kLd), vp = *reinterpret_cast<const u32x4_t *>(rp);
#pragma unroll
            for (int e = 0; e < 4; e++) {
                f2_t c = up2(vc[e]), p = up2(vp[e]);
                if (GENERIC) {
                    c *= mc;
                    p *= mp;
                }
                xc[e] = c;
                dx[e] = p - c;
            }
            bf16x8 af;
            auto lerp = [&](int b) {
                const u32x4_t mv = *reinterpret_cast<const u32x4_t *>(smu + b * D + kg * 16 + 8 * h);
                u32x4_t o;
#pragma unroll
                for (int e = 0; e < 4; e++) o[e] = pk2(__builtin_elementwise_fma(dx[e], up2(mv[e]), xc[e]));
                af = __builtin_bit_cast(bf16x8, o);
            };
            lerp(b0);
#pragma unroll
            for (int i = 0; i < NT; i++) {
                if (i == N0 && N0 < NT) lerp(b1);
                const bf16x8 wf = __builtin_bit_cast(bf16x8, wt[(i * 4 + ks) * 64]);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, af, acc[i], 0, 0, 0);
            }
        }
    };

    // stages in pairs (the two x register sets alternate); nstage = D / 64 is even
    {
        const uint16_t *p0 = xsrc_of(0);
        stash_x(ld16(p0), ld16(p0 + 64L * D), halo(ld16(hsrc_of(0))), 0);
        stash_w(*wsrc(0, 0), *wsrc(0, 1), *wsrc(0, 2), *wsrc(0, 3), *wsrc(0, 4), 0);
        const uint16_t *p1 = xsrc_of(1);
        xa0 = ld16(p1); xa1 = ld16(p1 + 64L * D); xah = halo(ld16(hsrc_of(1)));   // xa / wa: stage s + 1 at the top of an even stage s
        wa0 = *wsrc(1, 0); wa1 = *wsrc(1, 1); wa2 = *wsrc(1, 2); wa3 = *wsrc(1, 3); wa4 = *wsrc(1, 4);
    }
    __syncthreads();
    for (int s = 0; s < nstage; s += 2) {
        compute(s, std::integral_constant<int, 1>{});       // requests: W1 and x of stage s + 2 -> wb, xb
        __builtin_amdgcn_sched_barrier(0);
        stash_w(wa0, wa1, wa2, wa3, wa4, (s + 1) & 1);
        stash_x(xa0, xa1, xah, (s + 1) % 3);
        __syncthreads();
        compute(s + 1, std::integral_constant<int, 0>{});   // W1 and x of stage s + 3 -> wa, xa
        __builtin_amdgcn_sched_barrier(0);
        stash_w(wb0, wb1, wb2, wb3, wb4, s & 1);
        stash_x(xb0, xb1, xbh, (s + 2) % 3);
        __syncthreads();
    }

    // epilogue: lane = row rl of the tile, register 4 g + e = output column 8 g + 4 h + e of the 32-column tile.  Through a wave-private
    // LDS tile [32][32 + 8] (the x buffers are free behind the last barrier) so that a store instruction writes 16 rows x 64 bytes
    // instead of 32 rows x 16: a quarter of the lines per tile (the direct form was 7 us of the kernel).
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    uint16_t *et = sm + wave * (32 * 40);
#pragma unroll
    for (int i = 0; i < NT; i++) {
        const int b = i < N0 ? b0 : b1, rb = d.r[b], a = d.act[b];
        const int col0 = 32 * (t0 + i) - d.off[b];
#pragma unroll
        for (int g = 0; g < 4; g++) {
            // the projection's bf16 output, then the activation on it (what Linear + activation produce)
            const uint32_t p0 = cvt_pk(acc[i][4 * g], acc[i][4 * g + 1]), p1 = cvt_pk(acc[i][4 * g + 2], acc[i][4 * g + 3]);
            const float v0 = ld_act(a, __uint_as_float(p0 << 16)), v1 = ld_act(a, __uint_as_float(p0 & 0xffff0000u));
            const float v2 = ld_act(a, __uint_as_float(p1 << 16)), v3 = ld_act(a, __uint_as_float(p1 & 0xffff0000u));
            *reinterpret_cast<uint2 *>(et + rl * 40 + 8 * g + 4 * h) = make_uint2(cvt_pk(v0, v1), cvt_pk(v2, v3));
        }
        uint16_t *o = reinterpret_cast<uint16_t *>(d.out[b]) + (row0 + rowgrp * 32 + (lane >> 2)) * rb + col0 + (lane & 3) * 8;
        const uint4 lo = *reinterpret_cast<const uint4 *>(et + (lane >> 2) * 40 + (lane & 3) * 8);
        const uint4 hi = *reinterpret_cast<const uint4 *>(et + (16 + (lane >> 2)) * 40 + (lane & 3) * 8);
        *reinterpret_cast<uint4 *>(o) = lo;
        *reinterpret_cast<uint4 *>(o + 16L * rb) = hi;
    }
}

template <bool GENERIC>
__global__ __launch_bounds__(512) void lora_down_fwd_kernel(LoraDownDesc d, long M, int T, int D, const uint16_t *__restrict__ x,
                                                            const uint16_t *__restrict__ mask, const uint4 *__restrict__ packed) {
    extern __shared__ __attribute__((aligned(16))) uint16_t sm[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int rowgrp = wave & 3, nh = wave >> 2;   // waves w and w + 4 (the two column halves of one row group) share a SIMD
    const int t0 = d.tile0[nh], nt = d.tile0[nh + 1] - t0, ntall = d.tile0[2];
    // tiles of the half's first branch
    int n0 = 0;
    {
        const int b0 = ld_branch_of(d, 32 * t0);
        for (int i = 0; i < nt; i++) n0 += ld_branch_of(d, 32 * (t0 + i)) == b0;
    }
#define LD_CASE(NT_, N0_)                                                                                \
    if (nt == NT_ && n0 == N0_) {                                                                        \
        lora_down_body<GENERIC, NT_, N0_>(d, M, T, D, ntall, x, mask, packed, sm, t0, rowgrp);           \
        return;                                                                                          \
    }
    LD_CASE(4, 2) LD_CASE(5, 1) LD_CASE(4, 4) LD_CASE(5, 5) LD_CASE(5, 2) LD_CASE(5, 3) LD_CASE(5, 4) LD_CASE(4, 1) LD_CASE(4, 3)
    LD_CASE(3, 3) LD_CASE(3, 1) LD_CASE(3, 2) LD_CASE(2, 2) LD_CASE(2, 1) LD_CASE(1, 1)
#undef LD_CASE
}

template <bool GENERIC>
int launch_lora_down(const LoraDownDesc &d, long M, int T, int D, const void *x, const void *mask, const void *packed, hipStream_t st) {
    const size_t lds = (size_t)(3 * kXBuf + 2 * d.tile0[2] * 4 * 64 * 8 + d.nb * D) * 2;
    if (lds > 160 * 1024) return -4;
    static DynLdsOnce lds_once;
    auto kern = &lora_down_fwd_kernel<GENERIC>;
    if (hipError_t e = lds_once.ensure(reinterpret_cast<const void *>(kern), 160 * 1024); e != hipSuccess) return (int)e;
    (void)hipGetLastError();
    kern<<<dim3((unsigned)(M / kRows)), dim3(512), lds, st>>>(d, M, T, D, (const uint16_t *)x, (const uint16_t *)mask, (const uint4 *)packed);
    return (int)hipGetLastError();
}
}  // namespace

// cut the R / 32 column tiles into two contiguous halves (1..5 tiles and at most two branches each): minimise the heavier half, a
// half's weight = its tiles (MFMA, W1 fragment reads) plus 1.5 per branch in it (the lerp)
int lora_down_cut(LoraDownDesc &d, int R) {
    const int NT = R / 32;
    if (NT > kMaxTiles || NT < 2) return -1;
    auto branch = [&](int tile) {
        int b = 0;
        for (int j = 1; j < d.nb; j++)
            if (32 * tile >= d.off[j]) b = j;
        return b;
    };
    auto weight = [&](int a, int b) {   // tiles a .. b - 1
        if (b <= a || b - a > 5) return 1e9f;
        int nbr = 1;
        for (int t = a + 1; t < b; t++) nbr += branch(t) != branch(t - 1);
        if (nbr > 2) return 1e9f;
        return (float)(b - a) + 1.5f * nbr;
    };
    float best = 1e9f;
    int bc = -1;
    for (int c = 1; c < NT; c++) {
        const float w = fmaxf(weight(0, c), weight(c, NT));
        if (w < best) {
            best = w;
            bc = c;
        }
    }
    if (bc < 0 || best > 1e8f) return -1;
    d.tile0[0] = 0; d.tile0[1] = bc; d.tile0[2] = NT; d.tile0[3] = d.tile0[4] = NT;
    return NT;
}

int lora_down_pack(const LoraDownDesc &d, int R, int D, void *packed, hipStream_t st) {
    (void)hipGetLastError();
    const int items = (R / 32) * (D / 16);
    hipLaunchKernelGGL(lora_pack_kernel, dim3((items + 3) / 4), dim3(256), 0, st, d, R, D, (uint4 *)packed);
    return (int)hipGetLastError();
}

// M % 128 == 0, M % T == 0, D % 128 == 0, ranks multiples of 32, 2..10 column tiles that cut into two halves of at most two branches
int lora_down_fwd(LoraDownDesc &d, int R, long M, int T, int D, const void *x, const void *mask, const void *packed, hipStream_t st) {
    if (lora_down_cut(d, R) < 0) return -4;
    const bool generic = mask != nullptr || T % kRows != 0;
    return generic ? launch_lora_down<true>(d, M, T, D, x, mask, packed, st) : launch_lora_down<false>(d, M, T, D, x, mask, packed, st);
}

}  // namespace rwkv7
