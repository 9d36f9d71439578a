// rwkvtts_amd/csrc/decode_step.hip -- one greedy-decode step (T = 1, B <= 32 sequences) of the whole RWKV-7 stack on gfx950
// (BASELINE.json configs[4]: "persistent-state decode kernel").
//
// Reference: the per-token path RWKV_x070.forward_one (model/llm/rwkv_s2s_single_ffn.py:417-445) with
// RWKV_x070_TMix_one (:482-506) and RWKV_x070_CMix_one (:545-549); batched form forward_batch at T = 1
// (model/llm/rwkv_asr_cuda_whisper.py:438-472).  Module by module a step is ~18 launches per layer; even replayed from a
// hipGraph that is 430 launches and 4.0 ms for a step whose HBM traffic (0.65 GB of weights + 0.4 GB of state) would take
// 0.13 ms.
//
// Here the step is 7 grid-wide phases per layer:
//   P0 row    x += previous channel-mix output (K-split partials); h = LayerNorm1(x); six token-shift lerps -> bf16 rows;
//             att_x_prev <- h                                                      (rwkv_s2s_single_ffn.py:486-487)
//   P1 gemv   r, k, v projections and the four low-rank down projections in one sweep: [32 x K] . W^T on MFMA, the batch
//             rows are the 32-wide B operand, K split over waves and workgroups -> fp32 partials   (:489-491,497-500)
//   P2 head   per (head, 2 sequences): low-rank up projections (+ tanh / sigmoid), decay, value residual, kk
//             normalisation, the 64x64 fp32 state update in place, y, GroupNorm, bonus, gate -> bf16 rows   (:493-505)
//   P3 gemv   output projection -> partials                                                                  (:506)
//   P4 row    x += attention output; h = LayerNorm2(x); channel-mix lerp; ffn_x_prev <- h                    (:546-547)
//   P5 gemv   key projection, relu(.)^2 -> bf16 rows (no K split: the activation needs the whole sum)        (:548)
//   P6 gemv   value projection -> partials                                                                   (:548-549)
// and a final row phase (last residual add + model norm) and the head projection -> fp32 logits.  GEMV phases write fp32
// K-split partials [KS][32][N] that the consumer sums when it loads them, so no phase waits for a reduction.  A phase is a
// chain of load latencies, so every phase requests whatever does not depend on the previous phase (state rows, parameter
// vectors, up-projection rows) before it reads the activations, and sums partials with all loads of a round in flight.
//
// Two ways to run the phases (same bodies, bit-identical results, tests/test_decode_step_gpu.py):
//   persistent = 0  one launch per phase (7 L + 2), one kernel per phase, each sized to its item count.  Measured at configs[4]
//                   (0.4B, B = 32, tools/decode_phase_profile.py): row phases 3.1-3.6 us at best, GEMV phases 3.5-6.6 us, head
//                   phase 7.4-8.5 us (round 2: 13 us -- see head_phase), 1.05 ms per step in the replayed graph = 30.5 k
//                   tokens/s (module path: 4.0 ms, 8 k tokens/s).
//   persistent = 1  ONE launch of 256 resident workgroups that meet at a device-scope barrier between phases.  Measured:
//                   7.4 us per barrier -- 3.9 us for 256 arrivals + polling on one counter, 1.8 us for the agent-scope
//                   release (L2 write-back) and 1.6 us for the acquire (invalidate); the XCDs' L2s are not coherent with
//                   each other, so both are needed -- against ~1.5 us for a stream-ordered kernel boundary: 2.4 ms per step.
//                   Kept as an option (and as a cross-check of the phase bodies); the Python host uses persistent = 0.
#include "chunk_common.h"

namespace rwkv7 {

// order of the per-layer pointer table (include/rwkv7_hip.h: RWKV7_DEC_*)
enum DecPtr {
    DP_LN0_W, DP_LN0_B, DP_LN1_W, DP_LN1_B, DP_LN2_W, DP_LN2_B,
    DP_XR, DP_XW, DP_XK, DP_XV, DP_XA, DP_XG,
    DP_WR, DP_WK, DP_WV, DP_WO,
    DP_W1, DP_W2, DP_W0, DP_A1, DP_A2, DP_A0, DP_V1, DP_V2, DP_V0, DP_G1, DP_G2,
    DP_KK, DP_KA, DP_RK, DP_GNW, DP_GNB,
    DP_FXK, DP_WKEY, DP_WVAL,
    DP_ATT_XPREV, DP_ATT_KV, DP_FFN_XPREV,
    DP_COUNT
};

struct DecodeDesc {
    int B, D, H, L, F, V;
    int Rw, Ra, Rv, Rg;
    int ks_qkv, ks_o, ks_val;
    float ln_eps, gn_eps;
    const void *const *tbl;   // [L][DP_COUNT] device pointers
    const uint16_t *x_in;     // [B][D] bf16 embeddings of the current tokens
    const uint16_t *norm_w, *norm_b, *head_w, *head_b;
    float *logits;            // [B][V]
    // workspace
    float *xa, *xb, *vfirst, *p_qkv, *p_att, *p_val;
    uint16_t *mixed, *yg, *kx, *kact, *hfin;
    unsigned *bar;            // [0] arrival counter, [1] timeout flag
};

#ifdef WKV7C_TIMING
// profiling build only (python -m rwkvtts_amd.build --timing): cycle totals of the head phase's steps, workgroup 0,
// accumulated in the workspace's barrier block at byte offset 64 (tools/decode_phase_profile.py stamps)
#define DSTAMP(i)                                                                                   \
    do {                                                                                            \
        const long long now_ = __builtin_readcyclecounter();                                        \
        if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd((unsigned long long *)(d.bar + 16) + (i), (unsigned long long)(now_ - tprev_)); \
        tprev_ = now_;                                                                              \
    } while (0)
#define DSTAMP_INIT long long tprev_ = __builtin_readcyclecounter()
#else
#define DSTAMP(i) do { } while (0)
#define DSTAMP_INIT do { } while (0)
#endif

namespace {

constexpr int kDecThreads = 256;
constexpr int kRows = 32;                  // row capacity of every scratch matrix (the MFMA B operand is 32 wide)
constexpr int kMaxE = 16;                  // D <= 4096: elements per thread in the row phases
constexpr int kMaxR = 512;                 // Rw + Ra + Rv + Rg
constexpr unsigned kSpinLimit = 1u << 21;  // ~0.1 s: a barrier that is not met by then raises the flag instead of hanging the GPU

constexpr int kHidLD = kMaxR + 8;          // bf16 hidden rows, padded
constexpr int kUpFrags = 16;               // 16-wide k-steps of one up-projection job (a rank of 256)
struct HeadSm {
    __attribute__((aligned(16))) uint16_t hid[2][kHidLD];   // activated low-rank hidden vectors, bf16 like the reference's tensors
    float rkv[3][2][64];
    float up[4][2][64];
    float vec[6][2][64];   // r, decay, k2, v2, a_in, b_in
    float y[2][64];
    float dot[2];
};
union DecSmem {
    float part[3][64][17];
    HeadSm h;
    float red[32];
};

// Pointers read from the layer table are generic: loads through them are flat_load, which counts on BOTH memory counters and
// may return out of order -- the compiler then waits with vmcnt(0) lgkmcnt(0) everywhere (LDS reads behind state loads).  The
// table holds device-memory addresses only.
typedef const uint16_t __attribute__((address_space(1))) *gu16;
typedef float __attribute__((address_space(1))) *gf32;
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef uint32_t uint2v __attribute__((ext_vector_type(2)));
typedef uint16_t __attribute__((address_space(1))) *gu16m;   // written through (the token-shift rows)
#define G_U16M(p) ((gu16m)(p))
#define G_U16(p) ((gu16)(p))
#define G_F32(p) ((gf32)(p))
__device__ __forceinline__ float wave_sum(float x) {
    x = sum16(x);
    x += __shfl_xor(x, 16);
    x += __shfl_xor(x, 32);
    return x;
}

__device__ __forceinline__ float block_sum256(float v, float *red) {
    v = wave_sum(v);
    const int wave = threadIdx.x >> 6;
    __syncthreads();  // red may still be read from the previous call
    if ((threadIdx.x & 63) == 0) red[wave] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

__device__ __forceinline__ float sigm(float x) { return 1.f / (1.f + __expf(-x)); }
__device__ __forceinline__ float tanh_(float x) { return 1.f - 2.f / (__expf(2.f * x) + 1.f); }
// an opaque copy of a scalar: conditions derived from it cannot be hoisted out of the item loop (24 + 24 + 24 loop-invariant
// guards kept as SGPR pairs were 245 spilled SGPRs in the head phase)
__device__ __forceinline__ int fresh_s(int x) {
    asm volatile("" : "+s"(x));
    return x;
}
// log(1 + e^u) with the hardware log: for e^u below 2^-24 the sum rounds to 1 and the result to 0 instead of e^u -- an absolute
// error below 6e-8 in the decay exponent w (libm's log1pf is ~40 instructions with branches on the phase's critical path)
__device__ __forceinline__ float softplus_d(float u) { return u > 20.f ? u : __logf(1.f + __expf(u)); }

// One agent-scope release (L2 write-back) on arrival, a relaxed spin, one agent-scope acquire (cache invalidate) on exit: an
// acquire inside the spin loop would invalidate this XCD's L2 under the workgroups that are still computing.
// mode (debug): 1, 2 = full barrier; 3 = no fences; 4 = release only; 5 = acquire only
__device__ __forceinline__ void grid_barrier(unsigned *bar, unsigned &target, unsigned nwg, int mode = 1) {
    __syncthreads();
    if (threadIdx.x == 0) {
        target += nwg;
        if (mode <= 2 || mode == 4) __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        else __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if ((++spins & 63u) == 0u) {
                if (__hip_atomic_load(bar + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
                if (spins > kSpinLimit) {
                    __hip_atomic_store(bar + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
            }
        }
        if (mode <= 2 || mode == 5) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}

__device__ __forceinline__ float4 bf4(uint2 r) {
    return make_float4(__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u), __uint_as_float(r.y << 16),
                       __uint_as_float(r.y & 0xffff0000u));
}
__device__ __forceinline__ float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

// Every thread owns float4 column groups g = tid + 256 i, i < NG (NG = ceil(D / 1024): the host picks the instantiation).  All
// loads of the row (residual, the partial sums eight at a time, norm and lerp parameters, the shifted row) are unconditional --
// lanes beyond the row repeat its last group, partial sums beyond `nparts` repeat the last one and are dropped -- and are issued
// before the first reduction: the phase pays one memory latency.  (Round 2's form guarded each load by `g < D / 4` and
// `p < nparts`: every guarded load is waited for behind its issue, and the parameter rows came through generic pointers, i.e.
// flat_load -- four serialised latencies per phase.)
template <int NMIX, int NG>
__device__ __forceinline__ void row_phase(const DecodeDesc &d, int b, float *red, const float *x_old, const float *parts, int nparts,
                                          const uint16_t *x_in, gu16 ln0w, gu16 ln0b, float *x_out, gu16 lnw, gu16 lnb, gu16m x_prev,
                                          const gu16 *mixp, uint16_t *out) {
// the NG instantiations (per-phase kernels: the model's; persistent kernel: the widest) must round alike: no reassociation of the
// row sums under -ffast-math, no contraction left to the optimiser (it differs between the instantiations)
#pragma clang fp reassociate(off) contract(off)
    typedef const uint2v __attribute__((address_space(1))) *gq;
    const int D = d.D, tid = threadIdx.x, D4 = D >> 2;
    const float invD = 1.f / (float)D;
    const long rb = (long)b * D;
    float4 x[NG];
    uint2v wln[NG], bln[NG], xp[NG], mx[NMIX > 0 ? NMIX : 1][NG];
    bool live[NG];
    int col[NG];
#pragma unroll
    for (int i = 0; i < NG; i++) {
        const int g = tid + kDecThreads * i;
        live[i] = g < D4;
        col[i] = 4 * min(g, D4 - 1);
        wln[i] = *(gq)(lnw + col[i]);
        bln[i] = *(gq)(lnb + col[i]);
        if (NMIX > 0) {
            xp[i] = *(gq)(x_prev + rb + col[i]);
#pragma unroll
            for (int j = 0; j < NMIX; j++) mx[j][i] = *(gq)(mixp[j] + col[i]);
        }
    }
    float s = 0.f;
    if (x_in) {   // layer 0: the embeddings (scalar branch)
#pragma unroll
        for (int i = 0; i < NG; i++) x[i] = bf4(*reinterpret_cast<const uint2 *>(x_in + rb + col[i]));
    } else {
#pragma unroll
        for (int i = 0; i < NG; i++) x[i] = *reinterpret_cast<const float4 *>(x_old + rb + col[i]);
        for (int p0 = 0; p0 < nparts; p0 += 8) {
            float4 t[NG][8];
#pragma unroll
            for (int i = 0; i < NG; i++)
#pragma unroll
                for (int p = 0; p < 8; p++)
                    t[i][p] = *reinterpret_cast<const float4 *>(parts + ((long)min(p0 + p, nparts - 1) * kRows + b) * D + col[i]);
#pragma unroll
            for (int i = 0; i < NG; i++)
#pragma unroll
                for (int p = 0; p < 8; p++) {
                    const float m = p0 + p < nparts ? 1.f : 0.f;
                    x[i].x = fmaf(t[i][p].x, m, x[i].x); x[i].y = fmaf(t[i][p].y, m, x[i].y);
                    x[i].z = fmaf(t[i][p].z, m, x[i].z); x[i].w = fmaf(t[i][p].w, m, x[i].w);
                }
        }
    }
#pragma unroll
    for (int i = 0; i < NG; i++) s += live[i] ? (x[i].x + x[i].y) + (x[i].z + x[i].w) : 0.f;
    auto sqdev = [&](float mean) {
#pragma clang fp reassociate(off) contract(off)
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NG; i++) {
            const float a = x[i].x - mean, bb = x[i].y - mean, c = x[i].z - mean, e = x[i].w - mean;
            q += live[i] ? (a * a + bb * bb) + (c * c + e * e) : 0.f;
        }
        return q;
    };
    if (x_in) {  // pre_norm of the first block (rwkv_s2s_single_ffn.py:253-254); its output is a bf16 tensor
        uint2v w0[NG], b0[NG];
#pragma unroll
        for (int i = 0; i < NG; i++) {
            w0[i] = *(gq)(ln0w + col[i]);
            b0[i] = *(gq)(ln0b + col[i]);
        }
        const float mean = block_sum256(s, red) * invD;
        const float rstd = rsqrtf(block_sum256(sqdev(mean), red) * invD + d.ln_eps);
        s = 0.f;
#pragma unroll
        for (int i = 0; i < NG; i++) {
            const float4 w = bf4(make_uint2(w0[i].x, w0[i].y)), bi = bf4(make_uint2(b0[i].x, b0[i].y));
            const uint32_t lo = cvt_pk((x[i].x - mean) * rstd * w.x + bi.x, (x[i].y - mean) * rstd * w.y + bi.y);
            const uint32_t hi = cvt_pk((x[i].z - mean) * rstd * w.z + bi.z, (x[i].w - mean) * rstd * w.w + bi.w);
            x[i] = bf4(make_uint2(lo, hi));
            s += live[i] ? (x[i].x + x[i].y) + (x[i].z + x[i].w) : 0.f;
        }
    }
    const float mean = block_sum256(s, red) * invD;
    const float rstd = rsqrtf(block_sum256(sqdev(mean), red) * invD + d.ln_eps);
#pragma unroll
    for (int i = 0; i < NG; i++) {
        if (live[i]) {
            const int c = col[i];
            if (x_out) *reinterpret_cast<float4 *>(x_out + rb + c) = x[i];
            const float4 w = bf4(make_uint2(wln[i].x, wln[i].y)), bi = bf4(make_uint2(bln[i].x, bln[i].y));
            const uint2 hb = make_uint2(cvt_pk((x[i].x - mean) * rstd * w.x + bi.x, (x[i].y - mean) * rstd * w.y + bi.y),
                                        cvt_pk((x[i].z - mean) * rstd * w.z + bi.z, (x[i].w - mean) * rstd * w.w + bi.w));
            if (NMIX == 0) {
                *reinterpret_cast<uint2 *>(out + rb + c) = hb;
            } else {
                const float4 h = bf4(hb), pv = bf4(make_uint2(xp[i].x, xp[i].y));
                const float4 xx = make_float4(pv.x - h.x, pv.y - h.y, pv.z - h.z, pv.w - h.w);
#pragma unroll
                for (int j = 0; j < NMIX; j++) {
                    const float4 m = bf4(make_uint2(mx[j][i].x, mx[j][i].y));
                    *reinterpret_cast<uint2 *>(out + ((long)j * kRows + b) * D + c) =
                        make_uint2(cvt_pk(fmaf(xx.x, m.x, h.x), fmaf(xx.y, m.y, h.y)), cvt_pk(fmaf(xx.z, m.z, h.z), fmaf(xx.w, m.w, h.w)));
                }
                uint2v hv;
                hv.x = hb.x; hv.y = hb.y;
                *(uint2v __attribute__((address_space(1))) *)(x_prev + rb + c) = hv;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// GEMV phases: out[ks][n][col] = sum_{k in split ks} X[n][k] W[col][k]; one item = 32 columns x one K split.
// D[m][n]: m = output column inside the tile (A operand = W rows), n = sequence (B operand = X rows).
// ---------------------------------------------------------------------------------------------------------------------
struct GemvSeg {
    gu16 W;              // [ncols][K]; global address space: through a generic pointer the weight rows are flat_load, which the
                         // compiler drains with vmcnt(0) every two k-steps (four serialised latencies per sweep, round 2)
    const uint16_t *X;   // bf16 [32][K]
    int ntiles;          // 32-column tiles (the last one may be partial: ncols)
    int ncols;
};

#ifndef DEC_NT_WEIGHTS
#define DEC_NT_WEIGHTS 0
#endif
template <int KSTEPS>
__device__ __forceinline__ void gemv_steps(f32x16 &acc, gu16 wp, const uint16_t *xp) {
    bf16x8 a[KSTEPS], b[KSTEPS];
#pragma unroll
    for (int i = 0; i < KSTEPS; i++) {
#if DEC_NT_WEIGHTS
        // weight rows are read once per step by one workgroup (0.65 GB per step, more than L2 + MALL hold): non-temporal
        a[i] = __builtin_nontemporal_load((const bf16x8 __attribute__((address_space(1))) *)(wp + 16 * i));
#else
        a[i] = *(const bf16x8 __attribute__((address_space(1))) *)(wp + 16 * i);
#endif
        b[i] = *reinterpret_cast<const bf16x8 *>(xp + 16 * i);
    }
    // all loads of the round are issued before the first MFMA: left alone, the scheduler sinks each pair of loads to its MFMA
    // (shorter live ranges) and the sweep walks through its K range with 2.5 k-steps in flight
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < KSTEPS; i++) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[i], acc, 0, 0, 0);
}

// OUTMODE 0: fp32 partials out[ks][n][col] (+ bias);  1: KS = 1 and out is bf16 [n][col] = relu(.)^2 (the channel-mix key)
// TW: columns per tile.  16: the MFMA's 32 A rows hold each of the 16 weight rows twice (the upper half of the result is
// ignored) -- for the sweep that cannot split K (OUTMODE 1: F / 32 = 128 items would leave half of the CUs idle).
template <int OUTMODE, int NSEG, int TW = 32>
__device__ __forceinline__ void gemv_phase(const DecodeDesc &d, DecSmem &sm, const GemvSeg (&segs)[NSEG], int K, int KS, void *out_,
                                           int ldo, const uint16_t *bias) {
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    int ntiles = 0;
#pragma unroll
    for (int s = 0; s < NSEG; s++) ntiles += segs[s].ntiles;
    const int nitems = ntiles * KS;
    const int kw = K / KS / 4;  // K range of one wave (multiple of 16)
    const int nrow = min(lane & 31, d.B - 1);
    for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
        const int tile = item / KS, ks = item - tile * KS;
        // segment of this tile (unrolled with constant indices: the table stays in registers)
        GemvSeg sg = segs[0];
        int t = tile, col_base = 0, first_tile = 0, first_col = 0;
#pragma unroll
        for (int s = 1; s < NSEG; s++) {
            first_tile += segs[s - 1].ntiles;
            first_col += segs[s - 1].ncols;
            const bool here = tile >= first_tile;   // segments are in ascending tile order: the last match wins
            sg.W = here ? segs[s].W : sg.W;
            sg.X = here ? segs[s].X : sg.X;
            sg.ntiles = here ? segs[s].ntiles : sg.ntiles;
            sg.ncols = here ? segs[s].ncols : sg.ncols;
            t = here ? tile - first_tile : t;
            col_base = here ? first_col : col_base;
        }
        const int c0 = t * TW;                                  // first column of the tile inside its segment
        const int mrow = min(c0 + (lane & (TW - 1)), sg.ncols - 1);
        const int kbeg = ks * (K / KS) + wave * kw + (lane >> 5) * 8;
        gu16 wp = sg.W + (long)mrow * K + kbeg;
        const uint16_t *xp = sg.X + (long)nrow * K + kbeg;
        f32x16 acc = zero16();
        int k = 0;
        for (; k + 256 <= kw; k += 256) gemv_steps<16>(acc, wp + k, xp + k);   // un-split sweeps (key, head): one round of loads
        for (; k + 128 <= kw; k += 128) gemv_steps<8>(acc, wp + k, xp + k);
        for (; k + 32 <= kw; k += 32) gemv_steps<2>(acc, wp + k, xp + k);
        for (; k + 16 <= kw; k += 16) gemv_steps<1>(acc, wp + k, xp + k);
        __syncthreads();  // part[] of the previous item has been consumed
        if (wave > 0) {
#pragma unroll
            for (int r = 0; r < 16; r++) sm.part[wave - 1][lane][r] = acc[r];
        }
        __syncthreads();
        if (wave == 0) {
            const int n = lane & 31;
            float *op = (float *)out_ + ((long)ks * kRows + n) * ldo + col_base + c0;
            uint16_t *ob = (uint16_t *)out_ + (long)n * ldo + col_base + c0;
            const bool vec = (ldo & 3) == 0 && ((col_base + c0) & 3) == 0;
#pragma unroll
            for (int j = 0; j < TW / 8; j++) {
                const int c = 8 * j + 4 * (lane >> 5);  // 4 consecutive columns of the tile
                float v[4];
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    v[i] = acc[4 * j + i] + sm.part[0][lane][4 * j + i] + sm.part[1][lane][4 * j + i] + sm.part[2][lane][4 * j + i];
                    if (bias && c0 + c + i < sg.ncols) v[i] += bf2f(bias[col_base + c0 + c + i]);
                }
                if (OUTMODE == 1) {
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        v[i] = fmaxf(v[i], 0.f);
                        v[i] *= v[i];
                    }
                    if (n < d.B) *reinterpret_cast<uint2 *>(ob + c) = make_uint2(cvt_pk(v[0], v[1]), cvt_pk(v[2], v[3]));  // F % 64 == 0
                } else if (n < d.B) {
                    if (vec && c0 + c + 3 < sg.ncols) {   // write-through, like the state rows: read next by workgroups on other XCDs
                        f32x4v t4;
                        t4.x = v[0]; t4.y = v[1]; t4.z = v[2]; t4.w = v[3];
                        const unsigned long long sp = (unsigned long long)(op + c);
                        asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(sp), "v"(t4) : "memory");
                    } else {
#pragma unroll
                        for (int i = 0; i < 4; i++)
                            if (c0 + c + i < sg.ncols) op[c + i] = v[i];
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// head phase: everything between the projections and the output projection, for one head and two sequences.
// Every load of the phase is issued at its top, in the order of use (partial sums, up-projection rows as MFMA fragments,
// per-channel parameters, state rows), so the phase pays ONE memory latency (3.8k of its 10.9k cycles; tools/decode_head_timing.py).
// Round 2's form of this phase took 22.3k cycles for the same arithmetic; what the ISA showed (round 3):
//   * the wave id was a VGPR value, so the per-wave job sizes were divergent and each of the 24 guarded fragment loads / LDS reads /
//     MFMAs was an exec-masked region of its own;
//   * a guarded load (`if (i < n) x[i] = load`) is merged with "no value" by a phi and waited for right behind its issue: the
//     fragment loads and, in step B, every LDS read in front of its MFMA, were serialised latencies.  Now all loads are
//     unconditional with clamped slots (NF1 / NF2 template slots), only register-only work sits behind scalar branches;
//   * the layer's pointers were read inside the item loop, i.e. behind the kernel's stores: vector loads in the data's in-order
//     queue (partial sums -> wait -> pointers -> wait -> rows), and generic pointers, so the rows came through flat_load, which
//     ties the LDS counter to the global one.  Now scalar loads before the loop, cast to the global address space;
//   * one kernel held all nine phase bodies: 245 spilled SGPRs in this phase.  Now one kernel per phase.
// ---------------------------------------------------------------------------------------------------------------------
// NF1, NF2: fragment slots of a wave's two jobs (NF1 >= max(Rw, Ra) / 16, NF2 >= max(Rv, Rg) / 16; the host picks the smallest
// instantiation)
template <int NF1, int NF2, class LP>
__device__ __forceinline__ void head_phase(const DecodeDesc &d, DecSmem &smu, int l, const LP &lp) {
    HeadSm &sm = smu.h;
    const int tid = threadIdx.x, D = d.D, H = d.H;
    const int N2 = 3 * D + d.Rw + d.Ra + d.Rv + d.Rg;       // columns of the qkv/low-rank partials
    const int Rtot = d.Rw + d.Ra + d.Rv + d.Rg;
    const int oA = d.Rw, oV = d.Rw + d.Ra, oG = d.Rw + d.Ra + d.Rv;
    const int npair = (d.B + 1) / 2, nitems = H * npair;
    const bool first = l == 0;
    // thread roles: B (up projections on MFMA): wave 0/1 = 32-channel tile 0/1 of the w and v branches, wave 2/3 = tile 0/1 of
    // the a and g branches; C/E: waves 0,1 = sequence, lane = channel; D (state): 128 threads per sequence, 16 lanes x float4
    // = one state row, 8 rows per pass.  The wave id is made a scalar (readfirstlane): the wave's job sizes are then scalars and
    // everything that depends on them is scalar control flow.
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int bbD = wave >> 1, tt = tid & 127, k4 = (tt & 15) * 4, vr = tt >> 4;
    const int tileB = wave & 1;
    const bool wv = wave < 2;
    const int R1 = wv ? d.Rw : d.Ra, R2 = wv ? d.Rv : d.Rg;     // the wave's two jobs (layer 0 has no v branch: n2 = 0)
    const int off1 = wv ? 0 : oA, off2 = wv ? oV : oG;
    const int n1 = R1 >> 4, n2 = (wv && first) ? 0 : R2 >> 4;
    const int nhid = 2 * Rtot;   // activated hidden values of the item (<= 1024: four per thread)
    // The layer's pointers, read here -- before the first store of the kernel -- so that they are scalar loads (behind a store the
    // compiler must assume the table may have changed and reads it with vector loads, in the same in-order queue as the data).
    gu16 p_w2 = G_U16(lp[DP_W2]), p_a2 = G_U16(lp[DP_A2]);
    gu16 p_v2 = G_U16(lp[DP_V2]), p_g2 = G_U16(lp[DP_G2]);
    gu16 k_k = G_U16(lp[DP_KK]), k_a = G_U16(lp[DP_KA]), r_k = G_U16(lp[DP_RK]);
    gu16 gnw = G_U16(lp[DP_GNW]), gnb = G_U16(lp[DP_GNB]);
    gu16 w0 = G_U16(lp[DP_W0]), a0 = G_U16(lp[DP_A0]), p_v0 = G_U16(lp[DP_V0]);
    gf32 kv_all = G_F32(lp[DP_ATT_KV]);
    gu16 up1 = wv ? p_w2 : p_a2;
    gu16 up2 = wv ? (first ? p_w2 : p_v2) : p_g2;   // layer 0: null v pointers, never dereferenced
    gu16 v0 = first ? w0 : p_v0;
    for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
        const int h = item % H, bp = item / H;
        const int b0 = 2 * bp, b1 = min(b0 + 1, d.B - 1);
        // opaque copies: the guards below must not be hoisted out of the item loop as 2 x 24 live SGPR pairs
        const int m1 = fresh_s(n1), m2 = fresh_s(n2);
        DSTAMP_INIT;
        // ---- loads in the order of their use, all of them unconditional (clamped addresses): a guarded load is merged with "no
        // value" by a phi, and the compiler waits for it right behind its issue.
        // A: K-split partial sums of this head's r, k, v (384 values: e = tid and 256 + (tid & 127)) and of the low-rank hidden
        // vectors (idx = tid + 256 j < 2 Rtot)
        long addr[6];
        {
            const int e0 = tid, e1 = 256 + (tid & 127);
            addr[0] = (long)(((e0 >> 6) & 1) ? b1 : b0) * N2 + (e0 >> 7) * D + h * 64 + (e0 & 63);
            addr[1] = (long)(((e1 >> 6) & 1) ? b1 : b0) * N2 + (e1 >> 7) * D + h * 64 + (e1 & 63);
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int idx = min(tid + kDecThreads * j, nhid - 1);
                const int bb = idx >= Rtot, r = idx - (bb ? Rtot : 0);
                addr[2 + j] = (long)(bb ? b1 : b0) * N2 + 3 * D + r;
            }
        }
        float accA[6];
        const float *pp = d.p_qkv;
        const long ps = (long)kRows * N2;
        const bool two = d.ks_qkv > 1;
        const long ps1 = two ? ps : 0;   // one split: the second load repeats the first and is dropped below
        float t0[6], t1[6];
#pragma unroll
        for (int it = 0; it < 6; it++) {
            t0[it] = pp[addr[it]];
            t1[it] = pp[ps1 + addr[it]];
        }
        __builtin_amdgcn_sched_barrier(0);
        // B: up-projection rows as MFMA A fragments: lane = channel 32 tile + (lane & 31), k = 16 i + 8 (lane >> 5); slots beyond
        // the job's fragments repeat fragment 0
        bf16x8 wf1[NF1], wf2[NF2];
        {
            const int chB = h * 64 + tileB * 32 + (lane & 31);
            gu16 row1 = up1 + (long)chB * R1 + (lane >> 5) * 8;
            gu16 row2 = up2 + (long)chB * (m2 ? R2 : R1) + (lane >> 5) * 8;
#pragma unroll
            for (int i = 0; i < NF1; i++) wf1[i] = *(const bf16x8 __attribute__((address_space(1))) *)(row1 + 16 * (i < m1 ? i : 0));
#pragma unroll
            for (int i = 0; i < NF2; i++) wf2[i] = *(const bf16x8 __attribute__((address_space(1))) *)(row2 + 16 * (i < m2 ? i : 0));
        }
        __builtin_amdgcn_sched_barrier(0);
        // C, E: per-channel parameters (waves 0, 1 use them)
        const int chC = h * 64 + lane;
        const uint16_t q_kk = k_k[chC], q_ka = k_a[chC], q_rk = r_k[chC], q_gw = gnw[chC], q_gb = gnb[chC];
        const uint16_t q_w0 = w0[chC], q_a0 = a0[chC], q_v0 = v0[chC];
        const float q_vf = d.vfirst[(long)((wave & 1) ? b1 : b0) * D + chC];   // layer 0: stale values, not used
        __builtin_amdgcn_sched_barrier(0);
        // D: the state rows
        const bool liveD = b0 + bbD < d.B;
        gf32 S = kv_all + ((long)(bbD ? b1 : b0) * H + h) * 64 * 64;
        float4 st[8];
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const f32x4v t = *(const f32x4v __attribute__((address_space(1))) *)(S + (vr + 8 * i) * 64 + k4);
            st[i] = make_float4(t.x, t.y, t.z, t.w);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int it = 0; it < 6; it++) accA[it] = t0[it] + (two ? t1[it] : 0.f);
        for (int p = 2; p < d.ks_qkv; p += 2) {   // K splits beyond two (wide models): two per round of loads
            const float *pq = pp + (long)p * ps;
            float u0[6], u1[6];
#pragma unroll
            for (int it = 0; it < 6; it++) {
                u0[it] = pq[addr[it]];
                u1[it] = pq[ps + addr[it]];
            }
#pragma unroll
            for (int it = 0; it < 6; it++) accA[it] += u0[it] + u1[it];
        }
        __syncthreads();  // LDS of the previous item is free
        DSTAMP(0);
        // ---- A: low-rank hidden vectors (activation applied to the summed partials) and this head's r, k, v
        sm.rkv[tid >> 7][(tid >> 6) & 1][tid & 63] = accA[0];
        if (tid < 128) sm.rkv[2][(tid >> 6) & 1][tid & 63] = accA[1];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int idx = tid + kDecThreads * j;
            if (idx < nhid) {
                const int bb = idx >= Rtot, r = idx - (bb ? Rtot : 0);
                const float x = accA[2 + j];
                // tanh (decay branch), identity (a, v branches), sigmoid (gate branch): one exp, one reciprocal
                const float e = __expf(r < oA ? 2.f * x : -x), rc = 1.f / (e + 1.f);
                const float v = r < oA ? fmaf(-2.f, rc, 1.f) : (r >= oG ? rc : x);
                sm.hid[bb][r] = f2bf(v);
            }
        }
        __syncthreads();
        DSTAMP(1);
        // ---- B: up projections on MFMA: D[m = channel][n = sequence], only n = 0, 1 are real (the B operand repeats them)
        {
            const uint16_t *hp = &sm.hid[lane & 1][(lane >> 5) * 8];
            bf16x8 hf1[NF1], hf2[NF2];
#pragma unroll
            for (int i = 0; i < NF1; i++) hf1[i] = *reinterpret_cast<const bf16x8 *>(hp + off1 + 16 * (i < m1 ? i : 0));
#pragma unroll
            for (int i = 0; i < NF2; i++) hf2[i] = *reinterpret_cast<const bf16x8 *>(hp + off2 + 16 * (i < m2 ? i : 0));
            f32x16 acc1 = zero16(), acc2 = zero16();
#pragma unroll
            for (int i = 0; i < NF1; i++)   // scalar branches around register-only work
                if (i < m1) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf1[i], hf1[i], acc1, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < NF2; i++)
                if (i < m2) acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf2[i], hf2[i], acc2, 0, 0, 0);
            if ((lane & 31) < 2) {
                const int n = lane & 31;
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int m = tileB * 32 + d_row(r, lane);
                    sm.up[wv ? 0 : 1][n][m] = acc1[r];
                    sm.up[wv ? 2 : 3][n][m] = acc2[r];
                }
            }
        }
        __syncthreads();
        DSTAMP(2);
        // ---- C: decay, gates, value residual, kk normalisation (rwkv_s2s_single_ffn.py:493-500)
        if (wave < 2) {
            const int c = lane, bb = wave;
            const int b = bb ? b1 : b0, ch = h * 64 + c;
            const float p_kk = bf2f(q_kk), p_ka = bf2f(q_ka), p_rk = bf2f(q_rk);
            const float r = sm.rkv[0][bb][c], k = sm.rkv[1][bb][c];
            float v = sm.rkv[2][bb][c];
            const float w = -softplus_d(-(sm.up[0][bb][c] + bf2f(q_w0))) - 0.5f;
            const float a = sigm(sm.up[1][bb][c] + bf2f(q_a0));
            if (first) {
                if (b0 + bb < d.B) d.vfirst[(long)b * D + ch] = v;
            } else {
                v = fmaf(q_vf - v, sigm(sm.up[2][bb][c] + bf2f(q_v0)), v);
            }
            const float kkr = k * p_kk;
            const float ss = wave_sum(kkr * kkr);
            const float kk = kkr / fmaxf(sqrtf(ss), 1e-12f);
            const float k2 = k * fmaf(a - 1.f, p_ka, 1.f);
            const float dot = wave_sum(r * k2 * p_rk);
            sm.vec[0][bb][c] = r;
            sm.vec[1][bb][c] = __expf(-__expf(w));
            sm.vec[2][bb][c] = k2;
            sm.vec[3][bb][c] = v;
            sm.vec[4][bb][c] = -kk;
            sm.vec[5][bb][c] = kk * a;
            if (c == 0) sm.dot[bb] = dot;
        }
        __syncthreads();
        DSTAMP(3);
        // ---- D: state update in place
        {
            const float4 rr = *reinterpret_cast<const float4 *>(&sm.vec[0][bbD][k4]);
            const float4 dc = *reinterpret_cast<const float4 *>(&sm.vec[1][bbD][k4]);
            const float4 kk = *reinterpret_cast<const float4 *>(&sm.vec[2][bbD][k4]);
            const float4 aa = *reinterpret_cast<const float4 *>(&sm.vec[4][bbD][k4]);
            const float4 bv = *reinterpret_cast<const float4 *>(&sm.vec[5][bbD][k4]);
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const float vv = sm.vec[3][bbD][vr + 8 * i];
                const float sa = sum16(st[i].x * aa.x + st[i].y * aa.y + st[i].z * aa.z + st[i].w * aa.w);
                st[i].x = fmaf(st[i].x, dc.x, fmaf(sa, bv.x, vv * kk.x));
                st[i].y = fmaf(st[i].y, dc.y, fmaf(sa, bv.y, vv * kk.y));
                st[i].z = fmaf(st[i].z, dc.z, fmaf(sa, bv.z, vv * kk.z));
                st[i].w = fmaf(st[i].w, dc.w, fmaf(sa, bv.w, vv * kk.w));
                if (liveD) {
                    f32x4v t;
                    t.x = st[i].x; t.y = st[i].y; t.z = st[i].z; t.w = st[i].w;
                    // write-through (sc1): the 8 MB of state rows are not left dirty in the L2s at the kernel boundary (a boundary costs
                    // its predecessor's dirty bytes / 6 TB/s on top, MI355X_MICROARCH.md: 1.3 us here)
                    const unsigned long long sp = (unsigned long long)(S + (vr + 8 * i) * 64 + k4);
                    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(sp), "v"(t) : "memory");
                }
                const float y = sum16(st[i].x * rr.x + st[i].y * rr.y + st[i].z * rr.z + st[i].w * rr.w);
                if ((tt & 15) == 0) sm.y[bbD][vr + 8 * i] = y;
            }
        }
        __syncthreads();
        DSTAMP(4);
        // ---- E: GroupNorm over the head, bonus, gate (rwkv_s2s_single_ffn.py:504-505)
        if (wave < 2) {
            const int c = lane, bb = wave;
            const int ch = h * 64 + c;
            const float y = sm.y[bb][c];
            const float mean = wave_sum(y) * (1.f / 64.f);
            const float dv = y - mean;
            const float rstd = rsqrtf(wave_sum(dv * dv) * (1.f / 64.f) + d.gn_eps);
            const float o = (fmaf(dv * rstd, bf2f(q_gw), bf2f(q_gb)) + sm.dot[bb] * sm.vec[3][bb][c]) * sm.up[3][bb][c];
            if (b0 + bb < d.B) d.yg[(long)(b0 + bb) * D + ch] = f2bf(o);
        }
        DSTAMP(5);
    }
}

// the layer's pointers: a row of the device table (persistent kernel) ...
struct TblRow {
    const void *const *base;
    __device__ __forceinline__ const void *operator[](int i) const { return base[i]; }
};
// ... or, in the one-kernel-per-phase mode, the phase's own pointers as kernel arguments: no dependent table load between the
// kernel arguments and the first data.  (Round 2 measured all 38 pointers as arguments of the all-phases kernel SLOWER by 0.5-1.5 us
// per phase -- 300 bytes of kernarg and 76 more live SGPRs in a kernel that already spilled them; a phase needs 1 to 13.)
__host__ __device__ constexpr int dp_slot(int ph, int dp) {
    switch (ph) {
    case 0:
        switch (dp) {
        case DP_LN0_W: return 0; case DP_LN0_B: return 1; case DP_LN1_W: return 2; case DP_LN1_B: return 3; case DP_XR: return 4;
        case DP_XW: return 5; case DP_XK: return 6; case DP_XV: return 7; case DP_XA: return 8; case DP_XG: return 9;
        case DP_ATT_XPREV: return 10; default: return -1;
        }
    case 1:
        switch (dp) {
        case DP_WR: return 0; case DP_WK: return 1; case DP_WV: return 2; case DP_W1: return 3; case DP_A1: return 4; case DP_V1: return 5;
        case DP_G1: return 6; default: return -1;
        }
    case 2:
        switch (dp) {
        case DP_W2: return 0; case DP_A2: return 1; case DP_V2: return 2; case DP_G2: return 3; case DP_KK: return 4; case DP_KA: return 5;
        case DP_RK: return 6; case DP_GNW: return 7; case DP_GNB: return 8; case DP_W0: return 9; case DP_A0: return 10;
        case DP_V0: return 11; case DP_ATT_KV: return 12; default: return -1;
        }
    case 3: return dp == DP_WO ? 0 : -1;
    case 4:
        switch (dp) {
        case DP_LN2_W: return 0; case DP_LN2_B: return 1; case DP_FXK: return 2; case DP_FFN_XPREV: return 3; default: return -1;
        }
    case 5: return dp == DP_WKEY ? 0 : -1;
    case 6: return dp == DP_WVAL ? 0 : -1;
    default: return -1;
    }
}
__host__ __device__ constexpr int dp_count(int ph) {
    int n = 0;
    for (int dp = 0; dp < DP_COUNT; dp++) n += dp_slot(ph, dp) >= 0;
    return n;
}
template <int PH>
struct ArgRow {
    const void *p[dp_count(PH) ? dp_count(PH) : 1];
    __device__ __forceinline__ const void *operator[](int dp) const { return p[dp_slot(PH, dp) >= 0 ? dp_slot(PH, dp) : 0]; }
};

// PH 0-6: the phases of layer l; PH 7, 8: the tail (last residual add + model norm; head projection).  P1, P2: row phases:
// P1 = NG (float4 groups per thread); head phase: fragment slots NF1, NF2.
template <int PH, int P1, int P2, class LP>
__device__ __forceinline__ void run_phase(const DecodeDesc &d, DecSmem &sm, int l, const LP &lp) {
    const int D = d.D;
    if constexpr (PH == 7) {
        for (int b = blockIdx.x; b < d.B; b += gridDim.x)
            row_phase<0, P1>(d, b, sm.red, d.xa, d.p_val, d.ks_val, nullptr, nullptr, nullptr, nullptr, G_U16(d.norm_w), G_U16(d.norm_b),
                             nullptr, nullptr, d.hfin);
    } else if constexpr (PH == 8) {
        const GemvSeg seg[1] = {{G_U16(d.head_w), d.hfin, (d.V + 31) / 32, d.V}};
        gemv_phase<0, 1>(d, sm, seg, D, 1, d.logits, d.V, d.head_b);
    } else if constexpr (PH == 0) {
        const gu16 mixp[6] = {G_U16(lp[DP_XR]), G_U16(lp[DP_XW]), G_U16(lp[DP_XK]), G_U16(lp[DP_XV]), G_U16(lp[DP_XA]), G_U16(lp[DP_XG])};
        const gu16 ln0w = G_U16(lp[DP_LN0_W]), ln0b = G_U16(lp[DP_LN0_B]), ln1w = G_U16(lp[DP_LN1_W]), ln1b = G_U16(lp[DP_LN1_B]);
        const gu16m xprev = G_U16M(lp[DP_ATT_XPREV]);
        for (int b = blockIdx.x; b < d.B; b += gridDim.x)
            row_phase<6, P1>(d, b, sm.red, d.xa, d.p_val, d.ks_val, l == 0 ? d.x_in : nullptr, ln0w, ln0b, d.xb, ln1w, ln1b, xprev, mixp,
                             d.mixed);
    } else if constexpr (PH == 1) {
        const long RS = (long)kRows * D;  // one mixed plane: order r, w, k, v, a, g
        // layer 0 has no value-residual branch: its columns stay unwritten and unread
        const GemvSeg segs[7] = {{G_U16(lp[DP_WR]), d.mixed + 0 * RS, D / 32, D},
                                 {G_U16(lp[DP_WK]), d.mixed + 2 * RS, D / 32, D},
                                 {G_U16(lp[DP_WV]), d.mixed + 3 * RS, D / 32, D},
                                 {G_U16(lp[DP_W1]), d.mixed + 1 * RS, d.Rw / 32, d.Rw},
                                 {G_U16(lp[DP_A1]), d.mixed + 4 * RS, d.Ra / 32, d.Ra},
                                 {G_U16(l == 0 ? lp[DP_A1] : lp[DP_V1]), d.mixed + 3 * RS, l == 0 ? 0 : d.Rv / 32, d.Rv},
                                 {G_U16(lp[DP_G1]), d.mixed + 5 * RS, d.Rg / 32, d.Rg}};
        gemv_phase<0, 7>(d, sm, segs, D, d.ks_qkv, d.p_qkv, 3 * D + d.Rw + d.Ra + d.Rv + d.Rg, nullptr);
    } else if constexpr (PH == 2) {
        head_phase<P1, P2>(d, sm, l, lp);
    } else if constexpr (PH == 3) {
        const GemvSeg seg[1] = {{G_U16(lp[DP_WO]), d.yg, D / 32, D}};
        gemv_phase<0, 1>(d, sm, seg, D, d.ks_o, d.p_att, D, nullptr);
    } else if constexpr (PH == 4) {
        const gu16 mixp[1] = {G_U16(lp[DP_FXK])};
        const gu16 ln2w = G_U16(lp[DP_LN2_W]), ln2b = G_U16(lp[DP_LN2_B]);
        const gu16m xprev = G_U16M(lp[DP_FFN_XPREV]);
        for (int b = blockIdx.x; b < d.B; b += gridDim.x)
            row_phase<1, P1>(d, b, sm.red, d.xb, d.p_att, d.ks_o, nullptr, nullptr, nullptr, d.xa, ln2w, ln2b, xprev, mixp, d.kx);
    } else if constexpr (PH == 5) {   // P1 = columns per tile: 16 while F / 32 tiles would leave CUs idle (0.4B: 128), else 32
        const GemvSeg seg[1] = {{G_U16(lp[DP_WKEY]), d.kx, d.F / P1, d.F}};
        gemv_phase<1, 1, P1>(d, sm, seg, D, 1, d.kact, d.F, nullptr);
    } else {
        const GemvSeg seg[1] = {{G_U16(lp[DP_WVAL]), d.kact, D / 32, D}};
        gemv_phase<0, 1>(d, sm, seg, d.F, d.ks_val, d.p_val, D, nullptr);
    }
}

constexpr int kGrid = 256;
constexpr int kSplitCap = 256;   // workgroups a GEMV phase counts on (512 -- two per CU, finer K splits -- measured 15 % slower)

// Three instantiations of the width-dependent phases, shared by both launch modes so that they round alike (-ffast-math contracts
// and reassociates differently in different instantiations): NG float4 groups per thread in the row phases, NF1 / NF2 fragment
// slots in the head phase.  0: D <= 1024, ranks <= 64 / 128 (0.4B); 1: D <= 2048, ranks <= 128 / 256 (1.5B); 2: D <= 4096, ranks <= 256.
template <int V> struct Variant;
template <> struct Variant<0> { static constexpr int NG = 1, NF1 = 4, NF2 = 8; };
template <> struct Variant<1> { static constexpr int NG = 2, NF1 = 8, NF2 = 16; };
template <> struct Variant<2> { static constexpr int NG = kMaxE / 4, NF1 = kUpFrags, NF2 = kUpFrags; };

#ifdef RWKV7_LAB   // the one-launch variant lost (2.4 ms against 0.93 ms per step): lab build only (python -m rwkvtts_amd.build --lab)
// mode 1: the step; mode 2 (debug): the barriers alone
template <int V>
__global__ __launch_bounds__(kDecThreads) void decode_persistent_kernel(DecodeDesc d, int mode) {
    using W = Variant<V>;
    __shared__ DecSmem sm;
    unsigned target = 0;
    const int nphase = 7 * d.L + 2;
    for (int idx = 0; idx < nphase; idx++) {   // one call site per phase body
        const int l = idx / 7, ph = idx - 7 * l + (l == d.L ? 7 : 0);
        if (mode == 1) {
            const TblRow lp{d.tbl + (long)(l < d.L ? l : 0) * DP_COUNT};
            switch (ph) {
            case 0: run_phase<0, W::NG, 0>(d, sm, l, lp); break;
            case 1: run_phase<1, 0, 0>(d, sm, l, lp); break;
            case 2: run_phase<2, W::NF1, W::NF2>(d, sm, l, lp); break;
            case 3: run_phase<3, 0, 0>(d, sm, l, lp); break;
            case 4: run_phase<4, W::NG, 0>(d, sm, l, lp); break;
            case 5: run_phase<5, 16, 0>(d, sm, l, lp); break;
            case 6: run_phase<6, 0, 0>(d, sm, l, lp); break;
            case 7: run_phase<7, W::NG, 0>(d, sm, l, lp); break;
            default: run_phase<8, 0, 0>(d, sm, l, lp); break;
            }
        }
        if (idx + 1 < nphase) grid_barrier(d.bar, target, gridDim.x, mode);
    }
}
#endif

// One kernel per phase (round 3; until then one kernel with a switch over the phase: every phase paid for the registers and
// the SGPR spills of the largest one).  (Reading the descriptor from device memory through a 16-byte kernel argument instead was
// measured 3 % slower in the replayed graph: one more dependent load at the head of every phase.)
template <int PH, int P1, int P2>
__global__ __launch_bounds__(kDecThreads) void decode_phase_kernel(DecodeDesc d, int l, ArgRow<PH> row) {
    __shared__ DecSmem sm;
    run_phase<PH, P1, P2>(d, sm, l, row);
}
// the same phase reading its pointers from the device table (callers that have no host copy of it)
template <int PH, int P1, int P2>
__global__ __launch_bounds__(kDecThreads) void decode_phase_tbl_kernel(DecodeDesc d, int l) {
    __shared__ DecSmem sm;
    run_phase<PH, P1, P2>(d, sm, l, TblRow{d.tbl + (long)(PH < 7 ? l : 0) * DP_COUNT});
}

// host_tbl: the layer table in host memory, or nullptr
template <int PH, int P1 = 0, int P2 = 0>
inline void launch_phase(int items, hipStream_t st, const DecodeDesc &d, int l, const void *const *host_tbl) {
    if (!host_tbl) {
        decode_phase_tbl_kernel<PH, P1, P2><<<dim3(items), dim3(kDecThreads), 0, st>>>(d, l);
        return;
    }
    ArgRow<PH> row;
    row.p[0] = nullptr;
    if (PH < 7)
        for (int dp = 0; dp < DP_COUNT; dp++)
            if (dp_slot(PH, dp) >= 0) row.p[dp_slot(PH, dp)] = host_tbl[(long)l * DP_COUNT + dp];
    decode_phase_kernel<PH, P1, P2><<<dim3(items), dim3(kDecThreads), 0, st>>>(d, l, row);
}

template <int V>
void launch_phases(const int (&g_phase)[7], int items_l0_p1, int B, int L, int V_, hipStream_t st, const DecodeDesc &d,
                   const void *const *ht) {
    using W = Variant<V>;
    for (int l = 0; l < L; l++) {
        launch_phase<0, W::NG>(g_phase[0], st, d, l, ht);
        launch_phase<1>(l == 0 ? items_l0_p1 : g_phase[1], st, d, l, ht);
        launch_phase<2, W::NF1, W::NF2>(g_phase[2], st, d, l, ht);
        launch_phase<3>(g_phase[3], st, d, l, ht);
        launch_phase<4, W::NG>(g_phase[4], st, d, l, ht);
        if (d.F / 32 >= kSplitCap) launch_phase<5, 32>(d.F / 32, st, d, l, ht);
        else launch_phase<5, 16>(g_phase[5], st, d, l, ht);
        launch_phase<6>(g_phase[6], st, d, l, ht);
    }
    launch_phase<7, W::NG>(B, st, d, L, ht);
    launch_phase<8>((V_ + 31) / 32, st, d, L, ht);
}

inline int pick_variant(int D, int Rw, int Ra, int Rv, int Rg) {
    const int ng = (D / 4 + kDecThreads - 1) / kDecThreads, nf1 = max(Rw, Ra) / 16, nf2 = max(Rv, Rg) / 16;
    for (int v = 0; v < 2; v++) {
        const int NG = v == 0 ? Variant<0>::NG : Variant<1>::NG, NF1 = v == 0 ? Variant<0>::NF1 : Variant<1>::NF1,
                  NF2 = v == 0 ? Variant<0>::NF2 : Variant<1>::NF2;
        if (ng <= NG && nf1 <= NF1 && nf2 <= NF2) return v;
    }
    return 2;
}

// K split of a GEMV phase: minimise the work of the busiest workgroup, where an item costs its K range plus a fixed
// latency worth ~256 k (measured: a 16-way split of the r/k/v sweep -- 7 short items per workgroup -- took 13 us, the 2-way
// split 5 us); ties go to the smaller split (fewer partials to sum)
int pick_ks(int ntiles, int K, int grid) {
    int best = 0;
    long best_cost = -1;
    for (int ks = 1; ks <= 16; ks *= 2) {
        if (K % (ks * 4 * 16) != 0) continue;
        const long cost = (long)((ntiles * ks + grid - 1) / grid) * (K / ks + 256);
        if (best_cost < 0 || cost < best_cost) {
            best = ks;
            best_cost = cost;
        }
    }
    return best;
}

inline size_t al(size_t x) { return (x + 255) & ~(size_t)255; }

struct WsLayout {
    size_t xa, xb, vfirst, p_qkv, p_att, kact, p_val, mixed, yg, kx, hfin, bar, total;
    int ks_qkv, ks_o, ks_val;
};

bool ws_layout(int D, int F, int Rw, int Ra, int Rv, int Rg, WsLayout &w) {
    const int N2 = 3 * D + Rw + Ra + Rv + Rg;
    w.ks_qkv = pick_ks(N2 / 32, D, kSplitCap);
    w.ks_o = pick_ks(D / 32, D, kSplitCap);
    w.ks_val = pick_ks(D / 32, F, kSplitCap);
    if (!w.ks_qkv || !w.ks_o || !w.ks_val || D % 64 != 0) return false;
    size_t o = 0;
    auto take = [&](size_t bytes) { const size_t at = o; o += al(bytes); return at; };
    w.bar = take(256);   // first: [0] arrival counter, [1] timeout flag (the host reads byte offset 4)
    w.xa = take((size_t)kRows * D * 4);
    w.xb = take((size_t)kRows * D * 4);
    w.vfirst = take((size_t)kRows * D * 4);
    w.p_qkv = take((size_t)w.ks_qkv * kRows * N2 * 4);
    w.p_att = take((size_t)w.ks_o * kRows * D * 4);
    w.kact = take((size_t)kRows * F * 2);
    w.p_val = take((size_t)w.ks_val * kRows * D * 4);
    w.mixed = take((size_t)6 * kRows * D * 2);
    w.yg = take((size_t)kRows * D * 2);
    w.kx = take((size_t)kRows * D * 2);
    w.hfin = take((size_t)kRows * D * 2);
    w.total = o;
    return true;
}

bool shape_ok(int B, int D, int H, int F, int V, int Rw, int Ra, int Rv, int Rg) {
    auto r_ok = [](int r) { return r >= 32 && r % 32 == 0 && r <= 16 * kUpFrags; };
    return B >= 1 && B <= kRows && D == H * 64 && D % 64 == 0 && D <= kDecThreads * kMaxE && F % 64 == 0 && V >= 1 && r_ok(Rw) &&
           r_ok(Ra) && r_ok(Rv) && r_ok(Rg) && Rw + Ra + Rv + Rg <= kMaxR;
}

}  // namespace

int decode_layer_ptrs() { return DP_COUNT; }

size_t decode_workspace_bytes(int B, int D, int H, int F, int V, int Rw, int Ra, int Rv, int Rg) {
    WsLayout w;
    if (!shape_ok(B, D, H, F, V, Rw, Ra, Rv, Rg) || !ws_layout(D, F, Rw, Ra, Rv, Rg, w)) return 0;
    return w.total;
}

int decode_step_bf16(int B, int D, int H, int L, int F, int V, int Rw, int Ra, int Rv, int Rg, float ln_eps, float gn_eps,
                     const void *const *layer_tbl, const void *const *layer_tbl_host, const void *x_in, const void *norm_w,
                     const void *norm_b, const void *head_w, const void *head_b, float *logits, void *workspace, int persistent,
                     hipStream_t st) {
    WsLayout w;
    if (!shape_ok(B, D, H, F, V, Rw, Ra, Rv, Rg) || L < 1 || !ws_layout(D, F, Rw, Ra, Rv, Rg, w)) return -4;  // RWKV7_ESHAPE
    char *ws = (char *)workspace;
    DecodeDesc d;
    d.B = B; d.D = D; d.H = H; d.L = L; d.F = F; d.V = V;
    d.Rw = Rw; d.Ra = Ra; d.Rv = Rv; d.Rg = Rg;
    d.ks_qkv = w.ks_qkv; d.ks_o = w.ks_o; d.ks_val = w.ks_val;
    d.ln_eps = ln_eps; d.gn_eps = gn_eps;
    d.tbl = layer_tbl;
    d.x_in = (const uint16_t *)x_in;
    d.norm_w = (const uint16_t *)norm_w; d.norm_b = (const uint16_t *)norm_b;
    d.head_w = (const uint16_t *)head_w; d.head_b = (const uint16_t *)head_b;
    d.logits = logits;
    d.xa = (float *)(ws + w.xa); d.xb = (float *)(ws + w.xb); d.vfirst = (float *)(ws + w.vfirst);
    d.p_qkv = (float *)(ws + w.p_qkv); d.p_att = (float *)(ws + w.p_att); d.kact = (uint16_t *)(ws + w.kact);
    d.p_val = (float *)(ws + w.p_val);
    d.mixed = (uint16_t *)(ws + w.mixed); d.yg = (uint16_t *)(ws + w.yg); d.kx = (uint16_t *)(ws + w.kx);
    d.hfin = (uint16_t *)(ws + w.hfin);
    d.bar = (unsigned *)(ws + w.bar);
    (void)hipGetLastError();
    const int variant = pick_variant(D, Rw, Ra, Rv, Rg);
#ifdef RWKV7_LAB
    if (persistent) {
        int dev = 0, cus = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e == hipSuccess) e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        if (e != hipSuccess) return (int)e;
        // every workgroup must be resident at once: one per CU (14 KB of LDS; 4 waves of <= 512 VGPRs fit any CU)
        const int grid = cus < kGrid ? cus : kGrid;
        e = hipMemsetAsync(d.bar, 0, 8, st);
        if (e != hipSuccess) return (int)e;
        if (variant == 0) decode_persistent_kernel<0><<<dim3(grid), dim3(kDecThreads), 0, st>>>(d, persistent);
        else if (variant == 1) decode_persistent_kernel<1><<<dim3(grid), dim3(kDecThreads), 0, st>>>(d, persistent);
        else decode_persistent_kernel<2><<<dim3(grid), dim3(kDecThreads), 0, st>>>(d, persistent);
    } else
#else
    if (persistent) return -4;   // RWKV7_ESHAPE: the persistent variant exists in the lab build only
#endif
    {
        // one launch per phase, each sized to its own item count
        const int N2 = 3 * D + Rw + Ra + Rv + Rg;
        const int g_phase[7] = {B, (N2 / 32) * w.ks_qkv, H * ((B + 1) / 2), (D / 32) * w.ks_o, B, F / 16, (D / 32) * w.ks_val};
        const int p1_l0 = g_phase[1] - (Rv / 32) * w.ks_qkv;   // layer 0 has no value-residual columns
        if (variant == 0) launch_phases<0>(g_phase, p1_l0, B, L, V, st, d, layer_tbl_host);
        else if (variant == 1) launch_phases<1>(g_phase, p1_l0, B, L, V, st, d, layer_tbl_host);
        else launch_phases<2>(g_phase, p1_l0, B, L, V, st, d, layer_tbl_host);
    }
    return (int)hipGetLastError();
}

}  // namespace rwkv7
