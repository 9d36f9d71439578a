// rwkvtts_amd/csrc/decode_step.hip -- one greedy-decode step (T = 1, B <= 32 sequences) of the whole RWKV-7 stack on gfx950
// as ONE persistent kernel (BASELINE.json configs[4]: "persistent-state decode kernel").
//
// Reference: the per-token path RWKV_x070.forward_one (model/llm/rwkv_s2s_single_ffn.py:417-445) with
// RWKV_x070_TMix_one (:482-506) and RWKV_x070_CMix_one (:545-549); batched form forward_batch at T = 1
// (model/llm/rwkv_asr_cuda_whisper.py:438-472).  There a step is ~25 launches per layer; even replayed from a hipGraph every
// launch costs ~5 us of drain/fill, 430 of them make a 4 ms step whose HBM traffic (0.65 GB of weights + 0.4 GB of state)
// would take 0.13 ms.
//
// Here the step is 7 grid-wide phases per layer, executed by 256 resident workgroups (one per CU) that meet at a
// device-scope barrier between phases (agent-scope release/acquire: L2 write-back + invalidate, the XCDs' L2s are not
// coherent with each other):
//   P0 row    x += previous channel-mix output (K-split partials); h = LayerNorm1(x); six token-shift lerps -> bf16 rows;
//             att_x_prev <- h                                                      (rwkv_s2s_single_ffn.py:486-487)
//   P1 gemv   r, k, v projections and the four low-rank down projections in one sweep: [32 x K] . W^T on MFMA, the batch
//             rows are the 32-wide B operand, K split over waves and workgroups -> fp32 partials   (:489-491,497-500)
//   P2 head   per (head, 2 sequences): low-rank up projections (+ tanh / sigmoid), decay, value residual, kk
//             normalisation, the 64x64 fp32 state update in place, y, GroupNorm, bonus, gate -> bf16 rows   (:493-505)
//   P3 gemv   output projection -> partials                                                                  (:506)
//   P4 row    x += attention output; h = LayerNorm2(x); channel-mix lerp; ffn_x_prev <- h                    (:546-547)
//   P5 gemv   key projection -> partials                                                                     (:548)
//   P6 gemv   value projection of relu(.)^2 (applied while the partials are summed on load) -> partials      (:548-549)
// and a final row phase (last residual add + model norm) and the head projection -> fp32 logits.  Every GEMV phase writes
// fp32 K-split partials [KS][32][N] that the consumer sums when it loads them, so no phase waits for a reduction.
// The same phase bodies can be launched as 7 L + 2 separate kernels (persistent = 0): the safe mode, and the oracle for
// the barrier path in tests/test_decode_step_gpu.py.
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
    int ks_qkv, ks_o, ks_key, ks_val;
    float ln_eps, gn_eps;
    const void *const *tbl;   // [L][DP_COUNT] device pointers
    const uint16_t *x_in;     // [B][D] bf16 embeddings of the current tokens
    const uint16_t *norm_w, *norm_b, *head_w, *head_b;
    float *logits;            // [B][V]
    // workspace
    float *xa, *xb, *vfirst, *p_qkv, *p_att, *p_key, *p_val;
    uint16_t *mixed, *yg, *kx, *hfin;
    unsigned *bar;            // [0] arrival counter, [1] timeout flag
};

namespace {

constexpr int kDecThreads = 256;
constexpr int kRows = 32;                  // row capacity of every scratch matrix (the MFMA B operand is 32 wide)
constexpr int kMaxE = 16;                  // D <= 4096: elements per thread in the row phases
constexpr int kMaxR = 512;                 // Rw + Ra + Rv + Rg
constexpr unsigned kSpinLimit = 1u << 21;  // ~0.1 s: a barrier that is not met by then raises the flag instead of hanging the GPU

struct HeadSm {
    float hid[2][kMaxR];
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
__device__ __forceinline__ float softplus_d(float u) { return u > 20.f ? u : log1pf(__expf(u)); }

__device__ __forceinline__ void grid_barrier(unsigned *bar, unsigned &target, unsigned nwg) {
    __syncthreads();
    if (threadIdx.x == 0) {
        target += nwg;
        __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while (__hip_atomic_load(bar, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
            if (__hip_atomic_load(bar + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
            __builtin_amdgcn_s_sleep(2);
            if (++spins > kSpinLimit) {
                __hip_atomic_store(bar + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
    }
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------------------------------
// row phases: residual add, LayerNorm, token-shift lerps.  One workgroup per sequence.
//   x_new = x_old + sum_s parts[s]      (layer 0: x_new = LayerNorm0(x_in))      -> x_out
//   h = bf16(LayerNorm(x_new));   out_j = h + (x_prev - h) * mix_j ;   x_prev <- h        (NMIX = 0: h itself -> out)
// ---------------------------------------------------------------------------------------------------------------------
template <int NMIX>
__device__ __forceinline__ void row_phase(const DecodeDesc &d, int b, float *red, const float *x_old, const float *parts, int nparts,
                          const uint16_t *x_in, const uint16_t *ln0w, const uint16_t *ln0b, float *x_out,
                          const uint16_t *lnw, const uint16_t *lnb, uint16_t *x_prev, const uint16_t *const *mixp,
                          uint16_t *out) {
    const int D = d.D, tid = threadIdx.x;
    const float invD = 1.f / (float)D;
    float x[kMaxE];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < kMaxE; i++) {
        const int c = tid + kDecThreads * i;
        x[i] = 0.f;
        if (c < D) {
            if (x_in) {
                x[i] = bf2f(x_in[(long)b * D + c]);
            } else {
                float a = x_old[(long)b * D + c];
                for (int p = 0; p < nparts; p++) a += parts[((long)p * kRows + b) * D + c];
                x[i] = a;
            }
            s += x[i];
        }
    }
    if (x_in) {  // pre_norm of the first block (rwkv_s2s_single_ffn.py:253-254)
        const float mean = block_sum256(s, red) * invD;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < kMaxE; i++) {
            const int c = tid + kDecThreads * i;
            if (c < D) q += (x[i] - mean) * (x[i] - mean);
        }
        const float rstd = rsqrtf(block_sum256(q, red) * invD + d.ln_eps);
        s = 0.f;
#pragma unroll
        for (int i = 0; i < kMaxE; i++) {
            const int c = tid + kDecThreads * i;
            if (c < D) {
                x[i] = bf2f(f2bf((x[i] - mean) * rstd * bf2f(ln0w[c]) + bf2f(ln0b[c])));
                s += x[i];
            }
        }
    }
    const float mean = block_sum256(s, red) * invD;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < kMaxE; i++) {
        const int c = tid + kDecThreads * i;
        if (c < D) q += (x[i] - mean) * (x[i] - mean);
    }
    const float rstd = rsqrtf(block_sum256(q, red) * invD + d.ln_eps);
#pragma unroll
    for (int i = 0; i < kMaxE; i++) {
        const int c = tid + kDecThreads * i;
        if (c < D) {
            if (x_out) x_out[(long)b * D + c] = x[i];
            const uint16_t hb = f2bf((x[i] - mean) * rstd * bf2f(lnw[c]) + bf2f(lnb[c]));
            if (NMIX == 0) {
                out[(long)b * D + c] = hb;
            } else {
                const float h = bf2f(hb);
                const float xx = bf2f(x_prev[(long)b * D + c]) - h;
#pragma unroll
                for (int j = 0; j < NMIX; j++)
                    out[((long)j * kRows + b) * D + c] = f2bf(fmaf(xx, bf2f(mixp[j][c]), h));
                x_prev[(long)b * D + c] = hb;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// GEMV phases: out[ks][n][col] = sum_{k in split ks} X[n][k] W[col][k]; one item = 32 columns x one K split.
// D[m][n]: m = output column inside the tile (A operand = W rows), n = sequence (B operand = X rows).
// ---------------------------------------------------------------------------------------------------------------------
struct GemvSeg {
    const uint16_t *W;   // [ncols][K]
    const uint16_t *X;   // bf16 [32][K] (XMODE 0)
    int ntiles;          // 32-column tiles (the last one may be partial: ncols)
    int ncols;
};

template <int XMODE>  // 0: X is bf16; 1: X = relu(sum of nxp fp32 partials [nxp][32][K])^2
__device__ __forceinline__ bf16x8 load_x(const uint16_t *xb, const float *xf, int nxp, long xpstride, long off) {
    if constexpr (XMODE == 0) {
        return *reinterpret_cast<const bf16x8 *>(xb + off);
    } else {
        float4 a = *reinterpret_cast<const float4 *>(xf + off), b = *reinterpret_cast<const float4 *>(xf + off + 4);
        for (int p = 1; p < nxp; p++) {
            const float4 a2 = *reinterpret_cast<const float4 *>(xf + p * xpstride + off);
            const float4 b2 = *reinterpret_cast<const float4 *>(xf + p * xpstride + off + 4);
            a.x += a2.x; a.y += a2.y; a.z += a2.z; a.w += a2.w;
            b.x += b2.x; b.y += b2.y; b.z += b2.z; b.w += b2.w;
        }
        auto rs = [](float v) { v = fmaxf(v, 0.f); return v * v; };
        const uint4 o = make_uint4(cvt_pk(rs(a.x), rs(a.y)), cvt_pk(rs(a.z), rs(a.w)), cvt_pk(rs(b.x), rs(b.y)), cvt_pk(rs(b.z), rs(b.w)));
        return __builtin_bit_cast(bf16x8, o);
    }
}

template <int XMODE, int KSTEPS>
__device__ __forceinline__ void gemv_steps(f32x16 &acc, const uint16_t *wp, const uint16_t *xb, const float *xf, int nxp,
                                           long xpstride, long xoff) {
    bf16x8 a[KSTEPS], b[KSTEPS];
#pragma unroll
    for (int i = 0; i < KSTEPS; i++) {
        a[i] = *reinterpret_cast<const bf16x8 *>(wp + 16 * i);
        b[i] = load_x<XMODE>(xb, xf, nxp, xpstride, xoff + 16 * i);
    }
#pragma unroll
    for (int i = 0; i < KSTEPS; i++) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[i], acc, 0, 0, 0);
}

template <int XMODE, int NSEG>
__device__ __forceinline__ void gemv_phase(const DecodeDesc &d, DecSmem &sm, const GemvSeg (&segs)[NSEG], int K, int KS, const float *xf,
                           int nxp, float *out, int ldo, const uint16_t *bias) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
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
        const int c0 = t * 32;                                  // first column of the tile inside its segment
        const int mrow = min(c0 + (lane & 31), sg.ncols - 1);
        const int kbeg = ks * (K / KS) + wave * kw + (lane >> 5) * 8;
        const uint16_t *wp = sg.W + (long)mrow * K + kbeg;
        const long xoff = (long)nrow * K + kbeg;
        f32x16 acc = zero16();
        int k = 0;
        for (; k + 128 <= kw; k += 128) gemv_steps<XMODE, 8>(acc, wp + k, sg.X, xf, nxp, (long)kRows * K, xoff + k);
        for (; k + 32 <= kw; k += 32) gemv_steps<XMODE, 2>(acc, wp + k, sg.X, xf, nxp, (long)kRows * K, xoff + k);
        for (; k + 16 <= kw; k += 16) gemv_steps<XMODE, 1>(acc, wp + k, sg.X, xf, nxp, (long)kRows * K, xoff + k);
        __syncthreads();  // part[] of the previous item has been consumed
        if (wave > 0) {
#pragma unroll
            for (int r = 0; r < 16; r++) sm.part[wave - 1][lane][r] = acc[r];
        }
        __syncthreads();
        if (wave == 0) {
            const int n = lane & 31;
            float *op = out + ((long)ks * kRows + n) * ldo + col_base + c0;
            const bool vec = (ldo & 3) == 0 && ((col_base + c0) & 3) == 0;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int c = 8 * j + 4 * (lane >> 5);  // 4 consecutive columns of the tile
                float v[4];
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    v[i] = acc[4 * j + i] + sm.part[0][lane][4 * j + i] + sm.part[1][lane][4 * j + i] + sm.part[2][lane][4 * j + i];
                    if (bias && c0 + c + i < sg.ncols) v[i] += bf2f(bias[col_base + c0 + c + i]);
                }
                if (n < d.B) {
                    if (vec && c0 + c + 3 < sg.ncols) {
                        *reinterpret_cast<float4 *>(op + c) = make_float4(v[0], v[1], v[2], v[3]);
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
// head phase: everything between the projections and the output projection, for one head and two sequences
// ---------------------------------------------------------------------------------------------------------------------
__device__ void head_phase(const DecodeDesc &d, DecSmem &smu, int l, const void *const *lp) {
    HeadSm &sm = smu.h;
    const int tid = threadIdx.x, D = d.D, H = d.H;
    const int N2 = 3 * D + d.Rw + d.Ra + d.Rv + d.Rg;       // columns of the qkv/low-rank partials
    const int Rtot = d.Rw + d.Ra + d.Rv + d.Rg;
    const int oA = d.Rw, oV = d.Rw + d.Ra, oG = d.Rw + d.Ra + d.Rv;
    const int npair = (d.B + 1) / 2, nitems = H * npair;
    const uint16_t *w2 = (const uint16_t *)lp[DP_W2], *w0 = (const uint16_t *)lp[DP_W0];
    const uint16_t *a2 = (const uint16_t *)lp[DP_A2], *a0 = (const uint16_t *)lp[DP_A0];
    const uint16_t *v2w = (const uint16_t *)lp[DP_V2], *v0 = (const uint16_t *)lp[DP_V0];
    const uint16_t *g2 = (const uint16_t *)lp[DP_G2];
    const uint16_t *k_k = (const uint16_t *)lp[DP_KK], *k_a = (const uint16_t *)lp[DP_KA], *r_k = (const uint16_t *)lp[DP_RK];
    const uint16_t *gnw = (const uint16_t *)lp[DP_GNW], *gnb = (const uint16_t *)lp[DP_GNB];
    float *kv_all = (float *)lp[DP_ATT_KV];
    const bool first = l == 0;
    for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
        const int h = item % H, bp = item / H;
        const int b0 = 2 * bp;
        __syncthreads();  // LDS of the previous item is free
        // A: low-rank hidden vectors (activation applied to the summed partials) and this head's r, k, v
        for (int idx = tid; idx < 2 * Rtot; idx += kDecThreads) {
            const int bb = idx / Rtot, r = idx - bb * Rtot;
            const int b = min(b0 + bb, d.B - 1);
            float s = 0.f;
            for (int p = 0; p < d.ks_qkv; p++) s += d.p_qkv[((long)p * kRows + b) * N2 + 3 * D + r];
            if (r < oA) s = tanh_(s);
            else if (r >= oG) s = sigm(s);
            sm.hid[bb][r] = s;
        }
        for (int idx = tid; idx < 3 * 2 * 64; idx += kDecThreads) {
            const int which = idx / 128, bb = (idx >> 6) & 1, c = idx & 63;
            const int b = min(b0 + bb, d.B - 1);
            float s = 0.f;
            for (int p = 0; p < d.ks_qkv; p++) s += d.p_qkv[((long)p * kRows + b) * N2 + which * D + h * 64 + c];
            sm.rkv[which][bb][c] = s;
        }
        __syncthreads();
        // B: up projections for the 64 channels of the head.  wave = (sequence, half): half 0 -> w and a, half 1 -> v and g
        {
            const int c = tid & 63, bb = (tid >> 6) & 1, half = tid >> 7;
            const int ch = h * 64 + c;
            auto up = [&](const uint16_t *W2, int R, int off) {
                float acc = 0.f;
                const uint16_t *wr = W2 + (long)ch * R;
                for (int r = 0; r < R; r += 8) {
                    const uint4 w8 = *reinterpret_cast<const uint4 *>(wr + r);
                    const float *hp = &sm.hid[bb][off + r];
                    acc = fmaf(__uint_as_float(w8.x << 16), hp[0], acc);
                    acc = fmaf(__uint_as_float(w8.x & 0xffff0000u), hp[1], acc);
                    acc = fmaf(__uint_as_float(w8.y << 16), hp[2], acc);
                    acc = fmaf(__uint_as_float(w8.y & 0xffff0000u), hp[3], acc);
                    acc = fmaf(__uint_as_float(w8.z << 16), hp[4], acc);
                    acc = fmaf(__uint_as_float(w8.z & 0xffff0000u), hp[5], acc);
                    acc = fmaf(__uint_as_float(w8.w << 16), hp[6], acc);
                    acc = fmaf(__uint_as_float(w8.w & 0xffff0000u), hp[7], acc);
                }
                return acc;
            };
            if (half == 0) {
                sm.up[0][bb][c] = up(w2, d.Rw, 0) + bf2f(w0[ch]);
                sm.up[1][bb][c] = up(a2, d.Ra, oA) + bf2f(a0[ch]);
            } else {
                sm.up[2][bb][c] = first ? 0.f : up(v2w, d.Rv, oV) + bf2f(v0[ch]);
                sm.up[3][bb][c] = up(g2, d.Rg, oG);
            }
        }
        __syncthreads();
        // C: decay, gates, value residual, kk normalisation (rwkv_s2s_single_ffn.py:493-500); wave = sequence, lane = channel
        if (tid < 128) {
            const int c = tid & 63, bb = tid >> 6;
            const int b = min(b0 + bb, d.B - 1), ch = h * 64 + c;
            const float r = sm.rkv[0][bb][c], k = sm.rkv[1][bb][c];
            float v = sm.rkv[2][bb][c];
            const float w = -softplus_d(-sm.up[0][bb][c]) - 0.5f;
            const float a = sigm(sm.up[1][bb][c]);
            if (first) {
                if (b0 + bb < d.B) d.vfirst[(long)b * D + ch] = v;
            } else {
                v = fmaf(d.vfirst[(long)b * D + ch] - v, sigm(sm.up[2][bb][c]), v);
            }
            const float kkr = k * bf2f(k_k[ch]);
            const float ss = wave_sum(kkr * kkr);
            const float kk = kkr / fmaxf(sqrtf(ss), 1e-12f);
            const float k2 = k * fmaf(a - 1.f, bf2f(k_a[ch]), 1.f);
            const float dot = wave_sum(r * k2 * bf2f(r_k[ch]));
            sm.vec[0][bb][c] = r;
            sm.vec[1][bb][c] = __expf(-__expf(w));
            sm.vec[2][bb][c] = k2;
            sm.vec[3][bb][c] = v;
            sm.vec[4][bb][c] = -kk;
            sm.vec[5][bb][c] = kk * a;
            if (c == 0) sm.dot[bb] = dot;
        }
        __syncthreads();
        // D: state update in place.  128 threads per sequence; 16 lanes x float4 = one state row (value index), 8 rows per pass
        {
            const int bb = tid >> 7, tt = tid & 127;
            const int k4 = (tt & 15) * 4, vr = tt >> 4;
            if (b0 + bb < d.B) {
                float *S = kv_all + ((long)(b0 + bb) * H + h) * 64 * 64;
                float4 st[8];
#pragma unroll
                for (int i = 0; i < 8; i++) st[i] = *reinterpret_cast<const float4 *>(S + (vr + 8 * i) * 64 + k4);
                const float4 rr = *reinterpret_cast<const float4 *>(&sm.vec[0][bb][k4]);
                const float4 dc = *reinterpret_cast<const float4 *>(&sm.vec[1][bb][k4]);
                const float4 kk = *reinterpret_cast<const float4 *>(&sm.vec[2][bb][k4]);
                const float4 aa = *reinterpret_cast<const float4 *>(&sm.vec[4][bb][k4]);
                const float4 bv = *reinterpret_cast<const float4 *>(&sm.vec[5][bb][k4]);
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const float vv = sm.vec[3][bb][vr + 8 * i];
                    const float sa = sum16(st[i].x * aa.x + st[i].y * aa.y + st[i].z * aa.z + st[i].w * aa.w);
                    st[i].x = fmaf(st[i].x, dc.x, fmaf(sa, bv.x, vv * kk.x));
                    st[i].y = fmaf(st[i].y, dc.y, fmaf(sa, bv.y, vv * kk.y));
                    st[i].z = fmaf(st[i].z, dc.z, fmaf(sa, bv.z, vv * kk.z));
                    st[i].w = fmaf(st[i].w, dc.w, fmaf(sa, bv.w, vv * kk.w));
                    *reinterpret_cast<float4 *>(S + (vr + 8 * i) * 64 + k4) = st[i];
                    const float y = sum16(st[i].x * rr.x + st[i].y * rr.y + st[i].z * rr.z + st[i].w * rr.w);
                    if ((tt & 15) == 0) sm.y[bb][vr + 8 * i] = y;
                }
            }
        }
        __syncthreads();
        // E: GroupNorm over the head, bonus, gate (rwkv_s2s_single_ffn.py:504-505)
        if (tid < 128) {
            const int c = tid & 63, bb = tid >> 6;
            const int ch = h * 64 + c;
            const float y = sm.y[bb][c];
            const float mean = wave_sum(y) * (1.f / 64.f);
            const float dv = y - mean;
            const float rstd = rsqrtf(wave_sum(dv * dv) * (1.f / 64.f) + d.gn_eps);
            const float o = (fmaf(dv * rstd, bf2f(gnw[ch]), bf2f(gnb[ch])) + sm.dot[bb] * sm.vec[3][bb][c]) * sm.up[3][bb][c];
            if (b0 + bb < d.B) d.yg[(long)(b0 + bb) * D + ch] = f2bf(o);
        }
    }
}

__device__ void run_phase(const DecodeDesc &d, DecSmem &sm, int l, int ph) {
    const void *const *lp = d.tbl + (long)(l < d.L ? l : 0) * DP_COUNT;
    const int D = d.D;
    if (l == d.L) {  // tail: last residual add + model norm, then the head
        if (ph == 0) {
            for (int b = blockIdx.x; b < d.B; b += gridDim.x)
                row_phase<0>(d, b, sm.red, d.xa, d.p_val, d.ks_val, nullptr, nullptr, nullptr, nullptr, d.norm_w, d.norm_b, nullptr,
                             nullptr, d.hfin);
        } else {
            const GemvSeg seg[1] = {{d.head_w, d.hfin, (d.V + 31) / 32, d.V}};
            gemv_phase<0, 1>(d, sm, seg, D, 1, nullptr, 0, d.logits, d.V, d.head_b);
        }
        return;
    }
    switch (ph) {
    case 0: {
        const uint16_t *mixp[6] = {(const uint16_t *)lp[DP_XR], (const uint16_t *)lp[DP_XW], (const uint16_t *)lp[DP_XK],
                                   (const uint16_t *)lp[DP_XV], (const uint16_t *)lp[DP_XA], (const uint16_t *)lp[DP_XG]};
        for (int b = blockIdx.x; b < d.B; b += gridDim.x)
            row_phase<6>(d, b, sm.red, d.xa, d.p_val, d.ks_val, l == 0 ? d.x_in : nullptr, (const uint16_t *)lp[DP_LN0_W],
                         (const uint16_t *)lp[DP_LN0_B], d.xb, (const uint16_t *)lp[DP_LN1_W], (const uint16_t *)lp[DP_LN1_B],
                         (uint16_t *)lp[DP_ATT_XPREV], mixp, d.mixed);
        break;
    }
    case 1: {
        const long RS = (long)kRows * D;  // one mixed plane: order r, w, k, v, a, g
        // layer 0 has no value-residual branch: its columns stay unwritten and unread
        const GemvSeg segs[7] = {{(const uint16_t *)lp[DP_WR], d.mixed + 0 * RS, D / 32, D},
                                 {(const uint16_t *)lp[DP_WK], d.mixed + 2 * RS, D / 32, D},
                                 {(const uint16_t *)lp[DP_WV], d.mixed + 3 * RS, D / 32, D},
                                 {(const uint16_t *)lp[DP_W1], d.mixed + 1 * RS, d.Rw / 32, d.Rw},
                                 {(const uint16_t *)lp[DP_A1], d.mixed + 4 * RS, d.Ra / 32, d.Ra},
                                 {(const uint16_t *)(l == 0 ? lp[DP_A1] : lp[DP_V1]), d.mixed + 3 * RS, l == 0 ? 0 : d.Rv / 32, d.Rv},
                                 {(const uint16_t *)lp[DP_G1], d.mixed + 5 * RS, d.Rg / 32, d.Rg}};
        gemv_phase<0, 7>(d, sm, segs, D, d.ks_qkv, nullptr, 0, d.p_qkv, 3 * D + d.Rw + d.Ra + d.Rv + d.Rg, nullptr);
        break;
    }
    case 2:
        head_phase(d, sm, l, lp);
        break;
    case 3: {
        const GemvSeg seg[1] = {{(const uint16_t *)lp[DP_WO], d.yg, D / 32, D}};
        gemv_phase<0, 1>(d, sm, seg, D, d.ks_o, nullptr, 0, d.p_att, D, nullptr);
        break;
    }
    case 4: {
        const uint16_t *mixp[1] = {(const uint16_t *)lp[DP_FXK]};
        for (int b = blockIdx.x; b < d.B; b += gridDim.x)
            row_phase<1>(d, b, sm.red, d.xb, d.p_att, d.ks_o, nullptr, nullptr, nullptr, d.xa, (const uint16_t *)lp[DP_LN2_W],
                         (const uint16_t *)lp[DP_LN2_B], (uint16_t *)lp[DP_FFN_XPREV], mixp, d.kx);
        break;
    }
    case 5: {
        const GemvSeg seg[1] = {{(const uint16_t *)lp[DP_WKEY], d.kx, d.F / 32, d.F}};
        gemv_phase<0, 1>(d, sm, seg, D, d.ks_key, nullptr, 0, d.p_key, d.F, nullptr);
        break;
    }
    default: {
        const GemvSeg seg[1] = {{(const uint16_t *)lp[DP_WVAL], nullptr, D / 32, D}};
        gemv_phase<1, 1>(d, sm, seg, d.F, d.ks_val, d.p_key, d.ks_key, d.p_val, D, nullptr);
        break;
    }
    }
}

__global__ __launch_bounds__(kDecThreads) void decode_persistent_kernel(DecodeDesc d) {
    __shared__ DecSmem sm;
    unsigned target = 0;
    for (int l = 0; l < d.L; l++) {
        for (int ph = 0; ph < 7; ph++) {
            run_phase(d, sm, l, ph);
            grid_barrier(d.bar, target, gridDim.x);
        }
    }
    run_phase(d, sm, d.L, 0);
    grid_barrier(d.bar, target, gridDim.x);
    run_phase(d, sm, d.L, 1);
}

__global__ __launch_bounds__(kDecThreads) void decode_phase_kernel(DecodeDesc d, int l, int ph) {
    __shared__ DecSmem sm;
    run_phase(d, sm, l, ph);
}

// largest per-workgroup K work is minimised; ties go to the smaller split (fewer partials to sum)
int pick_ks(int ntiles, int K, int grid) {
    int best = 1;
    long best_cost = -1;
    for (int ks = 1; ks <= 16; ks *= 2) {
        if (K % (ks * 4 * 16) != 0) continue;
        const long cost = (long)((ntiles * ks + grid - 1) / grid) * (K / ks);
        if (best_cost < 0 || cost < best_cost) {
            best = ks;
            best_cost = cost;
        }
    }
    return best_cost < 0 ? 0 : best;
}

constexpr int kGrid = 256;
inline size_t al(size_t x) { return (x + 255) & ~(size_t)255; }

struct WsLayout {
    size_t xa, xb, vfirst, p_qkv, p_att, p_key, p_val, mixed, yg, kx, hfin, bar, total;
    int ks_qkv, ks_o, ks_key, ks_val;
};

bool ws_layout(int D, int F, int Rw, int Ra, int Rv, int Rg, WsLayout &w) {
    const int N2 = 3 * D + Rw + Ra + Rv + Rg;
    w.ks_qkv = pick_ks(N2 / 32, D, kGrid);
    w.ks_o = pick_ks(D / 32, D, kGrid);
    w.ks_key = pick_ks(F / 32, D, kGrid);
    w.ks_val = pick_ks(D / 32, F, kGrid);
    if (!w.ks_qkv || !w.ks_o || !w.ks_key || !w.ks_val) return false;
    size_t o = 0;
    auto take = [&](size_t bytes) { const size_t at = o; o += al(bytes); return at; };
    w.bar = take(256);   // first: [0] arrival counter, [1] timeout flag (the host reads byte offset 4)
    w.xa = take((size_t)kRows * D * 4);
    w.xb = take((size_t)kRows * D * 4);
    w.vfirst = take((size_t)kRows * D * 4);
    w.p_qkv = take((size_t)w.ks_qkv * kRows * N2 * 4);
    w.p_att = take((size_t)w.ks_o * kRows * D * 4);
    w.p_key = take((size_t)w.ks_key * kRows * F * 4);
    w.p_val = take((size_t)w.ks_val * kRows * D * 4);
    w.mixed = take((size_t)6 * kRows * D * 2);
    w.yg = take((size_t)kRows * D * 2);
    w.kx = take((size_t)kRows * D * 2);
    w.hfin = take((size_t)kRows * D * 2);
    w.total = o;
    return true;
}

bool shape_ok(int B, int D, int H, int F, int V, int Rw, int Ra, int Rv, int Rg) {
    auto r_ok = [](int r) { return r >= 32 && r % 32 == 0; };
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
                     const void *const *layer_tbl, const void *x_in, const void *norm_w, const void *norm_b, const void *head_w,
                     const void *head_b, float *logits, void *workspace, int persistent, hipStream_t st) {
    WsLayout w;
    if (!shape_ok(B, D, H, F, V, Rw, Ra, Rv, Rg) || L < 1 || !ws_layout(D, F, Rw, Ra, Rv, Rg, w)) return -4;  // RWKV7_ESHAPE
    char *ws = (char *)workspace;
    DecodeDesc d;
    d.B = B; d.D = D; d.H = H; d.L = L; d.F = F; d.V = V;
    d.Rw = Rw; d.Ra = Ra; d.Rv = Rv; d.Rg = Rg;
    d.ks_qkv = w.ks_qkv; d.ks_o = w.ks_o; d.ks_key = w.ks_key; d.ks_val = w.ks_val;
    d.ln_eps = ln_eps; d.gn_eps = gn_eps;
    d.tbl = layer_tbl;
    d.x_in = (const uint16_t *)x_in;
    d.norm_w = (const uint16_t *)norm_w; d.norm_b = (const uint16_t *)norm_b;
    d.head_w = (const uint16_t *)head_w; d.head_b = (const uint16_t *)head_b;
    d.logits = logits;
    d.xa = (float *)(ws + w.xa); d.xb = (float *)(ws + w.xb); d.vfirst = (float *)(ws + w.vfirst);
    d.p_qkv = (float *)(ws + w.p_qkv); d.p_att = (float *)(ws + w.p_att); d.p_key = (float *)(ws + w.p_key);
    d.p_val = (float *)(ws + w.p_val);
    d.mixed = (uint16_t *)(ws + w.mixed); d.yg = (uint16_t *)(ws + w.yg); d.kx = (uint16_t *)(ws + w.kx);
    d.hfin = (uint16_t *)(ws + w.hfin);
    d.bar = (unsigned *)(ws + w.bar);
    (void)hipGetLastError();
    if (persistent) {
        int dev = 0, cus = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e == hipSuccess) e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        if (e != hipSuccess) return (int)e;
        // every workgroup must be resident at once: one per CU (14 KB of LDS and < 128 VGPRs leave room for it anywhere)
        const int grid = cus < kGrid ? cus : kGrid;
        e = hipMemsetAsync(d.bar, 0, 8, st);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(decode_persistent_kernel, dim3(grid), dim3(kDecThreads), 0, st, d);
    } else {
        for (int l = 0; l <= L; l++)
            for (int ph = 0; ph < (l == L ? 2 : 7); ph++)
                hipLaunchKernelGGL(decode_phase_kernel, dim3(kGrid), dim3(kDecThreads), 0, st, d, l, ph);
    }
    return (int)hipGetLastError();
}

}  // namespace rwkv7
