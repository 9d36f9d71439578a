// rwkvtts_amd/csrc/wkv7_chunk_bwd.hip -- chunked (MFMA) WKV7 backward for gfx950, bf16 tensors.
//
// Same gradients as wkv7_bwd.hip (reference wkv7_cuda.cu:54-130), evaluated 32 steps at a time on the matrix cores.
// With H = S^T in R^{K x V} and the per-chunk quantities of chunk_common.h (q~, a~, k^, b^, g_C, T = (I - A_ab)^-1,
// W = T A~) the forward state obeys  H_{c+1} = M_c H_c + N_c  with  M_c = diag(g_C)(I + B^^T W), and the adjoint state
//     E_c = M_c^T E_{c+1} + N'_c ,   N'_c = Q~^T dY + W^T (A_qb^T dY)              (E_c = dL/dH at the START of chunk c)
// is the only sequential object of the backward pass (tools/chunked_proto2.py validates the algebra against the oracle).
// Three kernels:
//   wkv7c_bwd_pre_kernel   grid B*H*(T/32), parallel: M_c^T (bf16 hi/lo planes) and N'_c (fp32, MFMA tile layout)
//   wkv7c_state_kernel     grid B*H, sequential over chunks in reverse: E for every chunk (one 64x64x64 product each)
//   wkv7c_bwd_out_kernel   grid B*H*(T/32), parallel: dw,dq,dk,dv,da,db of a chunk from (H_c, E_{c+1}, U = sa, dY)
// Inputs that come from the forward pass: sa (= U) and the fp32 state checkpoints of wkv7_fwd.hip (every 16 steps,
// s[b,h,n][k][v]; the state at the start of chunk c is checkpoint 2c-1), and T from wkv7c_prep_kernel.
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
template <int K>
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
#pragma unroll
    for (int i = 0; i < NK; i++) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh[i], y[i], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xl[i], y[i], acc, 0, 0, 0);
    }
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
// the same 8 values into a channel-major plane pair: element j goes to row (pk + j), column pt
__device__ __forceinline__ void put_col8(uint16_t *Ph, uint16_t *Pl, int ld, int pk, int pt, const uint32_t (&hi)[4],
                                         const uint32_t (&lo)[4]) {
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int o = (pk + j) * ld + pt, sh = (j & 1) * 16;
        Ph[o] = (uint16_t)(hi[j >> 1] >> sh);
        Pl[o] = (uint16_t)(lo[j >> 1] >> sh);
    }
}

// inclusive cumulative sum of lw over the 32 steps of the chunk, per channel.  Thread (pt, pk) owns 8 channels of one
// step; returns G[j] = sum_{s <= pt} lw_s[pk + j].  sh_G [32][64] and sh_seg [4][64] are fp32 scratch.
__device__ __forceinline__ void chunk_cumsum(const float (&lw)[8], float (&G)[8], float *sh_G, float *sh_seg, int tid, int pt,
                                             int pk) {
    *reinterpret_cast<float4 *>(&sh_G[pt * kN + pk]) = make_float4(lw[0], lw[1], lw[2], lw[3]);
    *reinterpret_cast<float4 *>(&sh_G[pt * kN + pk + 4]) = make_float4(lw[4], lw[5], lw[6], lw[7]);
    lds_barrier();
    {
        const int ch = tid & 63, seg = tid >> 6;
        float run = 0.f;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            run += sh_G[(seg * 8 + i) * kN + ch];
            sh_G[(seg * 8 + i) * kN + ch] = run;
        }
        sh_seg[seg * kN + ch] = run;
    }
    lds_barrier();
    const float4 g0 = *reinterpret_cast<const float4 *>(&sh_G[pt * kN + pk]);
    const float4 g1 = *reinterpret_cast<const float4 *>(&sh_G[pt * kN + pk + 4]);
    G[0] = g0.x; G[1] = g0.y; G[2] = g0.z; G[3] = g0.w; G[4] = g1.x; G[5] = g1.y; G[6] = g1.z; G[7] = g1.w;
    for (int s = 0; s < (pt >> 3); s++) {
#pragma unroll
        for (int j = 0; j < 8; j++) G[j] += sh_seg[s * kN + pk + j];
    }
}

// T (or T^T) of the chunk, fp32 [32][32] in global memory -> bf16 hi/lo planes [32][LDC]
template <bool TRANSPOSE>
__device__ __forceinline__ void load_tm(const float *tp, uint16_t *Th, uint16_t *Tl, int tid) {
    const int tr = tid >> 3, tc = (tid & 7) * 4;
    const float4 x = *reinterpret_cast<const float4 *>(tp + tr * kC + tc);
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

struct PreSmem {  // offsets in uint16 units
    // time-major planes, dead after phase 1; G1T (phase 2+) is laid over them
    static constexpr int QTh = 0, QTl = QTh + kC * LDK, BHh = QTl + kC * LDK, BHl = BHh + kC * LDK;
    static constexpr int G1Th = QTh, G1Tl = G1Th + kN * LDC;  // 2 * 2560 <= 4 * 2304
    static constexpr int ATTh = BHl + kC * LDK, ATTl = ATTh + kN * LDC, QTTh = ATTl + kN * LDC, QTTl = QTTh + kN * LDC;
    static constexpr int BCTh = QTTl + kN * LDC, BCTl = BCTh + kN * LDC, DYT = BCTl + kN * LDC;
    static constexpr int TMh = DYT + kN * LDC, TMl = TMh + kC * LDC, QBTh = TMl + kC * LDC, QBTl = QBTh + kC * LDC;
    // W^T planes; the fp32 cumsum scratch of the prologue (sh_G 2048 + sh_seg 256 floats = 4608 u16) lies over them
    static constexpr int WTh = QBTl + kC * LDC, WTl = WTh + kN * LDC;
    static constexpr int scratch = WTh;
    static constexpr int gC = WTl + kN * LDC;  // 64 floats
    static constexpr int end16 = gC + 2 * kN;
    static constexpr size_t bytes = (size_t)end16 * 2;
};
static_assert(2 * kN * LDC >= (kC * kN + 4 * kN) * 2, "cumsum scratch must fit under the W^T planes");
static_assert(PreSmem::QTh % 8 == 0 && PreSmem::ATTh % 8 == 0 && PreSmem::WTh % 8 == 0 && PreSmem::gC % 8 == 0, "16-byte alignment");

}  // namespace

// ------------------------------------------------------------------------------------------------------------------
// pre: M_c^T planes and N'_c
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void wkv7c_bwd_pre_kernel(int T_, int H, const bf16_t *__restrict__ w_,
                                                            const bf16_t *__restrict__ q_, const bf16_t *__restrict__ a_,
                                                            const bf16_t *__restrict__ b_, const bf16_t *__restrict__ dy_,
                                                            const float *__restrict__ tinv_, uint16_t *__restrict__ mt_,
                                                            float *__restrict__ np_) {
    extern __shared__ __attribute__((aligned(16))) uint16_t sm[];
    using L = PreSmem;
    float *sh_G = reinterpret_cast<float *>(sm + L::scratch), *sh_seg = sh_G + kC * kN;
    float *sh_gC = reinterpret_cast<float *>(sm + L::gC);
    const int nc = T_ / kC;
    const int bh = blockIdx.x / nc, c = blockIdx.x - bh * nc;
    const int bb = bh / H, hh = bh - bb * H;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int pt = tid & 31, pk = (tid >> 5) * 8;
    const long tstride = (long)H * kN;
    const long off = ((long)bb * T_ * H + hh) * kN + (long)(c * kC + pt) * tstride + pk;

    // ---- prologue ---------------------------------------------------------------------------------------------------
    const Raw8 rw = ld8(w_ + off), rq = ld8(q_ + off), ra = ld8(a_ + off), rb = ld8(b_ + off), rdy = ld8(dy_ + off);
    load_tm<false>(tinv_ + (long)blockIdx.x * kC * kC, sm + L::TMh, sm + L::TMl, tid);
    float lw[8], G[8];
    cvt8(rw, lw);
#pragma unroll
    for (int j = 0; j < 8; j++) lw[j] = -fast_exp(lw[j]);
    chunk_cumsum(lw, G, sh_G, sh_seg, tid, pt, pk);
    if (pt == kC - 1) {
#pragma unroll
        for (int j = 0; j < 8; j++) sh_gC[pk + j] = fast_exp(G[j]);
    }
    lds_barrier();  // sh_gC complete; cumsum scratch free (W^T planes are written in phase 1)
    {
        float qv[8], av[8], bv[8], x[8];
        cvt8(rq, qv); cvt8(ra, av); cvt8(rb, bv);
        uint32_t hi[4], lo[4];
#pragma unroll
        for (int j = 0; j < 8; j++) x[j] = qv[j] * fast_exp(G[j]);                 // q~ = q gamma_t
        put_row8(sm + L::QTh, sm + L::QTl, pt * LDK + pk, x, hi, lo);
        put_col8(sm + L::QTTh, sm + L::QTTl, LDC, pk, pt, hi, lo);
#pragma unroll
        for (int j = 0; j < 8; j++) x[j] = bv[j] * fast_exp(-G[j]);                // b^ = b / gamma_t
        put_row8(sm + L::BHh, sm + L::BHl, pt * LDK + pk, x, hi, lo);
#pragma unroll
        for (int j = 0; j < 8; j++) x[j] *= sh_gC[pk + j];                          // b^ g_C (bounded by |b|)
#pragma unroll
        for (int j = 0; j < 4; j++) split_pk(x[2 * j], x[2 * j + 1], hi[j], lo[j]);
        put_col8(sm + L::BCTh, sm + L::BCTl, LDC, pk, pt, hi, lo);
#pragma unroll
        for (int j = 0; j < 8; j++) x[j] = av[j] * fast_exp(G[j] - lw[j]);         // a~ = a gamma_{t-1}
#pragma unroll
        for (int j = 0; j < 4; j++) split_pk(x[2 * j], x[2 * j + 1], hi[j], lo[j]);
        put_col8(sm + L::ATTh, sm + L::ATTl, LDC, pk, pt, hi, lo);
        const uint32_t dyr[4] = {rdy.r.x, rdy.r.y, rdy.r.z, rdy.r.w};            // dY is bf16: exact, one plane
#pragma unroll
        for (int j = 0; j < 8; j++) sm[L::DYT + (pk + j) * LDC + pt] = (uint16_t)(dyr[j >> 1] >> ((j & 1) * 16));
    }
    lds_barrier();
    // ---- phase 1: A_qb^T (wave 0), W = T A~ (waves 1, 2) ---------------------------------------------------------------
    if (wave == 0) {
        f32x16 acc = zero16();  // D[m = t][n = s] = q~_t . b^_s, kept for t >= s; stored as QBT[s][t]
        mma_tile3<kN>(acc, sm + L::QTh, sm + L::QTl, LDK, sm + L::BHh, sm + L::BHl, LDK, lane);
        mask_upper_T<false>(acc, lane);
        store_T_split(acc, sm + L::QBTh, sm + L::QBTl, LDC, lane);
    } else if (wave <= 2) {
        const int kt = wave - 1;
        f32x16 acc = zero16();  // D[m = t][n = k] = sum_s T[t][s] a~[s][k]; stored as WT[k][t]
        mma_tile3<kC>(acc, sm + L::TMh, sm + L::TMl, LDC, sm + L::ATTh + kt * 32 * LDC, sm + L::ATTl + kt * 32 * LDC, LDC, lane);
        store_T_split(acc, sm + L::WTh + kt * 32 * LDC, sm + L::WTl + kt * 32 * LDC, LDC, lane);
    }
    lds_barrier();
    // ---- phase 2: G1 = A_qb^T dY (waves 0, 1); M^T = diag(g_C) + W^T (B^ g_C) (waves 2, 3) ---------------------------
    if (wave <= 1) {
        const int vt = wave;
        f32x16 acc = zero16();  // D[m = s][n = v] = sum_t QBT[s][t] dY[t][v]; stored as G1T[v][s]
        mma_xs_ye<kC>(acc, sm + L::QBTh, sm + L::QBTl, LDC, sm + L::DYT + vt * 32 * LDC, LDC, lane);
        store_T_split(acc, sm + L::G1Th + vt * 32 * LDC, sm + L::G1Tl + vt * 32 * LDC, LDC, lane);
    } else {
        const int mt = wave - 2;  // rows k of M^T
        uint16_t *outh = mt_ + (long)blockIdx.x * 2 * kN * kN, *outl = outh + kN * kN;
#pragma unroll
        for (int nt = 0; nt < 2; nt++) {
            f32x16 acc = zero16();  // D[m = k][n = k'] = sum_t W[t][k] (b^ g_C)[t][k'] = (M^T - diag)[k][k']
            mma_tile3<kC>(acc, sm + L::WTh + mt * 32 * LDC, sm + L::WTl + mt * 32 * LDC, LDC, sm + L::BCTh + nt * 32 * LDC,
                          sm + L::BCTl + nt * 32 * LDC, LDC, lane);
            const int n = lane & 31;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int m = d_row(r, lane);
                float x = acc[r];
                if (mt == nt && m == n) x += sh_gC[mt * 32 + m];
                uint16_t xh, xl;
                split2(x, xh, xl);
                const int o = (mt * 32 + m) * kN + nt * 32 + n;
                outh[o] = xh;
                outl[o] = xl;
            }
        }
    }
    lds_barrier();
    // ---- phase 3: N' = Q~^T dY + W^T G1  (one 32x32 tile per wave, stored in MFMA register layout) ----------------------
    {
        const int mt = wave >> 1, nt = wave & 1;
        f32x16 acc = zero16();  // D[m = k][n = v]
        mma_xs_ye<kC>(acc, sm + L::QTTh + mt * 32 * LDC, sm + L::QTTl + mt * 32 * LDC, LDC, sm + L::DYT + nt * 32 * LDC, LDC, lane);
        mma_tile3<kC>(acc, sm + L::WTh + mt * 32 * LDC, sm + L::WTl + mt * 32 * LDC, LDC, sm + L::G1Th + nt * 32 * LDC,
                      sm + L::G1Tl + nt * 32 * LDC, LDC, lane);
        float *o = np_ + (((long)blockIdx.x * 4 + wave) * 64 + lane) * 16;
#pragma unroll
        for (int j = 0; j < 4; j++)
            *reinterpret_cast<float4 *>(o + 4 * j) = make_float4(acc[4 * j], acc[4 * j + 1], acc[4 * j + 2], acc[4 * j + 3]);
    }
}

// ------------------------------------------------------------------------------------------------------------------
// state: E_c = M_c^T E_{c+1} + N'_c, c = nc-1 .. 0; writes E_{c+1} (the adjoint state chunk c sees at its end) for every c
// in both orientations: e_vk[b,h,c][v][k] and e_kv[b,h,c][k][v]
// ------------------------------------------------------------------------------------------------------------------
namespace {
struct StateSmem {
    static constexpr int M0h = 0, M0l = M0h + kN * LDK, M1h = M0l + kN * LDK, M1l = M1h + kN * LDK;
    static constexpr int E0h = M1l + kN * LDK, E0l = E0h + kN * LDK, E1h = E0l + kN * LDK, E1l = E1h + kN * LDK;
    static constexpr int end16 = E1l + kN * LDK;
    static constexpr size_t bytes = (size_t)end16 * 2;
};
}  // namespace

__global__ __launch_bounds__(256) void wkv7c_state_kernel(int nc, const uint16_t *__restrict__ mt_, const float *__restrict__ np_,
                                                          float *__restrict__ e_vk, float *__restrict__ e_kv) {
    extern __shared__ __attribute__((aligned(16))) uint16_t sm[];
    using L = StateSmem;
    const int bh = blockIdx.x;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int mt = wave >> 1, nt = wave & 1;

    uint4 rm[4];
    float4 rn[4];
    auto issue = [&](int c) {
        const uint16_t *mp = mt_ + ((long)bh * nc + c) * 2 * kN * kN;
#pragma unroll
        for (int i = 0; i < 4; i++) rm[i] = *reinterpret_cast<const uint4 *>(mp + (tid + 256 * i) * 8);  // 1024 pieces of 8 u16
        const float *np = np_ + ((((long)bh * nc + c) * 4 + wave) * 64 + lane) * 16;
#pragma unroll
        for (int j = 0; j < 4; j++) rn[j] = *reinterpret_cast<const float4 *>(np + 4 * j);
    };
    auto commit = [&](int buf) {
        uint16_t *base = sm + (buf ? L::M1h : L::M0h);
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int p = tid + 256 * i;            // piece p: plane p >> 9, row (p >> 3) & 63, column 8 (p & 7)
            const int plane = p >> 9, row = (p >> 3) & 63, col = (p & 7) * 8;
            *reinterpret_cast<uint4 *>(base + plane * kN * LDK + row * LDK + col) = rm[i];
        }
    };
    for (int i = tid; i < 2 * kN * LDK; i += 256) sm[L::E0h + i] = 0;  // E_{nc} = 0
    issue(nc - 1);
    commit((nc - 1) & 1);
    f32x16 E = zero16();  // this wave's tile of the current E, D layout [m = k][n = v]
    float4 rn_cur[4];
#pragma unroll
    for (int j = 0; j < 4; j++) rn_cur[j] = rn[j];
    lds_barrier();
    int cur = 0;
    for (int c = nc - 1; c >= 0; c--) {
        if (c > 0) issue(c - 1);
        {
            // E_{c+1}: what chunk c receives from the future
            float *pv = e_vk + (((long)bh * nc + c) * kN + nt * 32 + (lane & 31)) * kN + mt * 32 + 4 * (lane >> 5);
#pragma unroll
            for (int j = 0; j < 4; j++)
                *reinterpret_cast<float4 *>(pv + 8 * j) = make_float4(E[4 * j], E[4 * j + 1], E[4 * j + 2], E[4 * j + 3]);
            float *pk = e_kv + ((long)bh * nc + c) * kN * kN + nt * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; r++) pk[(long)(mt * 32 + d_row(r, lane)) * kN] = E[r];
        }
        const uint16_t *Mh = sm + ((c & 1) ? L::M1h : L::M0h), *Ml = Mh + kN * LDK;
        const uint16_t *Eh = sm + (cur ? L::E1h : L::E0h), *El = Eh + kN * LDK;
        f32x16 acc = zero16();  // D[m = k][n = v] = sum_k' M^T[k][k'] E[k'][v]
        mma_tile3<kN>(acc, Mh + mt * 32 * LDK, Ml + mt * 32 * LDK, LDK, Eh + nt * 32 * LDK, El + nt * 32 * LDK, LDK, lane);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            acc[4 * j] += rn_cur[j].x; acc[4 * j + 1] += rn_cur[j].y; acc[4 * j + 2] += rn_cur[j].z; acc[4 * j + 3] += rn_cur[j].w;
        }
        E = acc;
        uint16_t *Oh = sm + (cur ? L::E0h : L::E1h), *Ol = Oh + kN * LDK;
        store_T_split(E, Oh + nt * 32 * LDK + mt * 32, Ol + nt * 32 * LDK + mt * 32, LDK, lane);  // planes [v][k]
        if (c > 0) {
            commit((c - 1) & 1);
#pragma unroll
            for (int j = 0; j < 4; j++) rn_cur[j] = rn[j];
        }
        lds_barrier();
        cur ^= 1;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------------------------
int chunk_bwd_pre_bf16(int B, int T_, int H, const void *w, const void *q, const void *a, const void *b, const void *dy,
                       const float *tinv, void *mt, float *np, hipStream_t st) {
    static bool attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&wkv7c_bwd_pre_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)PreSmem::bytes);
        if (e != hipSuccess) return (int)e;
        attr = true;
    }
    (void)hipGetLastError();
    hipLaunchKernelGGL(wkv7c_bwd_pre_kernel, dim3(B * H * (T_ / kC)), dim3(256), PreSmem::bytes, st, T_, H, (const bf16_t *)w,
                       (const bf16_t *)q, (const bf16_t *)a, (const bf16_t *)b, (const bf16_t *)dy, tinv, (uint16_t *)mt, np);
    return (int)hipGetLastError();
}

int chunk_state_bf16(int BH, int nc, const void *mt, const float *np, float *e_vk, float *e_kv, hipStream_t st) {
    static bool attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&wkv7c_state_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)StateSmem::bytes);
        if (e != hipSuccess) return (int)e;
        attr = true;
    }
    (void)hipGetLastError();
    hipLaunchKernelGGL(wkv7c_state_kernel, dim3(BH), dim3(256), StateSmem::bytes, st, nc, (const uint16_t *)mt, np, e_vk, e_kv);
    return (int)hipGetLastError();
}

}  // namespace rwkv7
