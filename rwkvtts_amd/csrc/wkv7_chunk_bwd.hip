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
// out: all six gradients of one chunk from (H_c, E_{c+1}, U, dY) -- tools/chunked_proto2.py:bwd3, third loop
//   G1 = A_qb^T dY + B^ (g_C E)          Z  = T^T G1                                  (Z_t = dL/du_t)
//   dV = A_qk^T dY + A_ak^T Z + K^ (g_C E)
//   dK = (P_vy Q~ + P_vz A~ + V (g_C E)^T) / gamma      dB = (P_uy Q~ + P_uz A~ + U (g_C E)^T) / gamma
//   dQ = (dY H0^T + P_vy^T K^ + P_uy^T B^) gamma        dA = (Z H0^T + P_vz^T K^ + P_uz^T B^) gamma_prev
//   P_vy = triu(V dY^T)  P_vz = triu(V Z^T, 1)  P_uy = triu(U dY^T)  P_uz = triu(U Z^T, 1)
//   dlw_t = sum_{s >= t} (q dQ - k dK - b dB)_s + sum_{s > t} (a dA)_s + rowsum(E * H_C) ;  dw = dlw * lw
// 35 tile products in 7 barrier-separated phases; the LDS map below is a union over the phases (160 KB exactly).
// ------------------------------------------------------------------------------------------------------------------
namespace {
struct OutSmem {  // offsets in uint16 units
    static constexpr int TM1 = kC * LDK, CM1 = kN * LDC, SQ1 = kN * LDK, A1 = kC * LDC, ST = kC * kN * 2;
    // fixed for the whole kernel
    static constexpr int QTTh = 0, QTTl = QTTh + CM1, ATTh = QTTl + CM1, ATTl = ATTh + CM1;
    static constexpr int KHTh = ATTl + CM1, KHTl = KHTh + CM1, BHTh = KHTl + CM1, BHTl = BHTh + CM1;
    static constexpr int Vp = BHTl + CM1, DYp = Vp + TM1, Uh = DYp + TM1, Ul = Uh + TM1, Zh = Ul + TM1, Zl = Zh + TM1;
    static constexpr int STG = Zl + TM1;                    // fp32 [32][64] staging tiles: dK, dB, dQ, dA
    static constexpr int sK = STG, sB = STG + ST, sQ = STG + 2 * ST, sA = STG + 3 * ST;
    static constexpr int gC = STG + 4 * ST, dterm = gC + 2 * kN;  // 64 floats each
    static constexpr int S = dterm + 2 * kN;                // phase scratch
    // phases A-D inside S
    static constexpr int KHh = S, KHl = KHh + TM1, BHh = KHl + TM1, BHl = BHh + TM1;
    static constexpr int QTh = BHl + TM1, QTl = QTh + TM1, ATh = QTl + TM1, ATl = ATh + TM1;   // dead after phase A
    static constexpr int G1Th = QTh, G1Tl = G1Th + CM1, sV = G1Tl + CM1;                       // laid over QT/AT
    static constexpr int EGh = ATl + TM1, EGl = EGh + SQ1;                                     // (g_C E)[v][k]
    static constexpr int DYT = EGl + SQ1;
    static constexpr int endAD = DYT + CM1;
    // phases A-D inside the (still unused) staging area
    static constexpr int TMTh = STG, TMTl = TMTh + A1, QBTh = TMTl + A1, QBTl = QBTh + A1, QKTh = QBTl + A1, QKTl = QKTh + A1;
    static constexpr int AKTh = QKTl + A1, AKTl = AKTh + A1, ZTh = AKTl + A1, ZTl = ZTh + CM1;
    static constexpr int scratch = ZTh;                     // prologue cumsum scratch (4608 u16)
    // phases E-F inside S
    static constexpr int XTh = S, XTl = XTh + SQ1;          // (g_C E)^T [k][v], later H0^T [k][v]
    static constexpr int P0 = EGh;                          // 4 pairs of [32][LDC] planes over the dead EG / DYT area
    static constexpr int aDA = sV;                          // epilogue: fp32 [32][64] a*dA
    static constexpr int end16 = endAD;
    static constexpr size_t bytes = (size_t)end16 * 2;
};
static_assert(OutSmem::sV + OutSmem::ST <= OutSmem::EGh, "dV staging must end before the E planes");
static_assert(OutSmem::ZTl + OutSmem::CM1 <= OutSmem::gC, "phase A-D planes must fit in the staging area");
static_assert(OutSmem::P0 + 8 * OutSmem::A1 <= OutSmem::endAD, "P planes");
static_assert(OutSmem::XTl + OutSmem::SQ1 <= OutSmem::sV, "E^T / H0^T planes must not reach the dV staging tile");
static_assert(OutSmem::bytes <= 160 * 1024, "LDS budget");
static_assert(OutSmem::S % 8 == 0 && OutSmem::EGh % 8 == 0 && OutSmem::DYT % 8 == 0 && OutSmem::STG % 8 == 0, "alignment");

// X exact (single plane), Y exact
template <int K>
__device__ __forceinline__ void mma_ee(f32x16 &acc, const uint16_t *X, int ldx, const uint16_t *Y, int ldy, int lane) {
    mma_tile<K>(acc, X, ldx, Y, ldy, lane);
}
// D tile -> fp32 staging [32][64], columns [32 ct, 32 ct + 32)
__device__ __forceinline__ void stage_tile(const f32x16 &acc, float *stg, int ct, int lane) {
#pragma unroll
    for (int r = 0; r < 16; r++) stg[d_row(r, lane) * kN + ct * 32 + (lane & 31)] = acc[r];
}
__device__ __forceinline__ void ld_stage8(const float *stg, int pt, int pk, float (&x)[8]) {
    const float4 a = *reinterpret_cast<const float4 *>(stg + pt * kN + pk), b = *reinterpret_cast<const float4 *>(stg + pt * kN + pk + 4);
    x[0] = a.x; x[1] = a.y; x[2] = a.z; x[3] = a.w; x[4] = b.x; x[5] = b.y; x[6] = b.z; x[7] = b.w;
}
__device__ __forceinline__ void st_bf16x8(bf16_t *p, const float (&x)[8]) {
    uint4 o;
    o.x = cvt_pk(x[0], x[1]); o.y = cvt_pk(x[2], x[3]); o.z = cvt_pk(x[4], x[5]); o.w = cvt_pk(x[6], x[7]);
    *reinterpret_cast<uint4 *>(p) = o;
}
// 4 consecutive fp32 of row `row` -> hi/lo planes [..][LDK]
__device__ __forceinline__ void put4(uint16_t *Ph, uint16_t *Pl, int row, int c4, float4 x, float scale_x, float scale_y,
                                     float scale_z, float scale_w) {
    uint32_t h0, l0, h1, l1;
    split_pk(x.x * scale_x, x.y * scale_y, h0, l0);
    split_pk(x.z * scale_z, x.w * scale_w, h1, l1);
    *reinterpret_cast<uint2 *>(Ph + row * LDK + c4) = make_uint2(h0, h1);
    *reinterpret_cast<uint2 *>(Pl + row * LDK + c4) = make_uint2(l0, l1);
}
}  // namespace

__global__ __launch_bounds__(256) void wkv7c_bwd_out_kernel(
    int T_, int H, const bf16_t *__restrict__ w_, const bf16_t *__restrict__ q_, const bf16_t *__restrict__ k_,
    const bf16_t *__restrict__ v_, const bf16_t *__restrict__ a_, const bf16_t *__restrict__ b_, const bf16_t *__restrict__ dy_,
    const float *__restrict__ s_, const float *__restrict__ sa_, const float *__restrict__ tinv_, const float *__restrict__ e_vk,
    const float *__restrict__ e_kv, bf16_t *__restrict__ dw_, bf16_t *__restrict__ dq_, bf16_t *__restrict__ dk_,
    bf16_t *__restrict__ dv_, bf16_t *__restrict__ da_, bf16_t *__restrict__ db_) {
    extern __shared__ __attribute__((aligned(16))) uint16_t sm[];
    using L = OutSmem;
    float *sh_gC = reinterpret_cast<float *>(sm + L::gC), *sh_dterm = reinterpret_cast<float *>(sm + L::dterm);
    const int nc = T_ / kC;
    const int bh = blockIdx.x / nc, c = blockIdx.x - bh * nc;
    const int bb = bh / H, hh = bh - bb * H;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int pt = tid & 31, pk = (tid >> 5) * 8;
    const long tstride = (long)H * kN;
    const long off = ((long)bb * T_ * H + hh) * kN + (long)(c * kC + pt) * tstride + pk;

    // ---- global loads ---------------------------------------------------------------------------------------------------
    const Raw8 rw = ld8(w_ + off), rq = ld8(q_ + off), rk = ld8(k_ + off), ra = ld8(a_ + off), rb = ld8(b_ + off);
    const Raw8 rv = ld8(v_ + off), rdy = ld8(dy_ + off);
    const float4 ru0 = *reinterpret_cast<const float4 *>(sa_ + off), ru1 = *reinterpret_cast<const float4 *>(sa_ + off + 4);
    // 64x64 fp32 matrices: piece p = tid + 256 i covers row p >> 4, columns 4 (p & 15) .. +4
    const float *evk = e_vk + (long)blockIdx.x * kN * kN, *ekv = e_kv + (long)blockIdx.x * kN * kN;
    const long nck = T_ / kChunk;  // scalar-forward checkpoints (every 16 steps), s[b,h,n][k][v]
    const float *h0p = s_ + ((long)bh * nck + (2 * c - 1)) * kN * kN, *hcp = s_ + ((long)bh * nck + (2 * c + 1)) * kN * kN;
    float4 rekv[4], rh0[4];
    {
        float part[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int p = tid + 256 * i;
            rekv[i] = *reinterpret_cast<const float4 *>(ekv + p * 4);
            rh0[i] = c > 0 ? *reinterpret_cast<const float4 *>(h0p + p * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            const float4 hc = *reinterpret_cast<const float4 *>(hcp + p * 4);
            part[i] = rekv[i].x * hc.x + rekv[i].y * hc.y + rekv[i].z * hc.z + rekv[i].w * hc.w;
        }
        // rowsum(E * H_C): the 16 lanes tid & 15 share a row
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const float t = sum16(part[i]);
            if ((tid & 15) == 0) sh_dterm[(tid + 256 * i) >> 4] = t;
        }
    }
    load_tm<true>(tinv_ + (long)blockIdx.x * kC * kC, sm + L::TMTh, sm + L::TMTl, tid);

    // ---- prologue: decay, scaled operands ----------------------------------------------------------------------------------
    float lw[8], G[8], qv[8], kv[8], av[8], bv[8], gam[8], gprev[8], igam[8];
    cvt8(rw, lw);
#pragma unroll
    for (int j = 0; j < 8; j++) lw[j] = -fast_exp(lw[j]);
    {
        float *sh_G = reinterpret_cast<float *>(sm + L::scratch);
        chunk_cumsum(lw, G, sh_G, sh_G + kC * kN, tid, pt, pk);
    }
    cvt8(rq, qv); cvt8(rk, kv); cvt8(ra, av); cvt8(rb, bv);
#pragma unroll
    for (int j = 0; j < 8; j++) {
        gam[j] = fast_exp(G[j]);
        gprev[j] = fast_exp(G[j] - lw[j]);
        igam[j] = fast_exp(-G[j]);
    }
    if (pt == kC - 1) {
#pragma unroll
        for (int j = 0; j < 8; j++) sh_gC[pk + j] = gam[j];
    }
    {
        float x[8];
        uint32_t hi[4], lo[4];
#pragma unroll
        for (int j = 0; j < 8; j++) x[j] = qv[j] * gam[j];
        put_row8(sm + L::QTh, sm + L::QTl, pt * LDK + pk, x, hi, lo);
        put_col8(sm + L::QTTh, sm + L::QTTl, LDC, pk, pt, hi, lo);
#pragma unroll
        for (int j = 0; j < 8; j++) x[j] = av[j] * gprev[j];
        put_row8(sm + L::ATh, sm + L::ATl, pt * LDK + pk, x, hi, lo);
        put_col8(sm + L::ATTh, sm + L::ATTl, LDC, pk, pt, hi, lo);
#pragma unroll
        for (int j = 0; j < 8; j++) x[j] = kv[j] * igam[j];
        put_row8(sm + L::KHh, sm + L::KHl, pt * LDK + pk, x, hi, lo);
        put_col8(sm + L::KHTh, sm + L::KHTl, LDC, pk, pt, hi, lo);
#pragma unroll
        for (int j = 0; j < 8; j++) x[j] = bv[j] * igam[j];
        put_row8(sm + L::BHh, sm + L::BHl, pt * LDK + pk, x, hi, lo);
        put_col8(sm + L::BHTh, sm + L::BHTl, LDC, pk, pt, hi, lo);
        const float u[8] = {ru0.x, ru0.y, ru0.z, ru0.w, ru1.x, ru1.y, ru1.z, ru1.w};
        put_row8(sm + L::Uh, sm + L::Ul, pt * LDK + pk, u, hi, lo);
        *reinterpret_cast<uint4 *>(sm + L::Vp + pt * LDK + pk) = rv.r;     // bf16 inputs are exact: single planes
        *reinterpret_cast<uint4 *>(sm + L::DYp + pt * LDK + pk) = rdy.r;
        const uint32_t dyr[4] = {rdy.r.x, rdy.r.y, rdy.r.z, rdy.r.w};
#pragma unroll
        for (int j = 0; j < 8; j++) sm[L::DYT + (pk + j) * LDC + pt] = (uint16_t)(dyr[j >> 1] >> ((j & 1) * 16));
    }
    lds_barrier();  // sh_gC visible
    {
        // (g_C E)[v][k] planes: row v, scale per column k
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int p = tid + 256 * i, row = p >> 4, c4 = (p & 15) * 4;
            const float4 e = *reinterpret_cast<const float4 *>(evk + p * 4);
            put4(sm + L::EGh, sm + L::EGl, row, c4, e, sh_gC[c4], sh_gC[c4 + 1], sh_gC[c4 + 2], sh_gC[c4 + 3]);
        }
    }
    lds_barrier();
    // ---- phase A: A_qb^T, A_qk^T, A_ak^T ------------------------------------------------------------------------------------
    if (wave == 0) {
        f32x16 acc = zero16();  // D[t][s] = q~_t . b^_s, t >= s -> QBT[s][t]
        mma_tile3<kN>(acc, sm + L::QTh, sm + L::QTl, LDK, sm + L::BHh, sm + L::BHl, LDK, lane);
        mask_upper_T<false>(acc, lane);
        store_T_split(acc, sm + L::QBTh, sm + L::QBTl, LDC, lane);
    } else if (wave == 1) {
        f32x16 acc = zero16();  // q~_t . k^_s, t >= s -> QKT[s][t]
        mma_tile3<kN>(acc, sm + L::QTh, sm + L::QTl, LDK, sm + L::KHh, sm + L::KHl, LDK, lane);
        mask_upper_T<false>(acc, lane);
        store_T_split(acc, sm + L::QKTh, sm + L::QKTl, LDC, lane);
    } else if (wave == 2) {
        f32x16 acc = zero16();  // a~_t . k^_s, t > s -> AKT[s][t]
        mma_tile3<kN>(acc, sm + L::ATh, sm + L::ATl, LDK, sm + L::KHh, sm + L::KHl, LDK, lane);
        mask_upper_T<true>(acc, lane);
        store_T_split(acc, sm + L::AKTh, sm + L::AKTl, LDC, lane);
    }
    lds_barrier();
    // ---- phase B: G1[s][v] = sum_t A_qb[t][s] dY[t][v] + sum_k b^[s][k] (g_C E)[k][v]  -> G1T[v][s] ---------------------------
    if (wave <= 1) {
        const int vt = wave;
        f32x16 acc = zero16();
        mma_xs_ye<kC>(acc, sm + L::QBTh, sm + L::QBTl, LDC, sm + L::DYT + vt * 32 * LDC, LDC, lane);
        mma_tile3<kN>(acc, sm + L::BHh, sm + L::BHl, LDK, sm + L::EGh + vt * 32 * LDK, sm + L::EGl + vt * 32 * LDK, LDK, lane);
        store_T_split(acc, sm + L::G1Th + vt * 32 * LDC, sm + L::G1Tl + vt * 32 * LDC, LDC, lane);
    }
    lds_barrier();
    // ---- phase C: Z = T^T G1 in both orientations ---------------------------------------------------------------------------------
    if (wave <= 1) {
        const int vt = wave;
        f32x16 acc = zero16();  // D[t][v] = sum_s T[s][t] G1[s][v] -> ZT[v][t]
        mma_tile3<kC>(acc, sm + L::TMTh, sm + L::TMTl, LDC, sm + L::G1Th + vt * 32 * LDC, sm + L::G1Tl + vt * 32 * LDC, LDC, lane);
        store_T_split(acc, sm + L::ZTh + vt * 32 * LDC, sm + L::ZTl + vt * 32 * LDC, LDC, lane);
    } else {
        const int vt = wave - 2;
        f32x16 acc = zero16();  // D[v][t] -> Z[t][v]
        mma_tile3<kC>(acc, sm + L::G1Th + vt * 32 * LDC, sm + L::G1Tl + vt * 32 * LDC, LDC, sm + L::TMTh, sm + L::TMTl, LDC, lane);
        store_T_split(acc, sm + L::Zh + vt * 32, sm + L::Zl + vt * 32, LDK, lane);
    }
    lds_barrier();
    // ---- phase D: dV[s][v] = sum_t A_qk[t][s] dY[t][v] + A_ak[t][s] Z[t][v] + sum_k k^[s][k] (g_C E)[k][v] -----------------------
    if (wave <= 1) {
        const int vt = wave;
        f32x16 acc = zero16();
        mma_xs_ye<kC>(acc, sm + L::QKTh, sm + L::QKTl, LDC, sm + L::DYT + vt * 32 * LDC, LDC, lane);
        mma_tile3<kC>(acc, sm + L::AKTh, sm + L::AKTl, LDC, sm + L::ZTh + vt * 32 * LDC, sm + L::ZTl + vt * 32 * LDC, LDC, lane);
        mma_tile3<kN>(acc, sm + L::KHh, sm + L::KHl, LDK, sm + L::EGh + vt * 32 * LDK, sm + L::EGl + vt * 32 * LDK, LDK, lane);
        stage_tile(acc, reinterpret_cast<float *>(sm + L::sV), vt, lane);
    }
    lds_barrier();
    // ---- phase E1: dV out; (g_C E)^T planes; the four P matrices of dK / dB --------------------------------------------------------
    {
        float x[8];
        ld_stage8(reinterpret_cast<const float *>(sm + L::sV), pt, pk, x);
        st_bf16x8(dv_ + off, x);
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int p = tid + 256 * i, row = p >> 4, c4 = (p & 15) * 4;
            const float g = sh_gC[row];
            put4(sm + L::XTh, sm + L::XTl, row, c4, rekv[i], g, g, g, g);
        }
        uint16_t *Ph = sm + L::P0 + wave * 2 * L::A1, *Pl = Ph + L::A1;
        f32x16 acc = zero16();  // D[m = s][n = t], kept for s >= t (P_vy, P_uy) or s > t (P_vz, P_uz); stored [t][s]
        if (wave == 0) {
            mma_ee<kN>(acc, sm + L::DYp, LDK, sm + L::Vp, LDK, lane);                       // dy_s . v_t
            mask_upper_T<false>(acc, lane);
        } else if (wave == 1) {
            mma_xs_ye<kN>(acc, sm + L::Zh, sm + L::Zl, LDK, sm + L::Vp, LDK, lane);         // z_s . v_t
            mask_upper_T<true>(acc, lane);
        } else if (wave == 2) {
            mma_tile2y<kN>(acc, sm + L::DYp, LDK, sm + L::Uh, sm + L::Ul, LDK, lane);       // dy_s . u_t
            mask_upper_T<false>(acc, lane);
        } else {
            mma_tile3<kN>(acc, sm + L::Zh, sm + L::Zl, LDK, sm + L::Uh, sm + L::Ul, LDK, lane);  // z_s . u_t
            mask_upper_T<true>(acc, lane);
        }
        store_T_split(acc, Ph, Pl, LDC, lane);
    }
    lds_barrier();
    // ---- phase F1: dK (waves 0,1) and dB (waves 2,3), unscaled, to staging ---------------------------------------------------------
    {
        const int kt = wave & 1;
        const uint16_t *P1h = sm + L::P0 + (wave < 2 ? 0 : 2) * 2 * L::A1, *P1l = P1h + L::A1, *P2h = P1h + 2 * L::A1, *P2l = P2h + L::A1;
        f32x16 acc = zero16();  // D[m = t][n = k]
        mma_tile3<kC>(acc, P1h, P1l, LDC, sm + L::QTTh + kt * 32 * LDC, sm + L::QTTl + kt * 32 * LDC, LDC, lane);
        mma_tile3<kC>(acc, P2h, P2l, LDC, sm + L::ATTh + kt * 32 * LDC, sm + L::ATTl + kt * 32 * LDC, LDC, lane);
        if (wave < 2) {
            mma_tile2y<kN>(acc, sm + L::Vp, LDK, sm + L::XTh + kt * 32 * LDK, sm + L::XTl + kt * 32 * LDK, LDK, lane);
            stage_tile(acc, reinterpret_cast<float *>(sm + L::sK), kt, lane);
        } else {
            mma_tile3<kN>(acc, sm + L::Uh, sm + L::Ul, LDK, sm + L::XTh + kt * 32 * LDK, sm + L::XTl + kt * 32 * LDK, LDK, lane);
            stage_tile(acc, reinterpret_cast<float *>(sm + L::sB), kt, lane);
        }
    }
    lds_barrier();
    // ---- phase E2: H0^T planes over (g_C E)^T; the four transposed P matrices of dQ / dA -------------------------------------------
    {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int p = tid + 256 * i, row = p >> 4, c4 = (p & 15) * 4;
            put4(sm + L::XTh, sm + L::XTl, row, c4, rh0[i], 1.f, 1.f, 1.f, 1.f);
        }
        uint16_t *Ph = sm + L::P0 + wave * 2 * L::A1, *Pl = Ph + L::A1;
        f32x16 acc = zero16();  // D[m = s][n = t'], kept for s <= t' (vy, uy) or s < t' (vz, uz); stored [t'][s]
        if (wave == 0) {
            mma_ee<kN>(acc, sm + L::Vp, LDK, sm + L::DYp, LDK, lane);                       // v_s . dy_t'
            mask_lower_T<false>(acc, lane);
        } else if (wave == 1) {
            mma_xs_ye<kN>(acc, sm + L::Uh, sm + L::Ul, LDK, sm + L::DYp, LDK, lane);        // u_s . dy_t'
            mask_lower_T<false>(acc, lane);
        } else if (wave == 2) {
            mma_tile2y<kN>(acc, sm + L::Vp, LDK, sm + L::Zh, sm + L::Zl, LDK, lane);        // v_s . z_t'
            mask_lower_T<true>(acc, lane);
        } else {
            mma_tile3<kN>(acc, sm + L::Uh, sm + L::Ul, LDK, sm + L::Zh, sm + L::Zl, LDK, lane);  // u_s . z_t'
            mask_lower_T<true>(acc, lane);
        }
        store_T_split(acc, Ph, Pl, LDC, lane);
    }
    lds_barrier();
    // ---- phase F2: dQ (waves 0,1) and dA (waves 2,3), unscaled, to staging ---------------------------------------------------------
    {
        const int kt = wave & 1;
        // dQ uses P planes 0 (vy) and 1 (uy); dA uses 2 (vz) and 3 (uz)
        const uint16_t *P1h = sm + L::P0 + (wave < 2 ? 0 : 2) * 2 * L::A1, *P1l = P1h + L::A1, *P2h = P1h + 2 * L::A1, *P2l = P2h + L::A1;
        f32x16 acc = zero16();  // D[m = t'][n = k]
        mma_tile3<kC>(acc, P1h, P1l, LDC, sm + L::KHTh + kt * 32 * LDC, sm + L::KHTl + kt * 32 * LDC, LDC, lane);
        mma_tile3<kC>(acc, P2h, P2l, LDC, sm + L::BHTh + kt * 32 * LDC, sm + L::BHTl + kt * 32 * LDC, LDC, lane);
        if (wave < 2) {
            mma_tile2y<kN>(acc, sm + L::DYp, LDK, sm + L::XTh + kt * 32 * LDK, sm + L::XTl + kt * 32 * LDK, LDK, lane);
            stage_tile(acc, reinterpret_cast<float *>(sm + L::sQ), kt, lane);
        } else {
            mma_tile3<kN>(acc, sm + L::Zh, sm + L::Zl, LDK, sm + L::XTh + kt * 32 * LDK, sm + L::XTl + kt * 32 * LDK, LDK, lane);
            stage_tile(acc, reinterpret_cast<float *>(sm + L::sA), kt, lane);
        }
    }
    lds_barrier();
    // ---- epilogue: decay scaling, decay gradient, stores ----------------------------------------------------------------------------
    float dQ[8], dK[8], dB[8], dA[8], e[8];
    ld_stage8(reinterpret_cast<const float *>(sm + L::sQ), pt, pk, dQ);
    ld_stage8(reinterpret_cast<const float *>(sm + L::sK), pt, pk, dK);
    ld_stage8(reinterpret_cast<const float *>(sm + L::sB), pt, pk, dB);
    ld_stage8(reinterpret_cast<const float *>(sm + L::sA), pt, pk, dA);
    float *sh_ada = reinterpret_cast<float *>(sm + L::aDA);
#pragma unroll
    for (int j = 0; j < 8; j++) {
        dQ[j] *= gam[j];
        dK[j] *= igam[j];
        dB[j] *= igam[j];
        dA[j] *= gprev[j];
        e[j] = qv[j] * dQ[j] - kv[j] * dK[j] - bv[j] * dB[j];
    }
    {
        float t[8];
#pragma unroll
        for (int j = 0; j < 8; j++) t[j] = av[j] * dA[j];
        *reinterpret_cast<float4 *>(&sh_ada[pt * kN + pk]) = make_float4(t[0], t[1], t[2], t[3]);
        *reinterpret_cast<float4 *>(&sh_ada[pt * kN + pk + 4]) = make_float4(t[4], t[5], t[6], t[7]);
    }
    st_bf16x8(dq_ + off, dQ);
    st_bf16x8(dk_ + off, dK);
    st_bf16x8(db_ + off, dB);
    st_bf16x8(da_ + off, dA);
    lds_barrier();
    if (pt < kC - 1) {
        const float4 n0 = *reinterpret_cast<const float4 *>(&sh_ada[(pt + 1) * kN + pk]);
        const float4 n1 = *reinterpret_cast<const float4 *>(&sh_ada[(pt + 1) * kN + pk + 4]);
        e[0] += n0.x; e[1] += n0.y; e[2] += n0.z; e[3] += n0.w; e[4] += n1.x; e[5] += n1.y; e[6] += n1.z; e[7] += n1.w;
    }
    {
        // suffix sums over the 32 steps (staging area is free now): dlw_t = sum_{s >= t} e_s + dterm
        float *sh_E = reinterpret_cast<float *>(sm + L::sK), *sh_seg = reinterpret_cast<float *>(sm + L::sQ);
        *reinterpret_cast<float4 *>(&sh_E[pt * kN + pk]) = make_float4(e[0], e[1], e[2], e[3]);
        *reinterpret_cast<float4 *>(&sh_E[pt * kN + pk + 4]) = make_float4(e[4], e[5], e[6], e[7]);
        lds_barrier();
        {
            const int ch = tid & 63, seg = tid >> 6;
            float run = 0.f;
#pragma unroll
            for (int i = 7; i >= 0; i--) {
                run += sh_E[(seg * 8 + i) * kN + ch];
                sh_E[(seg * 8 + i) * kN + ch] = run;
            }
            sh_seg[seg * kN + ch] = run;
        }
        lds_barrier();
        float dG[8];
        const float4 g0 = *reinterpret_cast<const float4 *>(&sh_E[pt * kN + pk]), g1 = *reinterpret_cast<const float4 *>(&sh_E[pt * kN + pk + 4]);
        dG[0] = g0.x; dG[1] = g0.y; dG[2] = g0.z; dG[3] = g0.w; dG[4] = g1.x; dG[5] = g1.y; dG[6] = g1.z; dG[7] = g1.w;
        for (int sgi = (pt >> 3) + 1; sgi < 4; sgi++) {
#pragma unroll
            for (int j = 0; j < 8; j++) dG[j] += sh_seg[sgi * kN + pk + j];
        }
#pragma unroll
        for (int j = 0; j < 8; j++) dG[j] = (dG[j] + sh_dterm[pk + j]) * lw[j];
        st_bf16x8(dw_ + off, dG);
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

int chunk_bwd_out_bf16(int B, int T_, int H, const void *w, const void *q, const void *k, const void *v, const void *a,
                       const void *b, const void *dy, const float *s, const float *sa, const float *tinv, const float *e_vk,
                       const float *e_kv, void *dw, void *dq, void *dk, void *dv, void *da, void *db, hipStream_t st) {
    static bool attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&wkv7c_bwd_out_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)OutSmem::bytes);
        if (e != hipSuccess) return (int)e;
        attr = true;
    }
    (void)hipGetLastError();
    hipLaunchKernelGGL(wkv7c_bwd_out_kernel, dim3(B * H * (T_ / kC)), dim3(256), OutSmem::bytes, st, T_, H, (const bf16_t *)w,
                       (const bf16_t *)q, (const bf16_t *)k, (const bf16_t *)v, (const bf16_t *)a, (const bf16_t *)b,
                       (const bf16_t *)dy, s, sa, tinv, e_vk, e_kv, (bf16_t *)dw, (bf16_t *)dq, (bf16_t *)dk, (bf16_t *)dv,
                       (bf16_t *)da, (bf16_t *)db);
    return (int)hipGetLastError();
}

}  // namespace rwkv7
