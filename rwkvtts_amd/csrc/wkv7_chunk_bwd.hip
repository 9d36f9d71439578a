// rwkvtts_amd/csrc/wkv7_chunk_bwd.hip -- chunked (MFMA) WKV7 backward for gfx950, bf16 tensors.
//
// Same gradients as wkv7_bwd.hip (reference wkv7_cuda.cu:54-130), evaluated 32 steps at a time on the matrix cores.
// With H = S^T in R^{K x V} and the per-chunk quantities of chunk_common.h (q~, a~, k^, b^, g_C, T = (I - A_ab)^-1,
// W = T A~) the forward state obeys  H_{c+1} = M_c H_c + N_c  with  M_c = diag(g_C)(I + B^^T W), and the adjoint state
//     E_c = M_c^T E_{c+1} + N'_c ,   N'_c = Q~^T dY + W^T (A_qb^T dY)              (E_c = dL/dH at the START of chunk c)
// is the only sequential object of the backward pass (tests/chunked_proto2.py validates the algebra against the oracle).
// Three kernels:
//   wkv7c_bwd_pre_kernel   grid B*H*(T/32), parallel: M_c^T (bf16 hi/lo planes) and N'_c (fp32, MFMA tile layout)
//   wkv7c_state_kernel     grid B*H, sequential over chunks in reverse: E for every chunk (one 64x64x64 product each)
//   wkv7c_bwd_out_kernel   grid B*H*(T/32), parallel: dw,dq,dk,dv,da,db of a chunk from (H_c, E_{c+1}, U = sa, dY)
// Inputs that come from the forward pass: sa (= U) and the fp32 state checkpoints of wkv7_fwd.hip (every 16 steps,
// s[b,h,n][k][v]; the state at the start of chunk c is checkpoint 2c-1), and T from wkv7c_prep_kernel.
#include "chunk_common.h"

namespace rwkv7 {

#ifdef WKV7C_TIMING
// profiling build only (python -m rwkvtts_amd.build --timing): per-phase cycle totals of workgroup 0, per wave
__device__ long long g_cbwd_timing[4 * 32];
#define BSTAMP(i)                                                                  \
    do {                                                                           \
        const long long now_ = __builtin_readcyclecounter();                       \
        if (blockIdx.x == 0 && lane == 0) g_cbwd_timing[wave * 32 + (i)] += now_ - tprev_; \
        tprev_ = now_;                                                             \
    } while (0)
#define BSTAMP_INIT long long tprev_ = __builtin_readcyclecounter()
#else
#define BSTAMP(i) do { } while (0)
#define BSTAMP_INIT do { } while (0)
#endif

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

constexpr int kChunksPerWG = 4;     // bwd_pre walks this many consecutive chunks per workgroup, prefetching the next one's inputs
constexpr int kOutChunksPerWG = 8;  // bwd_out (measured: 4 -> 8 is +4 % for pre, -1 % for out)

struct PreSmem {  // offsets in uint16 units
    // phase-1 inputs, contiguous: dead after phase 1 and overlaid by G1T and the M^T planes
    static constexpr int QTh = 0, QTl = QTh + kC * LDK, BHh = QTl + kC * LDK, BHl = BHh + kC * LDK;
    static constexpr int ATTh = BHl + kC * LDK, ATTl = ATTh + kN * LDC, TMh = ATTl + kN * LDC, TMl = TMh + kC * LDC;
    static constexpr int end1 = TMl + kC * LDC;
    static constexpr int G1Th = 0, G1Tl = G1Th + kN * LDC, MPh = G1Tl + kN * LDC, MPl = MPh + kN * LDK;   // M^T[k][k'] planes
    static constexpr int QTTh = end1, QTTl = QTTh + kN * LDC, BCTh = QTTl + kN * LDC, BCTl = BCTh + kN * LDC;
    static constexpr int DYT = BCTl + kN * LDC, QBTh = DYT + kN * LDC, QBTl = QBTh + kC * LDC;
    static constexpr int WTh = QBTl + kC * LDC, WTl = WTh + kN * LDC;   // W^T planes
    static constexpr int gC = WTl + kN * LDC;  // 64 floats
    static constexpr int end16 = gC + 2 * kN;
    static constexpr size_t bytes = (size_t)end16 * 2;
};
static_assert(PreSmem::MPl + kN * LDK <= PreSmem::end1, "G1T + M^T planes must fit over the phase-1 inputs");
static_assert(PreSmem::ATTh % 8 == 0 && PreSmem::QTTh % 8 == 0 && PreSmem::WTh % 8 == 0 && PreSmem::gC % 8 == 0 &&
                  PreSmem::MPh % 8 == 0, "16-byte alignment");
static_assert(PreSmem::bytes <= 80 * 1024, "two workgroups per CU");

}  // namespace

// ------------------------------------------------------------------------------------------------------------------
// pre: M_c^T (bf16 hi/lo, in MFMA A-fragment order: [chunk][k-tile][plane][k-step][lane][8]) and N'_c
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void wkv7c_bwd_pre_kernel(int T_, int H, int nchunks_total, const bf16_t *__restrict__ w_,
                                                            const bf16_t *__restrict__ q_, const bf16_t *__restrict__ a_,
                                                            const bf16_t *__restrict__ b_, const bf16_t *__restrict__ dy_,
                                                            const float *__restrict__ tinv_, uint16_t *__restrict__ mt_,
                                                            float *__restrict__ np_) {
    extern __shared__ __attribute__((aligned(16))) uint16_t sm[];
    using L = PreSmem;
    float *sh_gC = reinterpret_cast<float *>(sm + L::gC);
    const int nc = T_ / kC;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int pt = tid & 31, pk = (tid >> 5) * 8;
    const long tstride = (long)H * kN;

    struct In {
        Raw8 w, q, a, b, dy;
        float4 tm;
    };
    auto load = [&](int chunk) {
        const int bh = chunk / nc, c = chunk - bh * nc;
        const int bb = bh / H, hh = bh - bb * H;
        const long off = ((long)bb * T_ * H + hh) * kN + (long)(c * kC + pt) * tstride + pk;
        In r;
        r.w = ld8(w_ + off); r.q = ld8(q_ + off); r.a = ld8(a_ + off); r.b = ld8(b_ + off); r.dy = ld8(dy_ + off);
        r.tm = *reinterpret_cast<const float4 *>(tinv_ + (long)chunk * kC * kC + tid * 4);
        return r;
    };
    const int chunk0 = blockIdx.x * kChunksPerWG;
    In cur = load(chunk0);
    for (int ci = 0; ci < kChunksPerWG; ci++) {
        const int chunk = chunk0 + ci;
        if (chunk >= nchunks_total) break;
        In nxt = cur;
        if (ci + 1 < kChunksPerWG && chunk + 1 < nchunks_total) nxt = load(chunk + 1);
        // ---- prologue -----------------------------------------------------------------------------------------------
        put_tm<false>(cur.tm, sm + L::TMh, sm + L::TMl, tid);
        float lw[8], G[8];
        cvt8(cur.w, lw);
#pragma unroll
        for (int j = 0; j < 8; j++) lw[j] = -fast_exp(lw[j]);
        chunk_cumsum(lw, G);
        if (pt == kC - 1) {
#pragma unroll
            for (int j = 0; j < 8; j++) sh_gC[pk + j] = fast_exp(G[j]);
        }
        lds_barrier();  // sh_gC complete
        {
            float qv[8], av[8], bv[8], x[8];
            cvt8(cur.q, qv); cvt8(cur.a, av); cvt8(cur.b, bv);
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
            const uint32_t dyr[4] = {cur.dy.r.x, cur.dy.r.y, cur.dy.r.z, cur.dy.r.w};  // dY is bf16: exact, one plane
#pragma unroll
            for (int j = 0; j < 8; j++) sm[L::DYT + (pk + j) * LDC + pt] = (uint16_t)(dyr[j >> 1] >> ((j & 1) * 16));
        }
        lds_barrier();
        // ---- phase 1: A_qb^T (wave 0), W = T A~ (waves 1, 2) -----------------------------------------------------------
        if (wave == 0) {
            f32x16 acc = zero16();  // D[m = t][n = s] = q~_t . b^_s, kept for t >= s; stored as QBT[s][t]
            mma_tile3<kN, 2>(acc, sm + L::QTh, sm + L::QTl, LDK, sm + L::BHh, sm + L::BHl, LDK, lane);
            mask_upper_T<false>(acc, lane);
            store_T_split(acc, sm + L::QBTh, sm + L::QBTl, LDC, lane);
        } else if (wave <= 2) {
            const int kt = wave - 1;
            f32x16 acc = zero16();  // D[m = t][n = k] = sum_s T[t][s] a~[s][k]; stored as WT[k][t]
            mma_tile3<kC, 2>(acc, sm + L::TMh, sm + L::TMl, LDC, sm + L::ATTh + kt * 32 * LDC, sm + L::ATTl + kt * 32 * LDC, LDC, lane);
            store_T_split(acc, sm + L::WTh + kt * 32 * LDC, sm + L::WTl + kt * 32 * LDC, LDC, lane);
        }
        lds_barrier();
        // ---- phase 2: G1 = A_qb^T dY (waves 0, 1); M^T = diag(g_C) + W^T (B^ g_C) (waves 2, 3) -----------------------
        if (wave <= 1) {
            const int vt = wave;
            f32x16 acc = zero16();  // D[m = s][n = v] = sum_t QBT[s][t] dY[t][v]; stored as G1T[v][s]
            mma_xs_ye<kC, 2>(acc, sm + L::QBTh, sm + L::QBTl, LDC, sm + L::DYT + vt * 32 * LDC, LDC, lane);
            store_T_split(acc, sm + L::G1Th + vt * 32 * LDC, sm + L::G1Tl + vt * 32 * LDC, LDC, lane);
        } else {
            const int mt = wave - 2;  // rows k' of the product below
#pragma unroll
            for (int nt = 0; nt < 2; nt++) {
                f32x16 acc = zero16();  // D[m = k'][n = k] = sum_t (b^ g_C)[t][k'] W[t][k]; stored as M^T[k][k']
                mma_tile3<kC, 2>(acc, sm + L::BCTh + mt * 32 * LDC, sm + L::BCTl + mt * 32 * LDC, LDC, sm + L::WTh + nt * 32 * LDC,
                              sm + L::WTl + nt * 32 * LDC, LDC, lane);
                if (mt == nt) {
                    const int n = lane & 31;
#pragma unroll
                    for (int r = 0; r < 16; r++)
                        if (d_row(r, lane) == n) acc[r] += sh_gC[mt * 32 + n];
                }
                store_T_split(acc, sm + L::MPh + nt * 32 * LDK + mt * 32, sm + L::MPl + nt * 32 * LDK + mt * 32, LDK, lane);
            }
        }
        lds_barrier();
        // ---- phase 3: N' = Q~^T dY + W^T G1 (one tile per wave, MFMA register layout); M^T planes -> fragment order ---------
        {
            const int mt = wave >> 1, nt = wave & 1;
            f32x16 acc = zero16();  // D[m = k][n = v]
            mma_xs_ye<kC, 2>(acc, sm + L::QTTh + mt * 32 * LDC, sm + L::QTTl + mt * 32 * LDC, LDC, sm + L::DYT + nt * 32 * LDC, LDC, lane);
            mma_tile3<kC, 2>(acc, sm + L::WTh + mt * 32 * LDC, sm + L::WTl + mt * 32 * LDC, LDC, sm + L::G1Th + nt * 32 * LDC,
                          sm + L::G1Tl + nt * 32 * LDC, LDC, lane);
            float *o = np_ + (((long)chunk * 4 + wave) * 64 + lane) * 16;
#pragma unroll
            for (int j = 0; j < 4; j++)
                *reinterpret_cast<float4 *>(o + 4 * j) = make_float4(acc[4 * j], acc[4 * j + 1], acc[4 * j + 2], acc[4 * j + 3]);
            // wave = (k-tile mt, plane): the four 16-byte A fragments of its 32 rows of M^T
            const uint16_t *pl = sm + ((wave & 1) ? L::MPl : L::MPh) + (mt * 32 + (lane & 31)) * LDK + (lane >> 5) * 8;
            uint16_t *mo = mt_ + ((long)chunk * 4 + wave) * 4 * 512 + lane * 8;
#pragma unroll
            for (int i = 0; i < 4; i++) *reinterpret_cast<uint4 *>(mo + i * 512) = *reinterpret_cast<const uint4 *>(pl + 16 * i);
        }
        lds_barrier();  // the next chunk's prologue overwrites what phase 3 reads
        cur = nxt;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// state: E_c = M_c^T E_{c+1} + N'_c, c = nc-1 .. 0; writes E_{c+1} (the adjoint state chunk c sees at its end) for every c
// as e_kv[b,h,c][k][v].  M^T arrives in A-fragment order and N' in accumulator
// order, so both go from global memory straight into MFMA operands / accumulators; only E itself passes through LDS
// (accumulator layout -> B-operand planes).  The chain is one 64x64x64 product per chunk; inputs are prefetched three
// chunks ahead in registers so that the ~2 us HBM latency is off the critical path.
// ------------------------------------------------------------------------------------------------------------------
namespace {
struct StateSmem {  // E planes of one half of the value columns, [32 v][64 k], double buffered
    static constexpr int E0h = 0, E0l = E0h + kC * LDK, E1h = E0l + kC * LDK, E1l = E1h + kC * LDK;
    static constexpr int end16 = E1l + kC * LDK;
    static constexpr size_t bytes = (size_t)end16 * 2;
};
}  // namespace

__global__ __launch_bounds__(128) void wkv7c_state_kernel(int nc, int H, const uint16_t *__restrict__ mt_, const float *__restrict__ np_,
                                                          float *__restrict__ e_kv, const int *__restrict__ seq_off_) {
    extern __shared__ __attribute__((aligned(16))) uint16_t sm[];
    using L = StateSmem;
    // the value columns of E never mix (E_c = M^T E + N' acts on columns): one workgroup per (head, half of the value
    // columns), 2 waves = the two 32-row tiles of that half -> 2 B H workgroups keep all 256 CUs streaming
    // Workgroups are dealt round-robin to the 8 XCDs (each with its own L2): the two halves of a head get block ids g and
    // g + 8 so that they share an L2 and M_c^T is fetched from HBM once, not twice.
    int bh, nt;
    if ((gridDim.x & 15) == 0) {
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        bh = (j >> 1) * 8 + xcd;
        nt = j & 1;
    } else {
        bh = blockIdx.x >> 1;
        nt = blockIdx.x & 1;
    }
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int mt = wave;
    // packed rows: one workgroup pair per (sequence, head) walks only that sequence's chunks (see wkv7c_fwd_kernel)
    int c0 = 0, c1 = nc;
    if (seq_off_) {
        const int sq = bh / H, hh = bh - sq * H;
        const int g0 = seq_off_[sq], g1 = seq_off_[sq + 1];
        const int bb = g0 / nc;
        c0 = g0 - bb * nc;
        c1 = c0 + (g1 - g0);
        bh = bb * H + hh;
        if (c1 <= c0) return;
    }

    struct In {
        bf16x8 mh[4], ml[4];
        float4 n[4];
    };
    auto load = [&](int c) {
        In r;
        if (c >= c0) {
            const uint16_t *mp = mt_ + (((long)bh * nc + c) * 4 + mt * 2) * 4 * 512 + lane * 8;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                r.mh[i] = *reinterpret_cast<const bf16x8 *>(mp + i * 512);
                r.ml[i] = *reinterpret_cast<const bf16x8 *>(mp + 4 * 512 + i * 512);
            }
            const float *np = np_ + ((((long)bh * nc + c) * 4 + mt * 2 + nt) * 64 + lane) * 16;
#pragma unroll
            for (int j = 0; j < 4; j++) r.n[j] = *reinterpret_cast<const float4 *>(np + 4 * j);
        }
        return r;
    };
    for (int i = tid; i < 2 * kC * LDK; i += 128) sm[L::E0h + i] = 0;  // E_{nc} = 0
    f32x16 E = zero16();  // this wave's tile of the current E, accumulator layout [m = k][n = v]
    int cur = 0;
    auto step = [&](int c, const In &in) {
        {
            // E_{c+1}: what chunk c receives from the future
            float *pk = e_kv + ((long)bh * nc + c) * kN * kN + nt * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; r++) pk[(long)(mt * 32 + d_row(r, lane)) * kN] = E[r];
        }
        const uint16_t *Eh = sm + (cur ? L::E1h : L::E0h) + (lane & 31) * LDK + (lane >> 5) * 8, *El = Eh + kC * LDK;
        f32x16 acc;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            acc[4 * j] = in.n[j].x; acc[4 * j + 1] = in.n[j].y; acc[4 * j + 2] = in.n[j].z; acc[4 * j + 3] = in.n[j].w;
        }
        bf16x8 eh[4], el[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            eh[i] = *reinterpret_cast<const bf16x8 *>(Eh + 16 * i);
            el[i] = *reinterpret_cast<const bf16x8 *>(El + 16 * i);
        }
        // three independent accumulator chains (one per hi/lo term): this loop is the sequential critical path, and
        // back-to-back dependent MFMAs of a lone wave expose their latency (one chain 171 us, two or three 163 us)
        f32x16 acc_b = zero16(), acc_c = zero16();
#pragma unroll
        for (int i = 0; i < 4; i++) {  // D[m = k][n = v] += sum_k' M^T[k][k'] E[k'][v]
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(in.mh[i], eh[i], acc, 0, 0, 0);
            acc_b = __builtin_amdgcn_mfma_f32_32x32x16_bf16(in.mh[i], el[i], acc_b, 0, 0, 0);
            acc_c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(in.ml[i], eh[i], acc_c, 0, 0, 0);
        }
        E = acc + (acc_b + acc_c);
        uint16_t *Oh = sm + (cur ? L::E0h : L::E1h), *Ol = Oh + kC * LDK;
        store_T_split(E, Oh + mt * 32, Ol + mt * 32, LDK, lane);  // planes [v (this half)][k]
        lds_barrier();
        cur ^= 1;
    };
    In r0 = load(c1 - 1), r1 = load(c1 - 2), r2 = load(c1 - 3);
    lds_barrier();
    for (int c = c1 - 1; c >= c0; c -= 3) {
        step(c, r0);
        r0 = load(c - 3);
        if (c - 1 >= c0) {
            step(c - 1, r1);
            r1 = load(c - 4);
        }
        if (c - 2 >= c0) {
            step(c - 2, r2);
            r2 = load(c - 5);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// out: all six gradients of one chunk from (H_c, E_{c+1}, U, dY) -- tests/chunked_proto2.py:bwd3, third loop
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
    // staging tiles are fp32 [32][64 + 4]: with the step index across lanes (row stride 272 B) a thread's float4 read would
    // otherwise hit the same 4 banks in all 32 lanes (measured: 46 % of the kernel's LDS cycles were bank conflicts)
    static constexpr int kStLD = kN + 4;
    static constexpr int TM1 = kC * LDK, CM1 = kN * LDC, SQ1 = kN * LDK, A1 = kC * LDC, ST = kC * kStLD * 2;
    // fixed for the whole kernel: the scaled operands and V, dY, U, Z, all TIME-major [t][.]; products that contract over
    // time fetch them with LDS transpose reads (frag_tr), so no channel-major copies exist
    static constexpr int QTh = 0, QTl = QTh + TM1, ATh = QTl + TM1, ATl = ATh + TM1;
    static constexpr int KHh = ATl + TM1, KHl = KHh + TM1, BHh = KHl + TM1, BHl = BHh + TM1;
    static constexpr int Vp = BHl + TM1, DYp = Vp + TM1, Uh = DYp + TM1, Ul = Uh + TM1, Zh = Ul + TM1, Zl = Zh + TM1;
    static constexpr int STG = Zl + TM1;                    // fp32 [32][64] staging tiles: dK, dB, dQ, dA
    static constexpr int sK = STG, sB = STG + ST, sQ = STG + 2 * ST, sA = STG + 3 * ST;
    static constexpr int gC = STG + 4 * ST, dterm = gC + 2 * kN;  // 64 floats each
    static constexpr int S = dterm + 2 * kN;                // phase scratch
    // phases A-D inside S
    static constexpr int G1Th = S, G1Tl = G1Th + CM1, sV = G1Tl + CM1;
    static constexpr int XTh = sV + ST, XTl = XTh + SQ1;    // (g_C E)[k][v], after phase F1 H0^T [k][v]
    // phases A-D inside the (still unused) staging area
    static constexpr int TMTh = STG, TMTl = TMTh + A1, QBTh = TMTl + A1, QBTl = QBTh + A1, QKTh = QBTl + A1, QKTl = QKTh + A1;
    static constexpr int AKTh = QKTl + A1, AKTl = AKTh + A1;
    // phases E-F inside S
    static constexpr int P0 = XTl + SQ1;                    // 4 pairs of [32][LDC] planes
    static constexpr int end16 = P0 + 8 * A1;
    static constexpr size_t bytes = (size_t)end16 * 2;
};
static_assert(OutSmem::AKTl + OutSmem::A1 <= OutSmem::gC, "phase A-D planes must fit in the staging area");
static_assert(OutSmem::bytes <= 160 * 1024, "LDS budget");
static_assert(OutSmem::S + 7 * kC * LDK + kC * OutSmem::kStLD * 2 <= OutSmem::end16, "raw-row restaging must fit in the phase scratch");
static_assert(OutSmem::S + 6 * kC * LDK <= OutSmem::end16, "output restaging must fit in the phase scratch");
static_assert(OutSmem::S % 8 == 0 && OutSmem::XTh % 8 == 0 && OutSmem::P0 % 8 == 0 && OutSmem::STG % 8 == 0 && OutSmem::sV % 8 == 0, "alignment");

// X exact (single plane), Y exact
template <int K>
__device__ __forceinline__ void mma_ee(f32x16 &acc, const uint16_t *X, int ldx, const uint16_t *Y, int ldy, int lane) {
    mma_tile<K>(acc, X, ldx, Y, ldy, lane);
}
// D tile -> fp32 staging [32][64], columns [32 ct, 32 ct + 32)
__device__ __forceinline__ void stage_tile(const f32x16 &acc, float *stg, int ct, int lane) {
#pragma unroll
    for (int r = 0; r < 16; r++) stg[d_row(r, lane) * OutSmem::kStLD + ct * 32 + (lane & 31)] = acc[r];
}
__device__ __forceinline__ void ld_stage8(const float *stg, int pt, int pk, float (&x)[8]) {
    const float4 a = *reinterpret_cast<const float4 *>(stg + pt * OutSmem::kStLD + pk),
                 b = *reinterpret_cast<const float4 *>(stg + pt * OutSmem::kStLD + pk + 4);
    x[0] = a.x; x[1] = a.y; x[2] = a.z; x[3] = a.w; x[4] = b.x; x[5] = b.y; x[6] = b.z; x[7] = b.w;
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

// ck_mode 0: s_ = checkpoints of wkv7_fwd.hip (every 16 steps: H at the start of chunk c is entry 2c-1, at its end 2c+1);
// ck_mode 1: s_ = hs of wkv7_chunk_fwd.hip (state at the START of every 32-step chunk: entries c and c+1).  [k][v] both.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void wkv7c_bwd_out_kernel(
    int T_, int H, int nchunks_total, int ck_mode, const bf16_t *__restrict__ w_, const bf16_t *__restrict__ q_, const bf16_t *__restrict__ k_,
    const bf16_t *__restrict__ v_, const bf16_t *__restrict__ a_, const bf16_t *__restrict__ b_, const bf16_t *__restrict__ dy_,
    const float *__restrict__ s_, const float *__restrict__ sa_, const float *__restrict__ tinv_, const float *__restrict__ e_kv, bf16_t *__restrict__ dw_, bf16_t *__restrict__ dq_, bf16_t *__restrict__ dk_,
    bf16_t *__restrict__ dv_, bf16_t *__restrict__ da_, bf16_t *__restrict__ db_) {
    extern __shared__ __attribute__((aligned(16))) uint16_t sm[];
    using L = OutSmem;
    float *sh_gC = reinterpret_cast<float *>(sm + L::gC), *sh_dterm = reinterpret_cast<float *>(sm + L::dterm);
    const int nc = T_ / kC;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int pt = tid & 31, pk = (tid >> 5) * 8;   // compute mapping: step pt, channels pk .. pk+7
    const int lt = tid >> 3, lk = (tid & 7) * 8;    // global-memory mapping: step lt, channels lk .. lk+7
    const long tstride = (long)H * kN;
    const long nck = T_ / kChunk;  // scalar-forward checkpoints (every 16 steps), s[b,h,n][k][v]

    // Everything a chunk reads from global memory is requested one chunk ahead: the raw rows at the start of the previous
    // chunk, the three 64x64 fp32 matrices (48 registers) only after its phase D -- held any longer they push the phase A-D
    // working set into AGPR spills (30 % of the kernel's VALU instructions were v_accvgpr moves).
    struct Rows {
        Raw8 w, q, k, a, b, v, dy;
        float4 u0, u1, tm;
        long off;
    };
    struct Mats {
        float4 ekv[4], hc[4];  // 64x64 fp32: piece p = tid + 256 i = row p >> 4, columns 4 (p & 15) .. +4
    };
    auto load_rows = [&](int chunk) {
        const int bh = chunk / nc, c = chunk - bh * nc;
        const int bb = bh / H, hh = bh - bb * H;
        Rows r;
        // row-contiguous mapping for global memory (8 lanes = the 64 channels of one step); the compute mapping (step index
        // across lanes) is reached through the LDS restaging below, and left the same way for the six outputs
        r.off = ((long)bb * T_ * H + hh) * kN + (long)(c * kC + lt) * tstride + lk;
        r.w = ld8(w_ + r.off); r.q = ld8(q_ + r.off); r.k = ld8(k_ + r.off); r.a = ld8(a_ + r.off); r.b = ld8(b_ + r.off);
        r.v = ld8(v_ + r.off); r.dy = ld8(dy_ + r.off);
        r.u0 = *reinterpret_cast<const float4 *>(sa_ + r.off);
        r.u1 = *reinterpret_cast<const float4 *>(sa_ + r.off + 4);
        r.tm = *reinterpret_cast<const float4 *>(tinv_ + (long)chunk * kC * kC + tid * 4);
        return r;
    };
    const long n_ck = ck_mode ? nc : nck;
    // H at the start of a chunk (ck_mode 0: checkpoint 2c-1; 1: hs entry c)
    auto load_h0 = [&](int chunk, float4 (&h0)[4]) {
        const int bh = chunk / nc, c = chunk - bh * nc;
        const int i0 = ck_mode ? c : 2 * c - 1;
        const bool has0 = ck_mode ? true : c > 0;
        const float *h0p = s_ + ((long)bh * n_ck + i0) * kN * kN;
#pragma unroll
        for (int i = 0; i < 4; i++)
            h0[i] = has0 ? *reinterpret_cast<const float4 *>(h0p + (tid + 256 * i) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    auto load_mats = [&](int chunk, Mats &r) {
        const int bh = chunk / nc, c = chunk - bh * nc;
        const float *ekv = e_kv + (long)chunk * kN * kN;
        const int iC = ck_mode ? c + 1 : 2 * c + 1;
        const bool hasC = ck_mode ? c + 1 < nc : true;  // E = 0 after the last chunk: H_C unused
        const float *hcp = s_ + ((long)bh * n_ck + iC) * kN * kN;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int p = tid + 256 * i;
            r.ekv[i] = *reinterpret_cast<const float4 *>(ekv + p * 4);
            r.hc[i] = hasC ? *reinterpret_cast<const float4 *>(hcp + p * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    const int chunk0 = blockIdx.x * kOutChunksPerWG;
    Rows cur = load_rows(chunk0);
    Mats curm;
    load_mats(chunk0, curm);
    for (int ci = 0; ci < kOutChunksPerWG; ci++) {
    const int chunk = chunk0 + ci;
    if (chunk >= nchunks_total) break;
    const bool more = ci + 1 < kOutChunksPerWG && chunk + 1 < nchunks_total;
    BSTAMP_INIT;
    const long off = cur.off;
    Raw8 rw, rq, rk, ra, rb, rv, rdy;
    float4 ru0, ru1;
    {
        // raw rows: global mapping -> LDS (phase scratch, free at this point) -> compute mapping
        uint16_t *rs = sm + L::S;
        float *rsu = reinterpret_cast<float *>(rs + 7 * kC * LDK);
        const int wo = lt * LDK + lk, ro = pt * LDK + pk;
        *reinterpret_cast<uint4 *>(rs + 0 * kC * LDK + wo) = cur.w.r;
        *reinterpret_cast<uint4 *>(rs + 1 * kC * LDK + wo) = cur.q.r;
        *reinterpret_cast<uint4 *>(rs + 2 * kC * LDK + wo) = cur.k.r;
        *reinterpret_cast<uint4 *>(rs + 3 * kC * LDK + wo) = cur.a.r;
        *reinterpret_cast<uint4 *>(rs + 4 * kC * LDK + wo) = cur.b.r;
        *reinterpret_cast<uint4 *>(rs + 5 * kC * LDK + wo) = cur.v.r;
        *reinterpret_cast<uint4 *>(rs + 6 * kC * LDK + wo) = cur.dy.r;
        *reinterpret_cast<float4 *>(rsu + lt * L::kStLD + lk) = cur.u0;
        *reinterpret_cast<float4 *>(rsu + lt * L::kStLD + lk + 4) = cur.u1;
        lds_barrier();
        rw.r = *reinterpret_cast<const uint4 *>(rs + 0 * kC * LDK + ro);
        rq.r = *reinterpret_cast<const uint4 *>(rs + 1 * kC * LDK + ro);
        rk.r = *reinterpret_cast<const uint4 *>(rs + 2 * kC * LDK + ro);
        ra.r = *reinterpret_cast<const uint4 *>(rs + 3 * kC * LDK + ro);
        rb.r = *reinterpret_cast<const uint4 *>(rs + 4 * kC * LDK + ro);
        rv.r = *reinterpret_cast<const uint4 *>(rs + 5 * kC * LDK + ro);
        rdy.r = *reinterpret_cast<const uint4 *>(rs + 6 * kC * LDK + ro);
        ru0 = *reinterpret_cast<const float4 *>(rsu + pt * L::kStLD + pk);
        ru1 = *reinterpret_cast<const float4 *>(rsu + pt * L::kStLD + pk + 4);
    }
    float4 rekv[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        rekv[i] = curm.ekv[i];
        // rowsum(E * H_C): the 16 lanes tid & 15 share a row
        const float part = rekv[i].x * curm.hc[i].x + rekv[i].y * curm.hc[i].y + rekv[i].z * curm.hc[i].z + rekv[i].w * curm.hc[i].w;
        const float t = sum16(part);
        if ((tid & 15) == 0) sh_dterm[(tid + 256 * i) >> 4] = t;
    }
    put_tm<true>(cur.tm, sm + L::TMTh, sm + L::TMTl, tid);

    BSTAMP(0);
    // ---- prologue: decay, scaled operands ----------------------------------------------------------------------------------
    float lw[8], G[8], qv[8], kv[8], av[8], bv[8], gam[8], gprev[8], igam[8];
    cvt8(rw, lw);
#pragma unroll
    for (int j = 0; j < 8; j++) lw[j] = -fast_exp(lw[j]);
    chunk_cumsum(lw, G);
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
#pragma unroll
        for (int j = 0; j < 8; j++) x[j] = av[j] * gprev[j];
        put_row8(sm + L::ATh, sm + L::ATl, pt * LDK + pk, x, hi, lo);
#pragma unroll
        for (int j = 0; j < 8; j++) x[j] = kv[j] * igam[j];
        put_row8(sm + L::KHh, sm + L::KHl, pt * LDK + pk, x, hi, lo);
#pragma unroll
        for (int j = 0; j < 8; j++) x[j] = bv[j] * igam[j];
        put_row8(sm + L::BHh, sm + L::BHl, pt * LDK + pk, x, hi, lo);
        const float u[8] = {ru0.x, ru0.y, ru0.z, ru0.w, ru1.x, ru1.y, ru1.z, ru1.w};
        put_row8(sm + L::Uh, sm + L::Ul, pt * LDK + pk, u, hi, lo);
        *reinterpret_cast<uint4 *>(sm + L::Vp + pt * LDK + pk) = rv.r;     // bf16 inputs are exact: single planes
        *reinterpret_cast<uint4 *>(sm + L::DYp + pt * LDK + pk) = rdy.r;
    }
    lds_barrier();  // sh_gC visible
    {
        // (g_C E)[k][v] planes (row k scaled by g_C[k]); both uses -- B^ (g_C E), K^ (g_C E) contracting over k and
        // V (g_C E)^T, U (g_C E)^T contracting over v -- read this one orientation (k-major via frag_tr / row-major)
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int p = tid + 256 * i, row = p >> 4, c4 = (p & 15) * 4;
            const float g = sh_gC[row];
            put4(sm + L::XTh, sm + L::XTl, row, c4, rekv[i], g, g, g, g);
        }
    }
    lds_barrier();
    BSTAMP(2);
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
    BSTAMP(3);
    // ---- phase B: waves 0,1: G1[s][v] = sum_t A_qb[t][s] dY[t][v] + sum_k b^[s][k] (g_C E)[k][v]  -> G1T[v][s]
    //               waves 2,3: the part of dV that does not need Z: sum_t A_qk[t][s] dY[t][v] + sum_k k^[s][k] (g_C E)[k][v] -------------
    f32x16 accV = zero16();  // waves 2,3: dV tile D[m = s][n = v], finished in phase D
    if (wave <= 1) {
        const int vt = wave;
        f32x16 acc = zero16();
        mma_xs_yeK<kC>(acc, sm + L::QBTh, sm + L::QBTl, LDC, sm + L::DYp, LDK, vt * 32, lane);
        mma_gen<kN, false, true, true, true>(acc, sm + L::BHh, sm + L::BHl, LDK, 0, sm + L::XTh, sm + L::XTl, LDK, vt * 32, lane);
        store_T_split(acc, sm + L::G1Th + vt * 32 * LDC, sm + L::G1Tl + vt * 32 * LDC, LDC, lane);
    } else {
        const int vt = wave - 2;
        mma_xs_yeK<kC>(accV, sm + L::QKTh, sm + L::QKTl, LDC, sm + L::DYp, LDK, vt * 32, lane);
        mma_gen<kN, false, true, true, true>(accV, sm + L::KHh, sm + L::KHl, LDK, 0, sm + L::XTh, sm + L::XTl, LDK, vt * 32, lane);
    }
    lds_barrier();
    BSTAMP(4);
    // ---- phase C: waves 0,1: Z[t][v] = sum_s T[s][t] G1[s][v];  waves 2,3: P_vy, P_uy (need no Z) ---------------------------------
    {
        // P planes: index 0 = P_vy, 1 = P_vz (for dK), 2 = P_uy, 3 = P_uz (for dB); D[m = s][n = t] kept for s >= t (s > t), stored [t][s]
        if (wave <= 1) {
            const int vt = wave;
            f32x16 acc = zero16();  // D[m = v][n = t] -> stored as Z[t][v]
            mma_tile3<kC>(acc, sm + L::G1Th + vt * 32 * LDC, sm + L::G1Tl + vt * 32 * LDC, LDC, sm + L::TMTh, sm + L::TMTl, LDC, lane);
            store_T_split(acc, sm + L::Zh + vt * 32, sm + L::Zl + vt * 32, LDK, lane);
        } else {
            f32x16 acc = zero16();
            if (wave == 2) mma_ee<kN>(acc, sm + L::DYp, LDK, sm + L::Vp, LDK, lane);                    // dy_s . v_t
            else mma_tile2y<kN>(acc, sm + L::DYp, LDK, sm + L::Uh, sm + L::Ul, LDK, lane);              // dy_s . u_t
            mask_upper_T<false>(acc, lane);
            uint16_t *Ph = sm + L::P0 + (wave == 2 ? 0 : 2) * 2 * L::A1;
            store_T_split(acc, Ph, Ph + L::A1, LDC, lane);
        }
    }
    lds_barrier();
    BSTAMP(5);
    // ---- phase D: waves 2,3: dV += sum_t A_ak[t][s] Z[t][v] -> staging;  waves 0,1: P_vz, P_uz ------------------------------------
    if (wave >= 2) {
        const int vt = wave - 2;
        mma_tile3_yK<kC>(accV, sm + L::AKTh, sm + L::AKTl, LDC, sm + L::Zh, sm + L::Zl, LDK, vt * 32, lane);
        stage_tile(accV, reinterpret_cast<float *>(sm + L::sV), vt, lane);
    } else {
        f32x16 acc = zero16();
        if (wave == 0) mma_xs_ye<kN>(acc, sm + L::Zh, sm + L::Zl, LDK, sm + L::Vp, LDK, lane);         // z_s . v_t
        else mma_tile3<kN>(acc, sm + L::Zh, sm + L::Zl, LDK, sm + L::Uh, sm + L::Ul, LDK, lane);        // z_s . u_t
        mask_upper_T<true>(acc, lane);
        uint16_t *Ph = sm + L::P0 + (wave == 0 ? 1 : 3) * 2 * L::A1;
        store_T_split(acc, Ph, Ph + L::A1, LDC, lane);
    }
    lds_barrier();
    BSTAMP(6);
    // H0 of this chunk for phase E2 (an L2 hit: the previous chunk read it as its H_C); nothing else is held across the
    // product phases -- every register carried through them is one the compiler cannot use to keep a product's fragment
    // loads in flight (with 40 + 48 prefetch registers live it issued them four at a time between dependent MFMAs)
    float4 h0e[4];
    load_h0(chunk, h0e);
    BSTAMP(7);
    // ---- phase F1: dV out; dK (waves 0,1) and dB (waves 2,3), unscaled, to staging -------------------------------------------------
    float dVv[8];  // written out with the other gradients in the epilogue
    {
        ld_stage8(reinterpret_cast<const float *>(sm + L::sV), pt, pk, dVv);
        const int kt = wave & 1;
        const uint16_t *P1h = sm + L::P0 + (wave < 2 ? 0 : 2) * 2 * L::A1, *P1l = P1h + L::A1, *P2h = P1h + 2 * L::A1, *P2l = P2h + L::A1;
        f32x16 acc = zero16();  // D[m = t][n = k]
        mma_tile3_yK<kC>(acc, P1h, P1l, LDC, sm + L::QTh, sm + L::QTl, LDK, kt * 32, lane);
        mma_tile3_yK<kC>(acc, P2h, P2l, LDC, sm + L::ATh, sm + L::ATl, LDK, kt * 32, lane);
        if (wave < 2) {
            mma_tile2y<kN>(acc, sm + L::Vp, LDK, sm + L::XTh + kt * 32 * LDK, sm + L::XTl + kt * 32 * LDK, LDK, lane);
            stage_tile(acc, reinterpret_cast<float *>(sm + L::sK), kt, lane);
        } else {
            mma_tile3<kN>(acc, sm + L::Uh, sm + L::Ul, LDK, sm + L::XTh + kt * 32 * LDK, sm + L::XTl + kt * 32 * LDK, LDK, lane);
            stage_tile(acc, reinterpret_cast<float *>(sm + L::sB), kt, lane);
        }
    }
    lds_barrier();
    BSTAMP(8);
    // ---- phase E2: the four transposed P matrices of dQ / dA; H0^T planes over (g_C E)^T --------------------------------------------
    {
        uint16_t *Ph = sm + L::P0 + wave * 2 * L::A1, *Pl = Ph + L::A1;
        f32x16 acc = zero16();  // D[m = s][n = t'], kept for s <= t' (vy, uy) or s < t' (vz, uz); stored [t'][s]
        if (wave == 0) mma_ee<kN>(acc, sm + L::Vp, LDK, sm + L::DYp, LDK, lane);                        // v_s . dy_t'
        else if (wave == 1) mma_xs_ye<kN>(acc, sm + L::Uh, sm + L::Ul, LDK, sm + L::DYp, LDK, lane);    // u_s . dy_t'
        else if (wave == 2) mma_tile2y<kN>(acc, sm + L::Vp, LDK, sm + L::Zh, sm + L::Zl, LDK, lane);    // v_s . z_t'
        else mma_tile3<kN>(acc, sm + L::Uh, sm + L::Ul, LDK, sm + L::Zh, sm + L::Zl, LDK, lane);        // u_s . z_t'
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int p = tid + 256 * i, row = p >> 4, c4 = (p & 15) * 4;
            put4(sm + L::XTh, sm + L::XTl, row, c4, h0e[i], 1.f, 1.f, 1.f, 1.f);
        }
        if (wave <= 1) mask_lower_T<false>(acc, lane);
        else mask_lower_T<true>(acc, lane);
        store_T_split(acc, Ph, Pl, LDC, lane);
    }
    lds_barrier();
    BSTAMP(9);
    // the next chunk's raw rows: requested two phases (~4k cycles) ahead of their use
    Rows nxt = cur;
    if (more) nxt = load_rows(chunk + 1);
    // ---- phase F2: dQ (waves 0,1) and dA (waves 2,3), unscaled, to staging ---------------------------------------------------------
    {
        const int kt = wave & 1;
        // dQ uses P planes 0 (vy) and 1 (uy); dA uses 2 (vz) and 3 (uz)
        const uint16_t *P1h = sm + L::P0 + (wave < 2 ? 0 : 2) * 2 * L::A1, *P1l = P1h + L::A1, *P2h = P1h + 2 * L::A1, *P2l = P2h + L::A1;
        f32x16 acc = zero16();  // D[m = t'][n = k]
        mma_tile3_yK<kC>(acc, P1h, P1l, LDC, sm + L::KHh, sm + L::KHl, LDK, kt * 32, lane);
        mma_tile3_yK<kC>(acc, P2h, P2l, LDC, sm + L::BHh, sm + L::BHl, LDK, kt * 32, lane);
        if (wave < 2) {
            mma_tile2y<kN>(acc, sm + L::DYp, LDK, sm + L::XTh + kt * 32 * LDK, sm + L::XTl + kt * 32 * LDK, LDK, lane);
            stage_tile(acc, reinterpret_cast<float *>(sm + L::sQ), kt, lane);
        } else {
            mma_tile3<kN>(acc, sm + L::Zh, sm + L::Zl, LDK, sm + L::XTh + kt * 32 * LDK, sm + L::XTl + kt * 32 * LDK, LDK, lane);
            stage_tile(acc, reinterpret_cast<float *>(sm + L::sA), kt, lane);
        }
    }
    lds_barrier();
    BSTAMP(10);
    // the next chunk's E and H_C (needed ~1k cycles into its prologue)
    Mats nxtm = curm;
    if (more) load_mats(chunk + 1, nxtm);
    // ---- epilogue: decay scaling, decay gradient, stores ----------------------------------------------------------------------------
    float dQ[8], dK[8], dB[8], dA[8], e[8];
    ld_stage8(reinterpret_cast<const float *>(sm + L::sQ), pt, pk, dQ);
    ld_stage8(reinterpret_cast<const float *>(sm + L::sK), pt, pk, dK);
    ld_stage8(reinterpret_cast<const float *>(sm + L::sB), pt, pk, dB);
    ld_stage8(reinterpret_cast<const float *>(sm + L::sA), pt, pk, dA);
#pragma unroll
    for (int j = 0; j < 8; j++) {
        dQ[j] *= gam[j];
        dK[j] *= igam[j];
        dB[j] *= igam[j];
        dA[j] *= gprev[j];
        // e_t = (q dQ - k dK - b dB)_t + (a dA)_{t+1}
        e[j] = qv[j] * dQ[j] - kv[j] * dK[j] - bv[j] * dB[j] + next32(av[j] * dA[j], lane);
    }
    float dG[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
        // dlw_t = sum_{s >= t} e_s + rowsum(E * H_C) = total - (inclusive prefix - e_t) + dterm ;  dw = dlw * lw
        const float pre = scan32(e[j]);
        dG[j] = (last32(pre, lane) - pre + e[j] + sh_dterm[pk + j]) * lw[j];
    }
    {
        // the six gradients: compute mapping -> bf16 rows in LDS (phase scratch is free now) -> row-contiguous stores
        uint16_t *os = sm + L::S;
        auto put = [&](int i, const float (&x)[8]) {
            uint4 o;
            o.x = cvt_pk(x[0], x[1]); o.y = cvt_pk(x[2], x[3]); o.z = cvt_pk(x[4], x[5]); o.w = cvt_pk(x[6], x[7]);
            *reinterpret_cast<uint4 *>(os + i * kC * LDK + pt * LDK + pk) = o;
        };
        put(0, dG); put(1, dQ); put(2, dK); put(3, dVv); put(4, dA); put(5, dB);
        lds_barrier();
        bf16_t *const outs[6] = {dw_, dq_, dk_, dv_, da_, db_};
#pragma unroll
        for (int i = 0; i < 6; i++)
            *reinterpret_cast<uint4 *>(outs[i] + off) = *reinterpret_cast<const uint4 *>(os + i * kC * LDK + lt * LDK + lk);
    }
    BSTAMP(11);
    lds_barrier();  // the next chunk's prologue overwrites what the epilogue reads
    cur = nxt;
    curm = nxtm;
    }  // chunk loop
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
    const int total = B * H * (T_ / kC);
    hipLaunchKernelGGL(wkv7c_bwd_pre_kernel, dim3((total + kChunksPerWG - 1) / kChunksPerWG), dim3(256), PreSmem::bytes, st, T_, H,
                       total, (const bf16_t *)w, (const bf16_t *)q, (const bf16_t *)a, (const bf16_t *)b, (const bf16_t *)dy, tinv,
                       (uint16_t *)mt, np);
    return (int)hipGetLastError();
}

int chunk_state_bf16(int BH, int nc, int H, const void *mt, const float *np, float *e_kv, const int *seq_off, int nseq,
                     hipStream_t st) {
    static bool attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&wkv7c_state_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)StateSmem::bytes);
        if (e != hipSuccess) return (int)e;
        attr = true;
    }
    (void)hipGetLastError();
    hipLaunchKernelGGL(wkv7c_state_kernel, dim3((seq_off ? nseq * H : BH) * 2), dim3(128), StateSmem::bytes, st, nc, H, (const uint16_t *)mt,
                       np, e_kv, seq_off);
    return (int)hipGetLastError();
}

int chunk_bwd_out_bf16(int B, int T_, int H, int ck_mode, const void *w, const void *q, const void *k, const void *v, const void *a,
                       const void *b, const void *dy, const float *s, const float *sa, const float *tinv, const float *e_kv, void *dw, void *dq, void *dk, void *dv, void *da, void *db, hipStream_t st) {
    static bool attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&wkv7c_bwd_out_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)OutSmem::bytes);
        if (e != hipSuccess) return (int)e;
        attr = true;
    }
    (void)hipGetLastError();
    const int total = B * H * (T_ / kC);
    hipLaunchKernelGGL(wkv7c_bwd_out_kernel, dim3((total + kOutChunksPerWG - 1) / kOutChunksPerWG), dim3(256), OutSmem::bytes, st, T_, H,
                       total, ck_mode, (const bf16_t *)w,
                       (const bf16_t *)q, (const bf16_t *)k, (const bf16_t *)v, (const bf16_t *)a, (const bf16_t *)b,
                       (const bf16_t *)dy, s, sa, tinv, e_kv, (bf16_t *)dw, (bf16_t *)dq, (bf16_t *)dk, (bf16_t *)dv,
                       (bf16_t *)da, (bf16_t *)db);
    return (int)hipGetLastError();
}

#ifdef WKV7C_TIMING
extern "C" int rwkv7_debug_cbwd_timing(long long *out, int reset) {
    if (reset) {
        long long z[128] = {0};
        return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_cbwd_timing), z, sizeof(z));
    }
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_cbwd_timing), sizeof(long long) * 128);
}
#endif

}  // namespace rwkv7
