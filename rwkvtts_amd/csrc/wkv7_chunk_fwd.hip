// rwkvtts_amd/csrc/wkv7_chunk_fwd.hip -- chunked (MFMA) WKV7 forward for gfx950: the training fast path.
//
// Same operator as wkv7_fwd.hip (reference wkv7_cuda.cu:10-52) but evaluated 32 steps at a time as small matrix
// products on the matrix cores instead of 64x64 scalar FMAs per step -- see chunk_common.h for the algebra.
// The scalar recurrence needs ~9*64^2 flop per token-head at the fp32 VALU rate (157 TF/s chip-wide) and its
// per-step dependency chain leaves one wave per SIMD latency-bound; here the only sequential object is the
// 64x64 state between chunks, everything else is MFMA work and the part that does not depend on the state
// (the 32x32 triangular inverse) is hoisted into a fully parallel kernel.
//
//   wkv7c_prep_kernel : grid B*H*(T/32), one wave per chunk: a~, b^ -> A_ab -> T = (I - A_ab)^-1 (fp32, row-wise
//                       back substitution with v_readlane scalars) -> Tinv[b,h,c,32,32]   (also used by backward)
//   wkv7c_fwd_kernel  : grid B*H*2 (two workgroups per head, 32 value columns each -- value columns never
//                       interact in the forward), 4 waves; per chunk: decay cumsum + operand scaling into bf16
//                       hi/lo planes -> A_ak, A_qb, A_qk (3 waves) + T planes (4th) -> R -> U -> Y | state update.
// Saved for backward: U (= the scalar kernel's `sa`, fp32 [B,T,H,64]) and the state at the START of every chunk,
// hs fp32 [B,H,T/32,64(k),64(v)].
#include "chunk_common.h"
#include "launch_attr.h"

namespace rwkv7 {

#ifdef WKV7C_TIMING
// profiling build only (python -m rwkvtts_amd.build --timing): per-phase cycle totals of block 0, per wave
__device__ long long g_chunk_timing[4 * 16];
#define TSTAMP(i)                                                                  \
    do {                                                                           \
        const long long now_ = __builtin_readcyclecounter();                       \
        if (blockIdx.x == 0 && lane == 0) g_chunk_timing[wave * 16 + (i)] += now_ - tprev_; \
        tprev_ = now_;                                                             \
    } while (0)
#else
#define TSTAMP(i) do { } while (0)
#endif

template <typename T>
__device__ __forceinline__ float ld_scalar(const T *p);
template <>
__device__ __forceinline__ float ld_scalar<bf16_t>(const bf16_t *p) { return bf2f(p->x); }
template <>
__device__ __forceinline__ float ld_scalar<float>(const float *p) { return *p; }

// --------------------------------------------------------------------------------------------------------------
// T = (I - A_ab)^-1 per chunk
// --------------------------------------------------------------------------------------------------------------
// Round 5: the contraction over the 64 channels runs in TWO passes of 32 (lane = channel 32 p + (lane & 31), time half lane >> 5:
// every lane walks 16 steps per pass, the upper half starting from its twin's decay sum), so the operand planes are [32][32 + 8]
// instead of [32][64 + 8]: 10 KB of LDS per wave instead of 18 and 48 instead of 96 prefetched values per lane -- the kernel is a
// chain of latencies (two serial scans, 0.18 instructions per cycle and SIMD at the two waves per SIMD that 18 KB allowed), and its
// throughput is the number of chunks in flight per CU: 16 waves instead of 8.  Same instruction count per chunk.
template <typename T>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) void wkv7c_prep_kernel(int T_, int H, const T *__restrict__ w_, const T *__restrict__ a_,
                                                        const T *__restrict__ b_, float *__restrict__ tinv_) {
    constexpr int LD = kC + kPad;   // 32 channels of a pass + padding
    __shared__ __attribute__((aligned(16))) uint16_t planes[4 * kC * LD];
    uint16_t *ATh = planes, *ATl = planes + kC * LD, *BHh = planes + 2 * kC * LD, *BHl = planes + 3 * kC * LD;
    static_assert(2 * kC * LD * 2 >= kC * 2 * 16 * 4, "the back substitution's A^T copy must fit over the a~ planes");
    const int nc = T_ / kC;
    const int bh = blockIdx.x / nc, c = blockIdx.x - bh * nc;
    const int bb = bh / H, hh = bh - bb * H;
    const int lane = threadIdx.x;
    const int kc = lane & 31, th = lane >> 5;
    const long tstride = (long)H * kN;
    const long base = (((long)bb * T_ + (long)c * kC + 16 * th) * H + hh) * kN + kc;
    // both passes' loads are requested before the first use (2 x 48 per lane): the kernel pays one HBM latency
    float wv[2][16], av_[2][16], bv_[2][16];
#pragma unroll
    for (int p = 0; p < 2; p++)
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const long idx = base + 32 * p + j * tstride;
            wv[p][j] = ld_scalar<T>(w_ + idx);
            av_[p][j] = ld_scalar<T>(a_ + idx);
            bv_[p][j] = ld_scalar<T>(b_ + idx);
        }
    f32x16 acc = zero16();
#pragma unroll
    for (int p = 0; p < 2; p++) {
        float lw[16], tot = 0.f;
#pragma unroll
        for (int j = 0; j < 16; j++) {
            lw[j] = -fast_exp(wv[p][j]);
            tot += lw[j];
        }
        // steps 16..31 start from the decay sum of steps 0..15 of the same channel: the twin lane's total
        const float first = __shfl(tot, kc);
        float G = th ? first : 0.f;
        if (p == 1) __syncthreads();   // pass 0's fragments have been read (one wave: LDS is in order; the barrier is for the compiler)
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const int t = 16 * th + j;
            const float at = av_[p][j] * fast_exp(G);  // a * gamma_{t-1}
            G += lw[j];
            const float bh_ = bv_[p][j] * fast_exp(-G);  // b / gamma_t
            split2(at, ATh[t * LD + kc], ATl[t * LD + kc]);
            split2(bh_, BHh[t * LD + kc], BHl[t * LD + kc]);
        }
        __syncthreads();
        mma_tile3<kC>(acc, BHh, BHl, LD, ATh, ATl, LD, lane);  // D[m = s][n = t] += b^_s . a~_t over this pass's 32 channels = A_ab[t][s]
    }
    mask_lower_T<true>(acc, lane);
    // T = I + T A  =>  T[t][r] = delta(t,r) + sum_{q>r} T[t][q] A[q][r], r = 31..0, row t on lane t AND its twin t + 32: the twins
    // split the sum by the parity of q (round 4; before, both computed the whole row with A[q][r] fetched by v_readlane: 496 readlane
    // + 496 fma per lane, half the wave redundant -- 1k of the kernel's 2.1k instructions).  A goes through LDS, transposed and
    // parity-split, AtP[r][h][i] = A[2 i + h][r] (over the dead operand planes: same wave, LDS is in order), so that the 16-byte read
    // of four consecutive i is a two-address broadcast; lane (t, h) keeps Tq[i] = T[t][2 i + h].  272 fma + 80 reads + 32 exchanges.
    const int t = lane & 31, h = lane >> 5;
    float *AtP = reinterpret_cast<float *>(ATh);   // [32][2][16] floats = 4 KB
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int s0 = (r & 3) + 8 * (r >> 2) + 4 * h;   // acc[r] = A[t][s0]
        AtP[(s0 * 2 + (t & 1)) * 16 + (t >> 1)] = acc[r];
    }
    float Tq[16];
#pragma unroll
    for (int i = 0; i < 16; i++) Tq[i] = 0.f;
#pragma unroll
    for (int r = kC - 1; r >= 0; r--) {
        // terms i >= r / 2 (for even r and h = 0 that includes q = r: A[r][r] = 0 times the still-zero Tq[r / 2])
        const int i0 = r >> 1;
        float s0 = 0.f, s1 = 0.f;
        const float *ap = AtP + (r * 2 + h) * 16;
        // oldest columns first: the term with the column finished one step ago is the LAST link of its chain, not the first
#pragma unroll
        for (int i4 = 12; i4 >= (i0 & ~3); i4 -= 4) {
            const float4 a4 = *reinterpret_cast<const float4 *>(ap + i4);
            if (i4 + 3 >= i0) s1 = fmaf(Tq[i4 + 3], a4.w, s1);
            if (i4 + 2 >= i0) s0 = fmaf(Tq[i4 + 2], a4.z, s0);
            if (i4 + 1 >= i0) s1 = fmaf(Tq[i4 + 1], a4.y, s1);
            if (i4 + 0 >= i0) s0 = fmaf(Tq[i4 + 0], a4.x, s0);
        }
        const float part = s0 + s1;
        const float tot = part + __shfl_xor(part, 32) + ((t == r) ? 1.f : 0.f);
        Tq[i0] = ((r & 1) == h) ? tot : Tq[i0];
    }
    // lane (t, 0) stores columns 0..15 of its row, its twin 16..31: the twins exchange the halves they do not store
    float *out = tinv_ + ((long)blockIdx.x * kC + t) * kC + 16 * h;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        float c[4];
#pragma unroll
        for (int e = 0; e < 2; e++) {
            const int i = 2 * j + e;
            // h = 0 holds T[t][2 i] (keeps) and T[t][16 + 2 i] (gives); h = 1 holds T[t][2 i + 1] (gives) and T[t][17 + 2 i] (keeps)
            const float got = __shfl_xor(h == 0 ? Tq[8 + i] : Tq[i], 32);
            c[2 * e] = h == 0 ? Tq[i] : got;
            c[2 * e + 1] = h == 0 ? got : Tq[8 + i];
        }
        *reinterpret_cast<float4 *>(out + 4 * j) = make_float4(c[0], c[1], c[2], c[3]);
    }
}

// --------------------------------------------------------------------------------------------------------------
// forward
// --------------------------------------------------------------------------------------------------------------
namespace {
constexpr int LDK = kN + kPad;  // planes with K = 64 columns
constexpr int LDC = kC + kPad;  // planes with K = 32 columns
constexpr int VH = 32;          // value columns per workgroup

struct FwdSmem {  // all offsets in uint16 units; every plane 16-byte aligned
    static constexpr int QTh = 0, QTl = QTh + kC * LDK, ATh = QTl + kC * LDK, ATl = ATh + kC * LDK;
    static constexpr int KHh = ATl + kC * LDK, KHl = KHh + kC * LDK, BHh = KHl + kC * LDK, BHl = BHh + kC * LDK;
    // V[t][v] time-major like the other operands; products that contract over time read k^, b^, v with LDS transpose
    // reads (frag_tr) instead of keeping channel-major copies
    static constexpr int Vt = BHl + kC * LDK, Vtl = Vt + kC * LDC;  // Vtl: low part, fp32 inputs only
    static constexpr int Sh = Vtl + VH * LDC, Sl = Sh + VH * LDK;
    static constexpr int AKh = Sl + VH * LDK, AKl = AKh + kC * LDC, QBh = AKl + kC * LDC, QBl = QBh + kC * LDC;
    static constexpr int QKh = QBl + kC * LDC, QKl = QKh + kC * LDC, TMh = QKl + kC * LDC, TMl = TMh + kC * LDC;
    static constexpr int Rh = TMl + kC * LDC, Rl = Rh + VH * LDC, Uh = Rl + VH * LDC, Ul = Uh + VH * LDC;
    static constexpr int end16 = Ul + VH * LDC;
    // fp32 region (offsets in floats from the start of the fp32 area)
    static constexpr int fStage = 0, fGC = fStage + 2 * kC * 36, fend = fGC + kN;   // U, Y staging tiles [32][36]; g_C
    static constexpr size_t bytes = (size_t)end16 * 2 + (size_t)fend * 4;
    // + the raw input staging area (element type dependent)
    template <typename T>
    static constexpr size_t total() {
        return bytes + (size_t)(5 * kC * (kN + 16 / sizeof(T)) + kC * (VH + 16 / sizeof(T))) * sizeof(T);
    }
};
static_assert(FwdSmem::end16 % 8 == 0, "fp32 area must stay 16-byte aligned");
}  // namespace

template <typename T, bool SAVE>
__global__ __launch_bounds__(256) void wkv7c_fwd_kernel(int T_, int H, const T *__restrict__ w_, const T *__restrict__ q_,
                                                        const T *__restrict__ k_, const T *__restrict__ v_,
                                                        const T *__restrict__ a_, const T *__restrict__ b_,
                                                        const float *__restrict__ tinv_, T *__restrict__ y_,
                                                        float *__restrict__ sa_, uint16_t *__restrict__ hs_,
                                                        const int *__restrict__ seq_off_) {
    extern __shared__ __attribute__((aligned(16))) uint16_t sm[];
    float *fm = reinterpret_cast<float *>(sm + FwdSmem::end16);
    constexpr int kStageLD = 36;  // fp32 staging tiles [32][36]: conflict-free float4 reads with the step index across lanes
    float *sh_U = fm + FwdSmem::fStage, *sh_Y = sh_U + kC * kStageLD, *sh_gC = fm + FwdSmem::fGC;
    static_assert(2 * kC * kStageLD == FwdSmem::fGC - FwdSmem::fStage, "staging tiles");
    using L = FwdSmem;
    constexpr bool VEXACT = sizeof(T) == 2;  // bf16 tensors: v needs no hi/lo split
    // acc += X V^T-plane product with X split; V exact (bf16 I/O) or split (fp32 I/O)
    auto mma_xv = [&](f32x16 &acc, const uint16_t *Xh, const uint16_t *Xl, int lane_) {  // X[t][s] . V[s][v]
        mma_gen<kC, false, true, true, !VEXACT>(acc, Xh, Xl, LDC, 0, sm + L::Vt, sm + L::Vtl, LDC, 0, lane_);
    };

    // Workgroups are dealt round-robin to the 8 XCDs (each with its own L2): the two value halves of a head get block ids
    // g and g + 8, so they run on the same XCD and the 5 shared input streams come from HBM once.
    int vh, bh;
    if ((gridDim.x & 15) == 0) {
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        bh = (j >> 1) * 8 + xcd;
        vh = j & 1;
    } else {
        vh = blockIdx.x & 1;
        bh = blockIdx.x >> 1;
    }
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int nc = T_ / kC;
    // Packed rows (fla chunk_rwkv7's cu_seqlens): seq_off_[s] .. seq_off_[s+1] is the range of 32-step chunks (counted over the
    // whole [B][T/32] chunk space) of sequence s; the grid then has one workgroup pair per (sequence, head), each starting
    // from the zero state -- sequences of one row run in parallel instead of one after the other.
    int bb, hh, c0 = 0, c1 = nc;
    if (seq_off_) {
        const int sq = bh / H;
        hh = bh - sq * H;
        const int g0 = seq_off_[sq], g1 = seq_off_[sq + 1];
        bb = g0 / nc;
        c0 = g0 - bb * nc;
        c1 = c0 + (g1 - g0);
        bh = bb * H + hh;
        if (c1 <= c0) return;
    } else {
        bb = bh / H;
        hh = bh - bb * H;
    }
    const long tstride = (long)H * kN;
    const long head_base = ((long)bb * T_ * H + hh) * kN;

    // phase-1/2 roles: one time step x 8 channels per thread.  The time step is the FAST lane index so that the
    // channel-major (transposed) plane stores of a wave land on consecutive 2-byte addresses (conflict-free);
    // with the channel group fastest, the 8 groups alias onto one bank (any 16-B-aligned row stride x 8 rows is
    // a multiple of 128 B).
    const int pt = tid & 31, pk = (tid >> 5) * 8;
    const int pv = (tid >> 5) * 4;  // 4 value columns of this workgroup's half

    // zero the state planes (chunk 0 starts from S = 0)
    for (int i = tid; i < 2 * VH * LDK; i += 256) sm[L::Sh + i] = 0;

    f32x16 Smaster = zero16();  // waves 1,2: D-layout tile [k-tile rows][v cols] of the fp32 state

    // Global loads use a row-contiguous mapping (8 lanes cover the 64 channels of one step: every wave instruction touches
    // 8 rows), the compute mapping has the step index across lanes (32 rows per instruction, which kept the address unit
    // busy for ~2k cycles per chunk).  The raw rows pass through an LDS staging area to change mapping.
    constexpr int RS = kN + 16 / (int)sizeof(T), RSV = VH + 16 / (int)sizeof(T);  // padded row strides (elements)
    T *raw = reinterpret_cast<T *>(fm + FwdSmem::fend);                             // [5][32][RS] then [32][RSV]
    const int lt = tid >> 3, lk = (tid & 7) * 8, lv = (tid & 7) * 4;
    Raw4<T> gw[2], gq[2], gk[2], ga[2], gb[2], gv;
    auto issue = [&](int c) {
        const long off = head_base + (long)(c * kC + lt) * tstride;
#pragma unroll
        for (int i = 0; i < 2; i++) {
            gw[i] = ld4<T>(w_ + off + lk + 4 * i, true);
            gq[i] = ld4<T>(q_ + off + lk + 4 * i, true);
            gk[i] = ld4<T>(k_ + off + lk + 4 * i, true);
            ga[i] = ld4<T>(a_ + off + lk + 4 * i, true);
            gb[i] = ld4<T>(b_ + off + lk + 4 * i, true);
        }
        gv = ld4<T>(v_ + off + vh * VH + lv, true);
    };
    using RawVec = decltype(Raw4<T>::r);
    Raw4<T> rw[2], rq[2], rk[2], ra[2], rb[2], rv;
    auto restage = [&]() {  // load mapping -> LDS -> compute mapping (one barrier)
#pragma unroll
        for (int i = 0; i < 2; i++) {
            *reinterpret_cast<RawVec *>(raw + (0 * kC + lt) * RS + lk + 4 * i) = gw[i].r;
            *reinterpret_cast<RawVec *>(raw + (1 * kC + lt) * RS + lk + 4 * i) = gq[i].r;
            *reinterpret_cast<RawVec *>(raw + (2 * kC + lt) * RS + lk + 4 * i) = gk[i].r;
            *reinterpret_cast<RawVec *>(raw + (3 * kC + lt) * RS + lk + 4 * i) = ga[i].r;
            *reinterpret_cast<RawVec *>(raw + (4 * kC + lt) * RS + lk + 4 * i) = gb[i].r;
        }
        *reinterpret_cast<RawVec *>(raw + 5 * kC * RS + lt * RSV + lv) = gv.r;
        lds_barrier();
#pragma unroll
        for (int i = 0; i < 2; i++) {
            rw[i].r = *reinterpret_cast<const RawVec *>(raw + (0 * kC + pt) * RS + pk + 4 * i);
            rq[i].r = *reinterpret_cast<const RawVec *>(raw + (1 * kC + pt) * RS + pk + 4 * i);
            rk[i].r = *reinterpret_cast<const RawVec *>(raw + (2 * kC + pt) * RS + pk + 4 * i);
            ra[i].r = *reinterpret_cast<const RawVec *>(raw + (3 * kC + pt) * RS + pk + 4 * i);
            rb[i].r = *reinterpret_cast<const RawVec *>(raw + (4 * kC + pt) * RS + pk + 4 * i);
        }
        rv.r = *reinterpret_cast<const RawVec *>(raw + 5 * kC * RS + pt * RSV + pv);
    };
    issue(c0);
    lds_barrier();

#ifdef WKV7C_TIMING
    long long tprev_ = __builtin_readcyclecounter();
#endif
    for (int c = c0; c < c1; c++) {
        TSTAMP(0);
        restage();
        if (c + 1 < c1) issue(c + 1);  // next chunk's raw inputs fly during the whole chunk
        // ---- phase 1: log-decay and its cumulative sum over the chunk ---------------------------------------
        float lw[8], qv[8], kv[8], av[8], bv[8], vv[4];
        {
            const float4 w0 = cvt4(rw[0]), w1 = cvt4(rw[1]);
            const float wr[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
            for (int j = 0; j < 8; j++) lw[j] = -fast_exp(wr[j]);
            const float4 q0 = cvt4(rq[0]), q1 = cvt4(rq[1]), k0 = cvt4(rk[0]), k1 = cvt4(rk[1]);
            const float4 a0 = cvt4(ra[0]), a1 = cvt4(ra[1]), b0 = cvt4(rb[0]), b1 = cvt4(rb[1]), v0 = cvt4(rv);
            qv[0] = q0.x; qv[1] = q0.y; qv[2] = q0.z; qv[3] = q0.w; qv[4] = q1.x; qv[5] = q1.y; qv[6] = q1.z; qv[7] = q1.w;
            kv[0] = k0.x; kv[1] = k0.y; kv[2] = k0.z; kv[3] = k0.w; kv[4] = k1.x; kv[5] = k1.y; kv[6] = k1.z; kv[7] = k1.w;
            av[0] = a0.x; av[1] = a0.y; av[2] = a0.z; av[3] = a0.w; av[4] = a1.x; av[5] = a1.y; av[6] = a1.z; av[7] = a1.w;
            bv[0] = b0.x; bv[1] = b0.y; bv[2] = b0.z; bv[3] = b0.w; bv[4] = b1.x; bv[5] = b1.y; bv[6] = b1.z; bv[7] = b1.w;
            vv[0] = v0.x; vv[1] = v0.y; vv[2] = v0.z; vv[3] = v0.w;
        }
        // inclusive cumulative log-decay over the chunk: DPP prefix sum across the 32 lanes that hold the 32 steps
        float Gc[8];
#pragma unroll
        for (int j = 0; j < 8; j++) Gc[j] = scan32(lw[j]);
        TSTAMP(1);
        // ---- phase 2: scaled operands into bf16 hi/lo planes ------------------------------------------------
        {
            float G[8];
#pragma unroll
            for (int j = 0; j < 8; j++) G[j] = Gc[j];
            TSTAMP(12);
            float qs[8], as_[8], ks[8], bs[8];
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const float gam = fast_exp(G[j]), gprev = fast_exp(G[j] - lw[j]), ig = fast_exp(-G[j]);
                qs[j] = qv[j] * gam;
                as_[j] = av[j] * gprev;
                ks[j] = kv[j] * ig;
                bs[j] = bv[j] * ig;
                if (pt == kC - 1) sh_gC[pk + j] = gam;
            }
            uint32_t qh[4], ql[4], ah[4], al[4], kh[4], kl[4], bhh[4], bl[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                split_pk(qs[2 * j], qs[2 * j + 1], qh[j], ql[j]);
                split_pk(as_[2 * j], as_[2 * j + 1], ah[j], al[j]);
                split_pk(ks[2 * j], ks[2 * j + 1], kh[j], kl[j]);
                split_pk(bs[2 * j], bs[2 * j + 1], bhh[j], bl[j]);
            }
            TSTAMP(13);
            auto pack = [](const uint32_t (&x)[4]) { return make_uint4(x[0], x[1], x[2], x[3]); };
            const int o = pt * LDK + pk;
            *reinterpret_cast<uint4 *>(&sm[L::QTh + o]) = pack(qh);
            *reinterpret_cast<uint4 *>(&sm[L::QTl + o]) = pack(ql);
            *reinterpret_cast<uint4 *>(&sm[L::ATh + o]) = pack(ah);
            *reinterpret_cast<uint4 *>(&sm[L::ATl + o]) = pack(al);
            *reinterpret_cast<uint4 *>(&sm[L::KHh + o]) = pack(kh);
            *reinterpret_cast<uint4 *>(&sm[L::KHl + o]) = pack(kl);
            *reinterpret_cast<uint4 *>(&sm[L::BHh + o]) = pack(bhh);
            *reinterpret_cast<uint4 *>(&sm[L::BHl + o]) = pack(bl);
            TSTAMP(14);
            {
                uint32_t h0, l0, h1, l1;
                split_pk(vv[0], vv[1], h0, l0);
                split_pk(vv[2], vv[3], h1, l1);
                *reinterpret_cast<uint2 *>(&sm[L::Vt + pt * LDC + pv]) = make_uint2(h0, h1);
                if (!VEXACT) *reinterpret_cast<uint2 *>(&sm[L::Vtl + pt * LDC + pv]) = make_uint2(l0, l1);  // bf16 v is exact
            }
        }
        TSTAMP(2);
        lds_barrier();
        TSTAMP(3);
        // ---- phase 3: intra-chunk matrices (one per wave) ---------------------------------------------------
        if (wave == 0) {
            f32x16 acc = zero16();  // D[m = s][n = t] = k^_s . a~_t = A_ak[t][s]
            mma_tile3<kN>(acc, sm + L::KHh, sm + L::KHl, LDK, sm + L::ATh, sm + L::ATl, LDK, lane);
            mask_lower_T<true>(acc, lane);
            store_T_split(acc, sm + L::AKh, sm + L::AKl, LDC, lane);
        } else if (wave == 1) {
            f32x16 acc = zero16();  // b^_s . q~_t = A_qb[t][s]
            mma_tile3<kN>(acc, sm + L::BHh, sm + L::BHl, LDK, sm + L::QTh, sm + L::QTl, LDK, lane);
            mask_lower_T<false>(acc, lane);
            store_T_split(acc, sm + L::QBh, sm + L::QBl, LDC, lane);
        } else if (wave == 2) {
            f32x16 acc = zero16();  // k^_s . q~_t = A_qk[t][s]
            mma_tile3<kN>(acc, sm + L::KHh, sm + L::KHl, LDK, sm + L::QTh, sm + L::QTl, LDK, lane);
            mask_lower_T<false>(acc, lane);
            store_T_split(acc, sm + L::QKh, sm + L::QKl, LDC, lane);
        } else {
            // T = (I - A_ab)^-1 of this chunk (wkv7c_prep_kernel), fp32 [32][32] -> planes Tm[t][r]
            const float *tp = tinv_ + ((long)bh * nc + c) * kC * kC;
            const int tr = lane >> 1, tc = (lane & 1) * 16;
            uint32_t hi[8], lo[8];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const float4 x = *reinterpret_cast<const float4 *>(tp + tr * kC + tc + 4 * j);
                split_pk(x.x, x.y, hi[2 * j], lo[2 * j]);
                split_pk(x.z, x.w, hi[2 * j + 1], lo[2 * j + 1]);
            }
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const int o = tr * LDC + tc + 8 * j;
                *reinterpret_cast<uint4 *>(&sm[L::TMh + o]) = make_uint4(hi[4 * j], hi[4 * j + 1], hi[4 * j + 2], hi[4 * j + 3]);
                *reinterpret_cast<uint4 *>(&sm[L::TMl + o]) = make_uint4(lo[4 * j], lo[4 * j + 1], lo[4 * j + 2], lo[4 * j + 3]);
            }
        }
        TSTAMP(4);
        lds_barrier();
        TSTAMP(5);
        // ---- phase 4: R = A~ H0 + A_ak V   (D[t][v]) ---------------------------------------------------------
        f32x16 accY = zero16();  // wave 3: the part of Y that does not need U (Q~ H0 + A_qk V), finished in phase 6
        if (wave == 0) {
            f32x16 acc = zero16();
            mma_tile3<kN>(acc, sm + L::ATh, sm + L::ATl, LDK, sm + L::Sh, sm + L::Sl, LDK, lane);
            mma_xv(acc, sm + L::AKh, sm + L::AKl, lane);
            store_T_split(acc, sm + L::Rh, sm + L::Rl, LDC, lane);
        } else if (wave == 3) {
            mma_tile3<kN>(accY, sm + L::QTh, sm + L::QTl, LDK, sm + L::Sh, sm + L::Sl, LDK, lane);
            mma_xv(accY, sm + L::QKh, sm + L::QKl, lane);
        }
        TSTAMP(6);
        lds_barrier();
        TSTAMP(7);
        // ---- phase 5: U = T R ----------------------------------------------------------------------------------
        if (wave == 0) {
            f32x16 acc = zero16();
            mma_tile3<kC>(acc, sm + L::TMh, sm + L::TMl, LDC, sm + L::Rh, sm + L::Rl, LDC, lane);
            store_T_split(acc, sm + L::Uh, sm + L::Ul, LDC, lane);
            if (SAVE) {  // staged: written out by all threads after phase 6 with one 16-byte store each
#pragma unroll
                for (int r = 0; r < 16; r++) sh_U[d_row(r, lane) * kStageLD + (lane & 31)] = acc[r];
            }
        }
        TSTAMP(8);
        lds_barrier();
        TSTAMP(9);
        // ---- phase 6: Y += A_qb U (wave 3) and the state update (waves 1,2) -------------------------------------
        if (wave == 3) {
            mma_tile3<kC>(accY, sm + L::QBh, sm + L::QBl, LDC, sm + L::Uh, sm + L::Ul, LDC, lane);
#pragma unroll
            for (int r = 0; r < 16; r++) sh_Y[d_row(r, lane) * kStageLD + (lane & 31)] = accY[r];
        } else if (wave == 1 || wave == 2) {
            const int kt = wave - 1;  // rows (key channels) [32 kt, 32 kt + 32)
            // state at the START of chunk c as the backward's checkpoint: a q15 record (chunk_common.h) from the fp32 tile
            if (SAVE) q15_encode_tile(Smaster, hs_ + ((long)bh * nc + c) * kQRec, vh, kt, lane);
            f32x16 acc = zero16();  // D[m = k][n = v] = sum_t b^[t][k] U[t][v] + k^[t][k] V[t][v]
            mma_gen<kC, true, true, false, true>(acc, sm + L::BHh, sm + L::BHl, LDK, kt * 32, sm + L::Uh, sm + L::Ul, LDC, 0, lane);
            mma_gen<kC, true, true, true, !VEXACT>(acc, sm + L::KHh, sm + L::KHl, LDK, kt * 32, sm + L::Vt, sm + L::Vtl, LDC, 0, lane);
#pragma unroll
            for (int r = 0; r < 16; r++) Smaster[r] = sh_gC[kt * 32 + d_row(r, lane)] * (Smaster[r] + acc[r]);
        }
        TSTAMP(10);
        lds_barrier();
        TSTAMP(11);
        {
            // y (and sa) of this chunk: thread (pt, pv) owns 4 value columns of one step -> one 8/16-byte store per tensor
            // (32 scalar stores per lane from the accumulator layout used to sit in front of the next chunk's loads)
            const long o = head_base + (long)(c * kC + pt) * tstride + vh * VH + pv;
            const float4 yv = *reinterpret_cast<const float4 *>(&sh_Y[pt * kStageLD + pv]);
            if constexpr (sizeof(T) == 2) {
                *reinterpret_cast<uint2 *>(reinterpret_cast<uint16_t *>(y_) + o) = make_uint2(cvt_pk(yv.x, yv.y), cvt_pk(yv.z, yv.w));
            } else {
                *reinterpret_cast<float4 *>(reinterpret_cast<float *>(y_) + o) = yv;
            }
            if (SAVE) *reinterpret_cast<float4 *>(sa_ + o) = *reinterpret_cast<const float4 *>(&sh_U[pt * kStageLD + pv]);
        }
        // ---- phase 7: publish the new state planes S[v][k] ---------------------------------------------------
        if (wave == 1 || wave == 2) store_T_split(Smaster, sm + L::Sh + (wave - 1) * 32, sm + L::Sl + (wave - 1) * 32, LDK, lane);
        // (ordered against phase 4 of the next chunk by that chunk's phase-1/2/3 barriers)
    }
}

// --------------------------------------------------------------------------------------------------------------
// debug: one 32x32 tile product through the same primitives (GPU unit test of the fragment layouts)
// --------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void chunk_debug_mma_kernel(const float *__restrict__ X, const float *__restrict__ Y,
                                                             float *__restrict__ D, float *__restrict__ DT) {
    constexpr int LD = kN + kPad;
    __shared__ __attribute__((aligned(16))) uint16_t Xh[kC * LD], Xl[kC * LD], Yh[kC * LD], Yl[kC * LD];
    __shared__ __attribute__((aligned(16))) uint16_t Oh[kC * LDC], Ol[kC * LDC];
    const int lane = threadIdx.x;
    for (int i = lane; i < kC * kN; i += 64) {
        const int r = i / kN, c = i % kN;
        split2(X[i], Xh[r * LD + c], Xl[r * LD + c]);
        split2(Y[i], Yh[r * LD + c], Yl[r * LD + c]);
    }
    __syncthreads();
    f32x16 acc = zero16();
    mma_tile3<kN>(acc, Xh, Xl, LD, Yh, Yl, LD, lane);
#pragma unroll
    for (int r = 0; r < 16; r++) D[d_row(r, lane) * kC + (lane & 31)] = acc[r];
    store_T_split(acc, Oh, Ol, LDC, lane);
    __syncthreads();
    for (int i = lane; i < kC * kC; i += 64) DT[i] = bf2f(Oh[(i / kC) * LDC + i % kC]) + bf2f(Ol[(i / kC) * LDC + i % kC]);
}

// debug: semantics probe of ds_read_b64_tr_b16 (gfx950 LDS transpose read).  LDS is filled with in[0..4096); lane l
// passes the address of element addr[l]; out[l][0..4) = the four 16-bit values the lane receives.
typedef short v4s_t __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(64) void tr16_probe_kernel(const uint16_t *__restrict__ in, const int *__restrict__ addr,
                                                        uint16_t *__restrict__ out) {
    __shared__ __attribute__((aligned(16))) uint16_t sm[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) sm[i] = in[i];
    __syncthreads();
    const int lane = threadIdx.x;
    const v4s_t r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s_t __attribute__((address_space(3))) *)(sm + addr[lane]));
#pragma unroll
    for (int j = 0; j < 4; j++) out[lane * 4 + j] = (uint16_t)r[j];
}
int chunk_debug_tr16(const uint16_t *in, const int *addr, uint16_t *out, hipStream_t st) {
    (void)hipGetLastError();
    hipLaunchKernelGGL(tr16_probe_kernel, dim3(1), dim3(64), 0, st, in, addr, out);
    return (int)hipGetLastError();
}

// --------------------------------------------------------------------------------------------------------------
// launchers
// --------------------------------------------------------------------------------------------------------------
template <typename T>
static int launch_prep(int B, int T_, int H, const void *w, const void *a, const void *b, float *tinv, hipStream_t st) {
    (void)hipGetLastError();
    hipLaunchKernelGGL((wkv7c_prep_kernel<T>), dim3(B * H * (T_ / kC)), dim3(64), 0, st, T_, H, (const T *)w, (const T *)a,
                       (const T *)b, tinv);
    return (int)hipGetLastError();
}

template <typename T, bool SAVE>
static int launch_fwd_t(int B, int T_, int H, const void *w, const void *q, const void *k, const void *v, const void *a,
                        const void *b, const float *tinv, void *y, float *sa, void *hs, const int *seq_off, int nseq,
                        hipStream_t st) {
    static DynLdsOnce lds_once;
    if (hipError_t e = lds_once.ensure(reinterpret_cast<const void *>(&wkv7c_fwd_kernel<T, SAVE>), (int)FwdSmem::total<T>()); e != hipSuccess) return (int)e;
    (void)hipGetLastError();
    hipLaunchKernelGGL((wkv7c_fwd_kernel<T, SAVE>), dim3((seq_off ? nseq : B) * H * 2), dim3(256), FwdSmem::total<T>(), st, T_, H,
                       (const T *)w, (const T *)q, (const T *)k, (const T *)v, (const T *)a, (const T *)b, tinv, (T *)y, sa, (uint16_t *)hs, seq_off);
    return (int)hipGetLastError();
}

int chunk_prep_bf16(int B, int T_, int H, const void *w, const void *a, const void *b, float *tinv, hipStream_t st) {
    return launch_prep<bf16_t>(B, T_, H, w, a, b, tinv, st);
}
int chunk_prep_f32(int B, int T_, int H, const void *w, const void *a, const void *b, float *tinv, hipStream_t st) {
    return launch_prep<float>(B, T_, H, w, a, b, tinv, st);
}
// bf16 tensors run the 8-wave producer / consumer kernel (wkv7_chunk_fwd9.hip: 278 us against 495 us for this 4-wave kernel at
// B=8, T=4096, H=16).  fp32 tensors always run here.  The bf16 instantiation of this 4-wave kernel is an A/B twin and a cross-check:
// it is compiled into the lab build only (python -m rwkvtts_amd.build --lab, include/rwkv7_hip_lab.h).
int chunk_fwd9_bf16(int, int, int, const void *, const void *, const void *, const void *, const void *, const void *, const float *,
                    void *, float *, void *, const int *, int, hipStream_t);

int chunk_fwd_bf16(int B, int T_, int H, const void *w, const void *q, const void *k, const void *v, const void *a,
                   const void *b, const float *tinv, void *y, float *sa, void *hs, const int *seq_off, int nseq, hipStream_t st) {
    return chunk_fwd9_bf16(B, T_, H, w, q, k, v, a, b, tinv, y, sa, hs, seq_off, nseq, st);
}
#ifdef RWKV7_LAB
int chunk_fwd4_bf16(int B, int T_, int H, const void *w, const void *q, const void *k, const void *v, const void *a,
                    const void *b, const float *tinv, void *y, float *sa, void *hs, const int *seq_off, int nseq, hipStream_t st) {
    return (sa && hs) ? launch_fwd_t<bf16_t, true>(B, T_, H, w, q, k, v, a, b, tinv, y, sa, hs, seq_off, nseq, st)
                      : launch_fwd_t<bf16_t, false>(B, T_, H, w, q, k, v, a, b, tinv, y, nullptr, nullptr, seq_off, nseq, st);
}
#endif
int chunk_fwd_f32(int B, int T_, int H, const void *w, const void *q, const void *k, const void *v, const void *a,
                  const void *b, const float *tinv, void *y, float *sa, void *hs, const int *seq_off, int nseq, hipStream_t st) {
    return (sa && hs) ? launch_fwd_t<float, true>(B, T_, H, w, q, k, v, a, b, tinv, y, sa, hs, seq_off, nseq, st)
                      : launch_fwd_t<float, false>(B, T_, H, w, q, k, v, a, b, tinv, y, nullptr, nullptr, seq_off, nseq, st);
}
#ifdef WKV7C_TIMING
extern "C" int rwkv7_debug_chunk_timing(long long *out, int reset) {
    if (reset) {
        long long z[64] = {0};
        return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_chunk_timing), z, sizeof(z));
    }
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_chunk_timing), sizeof(long long) * 64);
}
#endif

int chunk_debug_mma(const float *X, const float *Y, float *D, float *DT, hipStream_t st) {
    (void)hipGetLastError();
    hipLaunchKernelGGL(chunk_debug_mma_kernel, dim3(1), dim3(64), 0, st, X, Y, D, DT);
    return (int)hipGetLastError();
}

}  // namespace rwkv7
