// rwkvtts_amd/csrc/wkv7_chunk_fwd8.hip -- chunked (MFMA) WKV7 forward, bf16 tensors, 8 waves: producer / consumer split.
//
// Same mathematics, inputs and outputs as wkv7c_fwd_kernel (wkv7_chunk_fwd.hip; reference wkv7_cuda.cu:10-52).  Per-phase
// stamps of that kernel (tools/chunk_timing.py) show that ~45 % of the sequential per-chunk path is the state-INDEPENDENT
// prologue: restaging the raw rows, exp / decay prefix sums, scaling, hi/lo splitting and writing the eight operand planes.
// Here four extra waves (the producer, waves 4-7) do that for chunk c + 1 into a second set of planes while waves 0-3 (the
// consumer) run the matrix phases of chunk c:
//     interval 1   consumer: A_ak, A_qb, A_qk                        producer: staged rows in compute mapping, next prefetch, T^-1 request, exp
//     interval 2   consumer: R = A~ H0 + A_ak V ; Q~ H0 + A_qk V    producer: decay prefix sums (DPP)
//     interval 3   consumer: U = T R                                producer: scaling, hi/lo splits of q~, a~ (kept in registers)
//     interval 4   consumer: Y += A_qb U ; state update             producer: splits of k^, b^; plane stores, V, g_C, T planes; next rows -> LDS staging
// The workgroup barrier is the only hardware barrier, so both groups pass the same four barriers per chunk; an interval
// lasts as long as its longer half.
// Measured (tools/bench_chunk_fwd_waves.py, B=8, T=4096, H=16): 335 us against 495 us for the 4-wave kernel.  Two things
// made the difference between a loss (570 us in the first cut) and this: (1) the producer's work is cut into four pieces
// that each fit beside a consumer phase -- read/convert/exp, prefix sums, scale/split (results held in registers), plane
// stores -- and the LDS staging of the NEXT chunk's rows moved to the end of interval 4; (2) the two roles are separate code
// paths (if / else around two loops with the same barrier count), so their loop-carried registers are allocated separately:
// one interleaved loop needed 256 registers plus 109 spilled to scratch, the split one 186 and none.
// LDS: 2 x 38.5 KB of operand planes + 39 KB of matrices + 9.5 KB fp32 + 25 KB staging = 151 KB, one workgroup (two
// value-column halves of a head -> two workgroups, as before) per CU.
#include "chunk_common.h"

namespace rwkv7 {

#ifdef WKV7C_TIMING
// profiling build only (python -m rwkvtts_amd.build --timing): cycle totals per interval (work, then barrier wait), workgroup 0,
// per wave (0-3 consumer, 4-7 producer); tools/cfwd8_timing.py
__device__ long long g_cfwd8_timing[8 * 12];
#define F8STAMP(i)                                              \
    do {                                                        \
        const long long now_ = __builtin_readcyclecounter();    \
        tacc_[i] += now_ - tprev_;                              \
        tprev_ = now_;                                          \
    } while (0)
#else
#define F8STAMP(i) do { } while (0)
#endif

namespace {
constexpr int LDK = kN + kPad;  // planes with K = 64 columns
constexpr int LDC = kC + kPad;  // planes with K = 32 columns
constexpr int VH = 32;          // value columns per workgroup

struct F8Smem {  // offsets in uint16 units; every plane 16-byte aligned
    static constexpr int PL = kC * LDK;
    // one producer buffer: the eight scaled operand planes, time-major, and V[t][v]
    static constexpr int QTh = 0, QTl = PL, ATh = 2 * PL, ATl = 3 * PL, KHh = 4 * PL, KHl = 5 * PL, BHh = 6 * PL, BHl = 7 * PL;
    static constexpr int Vt = 8 * PL;
    static constexpr int BUF = 8 * PL + kC * LDC;
    // single: state planes, intra-chunk matrices, T, R, U
    static constexpr int Sh = 2 * BUF, Sl = Sh + VH * LDK;
    static constexpr int AKh = Sl + VH * LDK, AKl = AKh + kC * LDC, QBh = AKl + kC * LDC, QBl = QBh + kC * LDC;
    static constexpr int QKh = QBl + kC * LDC, QKl = QKh + kC * LDC, TMh = QKl + kC * LDC, TMl = TMh + kC * LDC;
    static constexpr int Rh = TMl + kC * LDC, Rl = Rh + VH * LDC, Uh = Rl + VH * LDC, Ul = Uh + VH * LDC;
    static constexpr int end16 = Ul + VH * LDC;
    // fp32 region (offsets in floats): U, Y staging tiles [32][36]; g_C of both buffers
    static constexpr int fStage = 0, fGC = fStage + 2 * kC * 36, fend = fGC + 2 * kN;
    // raw input staging (bf16): 5 planes [32][64 + 8] and V [32][32 + 8]
    static constexpr int RS = kN + 8, RSV = VH + 8;
    static constexpr size_t bytes = (size_t)end16 * 2 + (size_t)fend * 4 + (size_t)(5 * kC * RS + kC * RSV) * 2;
};
static_assert(F8Smem::end16 % 8 == 0 && F8Smem::BUF % 8 == 0, "16-byte alignment");
static_assert(F8Smem::bytes <= 160 * 1024, "LDS budget");
}  // namespace

template <bool SAVE>
__global__ __launch_bounds__(512) void wkv7c_fwd8_kernel(int T_, int H, const bf16_t *__restrict__ w_, const bf16_t *__restrict__ q_,
                                                         const bf16_t *__restrict__ k_, const bf16_t *__restrict__ v_,
                                                         const bf16_t *__restrict__ a_, const bf16_t *__restrict__ b_,
                                                         const float *__restrict__ tinv_, bf16_t *__restrict__ y_,
                                                         float *__restrict__ sa_, uint16_t *__restrict__ hs_,
                                                         const int *__restrict__ seq_off_) {
    extern __shared__ __attribute__((aligned(16))) uint16_t sm[];
    using L = F8Smem;
    float *fm = reinterpret_cast<float *>(sm + L::end16);
    constexpr int kStageLD = 36;
    float *sh_U = fm + L::fStage, *sh_Y = sh_U + kC * kStageLD, *sh_gC2 = fm + L::fGC;
    bf16_t *raw = reinterpret_cast<bf16_t *>(fm + L::fend);
    constexpr int RS = L::RS, RSV = L::RSV;

    // workgroup -> (head, value half): the two halves of a head get block ids g and g + 8 (same XCD, shared L2)
    int vh, bh;
    if ((gridDim.x & 15) == 0) {
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        bh = (j >> 1) * 8 + xcd;
        vh = j & 1;
    } else {
        vh = blockIdx.x & 1;
        bh = blockIdx.x >> 1;
    }
    const int tid = threadIdx.x, ltid = tid & 255, lane = tid & 63;
    // scalar role / wave ids: derived from threadIdx they make every `if (wave == ..)` an exec-masked region that all waves walk
    const int role = __builtin_amdgcn_readfirstlane(tid >> 8), wave = __builtin_amdgcn_readfirstlane(ltid >> 6);
    const int nc = T_ / kC;
    int bb, hh, c0 = 0, c1 = nc;
    if (seq_off_) {  // packed rows: see wkv7c_fwd_kernel
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

    // per-thread roles inside a group of 256 threads (identical to wkv7c_fwd_kernel)
    const int pt = ltid & 31, pk = (ltid >> 5) * 8, pv = (ltid >> 5) * 4;
    const int lt = ltid >> 3, lk = (ltid & 7) * 8, lv = (ltid & 7) * 4;

    for (int i = tid; i < 2 * VH * LDK; i += 512) sm[L::Sh + i] = 0;  // chunk c0 starts from S = 0
    f32x16 Smaster = zero16();  // consumer waves 1,2: D-layout tile of the fp32 state

    // producer: raw rows of the next chunk, global -> registers (row-contiguous mapping)
    Raw4<bf16_t> gw[2], gq[2], gk[2], ga[2], gb[2], gv;
    auto issue = [&](int c) {
        const long off = head_base + (long)(c * kC + lt) * tstride;
#pragma unroll
        for (int i = 0; i < 2; i++) {
            gw[i] = ld4<bf16_t>(w_ + off + lk + 4 * i, true);
            gq[i] = ld4<bf16_t>(q_ + off + lk + 4 * i, true);
            gk[i] = ld4<bf16_t>(k_ + off + lk + 4 * i, true);
            ga[i] = ld4<bf16_t>(a_ + off + lk + 4 * i, true);
            gb[i] = ld4<bf16_t>(b_ + off + lk + 4 * i, true);
        }
        gv = ld4<bf16_t>(v_ + off + vh * VH + lv, true);
    };
    using RawVec = decltype(Raw4<bf16_t>::r);
    auto stage_raw = [&]() {  // prefetched rows -> LDS staging (read back in the compute mapping one barrier later)
#pragma unroll
        for (int i = 0; i < 2; i++) {
            *reinterpret_cast<RawVec *>(raw + (0 * kC + lt) * RS + lk + 4 * i) = gw[i].r;
            *reinterpret_cast<RawVec *>(raw + (1 * kC + lt) * RS + lk + 4 * i) = gq[i].r;
            *reinterpret_cast<RawVec *>(raw + (2 * kC + lt) * RS + lk + 4 * i) = gk[i].r;
            *reinterpret_cast<RawVec *>(raw + (3 * kC + lt) * RS + lk + 4 * i) = ga[i].r;
            *reinterpret_cast<RawVec *>(raw + (4 * kC + lt) * RS + lk + 4 * i) = gb[i].r;
        }
        *reinterpret_cast<RawVec *>(raw + 5 * kC * RS + lt * RSV + lv) = gv.r;
    };
    if (role == 1) {
        issue(c0);
        stage_raw();
    }
    // producer values carried between intervals
    float lw[8], Gc[8], qv[8], kv[8], av[8], bv[8], vv[4], gamL[8];
    uint4 pq[2], pa[2], pkk[2], pb[2];
    float ksL[8], bsL[8];
    float4 tmreg = make_float4(0.f, 0.f, 0.f, 0.f);
    lds_barrier();
#ifdef WKV7C_TIMING
    long long tacc_[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    long long tprev_ = __builtin_readcyclecounter();
#endif

    // The two roles run disjoint code (separate register allocation) with the same barrier sequence: four per iteration.
    if (role == 0) {
    for (int it = c0 - 1; it < c1; it++) {
            const int cc = it, pc = it + 1;
            uint16_t *bufc = sm + (cc & 1) * L::BUF, *bufp = sm + (pc & 1) * L::BUF;
            const float *gCc = sh_gC2 + (cc & 1) * kN;
            float *gCp = sh_gC2 + (pc & 1) * kN;
            // =============================================================== interval 1
            if (cc >= c0) {
                if (wave == 0) {
                    f32x16 acc = zero16();  // D[m = s][n = t] = k^_s . a~_t = A_ak[t][s]
                    mma_tile3<kN>(acc, bufc + L::KHh, bufc + L::KHl, LDK, bufc + L::ATh, bufc + L::ATl, LDK, lane);
                    mask_lower_T<true>(acc, lane);
                    store_T_split(acc, sm + L::AKh, sm + L::AKl, LDC, lane);
                } else if (wave == 1) {
                    f32x16 acc = zero16();  // b^_s . q~_t = A_qb[t][s]
                    mma_tile3<kN>(acc, bufc + L::BHh, bufc + L::BHl, LDK, bufc + L::QTh, bufc + L::QTl, LDK, lane);
                    mask_lower_T<false>(acc, lane);
                    store_T_split(acc, sm + L::QBh, sm + L::QBl, LDC, lane);
                } else if (wave == 2) {
                    f32x16 acc = zero16();  // k^_s . q~_t = A_qk[t][s]
                    mma_tile3<kN>(acc, bufc + L::KHh, bufc + L::KHl, LDK, bufc + L::QTh, bufc + L::QTl, LDK, lane);
                    mask_lower_T<false>(acc, lane);
                    store_T_split(acc, sm + L::QKh, sm + L::QKl, LDC, lane);
                }   // wave 3: idle here (the T planes come from the producer)
            }
            F8STAMP(0);
            lds_barrier();
            F8STAMP(1);
            // =============================================================== interval 2
            f32x16 accY = zero16();  // consumer wave 3: the part of Y that does not need U, finished in interval 4
            if (cc >= c0) {
                // R = A~ H0 + A_ak V   (D[t][v]) ; wave 3: Q~ H0 + A_qk V
                if (wave == 0) {
                    f32x16 acc = zero16();
                    mma_tile3<kN>(acc, bufc + L::ATh, bufc + L::ATl, LDK, sm + L::Sh, sm + L::Sl, LDK, lane);
                    mma_gen<kC, false, true, true, false>(acc, sm + L::AKh, sm + L::AKl, LDC, 0, bufc + L::Vt, bufc + L::Vt, LDC, 0, lane);
                    store_T_split(acc, sm + L::Rh, sm + L::Rl, LDC, lane);
                } else if (wave == 3) {
                    mma_tile3<kN>(accY, bufc + L::QTh, bufc + L::QTl, LDK, sm + L::Sh, sm + L::Sl, LDK, lane);
                    mma_gen<kC, false, true, true, false>(accY, sm + L::QKh, sm + L::QKl, LDC, 0, bufc + L::Vt, bufc + L::Vt, LDC, 0, lane);
                }
            }
            F8STAMP(2);
            lds_barrier();
            F8STAMP(3);
            // =============================================================== interval 3
            if (cc >= c0) {
                if (wave == 0) {  // U = T R
                    f32x16 acc = zero16();
                    mma_tile3<kC>(acc, sm + L::TMh, sm + L::TMl, LDC, sm + L::Rh, sm + L::Rl, LDC, lane);
                    store_T_split(acc, sm + L::Uh, sm + L::Ul, LDC, lane);
                    if (SAVE) {
    #pragma unroll
                        for (int r = 0; r < 16; r++) sh_U[d_row(r, lane) * kStageLD + (lane & 31)] = acc[r];
                    }
                }
            }
            F8STAMP(4);
            lds_barrier();
            F8STAMP(5);
            // =============================================================== interval 4
            if (cc >= c0) {
                if (wave == 3) {
                    mma_tile3<kC>(accY, sm + L::QBh, sm + L::QBl, LDC, sm + L::Uh, sm + L::Ul, LDC, lane);
    #pragma unroll
                    for (int r = 0; r < 16; r++) sh_Y[d_row(r, lane) * kStageLD + (lane & 31)] = accY[r];
                } else if (wave == 1 || wave == 2) {
                    const int kt = wave - 1;  // rows (key channels) [32 kt, 32 kt + 32)
                    // state at the START of chunk cc as the backward's checkpoint: a q15 record (chunk_common.h) straight from the
                    // fp32 accumulator tile -- 3 stores per lane (the fp32 [k][v] checkpoint: 16 scattered 4-byte stores, 16 KB)
                    if (SAVE) q15_encode_tile(Smaster, hs_ + ((long)bh * nc + cc) * kQRec, vh, kt, lane);
                    f32x16 acc = zero16();  // D[m = k][n = v] = sum_t b^[t][k] U[t][v] + k^[t][k] V[t][v]
                    mma_gen<kC, true, true, false, true>(acc, bufc + L::BHh, bufc + L::BHl, LDK, kt * 32, sm + L::Uh, sm + L::Ul, LDC, 0, lane);
                    mma_gen<kC, true, true, true, false>(acc, bufc + L::KHh, bufc + L::KHl, LDK, kt * 32, bufc + L::Vt, bufc + L::Vt, LDC, 0, lane);
    #pragma unroll
                    for (int r = 0; r < 16; r++) Smaster[r] = gCc[kt * 32 + d_row(r, lane)] * (Smaster[r] + acc[r]);
                }
            }
            F8STAMP(6);
            lds_barrier();
            F8STAMP(7);
            if (cc >= c0) {
                // y (and sa) of this chunk: thread (pt, pv) owns 4 value columns of one step
                const long o = head_base + (long)(cc * kC + pt) * tstride + vh * VH + pv;
                const float4 yv = *reinterpret_cast<const float4 *>(&sh_Y[pt * kStageLD + pv]);
                *reinterpret_cast<uint2 *>(reinterpret_cast<uint16_t *>(y_) + o) = make_uint2(cvt_pk(yv.x, yv.y), cvt_pk(yv.z, yv.w));
                if (SAVE) *reinterpret_cast<float4 *>(sa_ + o) = *reinterpret_cast<const float4 *>(&sh_U[pt * kStageLD + pv]);
                // publish the new state planes S[v][k] (read again in interval 2 of the next iteration, two barriers away)
                if (wave == 1 || wave == 2) store_T_split(Smaster, sm + L::Sh + (wave - 1) * 32, sm + L::Sl + (wave - 1) * 32, LDK, lane);
            }
        }
    } else {
    for (int it = c0 - 1; it < c1; it++) {
            const int cc = it, pc = it + 1;
            uint16_t *bufc = sm + (cc & 1) * L::BUF, *bufp = sm + (pc & 1) * L::BUF;
            const float *gCc = sh_gC2 + (cc & 1) * kN;
            float *gCp = sh_gC2 + (pc & 1) * kN;
            // =============================================================== interval 1
            if (pc < c1) {
                // rows of chunk pc in the compute mapping (staged one interval ago), the chunk after it requested from HBM
                Raw4<bf16_t> rw[2], rq[2], rk[2], ra[2], rb[2], rv;
    #pragma unroll
                for (int i = 0; i < 2; i++) {
                    rw[i].r = *reinterpret_cast<const RawVec *>(raw + (0 * kC + pt) * RS + pk + 4 * i);
                    rq[i].r = *reinterpret_cast<const RawVec *>(raw + (1 * kC + pt) * RS + pk + 4 * i);
                    rk[i].r = *reinterpret_cast<const RawVec *>(raw + (2 * kC + pt) * RS + pk + 4 * i);
                    ra[i].r = *reinterpret_cast<const RawVec *>(raw + (3 * kC + pt) * RS + pk + 4 * i);
                    rb[i].r = *reinterpret_cast<const RawVec *>(raw + (4 * kC + pt) * RS + pk + 4 * i);
                }
                rv.r = *reinterpret_cast<const RawVec *>(raw + 5 * kC * RS + pt * RSV + pv);
                if (pc + 1 < c1) issue(pc + 1);
                // T = (I - A_ab)^-1 of chunk pc (wkv7c_prep_kernel): requested here, written as planes in interval 4 -- after the
                // consumer's phase-5 read of the previous chunk's T and a whole iteration before it reads this one
                tmreg = *reinterpret_cast<const float4 *>(tinv_ + ((long)bh * nc + pc) * kC * kC + ltid * 4);
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
            F8STAMP(0);
            lds_barrier();
            F8STAMP(1);
            // =============================================================== interval 2
            if (pc < c1) {
                // inclusive cumulative log-decay over the chunk: DPP prefix sum across the 32 lanes that hold the 32 steps
    #pragma unroll
                for (int j = 0; j < 8; j++) Gc[j] = scan32(lw[j]);
            }
            F8STAMP(2);
            lds_barrier();
            F8STAMP(3);
            // =============================================================== interval 3
            if (pc < c1) {
                // scaled operands of chunk pc, split into bf16 hi/lo pairs (stored in interval 4)
                float qs[8], as_[8], ks[8], bs[8];
    #pragma unroll
                for (int j = 0; j < 8; j++) {
                    const float gam = fast_exp(Gc[j]), gprev = fast_exp(Gc[j] - lw[j]), ig = fast_exp(-Gc[j]);
                    qs[j] = qv[j] * gam;
                    as_[j] = av[j] * gprev;
                    ks[j] = kv[j] * ig;
                    bs[j] = bv[j] * ig;
                    gamL[j] = gam;
                }
                uint32_t qh[4], ql[4], ah[4], al[4];
    #pragma unroll
                for (int j = 0; j < 4; j++) {
                    split_pk(qs[2 * j], qs[2 * j + 1], qh[j], ql[j]);
                    split_pk(as_[2 * j], as_[2 * j + 1], ah[j], al[j]);
                }
                auto pack = [](const uint32_t (&x)[4]) { return make_uint4(x[0], x[1], x[2], x[3]); };
                pq[0] = pack(qh); pq[1] = pack(ql); pa[0] = pack(ah); pa[1] = pack(al);
    #pragma unroll
                for (int j = 0; j < 8; j++) {   // k^, b^ are split in interval 4: this interval is the producer's longest
                    ksL[j] = ks[j];
                    bsL[j] = bs[j];
                }
            }
            F8STAMP(4);
            lds_barrier();
            F8STAMP(5);
            // =============================================================== interval 4
            if (pc < c1) {
                // planes of chunk pc into its buffer (the consumer reads the other one); then the next chunk's raw rows -> staging
                {
                    uint32_t kh[4], kl[4], bhh[4], bl[4];
    #pragma unroll
                    for (int j = 0; j < 4; j++) {
                        split_pk(ksL[2 * j], ksL[2 * j + 1], kh[j], kl[j]);
                        split_pk(bsL[2 * j], bsL[2 * j + 1], bhh[j], bl[j]);
                    }
                    pkk[0] = make_uint4(kh[0], kh[1], kh[2], kh[3]); pkk[1] = make_uint4(kl[0], kl[1], kl[2], kl[3]);
                    pb[0] = make_uint4(bhh[0], bhh[1], bhh[2], bhh[3]); pb[1] = make_uint4(bl[0], bl[1], bl[2], bl[3]);
                }
                const int o = pt * LDK + pk;
                *reinterpret_cast<uint4 *>(&bufp[L::QTh + o]) = pq[0];
                *reinterpret_cast<uint4 *>(&bufp[L::QTl + o]) = pq[1];
                *reinterpret_cast<uint4 *>(&bufp[L::ATh + o]) = pa[0];
                *reinterpret_cast<uint4 *>(&bufp[L::ATl + o]) = pa[1];
                *reinterpret_cast<uint4 *>(&bufp[L::KHh + o]) = pkk[0];
                *reinterpret_cast<uint4 *>(&bufp[L::KHl + o]) = pkk[1];
                *reinterpret_cast<uint4 *>(&bufp[L::BHh + o]) = pb[0];
                *reinterpret_cast<uint4 *>(&bufp[L::BHl + o]) = pb[1];
                *reinterpret_cast<uint2 *>(&bufp[L::Vt + pt * LDC + pv]) = make_uint2(cvt_pk(vv[0], vv[1]), cvt_pk(vv[2], vv[3]));  // bf16 v: exact
                if (pt == kC - 1) {
    #pragma unroll
                    for (int j = 0; j < 8; j++) gCp[pk + j] = gamL[j];
                }
                {   // T planes Tm[t][r] of chunk pc: thread = row ltid >> 3, columns 4 (ltid & 7) .. +4
                    uint32_t h0, l0, h1, l1;
                    split_pk(tmreg.x, tmreg.y, h0, l0);
                    split_pk(tmreg.z, tmreg.w, h1, l1);
                    const int ot = (ltid >> 3) * LDC + (ltid & 7) * 4;
                    *reinterpret_cast<uint2 *>(&sm[L::TMh + ot]) = make_uint2(h0, h1);
                    *reinterpret_cast<uint2 *>(&sm[L::TMl + ot]) = make_uint2(l0, l1);
                }
                if (pc + 1 < c1) stage_raw();
            }
            F8STAMP(6);
            lds_barrier();
            F8STAMP(7);
            
        }
    }
#ifdef WKV7C_TIMING
    if (blockIdx.x == 0 && lane == 0)
        for (int i = 0; i < 12; i++) g_cfwd8_timing[(tid >> 6) * 12 + i] += tacc_[i];
#endif
}

static int launch_fwd8(bool save, int B, int T_, int H, const void *w, const void *q, const void *k, const void *v, const void *a,
                       const void *b, const float *tinv, void *y, float *sa, void *hs, const int *seq_off, int nseq, hipStream_t st) {
    static bool attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&wkv7c_fwd8_kernel<true>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)F8Smem::bytes);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(&wkv7c_fwd8_kernel<false>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)F8Smem::bytes);
        if (e != hipSuccess) return (int)e;
        attr = true;
    }
    (void)hipGetLastError();
    const dim3 grid((seq_off ? nseq : B) * H * 2), block(512);
    if (save)
        hipLaunchKernelGGL(wkv7c_fwd8_kernel<true>, grid, block, F8Smem::bytes, st, T_, H, (const bf16_t *)w, (const bf16_t *)q,
                           (const bf16_t *)k, (const bf16_t *)v, (const bf16_t *)a, (const bf16_t *)b, tinv, (bf16_t *)y, sa, (uint16_t *)hs, seq_off);
    else
        hipLaunchKernelGGL(wkv7c_fwd8_kernel<false>, grid, block, F8Smem::bytes, st, T_, H, (const bf16_t *)w, (const bf16_t *)q,
                           (const bf16_t *)k, (const bf16_t *)v, (const bf16_t *)a, (const bf16_t *)b, tinv, (bf16_t *)y, nullptr, nullptr,
                           seq_off);
    return (int)hipGetLastError();
}

int chunk_fwd8_bf16(int B, int T_, int H, const void *w, const void *q, const void *k, const void *v, const void *a, const void *b,
                    const float *tinv, void *y, float *sa, void *hs, const int *seq_off, int nseq, hipStream_t st) {
    return launch_fwd8(sa && hs, B, T_, H, w, q, k, v, a, b, tinv, y, sa, hs, seq_off, nseq, st);
}

}  // namespace rwkv7

#ifdef WKV7C_TIMING
extern "C" int rwkv7_debug_cfwd8_timing(long long *out, int reset) {
    if (reset) {
        long long z[96] = {0};
        return (int)hipMemcpyToSymbol(HIP_SYMBOL(rwkv7::g_cfwd8_timing), z, sizeof(z));
    }
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(rwkv7::g_cfwd8_timing), sizeof(long long) * 96);
}
#endif
