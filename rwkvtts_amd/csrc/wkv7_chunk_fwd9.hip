// rwkvtts_amd/csrc/wkv7_chunk_fwd9.hip -- chunked (MFMA) WKV7 forward, bf16 tensors, 8 waves, TWO dependent products per chunk.
//
// Reference: wkv7_cuda.cu:10-52.  Its predecessor (round 2's wkv7c_fwd8_kernel, removed in round 4: git log -- rwkvtts_amd/csrc/wkv7_chunk_fwd8.hip)
// walked a chunk in four barrier-separated intervals, three of them on the state's dependency chain:
//     S -> R = A~ S + A_ak V -> U = T R -> S' = g_C (S + B^^T U + K^^T V)
// (6.2k cycles per chunk, sequential over T/32 chunks, one workgroup per CU: the whole kernel is this chain).  Here the chain is
// cut to TWO products by moving T to the state-independent side:
//     W = T A~ (32 x 64),  X' = T A_ak (32 x 32)        -- no state in them: computed one chunk ahead, beside the chain
//     U = W S + X' V                                    -- interval a
//     S' = g_C (S + B^^T U + K^^T V)                    -- interval b
// and nothing else is left on it: A_qb / A_qk of a chunk are made in its own interval a by the two waves that wait for U, the
// state-independent half of Y (Q~ S) runs beside U, the rest of Y beside the state update.
//     interval a   wave 0: U = W S + X' V -> U planes        1: A_qb     2: A_qk     3: Y  = Q~ S        (0-3: y / sa of the
//                  previous chunk -> HBM first)
//                  waves 4-7 (producer): k^, b^ splits; the eight operand planes, V, g_C, T planes of the NEXT chunk; raw rows of
//                                        the chunk after it -> LDS staging; next global prefetch
//     interval b   waves 1,2: checkpoint, S' (two key tiles) 3: Y += A_qk V + A_qb U -> staging; W of the next chunk
//                  0: U -> staging (sa); A_ak, X' of the next chunk
//                  waves 4-7: rows of the chunk after the next in the compute mapping: exp, prefix sums, scaling, q~ / a~ splits
// Two barriers per chunk.  LDS: 2 x 38.5 KB operand planes + 48 KB matrices + 9.5 KB fp32 + 25 KB staging = 159.5 KB.
// Measured (tools/bench_chunk_fwd_waves.py, B=8, T=4096, H=16): 278 us against 335 us for the three-product kernel; interval stamps
// (tools/cfwd9_timing.py): 4.5k cycles per chunk = 2.25k + 2.25k, every wave within 10 % of the interval in both -- the chain
// waves' own work is 1.5k + 1.5k.  A fragment fetch of one 32x32x64 product (16 ds_read_b128) takes ~790 cycles with all eight
// waves on the LDS: the ~120 KB of LDS stores per chunk (operand planes 37, staging 25, split intermediates and fp32 tiles 55)
// go through a ~80 B/clk path (MI355X_MICROARCH.md, LDS table) and are half of the interval.  Producer waves at s_setprio 1:
// 5.3k cycles per chunk (the consumer waves lose the VALU slots) -- off.
#include "chunk_common.h"
#include "launch_attr.h"

#ifndef WKV7C_F9_PRODUCER_PRIO
#define WKV7C_F9_PRODUCER_PRIO 0
#endif

namespace rwkv7 {

#ifdef WKV7C_TIMING
// profiling build only (python -m rwkvtts_amd.build --timing): cycle totals per wave of workgroup 0, seven slots per chunk --
// interval a: products | epilogue + LDS drain | barrier wait; interval b: products | update + state planes | the rest | barrier wait
// (tools/cfwd9_timing.py; profiles/r06z_cfwd9_timing_fine.txt)
__device__ long long g_cfwd9_timing[8 * 8];
#define F9STAMP(i)                                              \
    do {                                                        \
        const long long now_ = __builtin_readcyclecounter();    \
        tacc_[i] += now_ - tprev_;                              \
        tprev_ = now_;                                          \
    } while (0)
#define F9FORCE(x) asm volatile("v_mov_b32 %0, %0" : "+v"(x))          // stamps behind a product: wait for the accumulator
#define F9DRAIN() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")   // ... behind LDS stores: wait for them
#else
#define F9STAMP(i) do { } while (0)
#define F9FORCE(x) do { } while (0)
#define F9DRAIN() do { } while (0)
#endif

namespace {
constexpr int LDK = kN + kPad;  // planes with K = 64 columns
constexpr int LDC = kC + kPad;  // planes with K = 32 columns
constexpr int VH = 32;          // value columns per workgroup

struct F9Smem {  // offsets in uint16 units; every plane 16-byte aligned
    static constexpr int PL = kC * LDK, PS = kC * LDC;
    // one producer buffer: the eight scaled operand planes, time-major, and V[t][v]
    static constexpr int QTh = 0, QTl = PL, ATh = 2 * PL, ATl = 3 * PL, KHh = 4 * PL, KHl = 5 * PL, BHh = 6 * PL, BHl = 7 * PL;
    static constexpr int Vt = 8 * PL;
    static constexpr int BUF = 8 * PL + PS;
    // single: state planes S[v][k], W[t][k], and the 32 x 32 matrices X'[t][s], A_ak[t][s], A_qb[t][s], A_qk[t][s], T[t][r], U[v][t]
    static constexpr int Sh = 2 * BUF, Sl = Sh + VH * LDK;
    static constexpr int Wh = Sl + VH * LDK, Wl = Wh + PL;
    static constexpr int XPh = Wl + PL, XPl = XPh + PS, AKh = XPl + PS, AKl = AKh + PS, QBh = AKl + PS, QBl = QBh + PS;
    static constexpr int QKh = QBl + PS, QKl = QKh + PS, TMh = QKl + PS, TMl = TMh + PS, Uh = TMl + PS, Ul = Uh + VH * LDC;
    static constexpr int end16 = Ul + VH * LDC;
    // fp32 region (offsets in floats): U, Y staging tiles [32][36]; g_C of both buffers
    static constexpr int fStage = 0, fGC = fStage + 2 * kC * 36, fend = fGC + 2 * kN;
    // raw input staging (bf16): 5 planes [32][64 + 8] and V [32][32 + 8]
    static constexpr int RS = kN + 8, RSV = VH + 8;
    static constexpr size_t bytes = (size_t)end16 * 2 + (size_t)fend * 4 + (size_t)(5 * kC * RS + kC * RSV) * 2;
};
static_assert(F9Smem::end16 % 8 == 0 && F9Smem::BUF % 8 == 0 && F9Smem::Wh % 8 == 0 && F9Smem::XPh % 8 == 0, "16-byte alignment");
static_assert(F9Smem::bytes <= 160 * 1024, "LDS budget");
}  // namespace

template <bool SAVE>
__global__ __launch_bounds__(512) void wkv7c_fwd9_kernel(int T_, int H, const bf16_t *__restrict__ w_, const bf16_t *__restrict__ q_,
                                                         const bf16_t *__restrict__ k_, const bf16_t *__restrict__ v_,
                                                         const bf16_t *__restrict__ a_, const bf16_t *__restrict__ b_,
                                                         const float *__restrict__ tinv_, bf16_t *__restrict__ y_,
                                                         float *__restrict__ sa_, uint16_t *__restrict__ hs_,
                                                         const int *__restrict__ seq_off_) {
    extern __shared__ __attribute__((aligned(16))) uint16_t sm[];
    using L = F9Smem;
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
    // scalar role / wave ids (as plain functions of threadIdx every `if (wave == ..)` is an exec-masked region that all waves walk)
    const int role = __builtin_amdgcn_readfirstlane(tid >> 8), wave = __builtin_amdgcn_readfirstlane(ltid >> 6);
    const int nc = T_ / kC;
    int bb, hh, c0 = 0, c1 = nc;
    if (seq_off_) {  // packed rows: one workgroup pair per (sequence, head) walks only that sequence's chunks
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

    // per-thread roles inside a group of 256 threads: compute mapping (step pt, 8 channels from pk / 4 value columns from pv)
    // and global mapping (row lt, 8 channels from lk / 4 value columns from lv: 8 lanes per 128-byte row)
    const int pt = ltid & 31, pk = (ltid >> 5) * 8, pv = (ltid >> 5) * 4;
    const int lt = ltid >> 3, lk = (ltid & 7) * 8, lv = (ltid & 7) * 4;

    for (int i = tid; i < 2 * VH * LDK; i += 512) sm[L::Sh + i] = 0;  // chunk c0 starts from S = 0
    using RawVec = decltype(Raw4<bf16_t>::r);
#ifdef WKV7C_TIMING
    long long tacc_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long tprev_ = __builtin_readcyclecounter();
#endif

    // The two roles run disjoint code (separate register allocation) with the same barrier sequence: two before the loop, two per
    // iteration.  Iteration `it`: the consumer works on chunk cc = it, the producer finishes chunk pc = it + 1 and starts pc + 1.
    if (role == 0) {
        // =================================================================================================== consumer
        f32x16 Smaster = zero16();  // waves 1, 2: D-layout tile (32 keys x 32 value columns) of the fp32 state
        // y (and sa) of a chunk, staged in its interval b, leave at the start of the next interval a: thread (pt, pv) owns 4 value
        // columns of one step.  On the consumer side: the producer's vector-memory queue holds its prefetches, and a store in front
        // of them makes every wait for a prefetched row a wait for the store as well (one in-order counter for loads and stores)
        auto store_out = [&](int c) {
            const long o = head_base + (long)(c * kC + pt) * tstride + vh * VH + pv;
            const float4 yv = *reinterpret_cast<const float4 *>(&sh_Y[pt * kStageLD + pv]);
            *reinterpret_cast<uint2 *>(reinterpret_cast<uint16_t *>(y_) + o) = make_uint2(cvt_pk(yv.x, yv.y), cvt_pk(yv.z, yv.w));
            if (SAVE) *reinterpret_cast<float4 *>(sa_ + o) = *reinterpret_cast<const float4 *>(&sh_U[pt * kStageLD + pv]);
        };
        if (SAVE && (wave == 1 || wave == 2)) q15_encode_tile(Smaster, hs_ + ((long)bh * nc + c0) * kQRec, vh, wave - 1, lane);   // S = 0
        lds_barrier();
        lds_barrier();
        for (int it = c0 - 1; it < c1; it++) {
            const int cc = it, pc = it + 1;
            const uint16_t *bufc = sm + (cc & 1) * L::BUF, *bufp = sm + (pc & 1) * L::BUF;
            const float *gCc = sh_gC2 + (cc & 1) * kN;
            // ----------------------------------------------------------------------------------------------- interval a
            if (cc - 1 >= c0) store_out(cc - 1);
            f32x16 accA = zero16();  // wave 0: U (-> sa staging in interval b); wave 3: Y
            if (cc >= c0) {
                if (wave == 0) {         // U = W S + X' V : D[t][v] -> U[v][t]
                    mma_tile3<kN>(accA, sm + L::Wh, sm + L::Wl, LDK, sm + L::Sh, sm + L::Sl, LDK, lane);
                    mma_gen<kC, false, true, true, false>(accA, sm + L::XPh, sm + L::XPl, LDC, 0, bufc + L::Vt, bufc + L::Vt, LDC, 0, lane);
                    F9FORCE(accA[0]); F9STAMP(0);
                    store_T_split(accA, sm + L::Uh, sm + L::Ul, LDC, lane);
                } else if (wave == 1) {  // D[m = s][n = t] = b^_s . q~_t = A_qb[t][s], s <= t
                    f32x16 acc = zero16();
                    mma_tile3<kN>(acc, bufc + L::BHh, bufc + L::BHl, LDK, bufc + L::QTh, bufc + L::QTl, LDK, lane);
                    F9FORCE(acc[0]); F9STAMP(0);
                    mask_lower_T<false>(acc, lane);
                    store_T_split(acc, sm + L::QBh, sm + L::QBl, LDC, lane);
                } else if (wave == 2) {  // k^_s . q~_t = A_qk[t][s], s <= t
                    f32x16 acc = zero16();
                    mma_tile3<kN>(acc, bufc + L::KHh, bufc + L::KHl, LDK, bufc + L::QTh, bufc + L::QTl, LDK, lane);
                    F9FORCE(acc[0]); F9STAMP(0);
                    mask_lower_T<false>(acc, lane);
                    store_T_split(acc, sm + L::QKh, sm + L::QKl, LDC, lane);
                } else {                 // the part of Y that needs neither U nor the A matrices
                    mma_tile3<kN>(accA, bufc + L::QTh, bufc + L::QTl, LDK, sm + L::Sh, sm + L::Sl, LDK, lane);
                    F9FORCE(accA[0]); F9STAMP(0);
                }
            }
            F9DRAIN();
            F9STAMP(1);
            lds_barrier();
            F9STAMP(2);
            // ----------------------------------------------------------------------------------------------- interval b
            if (cc >= c0) {
                if (wave == 1 || wave == 2) {
                    const int kt = wave - 1;  // key channels [32 kt, 32 kt + 32)
                    f32x16 acc = zero16();  // D[m = k][n = v] = sum_t b^[t][k] U[t][v] + k^[t][k] V[t][v]
                    mma_gen<kC, true, true, false, true>(acc, bufc + L::BHh, bufc + L::BHl, LDK, kt * 32, sm + L::Uh, sm + L::Ul, LDC, 0, lane);
                    mma_gen<kC, true, true, true, false>(acc, bufc + L::KHh, bufc + L::KHl, LDK, kt * 32, bufc + L::Vt, bufc + L::Vt, LDC, 0, lane);
                    F9FORCE(acc[0]); F9STAMP(3);
#pragma unroll
                    for (int r = 0; r < 16; r++) Smaster[r] = gCc[kt * 32 + d_row(r, lane)] * (Smaster[r] + acc[r]);
                    // new state planes S[v][k]: read by waves 0 and 3 in the next interval a (one barrier away)
                    store_T_split(Smaster, sm + L::Sh + kt * 32, sm + L::Sl + kt * 32, LDK, lane);
                    F9DRAIN(); F9STAMP(4);
                    // the state at the START of chunk cc + 1 = the backward's checkpoint: q15 record straight from the accumulator
                    // tile.  Here, behind the state planes (round 4; round 3 wrote it at the top of the next interval a): the interval
                    // stamps put waves 1, 2 at 2.1k of interval a's 2.26k cycles (the A_qb / A_qk products + this record, sharing their
                    // SIMDs with producer waves that are as long) and at 1.2k of interval b's -- the record moves to where the slack is
                    if (SAVE && cc + 1 < c1) q15_encode_tile(Smaster, hs_ + ((long)bh * nc + cc + 1) * kQRec, vh, kt, lane);
                } else if (wave == 3) {
                    mma_gen<kC, false, true, true, false>(accA, sm + L::QKh, sm + L::QKl, LDC, 0, bufc + L::Vt, bufc + L::Vt, LDC, 0, lane);
                    mma_tile3<kC>(accA, sm + L::QBh, sm + L::QBl, LDC, sm + L::Uh, sm + L::Ul, LDC, lane);
                    F9FORCE(accA[0]); F9STAMP(3);
#pragma unroll
                    for (int r = 0; r < 16; r++) sh_Y[d_row(r, lane) * kStageLD + (lane & 31)] = accA[r];
                } else if (wave == 0 && SAVE) {
#pragma unroll
                    for (int r = 0; r < 16; r++) sh_U[d_row(r, lane) * kStageLD + (lane & 31)] = accA[r];
                }
            }
            if (wave == 3 && pc < c1) {
                // next chunk, state-independent: W = T A~ : D[m = k][n = t] = sum_s a~[s][k] T[t][s] -> W[t][k], two key tiles
#pragma unroll
                for (int kt = 0; kt < 2; kt++) {
                    f32x16 acc = zero16();
                    mma_gen<kC, true, true, false, true>(acc, bufp + L::ATh, bufp + L::ATl, LDK, kt * 32, sm + L::TMh, sm + L::TMl, LDC, 0, lane);
                    store_T_split(acc, sm + L::Wh + kt * 32, sm + L::Wl + kt * 32, LDK, lane);
                }
            }
            if (wave == 0 && pc < c1) {
                // next chunk, state-independent: A_ak[t][s] (s < t), then X' = T A_ak (this wave reads back what it wrote:
                // LDS operations of one wave execute in order)
                f32x16 acc = zero16();  // D[m = s][n = t] = k^_s . a~_t
                mma_tile3<kN, 2>(acc, bufp + L::KHh, bufp + L::KHl, LDK, bufp + L::ATh, bufp + L::ATl, LDK, lane);
                mask_lower_T<true>(acc, lane);
                store_T_split(acc, sm + L::AKh, sm + L::AKl, LDC, lane);
                f32x16 acx = zero16();  // D[m = s][n = t] = sum_r A_ak[r][s] T[t][r] = X'[t][s]
                mma_gen<kC, true, true, false, true>(acx, sm + L::AKh, sm + L::AKl, LDC, 0, sm + L::TMh, sm + L::TMl, LDC, 0, lane);
                store_T_split(acx, sm + L::XPh, sm + L::XPl, LDC, lane);
            }
            F9DRAIN();
            F9STAMP(5);
            lds_barrier();
            F9STAMP(6);
        }
        store_out(c1 - 1);
    } else {
        // =================================================================================================== producer
#if WKV7C_F9_PRODUCER_PRIO
        __builtin_amdgcn_s_setprio(1);   // the second-dispatched half loses VALU arbitration by age (MI355X_MICROARCH.md, "Two waves per SIMD")
#endif
        // raw rows, global -> registers (row-contiguous mapping) -> LDS staging -> registers (compute mapping, one barrier later)
        uint4 gw, gq, gk, ga, gb;
        Raw4<bf16_t> gv;
        auto issue = [&](int c) {   // unconditional: the chunk index is clamped by the caller (the notes on conditional loads: DESIGN.md section 4, "compiler traps")
            const long off = head_base + (long)(c * kC + lt) * tstride;
            gw = *reinterpret_cast<const uint4 *>(w_ + off + lk);
            gq = *reinterpret_cast<const uint4 *>(q_ + off + lk);
            gk = *reinterpret_cast<const uint4 *>(k_ + off + lk);
            ga = *reinterpret_cast<const uint4 *>(a_ + off + lk);
            gb = *reinterpret_cast<const uint4 *>(b_ + off + lk);
            gv = ld4<bf16_t>(v_ + off + vh * VH + lv, true);
        };
        auto stage_raw = [&]() {
            *reinterpret_cast<uint4 *>(raw + (0 * kC + lt) * RS + lk) = gw;
            *reinterpret_cast<uint4 *>(raw + (1 * kC + lt) * RS + lk) = gq;
            *reinterpret_cast<uint4 *>(raw + (2 * kC + lt) * RS + lk) = gk;
            *reinterpret_cast<uint4 *>(raw + (3 * kC + lt) * RS + lk) = ga;
            *reinterpret_cast<uint4 *>(raw + (4 * kC + lt) * RS + lk) = gb;
            *reinterpret_cast<RawVec *>(raw + 5 * kC * RS + lt * RSV + lv) = gv.r;
        };
        auto clampc = [&](int c) { return c < c1 ? c : c1 - 1; };
        auto load_tm = [&](int c) {
            return *reinterpret_cast<const float4 *>(tinv_ + ((long)bh * nc + clampc(c)) * kC * kC + ltid * 4);
        };
        // values of the chunk in flight (first half -> second half of its prologue)
        float ksL[8], bsL[8], gamL[8];
        uint4 pq[2], pa[2];
        Raw4<bf16_t> rv;
        // staged rows in the compute mapping; decay logarithm, its inclusive prefix sum over the chunk (DPP), scaled operands;
        // q~ and a~ already as bf16 hi/lo pairs (k^, b^ are split in interval a: the producer's two halves are equally long that way)
        auto first_half = [&]() {
            float lw[8], Gc[8], wr[8], qv[8], kv[8], av[8], bv[8];
            {
                const uint4 rw = *reinterpret_cast<const uint4 *>(raw + (0 * kC + pt) * RS + pk);
                const uint4 rq = *reinterpret_cast<const uint4 *>(raw + (1 * kC + pt) * RS + pk);
                const uint4 rk = *reinterpret_cast<const uint4 *>(raw + (2 * kC + pt) * RS + pk);
                const uint4 ra = *reinterpret_cast<const uint4 *>(raw + (3 * kC + pt) * RS + pk);
                const uint4 rb = *reinterpret_cast<const uint4 *>(raw + (4 * kC + pt) * RS + pk);
                rv.r = *reinterpret_cast<const RawVec *>(raw + 5 * kC * RS + pt * RSV + pv);
                auto cvt8u = [](const uint4 r, float (&f)[8]) {
                    f[0] = __uint_as_float(r.x << 16); f[1] = __uint_as_float(r.x & 0xffff0000u);
                    f[2] = __uint_as_float(r.y << 16); f[3] = __uint_as_float(r.y & 0xffff0000u);
                    f[4] = __uint_as_float(r.z << 16); f[5] = __uint_as_float(r.z & 0xffff0000u);
                    f[6] = __uint_as_float(r.w << 16); f[7] = __uint_as_float(r.w & 0xffff0000u);
                };
                cvt8u(rw, wr); cvt8u(rq, qv); cvt8u(rk, kv); cvt8u(ra, av); cvt8u(rb, bv);
            }
#pragma unroll
            for (int j = 0; j < 8; j++) lw[j] = -fast_exp(wr[j]);
#pragma unroll
            for (int j = 0; j < 8; j++) Gc[j] = scan32(lw[j]);
            float qs[8], as_[8];
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const float gam = fast_exp(Gc[j]), gprev = prev32(gam, 1.f, ltid), ig = fast_exp(-Gc[j]);
                qs[j] = qv[j] * gam;
                as_[j] = av[j] * gprev;
                ksL[j] = kv[j] * ig;
                bsL[j] = bv[j] * ig;
                gamL[j] = gam;
            }
            uint32_t qh[4], ql[4], ah[4], al[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                split_pk(qs[2 * j], qs[2 * j + 1], qh[j], ql[j]);
                split_pk(as_[2 * j], as_[2 * j + 1], ah[j], al[j]);
            }
            pq[0] = make_uint4(qh[0], qh[1], qh[2], qh[3]); pq[1] = make_uint4(ql[0], ql[1], ql[2], ql[3]);
            pa[0] = make_uint4(ah[0], ah[1], ah[2], ah[3]); pa[1] = make_uint4(al[0], al[1], al[2], al[3]);
        };
        issue(c0);
        stage_raw();
        issue(clampc(c0 + 1));
        float4 tmreg = load_tm(c0);
        lds_barrier();
        first_half();   // chunk c0
        lds_barrier();  // staging is rewritten in the first interval a
        for (int it = c0 - 1; it < c1; it++) {
            const int cc = it, pc = it + 1;
            uint16_t *bufp = sm + (pc & 1) * L::BUF;
            float *gCp = sh_gC2 + (pc & 1) * kN;
            // ----------------------------------------------------------------------------------------------- interval a
            if (pc < c1) {
                // operand planes of chunk pc (q~, a~ split in the previous interval b; k^, b^ here), V, g_C, T planes
                const float4 v0 = cvt4(rv);
                uint32_t kh[4], kl[4], bhh[4], bl[4];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    split_pk(ksL[2 * j], ksL[2 * j + 1], kh[j], kl[j]);
                    split_pk(bsL[2 * j], bsL[2 * j + 1], bhh[j], bl[j]);
                }
                const int o = pt * LDK + pk;
                *reinterpret_cast<uint4 *>(&bufp[L::QTh + o]) = pq[0];
                *reinterpret_cast<uint4 *>(&bufp[L::QTl + o]) = pq[1];
                *reinterpret_cast<uint4 *>(&bufp[L::ATh + o]) = pa[0];
                *reinterpret_cast<uint4 *>(&bufp[L::ATl + o]) = pa[1];
                *reinterpret_cast<uint4 *>(&bufp[L::KHh + o]) = make_uint4(kh[0], kh[1], kh[2], kh[3]);
                *reinterpret_cast<uint4 *>(&bufp[L::KHl + o]) = make_uint4(kl[0], kl[1], kl[2], kl[3]);
                *reinterpret_cast<uint4 *>(&bufp[L::BHh + o]) = make_uint4(bhh[0], bhh[1], bhh[2], bhh[3]);
                *reinterpret_cast<uint4 *>(&bufp[L::BHl + o]) = make_uint4(bl[0], bl[1], bl[2], bl[3]);
                *reinterpret_cast<uint2 *>(&bufp[L::Vt + pt * LDC + pv]) = make_uint2(cvt_pk(v0.x, v0.y), cvt_pk(v0.z, v0.w));  // bf16 v: exact
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
            }
            // rows of chunk pc + 1 (requested one iteration ago) -> staging; T of pc + 1 and rows of pc + 2 requested (indices
            // clamped to the last chunk: a load under a condition is merged with the old value by a phi and waited for at the
            // loop header)
            stage_raw();
            __builtin_amdgcn_sched_barrier(0);
            tmreg = load_tm(pc + 1);
            issue(clampc(pc + 2));
            __builtin_amdgcn_sched_barrier(0);
            F9DRAIN();
            F9STAMP(1);
            lds_barrier();
            F9STAMP(2);
            // ----------------------------------------------------------------------------------------------- interval b
            if (pc + 1 < c1) first_half();   // chunk pc + 1
            F9STAMP(5);
            lds_barrier();
            F9STAMP(6);
        }
    }
#ifdef WKV7C_TIMING
    if (blockIdx.x == 0 && lane == 0)
        for (int i = 0; i < 8; i++) g_cfwd9_timing[(tid >> 6) * 8 + i] += tacc_[i];
#endif
}

static int launch_fwd9(bool save, int B, int T_, int H, const void *w, const void *q, const void *k, const void *v, const void *a,
                       const void *b, const float *tinv, void *y, float *sa, void *hs, const int *seq_off, int nseq, hipStream_t st) {
    static DynLdsOnce lds_once, lds_once2;
    if (hipError_t e = lds_once.ensure(reinterpret_cast<const void *>(&wkv7c_fwd9_kernel<true>), (int)F9Smem::bytes); e != hipSuccess) return (int)e;
    if (hipError_t e = lds_once2.ensure(reinterpret_cast<const void *>(&wkv7c_fwd9_kernel<false>), (int)F9Smem::bytes); e != hipSuccess) return (int)e;
    (void)hipGetLastError();
    const dim3 grid((seq_off ? nseq : B) * H * 2), block(512);
    if (save)
        hipLaunchKernelGGL(wkv7c_fwd9_kernel<true>, grid, block, F9Smem::bytes, st, T_, H, (const bf16_t *)w, (const bf16_t *)q,
                           (const bf16_t *)k, (const bf16_t *)v, (const bf16_t *)a, (const bf16_t *)b, tinv, (bf16_t *)y, sa, (uint16_t *)hs, seq_off);
    else
        hipLaunchKernelGGL(wkv7c_fwd9_kernel<false>, grid, block, F9Smem::bytes, st, T_, H, (const bf16_t *)w, (const bf16_t *)q,
                           (const bf16_t *)k, (const bf16_t *)v, (const bf16_t *)a, (const bf16_t *)b, tinv, (bf16_t *)y, nullptr, nullptr,
                           seq_off);
    return (int)hipGetLastError();
}

int chunk_fwd9_bf16(int B, int T_, int H, const void *w, const void *q, const void *k, const void *v, const void *a, const void *b,
                    const float *tinv, void *y, float *sa, void *hs, const int *seq_off, int nseq, hipStream_t st) {
    return launch_fwd9(sa && hs, B, T_, H, w, q, k, v, a, b, tinv, y, sa, hs, seq_off, nseq, st);
}

}  // namespace rwkv7

#ifdef WKV7C_TIMING
extern "C" int rwkv7_debug_cfwd9_timing(long long *out, int reset) {
    if (reset) {
        long long z[64] = {0};
        return (int)hipMemcpyToSymbol(HIP_SYMBOL(rwkv7::g_cfwd9_timing), z, sizeof(z));
    }
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(rwkv7::g_cfwd9_timing), sizeof(long long) * 64);
}
#endif
