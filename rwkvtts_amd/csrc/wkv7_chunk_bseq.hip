// rwkvtts_amd/csrc/wkv7_chunk_bseq.hip -- adjoint-state recurrence of the chunked (MFMA) WKV7 backward in ONE kernel, bf16 tensors.
//
// Reference: wkv7_cuda.cu:54-130.  Replaced round 2's wkv7c_bwd_pre_kernel + wkv7c_state_kernel (removed in round 4: git log --
// rwkvtts_amd/csrc/wkv7_chunk_bwd.hip).  Those two materialised
// M_c^T and N'_c (two 64 x 64 matrices per chunk, 18 KB of q15 records written by one kernel and read by the next: 0.6 GB per layer
// at B=8, T=4096, H=16) so that the sequential kernel is one product per chunk.  Here the recurrence
//     E_c = M_c^T E_{c+1} + N'_c ,   M_c = diag(g_C)(I + B^^T T A~) ,   N'_c = Q~^T dY + (T A~)^T (A_qb^T dY)
// is applied in factored form, the mirror image of the forward kernel wkv7c_fwd9_kernel (wkv7_chunk_fwd9.hip), chunks descending:
//     E' = g_C E_{c+1}
//     Z  = B" E' + X" dY        B" = T^T B^ (32 x 64),  X" = T^T A_qb^T (32 x 32)   -- no state in them: made one chunk ahead
//     E_c = E' + A~^T Z + Q~^T dY
// (Z_t = dL/du_t, the same Z the per-chunk gradient kernel uses).  Two dependent products per chunk, nothing but the raw rows read
// and nothing but the E records written: 20 KB in (16 of them shared by the two workgroups of a head through L2) + 9 KB out per
// chunk against 43 + 28.
//     interval a   wave 0: Z = B" E' + X" dY -> Z planes            wave 3: A_qb of the NEXT chunk (c - 1)
//                  waves 1,2: record of E_{c+1} -> e_vk[c] (and the Z tile of the previous chunk -> global)
//                  waves 4-7 (producer): hi/lo splits; operand planes q~, a~, b^, dY, g_C of the chunk after the next (c - 2): THREE
//                                        plane buffers since round 4, so that A_qb no longer waits for planes written in the same
//                                        interval; T planes of c - 1; raw rows -> LDS staging; next global prefetch
//     interval b   waves 1,2: E_c (two key tiles); E' planes of the next chunk
//                  wave 0: Z -> staging tile, B" key tile 0 of the next chunk      wave 3: X", B" key tile 1 of the next chunk
//                  waves 4-7: rows of chunk c - 2 in the compute mapping: exp, prefix sums, scaling
// One workgroup per (head, half of the value columns): the value columns of E never mix.
// Measured (tools/cbseq_timing.py, B=8, T=4096, H=16): round 3: 3.8k cycles per chunk = 1.9k + 1.9k, 0.22-0.24 ms with the Z store
// (0.29 ms for the pair of round 2).  Round 4 (same-box A/B, tools/ab_kernel.py): -6.3 % from the three-buffer pipeline above -- the
// intervals are bounded by the wave with the most state-INDEPENDENT work and its SIMD partner (waves w and w + 4 share a SIMD and the
// older wave wins the VALU arbitration), not by the chain product, so work was moved until the SIMDs are even.
// Tried and dropped: eight producer waves (768 threads, 4 channels per producer thread,
// three waves per SIMD, K = 64 products in two halves to stay inside 168 registers) -- the producer's share of each interval
// shrinks (1.5-1.9k -> 1.1-1.7k) but the chain waves, now sharing their SIMD with two producer waves, slow down by as much:
// 3.9k cycles per chunk; B" on the waves that compute E_c instead of wave 0: no change; an LDS flag instead of the barrier for the
// X" hand-off inside interval b: +4.2 % (profiles/experiments_r04/bseq_xpp_relocation.patch).
#include "chunk_common.h"
#include "launch_attr.h"

namespace rwkv7 {

#ifdef WKV7C_TIMING
__device__ long long g_cbseq_timing[8 * 4];
#define BSSTAMP(i)                                              \
    do {                                                        \
        const long long now_ = __builtin_readcyclecounter();    \
        tacc_[i] += now_ - tprev_;                              \
        tprev_ = now_;                                          \
    } while (0)
#else
#define BSSTAMP(i) do { } while (0)
#endif

namespace {
constexpr int LDK = kN + kPad;  // planes with K = 64 columns
constexpr int LDC = kC + kPad;  // planes with K = 32 columns
constexpr int VH = 32;          // value columns per workgroup

struct BSSmem {  // offsets in uint16 units; every plane 16-byte aligned
    static constexpr int PL = kC * LDK, PS = kC * LDC;
    // one producer buffer: six scaled operand planes, time-major, and dY[t][v]
    static constexpr int QTh = 0, QTl = PL, ATh = 2 * PL, ATl = 3 * PL, BHh = 4 * PL, BHl = 5 * PL, DYt = 6 * PL;
    static constexpr int BUF = 6 * PL + PS;
    // single: state planes E'[v][k], B"[r][k], and the 32 x 32 matrices X"[r][t], A_qb[t][s], T[t][r], Z[v][r]
    // THREE producer buffers (round 4: the planes of a chunk land one iteration earlier, so that A_qb of the next chunk can be formed in
    // interval a -- by wave 3, idle there until now -- and only X" is left behind it in interval b)
    static constexpr int NBUF = 3;
    static constexpr int Eh = NBUF * BUF, El = Eh + VH * LDK;
    static constexpr int BBh = El + VH * LDK, BBl = BBh + PL;
    static constexpr int XPh = BBl + PL, XPl = XPh + PS, QBh = XPl + PS, QBl = QBh + PS, TMh = QBl + PS, TMl = TMh + PS;
    static constexpr int Zh = TMl + PS, Zl = Zh + VH * LDC;
    static constexpr int end16 = Zl + VH * LDC;
    static constexpr int fGC = 0, fZ = NBUF * kN, fend = fZ + kC * 36;   // fp32: g_C of the three buffers; Z staging tile [32][36]
    // raw input staging (bf16): 4 planes [32][64 + 8] and dY [32][32 + 8]
    static constexpr int RS = kN + 8, RSV = VH + 8;
    static constexpr size_t bytes = (size_t)end16 * 2 + (size_t)fend * 4 + (size_t)(4 * kC * RS + kC * RSV) * 2;
};
static_assert(BSSmem::end16 % 8 == 0 && BSSmem::BUF % 8 == 0 && BSSmem::BBh % 8 == 0 && BSSmem::XPh % 8 == 0, "16-byte alignment");
static_assert(BSSmem::bytes <= 160 * 1024, "LDS budget");
}  // namespace

__global__ __launch_bounds__(512) void wkv7c_bseq_kernel(int T_, int H, const bf16_t *__restrict__ w_, const bf16_t *__restrict__ q_,
                                                         const bf16_t *__restrict__ a_, const bf16_t *__restrict__ b_,
                                                         const bf16_t *__restrict__ dy_, const float *__restrict__ tinv_,
                                                         uint16_t *__restrict__ e_vk, float *__restrict__ z_,
                                                         const int *__restrict__ seq_off_) {
    extern __shared__ __attribute__((aligned(16))) uint16_t sm[];
    using L = BSSmem;
    float *sh_gC2 = reinterpret_cast<float *>(sm + L::end16) + L::fGC, *sh_Z = reinterpret_cast<float *>(sm + L::end16) + L::fZ;
    constexpr int kStageLD = 36;
    bf16_t *raw = reinterpret_cast<bf16_t *>(reinterpret_cast<float *>(sm + L::end16) + L::fend);
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
    const int pt = ltid & 31, pk = (ltid >> 5) * 8, pv = (ltid >> 5) * 4;
    const int lt = ltid >> 3, lk = (ltid & 7) * 8, lv = (ltid & 7) * 4;

    for (int i = tid; i < 2 * VH * LDK; i += 512) sm[L::Eh + i] = 0;  // E after the last chunk is zero
    using RawVec = decltype(Raw4<bf16_t>::r);
#ifdef WKV7C_TIMING
    long long tacc_[4] = {0, 0, 0, 0};
    long long tprev_ = __builtin_readcyclecounter();
#endif

    // Iteration `it` (descending): the consumer works on chunk cc = it, the producer finishes chunk pc = it - 1 and starts pc - 1.
    if (role == 0) {
        // =================================================================================================== consumer
        f32x16 Emaster = zero16();  // waves 1, 2: D-layout tile (32 keys x 32 value columns) of E_{cc+1}, fp32, not yet decayed
        lds_barrier();
        lds_barrier();
        lds_barrier();
        lds_barrier();
        for (int it = c1; it >= c0; it--) {
            const int cc = it, pc = it - 1;
            const int bc = cc % L::NBUF, bp = (pc + L::NBUF) % L::NBUF;
            const uint16_t *bufc = sm + bc * L::BUF, *bufp = sm + bp * L::BUF;
            const float *gCc = sh_gC2 + bc * kN, *gCp = sh_gC2 + bp * kN;
            // ----------------------------------------------------------------------------------------------- interval a
            if (z_ && cc + 1 < c1) {
                // Z of the previous chunk (staged in its interval b) -> HBM, fp32 [B,T,H,64] like sa: thread (pt, pv) owns 4 value
                // columns of one step.  The per-chunk gradient kernel reads it instead of rebuilding A_qb, G1 and Z.
                const long o = head_base + (long)((cc + 1) * kC + pt) * tstride + vh * VH + pv;
                *reinterpret_cast<float4 *>(z_ + o) = *reinterpret_cast<const float4 *>(&sh_Z[pt * kStageLD + pv]);
            }
            // what chunk cc receives from its future: q15 record straight from the accumulator tile (here, not in interval b: waves
            // 1 and 2 have nothing else to do while wave 0 forms Z, and the record is not on the chain)
            if (cc < c1 && (wave == 1 || wave == 2)) q15_encode_tile(Emaster, e_vk + ((long)bh * nc + cc) * kQRec, vh, wave - 1, lane);
            f32x16 accZ = zero16();
            if (cc < c1 && wave == 0) {   // Z = B" E' + X" dY : D[r][v] -> Z[v][r]
                mma_tile3<kN>(accZ, sm + L::BBh, sm + L::BBl, LDK, sm + L::Eh, sm + L::El, LDK, lane);
                mma_gen<kC, false, true, true, false>(accZ, sm + L::XPh, sm + L::XPl, LDC, 0, bufc + L::DYt, bufc + L::DYt, LDC, 0, lane);
                store_T_split(accZ, sm + L::Zh, sm + L::Zl, LDC, lane);
            }
            if (wave == 3 && pc >= c0) {
                // next chunk: A_qb[t][s] (s <= t) from its planes, which landed an iteration ago (X" = T^T A_qb^T follows in interval b,
                // when T of that chunk has landed)
                f32x16 acc = zero16();  // D[m = s][n = t] = b^_s . q~_t
                mma_tile3<kN, 2>(acc, bufp + L::BHh, bufp + L::BHl, LDK, bufp + L::QTh, bufp + L::QTl, LDK, lane);
                mask_lower_T<false>(acc, lane);
                store_T_split(acc, sm + L::QBh, sm + L::QBl, LDC, lane);
            }
            BSSTAMP(0);
            lds_barrier();
            BSSTAMP(1);
            // ----------------------------------------------------------------------------------------------- interval b
            if (wave == 1 || wave == 2) {
                const int kt = wave - 1;  // key channels [32 kt, 32 kt + 32)
                if (cc < c1) {
                    f32x16 acc = zero16();  // D[m = k][n = v] = sum_r a~[r][k] Z[r][v] + sum_t q~[t][k] dY[t][v]
                    mma_gen<kC, true, true, false, true>(acc, bufc + L::ATh, bufc + L::ATl, LDK, kt * 32, sm + L::Zh, sm + L::Zl, LDC, 0, lane);
                    mma_gen<kC, true, true, true, false>(acc, bufc + L::QTh, bufc + L::QTl, LDK, kt * 32, bufc + L::DYt, bufc + L::DYt, LDC, 0, lane);
#pragma unroll
                    for (int r = 0; r < 16; r++) Emaster[r] = gCc[kt * 32 + d_row(r, lane)] * Emaster[r] + acc[r];
                }
                if (pc >= c0) {
                    // E' = g_C E for the next chunk (its g_C arrived in interval a), planes E'[v][k]
                    f32x16 Ep;
#pragma unroll
                    for (int r = 0; r < 16; r++) Ep[r] = gCp[kt * 32 + d_row(r, lane)] * Emaster[r];
                    store_T_split(Ep, sm + L::Eh + kt * 32, sm + L::El + kt * 32, LDK, lane);
                }
            } else if (wave == 0) {
                if (z_ && cc < c1) {
#pragma unroll
                    for (int r = 0; r < 16; r++) sh_Z[d_row(r, lane) * kStageLD + (lane & 31)] = accZ[r];
                }
                if (pc >= c0) {
                    // B" = T^T B^ of the next chunk, key tile 0: D[m = k][n = r] = sum_s b^[s][k] T[s][r] -> B"[r][k]
                    f32x16 acc = zero16();
                    mma_gen<kC, true, true, true, true>(acc, bufp + L::BHh, bufp + L::BHl, LDK, 0, sm + L::TMh, sm + L::TMl, LDC, 0, lane);
                    store_T_split(acc, sm + L::BBh, sm + L::BBl, LDK, lane);
                }
            } else if (wave == 3 && pc >= c0) {
                // next chunk: X"[r][t] = sum_s T[s][r] A_qb[t][s] (A_qb: this wave, interval a) and key tile 1 of B"
                f32x16 acx = zero16();  // D[m = t][n = r]
                mma_gen<kC, false, true, true, true>(acx, sm + L::QBh, sm + L::QBl, LDC, 0, sm + L::TMh, sm + L::TMl, LDC, 0, lane);
                store_T_split(acx, sm + L::XPh, sm + L::XPl, LDC, lane);
                f32x16 acb = zero16();
                mma_gen<kC, true, true, true, true>(acb, bufp + L::BHh, bufp + L::BHl, LDK, 32, sm + L::TMh, sm + L::TMl, LDC, 0, lane);
                store_T_split(acb, sm + L::BBh + 32, sm + L::BBl + 32, LDK, lane);
            }
            BSSTAMP(2);
            lds_barrier();
            BSSTAMP(3);
        }
        if (z_) {   // Z of the last chunk processed (c0)
            const long o = head_base + (long)(c0 * kC + pt) * tstride + vh * VH + pv;
            *reinterpret_cast<float4 *>(z_ + o) = *reinterpret_cast<const float4 *>(&sh_Z[pt * kStageLD + pv]);
        }
    } else {
        // =================================================================================================== producer
        uint4 gw, gq, ga, gb;
        Raw4<bf16_t> gdy;
        auto issue = [&](int c) {   // unconditional: the chunk index is clamped by the caller
            const long off = head_base + (long)(c * kC + lt) * tstride;
            gw = *reinterpret_cast<const uint4 *>(w_ + off + lk);
            gq = *reinterpret_cast<const uint4 *>(q_ + off + lk);
            ga = *reinterpret_cast<const uint4 *>(a_ + off + lk);
            gb = *reinterpret_cast<const uint4 *>(b_ + off + lk);
            gdy = ld4<bf16_t>(dy_ + off + vh * VH + lv, true);
        };
        auto stage_raw = [&]() {
            *reinterpret_cast<uint4 *>(raw + (0 * kC + lt) * RS + lk) = gw;
            *reinterpret_cast<uint4 *>(raw + (1 * kC + lt) * RS + lk) = gq;
            *reinterpret_cast<uint4 *>(raw + (2 * kC + lt) * RS + lk) = ga;
            *reinterpret_cast<uint4 *>(raw + (3 * kC + lt) * RS + lk) = gb;
            *reinterpret_cast<RawVec *>(raw + 4 * kC * RS + lt * RSV + lv) = gdy.r;
        };
        auto clampc = [&](int c) { return c > c0 ? c : c0; };
        auto load_tm = [&](int c) {
            return *reinterpret_cast<const float4 *>(tinv_ + ((long)bh * nc + clampc(c)) * kC * kC + ltid * 4);
        };
        float qsL[8], asL[8], bsL[8], gamL[8];
        RawVec rdy;
        auto first_half = [&]() {
            float lw[8], Gc[8], wr[8], qv[8], av[8], bv[8];
            {
                const uint4 rw = *reinterpret_cast<const uint4 *>(raw + (0 * kC + pt) * RS + pk);
                const uint4 rq = *reinterpret_cast<const uint4 *>(raw + (1 * kC + pt) * RS + pk);
                const uint4 ra = *reinterpret_cast<const uint4 *>(raw + (2 * kC + pt) * RS + pk);
                const uint4 rb = *reinterpret_cast<const uint4 *>(raw + (3 * kC + pt) * RS + pk);
                rdy = *reinterpret_cast<const RawVec *>(raw + 4 * kC * RS + pt * RSV + pv);
                auto cvt8u = [](const uint4 r, float (&f)[8]) {
                    f[0] = __uint_as_float(r.x << 16); f[1] = __uint_as_float(r.x & 0xffff0000u);
                    f[2] = __uint_as_float(r.y << 16); f[3] = __uint_as_float(r.y & 0xffff0000u);
                    f[4] = __uint_as_float(r.z << 16); f[5] = __uint_as_float(r.z & 0xffff0000u);
                    f[6] = __uint_as_float(r.w << 16); f[7] = __uint_as_float(r.w & 0xffff0000u);
                };
                cvt8u(rw, wr); cvt8u(rq, qv); cvt8u(ra, av); cvt8u(rb, bv);
            }
#pragma unroll
            for (int j = 0; j < 8; j++) lw[j] = -fast_exp(wr[j]);
#pragma unroll
            for (int j = 0; j < 8; j++) Gc[j] = scan32(lw[j]);
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const float gam = fast_exp(Gc[j]), gprev = prev32(gam, 1.f, ltid), ig = fast_exp(-Gc[j]);
                qsL[j] = qv[j] * gam;
                asL[j] = av[j] * gprev;
                bsL[j] = bv[j] * ig;
                gamL[j] = gam;
            }
        };
        // operand planes, dY and g_C of chunk c (whose scaled rows first_half() left in registers) -> buffer c % 3
        auto write_planes = [&](int c) {
            uint16_t *bufw = sm + (c % L::NBUF) * L::BUF;
            float *gCw = sh_gC2 + (c % L::NBUF) * kN;
            const int o = pt * LDK + pk;
            auto put = [&](const float (&x)[8], int ph, int pl) {
                uint32_t h[4], l[4];
#pragma unroll
                for (int j = 0; j < 4; j++) split_pk(x[2 * j], x[2 * j + 1], h[j], l[j]);
                *reinterpret_cast<uint4 *>(&bufw[ph + o]) = make_uint4(h[0], h[1], h[2], h[3]);
                *reinterpret_cast<uint4 *>(&bufw[pl + o]) = make_uint4(l[0], l[1], l[2], l[3]);
            };
            put(qsL, L::QTh, L::QTl);
            put(asL, L::ATh, L::ATl);
            put(bsL, L::BHh, L::BHl);
            *reinterpret_cast<RawVec *>(&bufw[L::DYt + pt * LDC + pv]) = rdy;  // bf16 dY: exact
            if (pt == kC - 1) {
#pragma unroll
                for (int j = 0; j < 8; j++) gCw[pk + j] = gamL[j];
            }
        };
        // Two chunks of lead (four barriers before the loop, like the consumer): planes of c1 - 1 written, rows of c1 - 2 scaled
        issue(c1 - 1);
        stage_raw();
        issue(clampc(c1 - 2));
        float4 tmreg = load_tm(c1 - 1);
        lds_barrier();
        first_half();   // chunk c1 - 1
        lds_barrier();  // staging is rewritten below
        write_planes(c1 - 1);
        stage_raw();    // rows of chunk c1 - 2
        issue(clampc(c1 - 3));
        lds_barrier();
        if (c1 - 2 >= c0) first_half();   // chunk c1 - 2
        lds_barrier();
        for (int it = c1; it >= c0; it--) {
            const int pc = it - 1, pp = it - 2;   // T planes of pc, operand planes of pp land in this interval a
            // ----------------------------------------------------------------------------------------------- interval a
            if (pp >= c0) write_planes(pp);
            if (pc >= c0) {
                {   // T planes Tm[t][r] of chunk pc: thread = row ltid >> 3, columns 4 (ltid & 7) .. +4
                    uint32_t h0, l0, h1, l1;
                    split_pk(tmreg.x, tmreg.y, h0, l0);
                    split_pk(tmreg.z, tmreg.w, h1, l1);
                    const int ot = (ltid >> 3) * LDC + (ltid & 7) * 4;
                    *reinterpret_cast<uint2 *>(&sm[L::TMh + ot]) = make_uint2(h0, h1);
                    *reinterpret_cast<uint2 *>(&sm[L::TMl + ot]) = make_uint2(l0, l1);
                }
            }
            stage_raw();   // rows of chunk pp - 1 (requested one iteration ago)
            __builtin_amdgcn_sched_barrier(0);
            tmreg = load_tm(pc - 1);
            issue(clampc(pp - 2));
            __builtin_amdgcn_sched_barrier(0);
            BSSTAMP(0);
            lds_barrier();
            BSSTAMP(1);
            // ----------------------------------------------------------------------------------------------- interval b
            if (pp - 1 >= c0) first_half();   // chunk pp - 1
            BSSTAMP(2);
            lds_barrier();
            BSSTAMP(3);
        }
    }
#ifdef WKV7C_TIMING
    if (blockIdx.x == 0 && lane == 0)
        for (int i = 0; i < 4; i++) g_cbseq_timing[(tid >> 6) * 4 + i] += tacc_[i];
#endif
}

int chunk_bseq_bf16(int B, int T_, int H, const void *w, const void *q, const void *a, const void *b, const void *dy, const float *tinv,
                    void *e_vk, float *z, const int *seq_off, int nseq, hipStream_t st) {
    static DynLdsOnce lds_once;
    if (hipError_t e = lds_once.ensure(reinterpret_cast<const void *>(&wkv7c_bseq_kernel), (int)BSSmem::bytes); e != hipSuccess) return (int)e;
    (void)hipGetLastError();
    hipLaunchKernelGGL(wkv7c_bseq_kernel, dim3((seq_off ? nseq : B) * H * 2), dim3(512), BSSmem::bytes, st, T_, H, (const bf16_t *)w,
                       (const bf16_t *)q, (const bf16_t *)a, (const bf16_t *)b, (const bf16_t *)dy, tinv, (uint16_t *)e_vk, z, seq_off);
    return (int)hipGetLastError();
}

}  // namespace rwkv7

#ifdef WKV7C_TIMING
extern "C" int rwkv7_debug_cbseq_timing(long long *out, int reset) {
    if (reset) {
        long long z[32] = {0};
        return (int)hipMemcpyToSymbol(HIP_SYMBOL(rwkv7::g_cbseq_timing), z, sizeof(z));
    }
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(rwkv7::g_cbseq_timing), sizeof(long long) * 32);
}
#endif
