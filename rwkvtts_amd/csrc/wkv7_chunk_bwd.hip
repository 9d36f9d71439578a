// rwkvtts_amd/csrc/wkv7_chunk_bwd.hip -- chunked (MFMA) WKV7 backward for gfx950, bf16 tensors.
//
// Same gradients as wkv7_bwd.hip (reference wkv7_cuda.cu:54-130), evaluated 32 steps at a time on the matrix cores.
// With H = S^T in R^{K x V} and the per-chunk quantities of chunk_common.h (q~, a~, k^, b^, g_C, T = (I - A_ab)^-1,
// W = T A~) the forward state obeys  H_{c+1} = M_c H_c + N_c  with  M_c = diag(g_C)(I + B^^T W), and the adjoint state
//     E_c = M_c^T E_{c+1} + N'_c ,   N'_c = Q~^T dY + W^T (A_qb^T dY)              (E_c = dL/dH at the START of chunk c)
// is the only sequential object of the backward pass (tests/chunked_proto2.py validates the algebra against the oracle).
// Three kernels:
//   wkv7c_bwd_pre_kernel   grid B*H*(T/32), parallel: M_c^T (q15 record, A-fragment order) and N'_c (q15 record, accumulator order)
//   wkv7c_state_kernel     grid B*H, sequential over chunks in reverse: E for every chunk (one 64x64x64 product each)
//   wkv7c_bwd_out8_kernel  grid B*H*(T/32), parallel: dw,dq,dk,dv,da,db of a chunk from (H_c, E_{c+1}, U = sa, dY)
//                          (wkv7_chunk_bwd8.hip)
// Inputs that come from the chunked forward pass (wkv7_chunk_fwd*.hip): sa (= U, fp32), hs = the state at the start of every
// chunk (bf16, [v][k]) and T from wkv7c_prep_kernel.  Both per-chunk states that cross kernels -- hs and the adjoint E -- are
// stored as the bf16 HI plane of their hi/lo pair, [v][k]: the recurrences themselves carry hi + lo (~16 mantissa bits), only
// the one-shot use inside a chunk sees the rounded copy, so nothing accumulates; 8 KB per chunk and state instead of 16.
#include "chunk_bwd_common.h"

namespace rwkv7 {

namespace {
constexpr int kChunksPerWG = 8;     // bwd_pre walks this many consecutive chunks per workgroup, prefetching the next one's inputs

struct PreSmem {  // offsets in uint16 units
    // the chunk's operands, every one stored ONCE, time-major [t][.] as its rows arrive: products that contract over time fetch
    // them with the LDS transpose read (mma_gen, operand "k-major") -- the channel-major copies of q~, a~, b^ g_C and dY cost 72
    // two-byte scattered LDS writes per thread and chunk (26 % of the kernel's cycles on the LDS pipe)
    static constexpr int TM1 = kC * LDK;
    static constexpr int QTh = 0, QTl = QTh + TM1, BHh = QTl + TM1, BHl = BHh + TM1, ATh = BHl + TM1, ATl = ATh + TM1;
    static constexpr int BCh = ATl + TM1, BCl = BCh + TM1, DY = BCl + TM1;
    static constexpr int TMh = DY + TM1, TMl = TMh + kC * LDC;                 // T
    static constexpr int QBTh = TMl + kC * LDC, QBTl = QBTh + kC * LDC;        // A_qb^T
    static constexpr int WTh = QBTl + kC * LDC, WTl = WTh + kN * LDC;          // W^T [k][t]
    static constexpr int G1Th = WTl + kN * LDC, G1Tl = G1Th + kN * LDC;        // G1^T [v][s]
    static constexpr int gC = G1Tl + kN * LDC;  // 64 floats
    static constexpr int end16 = gC + 2 * kN;
    static constexpr size_t bytes = (size_t)end16 * 2;
};
static_assert(PreSmem::TMh % 8 == 0 && PreSmem::QBTh % 8 == 0 && PreSmem::WTh % 8 == 0 && PreSmem::G1Th % 8 == 0 && PreSmem::gC % 8 == 0,
              "16-byte alignment");
static_assert(PreSmem::bytes <= 80 * 1024, "two workgroups per CU");

}  // namespace

// ------------------------------------------------------------------------------------------------------------------
// pre: M_c^T (q15 record whose tiles are MFMA A fragments: tile (k-tile, k'-tile) = [lane][k-steps 2 k'-tile, +1][8]) and N'_c
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void wkv7c_bwd_pre_kernel(int T_, int H, int nchunks_total, const bf16_t *__restrict__ w_,
                                                            const bf16_t *__restrict__ q_, const bf16_t *__restrict__ a_,
                                                            const bf16_t *__restrict__ b_, const bf16_t *__restrict__ dy_,
                                                            const float *__restrict__ tinv_, uint16_t *__restrict__ mt_,
                                                            uint16_t *__restrict__ np_) {
    extern __shared__ __attribute__((aligned(16))) uint16_t sm[];
    using L = PreSmem;
    float *sh_gC = reinterpret_cast<float *>(sm + L::gC);
    const int nc = T_ / kC;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: role branches must not become exec-masked regions
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
        // unconditional (clamped) prefetch: a conditional load makes the s_waitcnt at the top of the next iteration vmcnt(0), which
        // also waits for this iteration's record stores; the one wasted fetch per workgroup re-reads the last chunk (an L2 hit)
        const In nxt = load(chunk + 1 < nchunks_total && ci + 1 < kChunksPerWG ? chunk + 1 : chunk);
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
            const int o = pt * LDK + pk;
            put_row8(sm + L::QTh, sm + L::QTl, o, x, hi, lo);
#pragma unroll
            for (int j = 0; j < 8; j++) x[j] = bv[j] * fast_exp(-G[j]);                // b^ = b / gamma_t
            put_row8(sm + L::BHh, sm + L::BHl, o, x, hi, lo);
#pragma unroll
            for (int j = 0; j < 8; j++) x[j] *= sh_gC[pk + j];                          // b^ g_C (bounded by |b|)
            put_row8(sm + L::BCh, sm + L::BCl, o, x, hi, lo);
#pragma unroll
            for (int j = 0; j < 8; j++) x[j] = av[j] * fast_exp(G[j] - lw[j]);         // a~ = a gamma_{t-1}
            put_row8(sm + L::ATh, sm + L::ATl, o, x, hi, lo);
            *reinterpret_cast<uint4 *>(sm + L::DY + o) = cur.dy.r;                      // dY is bf16: exact, one plane
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
            f32x16 acc = zero16();  // D[m = t][n = k] = sum_s T[t][s] a~[s][k] (a~ time-major: transpose read); stored as WT[k][t]
            mma_gen<kC, false, true, true, true, 2>(acc, sm + L::TMh, sm + L::TMl, LDC, 0, sm + L::ATh, sm + L::ATl, LDK, kt * 32, lane);
            store_T_split(acc, sm + L::WTh + kt * 32 * LDC, sm + L::WTl + kt * 32 * LDC, LDC, lane);
        }
        lds_barrier();
        // ---- phase 2: G1 = A_qb^T dY (waves 0, 1); M^T = diag(g_C) + W^T (B^ g_C) (waves 2, 3) -----------------------
        if (wave <= 1) {
            const int vt = wave;
            f32x16 acc = zero16();  // D[m = s][n = v] = sum_t QBT[s][t] dY[t][v]; stored as G1T[v][s]
            mma_gen<kC, false, true, true, false, 2>(acc, sm + L::QBTh, sm + L::QBTl, LDC, 0, sm + L::DY, sm + L::DY, LDK, vt * 32, lane);
            store_T_split(acc, sm + L::G1Th + vt * 32 * LDC, sm + L::G1Tl + vt * 32 * LDC, LDC, lane);
        } else {
            const int mt = wave - 2;  // rows k' of the product below
#pragma unroll
            for (int nt = 0; nt < 2; nt++) {
                f32x16 acc = zero16();  // D[m = k'][n = k] = sum_t (b^ g_C)[t][k'] W[t][k]; stored as M^T[k][k']
                mma_gen<kC, true, true, false, true, 2>(acc, sm + L::BCh, sm + L::BCl, LDK, mt * 32, sm + L::WTh, sm + L::WTl, LDC, nt * 32, lane);
                if (mt == nt) {
                    const int n = lane & 31;
#pragma unroll
                    for (int r = 0; r < 16; r++) acc[r] += d_row(r, lane) == n ? sh_gC[mt * 32 + n] : 0.f;
                }
                // Lane (k, half h) holds M^T[k][k'] for k' = 32 mt + 8 q + 4 h + (0..3), q = 0..3 (registers 4q..4q+3); the state
                // kernel's A fragment of k-step i wants k' = 32 mt + 16 i + 8 h + (0..7).  v_permlane32_swap of register
                // groups q = 2i and q = 2i + 1 between the lane halves gives exactly that: afterwards acc[8i .. 8i+7] IS the
                // fragment of k-step 2 mt + i, and the tile goes out as a q15 record (tile = (k-tile nt, k'-tile mt)) without
                // touching LDS.
#pragma unroll
                for (int i = 0; i < 2; i++)
#pragma unroll
                    for (int t = 0; t < 4; t++) {
                        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[8 * i + t]), __float_as_uint(acc[8 * i + 4 + t]), false, false);
                        acc[8 * i + t] = __uint_as_float(sw[0]);
                        acc[8 * i + 4 + t] = __uint_as_float(sw[1]);
                    }
                q15_encode_tile(acc, mt_ + (long)chunk * kQRec, nt, mt, lane);
            }
        }
        lds_barrier();
        // ---- phase 3: N' = Q~^T dY + W^T G1 (one tile per wave, MFMA register layout) ----------------------------------------
        {
            const int mt = wave >> 1, nt = wave & 1;
            f32x16 acc = zero16();  // D[m = k][n = v]
            mma_gen<kC, true, true, true, false, 2>(acc, sm + L::QTh, sm + L::QTl, LDK, mt * 32, sm + L::DY, sm + L::DY, LDK, nt * 32, lane);
            mma_tile3<kC, 2>(acc, sm + L::WTh + mt * 32 * LDC, sm + L::WTl + mt * 32 * LDC, LDC, sm + L::G1Th + nt * 32 * LDC,
                          sm + L::G1Tl + nt * 32 * LDC, LDC, lane);
            // N' as a q15 record in accumulator order (chunk_common.h), tile = wave: 3 stores per lane, 9 KB instead of 16 KB fp32
            q15_encode_tile(acc, np_ + (long)chunk * kQRec, wave >> 1, wave & 1, lane);
        }
        lds_barrier();  // the next chunk's prologue overwrites what phase 3 reads
        cur = nxt;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// state: E_c = M_c^T E_{c+1} + N'_c, c = nc-1 .. 0; writes E_{c+1} (the adjoint state chunk c sees at its end) for every c
// as a q15 record (chunk_common.h) e_vk[b,h,c], straight from the accumulator tiles.  M^T arrives in A-fragment order and N' in accumulator
// order, both as q15 records (the step is bound by the bytes a CU can pull per cycle: 13.8 KB per workgroup and step instead of
// 20.5 KB with bf16 hi/lo planes), and go from registers straight into MFMA operands / accumulators; only E itself passes through LDS
// (accumulator layout -> B-operand planes).  The chain is one 64x64x64 product per chunk; inputs are prefetched three
// chunks ahead in registers so that the ~2 us HBM latency is off the critical path.
// ------------------------------------------------------------------------------------------------------------------
namespace {
constexpr int kStatePF = 4;   // recurrence steps whose inputs the helper waves keep in flight (registers)
struct StateSmem {  // offsets in uint16 units
    // E planes of this half of the value columns, [32 v][64 k] hi / lo, double buffered (B operand of the step's product)
    static constexpr int E0h = 0, E0l = E0h + kC * LDK, E1h = E0l + kC * LDK, E1l = E1h + kC * LDK;
    // what the helper waves hand to the product waves, per buffer and tile, lane-private (lane l writes what lane l reads):
    //   MF  M^T as bf16 A fragments  [plane hi/lo][k-step 4][lane 64][8]      NF  N' as fp32 accumulators [quarter 4][lane 64][4]
    // and what comes back:  EF  the new E tile, fp32 accumulators [quarter 4][lane 64][4] (becomes the q15 record of the next step)
    static constexpr int MFsz = 2 * 4 * 64 * 8, NFsz = 4 * 64 * 4 * 2;
    static constexpr int MF = E1l + kC * LDK;             // [buf 2][tile 2][MFsz]
    static constexpr int NF = MF + 4 * MFsz;              // [buf 2][tile 2][NFsz]
    static constexpr int EF = NF + 4 * NFsz;              // [buf 2][tile 2][NFsz]
    static constexpr int end16 = EF + 4 * NFsz;
    static constexpr size_t bytes = (size_t)end16 * 2;
};
static_assert(StateSmem::MF % 8 == 0 && StateSmem::NF % 8 == 0 && StateSmem::EF % 8 == 0, "16-byte alignment");
}  // namespace

#ifdef WKV7C_TIMING
// profiling build only (python -m rwkvtts_amd.build --timing): cycle totals of the segments of one recurrence step, workgroup
// 0, per wave (waves 0-1: product waves, 2-3: helpers).
__device__ long long g_cstate_timing[4 * 8];
#define STSTAMP(i)                                              \
    do {                                                        \
        const long long now_ = __builtin_readcyclecounter();    \
        tacc_[i] += now_ - tprev_;                              \
        tprev_ = now_;                                          \
    } while (0)
#else
#define STSTAMP(i) do { } while (0)
#endif

// One workgroup per (head, half of the value columns): the value columns of E never mix (E_c = M^T E + N' acts on columns).
// Waves 0-1 ("product") own the two 32-row tiles of that half and do nothing but the chain
//     E fragments, M^T fragments, N' from LDS -> 12 MFMAs -> sum -> E as fp32 + as hi/lo planes to LDS -> barrier;
// waves 2-3 ("helpers", one per tile) do everything that is not on that chain, one step ahead: the global prefetch ring, the
// decode of M^T / N' (q15 -> bf16 hi/lo fragments / fp32) and the q15 record of E (two helpers per tile measured no faster).  A lone wave issues one instruction per ~4
// cycles, and with everything on the product wave a step was ~330 VALU instructions = 2400 cycles whatever their order (three
// prefetch depths and MFMA-shadow scheduling all measured the same); the chain alone is ~130 issue slots.
__global__ __launch_bounds__(256) void wkv7c_state_kernel(int nc, int H, const uint16_t *__restrict__ mt_, const uint16_t *__restrict__ np_,
                                                          uint16_t *__restrict__ e_vk, const int *__restrict__ seq_off_) {
    extern __shared__ __attribute__((aligned(16))) uint16_t sm[];
    using L = StateSmem;
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
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: role branches must not become exec-masked regions
    const int mt = wave & 1;
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
    const int nsteps = c1 - c0;
    // E_{nc} = 0: planes of buffer 0 and the fp32 tile the first step's record is made from (EF buffer 1)
    for (int i = tid; i < 2 * kC * LDK; i += 256) sm[L::E0h + i] = 0;
    for (int i = tid; i < 2 * L::NFsz; i += 256) sm[L::EF + 2 * L::NFsz + i] = 0;
#ifdef WKV7C_TIMING
    long long tacc_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long tprev_ = __builtin_readcyclecounter();
#endif

    if (wave < 2) {
        // ------------------------------------------------------------------------------------------ product waves
        lds_barrier();   // helpers: fragments of the first step; zeros above
        for (int k = 0; k < nsteps; k++) {
            const int buf = k & 1;
            STSTAMP(0);
            const uint16_t *Eh = sm + (buf ? L::E1h : L::E0h) + (lane & 31) * LDK + (lane >> 5) * 8, *El = Eh + kC * LDK;
            const uint16_t *mf = sm + L::MF + (buf * 2 + mt) * L::MFsz + lane * 8;
            const float *nf = reinterpret_cast<const float *>(sm + L::NF + (buf * 2 + mt) * L::NFsz) + lane * 4;
            bf16x8 eh[4], el[4], mh[4], ml[4];
            f32x16 acc;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                mh[i] = *reinterpret_cast<const bf16x8 *>(mf + i * 512);
                eh[i] = *reinterpret_cast<const bf16x8 *>(Eh + 16 * i);
            }
#pragma unroll
            for (int i = 0; i < 4; i++) {
                el[i] = *reinterpret_cast<const bf16x8 *>(El + 16 * i);
                ml[i] = *reinterpret_cast<const bf16x8 *>(mf + 4 * 512 + i * 512);
            }
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const float4 t = *reinterpret_cast<const float4 *>(nf + q * 256);
                acc[4 * q] = t.x; acc[4 * q + 1] = t.y; acc[4 * q + 2] = t.z; acc[4 * q + 3] = t.w;
            }
            STSTAMP(1);
            // three independent accumulator chains (one per hi/lo term), the first one starting from N'
            f32x16 acc_b = zero16(), acc_c = zero16();
#pragma unroll
            for (int i = 0; i < 4; i++) {  // D[m = k][n = v] += sum_k' M^T[k][k'] E[k'][v]
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(mh[i], eh[i], acc, 0, 0, 0);
                acc_b = __builtin_amdgcn_mfma_f32_32x32x16_bf16(mh[i], el[i], acc_b, 0, 0, 0);
                acc_c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ml[i], eh[i], acc_c, 0, 0, 0);
            }
            const f32x16 E = acc + (acc_b + acc_c);
#ifdef WKV7C_TIMING
            asm volatile("" ::"v"(E[0]), "v"(E[15]));
#endif
            STSTAMP(2);
            float *ef = reinterpret_cast<float *>(sm + L::EF + (buf * 2 + mt) * L::NFsz) + lane * 4;
#pragma unroll
            for (int q = 0; q < 4; q++) *reinterpret_cast<float4 *>(ef + q * 256) = make_float4(E[4 * q], E[4 * q + 1], E[4 * q + 2], E[4 * q + 3]);
            uint16_t *Oh = sm + (buf ? L::E0h : L::E1h), *Ol = Oh + kC * LDK;
            store_T_split(E, Oh + mt * 32, Ol + mt * 32, LDK, lane);  // planes [v (this half)][k]
            STSTAMP(3);
            lds_barrier();
            STSTAMP(4);
        }
    } else {
        // ------------------------------------------------------------------------------------------ helper waves
        struct In {
            uint4 m[4];     // M^T: the lane's A fragments of k-steps 0..3 as int16 mantissas (two q15 tiles of 16) ...
            float ms[2];    // ... and the two tiles' scales
            uint4 n[2];     // N' tile: 16 int16 mantissas of this lane (q15 record, accumulator order) ...
            float ns;       // ... and their scale
        };
        // Unconditional (index clamped): a load inside a conditional makes the compiler's s_waitcnt bookkeeping take the minimum
        // over both paths at the join, i.e. wait for ALL outstanding loads (measured in the two-wave kernel: 1000-1270 of 2900
        // cycles per step).
        auto load = [&](int c) {
            In r;
            c = c < c0 ? c0 : c;
            const uint16_t *mrec = mt_ + ((long)bh * nc + c) * kQRec;
#pragma unroll
            for (int j = 0; j < 2; j++) {   // tile (k-tile mt, k'-tile j) = k-steps 2j, 2j + 1
                const int ms = (mt * 2 + j) * 64 + lane;
                r.m[2 * j] = *reinterpret_cast<const uint4 *>(mrec + ms * 16);
                r.m[2 * j + 1] = *reinterpret_cast<const uint4 *>(mrec + ms * 16 + 8);
                r.ms[j] = reinterpret_cast<const float *>(mrec + kQMant)[ms];
            }
            const uint16_t *nrec = np_ + ((long)bh * nc + c) * kQRec;
            const int slot = (mt * 2 + nt) * 64 + lane;
            r.n[0] = *reinterpret_cast<const uint4 *>(nrec + slot * 16);
            r.n[1] = *reinterpret_cast<const uint4 *>(nrec + slot * 16 + 8);
            r.ns = reinterpret_cast<const float *>(nrec + kQMant)[slot];
            return r;
        };
        // q15 -> what the product wave consumes, into buffer `buf`
        auto publish = [&](const In &in, int buf) {
            uint16_t *mf = sm + L::MF + (buf * 2 + mt) * L::MFsz + lane * 8;
#pragma unroll
            for (int i = 0; i < 4; i++) {   // int16 mantissas -> fp32 -> bf16 hi/lo pairs (the recurrence needs ~16 mantissa bits per operand)
                const uint32_t mw[4] = {in.m[i].x, in.m[i].y, in.m[i].z, in.m[i].w};
                const float sc = in.ms[i >> 1];
                uint32_t h4[4], l4[4];
#pragma unroll
                for (int j = 0; j < 4; j++)
                    split_pk((float)(int)(int16_t)(mw[j] & 0xffffu) * sc, (float)((int)mw[j] >> 16) * sc, h4[j], l4[j]);
                *reinterpret_cast<uint4 *>(mf + i * 512) = make_uint4(h4[0], h4[1], h4[2], h4[3]);
                *reinterpret_cast<uint4 *>(mf + 4 * 512 + i * 512) = make_uint4(l4[0], l4[1], l4[2], l4[3]);
            }
            float *nf = reinterpret_cast<float *>(sm + L::NF + (buf * 2 + mt) * L::NFsz) + lane * 4;
            const uint32_t nw[8] = {in.n[0].x, in.n[0].y, in.n[0].z, in.n[0].w, in.n[1].x, in.n[1].y, in.n[1].z, in.n[1].w};
#pragma unroll
            for (int q = 0; q < 4; q++) {
                float4 t;
                t.x = (float)(int)(int16_t)(nw[2 * q] & 0xffffu) * in.ns;
                t.y = (float)((int)nw[2 * q] >> 16) * in.ns;
                t.z = (float)(int)(int16_t)(nw[2 * q + 1] & 0xffffu) * in.ns;
                t.w = (float)((int)nw[2 * q + 1] >> 16) * in.ns;
                *reinterpret_cast<float4 *>(nf + q * 256) = t;
            }
        };
        // interval k (the product waves run step k = chunk c1-1-k): record of the E that ENTERS step k (what chunk c receives from
        // its future; made by step k-1, zero for k = 0) -> e_vk[c]; fragments of step k+1 -> the other buffer
        auto interval = [&](int k, const In &next) {
            STSTAMP(0);
            const int c = c1 - 1 - k;
            const float *ef = reinterpret_cast<const float *>(sm + L::EF + ((((k & 1) ^ 1) * 2 + mt) * L::NFsz)) + lane * 4;
            f32x16 E;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const float4 t = *reinterpret_cast<const float4 *>(ef + q * 256);
                E[4 * q] = t.x; E[4 * q + 1] = t.y; E[4 * q + 2] = t.z; E[4 * q + 3] = t.w;
            }
            q15_encode_tile(E, e_vk + ((long)bh * nc + c) * kQRec, nt, mt, lane);
            STSTAMP(1);
            publish(next, (k & 1) ^ 1);
            STSTAMP(2);
            lds_barrier();
            STSTAMP(4);
        };
        In r[kStatePF];
#pragma unroll
        for (int i = 0; i < kStatePF; i++) r[i] = load(c1 - 1 - i);   // r[i]: inputs of step i (mod kStatePF)
        publish(r[0], 0);
        __builtin_amdgcn_sched_barrier(0);
        r[0] = load(c1 - 1 - kStatePF);
        __builtin_amdgcn_sched_barrier(0);
        lds_barrier();
        // interval k publishes step k+1 from r[(k+1) % PF] and then refills that slot with step k+1+PF
        int k = 0;
        for (; k + kStatePF <= nsteps; k += kStatePF) {   // straight-line body: the waits count exactly the loads issued after the ones they need
#pragma unroll
            for (int i = 0; i < kStatePF; i++) {
                interval(k + i, r[(i + 1) % kStatePF]);
                // the sched_barriers keep each prefetch where it is written (the scheduler would sink them to the end of the body)
                __builtin_amdgcn_sched_barrier(0);
                r[(i + 1) % kStatePF] = load(c1 - 1 - (k + i + 1 + kStatePF));
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int i = 0; i < kStatePF - 1; i++)
            if (k + i < nsteps) interval(k + i, r[(i + 1) % kStatePF]);
    }
#ifdef WKV7C_TIMING
    if (blockIdx.x == 0 && lane == 0)
        for (int i = 0; i < 8; i++) g_cstate_timing[wave * 8 + i] += tacc_[i];
#endif
}

// ------------------------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------------------------
int chunk_bwd_pre_bf16(int B, int T_, int H, const void *w, const void *q, const void *a, const void *b, const void *dy,
                       const float *tinv, void *mt, void *np, hipStream_t st) {
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
                       (uint16_t *)mt, (uint16_t *)np);
    return (int)hipGetLastError();
}

int chunk_state_bf16(int BH, int nc, int H, const void *mt, const void *np, void *e_vk, const int *seq_off, int nseq,
                     hipStream_t st) {
    static bool attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&wkv7c_state_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)StateSmem::bytes);
        if (e != hipSuccess) return (int)e;
        attr = true;
    }
    (void)hipGetLastError();
    hipLaunchKernelGGL(wkv7c_state_kernel, dim3((seq_off ? nseq * H : BH) * 2), dim3(256), StateSmem::bytes, st, nc, H, (const uint16_t *)mt,
                       (const uint16_t *)np, (uint16_t *)e_vk, seq_off);
    return (int)hipGetLastError();
}

}  // namespace rwkv7

#ifdef WKV7C_TIMING
extern "C" int rwkv7_debug_cstate_timing(long long *out, int reset) {
    if (reset) {
        long long z[32] = {0};
        return (int)hipMemcpyToSymbol(HIP_SYMBOL(rwkv7::g_cstate_timing), z, sizeof(z));
    }
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(rwkv7::g_cstate_timing), sizeof(long long) * 32);
}
#endif
