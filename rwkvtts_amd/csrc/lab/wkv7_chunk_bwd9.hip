// rwkvtts_amd/csrc/wkv7_chunk_bwd9.hip -- per-chunk gradients of the chunked (MFMA) WKV7 backward from Z, bf16 tensors, 8 waves.
//
// Reference: wkv7_cuda.cu:54-130.  Successor of round 2's wkv7c_bwd_out8_kernel (removed in round 4: git log -- rwkvtts_amd/csrc/wkv7_chunk_bwd8.hip;
// the compiler traps found there -- hoisted LDS addresses, divergent wave ids, conditional loads -- are recorded in DESIGN.md section 4), except that
// Z (Z_t = dL/du_t) is an INPUT: the adjoint-state kernel wkv7c_bseq_kernel forms Z = (T^T B^) E' + (T^T A_qb^T) dY on its chain
// anyway and writes it (fp32 [B,T,H,64], like sa).  With Z given, the only dependency chain of the 8-wave kernel
// (A_qb -> G1 -> Z -> {P_vz, P_uz} -> last terms: five barrier-separated phases) is gone -- every tile product of the chunk depends
// on loaded operands and on the four P matrices only:
//   dV = A_qk^T dY + A_ak^T Z + K^ E'
//   dK = (P_vy Q~ + P_vz A~ + V E'^T) / gamma        dB = (P_uy Q~ + P_uz A~ + U E'^T) / gamma
//   dQ = (dY H0^T + P_vy^T K^ + P_uy^T B^) gamma      dA = (Z H0^T + P_vz^T K^ + P_uz^T B^) gamma_prev
//   P_vy = triu(V dY^T)  P_vz = triu(V Z^T, 1)  P_uy = triu(U dY^T)  P_uz = triu(U Z^T, 1)
//   dlw_t = sum_{s >= t} (q dQ - k dK - b dB)_s + sum_{s > t} (a dA)_s + rowsum(E * H_C) ;  dw = dlw * lw
// TWO matrix phases:
//     phase  wave 0               1              2        3        4        5        6        7
//       A    dQ0: dY H0^T         dQ1: dY H0^T   A_qk     A_ak     P_uz     P_uy     P_vz     P_vy
//       B    dK0; dQ0 +=          dK1; dQ1 +=    dA0      dA1      dV[0]    dV[1]    dB0      dB1       (all three terms each)
// (round 4: waves w and w + 4 share a SIMD; phase B's MFMA counts per SIMD were 56 / 56 / 46 / 46 with dB on 2-3, dA on 4-5, dV on
// 6-7 -- now 54 / 54 / 48 / 48, the heavier dA on the older waves, which win the VALU arbitration -- and P_vy moved from wave 0 to
// the wave that had nothing in phase A)
// then the ten accumulator tiles are staged (fp32, over the dead operand planes) and the epilogue runs as before.  T^-1, A_qb and
// G1 are not needed at all: 28 (rows) + 8 (u) + 8 (z) + 8.5 (E) + 8.5 (H_C) KB in, 24 KB out per chunk; LDS 133 KB.
#include "../chunk_bwd_common.h"
#include "../launch_attr.h"

// experiment ("measure operand by operand"): bit i set = the lo plane of operand i is written as zeros, i.e. that
// operand enters its products as ONE bf16 plane.  0 Q~  1 A~  2 K^  3 B^  4 U  5 Z  6 E'  7 H0  8 P_vy  9 P_vz  10 P_uy  11 P_uz  12 A_qk  13 A_ak
#ifndef WKV7C_B9_SINGLE
#define WKV7C_B9_SINGLE 0
#endif
#define B9S(bit) (((WKV7C_B9_SINGLE) >> (bit)) & 1)
#ifndef WKV7C_B9_YOUNG_PRIO
#define WKV7C_B9_YOUNG_PRIO 0   // measured: the two halves of the workgroup swap places, the chunk takes the same 14.3-14.6k cycles
#endif

namespace rwkv7 {

#ifdef WKV7C_TIMING
// profiling build only (python -m rwkvtts_amd.build --timing): per-phase cycle totals of workgroup 0, per wave
__device__ long long g_cbwd9_timing[8 * 16];
#define B9STAMP(i)                                                                     \
    do {                                                                               \
        const long long now_ = __builtin_readcyclecounter();                           \
        if (lane == 0) tacc_[wave * 16 + (i)] += now_ - tprev_;                         \
        tprev_ = now_;                                                                 \
    } while (0)
#define B9STAMP_INIT long long tprev_ = __builtin_readcyclecounter()
#define B9TIMING 1
#else
#define B9STAMP(i) do { } while (0)
#define B9STAMP_INIT do { } while (0)
#define B9TIMING 0
#endif

namespace {
// consecutive chunks per workgroup: a launch argument (round 4).  8 was fixed; with 16 384 chunks the same-box A/B gave 16: -2.2 %,
// 32: -3.1 %, 64 (one workgroup per CU): -3.8 % -- every workgroup pays one un-prefetched chunk at its start -- and 4: +4.0 %.
// The host picks the largest power of two in [8, 64] that still leaves >= 256 workgroups.
constexpr int kOut9MinChunksPerWG = 8, kOut9MaxChunksPerWG = 64;

struct Out9Smem {  // offsets in uint16 units
    static constexpr int kStLD = kN + 4;   // fp32 staging tiles [32][64 + 4]
    static constexpr int TM1 = kC * LDK, SQ1 = kN * LDK, A1 = kC * LDC, ST = kC * kStLD * 2;
    // operands of the whole chunk, TIME-major [t][.] (contractions over time fetch them with frag_tr)
    static constexpr int QTh = 0, QTl = QTh + TM1, ATh = QTl + TM1, ATl = ATh + TM1;
    static constexpr int KHh = ATl + TM1, KHl = KHh + TM1, BHh = KHl + TM1, BHl = BHh + TM1;
    static constexpr int Vp = BHl + TM1, DYp = Vp + TM1, Uh = DYp + TM1, Ul = Uh + TM1, Zh = Ul + TM1, Zl = Zh + TM1;
    // both 64x64 states, [v][k] (as they are stored in HBM)
    static constexpr int XTh = Zl + TM1, XTl = XTh + SQ1;     // E' = E g_C[k]
    static constexpr int HTh = XTl + SQ1, HTl = HTh + SQ1;    // H0
    // P planes: pair 0 = P_vy, 1 = P_vz, 2 = P_uy, 3 = P_uz; stored [t][s] (value kept for s >= t, s > t for the z ones)
    static constexpr int P0 = HTl + SQ1;
    static constexpr int AKTh = P0 + 8 * A1, AKTl = AKTh + A1, QKTh = AKTl + A1, QKTl = QKTh + A1;
    static constexpr int gC = QKTl + A1, dterm = gC + 2 * kN;   // 64 floats each
    static constexpr int sclE = dterm + 2 * kN, sclH = sclE + 2 * 256;   // q15 scales (256 floats per record): E; H [2 buffers]
    static constexpr int tacc = sclH + 2 * 2 * 256;              // timing build: 8 x 16 cycle counters (1 KB)
    static constexpr int end16 = tacc + (B9TIMING ? 8 * 16 * 4 : 0);
    static constexpr size_t bytes = (size_t)end16 * 2;
    // overlays
    static constexpr int RST = XTh;                      // raw rows in (restage: states / P / A matrices are written after its barrier)
    static constexpr int sQ = QTh, sK = QTh + ST, sB = QTh + 2 * ST, sA = QTh + 3 * ST, sV = QTh + 4 * ST;   // fp32 staging (operand planes dead after phase B)
    static constexpr int OUT = XTh;                      // six bf16 gradient planes [32][LDK] (states dead after phase B)
    static constexpr int DT8 = P0;                       // prologue 2: per-wave partial row sums of E * H_C, 8 x 64 floats (wave 0 reads them, then writes P_vy there)
};
static_assert(Out9Smem::bytes <= 160 * 1024, "LDS budget");
static_assert(Out9Smem::RST + 7 * kC * LDK + 2 * Out9Smem::ST <= Out9Smem::gC, "raw-row restaging must fit over the states / P / A matrices");
static_assert(Out9Smem::sV + Out9Smem::ST <= Out9Smem::XTh, "staging must fit over the operand planes");
static_assert(Out9Smem::OUT + 6 * kC * LDK <= Out9Smem::P0, "gradient rows must fit over the state planes");
static_assert(8 * kN * 2 <= 2 * Out9Smem::A1, "DT8 must fit in the P_vy planes");
static_assert(Out9Smem::XTh % 8 == 0 && Out9Smem::P0 % 8 == 0 && Out9Smem::AKTh % 8 == 0 && Out9Smem::gC % 8 == 0 && Out9Smem::ST % 8 == 0,
              "16-byte alignment");

// experiment helper: the lo plane of a transposed D tile as zeros (same addresses as store_T_split)
__device__ __forceinline__ void zero_T_lo(uint16_t *Ol, int ld, int lane) {
    const int n = lane & 31, h = lane >> 5;
#pragma unroll
    for (int j = 0; j < 4; j++) *reinterpret_cast<uint2 *>(Ol + n * ld + 8 * j + 4 * h) = make_uint2(0u, 0u);
}
// D tile -> fp32 staging [32][64 + 4], columns [32 ct, 32 ct + 32)
__device__ __forceinline__ void stage_tile9(const f32x16 &acc, float *stg, int ct, int lane) {
#pragma unroll
    for (int r = 0; r < 16; r++) stg[d_row(r, lane) * Out9Smem::kStLD + ct * 32 + (lane & 31)] = acc[r];
}
__device__ __forceinline__ float4 ld_stage4(const uint16_t *stg16, int pt, int pk) {
    return *reinterpret_cast<const float4 *>(reinterpret_cast<const float *>(stg16) + pt * Out9Smem::kStLD + pk);
}
// 4 consecutive fp32 of row `row` -> hi/lo planes [..][LDK], each scaled
__device__ __forceinline__ void put4s(uint16_t *Ph, uint16_t *Pl, int row, int c4, float4 x, float s) {
    uint32_t h0, l0, h1, l1;
    split_pk(x.x * s, x.y * s, h0, l0);
    split_pk(x.z * s, x.w * s, h1, l1);
    *reinterpret_cast<uint2 *>(Ph + row * LDK + c4) = make_uint2(h0, h1);
    *reinterpret_cast<uint2 *>(Pl + row * LDK + c4) = make_uint2(l0, l1);
}
// 4 fp32 -> hi/lo bf16, one 8-byte row segment of a time-major plane pair
__device__ __forceinline__ void put_row4(uint16_t *Ph, uint16_t *Pl, int off, float x0, float x1, float x2, float x3) {
    uint32_t h0, l0, h1, l1;
    split_pk(x0, x1, h0, l0);
    split_pk(x2, x3, h1, l1);
    *reinterpret_cast<uint2 *>(Ph + off) = make_uint2(h0, h1);
    *reinterpret_cast<uint2 *>(Pl + off) = make_uint2(l0, l1);
}
// K = 64 products in two halves: 8 operand fragments (32 registers) in flight instead of 16 -- with two waves per SIMD the
// second fragment wait hides behind the partner wave, while 64 fragment registers pushed the kernel over its 256-register
// budget (first cut: 113 registers spilled to scratch)
__device__ __forceinline__ void mma3_k64(f32x16 &acc, const uint16_t *Xh, const uint16_t *Xl, int ldx, const uint16_t *Yh, const uint16_t *Yl,
                                         int ldy, int lane) {   // X, Y row-major, both split
    mma_tile3<32>(acc, Xh, Xl, ldx, Yh, Yl, ldy, lane);
    mma_tile3<32>(acc, Xh + 32, Xl + 32, ldx, Yh + 32, Yl + 32, ldy, lane);
}
__device__ __forceinline__ void mma2y_k64(f32x16 &acc, const uint16_t *X, int ldx, const uint16_t *Yh, const uint16_t *Yl, int ldy,
                                          int lane) {  // X exact, Y split, row-major
    mma_tile2y<32>(acc, X, ldx, Yh, Yl, ldy, lane);
    mma_tile2y<32>(acc, X + 32, ldx, Yh + 32, Yl + 32, ldy, lane);
}
__device__ __forceinline__ void mma_xs_ye_k64(f32x16 &acc, const uint16_t *Xh, const uint16_t *Xl, int ldx, const uint16_t *Y, int ldy,
                                              int lane) {  // X split, Y exact, row-major
    mma_xs_ye<32>(acc, Xh, Xl, ldx, Y, ldy, lane);
    mma_xs_ye<32>(acc, Xh + 32, Xl + 32, ldx, Y + 32, ldy, lane);
}
// X row-major split (K = 64 contiguous), Y k-major split (rows = k): acc[m][n] += sum_k X[m][k] Y[k][ybase + n]
__device__ __forceinline__ void mma3_xr_yk_k64(f32x16 &acc, const uint16_t *Xh, const uint16_t *Xl, int ldx, const uint16_t *Yh,
                                               const uint16_t *Yl, int ldy, int ybase, int lane) {
    mma_gen<32, false, true, true, true>(acc, Xh, Xl, ldx, 0, Yh, Yl, ldy, ybase, lane);
    mma_gen<32, false, true, true, true>(acc, Xh + 32, Xl + 32, ldx, 0, Yh + 32 * ldy, Yl + 32 * ldy, ldy, ybase, lane);
}
// Y k-major (rows = contraction index) over K = 64 in two halves; X row-major
__device__ __forceinline__ void mma_xe_yks_k64(f32x16 &acc, const uint16_t *X, int ldx, const uint16_t *Yh, const uint16_t *Yl, int ldy,
                                               int ybase, int lane) {   // X exact, Y split
    mma_gen<32, false, false, true, true>(acc, X, X, ldx, 0, Yh, Yl, ldy, ybase, lane);
    mma_gen<32, false, false, true, true>(acc, X + 32, X + 32, ldx, 0, Yh + 32 * ldy, Yl + 32 * ldy, ldy, ybase, lane);
}
__device__ __forceinline__ void mma_xs_yke_k64(f32x16 &acc, const uint16_t *Xh, const uint16_t *Xl, int ldx, const uint16_t *Y, int ldy,
                                               int ybase, int lane) {   // X split, Y exact
    mma_gen<32, false, true, true, false>(acc, Xh, Xl, ldx, 0, Y, Y, ldy, ybase, lane);
    mma_gen<32, false, true, true, false>(acc, Xh + 32, Xl + 32, ldx, 0, Y + 32 * ldy, Y + 32 * ldy, ldy, ybase, lane);
}
__device__ __forceinline__ void mma_xe_yke_k64(f32x16 &acc, const uint16_t *X, int ldx, const uint16_t *Y, int ldy, int ybase,
                                               int lane) {   // both exact
    mma_gen<32, false, false, true, false>(acc, X, X, ldx, 0, Y, Y, ldy, ybase, lane);
    mma_gen<32, false, false, true, false>(acc, X + 32, X + 32, ldx, 0, Y + 32 * ldy, Y + 32 * ldy, ldy, ybase, lane);
}
// An opaque copy of the lane id.  Every LDS address of a tile product is a function of the lane id and of constants, i.e. loop
// invariant: hipcc hoists ~100 of them out of the chunk loop, runs out of registers and spills them to scratch once, reloading
// one before almost every product (first cut: 101 scratch stores at kernel entry, 101 reloads inside the loop, each an L1/L2
// round trip in front of an MFMA group).  Addresses derived from fresh(lane) cannot move above the call, so they are recomputed
// per phase -- two or three VALU instructions instead of a scratch load.
__device__ __forceinline__ int fresh(int x) {
    asm volatile("" : "+v"(x));
    return x;
}
__device__ __forceinline__ void cvt4u(const uint2 r, float (&f)[4]) {
    f[0] = __uint_as_float(r.x << 16); f[1] = __uint_as_float(r.x & 0xffff0000u);
    f[2] = __uint_as_float(r.y << 16); f[3] = __uint_as_float(r.y & 0xffff0000u);
}
}  // namespace

// hs_ = hs of wkv7_chunk_fwd*.hip (state at the START of every 32-step chunk: entries c and c+1), e_vk = E_{c+1} and z_ = Z of
// wkv7c_bseq_kernel; hs_ / e_vk q15 records [b,h,c] (chunk_common.h), sa_ / z_ fp32 [B,T,H,64].
__global__ __launch_bounds__(512) void wkv7c_bwd_out9_kernel(
    int T_, int H, int nchunks_total, int cpw, const bf16_t *__restrict__ w_, const bf16_t *__restrict__ q_, const bf16_t *__restrict__ k_,
    const bf16_t *__restrict__ v_, const bf16_t *__restrict__ a_, const bf16_t *__restrict__ b_, const bf16_t *__restrict__ dy_,
    const uint16_t *__restrict__ hs_, const float *__restrict__ sa_, const float *__restrict__ z_, const uint16_t *__restrict__ e_vk,
    bf16_t *__restrict__ dw_, bf16_t *__restrict__ dq_, bf16_t *__restrict__ dk_, bf16_t *__restrict__ dv_, bf16_t *__restrict__ da_,
    bf16_t *__restrict__ db_) {
    extern __shared__ __attribute__((aligned(16))) uint16_t sm[];
    using L = Out9Smem;
    float *sh_gC = reinterpret_cast<float *>(sm + L::gC), *sh_dterm = reinterpret_cast<float *>(sm + L::dterm);
    const int nc = T_ / kC;
    // wave and half are wave-uniform, and the compiler must know it (compiler traps)
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int pt = tid & 31, pk = (tid >> 5) * 4;                        // compute mapping: step pt, channels pk .. pk+3
    const int half = wave >> 2, ltid = tid & 255, lt = ltid >> 3, lk = (ltid & 7) * 8;   // global mapping: step lt, channels lk .. lk+7
    const long tstride = (long)H * kN;

    // global -> registers, one chunk ahead.  Rows: waves 0-3 fetch w, q, k, a and z; waves 4-7 fetch b, v, dy and u (= sa).
    struct Rows {
        uint4 x0, x1, x2, x3;   // waves 0-3: w, q, k, a          waves 4-7: b, v, dy, (dy again)
        float4 f0, f1;          //            z[0..3], z[4..7]               u[0..3], u[4..7]
        long off;
    };
    struct Mats {
        uint2 e[2], hc[2];   // q15 mantissas of (value row v = tid >> 3, keys 8 (tid & 7) .. +8): two 4-key pieces (chunk_common.h)
        float4 sc;           // threads 0-63: 4 of the 256 scales of E; threads 64-127: of H_C
    };
    // The same six 16-byte loads in both halves, through per-half pointers (wave-uniform selects), no branch, never skipped
    // (`valid` = false: every lane reads offset 0): the reasons are in DESIGN.md section 4 (compiler traps).
    auto load_rows = [&](int chunk, bool valid) {
        const int bh = chunk / nc, c = chunk - bh * nc;
        const int bb = bh / H, hh = bh - bb * H;
        Rows r;
        r.off = ((long)bb * T_ * H + hh) * kN + (long)(c * kC + lt) * tstride + lk;
        const long lo = valid ? r.off : 0;
        const bf16_t *p0 = half ? b_ : w_, *p1 = half ? v_ : q_, *p2 = half ? dy_ : k_, *p3 = half ? dy_ : a_;
        const float *pf = (half ? sa_ : z_) + lo;
        r.x0 = *reinterpret_cast<const uint4 *>(p0 + lo);
        r.x1 = *reinterpret_cast<const uint4 *>(p1 + lo);
        r.x2 = *reinterpret_cast<const uint4 *>(p2 + lo);
        r.x3 = *reinterpret_cast<const uint4 *>(p3 + lo);
        r.f0 = *reinterpret_cast<const float4 *>(pf);
        r.f1 = *reinterpret_cast<const float4 *>(pf + 4);
        return r;
    };
    const int st_v = tid >> 3, st_k8 = (tid & 7) * 8;   // this thread's piece of a 64x64 state: value row, keys st_k8 .. st_k8 + 7
    auto load_mats = [&](int chunk, bool valid) {
        Mats r;
        const int ch = valid ? chunk : 0;
        const int bh = ch / nc, c = ch - bh * nc;
        const uint16_t *er = e_vk + (long)ch * kQRec;
        const uint16_t *hr = hs_ + ((long)bh * nc + (c + 1 < nc ? c + 1 : c)) * kQRec;
        q15_load8(er, st_v, st_k8, r.e[0], r.e[1]);
        q15_load8(hr, st_v, st_k8, r.hc[0], r.hc[1]);
        r.sc = *reinterpret_cast<const float4 *>(reinterpret_cast<const float *>((tid < 64 ? er : hr) + kQMant) + (tid & 63) * 4);
        return r;
    };
#if B9TIMING
    long long *tacc_ = reinterpret_cast<long long *>(sm + L::tacc);
    if (tid < 128) tacc_[tid] = 0;
    lds_barrier();
#endif
#if WKV7C_B9_YOUNG_PRIO
    if (wave >= 4) __builtin_amdgcn_s_setprio(1);   // the second-dispatched half loses the per-SIMD VALU arbitration by age
#endif
    const int chunk0 = blockIdx.x * cpw;
    Rows cur = load_rows(chunk0, true);
    Mats curm = load_mats(chunk0, true);
    // H0 of a chunk = H_C of the chunk before it (a workgroup walks consecutive chunks, so the record is fetched once)
    float *sh_sE = reinterpret_cast<float *>(sm + L::sclE), *sh_sH = reinterpret_cast<float *>(sm + L::sclH);
    uint2 h0[2];
    q15_load8(hs_ + (long)chunk0 * kQRec, st_v, st_k8, h0[0], h0[1]);
    if (tid < 64)   // scales of the first H0 -> buffer 1 (chunk ci reads its H0 scales from buffer (ci & 1) ^ 1)
        *reinterpret_cast<float4 *>(sh_sH + 256 + tid * 4) =
            *reinterpret_cast<const float4 *>(reinterpret_cast<const float *>(hs_ + (long)chunk0 * kQRec + kQMant) + tid * 4);
    for (int ci = 0; ci < cpw; ci++) {
        const int chunk = chunk0 + ci;
        if (chunk >= nchunks_total) break;
        const bool more = ci + 1 < cpw && chunk + 1 < nchunks_total;
        const long off = cur.off;
        B9STAMP_INIT;
        // ---- raw rows: global mapping -> LDS -> compute mapping ---------------------------------------------------------------
        uint2 rw, rq, rk, ra, rb, rv, rdy;
        float4 ru, rz;
        {
            uint16_t *rs = sm + L::RST;
            float *rsu = reinterpret_cast<float *>(rs + 7 * kC * LDK), *rsz = rsu + kC * L::kStLD;
            const int wo = lt * LDK + lk, ro = pt * LDK + pk;
            if (half == 0) {
                *reinterpret_cast<uint4 *>(rs + 0 * kC * LDK + wo) = cur.x0;
                *reinterpret_cast<uint4 *>(rs + 1 * kC * LDK + wo) = cur.x1;
                *reinterpret_cast<uint4 *>(rs + 2 * kC * LDK + wo) = cur.x2;
                *reinterpret_cast<uint4 *>(rs + 3 * kC * LDK + wo) = cur.x3;
                *reinterpret_cast<float4 *>(rsz + lt * L::kStLD + lk) = cur.f0;
                *reinterpret_cast<float4 *>(rsz + lt * L::kStLD + lk + 4) = cur.f1;
            } else {
                *reinterpret_cast<uint4 *>(rs + 4 * kC * LDK + wo) = cur.x0;
                *reinterpret_cast<uint4 *>(rs + 5 * kC * LDK + wo) = cur.x1;
                *reinterpret_cast<uint4 *>(rs + 6 * kC * LDK + wo) = cur.x2;
                *reinterpret_cast<float4 *>(rsu + lt * L::kStLD + lk) = cur.f0;
                *reinterpret_cast<float4 *>(rsu + lt * L::kStLD + lk + 4) = cur.f1;
            }
            lds_barrier();
            rw = *reinterpret_cast<const uint2 *>(rs + 0 * kC * LDK + ro);
            rq = *reinterpret_cast<const uint2 *>(rs + 1 * kC * LDK + ro);
            rk = *reinterpret_cast<const uint2 *>(rs + 2 * kC * LDK + ro);
            ra = *reinterpret_cast<const uint2 *>(rs + 3 * kC * LDK + ro);
            rb = *reinterpret_cast<const uint2 *>(rs + 4 * kC * LDK + ro);
            rv = *reinterpret_cast<const uint2 *>(rs + 5 * kC * LDK + ro);
            rdy = *reinterpret_cast<const uint2 *>(rs + 6 * kC * LDK + ro);
            ru = *reinterpret_cast<const float4 *>(rsu + pt * L::kStLD + pk);
            rz = *reinterpret_cast<const float4 *>(rsz + pt * L::kStLD + pk);
        }
        B9STAMP(0);
        // ---- prologue 1: decay, scaled operands ------------------------------------------------------------------------------
        float lw[4], G[4], qv[4], kv[4], av[4], bv[4], gam[4], gprev[4], igam[4];
        cvt4u(rw, lw);
#pragma unroll
        for (int j = 0; j < 4; j++) lw[j] = -fast_exp(lw[j]);
#pragma unroll
        for (int j = 0; j < 4; j++) G[j] = scan32(lw[j]);
        cvt4u(rq, qv); cvt4u(rk, kv); cvt4u(ra, av); cvt4u(rb, bv);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            gam[j] = fast_exp(G[j]);
            gprev[j] = fast_exp(G[j] - lw[j]);
            igam[j] = fast_exp(-G[j]);
        }
        if (pt == kC - 1) {
#pragma unroll
            for (int j = 0; j < 4; j++) sh_gC[pk + j] = gam[j];
        }
        if (tid < 64) *reinterpret_cast<float4 *>(sh_sE + tid * 4) = curm.sc;
        else if (tid < 128) {
            const float z = (chunk % nc) + 1 < nc ? 1.f : 0.f;   // last chunk of a head: H_C = 0
            *reinterpret_cast<float4 *>(sh_sH + (ci & 1) * 256 + (tid - 64) * 4) = make_float4(curm.sc.x * z, curm.sc.y * z, curm.sc.z * z, curm.sc.w * z);
        }
        {
            const int o = pt * LDK + pk;
            put_row4(sm + L::QTh, sm + L::QTl, o, qv[0] * gam[0], qv[1] * gam[1], qv[2] * gam[2], qv[3] * gam[3]);
            if (B9S(0)) *reinterpret_cast<uint2 *>(sm + L::QTl + o) = make_uint2(0u, 0u);
            put_row4(sm + L::ATh, sm + L::ATl, o, av[0] * gprev[0], av[1] * gprev[1], av[2] * gprev[2], av[3] * gprev[3]);
            if (B9S(1)) *reinterpret_cast<uint2 *>(sm + L::ATl + o) = make_uint2(0u, 0u);
            put_row4(sm + L::KHh, sm + L::KHl, o, kv[0] * igam[0], kv[1] * igam[1], kv[2] * igam[2], kv[3] * igam[3]);
            if (B9S(2)) *reinterpret_cast<uint2 *>(sm + L::KHl + o) = make_uint2(0u, 0u);
            put_row4(sm + L::BHh, sm + L::BHl, o, bv[0] * igam[0], bv[1] * igam[1], bv[2] * igam[2], bv[3] * igam[3]);
            if (B9S(3)) *reinterpret_cast<uint2 *>(sm + L::BHl + o) = make_uint2(0u, 0u);
            put_row4(sm + L::Uh, sm + L::Ul, o, ru.x, ru.y, ru.z, ru.w);
            if (B9S(4)) *reinterpret_cast<uint2 *>(sm + L::Ul + o) = make_uint2(0u, 0u);
            put_row4(sm + L::Zh, sm + L::Zl, o, rz.x, rz.y, rz.z, rz.w);
            if (B9S(5)) *reinterpret_cast<uint2 *>(sm + L::Zl + o) = make_uint2(0u, 0u);
            *reinterpret_cast<uint2 *>(sm + L::Vp + o) = rv;     // bf16 inputs are exact: single planes
            *reinterpret_cast<uint2 *>(sm + L::DYp + o) = rdy;
        }
        B9STAMP(1);
        lds_barrier();  // sh_gC visible; the restaged rows are no longer read
        B9STAMP(2);
        // ---- prologue 2: the two states as planes, rowsum(E * H_C) ------------------------------------------------------------
        {
            const int row = st_v, k8 = st_k8;       // v = row, k = k8 .. k8 + 7
            const int slot = q15_slot(row, k8);  // scales: [slot] for keys k8 .. k8+3, [slot + 32] for k8+4 .. k8+7
            float gk[8], ev[8], hv[8], x[8], part[8];
            q15_decode8(curm.e[0], curm.e[1], sh_sE[slot], sh_sE[slot + 32], ev);
            const float *sHC = sh_sH + (ci & 1) * 256, *sH0 = sh_sH + ((ci & 1) ^ 1) * 256;
            q15_decode8(curm.hc[0], curm.hc[1], sHC[slot], sHC[slot + 32], hv);
            {
                const float4 a4 = *reinterpret_cast<const float4 *>(sh_gC + k8), b4 = *reinterpret_cast<const float4 *>(sh_gC + k8 + 4);
                gk[0] = a4.x; gk[1] = a4.y; gk[2] = a4.z; gk[3] = a4.w; gk[4] = b4.x; gk[5] = b4.y; gk[6] = b4.z; gk[7] = b4.w;
            }
#pragma unroll
            for (int j = 0; j < 8; j++) {
                x[j] = ev[j] * gk[j];          // E' = E g_C[k]
                part[j] = ev[j] * hv[j];       // E * H_C, summed over v below
            }
            uint32_t hi[4], lo[4];
            put_row8(sm + L::XTh, sm + L::XTl, row * LDK + k8, x, hi, lo);
            if (B9S(6)) *reinterpret_cast<uint4 *>(sm + L::XTl + row * LDK + k8) = make_uint4(0u, 0u, 0u, 0u);
            q15_decode8(h0[0], h0[1], sH0[slot], sH0[slot + 32], x);
            put_row8(sm + L::HTh, sm + L::HTl, row * LDK + k8, x, hi, lo);
            if (B9S(7)) *reinterpret_cast<uint4 *>(sm + L::HTl + row * LDK + k8) = make_uint4(0u, 0u, 0u, 0u);
            h0[0] = curm.hc[0];   // H_C of this chunk = H0 of the next (zeros across a head / sequence boundary on both sides);
            h0[1] = curm.hc[1];   // copied here, while no load is in flight
            // rowsum over v (= over the 8 lanes of this wave with the same tid & 7, then over the 8 waves): transposing pair sums
            float s4[4], s2[2];
#pragma unroll
            for (int j = 0; j < 4; j++) s4[j] = swap32_sum(part[2 * j], part[2 * j + 1]);   // lanes < 32: even k, lanes >= 32: odd k
#pragma unroll
            for (int j = 0; j < 2; j++) s2[j] = swap16_sum(s4[2 * j], s4[2 * j + 1]);       // even 16-rows: j pairs 0, odd: 1
            const float a0 = s2[0] + dpp_mov<0x128>(s2[0]), a1 = s2[1] + dpp_mov<0x128>(s2[1]);   // row_ror:8 brings lane l ^ 8
            const float red = (lane & 8) ? a1 : a0;
            const int kk = k8 + ((lane >> 5) & 1) + 2 * ((lane >> 4) & 1) + 4 * ((lane >> 3) & 1);
            reinterpret_cast<float *>(sm + L::DT8)[wave * kN + kk] = red;
        }
        B9STAMP(3);
        lds_barrier();
        if (tid < kN) {   // wave 0 only: it reads the DT8 area here and is the wave that overwrites it (P_vy) in phase A
            const float *d8 = reinterpret_cast<const float *>(sm + L::DT8);
            float t = 0.f;
#pragma unroll
            for (int wv = 0; wv < 8; wv++) t += d8[wv * kN + tid];
            sh_dterm[tid] = t;
        }
        B9STAMP(4);
        // ---- phase A --------------------------------------------------------------------------------------------------------------
        const int lnA = fresh(lane);
        // acc1: waves 0,1 dQ (A, B); waves 2,3 dA (B).  acc2: waves 0,1 dK; 6,7 dB; 4,5 dV (B).   D[m = t][n = k] / D[m = s][n = v]
        // (waves 0-3, dispatched first, win the VALU arbitration against their SIMD partners 4-7: they get the longer jobs)
        f32x16 acc1 = zero16(), acc2 = zero16();
        if (wave <= 1) {
            const int kt = wave;   // D[t][k] = sum_v dY[t][v] H0[v][k]
            mma_xe_yks_k64(acc1, sm + L::DYp, LDK, sm + L::HTh, sm + L::HTl, LDK, kt * 32, lnA);
        } else if (wave == 7) {   // (round 4: was wave 0's second job; wave 7 had none)
            f32x16 acc = zero16();  // D[m = s][n = t] = dy_s . v_t, s >= t -> P_vy[t][s]
            mma_tile<kN>(acc, sm + L::DYp, LDK, sm + L::Vp, LDK, lnA);
            mask_upper_T<false>(acc, lnA);
            store_T_split(acc, sm + L::P0 + 0 * 2 * L::A1, sm + L::P0 + 0 * 2 * L::A1 + L::A1, LDC, lnA);
            if (B9S(8)) zero_T_lo(sm + L::P0 + 0 * 2 * L::A1 + L::A1, LDC, lnA);
        } else if (wave == 2) {
            f32x16 acc = zero16();  // q~_t . k^_s, t >= s -> QKT[s][t]
            mma3_k64(acc, sm + L::QTh, sm + L::QTl, LDK, sm + L::KHh, sm + L::KHl, LDK, lnA);
            mask_upper_T<false>(acc, lnA);
            store_T_split(acc, sm + L::QKTh, sm + L::QKTl, LDC, lnA);
            if (B9S(12)) zero_T_lo(sm + L::QKTl, LDC, lnA);
        } else if (wave == 3) {
            f32x16 acc = zero16();  // a~_t . k^_s, t > s -> AKT[s][t]
            mma3_k64(acc, sm + L::ATh, sm + L::ATl, LDK, sm + L::KHh, sm + L::KHl, LDK, lnA);
            mask_upper_T<true>(acc, lnA);
            store_T_split(acc, sm + L::AKTh, sm + L::AKTl, LDC, lnA);
            if (B9S(13)) zero_T_lo(sm + L::AKTl, LDC, lnA);
        } else if (wave == 4) {
            f32x16 acc = zero16();  // z_s . u_t, s > t -> P_uz[t][s]
            mma3_k64(acc, sm + L::Zh, sm + L::Zl, LDK, sm + L::Uh, sm + L::Ul, LDK, lnA);
            mask_upper_T<true>(acc, lnA);
            store_T_split(acc, sm + L::P0 + 3 * 2 * L::A1, sm + L::P0 + 3 * 2 * L::A1 + L::A1, LDC, lnA);
            if (B9S(11)) zero_T_lo(sm + L::P0 + 3 * 2 * L::A1 + L::A1, LDC, lnA);
        } else if (wave == 5) {
            f32x16 acc = zero16();  // dy_s . u_t, s >= t -> P_uy[t][s]
            mma2y_k64(acc, sm + L::DYp, LDK, sm + L::Uh, sm + L::Ul, LDK, lnA);
            mask_upper_T<false>(acc, lnA);
            store_T_split(acc, sm + L::P0 + 2 * 2 * L::A1, sm + L::P0 + 2 * 2 * L::A1 + L::A1, LDC, lnA);
            if (B9S(10)) zero_T_lo(sm + L::P0 + 2 * 2 * L::A1 + L::A1, LDC, lnA);
        } else if (wave == 6) {
            f32x16 acz = zero16();  // z_s . v_t, s > t -> P_vz[t][s]
            mma_xs_ye_k64(acz, sm + L::Zh, sm + L::Zl, LDK, sm + L::Vp, LDK, lnA);
            mask_upper_T<true>(acz, lnA);
            store_T_split(acz, sm + L::P0 + 1 * 2 * L::A1, sm + L::P0 + 1 * 2 * L::A1 + L::A1, LDC, lnA);
            if (B9S(9)) zero_T_lo(sm + L::P0 + 1 * 2 * L::A1 + L::A1, LDC, lnA);
        }
        B9STAMP(5);
        lds_barrier();
        B9STAMP(6);
        // the next chunk's raw rows, two phases ahead of their use.  `cur` is dead since the restaging at the top of the iteration and
        // is overwritten in place
        cur = load_rows(chunk + 1, more);
        // ---- phase B --------------------------------------------------------------------------------------------------------------
        const int lnB = fresh(lane);
        if (wave <= 1) {
            const int kt = wave;   // dK: V E'^T + P_vy Q~ + P_vz A~ ;  dQ += P_vy^T K^ + P_uy^T B^   (X[t'][s] = P[s][t']: k-major fetch)
            mma_xe_yks_k64(acc2, sm + L::Vp, LDK, sm + L::XTh, sm + L::XTl, LDK, kt * 32, lnB);
            mma_tile3_yK<kC>(acc2, sm + L::P0, sm + L::P0 + L::A1, LDC, sm + L::QTh, sm + L::QTl, LDK, kt * 32, lnB);
            mma_tile3_yK<kC>(acc2, sm + L::P0 + 1 * 2 * L::A1, sm + L::P0 + 1 * 2 * L::A1 + L::A1, LDC, sm + L::ATh, sm + L::ATl, LDK, kt * 32, lnB);
            mma_gen<kC, true, true, true, true>(acc1, sm + L::P0, sm + L::P0 + L::A1, LDC, 0, sm + L::KHh, sm + L::KHl, LDK, kt * 32, lnB);
            mma_gen<kC, true, true, true, true>(acc1, sm + L::P0 + 2 * 2 * L::A1, sm + L::P0 + 2 * 2 * L::A1 + L::A1, LDC, 0, sm + L::BHh,
                                                sm + L::BHl, LDK, kt * 32, lnB);
        } else if (wave >= 6) {
            const int kt = wave - 6;   // dB: U E'^T + P_uy Q~ + P_uz A~
            mma3_xr_yk_k64(acc2, sm + L::Uh, sm + L::Ul, LDK, sm + L::XTh, sm + L::XTl, LDK, kt * 32, lnB);
            mma_tile3_yK<kC>(acc2, sm + L::P0 + 2 * 2 * L::A1, sm + L::P0 + 2 * 2 * L::A1 + L::A1, LDC, sm + L::QTh, sm + L::QTl, LDK,
                             kt * 32, lnB);
            mma_tile3_yK<kC>(acc2, sm + L::P0 + 3 * 2 * L::A1, sm + L::P0 + 3 * 2 * L::A1 + L::A1, LDC, sm + L::ATh, sm + L::ATl, LDK, kt * 32,
                             lnB);
        } else if (wave <= 3) {
            const int kt = wave - 2;    // dA: Z H0^T + P_vz^T K^ + P_uz^T B^
            mma3_xr_yk_k64(acc1, sm + L::Zh, sm + L::Zl, LDK, sm + L::HTh, sm + L::HTl, LDK, kt * 32, lnB);
            mma_gen<kC, true, true, true, true>(acc1, sm + L::P0 + 1 * 2 * L::A1, sm + L::P0 + 1 * 2 * L::A1 + L::A1, LDC, 0, sm + L::KHh,
                                                sm + L::KHl, LDK, kt * 32, lnB);
            mma_gen<kC, true, true, true, true>(acc1, sm + L::P0 + 3 * 2 * L::A1, sm + L::P0 + 3 * 2 * L::A1 + L::A1, LDC, 0, sm + L::BHh,
                                                sm + L::BHl, LDK, kt * 32, lnB);
        } else {
            const int vt = wave - 4;   // dV[s][v] = sum_t A_qk[t][s] dY[t][v] + sum_k k^[s][k] E'[k][v] + sum_t A_ak[t][s] Z[t][v]
            mma_xs_yeK<kC>(acc2, sm + L::QKTh, sm + L::QKTl, LDC, sm + L::DYp, LDK, vt * 32, lnB);
            mma3_k64(acc2, sm + L::KHh, sm + L::KHl, LDK, sm + L::XTh + vt * 32 * LDK, sm + L::XTl + vt * 32 * LDK, LDK, lnB);
            mma_tile3_yK<kC>(acc2, sm + L::AKTh, sm + L::AKTl, LDC, sm + L::Zh, sm + L::Zl, LDK, vt * 32, lnB);
        }
        B9STAMP(7);
        lds_barrier();   // every operand plane, P and state plane is dead from here on
        B9STAMP(8);
        // the next chunk's E, H_C (used ~1.5 phases into its prologue); curm is dead since prologue 2
        curm = load_mats(chunk + 1, more);
        // ---- the ten accumulator tiles -> fp32 staging over the operand planes ---------------------------------------------------------
        {
            const int lnS = fresh(lane);
            if (wave <= 1) {
                stage_tile9(acc2, reinterpret_cast<float *>(sm + L::sK), wave, lnS);
                stage_tile9(acc1, reinterpret_cast<float *>(sm + L::sQ), wave, lnS);
            } else if (wave <= 3) stage_tile9(acc1, reinterpret_cast<float *>(sm + L::sA), wave - 2, lnS);
            else if (wave <= 5) stage_tile9(acc2, reinterpret_cast<float *>(sm + L::sV), wave - 4, lnS);
            else stage_tile9(acc2, reinterpret_cast<float *>(sm + L::sB), wave - 6, lnS);
        }
        B9STAMP(9);
        lds_barrier();
        B9STAMP(10);
        // ---- epilogue: decay scaling, decay gradient, stores -------------------------------------------------------------------------
        {
            const float4 sQ4 = ld_stage4(sm + L::sQ, pt, pk), sK4 = ld_stage4(sm + L::sK, pt, pk), sB4 = ld_stage4(sm + L::sB, pt, pk),
                         sA4 = ld_stage4(sm + L::sA, pt, pk), sV4 = ld_stage4(sm + L::sV, pt, pk);
            float dQ[4] = {sQ4.x, sQ4.y, sQ4.z, sQ4.w}, dK[4] = {sK4.x, sK4.y, sK4.z, sK4.w}, dB[4] = {sB4.x, sB4.y, sB4.z, sB4.w},
                  dA[4] = {sA4.x, sA4.y, sA4.z, sA4.w}, e[4], dG[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                dQ[j] *= gam[j];
                dK[j] *= igam[j];
                dB[j] *= igam[j];
                dA[j] *= gprev[j];
                // e_t = (q dQ - k dK - b dB)_t + (a dA)_{t+1}
                e[j] = qv[j] * dQ[j] - kv[j] * dK[j] - bv[j] * dB[j] + next32(av[j] * dA[j], lane);
            }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                // dlw_t = sum_{s >= t} e_s + rowsum(E * H_C) = total - (inclusive prefix - e_t) + dterm ;  dw = dlw * lw
                const float pre = scan32(e[j]);
                dG[j] = (last32(pre, lane) - pre + e[j] + sh_dterm[pk + j]) * lw[j];
            }
            // the six gradients: compute mapping -> bf16 rows in LDS -> row-contiguous stores
            uint16_t *os = sm + L::OUT;
            auto put = [&](int i, float x0, float x1, float x2, float x3) {
                *reinterpret_cast<uint2 *>(os + i * kC * LDK + pt * LDK + pk) = make_uint2(cvt_pk(x0, x1), cvt_pk(x2, x3));
            };
            put(0, dG[0], dG[1], dG[2], dG[3]); put(1, dQ[0], dQ[1], dQ[2], dQ[3]); put(2, dK[0], dK[1], dK[2], dK[3]);
            put(3, sV4.x, sV4.y, sV4.z, sV4.w); put(4, dA[0], dA[1], dA[2], dA[3]); put(5, dB[0], dB[1], dB[2], dB[3]);
            lds_barrier();
            // per-half pointer selects, not an indexed pointer array: that loses the address space and the stores become flat_store
            bf16_t *const o0 = half ? dv_ : dw_, *const o1 = half ? da_ : dq_, *const o2 = half ? db_ : dk_;
            const uint16_t *src = os + half * 3 * kC * LDK + lt * LDK + lk;
            *reinterpret_cast<uint4 *>(o0 + off) = *reinterpret_cast<const uint4 *>(src);
            *reinterpret_cast<uint4 *>(o1 + off) = *reinterpret_cast<const uint4 *>(src + kC * LDK);
            *reinterpret_cast<uint4 *>(o2 + off) = *reinterpret_cast<const uint4 *>(src + 2 * kC * LDK);
        }
        B9STAMP(11);
        // the next chunk restages its rows over the state planes, where the gradient rows have just been read: every wave must be past
        // those reads first
        lds_barrier();
    }  // chunk loop
#if B9TIMING
    lds_barrier();
    if (blockIdx.x == 0 && tid < 128) g_cbwd9_timing[tid] += tacc_[tid];
#endif
}

int chunk_bwd_out9_bf16(int B, int T_, int H, const void *w, const void *q, const void *k, const void *v, const void *a, const void *b,
                        const void *dy, const void *hs, const float *sa, const float *z, const void *e_vk, void *dw, void *dq, void *dk,
                        void *dv, void *da, void *db, hipStream_t st) {
    static DynLdsOnce lds_once;
    if (hipError_t e = lds_once.ensure(reinterpret_cast<const void *>(&wkv7c_bwd_out9_kernel), (int)Out9Smem::bytes); e != hipSuccess) return (int)e;
    (void)hipGetLastError();
    const int total = B * H * (T_ / kC);
    int cpw = kOut9MinChunksPerWG;
    while (cpw < kOut9MaxChunksPerWG && total / (2 * cpw) >= 256) cpw *= 2;
    hipLaunchKernelGGL(wkv7c_bwd_out9_kernel, dim3((total + cpw - 1) / cpw), dim3(512), Out9Smem::bytes, st, T_, H,
                       total, cpw, (const bf16_t *)w, (const bf16_t *)q, (const bf16_t *)k, (const bf16_t *)v, (const bf16_t *)a,
                       (const bf16_t *)b, (const bf16_t *)dy, (const uint16_t *)hs, sa, z, (const uint16_t *)e_vk, (bf16_t *)dw,
                       (bf16_t *)dq, (bf16_t *)dk, (bf16_t *)dv, (bf16_t *)da, (bf16_t *)db);
    return (int)hipGetLastError();
}

#ifdef WKV7C_TIMING
extern "C" int rwkv7_debug_cbwd9_timing(long long *out, int reset) {
    if (reset) {
        long long z[128] = {0};
        return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_cbwd9_timing), z, sizeof(z));
    }
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_cbwd9_timing), sizeof(long long) * 128);
}
#endif

}  // namespace rwkv7
