// rwkvtts_amd/csrc/wkv7_chunk_bwd10.hip -- per-chunk gradients of the chunked (MFMA) WKV7 backward from Z, bf16 tensors, 8 waves.
//
// Reference: wkv7_cuda.cu:54-130.  Same arithmetic, same wave-to-tile table and same epilogue as wkv7c_bwd_out9_kernel
// (wkv7_chunk_bwd9.hip, whose header has the formulas); three structural changes (round 5):
//
//  (a) RAW ROWS BY LDS-DMA.  The nine row streams of a chunk (w q k a b v dy bf16, u = sa and z fp32: 44 KB) land in a dedicated
//      LDS area by global_load_lds_dwordx4 (44 pieces of 1 KB, 5-6 per wave; inline asm, see glds16), issued one chunk ahead, two
//      pieces at a time between the products of phase B (one burst queues at the texture addresser: 0.9-1.3k cycles per wave), and
//      waited for (vmcnt) in front of the gradient stores: no staging registers (24 VGPRs), no restage stores (44 KB through the
//      ~80 B/clk VGPR->LDS path), no restage barrier.  The landing image is lane-linear (base + lane * 16), so the swizzle that makes
//      the compute-mapping reads cheap sits on the SOURCE address (cdna_hip_programming.md rule 21).
//  (b) UNPADDED, XOR-SWIZZLED OPERAND PLANES.  Every plane is [rows][64] or [rows][32] bf16 without row padding; the 16-byte slot s
//      of row r lives at slot s ^ g(r), g64(r) = ((r >> 1) & 1) << 2 | (r >> 2) & 3, g32(r) = (r >> 2) & 3.  With the 72-element
//      rows of bwd_out9 the ds_read_b64_tr_b16 fragment reads (4 consecutive rows x 64 B per 32-lane group) hit every bank twice
//      (row stride 36 dwords: rows j and j + 2 overlap) -- the 31 % SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE of the round-4 PMC
//      pass; under the swizzle the tr reads AND the ds_read_b128 row fragments are conflict free, and the planes shrink by 11 %
//      (133 -> 115.5 KB), which is what pays for the landing area: 159.5 KB in all.
//  (c) STATE PROLOGUE AND PHASE A SHARE ONE BARRIER INTERVAL, IN OPPOSITE ORDER ON THE TWO WAVES OF A SIMD.  H0's planes move
//      into the first interval (they need no decay); then waves 0-3 decode E' (VALU + LDS stores) and run their phase-A product
//      afterwards, while their SIMD partners 4-7 run their product (LDS reads + MFMA) first and decode afterwards: the two
//      halves of a SIMD want different pipes at any time instead of the same one.  One barrier and one phase fewer per chunk.
//
//   interval  wave 0        1          2        3        4        5        6        7
//     I0      raw rows (landing -> registers), decay, scaled operands -> planes, H0 -> planes              (all waves alike)
//     I1      E' ; dQ0 = dY H0^T, dA0 = Z H0^T   (wave 1: dQ1, dA1)   E' ; A_qk  E' ; A_ak  P_uz ; E'  P_uy ; E'  P_vz ; E'  P_vy ; E'
//     I2      dQ0 +=, dA0 +=     dQ1 +=, dA1 +=     dK0      dK1      dV[0]    dV[1]    dB0      dB1       (phase B, the DMA pieces of the next chunk between its products)
//             MFMAs per SIMD (waves w, w + 4): I1 32 / 28 / 20 / 20, I2 46 / 46 / 44 / 44 (bwd_out9: phase A 20 / 20 / 20 / 16, B 54 / 54 / 48 / 48)
//     I3      accumulators -> fp32 staging ; I4 epilogue ; I5 gradient rows out
#include "chunk_bwd_common.h"
#include "launch_attr.h"

namespace rwkv7 {

#ifdef WKV7C_TIMING
__device__ long long g_cbwd10_timing[8 * 16];
#define B10STAMP(i)                                                                    \
    do {                                                                               \
        const long long now_ = __builtin_readcyclecounter();                           \
        if (lane == 0) tacc_[wave * 16 + (i)] += (unsigned)(now_ - tprev_);             \
        tprev_ = now_;                                                                 \
    } while (0)
#define B10STAMP_INIT long long tprev_ = __builtin_readcyclecounter()
#define B10TIMING 1
#else
#define B10STAMP(i) do { } while (0)
#define B10STAMP_INIT do { } while (0)
#define B10TIMING 0
#endif

#ifndef WKV7C_B10_K64_ONE
#define WKV7C_B10_K64_ONE 0   // experiment: the K = 64 products in one round of 16 + 16 fragments
#endif
namespace {
constexpr int kOut10MinChunksPerWG = 8, kOut10MaxChunksPerWG = 64;

// ---- swizzled planes ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int g64(int r) { return (((r >> 1) & 1) << 2) | ((r >> 2) & 3); }
__device__ __forceinline__ int sw64(int r, int c) { return r * 64 + ((((c >> 3) ^ g64(r)) << 3) | (c & 7)); }   // element offset in a [.][64] plane
__device__ __forceinline__ int g32(int r) { return (r >> 2) & 3; }
__device__ __forceinline__ int sw32(int r, int c) { return r * 32 + ((((c >> 3) ^ g32(r)) << 3) | (c & 7)); }   // [.][32] plane
// raw-row landing images (filled by LDS-DMA, read once in the compute mapping): bf16 [32][64] with slot ^ ((r >> 1) & 7),
// fp32 [32][64] (16 slots of 16 B per row) with slot ^ (r & 15)
__device__ __forceinline__ int raw_slot16(int r, int s) { return s ^ ((r >> 1) & 7); }
__device__ __forceinline__ int rawf_slot16(int r, int s) { return s ^ (r & 15); }

// Lane-constant parts of every fragment address (recomputed per phase from fresh(lane): cheap, and not hoistable out of the chunk loop):
//   row-major fragment i (k = 16 i + 8 (lane >> 5) .. + 8) of tile rows rb .. rb + 31:   P + rb * W + a64[i] / b32[i]
//   k-major fragment i (k rows 16 i + 8 (lane >> 5) + 0..7, columns nb .. nb + 31):       P + 16 i W + (t64[0|1] ^ nb) / t32[0|1]
// (g64 / g32 of the k-major rows do not depend on i, and the column base 32 is the XOR of bit 5: see the header)
struct SwLane {
    int a64[4], b32[2], t64[2], t32[2];
};
__device__ __forceinline__ SwLane sw_lane(int lane) {
    SwLane s;
    const int r = lane & 31, h = lane >> 5;
#pragma unroll
    for (int i = 0; i < 4; i++) s.a64[i] = sw64(r, 16 * i + 8 * h);
#pragma unroll
    for (int i = 0; i < 2; i++) s.b32[i] = sw32(r, 16 * i + 8 * h);
    const int i16 = lane & 15, col = 16 * ((lane >> 4) & 1) + 4 * (i16 & 3);
#pragma unroll
    for (int r2 = 0; r2 < 2; r2++) {
        const int row = 8 * h + (i16 >> 2) + 4 * r2;
        s.t64[r2] = sw64(row, col);
        s.t32[r2] = sw32(row, col);
    }
    return s;
}
template <int W>
__device__ __forceinline__ bf16x8 frag_rm(const uint16_t *P, int rb, const SwLane &s, int i) {
    return *reinterpret_cast<const bf16x8 *>(P + rb * W + (W == 64 ? s.a64[i] : s.b32[i]));
}
template <int W>
__device__ __forceinline__ bf16x8 frag_km(const uint16_t *P, int nb, const SwLane &s, int i) {
    using lds_ptr = bf16x4_t __attribute__((address_space(3))) *;
    const uint16_t *p0 = P + 16 * i * W + (W == 64 ? (s.t64[0] ^ nb) : s.t32[0]);
    const uint16_t *p1 = P + 16 * i * W + (W == 64 ? (s.t64[1] ^ nb) : s.t32[1]);
    const bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_ptr)(p0));
    const bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_ptr)(p1));
    const bf16x8e_t v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}
// acc[m][n] += sum over NK k-steps of 16, starting at k-step i0, of X[m][k] Y[n][k].  Each operand: plane width XW / YW (64 or 32),
// k-major or row-major, split (hi + lo) or exact (one plane; pass it twice).  xb / yb: first tile row (row-major) or first column
// (k-major).  Terms: Xh Yh (+ Xh Yl) (+ Xl Yh); fragments fetched first, MFMAs behind a scheduling barrier (chunk_common.h).
template <int NK, int XW, bool XKM, bool XSPLIT, int YW, bool YKM, bool YSPLIT>
__device__ __forceinline__ void mma_sw(f32x16 &acc, const uint16_t *Xh, const uint16_t *Xl, int xb, const uint16_t *Yh, const uint16_t *Yl,
                                       int yb, const SwLane &s, int i0 = 0) {
    bf16x8 xh[NK], xl[NK], yh[NK], yl[NK];
#pragma unroll
    for (int i = 0; i < NK; i++) {
        xh[i] = XKM ? frag_km<XW>(Xh, xb, s, i0 + i) : frag_rm<XW>(Xh, xb, s, i0 + i);
        yh[i] = YKM ? frag_km<YW>(Yh, yb, s, i0 + i) : frag_rm<YW>(Yh, yb, s, i0 + i);
        if (XSPLIT) xl[i] = XKM ? frag_km<XW>(Xl, xb, s, i0 + i) : frag_rm<XW>(Xl, xb, s, i0 + i);
        if (YSPLIT) yl[i] = YKM ? frag_km<YW>(Yl, yb, s, i0 + i) : frag_rm<YW>(Yl, yb, s, i0 + i);
    }
    __builtin_amdgcn_sched_barrier(0);  // fragment loads stay above, MFMAs below
#pragma unroll
    for (int i = 0; i < NK; i++) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh[i], yh[i], acc, 0, 0, 0);
        if (YSPLIT) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh[i], yl[i], acc, 0, 0, 0);
        if (XSPLIT) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xl[i], yh[i], acc, 0, 0, 0);
    }
}
// K = 64 in two halves of two k-steps (8 + 8 fragments in flight instead of 16: the register budget of bwd_out9)
template <int XW, bool XKM, bool XSPLIT, int YW, bool YKM, bool YSPLIT>
__device__ __forceinline__ void mma_sw_k64(f32x16 &acc, const uint16_t *Xh, const uint16_t *Xl, int xb, const uint16_t *Yh,
                                           const uint16_t *Yl, int yb, const SwLane &s) {
#if WKV7C_B10_K64_ONE
    mma_sw<4, XW, XKM, XSPLIT, YW, YKM, YSPLIT>(acc, Xh, Xl, xb, Yh, Yl, yb, s, 0);
#else
    mma_sw<2, XW, XKM, XSPLIT, YW, YKM, YSPLIT>(acc, Xh, Xl, xb, Yh, Yl, yb, s, 0);
    mma_sw<2, XW, XKM, XSPLIT, YW, YKM, YSPLIT>(acc, Xh, Xl, xb, Yh, Yl, yb, s, 2);
#endif
}

struct Out10Smem {  // offsets in uint16 units
    static constexpr int kStLD = kN + 4;   // fp32 staging tiles [32][64 + 4]
    static constexpr int TM1 = kC * kN, SQ1 = kN * kN, A1 = kC * kC, ST = kC * kStLD * 2;
    static constexpr int OLD = kN + kPad;   // gradient rows out [32][72] (bf16, row-contiguous reads)
    // operands of the whole chunk, TIME-major [t][.] swizzled (sw64)
    static constexpr int QTh = 0, QTl = QTh + TM1, ATh = QTl + TM1, ATl = ATh + TM1;
    static constexpr int KHh = ATl + TM1, KHl = KHh + TM1, BHh = KHl + TM1, BHl = BHh + TM1;
    static constexpr int Vp = BHl + TM1, DYp = Vp + TM1, Uh = DYp + TM1, Ul = Uh + TM1, Zh = Ul + TM1, Zl = Zh + TM1;
    // both 64x64 states, [v][k] swizzled (sw64)
    static constexpr int XTh = Zl + TM1, XTl = XTh + SQ1;     // E' = E g_C[k]
    static constexpr int HTh = XTl + SQ1, HTl = HTh + SQ1;    // H0
    // P planes [t][s] (sw32): pair 0 = P_vy, 1 = P_vz, 2 = P_uy, 3 = P_uz; then A_ak^T, A_qk^T [s][t]
    static constexpr int P0 = HTl + SQ1;
    static constexpr int AKTh = P0 + 8 * A1, AKTl = AKTh + A1, QKTh = AKTl + A1, QKTl = QKTh + A1;
    static constexpr int gC = QKTl + A1, dterm = gC + 2 * kN;   // 64 floats each
    static constexpr int sclE = dterm + 2 * kN, sclH = sclE + 2 * 256;   // q15 scales (256 floats per record): E; H [2 buffers]
    // raw-row landing area (LDS-DMA): 7 bf16 images of 4 KB (w q k a b v dy), then u and z fp32 images of 8 KB
    static constexpr int LAND = sclH + 2 * 2 * 256;
    static constexpr int LANDF = LAND + 7 * TM1;
    static constexpr int tacc = LANDF + 2 * 2 * TM1;             // timing build: 8 x 16 32-bit cycle counters (512 B)
    static constexpr int end16 = tacc + (B10TIMING ? 8 * 16 * 2 : 0);
    static constexpr size_t bytes = (size_t)end16 * 2;
    // overlays
    static constexpr int sQ = QTh, sK = QTh + ST, sB = QTh + 2 * ST, sA = QTh + 3 * ST, sV = QTh + 4 * ST;   // fp32 staging (operand planes dead after phase B)
    static constexpr int OUT = XTh;                      // six bf16 gradient planes [32][OLD] (states dead after phase B)
    static constexpr int DT8 = LAND;                     // per-wave partial row sums of E * H_C, 8 x 64 floats: the first 2 KB of the landing area, which
                                                         // wave 0 alone refills (pieces 0, 1) after it has summed them
};
static_assert(Out10Smem::bytes <= 160 * 1024, "LDS budget");
static_assert(Out10Smem::sV + Out10Smem::ST <= Out10Smem::XTh, "staging must fit over the operand planes");
static_assert(Out10Smem::OUT + 6 * kC * Out10Smem::OLD <= Out10Smem::P0, "gradient rows must fit over the state planes");
static_assert(Out10Smem::XTh % 8 == 0 && Out10Smem::P0 % 8 == 0 && Out10Smem::gC % 8 == 0 && Out10Smem::ST % 8 == 0 && Out10Smem::LAND % 8 == 0,
              "16-byte alignment");

__device__ __forceinline__ void stage_tile10(const f32x16 &acc, float *stg, int ct, int lane) {
#pragma unroll
    for (int r = 0; r < 16; r++) stg[d_row(r, lane) * Out10Smem::kStLD + ct * 32 + (lane & 31)] = acc[r];
}
__device__ __forceinline__ float4 ld_stage4_10(const uint16_t *stg16, int pt, int pk) {
    return *reinterpret_cast<const float4 *>(reinterpret_cast<const float *>(stg16) + pt * Out10Smem::kStLD + pk);
}
// 4 fp32 -> hi/lo bf16, one 8-byte row segment of a plane pair
__device__ __forceinline__ void put_row4_10(uint16_t *Ph, uint16_t *Pl, int off, float x0, float x1, float x2, float x3) {
    uint32_t h0, l0, h1, l1;
    split_pk(x0, x1, h0, l0);
    split_pk(x2, x3, h1, l1);
    *reinterpret_cast<uint2 *>(Ph + off) = make_uint2(h0, h1);
    *reinterpret_cast<uint2 *>(Pl + off) = make_uint2(l0, l1);
}
// D tile transposed into a hi/lo pair of [32][32] swizzled planes: OUT[n][m] = D[m][n]
__device__ __forceinline__ void store_T_split_sw32(const f32x16 &acc, uint16_t *Oh, uint16_t *Ol, int lane) {
    const int n = lane & 31, h = lane >> 5, g = g32(n), base = n * 32 + 4 * h;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        uint32_t h0, l0, h1, l1;
        split_pk(acc[4 * j + 0], acc[4 * j + 1], h0, l0);
        split_pk(acc[4 * j + 2], acc[4 * j + 3], h1, l1);
        const int off = base + ((j ^ g) << 3);
        *reinterpret_cast<uint2 *>(Oh + off) = make_uint2(h0, h1);
        *reinterpret_cast<uint2 *>(Ol + off) = make_uint2(l0, l1);
    }
}
// One LDS-DMA instruction, 16 B per lane: LDS destination = M0 (wave-uniform byte address) + lane * 16.  Inline asm, not
// __builtin_amdgcn_global_load_lds: hipcc's wait-count pass treats the builtin as a store to LDS that may alias every later ds_read of
// the one dynamic __shared__ array and put `s_waitcnt vmcnt(0)` in front of phase B's first fragment read -- every wave sat out the full
// HBM latency of the pieces it had just issued (first cut of this kernel: phase B 3.0k -> 4.0-5.5k cycles, stamps in profiles/).  The
// asm statement is opaque to that pass; the data is ordered by the explicit vmcnt(0) + barrier in front of the gradient stores.  M0 is
// compiler-reserved and saved / restored around the statement (cdna_hip_programming.md, inline-asm rules).
__device__ __forceinline__ void glds16(const void *gsrc, uint32_t lds_dst) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}
__device__ __forceinline__ int fresh10(int x) {   // opaque copy of the lane id (wkv7_chunk_bwd9.hip: compiler traps)
    asm volatile("" : "+v"(x));
    return x;
}
__device__ __forceinline__ void cvt4u10(const uint2 r, float (&f)[4]) {
    f[0] = __uint_as_float(r.x << 16); f[1] = __uint_as_float(r.x & 0xffff0000u);
    f[2] = __uint_as_float(r.y << 16); f[3] = __uint_as_float(r.y & 0xffff0000u);
}
}  // namespace

// the nine row streams of the LDS-DMA, by piece class: a kernel argument (kernarg memory), so that the wave-uniform index p >> 2 is one
// scalar load instead of a chain of pointer selects
struct Out10Rows {
    const void *p[9];   // w q k a b v dy (bf16) ; u = sa, z (fp32)
};

__global__ __launch_bounds__(512) void wkv7c_bwd_out10_kernel(
    Out10Rows rows_, int T_, int H, int nchunks_total, int cpw, const uint16_t *__restrict__ hs_, const uint16_t *__restrict__ e_vk,
    bf16_t *__restrict__ dw_, bf16_t *__restrict__ dq_, bf16_t *__restrict__ dk_, bf16_t *__restrict__ dv_, bf16_t *__restrict__ da_,
    bf16_t *__restrict__ db_) {
    extern __shared__ __attribute__((aligned(16))) uint16_t sm[];
    using L = Out10Smem;
    using gptr = const __attribute__((address_space(1))) void *;
    using lptr = __attribute__((address_space(3))) void *;
    float *sh_gC = reinterpret_cast<float *>(sm + L::gC), *sh_dterm = reinterpret_cast<float *>(sm + L::dterm);
    const int nc = T_ / kC;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int pt = tid & 31, pk = (tid >> 5) * 4;                        // compute mapping: step pt, channels pk .. pk+3
    const int half = wave >> 2, ltid = tid & 255, lt = ltid >> 3, lk = (ltid & 7) * 8;   // store mapping: step lt, channels lk .. lk+7
    const long tstride = (long)H * kN;

    struct Mats {
        uint2 e[2], hc[2];   // q15 mantissas of (value row v = tid >> 3, keys 8 (tid & 7) .. +8): two 4-key pieces (chunk_common.h)
        float4 sc;           // threads 0-63: 4 of the 256 scales of E; threads 64-127: of H_C
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
    auto chunk_base = [&](int chunk) -> long {   // element offset of (step 0, channel 0) of the chunk's head in a [B,T,H,64] tensor
        const int bh = chunk / nc, c = chunk - bh * nc;
        const int bb = bh / H, hh = bh - bb * H;
        return ((long)bb * T_ * H + hh) * kN + (long)(c * kC) * tstride;
    };
    // LDS-DMA of one chunk's raw rows: 44 pieces of 1 KB (64 lanes x 16 B).  Pieces 0-27: bf16 tensor p >> 2 (w q k a b v dy), rows
    // 8 (p & 3) .. + 8; pieces 28-43: fp32 tensor (p - 28) >> 3 (u, z), rows 4 ((p - 28) & 7) .. + 4.  Waves 0-3 own pieces 6 w .. 6 w + 5,
    // waves 4-7 pieces 24 + 5 (w - 4) .. + 4.  The swizzle is on the SOURCE address (the destination is base + lane * 16).
    const uint32_t lds0 = (uint32_t)(size_t)(lptr)(sm);   // LDS byte address of the dynamic array
    auto dma_pieces = [&](long cb, int j0, int j1) {   // pieces j0 .. j1 - 1 of this wave's 5 or 6; cb = chunk_base of the chunk (or 0)
        const int p0 = wave < 4 ? 6 * wave : 24 + 5 * (wave - 4);
        const int np = wave < 4 ? 6 : 5;
        // lane parts of the source address: bf16 piece: row 8 part + (lane >> 3), physical slot lane & 7 holds logical slot
        // (lane & 7) ^ ((row >> 1) & 7); fp32 piece: row 4 part + (lane >> 4), logical slot (lane & 15) ^ (row & 15)
        const int lr8 = lane >> 3, lr4 = lane >> 4;
#pragma unroll
        for (int j = j0; j < j1; j++) {
            if (j < np) {
                const int p = p0 + j;   // wave-uniform
                const bool isf = p >= 28;
                const int t = isf ? 7 + ((p - 28) >> 3) : (p >> 2), part = isf ? ((p - 28) & 7) : (p & 3);
                const char *src = reinterpret_cast<const char *>(rows_.p[t]);
                const int row = isf ? 4 * part + lr4 : 8 * part + lr8;
                const int slot = isf ? rawf_slot16(row, lane & 15) : raw_slot16(row, lane & 7);
                const long eoff = cb + (long)row * tstride;                       // elements
                const char *g = src + (isf ? eoff * 4 : eoff * 2) + slot * 16;
                const uint32_t dst = lds0 + 2 * (isf ? L::LANDF + (t - 7) * 2 * L::TM1 : L::LAND + t * L::TM1) + part * 1024;
                glds16(g, __builtin_amdgcn_readfirstlane(dst));
            }
        }
    };
#if B10TIMING
    unsigned *tacc_ = reinterpret_cast<unsigned *>(sm + L::tacc);
    if (tid < 128) tacc_[tid] = 0;
    lds_barrier();
#endif
    const int chunk0 = blockIdx.x * cpw;
    dma_pieces(chunk_base(chunk0), 0, 6);
    Mats curm = load_mats(chunk0, true);
    float *sh_sE = reinterpret_cast<float *>(sm + L::sclE), *sh_sH = reinterpret_cast<float *>(sm + L::sclH);
    uint2 h0[2];
    q15_load8(hs_ + (long)chunk0 * kQRec, st_v, st_k8, h0[0], h0[1]);
    if (tid < 64)   // scales of the first H0 -> buffer 1 (chunk ci reads its H0 scales from buffer (ci & 1) ^ 1)
        *reinterpret_cast<float4 *>(sh_sH + 256 + tid * 4) =
            *reinterpret_cast<const float4 *>(reinterpret_cast<const float *>(hs_ + (long)chunk0 * kQRec + kQMant) + tid * 4);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the first chunk's rows have landed (this wave's pieces) ...
    lds_barrier();                                      // ... everybody's
    // lane-constant LDS offsets of the compute mapping
    const int o_cm = sw64(pt, pk);                                                       // operand planes (element offset)
    const int o_raw = pt * 64 + raw_slot16(pt, pk >> 3) * 8 + (pk & 7);                  // bf16 landing images (element offset)
    const int o_rawf = pt * 64 + rawf_slot16(pt, pk >> 2) * 4;                           // fp32 landing images (float offset)
    const int o_st = sw64(st_v, st_k8);                                                  // state planes (element offset)
    for (int ci = 0; ci < cpw; ci++) {
        const int chunk = chunk0 + ci;
        if (chunk >= nchunks_total) break;
        const bool more = ci + 1 < cpw && chunk + 1 < nchunks_total;
        B10STAMP_INIT;
        // ---- I0: raw rows from the landing area (compute mapping), decay, scaled operands, H0 planes --------------------------------
        uint2 rw, rq, rk, ra, rb, rv, rdy;
        float4 ru, rz;
        {
            const uint16_t *ld = sm + L::LAND + o_raw;
            rw = *reinterpret_cast<const uint2 *>(ld + 0 * L::TM1);
            rq = *reinterpret_cast<const uint2 *>(ld + 1 * L::TM1);
            rk = *reinterpret_cast<const uint2 *>(ld + 2 * L::TM1);
            ra = *reinterpret_cast<const uint2 *>(ld + 3 * L::TM1);
            rb = *reinterpret_cast<const uint2 *>(ld + 4 * L::TM1);
            rv = *reinterpret_cast<const uint2 *>(ld + 5 * L::TM1);
            rdy = *reinterpret_cast<const uint2 *>(ld + 6 * L::TM1);
            const float *lf = reinterpret_cast<const float *>(sm + L::LANDF) + o_rawf;
            ru = *reinterpret_cast<const float4 *>(lf);
            rz = *reinterpret_cast<const float4 *>(lf + kC * kN);
        }
        B10STAMP(0);
        float lw[4], G[4], qv[4], kv[4], av[4], bv[4], gam[4], gprev[4], igam[4];
        cvt4u10(rw, lw);
#pragma unroll
        for (int j = 0; j < 4; j++) lw[j] = -fast_exp(lw[j]);
#pragma unroll
        for (int j = 0; j < 4; j++) G[j] = scan32(lw[j]);
        cvt4u10(rq, qv); cvt4u10(rk, kv); cvt4u10(ra, av); cvt4u10(rb, bv);
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
            const float zf = (chunk % nc) + 1 < nc ? 1.f : 0.f;   // last chunk of a head: H_C = 0
            *reinterpret_cast<float4 *>(sh_sH + (ci & 1) * 256 + (tid - 64) * 4) = make_float4(curm.sc.x * zf, curm.sc.y * zf, curm.sc.z * zf, curm.sc.w * zf);
        }
        {
            const int o = o_cm;
            put_row4_10(sm + L::QTh, sm + L::QTl, o, qv[0] * gam[0], qv[1] * gam[1], qv[2] * gam[2], qv[3] * gam[3]);
            put_row4_10(sm + L::ATh, sm + L::ATl, o, av[0] * gprev[0], av[1] * gprev[1], av[2] * gprev[2], av[3] * gprev[3]);
            put_row4_10(sm + L::KHh, sm + L::KHl, o, kv[0] * igam[0], kv[1] * igam[1], kv[2] * igam[2], kv[3] * igam[3]);
            put_row4_10(sm + L::BHh, sm + L::BHl, o, bv[0] * igam[0], bv[1] * igam[1], bv[2] * igam[2], bv[3] * igam[3]);
            put_row4_10(sm + L::Uh, sm + L::Ul, o, ru.x, ru.y, ru.z, ru.w);
            put_row4_10(sm + L::Zh, sm + L::Zl, o, rz.x, rz.y, rz.z, rz.w);
            *reinterpret_cast<uint2 *>(sm + L::Vp + o) = rv;     // bf16 inputs are exact: single planes
            *reinterpret_cast<uint2 *>(sm + L::DYp + o) = rdy;
        }
        const int slot = q15_slot(st_v, st_k8);  // q15 scales: [slot] for keys st_k8 .. +3, [slot + 32] for +4 .. +7
        {   // H0 -> planes (its scales were written one chunk ago; no decay involved)
            const float *sH0 = sh_sH + ((ci & 1) ^ 1) * 256;
            float x[8];
            uint32_t hi[4], lo[4];
            q15_decode8(h0[0], h0[1], sH0[slot], sH0[slot + 32], x);
            put_row8(sm + L::HTh, sm + L::HTl, o_st, x, hi, lo);
        }
        B10STAMP(1);
        lds_barrier();  // operand planes, H0 planes, sh_gC, scales visible; the landing area is no longer read
        B10STAMP(2);
        // ---- I1: E' planes + rowsum(E * H_C)  |  the eight single products -- in opposite order on the two waves of a SIMD -------------
        // acc1: waves 0-1 dQ.  acc3: waves 0-1 dA.  acc2: waves 2-3 dK, 4-5 dV, 6-7 dB.
        f32x16 acc1 = zero16(), acc2 = zero16(), acc3 = zero16();
        // the ten lane-constant fragment offsets of the swizzled planes, once per chunk from an opaque copy of the lane id (hoisted out of
        // the chunk loop they drag the derived addresses along: 256 registers + 56 spilled)
        const SwLane s = sw_lane(fresh10(lane));
        auto state_prologue = [&]() {
            float gk[8], ev[8], hv[8], x[8], part[8];
            q15_decode8(curm.e[0], curm.e[1], sh_sE[slot], sh_sE[slot + 32], ev);
            const float *sHC = sh_sH + (ci & 1) * 256;
            q15_decode8(curm.hc[0], curm.hc[1], sHC[slot], sHC[slot + 32], hv);
            {
                const float4 a4 = *reinterpret_cast<const float4 *>(sh_gC + st_k8), b4 = *reinterpret_cast<const float4 *>(sh_gC + st_k8 + 4);
                gk[0] = a4.x; gk[1] = a4.y; gk[2] = a4.z; gk[3] = a4.w; gk[4] = b4.x; gk[5] = b4.y; gk[6] = b4.z; gk[7] = b4.w;
            }
#pragma unroll
            for (int j = 0; j < 8; j++) {
                x[j] = ev[j] * gk[j];          // E' = E g_C[k]
                part[j] = ev[j] * hv[j];       // E * H_C, summed over v below
            }
            uint32_t hi[4], lo[4];
            put_row8(sm + L::XTh, sm + L::XTl, o_st, x, hi, lo);
            h0[0] = curm.hc[0];   // H_C of this chunk = H0 of the next (zeros across a head / sequence boundary on both sides)
            h0[1] = curm.hc[1];
            // rowsum over v (= over the 8 lanes of this wave with the same tid & 7, then over the 8 waves): transposing pair sums
            float s4[4], s2[2];
#pragma unroll
            for (int j = 0; j < 4; j++) s4[j] = swap32_sum(part[2 * j], part[2 * j + 1]);   // lanes < 32: even k, lanes >= 32: odd k
#pragma unroll
            for (int j = 0; j < 2; j++) s2[j] = swap16_sum(s4[2 * j], s4[2 * j + 1]);       // even 16-rows: j pairs 0, odd: 1
            const float a0 = s2[0] + dpp_mov<0x128>(s2[0]), a1 = s2[1] + dpp_mov<0x128>(s2[1]);   // row_ror:8 brings lane l ^ 8
            const float red = (lane & 8) ? a1 : a0;
            const int kk = st_k8 + ((lane >> 5) & 1) + 2 * ((lane >> 4) & 1) + 4 * ((lane >> 3) & 1);
            reinterpret_cast<float *>(sm + L::DT8)[wave * kN + kk] = red;
        };
        auto phase_a = [&]() {
            const int ln = fresh10(lane);
            if (wave <= 1) {   // both products on H0 (its planes are complete since I0): dQ: D[t][k] = sum_v dY[t][v] H0[v][k] ; dA: Z H0^T
                mma_sw_k64<64, false, false, 64, true, true>(acc1, sm + L::DYp, sm + L::DYp, 0, sm + L::HTh, sm + L::HTl, wave * 32, s);
                mma_sw_k64<64, false, true, 64, true, true>(acc3, sm + L::Zh, sm + L::Zl, 0, sm + L::HTh, sm + L::HTl, wave * 32, s);
            } else if (wave == 7) {
                f32x16 acc = zero16();  // D[m = s][n = t] = dy_s . v_t, s >= t -> P_vy[t][s]
                mma_sw_k64<64, false, false, 64, false, false>(acc, sm + L::DYp, sm + L::DYp, 0, sm + L::Vp, sm + L::Vp, 0, s);
                mask_upper_T<false>(acc, ln);
                store_T_split_sw32(acc, sm + L::P0 + 0 * 2 * L::A1, sm + L::P0 + 0 * 2 * L::A1 + L::A1, ln);
            } else if (wave == 2) {
                f32x16 acc = zero16();  // q~_t . k^_s, t >= s -> QKT[s][t]
                mma_sw_k64<64, false, true, 64, false, true>(acc, sm + L::QTh, sm + L::QTl, 0, sm + L::KHh, sm + L::KHl, 0, s);
                mask_upper_T<false>(acc, ln);
                store_T_split_sw32(acc, sm + L::QKTh, sm + L::QKTl, ln);
            } else if (wave == 3) {
                f32x16 acc = zero16();  // a~_t . k^_s, t > s -> AKT[s][t]
                mma_sw_k64<64, false, true, 64, false, true>(acc, sm + L::ATh, sm + L::ATl, 0, sm + L::KHh, sm + L::KHl, 0, s);
                mask_upper_T<true>(acc, ln);
                store_T_split_sw32(acc, sm + L::AKTh, sm + L::AKTl, ln);
            } else if (wave == 4) {
                f32x16 acc = zero16();  // z_s . u_t, s > t -> P_uz[t][s]
                mma_sw_k64<64, false, true, 64, false, true>(acc, sm + L::Zh, sm + L::Zl, 0, sm + L::Uh, sm + L::Ul, 0, s);
                mask_upper_T<true>(acc, ln);
                store_T_split_sw32(acc, sm + L::P0 + 3 * 2 * L::A1, sm + L::P0 + 3 * 2 * L::A1 + L::A1, ln);
            } else if (wave == 5) {
                f32x16 acc = zero16();  // dy_s . u_t, s >= t -> P_uy[t][s]
                mma_sw_k64<64, false, false, 64, false, true>(acc, sm + L::DYp, sm + L::DYp, 0, sm + L::Uh, sm + L::Ul, 0, s);
                mask_upper_T<false>(acc, ln);
                store_T_split_sw32(acc, sm + L::P0 + 2 * 2 * L::A1, sm + L::P0 + 2 * 2 * L::A1 + L::A1, ln);
            } else {   // wave 6
                f32x16 acz = zero16();  // z_s . v_t, s > t -> P_vz[t][s]
                mma_sw_k64<64, false, true, 64, false, false>(acz, sm + L::Zh, sm + L::Zl, 0, sm + L::Vp, sm + L::Vp, 0, s);
                mask_upper_T<true>(acz, ln);
                store_T_split_sw32(acz, sm + L::P0 + 1 * 2 * L::A1, sm + L::P0 + 1 * 2 * L::A1 + L::A1, ln);
            }
        };
        if (half == 0) {
            state_prologue();
            B10STAMP(3);
            phase_a();
            B10STAMP(4);
        } else {
            phase_a();
            B10STAMP(3);
            state_prologue();
            B10STAMP(4);
        }
        lds_barrier();
        B10STAMP(5);
        if (tid < kN) {   // wave 0: the eight partial row sums; it alone refills this part of the landing area, after these reads
            const float *d8 = reinterpret_cast<const float *>(sm + L::DT8);
            float t = 0.f;
#pragma unroll
            for (int wv = 0; wv < 8; wv++) t += d8[wv * kN + tid];
            sh_dterm[tid] = t;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        B10STAMP(12);
        // ---- I2: phase B, with the LDS-DMA of the next chunk's raw rows spread between its products (issued in one burst the 44
        // pieces queue at the texture addresser, 16 cycles each, and every wave stands in that queue: 0.9-1.3k cycles per chunk) ----------
        {
            const long cbn = more ? chunk_base(chunk + 1) : 0;
            const uint16_t *const Pvy = sm + L::P0, *const Pvz = sm + L::P0 + 2 * L::A1, *const Puy = sm + L::P0 + 4 * L::A1, *const Puz = sm + L::P0 + 6 * L::A1;
            dma_pieces(cbn, 0, 2);
            if (wave <= 1) {
                const int nb = wave * 32;   // dQ += P_vy^T K^ + P_uy^T B^ ;  dA += P_vz^T K^ + P_uz^T B^
                mma_sw<2, 32, true, true, 64, true, true>(acc1, Pvy, Pvy + L::A1, 0, sm + L::KHh, sm + L::KHl, nb, s);
                mma_sw<2, 32, true, true, 64, true, true>(acc3, Pvz, Pvz + L::A1, 0, sm + L::KHh, sm + L::KHl, nb, s);
                dma_pieces(cbn, 2, 4);
                mma_sw<2, 32, true, true, 64, true, true>(acc1, Puy, Puy + L::A1, 0, sm + L::BHh, sm + L::BHl, nb, s);
                dma_pieces(cbn, 4, 6);
                mma_sw<2, 32, true, true, 64, true, true>(acc3, Puz, Puz + L::A1, 0, sm + L::BHh, sm + L::BHl, nb, s);
            } else if (wave <= 3) {
                const int nb = (wave - 2) * 32;   // dK: V E'^T + P_vy Q~ + P_vz A~
                mma_sw_k64<64, false, false, 64, true, true>(acc2, sm + L::Vp, sm + L::Vp, 0, sm + L::XTh, sm + L::XTl, nb, s);
                dma_pieces(cbn, 2, 4);
                mma_sw<2, 32, false, true, 64, true, true>(acc2, Pvy, Pvy + L::A1, 0, sm + L::QTh, sm + L::QTl, nb, s);
                dma_pieces(cbn, 4, 6);
                mma_sw<2, 32, false, true, 64, true, true>(acc2, Pvz, Pvz + L::A1, 0, sm + L::ATh, sm + L::ATl, nb, s);
            } else if (wave >= 6) {
                const int nb = (wave - 6) * 32;   // dB: U E'^T + P_uy Q~ + P_uz A~
                mma_sw_k64<64, false, true, 64, true, true>(acc2, sm + L::Uh, sm + L::Ul, 0, sm + L::XTh, sm + L::XTl, nb, s);
                dma_pieces(cbn, 2, 4);
                mma_sw<2, 32, false, true, 64, true, true>(acc2, Puy, Puy + L::A1, 0, sm + L::QTh, sm + L::QTl, nb, s);
                dma_pieces(cbn, 4, 6);
                mma_sw<2, 32, false, true, 64, true, true>(acc2, Puz, Puz + L::A1, 0, sm + L::ATh, sm + L::ATl, nb, s);
            } else {
                const int vt = wave - 4;   // dV[s][v] = sum_t A_qk[t][s] dY[t][v] + sum_k k^[s][k] E'[k][v] + sum_t A_ak[t][s] Z[t][v]
                mma_sw<2, 32, false, true, 64, true, false>(acc2, sm + L::QKTh, sm + L::QKTl, 0, sm + L::DYp, sm + L::DYp, vt * 32, s);
                dma_pieces(cbn, 2, 4);
                mma_sw_k64<64, false, true, 64, false, true>(acc2, sm + L::KHh, sm + L::KHl, 0, sm + L::XTh, sm + L::XTl, vt * 32, s);
                dma_pieces(cbn, 4, 6);
                mma_sw<2, 32, false, true, 64, true, true>(acc2, sm + L::AKTh, sm + L::AKTl, 0, sm + L::Zh, sm + L::Zl, vt * 32, s);
            }
        }
        B10STAMP(6);
        lds_barrier();   // every operand plane, P and state plane is dead from here on
        B10STAMP(7);
        // the next chunk's E, H_C (used in its I0 / I1); curm is dead since I1
        curm = load_mats(chunk + 1, more);
        // ---- I3: the ten accumulator tiles -> fp32 staging over the operand planes ---------------------------------------------------------
        {
            const int lnS = fresh10(lane);
            if (wave <= 1) {
                stage_tile10(acc1, reinterpret_cast<float *>(sm + L::sQ), wave, lnS);
                stage_tile10(acc3, reinterpret_cast<float *>(sm + L::sA), wave, lnS);
            } else if (wave <= 3) stage_tile10(acc2, reinterpret_cast<float *>(sm + L::sK), wave - 2, lnS);
            else if (wave <= 5) stage_tile10(acc2, reinterpret_cast<float *>(sm + L::sV), wave - 4, lnS);
            else stage_tile10(acc2, reinterpret_cast<float *>(sm + L::sB), wave - 6, lnS);
        }
        B10STAMP(8);
        lds_barrier();
        B10STAMP(9);
        // ---- I4: epilogue: decay scaling, decay gradient, gradient rows -------------------------------------------------------------------
        {
            const float4 sQ4 = ld_stage4_10(sm + L::sQ, pt, pk), sK4 = ld_stage4_10(sm + L::sK, pt, pk), sB4 = ld_stage4_10(sm + L::sB, pt, pk),
                         sA4 = ld_stage4_10(sm + L::sA, pt, pk), sV4 = ld_stage4_10(sm + L::sV, pt, pk);
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
            uint16_t *os = sm + L::OUT;
            auto put = [&](int i, float x0, float x1, float x2, float x3) {
                *reinterpret_cast<uint2 *>(os + i * kC * L::OLD + pt * L::OLD + pk) = make_uint2(cvt_pk(x0, x1), cvt_pk(x2, x3));
            };
            put(0, dG[0], dG[1], dG[2], dG[3]); put(1, dQ[0], dQ[1], dQ[2], dQ[3]); put(2, dK[0], dK[1], dK[2], dK[3]);
            put(3, sV4.x, sV4.y, sV4.z, sV4.w); put(4, dA[0], dA[1], dA[2], dA[3]); put(5, dB[0], dB[1], dB[2], dB[3]);
        }
        B10STAMP(10);
        // this wave's pieces of the next chunk's rows have landed (issued ~6k cycles ago, the state records ~3k ago); the barrier below
        // makes everybody's visible before the next chunk reads them.  The gradient stores are issued behind the wait, so it never
        // waits for a store.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        lds_barrier();
        {
            const long off = chunk_base(chunk) + (long)lt * tstride + lk;
            // per-half pointer selects, not an indexed pointer array: that loses the address space and the stores become flat_store
            bf16_t *const o0 = half ? dv_ : dw_, *const o1 = half ? da_ : dq_, *const o2 = half ? db_ : dk_;
            const uint16_t *src = sm + L::OUT + half * 3 * kC * L::OLD + lt * L::OLD + lk;
            *reinterpret_cast<uint4 *>(o0 + off) = *reinterpret_cast<const uint4 *>(src);
            *reinterpret_cast<uint4 *>(o1 + off) = *reinterpret_cast<const uint4 *>(src + kC * L::OLD);
            *reinterpret_cast<uint4 *>(o2 + off) = *reinterpret_cast<const uint4 *>(src + 2 * kC * L::OLD);
        }
        B10STAMP(11);
        // the next chunk writes H0 planes where the gradient rows have just been read, and operand planes over the staging tiles
        lds_barrier();
    }  // chunk loop
#if B10TIMING
    lds_barrier();
    if (blockIdx.x == 0 && tid < 128) g_cbwd10_timing[tid] += tacc_[tid];
#endif
}

int chunk_bwd_out10_bf16(int B, int T_, int H, const void *w, const void *q, const void *k, const void *v, const void *a, const void *b,
                         const void *dy, const void *hs, const float *sa, const float *z, const void *e_vk, void *dw, void *dq, void *dk,
                         void *dv, void *da, void *db, hipStream_t st) {
    static DynLdsOnce lds_once;
    if (hipError_t e = lds_once.ensure(reinterpret_cast<const void *>(&wkv7c_bwd_out10_kernel), (int)Out10Smem::bytes); e != hipSuccess) return (int)e;
    (void)hipGetLastError();
    const int total = B * H * (T_ / kC);
    // chunks per workgroup: one workgroup per CU is resident (159.5 KB of LDS), so the launch runs in ceil(grid / CUs) rounds of cpw chunks
    // each.  More chunks per workgroup amortise its prologue (and the landing area is refilled one chunk ahead), but a grid just above a
    // multiple of the CU count pays a whole extra round for a few workgroups: 16 640 chunks (a packed row of 33 280 positions, H = 16) at
    // cpw = 64 are 260 workgroups = two rounds, 0.70 ms against 0.39 ms for 16 384 chunks.  Take the cpw with the fewest chunk-rounds
    // (+2 chunks' worth of prologue per round), the larger one on a tie.
    int cpw = kOut10MinChunksPerWG;
    {
        long best = -1;
        for (int c = kOut10MinChunksPerWG; c <= kOut10MaxChunksPerWG; c *= 2) {
            const long grid = (total + c - 1) / c, rounds = (grid + 255) / 256, cost = rounds * (c + 2);
            if (best < 0 || cost <= best) best = cost, cpw = c;
        }
    }
    Out10Rows rows;
    rows.p[0] = w; rows.p[1] = q; rows.p[2] = k; rows.p[3] = a; rows.p[4] = b; rows.p[5] = v; rows.p[6] = dy; rows.p[7] = sa; rows.p[8] = z;
    hipLaunchKernelGGL(wkv7c_bwd_out10_kernel, dim3((total + cpw - 1) / cpw), dim3(512), Out10Smem::bytes, st, rows, T_, H, total, cpw,
                       (const uint16_t *)hs, (const uint16_t *)e_vk, (bf16_t *)dw, (bf16_t *)dq, (bf16_t *)dk, (bf16_t *)dv, (bf16_t *)da,
                       (bf16_t *)db);
    return (int)hipGetLastError();
}

#ifdef WKV7C_TIMING
extern "C" int rwkv7_debug_cbwd10_timing(long long *out, int reset) {
    if (reset) {
        long long z[128] = {0};
        return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_cbwd10_timing), z, sizeof(z));
    }
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_cbwd10_timing), sizeof(long long) * 128);
}
#endif

}  // namespace rwkv7
