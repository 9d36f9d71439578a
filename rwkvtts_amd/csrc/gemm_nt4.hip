// rwkvtts_amd/csrc/gemm_nt4.hip -- bf16 GEMM  C[M][N] = epi(A[M][K] . W[N][K]^T), second generation (round 4) of the own MFMA GEMM
// (csrc/gemm_relusq.hip is the first).  Same problem and epilogues (0 plain, 1 relu(.)^2 = the channel-mix key activation,
// rwkv_s2s_single_ffn.py:228, 2 its backward as the epilogue of the value projection's input-gradient GEMM from the pre-activation h,
// 3 the same from the activation's output s, 4 the residual add behind a projection).
//
// What the first kernel's ablation and interval stamps showed (tools/gemm_lab, 32768 x 4096 x 1024): (i) the texture-address path
// costs ~2.5 cycles per distinct 128-byte line and instruction (64 B/clk of L1): 1280 cycles for the 64 KB of a 256 x 256 x 64 K
// tile -- 60 % of that tile's MFMA time -- and issued as one burst behind the barrier those cycles are EXPOSED; (ii) the 8-byte
// scattered epilogue stores touch 32 lines per instruction (10k cycles per tile) and, sharing vmcnt with the LDS-DMA, never drained
// under the next tile; (iii) 8 waves x (128 x 64) wave tiles read 192 KB of fragments per K tile: with the 64 KB of DMA writes that
// is the whole LDS bandwidth at MFMA peak.  Here:
//   * FOUR waves, 128 x 128 wave tiles (4 x 4 MFMA tiles of 32 x 32, 256 accumulator registers, one wave per SIMD with the whole
//     512-register file): 32 fragment reads per 64 MFMAs, 128 KB of LDS reads per K tile;
//   * a K tile is consumed in four QUADRANT phases (a0 b0, a0 b1, a1 b1, a1 b0; a_s / b_s = 64-row halves of the wave's A / W rows,
//     16 MFMAs = 2 x 2 tiles x K 64 each).  Every phase reads ONE half tile (128 rows x 64 k = 16 KB: the a_s or b_s rows of all
//     waves) into registers for a later phase and requests ONE half tile six phases ahead: the LDS is a ring of eight half-tile
//     slots, a slot is free again one phase after its only read, the DMA is issued 4 instructions per wave and phase between the
//     MFMAs instead of 16 in a burst, and every load has 1.5 K tiles of latency budget (counted vmcnt, never 0 inside the loop);
//   * ONE software pipeline over all K tiles of all tiles of a workgroup (persistent, XCD-aware tile patches as before);
//   * one barrier per TWO phases;
//   * epilogue through a wave-private LDS staging tile: full 128-byte lines leave, 8 lines per store instruction (2.5k instead of
//     10k address cycles per tile).  vmcnt is in order on gfx9 (loads and stores retire in issue order), so the DMA requested before
//     the epilogue is awaited with the epilogue's stores still in flight (vmcnt(20 + 32)); nothing waits for a store before the
//     seventh phase of the next tile.
// Measured (tools/gemm_lab/bench.py: operands rotating through 2 GB, variants interleaved in one process; 32768 x 4096 x 1024):
// library (hipBLASLt) 215-219 us, this kernel 231 us, the first generation 320 us; 32768 x 1024 x 1024: 59-62 / 64.6 / 80-90 us.  As a plain
// GEMM it stays 7 % behind the library; with the activation as epilogue it replaces GEMM + elementwise kernel pairs (fused.channel_mix).
// What bounds it (interval stamps, GEMM4_TIMING): 3.4k cycles per K tile for 2.0k of MFMA.  On a LONE wave nothing issues in the
// shadow of an MFMA -- two dummy VALU instructions behind every MFMA cost +12 %, four +41 % -- so the 32 fragment reads (~14 cycles),
// the 16 DMA instructions with their address arithmetic (~30), two barriers and the epilogue (5.2k cycles per tile) all ADD to the
// MFMA time.  Tried: other interleavings of reads / DMA / MFMAs (within 2 %); one barrier per two phases (-1.3 %, kept); the same ring
// with EIGHT waves of 128 x 64 (two per SIMD, so that one wave's issue hides behind the other's MFMAs): bit-identical, 267 us -- the
// 192 KB of fragment reads per K tile saturate the LDS (profiles/experiments_r04/gemm_nt8_ring_8waves.hip); the same eight waves as two
// groups staggered by one step, one reading fragments while the other runs MFMAs: 320 us (gemm_nt8s_staggered_groups.hip: a load step
// moves 64 KB through the LDS, as long as the MFMAs beside it, plus latency and a barrier); the DMA in its scalar-base form
// (inline asm, no 64-bit VALU add per instruction): +-0.
#include "chunk_common.h"
#include "launch_attr.h"
#ifndef GEMM4_EXP
#define GEMM4_EXP 0
#endif

namespace rwkv7 {
namespace {
constexpr int TM4 = 256, TN4 = 256, BK4 = 64;
constexpr int kRowB4 = BK4 * 2;            // bytes per LDS row
constexpr int kSlotB4 = 128 * kRowB4;      // a half tile: 128 rows x 64 k = 16 KB
constexpr int kStageB4 = 32 * 256;         // per wave: 32 rows x 128 columns bf16
constexpr size_t kLds4 = 8 * kSlotB4 + 4 * kStageB4;   // 160 KB
constexpr int kAhead = 6;                  // half tiles requested ahead of the one being read
constexpr int kDmaPerPhase = 4;            // DMA instructions per wave and half tile

__device__ __forceinline__ int swz4(int row) { return (row >> 1) & 7; }
__device__ __forceinline__ bf16x8 frag4(const char *slot, int row, int seg) {
    return *reinterpret_cast<const bf16x8 *>(slot + row * kRowB4 + ((seg ^ swz4(row)) << 4));
}
template <int I>
struct IC { static constexpr int value = I; };
template <int N, int I = 0, typename F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (I < N) {
        f(IC<I>{});
        static_for<N, I + 1>(f);
    }
}
template <int N>
__device__ __forceinline__ void wait_vm4() { asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory"); }

// half tile sequence: h = 0: A0(0), 1: B0(0), then 2 + 4 g + {0: B1(g), 1: A1(g), 2: A0(g + 1), 3: B0(g + 1)}: the order of the reads.
struct Frags { bf16x8 f[4][2]; };   // [k-step][32-row tile of the 64-row half]
}  // namespace

template <int EPI>
__global__ __launch_bounds__(256) void gemm_nt4_kernel(int M, int N, int K, const uint16_t *__restrict__ A, const uint16_t *__restrict__ W,
                                                       uint16_t *__restrict__ C, const uint16_t *__restrict__ aux) {
    using gptr = const __attribute__((address_space(1))) void *;
    using lptr = __attribute__((address_space(3))) void *;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int rl = lane & 31, h = lane >> 5;
    const int nbn = N / TN4, nbm = M / TM4, ntiles = nbn * nbm, nk = K / BK4;
    // id -> tile: ids b, b + 8, ... share an XCD (one L2), 32 at a time: those 32 form a (32 / pc) x pc patch of tiles
    const int pc = nbn % 8 == 0 ? 8 : 4, pr = 32 / pc;
    const bool patched = (gridDim.x & 7) == 0 && nbm % (8 * pr) == 0 && nbn % pc == 0;
    auto tile_origin = [&](int id, int &row0, int &col0) {
        int bm, bn;
        if (patched) {
            const int xcd = id & 7, j = id >> 3, nround_n = nbn / pc, r = j / 32, i = j % 32;
            bn = pc * (r % nround_n) + (i % pc);
            bm = xcd + 8 * (pr * (r / nround_n) + (i / pc));
        } else {
            bn = id % nbn;
            bm = id / nbn;
        }
        row0 = bm * TM4;
        col0 = bn * TN4;
    };
    // LDS-DMA of a half tile (sub = 0 / 1): slot row q (0..127) = tile row (q < 64 ? q : q + 64) + 64 sub -- the a_sub (b_sub) rows of
    // the waves with wm (wn) = 0, then 1.  Piece r (0..3) of a wave = slot rows (4 r + wave) * 8 .. + 7; lane -> row + lane / 8,
    // 16-byte segment (lane & 7), swizzled on the SOURCE side (the destination is wave-uniform base + lane * 16).
    uint32_t doff[2][kDmaPerPhase];
#pragma unroll
    for (int sub = 0; sub < 2; sub++)
#pragma unroll
        for (int r = 0; r < kDmaPerPhase; r++) {
            const int q = (4 * r + wave) * 8 + (lane >> 3);
            const int trow = (q < 64 ? q : q + 64) + 64 * sub;
            doff[sub][r] = (uint32_t)(trow * K + (((lane & 7) ^ swz4(q)) << 3)) * 2u;
        }
    // DMA cursor: the K tile whose half tiles are requested next (tile t, K tile kt of it); clamps at the end (the last K tile is
    // requested again into slots nobody reads any more: branch-free)
    const int my_tiles = (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    int d_t = 0, d_kt = 0, d_row0, d_col0;
    tile_origin(blockIdx.x, d_row0, d_col0);
    auto advance = [&]() {
        if (d_kt + 1 < nk) d_kt++;
        else if (d_t + 1 < my_tiles) {
            d_t++;
            d_kt = 0;
            tile_origin(blockIdx.x + d_t * gridDim.x, d_row0, d_col0);
        }
    };
    // one DMA instruction of the half tile `kind` (0 A0, 1 B0, 2 B1, 3 A1) of the cursor's K tile into slot `slot`
    auto dma1 = [&](int kind, int r, int slot) {
        const bool isW = kind == 1 || kind == 2;
        const int sub = kind >= 2;
        const char *base = reinterpret_cast<const char *>(isW ? W : A) + ((long)(isW ? d_col0 : d_row0) * K + d_kt * BK4) * 2;
        __builtin_amdgcn_global_load_lds((gptr)(base + doff[sub][r]), (lptr)(lds + slot * kSlotB4 + (4 * r + wave) * 8 * kRowB4), 16, 0, 0);
    };

    f32x16 acc[4][4];        // [n tile][m tile]: D[m' = lane & 31][n' = 8 g + 4 h + e], register 4 g + e
    Frags FA[2], FB[2];      // A halves a0, a1; W halves: FB[e] is b0 for even K tiles, b1 for odd ones
    auto read_half = [&](Frags &F, int slot, int w01) {   // the wave's 64 rows of the half tile in `slot`
        const char *sl = lds + slot * kSlotB4;
#pragma unroll
        for (int ks = 0; ks < 4; ks++)
#pragma unroll
            for (int tl = 0; tl < 2; tl++) F.f[ks][tl] = frag4(sl, w01 * 64 + tl * 32 + rl, 2 * ks + h);
    };
    // quadrant (a_sa, b_sb): acc[2 sb + i'][2 sa + j'] += b-frag(i') x a-frag(j') over the four k-steps
    auto quadrant = [&](const Frags &Fa, const Frags &Fb, int sa, int sb) {
#pragma unroll
        for (int ks = 0; ks < 4; ks++)
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int j = 0; j < 2; j++)
                    acc[2 * sb + i][2 * sa + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Fb.f[ks][i], Fa.f[ks][j], acc[2 * sb + i][2 * sa + j], 0, 0, 0);
    };

#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = zero16();
    // ---- prologue: half tiles 0..7 = K tiles 0 and 1 (A0, B0, B1, A1 each) -> slots 0..7; the first two read
#pragma unroll
    for (int kk = 0; kk < 2; kk++) {
#pragma unroll
        for (int kind = 0; kind < 4; kind++)
#pragma unroll
            for (int r = 0; r < kDmaPerPhase; r++) dma1(kind, r, 4 * kk + kind);
        advance();
    }
    wait_vm4<6 * kDmaPerPhase>();
    read_half(FA[0], 0, wm);
    read_half(FB[0], 1, wn);
#ifdef GEMM4_TIMING
    long long tseg[3] = {0, 0, 0};
    long long tprev = __builtin_readcyclecounter();
    const long long tstart_c = tprev, tstart_r = __builtin_amdgcn_s_memrealtime();
#define G4STAMP(i) { const long long now_ = __builtin_readcyclecounter(); tseg[i] += now_ - tprev; tprev = now_; }
#else
#define G4STAMP(i)
#endif
    constexpr int kInFlight = (kAhead - 1) * kDmaPerPhase;   // DMA instructions younger than the half tile a phase needs
    constexpr int kInFlight2 = (kAhead - 2) * kDmaPerPhase;  // ... than the half tiles a PAIR of phases needs
    constexpr int kStores = 32;                               // epilogue stores per thread
    bool stores_behind = false;   // the previous tile's stores are younger than the DMA the first six phases wait for

    for (int t = 0; t < my_tiles; t++) {
        int row0, col0;
        tile_origin(blockIdx.x + t * gridDim.x, row0, col0);
        for (int kt0 = 0; kt0 < nk; kt0 += 16) {
            static_for<16>([&](auto KT) {
                constexpr int kt = decltype(KT)::value, e = kt & 1;
                // half tile read in phase p (0..3) of this K tile: h = 2 + 4 g + p, slot h % 8 (16 K tiles per trip: 4 g = 4 kt mod 8)
                constexpr int s0 = (2 + 4 * kt) % 8;
                auto phase = [&](auto P) {
                    constexpr int p = decltype(P)::value;
                    constexpr int slot_r = (s0 + p) % 8, slot_w = (s0 + p + kAhead) % 8;
                    G4STAMP(1)
                    // one barrier per TWO phases: the half tiles of this phase and the next have landed (everybody's pieces), everybody is
                    // done with the slots the two phases' DMA will overwrite (read in the previous pair of phases)
                    if (p % 2 == 0) {
                        if (kt == 0 || (kt == 1 && p == 0)) {
                            if (stores_behind) wait_vm4<kInFlight2 + kStores>();
                            else wait_vm4<kInFlight2>();
                        } else {
                            wait_vm4<kInFlight2>();
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    G4STAMP(0)
                    // reads: p 0: B1(g) -> FB[1 - e];  1: A1(g) -> FA[1];  2: A0(g + 1) -> FA[0];  3: B0(g + 1) -> FB[1 - e]
                    if (p == 0) read_half(FB[1 - e], slot_r, wn);
                    if (p == 1) read_half(FA[1], slot_r, wm);
                    if (p == 2) read_half(FA[0], slot_r, wm);
                    if (p == 3) read_half(FB[1 - e], slot_r, wn);
                    // requested: half tile h + 6 = A0, B0, B1, A1 (p = 0..3) of the cursor's K tile (g + 2)
                    if (!(GEMM4_EXP & 2)) {
#pragma unroll
                        for (int r = 0; r < kDmaPerPhase; r++) dma1(p, r, slot_w);
                    }
                    if (p == 3) advance();
                    // MFMAs: p 0: a0 b0;  1: a0 b1;  2: a1 b1;  3: a1 b0     (b0 = FB[e], b1 = FB[1 - e])
                    if (p == 0) quadrant(FA[0], FB[e], 0, 0);
                    if (p == 1) quadrant(FA[0], FB[1 - e], 0, 1);
                    if (p == 2) quadrant(FA[1], FB[1 - e], 1, 1);
                    if (p == 3) quadrant(FA[1], FB[e], 1, 0);
                    // issue order: MFMA, ds_read (x 8), MFMA, DMA (x 4), 4 MFMAs -- the reads and the DMA go out in the shadow of the MFMAs
#pragma unroll
                    for (int q = 0; q < 8; q++) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
#pragma unroll
                    for (int q = 0; q < kDmaPerPhase; q++) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                    __builtin_amdgcn_sched_barrier(0);
                };
                phase(IC<0>{});
                phase(IC<1>{});
                phase(IC<2>{});
                phase(IC<3>{});
            });
            stores_behind = false;
        }
        G4STAMP(1)
        // ---- epilogue: 32 rows x 128 columns at a time through the wave's staging tile ([32][256 B], 16-byte segments XOR (row & 15))
        char *stg = lds + 8 * kSlotB4 + wave * kStageB4;
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        // EPI 2 / 3: the aux operand of row block j, requested two blocks ahead by INLINE-ASM loads counted by hand (beside LDS-DMA in
        // flight hipcc answers a visible global load with s_waitcnt vmcnt(0): the whole DMA ring would drain four times per tile)
        u32x4 ax[2][8];
        const uint16_t *atile = EPI >= 2 ? aux + (long)(row0 + wm * 128 + (lane >> 4)) * N + col0 + wn * 128 + (lane & 15) * 8 : nullptr;
        auto load_aux = [&](int j, u32x4 (&dst)[8]) {
#pragma unroll
            for (int q = 0; q < 8; q++)
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst[q]) : "v"(atile + (long)(j * 32 + 4 * q) * N) : "memory");
        };
        if (EPI >= 2) {
            load_aux(0, ax[0]);
            load_aux(1, ax[1]);
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int gq = 0; gq < 4; gq++) {
                    float x[4];
#pragma unroll
                    for (int e2 = 0; e2 < 4; e2++) x[e2] = acc[i][j][4 * gq + e2];
                    if (EPI == 1) {   // relu(bf16(x))^2, rounded again: what the two separate kernels produce
                        const uint32_t r0 = cvt_pk(x[0], x[1]), r1 = cvt_pk(x[2], x[3]);
                        x[0] = __uint_as_float(r0 << 16); x[1] = __uint_as_float(r0 & 0xffff0000u);
                        x[2] = __uint_as_float(r1 << 16); x[3] = __uint_as_float(r1 & 0xffff0000u);
#pragma unroll
                        for (int e2 = 0; e2 < 4; e2++) x[e2] = x[e2] * fmaxf(x[e2], 0.f);
                    }
                    // columns i * 32 + 8 gq + 4 h + (0..3) of row rl: segment (i * 4 + gq), half h
                    const int seg = i * 4 + gq;
                    const uint2 pk = make_uint2(cvt_pk(x[0], x[1]), cvt_pk(x[2], x[3]));
                    // inline asm: a compiler-visible ds_write with LDS-DMA in flight gets an s_waitcnt vmcnt(0) in front of it (alias rule)
                    asm volatile("ds_write_b64 %0, %1" ::"v"((uint32_t)(uintptr_t)(stg + rl * 256 + ((seg ^ (rl & 15)) << 4) + h * 8)), "v"(pk) : "memory");
                }
#pragma unroll
            for (int i = 0; i < 4; i++) acc[i][j] = zero16();
            // back: 4 rows x 256 B per instruction (lane -> row 4 q + lane / 16, segment lane & 15): two full lines per row
            uint16_t *cblk = C + (long)(row0 + wm * 128 + j * 32) * N + col0 + wn * 128;
            // the reads too are inline asm (a compiler-visible ds_read of this tile would be given the same vmcnt(0)); four at a time
            if (EPI >= 2) {
                // vmcnt is in order: behind aux(j) sit  j = 0: aux(1);  1: stores(0), aux(2);  2: stores(1), aux(3);  3: stores(2)
                u32x4(&a)[8] = ax[j & 1];
                if (j == 0 || j == 3)
                    asm volatile("s_waitcnt vmcnt(8)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])::"memory");
                else
                    asm volatile("s_waitcnt vmcnt(16)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])::"memory");
            }
#pragma unroll
            for (int qh = 0; qh < 2; qh++) {
                u32x4 v8[4];
#pragma unroll
                for (int q4 = 0; q4 < 4; q4++) {
                    const int row = 4 * (4 * qh + q4) + (lane >> 4), seg = lane & 15;
                    asm volatile("ds_read_b128 %0, %1" : "=v"(v8[q4]) : "v"((uint32_t)(uintptr_t)(stg + row * 256 + ((seg ^ (row & 15)) << 4))) : "memory");
                }
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v8[0]), "+v"(v8[1]), "+v"(v8[2]), "+v"(v8[3])::"memory");
#pragma unroll
                for (int q4 = 0; q4 < 4; q4++) {
                    const int q = 4 * qh + q4, row = 4 * q + (lane >> 4), seg = lane & 15;
                    uint4 v = make_uint4(v8[q4][0], v8[q4][1], v8[q4][2], v8[q4][3]);
                    if (EPI >= 2) {
                        // 2: ds (bf16) * 2 relu(h), aux = h: what the library GEMM + rwkv7_relusq_bwd produce
                        // 3: ds (bf16) * 2 sqrt(s), aux = s = relu(h)^2: what the library GEMM + rwkv7_relusq_bwd_s produce
                        uint32_t *pv = reinterpret_cast<uint32_t *>(&v);
#pragma unroll
                        for (int e2 = 0; e2 < 4; e2++) {
                            const uint32_t au = ax[j & 1][q][e2];
                            const float x0 = __uint_as_float(pv[e2] << 16), x1 = __uint_as_float(pv[e2] & 0xffff0000u);
                            const float h0 = __uint_as_float(au << 16), h1 = __uint_as_float(au & 0xffff0000u);
                            if (EPI == 2) pv[e2] = cvt_pk(h0 > 0.f ? 2.f * h0 * x0 : 0.f, h1 > 0.f ? 2.f * h1 * x1 : 0.f);
                            else if (EPI == 4) pv[e2] = cvt_pk(x0 + h0, x1 + h1);   // residual add: bf16(bf16(A W^T) + aux), what Linear + add produce
                            else {
                                // s = bf16(relu(h)^2) >= 0: no clamp (the four v_max per pair that fmaxf costs are paid for on a lone wave).
                                // (the factor 2 as the square root's output modifier does not work: IEEE mode ignores omod)
                                const float q0 = 2.f * __builtin_sqrtf(h0), q1 = 2.f * __builtin_sqrtf(h1);
                                pv[e2] = cvt_pk(q0 * x0, q1 * x1);
                            }
                        }
                    }
                    if (!(GEMM4_EXP & 4) || v.x == 0x12345u) *reinterpret_cast<uint4 *>(cblk + (long)row * N + seg * 8) = v;
                }
            }
            if (EPI >= 2 && j + 2 < 4) load_aux(j + 2, ax[j & 1]);
        }
        stores_behind = true;
        G4STAMP(2)
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef GEMM4_TIMING
    if (EPI == 0 && aux && lane == 0) {   // lab only: aux = long long [grid][4 waves][4]
        long long *o = reinterpret_cast<long long *>(const_cast<uint16_t *>(aux)) + ((long)blockIdx.x * 4 + wave) * 4;
        o[0] = tseg[0]; o[1] = tseg[1]; o[2] = tseg[2];
        o[3] = ((__builtin_readcyclecounter() - tstart_c) << 24) / (__builtin_amdgcn_s_memrealtime() - tstart_r);   // cycles per 10 ns, 24 fraction bits
    }
#endif
}

namespace {
template <int EPI>
int launch_gemm4(int M, int N, int K, const void *A, const void *W, void *C, const void *aux, hipStream_t st) {
    static DynLdsOnce lds_once;
    auto kern = &gemm_nt4_kernel<EPI>;
    if (hipError_t e = lds_once.ensure(reinterpret_cast<const void *>(kern), (int)kLds4); e != hipSuccess) return (int)e;
    (void)hipGetLastError();
    const int ntiles = (M / TM4) * (N / TN4);
    kern<<<dim3(ntiles < 256 ? ntiles : 256), dim3(256), kLds4, st>>>(M, N, K, (const uint16_t *)A, (const uint16_t *)W, (uint16_t *)C,
                                                                      (const uint16_t *)aux);
    return (int)hipGetLastError();
}
}  // namespace

// shapes: M, N multiples of 256, K a multiple of 1024 (16 K tiles per unrolled trip)
int gemm_nt4_bf16(int M, int N, int K, const void *A, const void *W, void *C, const void *aux, int epilogue, hipStream_t st) {
    if (M <= 0 || N <= 0 || K <= 0 || M % TM4 || N % TN4 || K % (16 * BK4)) return -1;
    if (epilogue == 0) return launch_gemm4<0>(M, N, K, A, W, C, aux, st);   // aux: lab timing buffer (GEMM4_TIMING builds) or null
    if (epilogue == 1) return launch_gemm4<1>(M, N, K, A, W, C, nullptr, st);
    if (epilogue == 2 && aux) return launch_gemm4<2>(M, N, K, A, W, C, aux, st);
    if (epilogue == 3 && aux) return launch_gemm4<3>(M, N, K, A, W, C, aux, st);
    if (epilogue == 4 && aux) return launch_gemm4<4>(M, N, K, A, W, C, aux, st);
    return -1;
}

}  // namespace rwkv7
