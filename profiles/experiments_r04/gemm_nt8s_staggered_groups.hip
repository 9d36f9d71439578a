// EXPERIMENT, not built: eight waves in two groups staggered by one step (a group reads its fragments and issues DMA while the other runs
// MFMAs; no fragment double buffering).  Bit-identical; 320 us against 231 us for csrc/gemm_nt4.hip (32768 x 4096 x 1024): a load step moves
// 64 KB through the LDS -- as long as the 512 cycles of MFMAs beside it -- plus its latency and a barrier per step.
// rwkvtts_amd/csrc/gemm_nt8.hip -- bf16 GEMM  C[M][N] = epi(A[M][K] . W[N][K]^T): the ring-of-half-tiles pipeline of csrc/gemm_nt4.hip with
// EIGHT waves (two per SIMD).  Why: with one wave per SIMD every instruction that is not an MFMA is serial overhead -- the interval
// stamps of gemm_nt4 show its K tile at 3.4k cycles for 2.0k cycles of MFMA: 32 fragment reads (~14 cycles each), 16 DMA instructions
// (~30 each), the barrier waits and the epilogue all add up on the single wave.  With two waves per SIMD one wave's reads / DMA /
// epilogue arithmetic issue while the other's MFMAs run.  The price is the smaller wave tile (128 x 64, 128 accumulator registers
// of the 256 a wave owns): 24 fragment reads per 32 MFMAs instead of 32 per 64, i.e. 192 KB of LDS reads per K tile instead of 128.
//
// Geometry: 256 x 256 x 64 tiles, waves 2 (m) x 4 (n).  Per K tile FOUR 16 KB units in request order: A0 (the first 64 rows of both
// wave rows), Blo (W rows 0..127: waves n = 0, 1), Bhi (W rows 128..255), A1 (the second 64 rows).  Two phases per K tile and wave:
//     P1(g): 16 MFMAs a0 x b;  reads A1(g) -> registers;          requests A0(g + 2), Blo(g + 2)
//     P2(g): 16 MFMAs a1 x b;  reads A0(g + 1), B(g + 1);         requests Bhi(g + 2), A1(g + 2)
// Ring of eight unit slots (unit u of K tile G -> slot (4 G + u) % 8); a unit is read exactly once per wave, one phase before its use;
// one barrier per phase; counted vmcnt (in order on gfx9): P1 waits with 8 younger DMA instructions outstanding, P2 with 6; the first
// three phases after an epilogue add the 16 stores of the epilogue to the count (the DMA they wait for was requested before it).
// Epilogues 0..3 as in gemm_nt4.hip, through a wave-private 4 KB staging tile: 8 full lines per store instruction.
#include "chunk_common.h"

namespace rwkv7 {
namespace {
constexpr int TM8 = 256, TN8 = 256, BK8 = 64;
constexpr int kRowB8 = BK8 * 2;            // bytes per LDS row
constexpr int kSlotB8 = 128 * kRowB8;      // a unit: 128 rows x 64 k = 16 KB
constexpr int kStageB8 = 32 * 128;         // per wave: 32 rows x 64 columns bf16
constexpr size_t kLds8 = 8 * kSlotB8 + 8 * kStageB8;   // 160 KB

__device__ __forceinline__ int swz8(int row) { return (row >> 1) & 7; }
__device__ __forceinline__ bf16x8 frag8(const char *slot, int row, int seg) {
    return *reinterpret_cast<const bf16x8 *>(slot + row * kRowB8 + ((seg ^ swz8(row)) << 4));
}
template <int I>
struct IC8 { static constexpr int value = I; };
template <int N, int I = 0, typename F>
__device__ __forceinline__ void static_for8(F &&f) {
    if constexpr (I < N) {
        f(IC8<I>{});
        static_for8<N, I + 1>(f);
    }
}
template <int N>
__device__ __forceinline__ void wait_vm8() { asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory"); }
struct Frags8 { bf16x8 f[4][2]; };   // [k-step][32-row tile of the 64 rows]
}  // namespace

template <int EPI>
__global__ __launch_bounds__(512) void gemm_nt8_kernel(int M, int N, int K, const uint16_t *__restrict__ A, const uint16_t *__restrict__ W,
                                                       uint16_t *__restrict__ C, const uint16_t *__restrict__ aux) {
    using gptr = const __attribute__((address_space(1))) void *;
    using lptr = __attribute__((address_space(3))) void *;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int rl = lane & 31, h = lane >> 5;
    const int nbn = N / TN8, nbm = M / TM8, ntiles = nbn * nbm, nk = K / BK8;
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
        row0 = bm * TM8;
        col0 = bn * TN8;
    };
    // LDS-DMA of a unit: piece r (0, 1) of a wave = slot rows (8 r + wave) * 8 .. + 7; lane -> row + lane / 8, 16-byte segment
    // (lane & 7), swizzled on the SOURCE side.  A units: slot row q = tile row (q < 64 ? q : q + 64) + 64 sub;  B units: q + 128 hi.
    uint32_t doffA[2][2], doffB[2][2];
#pragma unroll
    for (int s = 0; s < 2; s++)
#pragma unroll
        for (int r = 0; r < 2; r++) {
            const int q = (8 * r + wave) * 8 + (lane >> 3);
            const int sg = ((lane & 7) ^ swz8(q)) << 3;
            doffA[s][r] = (uint32_t)(((q < 64 ? q : q + 64) + 64 * s) * K + sg) * 2u;
            doffB[s][r] = (uint32_t)((q + 128 * s) * K + sg) * 2u;
        }
    const int my_tiles = (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    int d_t = 0, d_kt = 0, d_row0, d_col0;   // DMA cursor: the K tile whose units are requested next; clamps at the end
    tile_origin(blockIdx.x, d_row0, d_col0);
    auto advance = [&]() {
        if (d_kt + 1 < nk) d_kt++;
        else if (d_t + 1 < my_tiles) {
            d_t++;
            d_kt = 0;
            tile_origin(blockIdx.x + d_t * gridDim.x, d_row0, d_col0);
        }
    };
    // the two DMA instructions of unit u (0 A0, 1 Blo, 2 Bhi, 3 A1) of the cursor's K tile into slot `slot`
    auto dma_unit = [&](int u, int slot) {
        const bool isW = u == 1 || u == 2;
        const char *base = reinterpret_cast<const char *>(isW ? W : A) + ((long)(isW ? d_col0 : d_row0) * K + d_kt * BK8) * 2;
#pragma unroll
        for (int r = 0; r < 2; r++) {
            const uint32_t off = isW ? doffB[u == 2][r] : doffA[u == 3][r];
            __builtin_amdgcn_global_load_lds((gptr)(base + off), (lptr)(lds + slot * kSlotB8 + (8 * r + wave) * 8 * kRowB8), 16, 0, 0);
        }
    };

    f32x16 acc[2][4];        // [n tile][m tile]: D[m' = lane & 31][n' = 8 g + 4 h + e], register 4 g + e
    Frags8 FAx, FBx;         // the wave's 64 A rows (a0, then a1) and 64 W rows of the current K tile: no double buffering -- a wave reads
                             // its fragments in the step right before the MFMAs that use them, while the OTHER wave of its SIMD runs MFMAs
    // fragment addresses: one register per k-step, operand and 64 KB half of the ring; slot and 32-row tile are immediate offsets
    using lcp = const __attribute__((address_space(3))) char *;
    lcp pa[2][4], pb[2][4];
    {
        lcp l0 = (lcp)lds;
#pragma unroll
        for (int ks = 0; ks < 4; ks++) {
            const int sg = ((2 * ks + h) ^ swz8(rl)) << 4;
            pa[0][ks] = l0 + (wm * 64 + rl) * kRowB8 + sg;
            pa[1][ks] = pa[0][ks] + 4 * kSlotB8;
            pb[0][ks] = l0 + (wn >> 1) * kSlotB8 + ((wn & 1) * 64 + rl) * kRowB8 + sg;   // Blo / Bhi: slot 1 or 2 (+ 4)
            pb[1][ks] = pb[0][ks] + 4 * kSlotB8;
        }
    }
    auto ldfrag = [&](lcp p, int off) { return *reinterpret_cast<const __attribute__((address_space(3))) bf16x8 *>(p + off); };
    auto read_a = [&](int slot) {
#pragma unroll
        for (int ks = 0; ks < 4; ks++)
#pragma unroll
            for (int tl = 0; tl < 2; tl++) FAx.f[ks][tl] = ldfrag(pa[slot >> 2][ks], (slot & 3) * kSlotB8 + tl * 32 * kRowB8);
    };
    auto read_b = [&](int half) {
#pragma unroll
        for (int ks = 0; ks < 4; ks++)
#pragma unroll
            for (int tl = 0; tl < 2; tl++) FBx.f[ks][tl] = ldfrag(pb[half][ks], kSlotB8 + tl * 32 * kRowB8);
    };
    auto mma16 = [&](int sa) {   // acc[i][2 sa + j] += b(i) x a_sa(j), four k-steps
#pragma unroll
        for (int ks = 0; ks < 4; ks++)
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int j = 0; j < 2; j++)
                    acc[i][2 * sa + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FBx.f[ks][i], FAx.f[ks][j], acc[i][2 * sa + j], 0, 0, 0);
    };
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = zero16();
    // ---- prologue: K tile 0 (A0, Blo, Bhi, A1 -> slots 0..3), K tile 1 (A0, Blo, Bhi -> slots 4..6); cursor 1 = K tile 1, cursor 2 = K tile 2
#pragma unroll
    for (int u = 0; u < 4; u++) dma_unit(u, u);
    advance();
    int c1_row0 = d_row0, c1_kt = d_kt;   // cursor 1: the K tile whose A1 is requested next
#pragma unroll
    for (int u = 0; u < 3; u++) dma_unit(u, 4 + u);
    advance();
    auto dma_a1_c1 = [&](int slot) {   // A1 of cursor 1's K tile
        const char *base = reinterpret_cast<const char *>(A) + ((long)c1_row0 * K + c1_kt * BK8) * 2;
#pragma unroll
        for (int r = 0; r < 2; r++)
            __builtin_amdgcn_global_load_lds((gptr)(base + doffA[1][r]), (lptr)(lds + slot * kSlotB8 + (8 * r + wave) * 8 * kRowB8), 16, 0, 0);
    };
    wait_vm8<8>();
    const int grp = wm;   // waves w and w + 4 share a SIMD: group 1 runs ONE step behind group 0
    if (grp == 1) asm volatile("s_barrier" ::: "memory");
    constexpr int kStores = 16;   // epilogue stores per thread
    int sp = 0;   // DMA waits still to come whose awaited DMA is OLDER than this wave's last epilogue stores (two after every epilogue)
    int e_row0 = 0, e_col0 = 0;   // group 0: the tile whose epilogue is pending
    auto epilogue = [&](int row0, int col0) {
        char *stg = lds + 8 * kSlotB8 + wave * kStageB8;
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
            // EPI 2 / 3: aux of row block j, requested two blocks ahead by inline-asm loads counted by hand (see gemm_nt4.hip)
            u32x4 ax[2][4];
            const uint16_t *atile = EPI >= 2 ? aux + (long)(row0 + wm * 128 + (lane >> 3)) * N + col0 + wn * 64 + (lane & 7) * 8 : nullptr;
            auto load_aux = [&](int j, u32x4 (&dst)[4]) {
#pragma unroll
                for (int q = 0; q < 4; q++)
                    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst[q]) : "v"(atile + (long)(j * 32 + 8 * q) * N) : "memory");
            };
            if (EPI >= 2) {
                load_aux(0, ax[0]);
                load_aux(1, ax[1]);
            }
#pragma unroll
            for (int j = 0; j < 4; j++) {
#pragma unroll
                for (int i = 0; i < 2; i++)
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
                        asm volatile("ds_write_b64 %0, %1" ::"v"((uint32_t)(uintptr_t)(stg + rl * 128 + ((seg ^ swz8(rl)) << 4) + h * 8)), "v"(pk) : "memory");
                    }
#pragma unroll
                for (int i = 0; i < 2; i++) acc[i][j] = zero16();
                // back: 8 rows x 128 B per instruction (lane -> row 8 q + lane / 8, segment lane & 7): full lines
                uint16_t *cblk = C + (long)(row0 + wm * 128 + j * 32) * N + col0 + wn * 64;
                if (EPI >= 2) {
                    // vmcnt is in order: behind aux(j) sit  j = 0: aux(1);  1: stores(0), aux(2);  2: stores(1), aux(3);  3: stores(2)
                    u32x4(&a)[4] = ax[j & 1];
                    if (j == 0 || j == 3) asm volatile("s_waitcnt vmcnt(4)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3])::"memory");
                    else asm volatile("s_waitcnt vmcnt(8)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3])::"memory");
                }
                u32x4 v4[4];
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int row = 8 * q + (lane >> 3), seg = lane & 7;
                    asm volatile("ds_read_b128 %0, %1" : "=v"(v4[q]) : "v"((uint32_t)(uintptr_t)(stg + row * 128 + ((seg ^ swz8(row)) << 4))) : "memory");
                }
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v4[0]), "+v"(v4[1]), "+v"(v4[2]), "+v"(v4[3])::"memory");
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int row = 8 * q + (lane >> 3), seg = lane & 7;
                    uint4 v = make_uint4(v4[q][0], v4[q][1], v4[q][2], v4[q][3]);
                    if (EPI >= 2) {
                        // 2: ds (bf16) * 2 relu(h), aux = h;  3: ds (bf16) * 2 sqrt(s), aux = s = relu(h)^2
                        uint32_t *pv = reinterpret_cast<uint32_t *>(&v);
#pragma unroll
                        for (int e2 = 0; e2 < 4; e2++) {
                            const uint32_t au = ax[j & 1][q][e2];
                            const float x0 = __uint_as_float(pv[e2] << 16), x1 = __uint_as_float(pv[e2] & 0xffff0000u);
                            const float h0 = __uint_as_float(au << 16), h1 = __uint_as_float(au & 0xffff0000u);
                            if (EPI == 2) pv[e2] = cvt_pk(h0 > 0.f ? 2.f * h0 * x0 : 0.f, h1 > 0.f ? 2.f * h1 * x1 : 0.f);
                            else pv[e2] = cvt_pk(2.f * __builtin_sqrtf(fmaxf(h0, 0.f)) * x0, 2.f * __builtin_sqrtf(fmaxf(h1, 0.f)) * x1);
                        }
                    }
                    *reinterpret_cast<uint4 *>(cblk + (long)row * N + seg * 8) = v;
                }
                if (EPI >= 2 && j + 2 < 4) load_aux(j + 2, ax[j & 1]);
            }
    };
    auto step_end = [&](bool wait_dma) {
        // every step ends in a barrier; where this wave's pieces of the units read next (by group 0 first) must have landed, vmcnt(8)
        // in front of it (in order: 8 younger DMA instructions, + the 16 stores of an epilogue in between)
        if (wait_dma) {
            if (sp > 0) {
                wait_vm8<8 + kStores>();
                sp--;
            } else {
                wait_vm8<8>();
            }
        } else {
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
        __builtin_amdgcn_sched_barrier(0);
    };

    for (int t = 0; t < my_tiles; t++) {
        int row0, col0;
        tile_origin(blockIdx.x + t * gridDim.x, row0, col0);
        for (int kt0 = 0; kt0 < nk; kt0 += 16) {
            const bool first_trip = kt0 == 0, last_trip = kt0 + 16 >= nk;
            static_for8<16>([&](auto KT) {
                constexpr int kt = decltype(KT)::value, e = kt & 1;
                // ---- L1: fragments of a0 and b <- A0(g), B(g); request A1(g + 1) (slot of A1(g - 1), read two steps ago by group 1)
                if (grp == 0 && kt == 0 && first_trip && t > 0) {
                    epilogue(e_row0, e_col0);   // group 0's epilogue of the previous tile: in the same interval as group 1's (its last L4)
                    sp = 2;
                }
                read_a(4 * e + 0);
                read_b(e);
                dma_a1_c1(4 * (1 - e) + 3);
                step_end(grp == 1);
                // ---- L2: a0 x b
                mma16(0);
                step_end(grp == 0);
                // ---- L3: fragments of a1 <- A1(g); request A0, Blo, Bhi of K tile g + 2 (slots of K tile g, read in L1 by both groups)
                read_a(4 * e + 3);
                dma_unit(0, 4 * e + 0);
                dma_unit(1, 4 * e + 1);
                dma_unit(2, 4 * e + 2);
                c1_row0 = d_row0;
                c1_kt = d_kt;
                advance();
                step_end(grp == 1);
                // ---- L4: a1 x b
                mma16(1);
                if (grp == 1 && kt == 15 && last_trip) {
                    epilogue(row0, col0);
                    sp = 2;
                }
                step_end(grp == 0);
            });
        }
        e_row0 = row0;
        e_col0 = col0;
    }
    if (grp == 0) {
        epilogue(e_row0, e_col0);
        asm volatile("s_barrier" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

namespace {
template <int EPI>
int launch_gemm8(int M, int N, int K, const void *A, const void *W, void *C, const void *aux, hipStream_t st) {
    static bool attr = false;
    auto kern = &gemm_nt8_kernel<EPI>;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLds8);
        if (e != hipSuccess) return (int)e;
        attr = true;
    }
    (void)hipGetLastError();
    const int ntiles = (M / TM8) * (N / TN8);
    kern<<<dim3(ntiles < 256 ? ntiles : 256), dim3(512), kLds8, st>>>(M, N, K, (const uint16_t *)A, (const uint16_t *)W, (uint16_t *)C,
                                                                      (const uint16_t *)aux);
    return (int)hipGetLastError();
}
}  // namespace

// shapes: M, N multiples of 256, K a multiple of 1024 (16 K tiles per unrolled trip)
int gemm_nt8_bf16(int M, int N, int K, const void *A, const void *W, void *C, const void *aux, int epilogue, hipStream_t st) {
    if (M <= 0 || N <= 0 || K <= 0 || M % TM8 || N % TN8 || K % (16 * BK8)) return -1;
    if (epilogue == 0) return launch_gemm8<0>(M, N, K, A, W, C, nullptr, st);
    if (epilogue == 1) return launch_gemm8<1>(M, N, K, A, W, C, nullptr, st);
    if (epilogue == 2 && aux) return launch_gemm8<2>(M, N, K, A, W, C, aux, st);
    if (epilogue == 3 && aux) return launch_gemm8<3>(M, N, K, A, W, C, aux, st);
    return -1;
}

}  // namespace rwkv7
