// EXPERIMENT, not built: the eight-wave variant of csrc/gemm_nt4.hip (see that file's header).  Bit-identical results; 267 us against 231 us
// (32768 x 4096 x 1024, tools/gemm_lab): kept as the record of the measurement.
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
    Frags8 FAx, FBx;         // the 64 A rows / 64 W rows in use; a k-step's fragments are REPLACED by the next unit's right behind its MFMAs
                             // (explicitly the same variables: four independent sets did not fit the 256 registers of a wave)
    // fragment addresses: ONE register per k-step, operand and 64 KB half of the ring; slot and 32-row tile are immediate offsets of the
    // ds_read (left to itself hipcc kept 36 address registers and spilled inside the loop).  The XOR swizzle depends on the lane only:
    // (row >> 1) & 7 = (rl >> 1) & 7 for rows = multiple of 16 + rl.
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
    // the MFMAs of phase (a_sa x b) with, behind every k-step, the reads of the next A unit (slot_a) and, if bhalf >= 0, the next B unit
    // (slot 4 bhalf + 1 or + 2 by wave)
    auto phase_mma = [&](int sa, int slot_a, int bhalf) {
#pragma unroll
        for (int ks = 0; ks < 4; ks++) {
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int j = 0; j < 2; j++)
                    acc[i][2 * sa + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FBx.f[ks][i], FAx.f[ks][j], acc[i][2 * sa + j], 0, 0, 0);
#pragma unroll
            for (int tl = 0; tl < 2; tl++) FAx.f[ks][tl] = ldfrag(pa[slot_a >> 2][ks], (slot_a & 3) * kSlotB8 + tl * 32 * kRowB8);
            if (bhalf >= 0) {
#pragma unroll
                for (int tl = 0; tl < 2; tl++) FBx.f[ks][tl] = ldfrag(pb[bhalf][ks], kSlotB8 + tl * 32 * kRowB8);
            }
        }
    };
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = zero16();
    // ---- prologue: K tiles 0 and 1 -> slots 0..7; A0(0), B(0) read
#pragma unroll
    for (int kk = 0; kk < 2; kk++) {
#pragma unroll
        for (int u = 0; u < 4; u++) dma_unit(u, 4 * kk + u);
        advance();
    }
    wait_vm8<10>();
#pragma unroll
    for (int ks = 0; ks < 4; ks++)
#pragma unroll
        for (int tl = 0; tl < 2; tl++) {
            FAx.f[ks][tl] = ldfrag(pa[0][ks], tl * 32 * kRowB8);
            FBx.f[ks][tl] = ldfrag(pb[0][ks], kSlotB8 + tl * 32 * kRowB8);
        }
    constexpr int kStores = 16;   // epilogue stores per thread
    bool stores_behind = false;

    for (int t = 0; t < my_tiles; t++) {
        int row0, col0;
        tile_origin(blockIdx.x + t * gridDim.x, row0, col0);
        for (int kt0 = 0; kt0 < nk; kt0 += 16) {
            static_for8<16>([&](auto KT) {
                constexpr int kt = decltype(KT)::value, e = kt & 1;
                // ---- P1: a0 x b; read A1(g); request A0, Blo of K tile g + 2 (the slots of A0(g), Blo(g), read in P2(g - 1))
                if (kt <= 1) {
                    if (stores_behind) wait_vm8<8 + kStores>();
                    else wait_vm8<8>();
                } else {
                    wait_vm8<8>();
                }
                __builtin_amdgcn_sched_barrier(0);
                dma_unit(0, 4 * e + 0);
                dma_unit(1, 4 * e + 1);
                phase_mma(0, 4 * e + 3, -1);
#pragma unroll
                for (int q = 0; q < 4; q++) {   // per k-step: 4 MFMAs, then the 2 reads that reuse its dead A fragments, a DMA
                    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                // ---- P2: a1 x b; read A0(g + 1), B(g + 1); request Bhi, A1 of K tile g + 2 (slots of Bhi(g): read in P2(g - 1); A1(g): P1(g))
                if (kt == 0) {
                    if (stores_behind) wait_vm8<6 + kStores>();
                    else wait_vm8<6>();
                } else {
                    wait_vm8<6>();
                }
                __builtin_amdgcn_sched_barrier(0);
                dma_unit(2, 4 * e + 2);
                dma_unit(3, 4 * e + 3);
                advance();
                phase_mma(1, 4 * (1 - e) + 0, 1 - e);
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
                    __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            });
            stores_behind = false;
        }
        // ---- epilogue: 32 rows x 64 columns at a time through the wave's staging tile ([32][128 B], 16-byte segments XOR (row >> 1) & 7)
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
        stores_behind = true;
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
