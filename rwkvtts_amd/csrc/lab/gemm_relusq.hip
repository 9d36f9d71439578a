// rwkvtts_amd/csrc/gemm_relusq.hip -- bf16 GEMM  C[M][N] = epi(A[M][K] . W[N][K]^T)  with the channel-mix activation as epilogue
// (epi = relu(.)^2, rwkv_s2s_single_ffn.py:228), hand-written for gfx950: a measured experiment against
// the library GEMM + rwkv7_relusq_fwd pair that the channel-mix key projection runs today.
//
// 256 x 256 x 64 tiles, 8 waves (2 x 4: each 128 rows x 64 columns = 4 x 2 MFMA tiles of 32 x 32), fp32 accumulators (128 VGPRs),
// both operands K-contiguous ("NT"), staged through LDS by LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write
// pass), two LDS buffers of 64 KB, one barrier per K tile.  The LDS image of an operand tile is [256 rows][128 B] with the eight
// 16-byte segments of a row XOR-swizzled by (row >> 1) & 7 -- applied to the SOURCE address of the DMA (its destination is
// wave-uniform base + lane * 16) and to the fragment reads -- so that the ds_read_b128 of 32 consecutive rows is conflict free.
// The product is formed transposed (X = W rows, Y = A rows: lane = output row, 4 consecutive output columns per register group),
// so the epilogue stores 8 bytes per lane and a row's 64 columns of a wave leave as full 128-byte lines.
// Measured (tools/bench_gemm_relusq.py, 32768 x 4096 x 1024, same process as the library): first cut 811 TFLOP/s; 4 x 8 tile
// patches per XCD 872; persistent over the tiles (the 128 KB store tail of a tile drains under the next tile's K loop) + the next
// tile's DMA instructions spread over the k-steps 925; fragments one k-step ahead in registers: 0.905 of the library's rate on
// the same box (0.856 before).  K tile 32 with four buffers (three tiles in flight, counted vmcnt, raw barrier): slower (0.83) --
// it is not the DMA latency that limits the loop.  PMC: same HBM traffic and L2 misses as the library's kernel, 1.6x its L2
// requests.  Outcome of the experiment: fused.py (FUSED_KEY_RELUSQ) -- ties the library pair inside the training step, not adopted.
#include "../chunk_common.h"
#include "../launch_attr.h"

namespace rwkv7 {
namespace {
constexpr int GBM = 256, GBN = 256;

// K tile BK (64 or 32 elements = 8 or 4 sixteen-byte segments per row), NBUF LDS buffers of (A tile + W tile)
template <int BK, int NBUF>
struct GemmCfg {
    static constexpr int S = BK / 8;                  // segments per row
    static constexpr int kRowB = BK * 2;              // bytes per tile row
    static constexpr int kTileB = 256 * kRowB;        // one operand tile
    static constexpr int R = 64 / S;                  // rows per DMA instruction (64 lanes x 16 B)
    static constexpr int G = 256 / R / 8;             // DMA instructions per thread and operand tile (8 waves)
    static constexpr size_t lds = (size_t)NBUF * 2 * kTileB;
    // swizzle of the 16-byte segments of a row: ds_read_b128 serves lanes in groups of 16 that must cover the 16 bank quads
    static __device__ __forceinline__ int swz(int row) { return BK == 64 ? (row >> 1) & 7 : (row >> 2) & 3; }
};

// one operand tile (256 rows x BK) -> LDS by LDS-DMA: destination = wave-uniform base + lane * 16, swizzle on the SOURCE address
template <typename C>
__device__ __forceinline__ void stage_tile(const uint16_t *__restrict__ g, long ld, int row0, int k0, char *lds, int wave, int lane) {
    using gptr = const __attribute__((address_space(1))) void *;
    using lptr = __attribute__((address_space(3))) void *;
#pragma unroll
    for (int r = 0; r < C::G; r++) {
        const int rbase = (8 * r + wave) * C::R;
        const int row = rbase + lane / C::S;
        const int seg = (lane % C::S) ^ C::swz(row);
        const uint16_t *src = g + (long)(row0 + row) * ld + k0 + seg * 8;
        __builtin_amdgcn_global_load_lds((gptr)(src), (lptr)(lds + rbase * C::kRowB), 16, 0, 0);
    }
}
// piece p of the (A, W) tile pair: p < G -> A, else W; one DMA instruction
template <typename C>
__device__ __forceinline__ void stage_piece(int p, const uint16_t *__restrict__ A, const uint16_t *__restrict__ W, long ld, int row0, int col0,
                                            int k0, char *buf, int wave, int lane) {
    using gptr = const __attribute__((address_space(1))) void *;
    using lptr = __attribute__((address_space(3))) void *;
    const bool isW = p >= C::G;
    const int r = isW ? p - C::G : p;
    const int rbase = (8 * r + wave) * C::R;
    const int row = rbase + lane / C::S;
    const int seg = (lane % C::S) ^ C::swz(row);
    const uint16_t *src = (isW ? W + (long)(col0 + row) * ld : A + (long)(row0 + row) * ld) + k0 + seg * 8;
    __builtin_amdgcn_global_load_lds((gptr)(src), (lptr)(buf + (isW ? C::kTileB : 0) + rbase * C::kRowB), 16, 0, 0);
}
template <typename C>
__device__ __forceinline__ bf16x8 frag(const char *tile, int row, int seg) {
    return *reinterpret_cast<const bf16x8 *>(tile + row * C::kRowB + ((seg ^ C::swz(row)) << 4));
}
template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
}  // namespace

// EPI: 0 plain; 1 relu(.)^2 (channel-mix key, forward); 2 (round 4) C = bf16(A W^T) * 2 relu(aux) -- the backward of the activation
// as the epilogue of the value projection's input-gradient GEMM (ds = dy W_value, dh = ds * 2 relu(h), rwkv_s2s_single_ffn.py:228
// differentiated): ds never reaches HBM and rwkv7_relusq_bwd (805 MB, 140 us per layer) is not launched.  aux = h, [M][N] like C.
template <int EPI, int BK, int NBUF>
__global__ __launch_bounds__(512) void gemm_nt_bf16_kernel(int M, int N, int K, const uint16_t *__restrict__ A, const uint16_t *__restrict__ W,
                                                           uint16_t *__restrict__ C, const uint16_t *__restrict__ aux) {
    using Cfg = GemmCfg<BK, NBUF>;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    // Persistent: gridDim.x workgroups (one per CU) walk the tiles id = blockIdx.x, + gridDim.x, ...: the 128 KB of output stores of a
    // tile drain while the next tile's K loop runs (with one tile per workgroup and one workgroup per CU the store tail of every tile
    // was exposed: 8 x ~5 us of a 315 us kernel).
    // id -> tile.  Ids b, b + 8, ... run on one XCD (one 4 MB L2), 32 at a time: those 32 form a 4 (row panels) x 8 (column tiles)
    // patch, so that every k-slice of A is fetched once for 8 workgroups and every slice of W once for 4 while they walk K together
    // (a 2 x 16 patch streamed all of W through every L2 once per pair of row panels: 811 instead of 872 TFLOP/s)
    const int nbn = N / GBN, nbm = M / GBM, ntiles = nbn * nbm;
    // (round 4) narrow outputs (N = 1024: four column tiles): the patch is 8 row panels x 4 column tiles, so that the four workgroups
    // that share an A panel still sit on one XCD (id % nbn as the column put them on four different L2s: A fetched four times)
    const int pc = nbn % 8 == 0 ? 8 : 4, pr = 32 / pc;      // patch: pr row panels x pc column tiles
    const bool patched = (gridDim.x & 7) == 0 && nbm % (8 * pr) == 0 && nbn % pc == 0;
    const int nk = K / BK;
    const int rl = lane & 31, h = lane >> 5;
    constexpr int NP = 2 * Cfg::G;              // DMA instructions per thread and K tile
    constexpr int KS = BK / 16;                 // MFMA k-steps per K tile
    static_assert(NP % KS == 0, "pieces are spread evenly over the k-steps");
    for (int id = blockIdx.x; id < ntiles; id += gridDim.x) {
        int bm, bn;
        if (patched) {
            const int xcd = id & 7, j = id >> 3;
            const int nround_n = nbn / pc;
            const int r = j / 32, i = j % 32;
            bn = pc * (r % nround_n) + (i % pc);
            bm = xcd + 8 * (pr * (r / nround_n) + (i / pc));
        } else {
            bn = id % nbn;
            bm = id / nbn;
        }
        const int row0 = bm * GBM, col0 = bn * GBN;
        f32x16 acc[2][4];   // [n tile of the wave][m tile of the wave]: D[m' = column][n' = row]
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) acc[i][j] = zero16();
        // every wave is past its reads of the previous tile's last buffers before they are refilled
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
        for (int p = 0; p < NBUF - 1; p++)
            if (p < nk) {
#pragma unroll
                for (int q = 0; q < NP; q++) stage_piece<Cfg>(q, A, W, K, row0, col0, p * BK, lds + (p % NBUF) * 2 * Cfg::kTileB, wave, lane);
            }
        for (int kt = 0; kt < nk; kt++) {
            const char *bufA = lds + (kt % NBUF) * 2 * Cfg::kTileB, *bufW = bufA + Cfg::kTileB;
            // this wave's pieces of tile kt have landed: the DMAs of the (up to NBUF - 2) younger tiles stay in flight
            if (NBUF >= 4 && kt + 2 < nk) wait_vm<2 * NP>();
            else if (NBUF >= 3 && kt + 1 < nk) wait_vm<NP>();
            else wait_vm<0>();
            // raw barrier (no vmcnt drain): everybody's pieces have landed; everybody is done reading tile kt - 1, whose buffer the
            // next DMAs overwrite
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            const int nt = kt + NBUF - 1;
            const bool more = nt < nk;
            char *nbuf = lds + (nt % NBUF) * 2 * Cfg::kTileB;
            // fragments one k-step ahead in registers: the ds_reads of step ks + 1 are issued BEFORE the MFMAs of step ks (left to
            // itself hipcc issues them behind six of the eight MFMAs, and every k-step then opens with an LDS latency)
            bf16x8 fw[2][2], fa[2][4];
            auto load_frags = [&](int ks, int slot) {
#pragma unroll
                for (int i = 0; i < 2; i++) fw[slot][i] = frag<Cfg>(bufW, wn * 64 + i * 32 + rl, 2 * ks + h);
#pragma unroll
                for (int j = 0; j < 4; j++) fa[slot][j] = frag<Cfg>(bufA, wm * 128 + j * 32 + rl, 2 * ks + h);
            };
            load_frags(0, 0);
#pragma unroll
            for (int ks = 0; ks < KS; ks++) {
                if (ks + 1 < KS) load_frags(ks + 1, (ks + 1) & 1);
                // the next tile's DMA instructions, a few per k-step, between the MFMAs instead of as one burst behind the barrier
                if (more) {
#pragma unroll
                    for (int q = 0; q < NP / KS; q++) stage_piece<Cfg>(ks * (NP / KS) + q, A, W, K, row0, col0, nt * BK, nbuf, wave, lane);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 2; i++)
#pragma unroll
                    for (int j = 0; j < 4; j++)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[ks & 1][i], fa[ks & 1][j], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // epilogue: lane = output row inside the m tile, registers 4 g .. 4 g + 3 = columns 8 g + 4 h + (0..3) of the n tile
        uint2 ax[2][8];   // EPI 2: the aux values of row block j, requested one block ahead
        auto load_aux = [&](int j, uint2 (&dst)[8]) {
            const uint16_t *arow = aux + (long)(row0 + wm * 128 + j * 32 + rl) * N + col0 + wn * 64 + 4 * h;
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int g = 0; g < 4; g++) dst[i * 4 + g] = *reinterpret_cast<const uint2 *>(arow + i * 32 + 8 * g);
        };
        if (EPI == 2) load_aux(0, ax[0]);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if (EPI == 2 && j + 1 < 4) load_aux(j + 1, ax[(j + 1) & 1]);
            uint16_t *crow = C + (long)(row0 + wm * 128 + j * 32 + rl) * N + col0 + wn * 64 + 4 * h;
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    float x[4];
                    float hx[4] = {0.f, 0.f, 0.f, 0.f};
                    if (EPI == 2) {
                        const uint2 a2 = ax[j & 1][i * 4 + g];
                        hx[0] = __uint_as_float(a2.x << 16); hx[1] = __uint_as_float(a2.x & 0xffff0000u);
                        hx[2] = __uint_as_float(a2.y << 16); hx[3] = __uint_as_float(a2.y & 0xffff0000u);
                    }
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        float v = acc[i][j][4 * g + e];
                        if (EPI == 1) {          // relu(bf16(x))^2, rounded again: what the two separate kernels produce
                            v = bf2f(f2bf(v));
                            v = v > 0.f ? v * v : 0.f;
                        }
                        if (EPI == 2) {          // bf16(ds) * 2 relu(h): what the library GEMM + rwkv7_relusq_bwd produce
                            v = bf2f(f2bf(v));
                            v = hx[e] > 0.f ? 2.f * hx[e] * v : 0.f;
                        }
                        x[e] = v;
                    }
                    *reinterpret_cast<uint2 *>(crow + i * 32 + 8 * g) = make_uint2(cvt_pk(x[0], x[1]), cvt_pk(x[2], x[3]));
                }
        }
    }
}

namespace {
template <int EPI, int BK, int NBUF>
int launch_gemm(int M, int N, int K, const void *A, const void *W, void *C, hipStream_t st, const void *aux = nullptr) {
    static DynLdsOnce lds_once;
    auto kern = &gemm_nt_bf16_kernel<EPI, BK, NBUF>;
    constexpr size_t lds_bytes = GemmCfg<BK, NBUF>::lds;
    if (hipError_t e = lds_once.ensure(reinterpret_cast<const void *>(kern), (int)lds_bytes); e != hipSuccess) return (int)e;
    (void)hipGetLastError();
    const int ntiles = (M / GBM) * (N / GBN);
    kern<<<dim3(ntiles < 256 ? ntiles : 256), dim3(512), lds_bytes, st>>>(M, N, K, (const uint16_t *)A, (const uint16_t *)W, (uint16_t *)C,
                                                                          (const uint16_t *)aux);
    return (int)hipGetLastError();
}
}  // namespace

// variant: 0 = BK 64, two LDS buffers (one tile ahead); 1 = BK 32, four buffers (three tiles ahead, counted vmcnt)
int gemm_nt_bf16_variant(int M, int N, int K, const void *A, const void *W, void *C, int epilogue, int variant, hipStream_t st) {
    if (variant == 0)
        return epilogue ? launch_gemm<1, 64, 2>(M, N, K, A, W, C, st) : launch_gemm<0, 64, 2>(M, N, K, A, W, C, st);
    return epilogue ? launch_gemm<1, 32, 4>(M, N, K, A, W, C, st) : launch_gemm<0, 32, 4>(M, N, K, A, W, C, st);
}

int gemm_nt_relusq_bwd_bf16(int M, int N, int K, const void *A, const void *W, const void *aux, void *C, hipStream_t st) {
    return launch_gemm<2, 64, 2>(M, N, K, A, W, C, st, aux);
}

int gemm_nt_bf16(int M, int N, int K, const void *A, const void *W, void *C, int epilogue, hipStream_t st) {
    return gemm_nt_bf16_variant(M, N, K, A, W, C, epilogue, 0, st);   // K tile 64, two buffers: 297 us against 313 for variant 1
}

}  // namespace rwkv7
