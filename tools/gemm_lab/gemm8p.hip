// tools/gemm_lab/gemm8p.hip -- LAB: the "256^2 8-phase" bf16 GEMM structure of cdna_hip_programming.md (section 5, T3 + T4 + T5 + T2),
// written from its description, to measure what that structure gives on THIS path's shapes (C[M][N] = A[M][K] . W[N][K]^T with
// K = 1024 / 4096) against the library and csrc/gemm_nt4.hip.  Not linked into librwkv7_hip.so.
//
//   tile 256 x 256 x 64, 8 waves (2 in M x 4 in N), wave tile 128 x 64 = 8 x 4 MFMA tiles of 16 x 16 (v_mfma_f32_16x16x32_bf16),
//   128 accumulator registers; LDS 128 KB = 2 K-tile buffers x 4 half tiles (A0, A1, B0, B1: 128 rows x 64 k = 16 KB each, the
//   rows of every wave's m-half / n-half), XOR-swizzled on the SOURCE side of the LDS-DMA; per K tile four phases = the four
//   64 x 32 quadrants of the wave tile in the order (m0,n0) (m1,n0) (m1,n1) (m0,n1): phase = { ds_read of the operand half that
//   changes (B0 + A0 | A1 | B1 | A0), one half tile of LDS-DMA two phases behind that slot's last read, s_barrier, lgkmcnt(0),
//   setprio 1, 16 MFMA, setprio 0, s_barrier }; the two wave groups (M halves) run one barrier apart, so one group's MFMAs sit
//   beside the other's LDS reads and DMA issue; vmcnt(4) once per K tile (never 0 in the loop).
#include <hip/hip_runtime.h>
#include <stdint.h>

using bf16x8 = __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16;
using f32x4 = __attribute__((__vector_size__(4 * sizeof(float)))) float;
typedef __bf16 bf2_t __attribute__((ext_vector_type(2)));
typedef float f2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t cvt_pk(float a, float b) {
    const f2_t v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf2_t));
}

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int kHalf = 128 * BK * 2;          // 16 KB
constexpr int kBuf = 4 * kHalf;              // A0 A1 B0 B1
constexpr int kLds = 2 * kBuf;               // 128 KB
enum { HA0 = 0, HA1 = 1, HB0 = 2, HB1 = 3 };

__device__ __forceinline__ void glds16(const void *gsrc, uint32_t lds_dst) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}
__device__ __forceinline__ int swz(int row) { return (row >> 1) & 7; }

template <int PRIO, int STAGGER>
__global__ __launch_bounds__(512) void gemm8p_kernel(int M, int N, int K, const uint16_t *__restrict__ A, const uint16_t *__restrict__ W,
                                                     uint16_t *__restrict__ C) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;          // wave tile: rows wr * 128 .., columns wc * 64 ..
    const int nbn = N / BN, nbm = M / BM, ntiles = nbn * nbm, nk = K / BK;
    // XCD-aware tile order: ids b, b + 8, ... share an L2; give each XCD a contiguous run of tiles (row panels: the A rows are re-used)
    auto tile_origin = [&](int id, int &row0, int &col0) {
        int t = id;
        if (ntiles % 8 == 0) t = (id % 8) * (ntiles / 8) + id / 8;
        row0 = (t / nbn) * BM;
        col0 = (t % nbn) * BN;
    };
    // LDS-DMA of a half tile: 128 rows x 128 B = 16 pieces of 1 KB (8 rows); wave w issues pieces w and w + 8: slot rows 8 p + lane / 8,
    // 16-byte segment lane & 7 (swizzled on the source side).  Slot row q of an A half mh: tile row (q < 64 ? q : q + 64) + 64 mh
    // (the m-half rows of the waves with wr = 0, then wr = 1); of a B half nh: tile column 64 (q / 32) + 32 nh + q % 32.
    uint32_t offA[2][2], offB[2][2];
#pragma unroll
    for (int hf = 0; hf < 2; hf++)
#pragma unroll
        for (int p2 = 0; p2 < 2; p2++) {
            const int q = (wave + 8 * p2) * 8 + (lane >> 3);
            const int seg = ((lane & 7) ^ swz(q)) << 3;
            const int arow = (q < 64 ? q : q + 64) + 64 * hf;
            const int brow = 64 * (q >> 5) + 32 * hf + (q & 31);
            offA[hf][p2] = (uint32_t)(arow * K + seg) * 2u;
            offB[hf][p2] = (uint32_t)(brow * K + seg) * 2u;
        }
    const uint32_t lds0 = (uint32_t)(uintptr_t)lds;
    // fragment read addresses: A fragment of m-tile mt (0..3) of the wave's m-half: slot row wr * 64 + 16 mt + (lane & 15), k-step ks:
    // segment 4 ks + (lane >> 4) ... (16 B = 8 k); B fragment of n-tile nt (0..1): slot row wc * 32 + 16 nt + (lane & 15)
    auto frag = [&](int buf, int half, int row, int seg) -> bf16x8 {
        return *reinterpret_cast<const bf16x8 *>(lds + buf * kBuf + half * kHalf + row * 128 + ((seg ^ swz(row)) << 4));
    };
    const int ar = wr * 64 + (lane & 15), br = wc * 32 + (lane & 15), sg = lane >> 4;

    for (int tileid = blockIdx.x; tileid < ntiles; tileid += gridDim.x) {
        int row0, col0;
        tile_origin(tileid, row0, col0);
        const char *Ab = reinterpret_cast<const char *>(A) + (long)row0 * K * 2;
        const char *Wb = reinterpret_cast<const char *>(W) + (long)col0 * K * 2;
        auto stage = [&](int kt, int half, int buf) {   // one half tile of K tile kt (clamped) into buffer buf
            const int k2 = (kt < nk ? kt : nk - 1) * BK * 2;
            const bool isB = half >= 2;
            const int hf = half & 1;
            const char *base = (isB ? Wb : Ab) + k2;
#pragma unroll
            for (int p2 = 0; p2 < 2; p2++)
                glds16(base + (isB ? offB[hf][p2] : offA[hf][p2]), lds0 + buf * kBuf + half * kHalf + (wave + 8 * p2) * 1024);
        };
        f32x4 acc[4][8];   // [n tile 0..3][m tile 0..7]: D rows = n (4 consecutive per lane), columns = m (lane & 15)
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 8; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        // prologue: K tiles 0 and 1 complete
        __syncthreads();   // the previous tile's LDS reads are done
#pragma unroll
        for (int h = 0; h < 4; h++) stage(0, h, 0);
#pragma unroll
        for (int h = 0; h < 4; h++) stage(1, h, 1);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (STAGGER && wr == 1) __builtin_amdgcn_s_barrier();
        bf16x8 fa[4][2], fb[2][2];   // [tile][k-step]
        auto readA = [&](int buf, int mh) {
#pragma unroll
            for (int mt = 0; mt < 4; mt++)
#pragma unroll
                for (int ks = 0; ks < 2; ks++) fa[mt][ks] = frag(buf, mh ? HA1 : HA0, ar + 16 * mt, 4 * ks + sg);
        };
        auto readB = [&](int buf, int nh) {
#pragma unroll
            for (int nt = 0; nt < 2; nt++)
#pragma unroll
                for (int ks = 0; ks < 2; ks++) fb[nt][ks] = frag(buf, nh ? HB1 : HB0, br + 16 * nt, 4 * ks + sg);
        };
        auto mma = [&](int mh, int nh) {
            if (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int ks = 0; ks < 2; ks++)
#pragma unroll
                for (int nt = 0; nt < 2; nt++)
#pragma unroll
                    for (int mt = 0; mt < 4; mt++)
                        acc[2 * nh + nt][4 * mh + mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[nt][ks], fa[mt][ks], acc[2 * nh + nt][4 * mh + mt], 0, 0, 0);
            if (PRIO) __builtin_amdgcn_s_setprio(0);
        };
        auto bar_reads = [&]() {
            __builtin_amdgcn_s_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        };
        for (int kt = 0; kt < nk; kt++) {
            const int buf = kt & 1;
            // phase 1: quadrant (m0, n0): reads B0 + A0; stage B1 of K tile kt + 1 ... (slot last read in phase 3 of K tile kt - 1)
            readB(buf, 0);
            __builtin_amdgcn_sched_barrier(0);
            readA(buf, 0);
            if (kt >= 1) stage(kt + 1, HB1, buf ^ 1);
            bar_reads();
            mma(0, 0);
            __builtin_amdgcn_s_barrier();
            // phase 2: (m1, n0): reads A1; stage A0 of K tile kt + 1 (last read in phase 4 of kt - 1)
            readA(buf, 1);
            if (kt >= 1) stage(kt + 1, HA0, buf ^ 1);
            bar_reads();
            mma(1, 0);
            __builtin_amdgcn_s_barrier();
            // phase 3: (m1, n1): reads B1; stage B0 of K tile kt + 2 (last read in phase 1 of kt)
            readB(buf, 1);
            stage(kt + 2, HB0, buf);
            bar_reads();
            mma(1, 1);
            __builtin_amdgcn_s_barrier();
            // phase 4: (m0, n1): reads A0 again; stage A1 of K tile kt + 2 (last read in phase 2 of kt); K tile kt + 1 must have landed
            readA(buf, 0);
            stage(kt + 2, HA1, buf);
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            bar_reads();
            mma(0, 1);
            __builtin_amdgcn_s_barrier();
        }
        if (STAGGER && wr == 0) __builtin_amdgcn_s_barrier();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // epilogue: lane holds 4 consecutive columns of one row per MFMA tile: 8-byte stores (plain)
        uint16_t *cb = C + (long)(row0 + wr * 128) * N + col0 + wc * 64;
#pragma unroll
        for (int nt4 = 0; nt4 < 4; nt4++)
#pragma unroll
            for (int mt8 = 0; mt8 < 8; mt8++) {
                // acc[nt4][mt8]: n tile = (nt4 >> 1) half, (nt4 & 1) tile: columns 32 (nt4 >> 1) + 16 (nt4 & 1) + 4 (lane >> 4) + i; row 64 (mt8 >> 2) + 16 (mt8 & 3) + (lane & 15)
                const int r = 64 * (mt8 >> 2) + 16 * (mt8 & 3) + (lane & 15);
                const int c = 32 * (nt4 >> 1) + 16 * (nt4 & 1) + 4 * (lane >> 4);
                const f32x4 v = acc[nt4][mt8];
                *reinterpret_cast<uint2 *>(cb + (long)r * N + c) = make_uint2(cvt_pk(v[0], v[1]), cvt_pk(v[2], v[3]));
            }
    }
}

extern "C" int lab_gemm(int M, int N, int K, const void *A, const void *W, void *C, int epi, void *st) {
    if (M % BM || N % BN || K % BK || K / BK < 2) return -4;
    static bool attr = false;
#ifndef G8_PRIO
#define G8_PRIO 1
#endif
#ifndef G8_STAGGER
#define G8_STAGGER 1
#endif
    auto kern = &gemm8p_kernel<G8_PRIO, G8_STAGGER>;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
        if (e != hipSuccess) return (int)e;
        attr = true;
    }
    const int ntiles = (M / BM) * (N / BN);
    kern<<<dim3(ntiles < 256 ? ntiles : 256), dim3(512), kLds, (hipStream_t)st>>>(M, N, K, (const uint16_t *)A, (const uint16_t *)W, (uint16_t *)C);
    return (int)hipGetLastError();
}
