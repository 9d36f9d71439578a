// rwkvtts_amd/csrc/gemv32.hip -- weight-streaming product for the decode step on gfx950:  Y[M,N] = X[M,K] . W[N,K]^T (+ bias)
// with M <= 32 rows (the decode batch: BASELINE.json configs[4] is B = 32, one token per sequence per step).
//
// Reference call sites: every nn.Linear of the per-token path (model/llm/rwkv_s2s_single_ffn.py:482-506,545-549 -- r,k,v,o
// projections, the w/a/v/g low-rank pairs, the channel-mix key/value) and the head.  At M = 32 the product is pure weight
// streaming (2 N K bytes for 64 N K flop); the BLAS library picks a 32x32 macro tile with a serial K loop and needs ~11 us
// for a 1024 x 1024 weight (0.19 TB/s, rocprofv3 on tools/bench_decode.py) -- 24 x 9 such calls are 70 % of a decode step.
//
// Here one workgroup owns 32 output columns (32 rows of W); its 4 waves split K, each wave pulls its W and X fragments
// straight from global memory into MFMA operand registers (all loads of a wave are issued before the first MFMA), the
// four partial 32x32 tiles meet in LDS, and wave 0 adds the bias and stores bf16.  X (64 KB at K = 1024) is re-read by
// every workgroup from L2.  D[m][n]: m = output column inside the tile (A operand = W rows), n = batch row (B operand = X).
//
// Measured (rocprofv3, 0.4B, B = 32): 9.9 us average per call -- on par with the library, not the ~3 us the byte count
// suggests: every lane of a fragment load reads a different 2 KB-strided row (32 lines per instruction) and each call
// still pays ~5 us of launch/drain even inside a hipGraph.  The decode step went 4.4 -> 4.0 ms; the real lever is fewer,
// fused kernels per layer (655 launches per step).
#include "chunk_common.h"

namespace rwkv7 {

namespace {

template <int KSTEPS>  // 16-wide k-steps per wave and batch
__device__ __forceinline__ void gemv_part(f32x16 &acc, const uint16_t *wp, const uint16_t *xp) {
    bf16x8 a[KSTEPS], b[KSTEPS];
#pragma unroll
    for (int i = 0; i < KSTEPS; i++) {
        a[i] = *reinterpret_cast<const bf16x8 *>(wp + 16 * i);
        b[i] = *reinterpret_cast<const bf16x8 *>(xp + 16 * i);
    }
#pragma unroll
    for (int i = 0; i < KSTEPS; i++) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[i], acc, 0, 0, 0);
}

__global__ __launch_bounds__(256) void gemv32_kernel(int M, int N, int K, const uint16_t *__restrict__ X,
                                                     const uint16_t *__restrict__ W, const uint16_t *__restrict__ bias,
                                                     uint16_t *__restrict__ Y) {
    __shared__ __attribute__((aligned(16))) float part[3][64][17];  // waves 1..3 -> wave 0 (padded: conflict-free)
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int col0 = blockIdx.x * 32;
    const int mrow = min(col0 + (lane & 31), N - 1);   // W row of this lane (clamped in the last tile)
    const int nrow = min(lane & 31, M - 1);            // X row of this lane (clamped when M < 32)
    const int kw = K / 4;                              // K range of this wave; K % 64 == 0
    const uint16_t *wp = W + (long)mrow * K + wave * kw + (lane >> 5) * 8;
    const uint16_t *xp = X + (long)nrow * K + wave * kw + (lane >> 5) * 8;
    f32x16 acc = zero16();
    int k = 0;
    for (; k + 256 <= kw; k += 256) gemv_part<16>(acc, wp + k, xp + k);
    for (; k + 64 <= kw; k += 64) gemv_part<4>(acc, wp + k, xp + k);
    for (; k + 16 <= kw; k += 16) gemv_part<1>(acc, wp + k, xp + k);
    if (wave > 0) {
#pragma unroll
        for (int r = 0; r < 16; r++) part[wave - 1][lane][r] = acc[r];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int r = 0; r < 16; r++) acc[r] += part[0][lane][r] + part[1][lane][r] + part[2][lane][r];
        const int n = lane & 31;
        if (n < M) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int c = col0 + 8 * j + 4 * (lane >> 5);  // 4 consecutive output columns
                float v[4];
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    v[i] = acc[4 * j + i];
                    if (bias && c + i < N) v[i] += bf2f(bias[c + i]);
                }
                uint16_t *yp = Y + (long)n * N + c;
                if (c + 3 < N && (N & 3) == 0) {
                    *reinterpret_cast<uint2 *>(yp) = make_uint2(cvt_pk(v[0], v[1]), cvt_pk(v[2], v[3]));
                } else {
#pragma unroll
                    for (int i = 0; i < 4; i++)
                        if (c + i < N) yp[i] = (uint16_t)cvt_pk(v[i], 0.f);
                }
            }
        }
    }
}

}  // namespace

int gemv32_bf16(int M, int N, int K, const void *x, const void *w, const void *bias, void *y, hipStream_t st) {
    (void)hipGetLastError();
    hipLaunchKernelGGL(gemv32_kernel, dim3((N + 31) / 32), dim3(256), 0, st, M, N, K, (const uint16_t *)x, (const uint16_t *)w,
                       (const uint16_t *)bias, (uint16_t *)y);
    return (int)hipGetLastError();
}

}  // namespace rwkv7
