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

// Low-rank pair in one launch for the decode batch: Y[M,N] = act(X[M,K] . W1[R,K]^T) . W2[N,R]^T (+ bias), R in {32,64,128}.
// Every workgroup (32 output columns) recomputes the small [M,R] intermediate -- W1 is at most 256 KB and comes from L2 --
// so the three launches of nn.Sequential(Linear, act, Linear) become one.  The intermediate is rounded to bf16 like the
// tensor it replaces.
template <int ACT>
__device__ __forceinline__ float lora32_act(float x) {
    if constexpr (ACT == 1) return 1.f - 2.f / (__expf(2.f * x) + 1.f);   // tanh
    if constexpr (ACT == 2) return 1.f / (1.f + __expf(-x));              // sigmoid
    return x;
}

template <int ACT>
__global__ __launch_bounds__(256) void lora32_kernel(int M, int N, int K, int R, const uint16_t *__restrict__ X,
                                                     const uint16_t *__restrict__ W1, const uint16_t *__restrict__ W2,
                                                     const uint16_t *__restrict__ bias, uint16_t *__restrict__ Y) {
    __shared__ __attribute__((aligned(16))) float part[3][64][17];
    __shared__ __attribute__((aligned(16))) uint16_t Ap[32 * (128 + 8)];   // act(X W1^T) as a bf16 plane [n][R + 8]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int nrow = min(lane & 31, M - 1);
    const int kw = K / 4, ldA = R + 8;
    const uint16_t *xp = X + (long)nrow * K + wave * kw + (lane >> 5) * 8;
    for (int rt = 0; rt < R / 32; rt++) {
        const uint16_t *wp = W1 + (long)(rt * 32 + (lane & 31)) * K + wave * kw + (lane >> 5) * 8;
        f32x16 acc = zero16();  // D[m = r][n = batch row]
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
            for (int j = 0; j < 4; j++) {
                float v[4];
#pragma unroll
                for (int i = 0; i < 4; i++)
                    v[i] = lora32_act<ACT>(acc[4 * j + i] + part[0][lane][4 * j + i] + part[1][lane][4 * j + i] + part[2][lane][4 * j + i]);
                *reinterpret_cast<uint2 *>(&Ap[(lane & 31) * ldA + rt * 32 + 8 * j + 4 * (lane >> 5)]) =
                    make_uint2(cvt_pk(v[0], v[1]), cvt_pk(v[2], v[3]));
            }
        }
        __syncthreads();
    }
    if (wave == 0) {
        const int col0 = blockIdx.x * 32;
        const int mrow = min(col0 + (lane & 31), N - 1);
        const uint16_t *w2 = W2 + (long)mrow * R + (lane >> 5) * 8;
        const uint16_t *ap = Ap + (lane & 31) * ldA + (lane >> 5) * 8;
        f32x16 acc = zero16();  // D[m = output column][n = batch row]
        for (int k = 0; k < R; k += 16) {
            const bf16x8 a = *reinterpret_cast<const bf16x8 *>(w2 + k), b = *reinterpret_cast<const bf16x8 *>(ap + k);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
        }
        const int n = lane & 31;
        if (n < M) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int c = col0 + 8 * j + 4 * (lane >> 5);
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

int lora32_bf16(int M, int N, int K, int R, int act, const void *x, const void *w1, const void *w2, const void *bias, void *y,
                hipStream_t st) {
    (void)hipGetLastError();
    const dim3 grid((N + 31) / 32), block(256);
    const uint16_t *X = (const uint16_t *)x, *W1 = (const uint16_t *)w1, *W2 = (const uint16_t *)w2, *Bi = (const uint16_t *)bias;
    if (act == 1) hipLaunchKernelGGL(lora32_kernel<1>, grid, block, 0, st, M, N, K, R, X, W1, W2, Bi, (uint16_t *)y);
    else if (act == 2) hipLaunchKernelGGL(lora32_kernel<2>, grid, block, 0, st, M, N, K, R, X, W1, W2, Bi, (uint16_t *)y);
    else hipLaunchKernelGGL(lora32_kernel<0>, grid, block, 0, st, M, N, K, R, X, W1, W2, Bi, (uint16_t *)y);
    return (int)hipGetLastError();
}

int gemv32_bf16(int M, int N, int K, const void *x, const void *w, const void *bias, void *y, hipStream_t st) {
    (void)hipGetLastError();
    hipLaunchKernelGGL(gemv32_kernel, dim3((N + 31) / 32), dim3(256), 0, st, M, N, K, (const uint16_t *)x, (const uint16_t *)w,
                       (const uint16_t *)bias, (uint16_t *)y);
    return (int)hipGetLastError();
}

}  // namespace rwkv7
