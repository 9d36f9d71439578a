// rwkvtts_amd/csrc/lora.hip -- the skinny products of the RWKV-7 low-rank branches on gfx950 MFMA.
//
// Reference maths (model/llm/rwkv_s2s_single_ffn.py:172-181):  w = w0 + tanh(xw @ w1) @ w2,  a = a0 + (xa @ a1) @ a2,
// v-residual gate v0 + (xv @ v1) @ v2,  g = sigmoid(xg @ g1) @ g2,  with rank R in {32, 64, 128} << D = 1024.
// Two of the four products per branch have a [M x D] operand and a [M x R] result (M = B*T = 32768 rows at
// BASELINE configs[1]):
//     down   : A  = act(X  @ W1^T)          X  [M,D], W1 [R,D]  ->  A  [M,R]         (forward)
//     dgradup: dY = (dZ @ W2 ) * act'(A)    dZ [M,D], W2^T [R,D] -> dY [M,R]         (backward)
// They stream 2*M*D bytes (67 MB) for ~0.5 flop/byte: pure HBM work, which the BLAS heuristics serve badly here
// (measured with rocprofv3: 115 us forward, 140-350 us backward per call, i.e. 0.2-0.6 TB/s).  This kernel does
// both with one code path:  Out[M,R] = epilogue(X[M,K] . W[R,K]^T).
//
// Workgroup = 64 rows of X, 4 waves; K is walked in chunks of 64 that are staged through LDS as bf16 planes
// [rows][64 + 8] (the padding makes the 16-byte fragment reads of 32 consecutive rows conflict free) while the next
// chunk's global loads are already in flight (register prefetch).  v_mfma_f32_32x32x16_bf16, fp32 accumulate, tiles
// 32x32: tile t = (row half t & 1, column block t >> 1) goes to wave t & 3.  The activation (and in backward its
// derivative, evaluated from the saved post-activation A) is applied on the accumulator and the result is rounded
// to bf16 once.
#include "chunk_common.h"

namespace rwkv7 {

namespace {

constexpr int kLoraRows = 64;
constexpr int kLoraKC = 64;
constexpr int kLoraLD = kLoraKC + kPad;

enum { ACT_NONE = 0, ACT_TANH = 1, ACT_SIGMOID = 2 };

template <int ACT>
__device__ __forceinline__ float lora_act(float x) {
    if constexpr (ACT == ACT_TANH) return 1.f - 2.f / (__expf(2.f * x) + 1.f);
    if constexpr (ACT == ACT_SIGMOID) return 1.f / (1.f + __expf(-x));
    return x;
}
template <int ACT>
__device__ __forceinline__ float lora_dact(float a) {  // derivative expressed with the activation's OUTPUT a
    if constexpr (ACT == ACT_TANH) return 1.f - a * a;
    if constexpr (ACT == ACT_SIGMOID) return a * (1.f - a);
    return 1.f;
}

// MODE 0: Out = act(X W^T)          MODE 1: Out = (X W^T) * act'(A)
template <int R, int ACT, int MODE>
__global__ __launch_bounds__(256) void lora_skinny_kernel(long M, int K, const uint16_t *__restrict__ X,
                                                          const uint16_t *__restrict__ W,
                                                          const uint16_t *__restrict__ A, uint16_t *__restrict__ Out) {
    constexpr int NT = 2 * (R / 32);            // 32x32 output tiles of this workgroup
    constexpr int NTW = NT >= 4 ? NT / 4 : 1;   // tiles per wave
    constexpr int WP = R * 8 / 256 > 0 ? R * 8 / 256 : 1;  // 16-byte W pieces per thread and chunk
    __shared__ __attribute__((aligned(16))) uint16_t sX[kLoraRows * kLoraLD];
    __shared__ __attribute__((aligned(16))) uint16_t sW[R * kLoraLD];

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const long m0 = (long)blockIdx.x * kLoraRows;

    // staging map: piece p = 16 bytes = 8 bf16 of one row; 8 consecutive lanes cover one 128-byte row chunk
    uint4 rx[2], rw[WP];
    auto issue = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int p = tid + 256 * i, row = p >> 3, c8 = p & 7;
            const long m = m0 + row;
            rx[i] = m < M ? *reinterpret_cast<const uint4 *>(X + m * K + k0 + c8 * 8) : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < WP; i++) {
            const int p = tid + 256 * i, row = p >> 3, c8 = p & 7;
            if (R * 8 >= 256 || p < R * 8) rw[i] = *reinterpret_cast<const uint4 *>(W + (long)row * K + k0 + c8 * 8);
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int p = tid + 256 * i, row = p >> 3, c8 = p & 7;
            *reinterpret_cast<uint4 *>(sX + row * kLoraLD + c8 * 8) = rx[i];
        }
#pragma unroll
        for (int i = 0; i < WP; i++) {
            const int p = tid + 256 * i, row = p >> 3, c8 = p & 7;
            if (R * 8 >= 256 || p < R * 8) *reinterpret_cast<uint4 *>(sW + row * kLoraLD + c8 * 8) = rw[i];
        }
    };

    f32x16 acc[NTW];
#pragma unroll
    for (int i = 0; i < NTW; i++) acc[i] = zero16();
    const bool active = wave < NT;
    const int rt = wave & 1, ct0 = wave >> 1;

    issue(0);
    for (int k0 = 0; k0 < K; k0 += kLoraKC) {
        commit();
        lds_barrier();
        if (k0 + kLoraKC < K) issue(k0 + kLoraKC);
        if (active) {
#pragma unroll
            for (int i = 0; i < NTW; i++)
                mma_tile<kLoraKC>(acc[i], sX + rt * 32 * kLoraLD, kLoraLD, sW + (ct0 + 2 * i) * 32 * kLoraLD, kLoraLD, lane);
        }
        lds_barrier();
    }
    if (!active) return;
#pragma unroll
    for (int i = 0; i < NTW; i++) {
        const int col = (ct0 + 2 * i) * 32 + (lane & 31);
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const long m = m0 + rt * 32 + d_row(r, lane);
            if (m < M) {
                float o;
                if constexpr (MODE == 0) {
                    o = lora_act<ACT>(acc[i][r]);
                } else {
                    o = acc[i][r] * lora_dact<ACT>(bf2f(A[m * R + col]));
                }
                Out[m * R + col] = (uint16_t)cvt_pk(o, 0.f);
            }
        }
    }
}

template <int R, int ACT, int MODE>
int launch_skinny(long M, int K, const void *X, const void *W, const void *A, void *Out, hipStream_t st) {
    const long nblk = (M + kLoraRows - 1) / kLoraRows;
    (void)hipGetLastError();
    hipLaunchKernelGGL((lora_skinny_kernel<R, ACT, MODE>), dim3((unsigned)nblk), dim3(256), 0, st, M, K,
                       (const uint16_t *)X, (const uint16_t *)W, (const uint16_t *)A, (uint16_t *)Out);
    return (int)hipGetLastError();
}

template <int MODE>
int dispatch(long M, int K, int R, int act, const void *X, const void *W, const void *A, void *Out, hipStream_t st) {
#define CASE(RR, AA) \
    if (R == RR && act == AA) return launch_skinny<RR, AA, MODE>(M, K, X, W, A, Out, st);
    CASE(32, 0) CASE(32, 1) CASE(32, 2) CASE(64, 0) CASE(64, 1) CASE(64, 2) CASE(128, 0) CASE(128, 1) CASE(128, 2)
#undef CASE
    return -4;  // RWKV7_ESHAPE
}

}  // namespace

int lora_down_bf16(long M, int K, int R, int act, const void *x, const void *w1, void *a_out, hipStream_t st) {
    return dispatch<0>(M, K, R, act, x, w1, nullptr, a_out, st);
}
int lora_dgrad_up_bf16(long M, int K, int R, int act, const void *dz, const void *w2t, const void *a, void *dy,
                       hipStream_t st) {
    return dispatch<1>(M, K, R, act, dz, w2t, a, dy, st);
}

}  // namespace rwkv7
