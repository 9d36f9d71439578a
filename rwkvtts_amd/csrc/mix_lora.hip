// rwkvtts_amd/csrc/mix_lora.hip -- the small kernels around fused.mix_lora (round 4): the low-rank branches of the time-mix block
// (rwkv_s2s_single_ffn.py:171-190: w, a, v, g = Linear(D, r) -> act -> Linear(r, D)) take their inputs THROUGH the token-shift lerp
// (:160-169),   (xm (1 - mu) + shift(xm) mu) W1^T = xm (W1 * (1 - mu))^T + shift(xm) (W1 * mu)^T ,
// so one GEMM G = x [W_a ; W_b]^T on the LayerNorm output replaces four mixed [M, D] tensors and four projections.  Here:
//   wcat_fwd / wcat_bwd     W_a = W1 (1 - mu), W_b = W1 mu for all branches as one [2 R, D] matrix, and the gradients of W1, mu from d[W_a ; W_b]
//   combine_fwd / _bwd      h[t] = m_t G_a[t] + m_{t-1} G_b[t - 1] (nothing before the first step of a sequence), rounded to bf16 as the
//                           projection's output would be, then the branch's activation (none / tanh / sigmoid), written per branch [M, r_i]
// All bf16, fp32 arithmetic; every tensor here is at most [M, 2 R] with R <= 288 columns: what was six [M, D] streams is on the small side.
#include "wkv7_common.h"

namespace rwkv7 {

struct MixLoraDesc {
    int nb;               // branches (3 in layer 0: w, a, g; else 4: w, a, v, g)
    int r[4], off[4];     // rank and first column of each branch inside R
    int act[4];           // 0 none, 1 tanh, 2 sigmoid
    const void *w1[4];    // Linear(D, r_i).weight  [r_i][D]
    const void *mu[4];    // lerp coefficient [D]
    void *out[4];         // fwd: activation outputs [M][r_i];  wcat_bwd: dW1_i [r_i][D]
    void *out2[4];        // wcat_bwd: dmu_i [D];  combine_bwd: the incoming gradients d a_i [M][r_i]
};

namespace {
__device__ __forceinline__ int branch_of(const MixLoraDesc &d, int col) {
    int i = 0;
#pragma unroll
    for (int j = 1; j < 4; j++)
        if (j < d.nb && col >= d.off[j]) i = j;
    return i;
}

// wcat[row][c] = W1_i[row - off_i][c] * (1 - mu_i[c]) ; wcat[R + row][c] = W1_i[..][c] * mu_i[c]
__global__ void wcat_fwd_kernel(MixLoraDesc d, int R, int D, uint16_t *__restrict__ wcat) {
    const int row = blockIdx.x, i = branch_of(d, row);
    const uint16_t *w = reinterpret_cast<const uint16_t *>(d.w1[i]) + (long)(row - d.off[i]) * D;
    const uint16_t *mu = reinterpret_cast<const uint16_t *>(d.mu[i]);
    for (int c = threadIdx.x; c < D; c += blockDim.x) {
        const float wv = bf2f(w[c]), m = bf2f(mu[c]);
        wcat[(long)row * D + c] = f2bf(wv * (1.f - m));
        wcat[(long)(R + row) * D + c] = f2bf(wv * m);
    }
}

// dW1_i[row][c] = dWa (1 - mu) + dWb mu ;  dmu_i[c] = sum_rows W1 (dWb - dWa).  Block = 64 columns x 16 row groups of branch blockIdx.y
// (first cut: one thread per column walking all rows, 16 blocks in all: 88 us for 0.6 M elements; round 4: 4 row groups, a loop of up
// to 32 dependent load rounds per thread: 25 us; round 5: 16 row groups, the <= 8 rows of a thread unrolled with clamped, unconditional
// loads -- one memory latency)
constexpr int kWcatRG = 16, kWcatRows = 8;
__global__ __launch_bounds__(64 * kWcatRG) void wcat_bwd_kernel(MixLoraDesc d, int R, int D, const uint16_t *__restrict__ dwcat) {
    __shared__ float red[kWcatRG][64];
    const int i = blockIdx.y, cl = threadIdx.x & 63, rg = threadIdx.x >> 6, c = min(blockIdx.x * 64 + cl, D - 1);
    const bool live = blockIdx.x * 64 + cl < D;
    const uint16_t *w = reinterpret_cast<const uint16_t *>(d.w1[i]);
    uint16_t *dw = reinterpret_cast<uint16_t *>(d.out[i]);
    const int ri = d.r[i];
    const float m = bf2f(reinterpret_cast<const uint16_t *>(d.mu[i])[c]);
    float acc = 0.f;
    for (int r0 = 0; r0 < ri; r0 += kWcatRG * kWcatRows) {   // 128 rows per trip (one trip up to rank 128)
        uint16_t ra[kWcatRows], rb[kWcatRows], rw[kWcatRows];
#pragma unroll
        for (int j = 0; j < kWcatRows; j++) {
            const int rr = min(r0 + rg + kWcatRG * j, ri - 1);
            ra[j] = dwcat[(long)(d.off[i] + rr) * D + c];
            rb[j] = dwcat[(long)(R + d.off[i] + rr) * D + c];
            rw[j] = w[(long)rr * D + c];
        }
#pragma unroll
        for (int j = 0; j < kWcatRows; j++) {
            const int rr = r0 + rg + kWcatRG * j;
            const float da = bf2f(ra[j]), db = bf2f(rb[j]);
            if (rr < ri && live) {
                dw[(long)rr * D + c] = f2bf(da * (1.f - m) + db * m);
                acc += bf2f(rw[j]) * (db - da);
            }
        }
    }
    red[rg][cl] = acc;
    __syncthreads();
    if (rg == 0 && live) {
        float t = 0.f;
#pragma unroll
        for (int g = 0; g < kWcatRG; g++) t += red[g][cl];
        reinterpret_cast<uint16_t *>(d.out2[i])[c] = f2bf(t);
    }
}

__device__ __forceinline__ float act_fwd(int a, float x) {
    if (a == 1) return tanhf(x);
    if (a == 2) return 1.f / (1.f + __expf(-x));
    return x;
}
__device__ __forceinline__ float act_bwd(int a, float y, float g) {   // from the activation's OUTPUT y
    if (a == 1) return g * (1.f - y * y);
    if (a == 2) return g * (y * (1.f - y));
    return g;
}

// item = (row t, group of 8 columns of R)
__global__ void combine_fwd_kernel(MixLoraDesc d, long M, int T, int R, const uint16_t *__restrict__ G, const uint16_t *__restrict__ mask) {
    const int groups = R / 8;
    for (long it = (long)blockIdx.x * blockDim.x + threadIdx.x; it < M * groups; it += (long)gridDim.x * blockDim.x) {
        const long t = it / groups;
        const int c0 = (int)(it - t * groups) * 8;
        const int i = branch_of(d, c0);
        const float mt = mask ? bf2f(mask[t]) : 1.f;
        const bool first = (t % T) == 0;
        const float mp = first ? 0.f : (mask ? bf2f(mask[t - 1]) : 1.f);
        const uint4 ga = *reinterpret_cast<const uint4 *>(G + t * 2 * R + c0);
        uint4 gb = make_uint4(0u, 0u, 0u, 0u);
        if (!first) gb = *reinterpret_cast<const uint4 *>(G + (t - 1) * 2 * R + R + c0);
        const uint32_t *pa = reinterpret_cast<const uint32_t *>(&ga), *pb = reinterpret_cast<const uint32_t *>(&gb);
        uint32_t o[4];
#pragma unroll
        for (int e = 0; e < 4; e++) {
            float h0 = mt * __uint_as_float(pa[e] << 16) + mp * __uint_as_float(pb[e] << 16);
            float h1 = mt * __uint_as_float(pa[e] & 0xffff0000u) + mp * __uint_as_float(pb[e] & 0xffff0000u);
            h0 = bf2f(f2bf(h0));   // the projection's bf16 output, then the activation on it
            h1 = bf2f(f2bf(h1));
            o[e] = (uint32_t)f2bf(act_fwd(d.act[i], h0)) | ((uint32_t)f2bf(act_fwd(d.act[i], h1)) << 16);
        }
        *reinterpret_cast<uint4 *>(reinterpret_cast<uint16_t *>(d.out[i]) + t * d.r[i] + (c0 - d.off[i])) = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

// dG_a[t] = m_t dh[t] ; dG_b[t] = m_t dh[t + 1] (0 behind the last step of a sequence) ; dh = act'(y) da, rounded to bf16 like the separate node
__global__ void combine_bwd_kernel(MixLoraDesc d, long M, int T, int R, const uint16_t *__restrict__ mask, uint16_t *__restrict__ dG) {
    const int groups = R / 8;
    for (long it = (long)blockIdx.x * blockDim.x + threadIdx.x; it < M * groups; it += (long)gridDim.x * blockDim.x) {
        const long t = it / groups;
        const int c0 = (int)(it - t * groups) * 8;
        const int i = branch_of(d, c0), ri = d.r[i], cc = c0 - d.off[i];
        const float mt = mask ? bf2f(mask[t]) : 1.f;
        const bool last = (t % T) == T - 1;
        const uint16_t *y = reinterpret_cast<const uint16_t *>(d.out[i]), *da = reinterpret_cast<const uint16_t *>(d.out2[i]);
        const uint4 y0 = *reinterpret_cast<const uint4 *>(y + t * ri + cc), g0 = *reinterpret_cast<const uint4 *>(da + t * ri + cc);
        uint4 y1 = make_uint4(0u, 0u, 0u, 0u), g1 = y1;
        if (!last) {
            y1 = *reinterpret_cast<const uint4 *>(y + (t + 1) * ri + cc);
            g1 = *reinterpret_cast<const uint4 *>(da + (t + 1) * ri + cc);
        }
        const uint32_t *py0 = reinterpret_cast<const uint32_t *>(&y0), *pg0 = reinterpret_cast<const uint32_t *>(&g0);
        const uint32_t *py1 = reinterpret_cast<const uint32_t *>(&y1), *pg1 = reinterpret_cast<const uint32_t *>(&g1);
        uint32_t oa[4], ob[4];
#pragma unroll
        for (int e = 0; e < 4; e++) {
            float a0 = act_bwd(d.act[i], __uint_as_float(py0[e] << 16), __uint_as_float(pg0[e] << 16));
            float a1 = act_bwd(d.act[i], __uint_as_float(py0[e] & 0xffff0000u), __uint_as_float(pg0[e] & 0xffff0000u));
            float b0 = act_bwd(d.act[i], __uint_as_float(py1[e] << 16), __uint_as_float(pg1[e] << 16));
            float b1 = act_bwd(d.act[i], __uint_as_float(py1[e] & 0xffff0000u), __uint_as_float(pg1[e] & 0xffff0000u));
            a0 = bf2f(f2bf(a0)); a1 = bf2f(f2bf(a1)); b0 = bf2f(f2bf(b0)); b1 = bf2f(f2bf(b1));
            oa[e] = (uint32_t)f2bf(mt * a0) | ((uint32_t)f2bf(mt * a1) << 16);
            ob[e] = last ? 0u : ((uint32_t)f2bf(mt * b0) | ((uint32_t)f2bf(mt * b1) << 16));
        }
        *reinterpret_cast<uint4 *>(dG + t * 2 * R + c0) = make_uint4(oa[0], oa[1], oa[2], oa[3]);
        *reinterpret_cast<uint4 *>(dG + t * 2 * R + R + c0) = make_uint4(ob[0], ob[1], ob[2], ob[3]);
    }
}
}  // namespace

int mix_lora_wcat_fwd(const MixLoraDesc &d, int R, int D, void *wcat, hipStream_t st) {
    (void)hipGetLastError();
    hipLaunchKernelGGL(wcat_fwd_kernel, dim3(R), dim3(256), 0, st, d, R, D, (uint16_t *)wcat);
    return (int)hipGetLastError();
}
int mix_lora_wcat_bwd(const MixLoraDesc &d, int R, int D, const void *dwcat, hipStream_t st) {
    (void)hipGetLastError();
    hipLaunchKernelGGL(wcat_bwd_kernel, dim3((D + 63) / 64, d.nb), dim3(64 * kWcatRG), 0, st, d, R, D, (const uint16_t *)dwcat);
    return (int)hipGetLastError();
}
int mix_lora_combine_fwd(const MixLoraDesc &d, long M, int T, int R, const void *G, const void *mask, hipStream_t st) {
    (void)hipGetLastError();
    const long items = M * (R / 8);
    const int grid = (int)((items + 255) / 256 < 8192 ? (items + 255) / 256 : 8192);
    hipLaunchKernelGGL(combine_fwd_kernel, dim3(grid), dim3(256), 0, st, d, M, T, R, (const uint16_t *)G, (const uint16_t *)mask);
    return (int)hipGetLastError();
}
int mix_lora_combine_bwd(const MixLoraDesc &d, long M, int T, int R, const void *mask, void *dG, hipStream_t st) {
    (void)hipGetLastError();
    const long items = M * (R / 8);
    const int grid = (int)((items + 255) / 256 < 8192 ? (items + 255) / 256 : 8192);
    hipLaunchKernelGGL(combine_bwd_kernel, dim3(grid), dim3(256), 0, st, d, M, T, R, (const uint16_t *)mask, (uint16_t *)dG);
    return (int)hipGetLastError();
}

}  // namespace rwkv7
