// rwkvtts_amd/csrc/wkv7_fwd.hip -- WKV7 forward scan for gfx950 (MI355X), written from scratch.
//
// Operator contract: torch.ops.wind_backstepping.forward (reference model/llm/cuda/wkv7_op.cpp:21-22,
// kernel wkv7_cuda.cu:10-52) and torch.ops.rwkv7_state_fwd_fp16.forward / wkv7s.forward
// (rwkv7_state_fwd_fp16.cu:9-57).  Per (batch, head), for t = 0..T-1:
//     w~ = exp(-exp(w_t));  sa_i = sum_j a_j S_ij;  S_ij = S_ij w~_j + sa_i b_j + v_i k_j;  y_i = sum_j S_ij q_j
//
// MI355X mapping (NOT the reference's one-thread-per-row/64-thread-block shape, which would leave 7/8 of
// the 1024 SIMDs idle at B*H = 128):
//   * the 64 state rows of a head never talk to each other in the forward recurrence, so a head is split
//     over 2 workgroups x 4 wavefronts = 8 wavefronts (1024 wavefronts at B*H=128: one per SIMD);
//   * inside a wavefront lane = (row group rg = lane>>3, column group cg = lane&7): each lane carries
//     one state row x 8 columns in VGPRs (fp32); sa_i and y_i are 8-lane sums done with DPP
//     (quad_perm, quad_perm, row_half_mirror) -- no LDS crossbar, no barrier per step;
//   * w,q,k,a,b,v arrive by coalesced 8-byte (bf16) / 16-byte (fp32) global loads, 16 time steps per
//     stage, prefetched into registers one stage ahead, converted once (exp(-exp(w)) once per element,
//     not once per row) and parked in LDS as fp32; lanes read their 8 columns with 2 ds_read_b128 per
//     vector (8 distinct 32-B segments per instruction -> conflict free, rest is broadcast);
//   * y / sa leave through an LDS staging tile so the global stores are contiguous per time step.
#include "wkv7_common.h"

namespace rwkv7 {

template <typename T, bool SAVE, bool STATE>
__global__ __launch_bounds__(256) void wkv7_fwd_kernel(int T_, int H, const T *__restrict__ w_,
                                                       const T *__restrict__ q_, const T *__restrict__ k_,
                                                       const T *__restrict__ v_, const T *__restrict__ a_,
                                                       const T *__restrict__ b_, T *__restrict__ y_,
                                                       float *__restrict__ s_, float *__restrict__ sa_,
                                                       float *__restrict__ state_) {
    __shared__ __attribute__((aligned(16))) float sh_vec[kTB][5][kN];  // w~, q, k, a, b  (20 KiB)
    __shared__ __attribute__((aligned(16))) float sh_v[kTB][32];
    __shared__ __attribute__((aligned(16))) float sh_y[kTB][32];
    __shared__ __attribute__((aligned(16))) float sh_sa[kTB][32];

    const int half = blockIdx.x & 1;  // which 32 rows of the head
    const int bh = blockIdx.x >> 1;
    const int bb = bh / H, hh = bh - bb * H;
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int rg = lane >> 3, cg = lane & 7;
    const int rib = wave * 8 + rg;   // row inside this block's 32
    const int row = half * 32 + rib; // row inside the head
    const int c0 = cg * 8;

    // staging roles: 16 time steps x 16 column quads / 16 row pairs
    const int st = tid >> 4;
    const int sc = (tid & 15) * 4;
    const int sr = (tid & 15) * 2;

    const long tstride = (long)H * kN;
    const long head_base = ((long)bb * T_ * H + hh) * kN;

    float S[8];
    if (STATE) {
        const float *sp = state_ + ((long)bh * kN + row) * kN + c0;
        const float4 s0 = *reinterpret_cast<const float4 *>(sp);
        const float4 s1 = *reinterpret_cast<const float4 *>(sp + 4);
        S[0] = s0.x; S[1] = s0.y; S[2] = s0.z; S[3] = s0.w;
        S[4] = s1.x; S[5] = s1.y; S[6] = s1.z; S[7] = s1.w;
    } else {
#pragma unroll
        for (int c = 0; c < 8; c++) S[c] = 0.f;
    }

    Raw4<T> rv[5];
    Raw2<T> rvv;

    auto issue = [&](int t0) {
        const int t = t0 + st;
        const bool ok = t < T_;
        const long off = head_base + (long)(ok ? t : 0) * tstride;
        rv[0] = ld4<T>(w_ + off + sc, ok);
        rv[1] = ld4<T>(q_ + off + sc, ok);
        rv[2] = ld4<T>(k_ + off + sc, ok);
        rv[3] = ld4<T>(a_ + off + sc, ok);
        rv[4] = ld4<T>(b_ + off + sc, ok);
        rvv = ld2<T>(v_ + off + half * 32 + sr, ok);
    };
    auto stage = [&]() {
        float4 f = cvt4(rv[0]);
        f.x = fast_exp(-fast_exp(f.x));
        f.y = fast_exp(-fast_exp(f.y));
        f.z = fast_exp(-fast_exp(f.z));
        f.w = fast_exp(-fast_exp(f.w));
        *reinterpret_cast<float4 *>(&sh_vec[st][0][sc]) = f;
#pragma unroll
        for (int i = 1; i < 5; i++) *reinterpret_cast<float4 *>(&sh_vec[st][i][sc]) = cvt4(rv[i]);
        *reinterpret_cast<float2 *>(&sh_v[st][sr]) = cvt2(rvv);
    };

    const int nblk = (T_ + kTB - 1) / kTB;
    issue(0);
    stage();
    __syncthreads();

    for (int n = 0; n < nblk; n++) {
        const int t0 = n * kTB;
        if (n + 1 < nblk) issue(t0 + kTB);
        const int steps = min(kTB, T_ - t0);

        for (int tt = 0; tt < steps; tt++) {
            const float4 w0 = *reinterpret_cast<const float4 *>(&sh_vec[tt][0][c0]);
            const float4 w1 = *reinterpret_cast<const float4 *>(&sh_vec[tt][0][c0 + 4]);
            const float4 q0 = *reinterpret_cast<const float4 *>(&sh_vec[tt][1][c0]);
            const float4 q1 = *reinterpret_cast<const float4 *>(&sh_vec[tt][1][c0 + 4]);
            const float4 k0 = *reinterpret_cast<const float4 *>(&sh_vec[tt][2][c0]);
            const float4 k1 = *reinterpret_cast<const float4 *>(&sh_vec[tt][2][c0 + 4]);
            const float4 a0 = *reinterpret_cast<const float4 *>(&sh_vec[tt][3][c0]);
            const float4 a1 = *reinterpret_cast<const float4 *>(&sh_vec[tt][3][c0 + 4]);
            const float4 b0 = *reinterpret_cast<const float4 *>(&sh_vec[tt][4][c0]);
            const float4 b1 = *reinterpret_cast<const float4 *>(&sh_vec[tt][4][c0 + 4]);
            const float vv = sh_v[tt][rib];
            const float wv[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
            const float qv[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
            const float kv[8] = {k0.x, k0.y, k0.z, k0.w, k1.x, k1.y, k1.z, k1.w};
            const float av[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
            const float bv[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};

            float sa0 = 0.f, sa1 = 0.f;
#pragma unroll
            for (int c = 0; c < 8; c += 2) {
                sa0 = fmaf(av[c], S[c], sa0);
                sa1 = fmaf(av[c + 1], S[c + 1], sa1);
            }
            const float sa = sum8(sa0 + sa1);

            float y0 = 0.f, y1 = 0.f;
#pragma unroll
            for (int c = 0; c < 8; c += 2) {
                S[c] = fmaf(S[c], wv[c], fmaf(sa, bv[c], kv[c] * vv));
                S[c + 1] = fmaf(S[c + 1], wv[c + 1], fmaf(sa, bv[c + 1], kv[c + 1] * vv));
                y0 = fmaf(S[c], qv[c], y0);
                y1 = fmaf(S[c + 1], qv[c + 1], y1);
            }
            const float y = sum8(y0 + y1);
            if (cg == 0) {
                sh_y[tt][rib] = y;
                if (SAVE) sh_sa[tt][rib] = sa;
            }
            if (SAVE) {
                const int t = t0 + tt;
                if (((t + 1) & (kChunk - 1)) == 0) {  // wkv7_cuda.cu:44-50, transposed layout [j][i]
                    float *sp = s_ + (((long)bh * (T_ / kChunk) + t / kChunk) * kN + c0) * kN + row;
#pragma unroll
                    for (int c = 0; c < 8; c++) sp[(long)c * kN] = S[c];
                }
            }
        }
        __syncthreads();
        {
            const int t = t0 + st;
            if (t < T_) {
                const long off = head_base + (long)t * tstride + half * 32 + sr;
                st2(y_ + off, *reinterpret_cast<const float2 *>(&sh_y[st][sr]));
                if (SAVE) *reinterpret_cast<float2 *>(sa_ + off) = *reinterpret_cast<const float2 *>(&sh_sa[st][sr]);
            }
        }
        if (n + 1 < nblk) stage();
        __syncthreads();
    }

    if (STATE) {
        float *sp = state_ + ((long)bh * kN + row) * kN + c0;
        *reinterpret_cast<float4 *>(sp) = make_float4(S[0], S[1], S[2], S[3]);
        *reinterpret_cast<float4 *>(sp + 4) = make_float4(S[4], S[5], S[6], S[7]);
    }
}

template <typename T>
static int launch_fwd(int B, int T_, int H, const void *w, const void *q, const void *k, const void *v,
                      const void *a, const void *b, void *y, float *s, float *sa, float *state,
                      hipStream_t stream) {
    const dim3 grid(B * H * 2), block(256);
    const T *W = (const T *)w, *Q = (const T *)q, *K = (const T *)k, *V = (const T *)v, *A = (const T *)a,
            *Bv = (const T *)b;
    if (state) {
        hipLaunchKernelGGL((wkv7_fwd_kernel<T, false, true>), grid, block, 0, stream, T_, H, W, Q, K, V, A, Bv,
                           (T *)y, nullptr, nullptr, state);
    } else if (s && sa) {
        hipLaunchKernelGGL((wkv7_fwd_kernel<T, true, false>), grid, block, 0, stream, T_, H, W, Q, K, V, A, Bv,
                           (T *)y, s, sa, nullptr);
    } else {
        hipLaunchKernelGGL((wkv7_fwd_kernel<T, false, false>), grid, block, 0, stream, T_, H, W, Q, K, V, A, Bv,
                           (T *)y, nullptr, nullptr, nullptr);
    }
    return (int)hipGetLastError();
}

int wkv_fwd_bf16(int B, int T_, int H, const void *w, const void *q, const void *k, const void *v, const void *a,
                 const void *b, void *y, float *s, float *sa, float *state, hipStream_t stream) {
    return launch_fwd<bf16_t>(B, T_, H, w, q, k, v, a, b, y, s, sa, state, stream);
}
int wkv_fwd_f32(int B, int T_, int H, const void *w, const void *q, const void *k, const void *v, const void *a,
                const void *b, void *y, float *s, float *sa, float *state, hipStream_t stream) {
    return launch_fwd<float>(B, T_, H, w, q, k, v, a, b, y, s, sa, state, stream);
}

}  // namespace rwkv7
