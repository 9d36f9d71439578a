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
    __shared__ __attribute__((aligned(16))) float sh_y[2][kTB][32];   // double-buffered: stored one stage late
    __shared__ __attribute__((aligned(16))) float sh_sa[2][kTB][32];

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
        // rows past T (ragged tail of the state-carrying op) re-read row T-1: valid memory, never used
        const int t = min(t0 + st, T_ - 1);
        const long off = head_base + (long)t * tstride;
        rv[0] = ld4<T>(w_ + off + sc, true);
        rv[1] = ld4<T>(q_ + off + sc, true);
        rv[2] = ld4<T>(k_ + off + sc, true);
        rv[3] = ld4<T>(a_ + off + sc, true);
        rv[4] = ld4<T>(b_ + off + sc, true);
        rvv = ld2<T>(v_ + off + half * 32 + sr, true);
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

    float Sck[8];
    auto flush = [&](int n) {
        if (SAVE) {
            // reference layout is transposed, s[j][i] (wkv7_cuda.cu:44-50)
            float *sp = s_ + (((long)bh * (T_ / kChunk) + n) * kN + c0) * kN + row;
#pragma unroll
            for (int c = 0; c < 8; c++) sp[(long)c * kN] = Sck[c];
        }
        const int t = n * kTB + st;
        if (t < T_) {
            const long off = head_base + (long)t * tstride + half * 32 + sr;
            st2(y_ + off, *reinterpret_cast<const float2 *>(&sh_y[n & 1][st][sr]));
            if (SAVE) *reinterpret_cast<float2 *>(sa_ + off) = *reinterpret_cast<const float2 *>(&sh_sa[n & 1][st][sr]);
        }
    };

    const int nblk = (T_ + kTB - 1) / kTB;
    issue(0);
    stage();
    __syncthreads();

    for (int n = 0; n < nblk; n++) {
        const int t0 = n * kTB;
        if (n + 1 < nblk) issue(t0 + kTB);
        // y/sa of the PREVIOUS stage go out now, so that their write latency runs under this stage's
        // recurrence: gfx950 has one vmcnt for loads and stores, and the next s_waitcnt vmcnt (for the
        // prefetched inputs, a whole stage from here) would otherwise also drain stores issued just before it.
        if (n > 0) flush(n - 1);
        const int steps = min(kTB, T_ - t0);

        // Operands of one time step for this lane: its 8 columns of w~,q,k,a,b and its row's v.
        struct StepOps {
            float4 w0, w1, q0, q1, k0, k1, a0, a1, b0, b1;
            float v;
        };
        auto load_ops = [&](const int tt) {
            StepOps o;
            o.w0 = *reinterpret_cast<const float4 *>(&sh_vec[tt][0][c0]);
            o.w1 = *reinterpret_cast<const float4 *>(&sh_vec[tt][0][c0 + 4]);
            o.q0 = *reinterpret_cast<const float4 *>(&sh_vec[tt][1][c0]);
            o.q1 = *reinterpret_cast<const float4 *>(&sh_vec[tt][1][c0 + 4]);
            o.k0 = *reinterpret_cast<const float4 *>(&sh_vec[tt][2][c0]);
            o.k1 = *reinterpret_cast<const float4 *>(&sh_vec[tt][2][c0 + 4]);
            o.a0 = *reinterpret_cast<const float4 *>(&sh_vec[tt][3][c0]);
            o.a1 = *reinterpret_cast<const float4 *>(&sh_vec[tt][3][c0 + 4]);
            o.b0 = *reinterpret_cast<const float4 *>(&sh_vec[tt][4][c0]);
            o.b1 = *reinterpret_cast<const float4 *>(&sh_vec[tt][4][c0 + 4]);
            o.v = sh_v[tt][rib];
            return o;
        };
        // y/sa of step tt are kept by lane cg == tt%8 of the row group (two register slots for the
        // 16 steps) and written to LDS once per stage: no exec-masked store, hence no branch, inside the
        // recurrence -- a branch per step would pin every step's ds_reads behind the previous step.
        float ykeep0 = 0.f, ykeep1 = 0.f, sakeep0 = 0.f, sakeep1 = 0.f;
        auto step = [&](const StepOps &o, const int tt) {
            const float wv[8] = {o.w0.x, o.w0.y, o.w0.z, o.w0.w, o.w1.x, o.w1.y, o.w1.z, o.w1.w};
            const float qv[8] = {o.q0.x, o.q0.y, o.q0.z, o.q0.w, o.q1.x, o.q1.y, o.q1.z, o.q1.w};
            const float kv[8] = {o.k0.x, o.k0.y, o.k0.z, o.k0.w, o.k1.x, o.k1.y, o.k1.z, o.k1.w};
            const float av[8] = {o.a0.x, o.a0.y, o.a0.z, o.a0.w, o.a1.x, o.a1.y, o.a1.z, o.a1.w};
            const float bv[8] = {o.b0.x, o.b0.y, o.b0.z, o.b0.w, o.b1.x, o.b1.y, o.b1.z, o.b1.w};
            const float vv = o.v;

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
            const bool mine = (cg == (tt & 7));
            if (tt < 8) {
                ykeep0 = mine ? y : ykeep0;
                if (SAVE) sakeep0 = mine ? sa : sakeep0;
            } else {
                ykeep1 = mine ? y : ykeep1;
                if (SAVE) sakeep1 = mine ? sa : sakeep1;
            }
        };
        if (steps == kTB) {
            // full stage: unrolled, operands of step tt+1 are fetched from LDS under the FMA chain of step tt
            StepOps cur = load_ops(0);
#pragma unroll
            for (int tt = 0; tt < kTB; tt++) {
                StepOps nxt = cur;
                if (tt + 1 < kTB) nxt = load_ops(tt + 1);
                // pin the order: hipcc's scheduler otherwise sinks these ds_reads down to their first use
                // in step tt+1 and the wave (alone on its SIMD) sits in s_waitcnt for the LDS latency
                __builtin_amdgcn_sched_barrier(0);
                step(cur, tt);
                __builtin_amdgcn_sched_barrier(0);
                cur = nxt;
            }
        } else {
            for (int tt = 0; tt < steps; tt++) step(load_ops(tt), tt);
        }
        sh_y[n & 1][cg][rib] = ykeep0;
        sh_y[n & 1][cg + 8][rib] = ykeep1;
        if (SAVE) {
            sh_sa[n & 1][cg][rib] = sakeep0;
            sh_sa[n & 1][cg + 8][rib] = sakeep1;
            // state checkpoint after every 16th step (t0 is a multiple of 16 and T % 16 == 0 here): copied
            // to spare registers now, stored with the delayed y/sa flush
#pragma unroll
            for (int c = 0; c < 8; c++) Sck[c] = S[c];
        }
        __syncthreads();
        if (n + 1 < nblk) stage();
        __syncthreads();
    }
    flush(nblk - 1);

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
    (void)hipGetLastError();  // drop any stale error left by an earlier runtime call of the host program
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
