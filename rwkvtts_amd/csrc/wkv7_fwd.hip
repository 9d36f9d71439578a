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

// CW = state columns per lane.  CW = 8: 8 lanes per row, 4 waves (256 threads) per 32 rows -- one wave per SIMD at
// B*H = 128, where the lone wave sits in LDS/DPP latency for ~70% of the cycles (rocprofv3 PMC: VALU busy 29%).
// CW = 4: 16 lanes per row, 8 waves (512 threads) per 32 rows -- two waves per SIMD cover each other's latency at
// the price of one more DPP step per reduction; picked by the launcher when B*H is too small to fill the SIMDs twice.
template <typename T, bool SAVE, bool STATE, int CW>
__global__ __launch_bounds__(2048 / CW) void wkv7_fwd_kernel(int T_, int H, const T *__restrict__ w_,
                                                             const T *__restrict__ q_, const T *__restrict__ k_,
                                                             const T *__restrict__ v_, const T *__restrict__ a_,
                                                             const T *__restrict__ b_, T *__restrict__ y_,
                                                             float *__restrict__ s_, float *__restrict__ sa_,
                                                             float *__restrict__ state_) {
    constexpr int LPR = kN / CW;        // lanes per state row (8 or 16)
    constexpr int RPW = 64 / LPR;       // rows per wave (8 or 4)
    constexpr int NSLOT = kTB / LPR;    // y/sa keep slots per lane (2 or 1)
    constexpr bool WIDE = CW == 4;      // 512 threads: the staging work is split between the two halves
    __shared__ __attribute__((aligned(16))) float sh_vec[kTB][5][kN];  // w~, q, k, a, b  (20 KiB)
    __shared__ __attribute__((aligned(16))) float sh_v[kTB][32];
    __shared__ __attribute__((aligned(16))) float sh_y[2][kTB][32];   // double-buffered: stored one stage late
    __shared__ __attribute__((aligned(16))) float sh_sa[2][kTB][32];

    // which 32 rows of which head.  Workgroups are dealt round-robin to the 8 XCDs; the two halves of a head get block ids
    // g and g + 8 so that they share an L2 and the input streams are fetched from HBM once.
    int half, bh;
    if ((gridDim.x & 15) == 0) {
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        bh = (j >> 1) * 8 + xcd;
        half = j & 1;
    } else {
        half = blockIdx.x & 1;
        bh = blockIdx.x >> 1;
    }
    const int bb = bh / H, hh = bh - bb * H;
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int rg = lane / LPR, cg = lane % LPR;
    const int rib = wave * RPW + rg;  // row inside this block's 32
    const int row = half * 32 + rib;  // row inside the head
    const int c0 = cg * CW;

    // staging roles: 16 time steps x 16 column quads / 16 row pairs; with 512 threads the lower 256 take w,q,k and
    // the y/sa write-out, the upper 256 take a,b,v
    const int sub = tid >> 8;
    const int st = (tid & 255) >> 4;
    const int sc = (tid & 15) * 4;
    const int sr = (tid & 15) * 2;
    const bool lo_half = !WIDE || sub == 0, hi_half = !WIDE || sub == 1;

    const long tstride = (long)H * kN;
    const long head_base = ((long)bb * T_ * H + hh) * kN;

    float S[CW];
    if (STATE) {
        const float *sp = state_ + ((long)bh * kN + row) * kN + c0;
#pragma unroll
        for (int c = 0; c < CW; c += 4) {
            const float4 s0 = *reinterpret_cast<const float4 *>(sp + c);
            S[c] = s0.x; S[c + 1] = s0.y; S[c + 2] = s0.z; S[c + 3] = s0.w;
        }
    } else {
#pragma unroll
        for (int c = 0; c < CW; c++) S[c] = 0.f;
    }

    Raw4<T> rv[5];
    Raw2<T> rvv;

    auto issue = [&](int t0) {
        // rows past T (ragged tail of the state-carrying op) re-read row T-1: valid memory, never used
        const int t = min(t0 + st, T_ - 1);
        const long off = head_base + (long)t * tstride;
        if (lo_half) {
            rv[0] = ld4<T>(w_ + off + sc, true);
            rv[1] = ld4<T>(q_ + off + sc, true);
            rv[2] = ld4<T>(k_ + off + sc, true);
        }
        if (hi_half) {
            rv[3] = ld4<T>(a_ + off + sc, true);
            rv[4] = ld4<T>(b_ + off + sc, true);
            rvv = ld2<T>(v_ + off + half * 32 + sr, true);
        }
    };
    auto stage = [&]() {
        if (lo_half) {
            float4 f = cvt4(rv[0]);
            f.x = fast_exp(-fast_exp(f.x));
            f.y = fast_exp(-fast_exp(f.y));
            f.z = fast_exp(-fast_exp(f.z));
            f.w = fast_exp(-fast_exp(f.w));
            *reinterpret_cast<float4 *>(&sh_vec[st][0][sc]) = f;
            *reinterpret_cast<float4 *>(&sh_vec[st][1][sc]) = cvt4(rv[1]);
            *reinterpret_cast<float4 *>(&sh_vec[st][2][sc]) = cvt4(rv[2]);
        }
        if (hi_half) {
            *reinterpret_cast<float4 *>(&sh_vec[st][3][sc]) = cvt4(rv[3]);
            *reinterpret_cast<float4 *>(&sh_vec[st][4][sc]) = cvt4(rv[4]);
            *reinterpret_cast<float2 *>(&sh_v[st][sr]) = cvt2(rvv);
        }
    };

    float Sck[CW];
    auto flush = [&](int n) {
        if (SAVE) {
            // reference layout is transposed, s[j][i] (wkv7_cuda.cu:44-50)
            float *sp = s_ + (((long)bh * (T_ / kChunk) + n) * kN + c0) * kN + row;
#pragma unroll
            for (int c = 0; c < CW; c++) sp[(long)c * kN] = Sck[c];
        }
        const int t = n * kTB + st;
        if (lo_half && t < T_) {
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

        // Operands of one time step for this lane: its CW columns of w~,q,k,a,b and its row's v.
        struct StepOps {
            float4 w[CW / 4], q[CW / 4], k[CW / 4], a[CW / 4], b[CW / 4];
            float v;
        };
        auto load_ops = [&](const int tt) {
            StepOps o;
#pragma unroll
            for (int i = 0; i < CW / 4; i++) {
                o.w[i] = *reinterpret_cast<const float4 *>(&sh_vec[tt][0][c0 + 4 * i]);
                o.q[i] = *reinterpret_cast<const float4 *>(&sh_vec[tt][1][c0 + 4 * i]);
                o.k[i] = *reinterpret_cast<const float4 *>(&sh_vec[tt][2][c0 + 4 * i]);
                o.a[i] = *reinterpret_cast<const float4 *>(&sh_vec[tt][3][c0 + 4 * i]);
                o.b[i] = *reinterpret_cast<const float4 *>(&sh_vec[tt][4][c0 + 4 * i]);
            }
            o.v = sh_v[tt][rib];
            return o;
        };
        // y/sa of step tt are kept by lane cg == tt % LPR of the row group (NSLOT register slots for the
        // 16 steps) and written to LDS once per stage: no exec-masked store, hence no branch, inside the
        // recurrence -- a branch per step would pin every step's ds_reads behind the previous step.
        float ykeep[NSLOT], sakeep[NSLOT];
#pragma unroll
        for (int i = 0; i < NSLOT; i++) ykeep[i] = sakeep[i] = 0.f;
        auto step = [&](const StepOps &o, const int tt) {
            // This body is instantiated twice (unrolled full stage, looped ragged tail).  Under -ffast-math the two copies were
            // contracted differently: y of a step differed in the last fp32 bit (one bf16 ulp on ~1e-4 of the outputs) depending on
            // whether the step fell into a full stage -- i.e. on where a caller splits a sequence between two state-carrying calls,
            // which the reference's single loop cannot do (rwkv7_state_fwd_fp16.cu:23-52).  The arithmetic is pinned as written.
#pragma clang fp reassociate(off) contract(off)
            float wv[CW], qv[CW], kv[CW], av[CW], bv[CW];
#pragma unroll
            for (int i = 0; i < CW / 4; i++) {
                wv[4 * i] = o.w[i].x; wv[4 * i + 1] = o.w[i].y; wv[4 * i + 2] = o.w[i].z; wv[4 * i + 3] = o.w[i].w;
                qv[4 * i] = o.q[i].x; qv[4 * i + 1] = o.q[i].y; qv[4 * i + 2] = o.q[i].z; qv[4 * i + 3] = o.q[i].w;
                kv[4 * i] = o.k[i].x; kv[4 * i + 1] = o.k[i].y; kv[4 * i + 2] = o.k[i].z; kv[4 * i + 3] = o.k[i].w;
                av[4 * i] = o.a[i].x; av[4 * i + 1] = o.a[i].y; av[4 * i + 2] = o.a[i].z; av[4 * i + 3] = o.a[i].w;
                bv[4 * i] = o.b[i].x; bv[4 * i + 1] = o.b[i].y; bv[4 * i + 2] = o.b[i].z; bv[4 * i + 3] = o.b[i].w;
            }
            const float vv = o.v;

            float sa0 = 0.f, sa1 = 0.f;
#pragma unroll
            for (int c = 0; c < CW; c += 2) {
                sa0 = fmaf(av[c], S[c], sa0);
                sa1 = fmaf(av[c + 1], S[c + 1], sa1);
            }
            const float sa = LPR == 8 ? sum8(sa0 + sa1) : sum16(sa0 + sa1);

            float y0 = 0.f, y1 = 0.f;
#pragma unroll
            for (int c = 0; c < CW; c += 2) {
                S[c] = fmaf(S[c], wv[c], fmaf(sa, bv[c], kv[c] * vv));
                S[c + 1] = fmaf(S[c + 1], wv[c + 1], fmaf(sa, bv[c + 1], kv[c + 1] * vv));
                y0 = fmaf(S[c], qv[c], y0);
                y1 = fmaf(S[c + 1], qv[c + 1], y1);
            }
            const float y = LPR == 8 ? sum8(y0 + y1) : sum16(y0 + y1);
            const bool mine = (cg == (tt % LPR));
#pragma unroll
            for (int i = 0; i < NSLOT; i++) {
                if (tt / LPR == i) {  // compile-time inside the unrolled stage, a uniform branch in the ragged tail
                    ykeep[i] = mine ? y : ykeep[i];
                    if (SAVE) sakeep[i] = mine ? sa : sakeep[i];
                }
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
                // in step tt+1 and the wave sits in s_waitcnt for the LDS latency
                __builtin_amdgcn_sched_barrier(0);
                step(cur, tt);
                __builtin_amdgcn_sched_barrier(0);
                cur = nxt;
            }
        } else {
            for (int tt = 0; tt < steps; tt++) step(load_ops(tt), tt);
        }
#pragma unroll
        for (int i = 0; i < NSLOT; i++) {
            sh_y[n & 1][cg + LPR * i][rib] = ykeep[i];
            if (SAVE) sh_sa[n & 1][cg + LPR * i][rib] = sakeep[i];
        }
        if (SAVE) {
            // state checkpoint after every 16th step (t0 is a multiple of 16 and T % 16 == 0 here): copied
            // to spare registers now, stored with the delayed y/sa flush
#pragma unroll
            for (int c = 0; c < CW; c++) Sck[c] = S[c];
        }
        __syncthreads();
        if (n + 1 < nblk) stage();
        __syncthreads();
    }
    flush(nblk - 1);

    if (STATE) {
        float *sp = state_ + ((long)bh * kN + row) * kN + c0;
#pragma unroll
        for (int c = 0; c < CW; c += 4) *reinterpret_cast<float4 *>(sp + c) = make_float4(S[c], S[c + 1], S[c + 2], S[c + 3]);
    }
}

// Heads per launch below which the 512-thread / 4-columns-per-lane shape is used: with B*H >= 256 heads the
// 256-thread shape already puts >= 2 waves on every SIMD and does less reduction work.
constexpr int kWideBelowHeads = 256;

template <typename T, int CW>
static void launch_fwd_cw(int B, int T_, int H, const T *W, const T *Q, const T *K, const T *V, const T *A, const T *Bv,
                          T *y, float *s, float *sa, float *state, hipStream_t stream) {
    const dim3 grid(B * H * 2), block(2048 / CW);
    if (state) {
        hipLaunchKernelGGL((wkv7_fwd_kernel<T, false, true, CW>), grid, block, 0, stream, T_, H, W, Q, K, V, A, Bv, y,
                           nullptr, nullptr, state);
    } else if (s && sa) {
        hipLaunchKernelGGL((wkv7_fwd_kernel<T, true, false, CW>), grid, block, 0, stream, T_, H, W, Q, K, V, A, Bv, y, s,
                           sa, nullptr);
    } else {
        hipLaunchKernelGGL((wkv7_fwd_kernel<T, false, false, CW>), grid, block, 0, stream, T_, H, W, Q, K, V, A, Bv, y,
                           nullptr, nullptr, nullptr);
    }
}

template <typename T>
static int launch_fwd(int B, int T_, int H, const void *w, const void *q, const void *k, const void *v,
                      const void *a, const void *b, void *y, float *s, float *sa, float *state, int force_cw,
                      hipStream_t stream) {
    // force_cw: 0 = automatic by B*H; 4 / 8 = that many state columns per lane (the *_variant entry points: A/B measurements)
    (void)hipGetLastError();  // drop any stale error left by an earlier runtime call of the host program
    const T *W = (const T *)w, *Q = (const T *)q, *K = (const T *)k, *V = (const T *)v, *A = (const T *)a,
            *Bv = (const T *)b;
    const bool wide = force_cw ? force_cw == 4 : (long)B * H < kWideBelowHeads;
    if (wide)
        launch_fwd_cw<T, 4>(B, T_, H, W, Q, K, V, A, Bv, (T *)y, s, sa, state, stream);
    else
        launch_fwd_cw<T, 8>(B, T_, H, W, Q, K, V, A, Bv, (T *)y, s, sa, state, stream);
    return (int)hipGetLastError();
}

int wkv_fwd_bf16(int B, int T_, int H, const void *w, const void *q, const void *k, const void *v, const void *a,
                 const void *b, void *y, float *s, float *sa, float *state, int force_cw, hipStream_t stream) {
    return launch_fwd<bf16_t>(B, T_, H, w, q, k, v, a, b, y, s, sa, state, force_cw, stream);
}
int wkv_fwd_f32(int B, int T_, int H, const void *w, const void *q, const void *k, const void *v, const void *a,
                const void *b, void *y, float *s, float *sa, float *state, int force_cw, hipStream_t stream) {
    return launch_fwd<float>(B, T_, H, w, q, k, v, a, b, y, s, sa, state, force_cw, stream);
}

}  // namespace rwkv7
