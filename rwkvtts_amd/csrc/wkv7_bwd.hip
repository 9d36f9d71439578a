// rwkvtts_amd/csrc/wkv7_bwd.hip -- WKV7 backward scan for gfx950 (MI355X), written from scratch.
//
// Operator contract: torch.ops.wind_backstepping.backward (reference model/llm/cuda/wkv7_op.cpp:23-24,
// kernel wkv7_cuda.cu:54-130).  Reverse time; the state is reconstructed step by step from the fp32
// checkpoint taken every 16 steps (S_{t-1} = (S_t - v k^T - sa b^T) / w~, wkv7_cuda.cu:91-94) while the
// adjoint state dS is carried backwards:
//     dq_j  = sum_i S_t[i][j] dy_i
//     dS   += dy q^T
//     dw_j  = (sum_i dS[i][j] S_{t-1}[i][j]) * w~_j * (-exp(w_j))          dk_j = sum_i dS[i][j] v_i
//     dv_i  = sum_j dS[i][j] k_j      dSb_i = sum_j dS[i][j] b_j            db_j = sum_i dS[i][j] sa_i
//     da_j  = sum_i S_{t-1}[i][j] dSb_i
//     dS    = dS diag(w~) + dSb a^T
//
// MI355X mapping: the reference keeps THREE 64-float vectors per thread (a column of S, a row and a column
// of dS) so that every sum is thread-local, paying 2x the dS arithmetic on 64 threads per head.  Here one
// workgroup of 4 wavefronts owns a head; each lane carries one 4x4 tile of S and one of dS (single copy):
//   lane = (til = lane>>4, tj = lane&15), rows 16*wave + 4*til + {0..3}, columns 4*tj + {0..3}.
//   * row sums (dv, dSb) run over the 16 lanes of a DPP row: 4 DPP adds each, all lanes get the total,
//     so the recurrence itself needs no LDS and no barrier;
//   * column sums (dq, dw, dk, db, da) are pure outputs: reduced over the 4 `til` groups of the wave with
//     two cross-row exchanges, parked per wave in LDS, and combined across the 4 waves once per 16 steps
//     when the whole stage is written out with coalesced stores;
//   * the 8 input streams are prefetched one 16-step stage ahead into registers, converted once, and read
//     back from LDS as float4 per lane (its 4 columns / 4 rows).
#include "wkv7_common.h"

namespace rwkv7 {

namespace {
constexpr int CV_W = 0, CV_IW = 1, CV_WS = 2, CV_Q = 3, CV_K = 4, CV_A = 5, CV_B = 6;  // per-column streams
constexpr int RV_V = 0, RV_DY = 1, RV_SA = 2;                                          // per-row streams
constexpr int NCV = 7, NRV = 3, NOUT = 5;  // outputs per column: dq, dw, dk, db, da
}  // namespace

template <typename T>
__global__ __launch_bounds__(256) void wkv7_bwd_kernel(
    int T_, int H, const T *__restrict__ w_, const T *__restrict__ q_, const T *__restrict__ k_,
    const T *__restrict__ v_, const T *__restrict__ a_, const T *__restrict__ b_, const T *__restrict__ dy_,
    const float *__restrict__ s_, const float *__restrict__ sa_, T *__restrict__ dw_, T *__restrict__ dq_,
    T *__restrict__ dk_, T *__restrict__ dv_, T *__restrict__ da_, T *__restrict__ db_) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float(*sh_cv)[NCV][kN] = reinterpret_cast<float(*)[NCV][kN]>(smem);                       // [kTB][7][64]
    float(*sh_rv)[NRV][kN] = reinterpret_cast<float(*)[NRV][kN]>(smem + kTB * NCV * kN);      // [kTB][3][64]
    float(*sh_part)[kTB][NOUT][kN] =
        reinterpret_cast<float(*)[kTB][NOUT][kN]>(smem + kTB * (NCV + NRV) * kN);             // [4][kTB][5][64]
    float(*sh_dv)[kN] = reinterpret_cast<float(*)[kN]>(smem + kTB * (NCV + NRV) * kN + 4 * kTB * NOUT * kN);

    const int bh = blockIdx.x;
    const int bb = bh / H, hh = bh - bb * H;
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int til = lane >> 4, tj = lane & 15;
    const int r0 = wave * 16 + til * 4;
    const int c0 = tj * 4;

    const int st = tid >> 4;          // staging: time step inside the stage
    const int sc = (tid & 15) * 4;    // staging: element quad

    const long tstride = (long)H * kN;
    const long head_base = ((long)bb * T_ * H + hh) * kN;
    const int nchunk = T_ / kChunk;

    float S[4][4], dS[4][4];
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
        for (int c = 0; c < 4; c++) S[r][c] = dS[r][c] = 0.f;

    Raw4<T> rw, rq, rk, ra, rb, rvv, rdy;
    float4 rsa;

    auto issue = [&](int t0) {
        const long off = head_base + (long)(t0 + st) * tstride + sc;
        rw = ld4<T>(w_ + off, true);
        rq = ld4<T>(q_ + off, true);
        rk = ld4<T>(k_ + off, true);
        ra = ld4<T>(a_ + off, true);
        rb = ld4<T>(b_ + off, true);
        rvv = ld4<T>(v_ + off, true);
        rdy = ld4<T>(dy_ + off, true);
        rsa = *reinterpret_cast<const float4 *>(sa_ + off);
    };
    auto stage = [&]() {
        const float4 wr = cvt4(rw);
        float4 wf, wt, iw, ws;
        wf.x = -fast_exp(wr.x); wf.y = -fast_exp(wr.y); wf.z = -fast_exp(wr.z); wf.w = -fast_exp(wr.w);
        wt.x = fast_exp(wf.x); wt.y = fast_exp(wf.y); wt.z = fast_exp(wf.z); wt.w = fast_exp(wf.w);
        iw.x = 1.0f / wt.x; iw.y = 1.0f / wt.y; iw.z = 1.0f / wt.z; iw.w = 1.0f / wt.w;
        ws.x = wt.x * wf.x; ws.y = wt.y * wf.y; ws.z = wt.z * wf.z; ws.w = wt.w * wf.w;
        *reinterpret_cast<float4 *>(&sh_cv[st][CV_W][sc]) = wt;
        *reinterpret_cast<float4 *>(&sh_cv[st][CV_IW][sc]) = iw;
        *reinterpret_cast<float4 *>(&sh_cv[st][CV_WS][sc]) = ws;
        *reinterpret_cast<float4 *>(&sh_cv[st][CV_Q][sc]) = cvt4(rq);
        *reinterpret_cast<float4 *>(&sh_cv[st][CV_K][sc]) = cvt4(rk);
        *reinterpret_cast<float4 *>(&sh_cv[st][CV_A][sc]) = cvt4(ra);
        *reinterpret_cast<float4 *>(&sh_cv[st][CV_B][sc]) = cvt4(rb);
        *reinterpret_cast<float4 *>(&sh_rv[st][RV_V][sc]) = cvt4(rvv);
        *reinterpret_cast<float4 *>(&sh_rv[st][RV_DY][sc]) = cvt4(rdy);
        *reinterpret_cast<float4 *>(&sh_rv[st][RV_SA][sc]) = rsa;
    };

    const int nblk = T_ / kTB;  // T % 16 == 0 is checked on the host
    issue((nblk - 1) * kTB);
    stage();
    __syncthreads();

    for (int n = nblk - 1; n >= 0; n--) {
        const int t0 = n * kTB;
        if (n > 0) issue(t0 - kTB);

        {
            // a stage is exactly one checkpoint chunk (kTB == kChunk): reload S_{t0+15} from the fp32
            // checkpoint, stored transposed s[j][i] (wkv7_cuda.cu:45-48,76-82)
            const float *sp = s_ + (((long)bh * nchunk + n) * kN + c0) * kN + r0;
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const float4 x = *reinterpret_cast<const float4 *>(sp + (long)c * kN);
                S[0][c] = x.x; S[1][c] = x.y; S[2][c] = x.z; S[3][c] = x.w;
            }
        }
#pragma unroll 4
        for (int tt = kTB - 1; tt >= 0; tt--) {
            const float4 wt4 = *reinterpret_cast<const float4 *>(&sh_cv[tt][CV_W][c0]);
            const float4 iw4 = *reinterpret_cast<const float4 *>(&sh_cv[tt][CV_IW][c0]);
            const float4 q4 = *reinterpret_cast<const float4 *>(&sh_cv[tt][CV_Q][c0]);
            const float4 k4 = *reinterpret_cast<const float4 *>(&sh_cv[tt][CV_K][c0]);
            const float4 a4 = *reinterpret_cast<const float4 *>(&sh_cv[tt][CV_A][c0]);
            const float4 b4 = *reinterpret_cast<const float4 *>(&sh_cv[tt][CV_B][c0]);
            const float4 v4 = *reinterpret_cast<const float4 *>(&sh_rv[tt][RV_V][r0]);
            const float4 dy4 = *reinterpret_cast<const float4 *>(&sh_rv[tt][RV_DY][r0]);
            const float4 sa4 = *reinterpret_cast<const float4 *>(&sh_rv[tt][RV_SA][r0]);
            const float wt[4] = {wt4.x, wt4.y, wt4.z, wt4.w};
            const float iw[4] = {iw4.x, iw4.y, iw4.z, iw4.w};
            const float qv[4] = {q4.x, q4.y, q4.z, q4.w};
            const float kv[4] = {k4.x, k4.y, k4.z, k4.w};
            const float av[4] = {a4.x, a4.y, a4.z, a4.w};
            const float bv[4] = {b4.x, b4.y, b4.z, b4.w};
            const float vv[4] = {v4.x, v4.y, v4.z, v4.w};
            const float dyv[4] = {dy4.x, dy4.y, dy4.z, dy4.w};
            const float sav[4] = {sa4.x, sa4.y, sa4.z, sa4.w};

            float colp[NOUT][4];  // dq, dw, dk, db, da partial sums over this lane's 4 rows
#pragma unroll
            for (int c = 0; c < 4; c++) {
                float dq = 0.f;
#pragma unroll
                for (int r = 0; r < 4; r++) dq = fmaf(S[r][c], dyv[r], dq);
                colp[0][c] = dq;
            }
            float dvp[4] = {0.f, 0.f, 0.f, 0.f}, dsbp[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < 4; c++) {
                float dw = 0.f, dk = 0.f, db = 0.f;
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    // un-do step t: S_{t-1} = (S_t - v k^T - sa b^T) / w~
                    S[r][c] = (S[r][c] - kv[c] * vv[r] - bv[c] * sav[r]) * iw[c];
                    dS[r][c] = fmaf(dyv[r], qv[c], dS[r][c]);
                    dw = fmaf(dS[r][c], S[r][c], dw);
                    dk = fmaf(dS[r][c], vv[r], dk);
                    db = fmaf(dS[r][c], sav[r], db);
                    dvp[r] = fmaf(dS[r][c], kv[c], dvp[r]);
                    dsbp[r] = fmaf(dS[r][c], bv[c], dsbp[r]);
                }
                colp[1][c] = dw;
                colp[2][c] = dk;
                colp[3][c] = db;
            }
            float dvv[4], dsb[4];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                dvv[r] = sum16(dvp[r]);
                dsb[r] = sum16(dsbp[r]);
            }
#pragma unroll
            for (int c = 0; c < 4; c++) {
                float da = 0.f;
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    da = fmaf(S[r][c], dsb[r], da);
                    dS[r][c] = fmaf(dS[r][c], wt[c], dsb[r] * av[c]);
                }
                colp[4][c] = da;
            }
            // column partials -> wave totals, over the 4 row groups (til) of this wave.  Transposing
            // butterfly: v_permlane32_swap pairs column c with c+2 (lanes <32 keep c, lanes >=32 keep c+2),
            // v_permlane16_swap pairs c with c+1 (even DPP rows keep c, odd rows keep c+1): 15 swaps + 15
            // adds for 20 values, and row `til` ends up owning column c0 + til of all 5 outputs -- every lane
            // then stores its 5 totals (64 distinct consecutive addresses per output), no exec-masked store.
#pragma unroll
            for (int o = 0; o < NOUT; o++) {
                const float x0 = swap32_sum(colp[o][0], colp[o][2]);
                const float x1 = swap32_sum(colp[o][1], colp[o][3]);
                sh_part[wave][tt][o][c0 + til] = swap16_sum(x0, x1);
            }
            {
                // dv[r] totals are replicated over the 16 lanes of the DPP row: lane tj stores row r0 + (tj&3)
                const int rs = tj & 3;
                const float d = rs == 0 ? dvv[0] : rs == 1 ? dvv[1] : rs == 2 ? dvv[2] : dvv[3];
                sh_dv[tt][r0 + rs] = d;
            }
        }
        __syncthreads();
        {
            // stage write-out: thread (st, sc) sums the 4 wave partials of its column quad
            const long off = head_base + (long)(t0 + st) * tstride + sc;
            float4 o[NOUT];
#pragma unroll
            for (int i = 0; i < NOUT; i++) {
                float4 acc = *reinterpret_cast<const float4 *>(&sh_part[0][st][i][sc]);
#pragma unroll
                for (int wv = 1; wv < 4; wv++) {
                    const float4 p = *reinterpret_cast<const float4 *>(&sh_part[wv][st][i][sc]);
                    acc.x += p.x; acc.y += p.y; acc.z += p.z; acc.w += p.w;
                }
                o[i] = acc;
            }
            const float4 ws = *reinterpret_cast<const float4 *>(&sh_cv[st][CV_WS][sc]);
            o[1].x *= ws.x; o[1].y *= ws.y; o[1].z *= ws.z; o[1].w *= ws.w;  // dw * w~ * (-exp(w)), :108
            st4(dq_ + off, o[0]);
            st4(dw_ + off, o[1]);
            st4(dk_ + off, o[2]);
            st4(db_ + off, o[3]);
            st4(da_ + off, o[4]);
            st4(dv_ + off, *reinterpret_cast<const float4 *>(&sh_dv[st][sc]));
        }
        __syncthreads();
        if (n > 0) {
            stage();
            __syncthreads();
        }
    }
}

constexpr size_t kBwdSmemBytes = (size_t)(kTB * (NCV + NRV) * kN + 4 * kTB * NOUT * kN + kTB * kN) * sizeof(float);

template <typename T>
static int launch_bwd(int B, int T_, int H, const void *w, const void *q, const void *k, const void *v,
                      const void *a, const void *b, const void *dy, const float *s, const float *sa, void *dw,
                      void *dq, void *dk, void *dv, void *da, void *db, hipStream_t stream) {
    static bool attr_set = false;  // benign race: idempotent
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&wkv7_bwd_kernel<T>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)kBwdSmemBytes);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    (void)hipGetLastError();  // drop any stale error left by an earlier runtime call of the host program
    hipLaunchKernelGGL((wkv7_bwd_kernel<T>), dim3(B * H), dim3(256), kBwdSmemBytes, stream, T_, H, (const T *)w,
                       (const T *)q, (const T *)k, (const T *)v, (const T *)a, (const T *)b, (const T *)dy, s, sa,
                       (T *)dw, (T *)dq, (T *)dk, (T *)dv, (T *)da, (T *)db);
    return (int)hipGetLastError();
}

int wkv_bwd_bf16(int B, int T_, int H, const void *w, const void *q, const void *k, const void *v, const void *a,
                 const void *b, const void *dy, const float *s, const float *sa, void *dw, void *dq, void *dk,
                 void *dv, void *da, void *db, hipStream_t stream) {
    return launch_bwd<bf16_t>(B, T_, H, w, q, k, v, a, b, dy, s, sa, dw, dq, dk, dv, da, db, stream);
}
int wkv_bwd_f32(int B, int T_, int H, const void *w, const void *q, const void *k, const void *v, const void *a,
                const void *b, const void *dy, const float *s, const float *sa, void *dw, void *dq, void *dk,
                void *dv, void *da, void *db, hipStream_t stream) {
    return launch_bwd<float>(B, T_, H, w, q, k, v, a, b, dy, s, sa, dw, dq, dk, dv, da, db, stream);
}

}  // namespace rwkv7
