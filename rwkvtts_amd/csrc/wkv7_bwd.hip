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

// Output pointers.  With RT = 4 one workgroup owns all 64 state rows of a head and set [0] receives the final
// gradients (the reference-schema op).  With RT = 2 the head is split over TWO workgroups (32 rows each, so that all
// 256 CUs work at B*H = 128): dv is complete per row, the column sums dq,dw,dk,db,da are partial over the workgroup's
// rows and go to set [part]; the consumer adds the two sets (rwkv7_tmix_prepare_bwd takes both).
template <typename T>
struct BwdOuts {
    T *dw[2], *dq[2], *dk[2], *db[2], *da[2];
    T *dv;
};

template <typename T, int RT>
__global__ __launch_bounds__(256) void wkv7_bwd_kernel(
    int T_, int H, const T *__restrict__ w_, const T *__restrict__ q_, const T *__restrict__ k_,
    const T *__restrict__ v_, const T *__restrict__ a_, const T *__restrict__ b_, const T *__restrict__ dy_,
    const float *__restrict__ s_, const float *__restrict__ sa_, BwdOuts<T> outs) {
    constexpr int NSPLIT = 4 / RT;       // workgroups per head
    constexpr int ROWS_WG = 16 * RT;     // state rows per workgroup
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float(*sh_cv)[NCV][kN] = reinterpret_cast<float(*)[NCV][kN]>(smem);                       // [kTB][7][64]
    float(*sh_rv)[NRV][kN] = reinterpret_cast<float(*)[NRV][kN]>(smem + kTB * NCV * kN);      // [kTB][3][64]
    float(*sh_part)[kTB][NOUT][kN] =
        reinterpret_cast<float(*)[kTB][NOUT][kN]>(smem + kTB * (NCV + NRV) * kN);             // [4][kTB][5][64]
    float(*sh_dv)[kN] = reinterpret_cast<float(*)[kN]>(smem + kTB * (NCV + NRV) * kN + 4 * kTB * NOUT * kN);

    const int part = blockIdx.x % NSPLIT;
    const int bh = blockIdx.x / NSPLIT;
    const int bb = bh / H, hh = bh - bb * H;
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int til = lane >> 4, tj = lane & 15;
    const int r0l = wave * (4 * RT) + til * RT;  // first row of this lane's tile, inside the workgroup
    const int r0 = part * ROWS_WG + r0l;         // ... inside the head
    const int c0 = tj * 4;

    const int st = tid >> 4;          // staging: time step inside the stage
    const int sc = (tid & 15) * 4;    // staging: element quad

    const long tstride = (long)H * kN;
    const long head_base = ((long)bb * T_ * H + hh) * kN;
    const int nchunk = T_ / kChunk;

    float S[RT][4], dS[RT][4];
#pragma unroll
    for (int r = 0; r < RT; r++)
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
    auto ld_rows = [&](const float *p, float (&o)[RT]) {  // RT consecutive floats, RT-aligned
        if constexpr (RT == 4) {
            const float4 x = *reinterpret_cast<const float4 *>(p);
            o[0] = x.x; o[1] = x.y; o[2] = x.z; o[3] = x.w;
        } else {
            const float2 x = *reinterpret_cast<const float2 *>(p);
            o[0] = x.x; o[1] = x.y;
        }
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
                float x[RT];
                ld_rows(sp + (long)c * kN, x);
#pragma unroll
                for (int r = 0; r < RT; r++) S[r][c] = x[r];
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
            const float wt[4] = {wt4.x, wt4.y, wt4.z, wt4.w};
            const float iw[4] = {iw4.x, iw4.y, iw4.z, iw4.w};
            const float qv[4] = {q4.x, q4.y, q4.z, q4.w};
            const float kv[4] = {k4.x, k4.y, k4.z, k4.w};
            const float av[4] = {a4.x, a4.y, a4.z, a4.w};
            const float bv[4] = {b4.x, b4.y, b4.z, b4.w};
            float vv[RT], dyv[RT], sav[RT];
            ld_rows(&sh_rv[tt][RV_V][r0], vv);
            ld_rows(&sh_rv[tt][RV_DY][r0], dyv);
            ld_rows(&sh_rv[tt][RV_SA][r0], sav);

            float colp[NOUT][4];  // dq, dw, dk, db, da partial sums over this lane's RT rows
#pragma unroll
            for (int c = 0; c < 4; c++) {
                float dq = 0.f;
#pragma unroll
                for (int r = 0; r < RT; r++) dq = fmaf(S[r][c], dyv[r], dq);
                colp[0][c] = dq;
            }
            float dvp[RT], dsbp[RT];
#pragma unroll
            for (int r = 0; r < RT; r++) dvp[r] = dsbp[r] = 0.f;
#pragma unroll
            for (int c = 0; c < 4; c++) {
                float dw = 0.f, dk = 0.f, db = 0.f;
#pragma unroll
                for (int r = 0; r < RT; r++) {
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
            float dvv[RT], dsb[RT];
#pragma unroll
            for (int r = 0; r < RT; r++) {
                dvv[r] = sum16(dvp[r]);
                dsb[r] = sum16(dsbp[r]);
            }
#pragma unroll
            for (int c = 0; c < 4; c++) {
                float da = 0.f;
#pragma unroll
                for (int r = 0; r < RT; r++) {
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
                // dv[r] totals are replicated over the 16 lanes of the DPP row: lane tj stores local row r0l + tj % RT
                const int rs = tj & (RT - 1);
                float d = dvv[0];
#pragma unroll
                for (int r = 1; r < RT; r++) d = rs == r ? dvv[r] : d;
                sh_dv[tt][r0l + rs] = d;
            }
        }
        __syncthreads();
        {
            // stage write-out: thread (st, sc) sums the 4 wave partials of its column quad
            const long off_t = head_base + (long)(t0 + st) * tstride;
            const long off = off_t + sc;
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
            st4(outs.dq[part] + off, o[0]);
            st4(outs.dw[part] + off, o[1]);
            st4(outs.dk[part] + off, o[2]);
            st4(outs.db[part] + off, o[3]);
            st4(outs.da[part] + off, o[4]);
            if (sc < ROWS_WG) st4(outs.dv + off_t + part * ROWS_WG + sc, *reinterpret_cast<const float4 *>(&sh_dv[st][sc]));
        }
        __syncthreads();
        if (n > 0) {
            stage();
            __syncthreads();
        }
    }
}

constexpr size_t kBwdSmemBytes = (size_t)(kTB * (NCV + NRV) * kN + 4 * kTB * NOUT * kN + kTB * kN) * sizeof(float);

template <typename T, int RT>
static int launch_bwd(int B, int T_, int H, const void *w, const void *q, const void *k, const void *v,
                      const void *a, const void *b, const void *dy, const float *s, const float *sa,
                      const BwdOuts<T> &outs, hipStream_t stream) {
    static bool attr_set = false;  // benign race: idempotent
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&wkv7_bwd_kernel<T, RT>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)kBwdSmemBytes);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    (void)hipGetLastError();  // drop any stale error left by an earlier runtime call of the host program
    hipLaunchKernelGGL((wkv7_bwd_kernel<T, RT>), dim3(B * H * (4 / RT)), dim3(256), kBwdSmemBytes, stream, T_, H,
                       (const T *)w, (const T *)q, (const T *)k, (const T *)v, (const T *)a, (const T *)b,
                       (const T *)dy, s, sa, outs);
    return (int)hipGetLastError();
}

template <typename T>
static int bwd_full(int B, int T_, int H, const void *w, const void *q, const void *k, const void *v, const void *a,
                    const void *b, const void *dy, const float *s, const float *sa, void *dw, void *dq, void *dk,
                    void *dv, void *da, void *db, hipStream_t stream) {
    BwdOuts<T> o;
    o.dw[0] = o.dw[1] = (T *)dw; o.dq[0] = o.dq[1] = (T *)dq; o.dk[0] = o.dk[1] = (T *)dk;
    o.db[0] = o.db[1] = (T *)db; o.da[0] = o.da[1] = (T *)da; o.dv = (T *)dv;
    return launch_bwd<T, 4>(B, T_, H, w, q, k, v, a, b, dy, s, sa, o, stream);
}
template <typename T>
static int bwd_split(int B, int T_, int H, const void *w, const void *q, const void *k, const void *v, const void *a,
                     const void *b, const void *dy, const float *s, const float *sa, void *const *dw, void *const *dq,
                     void *const *dk, void *dv, void *const *da, void *const *db, hipStream_t stream) {
    BwdOuts<T> o;
    for (int i = 0; i < 2; i++) {
        o.dw[i] = (T *)dw[i]; o.dq[i] = (T *)dq[i]; o.dk[i] = (T *)dk[i]; o.db[i] = (T *)db[i]; o.da[i] = (T *)da[i];
    }
    o.dv = (T *)dv;
    return launch_bwd<T, 2>(B, T_, H, w, q, k, v, a, b, dy, s, sa, o, stream);
}

int wkv_bwd_bf16(int B, int T_, int H, const void *w, const void *q, const void *k, const void *v, const void *a,
                 const void *b, const void *dy, const float *s, const float *sa, void *dw, void *dq, void *dk,
                 void *dv, void *da, void *db, hipStream_t stream) {
    return bwd_full<bf16_t>(B, T_, H, w, q, k, v, a, b, dy, s, sa, dw, dq, dk, dv, da, db, stream);
}
int wkv_bwd_f32(int B, int T_, int H, const void *w, const void *q, const void *k, const void *v, const void *a,
                const void *b, const void *dy, const float *s, const float *sa, void *dw, void *dq, void *dk,
                void *dv, void *da, void *db, hipStream_t stream) {
    return bwd_full<float>(B, T_, H, w, q, k, v, a, b, dy, s, sa, dw, dq, dk, dv, da, db, stream);
}
int wkv_bwd_split_bf16(int B, int T_, int H, const void *w, const void *q, const void *k, const void *v, const void *a,
                       const void *b, const void *dy, const float *s, const float *sa, void *const *dw,
                       void *const *dq, void *const *dk, void *dv, void *const *da, void *const *db, hipStream_t st) {
    return bwd_split<bf16_t>(B, T_, H, w, q, k, v, a, b, dy, s, sa, dw, dq, dk, dv, da, db, st);
}
int wkv_bwd_split_f32(int B, int T_, int H, const void *w, const void *q, const void *k, const void *v, const void *a,
                      const void *b, const void *dy, const float *s, const float *sa, void *const *dw, void *const *dq,
                      void *const *dk, void *dv, void *const *da, void *const *db, hipStream_t st) {
    return bwd_split<float>(B, T_, H, w, q, k, v, a, b, dy, s, sa, dw, dq, dk, dv, da, db, st);
}

}  // namespace rwkv7
