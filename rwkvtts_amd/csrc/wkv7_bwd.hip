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
#include "launch_attr.h"

namespace rwkv7 {

typedef float f2_t __attribute__((ext_vector_type(2)));

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

// RT = state rows per lane tile, NW = wavefronts per workgroup; a workgroup owns NW*4*RT rows of a head:
//   <4,4> 64 rows, 256 threads: the whole head (reference-schema op)
//   <2,4> 32 rows, 256 threads: two workgroups per head
//   <1,8> 32 rows, 512 threads: two workgroups per head AND two waves per SIMD (measured slower, see g_bwd_split_wide)
// The per-wave column partials are combined every FL = 64/NW steps (16 or 8), which keeps sh_part at 80 KB.
template <typename T, int RT, int NW>
__global__ __launch_bounds__(64 * NW) void wkv7_bwd_kernel(
    int T_, int H, const T *__restrict__ w_, const T *__restrict__ q_, const T *__restrict__ k_,
    const T *__restrict__ v_, const T *__restrict__ a_, const T *__restrict__ b_, const T *__restrict__ dy_,
    const float *__restrict__ s_, const float *__restrict__ sa_, BwdOuts<T> outs) {
    constexpr int ROWS_WG = NW * 4 * RT;  // state rows per workgroup
    constexpr int NSPLIT = kN / ROWS_WG;  // workgroups per head
    constexpr int FL = 64 / NW;           // steps between two write-outs
    constexpr bool WIDE = NW == 8;        // 512 threads: staging and write-out work is split between the halves
    static_assert(NW == 4 || NW == 8, "4 or 8 wavefronts");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float(*sh_cv)[NCV][kN] = reinterpret_cast<float(*)[NCV][kN]>(smem);                       // [kTB][7][64]
    float(*sh_rv)[NRV][kN] = reinterpret_cast<float(*)[NRV][kN]>(smem + kTB * NCV * kN);      // [kTB][3][64]
    float(*sh_part)[FL][NOUT][kN] =
        reinterpret_cast<float(*)[FL][NOUT][kN]>(smem + kTB * (NCV + NRV) * kN);              // [NW][FL][5][64]
    float(*sh_dv)[kN] = reinterpret_cast<float(*)[kN]>(smem + kTB * (NCV + NRV) * kN + NW * FL * NOUT * kN);

    int part = blockIdx.x % NSPLIT, bh = blockIdx.x / NSPLIT;
    if (NSPLIT == 2 && (gridDim.x & 15) == 0) {  // both halves of a head on one XCD (shared L2): block ids g and g + 8
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        bh = (j >> 1) * 8 + xcd;
        part = j & 1;
    }
    const int bb = bh / H, hh = bh - bb * H;
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int til = lane >> 4, tj = lane & 15;
    const int r0l = wave * (4 * RT) + til * RT;  // first row of this lane's tile, inside the workgroup
    const int r0 = part * ROWS_WG + r0l;         // ... inside the head
    const int c0 = tj * 4;

    // staging roles: 16 steps x 16 element quads; with 512 threads the lower half takes w,q,k,a and the upper b,v,dy,sa
    const int st = (tid & 255) >> 4;
    const int sc = (tid & 15) * 4;
    const bool lo_half = !WIDE || tid < 256, hi_half = !WIDE || tid >= 256;

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
        if (lo_half) {
            rw = ld4<T>(w_ + off, true);
            rq = ld4<T>(q_ + off, true);
            rk = ld4<T>(k_ + off, true);
            ra = ld4<T>(a_ + off, true);
        }
        if (hi_half) {
            rb = ld4<T>(b_ + off, true);
            rvv = ld4<T>(v_ + off, true);
            rdy = ld4<T>(dy_ + off, true);
            rsa = *reinterpret_cast<const float4 *>(sa_ + off);
        }
    };
    auto stage = [&]() {
        if (lo_half) {
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
        }
        if (hi_half) {
            *reinterpret_cast<float4 *>(&sh_cv[st][CV_B][sc]) = cvt4(rb);
            *reinterpret_cast<float4 *>(&sh_rv[st][RV_V][sc]) = cvt4(rvv);
            *reinterpret_cast<float4 *>(&sh_rv[st][RV_DY][sc]) = cvt4(rdy);
            *reinterpret_cast<float4 *>(&sh_rv[st][RV_SA][sc]) = rsa;
        }
    };
    auto ld_rows = [&](const float *p, float (&o)[RT]) {  // RT consecutive floats, RT-aligned
        if constexpr (RT == 4) {
            const float4 x = *reinterpret_cast<const float4 *>(p);
            o[0] = x.x; o[1] = x.y; o[2] = x.z; o[3] = x.w;
        } else if constexpr (RT == 2) {
            const float2 x = *reinterpret_cast<const float2 *>(p);
            o[0] = x.x; o[1] = x.y;
        } else {
            o[0] = *p;
        }
    };

    // one reverse time step; column partials go to slot `fs` of this wave's sh_part.
    // The 4 columns of the lane tile are handled as two float pairs so that the whole state arithmetic is
    // v_pk_fma_f32 / v_pk_mul_f32 (2 fp32 lanes per instruction; the row scalars v, dy, sa, dSb are broadcast
    // through op_sel): 13 packed instructions per row and column pair instead of ~20 scalar ones.
    auto step = [&](const int tt, const int fs) {
        const float4 wt4 = *reinterpret_cast<const float4 *>(&sh_cv[tt][CV_W][c0]);
        const float4 iw4 = *reinterpret_cast<const float4 *>(&sh_cv[tt][CV_IW][c0]);
        const float4 q4 = *reinterpret_cast<const float4 *>(&sh_cv[tt][CV_Q][c0]);
        const float4 k4 = *reinterpret_cast<const float4 *>(&sh_cv[tt][CV_K][c0]);
        const float4 a4 = *reinterpret_cast<const float4 *>(&sh_cv[tt][CV_A][c0]);
        const float4 b4 = *reinterpret_cast<const float4 *>(&sh_cv[tt][CV_B][c0]);
        const f2_t wt[2] = {{wt4.x, wt4.y}, {wt4.z, wt4.w}};
        const f2_t iw[2] = {{iw4.x, iw4.y}, {iw4.z, iw4.w}};
        const f2_t qv[2] = {{q4.x, q4.y}, {q4.z, q4.w}};
        const f2_t kv[2] = {{k4.x, k4.y}, {k4.z, k4.w}};
        const f2_t av[2] = {{a4.x, a4.y}, {a4.z, a4.w}};
        const f2_t bv[2] = {{b4.x, b4.y}, {b4.z, b4.w}};
        float vv[RT], dyv[RT], sav[RT];
        ld_rows(&sh_rv[tt][RV_V][r0], vv);
        ld_rows(&sh_rv[tt][RV_DY][r0], dyv);
        ld_rows(&sh_rv[tt][RV_SA][r0], sav);

        f2_t colp[NOUT][2];  // dq, dw, dk, db, da partial sums over this lane's RT rows, per column pair
        f2_t dvp[RT], dsbp[RT];
#pragma unroll
        for (int r = 0; r < RT; r++) dvp[r] = dsbp[r] = f2_t{0.f, 0.f};
#pragma unroll
        for (int cp = 0; cp < 2; cp++) {
            f2_t dq = {0.f, 0.f}, dw = {0.f, 0.f}, dk = {0.f, 0.f}, db = {0.f, 0.f};
#pragma unroll
            for (int r = 0; r < RT; r++) {
                f2_t &Sx = *reinterpret_cast<f2_t *>(&S[r][2 * cp]);
                f2_t &dSx = *reinterpret_cast<f2_t *>(&dS[r][2 * cp]);
                dq = Sx * dyv[r] + dq;
                // un-do step t: S_{t-1} = (S_t - v k^T - sa b^T) / w~
                Sx = (Sx - kv[cp] * vv[r] - bv[cp] * sav[r]) * iw[cp];
                dSx = qv[cp] * dyv[r] + dSx;
                dw = dSx * Sx + dw;
                dk = dSx * vv[r] + dk;
                db = dSx * sav[r] + db;
                dvp[r] = dSx * kv[cp] + dvp[r];
                dsbp[r] = dSx * bv[cp] + dsbp[r];
            }
            colp[0][cp] = dq;
            colp[1][cp] = dw;
            colp[2][cp] = dk;
            colp[3][cp] = db;
        }
        float dvv[RT], dsb[RT];
#pragma unroll
        for (int r = 0; r < RT; r++) {
            dvv[r] = sum16(dvp[r].x + dvp[r].y);
            dsb[r] = sum16(dsbp[r].x + dsbp[r].y);
        }
#pragma unroll
        for (int cp = 0; cp < 2; cp++) {
            f2_t da = {0.f, 0.f};
#pragma unroll
            for (int r = 0; r < RT; r++) {
                f2_t &Sx = *reinterpret_cast<f2_t *>(&S[r][2 * cp]);
                f2_t &dSx = *reinterpret_cast<f2_t *>(&dS[r][2 * cp]);
                da = Sx * dsb[r] + da;
                dSx = dSx * wt[cp] + av[cp] * dsb[r];
            }
            colp[4][cp] = da;
        }
        // column partials -> wave totals, over the 4 row groups (til) of this wave.  Transposing
        // butterfly: v_permlane32_swap pairs column c with c+2 (lanes <32 keep c, lanes >=32 keep c+2),
        // v_permlane16_swap pairs c with c+1 (even DPP rows keep c, odd rows keep c+1): 15 swaps + 15
        // adds for 20 values, and row `til` ends up owning column c0 + til of all 5 outputs -- every lane
        // then stores its 5 totals (64 distinct consecutive addresses per output), no exec-masked store.
#pragma unroll
        for (int o = 0; o < NOUT; o++) {
            const float x0 = swap32_sum(colp[o][0].x, colp[o][1].x);
            const float x1 = swap32_sum(colp[o][0].y, colp[o][1].y);
            sh_part[wave][fs][o][c0 + til] = swap16_sum(x0, x1);
        }
        {
            // dv[r] totals are replicated over the 16 lanes of the DPP row: lane tj stores local row r0l + tj % RT
            const int rs = tj & (RT - 1);
            float d = dvv[0];
#pragma unroll
            for (int r = 1; r < RT; r++) d = rs == r ? dvv[r] : d;
            sh_dv[fs][r0l + rs] = d;
        }
    };

    // write-out of FL steps starting at stage step tb: sum the NW wave partials of a column quad.  256 threads take
    // part; with FL = 8 they form two groups (outputs dq,dw,dk | db,da,dv), with FL = 16 every thread does all six.
    auto write_out = [&](const int t0, const int tb) {
        constexpr int NOG = 16 / FL;  // output groups
        if (tid >= 256) return;
        const int fs = (tid >> 4) % FL, og = (tid >> 4) / FL;
        const int tt = tb + fs;
        const long off_t = head_base + (long)(t0 + tt) * tstride;
        const long off = off_t + sc;
        auto total = [&](const int i) {
            float4 acc = *reinterpret_cast<const float4 *>(&sh_part[0][fs][i][sc]);
#pragma unroll
            for (int wv = 1; wv < NW; wv++) {
                const float4 p = *reinterpret_cast<const float4 *>(&sh_part[wv][fs][i][sc]);
                acc.x += p.x; acc.y += p.y; acc.z += p.z; acc.w += p.w;
            }
            return acc;
        };
        if (NOG == 1 || og == 0) {
            st4(outs.dq[part] + off, total(0));
            float4 o1 = total(1);
            const float4 ws = *reinterpret_cast<const float4 *>(&sh_cv[tt][CV_WS][sc]);
            o1.x *= ws.x; o1.y *= ws.y; o1.z *= ws.z; o1.w *= ws.w;  // dw * w~ * (-exp(w)), wkv7_cuda.cu:108
            st4(outs.dw[part] + off, o1);
            st4(outs.dk[part] + off, total(2));
        }
        if (NOG == 1 || og == 1) {
            st4(outs.db[part] + off, total(3));
            st4(outs.da[part] + off, total(4));
            if (sc < ROWS_WG)
                st4(outs.dv + off_t + part * ROWS_WG + sc, *reinterpret_cast<const float4 *>(&sh_dv[fs][sc]));
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
#pragma unroll 1
        for (int tb = kTB - FL; tb >= 0; tb -= FL) {
#pragma unroll 4
            for (int fs = FL - 1; fs >= 0; fs--) step(tb + fs, fs);
            __syncthreads();
            write_out(t0, tb);
            __syncthreads();
        }
        if (n > 0) {
            stage();
            __syncthreads();
        }
    }
}

template <int NW>
constexpr size_t bwd_smem_bytes() {
    return (size_t)(kTB * (NCV + NRV) * kN + NW * (64 / NW) * NOUT * kN + (64 / NW) * kN) * sizeof(float);
}

template <typename T, int RT, int NW>
static int launch_bwd(int B, int T_, int H, const void *w, const void *q, const void *k, const void *v,
                      const void *a, const void *b, const void *dy, const float *s, const float *sa,
                      const BwdOuts<T> &outs, hipStream_t stream) {
    constexpr size_t smem = bwd_smem_bytes<NW>();
    static DynLdsOnce lds_once;
    if (hipError_t e = lds_once.ensure(reinterpret_cast<const void *>(&wkv7_bwd_kernel<T, RT, NW>), (int)smem); e != hipSuccess) return (int)e;
    (void)hipGetLastError();  // drop any stale error left by an earlier runtime call of the host program
    hipLaunchKernelGGL((wkv7_bwd_kernel<T, RT, NW>), dim3(B * H * (kN / (NW * 4 * RT))), dim3(64 * NW), smem, stream,
                       T_, H, (const T *)w, (const T *)q, (const T *)k, (const T *)v, (const T *)a, (const T *)b,
                       (const T *)dy, s, sa, outs);
    return (int)hipGetLastError();
}

// wide 0: <2,4> (256 threads, default), 1: <1,8> (512 threads; rwkv7_wkv_bwd_split_variant_*).  Measured at B=8,T=4096,H=16:
// <2,4> 2.11 ms, <1,8> 2.56 ms -- with 8 waves the 9 ds_read_b128 per wave and step saturate the LDS return path
// (8 x ~60 cycles per step per CU), so the second wave per SIMD buys nothing.

template <typename T>
static int bwd_full(int B, int T_, int H, const void *w, const void *q, const void *k, const void *v, const void *a,
                    const void *b, const void *dy, const float *s, const float *sa, void *dw, void *dq, void *dk,
                    void *dv, void *da, void *db, hipStream_t stream) {
    BwdOuts<T> o;
    o.dw[0] = o.dw[1] = (T *)dw; o.dq[0] = o.dq[1] = (T *)dq; o.dk[0] = o.dk[1] = (T *)dk;
    o.db[0] = o.db[1] = (T *)db; o.da[0] = o.da[1] = (T *)da; o.dv = (T *)dv;
    return launch_bwd<T, 4, 4>(B, T_, H, w, q, k, v, a, b, dy, s, sa, o, stream);
}
template <typename T>
static int bwd_split(int B, int T_, int H, const void *w, const void *q, const void *k, const void *v, const void *a,
                     const void *b, const void *dy, const float *s, const float *sa, void *const *dw, void *const *dq,
                     void *const *dk, void *dv, void *const *da, void *const *db, int wide, hipStream_t stream) {
    BwdOuts<T> o;
    for (int i = 0; i < 2; i++) {
        o.dw[i] = (T *)dw[i]; o.dq[i] = (T *)dq[i]; o.dk[i] = (T *)dk[i]; o.db[i] = (T *)db[i]; o.da[i] = (T *)da[i];
    }
    o.dv = (T *)dv;
    if (wide) return launch_bwd<T, 1, 8>(B, T_, H, w, q, k, v, a, b, dy, s, sa, o, stream);
    return launch_bwd<T, 2, 4>(B, T_, H, w, q, k, v, a, b, dy, s, sa, o, stream);
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
                       void *const *dq, void *const *dk, void *dv, void *const *da, void *const *db, int wide, hipStream_t st) {
    return bwd_split<bf16_t>(B, T_, H, w, q, k, v, a, b, dy, s, sa, dw, dq, dk, dv, da, db, wide, st);
}
int wkv_bwd_split_f32(int B, int T_, int H, const void *w, const void *q, const void *k, const void *v, const void *a,
                      const void *b, const void *dy, const float *s, const float *sa, void *const *dw, void *const *dq,
                      void *const *dk, void *dv, void *const *da, void *const *db, int wide, hipStream_t st) {
    return bwd_split<float>(B, T_, H, w, q, k, v, a, b, dy, s, sa, dw, dq, dk, dv, da, db, wide, st);
}

}  // namespace rwkv7
