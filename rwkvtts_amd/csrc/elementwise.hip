// rwkvtts_amd/csrc/elementwise.hip -- the HBM-bound stages around the WKV7 scan, fused for gfx950.
//
// What they replace (reference: model/llm/rwkv_s2s_single_ffn.py, ~20 separate elementwise launches per layer,
// each re-reading [B,T,D] activations from HBM):
//   mix        :162-169 / :225-227   token shift + 6 (time-mix) or 1 (channel-mix) lerps, with the mask multiply (:160)
//   tmix_prepare :172-190            decay soft-clamp, masks, value residual, a/kk/k' and the scan's a,b operands
//   tmix_post  :192-195              GroupNorm over each head + (r.k.r_k) v bonus + gate
//   relusq     :228                  relu(x)^2
// each with its backward.  Layout: activations are [rows = B*T, D] row-major; one workgroup walks rows with
// D/8 threads, each thread owning 8 consecutive channels (one 16-byte bf16 load/store per tensor per row --
// the coalescing sweet spot, cdna_hip_programming.md G13); a head is 64 channels = 8 consecutive lanes, so all
// per-head reductions (l2 norm, GroupNorm moments, bonus dot product) are 8-lane DPP sums, no LDS.
// Parameter gradients are accumulated per thread over the rows a workgroup walks and written as per-workgroup
// partials [nblocks, P, D] (fp32); the host sums the partials (deterministic, no atomics).
#include "wkv7_common.h"

namespace rwkv7 {

// One thread per 8 columns of a row: D <= 4096.  Declaring the bound lets the register allocator use 256 VGPRs
// (the default assumes 1024-thread blocks = 128 VGPRs, which made mix_bwd<6> spill 392 bytes of scratch per lane
// and serialised its loads: 1.1 TB/s).
constexpr int kEwMaxThreads = 512;

// two fp32 -> packed bf16 pair (low half = a), round-to-nearest-even in ONE v_cvt_pk_bf16_f32 (gfx950) instead of the ~14 integer
// instructions of two f2bf(): the row-stream kernels below round 8-24 values per thread and row
typedef __bf16 pkbf2_t __attribute__((ext_vector_type(2)));
typedef float pkf2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_bf16(float a, float b) {
    const pkf2_t v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, pkbf2_t));
}

template <typename T>
struct V8;
template <>
struct V8<bf16_t> {
    static __device__ __forceinline__ void ld(const bf16_t *p, float (&f)[8]) {
        const uint4 r = *reinterpret_cast<const uint4 *>(p);
        f[0] = __uint_as_float(r.x << 16); f[1] = __uint_as_float(r.x & 0xffff0000u);
        f[2] = __uint_as_float(r.y << 16); f[3] = __uint_as_float(r.y & 0xffff0000u);
        f[4] = __uint_as_float(r.z << 16); f[5] = __uint_as_float(r.z & 0xffff0000u);
        f[6] = __uint_as_float(r.w << 16); f[7] = __uint_as_float(r.w & 0xffff0000u);
    }
    static __device__ __forceinline__ void st(bf16_t *p, const float (&f)[8]) {
        uint4 r;
        r.x = pk_bf16(f[0], f[1]);
        r.y = pk_bf16(f[2], f[3]);
        r.z = pk_bf16(f[4], f[5]);
        r.w = pk_bf16(f[6], f[7]);
        *reinterpret_cast<uint4 *>(p) = r;
    }
    static __device__ __forceinline__ float ld1(const bf16_t *p) { return bf2f(p->x); }
};
template <>
struct V8<float> {
    static __device__ __forceinline__ void ld(const float *p, float (&f)[8]) {
        const float4 a = *reinterpret_cast<const float4 *>(p), b = *reinterpret_cast<const float4 *>(p + 4);
        f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
    }
    static __device__ __forceinline__ void st(float *p, const float (&f)[8]) {
        *reinterpret_cast<float4 *>(p) = make_float4(f[0], f[1], f[2], f[3]);
        *reinterpret_cast<float4 *>(p + 4) = make_float4(f[4], f[5], f[6], f[7]);
    }
    static __device__ __forceinline__ float ld1(const float *p) { return *p; }
};

// the 16 (bf16) / 32 (fp32) bytes of 8 consecutive channels as they come from memory: loaded early, converted at the use
template <typename T>
struct Raw8;
template <>
struct Raw8<bf16_t> {
    uint4 r;
    __device__ __forceinline__ void load(const bf16_t *p) { r = *reinterpret_cast<const uint4 *>(p); }
    __device__ __forceinline__ void get(float (&f)[8]) const {
        f[0] = __uint_as_float(r.x << 16); f[1] = __uint_as_float(r.x & 0xffff0000u);
        f[2] = __uint_as_float(r.y << 16); f[3] = __uint_as_float(r.y & 0xffff0000u);
        f[4] = __uint_as_float(r.z << 16); f[5] = __uint_as_float(r.z & 0xffff0000u);
        f[6] = __uint_as_float(r.w << 16); f[7] = __uint_as_float(r.w & 0xffff0000u);
    }
};
template <>
struct Raw8<float> {
    float4 a, b;
    __device__ __forceinline__ void load(const float *p) {
        a = *reinterpret_cast<const float4 *>(p);
        b = *reinterpret_cast<const float4 *>(p + 4);
    }
    __device__ __forceinline__ void get(float (&f)[8]) const {
        f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
    }
};

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }

// ------------------------------------------------------------------------------------------------------
// mix: out_i[t] = xm[t] + (xm[t-1] - xm[t]) * p_i,   xm = x * mask,   xm[-1] = x_prev (or 0)
// ------------------------------------------------------------------------------------------------------
template <typename T, int NMIX>
__global__ __launch_bounds__(kEwMaxThreads) void mix_fwd_kernel(int B, int T_, int D, const T *__restrict__ x, const T *__restrict__ x_prev,
                               const T *__restrict__ mask, const T *__restrict__ params, T *__restrict__ out) {
    const int c = threadIdx.x * 8;
    const long rows = (long)B * T_;
    float p[NMIX][8];
#pragma unroll
    for (int i = 0; i < NMIX; i++) V8<T>::ld(params + (long)i * D + c, p[i]);
    // (round 4) both rows and both mask values are loaded unconditionally at the top -- a load behind `if (t > 0)` / `if (mask)` is
    // waited for at its issue -- and dropped by a multiply; only the carried-state row x_prev (inference) stays behind its branch
    const bool has_mask = mask != nullptr;
    const T *const maskq = has_mask ? mask : params;
    for (long row = blockIdx.x; row < rows; row += gridDim.x) {
        const int t = (int)(row % T_);
        const long rp = row > 0 ? row - 1 : 0;
        Raw8<T> qc, qp;
        qc.load(x + row * D + c);
        qp.load(x + rp * D + c);
        const float mc0 = V8<T>::ld1(maskq + (has_mask ? row : 0)), mp0 = V8<T>::ld1(maskq + (has_mask ? rp : 0));
        float xc[8], xp[8];
        qc.get(xc);
        qp.get(xp);
        const float mc = has_mask ? mc0 : 1.f, mp = (has_mask ? mp0 : 1.f) * (t > 0 ? 1.f : 0.f);
#pragma unroll
        for (int j = 0; j < 8; j++) {
            xc[j] *= mc;
            xp[j] *= mp;
        }
        if (t == 0 && x_prev) V8<T>::ld(x_prev + (row / T_) * D + c, xp);
#pragma unroll
        for (int i = 0; i < NMIX; i++) {
            float o[8];
#pragma unroll
            for (int j = 0; j < 8; j++) o[j] = fmaf(xp[j] - xc[j], p[i][j], xc[j]);
            V8<T>::st(out + ((long)i * rows + row) * D + c, o);
        }
    }
}

// dxm[t] = sum_i g_i[t] (1 - p_i) + sum_i g_i[t+1] p_i ; dx = dxm * mask ; dp_i = sum_rows g_i[t] (xm[t-1] - xm[t])
// Each workgroup walks a CONTIGUOUS range of rows downwards, so g_i[t+1] and xm[t-1] are carried in registers from
// one row to the next (7 row loads per row instead of 14).  The nmix gradients are separate tensors (one per
// output of the forward), passed as an array of pointers: no stacking copy on the autograd side.
template <int NMIX>
struct MixGrads {
    const void *g[NMIX];
};

// Rows are walked in short descending runs (RUN rows: the t+1 gradients and x[t-1] are carried in registers inside a
// run), and run j of workgroup b is run b + j * gridDim: at any moment the workgroups together stream one contiguous
// window of the 8 tensors.  (Giving each workgroup ONE long contiguous range instead -- 2048 ranges 32 KB apart --
// measured 1.15 TB/s: thousands of concurrent 2 KB streams defeat the DRAM row buffers.)
template <typename T, int NMIX>
__global__ __launch_bounds__(kEwMaxThreads) void mix_bwd_kernel(int B, int T_, int D, int run_len, MixGrads<NMIX> gs, const T *__restrict__ x,
                               const T *__restrict__ x_prev, const T *__restrict__ mask, const T *__restrict__ params,
                               T *__restrict__ dx, float *__restrict__ dpart) {
    const int c = threadIdx.x * 8;
    const long rows = (long)B * T_;
    float p[NMIX][8], dp[NMIX][8], gn[NMIX][8];
#pragma unroll
    for (int i = 0; i < NMIX; i++) {
        V8<T>::ld(params + (long)i * D + c, p[i]);
#pragma unroll
        for (int j = 0; j < 8; j++) dp[i][j] = 0.f;
    }
    // Round 4: the loads of a row iteration are unconditional (clamped rows; a NULL mask is read through a valid pointer and
    // dropped) and issued one iteration ahead -- behind conditions hipcc waited for each of them at its issue (see
    // mix_add_ln_bwd_kernel).  Only the carried-state row x_prev (inference) stays behind its branch.
    const bool has_mask = mask != nullptr;
    const T *const maskq = has_mask ? mask : params;
    struct Pre {
        Raw8<T> xp, g[NMIX];
        float m_prev, m_cur;
    };
    auto fetch = [&](long row) {   // x / mask of row - 1, g and mask of row
        Pre f;
        const long rp = row > 0 ? row - 1 : 0;
        f.xp.load(x + rp * D + c);
#pragma unroll
        for (int i = 0; i < NMIX; i++) f.g[i].load(reinterpret_cast<const T *>(gs.g[i]) + row * D + c);
        f.m_prev = V8<T>::ld1(maskq + (has_mask ? rp : 0));
        f.m_cur = V8<T>::ld1(maskq + (has_mask ? row : 0));
        return f;
    };
    for (long r_lo = (long)blockIdx.x * run_len; r_lo < rows; r_lo += (long)gridDim.x * run_len) {
        const long r_hi = r_lo + run_len < rows ? r_lo + run_len : rows;
        Pre A = fetch(r_hi - 1);
        // the carried values: g of row r_hi (the row after this run) if it belongs to the same sequence, xm of row r_hi - 1
        const bool has_next = r_hi < rows && (r_hi % T_) != 0;
        float xc[8];
        {
            const long rn = has_next ? r_hi : r_hi - 1;
            Raw8<T> gr[NMIX], xr;
#pragma unroll
            for (int i = 0; i < NMIX; i++) gr[i].load(reinterpret_cast<const T *>(gs.g[i]) + rn * D + c);
            xr.load(x + (r_hi - 1) * D + c);
            const float m0 = V8<T>::ld1(maskq + (has_mask ? r_hi - 1 : 0));
            const float m = has_mask ? m0 : 1.f, keep = has_next ? 1.f : 0.f;
#pragma unroll
            for (int i = 0; i < NMIX; i++) {
                gr[i].get(gn[i]);
#pragma unroll
                for (int j = 0; j < 8; j++) gn[i][j] *= keep;
            }
            xr.get(xc);
            if (has_mask) {
#pragma unroll
                for (int j = 0; j < 8; j++) xc[j] *= m;
            }
        }
        int t = (int)((r_hi - 1) % T_) + 1;   // one 64-bit division per run
        for (long row = r_hi - 1; row >= r_lo; row--) {
            const Pre Bn = fetch(row > r_lo ? row - 1 : r_lo);
            __builtin_amdgcn_sched_barrier(0);
            t = t > 0 ? t - 1 : T_ - 1;
            float xq[8], xp[8], acc[8];
            A.xp.get(xq);                       // xm[row - 1]: the next iteration's row whatever t is
            if (has_mask) {
#pragma unroll
                for (int j = 0; j < 8; j++) xq[j] *= A.m_prev;
            }
            const float first = t > 0 ? 1.f : 0.f;
#pragma unroll
            for (int j = 0; j < 8; j++) xp[j] = xq[j] * first;
            if (t == 0 && x_prev) V8<T>::ld(x_prev + (row / T_) * D + c, xp);
            const float m = has_mask ? A.m_cur : 1.f;
#pragma unroll
            for (int j = 0; j < 8; j++) acc[j] = 0.f;
#pragma unroll
            for (int i = 0; i < NMIX; i++) {
                float gc[8];
                A.g[i].get(gc);
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    acc[j] = fmaf(gc[j], 1.f - p[i][j], acc[j]);
                    acc[j] = fmaf(gn[i][j], p[i][j], acc[j]);  // g_i[t+1] (zero past the end of the sequence)
                    dp[i][j] = fmaf(gc[j], xp[j] - xc[j], dp[i][j]);
                    gn[i][j] = gc[j] * first;                   // row-1 is the last row of the previous sequence if t == 0
                }
            }
#pragma unroll
            for (int j = 0; j < 8; j++) {
                acc[j] *= m;
                xc[j] = xq[j];
            }
            V8<T>::st(dx + row * D + c, acc);
            A = Bn;
        }
    }
#pragma unroll
    for (int i = 0; i < NMIX; i++) V8<float>::st(dpart + ((long)blockIdx.x * NMIX + i) * D + c, dp[i]);
}

// ------------------------------------------------------------------------------------------------------
// tmix_prepare (rwkv_s2s_single_ffn.py:172-190)
// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float softplus_(float u) { return u > 20.f ? u : log1pf(__expf(u)); }

template <typename T>
__global__ __launch_bounds__(kEwMaxThreads) void tmix_prepare_fwd_kernel(long rows, int D, const T *__restrict__ w_pre, const T *__restrict__ k,
                                        const T *__restrict__ v, const T *__restrict__ a_pre,
                                        const T *__restrict__ v_pre, const T *__restrict__ v_first,
                                        const T *__restrict__ mask, const T *__restrict__ k_k,
                                        const T *__restrict__ k_a, T *__restrict__ w_out, T *__restrict__ k_out,
                                        T *__restrict__ v_out, T *__restrict__ a_out, T *__restrict__ b_out) {
    const int c = threadIdx.x * 8;
    float kk_p[8], ka_p[8];
    V8<T>::ld(k_k + c, kk_p);
    V8<T>::ld(k_a + c, ka_p);
    for (long row = blockIdx.x; row < rows; row += gridDim.x) {
        const long o = row * D + c;
        const float m = mask ? V8<T>::ld1(mask + row) : 1.f;
        float z[8], kx[8], vx[8], ap[8];
        V8<T>::ld(w_pre + o, z);
        V8<T>::ld(k + o, kx);
        V8<T>::ld(v + o, vx);
        V8<T>::ld(a_pre + o, ap);
        float wo[8], ko[8], vo[8], ao[8], bo[8], kkr[8];
        float ss = 0.f;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            wo[j] = (-softplus_(-z[j]) - 0.5f) * m;
            kx[j] *= m;
            vx[j] *= m;
            kkr[j] = kx[j] * kk_p[j];
            ss = fmaf(kkr[j], kkr[j], ss);
        }
        if (v_pre) {
            float vp[8], vf[8];
            V8<T>::ld(v_pre + o, vp);
            V8<T>::ld(v_first + o, vf);
#pragma unroll
            for (int j = 0; j < 8; j++) vx[j] = fmaf(vf[j] - vx[j], sigmoidf_(vp[j]), vx[j]);
        }
        ss = sum8(ss);
        const float inv = m / fmaxf(sqrtf(ss), 1e-12f);  // F.normalize eps, then kk * mask
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const float a = sigmoidf_(ap[j]);
            const float kk = kkr[j] * inv;
            ko[j] = kx[j] * fmaf(a - 1.f, ka_p[j], 1.f);
            vo[j] = vx[j] * m;
            ao[j] = -kk;
            bo[j] = kk * a;
        }
        V8<T>::st(w_out + o, wo);
        V8<T>::st(k_out + o, ko);
        V8<T>::st(v_out + o, vo);
        V8<T>::st(a_out + o, ao);
        V8<T>::st(b_out + o, bo);
    }
}

// Extra addends of the fused time-mix backward (rwkvtts_amd/fused.py:_TmixCore): the row-split WKV7 backward
// leaves two partial sets of dw,dk,da,db (and dq), and tmix_post's backward contributes to k2, v2 and r.  They are
// summed here in fp32 on load instead of by separate elementwise add kernels.
template <typename T>
struct PrepBwdExtra {
    const T *d_w_b, *d_k2_b, *d_k2_c, *d_v2_b, *d_a_b, *d_b_b;  // added to d_w, d_k2, d_k2, d_v2, d_ain, d_bin
    const T *d_r_a, *d_r_b, *d_r_c;                              // d_r = a + b + c
    T *d_r;
    const T *d_vfirst_in;   // may be NULL: what the layers after this one contributed to d v_first, added before d_vfirst is stored
    // compact hand-off from tmix_post's backward (round 4; all NULL = the three full tensors d_k2_c, d_v2_b, d_r_c above): the bonus
    // term's contributions are rank-1 per head -- d_v2 += dt * dot_h, d_k2 += ds_h r r_k, d_r += ds_h k2 r_k -- so the post backward
    // writes ONE tensor (dt = dL/d(GroupNorm + bonus)) and two scalars per (row, head) instead of three tensors, and they are
    // rebuilt here from r (one more stream in) and k2 (recomputed from k, a, k_a as the forward did): 3 x [rows, D] fewer streams
    const T *dt, *r, *r_k;
    const float *hscal;     // [rows][H][2]: dot_h = sum_head r k2 r_k, ds_h = sum_head dt v2
};

template <typename T, bool MULTI>
__global__ __launch_bounds__(kEwMaxThreads) void tmix_prepare_bwd_kernel(long rows, int D, const T *__restrict__ w_pre, const T *__restrict__ k,
                                        const T *__restrict__ v, const T *__restrict__ a_pre,
                                        const T *__restrict__ v_pre, const T *__restrict__ v_first,
                                        const T *__restrict__ mask, const T *__restrict__ k_k,
                                        const T *__restrict__ k_a, const T *__restrict__ d_w,
                                        const T *__restrict__ d_k2, const T *__restrict__ d_v2,
                                        const T *__restrict__ d_ain, const T *__restrict__ d_bin,
                                        T *__restrict__ d_wpre, T *__restrict__ d_k, T *__restrict__ d_v,
                                        T *__restrict__ d_apre, T *__restrict__ d_vpre, T *__restrict__ d_vfirst,
                                        float *__restrict__ dpart /* [nblk][5][D]: dk_k, dk_a, column sums of d_wpre, d_apre, d_vpre */,
                                        PrepBwdExtra<T> ex) {
    const int c = threadIdx.x * 8;
    float kk_p[8], ka_p[8], dkk_acc[8], dka_acc[8], sw_acc[8], sa_acc[8], sv_acc[8];
    V8<T>::ld(k_k + c, kk_p);
    V8<T>::ld(k_a + c, ka_p);
#pragma unroll
    for (int j = 0; j < 8; j++) dkk_acc[j] = dka_acc[j] = sw_acc[j] = sa_acc[j] = sv_acc[j] = 0.f;
    for (long row = blockIdx.x; row < rows; row += gridDim.x) {
        const long o = row * D + c;
        const float m = mask ? V8<T>::ld1(mask + row) : 1.f;
        float z[8], kx[8], vx[8], ap[8], gw[8], gk2[8], gv2[8], ga[8], gb[8];
        V8<T>::ld(w_pre + o, z);
        V8<T>::ld(k + o, kx);
        V8<T>::ld(v + o, vx);
        V8<T>::ld(a_pre + o, ap);
        V8<T>::ld(d_w + o, gw);
        V8<T>::ld(d_k2 + o, gk2);
        V8<T>::ld(d_v2 + o, gv2);
        V8<T>::ld(d_ain + o, ga);
        V8<T>::ld(d_bin + o, gb);
        if constexpr (MULTI) {
            // second partial set: present after the row-split scan backward, absent (NULL) after the chunked one
            auto add = [&](const T *p, float (&acc)[8]) {
                if (p) {
                    float t[8];
                    V8<T>::ld(p + o, t);
#pragma unroll
                    for (int j = 0; j < 8; j++) acc[j] += t[j];
                }
            };
            add(ex.d_w_b, gw);
            add(ex.d_k2_b, gk2);
            add(ex.d_k2_c, gk2);
            add(ex.d_v2_b, gv2);
            add(ex.d_a_b, ga);
            add(ex.d_b_b, gb);
            float r3[8];
            V8<T>::ld(ex.d_r_a + o, r3);
            add(ex.d_r_b, r3);
            add(ex.d_r_c, r3);
            if (ex.dt) {   // compact hand-off: the bonus term's contributions, rebuilt (a head = this thread's 8 channels' group of 8 lanes)
                float dtv[8], rr[8], rk[8];
                V8<T>::ld(ex.dt + o, dtv);
                V8<T>::ld(ex.r + o, rr);
                V8<T>::ld(ex.r_k + c, rk);
                const float2 hs = *reinterpret_cast<const float2 *>(ex.hscal + (row * (D >> 6) + (threadIdx.x >> 3)) * 2);
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const float k2 = kx[j] * m * fmaf(sigmoidf_(ap[j]) - 1.f, ka_p[j], 1.f);   // what tmix_prepare's forward stored
                    gv2[j] = fmaf(dtv[j], hs.x, gv2[j]);
                    gk2[j] = fmaf(hs.y * rr[j], rk[j], gk2[j]);
                    r3[j] = fmaf(hs.y * k2, rk[j], r3[j]);
                }
            }
            V8<T>::st(ex.d_r + o, r3);
        }
        float kkr[8], a[8], du[8], u[8], o1[8], o2[8];
        float ss = 0.f;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            kx[j] *= m;
            vx[j] *= m;
            kkr[j] = kx[j] * kk_p[j];
            ss = fmaf(kkr[j], kkr[j], ss);
            a[j] = sigmoidf_(ap[j]);
        }
        ss = sum8(ss);
        const float rn = 1.0f / fmaxf(sqrtf(ss), 1e-12f);
        float dot = 0.f;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            u[j] = kkr[j] * rn;                            // unit vector; kk = u * m
            du[j] = (gb[j] * a[j] - ga[j]) * m;            // dL/du
            dot = fmaf(du[j], u[j], dot);
        }
        dot = sum8(dot);
        // d_wpre
#pragma unroll
        for (int j = 0; j < 8; j++) {
            o1[j] = gw[j] * m * sigmoidf_(-z[j]);
            sw_acc[j] += o1[j];   // the low-rank branches' bias gradients are these column sums
        }
        V8<T>::st(d_wpre + o, o1);
        // d_k, d_apre, parameter partials
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const float dkkr = (du[j] - u[j] * dot) * rn;
            const float kk = u[j] * m;
            const float da = gk2[j] * kx[j] * ka_p[j] + gb[j] * kk;
            const float dkx = gk2[j] * fmaf(a[j] - 1.f, ka_p[j], 1.f) + dkkr * kk_p[j];
            dkk_acc[j] = fmaf(dkkr, kx[j], dkk_acc[j]);
            dka_acc[j] = fmaf(gk2[j] * kx[j], a[j] - 1.f, dka_acc[j]);
            o1[j] = dkx * m;
            o2[j] = da * a[j] * (1.f - a[j]);
            sa_acc[j] += o2[j];
        }
        V8<T>::st(d_k + o, o1);
        V8<T>::st(d_apre + o, o2);
        // value path
        if (v_pre) {
            float vp[8], vf[8], o3[8];
            V8<T>::ld(v_pre + o, vp);
            V8<T>::ld(v_first + o, vf);
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const float s = sigmoidf_(vp[j]);
                const float g2 = gv2[j] * m;
                o1[j] = g2 * (1.f - s) * m;              // d_v
                o2[j] = g2 * (vf[j] - vx[j]) * s * (1.f - s);  // d_vpre
                o3[j] = g2 * s;                          // d_vfirst
                sv_acc[j] += o2[j];
            }
            if (MULTI && ex.d_vfirst_in) {
                float vin[8];
                V8<T>::ld(ex.d_vfirst_in + o, vin);
#pragma unroll
                for (int j = 0; j < 8; j++) o3[j] += vin[j];
            }
            V8<T>::st(d_v + o, o1);
            V8<T>::st(d_vpre + o, o2);
            V8<T>::st(d_vfirst + o, o3);
        } else {
#pragma unroll
            for (int j = 0; j < 8; j++) o1[j] = gv2[j] * m * m;
            V8<T>::st(d_v + o, o1);
        }
    }
    V8<float>::st(dpart + ((long)blockIdx.x * 5 + 0) * D + c, dkk_acc);
    V8<float>::st(dpart + ((long)blockIdx.x * 5 + 1) * D + c, dka_acc);
    V8<float>::st(dpart + ((long)blockIdx.x * 5 + 2) * D + c, sw_acc);
    V8<float>::st(dpart + ((long)blockIdx.x * 5 + 3) * D + c, sa_acc);
    V8<float>::st(dpart + ((long)blockIdx.x * 5 + 4) * D + c, sv_acc);
}

// Round 4: the SAME arithmetic for the configuration bf16 training runs (chunked scan backward: one gradient set; compact hand-off
// from tmix_post's backward) with every load of a row issued at the top of the iteration, unconditionally.  In the general kernel
// above each optional addend sits behind `if (pointer)`, the loads behind it are merged with "no value" by a phi, and hipcc waits
// for every load right behind its issue: the ISA showed 18 global loads each followed by s_waitcnt vmcnt(0) -- eighteen serialized
// HBM latencies per row on a kernel with eight waves per CU.  Optional tensors that remain (mask, d_vfirst_in) are read through a
// valid substitute pointer and dropped by a multiply; HAS_V (layers >= 1) is a template parameter.
template <typename T, bool HAS_V>
__global__ __launch_bounds__(kEwMaxThreads) void tmix_prepare_bwd_fast_kernel(
    long rows, int D, const T *__restrict__ w_pre, const T *__restrict__ k, const T *__restrict__ v, const T *__restrict__ a_pre,
    const T *__restrict__ v_pre, const T *__restrict__ v_first, const T *__restrict__ mask, const T *__restrict__ k_k,
    const T *__restrict__ k_a, const T *__restrict__ d_w, const T *__restrict__ d_k2, const T *__restrict__ d_v2,
    const T *__restrict__ d_ain, const T *__restrict__ d_bin, T *__restrict__ d_wpre, T *__restrict__ d_k, T *__restrict__ d_v,
    T *__restrict__ d_apre, T *__restrict__ d_vpre, T *__restrict__ d_vfirst, float *__restrict__ dpart, const T *__restrict__ d_r_a,
    T *__restrict__ d_r, const T *__restrict__ d_vfirst_in, const T *__restrict__ dt_, const T *__restrict__ r_,
    const T *__restrict__ r_k, const float *__restrict__ hscal) {
    const int c = threadIdx.x * 8;
    float kk_p[8], ka_p[8], rk[8], dkk_acc[8], dka_acc[8], sw_acc[8], sa_acc[8], sv_acc[8];
    V8<T>::ld(k_k + c, kk_p);
    V8<T>::ld(k_a + c, ka_p);
    V8<T>::ld(r_k + c, rk);
#pragma unroll
    for (int j = 0; j < 8; j++) dkk_acc[j] = dka_acc[j] = sw_acc[j] = sa_acc[j] = sv_acc[j] = 0.f;
    const bool has_mask = mask != nullptr, has_vin = d_vfirst_in != nullptr;
    const T *const maskq = has_mask ? mask : k_k;
    const T *const vinq = has_vin ? d_vfirst_in : d_w;
    for (long row = blockIdx.x; row < rows; row += gridDim.x) {
        const long o = row * D + c;
        Raw8<T> q_z, q_k, q_v, q_ap, q_gw, q_gk2, q_gv2, q_ga, q_gb, q_r3, q_dt, q_rr, q_vp, q_vf, q_vin;
        q_z.load(w_pre + o); q_k.load(k + o); q_v.load(v + o); q_ap.load(a_pre + o);
        q_gw.load(d_w + o); q_gk2.load(d_k2 + o); q_gv2.load(d_v2 + o); q_ga.load(d_ain + o); q_gb.load(d_bin + o);
        q_r3.load(d_r_a + o); q_dt.load(dt_ + o); q_rr.load(r_ + o);
        if constexpr (HAS_V) {
            q_vp.load(v_pre + o); q_vf.load(v_first + o); q_vin.load(vinq + o);
        }
        const float2 hs = *reinterpret_cast<const float2 *>(hscal + (row * (D >> 6) + (threadIdx.x >> 3)) * 2);
        const float m0 = V8<T>::ld1(maskq + (has_mask ? row : 0));
        __builtin_amdgcn_sched_barrier(0);
        const float m = has_mask ? m0 : 1.f;
        float z[8], kx[8], vx[8], ap[8], gw[8], gk2[8], gv2[8], ga[8], gb[8], r3[8];
        q_z.get(z); q_k.get(kx); q_v.get(vx); q_ap.get(ap); q_gw.get(gw); q_gk2.get(gk2); q_gv2.get(gv2); q_ga.get(ga); q_gb.get(gb);
        q_r3.get(r3);
        {   // compact hand-off: the bonus term's contributions, rebuilt (a head = this thread's 8 channels' group of 8 lanes)
            float dtv[8], rr[8];
            q_dt.get(dtv); q_rr.get(rr);
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const float k2 = kx[j] * m * fmaf(sigmoidf_(ap[j]) - 1.f, ka_p[j], 1.f);   // what tmix_prepare's forward stored
                gv2[j] = fmaf(dtv[j], hs.x, gv2[j]);
                gk2[j] = fmaf(hs.y * rr[j], rk[j], gk2[j]);
                r3[j] = fmaf(hs.y * k2, rk[j], r3[j]);
            }
        }
        V8<T>::st(d_r + o, r3);
        float kkr[8], a[8], du[8], u[8], o1[8], o2[8];
        float ss = 0.f;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            kx[j] *= m;
            vx[j] *= m;
            kkr[j] = kx[j] * kk_p[j];
            ss = fmaf(kkr[j], kkr[j], ss);
            a[j] = sigmoidf_(ap[j]);
        }
        ss = sum8(ss);
        const float rn = 1.0f / fmaxf(sqrtf(ss), 1e-12f);
        float dot = 0.f;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            u[j] = kkr[j] * rn;                            // unit vector; kk = u * m
            du[j] = (gb[j] * a[j] - ga[j]) * m;            // dL/du
            dot = fmaf(du[j], u[j], dot);
        }
        dot = sum8(dot);
#pragma unroll
        for (int j = 0; j < 8; j++) {
            o1[j] = gw[j] * m * sigmoidf_(-z[j]);
            sw_acc[j] += o1[j];   // the low-rank branches' bias gradients are these column sums
        }
        V8<T>::st(d_wpre + o, o1);
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const float dkkr = (du[j] - u[j] * dot) * rn;
            const float kk = u[j] * m;
            const float da = gk2[j] * kx[j] * ka_p[j] + gb[j] * kk;
            const float dkx = gk2[j] * fmaf(a[j] - 1.f, ka_p[j], 1.f) + dkkr * kk_p[j];
            dkk_acc[j] = fmaf(dkkr, kx[j], dkk_acc[j]);
            dka_acc[j] = fmaf(gk2[j] * kx[j], a[j] - 1.f, dka_acc[j]);
            o1[j] = dkx * m;
            o2[j] = da * a[j] * (1.f - a[j]);
            sa_acc[j] += o2[j];
        }
        V8<T>::st(d_k + o, o1);
        V8<T>::st(d_apre + o, o2);
        if constexpr (HAS_V) {
            float vp[8], vf[8], o3[8], vin[8];
            q_vp.get(vp); q_vf.get(vf); q_vin.get(vin);
            const float keep = has_vin ? 1.f : 0.f;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const float s = sigmoidf_(vp[j]);
                const float g2 = gv2[j] * m;
                o1[j] = g2 * (1.f - s) * m;              // d_v
                o2[j] = g2 * (vf[j] - vx[j]) * s * (1.f - s);  // d_vpre
                o3[j] = g2 * s;                          // d_vfirst
                sv_acc[j] += o2[j];
            }
#pragma unroll
            for (int j = 0; j < 8; j++) o3[j] += vin[j] * keep;
            V8<T>::st(d_v + o, o1);
            V8<T>::st(d_vpre + o, o2);
            V8<T>::st(d_vfirst + o, o3);
        } else {
#pragma unroll
            for (int j = 0; j < 8; j++) o1[j] = gv2[j] * m * m;
            V8<T>::st(d_v + o, o1);
        }
    }
    V8<float>::st(dpart + ((long)blockIdx.x * 5 + 0) * D + c, dkk_acc);
    V8<float>::st(dpart + ((long)blockIdx.x * 5 + 1) * D + c, dka_acc);
    V8<float>::st(dpart + ((long)blockIdx.x * 5 + 2) * D + c, sw_acc);
    V8<float>::st(dpart + ((long)blockIdx.x * 5 + 3) * D + c, sa_acc);
    V8<float>::st(dpart + ((long)blockIdx.x * 5 + 4) * D + c, sv_acc);
}

// ------------------------------------------------------------------------------------------------------
// tmix_post (rwkv_s2s_single_ffn.py:192-195): out = (GroupNorm_head(y) + (sum_head r k r_k) v) * g
// ------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(kEwMaxThreads) void tmix_post_fwd_kernel(long rows, int D, const T *__restrict__ y, const T *__restrict__ r,
                                     const T *__restrict__ k, const T *__restrict__ v, const T *__restrict__ g,
                                     const T *__restrict__ gn_w, const T *__restrict__ gn_b,
                                     const T *__restrict__ r_k, float eps, T *__restrict__ out) {
    const int c = threadIdx.x * 8;
    float gw[8], gb[8], rk[8];
    V8<T>::ld(gn_w + c, gw);
    V8<T>::ld(gn_b + c, gb);
    V8<T>::ld(r_k + c, rk);
    for (long row = blockIdx.x; row < rows; row += gridDim.x) {
        const long o = row * D + c;
        float yy[8], rr[8], kk[8], vv[8], gg[8], oo[8];
        V8<T>::ld(y + o, yy);
        V8<T>::ld(r + o, rr);
        V8<T>::ld(k + o, kk);
        V8<T>::ld(v + o, vv);
        V8<T>::ld(g + o, gg);
        float s1 = 0.f, dot = 0.f;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            s1 += yy[j];
            dot = fmaf(rr[j] * kk[j], rk[j], dot);
        }
        const float mean = sum8(s1) * (1.0f / 64.0f);
        dot = sum8(dot);
        float s2 = 0.f;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const float d = yy[j] - mean;
            s2 = fmaf(d, d, s2);
        }
        const float rstd = rsqrtf(sum8(s2) * (1.0f / 64.0f) + eps);
#pragma unroll
        for (int j = 0; j < 8; j++)
            oo[j] = (fmaf((yy[j] - mean) * rstd, gw[j], gb[j]) + dot * vv[j]) * gg[j];
        V8<T>::st(out + o, oo);
    }
}

// COMPACT (round 4): d_v receives dt = dL/d(GroupNorm + bonus) instead of dt * dot, d_r / d_k are not written, and (dot_h, ds_h)
// go to hscal[rows][H][2]: tmix_prepare_bwd_kernel rebuilds the three bonus contributions from them (PrepBwdExtra)
template <typename T, bool COMPACT>
__global__ __launch_bounds__(kEwMaxThreads) void tmix_post_bwd_kernel(long rows, int D, const T *__restrict__ dout, const T *__restrict__ y,
                                     const T *__restrict__ r, const T *__restrict__ k, const T *__restrict__ v,
                                     const T *__restrict__ g, const T *__restrict__ gn_w, const T *__restrict__ gn_b,
                                     const T *__restrict__ r_k, float eps, T *__restrict__ d_y, T *__restrict__ d_r,
                                     T *__restrict__ d_k, T *__restrict__ d_v, T *__restrict__ d_g,
                                     float *__restrict__ dpart /* [nblk][3][D]: d gn_w, d gn_b, d r_k */,
                                     float *__restrict__ hscal) {
    const int c = threadIdx.x * 8;
    float gw[8], gb[8], rk[8], a_w[8], a_b[8], a_rk[8];
    V8<T>::ld(gn_w + c, gw);
    V8<T>::ld(gn_b + c, gb);
    V8<T>::ld(r_k + c, rk);
#pragma unroll
    for (int j = 0; j < 8; j++) a_w[j] = a_b[j] = a_rk[j] = 0.f;
    for (long row = blockIdx.x; row < rows; row += gridDim.x) {
        const long o = row * D + c;
        float go[8], yy[8], rr[8], kk[8], vv[8], gg[8];
        V8<T>::ld(dout + o, go);
        V8<T>::ld(y + o, yy);
        V8<T>::ld(r + o, rr);
        V8<T>::ld(k + o, kk);
        V8<T>::ld(v + o, vv);
        V8<T>::ld(g + o, gg);
        float s1 = 0.f, dot = 0.f;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            s1 += yy[j];
            dot = fmaf(rr[j] * kk[j], rk[j], dot);
        }
        const float mean = sum8(s1) * (1.0f / 64.0f);
        dot = sum8(dot);
        float s2 = 0.f;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const float d = yy[j] - mean;
            s2 = fmaf(d, d, s2);
        }
        const float rstd = rsqrtf(sum8(s2) * (1.0f / 64.0f) + eps);
        float xh[8], dt[8], dxh[8], o1[8], o2[8];
        float ds = 0.f, m1 = 0.f, m2 = 0.f;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            xh[j] = (yy[j] - mean) * rstd;
            const float tval = fmaf(xh[j], gw[j], gb[j]) + dot * vv[j];
            o1[j] = go[j] * tval;          // d_g
            dt[j] = go[j] * gg[j];         // dL/d(yn + bonus)
            ds = fmaf(dt[j], vv[j], ds);
            dxh[j] = dt[j] * gw[j];
            m1 += dxh[j];
            m2 = fmaf(dxh[j], xh[j], m2);
            a_w[j] = fmaf(dt[j], xh[j], a_w[j]);
            a_b[j] += dt[j];
        }
        V8<T>::st(d_g + o, o1);
        ds = sum8(ds);
        m1 = sum8(m1) * (1.0f / 64.0f);
        m2 = sum8(m2) * (1.0f / 64.0f);
#pragma unroll
        for (int j = 0; j < 8; j++) {
            o1[j] = rstd * (dxh[j] - m1 - xh[j] * m2);  // d_y
            o2[j] = COMPACT ? dt[j] : dt[j] * dot;      // d_v (COMPACT: dt itself)
        }
        V8<T>::st(d_y + o, o1);
        V8<T>::st(d_v + o, o2);
#pragma unroll
        for (int j = 0; j < 8; j++) {
            o1[j] = ds * kk[j] * rk[j];  // d_r
            o2[j] = ds * rr[j] * rk[j];  // d_k
            a_rk[j] = fmaf(ds, rr[j] * kk[j], a_rk[j]);
        }
        if (COMPACT) {
            if ((threadIdx.x & 7) == 0) *reinterpret_cast<float2 *>(hscal + (row * (D >> 6) + (threadIdx.x >> 3)) * 2) = make_float2(dot, ds);
        } else {
            V8<T>::st(d_r + o, o1);
            V8<T>::st(d_k + o, o2);
        }
    }
    V8<float>::st(dpart + ((long)blockIdx.x * 3 + 0) * D + c, a_w);
    V8<float>::st(dpart + ((long)blockIdx.x * 3 + 1) * D + c, a_b);
    V8<float>::st(dpart + ((long)blockIdx.x * 3 + 2) * D + c, a_rk);
}

// ------------------------------------------------------------------------------------------------------
// relu(x)^2 and its derivative 2 relu(x) dy
// ------------------------------------------------------------------------------------------------------
template <typename T>
__global__ void relusq_fwd_kernel(long n8, const T *__restrict__ x, T *__restrict__ y) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
        float f[8];
        V8<T>::ld(x + i * 8, f);
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const float r = fmaxf(f[j], 0.f);
            f[j] = r * r;
        }
        V8<T>::st(y + i * 8, f);
    }
}
template <typename T>
__global__ void relusq_bwd_kernel(long n8, const T *__restrict__ x, const T *__restrict__ dy, T *__restrict__ dx) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
        float f[8], g[8];
        V8<T>::ld(x + i * 8, f);
        V8<T>::ld(dy + i * 8, g);
#pragma unroll
        for (int j = 0; j < 8; j++) f[j] = 2.f * fmaxf(f[j], 0.f) * g[j];
        V8<T>::st(dx + i * 8, f);
    }
}

// backward of relu(x)^2 from the OUTPUT s = relu(x)^2 (the GEMM with the activation as its epilogue, gemm_relusq.hip, never
// writes x):  dx = dy * 2 relu(x) = dy * 2 sqrt(s)
template <typename T>
__global__ void relusq_bwd_s_kernel(long n8, const T *__restrict__ s, const T *__restrict__ dy, T *__restrict__ dx) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
        float f[8], g[8];
        V8<T>::ld(s + i * 8, f);
        V8<T>::ld(dy + i * 8, g);
#pragma unroll
        for (int j = 0; j < 8; j++) f[j] = 2.f * __builtin_sqrtf(fmaxf(f[j], 0.f)) * g[j];
        V8<T>::st(dx + i * 8, f);
    }
}

// ------------------------------------------------------------------------------------------------------
// host launchers
// ------------------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------------------
// residual add + LayerNorm (block structure of rwkv_s2s_single_ffn.py:262-276: x = x + att(ln1(x)); x = x + ffn(ln2(x)))
//   forward : x1 = x + branch (rounded to T, as the separate add would) ; h = LN(x1) * gamma + beta
//   backward: dx1 = d_resid + LN'(dh)          (d_resid = gradient arriving at x1 through the residual path)
// One workgroup (D/8 threads) per row; row sums go through LDS (one slot per 8-lane group).
// ------------------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ float round_to(float v);
template <>
__device__ __forceinline__ float round_to<bf16_t>(float v) { return __uint_as_float(pk_bf16(v, v) & 0xffff0000u); }
template <>
__device__ __forceinline__ float round_to<float>(float v) { return v; }

// a + b that -ffast-math cannot reassociate: the LayerNorm sums of the one-pass kernels and of the separate stages must come out
// bit for bit the same (tests/test_fused_gpu.py compares the two routes with torch.equal), whatever shape the compiler gives each loop
__device__ __forceinline__ float add_pinned(float a, float b) {
    float r;
    asm("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// red[0 .. ngroups) summed in index order (slots past ngroups hold zeros, red_init: whole float4 groups are read)
__device__ __forceinline__ float sum_slots(const float *red, int ngroups) {
    float t = 0.f;
    for (int i = 0; i < ngroups; i += 4) {
        const float4 v = *reinterpret_cast<const float4 *>(red + i);
        t = add_pinned(add_pinned(add_pinned(add_pinned(t, v.x), v.y), v.z), v.w);
    }
    return t;
}
// sum of v over the D/8 threads of the workgroup; red has D/64 slots; every thread gets the total
__device__ __forceinline__ float block_sum(float v, float *red, int ngroups) {
    v = sum8(v);
    if ((threadIdx.x & 7) == 0) red[threadIdx.x >> 3] = v;
    __syncthreads();
    return sum_slots(red, ngroups);
}
// every reduction slot zero before the first block_sum / block_sum2 (they read whole float4 groups)
template <int ROWS>
__device__ __forceinline__ void red_init(float (*red)[kEwMaxThreads / 8]) {   // red[ROWS][kEwMaxThreads / 8]
    for (int i = threadIdx.x; i < ROWS * (kEwMaxThreads / 8); i += blockDim.x) (&red[0][0])[i] = 0.f;
    __syncthreads();
}

template <typename T>
__global__ __launch_bounds__(kEwMaxThreads) void add_ln_fwd_kernel(long rows, int D, const T *__restrict__ x,
                                                                   const T *__restrict__ branch,
                                                                   const T *__restrict__ gamma,
                                                                   const T *__restrict__ beta, float eps,
                                                                   T *__restrict__ x_out, T *__restrict__ h,
                                                                   float *__restrict__ mean, float *__restrict__ rstd) {
    __shared__ __attribute__((aligned(16))) float red[2][kEwMaxThreads / 8];
    red_init<2>(red);
    const int c = threadIdx.x * 8, ng = D / 64;
    const float inv_d = 1.0f / (float)D;
    float gm[8], bt[8];
    V8<T>::ld(gamma + c, gm);
    if (beta) {
        V8<T>::ld(beta + c, bt);
    } else {
#pragma unroll
        for (int j = 0; j < 8; j++) bt[j] = 0.f;
    }
    const bool has_branch = branch != nullptr;
    const T *const brq = has_branch ? branch : x;   // (round 4) both rows loaded unconditionally: no wait between them
    for (long row = blockIdx.x; row < rows; row += gridDim.x) {
        const long o = row * D + c;
        Raw8<T> qx, qb;
        qx.load(x + o);
        qb.load(brq + o);
        float v[8], b[8];
        qx.get(v);
        qb.get(b);
        const float keep_b = has_branch ? 1.f : 0.f;   // no branch around the use: the compiler would sink the load into it
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = round_to<T>(fmaf(b[j], keep_b, v[j]));   // (x is exact in T: the rounding is the identity without a branch)
        if (has_branch) V8<T>::st(x_out + o, v);
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 8; j++) s += v[j];
        const float mu = block_sum(s, red[0], ng) * inv_d;
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            v[j] -= mu;
            q = fmaf(v[j], v[j], q);
        }
        const float rs = rsqrtf(block_sum(q, red[1], ng) * inv_d + eps);
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = fmaf(v[j] * rs, gm[j], bt[j]);
        V8<T>::st(h + o, v);
        if (threadIdx.x == 0) {
            mean[row] = mu;
            rstd[row] = rs;
        }
    }
}

template <typename T>
__global__ __launch_bounds__(kEwMaxThreads) void add_ln_bwd_kernel(long rows, int D, const T *__restrict__ dh,
                                                                   const T *__restrict__ d_resid,
                                                                   const T *__restrict__ x1,
                                                                   const float *__restrict__ mean,
                                                                   const float *__restrict__ rstd,
                                                                   const T *__restrict__ gamma, T *__restrict__ dx,
                                                                   float *__restrict__ dpart /* [nblk][2][D] */) {
    __shared__ __attribute__((aligned(16))) float red[4][kEwMaxThreads / 8];
    red_init<4>(red);
    const int c = threadIdx.x * 8, ng = D / 64;
    const float inv_d = 1.0f / (float)D;
    float gm[8], dg[8], db[8];
    V8<T>::ld(gamma + c, gm);
#pragma unroll
    for (int j = 0; j < 8; j++) dg[j] = db[j] = 0.f;
    int ph = 0;
    const bool has_dr = d_resid != nullptr;
    const T *const drq = has_dr ? d_resid : dh;   // (round 4) the three rows of an iteration are loaded together, one row ahead
    struct Pre {
        Raw8<T> g, x, r;
        float mu, rs;
    };
    auto fetch = [&](long row) {
        Pre f;
        const long o = row * D + c;
        f.g.load(dh + o);
        f.x.load(x1 + o);
        f.r.load(drq + o);
        f.mu = mean[row];
        f.rs = rstd[row];
        return f;
    };
    Pre A = fetch(blockIdx.x < rows ? blockIdx.x : 0);
    for (long row = blockIdx.x; row < rows; row += gridDim.x, ph ^= 2) {
        const long o = row * D + c;
        const Pre Bn = fetch(row + gridDim.x < rows ? row + gridDim.x : row);
        __builtin_amdgcn_sched_barrier(0);
        float g[8], xh[8];
        A.g.get(g);
        A.x.get(xh);
        const float mu = A.mu, rs = A.rs;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            xh[j] = (xh[j] - mu) * rs;
            dg[j] = fmaf(g[j], xh[j], dg[j]);
            db[j] += g[j];
            g[j] *= gm[j];
            s1 += g[j];
            s2 = fmaf(g[j], xh[j], s2);
        }
        // two sums, one barrier (slots alternate between rows so that a fast wave cannot overwrite a slot a slow
        // wave is still reading)
        s1 = sum8(s1);
        s2 = sum8(s2);
        if ((threadIdx.x & 7) == 0) {
            red[ph][threadIdx.x >> 3] = s1;
            red[ph + 1][threadIdx.x >> 3] = s2;
        }
        __syncthreads();
        float m1 = sum_slots(red[ph], ng), m2 = sum_slots(red[ph + 1], ng);
        m1 *= inv_d;
        m2 *= inv_d;
        float r[8];
        A.r.get(r);
        const float keep_r = has_dr ? 1.f : 0.f;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            r[j] *= keep_r;
            r[j] += rs * (g[j] - m1 - xh[j] * m2);
        }
        V8<T>::st(dx + o, r);
        A = Bn;
    }
    V8<float>::st(dpart + ((long)blockIdx.x * 2 + 0) * D + c, dg);
    V8<float>::st(dpart + ((long)blockIdx.x * 2 + 1) * D + c, db);
}

// ------------------------------------------------------------------------------------------------------
// add + LayerNorm + token-shift mix in one pass (training path, no carried state):
//   x1 = x + branch ; h = LN(x1) ; hm = h * mask ; out_i[t] = hm[t] + (hm[t-1] - hm[t]) p_i      (rwkv_s2s_single_ffn.py:251-259, 160-169, 223-226)
// The separate kernels write h (add_ln_fwd) and read it back (mix_fwd), and in the backward write dh (mix_bwd) and read it back
// (add_ln_bwd): 2 of 11 / 3 of 12 tensor passes over [B*T, D].  h is rounded to the tensor type before it is mixed, as the
// separate path stores it, so both paths produce the same values.  Rows are walked in short runs (run j of workgroup b is run
// b + j gridDim: together the workgroups stream one contiguous window); inside a run the neighbour row is carried in registers,
// at the start of a run it is recomputed from x1's inputs (forward) or from x1 (backward).
// ------------------------------------------------------------------------------------------------------
// two sums over the D/8 threads of the workgroup with ONE barrier; slots alternate (ph = 0 / 2) between consecutive calls so that
// a fast wave cannot overwrite a slot a slow wave is still reading
__device__ __forceinline__ void block_sum2(float &a, float &b, float (*red)[kEwMaxThreads / 8], int ph, int ngroups) {
    a = sum8(a);
    b = sum8(b);
    if ((threadIdx.x & 7) == 0) {
        red[ph][threadIdx.x >> 3] = a;
        red[ph + 1][threadIdx.x >> 3] = b;
    }
    __syncthreads();
    const float ta = sum_slots(red[ph], ngroups), tb = sum_slots(red[ph + 1], ngroups);
    a = ta;
    b = tb;
}

template <typename T, int NMIX>
__global__ __launch_bounds__(kEwMaxThreads) void add_ln_mix_fwd_kernel(int B, int T_, int D, int run_len, const T *__restrict__ x,
                                                                       const T *__restrict__ branch, const T *__restrict__ gamma,
                                                                       const T *__restrict__ beta, float eps, const T *__restrict__ mask,
                                                                       const T *__restrict__ params, T *__restrict__ x_out,
                                                                       T *__restrict__ out, float *__restrict__ mean,
                                                                       float *__restrict__ rstd, T *__restrict__ h_out) {
    // h_out (may be NULL): also store h = LayerNorm(x1) (unmasked), for a backward that runs as the two separate kernels
    __shared__ __attribute__((aligned(16))) float red[4][kEwMaxThreads / 8];
    red_init<4>(red);
    const int c = threadIdx.x * 8, ng = D / 64;
    const long rows = (long)B * T_;
    const float inv_d = 1.0f / (float)D;
    float gm[8], bt[8], p[NMIX][8];
    V8<T>::ld(gamma + c, gm);
    if (beta) {
        V8<T>::ld(beta + c, bt);
    } else {
#pragma unroll
        for (int j = 0; j < 8; j++) bt[j] = 0.f;
    }
#pragma unroll
    for (int i = 0; i < NMIX; i++) V8<T>::ld(params + (long)i * D + c, p[i]);
    int ph = 0;
    // Round 4: unconditional loads issued one row ahead (see mix_add_ln_bwd_kernel below: behind conditions hipcc waited for every load
    // right behind its issue -- three serialized HBM latencies per row of a run)
    const bool has_branch = branch != nullptr, has_mask = mask != nullptr;
    const T *const brq = has_branch ? branch : x;
    const T *const maskq = has_mask ? mask : gamma;
    struct Pre {
        Raw8<T> xv, bv;
        float m;
    };
    auto fetch = [&](long row) {
        Pre f;
        f.xv.load(x + row * D + c);
        f.bv.load(brq + row * D + c);
        f.m = V8<T>::ld1(maskq + (has_mask ? row : 0));
        return f;
    };
    // hm of one row; WRITE: also x1 and the statistics (mean, then the centred squares: two barriers, as add_ln_fwd_kernel)
    auto ln_row = [&](const Pre &f, long row, bool write, float (&hm)[8]) {
        const long o = row * D + c;
        float v[8];
        f.xv.get(v);
        if (has_branch) {
            float b[8];
            f.bv.get(b);
#pragma unroll
            for (int j = 0; j < 8; j++) v[j] = round_to<T>(v[j] + b[j]);
            if (write) V8<T>::st(x_out + o, v);
        }
        float s = 0.f, dummy = 0.f;
#pragma unroll
        for (int j = 0; j < 8; j++) s += v[j];
        block_sum2(s, dummy, red, ph, ng);
        ph ^= 2;
        const float mu = s * inv_d;
        float q = 0.f;
        dummy = 0.f;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            v[j] -= mu;
            q = fmaf(v[j], v[j], q);
        }
        block_sum2(q, dummy, red, ph, ng);
        ph ^= 2;
        const float rs = rsqrtf(q * inv_d + eps);
        const float m = has_mask ? f.m : 1.f;
#pragma unroll
        for (int j = 0; j < 8; j++) hm[j] = round_to<T>(fmaf(v[j] * rs, gm[j], bt[j]));
        if (write && h_out) V8<T>::st(h_out + o, hm);
#pragma unroll
        for (int j = 0; j < 8; j++) hm[j] *= m;
        if (write && threadIdx.x == 0) {
            mean[row] = mu;
            rstd[row] = rs;
        }
    };
    for (long r_lo = (long)blockIdx.x * run_len; r_lo < rows; r_lo += (long)gridDim.x * run_len) {
        const long r_hi = r_lo + run_len < rows ? r_lo + run_len : rows;
        int t = (int)(r_lo % T_);                       // one 64-bit division per run
        const long first = t != 0 ? r_lo - 1 : r_lo;    // the neighbour row is re-normalised at the start of a run
        t = t != 0 ? t - 1 : 0;
        float hp[8];
#pragma unroll
        for (int j = 0; j < 8; j++) hp[j] = 0.f;
        Pre A = fetch(first);
        for (long row = first; row < r_hi; row++) {
            const Pre Bn = fetch(row + 1 < r_hi ? row + 1 : row);
            __builtin_amdgcn_sched_barrier(0);
            const bool write = row >= r_lo;
            float hc[8];
            ln_row(A, row, write, hc);
            if (write) {
                const float keep = t != 0 ? 1.f : 0.f;   // first step of a sequence: shift(x) = 0
#pragma unroll
                for (int j = 0; j < 8; j++) hp[j] *= keep;
#pragma unroll
                for (int i = 0; i < NMIX; i++) {
                    float o[8];
#pragma unroll
                    for (int j = 0; j < 8; j++) o[j] = fmaf(hp[j] - hc[j], p[i][j], hc[j]);
                    V8<T>::st(out + ((long)i * rows + row) * D + c, o);
                }
            }
#pragma unroll
            for (int j = 0; j < 8; j++) hp[j] = hc[j];
            t = t + 1 < T_ ? t + 1 : 0;
            A = Bn;
        }
    }
}

// dhm[t] = sum_i g_i[t] (1 - p_i) + g_i[t+1] p_i ; dh = dhm * mask ; dx = LN'(dh) + d_resid ; dp_i = sum g_i[t] (hm[t-1] - hm[t]) ;
// dgamma = sum dh xhat ; dbeta = sum dh.   dpart [nblocks][NMIX + 2][D]: dp_0.., dgamma, dbeta.
template <typename T, int NMIX>
__global__ __launch_bounds__(kEwMaxThreads) void mix_add_ln_bwd_kernel(int B, int T_, int D, int run_len, MixGrads<NMIX> gs,
                                                                       const T *__restrict__ d_resid, const T *__restrict__ x1,
                                                                       const float *__restrict__ mean, const float *__restrict__ rstd,
                                                                       const T *__restrict__ gamma, const T *__restrict__ beta,
                                                                       const T *__restrict__ mask, const T *__restrict__ params,
                                                                       T *__restrict__ dx, float *__restrict__ dpart) {
    __shared__ __attribute__((aligned(16))) float red[4][kEwMaxThreads / 8];
    red_init<4>(red);
    const int c = threadIdx.x * 8, ng = D / 64;
    const long rows = (long)B * T_;
    const float inv_d = 1.0f / (float)D;
    float gm[8], bt[8], dg[8], db[8], p[NMIX][8], dp[NMIX][8], gn[NMIX][8];
    V8<T>::ld(gamma + c, gm);
    if (beta) {
        V8<T>::ld(beta + c, bt);
    } else {
#pragma unroll
        for (int j = 0; j < 8; j++) bt[j] = 0.f;
    }
#pragma unroll
    for (int j = 0; j < 8; j++) dg[j] = db[j] = 0.f;
#pragma unroll
    for (int i = 0; i < NMIX; i++) {
        V8<T>::ld(params + (long)i * D + c, p[i]);
#pragma unroll
        for (int j = 0; j < 8; j++) dp[i][j] = 0.f;
    }
    // Round 4: every load of a row iteration is UNCONDITIONAL (clamped rows, NULL pointers replaced by a valid one and the value
    // dropped by a select) and issued ONE ITERATION AHEAD.  The first form loaded x1[row - 1], the mask, g and d_resid behind
    // conditions, each merged with "no value" by a phi, so hipcc waited for every one of them right behind its issue (s_waitcnt
    // vmcnt(0) after each global_load in the ISA): four to five serialized HBM latencies per row and run -- the kernel streamed at
    // 2.3 TB/s where the one-row-per-workgroup stages reach 4-5.7 by sheer parallelism.
    const bool has_dr = d_resid != nullptr, has_mask = mask != nullptr;
    const T *const drq = has_dr ? d_resid : x1;
    const T *const maskq = has_mask ? mask : gamma;
    struct Pre {
        Raw8<T> x1p, dr, g[NMIX];
        float mu, rs, m_prev, m_cur;
    };
    auto fetch = [&](long row) {   // what the iteration of `row` needs: x1 / statistics / mask of row - 1, g, d_resid and mask of row
        Pre f;
        const long rp = row > 0 ? row - 1 : 0;
        f.x1p.load(x1 + rp * D + c);
#pragma unroll
        for (int i = 0; i < NMIX; i++) f.g[i].load(reinterpret_cast<const T *>(gs.g[i]) + row * D + c);
        f.dr.load(drq + row * D + c);
        f.mu = mean[rp];
        f.rs = rstd[rp];
        f.m_prev = V8<T>::ld1(maskq + (has_mask ? rp : 0));
        f.m_cur = V8<T>::ld1(maskq + (has_mask ? row : 0));
        return f;
    };
    int ph = 0;
    for (long r_lo = (long)blockIdx.x * run_len; r_lo < rows; r_lo += (long)gridDim.x * run_len) {
        const long r_hi = r_lo + run_len < rows ? r_lo + run_len : rows;
        Pre A = fetch(r_hi - 1);
        // the carried values: g of row r_hi (the row after this run) if it belongs to the same sequence, xhat / hm of row r_hi - 1
        const bool has_next = r_hi < rows && (r_hi % T_) != 0;
        float xc[8], hc[8], rs_c;
        {
            const long rn = has_next ? r_hi : r_hi - 1;
            Raw8<T> gr[NMIX], xr;
#pragma unroll
            for (int i = 0; i < NMIX; i++) gr[i].load(reinterpret_cast<const T *>(gs.g[i]) + rn * D + c);
            xr.load(x1 + (r_hi - 1) * D + c);
            const float mu = mean[r_hi - 1], m0 = V8<T>::ld1(maskq + (has_mask ? r_hi - 1 : 0));
            rs_c = rstd[r_hi - 1];
            const float m = has_mask ? m0 : 1.f, keep = has_next ? 1.f : 0.f;
#pragma unroll
            for (int i = 0; i < NMIX; i++) {
                gr[i].get(gn[i]);
#pragma unroll
                for (int j = 0; j < 8; j++) gn[i][j] *= keep;
            }
            xr.get(xc);
#pragma unroll
            for (int j = 0; j < 8; j++) {
                xc[j] = (xc[j] - mu) * rs_c;
                hc[j] = round_to<T>(fmaf(xc[j], gm[j], bt[j])) * m;
            }
        }
        int t = (int)((r_hi - 1) % T_) + 1;                      // one 64-bit division per run, not per row
        for (long row = r_hi - 1; row >= r_lo; row--) {
            const Pre Bn = fetch(row > r_lo ? row - 1 : r_lo);   // the next iteration's rows (the last one re-reads row r_lo: dropped)
            __builtin_amdgcn_sched_barrier(0);                   // the scheduler would sink these loads to their use
            t = t > 0 ? t - 1 : T_ - 1;
            // row - 1 from what the forward saved (the row before a sequence start is normalised too: it is the next iteration's row)
            float xq[8], hq[8], hp[8];
            A.x1p.get(xq);
            const float mq = has_mask ? A.m_prev : 1.f, first = t > 0 ? 1.f : 0.f;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                xq[j] = (xq[j] - A.mu) * A.rs;
                hq[j] = round_to<T>(fmaf(xq[j], gm[j], bt[j])) * mq;
                hp[j] = hq[j] * first;                           // first step of a sequence: shift(hm) = 0
            }
            const float m = has_mask ? A.m_cur : 1.f;
            float g[8];
#pragma unroll
            for (int j = 0; j < 8; j++) g[j] = 0.f;
#pragma unroll
            for (int i = 0; i < NMIX; i++) {
                float gc[8];
                A.g[i].get(gc);
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    g[j] = fmaf(gc[j], 1.f - p[i][j], g[j]);
                    g[j] = fmaf(gn[i][j], p[i][j], g[j]);     // g_i[t+1] (zero past the end of the sequence)
                    dp[i][j] = fmaf(gc[j], hp[j] - hc[j], dp[i][j]);
                    gn[i][j] = gc[j] * first;                  // row - 1 is the last row of the previous sequence if t == 0
                }
            }
            // dh = g * mask, rounded to the tensor type as the separate mix_bwd kernel stores it
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                g[j] = round_to<T>(g[j] * m);
                dg[j] = fmaf(g[j], xc[j], dg[j]);
                db[j] += g[j];
                g[j] *= gm[j];
                s1 += g[j];
                s2 = fmaf(g[j], xc[j], s2);
            }
            block_sum2(s1, s2, red, ph, ng);
            ph ^= 2;
            s1 *= inv_d;
            s2 *= inv_d;
            float r[8];
            A.dr.get(r);
            const float keep_r = has_dr ? 1.f : 0.f;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                r[j] *= keep_r;
                r[j] += rs_c * (g[j] - s1 - xc[j] * s2);   // (the expression of the first form: same contraction, same bits)
            }
            V8<T>::st(dx + row * D + c, r);
#pragma unroll
            for (int j = 0; j < 8; j++) {
                xc[j] = xq[j];
                hc[j] = hq[j];
            }
            rs_c = A.rs;
            A = Bn;
        }
    }
#pragma unroll
    for (int i = 0; i < NMIX; i++) V8<float>::st(dpart + ((long)blockIdx.x * (NMIX + 2) + i) * D + c, dp[i]);
    V8<float>::st(dpart + ((long)blockIdx.x * (NMIX + 2) + NMIX) * D + c, dg);
    V8<float>::st(dpart + ((long)blockIdx.x * (NMIX + 2) + NMIX + 1) * D + c, db);
}

// ------------------------------------------------------------------------------------------------------
// AdamW on fp32 master weights with bf16 gradients and a bf16 copy of the updated weights in one pass
// (train_scripts/train_spark_rwkv7speech.py:178-197 builds torch.optim.AdamW / DeepSpeed FusedAdam; DeepSpeed's bf16
// optimizer keeps fp32 masters the same way).  Update rule = torch.optim.AdamW (decoupled weight decay):
//   p *= 1 - lr wd ;  m = b1 m + (1-b1) g ;  v = b2 v + (1-b2) g^2 ;  p -= (lr / bc1) m / (sqrt(v) / sqrt(bc2) + eps)
// 4 floats per thread and iteration: 2 + 3*4 bytes read, 3*4 + 2 written per parameter (28 B) -- nothing else touches HBM.
// ------------------------------------------------------------------------------------------------------
// GROUPS: per-slab parameter groups (train_cosy_rwkv7speech_multiple_dataset.py:162-202): slab_group[e / 128] indexes
// group_tab[g] = {lr scale, weight decay}; the trainer's flat buffer aligns every parameter to 128 elements, so a float4 never
// straddles two parameters.  skip_flag (device, may be NULL): != 0 -> the step runs on a ZERO gradient (the reference's
// NaN-loss step, train_spark_rwkv7speech.py:664-687), decided on the device so that the host never waits for the flag.
template <bool GROUPS>
__global__ __launch_bounds__(256) void adamw_kernel(long n4, float *__restrict__ p32, const bf16_t *__restrict__ g16,
                                                    float *__restrict__ m, float *__restrict__ v, bf16_t *__restrict__ p16,
                                                    const uint8_t *__restrict__ slab_group, const float2 *__restrict__ group_tab,
                                                    const float *__restrict__ skip_flag, float lr, float beta1, float beta2,
                                                    float eps, float wd, float inv_bc1, float inv_sqrt_bc2) {
    const bool skip = skip_flag != nullptr && *skip_flag != 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        float lr_i = lr, wd_i = wd;
        if (GROUPS) {
            const float2 g = group_tab[slab_group[i >> 5]];
            lr_i = lr * g.x;
            wd_i = g.y;
        }
        const float decay = 1.f - lr_i * wd_i, step = lr_i * inv_bc1;
        float4 p = reinterpret_cast<float4 *>(p32)[i], mm = reinterpret_cast<float4 *>(m)[i], vv = reinterpret_cast<float4 *>(v)[i];
        float4 g = cvt4(ld4<bf16_t>(g16 + 4 * i, true));
        if (skip) g = make_float4(0.f, 0.f, 0.f, 0.f);
        auto upd = [&](float &pp, float &m1, float &v1, float gg) {
            pp *= decay;
            m1 = fmaf(beta1, m1, (1.f - beta1) * gg);
            v1 = fmaf(beta2, v1, (1.f - beta2) * gg * gg);
            pp -= step * m1 / (sqrtf(v1) * inv_sqrt_bc2 + eps);
        };
        upd(p.x, mm.x, vv.x, g.x); upd(p.y, mm.y, vv.y, g.y); upd(p.z, mm.z, vv.z, g.z); upd(p.w, mm.w, vv.w, g.w);
        reinterpret_cast<float4 *>(p32)[i] = p;
        reinterpret_cast<float4 *>(m)[i] = mm;
        reinterpret_cast<float4 *>(v)[i] = vv;
        st4(p16 + 4 * i, p);
    }
}

int adamw_step(long n, float *p32, const void *g16, float *m, float *v, void *p16, const uint8_t *slab_group, const float *group_tab,
               const float *skip_flag, float lr, float beta1, float beta2, float eps, float wd, float inv_bc1, float inv_sqrt_bc2,
               hipStream_t st) {
    (void)hipGetLastError();
    const long n4 = n / 4;
    const int grid = (int)((n4 + 255) / 256 < 16384 ? (n4 + 255) / 256 : 16384);
    if (slab_group)
        hipLaunchKernelGGL(adamw_kernel<true>, dim3(grid), dim3(256), 0, st, n4, p32, (const bf16_t *)g16, m, v, (bf16_t *)p16,
                           slab_group, reinterpret_cast<const float2 *>(group_tab), skip_flag, lr, beta1, beta2, eps, wd, inv_bc1,
                           inv_sqrt_bc2);
    else
        hipLaunchKernelGGL(adamw_kernel<false>, dim3(grid), dim3(256), 0, st, n4, p32, (const bf16_t *)g16, m, v, (bf16_t *)p16,
                           slab_group, reinterpret_cast<const float2 *>(group_tab), skip_flag, lr, beta1, beta2, eps, wd, inv_bc1,
                           inv_sqrt_bc2);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------
// softmax cross-entropy of one chunk of head logits, forward and backward in one kernel
// (model/llm/spark_llm.py:146-160 uses rwkvfla's FusedLinearCrossEntropyLoss; the chunking over tokens lives in
// rwkvtts_amd/losses.py).  One wave per row: online max / sum-of-exp over the row (bf16 logits as the GEMM wrote them,
// no fp32 copy), then the row is rewritten IN PLACE with d loss / d logits = (softmax - onehot(label)) * scale, or 0
// for ignored rows; loss_rows[row] = (logsumexp - logit[label]) * valid.  Two reads (the second from L2) + one write.
// ------------------------------------------------------------------------------------------------------
// Round 4: the row is walked in 16-byte pieces (8 logits per lane and load; V is odd for the Spark head -- 8 193 -- so a row starts
// on any 2-byte boundary: up to 7 single elements in front of the first aligned piece and behind the last), one running-max rescale per
// piece instead of one per element (9 exponentials per 8 logits instead of 16), the gradients leave as 16-byte stores through
// v_cvt_pk_bf16_f32.  The first form read and wrote 2 bytes per lane and instruction: 98 us per 4 096 x 8 193 chunk.
// label smoothing ls (torch.nn.CrossEntropyLoss(label_smoothing), xy_llm.py:233-240): loss = (1 - ls)(lse - x[label]) + ls (lse - mean x),
// d loss / d x_j = softmax_j - ls / V - (1 - ls) [j == label]; ls = 0 is the plain form, bit for bit what it was.
// ld: elements between rows (>= V; round 6: the Spark head's logits live in a buffer padded to a multiple of 256 columns, so rows are 16-byte
// aligned and the head GEMMs see aligned leading dimensions; the padding columns are neither read nor written here)
__global__ __launch_bounds__(256) void ce_fwd_bwd_kernel(long rows, int V, long ld, bf16_t *__restrict__ logits, const long *__restrict__ labels,
                                                         long ignore_index, float scale, float *__restrict__ loss_rows, float ls) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    uint16_t *x = reinterpret_cast<uint16_t *>(logits) + row * ld;
    const long lab = labels[row];
    const bool valid = lab != ignore_index;
    // [0, head) singles, [head, head + 8 nv) aligned pieces, [head + 8 nv, V) singles
    int head = (int)(((16 - (reinterpret_cast<uintptr_t>(x) & 15)) & 15) >> 1);
    head = head < V ? head : V;
    const int nv = (V - head) >> 3, tail0 = head + 8 * nv;
    const uint4 *xv = reinterpret_cast<const uint4 *>(x + head);
    float m = -INFINITY, ssum = 0.f, xsum = 0.f;   // xsum: sum of the logits (the smoothing term)
    auto one = [&](float v) {
        const float mn = fmaxf(m, v);
        ssum = ssum * __expf(m - mn) + __expf(v - mn);
        m = mn;
        xsum += v;
    };
    if (lane < head) one(bf2f(x[lane]));
    if (tail0 + lane < V) one(bf2f(x[tail0 + lane]));
    for (int i = lane; i < nv; i += 64) {
        const uint4 r = xv[i];
        float f[8];
        f[0] = __uint_as_float(r.x << 16); f[1] = __uint_as_float(r.x & 0xffff0000u);
        f[2] = __uint_as_float(r.y << 16); f[3] = __uint_as_float(r.y & 0xffff0000u);
        f[4] = __uint_as_float(r.z << 16); f[5] = __uint_as_float(r.z & 0xffff0000u);
        f[6] = __uint_as_float(r.w << 16); f[7] = __uint_as_float(r.w & 0xffff0000u);
        float mx = fmaxf(fmaxf(fmaxf(f[0], f[1]), fmaxf(f[2], f[3])), fmaxf(fmaxf(f[4], f[5]), fmaxf(f[6], f[7])));
        const float mn = fmaxf(m, mx);
        float e = 0.f;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            e += __expf(f[j] - mn);
            xsum += f[j];
        }
        ssum = ssum * __expf(m - mn) + e;
        m = mn;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        xsum += __shfl_xor(xsum, off);
        const float mo = __shfl_xor(m, off), so = __shfl_xor(ssum, off);
        const float mn = fmaxf(m, mo);
        // a lane that saw no element carries (m, ssum) = (-inf, 0): exp(-inf - mn) = 0 keeps it out, and two such lanes give exp(nan) * 0
        ssum = (m == mn ? ssum : ssum * __expf(m - mn)) + (mo == mn ? so : so * __expf(mo - mn));
        m = mn;
    }
    const float lse = m + __logf(ssum);
    if (lane == 0) {
        const float nll = lse - bf2f(x[lab < 0 ? 0 : lab]);
        loss_rows[row] = valid ? (ls > 0.f ? (1.f - ls) * nll + ls * (lse - xsum / (float)V) : nll) : 0.f;
    }
    const float sc = valid ? scale : 0.f, hit = 1.f - ls, off_all = ls / (float)V;
    auto grad1 = [&](int j) {
        const float p = __expf(bf2f(x[j]) - lse) - off_all - (j == lab ? hit : 0.f);
        x[j] = (uint16_t)pk_bf16(p * sc, 0.f);
    };
    if (lane < head) grad1(lane);
    if (tail0 + lane < V) grad1(tail0 + lane);
    uint4 *xw = reinterpret_cast<uint4 *>(x + head);
    const int labv = (int)lab - head;   // position of the label among the aligned elements (anything outside [0, 8 nv): no hit)
    for (int i = lane; i < nv; i += 64) {
        const uint4 r = xv[i];
        float f[8];
        f[0] = __uint_as_float(r.x << 16); f[1] = __uint_as_float(r.x & 0xffff0000u);
        f[2] = __uint_as_float(r.y << 16); f[3] = __uint_as_float(r.y & 0xffff0000u);
        f[4] = __uint_as_float(r.z << 16); f[5] = __uint_as_float(r.z & 0xffff0000u);
        f[6] = __uint_as_float(r.w << 16); f[7] = __uint_as_float(r.w & 0xffff0000u);
#pragma unroll
        for (int j = 0; j < 8; j++) f[j] = (__expf(f[j] - lse) - off_all - (8 * i + j == labv ? hit : 0.f)) * sc;
        xw[i] = make_uint4(pk_bf16(f[0], f[1]), pk_bf16(f[2], f[3]), pk_bf16(f[4], f[5]), pk_bf16(f[6], f[7]));
    }
}

int ce_fwd_bwd(long rows, int V, long ld, void *logits, const long *labels, long ignore_index, float scale, float *loss_rows,
               float label_smoothing, hipStream_t st) {
    (void)hipGetLastError();
    hipLaunchKernelGGL(ce_fwd_bwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, rows, V, ld, (bf16_t *)logits, labels,
                       ignore_index, scale, loss_rows, label_smoothing);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------
// out[n] (bf16) = (accumulate ? out[n] : 0) + sum_s parts[s][n] (fp32): the last step of the split weight gradient
// (fused.wgrad_splitk) written straight into the gradient buffer -- replaces reduce + cast + accumulate launches
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sum_slabs_kernel(long n4, int S, const float *__restrict__ parts, bf16_t *__restrict__ out,
                                                        int accumulate) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        if (accumulate) a = cvt4(ld4<bf16_t>(out + i * 4, true));
        for (int s0 = 0; s0 < S; s0 += 4) {
            float4 t[4];
#pragma unroll
            for (int s = 0; s < 4; s++)
                t[s] = s0 + s < S ? *reinterpret_cast<const float4 *>(parts + ((long)(s0 + s) * n4 + i) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int s = 0; s < 4; s++) {
                a.x += t[s].x; a.y += t[s].y; a.z += t[s].z; a.w += t[s].w;
            }
        }
        st4(out + i * 4, a);
    }
}

// many slabs over a short vector (the low-rank weight gradients: S = 64 slabs of 32K-128K elements): one thread per (float4
// column, quarter of the slabs), eight loads in flight, the four quarters combined through LDS -- the kernel above would walk
// the 64 slabs in 16 dependent rounds of loads with 64 workgroups
__global__ __launch_bounds__(256) void sum_slabs_wide_kernel(long n4, int S, const float *__restrict__ parts, bf16_t *__restrict__ out,
                                                             int accumulate) {
    __shared__ float4 red[3][64];
    const int c = threadIdx.x & 63, g = threadIdx.x >> 6;
    const long i = (long)blockIdx.x * 64 + c;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < n4) {
        for (int s0 = g; s0 < S; s0 += 32) {
            float4 t[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int s = s0 + 4 * u;
                t[u] = s < S ? *reinterpret_cast<const float4 *>(parts + ((long)s * n4 + i) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                a.x += t[u].x; a.y += t[u].y; a.z += t[u].z; a.w += t[u].w;
            }
        }
    }
    if (g > 0) red[g - 1][c] = a;
    __syncthreads();
    if (g == 0 && i < n4) {
#pragma unroll
        for (int u = 0; u < 3; u++) {
            const float4 t = red[u][c];
            a.x += t.x; a.y += t.y; a.z += t.z; a.w += t.w;
        }
        if (accumulate) {
            const float4 o = cvt4(ld4<bf16_t>(out + i * 4, true));
            a.x += o.x; a.y += o.y; a.z += o.z; a.w += o.w;
        }
        st4(out + i * 4, a);
    }
}

// many slabs over a SHORT vector, S in the hundreds or thousands (round 4: the column sums of the fused stages' parameter-gradient
// partials, [S = 1024..2048 workgroup rows][2..8 x D] fp32 -- torch's reduce + cast pair was 15 + 5 us, 122 times per step): a
// workgroup owns 32 columns (one 128-byte line per slab row) and walks ALL slabs, thread = (float4 column, slab lane of 64), 16
// independent loads in flight per thread, then a fixed-order tree over the 64 slab lanes in LDS -- one launch, deterministic
__global__ __launch_bounds__(512) void sum_slabs_tall_kernel(long n4, int S, const float *__restrict__ parts, bf16_t *__restrict__ out,
                                                             int accumulate) {
    __shared__ float4 red[64][8];
    const int cl = threadIdx.x & 7, rl = threadIdx.x >> 3;
    const long i = (long)blockIdx.x * 8 + cl;
    const long ic = i < n4 ? i : n4 - 1;   // clamped, never skipped (a guarded load is waited for at its issue)
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int s0 = rl; s0 < S; s0 += 64 * 16) {
        float4 t[16];
#pragma unroll
        for (int u = 0; u < 16; u++) {
            const int s = s0 + 64 * u;
            t[u] = *reinterpret_cast<const float4 *>(parts + ((long)(s < S ? s : S - 1) * n4 + ic) * 4);
        }
#pragma unroll
        for (int u = 0; u < 16; u++) {
            const float m = s0 + 64 * u < S ? 1.f : 0.f;
            a.x = fmaf(t[u].x, m, a.x); a.y = fmaf(t[u].y, m, a.y); a.z = fmaf(t[u].z, m, a.z); a.w = fmaf(t[u].w, m, a.w);
        }
    }
    red[rl][cl] = a;
    __syncthreads();
#pragma unroll
    for (int h = 32; h >= 1; h >>= 1) {
        if (rl < h) {
            const float4 b = red[rl + h][cl];
            a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
            red[rl][cl] = a;
        }
        __syncthreads();
    }
    if (rl == 0 && i < n4) {
        if (accumulate) {
            const float4 o = cvt4(ld4<bf16_t>(out + i * 4, true));
            a.x += o.x; a.y += o.y; a.z += o.z; a.w += o.w;
        }
        st4(out + i * 4, a);
    }
}

int sum_slabs_bf16(long n, int S, const float *parts, void *out, int accumulate, hipStream_t st) {
    (void)hipGetLastError();
    const long n4 = n / 4;
    if (S >= 256 && n4 <= 8192) {
        hipLaunchKernelGGL(sum_slabs_tall_kernel, dim3((int)((n4 + 7) / 8)), dim3(512), 0, st, n4, S, parts, (bf16_t *)out, accumulate);
        return (int)hipGetLastError();
    }
    if (S >= 16 && n4 <= 64L * 4096) {
        hipLaunchKernelGGL(sum_slabs_wide_kernel, dim3((int)((n4 + 63) / 64)), dim3(256), 0, st, n4, S, parts, (bf16_t *)out, accumulate);
        return (int)hipGetLastError();
    }
    const int grid = (int)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096);
    hipLaunchKernelGGL(sum_slabs_kernel, dim3(grid), dim3(256), 0, st, n4, S, parts, (bf16_t *)out, accumulate);
    return (int)hipGetLastError();
}

// out[c][r] = in[r][c] for a row-major [R][C] matrix of 16-bit elements: 64 x 64 tiles through LDS, 16-byte global accesses both ways
// (round 4: the channel-mix backward's NT operand W_value^T, 8 MiB per layer and step -- torch's strided copy took 27 us)
__global__ __launch_bounds__(256) void transpose16_kernel(int R, int C, const uint16_t *__restrict__ in, uint16_t *__restrict__ out) {
    __shared__ uint16_t tile[64][64 + 2];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tr = threadIdx.x >> 3, tc = (threadIdx.x & 7) * 8;   // 32 rows x 8 pieces of 8 elements per pass
#pragma unroll
    for (int p = 0; p < 2; p++) {
        const int r = tr + 32 * p;
        const uint4 v = *reinterpret_cast<const uint4 *>(in + (long)(r0 + r) * C + c0 + tc);
        const uint16_t *e = reinterpret_cast<const uint16_t *>(&v);
#pragma unroll
        for (int j = 0; j < 8; j++) tile[r][tc + j] = e[j];
    }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 2; p++) {
        const int c = tr + 32 * p;   // output row = input column
        uint4 v;
        uint16_t *e = reinterpret_cast<uint16_t *>(&v);
#pragma unroll
        for (int j = 0; j < 8; j++) e[j] = tile[tc + j][c];
        *reinterpret_cast<uint4 *>(out + (long)(c0 + c) * R + r0 + tc) = v;
    }
}
int transpose_bf16(int R, int C, const void *in, void *out, hipStream_t st) {
    (void)hipGetLastError();
    hipLaunchKernelGGL(transpose16_kernel, dim3(C / 64, R / 64), dim3(256), 0, st, R, C, (const uint16_t *)in, (uint16_t *)out);
    return (int)hipGetLastError();
}

// out[r][0..D) = idx[r] >= 0 ? src[idx[r]][0..D) : 0 -- 16-bit elements, D % 8 == 0.  The re-layout of a packed cu_seqlens row into the
// 32-aligned row of RWKV7Model._forward_packed and back (train_spark_rwkv7speech.py:238-239): both directions and both gradients are
// GATHERS (the position maps are injective), one pass each instead of zero-fill + index_copy / index_select + mask multiply.
__global__ __launch_bounds__(256) void gather_rows16_kernel(long n_out, int D8, const uint4 *__restrict__ src, const int *__restrict__ idx,
                                                            uint4 *__restrict__ out) {
    const long total = n_out * D8;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / D8;
        const int c = (int)(i - r * D8);
        const int s_ = idx[r];
        out[i] = s_ >= 0 ? src[(long)s_ * D8 + c] : make_uint4(0u, 0u, 0u, 0u);
    }
}
int gather_rows16(long n_out, int D, const void *src, const int *idx, void *out, hipStream_t st) {
    (void)hipGetLastError();
    const long total = n_out * (D / 8);
    const int grid = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    hipLaunchKernelGGL(gather_rows16_kernel, dim3(grid), dim3(256), 0, st, n_out, D / 8, (const uint4 *)src, idx, (uint4 *)out);
    return (int)hipGetLastError();
}

static inline int finish() { return (int)hipGetLastError(); }

template <typename T>
int mix_fwd(int B, int T_, int D, int nmix, const void *x, const void *x_prev, const void *mask, const void *params,
            void *out, int nblocks, hipStream_t st) {
    (void)hipGetLastError();
    const dim3 grid(nblocks), block(D / 8);
    if (nmix == 6)
        hipLaunchKernelGGL((mix_fwd_kernel<T, 6>), grid, block, 0, st, B, T_, D, (const T *)x, (const T *)x_prev,
                           (const T *)mask, (const T *)params, (T *)out);
    else if (nmix == 3)   // x_r, x_k, x_v only: the low-rank branches take their inputs through the lerp (fused.mix_lora)
        hipLaunchKernelGGL((mix_fwd_kernel<T, 3>), grid, block, 0, st, B, T_, D, (const T *)x, (const T *)x_prev,
                           (const T *)mask, (const T *)params, (T *)out);
    else
        hipLaunchKernelGGL((mix_fwd_kernel<T, 1>), grid, block, 0, st, B, T_, D, (const T *)x, (const T *)x_prev,
                           (const T *)mask, (const T *)params, (T *)out);
    return finish();
}
template <typename T>
int mix_bwd(int B, int T_, int D, int nmix, const void *const *g, const void *x, const void *x_prev, const void *mask,
            const void *params, void *dx, float *dpart, int nblocks, int run_len, hipStream_t st) {
    (void)hipGetLastError();
    const dim3 grid(nblocks), block(D / 8);
    if (nmix == 6) {
        MixGrads<6> gs;
        for (int i = 0; i < 6; i++) gs.g[i] = g[i];
        hipLaunchKernelGGL((mix_bwd_kernel<T, 6>), grid, block, 0, st, B, T_, D, run_len, gs, (const T *)x,
                           (const T *)x_prev,
                           (const T *)mask, (const T *)params, (T *)dx, dpart);
    } else if (nmix == 3) {
        MixGrads<3> gs;
        for (int i = 0; i < 3; i++) gs.g[i] = g[i];
        hipLaunchKernelGGL((mix_bwd_kernel<T, 3>), grid, block, 0, st, B, T_, D, run_len, gs, (const T *)x,
                           (const T *)x_prev,
                           (const T *)mask, (const T *)params, (T *)dx, dpart);
    } else {
        MixGrads<1> gs;
        gs.g[0] = g[0];
        hipLaunchKernelGGL((mix_bwd_kernel<T, 1>), grid, block, 0, st, B, T_, D, run_len, gs, (const T *)x,
                           (const T *)x_prev,
                           (const T *)mask, (const T *)params, (T *)dx, dpart);
    }
    return finish();
}
template <typename T>
int add_ln_mix_fwd(int B, int T_, int D, int nmix, void *h_out, const void *x, const void *branch, const void *gamma, const void *beta, float eps,
                   const void *mask, const void *params, void *x_out, void *out, float *mean, float *rstd, int nblocks, int run_len,
                   hipStream_t st) {
    (void)hipGetLastError();
    const dim3 grid(nblocks), block(D / 8);
    if (nmix == 6)
        hipLaunchKernelGGL((add_ln_mix_fwd_kernel<T, 6>), grid, block, 0, st, B, T_, D, run_len, (const T *)x, (const T *)branch,
                           (const T *)gamma, (const T *)beta, eps, (const T *)mask, (const T *)params, (T *)x_out, (T *)out, mean, rstd, (T *)h_out);
    else if (nmix == 3)   // x_r, x_k, x_v: the time-mix side when the low-rank branches go through the lerp (fused.add_layer_norm_mix_lora)
        hipLaunchKernelGGL((add_ln_mix_fwd_kernel<T, 3>), grid, block, 0, st, B, T_, D, run_len, (const T *)x, (const T *)branch,
                           (const T *)gamma, (const T *)beta, eps, (const T *)mask, (const T *)params, (T *)x_out, (T *)out, mean, rstd, (T *)h_out);
    else
        hipLaunchKernelGGL((add_ln_mix_fwd_kernel<T, 1>), grid, block, 0, st, B, T_, D, run_len, (const T *)x, (const T *)branch,
                           (const T *)gamma, (const T *)beta, eps, (const T *)mask, (const T *)params, (T *)x_out, (T *)out, mean, rstd, (T *)h_out);
    return finish();
}
template <typename T>
int mix_add_ln_bwd(int B, int T_, int D, int nmix, const void *const *g, const void *d_resid, const void *x1, const float *mean,
                   const float *rstd, const void *gamma, const void *beta, const void *mask, const void *params, void *dx, float *dpart,
                   int nblocks, int run_len, hipStream_t st) {
    (void)hipGetLastError();
    const dim3 grid(nblocks), block(D / 8);
    if (nmix == 6) {
        MixGrads<6> gs;
        for (int i = 0; i < 6; i++) gs.g[i] = g[i];
        hipLaunchKernelGGL((mix_add_ln_bwd_kernel<T, 6>), grid, block, 0, st, B, T_, D, run_len, gs, (const T *)d_resid, (const T *)x1,
                           mean, rstd, (const T *)gamma, (const T *)beta, (const T *)mask, (const T *)params, (T *)dx, dpart);
    } else {
        MixGrads<1> gs;
        gs.g[0] = g[0];
        hipLaunchKernelGGL((mix_add_ln_bwd_kernel<T, 1>), grid, block, 0, st, B, T_, D, run_len, gs, (const T *)d_resid, (const T *)x1,
                           mean, rstd, (const T *)gamma, (const T *)beta, (const T *)mask, (const T *)params, (T *)dx, dpart);
    }
    return finish();
}
template <typename T>
int tmix_prepare_fwd(long rows, int D, const void *w_pre, const void *k, const void *v, const void *a_pre,
                     const void *v_pre, const void *v_first, const void *mask, const void *k_k, const void *k_a,
                     void *w, void *k2, void *v2, void *ain, void *bin, int nblocks, hipStream_t st) {
    (void)hipGetLastError();
    hipLaunchKernelGGL((tmix_prepare_fwd_kernel<T>), dim3(nblocks), dim3(D / 8), 0, st, rows, D, (const T *)w_pre,
                       (const T *)k, (const T *)v, (const T *)a_pre, (const T *)v_pre, (const T *)v_first,
                       (const T *)mask, (const T *)k_k, (const T *)k_a, (T *)w, (T *)k2, (T *)v2, (T *)ain, (T *)bin);
    return finish();
}
template <typename T>
int tmix_prepare_bwd(long rows, int D, const void *w_pre, const void *k, const void *v, const void *a_pre,
                     const void *v_pre, const void *v_first, const void *mask, const void *k_k, const void *k_a,
                     const void *d_w, const void *d_k2, const void *d_v2, const void *d_ain, const void *d_bin,
                     void *d_wpre, void *d_k, void *d_v, void *d_apre, void *d_vpre, void *d_vfirst, float *dpart,
                     int nblocks, hipStream_t st) {
    (void)hipGetLastError();
    hipLaunchKernelGGL((tmix_prepare_bwd_kernel<T, false>), dim3(nblocks), dim3(D / 8), 0, st, rows, D,
                       (const T *)w_pre, (const T *)k, (const T *)v, (const T *)a_pre, (const T *)v_pre,
                       (const T *)v_first, (const T *)mask, (const T *)k_k, (const T *)k_a, (const T *)d_w,
                       (const T *)d_k2, (const T *)d_v2, (const T *)d_ain, (const T *)d_bin, (T *)d_wpre, (T *)d_k,
                       (T *)d_v, (T *)d_apre, (T *)d_vpre, (T *)d_vfirst, dpart, PrepBwdExtra<T>{});
    return finish();
}
// gsum: HOST array of 15 device pointers {d_w a,b; d_k2 a,b,c; d_v2 a,b; d_ain a,b; d_bin a,b; d_r a,b,c; d_vfirst_in}
template <typename T>
int tmix_prepare_bwd_sum(long rows, int D, const void *w_pre, const void *k, const void *v, const void *a_pre,
                         const void *v_pre, const void *v_first, const void *mask, const void *k_k, const void *k_a,
                         const void *const *gsum, int ngsum, void *d_wpre, void *d_k, void *d_v, void *d_apre, void *d_vpre,
                         void *d_vfirst, void *d_r, float *dpart, int nblocks, hipStream_t st) {
    (void)hipGetLastError();
    const T *const *g = reinterpret_cast<const T *const *>(gsum);
    // gsum[15..18] (compact hand-off, see PrepBwdExtra): dt, r, r_k, hscal (fp32); ngsum = 15 without them
    PrepBwdExtra<T> ex{g[1], g[3], g[4], g[6], g[8], g[10], g[11], g[12], g[13], (T *)d_r, g[14],
                       ngsum > 15 ? g[15] : nullptr, ngsum > 15 ? g[16] : nullptr, ngsum > 15 ? g[17] : nullptr,
                       ngsum > 15 ? reinterpret_cast<const float *>(gsum[18]) : nullptr};
    const bool one_set = !ex.d_w_b && !ex.d_k2_b && !ex.d_k2_c && !ex.d_v2_b && !ex.d_a_b && !ex.d_b_b && !ex.d_r_b && !ex.d_r_c;
    if (one_set && ex.dt && ex.r && ex.r_k && ex.hscal && ex.d_r_a && (v_pre == nullptr) == (v_first == nullptr)) {
        // what bf16 training launches: one gradient set from the chunked scan backward + the compact hand-off (loads hoisted)
        if (v_pre)
            hipLaunchKernelGGL((tmix_prepare_bwd_fast_kernel<T, true>), dim3(nblocks), dim3(D / 8), 0, st, rows, D, (const T *)w_pre,
                               (const T *)k, (const T *)v, (const T *)a_pre, (const T *)v_pre, (const T *)v_first, (const T *)mask,
                               (const T *)k_k, (const T *)k_a, g[0], g[2], g[5], g[7], g[9], (T *)d_wpre, (T *)d_k, (T *)d_v,
                               (T *)d_apre, (T *)d_vpre, (T *)d_vfirst, dpart, ex.d_r_a, ex.d_r, ex.d_vfirst_in, ex.dt, ex.r, ex.r_k,
                               ex.hscal);
        else
            hipLaunchKernelGGL((tmix_prepare_bwd_fast_kernel<T, false>), dim3(nblocks), dim3(D / 8), 0, st, rows, D, (const T *)w_pre,
                               (const T *)k, (const T *)v, (const T *)a_pre, (const T *)v_pre, (const T *)v_first, (const T *)mask,
                               (const T *)k_k, (const T *)k_a, g[0], g[2], g[5], g[7], g[9], (T *)d_wpre, (T *)d_k, (T *)d_v,
                               (T *)d_apre, (T *)d_vpre, (T *)d_vfirst, dpart, ex.d_r_a, ex.d_r, ex.d_vfirst_in, ex.dt, ex.r, ex.r_k,
                               ex.hscal);
        return finish();
    }
    hipLaunchKernelGGL((tmix_prepare_bwd_kernel<T, true>), dim3(nblocks), dim3(D / 8), 0, st, rows, D,
                       (const T *)w_pre, (const T *)k, (const T *)v, (const T *)a_pre, (const T *)v_pre,
                       (const T *)v_first, (const T *)mask, (const T *)k_k, (const T *)k_a, g[0], g[2], g[5], g[7],
                       g[9], (T *)d_wpre, (T *)d_k, (T *)d_v, (T *)d_apre, (T *)d_vpre, (T *)d_vfirst, dpart, ex);
    return finish();
}
template <typename T>
int tmix_post_fwd(long rows, int D, const void *y, const void *r, const void *k, const void *v, const void *g,
                  const void *gn_w, const void *gn_b, const void *r_k, float eps, void *out, int nblocks,
                  hipStream_t st) {
    (void)hipGetLastError();
    hipLaunchKernelGGL((tmix_post_fwd_kernel<T>), dim3(nblocks), dim3(D / 8), 0, st, rows, D, (const T *)y,
                       (const T *)r, (const T *)k, (const T *)v, (const T *)g, (const T *)gn_w, (const T *)gn_b,
                       (const T *)r_k, eps, (T *)out);
    return finish();
}
template <typename T>
int tmix_post_bwd(long rows, int D, const void *dout, const void *y, const void *r, const void *k, const void *v,
                  const void *g, const void *gn_w, const void *gn_b, const void *r_k, float eps, void *d_y, void *d_r,
                  void *d_k, void *d_v, void *d_g, float *dpart, float *hscal, int nblocks, hipStream_t st) {
    (void)hipGetLastError();
    if (hscal)   // compact: d_v receives dt, d_r / d_k are not written
        hipLaunchKernelGGL((tmix_post_bwd_kernel<T, true>), dim3(nblocks), dim3(D / 8), 0, st, rows, D, (const T *)dout,
                           (const T *)y, (const T *)r, (const T *)k, (const T *)v, (const T *)g, (const T *)gn_w,
                           (const T *)gn_b, (const T *)r_k, eps, (T *)d_y, (T *)d_r, (T *)d_k, (T *)d_v, (T *)d_g,
                           dpart, hscal);
    else
        hipLaunchKernelGGL((tmix_post_bwd_kernel<T, false>), dim3(nblocks), dim3(D / 8), 0, st, rows, D, (const T *)dout,
                           (const T *)y, (const T *)r, (const T *)k, (const T *)v, (const T *)g, (const T *)gn_w,
                           (const T *)gn_b, (const T *)r_k, eps, (T *)d_y, (T *)d_r, (T *)d_k, (T *)d_v, (T *)d_g,
                           dpart, hscal);
    return finish();
}
template <typename T>
int add_ln_fwd(long rows, int D, const void *x, const void *branch, const void *gamma, const void *beta, float eps,
               void *x_out, void *h, float *mean, float *rstd, int nblocks, hipStream_t st) {
    (void)hipGetLastError();
    hipLaunchKernelGGL((add_ln_fwd_kernel<T>), dim3(nblocks), dim3(D / 8), 0, st, rows, D, (const T *)x,
                       (const T *)branch, (const T *)gamma, (const T *)beta, eps, (T *)x_out, (T *)h, mean, rstd);
    return finish();
}
template <typename T>
int add_ln_bwd(long rows, int D, const void *dh, const void *d_resid, const void *x1, const float *mean,
               const float *rstd, const void *gamma, void *dx, float *dpart, int nblocks, hipStream_t st) {
    (void)hipGetLastError();
    hipLaunchKernelGGL((add_ln_bwd_kernel<T>), dim3(nblocks), dim3(D / 8), 0, st, rows, D, (const T *)dh,
                       (const T *)d_resid, (const T *)x1, mean, rstd, (const T *)gamma, (T *)dx, dpart);
    return finish();
}
template <typename T>
int relusq_fwd(long n, const void *x, void *y, hipStream_t st) {
    (void)hipGetLastError();
    const long n8 = n / 8;
    const int grid = (int)((n8 + 255) / 256 < 8192 ? (n8 + 255) / 256 : 8192);
    hipLaunchKernelGGL((relusq_fwd_kernel<T>), dim3(grid), dim3(256), 0, st, n8, (const T *)x, (T *)y);
    return finish();
}
template <typename T>
int relusq_bwd(long n, const void *x, const void *dy, void *dx, hipStream_t st) {
    (void)hipGetLastError();
    const long n8 = n / 8;
    const int grid = (int)((n8 + 255) / 256 < 8192 ? (n8 + 255) / 256 : 8192);
    hipLaunchKernelGGL((relusq_bwd_kernel<T>), dim3(grid), dim3(256), 0, st, n8, (const T *)x, (const T *)dy, (T *)dx);
    return finish();
}
template <typename T>
int relusq_bwd_s(long n, const void *x, const void *dy, void *dx, hipStream_t st) {
    (void)hipGetLastError();
    const long n8 = n / 8;
    const int grid = (int)((n8 + 255) / 256 < 8192 ? (n8 + 255) / 256 : 8192);
    hipLaunchKernelGGL((relusq_bwd_s_kernel<T>), dim3(grid), dim3(256), 0, st, n8, (const T *)x, (const T *)dy, (T *)dx);
    return finish();
}

#define INSTANTIATE(T)                                                                                              \
    template int mix_fwd<T>(int, int, int, int, const void *, const void *, const void *, const void *, void *, int, \
                            hipStream_t);                                                                           \
    template int mix_bwd<T>(int, int, int, int, const void *const *, const void *, const void *, const void *,       \
                            const void *, void *, float *, int, int, hipStream_t);                                                     \
    template int tmix_prepare_fwd<T>(long, int, const void *, const void *, const void *, const void *, const void *, \
                                     const void *, const void *, const void *, const void *, void *, void *, void *, \
                                     void *, void *, int, hipStream_t);                                             \
    template int add_ln_fwd<T>(long, int, const void *, const void *, const void *, const void *, float, void *, void *, \
                               float *, float *, int, hipStream_t);                                                 \
    template int add_ln_bwd<T>(long, int, const void *, const void *, const void *, const float *, const float *,    \
                               const void *, void *, float *, int, hipStream_t);                                     \
    template int add_ln_mix_fwd<T>(int, int, int, int, void *, const void *, const void *, const void *, const void *, float, const void *, \
                                   const void *, void *, void *, float *, float *, int, int, hipStream_t);           \
    template int mix_add_ln_bwd<T>(int, int, int, int, const void *const *, const void *, const void *, const float *, const float *, \
                                   const void *, const void *, const void *, const void *, void *, float *, int, int, hipStream_t); \
    template int tmix_prepare_bwd_sum<T>(long, int, const void *, const void *, const void *, const void *,          \
                                         const void *, const void *, const void *, const void *, const void *,       \
                                         const void *const *, int, void *, void *, void *, void *, void *, void *, void *, \
                                         float *, int, hipStream_t);                                                 \
    template int tmix_prepare_bwd<T>(long, int, const void *, const void *, const void *, const void *, const void *, \
                                     const void *, const void *, const void *, const void *, const void *,          \
                                     const void *, const void *, const void *, const void *, void *, void *, void *, \
                                     void *, void *, void *, float *, int, hipStream_t);                            \
    template int tmix_post_fwd<T>(long, int, const void *, const void *, const void *, const void *, const void *,  \
                                  const void *, const void *, const void *, float, void *, int, hipStream_t);       \
    template int tmix_post_bwd<T>(long, int, const void *, const void *, const void *, const void *, const void *,  \
                                  const void *, const void *, const void *, const void *, float, void *, void *,    \
                                  void *, void *, void *, float *, float *, int, hipStream_t);                               \
    template int relusq_fwd<T>(long, const void *, void *, hipStream_t);                                            \
    template int relusq_bwd<T>(long, const void *, const void *, void *, hipStream_t);                              \
    template int relusq_bwd_s<T>(long, const void *, const void *, void *, hipStream_t);
INSTANTIATE(bf16_t)
INSTANTIATE(float)

}  // namespace rwkv7
