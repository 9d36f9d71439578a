/*
 * oracle/wkv7_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Scalar CPU restatement of the reference's three WKV7 operators, used only as the
 * checker in tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
 * Nothing under rwkvtts_amd/ may import, link or call this file.
 *
 * What it follows (paths relative to /root/reference):
 *   wkv7_fwd_*   : model/llm/cuda/wkv7_cuda.cu:10-52   (forward_kernel)
 *   wkv7_bwd_*   : model/llm/cuda/wkv7_cuda.cu:54-130  (backward_kernel)
 *   wkv7_state_* : model/llm/cuda/rwkv7_state_fwd_fp16.cu:9-57 (kernel_forward; wkv7s.cu is its B=1 twin)
 *
 * Arithmetic is kept in the reference's order: one "thread" i per state row (forward) or
 * per state column (backward), j-loops ascending, fp32 accumulation, bf16 round-to-nearest-even
 * exactly at the reference's store points (wkv7_cuda.cu:42,89,108-111,122).  The only deliberate
 * difference: the reference is built with --use_fast_math (__expf); this file calls expf().
 *
 * Parity pin: the reference ships no golden vectors for this path, and its
 * CUDA cannot run here.  The pin is the reference's own pure-torch recurrence
 * (model/llm/rwkv_s2s_single_ffn.py:482-506, lines 499-502) imported in the authoring container by
 * oracle/pin_against_reference.py; the vectors it produced are committed under tests/golden/.
 *
 * The *_f32 entry points are the same loops with fp32 tensors in and out (no bf16 rounding);
 * they are what the fp32 logit-parity tests use.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define N_ 64          /* head size: -D_C_=64 / -D_N_=64, rwkv_s2s_single_ffn.py:10-12,43 */
#define CHUNK_ 16      /* _CHUNK_LEN_, rwkv_s2s_single_ffn.py:11 */

typedef uint16_t bf16_t;

static inline float bf2f(bf16_t u) {
    uint32_t x = ((uint32_t)u) << 16;
    float f;
    memcpy(&f, &x, 4);
    return f;
}

/* __float2bfloat16_rn */
static inline bf16_t f2bf(float f) {
    uint32_t x;
    memcpy(&x, &f, 4);
    if ((x & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((x >> 16) | 0x0040u); /* quiet NaN */
    uint32_t lsb = (x >> 16) & 1u;
    x += 0x7fffu + lsb;
    return (bf16_t)(x >> 16);
}

/* ------------------------------------------------------------------------------------------
 * Generic loops, instantiated for bf16 and fp32 I/O through two tiny macros.
 * ------------------------------------------------------------------------------------------ */
#define LD_BF(p, idx) bf2f(((const bf16_t *)(p))[idx])
#define ST_BF(p, idx, val) (((bf16_t *)(p))[idx] = f2bf(val))
#define LD_F32(p, idx) (((const float *)(p))[idx])
#define ST_F32(p, idx, val) (((float *)(p))[idx] = (val))

#define DEFINE_FWD(NAME, LD, ST)                                                                  \
    int NAME(int B, int T, int H, const void *w_, const void *q_, const void *k_, const void *v_, \
             const void *a_, const void *b_, void *y_, float *s_, float *sa_) {                   \
        const int C = N_;                                                                         \
        if (T % CHUNK_ != 0) return -1; /* wkv7_cuda.cu:136, rwkv_s2s_single_ffn.py:19 */        \
        for (int bb = 0; bb < B; bb++)                                                            \
            for (int hh = 0; hh < H; hh++) {                                                      \
                float state[N_][N_]; /* state[i][j]: row i = value index, col j = key index */    \
                memset(state, 0, sizeof(state));                                                  \
                float q[N_], k[N_], w[N_], a[N_], b[N_];                                          \
                for (int t = 0; t < T; t++) {                                                     \
                    const long base = ((long)bb * T * H + (long)t * H + hh) * C;                  \
                    for (int i = 0; i < C; i++) {                                                 \
                        q[i] = LD(q_, base + i);                                                  \
                        w[i] = expf(-expf(LD(w_, base + i))); /* :21 */                           \
                        k[i] = LD(k_, base + i);                                                  \
                        a[i] = LD(a_, base + i);                                                  \
                        b[i] = LD(b_, base + i);                                                  \
                    }                                                                             \
                    for (int i = 0; i < C; i++) {                                                 \
                        float sa = 0; /* :27-32 */                                                \
                        for (int j = 0; j < C; j++) sa += a[j] * state[i][j];                     \
                        if (sa_) sa_[base + i] = sa;                                              \
                        const float v = LD(v_, base + i);                                         \
                        float y = 0; /* :35-42 */                                                 \
                        for (int j = 0; j < C; j++) {                                             \
                            float s = state[i][j];                                                \
                            s = s * w[j] + sa * b[j] + k[j] * v;                                  \
                            state[i][j] = s;                                                      \
                            y += s * q[j];                                                        \
                        }                                                                         \
                        ST(y_, base + i, y);                                                      \
                    }                                                                             \
                    if (s_ && (t + 1) % CHUNK_ == 0) { /* :44-50, transposed store */             \
                        const long sb = (((long)bb * H + hh) * (T / CHUNK_) + t / CHUNK_) * C * C; \
                        for (int i = 0; i < C; i++)                                               \
                            for (int j = 0; j < C; j++) s_[sb + (long)j * C + i] = state[i][j];   \
                    }                                                                             \
                }                                                                                 \
            }                                                                                     \
        return 0;                                                                                 \
    }

DEFINE_FWD(wkv7_fwd_bf16, LD_BF, ST_BF)
DEFINE_FWD(wkv7_fwd_f32, LD_F32, ST_F32)

/*
 * Backward, wkv7_cuda.cu:54-130.  Per reference thread i:
 *   stateT[j] = S[j][i] (column i), dstate[j] = dS[i][j] (row i), dstateT[j] = dS[j][i] (column i).
 * Restated with full matrices S (row=value,col=key) and dS; every sum keeps the reference's
 * j-ascending order, so results are the reference's up to expf-vs-__expf.
 */
#define DEFINE_BWD(NAME, LD, ST)                                                                    \
    int NAME(int B, int T, int H, const void *w_, const void *q_, const void *k_, const void *v_,   \
             const void *a_, const void *b_, const void *dy_, const float *s_, const float *sa_,    \
             void *dw_, void *dq_, void *dk_, void *dv_, void *da_, void *db_) {                    \
        const int C = N_;                                                                           \
        if (T % CHUNK_ != 0) return -1;                                                             \
        float(*S)[N_] = malloc(sizeof(float) * N_ * N_);                                            \
        float(*dS)[N_] = malloc(sizeof(float) * N_ * N_);                                           \
        for (int bb = 0; bb < B; bb++)                                                              \
            for (int hh = 0; hh < H; hh++) {                                                        \
                memset(S, 0, sizeof(float) * N_ * N_);                                              \
                memset(dS, 0, sizeof(float) * N_ * N_);                                             \
                float w[N_], wfac[N_], q[N_], k[N_], v[N_], a[N_], b[N_], dy[N_], sa[N_], dSb[N_]; \
                for (int t = T - 1; t >= 0; t--) {                                                  \
                    const long base = ((long)bb * T * H + (long)t * H + hh) * C;                    \
                    for (int i = 0; i < C; i++) {                                                   \
                        q[i] = LD(q_, base + i);                                                    \
                        wfac[i] = -expf(LD(w_, base + i)); /* :67 */                                \
                        w[i] = expf(wfac[i]);                                                       \
                        k[i] = LD(k_, base + i);                                                    \
                        a[i] = LD(a_, base + i);                                                    \
                        b[i] = LD(b_, base + i);                                                    \
                        v[i] = LD(v_, base + i);                                                    \
                        dy[i] = LD(dy_, base + i);                                                  \
                        sa[i] = sa_[base + i];                                                      \
                    }                                                                               \
                    if ((t + 1) % CHUNK_ == 0) { /* :76-82: stateT[j] = s_[base + i*C + j] */       \
                        const long sb = (((long)bb * H + hh) * (T / CHUNK_) + t / CHUNK_) * C * C;  \
                        for (int i = 0; i < C; i++)                                                 \
                            for (int j = 0; j < C; j++) S[j][i] = s_[sb + (long)i * C + j];         \
                    }                                                                               \
                    for (int i = 0; i < C; i++) { /* :84-89 dq_i = sum_j S[j][i] dy_j */            \
                        float dq = 0;                                                               \
                        for (int j = 0; j < C; j++) dq += S[j][i] * dy[j];                          \
                        ST(dq_, base + i, dq);                                                      \
                    }                                                                               \
                    for (int i = 0; i < C; i++) { /* :91-97 */                                      \
                        const float iwi = 1.0f / w[i];                                              \
                        for (int j = 0; j < C; j++) {                                               \
                            S[j][i] = (S[j][i] - k[i] * v[j] - b[i] * sa[j]) * iwi;                 \
                            dS[j][i] += q[i] * dy[j]; /* dstateT and dstate are the same matrix */  \
                        }                                                                           \
                    }                                                                               \
                    for (int i = 0; i < C; i++) { /* :99-111 */                                     \
                        float dw = 0, dk = 0, dv = 0, db = 0, dsb = 0;                              \
                        for (int j = 0; j < C; j++) {                                               \
                            dw += dS[j][i] * S[j][i];                                               \
                            dk += dS[j][i] * v[j];                                                  \
                            dv += dS[i][j] * k[j];                                                  \
                            dsb += dS[i][j] * b[j];                                                 \
                            db += dS[j][i] * sa[j];                                                 \
                        }                                                                           \
                        ST(dw_, base + i, dw * w[i] * wfac[i]);                                     \
                        ST(dk_, base + i, dk);                                                      \
                        ST(dv_, base + i, dv);                                                      \
                        ST(db_, base + i, db);                                                      \
                        dSb[i] = dsb;                                                               \
                    }                                                                               \
                    for (int i = 0; i < C; i++) { /* :117-122 da_i = sum_j S[j][i] dSb_j */         \
                        float da = 0;                                                               \
                        for (int j = 0; j < C; j++) da += S[j][i] * dSb[j];                         \
                        ST(da_, base + i, da);                                                      \
                    }                                                                               \
                    for (int i = 0; i < C; i++) /* :124-128 */                                      \
                        for (int j = 0; j < C; j++) dS[i][j] = dS[i][j] * w[j] + dSb[i] * a[j];     \
                }                                                                                   \
            }                                                                                       \
        free(S);                                                                                    \
        free(dS);                                                                                   \
        return 0;                                                                                   \
    }

DEFINE_BWD(wkv7_bwd_bf16, LD_BF, ST_BF)
DEFINE_BWD(wkv7_bwd_f32, LD_F32, ST_F32)

/*
 * State-carrying forward, rwkv7_state_fwd_fp16.cu:9-57: r,w,k,v,a,b,y are [B,T,C] with C = H*64,
 * state is fp32 [B,H,64,64] (row = value index, col = key index, :16), read at entry and written
 * back at exit (:18-21,54-56).  Update order inside the step is s*w + k*v + sa*b (:48).
 */
#define DEFINE_STATE(NAME, LD, ST)                                                                 \
    int NAME(int B, int T, int C, int H, float *state_, const void *r_, const void *w_,            \
             const void *k_, const void *v_, const void *a_, const void *b_, void *y_) {           \
        if (H * N_ != C) return -1; /* rwkv7_state_fwd_fp16.cu:61 */                              \
        for (int bb = 0; bb < B; bb++)                                                             \
            for (int h = 0; h < H; h++) {                                                          \
                float *st = state_ + (long)bb * C * N_ + (long)h * N_ * N_;                        \
                float r[N_], k[N_], w[N_], a[N_], b[N_];                                           \
                for (int t = 0; t < T; t++) {                                                      \
                    const long base = (long)bb * T * C + (long)t * C + (long)h * N_;               \
                    for (int i = 0; i < N_; i++) {                                                 \
                        r[i] = LD(r_, base + i);                                                   \
                        w[i] = expf(-expf(LD(w_, base + i)));                                      \
                        k[i] = LD(k_, base + i);                                                   \
                        a[i] = LD(a_, base + i);                                                   \
                        b[i] = LD(b_, base + i);                                                   \
                    }                                                                              \
                    for (int i = 0; i < N_; i++) {                                                 \
                        float *row = st + (long)i * N_;                                            \
                        float sa = 0;                                                              \
                        for (int j = 0; j < N_; j++) sa += a[j] * row[j];                          \
                        const float vv = LD(v_, base + i);                                         \
                        float y = 0;                                                               \
                        for (int j = 0; j < N_; j++) {                                             \
                            float s = row[j];                                                      \
                            s = s * w[j] + k[j] * vv + sa * b[j];                                  \
                            row[j] = s;                                                            \
                            y += s * r[j];                                                         \
                        }                                                                          \
                        ST(y_, base + i, y);                                                       \
                    }                                                                              \
                }                                                                                  \
            }                                                                                      \
        return 0;                                                                                  \
    }

DEFINE_STATE(wkv7_state_fwd_bf16, LD_BF, ST_BF)
DEFINE_STATE(wkv7_state_fwd_f32, LD_F32, ST_F32)

/* helpers exported for the Python side of the oracle */
void oracle_f32_to_bf16(const float *src, bf16_t *dst, long n) {
    for (long i = 0; i < n; i++) dst[i] = f2bf(src[i]);
}
void oracle_bf16_to_f32(const bf16_t *src, float *dst, long n) {
    for (long i = 0; i < n; i++) dst[i] = bf2f(src[i]);
}
