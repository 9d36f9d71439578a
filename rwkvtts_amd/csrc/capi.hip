// rwkvtts_amd/csrc/capi.hip -- extern "C" surface of librwkv7_hip.so (declared in include/rwkv7_hip.h).
// Argument checking mirrors the reference's asserts (wkv7_cuda.cu:136, rwkv7_state_fwd_fp16.cu:61,
// rwkv_s2s_single_ffn.py:19-21) but reports through the return code instead of aborting the process.
#include <hip/hip_runtime.h>
#include <math.h>

#include "../../include/rwkv7_hip.h"

namespace rwkv7 {
int wkv_fwd_bf16(int, int, int, const void *, const void *, const void *, const void *, const void *, const void *,
                 void *, float *, float *, float *, int, hipStream_t);
int wkv_fwd_f32(int, int, int, const void *, const void *, const void *, const void *, const void *, const void *,
                void *, float *, float *, float *, int, hipStream_t);
int wkv_bwd_bf16(int, int, int, const void *, const void *, const void *, const void *, const void *, const void *,
                 const void *, const float *, const float *, void *, void *, void *, void *, void *, void *,
                 hipStream_t);
int wkv_bwd_f32(int, int, int, const void *, const void *, const void *, const void *, const void *, const void *,
                const void *, const float *, const float *, void *, void *, void *, void *, void *, void *,
                hipStream_t);

int wkv_bwd_split_bf16(int, int, int, const void *, const void *, const void *, const void *, const void *, const void *,
                       const void *, const float *, const float *, void *const *, void *const *, void *const *, void *,
                       void *const *, void *const *, int, hipStream_t);
int wkv_bwd_split_f32(int, int, int, const void *, const void *, const void *, const void *, const void *, const void *,
                      const void *, const float *, const float *, void *const *, void *const *, void *const *, void *,
                      void *const *, void *const *, int, hipStream_t);
int chunk_prep_bf16(int, int, int, const void *, const void *, const void *, float *, hipStream_t);
int chunk_prep_f32(int, int, int, const void *, const void *, const void *, float *, hipStream_t);
int chunk_fwd_bf16(int, int, int, const void *, const void *, const void *, const void *, const void *, const void *,
                   const float *, void *, float *, void *, const int *, int, hipStream_t);
int chunk_fwd_f32(int, int, int, const void *, const void *, const void *, const void *, const void *, const void *,
                  const float *, void *, float *, void *, const int *, int, hipStream_t);
int chunk_debug_mma(const float *, const float *, float *, float *, hipStream_t);
int chunk_debug_tr16(const uint16_t *, const int *, uint16_t *, hipStream_t);
int gemv32_bf16(int, int, int, const void *, const void *, const void *, void *, hipStream_t);
int ce_fwd_bwd(long, int, long, void *, const long *, long, float, float *, float, hipStream_t);
int adamw_step(long, float *, const void *, float *, float *, void *, const uint8_t *, const float *, const float *, float, float, float, float,
               float, float, float, hipStream_t);
int lora32_bf16(int, int, int, int, int, const void *, const void *, const void *, const void *, void *, hipStream_t);
int chunk_bseq_bf16(int, int, int, const void *, const void *, const void *, const void *, const void *, const float *, void *,
                    float *, const int *, int, hipStream_t);
int gemm_nt4_bf16(int, int, int, const void *, const void *, void *, const void *, int, hipStream_t);
struct MixLoraDesc {
    int nb;
    int r[4], off[4];
    int act[4];
    const void *w1[4];
    const void *mu[4];
    void *out[4];
    void *out2[4];
};
int mix_lora_wcat_fwd(const MixLoraDesc &, int, int, void *, hipStream_t);
int mix_lora_wcat_bwd(const MixLoraDesc &, int, int, const void *, hipStream_t);
int mix_lora_combine_fwd(const MixLoraDesc &, long, int, int, const void *, const void *, hipStream_t);
int mix_lora_combine_bwd(const MixLoraDesc &, long, int, int, const void *, void *, hipStream_t);
struct LoraDownDesc {
    int nb;
    int r[4], off[4];
    int act[4];
    const void *w1[4];
    const void *mu[4];
    void *out[4];
    int tile0[5];
};
int lora_down_pack(const LoraDownDesc &, int, int, void *, hipStream_t);
int lora_down_fwd(LoraDownDesc &, int, long, int, int, const void *, const void *, const void *, hipStream_t);
int chunk_bwd_out10_bf16(int, int, int, const void *, const void *, const void *, const void *, const void *, const void *, const void *,
                         const void *, const float *, const float *, const void *, void *, void *, void *, void *, void *, void *, hipStream_t);
int sum_slabs_bf16(long, int, const float *, void *, int, hipStream_t);
int transpose_bf16(int, int, const void *, void *, hipStream_t);
int gather_rows16(long, int, const void *, const int *, void *, hipStream_t);
int wgrad_skinny_bf16(long, int, int, int, const void *, const void *, float *, hipStream_t);
int wgrad_mid_bf16(long, int, int, int, const void *, const void *, float *, hipStream_t);
int sample_rows_f32(int, int, const float *, long, const int *, const int *, const int *, const int *, const int *, int, int, int, int, float,
                    float, unsigned long long, const long *, long *, const void *, int, long, hipStream_t);
int ras_step_f32(int, const float *, long *, long *, long *, long *, long, int, float, int, int, float, unsigned long long, hipStream_t);
int xy_frame_step(int, int, int, long, long, long, long, long, const long *, int, int, const long *, long *, long *, long *, long *, long *,
                  unsigned char *, long *, hipStream_t);
int xy_embed_bf16(int, int, int, const void *const *, const long *, void *, hipStream_t);
int decode_layer_ptrs();
size_t decode_workspace_bytes(int, int, int, int, int, int, int, int, int);
int decode_step_bf16(int, int, int, int, int, int, int, int, int, int, float, float, const void *const *, const void *const *, const void *,
                     const void *, const void *, const void *, const void *, float *, void *, int, hipStream_t);
struct bf16_t;
template <typename T> int mix_fwd(int, int, int, int, const void *, const void *, const void *, const void *, void *, int, hipStream_t);
template <typename T> int mix_bwd(int, int, int, int, const void *const *, const void *, const void *, const void *, const void *, void *, float *, int, int, hipStream_t);
template <typename T> int tmix_prepare_fwd(long, int, const void *, const void *, const void *, const void *, const void *, const void *, const void *, const void *, const void *, void *, void *, void *, void *, void *, int, hipStream_t);
template <typename T> int tmix_prepare_bwd(long, int, const void *, const void *, const void *, const void *, const void *, const void *, const void *, const void *, const void *, const void *, const void *, const void *, const void *, const void *, void *, void *, void *, void *, void *, void *, float *, int, hipStream_t);
template <typename T> int tmix_prepare_bwd_sum(long, int, const void *, const void *, const void *, const void *, const void *, const void *, const void *, const void *, const void *, const void *const *, int, void *, void *, void *, void *, void *, void *, void *, float *, int, hipStream_t);
template <typename T> int tmix_post_fwd(long, int, const void *, const void *, const void *, const void *, const void *, const void *, const void *, const void *, float, void *, int, hipStream_t);
template <typename T> int tmix_post_bwd(long, int, const void *, const void *, const void *, const void *, const void *, const void *, const void *, const void *, const void *, float, void *, void *, void *, void *, void *, float *, float *, int, hipStream_t);
template <typename T> int add_ln_fwd(long, int, const void *, const void *, const void *, const void *, float, void *, void *, float *, float *, int, hipStream_t);
template <typename T> int add_ln_bwd(long, int, const void *, const void *, const void *, const float *, const float *, const void *, void *, float *, int, hipStream_t);
template <typename T> int add_ln_mix_fwd(int, int, int, int, void *, const void *, const void *, const void *, const void *, float, const void *, const void *, void *, void *, float *, float *, int, int, hipStream_t);
template <typename T> int mix_add_ln_bwd(int, int, int, int, const void *const *, const void *, const void *, const float *, const float *, const void *, const void *, const void *, const void *, void *, float *, int, int, hipStream_t);
template <typename T> int relusq_fwd(long, const void *, void *, hipStream_t);
template <typename T> int relusq_bwd(long, const void *, const void *, void *, hipStream_t);
template <typename T> int relusq_bwd_s(long, const void *, const void *, void *, hipStream_t);
}  // namespace rwkv7

namespace {
inline bool any_null(std::initializer_list<const void *> ps) {
    for (const void *p : ps)
        if (!p) return true;
    return false;
}
}  // namespace

extern "C" {

const char *rwkv7_version(void) { return "rwkv7_hip 0.1.0 gfx950"; }

#define FWD_BODY(IMPL, CW)                                                                   \
    if (B <= 0 || T <= 0 || H <= 0 || any_null({w, q, k, v, a, b, y})) return RWKV7_EINVAL; \
    if ((s == nullptr) != (sa == nullptr)) return RWKV7_EINVAL;                              \
    if (T % RWKV7_CHUNK_LEN != 0) return RWKV7_ECHUNK;                                       \
    if ((CW) != 0 && (CW) != 4 && (CW) != 8) return RWKV7_ESHAPE;                            \
    return rwkv7::IMPL(B, T, H, w, q, k, v, a, b, y, s, sa, nullptr, CW, (hipStream_t)stream);

int rwkv7_wkv_fwd_bf16(int B, int T, int H, const void *w, const void *q, const void *k, const void *v,
                       const void *a, const void *b, void *y, float *s, float *sa, rwkv7_stream_t stream) {
    FWD_BODY(wkv_fwd_bf16, 0)
}
int rwkv7_wkv_fwd_f32(int B, int T, int H, const void *w, const void *q, const void *k, const void *v,
                      const void *a, const void *b, void *y, float *s, float *sa, rwkv7_stream_t stream) {
    FWD_BODY(wkv_fwd_f32, 0)
}
int rwkv7_wkv_fwd_variant_bf16(int B, int T, int H, const void *w, const void *q, const void *k, const void *v,
                               const void *a, const void *b, void *y, float *s, float *sa, int cols_per_lane, rwkv7_stream_t stream) {
    FWD_BODY(wkv_fwd_bf16, cols_per_lane)
}
int rwkv7_wkv_fwd_variant_f32(int B, int T, int H, const void *w, const void *q, const void *k, const void *v,
                              const void *a, const void *b, void *y, float *s, float *sa, int cols_per_lane, rwkv7_stream_t stream) {
    FWD_BODY(wkv_fwd_f32, cols_per_lane)
}

#define BWD_BODY(IMPL)                                                                                          \
    if (B <= 0 || T <= 0 || H <= 0 || any_null({w, q, k, v, a, b, dy, s, sa, dw, dq, dk, dv, da, db}))         \
        return RWKV7_EINVAL;                                                                                    \
    if (T % RWKV7_CHUNK_LEN != 0) return RWKV7_ECHUNK;                                                          \
    return rwkv7::IMPL(B, T, H, w, q, k, v, a, b, dy, s, sa, dw, dq, dk, dv, da, db, (hipStream_t)stream);

int rwkv7_wkv_bwd_bf16(int B, int T, int H, const void *w, const void *q, const void *k, const void *v,
                       const void *a, const void *b, const void *dy, const float *s, const float *sa, void *dw,
                       void *dq, void *dk, void *dv, void *da, void *db, rwkv7_stream_t stream) {
    BWD_BODY(wkv_bwd_bf16)
}
int rwkv7_wkv_bwd_f32(int B, int T, int H, const void *w, const void *q, const void *k, const void *v,
                      const void *a, const void *b, const void *dy, const float *s, const float *sa, void *dw,
                      void *dq, void *dk, void *dv, void *da, void *db, rwkv7_stream_t stream) {
    BWD_BODY(wkv_bwd_f32)
}

// the reference-schema pair on the chunked (MFMA) kernels: `s` as an arena [hs | T^-1 | e_vk] (include/rwkv7_hip.h)
namespace {
constexpr size_t kArenaRec = (size_t)RWKV7_Q15_REC * 2, kArenaTinv = 32 * 32 * sizeof(float);
}
int rwkv7_wkv_fwd_fast_bf16(int B, int T, int H, const void *w, const void *q, const void *k, const void *v,
                            const void *a, const void *b, void *y, float *s, float *sa, rwkv7_stream_t stream) {
    if (B <= 0 || T <= 0 || H <= 0 || any_null({w, q, k, v, a, b, y, s, sa})) return RWKV7_EINVAL;
    if (T % RWKV7_CHUNK_T != 0) return RWKV7_ECHUNK;
    const size_t n = (size_t)B * H * (T / RWKV7_CHUNK_T);
    char *base = reinterpret_cast<char *>(s);
    float *tinv = reinterpret_cast<float *>(base + n * kArenaRec);
    const int rc = rwkv7::chunk_prep_bf16(B, T, H, w, a, b, tinv, (hipStream_t)stream);
    if (rc != 0) return rc;
    return rwkv7::chunk_fwd_bf16(B, T, H, w, q, k, v, a, b, tinv, y, sa, base, nullptr, 0, (hipStream_t)stream);
}
int rwkv7_wkv_bwd_fast_bf16(int B, int T, int H, const void *w, const void *q, const void *k, const void *v,
                            const void *a, const void *b, const void *dy, float *s, const float *sa, void *dw,
                            void *dq, void *dk, void *dv, void *da, void *db, rwkv7_stream_t stream) {
    if (B <= 0 || T <= 0 || H <= 0 || any_null({w, q, k, v, a, b, dy, s, sa, dw, dq, dk, dv, da, db})) return RWKV7_EINVAL;
    if (T % RWKV7_CHUNK_T != 0) return RWKV7_ECHUNK;
    const size_t n = (size_t)B * H * (T / RWKV7_CHUNK_T);
    char *base = reinterpret_cast<char *>(s);
    const float *tinv = reinterpret_cast<const float *>(base + n * kArenaRec);
    void *e_vk = base + n * (kArenaRec + kArenaTinv);
    float *z = reinterpret_cast<float *>(base + n * (2 * kArenaRec + kArenaTinv));   // fp32 [B,T,H,64]: 8192 B per chunk and head
    const int rc = rwkv7::chunk_bseq_bf16(B, T, H, w, q, a, b, dy, tinv, e_vk, z, nullptr, 0, (hipStream_t)stream);
    if (rc != 0) return rc;
    return rwkv7::chunk_bwd_out10_bf16(B, T, H, w, q, k, v, a, b, dy, base, sa, z, e_vk, dw, dq, dk, dv, da, db, (hipStream_t)stream);
}

#define BWD2_BODY(IMPL, WIDE)                                                                                 \
    if (B <= 0 || T <= 0 || H <= 0 || any_null({w, q, k, v, a, b, dy, s, sa, dw, dq, dk, dv, da, db}))   \
        return RWKV7_EINVAL;                                                                               \
    for (int i = 0; i < 2; i++)                                                                            \
        if (!dw[i] || !dq[i] || !dk[i] || !da[i] || !db[i]) return RWKV7_EINVAL;                           \
    if (T % RWKV7_CHUNK_LEN != 0) return RWKV7_ECHUNK;                                                     \
    return rwkv7::IMPL(B, T, H, w, q, k, v, a, b, dy, s, sa, dw, dq, dk, dv, da, db, (WIDE) ? 1 : 0, (hipStream_t)stream);

int rwkv7_wkv_bwd_split_bf16(int B, int T, int H, const void *w, const void *q, const void *k, const void *v,
                             const void *a, const void *b, const void *dy, const float *s, const float *sa,
                             void *const *dw, void *const *dq, void *const *dk, void *dv, void *const *da,
                             void *const *db, rwkv7_stream_t stream) {
    BWD2_BODY(wkv_bwd_split_bf16, 0)
}
int rwkv7_wkv_bwd_split_f32(int B, int T, int H, const void *w, const void *q, const void *k, const void *v,
                            const void *a, const void *b, const void *dy, const float *s, const float *sa,
                            void *const *dw, void *const *dq, void *const *dk, void *dv, void *const *da,
                            void *const *db, rwkv7_stream_t stream) {
    BWD2_BODY(wkv_bwd_split_f32, 0)
}
int rwkv7_wkv_bwd_split_variant_bf16(int B, int T, int H, const void *w, const void *q, const void *k, const void *v,
                                     const void *a, const void *b, const void *dy, const float *s, const float *sa,
                                     void *const *dw, void *const *dq, void *const *dk, void *dv, void *const *da,
                                     void *const *db, int wide, rwkv7_stream_t stream) {
    BWD2_BODY(wkv_bwd_split_bf16, wide)
}

int rwkv7_wkv_workspace_bytes(int B, int T, int H, size_t *s_bytes, size_t *sa_bytes) {
    if (B <= 0 || T <= 0 || H <= 0 || !s_bytes || !sa_bytes) return RWKV7_EINVAL;
    if (T % RWKV7_CHUNK_LEN != 0) return RWKV7_ECHUNK;
    *s_bytes = (size_t)B * H * (T / RWKV7_CHUNK_LEN) * RWKV7_HEAD_SIZE * RWKV7_HEAD_SIZE * sizeof(float);
    *sa_bytes = (size_t)B * T * H * RWKV7_HEAD_SIZE * sizeof(float);
    return 0;
}

#define STATE_BODY(IMPL, CW)                                                                          \
    if (B <= 0 || T <= 0 || H <= 0 || any_null({state, r, w, k, v, a, b, y})) return RWKV7_EINVAL;   \
    if (H * RWKV7_HEAD_SIZE != C) return RWKV7_EHEAD;                                                 \
    if ((CW) != 0 && (CW) != 4 && (CW) != 8) return RWKV7_ESHAPE;                                     \
    return rwkv7::IMPL(B, T, H, w, r, k, v, a, b, y, nullptr, nullptr, state, CW, (hipStream_t)stream);

int rwkv7_wkv_state_fwd_bf16(int B, int T, int C, int H, float *state, const void *r, const void *w,
                             const void *k, const void *v, const void *a, const void *b, void *y,
                             rwkv7_stream_t stream) {
    STATE_BODY(wkv_fwd_bf16, 0)
}
int rwkv7_wkv_state_fwd_f32(int B, int T, int C, int H, float *state, const void *r, const void *w,
                            const void *k, const void *v, const void *a, const void *b, void *y,
                            rwkv7_stream_t stream) {
    STATE_BODY(wkv_fwd_f32, 0)
}
int rwkv7_wkv_state_fwd_variant_bf16(int B, int T, int C, int H, float *state, const void *r, const void *w,
                                     const void *k, const void *v, const void *a, const void *b, void *y, int cols_per_lane,
                                     rwkv7_stream_t stream) {
    STATE_BODY(wkv_fwd_bf16, cols_per_lane)
}


// ---- fused elementwise stages ----------------------------------------------------------------------------------
#define SHAPE_OK(D) ((D) > 0 && (D) % 64 == 0 && (D) <= 4096)  /* D/8 threads per row, kEwMaxThreads = 512 */
#define EW_DEFINE(SFX, TY)                                                                                           \
    int rwkv7_mix_fwd_##SFX(int B, int T, int D, int nmix, const void *x, const void *x_prev, const void *mask,       \
                            const void *params, void *out, int nblocks, rwkv7_stream_t stream) {                      \
        if (B <= 0 || T <= 0 || nblocks <= 0 || any_null({x, params, out})) return RWKV7_EINVAL;                      \
        if (!SHAPE_OK(D) || (nmix != 1 && nmix != 3 && nmix != 6)) return RWKV7_ESHAPE;                                            \
        return rwkv7::mix_fwd<TY>(B, T, D, nmix, x, x_prev, mask, params, out, nblocks, (hipStream_t)stream);         \
    }                                                                                                                 \
    int rwkv7_mix_bwd_##SFX(int B, int T, int D, int nmix, const void *const *g, const void *x, const void *x_prev,   \
                            const void *mask, const void *params, void *dx, float *dpart, int nblocks,                \
                            int run_len, rwkv7_stream_t stream) {                                                     \
        if (B <= 0 || T <= 0 || nblocks <= 0 || run_len <= 0 || any_null({g, x, params, dx, dpart}))                  \
            return RWKV7_EINVAL;                                                                                      \
        if (!SHAPE_OK(D) || (nmix != 1 && nmix != 3 && nmix != 6)) return RWKV7_ESHAPE;                                            \
        for (int i = 0; i < nmix; i++)                                                                                \
            if (!g[i]) return RWKV7_EINVAL;                                                                           \
        return rwkv7::mix_bwd<TY>(B, T, D, nmix, g, x, x_prev, mask, params, dx, dpart, nblocks, run_len,            \
                                  (hipStream_t)stream); \
    }                                                                                                                 \
    int rwkv7_tmix_prepare_fwd_##SFX(long rows, int D, const void *w_pre, const void *k, const void *v,               \
                                     const void *a_pre, const void *v_pre, const void *v_first, const void *mask,     \
                                     const void *k_k, const void *k_a, void *w, void *k2, void *v2, void *ain,        \
                                     void *bin, int nblocks, rwkv7_stream_t stream) {                                 \
        if (rows <= 0 || nblocks <= 0 || any_null({w_pre, k, v, a_pre, k_k, k_a, w, k2, v2, ain, bin}))               \
            return RWKV7_EINVAL;                                                                                      \
        if ((v_pre == nullptr) != (v_first == nullptr)) return RWKV7_EINVAL;                                          \
        if (!SHAPE_OK(D)) return RWKV7_ESHAPE;                                                                        \
        return rwkv7::tmix_prepare_fwd<TY>(rows, D, w_pre, k, v, a_pre, v_pre, v_first, mask, k_k, k_a, w, k2, v2,    \
                                           ain, bin, nblocks, (hipStream_t)stream);                                   \
    }                                                                                                                 \
    int rwkv7_tmix_prepare_bwd_##SFX(long rows, int D, const void *w_pre, const void *k, const void *v,               \
                                     const void *a_pre, const void *v_pre, const void *v_first, const void *mask,     \
                                     const void *k_k, const void *k_a, const void *d_w, const void *d_k2,             \
                                     const void *d_v2, const void *d_ain, const void *d_bin, void *d_wpre, void *d_k, \
                                     void *d_v, void *d_apre, void *d_vpre, void *d_vfirst, float *dpart,             \
                                     int nblocks, rwkv7_stream_t stream) {                                            \
        if (rows <= 0 || nblocks <= 0 ||                                                                              \
            any_null({w_pre, k, v, a_pre, k_k, k_a, d_w, d_k2, d_v2, d_ain, d_bin, d_wpre, d_k, d_v, d_apre, dpart})) \
            return RWKV7_EINVAL;                                                                                      \
        if ((v_pre == nullptr) != (v_first == nullptr)) return RWKV7_EINVAL;                                          \
        if (v_pre && (!d_vpre || !d_vfirst)) return RWKV7_EINVAL;                                                     \
        if (!SHAPE_OK(D)) return RWKV7_ESHAPE;                                                                        \
        return rwkv7::tmix_prepare_bwd<TY>(rows, D, w_pre, k, v, a_pre, v_pre, v_first, mask, k_k, k_a, d_w, d_k2,    \
                                           d_v2, d_ain, d_bin, d_wpre, d_k, d_v, d_apre, d_vpre, d_vfirst, dpart,     \
                                           nblocks, (hipStream_t)stream);                                             \
    }                                                                                                                 \
    int rwkv7_tmix_prepare_bwd_sum_##SFX(long rows, int D, const void *w_pre, const void *k, const void *v,           \
                                         const void *a_pre, const void *v_pre, const void *v_first, const void *mask, \
                                         const void *k_k, const void *k_a, const void *const *gsum, void *d_wpre,     \
                                         void *d_k, void *d_v, void *d_apre, void *d_vpre, void *d_vfirst, void *d_r, \
                                         float *dpart, int nblocks, rwkv7_stream_t stream) {                          \
        if (rows <= 0 || nblocks <= 0 ||                                                                              \
            any_null({w_pre, k, v, a_pre, k_k, k_a, (const void *)gsum, d_wpre, d_k, d_v, d_apre, d_r, dpart}))       \
            return RWKV7_EINVAL;                                                                                      \
        for (int i = 0; i < 14; i++) /* the second partials {1,3,8,10,12} may be NULL */                              \
            if (!gsum[i] && i != 1 && i != 3 && i != 8 && i != 10 && i != 12) return RWKV7_EINVAL;                    \
        if ((v_pre == nullptr) != (v_first == nullptr)) return RWKV7_EINVAL;                                          \
        if (v_pre && (!d_vpre || !d_vfirst)) return RWKV7_EINVAL;                                                     \
        if (!SHAPE_OK(D)) return RWKV7_ESHAPE;                                                                        \
        return rwkv7::tmix_prepare_bwd_sum<TY>(rows, D, w_pre, k, v, a_pre, v_pre, v_first, mask, k_k, k_a, gsum, 15, \
                                               d_wpre, d_k, d_v, d_apre, d_vpre, d_vfirst, d_r, dpart, nblocks,       \
                                               (hipStream_t)stream);                                                  \
    }                                                                                                                 \
    int rwkv7_tmix_prepare_bwd_sum_compact_##SFX(long rows, int D, const void *w_pre, const void *k, const void *v,   \
                                         const void *a_pre, const void *v_pre, const void *v_first, const void *mask, \
                                         const void *k_k, const void *k_a, const void *const *gsum, void *d_wpre,     \
                                         void *d_k, void *d_v, void *d_apre, void *d_vpre, void *d_vfirst, void *d_r, \
                                         float *dpart, int nblocks, rwkv7_stream_t stream) {                          \
        if (rows <= 0 || nblocks <= 0 ||                                                                              \
            any_null({w_pre, k, v, a_pre, k_k, k_a, (const void *)gsum, d_wpre, d_k, d_v, d_apre, d_r, dpart}))       \
            return RWKV7_EINVAL;                                                                                      \
        /* gsum: 19 pointers; {0,2,5,7,9,11} and the hand-off {15..18} are required, the post tensors {4,6,13} are unused */ \
        for (int i : {0, 2, 5, 7, 9, 11, 15, 16, 17, 18})                                                             \
            if (!gsum[i]) return RWKV7_EINVAL;                                                                        \
        if ((v_pre == nullptr) != (v_first == nullptr)) return RWKV7_EINVAL;                                          \
        if (v_pre && (!d_vpre || !d_vfirst)) return RWKV7_EINVAL;                                                     \
        if (!SHAPE_OK(D)) return RWKV7_ESHAPE;                                                                        \
        return rwkv7::tmix_prepare_bwd_sum<TY>(rows, D, w_pre, k, v, a_pre, v_pre, v_first, mask, k_k, k_a, gsum, 19, \
                                               d_wpre, d_k, d_v, d_apre, d_vpre, d_vfirst, d_r, dpart, nblocks,       \
                                               (hipStream_t)stream);                                                  \
    }                                                                                                                 \
    int rwkv7_add_ln_fwd_##SFX(long rows, int D, const void *x, const void *branch, const void *gamma,                \
                               const void *beta, float eps, void *x_out, void *h, float *mean, float *rstd,           \
                               int nblocks, rwkv7_stream_t stream) {                                                  \
        if (rows <= 0 || nblocks <= 0 || any_null({x, gamma, h, (const void *)mean, (const void *)rstd}))             \
            return RWKV7_EINVAL;                                                                                      \
        if (branch && !x_out) return RWKV7_EINVAL;                                                                    \
        if (!SHAPE_OK(D)) return RWKV7_ESHAPE;                                                                        \
        return rwkv7::add_ln_fwd<TY>(rows, D, x, branch, gamma, beta, eps, x_out, h, mean, rstd, nblocks,             \
                                     (hipStream_t)stream);                                                            \
    }                                                                                                                 \
    int rwkv7_add_ln_bwd_##SFX(long rows, int D, const void *dh, const void *d_resid, const void *x1,                 \
                               const float *mean, const float *rstd, const void *gamma, void *dx, float *dpart,       \
                               int nblocks, rwkv7_stream_t stream) {                                                  \
        if (rows <= 0 || nblocks <= 0 ||                                                                              \
            any_null({dh, x1, (const void *)mean, (const void *)rstd, gamma, dx, (const void *)dpart}))               \
            return RWKV7_EINVAL;                                                                                      \
        if (!SHAPE_OK(D)) return RWKV7_ESHAPE;                                                                        \
        return rwkv7::add_ln_bwd<TY>(rows, D, dh, d_resid, x1, mean, rstd, gamma, dx, dpart, nblocks,                 \
                                     (hipStream_t)stream);                                                            \
    }                                                                                                                 \
    int rwkv7_add_ln_mix_fwd_##SFX(int B, int T, int D, int nmix, const void *x, const void *branch, const void *gamma, \
                                   const void *beta, float eps, const void *mask, const void *params, void *x_out,   \
                                   void *out, float *mean, float *rstd, int nblocks, int run_len,                     \
                                   rwkv7_stream_t stream) {                                                           \
        if (B <= 0 || T <= 0 || nblocks <= 0 || run_len <= 0 ||                                                       \
            any_null({x, gamma, params, out, (const void *)mean, (const void *)rstd}))                                \
            return RWKV7_EINVAL;                                                                                      \
        if (branch && !x_out) return RWKV7_EINVAL;                                                                    \
        if (!SHAPE_OK(D) || (nmix != 1 && nmix != 3 && nmix != 6)) return RWKV7_ESHAPE;                               \
        return rwkv7::add_ln_mix_fwd<TY>(B, T, D, nmix, nullptr, x, branch, gamma, beta, eps, mask, params, x_out,    \
                                         out, mean, rstd, nblocks, run_len, (hipStream_t)stream);                     \
    }                                                                                                                 \
    int rwkv7_add_ln_mix_fwd_h_##SFX(int B, int T, int D, int nmix, const void *x, const void *branch,                \
                                     const void *gamma, const void *beta, float eps, const void *mask,                \
                                     const void *params, void *x_out, void *out, void *h, float *mean, float *rstd,   \
                                     int nblocks, int run_len, rwkv7_stream_t stream) {                               \
        if (B <= 0 || T <= 0 || nblocks <= 0 || run_len <= 0 ||                                                       \
            any_null({x, gamma, params, out, (const void *)h, (const void *)mean, (const void *)rstd}))                \
            return RWKV7_EINVAL;                                                                                      \
        if (branch && !x_out) return RWKV7_EINVAL;                                                                    \
        if (!SHAPE_OK(D) || (nmix != 1 && nmix != 3 && nmix != 6)) return RWKV7_ESHAPE;                               \
        return rwkv7::add_ln_mix_fwd<TY>(B, T, D, nmix, h, x, branch, gamma, beta, eps, mask, params, x_out, out,     \
                                         mean, rstd, nblocks, run_len, (hipStream_t)stream);                          \
    }                                                                                                                 \
    int rwkv7_mix_add_ln_bwd_##SFX(int B, int T, int D, int nmix, const void *const *g, const void *d_resid,          \
                                   const void *x1, const float *mean, const float *rstd, const void *gamma,           \
                                   const void *beta, const void *mask, const void *params, void *dx, float *dpart,    \
                                   int nblocks, int run_len, rwkv7_stream_t stream) {                                 \
        if (B <= 0 || T <= 0 || nblocks <= 0 || run_len <= 0 ||                                                       \
            any_null({(const void *)g, x1, (const void *)mean, (const void *)rstd, gamma, params, dx,                 \
                      (const void *)dpart}))                                                                          \
            return RWKV7_EINVAL;                                                                                      \
        if (!SHAPE_OK(D) || (nmix != 1 && nmix != 6)) return RWKV7_ESHAPE;                                            \
        for (int i = 0; i < nmix; i++)                                                                                \
            if (!g[i]) return RWKV7_EINVAL;                                                                           \
        return rwkv7::mix_add_ln_bwd<TY>(B, T, D, nmix, g, d_resid, x1, mean, rstd, gamma, beta, mask, params, dx,    \
                                         dpart, nblocks, run_len, (hipStream_t)stream);                               \
    }                                                                                                                 \
    int rwkv7_tmix_post_fwd_##SFX(long rows, int D, const void *y, const void *r, const void *k, const void *v,       \
                                  const void *g, const void *gn_w, const void *gn_b, const void *r_k, float eps,      \
                                  void *out, int nblocks, rwkv7_stream_t stream) {                                    \
        if (rows <= 0 || nblocks <= 0 || any_null({y, r, k, v, g, gn_w, gn_b, r_k, out})) return RWKV7_EINVAL;        \
        if (!SHAPE_OK(D)) return RWKV7_ESHAPE;                                                                        \
        return rwkv7::tmix_post_fwd<TY>(rows, D, y, r, k, v, g, gn_w, gn_b, r_k, eps, out, nblocks,                   \
                                        (hipStream_t)stream);                                                         \
    }                                                                                                                 \
    int rwkv7_tmix_post_bwd_##SFX(long rows, int D, const void *dout, const void *y, const void *r, const void *k,    \
                                  const void *v, const void *g, const void *gn_w, const void *gn_b, const void *r_k,  \
                                  float eps, void *d_y, void *d_r, void *d_k, void *d_v, void *d_g, float *dpart,     \
                                  int nblocks, rwkv7_stream_t stream) {                                               \
        if (rows <= 0 || nblocks <= 0 ||                                                                              \
            any_null({dout, y, r, k, v, g, gn_w, gn_b, r_k, d_y, d_r, d_k, d_v, d_g, dpart}))                         \
            return RWKV7_EINVAL;                                                                                      \
        if (!SHAPE_OK(D)) return RWKV7_ESHAPE;                                                                        \
        return rwkv7::tmix_post_bwd<TY>(rows, D, dout, y, r, k, v, g, gn_w, gn_b, r_k, eps, d_y, d_r, d_k, d_v, d_g,  \
                                        dpart, nullptr, nblocks, (hipStream_t)stream);                                \
    }                                                                                                                 \
    int rwkv7_tmix_post_bwd_compact_##SFX(long rows, int D, const void *dout, const void *y, const void *r,           \
                                  const void *k, const void *v, const void *g, const void *gn_w, const void *gn_b,    \
                                  const void *r_k, float eps, void *d_y, void *dt, void *d_g, float *hscal,           \
                                  float *dpart, int nblocks, rwkv7_stream_t stream) {                                 \
        if (rows <= 0 || nblocks <= 0 ||                                                                              \
            any_null({dout, y, r, k, v, g, gn_w, gn_b, r_k, d_y, dt, d_g, (const void *)hscal, (const void *)dpart})) \
            return RWKV7_EINVAL;                                                                                      \
        if (!SHAPE_OK(D)) return RWKV7_ESHAPE;                                                                        \
        return rwkv7::tmix_post_bwd<TY>(rows, D, dout, y, r, k, v, g, gn_w, gn_b, r_k, eps, d_y, nullptr, nullptr,    \
                                        dt, d_g, dpart, hscal, nblocks, (hipStream_t)stream);                         \
    }                                                                                                                 \
    int rwkv7_relusq_fwd_##SFX(long n, const void *x, void *y, rwkv7_stream_t stream) {                               \
        if (n <= 0 || any_null({x, y})) return RWKV7_EINVAL;                                                          \
        if (n % 8 != 0) return RWKV7_ESHAPE;                                                                          \
        return rwkv7::relusq_fwd<TY>(n, x, y, (hipStream_t)stream);                                                   \
    }                                                                                                                 \
    int rwkv7_relusq_bwd_##SFX(long n, const void *x, const void *dy, void *dx, rwkv7_stream_t stream) {              \
        if (n <= 0 || any_null({x, dy, dx})) return RWKV7_EINVAL;                                                     \
        if (n % 8 != 0) return RWKV7_ESHAPE;                                                                          \
        return rwkv7::relusq_bwd<TY>(n, x, dy, dx, (hipStream_t)stream);                                              \
    }                                                                                                                 \
    int rwkv7_relusq_bwd_s_##SFX(long n, const void *s, const void *dy, void *dx, rwkv7_stream_t stream) {            \
        if (n <= 0 || any_null({s, dy, dx})) return RWKV7_EINVAL;                                                     \
        if (n % 8 != 0) return RWKV7_ESHAPE;                                                                          \
        return rwkv7::relusq_bwd_s<TY>(n, s, dy, dx, (hipStream_t)stream);                                            \
    }

EW_DEFINE(bf16, rwkv7::bf16_t)
EW_DEFINE(f32, float)


// ---- chunked (MFMA) WKV7 -----------------------------------------------------------------------------------------
#define CHUNK_DEFINE(SFX)                                                                                          \
    int rwkv7_wkv_chunk_prep_##SFX(int B, int T, int H, const void *w, const void *a, const void *b, float *tinv,   \
                                   rwkv7_stream_t stream) {                                                         \
        if (B <= 0 || T <= 0 || H <= 0 || any_null({w, a, b, tinv})) return RWKV7_EINVAL;                           \
        if (T % RWKV7_CHUNK_T != 0) return RWKV7_ECHUNK;                                                            \
        return rwkv7::chunk_prep_##SFX(B, T, H, w, a, b, tinv, (hipStream_t)stream);                                \
    }                                                                                                               \
    int rwkv7_wkv_chunk_fwd_##SFX(int B, int T, int H, const void *w, const void *q, const void *k, const void *v,  \
                                  const void *a, const void *b, const float *tinv, void *y, float *sa, void *hs,   \
                                  rwkv7_stream_t stream) {                                                          \
        if (B <= 0 || T <= 0 || H <= 0 || any_null({w, q, k, v, a, b, tinv, y})) return RWKV7_EINVAL;               \
        if ((sa == nullptr) != (hs == nullptr)) return RWKV7_EINVAL;                                                \
        if (T % RWKV7_CHUNK_T != 0) return RWKV7_ECHUNK;                                                            \
        return rwkv7::chunk_fwd_##SFX(B, T, H, w, q, k, v, a, b, tinv, y, sa, hs, nullptr, 0, (hipStream_t)stream); \
    }                                                                                                               \
    int rwkv7_wkv_chunk_fwd_seq_##SFX(int B, int T, int H, const void *w, const void *q, const void *k, const void *v, \
                                      const void *a, const void *b, const float *tinv, void *y, float *sa, void *hs, \
                                      const int *seq_chunk_off, int nseq, rwkv7_stream_t stream) {                  \
        if (B <= 0 || T <= 0 || H <= 0 || any_null({w, q, k, v, a, b, tinv, y})) return RWKV7_EINVAL;               \
        if ((sa == nullptr) != (hs == nullptr) || (seq_chunk_off != nullptr && nseq <= 0)) return RWKV7_EINVAL;    \
        if (T % RWKV7_CHUNK_T != 0) return RWKV7_ECHUNK;                                                            \
        return rwkv7::chunk_fwd_##SFX(B, T, H, w, q, k, v, a, b, tinv, y, sa, hs, seq_chunk_off, nseq, (hipStream_t)stream); \
    }
CHUNK_DEFINE(bf16)
CHUNK_DEFINE(f32)
// chunked backward (bf16): csrc/wkv7_chunk_bseq.hip (adjoint recurrence, writes E and Z) + csrc/wkv7_chunk_bwd10.hip (per-chunk gradients)
int rwkv7_wkv_chunk_bseq_bf16(int B, int T, int H, const void *w, const void *q, const void *a, const void *b, const void *dy,
                              const float *tinv, void *e_vk, float *z, const int *seq_chunk_off, int nseq, rwkv7_stream_t stream) {
    if (B <= 0 || T <= 0 || H <= 0 || any_null({w, q, a, b, dy, (const void *)tinv, (const void *)e_vk})) return RWKV7_EINVAL;
    if (seq_chunk_off != nullptr && nseq <= 0) return RWKV7_EINVAL;
    if (T % 32 != 0) return RWKV7_ECHUNK;
    return rwkv7::chunk_bseq_bf16(B, T, H, w, q, a, b, dy, tinv, e_vk, z, seq_chunk_off, nseq, (hipStream_t)stream);
}
int rwkv7_wkv_chunk_bwd_out_z_bf16(int B, int T, int H, const void *w, const void *q, const void *k, const void *v,
                                   const void *a, const void *b, const void *dy, const void *hs, const float *sa,
                                   const float *z, const void *e_vk, void *dw, void *dq, void *dk, void *dv,
                                   void *da, void *db, rwkv7_stream_t stream) {
    if (B <= 0 || T <= 0 || H <= 0 ||
        any_null({w, q, k, v, a, b, dy, hs, (const void *)sa, (const void *)z, e_vk, dw, dq, dk, dv, da, db}))
        return RWKV7_EINVAL;
    if (T % 32 != 0) return RWKV7_ECHUNK;
    return rwkv7::chunk_bwd_out10_bf16(B, T, H, w, q, k, v, a, b, dy, hs, sa, z, e_vk, dw, dq, dk, dv, da, db, (hipStream_t)stream);
}
int rwkv7_gemm_nt_bf16(int M, int N, int K, const void *A, const void *W, void *C, int epilogue, rwkv7_stream_t stream) {
    if (any_null({A, W, (const void *)C})) return RWKV7_EINVAL;
    if (M <= 0 || N <= 0 || K <= 0 || M % 256 != 0 || N % 256 != 0 || K % 1024 != 0 || epilogue < 0 || epilogue > 1) return RWKV7_ESHAPE;
    return rwkv7::gemm_nt4_bf16(M, N, K, A, W, C, nullptr, epilogue, (hipStream_t)stream);
}
int rwkv7_gemm_nt_relusq_bwd_bf16(int M, int N, int K, const void *A, const void *W, const void *aux, void *C, rwkv7_stream_t stream) {
    if (any_null({A, W, aux, (const void *)C})) return RWKV7_EINVAL;
    if (M <= 0 || N <= 0 || K <= 0 || M % 256 != 0 || N % 256 != 0 || K % 1024 != 0) return RWKV7_ESHAPE;
    return rwkv7::gemm_nt4_bf16(M, N, K, A, W, C, aux, 2, (hipStream_t)stream);
}
int rwkv7_gemm_nt_relusq_bwd_s_bf16(int M, int N, int K, const void *A, const void *W, const void *s, void *C, rwkv7_stream_t stream) {
    if (any_null({A, W, s, (const void *)C})) return RWKV7_EINVAL;
    if (M <= 0 || N <= 0 || K <= 0 || M % 256 != 0 || N % 256 != 0 || K % 1024 != 0) return RWKV7_ESHAPE;
    return rwkv7::gemm_nt4_bf16(M, N, K, A, W, C, s, 3, (hipStream_t)stream);
}
int rwkv7_gemm_nt_add_bf16(int M, int N, int K, const void *A, const void *W, const void *resid, void *C, rwkv7_stream_t stream) {
    if (any_null({A, W, resid, (const void *)C})) return RWKV7_EINVAL;
    if (M <= 0 || N <= 0 || K <= 0 || M % 256 != 0 || N % 256 != 0 || K % 1024 != 0) return RWKV7_ESHAPE;
    return rwkv7::gemm_nt4_bf16(M, N, K, A, W, C, resid, 4, (hipStream_t)stream);
}
namespace {
// ranks: multiples of 8, 1 <= nb <= 4; fills nb, r, off; returns R or a negative error
int mix_lora_desc(rwkv7::MixLoraDesc &d, int nb, const int *ranks) {
    if (nb < 1 || nb > 4 || ranks == nullptr) return RWKV7_EINVAL;
    d.nb = nb;
    int R = 0;
    for (int i = 0; i < 4; i++) {
        d.r[i] = i < nb ? ranks[i] : 0;
        d.off[i] = R;
        d.act[i] = 0;
        d.w1[i] = d.mu[i] = nullptr;
        d.out[i] = d.out2[i] = nullptr;
        if (i < nb) {
            if (ranks[i] <= 0 || ranks[i] % 8 != 0) return RWKV7_ESHAPE;
            R += ranks[i];
        }
    }
    return R;
}
}  // namespace
int rwkv7_mix_lora_wcat_fwd_bf16(int nb, const int *ranks, const void *const *w1, const void *const *mu, int D, void *wcat, rwkv7_stream_t stream) {
    rwkv7::MixLoraDesc d;
    const int R = mix_lora_desc(d, nb, ranks);
    if (R < 0) return R;
    if (w1 == nullptr || mu == nullptr || wcat == nullptr) return RWKV7_EINVAL;
    if (D <= 0) return RWKV7_ESHAPE;
    for (int i = 0; i < nb; i++) {
        if (w1[i] == nullptr || mu[i] == nullptr) return RWKV7_EINVAL;
        d.w1[i] = w1[i];
        d.mu[i] = mu[i];
    }
    return rwkv7::mix_lora_wcat_fwd(d, R, D, wcat, (hipStream_t)stream);
}
int rwkv7_mix_lora_wcat_bwd_bf16(int nb, const int *ranks, const void *const *w1, const void *const *mu, int D, const void *dwcat,
                                 void *const *dw1, void *const *dmu, rwkv7_stream_t stream) {
    rwkv7::MixLoraDesc d;
    const int R = mix_lora_desc(d, nb, ranks);
    if (R < 0) return R;
    if (w1 == nullptr || mu == nullptr || dwcat == nullptr || dw1 == nullptr || dmu == nullptr) return RWKV7_EINVAL;
    if (D <= 0) return RWKV7_ESHAPE;
    for (int i = 0; i < nb; i++) {
        if (w1[i] == nullptr || mu[i] == nullptr || dw1[i] == nullptr || dmu[i] == nullptr) return RWKV7_EINVAL;
        d.w1[i] = w1[i];
        d.mu[i] = mu[i];
        d.out[i] = dw1[i];
        d.out2[i] = dmu[i];
    }
    return rwkv7::mix_lora_wcat_bwd(d, R, D, dwcat, (hipStream_t)stream);
}
int rwkv7_mix_lora_combine_fwd_bf16(int nb, const int *ranks, const int *acts, long M, int T, const void *G, const void *mask,
                                    void *const *out, rwkv7_stream_t stream) {
    rwkv7::MixLoraDesc d;
    const int R = mix_lora_desc(d, nb, ranks);
    if (R < 0) return R;
    if (acts == nullptr || G == nullptr || out == nullptr) return RWKV7_EINVAL;
    if (M <= 0 || T <= 0 || M % T != 0) return RWKV7_ESHAPE;
    for (int i = 0; i < nb; i++) {
        if (out[i] == nullptr || acts[i] < 0 || acts[i] > 2) return RWKV7_EINVAL;
        d.act[i] = acts[i];
        d.out[i] = out[i];
    }
    return rwkv7::mix_lora_combine_fwd(d, M, T, R, G, mask, (hipStream_t)stream);
}
int rwkv7_mix_lora_combine_bwd_bf16(int nb, const int *ranks, const int *acts, long M, int T, const void *mask, const void *const *y,
                                    const void *const *dy, void *dG, rwkv7_stream_t stream) {
    rwkv7::MixLoraDesc d;
    const int R = mix_lora_desc(d, nb, ranks);
    if (R < 0) return R;
    if (acts == nullptr || y == nullptr || dy == nullptr || dG == nullptr) return RWKV7_EINVAL;
    if (M <= 0 || T <= 0 || M % T != 0) return RWKV7_ESHAPE;
    for (int i = 0; i < nb; i++) {
        if (y[i] == nullptr || dy[i] == nullptr || acts[i] < 0 || acts[i] > 2) return RWKV7_EINVAL;
        d.act[i] = acts[i];
        d.out[i] = const_cast<void *>(y[i]);
        d.out2[i] = const_cast<void *>(dy[i]);
    }
    return rwkv7::mix_lora_combine_bwd(d, M, T, R, mask, dG, (hipStream_t)stream);
}
namespace {
// ranks: multiples of 32, 1 <= nb <= 4, at most 16 column tiles; fills nb, r, off; returns R or a negative error
int lora_down_desc(rwkv7::LoraDownDesc &d, int nb, const int *ranks) {
    if (nb < 1 || nb > 4 || ranks == nullptr) return RWKV7_EINVAL;
    d.nb = nb;
    int R = 0;
    for (int i = 0; i < 4; i++) {
        d.r[i] = i < nb ? ranks[i] : 0;
        d.off[i] = R;
        d.act[i] = 0;
        d.w1[i] = d.mu[i] = nullptr;
        d.out[i] = nullptr;
        if (i < nb) {
            if (ranks[i] <= 0 || ranks[i] % 32 != 0) return RWKV7_ESHAPE;
            R += ranks[i];
        }
    }
    for (int i = 0; i < 5; i++) d.tile0[i] = 0;
    return R <= 512 ? R : RWKV7_ESHAPE;
}
}  // namespace
int rwkv7_lora_down_pack_bf16(int nb, const int *ranks, const void *const *w1, int D, void *packed, rwkv7_stream_t stream) {
    rwkv7::LoraDownDesc d;
    const int R = lora_down_desc(d, nb, ranks);
    if (R < 0) return R;
    if (w1 == nullptr || packed == nullptr) return RWKV7_EINVAL;
    if (D <= 0 || D % 128 != 0) return RWKV7_ESHAPE;
    for (int i = 0; i < nb; i++) {
        if (w1[i] == nullptr) return RWKV7_EINVAL;
        d.w1[i] = w1[i];
    }
    return rwkv7::lora_down_pack(d, R, D, packed, (hipStream_t)stream);
}
int rwkv7_lora_down_fwd_bf16(int nb, const int *ranks, const int *acts, long M, int T, int D, const void *x, const void *mask,
                             const void *const *mu, const void *packed, void *const *out, rwkv7_stream_t stream) {
    rwkv7::LoraDownDesc d;
    const int R = lora_down_desc(d, nb, ranks);
    if (R < 0) return R;
    if (acts == nullptr || x == nullptr || mu == nullptr || packed == nullptr || out == nullptr) return RWKV7_EINVAL;
    if (M <= 0 || T <= 0 || M % T != 0 || M % 128 != 0 || D <= 0 || D % 128 != 0 || D > 4096) return RWKV7_ESHAPE;
    for (int i = 0; i < nb; i++) {
        if (mu[i] == nullptr || out[i] == nullptr || acts[i] < 0 || acts[i] > 2) return RWKV7_EINVAL;
        d.act[i] = acts[i];
        d.mu[i] = mu[i];
        d.out[i] = out[i];
    }
    const int rc = rwkv7::lora_down_fwd(d, R, M, T, D, x, mask, packed, (hipStream_t)stream);
    return rc == -4 ? RWKV7_ESHAPE : rc;
}
int rwkv7_gemv32_bf16(int M, int N, int K, const void *x, const void *w, const void *bias, void *y, rwkv7_stream_t stream) {
    if (any_null({x, w, y})) return RWKV7_EINVAL;
    if (M <= 0 || M > 32 || N <= 0 || K <= 0 || K % 64 != 0) return RWKV7_ESHAPE;
    return rwkv7::gemv32_bf16(M, N, K, x, w, bias, y, (hipStream_t)stream);
}
int rwkv7_lora32_bf16(int M, int N, int K, int R, int act, const void *x, const void *w1, const void *w2, const void *bias,
                      void *y, rwkv7_stream_t stream) {
    if (any_null({x, w1, w2, y})) return RWKV7_EINVAL;
    if (M <= 0 || M > 32 || N <= 0 || K <= 0 || K % 64 != 0 || (R != 32 && R != 64 && R != 128) || act < 0 || act > 2)
        return RWKV7_ESHAPE;
    return rwkv7::lora32_bf16(M, N, K, R, act, x, w1, w2, bias, y, (hipStream_t)stream);
}
int rwkv7_adamw_groups_bf16(long n, float *p32, const void *g16, float *m, float *v, void *p16, const unsigned char *slab_group,
                            const float *group_tab, int ngroups, const float *skip_flag, float lr, float beta1, float beta2, float eps,
                            int step, rwkv7_stream_t stream) {
    if (n <= 0 || step <= 0 || any_null({(const void *)p32, g16, (const void *)m, (const void *)v, p16})) return RWKV7_EINVAL;
    if ((slab_group == nullptr) != (group_tab == nullptr) || (slab_group && (ngroups <= 0 || ngroups > 256))) return RWKV7_EINVAL;
    if (n % 4 != 0 || (slab_group && n % 128 != 0)) return RWKV7_ESHAPE;
    const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
    return rwkv7::adamw_step(n, p32, g16, m, v, p16, slab_group, group_tab, skip_flag, lr, beta1, beta2, eps, 0.f, (float)(1.0 / bc1),
                             (float)(1.0 / sqrt(bc2)), (hipStream_t)stream);
}
int rwkv7_adamw_bf16(long n, float *p32, const void *g16, float *m, float *v, void *p16, float lr, float beta1, float beta2,
                     float eps, float weight_decay, int step, rwkv7_stream_t stream) {
    if (n <= 0 || step <= 0 || any_null({(const void *)p32, g16, (const void *)m, (const void *)v, p16})) return RWKV7_EINVAL;
    if (n % 4 != 0) return RWKV7_ESHAPE;
    const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
    return rwkv7::adamw_step(n, p32, g16, m, v, p16, nullptr, nullptr, nullptr, lr, beta1, beta2, eps, weight_decay,
                             (float)(1.0 / bc1), (float)(1.0 / sqrt(bc2)), (hipStream_t)stream);
}
int rwkv7_ce_fwd_bwd_bf16(long rows, int V, void *logits, const long *labels, long ignore_index, float scale, float *loss_rows,
                          rwkv7_stream_t stream) {
    if (rows <= 0 || V <= 0 || any_null({(const void *)logits, (const void *)labels, (const void *)loss_rows})) return RWKV7_EINVAL;
    return rwkv7::ce_fwd_bwd(rows, V, V, logits, labels, ignore_index, scale, loss_rows, 0.f, (hipStream_t)stream);
}
int rwkv7_ce_fwd_bwd_ls_bf16(long rows, int V, void *logits, const long *labels, long ignore_index, float scale, float *loss_rows,
                             float label_smoothing,
                          rwkv7_stream_t stream) {
    if (rows <= 0 || V <= 0 || any_null({(const void *)logits, (const void *)labels, (const void *)loss_rows})) return RWKV7_EINVAL;
    if (!(label_smoothing >= 0.f && label_smoothing < 1.f)) return RWKV7_EINVAL;
    return rwkv7::ce_fwd_bwd(rows, V, V, logits, labels, ignore_index, scale, loss_rows, label_smoothing, (hipStream_t)stream);
}
int rwkv7_ce_fwd_bwd_ld_bf16(long rows, int V, long ld, void *logits, const long *labels, long ignore_index, float scale, float *loss_rows,
                             float label_smoothing, rwkv7_stream_t stream) {
    if (rows <= 0 || V <= 0 || ld < V || any_null({(const void *)logits, (const void *)labels, (const void *)loss_rows})) return RWKV7_EINVAL;
    if (!(label_smoothing >= 0.f && label_smoothing < 1.f)) return RWKV7_EINVAL;
    return rwkv7::ce_fwd_bwd(rows, V, ld, logits, labels, ignore_index, scale, loss_rows, label_smoothing, (hipStream_t)stream);
}
int rwkv7_wgrad_skinny_bf16(long M, int N, int K, int S, const void *dy, const void *x, float *parts, rwkv7_stream_t stream) {
    if (any_null({dy, x, parts})) return RWKV7_EINVAL;
    const int rank = N > K ? K : N, wide = N > K ? N : K;
    if (M <= 0 || S <= 0 || M % S != 0 || (M / S) % 128 != 0 || wide % 256 != 0 || (rank != 32 && rank != 64 && rank != 128))
        return RWKV7_ESHAPE;
    return rwkv7::wgrad_skinny_bf16(M, N, K, S, dy, x, parts, (hipStream_t)stream);
}
int rwkv7_wgrad_mid_bf16(long M, int N, int K, int S, const void *dy, const void *x, float *parts, rwkv7_stream_t stream) {
    if (any_null({dy, x, parts})) return RWKV7_EINVAL;
    if (M <= 0 || S <= 0 || M % S != 0 || (M / S) % 64 != 0 || K <= 0 || K % 256 != 0 || (N != 512 && N != 576)) return RWKV7_ESHAPE;
    return rwkv7::wgrad_mid_bf16(M, N, K, S, dy, x, parts, (hipStream_t)stream);
}
int rwkv7_sum_slabs_bf16(long n, int S, const float *parts, void *out, int accumulate, rwkv7_stream_t stream) {
    if (n <= 0 || S <= 0 || any_null({(const void *)parts, (const void *)out})) return RWKV7_EINVAL;
    if (n % 4 != 0) return RWKV7_ESHAPE;
    return rwkv7::sum_slabs_bf16(n, S, parts, out, accumulate, (hipStream_t)stream);
}
int rwkv7_gather_rows_bf16(long n_out, int D, const void *src, const int *idx, void *out, rwkv7_stream_t stream) {
    if (n_out <= 0 || D <= 0 || any_null({src, (const void *)idx, (const void *)out})) return RWKV7_EINVAL;
    if (D % 8 != 0) return RWKV7_ESHAPE;
    return rwkv7::gather_rows16(n_out, D, src, idx, out, (hipStream_t)stream);
}
int rwkv7_transpose_bf16(int R, int C, const void *in, void *out, rwkv7_stream_t stream) {
    if (R <= 0 || C <= 0 || any_null({in, (const void *)out})) return RWKV7_EINVAL;
    if (R % 64 != 0 || C % 64 != 0) return RWKV7_ESHAPE;
    return rwkv7::transpose_bf16(R, C, in, out, (hipStream_t)stream);
}
int rwkv7_decode_layer_ptrs(void) { return rwkv7::decode_layer_ptrs(); }
size_t rwkv7_decode_workspace_bytes(const rwkv7_decode_dims *dm) {
    if (!dm) return 0;
    return rwkv7::decode_workspace_bytes(dm->B, dm->D, dm->H, dm->F, dm->V, dm->Rw, dm->Ra, dm->Rv, dm->Rg);
}
int rwkv7_decode_step_bf16(const rwkv7_decode_dims *dm, const void *const *layer_tbl, const void *x_in, const void *norm_w,
                           const void *norm_b, const void *head_w, const void *head_b, float *logits, void *workspace,
                           int persistent, rwkv7_stream_t stream) {
    if (!dm || any_null({(const void *)layer_tbl, x_in, norm_w, norm_b, head_w, (const void *)logits, (const void *)workspace}))
        return RWKV7_EINVAL;
    if (dm->H * RWKV7_HEAD_SIZE != dm->D) return RWKV7_EHEAD;
    return rwkv7::decode_step_bf16(dm->B, dm->D, dm->H, dm->L, dm->F, dm->V, dm->Rw, dm->Ra, dm->Rv, dm->Rg, dm->ln_eps, dm->gn_eps,
                                   layer_tbl, nullptr, x_in, norm_w, norm_b, head_w, head_b, logits, workspace, persistent,
                                   (hipStream_t)stream);
}
int rwkv7_decode_step_tbl_bf16(const rwkv7_decode_dims *dm, const void *const *layer_tbl, const void *const *layer_tbl_host,
                               const void *x_in, const void *norm_w, const void *norm_b, const void *head_w, const void *head_b,
                               float *logits, void *workspace, int persistent, rwkv7_stream_t stream) {
    if (!dm || any_null({(const void *)layer_tbl, (const void *)layer_tbl_host, x_in, norm_w, norm_b, head_w, (const void *)logits,
                         (const void *)workspace}))
        return RWKV7_EINVAL;
    if (dm->H * RWKV7_HEAD_SIZE != dm->D) return RWKV7_EHEAD;
    return rwkv7::decode_step_bf16(dm->B, dm->D, dm->H, dm->L, dm->F, dm->V, dm->Rw, dm->Ra, dm->Rv, dm->Rg, dm->ln_eps, dm->gn_eps,
                                   layer_tbl, layer_tbl_host, x_in, norm_w, norm_b, head_w, head_b, logits, workspace, persistent,
                                   (hipStream_t)stream);
}
int rwkv7_sample_rows_f32(int rows, int nseg, const float *logits, long ld, const int *seg_off, const int *seg_len, const int *allow_lo,
                          const int *allow_hi, const int *suppress, int nsuppress, int max_domain, int do_sample, int top_k, float top_p,
                          float temperature, unsigned long long seed, const long *step, long *out, rwkv7_stream_t stream) {
    if (rows <= 0 || nseg <= 0 || max_domain <= 0 || nsuppress < 0 ||
        any_null({(const void *)logits, (const void *)seg_off, (const void *)seg_len, (const void *)step, (const void *)out}) ||
        (nsuppress > 0 && !suppress))
        return RWKV7_EINVAL;
    return rwkv7::sample_rows_f32(rows, nseg, logits, ld, seg_off, seg_len, allow_lo, allow_hi, suppress, nsuppress, max_domain, do_sample,
                                  top_k, top_p, temperature, seed, step, out, nullptr, -1, 0, (hipStream_t)stream);
}
int rwkv7_sample_rows_tail_f32(int rows, const float *logits, long ld, const int *seg_off, const int *seg_len, const int *allow_lo,
                               const int *allow_hi, const int *suppress, int nsuppress, int max_domain, int do_sample, int top_k, float top_p,
                               float temperature, unsigned long long seed, const long *step, long *out, const rwkv7_sample_tail *tail,
                               int min_eos_id, long min_eos_until, rwkv7_stream_t stream) {
    if (rows <= 0 || max_domain <= 0 || nsuppress < 0 ||
        any_null({(const void *)logits, (const void *)seg_off, (const void *)seg_len, (const void *)step, (const void *)out}) ||
        (nsuppress > 0 && !suppress))
        return RWKV7_EINVAL;
    return rwkv7::sample_rows_f32(rows, 1, logits, ld, seg_off, seg_len, allow_lo, allow_hi, suppress, nsuppress, max_domain, do_sample, top_k,
                                  top_p, temperature, seed, step, out, tail, min_eos_id, min_eos_until, (hipStream_t)stream);
}
int rwkv7_ras_step_f32(int V, const float *logits, long *tok, long *recent, long *ptr, long *step_i, long n_ignore, int eos, float top_p,
                       int top_k, int win_size, float tau_r, unsigned long long seed, rwkv7_stream_t stream) {
    if (V <= 0 || any_null({(const void *)logits, (const void *)tok, (const void *)recent, (const void *)ptr, (const void *)step_i}))
        return RWKV7_EINVAL;
    return rwkv7::ras_step_f32(V, logits, tok, recent, ptr, step_i, n_ignore, eos, top_p, top_k, win_size, tau_r, seed, (hipStream_t)stream);
}
int rwkv7_xy_frame_step(int B, int C, int rows, long text_shift, long speech_vocab, long pad, long eos0, long total, const long *eos_list,
                        int n_eos, int reference_termination, const long *nt, long *out, long *row, long *pos, long *unfinished, long *needs,
                        unsigned char *all_done, long *n_rows, rwkv7_stream_t stream) {
    if (B <= 0 || n_eos < 0 || (n_eos > 0 && !eos_list) ||
        any_null({(const void *)nt, (const void *)out, (const void *)row, (const void *)pos, (const void *)unfinished, (const void *)needs,
                  (const void *)all_done, (const void *)n_rows}))
        return RWKV7_EINVAL;
    return rwkv7::xy_frame_step(B, C, rows, text_shift, speech_vocab, pad, eos0, total, eos_list, n_eos, reference_termination, nt, out, row,
                                pos, unfinished, needs, all_done, n_rows, (hipStream_t)stream);
}
int rwkv7_xy_embed_bf16(int B, int C, int D, const void *const *tables_host, const long *row, void *x, rwkv7_stream_t stream) {
    if (B <= 0 || D <= 0 || any_null({(const void *)tables_host, (const void *)row, (const void *)x})) return RWKV7_EINVAL;
    for (int c = 0; c < C && c < 16; c++)
        if (!tables_host[c]) return RWKV7_EINVAL;
    return rwkv7::xy_embed_bf16(B, C, D, tables_host, row, x, (hipStream_t)stream);
}
int rwkv7_debug_tr16(const void *in, const int *addr, void *out, rwkv7_stream_t stream) {
    if (!in || !addr || !out) return RWKV7_EINVAL;
    return rwkv7::chunk_debug_tr16((const uint16_t *)in, addr, (uint16_t *)out, (hipStream_t)stream);
}
int rwkv7_debug_mma32(const float *X, const float *Y, float *D, float *DT, rwkv7_stream_t stream) {
    if (any_null({X, Y, D, DT})) return RWKV7_EINVAL;
    return rwkv7::chunk_debug_mma(X, Y, D, DT, (hipStream_t)stream);
}

}  // extern "C"
