// rwkvtts_amd/csrc/capi.hip -- extern "C" surface of librwkv7_hip.so (declared in include/rwkv7_hip.h).
// Argument checking mirrors the reference's asserts (wkv7_cuda.cu:136, rwkv7_state_fwd_fp16.cu:61,
// rwkv_s2s_single_ffn.py:19-21) but reports through the return code instead of aborting the process.
#include <hip/hip_runtime.h>

#include "../../include/rwkv7_hip.h"

namespace rwkv7 {
int wkv_fwd_bf16(int, int, int, const void *, const void *, const void *, const void *, const void *, const void *,
                 void *, float *, float *, float *, hipStream_t);
int wkv_fwd_f32(int, int, int, const void *, const void *, const void *, const void *, const void *, const void *,
                void *, float *, float *, float *, hipStream_t);
int wkv_bwd_bf16(int, int, int, const void *, const void *, const void *, const void *, const void *, const void *,
                 const void *, const float *, const float *, void *, void *, void *, void *, void *, void *,
                 hipStream_t);
int wkv_bwd_f32(int, int, int, const void *, const void *, const void *, const void *, const void *, const void *,
                const void *, const float *, const float *, void *, void *, void *, void *, void *, void *,
                hipStream_t);
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

#define FWD_BODY(IMPL)                                                                       \
    if (B <= 0 || T <= 0 || H <= 0 || any_null({w, q, k, v, a, b, y})) return RWKV7_EINVAL; \
    if ((s == nullptr) != (sa == nullptr)) return RWKV7_EINVAL;                              \
    if (T % RWKV7_CHUNK_LEN != 0) return RWKV7_ECHUNK;                                       \
    return rwkv7::IMPL(B, T, H, w, q, k, v, a, b, y, s, sa, nullptr, (hipStream_t)stream);

int rwkv7_wkv_fwd_bf16(int B, int T, int H, const void *w, const void *q, const void *k, const void *v,
                       const void *a, const void *b, void *y, float *s, float *sa, rwkv7_stream_t stream) {
    FWD_BODY(wkv_fwd_bf16)
}
int rwkv7_wkv_fwd_f32(int B, int T, int H, const void *w, const void *q, const void *k, const void *v,
                      const void *a, const void *b, void *y, float *s, float *sa, rwkv7_stream_t stream) {
    FWD_BODY(wkv_fwd_f32)
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

#define STATE_BODY(IMPL)                                                                              \
    if (B <= 0 || T <= 0 || H <= 0 || any_null({state, r, w, k, v, a, b, y})) return RWKV7_EINVAL;   \
    if (H * RWKV7_HEAD_SIZE != C) return RWKV7_EHEAD;                                                 \
    return rwkv7::IMPL(B, T, H, w, r, k, v, a, b, y, nullptr, nullptr, state, (hipStream_t)stream);

int rwkv7_wkv_state_fwd_bf16(int B, int T, int C, int H, float *state, const void *r, const void *w,
                             const void *k, const void *v, const void *a, const void *b, void *y,
                             rwkv7_stream_t stream) {
    STATE_BODY(wkv_fwd_bf16)
}
int rwkv7_wkv_state_fwd_f32(int B, int T, int C, int H, float *state, const void *r, const void *w,
                            const void *k, const void *v, const void *a, const void *b, void *y,
                            rwkv7_stream_t stream) {
    STATE_BODY(wkv_fwd_f32)
}

}  // extern "C"
