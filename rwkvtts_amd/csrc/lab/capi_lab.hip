// rwkvtts_amd/csrc/lab/capi_lab.hip -- C entry points of the LAB build only (include/rwkv7_hip_lab.h).
//
// Superseded kernels kept as A/B twins and cross-checks of the shipped ones: the round-3/4 per-chunk gradient kernel
// (lab/wkv7_chunk_bwd9.hip), the first-generation own GEMM (lab/gemm_relusq.hip), the bf16 instantiation of the 4-wave chunked forward
// (csrc/wkv7_chunk_fwd.hip under RWKV7_LAB) and the one-launch decode step (csrc/decode_step.hip under RWKV7_LAB; reached through the
// `persistent` argument of rwkv7_decode_step_*_bf16).  None of this is linked into rwkvtts_amd/lib/librwkv7_hip.so: `python -m
// rwkvtts_amd.build --lab` writes rwkvtts_amd/lib/librwkv7_hip_lab.so (every shipped entry + the ones below), which tools/ab_*.py load
// through RWKV7_HIP_SO.  Each variant is its own entry point: the library has no process-wide switches (threading row).
#include <hip/hip_runtime.h>

#include <initializer_list>

#include "../../../include/rwkv7_hip_lab.h"

namespace rwkv7 {
int chunk_bwd_out9_bf16(int, int, int, const void *, const void *, const void *, const void *, const void *, const void *, const void *,
                        const void *, const float *, const float *, const void *, void *, void *, void *, void *, void *, void *, hipStream_t);
int chunk_fwd4_bf16(int, int, int, const void *, const void *, const void *, const void *, const void *, const void *, const float *, void *,
                    float *, void *, const int *, int, hipStream_t);
int gemm_nt_bf16_variant(int, int, int, const void *, const void *, void *, int, int, hipStream_t);
int gemm_nt_relusq_bwd_bf16(int, int, int, const void *, const void *, const void *, void *, hipStream_t);
}  // namespace rwkv7

namespace {
bool any_null(std::initializer_list<const void *> ps) {
    for (const void *p : ps)
        if (p == nullptr) return true;
    return false;
}
}  // namespace

extern "C" {
int rwkv7_lab_wkv_chunk_bwd_out9_z_bf16(int B, int T, int H, const void *w, const void *q, const void *k, const void *v, const void *a,
                                        const void *b, const void *dy, const void *hs, const float *sa, const float *z, const void *e_vk,
                                        void *dw, void *dq, void *dk, void *dv, void *da, void *db, rwkv7_stream_t stream) {
    if (B <= 0 || T <= 0 || H <= 0 ||
        any_null({w, q, k, v, a, b, dy, hs, (const void *)sa, (const void *)z, e_vk, dw, dq, dk, dv, da, db}))
        return RWKV7_EINVAL;
    if (T % 32 != 0) return RWKV7_ECHUNK;
    return rwkv7::chunk_bwd_out9_bf16(B, T, H, w, q, k, v, a, b, dy, hs, sa, z, e_vk, dw, dq, dk, dv, da, db, (hipStream_t)stream);
}
int rwkv7_lab_wkv_chunk_fwd4_seq_bf16(int B, int T, int H, const void *w, const void *q, const void *k, const void *v, const void *a,
                                      const void *b, const float *tinv, void *y, float *sa, void *hs, const int *seq_chunk_off, int nseq,
                                      rwkv7_stream_t stream) {
    if (B <= 0 || T <= 0 || H <= 0 || any_null({w, q, k, v, a, b, tinv, y})) return RWKV7_EINVAL;
    if ((sa == nullptr) != (hs == nullptr) || (seq_chunk_off != nullptr && nseq <= 0)) return RWKV7_EINVAL;
    if (T % RWKV7_CHUNK_T != 0) return RWKV7_ECHUNK;
    return rwkv7::chunk_fwd4_bf16(B, T, H, w, q, k, v, a, b, tinv, y, sa, hs, seq_chunk_off, nseq, (hipStream_t)stream);
}
int rwkv7_lab_gemm_nt_gen1_bf16(int M, int N, int K, const void *A, const void *W, void *C, int epilogue, int variant,
                                rwkv7_stream_t stream) {
    if (any_null({A, W, (const void *)C})) return RWKV7_EINVAL;
    if (M <= 0 || N <= 0 || K <= 0 || M % 256 != 0 || N % 256 != 0 || K % 64 != 0 || epilogue < 0 || epilogue > 1 || variant < 0 || variant > 1)
        return RWKV7_ESHAPE;
    return rwkv7::gemm_nt_bf16_variant(M, N, K, A, W, C, epilogue, variant, (hipStream_t)stream);
}
int rwkv7_lab_gemm_nt_relusq_bwd_gen1_bf16(int M, int N, int K, const void *A, const void *W, const void *aux, void *C,
                                           rwkv7_stream_t stream) {
    if (any_null({A, W, aux, (const void *)C})) return RWKV7_EINVAL;
    if (M <= 0 || N <= 0 || K <= 0 || M % 256 != 0 || N % 256 != 0 || K % 64 != 0) return RWKV7_ESHAPE;
    return rwkv7::gemm_nt_relusq_bwd_bf16(M, N, K, A, W, aux, C, (hipStream_t)stream);
}
}  // extern "C"
