/*
 * include/rwkv7_hip_lab.h -- extra C entry points of the LAB build (rwkvtts_amd/lib/librwkv7_hip_lab.so,
 * `python -m rwkvtts_amd.build --lab`).  NOT part of the drop-in boundary and NOT in the shipped library.
 *
 * The lab library contains every entry of include/rwkv7_hip.h plus the superseded kernels below, each under its OWN entry
 * point (no process-wide switches: same contract as the shipped ABI -- stream-ordered, re-entrant, caller-owned memory,
 * int return).  tools/ab_bwd_out.py, tools/ab_kernel.py and the lab cases of tests/ load it through RWKV7_HIP_SO.
 * With the lab build, `persistent = 1` of rwkv7_decode_step_*_bf16 runs the one-launch decode step (csrc/decode_step.hip).
 */
#ifndef RWKV7_HIP_LAB_H
#define RWKV7_HIP_LAB_H

#include "rwkv7_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* the round-3/4 per-chunk gradient kernel (csrc/lab/wkv7_chunk_bwd9.hip): arguments of rwkv7_wkv_chunk_bwd_out_z_bf16, same arithmetic */
int rwkv7_lab_wkv_chunk_bwd_out9_z_bf16(int B, int T, int H, const void *w, const void *q, const void *k, const void *v, const void *a,
                                        const void *b, const void *dy, const void *hs, const float *sa, const float *z, const void *e_vk,
                                        void *dw, void *dq, void *dk, void *dv, void *da, void *db, rwkv7_stream_t stream);
/* the bf16 instantiation of the 4-wave chunked forward (csrc/wkv7_chunk_fwd.hip; fp32 tensors run that kernel in the shipped library):
 * arguments of rwkv7_wkv_chunk_fwd_seq_bf16 */
int rwkv7_lab_wkv_chunk_fwd4_seq_bf16(int B, int T, int H, const void *w, const void *q, const void *k, const void *v, const void *a,
                                      const void *b, const float *tinv, void *y, float *sa, void *hs, const int *seq_chunk_off, int nseq,
                                      rwkv7_stream_t stream);
/* the first-generation own GEMM (csrc/lab/gemm_relusq.hip; 256 x 256 x 64 tiles, K % 64 == 0): C = epi(A . W^T), epilogue 0 none /
 * 1 relu(.)^2; variant 0 = K tile 64, two LDS buffers; 1 = K tile 32, four buffers, three tiles in flight */
int rwkv7_lab_gemm_nt_gen1_bf16(int M, int N, int K, const void *A, const void *W, void *C, int epilogue, int variant,
                                rwkv7_stream_t stream);
/* C = bf16(A . W^T) * 2 relu(aux) on the first-generation kernel */
int rwkv7_lab_gemm_nt_relusq_bwd_gen1_bf16(int M, int N, int K, const void *A, const void *W, const void *aux, void *C,
                                           rwkv7_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* RWKV7_HIP_LAB_H */
