/*
 * include/rwkv7_hip.h -- C ABI of librwkv7_hip.so: the MI355X (gfx950) RWKV-7 time-mix hot path.
 *
 * This is the drop-in boundary.  Every entry point takes plain device pointers, sizes and a HIP
 * stream; nothing here knows about torch.  The caller owns ALL memory including outputs and scratch
 * (as in the reference: y/s/sa/grads are torch.empty_like'd by the Python wrapper,
 * model/llm/rwkv_s2s_single_ffn.py:22-24,33) and the ops mutate in place and return an int:
 *      0   success
 *     <0   argument error (RWKV7_E*)          -- nothing was launched
 *     >0   hipError_t from the launch          -- hipGetLastError() after the launch
 * Kernels are stream-ordered, re-entrant and keep no global state, so they may be called from
 * several host threads on different streams (service/tts_service.py:42-60 runs one thread per engine).
 *
 * Tensor conventions (reference: model/llm/cuda/wkv7_cuda.cu:18, rwkv7_state_fwd_fp16.cu:16,27):
 *   w,q,k,v,a,b,y,dy,d*  [B,T,H,64] contiguous  == [B,T,C] with C = H*64
 *   w is the PRE-activation: the decay used is exp(-exp(w))            (wkv7_cuda.cu:21)
 *   s   fp32 [B,H,T/16,64,64], checkpoint of the state after every 16th step, stored TRANSPOSED
 *       (s[..., j, i] = S[i][j], wkv7_cuda.cu:45-48)
 *   sa  fp32 [B,T,H,64],  sa[i] = sum_j a[j] * S_{t-1}[i][j]             (wkv7_cuda.cu:27-32)
 *   state fp32 [B,H,64,64], row = value index, col = key index          (rwkv7_state_fwd_fp16.cu:16)
 * The *_bf16 functions take bf16 tensors (the reference's only dtype); the *_f32 twins take fp32
 * tensors and exist for the fp32 logit-parity path.
 */
#ifndef RWKV7_HIP_H
#define RWKV7_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RWKV7_OK 0
#define RWKV7_EINVAL (-1)   /* null pointer / non-positive size */
#define RWKV7_ECHUNK (-2)   /* T % 16 != 0   (assert, wkv7_cuda.cu:136; rwkv_s2s_single_ffn.py:19) */
#define RWKV7_EHEAD  (-3)   /* H*64 != C     (assert, rwkv7_state_fwd_fp16.cu:61) */
#define RWKV7_ESHAPE (-4)   /* unsupported size for a fused elementwise op */

#define RWKV7_HEAD_SIZE 64
#define RWKV7_CHUNK_LEN 16

typedef void *rwkv7_stream_t; /* hipStream_t */

/* library identification: "rwkv7_hip <version> gfx950" */
const char *rwkv7_version(void);

/* ---- WKV7 training forward: torch.ops.wind_backstepping.forward (model/llm/cuda/wkv7_op.cpp:21-22,
 *      kernel wkv7_cuda.cu:10-52).  Zero initial state.  s and sa may both be NULL (inference). ---- */
int rwkv7_wkv_fwd_bf16(int B, int T, int H, const void *w, const void *q, const void *k, const void *v,
                       const void *a, const void *b, void *y, float *s, float *sa, rwkv7_stream_t stream);
int rwkv7_wkv_fwd_f32(int B, int T, int H, const void *w, const void *q, const void *k, const void *v,
                      const void *a, const void *b, void *y, float *s, float *sa, rwkv7_stream_t stream);

/* ---- WKV7 training backward: torch.ops.wind_backstepping.backward (wkv7_op.cpp:23-24,
 *      kernel wkv7_cuda.cu:54-130).  Output order dw,dq,dk,dv,da,db == reference dw,dq,dk,dv,dz,da. ---- */
int rwkv7_wkv_bwd_bf16(int B, int T, int H, const void *w, const void *q, const void *k, const void *v,
                       const void *a, const void *b, const void *dy, const float *s, const float *sa,
                       void *dw, void *dq, void *dk, void *dv, void *da, void *db, rwkv7_stream_t stream);
int rwkv7_wkv_bwd_f32(int B, int T, int H, const void *w, const void *q, const void *k, const void *v,
                      const void *a, const void *b, const void *dy, const float *s, const float *sa,
                      void *dw, void *dq, void *dk, void *dv, void *da, void *db, rwkv7_stream_t stream);

/* ---- The same pair on the matrix cores ("fast" twins; bf16, T % 32 == 0): what torch.ops.wind_backstepping.forward / .backward
 *      launch for bf16 tensors whose T is a multiple of 32 (rwkvtts_amd/ops.py), i.e. what a maintainer who binds the reference's op
 *      (wkv7_op.cpp:21-29) gets.  Same schema, same caller-allocated tensors, same y / sa / gradients (sa in the reference's
 *      layout, fp32 [B,T,H,64]).  `s` -- in the reference a private forward -> backward scratch of state checkpoints (its content is
 *      read by nothing but the backward kernel, rwkv_s2s_single_ffn.py:22-35) -- is used as an opaque arena of the SAME size:
 *      per 32-step chunk and head 9216 B of state checkpoint (q15 record, see rwkv7_wkv_chunk_fwd_bf16) + 4096 B of
 *      T = (I - A_ab)^-1 + 9216 B of adjoint state and 8192 B of Z written by the backward = 30 720 of the 32 768 B the
 *      reference's layout has there.  Launches: chunk_prep + chunk_fwd (forward); chunk_bseq + chunk_bwd_out_z (backward).  RWKV7_ECHUNK when T % 32 != 0:
 *      fall back to rwkv7_wkv_fwd_bf16 / rwkv7_wkv_bwd_bf16 (T % 16 == 0), whose `s` holds the reference's checkpoints. ---- */
int rwkv7_wkv_fwd_fast_bf16(int B, int T, int H, const void *w, const void *q, const void *k, const void *v,
                            const void *a, const void *b, void *y, float *s, float *sa, rwkv7_stream_t stream);
int rwkv7_wkv_bwd_fast_bf16(int B, int T, int H, const void *w, const void *q, const void *k, const void *v,
                            const void *a, const void *b, const void *dy, float *s, const float *sa,
                            void *dw, void *dq, void *dk, void *dv, void *da, void *db, rwkv7_stream_t stream);

/* Sizes of the two caller-allocated scratch tensors of the training forward/backward pair (the reference allocates
 * them in Python, rwkv_s2s_single_ffn.py:22-24): s = fp32 [B,H,T/16,64,64] state checkpoints, sa = fp32 [B,T,H,64]. */
int rwkv7_wkv_workspace_bytes(int B, int T, int H, size_t *s_bytes, size_t *sa_bytes);

/* Measurement / cross-check variants.  The library keeps NO process-global state: every entry point is re-entrant and safe
 * from several host threads on different streams (threading row); which kernel shape runs is an explicit
 * argument of these `_variant` twins, never a hidden switch that changes what the plain entry points above launch.
 *   cols_per_lane : state columns per lane of the scalar forward kernel -- 0 automatic by B*H (what the plain entry does), 4, 8
 *   wide          : row-split backward -- 0 = 256 threads, 2 state rows per lane tile (plain entry); 1 = 512 threads, 1 row
 *   (The chunked bf16 forward has one shipped kernel, csrc/wkv7_chunk_fwd9.hip; the bf16 instantiation of the 4-wave kernel that fp32
 *   tensors run is a lab-build entry, include/rwkv7_hip_lab.h.) */
int rwkv7_wkv_fwd_variant_bf16(int B, int T, int H, const void *w, const void *q, const void *k, const void *v,
                               const void *a, const void *b, void *y, float *s, float *sa, int cols_per_lane, rwkv7_stream_t stream);
int rwkv7_wkv_fwd_variant_f32(int B, int T, int H, const void *w, const void *q, const void *k, const void *v,
                              const void *a, const void *b, void *y, float *s, float *sa, int cols_per_lane, rwkv7_stream_t stream);
int rwkv7_wkv_state_fwd_variant_bf16(int B, int T, int C, int H, float *state, const void *r, const void *w,
                                     const void *k, const void *v, const void *a, const void *b, void *y, int cols_per_lane,
                                     rwkv7_stream_t stream);
int rwkv7_wkv_bwd_split_variant_bf16(int B, int T, int H, const void *w, const void *q, const void *k, const void *v,
                                     const void *a, const void *b, const void *dy, const float *s, const float *sa,
                                     void *const *dw, void *const *dq, void *const *dk, void *dv, void *const *da,
                                     void *const *db, int wide, rwkv7_stream_t stream);

/* ---- same backward with each head split over two workgroups (32 state rows each) so that 256 CUs are busy at
 *      B*H = 128.  dv is complete; dw,dq,dk,da,db are HOST arrays of 2 device pointers receiving the two partial
 *      column sums (final gradient = [0] + [1]); rwkv7_tmix_prepare_bwd_* can consume the pairs directly. ---- */
int rwkv7_wkv_bwd_split_bf16(int B, int T, int H, const void *w, const void *q, const void *k, const void *v,
                             const void *a, const void *b, const void *dy, const float *s, const float *sa,
                             void *const *dw, void *const *dq, void *const *dk, void *dv, void *const *da,
                             void *const *db, rwkv7_stream_t stream);
int rwkv7_wkv_bwd_split_f32(int B, int T, int H, const void *w, const void *q, const void *k, const void *v,
                            const void *a, const void *b, const void *dy, const float *s, const float *sa,
                            void *const *dw, void *const *dq, void *const *dk, void *dv, void *const *da,
                            void *const *db, rwkv7_stream_t stream);

/* ---- state-carrying forward: torch.ops.rwkv7_state_fwd_fp16.forward (rwkv7_state_fwd_fp16.cpp:8-14,
 *      kernel rwkv7_state_fwd_fp16.cu:9-57) and its B=1 twin torch.ops.wkv7s.forward
 *      (wkv7s_op.cpp:9-15).  state is read at entry and overwritten at exit; any T >= 1. ---- */
int rwkv7_wkv_state_fwd_bf16(int B, int T, int C, int H, float *state, const void *r, const void *w,
                             const void *k, const void *v, const void *a, const void *b, void *y,
                             rwkv7_stream_t stream);
int rwkv7_wkv_state_fwd_f32(int B, int T, int C, int H, float *state, const void *r, const void *w,
                            const void *k, const void *v, const void *a, const void *b, void *y,
                            rwkv7_stream_t stream);

/* =====================================================================================================
 * Fused elementwise stages of the time-mix / channel-mix blocks.  Activations are [rows = B*T, D]
 * row-major, same dtype as the parameters (bf16 or fp32); `mask` is [rows] of that dtype or NULL;
 * D % 64 == 0 and D <= 8192.  `nblocks` = number of workgroups walking the rows; backward kernels
 * write per-workgroup fp32 partials of the parameter gradients, dparams_partial[nblocks][P][D], which
 * the caller sums over the first axis.
 * ===================================================================================================== */

/* token shift + nmix lerps (nmix = 6: x_r,x_w,x_k,x_v,x_a,x_g -- rwkv_s2s_single_ffn.py:160-169; nmix = 3: x_r,x_k,x_v when the
 * low-rank branches take their inputs through the lerp, fused.mix_lora;
 * nmix = 1: channel-mix x_k -- :224-227):  xm = x*mask ; out[i] = xm + (shift(xm) - xm) * params[i].
 * x_prev [B,D] (carried token-shift state, rwkv_asr_cuda_whisper.py:185) or NULL = zeros.
 * out is [nmix][rows][D]. */
int rwkv7_mix_fwd_bf16(int B, int T, int D, int nmix, const void *x, const void *x_prev, const void *mask,
                       const void *params, void *out, int nblocks, rwkv7_stream_t stream);
int rwkv7_mix_fwd_f32(int B, int T, int D, int nmix, const void *x, const void *x_prev, const void *mask,
                      const void *params, void *out, int nblocks, rwkv7_stream_t stream);
/* grad_outs: HOST array of nmix device pointers, one [rows][D] gradient per output of the forward.
 * Workgroup b walks the runs of run_len consecutive rows number b, b + nblocks, b + 2 nblocks, ...;
 * dparams_partial is [nblocks][nmix][D] fp32 (summed by the caller). */
int rwkv7_mix_bwd_bf16(int B, int T, int D, int nmix, const void *const *grad_outs, const void *x, const void *x_prev,
                       const void *mask, const void *params, void *dx, float *dparams_partial, int nblocks,
                       int run_len, rwkv7_stream_t stream);
int rwkv7_mix_bwd_f32(int B, int T, int D, int nmix, const void *const *grad_outs, const void *x, const void *x_prev,
                      const void *mask, const void *params, void *dx, float *dparams_partial, int nblocks,
                      int run_len, rwkv7_stream_t stream);

/* everything between the projections and the scan (rwkv_s2s_single_ffn.py:172-190):
 *   w = (-softplus(-w_pre) - 0.5)*mask ; k,v *= mask ; v += (v_first - v)*sigmoid(v_pre) (v_pre != NULL)
 *   a = sigmoid(a_pre) ; kk = l2norm_head(k*k_k)*mask ; k2 = k*(1 + (a-1)*k_a) ; v2 = v*mask
 * outputs w, k2, v2, ain = -kk, bin = kk*a (the scan's w,k,v,a,b).  v_pre/v_first NULL for layer 0.
 * backward partials: P = 5 (dk_k, dk_a, and the column sums of d_wpre, d_apre, d_vpre -- the bias gradients of the
 * low-rank branches that produced w_pre, a_pre, v_pre; the last is zero when v_pre is NULL). */
int rwkv7_tmix_prepare_fwd_bf16(long rows, int D, const void *w_pre, const void *k, const void *v, const void *a_pre,
                                const void *v_pre, const void *v_first, const void *mask, const void *k_k,
                                const void *k_a, void *w, void *k2, void *v2, void *ain, void *bin, int nblocks,
                                rwkv7_stream_t stream);
int rwkv7_tmix_prepare_fwd_f32(long rows, int D, const void *w_pre, const void *k, const void *v, const void *a_pre,
                               const void *v_pre, const void *v_first, const void *mask, const void *k_k,
                               const void *k_a, void *w, void *k2, void *v2, void *ain, void *bin, int nblocks,
                               rwkv7_stream_t stream);
int rwkv7_tmix_prepare_bwd_bf16(long rows, int D, const void *w_pre, const void *k, const void *v, const void *a_pre,
                                const void *v_pre, const void *v_first, const void *mask, const void *k_k,
                                const void *k_a, const void *d_w, const void *d_k2, const void *d_v2,
                                const void *d_ain, const void *d_bin, void *d_wpre, void *d_k, void *d_v,
                                void *d_apre, void *d_vpre, void *d_vfirst, float *dparams_partial, int nblocks,
                                rwkv7_stream_t stream);
int rwkv7_tmix_prepare_bwd_f32(long rows, int D, const void *w_pre, const void *k, const void *v, const void *a_pre,
                               const void *v_pre, const void *v_first, const void *mask, const void *k_k,
                               const void *k_a, const void *d_w, const void *d_k2, const void *d_v2,
                               const void *d_ain, const void *d_bin, void *d_wpre, void *d_k, void *d_v,
                               void *d_apre, void *d_vpre, void *d_vfirst, float *dparams_partial, int nblocks,
                               rwkv7_stream_t stream);

/* Same backward with the incoming gradients given as sums, added in fp32 on load (no separate add kernels): gsum is a
 * HOST array of 15 device pointers {d_w a,b; d_k2 a,b,c; d_v2 a,b; d_ain a,b; d_bin a,b; d_r a,b,c; d_vfirst_in}, and
 * d_r = d_r a+b+c is written as well.  Consumes the two partial sets of rwkv7_wkv_bwd_split_* plus tmix_post's
 * contributions to k2, v2 and r.  The second partials (d_w b, d_k2 b, d_ain b, d_bin b, d_r b) may be NULL: complete
 * gradients, as the chunked backward produces them.  d_vfirst_in (may be NULL): the gradient of v_first collected by the
 * layers after this one (v_first of layer 0 feeds every later layer, rwkv_s2s_single_ffn.py:179-182); d_vfirst = own + d_vfirst_in,
 * so the sum over the layers is formed layer by layer inside this kernel instead of by one [rows, D] add pass per layer. */
int rwkv7_tmix_prepare_bwd_sum_bf16(long rows, int D, const void *w_pre, const void *k, const void *v,
                                    const void *a_pre, const void *v_pre, const void *v_first, const void *mask,
                                    const void *k_k, const void *k_a, const void *const *gsum, void *d_wpre, void *d_k,
                                    void *d_v, void *d_apre, void *d_vpre, void *d_vfirst, void *d_r,
                                    float *dparams_partial, int nblocks, rwkv7_stream_t stream);
int rwkv7_tmix_prepare_bwd_sum_f32(long rows, int D, const void *w_pre, const void *k, const void *v,
                                   const void *a_pre, const void *v_pre, const void *v_first, const void *mask,
                                   const void *k_k, const void *k_a, const void *const *gsum, void *d_wpre, void *d_k,
                                   void *d_v, void *d_apre, void *d_vpre, void *d_vfirst, void *d_r,
                                   float *dparams_partial, int nblocks, rwkv7_stream_t stream);
/* The same with the COMPACT hand-off from rwkv7_tmix_post_bwd_compact_* (what bf16 training runs since round 4): the bonus term of
 * tmix_post contributes rank-1 per head -- d_v2 += dt * dot_h, d_k2 += ds_h r r_k, d_r += ds_h k2 r_k -- so the post backward writes
 * one tensor and two scalars per (row, head) instead of three tensors, and this kernel rebuilds them (k2 recomputed from k, a_pre,
 * k_a as the forward did).  gsum: HOST array of 19 device pointers = the 15 above (entries 4, 6, 13 -- the post tensors -- unused,
 * may be NULL) + {dt [rows,D], r [rows,D], r_k [D], hscal fp32 [rows][D/64][2] = (dot_h, ds_h)}. */
int rwkv7_tmix_prepare_bwd_sum_compact_bf16(long rows, int D, const void *w_pre, const void *k, const void *v,
                                            const void *a_pre, const void *v_pre, const void *v_first, const void *mask,
                                            const void *k_k, const void *k_a, const void *const *gsum, void *d_wpre, void *d_k,
                                            void *d_v, void *d_apre, void *d_vpre, void *d_vfirst, void *d_r,
                                            float *dparams_partial, int nblocks, rwkv7_stream_t stream);
int rwkv7_tmix_prepare_bwd_sum_compact_f32(long rows, int D, const void *w_pre, const void *k, const void *v,
                                           const void *a_pre, const void *v_pre, const void *v_first, const void *mask,
                                           const void *k_k, const void *k_a, const void *const *gsum, void *d_wpre, void *d_k,
                                           void *d_v, void *d_apre, void *d_vpre, void *d_vfirst, void *d_r,
                                           float *dparams_partial, int nblocks, rwkv7_stream_t stream);

/* ---- residual add + LayerNorm (block wiring of rwkv_s2s_single_ffn.py:262-276: x = x + att(ln1(x)); x = x + ffn(ln2(x));
 *      rwkvfla RWKV7Block attn_norm / ffn_norm / pre_norm, model-level norm) ----
 *   fwd: x_out = x + branch (rounded to the tensor type) ; h = LayerNorm(x_out; gamma, beta, eps) ; mean/rstd [rows] fp32.
 *        branch NULL: plain LayerNorm of x (x_out unused).  beta NULL: no bias.
 *   bwd: dx = d_resid + dLayerNorm(dh)   (d_resid NULL = 0) ; dparams_partial [nblocks][2][D] fp32 = dgamma, dbeta */
int rwkv7_add_ln_fwd_bf16(long rows, int D, const void *x, const void *branch, const void *gamma, const void *beta,
                          float eps, void *x_out, void *h, float *mean, float *rstd, int nblocks,
                          rwkv7_stream_t stream);
int rwkv7_add_ln_fwd_f32(long rows, int D, const void *x, const void *branch, const void *gamma, const void *beta,
                         float eps, void *x_out, void *h, float *mean, float *rstd, int nblocks,
                         rwkv7_stream_t stream);
int rwkv7_add_ln_bwd_bf16(long rows, int D, const void *dh, const void *d_resid, const void *x1, const float *mean,
                          const float *rstd, const void *gamma, void *dx, float *dparams_partial, int nblocks,
                          rwkv7_stream_t stream);
int rwkv7_add_ln_bwd_f32(long rows, int D, const void *dh, const void *d_resid, const void *x1, const float *mean,
                         const float *rstd, const void *gamma, void *dx, float *dparams_partial, int nblocks,
                         rwkv7_stream_t stream);
/* ---- the two stages above in one pass each (training path without carried state): residual add + LayerNorm + token-shift
 *      lerps, i.e. everything between a block's branch output and the inputs of the next projections
 *      (rwkv_s2s_single_ffn.py:251-259 with :160-169 / :223-226).
 *   fwd: x_out = x + branch ; h = LayerNorm(x_out) rounded to the tensor type ; hm = h*mask ;
 *        out[i] = hm + (shift(hm) - hm) * params[i], out [nmix][rows][D] ; mean/rstd [rows].  branch NULL: x_out unused.
 *   bwd: from the nmix output gradients (HOST array of device pointers) and the gradient d_resid arriving at x_out:
 *        dx (= gradient of x and of branch) ; dparams_partial [nblocks][nmix + 2][D] fp32 = dparams[0..nmix), dgamma, dbeta.
 *   Workgroup b walks the runs of run_len consecutive rows number b, b + nblocks, ... ---- */
int rwkv7_add_ln_mix_fwd_bf16(int B, int T, int D, int nmix, const void *x, const void *branch, const void *gamma,
                              const void *beta, float eps, const void *mask, const void *params, void *x_out, void *out,
                              float *mean, float *rstd, int nblocks, int run_len, rwkv7_stream_t stream);
int rwkv7_add_ln_mix_fwd_f32(int B, int T, int D, int nmix, const void *x, const void *branch, const void *gamma,
                             const void *beta, float eps, const void *mask, const void *params, void *x_out, void *out,
                             float *mean, float *rstd, int nblocks, int run_len, rwkv7_stream_t stream);
/*   fwd_h: the same forward that ALSO stores h [rows][D] (unmasked), for callers whose backward runs as the two separate kernels
 *        rwkv7_mix_bwd_* + rwkv7_add_ln_bwd_* (the six-lerp side: its one-pass backward carries 6 x 3 x 8 values per thread and
 *        spills; the one-pass forward is 18 us per layer ahead of the two stages even with the extra store). */
int rwkv7_add_ln_mix_fwd_h_bf16(int B, int T, int D, int nmix, const void *x, const void *branch, const void *gamma,
                                const void *beta, float eps, const void *mask, const void *params, void *x_out, void *out, void *h,
                                float *mean, float *rstd, int nblocks, int run_len, rwkv7_stream_t stream);
int rwkv7_add_ln_mix_fwd_h_f32(int B, int T, int D, int nmix, const void *x, const void *branch, const void *gamma,
                               const void *beta, float eps, const void *mask, const void *params, void *x_out, void *out, void *h,
                               float *mean, float *rstd, int nblocks, int run_len, rwkv7_stream_t stream);
int rwkv7_mix_add_ln_bwd_bf16(int B, int T, int D, int nmix, const void *const *grad_outs, const void *d_resid, const void *x1,
                              const float *mean, const float *rstd, const void *gamma, const void *beta, const void *mask,
                              const void *params, void *dx, float *dparams_partial, int nblocks, int run_len,
                              rwkv7_stream_t stream);
int rwkv7_mix_add_ln_bwd_f32(int B, int T, int D, int nmix, const void *const *grad_outs, const void *d_resid, const void *x1,
                             const float *mean, const float *rstd, const void *gamma, const void *beta, const void *mask,
                             const void *params, void *dx, float *dparams_partial, int nblocks, int run_len,
                             rwkv7_stream_t stream);

/* after the scan (rwkv_s2s_single_ffn.py:192-195): out = (GroupNorm_H(y; gn_w, gn_b, eps) + (sum_head r*k*r_k) v) * g.
 * r_k is [H*64] flattened.  backward partials: P = 3 (d gn_w, d gn_b, d r_k). */
int rwkv7_tmix_post_fwd_bf16(long rows, int D, const void *y, const void *r, const void *k, const void *v,
                             const void *g, const void *gn_w, const void *gn_b, const void *r_k, float eps, void *out,
                             int nblocks, rwkv7_stream_t stream);
int rwkv7_tmix_post_fwd_f32(long rows, int D, const void *y, const void *r, const void *k, const void *v,
                            const void *g, const void *gn_w, const void *gn_b, const void *r_k, float eps, void *out,
                            int nblocks, rwkv7_stream_t stream);
int rwkv7_tmix_post_bwd_bf16(long rows, int D, const void *dout, const void *y, const void *r, const void *k,
                             const void *v, const void *g, const void *gn_w, const void *gn_b, const void *r_k,
                             float eps, void *d_y, void *d_r, void *d_k, void *d_v, void *d_g, float *dparams_partial,
                             int nblocks, rwkv7_stream_t stream);
int rwkv7_tmix_post_bwd_f32(long rows, int D, const void *dout, const void *y, const void *r, const void *k,
                            const void *v, const void *g, const void *gn_w, const void *gn_b, const void *r_k,
                            float eps, void *d_y, void *d_r, void *d_k, void *d_v, void *d_g, float *dparams_partial,
                            int nblocks, rwkv7_stream_t stream);
/* compact form: d_y, d_g as above; dt = dL/d(GroupNorm(y) + bonus) [rows,D] and hscal fp32 [rows][D/64][2] = (sum_head r k r_k,
 * sum_head dt v) instead of d_r, d_k, d_v (rebuilt by rwkv7_tmix_prepare_bwd_sum_compact_*) */
int rwkv7_tmix_post_bwd_compact_bf16(long rows, int D, const void *dout, const void *y, const void *r, const void *k,
                                     const void *v, const void *g, const void *gn_w, const void *gn_b, const void *r_k,
                                     float eps, void *d_y, void *dt, void *d_g, float *hscal, float *dparams_partial,
                                     int nblocks, rwkv7_stream_t stream);
int rwkv7_tmix_post_bwd_compact_f32(long rows, int D, const void *dout, const void *y, const void *r, const void *k,
                                    const void *v, const void *g, const void *gn_w, const void *gn_b, const void *r_k,
                                    float eps, void *d_y, void *dt, void *d_g, float *hscal, float *dparams_partial,
                                    int nblocks, rwkv7_stream_t stream);

/* channel-mix activation relu(x)^2 (rwkv_s2s_single_ffn.py:228) and dx = 2 relu(x) dy; n % 8 == 0 */
int rwkv7_relusq_fwd_bf16(long n, const void *x, void *y, rwkv7_stream_t stream);
int rwkv7_relusq_fwd_f32(long n, const void *x, void *y, rwkv7_stream_t stream);
int rwkv7_relusq_bwd_bf16(long n, const void *x, const void *dy, void *dx, rwkv7_stream_t stream);
int rwkv7_relusq_bwd_f32(long n, const void *x, const void *dy, void *dx, rwkv7_stream_t stream);
/* the same backward from the OUTPUT s = relu(x)^2 (dx = 2 sqrt(s) dy): rwkv7_gemm_nt_bf16 with the activation as its epilogue never
 * writes x */
int rwkv7_relusq_bwd_s_bf16(long n, const void *s, const void *dy, void *dx, rwkv7_stream_t stream);
int rwkv7_relusq_bwd_s_f32(long n, const void *s, const void *dy, void *dx, rwkv7_stream_t stream);

/* =====================================================================================================
 * Chunked (MFMA) WKV7 -- the training fast path.  Same operator as rwkv7_wkv_fwd/bwd (reference
 * wkv7_cuda.cu:10-130), evaluated 32 steps at a time on the matrix cores (algebra: csrc/chunk_common.h).
 * Requires T % 32 == 0 and exp(w) <= ~2 per step (the model's soft-clamped decay satisfies it:
 * w <= -0.5, rwkv_s2s_single_ffn.py:172).  Scratch / saved tensors, all caller-allocated fp32:
 *   tinv [B,H,T/32,32,32]   (I - A_ab)^-1 per chunk, written by _prep, read by _fwd and _bwd
 *   sa   [B,T,H,64]         u_t = S_{t-1} a_t (same meaning as the scalar op's `sa`)
 *   hs   [B,H,T/32,64,64]   state at the START of each chunk, bf16, [value][key]: the backward's checkpoint (the forward itself
 *                           carries the state in fp32; the checkpoint is its bf16 rounding, 8 KB per chunk and head)
 * sa and hs may both be NULL in _fwd (inference).
 * ===================================================================================================== */
#define RWKV7_CHUNK_T 32
/* One 64x64 fp32 matrix handed between the chunk kernels (hs, e_vk, np) as a "q15" record: 4096 int16 mantissas in MFMA
 * accumulator order [tile][lane][16] followed by 256 fp32 scales [tile][lane] (x = mantissa * scale of its lane); see
 * csrc/chunk_common.h.  Sizes of hs / e_vk / np buffers: RWKV7_Q15_REC uint16 per (batch, head, chunk). */
#define RWKV7_Q15_REC (64 * 64 + 2 * 256)
int rwkv7_wkv_chunk_prep_bf16(int B, int T, int H, const void *w, const void *a, const void *b, float *tinv,
                              rwkv7_stream_t stream);
int rwkv7_wkv_chunk_prep_f32(int B, int T, int H, const void *w, const void *a, const void *b, float *tinv,
                             rwkv7_stream_t stream);
int rwkv7_wkv_chunk_fwd_bf16(int B, int T, int H, const void *w, const void *q, const void *k, const void *v,
                             const void *a, const void *b, const float *tinv, void *y, float *sa, void *hs,
                             rwkv7_stream_t stream);
int rwkv7_wkv_chunk_fwd_f32(int B, int T, int H, const void *w, const void *q, const void *k, const void *v,
                            const void *a, const void *b, const float *tinv, void *y, float *sa, void *hs,
                            rwkv7_stream_t stream);
/* Packed variable-length rows (fla chunk_rwkv7's `cu_seqlens`; the reference passes it for its packed Spark batches,
 * train_spark_rwkv7speech.py:238-239, data/utils/spark_dataset.py:111-162): the caller lays the sequences out 32-aligned
 * and passes seq_chunk_off, int32 [nseq + 1] on the device: sequence s owns the 32-step chunks seq_chunk_off[s] ..
 * seq_chunk_off[s+1] - 1, counted over the whole [B][T/32] chunk space (a sequence does not span rows).  Every sequence
 * starts from the zero state and gets its own workgroups, so the sequences of a row run in parallel; chunks that belong to
 * no sequence are left untouched.  seq_chunk_off == NULL: the plain ops above (one sequence per row). */
int rwkv7_wkv_chunk_fwd_seq_bf16(int B, int T, int H, const void *w, const void *q, const void *k, const void *v,
                                 const void *a, const void *b, const float *tinv, void *y, float *sa, void *hs,
                                 const int *seq_chunk_off, int nseq, rwkv7_stream_t stream);
int rwkv7_wkv_chunk_fwd_seq_f32(int B, int T, int H, const void *w, const void *q, const void *k, const void *v,
                                const void *a, const void *b, const float *tinv, void *y, float *sa, void *hs,
                                const int *seq_chunk_off, int nseq, rwkv7_stream_t stream);
/* ---- chunked backward, bf16 (csrc/wkv7_chunk_bseq.hip, wkv7_chunk_bwd10.hip; reference wkv7_cuda.cu:54-130).  With H = S^T and the
 *      chunk quantities above, the adjoint state obeys E_c = M_c^T E_{c+1} + N'_c.  T % 32 == 0.
 *   bseq    : sequential over chunks (reverse), one workgroup per (head, half of the value columns): the recurrence in factored
 *             form, E_c = E' + A~^T Z + Q~^T dY with Z = (T^T B^) E' + (T^T A_qb^T) dY, E' = g_C E_{c+1} -- M_c^T and N'_c are
 *             never formed and never reach HBM.  e_vk[b,h,c] = E_{c+1}, what chunk c receives from its future, one q15 record
 *             per chunk (int16 [4 tiles][64 lanes][16] + fp32 scale [4][64]; RWKV7_Q15_REC uint16 units; the recurrence itself
 *             carries ~16 mantissa bits, the per-chunk kernel reads this rounded copy once); seq_chunk_off / nseq as in
 *             rwkv7_wkv_chunk_fwd_seq_bf16 (NULL / 0 for plain rows). ---- */
int rwkv7_wkv_chunk_bseq_bf16(int B, int T, int H, const void *w, const void *q, const void *a, const void *b, const void *dy,
                              const float *tinv, void *e_vk, float *z, const int *seq_chunk_off, int nseq, rwkv7_stream_t stream);
/*             z (may be NULL): fp32 [B,T,H,64], Z_t = dL/du_t (u = sa), which the recurrence forms anyway.  With it the per-chunk
 *             gradient kernel needs neither T^-1 nor an A_qb -> G1 -> Z chain of its own.
 *   bwd_out_z : parallel over chunks: the six gradients (the contract of wind_backstepping::backward) from what the chunked
 *             forward saved (hs, sa) and e_vk, z of `bseq`; two matrix phases per chunk (csrc/wkv7_chunk_bwd10.hip: raw rows by
 *             LDS-DMA, unpadded XOR-swizzled planes, state prologue and phase A in one barrier interval). */
int rwkv7_wkv_chunk_bwd_out_z_bf16(int B, int T, int H, const void *w, const void *q, const void *k, const void *v,
                                   const void *a, const void *b, const void *dy, const void *hs, const float *sa,
                                   const float *z, const void *e_vk, void *dw, void *dq, void *dk, void *dv,
                                   void *da, void *db, rwkv7_stream_t stream);
/*      (The round-3/4 per-chunk gradient kernel, csrc/lab/wkv7_chunk_bwd9.hip, is an A/B twin with its own entry point in the lab build:
 *      include/rwkv7_hip_lab.h.  There are no process-wide switches in this library.) */
/* ---- head loss: softmax cross-entropy of a chunk of bf16 logits [rows,V], forward and backward in one pass
 *      (spark_llm.py:146-160, FusedLinearCrossEntropyLoss).  labels int64 [rows]; rows with label == ignore_index give 0.
 *      loss_rows[rows] = logsumexp - logit[label]; logits are REPLACED by (softmax - onehot) * scale. ---- */
int rwkv7_ce_fwd_bwd_bf16(long rows, int V, void *logits, const long *labels, long ignore_index, float scale, float *loss_rows,
                          rwkv7_stream_t stream);
/*      The same with label smoothing (torch.nn.CrossEntropyLoss(label_smoothing = ls), the XY heads: xy_llm.py:233-240):
 *      loss_rows = (1 - ls)(logsumexp - logit[label]) + ls (logsumexp - mean logit); logits REPLACED by
 *      (softmax - ls / V - (1 - ls) onehot) * scale.  0 <= ls < 1; ls = 0 is the entry above. */
int rwkv7_ce_fwd_bwd_ls_bf16(long rows, int V, void *logits, const long *labels, long ignore_index, float scale, float *loss_rows,
                             float label_smoothing, rwkv7_stream_t stream);
/*      The same on logits with a leading dimension ld >= V elements (rows of a buffer padded to aligned rows; columns V .. ld - 1 are
 *      neither read nor written). */
int rwkv7_ce_fwd_bwd_ld_bf16(long rows, int V, long ld, void *logits, const long *labels, long ignore_index, float scale, float *loss_rows,
                             float label_smoothing, rwkv7_stream_t stream);

/* ---- optimizer step (train_spark_rwkv7speech.py:178-197; torch.optim.AdamW update rule, decoupled weight decay) on a
 *      flat parameter buffer: fp32 master weights p32 and moments m, v updated in place from bf16 gradients g16; the
 *      bf16 working copy p16 is rewritten in the same pass.  n % 4 == 0; step = 1 for the first update. ---- */
int rwkv7_adamw_bf16(long n, float *p32, const void *g16, float *m, float *v, void *p16, float lr, float beta1, float beta2,
                     float eps, float weight_decay, int step, rwkv7_stream_t stream);
/*      Parameter groups (train_scripts/train_cosy_rwkv7speech_multiple_dataset.py:162-202: `lr_2x` for
 *      'attn.w_lora.lora.2.bias', a weight-decay group for >= 2-D non-LoRA `.weight`, `my_lr_scale`): slab_group[n / 128]
 *      (uint8, one entry per 128 consecutive elements -- the host aligns every parameter to 128 elements) indexes
 *      group_tab[ngroups][2] = {lr scale, weight decay}, both in device memory; both NULL = one group {1, 0}.
 *      skip_flag: device float or NULL; != 0 makes the step run on a zero gradient (the reference's NaN-loss step,
 *      train_spark_rwkv7speech.py:664-687) without the host ever reading the flag.  n % 128 == 0 with groups. ---- */
int rwkv7_adamw_groups_bf16(long n, float *p32, const void *g16, float *m, float *v, void *p16, const unsigned char *slab_group,
                            const float *group_tab, int ngroups, const float *skip_flag, float lr, float beta1, float beta2,
                            float eps, int step, rwkv7_stream_t stream);

/* ---- last step of a split weight gradient (the dW of nn.Linear under autograd, e.g. rwkv_s2s_single_ffn.py:171-174,195,
 *      228-229, reduced over B*T in S row slabs with fp32 partials): out[n] (bf16) = (accumulate ? out[n] : 0) +
 *      sum_s parts[s][n].  out may be the parameter's slice of the flat gradient buffer.  n % 4 == 0. ---- */
int rwkv7_sum_slabs_bf16(long n, int S, const float *parts, void *out, int accumulate, rwkv7_stream_t stream);
/*   out[C][R] = in[R][C]^T, 16-bit elements, R % 64 == 0 and C % 64 == 0: the NT operand W_value^T of the channel-mix backward's
 *   input-gradient GEMM (rwkv7_gemm_nt_relusq_bwd_s_bf16; autograd of rwkv_s2s_single_ffn.py:229). */
int rwkv7_transpose_bf16(int R, int C, const void *in, void *out, rwkv7_stream_t stream);
/*   out[r][0..D) = idx[r] >= 0 ? src[idx[r]][0..D) : 0 for r < n_out -- 16-bit rows, D % 8 == 0, idx int32 on the device.  The re-layout
 *   of a packed `cu_seqlens` row (train_spark_rwkv7speech.py:238-239, data/utils/spark_dataset.py:111-162) into the 32-aligned row the
 *   chunked WKV7 kernels walk, and back: the maps are injective, so forward and backward of both directions are this one gather. */
int rwkv7_gather_rows_bf16(long n_out, int D, const void *src, const int *idx, void *out, rwkv7_stream_t stream);
/*   Weight gradient of a low-rank projection (rwkv_s2s_single_ffn.py:172-184, autograd of x @ w1 / h @ w2): for y = x W^T,
 *   parts[s][N][K] (fp32) = dy[slab s][N]^T x[slab s][K], slab = M / S consecutive rows (a multiple of 128); one of N, K is the
 *   rank (32, 64 or 128), the other a multiple of 256.  bf16 operands, fp32 accumulation on MFMA; finish with
 *   rwkv7_sum_slabs_bf16(N * K, S, parts, dW, 0). */
int rwkv7_wgrad_skinny_bf16(long M, int N, int K, int S, const void *dy, const void *x, float *parts, rwkv7_stream_t stream);
/*   wgrad_mid: the same partials for the [W_a ; W_b] gradient of the through-the-lerp projections (dy = dG [M][N], N = 2 R in {512, 576};
 *   K a multiple of 256; rows per slab a multiple of 64): neither side is skinny, a workgroup holds a [N / 2][256] fp32 tile. */
int rwkv7_wgrad_mid_bf16(long M, int N, int K, int S, const void *dy, const void *x, float *parts, rwkv7_stream_t stream);

/* ---- C[M][N] = epi(A[M][K] . W[N][K]^T), bf16, fp32 accumulation: the channel-mix key projection with its activation as
 *      the epilogue (epilogue 1: relu(.)^2, rwkv_s2s_single_ffn.py:228; 0: none).  Hand-written persistent MFMA kernel fed by
 *      LDS-DMA (csrc/gemm_nt4.hip); M, N multiples of 256, K of 1024 (RWKV7_ESHAPE otherwise); bit-identical to the library
 *      GEMM + rwkv7_relusq_fwd pair. ---- */
int rwkv7_gemm_nt_bf16(int M, int N, int K, const void *A, const void *W, void *C, int epilogue, rwkv7_stream_t stream);
/*      the backward of the activation as the epilogue of the value projection's input-gradient GEMM (round 4):
 *      C[M][N] = bf16(A[M][K] . W[N][K]^T) * 2 relu(aux[M][N]) -- A = dy, W = value.weight^T (contiguous [N = F][K = D]), aux = the key
 *      projection's output h: dh without ds ever reaching HBM and without rwkv7_relusq_bwd_*. */
int rwkv7_gemm_nt_relusq_bwd_bf16(int M, int N, int K, const void *A, const void *W, const void *aux, void *C, rwkv7_stream_t stream);
/*      the same from the activation's OUTPUT: C[M][N] = bf16(A . W^T) * 2 sqrt(s[M][N]), s = relu(h)^2 as written by rwkv7_gemm_nt_bf16 with
 *      epilogue 1 (what the library GEMM + rwkv7_relusq_bwd_s_* produce, bit for bit): with it the channel mix never materialises h or
 *      ds (fused.channel_mix).  csrc/gemm_nt4.hip only: K % 1024 == 0. */
int rwkv7_gemm_nt_relusq_bwd_s_bf16(int M, int N, int K, const void *A, const void *W, const void *s, void *C, rwkv7_stream_t stream);
/*      a projection with the residual add that follows it as the epilogue: C[M][N] = bf16(bf16(A . W^T) + resid[M][N]) -- the block wiring
 *      x = x + att(...) of rwkv_s2s_single_ffn.py:262-276 for the output projection (fused.linear_add); what nn.Linear followed by the
 *      add of the two bf16 tensors produces, bit for bit.  csrc/gemm_nt4.hip only: K % 1024 == 0.  C must not alias resid. */
int rwkv7_gemm_nt_add_bf16(int M, int N, int K, const void *A, const void *W, const void *resid, void *C, rwkv7_stream_t stream);
/*      All four entries run csrc/gemm_nt4.hip (four waves, quadrant phases, ring of eight half-tile LDS slots): M, N multiples of 256,
 *      K a multiple of 1024, RWKV7_ESHAPE otherwise.  The first-generation kernel (csrc/lab/gemm_relusq.hip, K % 64 == 0) lives in the lab
 *      build under its own entry points (include/rwkv7_hip_lab.h). */

/* ---- the low-rank branches of the time-mix block taken THROUGH the token-shift lerp (fused.mix_lora; rwkv_s2s_single_ffn.py:160-190):
 *      (xm (1 - mu) + shift(xm) mu) W1^T = xm (W1 * (1 - mu))^T + shift(xm) (W1 * mu)^T, so one GEMM G = x [W_a ; W_b]^T on the LayerNorm
 *      output ([M, D] x [D, 2 R]) replaces the mixed inputs x_w, x_a, x_v, x_g and their four Linear(D, r_i).  nb <= 4 branches with ranks
 *      r_i (multiples of 8, R = sum r_i), HOST arrays of device pointers, bf16.
 *   wcat_fwd:     wcat [2 R][D]: rows off_i .. = W1_i * (1 - mu_i), rows R + off_i .. = W1_i * mu_i           (W1_i [r_i][D], mu_i [D])
 *   wcat_bwd:     dW1_i = dW_a (1 - mu_i) + dW_b mu_i ;  dmu_i[c] = sum_rows W1_i (dW_b - dW_a)                from dwcat [2 R][D]
 *   combine_fwd:  out_i[t] = act_i(bf16(m_t G[t][off_i ..] + m_{t-1} G[t - 1][R + off_i ..])), nothing from t - 1 at the first step of a
 *                 sequence (rows are [B][T]); acts: 0 none, 1 tanh, 2 sigmoid; mask [M] or NULL; out_i [M][r_i]
 *   combine_bwd:  dG [M][2 R] from the activation OUTPUTS y_i and their gradients dy_i ---- */
int rwkv7_mix_lora_wcat_fwd_bf16(int nb, const int *ranks, const void *const *w1, const void *const *mu, int D, void *wcat,
                                 rwkv7_stream_t stream);
int rwkv7_mix_lora_wcat_bwd_bf16(int nb, const int *ranks, const void *const *w1, const void *const *mu, int D, const void *dwcat,
                                 void *const *dw1, void *const *dmu, rwkv7_stream_t stream);
int rwkv7_mix_lora_combine_fwd_bf16(int nb, const int *ranks, const int *acts, long M, int T, const void *G, const void *mask,
                                    void *const *out, rwkv7_stream_t stream);
int rwkv7_mix_lora_combine_bwd_bf16(int nb, const int *ranks, const int *acts, long M, int T, const void *mask, const void *const *y,
                                    const void *const *dy, void *dG, rwkv7_stream_t stream);

/* ---- the same low-rank branches' down projections with the lerp as the GEMM's A prologue (csrc/lora_down.hip; rwkv_s2s_single_ffn.py:160-190):
 *      out_i[t] = act_i(bf16( bf16(xm[t] + (xm[t - 1] - xm[t]) mu_i) W1_i^T )), xm = x * mask, nothing from t - 1 at the first step of a
 *      sequence (rows are [B][T]) -- the reference's own rounding points; one kernel streams x [M][D] once and writes every branch's
 *      [M][r_i].  nb <= 4 branches, ranks multiples of 32 with sum <= 512, D a multiple of 128 (<= 4096), M a multiple of 128 and of T.
 *   pack:  W1_i [r_i][D] -> packed [sum r_i][D] in MFMA fragment order (same bytes, one coalesced 1 KB wave load per fragment); once
 *          per weight update
 *   fwd:   acts: 0 none, 1 tanh, 2 sigmoid; mask [M] or NULL; mu / out: HOST arrays of device pointers (mu_i [D], out_i [M][r_i]).
 *      The gradient is the one of the through-the-lerp form above (combine_bwd, wcat_*). ---- */
int rwkv7_lora_down_pack_bf16(int nb, const int *ranks, const void *const *w1, int D, void *packed, rwkv7_stream_t stream);
int rwkv7_lora_down_fwd_bf16(int nb, const int *ranks, const int *acts, long M, int T, int D, const void *x, const void *mask,
                             const void *const *mu, const void *packed, void *const *out, rwkv7_stream_t stream);

/* ---- decode-step linear layers: y[M,N] = x[M,K] @ w[N,K]^T (+ bias[N]), bf16, M <= 32 rows (one token per sequence and
 *      step), K % 64 == 0.  Replaces the nn.Linear calls of the per-token path (rwkv_s2s_single_ffn.py:482-506,545-549)
 *      for the decode batch; weight-streaming on MFMA, see csrc/gemv32.hip.  bias may be NULL. ---- */
int rwkv7_gemv32_bf16(int M, int N, int K, const void *x, const void *w, const void *bias, void *y, rwkv7_stream_t stream);

/* ---- one whole decode step (T = 1, B <= 32 sequences, bf16 weights) in ONE persistent kernel: replaces the per-token loop
 *      body RWKV_x070.forward_one (model/llm/rwkv_s2s_single_ffn.py:417-445; TMix_one :482-506, CMix_one :545-549) /
 *      forward_batch at T = 1 (model/llm/rwkv_asr_cuda_whisper.py:438-472) including the final norm and the head
 *      projection.  See csrc/decode_step.hip.
 *
 *      layer_tbl: DEVICE array [L][RWKV7_DEC_COUNT] of device pointers in the order below (bf16 unless noted; entries a
 *      layer does not have -- LN0 outside layer 0, the v branch in layer 0 -- may be NULL).  Weights are torch/rwkvfla
 *      layouts: Linear [out,in], low-rank lora.0.weight [R,D] ("1"), lora.2.weight [D,R] ("2"), lora.2.bias [D] ("0").
 *      State tensors are updated in place: att_x_prev/ffn_x_prev bf16 [B,D], att_kv fp32 [B,H,64,64] (row = value index).
 *      x_in bf16 [B,D] = embeddings of the current tokens; logits fp32 [B,V] (head_b may be NULL).
 *      workspace: rwkv7_decode_workspace_bytes() bytes, 256-byte aligned, owned by the caller, reused step after step;
 *      its first 8 bytes are the barrier words {arrivals, timeout flag}: flag != 0 after a step means a grid barrier was not
 *      met within ~0.1 s (the kernel then exits instead of hanging; the step's results are invalid).
 *      persistent = 0: ordinary launches, one kernel per phase (a stream-ordered kernel boundary costs ~1.5 us).  persistent = 1
 *      (ONE launch, the same phase bodies separated by device-scope barriers; measured 7.4 us per barrier on MI355X -- 3.9 us of
 *      arrivals/polling on one counter plus 1.8 us L2 write-back and 1.6 us invalidate, the XCD L2s not being coherent -- 2.4 ms per
 *      step) exists in the lab build only (python -m rwkvtts_amd.build --lab): the shipped library answers RWKV7_ESHAPE.
 *      Errors: RWKV7_ESHAPE unless B in [1,32], D = 64 H <= 4096, F % 64 == 0, every rank a multiple of 32 and <= 256, ranks sum <= 512. */
enum {
    RWKV7_DEC_LN0_W, RWKV7_DEC_LN0_B, RWKV7_DEC_LN1_W, RWKV7_DEC_LN1_B, RWKV7_DEC_LN2_W, RWKV7_DEC_LN2_B,
    RWKV7_DEC_XR, RWKV7_DEC_XW, RWKV7_DEC_XK, RWKV7_DEC_XV, RWKV7_DEC_XA, RWKV7_DEC_XG,
    RWKV7_DEC_WR, RWKV7_DEC_WK, RWKV7_DEC_WV, RWKV7_DEC_WO,
    RWKV7_DEC_W1, RWKV7_DEC_W2, RWKV7_DEC_W0, RWKV7_DEC_A1, RWKV7_DEC_A2, RWKV7_DEC_A0,
    RWKV7_DEC_V1, RWKV7_DEC_V2, RWKV7_DEC_V0, RWKV7_DEC_G1, RWKV7_DEC_G2,
    RWKV7_DEC_KK, RWKV7_DEC_KA, RWKV7_DEC_RK, RWKV7_DEC_GNW, RWKV7_DEC_GNB,
    RWKV7_DEC_FXK, RWKV7_DEC_WKEY, RWKV7_DEC_WVAL,
    RWKV7_DEC_ATT_XPREV, RWKV7_DEC_ATT_KV /* fp32 */, RWKV7_DEC_FFN_XPREV,
    RWKV7_DEC_COUNT
};
typedef struct rwkv7_decode_dims {
    int B, D, H, L, F, V;      /* sequences, hidden, heads, layers, channel-mix width, head rows */
    int Rw, Ra, Rv, Rg;        /* low-rank sizes of the decay / a / value-residual / gate branches */
    float ln_eps, gn_eps;      /* LayerNorm eps; GroupNorm eps (64e-5, rwkv_s2s_single_ffn.py:504) */
} rwkv7_decode_dims;
int rwkv7_decode_layer_ptrs(void);   /* == RWKV7_DEC_COUNT of the library that was loaded */
size_t rwkv7_decode_workspace_bytes(const rwkv7_decode_dims *dims);   /* 0: unsupported shape */
int rwkv7_decode_step_bf16(const rwkv7_decode_dims *dims, const void *const *layer_tbl, const void *x_in, const void *norm_w,
                           const void *norm_b, const void *head_w, const void *head_b, float *logits, void *workspace,
                           int persistent, rwkv7_stream_t stream);
/* the same step for a caller that also holds the table in HOST memory (layer_tbl_host: the same [L][RWKV7_DEC_COUNT] addresses,
 * read during the call only): with persistent = 0 every phase kernel then receives its 1-13 pointers as kernel arguments instead of
 * fetching its table row first -- one dependent load less at the head of each of the 7 L + 2 launches (0.3-0.4 us each, measured).
 * Results are bit-identical to rwkv7_decode_step_bf16. */
int rwkv7_decode_step_tbl_bf16(const rwkv7_decode_dims *dims, const void *const *layer_tbl, const void *const *layer_tbl_host,
                               const void *x_in, const void *norm_w, const void *norm_b, const void *head_w, const void *head_b,
                               float *logits, void *workspace, int persistent, rwkv7_stream_t stream);

/* ---- token draws of the generation loops, one launch each (csrc/sampling.hip).
 *
 * rwkv7_sample_rows_f32: for every row r < rows and segment s < nseg one id from logits[r * ld + seg_off[s] .. + seg_len[s])
 *      (fp32), restricted to the ids [allow_lo[s], allow_hi[s]) of the segment (NULL: the whole segment) and never one of
 *      `suppress` (segment-relative ids, nsuppress <= 256).  do_sample = 0: argmax (first maximum).  do_sample = 1: the warper
 *      chain the reference's generate() runs (HF temperature -> top-k -> top-p -> softmax -> multinomial; utils/utilities.py:101-117,
 *      model/llm/xy_llm.py:88-101 for the eight channels of a frame): top_k in [1, 64] with any top_p in (0, 1], or top_k = 0 with
 *      top_p = 1 (plain multinomial); temperature > 0.  seg_off / seg_len / allow_* / suppress are DEVICE int arrays; max_domain =
 *      the largest allow_hi - allow_lo (or seg_len), <= 15360.  out: [rows][nseg] int64 ids (segment-relative).
 *      Randomness: Philox4x32-10 keyed by `seed`, counter (*step, workgroup): `step` is a DEVICE int64 the caller advances between
 *      calls (the position counter of the decode loop), so a captured launch draws fresh ids on every replay and (seed, *step)
 *      fixes them.  RWKV7_ESHAPE for parameter combinations outside the above (the Python host then runs the torch chain).  At most
 *      128 candidates survive top-k: a row with more than 128 ids tied at the k-th value keeps the 128 smallest of them.
 * rwkv7_ras_step_f32: one token of CosyVoice's streaming loop (B = 1): ras_sampling (third_party/cosyvoice/utils/common.py:109-137:
 *      nucleus top_p / top_k, and random_sampling when the candidate already occurs >= win_size * tau_r times in `recent`) with the
 *      EOS rejection of sampling_ids (model/llm/llm.py:160-176) while *step_i < n_ignore, followed by the loop's bookkeeping:
 *      *tok = id; if id != eos: recent[*ptr] = id, *ptr = (*ptr + 1) % win_size; *step_i += 1.  logits fp32 [V] (any shift of the
 *      log-probabilities), V <= 15360, top_k <= 128, win_size <= 128; tok / recent [win_size] / ptr / step_i DEVICE int64. */
int rwkv7_sample_rows_f32(int rows, int nseg, const float *logits, long ld, const int *seg_off, const int *seg_len, const int *allow_lo,
                          const int *allow_hi, const int *suppress, int nsuppress, int max_domain, int do_sample, int top_k, float top_p,
                          float temperature, unsigned long long seed, const long *step, long *out, rwkv7_stream_t stream);
/* the one-segment draw followed by what the decode loop does with the id (decode.GraphDecoder._step; the reference's generate():
 * finished sequences emit `pad`, `eos` ends a sequence, the id is stored in the output row at column *step and is the next input):
 * unfinished [rows] bytes or NULL (no EOS handling), ids [rows] (required), seq [rows][seq_ld] or NULL, emb bf16 [V][D] + x bf16
 * [rows][D] or NULL: the embedding row of the id is copied to x, where the next decode step reads its input.  tail may be NULL
 * (the draw alone).  min_eos_id >= 0: that id cannot be drawn while *step < min_eos_until (min_new_tokens of the reference's
 * generate(): utils/utilities.py:106). */
typedef struct rwkv7_sample_tail {
    unsigned char *unfinished;
    long eos, pad;
    long *ids;
    long *seq;
    long seq_ld;
    const void *emb;
    void *x;
    int D;
} rwkv7_sample_tail;
int rwkv7_sample_rows_tail_f32(int rows, const float *logits, long ld, const int *seg_off, const int *seg_len, const int *allow_lo,
                               const int *allow_hi, const int *suppress, int nsuppress, int max_domain, int do_sample, int top_k, float top_p,
                               float temperature, unsigned long long seed, const long *step, long *out, const rwkv7_sample_tail *tail,
                               int min_eos_id, long min_eos_until, rwkv7_stream_t stream);
int rwkv7_ras_step_f32(int V, const float *logits, long *tok, long *recent, long *ptr, long *step_i, long n_ignore, int eos, float top_p,
                       int top_k, int win_size, float tau_r, unsigned long long seed, rwkv7_stream_t stream);

/* ---- one frame of the XY generation loop (model/llm/xy_llm.py:39-146, CustomGenerationMixin._sample) after the eight draws:
 * rwkv7_xy_frame_step: flush countdown, EOS / pad substitution, append of the row, stopping criteria and the `all finished` flag for
 *      B <= 64 sequences of C channels, on DEVICE int64 state (out [B][rows][C], row [B][C] = the row just written, pos / n_rows
 *      scalars, unfinished / needs [B], all_done one byte), from the drawn ids nt [B][C] (channel 0 in vocabulary ids).  eos0 < 0:
 *      no EOS id; total < 0: no length bound; eos_list: n_eos DEVICE ids for the stopping criterion; reference_termination = 1
 *      reproduces lines :139-140 literally.  The torch form of the same rules is rwkvtts_amd/xy_llm.py _XYFrameState.step.
 * rwkv7_xy_embed_bf16: x[b] = sum_c table_c[row[b][c]] (bf16 rows of D, added in channel order with bf16 rounding after every
 *      addition, like the module's chain of tensor additions; xy_llm.py:189-200); tables_host: HOST array of C <= 16 DEVICE pointers. */
int rwkv7_xy_frame_step(int B, int C, int rows, long text_shift, long speech_vocab, long pad, long eos0, long total, const long *eos_list,
                        int n_eos, int reference_termination, const long *nt, long *out, long *row, long *pos, long *unfinished, long *needs,
                        unsigned char *all_done, long *n_rows, rwkv7_stream_t stream);
int rwkv7_xy_embed_bf16(int B, int C, int D, const void *const *tables_host, const long *row, void *x, rwkv7_stream_t stream);

/* the low-rank pair of the decode step in one launch: y[M,N] = act(x[M,K] @ w1[R,K]^T) @ w2[N,R]^T (+ bias); M <= 32,
 * K % 64 == 0, R in {32,64,128}, act 0 none / 1 tanh / 2 sigmoid (rwkv_s2s_single_ffn.py:497-500: w, a, v, g branches) */
int rwkv7_lora32_bf16(int M, int N, int K, int R, int act, const void *x, const void *w1, const void *w2, const void *bias,
                      void *y, rwkv7_stream_t stream);

/* probe of ds_read_b64_tr_b16 (LDS transpose read): in = 4096 u16 copied to LDS, addr[64] = element index each lane
 * points at, out[64][4] = what each lane receives */
int rwkv7_debug_tr16(const void *in, const int *addr, void *out, rwkv7_stream_t stream);
/* unit-test hook for the MFMA fragment layouts: D[32][32] = X[32][64] Y[32][64]^T (fp32 in, bf16-split MFMA),
 * DT = the same tile after the transposed LDS write-back (hi+lo planes re-joined). */
int rwkv7_debug_mma32(const float *X, const float *Y, float *D, float *DT, rwkv7_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* RWKV7_HIP_H */
