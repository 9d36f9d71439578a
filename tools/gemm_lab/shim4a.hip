#include <hip/hip_runtime.h>
namespace rwkv7 { int gemm_nt4_bf16(int M, int N, int K, const void *A, const void *W, void *C, const void *aux, int epilogue, hipStream_t st); }
extern "C" int lab_gemm(int M, int N, int K, const void *A, const void *W, void *C, int epi, void *st) {
    return rwkv7::gemm_nt4_bf16(M, N, K, A, W, C, nullptr, epi, (hipStream_t)st);
}
extern "C" int lab_gemm_aux(int M, int N, int K, const void *A, const void *W, void *C, const void *aux, int epi, void *st) {
    return rwkv7::gemm_nt4_bf16(M, N, K, A, W, C, aux, epi, (hipStream_t)st);
}
