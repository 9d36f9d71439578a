// tools/probes/copy_probe.hip -- a typical elementwise pass (read NR tensors, write NW) with plain and non-temporal accesses:
// does `nt` help a kernel whose inputs are read once and whose outputs are read by a later kernel?
//   hipcc --offload-arch=gfx950 -O3 -o copy_probe tools/probes/copy_probe.hip && ./copy_probe
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ __forceinline__ uint4 ld(const uint4 *p, bool nt) {
    if (!nt) return *p;
    uint4 v;
    v.x = __builtin_nontemporal_load(&p->x); v.y = __builtin_nontemporal_load(&p->y);
    v.z = __builtin_nontemporal_load(&p->z); v.w = __builtin_nontemporal_load(&p->w);
    return v;
}
__device__ __forceinline__ void st(uint4 *p, uint4 v, bool nt) {
    if (!nt) { *p = v; return; }
    __builtin_nontemporal_store(v.x, &p->x); __builtin_nontemporal_store(v.y, &p->y);
    __builtin_nontemporal_store(v.z, &p->z); __builtin_nontemporal_store(v.w, &p->w);
}

template <int NR, int NW, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void ew_kernel(const uint4 *__restrict__ src, uint4 *__restrict__ dst, long n16) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (long)gridDim.x * blockDim.x) {
        uint4 acc = make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < NR; r++) {
            const uint4 v = ld(src + (long)r * n16 + i, NTL);
            acc.x += v.x; acc.y ^= v.y; acc.z += v.z; acc.w ^= v.w;
        }
#pragma unroll
        for (int w = 0; w < NW; w++) {
            acc.x += w;
            st(dst + (long)w * n16 + i, acc, NTS);
        }
    }
}

template <int NR, int NW, bool NTL, bool NTS>
float run(const uint4 *src, uint4 *dst, long n16, int grid) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int i = 0; i < 2; i++) ew_kernel<NR, NW, NTL, NTS><<<grid, 256>>>(src, dst, n16);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int i = 0; i < 5; i++) ew_kernel<NR, NW, NTL, NTS><<<grid, 256>>>(src, dst, n16);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / 5;
}

template <int NR, int NW>
void sweep(const uint4 *src, uint4 *dst, long n16, int grid) {
    const double gb = (double)(NR + NW) * n16 * 16 / 1e9;
    const float a = run<NR, NW, false, false>(src, dst, n16, grid), b = run<NR, NW, true, false>(src, dst, n16, grid),
                c = run<NR, NW, false, true>(src, dst, n16, grid), d = run<NR, NW, true, true>(src, dst, n16, grid);
    printf("read %d write %d, grid %5d: plain %6.1f us (%4.2f TB/s) | nt loads %6.1f us (%4.2f) | nt stores %6.1f us (%4.2f) | both %6.1f us (%4.2f)\n",
           NR, NW, grid, a * 1e3, gb / a, b * 1e3, gb / b, c * 1e3, gb / c, d * 1e3, gb / d);
}

int main() {
    const long n16 = 32768L * 1024 * 2 / 16;   // one [32768, 1024] bf16 tensor = 64 MiB
    uint4 *src, *dst;
    (void)hipMalloc(&src, 8 * n16 * 16);
    (void)hipMalloc(&dst, 8 * n16 * 16);
    (void)hipMemset(src, 1, 8 * n16 * 16);
    for (int grid : {2048, 8192, 32768}) {
        sweep<1, 1>(src, dst, n16, grid);
        sweep<2, 1>(src, dst, n16, grid);
        sweep<2, 2>(src, dst, n16, grid);
        sweep<1, 6>(src, dst, n16, grid);
        sweep<7, 1>(src, dst, n16, grid);
        sweep<6, 5>(src, dst, n16, grid);
    }
    return 0;
}
