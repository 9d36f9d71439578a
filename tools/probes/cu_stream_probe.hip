// tools/probes/cu_stream_probe.hip -- how many bytes per cycle can ONE workgroup per CU pull from HBM, as a function of the
// waves it has and of the 16-byte loads each lane keeps in flight?  (The chunked WKV7 kernels run one 8-wave workgroup per CU
// and stall at the ISSUE of their prefetch loads)
//   hipcc --offload-arch=gfx950 -O3 -o cu_stream_probe tools/probes/cu_stream_probe.hip && ./cu_stream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int INFLIGHT, bool NT>
__global__ __launch_bounds__(512) void stream_kernel(const uint4 *__restrict__ src, uint4 *__restrict__ sink, long per_wg16, int iters) {
    extern __shared__ char lds[];   // requested size keeps it to one workgroup per CU
    const long base = (long)blockIdx.x * per_wg16;
    const int nthr = blockDim.x;
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (int it = 0; it < iters; it++) {
        for (long i = threadIdx.x; i + (long)(INFLIGHT - 1) * nthr < per_wg16; i += (long)INFLIGHT * nthr) {
            uint4 v[INFLIGHT];
#pragma unroll
            for (int u = 0; u < INFLIGHT; u++) {
                const uint4 *p = src + base + i + (long)u * nthr;
                if (NT) {
                    v[u].x = __builtin_nontemporal_load(&p->x); v[u].y = __builtin_nontemporal_load(&p->y);
                    v[u].z = __builtin_nontemporal_load(&p->z); v[u].w = __builtin_nontemporal_load(&p->w);
                } else {
                    v[u] = *p;
                }
            }
#pragma unroll
            for (int u = 0; u < INFLIGHT; u++) {
                acc.x ^= v[u].x; acc.y ^= v[u].y; acc.z ^= v[u].z; acc.w ^= v[u].w;
            }
        }
    }
    if (acc.x == 0x12345678u && acc.y == 0x9abcdef0u) sink[blockIdx.x * nthr + threadIdx.x] = acc;   // never true: keeps the loads
    (void)lds;
}

template <int INFLIGHT, bool NT>
float run(int waves, const uint4 *src, uint4 *sink, long per_wg16, int nwg) {
    hipFuncSetAttribute(reinterpret_cast<const void *>(&stream_kernel<INFLIGHT, NT>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    stream_kernel<INFLIGHT, NT><<<nwg, waves * 64, 100 * 1024>>>(src, sink, per_wg16, 1);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    stream_kernel<INFLIGHT, NT><<<nwg, waves * 64, 100 * 1024>>>(src, sink, per_wg16, 1);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    const int nwg = 256;
    const long per_wg_bytes = 16L << 20;                 // 16 MiB per workgroup, 4 GiB in all: nothing stays in a cache
    const long per_wg16 = per_wg_bytes / 16;
    uint4 *src, *sink;
    hipMalloc(&src, per_wg_bytes * nwg);
    hipMalloc(&sink, 512L * nwg * 16);
    hipMemset(src, 1, per_wg_bytes * nwg);
    printf("one workgroup per CU (100 KB of LDS requested), 16 MiB per workgroup, plain / non-temporal 16-byte loads\n");
    printf("%6s %9s %12s %12s %14s\n", "waves", "in flight", "GB/s plain", "GB/s nt", "B/cycle/CU@2.4");
    for (int waves : {1, 2, 4, 8}) {
        float t[5][2];
        t[0][0] = run<1, false>(waves, src, sink, per_wg16, nwg); t[0][1] = run<1, true>(waves, src, sink, per_wg16, nwg);
        t[1][0] = run<2, false>(waves, src, sink, per_wg16, nwg); t[1][1] = run<2, true>(waves, src, sink, per_wg16, nwg);
        t[2][0] = run<4, false>(waves, src, sink, per_wg16, nwg); t[2][1] = run<4, true>(waves, src, sink, per_wg16, nwg);
        t[3][0] = run<8, false>(waves, src, sink, per_wg16, nwg); t[3][1] = run<8, true>(waves, src, sink, per_wg16, nwg);
        t[4][0] = run<16, false>(waves, src, sink, per_wg16, nwg); t[4][1] = run<16, true>(waves, src, sink, per_wg16, nwg);
        const int infl[5] = {1, 2, 4, 8, 16};
        for (int k = 0; k < 5; k++) {
            const double gb = (double)per_wg_bytes * nwg / 1e9;
            printf("%6d %9d %12.0f %12.0f %14.2f\n", waves, infl[k], gb / (t[k][0] * 1e-3), gb / (t[k][1] * 1e-3),
                   gb * 1e9 / (t[k][0] * 1e-3) / 256 / 2.4e9);
        }
    }
    return 0;
}
