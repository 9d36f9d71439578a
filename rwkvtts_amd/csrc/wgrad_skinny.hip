// rwkvtts_amd/csrc/wgrad_skinny.hip -- weight gradients of the low-rank (LoRA) projections, bf16, gfx950.
//
// Reference arithmetic: autograd of `tanh(xw @ w1) @ w2`, `xa @ a1 @ a2`, `xv @ v1 @ v2`, `sigmoid(xg @ g1) @ g2`
// (model/llm/rwkv_s2s_single_ffn.py:172-184; rwkvfla LoRA modules *.lora.0 / *.lora.2): for a Linear y = x W^T the weight
// gradient is dW[N][K] = sum_t dy[t][N] x[t][K] with t over the B*T rows.  Eight of them per layer have one side of 32..128
// (the rank) and the other of D: 4 GFLOP each, but 67 MB of activations to stream -- HBM-bound (11 us at 6.3 TB/s).  The BLAS
// route (8 row slabs as a batched GEMM with fp32 outputs + a reduction) takes 26-32 us + 7 us each, 6.8 ms per training step.
//
// Here: grid = (wide side / 256) x S row slabs; a workgroup streams its [rows][256] piece of the wide operand and the whole
// narrow operand through LDS in 32-row steps (registers prefetch the next step) and keeps its [rank][256] (or [256][rank]) fp32
// tile in MFMA accumulators; both operands are stored as they come (row = token, i.e. k-major) and fetched with the LDS
// transpose read (frag_tr), so nothing is transposed in memory.  Row strides of the planes are = 16 (mod 64) dwords: the four
// token rows a transpose read touches then fall into disjoint bank windows.  Partials [S][N][K] fp32 are summed (and rounded to
// bf16, straight into the gradient buffer) by rwkv7_sum_slabs_bf16.
#include "chunk_common.h"

namespace rwkv7 {

namespace {
constexpr int kWide = 256;           // columns of the wide operand per workgroup
constexpr int kStep = 32;            // rows (tokens) per step
constexpr int kLdWide = kWide + 32;  // 288 elements = 144 dwords = 16 (mod 64)
constexpr int kPF = 4;               // steps in flight (the loop body names four register sets)
__host__ __device__ constexpr int ld_narrow(int rs) { return rs == 4 ? 160 : 96; }   // 80 / 48 dwords = 16 / 48 (mod 64)

// RS = rank / 32.  WIDE_ROWS: the wide side indexes the ROWS of the output (dW[D][rank], the up projection), else its columns.
template <int RS, bool WIDE_ROWS>
// waves_per_eu(1, 2): without it hipcc budgets registers for the 6 workgroups per CU that the 24 KB of LDS would allow (80 VGPRs)
// and spills the prefetch ring to scratch memory
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2))) void wgrad_skinny_kernel(int N, int K, int rows_per_slab, const bf16_t *__restrict__ dy_,
                                                           const bf16_t *__restrict__ x_, float *__restrict__ part) {
    constexpr int R = 32 * RS, LDN = ld_narrow(RS);
    __shared__ __attribute__((aligned(16))) uint16_t s_wide[kStep * kLdWide];
    __shared__ __attribute__((aligned(16))) uint16_t s_narrow[kStep * LDN];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wide_dim = WIDE_ROWS ? N : K;
    const int nblk = wide_dim / kWide;
    const int blk = blockIdx.x % nblk, slab = blockIdx.x / nblk;
    const long row0 = (long)slab * rows_per_slab;
    // wide operand: dy (WIDE_ROWS) or x; narrow operand: the other one
    const uint16_t *wide = reinterpret_cast<const uint16_t *>(WIDE_ROWS ? dy_ : x_) + row0 * wide_dim + blk * kWide;
    const uint16_t *narrow = reinterpret_cast<const uint16_t *>(WIDE_ROWS ? x_ : dy_) + row0 * R;
    // per-thread pieces of one step: 4 x 16 bytes of the wide slab, up to 2 x 16 bytes of the narrow one
    constexpr int NCH = kStep * R / 8;                 // 16-byte chunks of the narrow slab: 128 RS
    constexpr int NPT = (NCH + 255) / 256;             // per thread: 1 (RS <= 2) or 2
    // kPF steps in flight (registers): one workgroup per CU and ~1.5 us of HBM latency per dependent step would otherwise
    // bound the kernel (first cut, one step ahead: 35-46 us, no faster than the BLAS route)
    struct Regs {   // scalar fields: arrays inside the struct ended up in scratch memory (416 bytes per lane, 87-100 us)
        uint4 w0, w1, w2, w3, n0, n1;
    };
    const int nsteps = rows_per_slab / kStep;
    const int wrow = tid >> 5, wcol = (tid & 31) * 8;                 // wide slab: chunk tid + 256 i = row (tid >> 5) + 8 i
    const int nidx0 = tid % NCH, nidx1 = (tid + 256) % NCH;           // narrow slab chunks (wrap for RS == 1: threads 128.. repeat 0..127)
    auto fetch = [&](int step) {
        Regs r;
        step = step < nsteps ? step : nsteps - 1;   // unconditional (clamped): see wkv7_chunk_bwd.hip on conditional loads
        const uint16_t *w = wide + ((long)step * kStep + wrow) * wide_dim + wcol, *n = narrow + (long)step * kStep * R;
        r.w0 = *reinterpret_cast<const uint4 *>(w);
        r.w1 = *reinterpret_cast<const uint4 *>(w + 8L * wide_dim);
        r.w2 = *reinterpret_cast<const uint4 *>(w + 16L * wide_dim);
        r.w3 = *reinterpret_cast<const uint4 *>(w + 24L * wide_dim);
        r.n0 = *reinterpret_cast<const uint4 *>(n + nidx0 * 8);
        r.n1 = *reinterpret_cast<const uint4 *>(n + nidx1 * 8);   // RS < 4: a second copy of chunk nidx0 / an unused chunk
        return r;
    };
    auto stage = [&](const Regs r) {
        uint16_t *sw = s_wide + wrow * kLdWide + wcol;
        *reinterpret_cast<uint4 *>(sw) = r.w0;
        *reinterpret_cast<uint4 *>(sw + 8 * kLdWide) = r.w1;
        *reinterpret_cast<uint4 *>(sw + 16 * kLdWide) = r.w2;
        *reinterpret_cast<uint4 *>(sw + 24 * kLdWide) = r.w3;
        *reinterpret_cast<uint4 *>(s_narrow + (nidx0 / (R / 8)) * LDN + (nidx0 % (R / 8)) * 8) = r.n0;
        if (NPT == 2) *reinterpret_cast<uint4 *>(s_narrow + (nidx1 / (R / 8)) * LDN + (nidx1 % (R / 8)) * 8) = r.n1;
    };
    f32x16 acc[RS][2];
#pragma unroll
    for (int i = 0; i < RS; i++) acc[i][0] = acc[i][1] = zero16();
    auto products = [&]() {
#pragma unroll
        for (int i = 0; i < RS; i++)
#pragma unroll
            for (int j = 0; j < 2; j++) {
                if (WIDE_ROWS)   // D[m][n]: m = wide (rows of dW), n = narrow
                    mma_gen<kStep, true, false, true, false>(acc[i][j], s_wide, s_wide, kLdWide, wave * 64 + j * 32, s_narrow, s_narrow, LDN, i * 32, lane);
                else             // m = narrow (rows of dW), n = wide
                    mma_gen<kStep, true, false, true, false>(acc[i][j], s_narrow, s_narrow, LDN, i * 32, s_wide, s_wide, kLdWide, wave * 64 + j * 32, lane);
            }
    };
    // four named register sets, not an array of structs handed to a lambda by reference: that form lands in scratch memory
    // (496 bytes per lane, 87-100 us)
    Regs r0 = fetch(0), r1 = fetch(1), r2 = fetch(2), r3 = fetch(3);
#define WGS_STEP(R_, U_)                                                                                            \
    stage(R_);                                                                                                      \
    lds_barrier();                                                                                                  \
    __builtin_amdgcn_sched_barrier(0); /* keep the prefetch here (the scheduler would sink it to the end of the body) */ \
    R_ = fetch(step + (U_) + kPF);                                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                                              \
    products();                                                                                                     \
    lds_barrier(); /* the planes are rewritten by the next stage() */
    for (int step = 0; step < nsteps; step += kPF) {   // nsteps is a multiple of kPF (launcher)
        WGS_STEP(r0, 0)
        WGS_STEP(r1, 1)
        WGS_STEP(r2, 2)
        WGS_STEP(r3, 3)
    }
#undef WGS_STEP
    float *p = part + (long)slab * N * K;
#pragma unroll
    for (int i = 0; i < RS; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int wbase = blk * kWide + wave * 64 + j * 32, nbase = i * 32;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int m = d_row(r, lane), n = lane & 31;
                const long o = WIDE_ROWS ? (long)(wbase + m) * K + nbase + n : (long)(nbase + m) * K + wbase + n;
                p[o] = acc[i][j][r];
            }
        }
}

template <int RS>
void launch(bool wide_rows, int grid, int N, int K, int rows_per_slab, const void *dy, const void *x, float *part, hipStream_t st) {
    if (wide_rows)
        hipLaunchKernelGGL((wgrad_skinny_kernel<RS, true>), dim3(grid), dim3(256), 0, st, N, K, rows_per_slab, (const bf16_t *)dy,
                           (const bf16_t *)x, part);
    else
        hipLaunchKernelGGL((wgrad_skinny_kernel<RS, false>), dim3(grid), dim3(256), 0, st, N, K, rows_per_slab, (const bf16_t *)dy,
                           (const bf16_t *)x, part);
}
}  // namespace

// ---- the [W_a ; W_b] weight gradient of fused.mix_lora (round 6, late): dy = dG [M][N], N = 2 R = 576 (512 in layer 0), x [M][K = D] ----
// Neither side is skinny: the batched-library route (32 row slabs, fp32 partials) ran a 256 x 192 tile kernel for 75 us per layer
// (0.52 PF/s).  Here: grid = 2 halves of N x (K / 256) column blocks x S slabs; a workgroup of eight waves keeps its [N / 2][256] fp32
// tile in MFMA accumulators: wave = 64 columns x five (or four) of the half's nine 32-row tiles (160 registers).  Both operands come as
// they are stored (row = token, i.e. k-major) through LDS in 32-token steps, fetched with the transpose read; register prefetch kMidPF
// steps ahead.  The eight workgroups of a slab share an XCD (dG rows are read by four of them, x rows by two: L2).
namespace {
constexpr int kMidPF = 2;
constexpr int kMidLdN = 288;   // narrow plane [32][<= 288]: 144 dwords = 16 (mod 64)
template <int NTN>             // 32-row tiles of dy per half (8 or 9)
__global__ __launch_bounds__(512) void wgrad_mid_kernel(int N, int K, int rows_per_slab, const bf16_t *__restrict__ dy_, const bf16_t *__restrict__ x_,
                                                        float *__restrict__ part) {
    constexpr int NH = 32 * NTN;                 // dy columns per half
    constexpr int NTA = (NTN + 1) / 2;           // tiles of the waves 0-3; waves 4-7 take the rest
    __shared__ __attribute__((aligned(16))) uint16_t s_wide[kStep * kLdWide];
    __shared__ __attribute__((aligned(16))) uint16_t s_narrow[kStep * kMidLdN];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), cg = wave & 3, rh = wave >> 2;
    const int ncb = K / kWide;
    // block id -> (slab, half, column block): ids b, b + 8, ... share an XCD; the 2 ncb workgroups of a slab are consecutive there
    int slab, inner;
    {
        const int per = 2 * ncb;
        if ((gridDim.x & 7) == 0 && ((gridDim.x / per) & 7) == 0) {
            const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
            slab = (j / per) * 8 + xcd;
            inner = j % per;
        } else {
            slab = blockIdx.x / per;
            inner = blockIdx.x % per;
        }
    }
    const int nh = inner / ncb, blk = inner % ncb;
    const long row0 = (long)slab * rows_per_slab;
    const uint16_t *wide = reinterpret_cast<const uint16_t *>(x_) + row0 * K + blk * kWide;
    const uint16_t *narrow = reinterpret_cast<const uint16_t *>(dy_) + row0 * N + nh * NH;
    // per-thread pieces of a step: wide 32 x 32 chunks of 16 bytes = 2 per thread; narrow 32 x (NH / 8) chunks = up to 3 per thread
    constexpr int NCW = NH / 8, NCH = kStep * NCW;     // 36 (32) chunks per row; 1152 (1024) per step
    const int wrow = tid >> 5, wcol = (tid & 31) * 8;
    const int q0 = tid, q1 = tid + 512, q2 = tid + 1024 < NCH ? tid + 1024 : NCH - 1;   // (the third clamped: unconditional load, guarded store)
    const int nr0 = q0 / NCW, nc0 = (q0 - nr0 * NCW) * 8, nr1 = q1 / NCW, nc1 = (q1 - nr1 * NCW) * 8, nr2 = q2 / NCW, nc2 = (q2 - nr2 * NCW) * 8;
    const int nsteps = rows_per_slab / kStep;
    struct Regs { uint4 w0, w1, n0, n1, n2; };   // scalar fields (see wgrad_skinny_kernel)
    auto fetch = [&](int step) {
        Regs r;
        step = step < nsteps ? step : nsteps - 1;
        const uint16_t *w = wide + ((long)step * kStep + wrow) * K + wcol, *n = narrow + (long)step * kStep * N;
        r.w0 = *reinterpret_cast<const uint4 *>(w);
        r.w1 = *reinterpret_cast<const uint4 *>(w + 16L * K);
        r.n0 = *reinterpret_cast<const uint4 *>(n + (long)nr0 * N + nc0);
        r.n1 = *reinterpret_cast<const uint4 *>(n + (long)nr1 * N + nc1);
        r.n2 = *reinterpret_cast<const uint4 *>(n + (long)nr2 * N + nc2);
        return r;
    };
    auto stage = [&](const Regs r) {
        uint16_t *sw = s_wide + wrow * kLdWide + wcol;
        *reinterpret_cast<uint4 *>(sw) = r.w0;
        *reinterpret_cast<uint4 *>(sw + 16 * kLdWide) = r.w1;
        *reinterpret_cast<uint4 *>(s_narrow + nr0 * kMidLdN + nc0) = r.n0;
        *reinterpret_cast<uint4 *>(s_narrow + nr1 * kMidLdN + nc1) = r.n1;
        if (tid + 1024 < NCH) *reinterpret_cast<uint4 *>(s_narrow + nr2 * kMidLdN + nc2) = r.n2;
    };
    f32x16 acc[NTA][2];
#pragma unroll
    for (int i = 0; i < NTA; i++) acc[i][0] = acc[i][1] = zero16();
    const int tbase = rh ? NTA : 0, ntl = rh ? NTN - NTA : NTA;
    // every fragment of a 16-token k step is fetched ONCE (first cut: one mma_gen call per tile pair, each fetching its own four
    // fragments -- 40 KB of LDS reads per wave and step instead of 14: 57.8 us); waves 4-7 of the nine-tile half run their fifth
    // product on a clamped tile (not stored): no branch inside the step
    auto products = [&]() {
#pragma unroll
        for (int ks = 0; ks < kStep / 16; ks++) {
            bf16x8 xf[NTA], yf[2];
#pragma unroll
            for (int j = 0; j < 2; j++) yf[j] = frag_tr(s_wide, kLdWide, 16 * ks, cg * 64 + j * 32, lane);
#pragma unroll
            for (int i = 0; i < NTA; i++) xf[i] = frag_tr(s_narrow, kMidLdN, 16 * ks, (tbase + (i < ntl ? i : ntl - 1)) * 32, lane);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < NTA; i++)
#pragma unroll
                for (int j = 0; j < 2; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xf[i], yf[j], acc[i][j], 0, 0, 0);   // m = dy column, n = x column
        }
    };
    Regs r0 = fetch(0), r1 = fetch(1);
#define WGM_STEP(R_, U_)                                                                                            \
    stage(R_);                                                                                                      \
    lds_barrier();                                                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                                              \
    R_ = fetch(step + (U_) + kMidPF);                                                                               \
    __builtin_amdgcn_sched_barrier(0);                                                                              \
    products();                                                                                                     \
    lds_barrier();
    for (int step = 0; step < nsteps; step += kMidPF) {   // nsteps is a multiple of kMidPF (launcher)
        WGM_STEP(r0, 0)
        WGM_STEP(r1, 1)
    }
#undef WGM_STEP
    float *p = part + (long)slab * N * K;
#pragma unroll
    for (int i = 0; i < NTA; i++) {
        if (i < ntl) {
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const int nbase = nh * NH + (tbase + i) * 32, wbase = blk * kWide + cg * 64 + j * 32;
#pragma unroll
                for (int r = 0; r < 16; r++) p[(long)(nbase + d_row(r, lane)) * K + wbase + (lane & 31)] = acc[i][j][r];
            }
        }
    }
}
}  // namespace

// part[s][N][K] (fp32) = dy[rows of slab s][N]^T x[rows of slab s][K]; N in {512, 576}, K a multiple of 256, rows per slab a multiple of 64
int wgrad_mid_bf16(long M, int N, int K, int S, const void *dy, const void *x, float *part, hipStream_t st) {
    (void)hipGetLastError();
    const int rows_per_slab = (int)(M / S);
    const int grid = 2 * (K / kWide) * S;
    if (N == 576)
        hipLaunchKernelGGL((wgrad_mid_kernel<9>), dim3(grid), dim3(512), 0, st, N, K, rows_per_slab, (const bf16_t *)dy, (const bf16_t *)x, part);
    else if (N == 512)
        hipLaunchKernelGGL((wgrad_mid_kernel<8>), dim3(grid), dim3(512), 0, st, N, K, rows_per_slab, (const bf16_t *)dy, (const bf16_t *)x, part);
    else
        return -4;
    return (int)hipGetLastError();
}

// part[s][N][K] (fp32) = dy[rows of slab s][N]^T x[rows of slab s][K]; one of N, K in {32, 64, 128}, the other a multiple of 256;
// M = S * rows_per_slab, rows_per_slab a multiple of 128 (4 steps of 32 rows in flight).
int wgrad_skinny_bf16(long M, int N, int K, int S, const void *dy, const void *x, float *part, hipStream_t st) {
    (void)hipGetLastError();
    const bool wide_rows = N > K;
    const int rank = wide_rows ? K : N, wide = wide_rows ? N : K;
    const int rows_per_slab = (int)(M / S);
    const int grid = wide / kWide * S;
    switch (rank) {
    case 32: launch<1>(wide_rows, grid, N, K, rows_per_slab, dy, x, part, st); break;
    case 64: launch<2>(wide_rows, grid, N, K, rows_per_slab, dy, x, part, st); break;
    case 128: launch<4>(wide_rows, grid, N, K, rows_per_slab, dy, x, part, st); break;
    default: return -4;
    }
    return (int)hipGetLastError();
}

}  // namespace rwkv7
