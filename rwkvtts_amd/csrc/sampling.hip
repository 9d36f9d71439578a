// rwkvtts_amd/csrc/sampling.hip -- the token draws of the generation loops as ONE launch each (sampling / generation
// control on the device).
//
// Reference semantics:
//   * sample_rows: HF's warper chain that the reference's generate() runs (utils/utilities.py:101-117 -> transformers
//     TemperatureLogitsWarper -> TopKLogitsWarper -> TopPLogitsWarper -> softmax -> multinomial), model/llm/xy_llm.py:88-101 for the
//     eight channels of an XY frame, or argmax.  As torch operations that chain is ~15 launches per call (topk, sort, softmax,
//     cumsum, scatter, multinomial ...): 0.45 ms per Spark step at B = 32 and 1.8 ms per XY frame (8 calls) on MI355X, against a
//     decode step of 0.98 ms.  Here: one workgroup per (row, channel), the segment's logits in registers (5 / 33 / 60 per thread),
//     the k largest by repeated block-wide maxima of (value, index) keys -- DPP inside the rows of 16, one barrier per round, the
//     owning thread drops its element and rescans its registers -- the nucleus and the draw on one thread.
//   * ras_step: CosyVoice's repetition-aware sampler (third_party/cosyvoice/utils/common.py:109-137: nucleus_sampling + the
//     win_size / tau_r fallback to random_sampling) with the EOS rejection of sampling_ids (model/llm/llm.py:160-176) and the
//     bookkeeping of the streaming loop (previous id, ring of recent ids, step index) in the same launch.
// The draws use a counter-based generator (Philox4x32-10; key = seed, counter = step index, workgroup, draw): the same (seed,
// step) gives the same id in eager runs and in hipGraph replays -- the step index lives in device memory and is advanced by the
// loop itself.  Same distributions as the torch chains (tests/test_sampling_gpu.py), not the same consumption of torch's stream.
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rwkv7 {
namespace {

constexpr int kSmpThreads = 256;
constexpr int kSmpMaxN = 15360;    // ids of one segment (after the allowed range is applied): 60 per thread, in registers
constexpr int kSmpMaxCand = 128;   // top-k candidates (k <= 64 plus ties at the k-th value)

__device__ __forceinline__ uint2 mulhilo(uint32_t a, uint32_t b) {
    const uint64_t p = (uint64_t)a * b;
    return make_uint2((uint32_t)(p >> 32), (uint32_t)p);
}
// Philox4x32-10 (Salmon et al. 2011)
__device__ __forceinline__ uint4 philox(uint4 c, uint2 k) {
#pragma unroll
    for (int r = 0; r < 10; r++) {
        const uint2 p0 = mulhilo(0xD2511F53u, c.x), p1 = mulhilo(0xCD9E8D57u, c.z);
        c = make_uint4(p1.x ^ c.y ^ k.x, p1.y, p0.x ^ c.w ^ k.y, p0.y);
        k.x += 0x9E3779B9u;
        k.y += 0xBB67AE85u;
    }
    return c;
}
__device__ __forceinline__ float u01(uint32_t x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }   // [0, 1)

struct SmpShared {
    float cand_v[kSmpMaxCand];
    int cand_i[kSmpMaxCand];
    unsigned long long key[2][4];   // per-wave maxima, two sets: one barrier per selection round
    float red[4];
    float part[kSmpThreads];
    int pick[2];
    unsigned hist[256];                       // value histogram (select_bins) / one byte of the value keys per pass (select_radix)
    unsigned long long gath[kSmpThreads];     // the candidates before they are ranked
    unsigned sel[4];                          // [0] bin, [1] still needed inside it, [2] gather counter
    float wtot[4];                            // block_scan: the waves' totals
};

// (value, index) as one sortable key: larger value first, ties to the smaller index
__device__ __forceinline__ unsigned long long mk_key(float v, int i) {
    const uint32_t b = __float_as_uint(v);
    const uint32_t o = (b & 0x80000000u) ? ~b : (b | 0x80000000u);
    return ((unsigned long long)o << 32) | (uint32_t)(0x7fffffff - i);
}
__device__ __forceinline__ float key_val(unsigned long long k) {
    const uint32_t o = (uint32_t)(k >> 32);
    return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o);
}
__device__ __forceinline__ int key_idx(unsigned long long k) { return 0x7fffffff - (int)(uint32_t)k; }
template <int CTRL>
__device__ __forceinline__ unsigned long long dpp64(unsigned long long k) {
    const uint32_t lo = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)k, CTRL, 0xF, 0xF, true);
    const uint32_t hi = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)(k >> 32), CTRL, 0xF, 0xF, true);
    return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ unsigned long long shfl64(unsigned long long k, int m) {
    const uint32_t lo = __shfl_xor((int)(uint32_t)k, m), hi = __shfl_xor((int)(uint32_t)(k >> 32), m);
    return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ unsigned long long umax(unsigned long long a, unsigned long long b) { return a > b ? a : b; }
// every lane gets the wave's maximum: four DPP steps inside the rows of 16, two cross-row exchanges
__device__ __forceinline__ unsigned long long wave_max(unsigned long long k) {
    k = umax(k, dpp64<0xB1>(k));    // quad_perm [1,0,3,2]
    k = umax(k, dpp64<0x4E>(k));    // quad_perm [2,3,0,1]
    k = umax(k, dpp64<0x141>(k));   // row_half_mirror
    k = umax(k, dpp64<0x140>(k));   // row_mirror
    k = umax(k, shfl64(k, 16));
    k = umax(k, shfl64(k, 32));
    return k;
}
// block maximum with ONE barrier (the per-wave slots alternate between two sets)
__device__ __forceinline__ unsigned long long block_max(unsigned long long k, SmpShared &sm, int set) {
    k = wave_max(k);
    if ((threadIdx.x & 63) == 0) sm.key[set][threadIdx.x >> 6] = k;
    __syncthreads();
    return umax(umax(sm.key[set][0], sm.key[set][1]), umax(sm.key[set][2], sm.key[set][3]));
}
__device__ __forceinline__ float block_sum(float v, SmpShared &sm) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sm.red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (sm.red[0] + sm.red[1]) + (sm.red[2] + sm.red[3]);
}

// A thread's share of the segment: elements tid + 256 e, e < EPT, in registers (-inf beyond the segment).
template <int EPT>
struct Vals {
    float v[EPT];
    __device__ __forceinline__ unsigned long long local_max() const {   // the value first, then the first slot that holds it
        float m = v[0];
#pragma unroll
        for (int e = 1; e < EPT; e++) m = fmaxf(m, v[e]);
        int slot = 0;
#pragma unroll
        for (int e = EPT - 1; e >= 0; e--) slot = v[e] == m ? e : slot;
        return mk_key(m, (int)threadIdx.x + kSmpThreads * slot);
    }
    __device__ __forceinline__ void drop(int idx) {   // idx belongs to this thread
        const int slot = idx / kSmpThreads;
#pragma unroll
        for (int e = 0; e < EPT; e++) v[e] = e == slot ? -INFINITY : v[e];
    }
};

// the `want` largest values in descending order (ties: smaller index first) -> cand_v / cand_i; with `ties`, further elements equal
// to the last one are taken too (TopKLogitsWarper removes `scores < kth value`: ties at the threshold stay).  Destroys x.
template <int EPT>
__device__ __forceinline__ int select_top(Vals<EPT> &x, SmpShared &sm, int want, bool ties) {
    unsigned long long lk = x.local_max();
    float last = 0.f;
    int n = 0;
    for (;;) {
        const unsigned long long k = block_max(lk, sm, n & 1);
        const float v = key_val(k);
        const int i = key_idx(k);
        if (v == -INFINITY || n == kSmpMaxCand) break;
        if (n >= want && !(ties && v == last)) break;
        if (n < want) last = v;
        if (threadIdx.x == 0) {
            sm.cand_v[n] = v;
            sm.cand_i[n] = i;
        }
        if ((i & (kSmpThreads - 1)) == (int)threadIdx.x) {   // the owner drops it and looks again
            x.drop(i);
            lk = x.local_max();
        }
        n++;
    }
    __syncthreads();
    return n;
}
// inclusive prefix sum over the 256 threads (thread order) and the block total
__device__ __forceinline__ float block_scan(float v, float &total, SmpShared &sm) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const float t = __shfl_up(v, d);
        v += lane >= d ? t : 0.f;
    }
    __syncthreads();
    if (lane == 63) sm.wtot[wave] = v;
    __syncthreads();
    const float w0 = sm.wtot[0], w1 = sm.wtot[1], w2 = sm.wtot[2], w3 = sm.wtot[3];
    total = ((w0 + w1) + w2) + w3;
    return v + (wave > 0 ? w0 : 0.f) + (wave > 1 ? w1 : 0.f) + (wave > 2 ? w2 : 0.f);
}

// The same candidate list without a round per candidate (a round of select_top is a block-wide reduction and a barrier, ~0.7-1.1 us:
// 59 us for k = 50 at 8193 ids): the k-th largest VALUE by a radix select over the order-preserving 32-bit image of the floats
// (four passes over one byte each: LDS histogram, suffix scan by one wave), then every element at or above it is gathered
// (with ties at the threshold that is what TopKLogitsWarper keeps) and ranked by counting the larger (value, index) keys.
// More than kSmpMaxCand elements at or above the threshold (a row of equal logits): the round-per-candidate form.
template <int EPT>
__device__ __forceinline__ int select_radix(Vals<EPT> &x, SmpShared &sm, int want, bool ties) {
    const int tid = threadIdx.x;
    uint32_t o[EPT];
#pragma unroll
    for (int e = 0; e < EPT; e++) {
        const uint32_t b = __float_as_uint(x.v[e]);
        o[e] = x.v[e] == -INFINITY ? 0u : ((b & 0x80000000u) ? ~b : (b | 0x80000000u));   // 0: never a candidate
    }
    uint32_t prefix = 0, need = (uint32_t)want;
#pragma unroll
    for (int pass = 0; pass < 4; pass++) {
        const int shift = 24 - 8 * pass;
        sm.hist[tid] = 0;
        __syncthreads();
#pragma unroll
        for (int e = 0; e < EPT; e++)
            if (o[e] != 0u && (pass == 0 || (o[e] >> (shift + 8)) == prefix)) atomicAdd(&sm.hist[(o[e] >> shift) & 255u], 1u);
        __syncthreads();
        if (tid < 64) {   // bins in descending order: lane l holds bins 255 - 4 l .. 252 - 4 l; the bin where the count reaches `need`
            const int b0 = 255 - 4 * tid;
            const unsigned c0 = sm.hist[b0], c1 = sm.hist[b0 - 1], c2 = sm.hist[b0 - 2], c3 = sm.hist[b0 - 3];
            unsigned incl = c0 + c1 + c2 + c3;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const unsigned t = __shfl_up(incl, d);
                incl += tid >= d ? t : 0u;
            }
            const unsigned before = incl - (c0 + c1 + c2 + c3);
            if (before < need && incl >= need) {
                unsigned cum = before;
                int bin = b0;
                unsigned left = need;
                const unsigned cs[4] = {c0, c1, c2, c3};
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    if (cum < need && cum + cs[q] >= need) {
                        bin = b0 - q;
                        left = need - cum;
                    }
                    cum += cs[q];
                }
                sm.sel[0] = (unsigned)bin;
                sm.sel[1] = left;                 // how many are still wanted inside this bin
                sm.sel[3] = need - left;          // taken from the bins above it in this pass
            }
            if (tid == 63 && incl < need) {       // fewer valid elements than wanted: everything is a candidate
                sm.sel[0] = 0xffffffffu;
                sm.sel[3] = incl;
            }
        }
        __syncthreads();
        if (sm.sel[0] == 0xffffffffu) {
            prefix = 0;
            need = 0;
            break;
        }
        prefix = (prefix << 8) | sm.sel[0];
        need = sm.sel[1];
        __syncthreads();
    }
    // prefix = the key of the want-th largest value (need == 0: take every valid element)
    const uint32_t thr = need == 0 ? 1u : prefix;
    unsigned mine = 0;
#pragma unroll
    for (int e = 0; e < EPT; e++) mine += o[e] >= thr && o[e] != 0u;
    if (tid == 0) sm.sel[2] = 0;
    __syncthreads();
    const float total_f = block_sum((float)mine, sm);   // exact: counts far below 2^24
    const int total = (int)total_f;
    if (total > kSmpMaxCand) return select_top(x, sm, want, ties);
#pragma unroll
    for (int e = 0; e < EPT; e++)
        if (o[e] >= thr && o[e] != 0u) {
            const unsigned slot = atomicAdd(&sm.sel[2], 1u);
            sm.gath[slot] = ((unsigned long long)o[e] << 32) | (uint32_t)(0x7fffffff - (tid + kSmpThreads * e));
        }
    __syncthreads();
    if (tid < total) {   // rank = number of larger keys (value first, then the smaller index)
        const unsigned long long k = sm.gath[tid];
        int rank = 0;
        for (int j = 0; j < total; j++) rank += sm.gath[j] > k;
        sm.cand_v[rank] = key_val(k);
        sm.cand_i[rank] = key_idx(k);
    }
    __syncthreads();
    return ties ? total : min(total, want);
}
// ... and the common case in one histogram pass: the values are binned linearly between the segment's minimum and maximum (a
// monotone map: the k-th largest value and everything above it lie in the top bins; the order-preserving bit image used above has
// nearly all logits in a handful of exponent bins, i.e. 64-way same-address LDS atomics), everything from the k-th value's bin up is
// gathered (<= 256 elements, else select_radix) and ranked exactly by its (value, index) key.
template <int EPT>
__device__ __forceinline__ int select_bins(Vals<EPT> &x, SmpShared &sm, int want, bool ties) {
    const int tid = threadIdx.x;
    float lmax = -INFINITY, lmin = INFINITY;
#pragma unroll
    for (int e = 0; e < EPT; e++) {
        lmax = fmaxf(lmax, x.v[e]);
        lmin = x.v[e] == -INFINITY ? lmin : fminf(lmin, x.v[e]);
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        lmax = fmaxf(lmax, __shfl_xor(lmax, o));
        lmin = fminf(lmin, __shfl_xor(lmin, o));
    }
    sm.hist[tid] = 0;
    if ((tid & 63) == 0) {
        sm.red[tid >> 6] = lmax;
        sm.wtot[tid >> 6] = lmin;
    }
    __syncthreads();
    const float mx = fmaxf(fmaxf(sm.red[0], sm.red[1]), fmaxf(sm.red[2], sm.red[3]));
    const float mn = fminf(fminf(sm.wtot[0], sm.wtot[1]), fminf(sm.wtot[2], sm.wtot[3]));
    const float scale = mx > mn ? 255.5f / (mx - mn) : 0.f;
    int bin[EPT];
#pragma unroll
    for (int e = 0; e < EPT; e++) {
        bin[e] = x.v[e] == -INFINITY ? -1 : min(255, (int)((x.v[e] - mn) * scale));
        if (bin[e] >= 0) atomicAdd(&sm.hist[bin[e]], 1u);
    }
    __syncthreads();
    if (tid < 64) {   // bins in descending order, four per lane: the bin in which the count from the top reaches `want`
        const int b0 = 255 - 4 * tid;
        const unsigned cs[4] = {sm.hist[b0], sm.hist[b0 - 1], sm.hist[b0 - 2], sm.hist[b0 - 3]};
        const unsigned own = (cs[0] + cs[1]) + (cs[2] + cs[3]);
        unsigned incl = own;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned t = __shfl_up(incl, d);
            incl += tid >= d ? t : 0u;
        }
        unsigned cum = incl - own;
        if (cum < (unsigned)want && incl >= (unsigned)want) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                if (cum < (unsigned)want && cum + cs[q] >= (unsigned)want) {
                    sm.sel[0] = (unsigned)(b0 - q);
                    sm.sel[1] = cum + cs[q];      // elements in this bin and above
                }
                cum += cs[q];
            }
        }
        if (tid == 63 && incl < (unsigned)want) {   // fewer valid elements than wanted: all of them
            sm.sel[0] = 0u;
            sm.sel[1] = incl;
        }
        if (tid == 0) sm.sel[2] = 0u;
    }
    __syncthreads();
    const int bsel = (int)sm.sel[0], total = (int)sm.sel[1];
    if (total > kSmpThreads) return select_radix(x, sm, want, ties);
#pragma unroll
    for (int e = 0; e < EPT; e++)
        if (bin[e] >= bsel) sm.gath[atomicAdd(&sm.sel[2], 1u)] = mk_key(x.v[e], tid + kSmpThreads * e);
    __syncthreads();
    if (tid < total) {   // rank = number of larger keys (value first, then the smaller index)
        const unsigned long long k = sm.gath[tid];
        int rank = 0;
        for (int j = 0; j < total; j++) rank += sm.gath[j] > k;
        if (rank < kSmpMaxCand) {
            sm.cand_v[rank] = key_val(k);
            sm.cand_i[rank] = key_idx(k);
        }
    }
    __syncthreads();
    const int cap = min(total, kSmpMaxCand);
    int n = min(want, cap);
    if (ties && n > 0) {
        const float last = sm.cand_v[n - 1];
        while (n < cap && sm.cand_v[n] == last) n++;
    }
    return n;
}
// index drawn from weights w(j) = exp(x[j] - mx) over the segment (excluding `skip`, -1 = none) for two uniforms: per-thread sums over
// the thread's elements, a prefix sum over the 256 sums, the owning thread walks its elements
template <int EPT>
__device__ __forceinline__ void draw_full(const Vals<EPT> &x, SmpShared &sm, float mx, int skip, float u0, float u1, int &i0, int &i1) {
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < EPT; e++) s += (int)threadIdx.x + kSmpThreads * e == skip ? 0.f : __expf(x.v[e] - mx);
    if (threadIdx.x < 2) sm.pick[threadIdx.x] = -1;
    float total;
    const float incl = block_scan(s, total, sm);
    const float before = incl - s;
#pragma unroll
    for (int d = 0; d < 2; d++) {
        const float target = (d ? u1 : u0) * total;
        // the last thread with weight takes targets that rounding pushes past the end
        if (s > 0.f && target >= before && (target < incl || incl >= total)) {
            float c = before;
            int last = -1;
            bool done = false;
#pragma unroll
            for (int e = 0; e < EPT; e++) {
                const int j = (int)threadIdx.x + kSmpThreads * e;
                const float w = j == skip ? 0.f : __expf(x.v[e] - mx);
                if (!done && w > 0.f) {
                    last = j;
                    c += w;
                    done = c > target;
                }
            }
            if (last >= 0) atomicMax(&sm.pick[d], last);   // (two threads can only both qualify at a rounding boundary)
        }
    }
    __syncthreads();
    i0 = sm.pick[0];
    i1 = sm.pick[1];
}

// Optional tail of a one-segment draw: what the decode loop does with the id before the next step (decode.GraphDecoder._step:
// finished sequences emit the pad id, EOS ends a sequence, the id goes into the output row at column *step and becomes the next
// input -- whose embedding row is copied here, so the next step's first kernel finds its input in place).
struct SmpTail {
    unsigned char *unfinished;   // [rows] bool, or null (no EOS handling)
    long eos, pad;
    long *ids;                   // [rows] the ids as the loop keeps them, or null: no tail at all
    long *seq;                   // [rows][seq_ld] generated ids, or null
    long seq_ld;
    const uint16_t *emb;         // [V][D] bf16, or null
    uint16_t *x;                 // [rows][D]
    int D;
};

// ---- one workgroup per (row, segment) ----------------------------------------------------------------------------------------------
template <int EPT>
__global__ __launch_bounds__(kSmpThreads) void sample_rows_kernel(int nseg, const float *__restrict__ logits, long ld,
                                                                  const int *__restrict__ seg_off, const int *__restrict__ seg_len,
                                                                  const int *__restrict__ allow_lo, const int *__restrict__ allow_hi,
                                                                  const int *__restrict__ suppress, int nsuppress, int do_sample, int top_k,
                                                                  float top_p, float inv_temp, uint2 key, const long *__restrict__ step,
                                                                  long *__restrict__ out, SmpTail tail, int min_id, long min_until) {
    __shared__ SmpShared sm;
    const int seg = blockIdx.x % nseg, row = blockIdx.x / nseg, tid = threadIdx.x;
    const float *xg = logits + (long)row * ld + seg_off[seg];
    const int n = seg_len[seg];
    const int lo = allow_lo ? max(allow_lo[seg], 0) : 0, hi = allow_hi ? min(allow_hi[seg], n) : n;
    const int m = hi - lo;
    Vals<EPT> x;
#pragma unroll
    for (int e = 0; e < EPT; e++) {
        const int j = tid + kSmpThreads * e;
        const float t = xg[lo + min(j, m - 1)];   // unconditional (clamped): a guarded load is waited for behind its issue
        x.v[e] = j < m ? t * inv_temp : -INFINITY;
    }
    for (int t = 0; t < nsuppress; t++) {   // a handful of ids
        const int sidx = suppress[t] - lo;
        if (sidx >= 0 && sidx < m && (sidx & (kSmpThreads - 1)) == tid) x.drop(sidx);
    }
    if (min_id >= 0 && *step < min_until) {   // min_new_tokens: no EOS before that many draws (HF MinNewTokensLengthLogitsProcessor)
        const int sidx = min_id - lo;
        if (sidx >= 0 && sidx < m && (sidx & (kSmpThreads - 1)) == tid) x.drop(sidx);
    }
    int choice;
    if (!do_sample) {
        choice = key_idx(block_max(x.local_max(), sm, 0));
    } else {
        const long st = *step;
        const uint4 r = philox(make_uint4((uint32_t)st, (uint32_t)((uint64_t)st >> 32), blockIdx.x, 0x5a17u), key);
        if (top_k > 0) {
            const int nc = select_bins(x, sm, min(top_k, m), true);
            // probabilities over the candidates (softmax of the filtered logits), nucleus = the ranks whose strictly-higher mass is
            // below top_p (TopPLogitsWarper: ascending cumulative <= 1 - top_p is removed, the largest always stays)
            const float ej = tid < nc ? __expf(sm.cand_v[tid] - sm.cand_v[0]) : 0.f;
            float z;
            const float cj = block_scan(ej, z, sm);                     // inclusive mass of ranks 0 .. tid
            const bool keep = tid < nc && (tid == 0 || cj - ej < top_p * z);   // the kept ranks are a prefix
            float kept;
            (void)block_scan(keep ? ej : 0.f, kept, sm);
            const float target = u01(r.x) * kept;
            if (tid == 0) sm.pick[1] = 0;
            __syncthreads();
            if (keep && cj <= target) atomicAdd(&sm.pick[1], 1);        // ranks passed before the draw's rank
            __syncthreads();
            int nk = 0;                                                 // (kept count: the last kept rank catches rounding)
            {
                float f;
                nk = (int)block_scan(keep ? 1.f : 0.f, f, sm);
                nk = (int)f;
            }
            // no candidate survived (every allowed logit is -inf / NaN, or suppression + min_eos cover the allowed range): the torch
            // chain would raise; here the draw falls back to the first allowed id instead of reading cand_i[-1]
            if (tid == 0) sm.pick[0] = nk > 0 ? sm.cand_i[min(sm.pick[1], nk - 1)] : 0;
            __syncthreads();
            choice = sm.pick[0];
        } else {   // plain multinomial over the whole segment
            const unsigned long long k = block_max(x.local_max(), sm, 0);
            int i0, i1;
            draw_full(x, sm, key_val(k), -1, u01(r.x), 0.f, i0, i1);
            choice = i0 >= 0 ? i0 : key_idx(k);
        }
    }
    choice = min(max(choice, 0), m - 1);   // an id outside [lo, lo + m) never reaches the output or the embedding lookup below
    if (tid == 0) out[(long)row * nseg + seg] = lo + choice;
    if (tail.ids) {   // nseg == 1
        __syncthreads();
        if (tid == 0) {
            long id = lo + choice;
            if (tail.unfinished) {
                const bool u = tail.unfinished[row] != 0;
                id = u ? id : tail.pad;
                tail.unfinished[row] = u && id != tail.eos;
            }
            tail.ids[row] = id;
            const long col = *step;
            if (tail.seq && col < tail.seq_ld) tail.seq[(long)row * tail.seq_ld + col] = id;
            sm.sel[0] = (unsigned)id;
        }
        __syncthreads();
        if (tail.emb) {
            const uint16_t *src = tail.emb + (long)sm.sel[0] * tail.D;
            uint16_t *dst = tail.x + (long)row * tail.D;
            for (int d = tid * 8; d < tail.D; d += kSmpThreads * 8) *reinterpret_cast<uint4 *>(dst + d) = *reinterpret_cast<const uint4 *>(src + d);
        }
    }
}

// ---- CosyVoice streaming step: draw + bookkeeping --------------------------------------------------------------------------------
template <int EPT>
__global__ __launch_bounds__(kSmpThreads) void ras_step_kernel(int V, const float *__restrict__ logits, long *__restrict__ tok,
                                                               long *__restrict__ recent, long *__restrict__ ptr, long *__restrict__ step_i,
                                                               long n_ignore, int eos, float top_p, int top_k, int win_size, float tau_r,
                                                               uint2 key) {
    __shared__ SmpShared sm;
    const int tid = threadIdx.x;
    Vals<EPT> x;
#pragma unroll
    for (int e = 0; e < EPT; e++) {
        const int j = tid + kSmpThreads * e;
        const float t = logits[min(j, V - 1)];
        x.v[e] = j < V ? t : -INFINITY;
    }
    const long st = *step_i;
    long rc[2];   // the ring of recent ids, one entry per lane of wave 0 (win_size <= 128)
    rc[0] = recent[min(tid, win_size - 1)];
    rc[1] = recent[min(64 + tid, win_size - 1)];
    const bool ignore_eos = st < n_ignore;
    const uint4 r = philox(make_uint4((uint32_t)st, (uint32_t)((uint64_t)st >> 32), 0u, 0x7a5u), key);
    const unsigned long long kmx = block_max(x.local_max(), sm, 0);
    const float mx = key_val(kmx);
    const int imx = key_idx(kmx);
    // softmax denominator over everything (the nucleus is cut on these probabilities, EOS included)
    float zs = 0.f;
#pragma unroll
    for (int e = 0; e < EPT; e++) zs += __expf(x.v[e] - mx);
    const float z = block_sum(zs, sm);
    // random_sampling (and the "nucleus is EOS alone" case) from the full distribution, without EOS while it is being rejected
    int full, alt;
    draw_full(x, sm, mx, ignore_eos ? eos : -1, u01(r.x), u01(r.y), full, alt);
    if (full < 0) full = imx;
    if (alt < 0) alt = imx;
    const int nc = select_bins(x, sm, min(top_k, V), false);
    // nucleus_sampling: the sorted prefix with cumulative probability (before adding) < top_p and rank < top_k, without EOS while it
    // is being rejected; one thread per candidate
    const float pj = tid < nc ? __expf(sm.cand_v[tid] - mx) / z : 0.f;
    float tot;
    const float cj = block_scan(pj, tot, sm);
    const bool keep = tid < nc && cj - pj < top_p;
    const float wj = keep && !(ignore_eos && sm.cand_i[min(tid, kSmpMaxCand - 1)] == eos) ? pj : 0.f;
    float mass;
    const float mj = block_scan(wj, mass, sm);
    if (tid == 0) {
        sm.pick[0] = 0x7fffffff;   // first kept rank whose inclusive mass exceeds the target
        sm.pick[1] = -1;           // last kept rank (a target that rounding pushed past the end)
    }
    __syncthreads();
    const float target = u01(r.z) * mass;
    if (wj > 0.f) {
        if (mj > target) atomicMin(&sm.pick[0], tid);
        atomicMax(&sm.pick[1], tid);
    }
    __syncthreads();
    if (tid < 64) {
        int cand = alt;   // the kept set holds nothing but EOS
        if (mass > 0.f) cand = sm.cand_i[sm.pick[0] != 0x7fffffff ? sm.pick[0] : sm.pick[1]];
        int rep = 0;
        for (int w0 = 0; w0 < win_size; w0 += 64) rep += __popcll(__ballot(w0 + tid < win_size && rc[w0 / 64] == (long)cand));
        if (tid == 0) {
            const long id = (float)rep >= (float)win_size * tau_r ? (long)full : (long)cand;
            *tok = id;
            if (id != (long)eos) {   // the reference appends emitted ids only
                const long p = *ptr;
                recent[p] = id;
                *ptr = (p + 1) % win_size;
            }
            *step_i = st + 1;
        }
    }
}

// ---- XY frame: the bookkeeping of CustomGenerationMixin._sample (model/llm/xy_llm.py:104-146) for one frame --------------------------
// One wave, lane = sequence (B <= 64).  The torch form of this (xy_llm._XYFrameState.step) is ~45 launches per frame.  Semantics
// (identical to that function, which stays as the reference path and is compared id for id in tests/test_heads_gpu.py):
//   a non-audio id on channel 0 starts a countdown of C - 1 further rows (needs = C - 1 .. -1); while it runs channel 0 carries EOS
//   (if configured) and channel i keeps its drawn ids for i more rows, then the pad id; finished sequences emit EOS / pad; the row
//   is appended at `pos` while the reference's loop would still be running (`all_done` not yet set); stop = length bound | EOS hit
//   (flushing sequences excepted unless reference_termination); unfinished &= ~stop & ~(needs == -1 [& flushing]).
struct XYFrameArgs {
    int B, C, rows;                 // out is [B][rows][C]
    long text_shift, speech_vocab, pad, eos0, total;   // eos0 < 0: no EOS id; total < 0: no length bound
    int n_eos, reference_termination;
};
__global__ __launch_bounds__(64) void xy_frame_kernel(XYFrameArgs a, const long *__restrict__ nt, const long *__restrict__ eos_list,
                                                      long *__restrict__ out, long *__restrict__ row, long *__restrict__ pos,
                                                      long *__restrict__ unfinished, long *__restrict__ needs_, unsigned char *__restrict__ all_done,
                                                      long *__restrict__ n_rows) {
    const int b = threadIdx.x, C = a.C;
    const bool live = b < a.B;
    const bool running = *all_done == 0;
    const long p0 = *pos;
    const long pw = min(p0, (long)a.rows - 1);                 // where the row goes (clamped like the torch form)
    const long p1 = p0 + (running ? 1 : 0);
    long u = 1, needs = -1;
    if (live) {
        u = unfinished[b];
        needs = needs_[b];
        const long *t = nt + (long)b * C;
        const long t0 = t[0];
        const bool is_audio = t0 >= a.text_shift && t0 < a.text_shift + a.speech_vocab;
        if (!is_audio && needs < 0) needs = C - 1;
        const bool flushing = needs >= 0;
        const long pddp_text = a.eos0 >= 0 ? a.eos0 : 0;
        long c0 = (a.eos0 >= 0 && flushing) ? a.eos0 : t0;
        c0 = u ? c0 : pddp_text;
        long *ob = out + ((long)b * a.rows + pw) * C, *rb = row + (long)b * C;
        if (running) ob[0] = c0;
        rb[0] = c0;
        for (int i = 1; i < C; i++) {
            long ci = (flushing && needs < C - i) ? a.pad : t[i];
            ci = u ? ci : a.pad;
            if (running) ob[i] = ci;
            rb[i] = ci;
        }
        if (flushing) needs -= 1;
        bool stop = a.total >= 0 && p1 >= a.total;
        bool hit = false;
        for (int e = 0; e < a.n_eos; e++) hit |= c0 == eos_list[e];
        stop |= a.reference_termination ? hit : (hit && !flushing);
        const bool gone = a.reference_termination ? needs == -1 : (needs == -1 && flushing);
        const long un = (u && !stop && !gone) ? 1 : 0;
        if (running) {
            unfinished[b] = un;
            needs_[b] = needs;
            u = un;
        }
    }
    const bool any = __ballot(live && u != 0) != 0;            // u: the value `unfinished` holds after this frame
    if (b == 0) {
        *pos = p1;
        *n_rows += running ? 1 : 0;
        if (!any) *all_done = 1;
    }
}

// the next frame's input: x[b] = emb_0[row[b][0]] + emb_1[row[b][1]] + ... in bf16, added in channel order like the module
// (model/llm/xy_llm.py:189-200: a chain of bf16 tensor additions)
constexpr int kXYMaxC = 16;
struct XYTables {
    const uint16_t *t[kXYMaxC];
};
__global__ __launch_bounds__(256) void xy_embed_kernel(int C, int D, XYTables tb, const long *__restrict__ row, uint16_t *__restrict__ x) {
    const int b = blockIdx.x;
    for (int d = threadIdx.x * 8; d < D; d += 256 * 8) {
        float acc[8];
#pragma unroll
        for (int c = 0; c < kXYMaxC; c++) {
            if (c < C) {
                const uint4 r = *reinterpret_cast<const uint4 *>(tb.t[c] + row[(long)b * C + c] * D + d);
                const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const float lo = __uint_as_float(w[j] << 16), hi = __uint_as_float(w[j] & 0xffff0000u);
                    if (c == 0) {
                        acc[2 * j] = lo;
                        acc[2 * j + 1] = hi;
                    } else {   // bf16 + bf16 -> bf16 (round to nearest even), as torch adds two bf16 tensors
                        float s0 = acc[2 * j] + lo, s1 = acc[2 * j + 1] + hi;
                        uint32_t u0 = __float_as_uint(s0), u1 = __float_as_uint(s1);
                        u0 += 0x7fffu + ((u0 >> 16) & 1u);
                        u1 += 0x7fffu + ((u1 >> 16) & 1u);
                        acc[2 * j] = __uint_as_float(u0 & 0xffff0000u);
                        acc[2 * j + 1] = __uint_as_float(u1 & 0xffff0000u);
                    }
                }
            }
        }
        uint4 o;
        uint32_t *ow = reinterpret_cast<uint32_t *>(&o);
#pragma unroll
        for (int j = 0; j < 4; j++) ow[j] = (__float_as_uint(acc[2 * j]) >> 16) | (__float_as_uint(acc[2 * j + 1]) & 0xffff0000u);
        *reinterpret_cast<uint4 *>(x + (long)b * D + d) = o;
    }
}

// elements per thread: the XY channels (1025 ids), the Spark / Cosy vocabularies (8193, 6562), the LDS-free maximum
constexpr int kEptS = 5, kEptM = 33, kEptL = kSmpMaxN / kSmpThreads;

}  // namespace

int sample_rows_f32(int rows, int nseg, const float *logits, long ld, const int *seg_off, const int *seg_len, const int *allow_lo,
                    const int *allow_hi, const int *suppress, int nsuppress, int max_domain, int do_sample, int top_k, float top_p,
                    float temperature, unsigned long long seed, const long *step, long *out, const void *tail_, int min_id, long min_until,
                    hipStream_t st) {
    SmpTail tail = {};
    if (tail_) {
        tail = *(const SmpTail *)tail_;
        if (nseg != 1 || !tail.ids || (tail.emb && (tail.D % 8 != 0 || !tail.x))) return -4;
    }
    if (max_domain > kSmpMaxN || nsuppress > 256) return -4;                      // RWKV7_ESHAPE
    if (do_sample && (top_k < 0 || top_k > 64 || (top_k == 0 && top_p < 1.f) || !(temperature > 0.f))) return -4;
    (void)hipGetLastError();
    const float it = do_sample ? 1.f / temperature : 1.f;
    const uint2 key = make_uint2((uint32_t)seed, (uint32_t)(seed >> 32));
    const dim3 grid(rows * nseg), block(kSmpThreads);
    if (max_domain <= kEptS * kSmpThreads)
        sample_rows_kernel<kEptS><<<grid, block, 0, st>>>(nseg, logits, ld, seg_off, seg_len, allow_lo, allow_hi, suppress, nsuppress, do_sample,
                                                          top_k, top_p, it, key, step, out, tail, min_id, min_until);
    else if (max_domain <= kEptM * kSmpThreads)
        sample_rows_kernel<kEptM><<<grid, block, 0, st>>>(nseg, logits, ld, seg_off, seg_len, allow_lo, allow_hi, suppress, nsuppress, do_sample,
                                                          top_k, top_p, it, key, step, out, tail, min_id, min_until);
    else
        sample_rows_kernel<kEptL><<<grid, block, 0, st>>>(nseg, logits, ld, seg_off, seg_len, allow_lo, allow_hi, suppress, nsuppress, do_sample,
                                                          top_k, top_p, it, key, step, out, tail, min_id, min_until);
    return (int)hipGetLastError();
}

int ras_step_f32(int V, const float *logits, long *tok, long *recent, long *ptr, long *step_i, long n_ignore, int eos, float top_p,
                 int top_k, int win_size, float tau_r, unsigned long long seed, hipStream_t st) {
    if (V > kSmpMaxN || top_k < 1 || top_k > kSmpMaxCand || win_size < 1 || win_size > 128) return -4;
    (void)hipGetLastError();
    const uint2 key = make_uint2((uint32_t)seed, (uint32_t)(seed >> 32));
    if (V <= kEptS * kSmpThreads)
        ras_step_kernel<kEptS><<<dim3(1), dim3(kSmpThreads), 0, st>>>(V, logits, tok, recent, ptr, step_i, n_ignore, eos, top_p, top_k, win_size, tau_r, key);
    else if (V <= kEptM * kSmpThreads)
        ras_step_kernel<kEptM><<<dim3(1), dim3(kSmpThreads), 0, st>>>(V, logits, tok, recent, ptr, step_i, n_ignore, eos, top_p, top_k, win_size, tau_r, key);
    else
        ras_step_kernel<kEptL><<<dim3(1), dim3(kSmpThreads), 0, st>>>(V, logits, tok, recent, ptr, step_i, n_ignore, eos, top_p, top_k, win_size, tau_r, key);
    return (int)hipGetLastError();
}

int xy_frame_step(int B, int C, int rows, long text_shift, long speech_vocab, long pad, long eos0, long total, const long *eos_list, int n_eos,
                  int reference_termination, const long *nt, long *out, long *row, long *pos, long *unfinished, long *needs,
                  unsigned char *all_done, long *n_rows, hipStream_t st) {
    if (B > 64 || C < 1 || rows < 1) return -4;
    XYFrameArgs a;
    a.B = B; a.C = C; a.rows = rows;
    a.text_shift = text_shift; a.speech_vocab = speech_vocab; a.pad = pad; a.eos0 = eos0; a.total = total;
    a.n_eos = n_eos; a.reference_termination = reference_termination;
    (void)hipGetLastError();
    xy_frame_kernel<<<dim3(1), dim3(64), 0, st>>>(a, nt, eos_list, out, row, pos, unfinished, needs, all_done, n_rows);
    return (int)hipGetLastError();
}

int xy_embed_bf16(int B, int C, int D, const void *const *tables_host, const long *row, void *x, hipStream_t st) {
    if (C < 1 || C > kXYMaxC || D % 8 != 0) return -4;
    XYTables tb;
    for (int c = 0; c < kXYMaxC; c++) tb.t[c] = (const uint16_t *)tables_host[c < C ? c : 0];
    (void)hipGetLastError();
    xy_embed_kernel<<<dim3(B), dim3(256), 0, st>>>(C, D, tb, row, (uint16_t *)x);
    return (int)hipGetLastError();
}

}  // namespace rwkv7
