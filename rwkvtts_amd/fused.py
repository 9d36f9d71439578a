"""Fused elementwise stages around the WKV7 scan, served by hand-written HIP kernels in librwkv7_hip.so
(rwkvtts_amd/csrc/elementwise.hip; C ABI in include/rwkv7_hip.h).

Reference formulas: model/llm/rwkv_s2s_single_ffn.py:160-169 (token shift + 6 lerps), :172-190 (decay / value
residual / kk / k' / scan operands), :192-195 (GroupNorm + bonus + gate), :224-228 (channel-mix shift, relu^2).
Each stage is one kernel forward and one backward (torch.autograd.Function); parameter gradients come back as
per-workgroup fp32 partials that are summed here.  HIP tensors only -- there is no CPU or eager route.
"""
import ctypes
import os

import torch

from . import _lib

_FWD_BLOCKS = 32768  # workgroups walking the rows (D/8 threads each): one row per workgroup at B*T = 32768 beats a row loop
                     # (tools/bench_fwd_blocks.py: prepare 178 -> 165 us, post 80 -> 71, add+LayerNorm 48 -> 44.5)
_MIX_FWD_BLOCKS = 2048   # the token-shift kernel re-reads the previous row at the start of a run: fewer, longer runs (106 -> 103 us)
_BWD_BLOCKS = 1024   # also the number of parameter-gradient partials (round 2, same-box A/B: 2048 -> 1024 -0.6...-0.9 ms per step)
_MIX_BWD_ROWS = 4     # rows per run in mix_bwd (neighbours carried in registers inside a run)
_MIX_BWD_BLOCKS = 1024


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream(t):
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _sfx(t):
    if not t.is_cuda:
        raise NotImplementedError("fused RWKV-7 stages run on the HIP device only (no CPU path)")
    if t.dtype == torch.bfloat16:
        return "bf16"
    if t.dtype == torch.float32:
        return "f32"
    raise TypeError(f"bf16 or fp32 expected, got {t.dtype}")


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


def _call(name, ref, *args):
    with torch.cuda.device_of(ref):
        rc = getattr(_lib.lib(), f"rwkv7_{name}_{_sfx(ref)}")(*args, _stream(ref))
    _lib.check(rc, name)


COLSUM_KERNEL = os.environ.get("RWKV7_COLSUM_KERNEL", "1") == "1"


def _colsum(part, dtype):
    """Column sums of a stage's parameter-gradient partials, [nb, ..., D] fp32 -> [..., D] `dtype`.  bf16 on the device: one launch
    (rwkv7_sum_slabs_bf16's tall shape, fixed summation order) instead of torch's reduce + cast pair (15 + 5 us, ~120 per step)."""
    if COLSUM_KERNEL and part.is_cuda and dtype == torch.bfloat16 and part.dtype == torch.float32 and part.shape[0] >= 256:
        n = part[0].numel()
        if n % 4 == 0 and n <= 32768:
            out = torch.empty(part.shape[1:], dtype=dtype, device=part.device)
            with torch.cuda.device_of(part):
                rc = _lib.lib().rwkv7_sum_slabs_bf16(ctypes.c_long(n), part.shape[0], _p(part), _p(out), 0, _stream(part))
            _lib.check(rc, "sum_slabs(colsum)")
            return out
    return part.sum(0).to(dtype)


def _mask_rows(mask, like):
    """mask [B,T,1] / [B,T] / None  ->  contiguous [B*T] tensor of like.dtype, or None."""
    if mask is None:
        return None
    return mask.reshape(-1).to(like.dtype).contiguous()


class _Mix(torch.autograd.Function):
    """nmix outputs are separate autograd outputs (slices of one buffer), so backward receives nmix separate
    gradients and hands their pointers to the kernel -- no torch.stack of 6 x [B,T,D] on the way back."""

    @staticmethod
    def forward(ctx, x, x_prev, mask, params):
        B, T, D = x.shape
        x, params = _c(x), _c(params)
        nmix = params.shape[0]
        out = torch.empty(nmix, B, T, D, dtype=x.dtype, device=x.device)
        xp = None if x_prev is None else _c(x_prev.to(x.dtype))
        _call("mix_fwd", x, B, T, D, nmix, _p(x), _p(xp), _p(mask), _p(params), _p(out), min(B * T, _MIX_FWD_BLOCKS))
        ctx.save_for_backward(x, xp, mask, params)
        return tuple(out[i] for i in range(nmix))

    @staticmethod
    def backward(ctx, *gs):
        x, xp, mask, params = ctx.saved_tensors
        B, T, D = x.shape
        nmix = params.shape[0]
        gs = [torch.zeros_like(x) if g is None else _c(g) for g in gs]
        # workgroup b walks the _MIX_BWD_ROWS-row runs b, b + nb, b + 2 nb, ... (see mix_bwd_kernel)
        nb = max(1, min(-(-B * T // _MIX_BWD_ROWS), _MIX_BWD_BLOCKS))
        dx = torch.empty_like(x)
        part = torch.empty(nb, nmix, D, dtype=torch.float32, device=x.device)
        ptrs = (ctypes.c_void_p * nmix)(*[g.data_ptr() for g in gs])
        _call("mix_bwd", x, B, T, D, nmix, ptrs, _p(x), _p(xp), _p(mask), _p(params), _p(dx), _p(part), nb, _MIX_BWD_ROWS)
        return dx, None, None, _colsum(part, params.dtype)


# Round 4: the low-rank branches' DOWN projections commute with the token-shift lerp,
#     (xm (1 - mu) + shift(xm) mu) W1^T = xm (W1 * (1 - mu))^T + shift(xm) (W1 * mu)^T ,
# so the four mixed inputs x_w, x_a, x_v(branch), x_g (rwkv_s2s_single_ffn.py:160-169, consumed only by Linear(D, r) of the w / a / v / g
# branches, :171-190) never have to exist: ONE GEMM G = x [W_a ; W_b]^T ([M, D] x [D, 2 R], R = 64 + 64 + 32 + 128) on the LayerNorm
# output, then h_i[t] = m_t G_a[t] + m_{t-1} G_b[t - 1] on the small side (the shift, the mask and the sequence start are applied to
# [M, R] instead of [M, D]).  Per layer, forward: three lerp outputs instead of six, one projection GEMM instead of four; backward: the
# lerp stage reads three gradients instead of six, one input-gradient GEMM accumulating into the lerp stage's dx (beta = 1) instead of
# four that each write [M, D], one weight-gradient reduction over M instead of four.  The gradients of W1 and mu come out of
# d[W_a ; W_b] through the two small products above (autograd).  The mixed inputs are not rounded to bf16 on the way (the reference
# rounds x_i, then projects): one rounding less, inside the parity bars of tests/test_model_gpu.py.  RWKV7_FUSED_MIX_LORA=0: off.
FUSED_MIX_LORA = os.environ.get("RWKV7_FUSED_MIX_LORA", "1") == "1"
FUSED_MIX_LORA_HITS = [0]
WGRAD_SLABS_WCAT = 32


class _MixLora(torch.autograd.Function):
    """(x_r, x_k, x_v, G) = (the three token-shift lerps that feed full projections, x @ wcat^T)."""

    @staticmethod
    def forward(ctx, x, mask, params, wcat):
        B, T, D = x.shape
        x, params, wcat = _c(x), _c(params), _c(wcat)
        nmix = params.shape[0]
        out = torch.empty(nmix, B, T, D, dtype=x.dtype, device=x.device)
        _call("mix_fwd", x, B, T, D, nmix, _p(x), _p(None), _p(mask), _p(params), _p(out), min(B * T, _MIX_FWD_BLOCKS))
        G = torch.mm(x.view(-1, D), wcat.t())
        ctx.save_for_backward(x, mask, params, wcat)
        FUSED_MIX_LORA_HITS[0] += 1
        return (*[out[i] for i in range(nmix)], G.view(B, T, -1))

    @staticmethod
    def backward(ctx, *gs):
        x, mask, params, wcat = ctx.saved_tensors
        B, T, D = x.shape
        nmix = params.shape[0]
        dG = gs[nmix]
        gs = [torch.zeros_like(x) if g is None else _c(g) for g in gs[:nmix]]
        nb = max(1, min(-(-B * T // _MIX_BWD_ROWS), _MIX_BWD_BLOCKS))
        dx = torch.empty_like(x)
        part = torch.empty(nb, nmix, D, dtype=torch.float32, device=x.device)
        ptrs = (ctypes.c_void_p * nmix)(*[g.data_ptr() for g in gs])
        _call("mix_bwd", x, B, T, D, nmix, ptrs, _p(x), _p(None), _p(mask), _p(params), _p(dx), _p(part), nb, _MIX_BWD_ROWS)
        dwcat = None
        if dG is not None:
            dG2 = _c(dG).view(-1, dG.shape[-1])
            dx.view(-1, D).addmm_(dG2, wcat)          # the branches' input gradient accumulates into the lerp stage's dx
            dwcat = wgrad_splitk(dG2, x.view(-1, D), slabs=WGRAD_SLABS_WCAT)   # [2 R, D] = [576, 1024]: 69 us with 32 slabs, 100 with 8
        return dx, None, _colsum(part, params.dtype), dwcat


def mix_lora_supported(x, state, seq_start, w1s=None, mask=None):
    """w1s: the down-projection weights of the low-rank branches that would ride on the fused path (a list, or a callable that builds
    it -- called only when everything else already holds, so eager decode / packed rows do not pay for the lists); what the C entry points
    require of them (rwkv7_mix_lora_wcat_*: at most 4 branches, every rank a multiple of 8, bf16, contiguous) is checked HERE so
    that an unusual config (rank 20, fp32 LoRA) falls back to token_shift_mix6 instead of raising RWKV7_ESHAPE mid-training."""
    # packed rows (seq_start): the 32-aligned layout of RWKV7Model._forward_packed puts at least one MASKED position in front of every
    # sequence, so "nothing from t - 1 at a sequence start" is the ordinary mask handling of the combine kernels (m_{t-1} = 0) -- the
    # fused path applies as long as the mask is there (round 6: the packed step had been running the six-lerp path, +8 ms per step)
    if not (FUSED_MIX_LORA and state is None and (seq_start is None or mask is not None) and x.is_cuda and x.dtype == torch.bfloat16 and torch.is_grad_enabled()
            and x.requires_grad and x.dim() == 3 and x.shape[0] * x.shape[1] >= WGRAD_MIN_ROWS and x.shape[-1] % 8 == 0):
        return False     # the cheap rejections first (decode, packed rows, no grad): the weights are looked at only on the training path
    if w1s is not None:
        w1s = w1s() if callable(w1s) else w1s
        return 0 < len(w1s) <= 4 and all(w.dtype == torch.bfloat16 and w.is_contiguous() and w.dim() == 2 and w.shape[0] % 8 == 0 for w in w1s)
    return True


def _ptr_array(ts):
    return (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])


class _WcatBuild(torch.autograd.Function):
    """wcat [2 R, D] = [W1_i * (1 - mu_i) ; W1_i * mu_i] of all branches (rwkv7_mix_lora_wcat_*: one small kernel each way instead of
    a dozen tensor ops per branch)."""

    @staticmethod
    def forward(ctx, nb, *ts):
        w1s, mus = [_c(t) for t in ts[:nb]], [_c(t) for t in ts[nb:]]
        D = w1s[0].shape[1]
        ranks = (ctypes.c_int * nb)(*[w.shape[0] for w in w1s])
        R = sum(w.shape[0] for w in w1s)
        wcat = torch.empty(2 * R, D, dtype=w1s[0].dtype, device=w1s[0].device)
        with torch.cuda.device_of(wcat):
            rc = _lib.lib().rwkv7_mix_lora_wcat_fwd_bf16(nb, ranks, _ptr_array(w1s), _ptr_array(mus), D, _p(wcat), _stream(wcat))
        _lib.check(rc, "mix_lora_wcat_fwd")
        ctx.save_for_backward(*w1s, *mus)
        ctx.nb = nb
        ctx.mu_shapes = [t.shape for t in ts[nb:]]
        return wcat

    @staticmethod
    def backward(ctx, dwcat):
        nb = ctx.nb
        ts = ctx.saved_tensors
        w1s, mus = ts[:nb], ts[nb:]
        D = w1s[0].shape[1]
        ranks = (ctypes.c_int * nb)(*[w.shape[0] for w in w1s])
        dwcat = _c(dwcat)
        dw1 = [torch.empty_like(w) for w in w1s]
        dmu = [torch.empty(D, dtype=w1s[0].dtype, device=w1s[0].device) for _ in range(nb)]
        with torch.cuda.device_of(dwcat):
            rc = _lib.lib().rwkv7_mix_lora_wcat_bwd_bf16(nb, ranks, _ptr_array(w1s), _ptr_array(mus), D, _p(dwcat), _ptr_array(dw1),
                                                         _ptr_array(dmu), _stream(dwcat))
        _lib.check(rc, "mix_lora_wcat_bwd")
        return (None, *dw1, *[g.view(sh) for g, sh in zip(dmu, ctx.mu_shapes)])


_ACT_CODE = {None: 0, "tanh": 1, "sigmoid": 2}


class _CombineAct(torch.autograd.Function):
    """a_i[t] = act_i(bf16(m_t G_a[t] + m_{t-1} G_b[t - 1])) per branch, contiguous [B, T, r_i] (rwkv7_mix_lora_combine_*)."""

    @staticmethod
    def forward(ctx, G, mask, ranks, acts):
        B, T, R2 = G.shape
        G = _c(G)
        nb = len(ranks)
        outs = [torch.empty(B, T, r, dtype=G.dtype, device=G.device) for r in ranks]
        cr, ca = (ctypes.c_int * nb)(*ranks), (ctypes.c_int * nb)(*acts)
        with torch.cuda.device_of(G):
            rc = _lib.lib().rwkv7_mix_lora_combine_fwd_bf16(nb, cr, ca, ctypes.c_long(B * T), T, _p(G), _p(mask), _ptr_array(outs), _stream(G))
        _lib.check(rc, "mix_lora_combine_fwd")
        ctx.save_for_backward(mask, *outs)
        ctx.ranks, ctx.acts, ctx.shape = ranks, acts, (B, T, R2)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *das):
        mask, *outs = ctx.saved_tensors
        B, T, R2 = ctx.shape
        nb = len(ctx.ranks)
        das = [torch.zeros_like(o) if g is None else _c(g) for g, o in zip(das, outs)]
        dG = torch.empty(B, T, R2, dtype=outs[0].dtype, device=outs[0].device)
        cr, ca = (ctypes.c_int * nb)(*ctx.ranks), (ctypes.c_int * nb)(*ctx.acts)
        with torch.cuda.device_of(dG):
            rc = _lib.lib().rwkv7_mix_lora_combine_bwd_bf16(nb, cr, ca, ctypes.c_long(B * T), T, _p(mask), _ptr_array(outs), _ptr_array(das),
                                                            _p(dG), _stream(dG))
        _lib.check(rc, "mix_lora_combine_bwd")
        return dG, None, None, None


# Round 6 (late): the branches' down projections with the lerp as the A PROLOGUE of one own MFMA kernel (csrc/lora_down.hip) instead of the
# library GEMM on [W_a ; W_b] + the combine kernel: the N = 2 R = 576 GEMM runs at 0.63 PF/s in the library (61 us per layer, the time of
# N = 1024) and the combine kernel adds 20 us; the own kernel streams x once, rounds the mixed inputs to bf16 where the reference does
# (one rounding the through-the-lerp form skipped) and writes the activated [M, r_i] directly.  Forward only: the backward below is the
# through-the-lerp gradient unchanged (combine_bwd -> dG; dx += dG wcat; d wcat = dG^T x; wcat_bwd), with wcat rebuilt there.
# RWKV7_LORA_DOWN_DIRECT=0: the round-4 pair.
LORA_DOWN_DIRECT = os.environ.get("RWKV7_LORA_DOWN_DIRECT", "1") == "1"
LORA_DOWN_DIRECT_HITS = [0]


def lora_down_direct_supported(x, w1s):
    """What rwkv7_lora_down_fwd_bf16 takes (csrc/lora_down.hip): rows a multiple of 128, D of 128, ranks multiples of 32, 2..10 column tiles of
    32 that cut into two contiguous halves of at most five tiles and two branches each (lora_down_cut) -- else the round-4 pair runs."""
    B, T, D = x.shape
    ranks = [w.shape[0] for w in w1s]
    if not (LORA_DOWN_DIRECT and (B * T) % 128 == 0 and D % 128 == 0 and D <= 4096 and all(r % 32 == 0 for r in ranks)):
        return False
    tiles = [b for b, r in enumerate(ranks) for _ in range(r // 32)]     # branch of every 32-column tile
    ok = lambda h: 1 <= len(h) <= 5 and len(set(h)) <= 2
    lds = 2 * (3 * 129 * 72 + 2 * len(tiles) * 2048 + len(ranks) * D)      # x buffers + two W1 stage buffers + the coefficients (launch_lora_down)
    return 2 <= len(tiles) <= 10 and lds <= 160 * 1024 and any(ok(tiles[:c]) and ok(tiles[c:]) for c in range(1, len(tiles)))


class _MixLoraDirect(torch.autograd.Function):
    """(x_r, x_k, x_v, a_1 .. a_nb) = the three lerps that feed full projections and the branches' activated hidden states."""

    @staticmethod
    def forward(ctx, x, mask, params, nb, acts, *ts):
        B, T, D = x.shape
        x, params = _c(x), _c(params)
        w1s, mus = [_c(t) for t in ts[:nb]], [_c(t.reshape(-1)) for t in ts[nb:]]
        nmix = params.shape[0]
        out = torch.empty(nmix, B, T, D, dtype=x.dtype, device=x.device)
        _call("mix_fwd", x, B, T, D, nmix, _p(x), _p(None), _p(mask), _p(params), _p(out), min(B * T, _MIX_FWD_BLOCKS))
        ranks = [w.shape[0] for w in w1s]
        cr, ca = (ctypes.c_int * nb)(*ranks), (ctypes.c_int * nb)(*acts)
        packed = torch.empty(sum(ranks), D, dtype=x.dtype, device=x.device)
        hs = [torch.empty(B, T, r, dtype=x.dtype, device=x.device) for r in ranks]
        with torch.cuda.device_of(x):
            rc = _lib.lib().rwkv7_lora_down_pack_bf16(nb, cr, _ptr_array(w1s), D, _p(packed), _stream(x))
            _lib.check(rc, "lora_down_pack")
            rc = _lib.lib().rwkv7_lora_down_fwd_bf16(nb, cr, ca, ctypes.c_long(B * T), T, D, _p(x), _p(mask), _ptr_array(mus), _p(packed),
                                                     _ptr_array(hs), _stream(x))
            _lib.check(rc, "lora_down_fwd")
        ctx.save_for_backward(x, mask, params, *w1s, *mus, *hs)
        ctx.nb, ctx.acts, ctx.mu_shapes = nb, acts, [t.shape for t in ts[nb:]]
        LORA_DOWN_DIRECT_HITS[0] += 1
        FUSED_MIX_LORA_HITS[0] += 1
        return (*[out[i] for i in range(nmix)], *hs)

    @staticmethod
    def backward(ctx, *gs):
        nb = ctx.nb
        x, mask, params, *rest = ctx.saved_tensors
        w1s, mus, hs = rest[:nb], rest[nb:2 * nb], rest[2 * nb:]
        B, T, D = x.shape
        nmix = params.shape[0]
        ranks = [w.shape[0] for w in w1s]
        R = sum(ranks)
        cr, ca = (ctypes.c_int * nb)(*ranks), (ctypes.c_int * nb)(*ctx.acts)
        das = [torch.zeros_like(h) if g is None else _c(g) for g, h in zip(gs[nmix:], hs)]
        dG = torch.empty(B * T, 2 * R, dtype=x.dtype, device=x.device)
        with torch.cuda.device_of(x):
            rc = _lib.lib().rwkv7_mix_lora_combine_bwd_bf16(nb, cr, ca, ctypes.c_long(B * T), T, _p(mask), _ptr_array(hs), _ptr_array(das),
                                                            _p(dG), _stream(x))
            _lib.check(rc, "mix_lora_combine_bwd")
        g3 = [torch.zeros_like(x) if g is None else _c(g) for g in gs[:nmix]]
        nblk = max(1, min(-(-B * T // _MIX_BWD_ROWS), _MIX_BWD_BLOCKS))
        dx = torch.empty_like(x)
        part = torch.empty(nblk, nmix, D, dtype=torch.float32, device=x.device)
        ptrs = (ctypes.c_void_p * nmix)(*[g.data_ptr() for g in g3])
        _call("mix_bwd", x, B, T, D, nmix, ptrs, _p(x), _p(None), _p(mask), _p(params), _p(dx), _p(part), nblk, _MIX_BWD_ROWS)
        wcat = torch.empty(2 * R, D, dtype=x.dtype, device=x.device)
        with torch.cuda.device_of(x):
            rc = _lib.lib().rwkv7_mix_lora_wcat_fwd_bf16(nb, cr, _ptr_array(w1s), _ptr_array(mus), D, _p(wcat), _stream(x))
            _lib.check(rc, "mix_lora_wcat_fwd")
        # the library's addmm_ stays: an own kernel of this family did the shape in 69 us against 51.5 (profiles/r06zz_lora_dx_ab.txt)
        dx.view(-1, D).addmm_(dG, wcat)
        dwcat = _c(wgrad_splitk(dG, x.view(-1, D), slabs=WGRAD_SLABS_WCAT))
        dw1 = [torch.empty_like(w) for w in w1s]
        dmu = [torch.empty(D, dtype=x.dtype, device=x.device) for _ in range(nb)]
        with torch.cuda.device_of(x):
            rc = _lib.lib().rwkv7_mix_lora_wcat_bwd_bf16(nb, cr, _ptr_array(w1s), _ptr_array(mus), D, _p(dwcat), _ptr_array(dw1),
                                                         _ptr_array(dmu), _stream(x))
            _lib.check(rc, "mix_lora_wcat_bwd")
        return (dx, None, _colsum(part, params.dtype), None, None, *dw1, *[g.view(sh) for g, sh in zip(dmu, ctx.mu_shapes)])


def mix_lora(x, mask, x_r, x_k, x_v, mus, w1s, acts):
    """x [B,T,D] (LayerNorm output); mus / w1s / acts: the lerp coefficient, Linear(D, r_i) weight and activation name of each low-rank
    branch.  Returns (x_r, x_k, x_v, [a_i]): the three lerps that feed full projections and the branches' ACTIVATED hidden states
    [B,T,r_i] (the inputs of their Linear(r_i, D))."""
    B, T, D = x.shape
    params = torch.cat([p.reshape(1, D) for p in (x_r, x_k, x_v)], 0).to(x.dtype)
    mr = _mask_rows(mask, x)
    if lora_down_direct_supported(x, w1s):
        outs = _MixLoraDirect.apply(x, mr, params, len(w1s), tuple(_ACT_CODE[a] for a in acts), *w1s, *mus)
        return outs[0], outs[1], outs[2], list(outs[3:])
    wcat = _WcatBuild.apply(len(w1s), *w1s, *mus)
    xr, xk, xv, G = _MixLora.apply(x, mr, params, wcat)
    hs = _CombineAct.apply(G, mr, tuple(w.shape[0] for w in w1s), tuple(_ACT_CODE[a] for a in acts))
    return xr, xk, xv, list(hs)


def token_shift_mix6(x, x_prev, x_r, x_w, x_k, x_v, x_a, x_g, mask=None, stacked=None):
    """xm = x*mask ; xx = shift(xm) - xm ; returns xm + xx*x_? for ? in r,w,k,v,a,g  (6 tensors [B,T,D]).
    stacked: the six coefficient vectors already stacked [6,D] (inference caches it; one launch less per call)."""
    D = x.shape[-1]
    params = stacked if stacked is not None else torch.cat([p.reshape(1, D) for p in (x_r, x_w, x_k, x_v, x_a, x_g)], 0).to(x.dtype)
    return _Mix.apply(x, x_prev, _mask_rows(mask, x), params)


def token_shift_mix1(x, x_prev, x_k, mask=None):
    return _Mix.apply(x, x_prev, _mask_rows(mask, x), x_k.reshape(1, -1).to(x.dtype))[0]


class _ReluSq(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = _c(x)
        y = torch.empty_like(x)
        _call("relusq_fwd", x, ctypes.c_long(x.numel()), _p(x), _p(y))
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        dy = _c(dy)
        dx = torch.empty_like(x)
        _call("relusq_bwd", x, ctypes.c_long(x.numel()), _p(x), _p(dy), _p(dx))
        return dx


def relu_sq(x):
    return _ReluSq.apply(x)


# The channel-mix key projection with relu(.)^2 as the GEMM's epilogue (round 2: csrc/lab/gemm_relusq.hip, hand-written 256 x 256 x 64 MFMA kernel
# fed by LDS-DMA, persistent over the output tiles; bit-identical output; the pre-activation is never written and the backward takes
# 2 relu(x) = 2 sqrt(s) from the output, rwkv7_relusq_bwd_s).  A measured experiment, OFF by default:
# isolated (tools/bench_gemm_relusq.py, 32768 x 4096 x 1024) 297-302 us against 254-274 + 87 us for the library GEMM + rwkv7_relusq_fwd
# -- but inside the training step the library runs this GEMM at 223 us (1.23 PFLOP/s; other kernel / other clocks than in the
# loop benchmark) and the own kernel at 313 us (0.88 PFLOP/s): 313 against 310 us per layer, -0.13 ms per step in the same-box A/B
# (tools/ab_step.py) -- no gain, so the pair stays.  The bar was the library's rate on this shape in the same process.
# Round 4: the entry now runs the second-generation kernel (csrc/gemm_nt4.hip) for K % 1024 == 0 and the pair of fusions lives in
# channel_mix below (on by default); this stand-alone node stays as the A/B switch of the forward half (-1.02 ms on its own).
# bf16, M and N multiples of 256, K of 1024 (csrc/gemm_nt4.hip); RWKV7_FUSED_KEY_RELUSQ=1 (or the attribute) switches it on.
FUSED_KEY_RELUSQ = os.environ.get("RWKV7_FUSED_KEY_RELUSQ", "0") == "1"
FUSED_KEY_RELUSQ_HITS = [0]


def key_relusq_eligible(x, weight):
    M = x.numel() // x.shape[-1]
    return (FUSED_KEY_RELUSQ and x.is_cuda and x.dtype == torch.bfloat16 and weight.dtype == torch.bfloat16 and M % 256 == 0
            and weight.shape[0] % 256 == 0 and weight.shape[1] % 1024 == 0)


class _KeyReluSq(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight):
        x2 = _c(x).view(-1, x.shape[-1])
        w = _c(weight)
        M, K = x2.shape
        N = w.shape[0]
        s = torch.empty(M, N, dtype=x.dtype, device=x.device)
        with torch.cuda.device_of(x):
            rc = _lib.lib().rwkv7_gemm_nt_bf16(M, N, K, _p(x2), _p(w), _p(s), 1, _stream(x))
        _lib.check(rc, "gemm_nt_relusq")
        FUSED_KEY_RELUSQ_HITS[0] += 1
        ctx.save_for_backward(x, weight, s)
        ctx.wparam = weight
        return s.view(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, ds):
        x, weight, s = ctx.saved_tensors
        ds2 = _c(ds).view(-1, ds.shape[-1])
        dk = torch.empty_like(s)
        _call("relusq_bwd_s", s, ctypes.c_long(s.numel()), _p(s), _p(ds2), _p(dk))
        x2 = _c(x).view(-1, x.shape[-1])
        dx = _dgrad(dk, weight).view(x.shape) if ctx.needs_input_grad[0] else None
        dw = None
        if ctx.needs_input_grad[1]:
            dw = _wgrad(dk, x2, ctx.wparam)
        return dx, dw


# Round 4: the OTHER end of the activation.  value(relu(h)^2): forward unchanged (rwkv7_relusq_fwd + library GEMM); backward: the
# input gradient of the value projection and the activation's backward as ONE launch -- dh = bf16(dy W_value) * 2 relu(h) through the
# own MFMA GEMM with h as an auxiliary epilogue operand (rwkv7_gemm_nt_relusq_bwd_bf16): ds (256 MiB per layer) is neither written nor
# read back and rwkv7_relusq_bwd (140 us per layer) is not launched; the own GEMM is slower than the library's on this shape
# (0.30 against 0.22-0.25 ms), the pair it replaces is 0.39 ms on paper.  Measured (tools/ab_step.py, "relu^2 backward in the value
# dgrad", same box, 132.98 ms without / 133.09 ms with): a tie, like the forward twin above -- what the launch saves, the slower GEMM
# and the transposed copy of the weight give back.  On the second-generation GEMM (round 4, later) the same switch is -2.18 ms; it is
# subsumed by channel_mix below and stays as the A/B switch of the backward half.  Off; RWKV7_FUSED_RELUSQ_VALUE_BWD=1 switches it on.
FUSED_RELUSQ_VALUE_BWD = os.environ.get("RWKV7_FUSED_RELUSQ_VALUE_BWD", "0") == "1"


def relusq_value_eligible(h, weight):
    M = h.numel() // h.shape[-1]
    return (FUSED_RELUSQ_VALUE_BWD and h.is_cuda and h.dtype == torch.bfloat16 and weight.dtype == torch.bfloat16 and M % 256 == 0
            and weight.shape[1] % 256 == 0 and weight.shape[0] % 1024 == 0 and torch.is_grad_enabled())


class _ReluSqValue(torch.autograd.Function):
    """out = relu(h)^2 @ weight^T  (weight = value.weight [D, F])."""

    @staticmethod
    def forward(ctx, h, weight):
        h2 = _c(h).view(-1, h.shape[-1])
        s = torch.empty_like(h2)
        _call("relusq_fwd", h2, ctypes.c_long(h2.numel()), _p(h2), _p(s))
        out = torch.nn.functional.linear(s, weight)
        ctx.save_for_backward(h2, s, weight)
        ctx.wparam = weight
        ctx.shape = h.shape
        return out.view(*h.shape[:-1], weight.shape[0])

    @staticmethod
    def backward(ctx, dout):
        h2, s, weight = ctx.saved_tensors
        d2 = _c(dout).view(-1, dout.shape[-1])
        M, F = h2.shape
        D = weight.shape[0]
        dh = None
        if ctx.needs_input_grad[0]:
            wt = weight.detach().t().contiguous()     # [F, D]: the NT operand (8 MiB at 0.4B, one copy per layer and step)
            dh = torch.empty_like(h2)
            with torch.cuda.device_of(h2):
                rc = _lib.lib().rwkv7_gemm_nt_relusq_bwd_bf16(M, F, D, _p(d2), _p(wt), _p(h2), _p(dh), _stream(h2))
            _lib.check(rc, "gemm_nt_relusq_bwd")
            dh = dh.view(ctx.shape)
        dw = _wgrad(d2, s, ctx.wparam) if ctx.needs_input_grad[1] else None
        return dh, dw


# Round 4, second half: BOTH ends at once on the second-generation own GEMM (csrc/gemm_nt4.hip: four waves, quadrant phases, ring of
# half-tile slots; 0.89 of the library's rate as a plain GEMM, but the activation rides for free).  forward: s = relu(x W_key^T)^2 as
# the key GEMM's epilogue (h never written), out = s W_value^T (library); backward: dk = bf16(dout W_value) * 2 sqrt(s) as the epilogue of
# the value projection's input-gradient GEMM (ds never written), then the key projection's gradients as before.  Neither rwkv7_relusq_fwd
# nor rwkv7_relusq_bwd runs, and one [rows, F] activation less is kept per layer.  Same-box A/B (tools/ab_step.py): see DESIGN.md section 4.
FUSED_CMIX = os.environ.get("RWKV7_FUSED_CMIX", "1") == "1"
FUSED_CMIX_HITS = [0]
# Round 5: input gradients dx = dy @ W (W [N_out, K_in] as nn.Linear stores it) are an "NN" product for the library -- the contraction
# index is the ROW index of W -- and the kernels it picks for that layout are slower than the "NT" ones of the forward product (both
# operands contraction-contiguous): 76.5 against 63.7 us at 32768 x 1024 x 1024, 246 against 181 us for the channel-mix key's input
# gradient (profiles/r05m_dgrad_layout_probe.txt), with bit-identical results.  W^T is one small transpose (2-8 MB,
# rwkv7_transpose_bf16) per weight and step.  RWKV7_DGRAD_NT=0: off.
# Inside the training step it pays for the long contraction only (N_out = 4096: 24 x 65 us per step); the 1024-wide input gradients run
# at ~59 us in the step either way (profiles/r05o_step_busy.txt), so they keep the library's layout and save the transpose.
DGRAD_NT = os.environ.get("RWKV7_DGRAD_NT", "1") == "1"
DGRAD_NT_MIN_N = int(os.environ.get("RWKV7_DGRAD_NT_MIN_N", "2048"))
DGRAD_NT_HITS = [0]


def _dgrad(dy2, weight):
    """dy2 [M, N] @ weight [N, K] -> [M, K]."""
    N, K = weight.shape
    if (DGRAD_NT and dy2.is_cuda and dy2.dtype == torch.bfloat16 and weight.dtype == torch.bfloat16 and dy2.shape[0] >= WGRAD_MIN_ROWS
            and N % 64 == 0 and K % 64 == 0 and N >= DGRAD_NT_MIN_N and K >= 256 and weight.is_contiguous() and dy2.is_contiguous()):
        wt = torch.empty(K, N, dtype=weight.dtype, device=weight.device)
        with torch.cuda.device_of(dy2):
            rc = _lib.lib().rwkv7_transpose_bf16(N, K, _p(weight), _p(wt), _stream(dy2))
        _lib.check(rc, "transpose(dgrad)")
        DGRAD_NT_HITS[0] += 1
        return torch.mm(dy2, wt.t())
    return torch.mm(dy2, weight)


TRANSPOSE_KERNEL = True   # W_value^T through rwkv7_transpose_bf16 instead of torch's strided copy (27 -> ~6 us per layer)


def cmix_eligible(x, wk, wv):
    M = x.numel() // x.shape[-1]
    return (FUSED_CMIX and x.is_cuda and x.dtype == torch.bfloat16 and wk.dtype == torch.bfloat16 and wv.dtype == torch.bfloat16
            and M % 256 == 0 and wk.shape[0] % 256 == 0 and wk.shape[1] % 1024 == 0 and wv.shape[0] == wk.shape[1] and wv.shape[1] == wk.shape[0])


class _ChannelMix(torch.autograd.Function):
    """out = relu(x @ wk^T)^2 @ wv^T   (wk = key.weight [F, D], wv = value.weight [D, F]); rwkv_s2s_single_ffn.py:226-229."""

    @staticmethod
    def forward(ctx, x, wk, wv):
        x2 = _c(x).view(-1, x.shape[-1])
        wkc = _c(wk)
        M, D = x2.shape
        F = wkc.shape[0]
        s = torch.empty(M, F, dtype=x.dtype, device=x.device)
        with torch.cuda.device_of(x):
            rc = _lib.lib().rwkv7_gemm_nt_bf16(M, F, D, _p(x2), _p(wkc), _p(s), 1, _stream(x))
        _lib.check(rc, "gemm_nt_relusq")
        FUSED_CMIX_HITS[0] += 1
        out = torch.nn.functional.linear(s, wv)
        ctx.save_for_backward(x2, s, wk, wv)
        ctx.wk, ctx.wv = wk, wv
        ctx.shape = x.shape
        return out.view(*x.shape[:-1], wv.shape[0])

    @staticmethod
    def backward(ctx, dout):
        x2, s, wk, wv = ctx.saved_tensors
        d2 = _c(dout).view(-1, dout.shape[-1])
        M, F = s.shape
        D = wv.shape[0]
        if TRANSPOSE_KERNEL and D % 64 == 0 and F % 64 == 0 and wv.is_contiguous():
            wt = torch.empty(F, D, dtype=wv.dtype, device=wv.device)      # [F, D]: the NT operand (8 MiB at 0.4B, once per layer and step)
            with torch.cuda.device_of(s):
                rc = _lib.lib().rwkv7_transpose_bf16(D, F, _p(wv), _p(wt), _stream(s))
            _lib.check(rc, "transpose")
        else:
            wt = wv.detach().t().contiguous()
        dk = torch.empty_like(s)
        with torch.cuda.device_of(s):
            rc = _lib.lib().rwkv7_gemm_nt_relusq_bwd_s_bf16(M, F, D, _p(d2), _p(wt), _p(s), _p(dk), _stream(s))
        _lib.check(rc, "gemm_nt_relusq_bwd_s")
        dx = _dgrad(dk, wk).view(ctx.shape) if ctx.needs_input_grad[0] else None
        dwk = _wgrad(dk, x2, ctx.wk) if ctx.needs_input_grad[1] else None
        dwv = _wgrad(d2, s, ctx.wv) if ctx.needs_input_grad[2] else None
        return dx, dwk, dwv


def channel_mix(x, wk, wv):
    """value(relu(key(x))^2) with the activation inside both GEMMs, or None (shapes / dtype outside the own GEMM's range)."""
    if cmix_eligible(x, wk, wv):
        return _ChannelMix.apply(x, wk, wv)
    return None


def relu_sq_value(h, weight):
    """value(relu(h)^2) with the activation's backward inside the value projection's input-gradient GEMM, or None (shapes / dtype
    outside the own GEMM's range: the caller runs relu_sq + Linear)."""
    if relusq_value_eligible(h, weight):
        return _ReluSqValue.apply(h, weight)
    return None


def key_relu_sq(x, weight):
    """relu(x @ weight^T)^2 -- one kernel when the shapes allow it (see above), the library GEMM + relu_sq otherwise."""
    if key_relusq_eligible(x, weight):
        return _KeyReluSq.apply(x, weight)
    return None


class _TmixPrepare(torch.autograd.Function):
    @staticmethod
    def forward(ctx, w_pre, k, v, a_pre, v_pre, v_first, k_k, k_a, mask):
        B, T, D = k.shape
        w_pre, k, v, a_pre = _c(w_pre), _c(k), _c(v), _c(a_pre)
        v_pre = None if v_pre is None else _c(v_pre)
        v_first = None if v_first is None else _c(v_first)
        k_k, k_a = _c(k_k.to(k.dtype)), _c(k_a.to(k.dtype))
        outs = [torch.empty_like(k) for _ in range(5)]
        rows = B * T
        _call("tmix_prepare_fwd", k, ctypes.c_long(rows), D, _p(w_pre), _p(k), _p(v), _p(a_pre), _p(v_pre), _p(v_first), _p(mask),
              _p(k_k), _p(k_a), *[_p(o) for o in outs], min(rows, _FWD_BLOCKS))
        ctx.save_for_backward(w_pre, k, v, a_pre, v_pre, v_first, k_k, k_a, mask)
        return tuple(outs)

    @staticmethod
    def backward(ctx, d_w, d_k2, d_v2, d_a, d_b):
        w_pre, k, v, a_pre, v_pre, v_first, k_k, k_a, mask = ctx.saved_tensors
        B, T, D = k.shape
        rows = B * T
        nb = min(rows, _BWD_BLOCKS)
        gs = [_c(g) for g in (d_w, d_k2, d_v2, d_a, d_b)]
        d_wpre, d_k, d_v, d_apre = [torch.empty_like(k) for _ in range(4)]
        d_vpre = torch.empty_like(k) if v_pre is not None else None
        d_vf = torch.empty_like(k) if v_pre is not None else None
        part = torch.empty(nb, 5, D, dtype=torch.float32, device=k.device)
        _call("tmix_prepare_bwd", k, ctypes.c_long(rows), D, _p(w_pre), _p(k), _p(v), _p(a_pre), _p(v_pre), _p(v_first), _p(mask),
              _p(k_k), _p(k_a), *[_p(g) for g in gs], _p(d_wpre), _p(d_k), _p(d_v), _p(d_apre), _p(d_vpre), _p(d_vf),
              _p(part), nb)
        dp = _colsum(part, k.dtype)
        _attach_colsums(dp, d_wpre, d_apre, d_vpre)
        return d_wpre, d_k, d_v, d_apre, d_vpre, d_vf, dp[0], dp[1], None


def tmix_prepare(w_pre, k, v, a_pre, v_pre, v_first, k_k, k_a, mask, H, is_layer0):
    """returns w, k2, v2, -kk, kk*a : the scan's w, k, v, a, b operands (see module docstring)."""
    assert k.shape[-1] == H * 64
    if is_layer0:
        v_pre = v_first = None
    return _TmixPrepare.apply(w_pre, k, v, a_pre, v_pre, v_first, k_k, k_a, _mask_rows(mask, k))


class _TmixPost(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y, r, k, v, g, gn_w, gn_b, r_k, eps):
        B, T, D = y.shape
        y, r, k, v, g = _c(y), _c(r), _c(k), _c(v), _c(g)
        gn_w, gn_b, r_k = _c(gn_w.to(y.dtype)), _c(gn_b.to(y.dtype)), _c(r_k.reshape(-1).to(y.dtype))
        out = torch.empty_like(y)
        rows = B * T
        _call("tmix_post_fwd", y, ctypes.c_long(rows), D, _p(y), _p(r), _p(k), _p(v), _p(g), _p(gn_w), _p(gn_b), _p(r_k),
              ctypes.c_float(eps), _p(out), min(rows, _FWD_BLOCKS))
        ctx.save_for_backward(y, r, k, v, g, gn_w, gn_b, r_k)
        ctx.eps = eps
        return out

    @staticmethod
    def backward(ctx, dout):
        y, r, k, v, g, gn_w, gn_b, r_k = ctx.saved_tensors
        B, T, D = y.shape
        rows = B * T
        nb = min(rows, _BWD_BLOCKS)
        dout = _c(dout)
        d_y, d_r, d_k, d_v, d_g = [torch.empty_like(y) for _ in range(5)]
        part = torch.empty(nb, 3, D, dtype=torch.float32, device=y.device)
        _call("tmix_post_bwd", y, ctypes.c_long(rows), D, _p(dout), _p(y), _p(r), _p(k), _p(v), _p(g), _p(gn_w), _p(gn_b), _p(r_k),
              ctypes.c_float(ctx.eps), _p(d_y), _p(d_r), _p(d_k), _p(d_v), _p(d_g), _p(part), nb)
        dp = _colsum(part, y.dtype)
        return d_y, d_r, d_k, d_v, d_g, dp[0], dp[1], dp[2], None


def tmix_post(y, r, k, v, g, gn_weight, gn_bias, r_k, H, eps):
    """(GroupNorm_H(y) + (sum_head r*k*r_k) * v) * g ; r_k is [H,64]."""
    return _TmixPost.apply(y, r, k, v, g, gn_weight, gn_bias, r_k.reshape(-1), eps)


CHAIN_VFIRST_GRAD = True   # v_first is handed from layer to layer through the time-mix node (its gradient is summed inside the prepare backward)
CHUNKED_WKV_BWD = True   # bf16 training: scan forward + backward on the matrix cores (csrc/wkv7_chunk_*.hip); the pair goes
CHUNKED_WKV_FWD = True   # together (the backward consumes the forward's checkpoints): set BOTH False for the scalar kernels
VIA_REFERENCE_OP = False  # A/B (bench.py --via-reference-op): the scan through torch.ops.wind_backstepping.{forward,backward}, the
                          # reference's own plug-in point (wkv7_op.cpp:21-29), instead of the direct calls below


class _TmixCore(torch.autograd.Function):
    """prepare -> WKV7 scan -> GroupNorm/bonus/gate as ONE autograd node for training (zero initial state).

    Same kernels as tmix_prepare / RUN_CUDA_RWKV7g / tmix_post in forward.  In backward the node owns the whole
    gradient flow between the three stages, so (a) the scan backward can run row-split over 2 workgroups per head
    (ops.wkv7_backward_split) and (b) the sums  d_k2 = scan + post,  d_v2 = scan + post,  d_r = scan + post  and the
    two partial sets are added in fp32 inside rwkv7_tmix_prepare_bwd_sum instead of by autograd's bf16 add kernels."""

    @staticmethod
    def forward(ctx, r, w_pre, k, v, a_pre, g, v_pre, v_first, k_k, k_a, gn_w, gn_b, r_k, mask, H, eps, seq_start=None):
        from . import ops
        B, T, D = k.shape
        if T % ops.CHUNK_LEN != 0:
            raise ValueError(f"T={T} must be a multiple of {ops.CHUNK_LEN}")
        r, w_pre, k, v, a_pre, g = _c(r), _c(w_pre), _c(k), _c(v), _c(a_pre), _c(g)
        v_pre = None if v_pre is None else _c(v_pre)
        v_first = None if v_first is None else _c(v_first)
        k_k, k_a = _c(k_k.to(k.dtype)), _c(k_a.to(k.dtype))
        gn_w, gn_b, r_k = _c(gn_w.to(k.dtype)), _c(gn_b.to(k.dtype)), _c(r_k.reshape(-1).to(k.dtype))
        rows = B * T
        w, k2, v2, a_in, b_in = [torch.empty_like(k) for _ in range(5)]
        _call("tmix_prepare_fwd", k, ctypes.c_long(rows), D, _p(w_pre), _p(k), _p(v), _p(a_pre), _p(v_pre), _p(v_first),
              _p(mask), _p(k_k), _p(k_a), _p(w), _p(k2), _p(v2), _p(a_in), _p(b_in), min(rows, _FWD_BLOCKS))
        v4 = lambda t: t.view(B, T, H, 64)
        via_op = VIA_REFERENCE_OP and seq_start is None
        chunked = CHUNKED_WKV_FWD and CHUNKED_WKV_BWD and k.dtype == torch.bfloat16 and T % ops.CHUNK_T == 0 and not via_op
        if seq_start is not None and not chunked:
            raise ValueError("packed rows (seq_start) need the chunked WKV7 kernels: bf16 tensors, T % 32 == 0")
        if chunked:
            # all-MFMA pair: saves T^-1, sa and the state at the start of every 32-step chunk (half the checkpoint bytes)
            y4, tinv, sa, s = ops.wkv7_chunk_forward(v4(w), v4(r), v4(k2), v4(v2), v4(a_in), v4(b_in), seq_off=seq_start)
            y = y4.view(B, T, D)
        else:
            tinv = None
            y = torch.empty_like(k)
            s = torch.empty(B, H, T // ops.CHUNK_LEN, 64, 64, dtype=torch.float32, device=k.device)
            sa = torch.empty(B, T, H, 64, dtype=torch.float32, device=k.device)
            if via_op:   # whatever the op launches for this dtype / length; its `s` is only good for the op's own backward
                torch.ops.wind_backstepping.forward(v4(w), v4(r), v4(k2), v4(v2), v4(a_in), v4(b_in), v4(y), s, sa)
            else:
                ops.wkv7_forward_scalar(v4(w), v4(r), v4(k2), v4(v2), v4(a_in), v4(b_in), v4(y), s, sa)
        out = torch.empty_like(k)
        _call("tmix_post_fwd", k, ctypes.c_long(rows), D, _p(y), _p(r), _p(k2), _p(v2), _p(g), _p(gn_w), _p(gn_b), _p(r_k),
              ctypes.c_float(eps), _p(out), min(rows, _FWD_BLOCKS))
        ctx.save_for_backward(r, w_pre, k, v, a_pre, g, v_pre, v_first, k_k, k_a, gn_w, gn_b, r_k, mask,
                              w, k2, v2, a_in, b_in, y, s, sa, tinv)
        ctx.H, ctx.eps, ctx.chunked_fwd, ctx.seq_start, ctx.via_op = H, eps, chunked, seq_start, via_op
        ctx.set_materialize_grads(False)   # the last layer's handed-on v_first has no consumer: None, not a [B,T,D] zero tensor
        if v_first is None or not CHAIN_VFIRST_GRAD:
            return out
        # v_first is handed on to the next layer THROUGH this node (a second output aliasing the input): its gradient then arrives
        # here already summed over the later layers and is added inside rwkv7_tmix_prepare_bwd_sum -- as a plain fan-out of one
        # tensor into 23 nodes autograd forms that sum with one [B*T, D] add kernel per layer
        return out, v_first.view_as(v_first)

    @staticmethod
    def backward(ctx, dout, d_vf_next=None):
        from . import ops
        (r, w_pre, k, v, a_pre, g, v_pre, v_first, k_k, k_a, gn_w, gn_b, r_k, mask,
         w, k2, v2, a_in, b_in, y, s, sa, tinv) = ctx.saved_tensors
        B, T, D = k.shape
        H = ctx.H
        rows = B * T
        nb = min(rows, _BWD_BLOCKS)
        dout = _c(dout)
        # 1. GroupNorm / bonus / gate
        part_post = torch.empty(nb, 3, D, dtype=torch.float32, device=k.device)
        compact = COMPACT_POST_BWD
        if compact:
            # the bonus term's contributions to r, k2, v2 are rank-1 per head: one tensor (dt) + two scalars per (row, head) leave
            # this stage instead of three tensors, and stage 3 rebuilds them (three [rows, D] streams fewer per layer)
            d_y, dt_post, d_g = [torch.empty_like(k) for _ in range(3)]
            hscal = torch.empty(rows, H, 2, dtype=torch.float32, device=k.device)
            d_r_post = d_k2_post = d_v2_post = None
            _call("tmix_post_bwd_compact", k, ctypes.c_long(rows), D, _p(dout), _p(y), _p(r), _p(k2), _p(v2), _p(g), _p(gn_w),
                  _p(gn_b), _p(r_k), ctypes.c_float(ctx.eps), _p(d_y), _p(dt_post), _p(d_g), _p(hscal), _p(part_post), nb)
        else:
            d_y, d_r_post, d_k2_post, d_v2_post, d_g = [torch.empty_like(k) for _ in range(5)]
            _call("tmix_post_bwd", k, ctypes.c_long(rows), D, _p(dout), _p(y), _p(r), _p(k2), _p(v2), _p(g), _p(gn_w), _p(gn_b),
                  _p(r_k), ctypes.c_float(ctx.eps), _p(d_y), _p(d_r_post), _p(d_k2_post), _p(d_v2_post), _p(d_g),
                  _p(part_post), nb)
        # 2. scan: chunked MFMA backward (bf16, T % 32 == 0) or the scalar kernel with two workgroups per head
        if WGRAD_SYNC_BEFORE_SCAN:
            wgrad_side_sync(k.device)
        v4 = lambda t: t.view(B, T, H, 64)
        if ctx.chunked_fwd:   # the chunked backward consumes what the chunked forward saved (hs bf16, sa, tinv)
            dw, dq, dk, dv, da, db = ops.wkv7_chunk_backward(v4(w), v4(r), v4(k2), v4(v2), v4(a_in), v4(b_in), v4(d_y), s, sa,
                                                             tinv, seq_off=ctx.seq_start)
            dw2, dq2, dk2, da2, db2 = [(g, None) for g in (dw, dq, dk, da, db)]
        elif ctx.via_op:
            dw, dq, dk, dv, da, db = [torch.empty_like(k).view(B, T, H, 64) for _ in range(6)]
            torch.ops.wind_backstepping.backward(v4(w), v4(r), v4(k2), v4(v2), v4(a_in), v4(b_in), v4(d_y), s, sa, dw, dq, dk, dv, da, db)
            dw2, dq2, dk2, da2, db2 = [(g, None) for g in (dw, dq, dk, da, db)]
        else:
            dw2, dq2, dk2, dv, da2, db2 = ops.wkv7_backward_split(v4(w), v4(r), v4(k2), v4(v2), v4(a_in), v4(b_in),
                                                                  v4(d_y), s, sa)
        # 3. decay / kk / k' / value residual, summing all contributions on load
        d_wpre, d_k, d_v, d_apre, d_r = [torch.empty_like(k) for _ in range(5)]
        d_vpre = torch.empty_like(k) if v_pre is not None else None
        d_vf = torch.empty_like(k) if v_pre is not None else None
        part = torch.empty(nb, 5, D, dtype=torch.float32, device=k.device)
        d_vf_next = None if (d_vf_next is None or v_pre is None) else _c(d_vf_next)
        gsum = [dw2[0], dw2[1], dk2[0], dk2[1], d_k2_post, dv, d_v2_post, da2[0], da2[1], db2[0], db2[1],
                dq2[0], dq2[1], d_r_post, d_vf_next]
        if compact:
            gsum += [dt_post, r, r_k, hscal]
        ptrs = (ctypes.c_void_p * len(gsum))(*[None if t is None else t.data_ptr() for t in gsum])
        _call("tmix_prepare_bwd_sum_compact" if compact else "tmix_prepare_bwd_sum", k, ctypes.c_long(rows), D, _p(w_pre), _p(k), _p(v), _p(a_pre), _p(v_pre), _p(v_first),
              _p(mask), _p(k_k), _p(k_a), ptrs, _p(d_wpre), _p(d_k), _p(d_v), _p(d_apre), _p(d_vpre), _p(d_vf),
              _p(d_r), _p(part), nb)
        dp = _colsum(part, k.dtype)
        dpp = _colsum(part_post, k.dtype)
        _attach_colsums(dp, d_wpre, d_apre, d_vpre)
        return (d_r, d_wpre, d_k, d_v, d_apre, d_g, d_vpre, d_vf, dp[0], dp[1], dpp[0], dpp[1], dpp[2], None, None,
                None, None)


def tmix_core(r, w_pre, k, v, a_pre, g, v_pre, v_first, k_k, k_a, gn_weight, gn_bias, r_k, mask, H, eps, is_layer0,
              seq_start=None):
    """Training-time time-mix core: tmix_post(RUN_CUDA_RWKV7g(r, *tmix_prepare(...)), ...) as one autograd node.
    r must already be masked (r * mask) when a mask is used.  seq_start: ops.wkv7_chunk_forward's seq_off (packed rows)."""
    assert k.shape[-1] == H * 64
    if is_layer0:
        v_pre = v_first = None
    res = _TmixCore.apply(r, w_pre, k, v, a_pre, g, v_pre, v_first, k_k, k_a, gn_weight, gn_bias, r_k.reshape(-1),
                          _mask_rows(mask, k), H, eps, seq_start)
    if torch.is_tensor(res):
        return res, None
    return res   # (y, v_first for the next layer)


_ACT_ID = {None: 0, "tanh": 1, "sigmoid": 2}


def lora_decode_supported(x, rank):
    rows = x.numel() // x.shape[-1]
    return (GEMV_MAX_ROWS and x.is_cuda and x.dtype == torch.bfloat16 and rows <= GEMV_MAX_ROWS and rank in (32, 64, 128)
            and x.shape[-1] % 64 == 0 and not torch.is_grad_enabled())


def lora_decode(x, w1, w2, bias, activation):
    """act(x @ w1.T) @ w2.T + bias for at most 32 rows, one launch (rwkv7_lora32_bf16)."""
    K, N, R = x.shape[-1], w2.shape[0], w1.shape[0]
    rows = x.numel() // K
    x2 = _c(x).view(rows, K)
    b2 = None if bias is None else _c(bias)
    y = torch.empty(rows, N, dtype=torch.bfloat16, device=x.device)
    with torch.cuda.device_of(x):
        rc = _lib.lib().rwkv7_lora32_bf16(rows, N, K, R, _ACT_ID[activation], _p(x2), _p(_c(w1)), _p(_c(w2)), _p(b2), _p(y),
                                          _stream(x))
    _lib.check(rc, "lora32")
    return y.view(*x.shape[:-1], N)


# ------------------------------------------------------------------------------------------------------
# nn.Linear with a split-M weight gradient
# ------------------------------------------------------------------------------------------------------
WGRAD_MIN_ROWS = 4096
WGRAD_SLABS_SMALL, WGRAD_SLABS_BIG = 8, 4   # row slabs of the batched weight-gradient GEMM: outputs up to 1024 x 1024 / larger
COMPACT_POST_BWD = True   # tmix_post's backward hands dt + (dot, ds) per head to tmix_prepare's backward instead of d_r, d_k2, d_v2
                          # (A/B switch for tools/ab_step.py)
SKINNY_WGRAD = True   # low-rank weight gradients through rwkv7_wgrad_skinny_bf16 (A/B switch for tools/ab_step.py)


# Round 6 (late): the weight gradient of [W_a ; W_b] (N = 2 R = 576 / 512) on an own kernel instead of 32 library slabs (75 us per layer on a
# 256 x 192 tile kernel).  RWKV7_MID_WGRAD=0: the library slabs.
MID_WGRAD = os.environ.get("RWKV7_MID_WGRAD", "1") == "1"
MID_WGRAD_SLABS = 32
MID_WGRAD_HITS = [0]


def wgrad_splitk(dy2, x2, out=None, slabs=None):
    """dy2[M,N]^T @ x2[M,K] -> [N,K].  The reduction runs over M = B*T rows (32768 at BASELINE configs[1]) while the
    result is at most a few 256x256 tiles, so one BLAS call leaves most CUs idle (measured on MI355X, tools/
    bench_wgrad_splitk.py: 1024x1024 209 us, 64x1024 115 us).  Batched over S slabs of rows with fp32 partials,
    then one reduction (rwkv7_sum_slabs_bf16): 82 us and 31 us.  `out`: a contiguous bf16 [N,K] tensor to write into
    (the parameter's slice of the trainer's flat gradient buffer), else a new tensor."""
    M, N = dy2.shape
    K = x2.shape[1]
    rank, wide = min(N, K), max(N, K)
    if (SKINNY_WGRAD and dy2.dtype == torch.bfloat16 and x2.dtype == torch.bfloat16 and rank in (32, 64, 128) and wide % 256 == 0
            and M >= WGRAD_MIN_ROWS and M % 512 == 0):
        # low-rank projections: HBM-bound (the [M, wide] operand is streamed once), rwkv7_wgrad_skinny_bf16 + one reduction
        S = M // 512
        part = torch.empty(S, N, K, dtype=torch.float32, device=dy2.device)
        if out is None:
            out = torch.empty(N, K, dtype=torch.bfloat16, device=dy2.device)
        with torch.cuda.device_of(part):
            rc = _lib.lib().rwkv7_wgrad_skinny_bf16(ctypes.c_long(M), N, K, S, _p(dy2), _p(x2), _p(part), _stream(part))
            _lib.check(rc, "wgrad_skinny")
            rc = _lib.lib().rwkv7_sum_slabs_bf16(ctypes.c_long(N * K), S, _p(part), _p(out), 0, _stream(part))
        _lib.check(rc, "sum_slabs")
        return out
    if (MID_WGRAD and dy2.dtype == torch.bfloat16 and x2.dtype == torch.bfloat16 and N in (512, 576) and K % 256 == 0 and M >= WGRAD_MIN_ROWS
            and M % (MID_WGRAD_SLABS * 64) == 0 and dy2.is_contiguous() and x2.is_contiguous()):
        # the [W_a ; W_b] gradient of fused.mix_lora: rwkv7_wgrad_mid_bf16 (csrc/wgrad_skinny.hip) + one reduction
        S = MID_WGRAD_SLABS
        part = torch.empty(S, N, K, dtype=torch.float32, device=dy2.device)
        if out is None:
            out = torch.empty(N, K, dtype=torch.bfloat16, device=dy2.device)
        with torch.cuda.device_of(part):
            rc = _lib.lib().rwkv7_wgrad_mid_bf16(ctypes.c_long(M), N, K, S, _p(dy2), _p(x2), _p(part), _stream(part))
            _lib.check(rc, "wgrad_mid")
            rc = _lib.lib().rwkv7_sum_slabs_bf16(ctypes.c_long(N * K), S, _p(part), _p(out), 0, _stream(part))
        _lib.check(rc, "sum_slabs")
        MID_WGRAD_HITS[0] += 1
        return out
    S = slabs if slabs else (WGRAD_SLABS_SMALL if N * K <= 1024 * 1024 else WGRAD_SLABS_BIG)
    if M < WGRAD_MIN_ROWS or M % (S * 8) != 0:
        res = torch.mm(dy2.t(), x2)
        if out is not None:
            out.copy_(res)
            return out
        return res
    part = torch.bmm(dy2.view(S, M // S, N).transpose(1, 2), x2.view(S, M // S, K), out_dtype=torch.float32)
    if dy2.dtype != torch.bfloat16 or (N * K) % 4 != 0:
        res = part.sum(0).to(dy2.dtype)
        if out is not None:
            out.copy_(res)
            return out
        return res
    if out is None:
        out = torch.empty(N, K, dtype=torch.bfloat16, device=dy2.device)
    with torch.cuda.device_of(part):
        rc = _lib.lib().rwkv7_sum_slabs_bf16(ctypes.c_long(N * K), S, _p(part), _p(out), 0, _stream(part))
    _lib.check(rc, "sum_slabs")
    return out


GATHER_ROWS_KERNEL = os.environ.get("RWKV7_GATHER_ROWS_KERNEL", "1") == "1"   # A/B switch: 0 = torch index_select + mask multiply


class _GatherRows(torch.autograd.Function):
    """out[r] = src[idx[r]] (zeros where idx[r] < 0); inv[j] = the output row that took source row j, or -1: the gradient is the same
    gather the other way (rwkv7_gather_rows_bf16).  src [n_in, D] bf16, idx int32 [n_out], inv int32 [n_in]."""

    @staticmethod
    def forward(ctx, src, idx, inv):
        src = _c(src)
        out = torch.empty(idx.numel(), src.shape[1], dtype=src.dtype, device=src.device)
        with torch.cuda.device_of(src):
            rc = _lib.lib().rwkv7_gather_rows_bf16(ctypes.c_long(idx.numel()), src.shape[1], _p(src), _p(idx), _p(out), _stream(src))
        _lib.check(rc, "gather_rows")
        ctx.save_for_backward(idx, inv)
        return out

    @staticmethod
    def backward(ctx, g):
        idx, inv = ctx.saved_tensors
        g = _c(g)
        dsrc = torch.empty(inv.numel(), g.shape[1], dtype=g.dtype, device=g.device)
        with torch.cuda.device_of(g):
            rc = _lib.lib().rwkv7_gather_rows_bf16(ctypes.c_long(inv.numel()), g.shape[1], _p(g), _p(inv), _p(dsrc), _stream(g))
        _lib.check(rc, "gather_rows(bwd)")
        return dsrc, None, None


def gather_rows(src, idx, inv):
    """Rows of `src` [n_in, D] re-ordered by the injective map idx (int32 [n_out], -1 = a zero row), with its inverse `inv` (int32 [n_in],
    -1 = the row is dropped) for the backward.  bf16 on the HIP device, D % 8 == 0; anything else takes torch's index ops."""
    if (GATHER_ROWS_KERNEL and src.is_cuda and src.dtype == torch.bfloat16 and src.shape[1] % 8 == 0 and idx.dtype == torch.int32
            and inv.dtype == torch.int32):
        return _GatherRows.apply(src, idx, inv)
    out = src.index_select(0, idx.clamp(min=0).long())
    return out * (idx >= 0).unsqueeze(-1).to(out.dtype)


# Weight gradients on a second stream.  A layer's dW = dy^T x is needed by nobody until the optimizer (or the bucket's collective):
# queued on the stream that runs backward it sits between the input-gradient GEMM and the next fused stage and both wait for it.
# On a side stream it runs beside the stages that follow -- HBM-bound kernels that leave the matrix cores idle and co-reside with
# a GEMM's workgroups.  Order: the side stream waits for the backward stream at the point of the call (dy and x exist), the
# tensors are marked as in use on the side stream (the allocator keeps them until it has passed), and the backward stream waits
# for the side stream before anything reads a gradient slice: wgrad_side_sync() -- trainer.FlatBuffers.finish_backward, the
# bucket launch of BucketedAllReduce, and a second use of a parameter in the same pass.  Only gradients written straight into the
# trainer's flat buffer go this way (a gradient handed back to autograd as a fresh tensor may be consumed at once).
# Off by default: the step is 2 ms (1.5 %) shorter with it, but every kernel that shares the GPU with the side stream measures
# longer (the WKV7 group ~5 %, small side-stream kernels several times), which blurs per-kernel profiles and the roofline figures
# of bench.py; RWKV7_WGRAD_SIDE_STREAM=1 / bench.py --wgrad-side-stream turn it on.
WGRAD_SIDE_STREAM = os.environ.get("RWKV7_WGRAD_SIDE_STREAM", "0") == "1"
WGRAD_SYNC_BEFORE_SCAN = False   # drain the side stream before the WKV7 backward kernels: +1.4 ms per step (A/B) -- most of the
                                 # overlap IS with those kernels (latency-bound, matrix cores and HBM mostly idle)
_SIDE = {}   # device index -> [stream, work pending]


def wgrad_side_sync(device=None):
    """Make the current stream wait for the weight gradients queued on the side stream (no-op when none are pending)."""
    for idx, ent in _SIDE.items():
        if ent[1] and (device is None or device.index == idx):
            torch.cuda.current_stream(ent[0].device).wait_stream(ent[0])
            ent[1] = False


def _wgrad(d2, x2, param):
    """dW for `param` from dy [M, N] and x [M, K]: into the trainer's slice on the side stream when there is one, else as a tensor."""
    slot = _grad_slot(param)
    if slot is None:
        wgrad_side_sync(d2.device)   # (a second use of a parameter: its first gradient may still be in flight)
        return wgrad_splitk(d2, x2)
    if not WGRAD_SIDE_STREAM:
        wgrad_splitk(d2, x2, out=slot)
        return slot.view_as(param)   # a fresh view object: autograd adopts it as .grad without a copy
    ent = _SIDE.get(d2.device.index)
    if ent is None:
        ent = _SIDE[d2.device.index] = [torch.cuda.Stream(d2.device), False]
    side = ent[0]
    side.wait_stream(torch.cuda.current_stream(d2.device))
    with torch.cuda.stream(side):
        wgrad_splitk(d2, x2, out=slot)
    d2.record_stream(side)
    x2.record_stream(side)
    ent[1] = True
    return slot.view_as(param)


def _grad_slot(param):
    """The trainer's slice of the flat gradient buffer for `param`, if this backward pass may write the gradient there
    directly (trainer.FlatBuffers hands the slices out; first writer per backward pass only -- a second use of the same
    parameter goes through a fresh tensor and autograd's own accumulation)."""
    slot = getattr(param, "_grad_slot", None)
    if slot is None or getattr(param, "_grad_slot_used", True) or param.grad is not None:
        return None
    param._grad_slot_used = True
    return slot


def _attach_colsums(dp, d_wpre, d_apre, d_vpre):
    """The prepare backward already walks every row of d_wpre / d_apre / d_vpre; it leaves their column sums in rows 2..4
    of its partials.  They are the bias gradients of the low-rank branches whose Linear produced w_pre / a_pre / v_pre, so
    they ride along on the gradient tensors (`_colsum`) and _Linear.backward takes them instead of launching a reduction
    over B*T rows.  If autograd hands _Linear a different tensor object (the gradient was accumulated with another one,
    hooks, ...), the attribute is simply absent and the reduction runs."""
    d_wpre._colsum = dp[2]
    d_apre._colsum = dp[3]
    if d_vpre is not None:
        d_vpre._colsum = dp[4]


COLSUM_HITS = [0]   # how often _Linear.backward found a ready column sum (tests)


class _Linear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        ctx.wparam = weight
        return torch.nn.functional.linear(x, weight, bias)

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        K = x.shape[-1]
        dy2 = _c(dy).view(-1, dy.shape[-1])
        x2 = _c(x).view(-1, K)
        dx = _dgrad(dy2, weight).view(x.shape) if ctx.needs_input_grad[0] else None
        dw = None
        if ctx.needs_input_grad[1]:   # (queued before the input-gradient GEMM instead: +0.8 ms per step, same-box A/B)
            dw = _wgrad(dy2, x2, ctx.wparam)
        db = None
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = getattr(dy, "_colsum", None)
            if db is not None and db.shape[0] == dy2.shape[1] and db.dtype == dy2.dtype:
                COLSUM_HITS[0] += 1
            else:
                db = dy2.sum(0)
        return dx, dw, db


class _DualLinear(torch.autograd.Function):
    """(x W_a^T, x W_b^T) for two bias-free Linears that read the SAME input (the value projection and the down projection of the
    value-residual LoRA both read xv, rwkv_s2s_single_ffn.py:174,180): as two autograd nodes the input gradient is two GEMMs plus
    an [M, D] add kernel; here the second GEMM accumulates into the first one's output (beta = 1)."""

    @staticmethod
    def forward(ctx, x, wa, wb):
        ctx.save_for_backward(x, wa, wb)
        ctx.params = (wa, wb)
        ctx.set_materialize_grads(False)   # an unused output arrives as None (handled below), not as a zero tensor
        return torch.nn.functional.linear(x, wa), torch.nn.functional.linear(x, wb)

    @staticmethod
    def backward(ctx, dya, dyb):
        x, wa, wb = ctx.saved_tensors
        K = x.shape[-1]
        x2 = _c(x).view(-1, K)
        da2 = None if dya is None else _c(dya).view(-1, wa.shape[0])
        db2 = None if dyb is None else _c(dyb).view(-1, wb.shape[0])
        dx = None
        if ctx.needs_input_grad[0]:
            if da2 is not None and db2 is not None:
                dx = _dgrad(da2, wa)
                dx.addmm_(db2, wb)
            elif da2 is not None:
                dx = _dgrad(da2, wa)
            elif db2 is not None:
                dx = torch.mm(db2, wb)
            dx = None if dx is None else dx.view(x.shape)
        grads = []
        for i, (d2, w) in enumerate(((da2, wa), (db2, wb))):
            g = None
            if d2 is not None and ctx.needs_input_grad[1 + i]:
                g = _wgrad(d2, x2, ctx.params[i])
            grads.append(g)
        return dx, grads[0], grads[1]


def dual_linear_supported(x, wa, wb):
    return (x.is_cuda and x.dtype == torch.bfloat16 and torch.is_grad_enabled() and wa.requires_grad and wb.requires_grad
            and x.numel() // x.shape[-1] >= WGRAD_MIN_ROWS)


def dual_linear(x, wa, wb):
    return _DualLinear.apply(x, wa, wb)


GEMV_MAX_ROWS = 32   # decode batches: rwkv7_gemv32_bf16 instead of the BLAS library


def linear(x, weight, bias=None):
    """F.linear with two specialisations for bf16 HIP tensors: with many rows and autograd on, the backward computes the
    weight gradient with wgrad_splitk; with at most 32 rows and autograd off (decode steps), the product runs on the
    weight-streaming kernel rwkv7_gemv32_bf16.  Anything else goes to F.linear unchanged."""
    rows = x.numel() // x.shape[-1]
    if (GEMV_MAX_ROWS and x.is_cuda and x.dtype == torch.bfloat16 and rows <= GEMV_MAX_ROWS and x.shape[-1] % 64 == 0
            and not (torch.is_grad_enabled() and (x.requires_grad or weight.requires_grad)) and weight.dtype == torch.bfloat16):
        K, N = x.shape[-1], weight.shape[0]
        x2, w2 = _c(x).view(rows, K), _c(weight)
        b2 = None if bias is None else _c(bias.to(torch.bfloat16))
        y = torch.empty(rows, N, dtype=torch.bfloat16, device=x.device)
        with torch.cuda.device_of(x):
            rc = _lib.lib().rwkv7_gemv32_bf16(rows, N, K, _p(x2), _p(w2), _p(b2), _p(y), _stream(x))
        _lib.check(rc, "gemv32")
        return y.view(*x.shape[:-1], N)
    if (x.is_cuda and x.dtype == torch.bfloat16 and torch.is_grad_enabled() and weight.requires_grad
            and x.numel() // x.shape[-1] >= WGRAD_MIN_ROWS):
        return _Linear.apply(x, weight, bias)
    return torch.nn.functional.linear(x, weight, bias)


# Round 4: a projection with the residual add that follows it as the GEMM's epilogue (csrc/gemm_nt4.hip epilogue 4): x1 = x + y W^T leaves
# the GEMM, so the add + LayerNorm (+ lerp) stage behind it reads one [rows, D] stream less and does not write x1.  Used for the output
# projection of the time-mix block (K = D = 1024: the own GEMM is 5 us behind the library there, the stage saves two streams); the
# channel-mix value projection (K = 4096) stays with the library, where the own kernel is 25 us behind.  RWKV7_FUSED_OPROJ_ADD=0: off.
FUSED_OPROJ_ADD = os.environ.get("RWKV7_FUSED_OPROJ_ADD", "1") == "1"


def linear_add_eligible(y, weight, resid):
    M = y.numel() // y.shape[-1]
    return (FUSED_OPROJ_ADD and y.is_cuda and y.dtype == torch.bfloat16 and weight.dtype == torch.bfloat16 and resid.dtype == torch.bfloat16
            and M % 256 == 0 and weight.shape[0] % 256 == 0 and weight.shape[1] % 1024 == 0 and resid.shape == (*y.shape[:-1], weight.shape[0]))


class _LinearAdd(torch.autograd.Function):
    """x1 = resid + y @ weight^T  (bf16(bf16(y W^T) + resid): the bits of nn.Linear followed by the add)."""

    @staticmethod
    def forward(ctx, y, weight, resid):
        y2 = _c(y).view(-1, y.shape[-1])
        w = _c(weight)
        r2 = _c(resid).view(-1, resid.shape[-1])
        M, K = y2.shape
        N = w.shape[0]
        out = torch.empty(M, N, dtype=y.dtype, device=y.device)
        with torch.cuda.device_of(y):
            rc = _lib.lib().rwkv7_gemm_nt_add_bf16(M, N, K, _p(y2), _p(w), _p(r2), _p(out), _stream(y))
        _lib.check(rc, "gemm_nt_add")
        ctx.save_for_backward(y2, weight)
        ctx.wparam = weight
        ctx.yshape = y.shape
        return out.view(resid.shape)

    @staticmethod
    def backward(ctx, d1):
        y2, weight = ctx.saved_tensors
        d2 = _c(d1).view(-1, d1.shape[-1])
        dy = _dgrad(d2, weight).view(ctx.yshape) if ctx.needs_input_grad[0] else None
        dw = _wgrad(d2, y2, ctx.wparam) if ctx.needs_input_grad[1] else None
        return dy, dw, (d1 if ctx.needs_input_grad[2] else None)


def linear_add(y, weight, resid):
    """resid + y @ weight^T as one GEMM, or None (shapes / dtype outside the own GEMM's range: the caller adds)."""
    if linear_add_eligible(y, weight, resid):
        return _LinearAdd.apply(y, weight, resid)
    return None


# ------------------------------------------------------------------------------------------------------
# residual add + LayerNorm
# ------------------------------------------------------------------------------------------------------
class _AddLN(torch.autograd.Function):
    """(x1, h) = (x + branch, LayerNorm(x + branch)); with branch None: h = LayerNorm(x) only.
    backward folds the gradient arriving at x1 through the residual path into the LayerNorm backward kernel."""

    @staticmethod
    def forward(ctx, x, branch, gamma, beta, eps):
        D = x.shape[-1]
        x = _c(x)
        rows = x.numel() // D
        gamma_c = _c(gamma.to(x.dtype))
        beta_c = None if beta is None else _c(beta.to(x.dtype))
        h = torch.empty_like(x)
        mean = torch.empty(rows, dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        if branch is not None:
            branch = _c(branch)
            x1 = torch.empty_like(x)
        else:
            x1 = None
        _call("add_ln_fwd", x, ctypes.c_long(rows), D, _p(x), _p(branch), _p(gamma_c), _p(beta_c), ctypes.c_float(eps),
              _p(x1), _p(h), _p(mean), _p(rstd), min(rows, _FWD_BLOCKS))
        ctx.has_branch, ctx.has_beta = branch is not None, beta is not None
        ctx.save_for_backward(x1 if branch is not None else x, mean, rstd, gamma_c)
        if branch is None:
            return h
        return x1, h

    @staticmethod
    def backward(ctx, *grads):
        xs, mean, rstd, gamma = ctx.saved_tensors
        D = xs.shape[-1]
        rows = xs.numel() // D
        if ctx.has_branch:
            d_x1, dh = grads
        else:
            d_x1, dh = None, grads[0]
        if dh is None:  # h unused downstream: only the residual path carries gradient
            dx = d_x1
            return dx, (dx if ctx.has_branch else None), None, None, None
        dh = _c(dh)
        d_x1 = None if d_x1 is None else _c(d_x1)
        nb = min(rows, _BWD_BLOCKS)
        dx = torch.empty_like(xs)
        part = torch.empty(nb, 2, D, dtype=torch.float32, device=xs.device)
        _call("add_ln_bwd", xs, ctypes.c_long(rows), D, _p(dh), _p(d_x1), _p(xs), _p(mean), _p(rstd), _p(gamma), _p(dx),
              _p(part), nb)
        dp = _colsum(part, xs.dtype)
        return dx, (dx if ctx.has_branch else None), dp[0], (dp[1] if ctx.has_beta else None), None


_ADD_LN_MIX_RUN = 4       # rows per run of the fused add + LayerNorm + token-shift kernels (one neighbour row recomputed per run)
_ADD_LN_MIX_BLOCKS = 8192
_ADD_LN_MIX_BWD_BLOCKS = 2048   # also the number of parameter-gradient partials


class _AddLNMix(torch.autograd.Function):
    """(x1, out_0 .. out_{n-1}): x1 = x + branch (or x), h = LayerNorm(x1), out_i = token-shift lerp i of h * mask -- _AddLN
    followed by _Mix without h (forward) and dh (backward) ever touching HBM (rwkv7_add_ln_mix_fwd / rwkv7_mix_add_ln_bwd).
    Training path only: no carried token-shift state."""

    @staticmethod
    def forward(ctx, x, branch, gamma, beta, eps, mask, params):
        B, T, D = x.shape
        x, params = _c(x), _c(params)
        nmix = params.shape[0]
        gamma_c = _c(gamma.to(x.dtype))
        beta_c = None if beta is None else _c(beta.to(x.dtype))
        rows = B * T
        out = torch.empty(nmix, B, T, D, dtype=x.dtype, device=x.device)
        mean = torch.empty(rows, dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        if branch is not None:
            branch = _c(branch)
            x1 = torch.empty_like(x)
        else:
            x1 = None
        nb = max(1, min(-(-rows // _ADD_LN_MIX_RUN), _ADD_LN_MIX_BLOCKS))
        _call("add_ln_mix_fwd", x, B, T, D, nmix, _p(x), _p(branch), _p(gamma_c), _p(beta_c), ctypes.c_float(eps), _p(mask),
              _p(params), _p(x1), _p(out), _p(mean), _p(rstd), nb, _ADD_LN_MIX_RUN)
        ctx.has_branch, ctx.has_beta = branch is not None, beta is not None
        ctx.save_for_backward(x1 if branch is not None else x, mean, rstd, gamma_c, beta_c, mask, params)
        return (x1 if branch is not None else x,) + tuple(out[i] for i in range(nmix))

    @staticmethod
    def backward(ctx, d_x1, *gs):
        xs, mean, rstd, gamma, beta, mask, params = ctx.saved_tensors
        B, T, D = xs.shape
        nmix = params.shape[0]
        gs = [torch.zeros_like(xs) if g is None else _c(g) for g in gs]
        d_x1 = None if d_x1 is None else _c(d_x1)
        rows = B * T
        nb = max(1, min(-(-rows // _ADD_LN_MIX_RUN), _ADD_LN_MIX_BWD_BLOCKS))
        dx = torch.empty_like(xs)
        part = torch.empty(nb, nmix + 2, D, dtype=torch.float32, device=xs.device)
        ptrs = (ctypes.c_void_p * nmix)(*[g.data_ptr() for g in gs])
        _call("mix_add_ln_bwd", xs, B, T, D, nmix, ptrs, _p(d_x1), _p(xs), _p(mean), _p(rstd), _p(gamma), _p(beta), _p(mask),
              _p(params), _p(dx), _p(part), nb, _ADD_LN_MIX_RUN)
        dp = _colsum(part, xs.dtype) if params.dtype == xs.dtype else part.sum(0)
        return (dx, (dx if ctx.has_branch else None), dp[nmix].to(xs.dtype), (dp[nmix + 1].to(xs.dtype) if ctx.has_beta else None),
                None, None, dp[:nmix].to(params.dtype))


class _AddLNMixFwd(torch.autograd.Function):
    """Forward of _AddLNMix (one pass: residual add + LayerNorm + lerps, rwkv7_add_ln_mix_fwd_h, which also stores h), backward as
    the two SEPARATE kernels (rwkv7_mix_bwd, rwkv7_add_ln_bwd).  For the time-mix side: its one-pass backward spills (backbone.py,
    FUSED_ADD_LN_MIX6), its one-pass forward does not."""

    @staticmethod
    def forward(ctx, x, branch, gamma, beta, eps, mask, params):
        B, T, D = x.shape
        x, params = _c(x), _c(params)
        nmix = params.shape[0]
        gamma_c = _c(gamma.to(x.dtype))
        beta_c = None if beta is None else _c(beta.to(x.dtype))
        rows = B * T
        out = torch.empty(nmix, B, T, D, dtype=x.dtype, device=x.device)
        h = torch.empty_like(x)
        mean = torch.empty(rows, dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        if branch is not None:
            branch = _c(branch)
            x1 = torch.empty_like(x)
        else:
            x1 = None
        nb = max(1, min(-(-rows // _ADD_LN_MIX_RUN), _ADD_LN_MIX_BLOCKS))
        _call("add_ln_mix_fwd_h", x, B, T, D, nmix, _p(x), _p(branch), _p(gamma_c), _p(beta_c), ctypes.c_float(eps), _p(mask),
              _p(params), _p(x1), _p(out), _p(h), _p(mean), _p(rstd), nb, _ADD_LN_MIX_RUN)
        ctx.has_branch, ctx.has_beta = branch is not None, beta is not None
        ctx.save_for_backward(x1 if branch is not None else x, h, mean, rstd, gamma_c, mask, params)
        return (x1 if branch is not None else x,) + tuple(out[i] for i in range(nmix))

    @staticmethod
    def backward(ctx, d_x1, *gs):
        xs, h, mean, rstd, gamma, mask, params = ctx.saved_tensors
        B, T, D = xs.shape
        nmix = params.shape[0]
        rows = B * T
        gs = [torch.zeros_like(xs) if g is None else _c(g) for g in gs]
        nb = max(1, min(-(-rows // _MIX_BWD_ROWS), _MIX_BWD_BLOCKS))
        dh = torch.empty_like(xs)
        part_m = torch.empty(nb, nmix, D, dtype=torch.float32, device=xs.device)
        ptrs = (ctypes.c_void_p * nmix)(*[g.data_ptr() for g in gs])
        _call("mix_bwd", xs, B, T, D, nmix, ptrs, _p(h), _p(None), _p(mask), _p(params), _p(dh), _p(part_m), nb, _MIX_BWD_ROWS)
        d_x1 = None if d_x1 is None else _c(d_x1)
        nb2 = min(rows, _BWD_BLOCKS)
        dx = torch.empty_like(xs)
        part = torch.empty(nb2, 2, D, dtype=torch.float32, device=xs.device)
        _call("add_ln_bwd", xs, ctypes.c_long(rows), D, _p(dh), _p(d_x1), _p(xs), _p(mean), _p(rstd), _p(gamma), _p(dx), _p(part), nb2)
        dp = _colsum(part, xs.dtype)
        return (dx, (dx if ctx.has_branch else None), dp[0], (dp[1] if ctx.has_beta else None),
                None, None, _colsum(part_m, params.dtype))


# measured (round 4, same-box A/B, tools/ab_step.py): 126.25 -> 126.83 ms per step, +0.58 ms -- the one-pass kernel re-normalises a
# neighbour row per run of four and holds two workgroup barriers per row; with five output streams that costs more than the one
# [rows, D] read it saves.  Off; bit-identical to the two stages (tests/test_fused_gpu.py), kept as the recorded experiment.
FUSED_ADD_LN_MIX_LORA_FWD = os.environ.get("RWKV7_FUSED_ADD_LN_MIX_LORA_FWD", "0") == "1"


class _AddLNMixLora(torch.autograd.Function):
    """(x1, x_r, x_k, x_v, G): _AddLN followed by _MixLora with ONE forward kernel -- residual add + LayerNorm + the three lerps that
    feed full projections, h = LayerNorm(x1) stored because the branches' GEMM G = h wcat^T (and the backward) read it anyway
    (rwkv7_add_ln_mix_fwd_h, nmix = 3).  The backward is the separate stages' own: mix_bwd -> + dG wcat -> add_ln_bwd."""

    @staticmethod
    def forward(ctx, x, branch, gamma, beta, eps, mask, params, wcat):
        B, T, D = x.shape
        x, params, wcat = _c(x), _c(params), _c(wcat)
        nmix = params.shape[0]
        gamma_c = _c(gamma.to(x.dtype))
        beta_c = None if beta is None else _c(beta.to(x.dtype))
        rows = B * T
        out = torch.empty(nmix, B, T, D, dtype=x.dtype, device=x.device)
        h = torch.empty_like(x)
        mean = torch.empty(rows, dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        if branch is not None:
            branch = _c(branch)
            x1 = torch.empty_like(x)
        else:
            x1 = None
        nb = max(1, min(-(-rows // _ADD_LN_MIX_RUN), _ADD_LN_MIX_BLOCKS))
        _call("add_ln_mix_fwd_h", x, B, T, D, nmix, _p(x), _p(branch), _p(gamma_c), _p(beta_c), ctypes.c_float(eps), _p(mask),
              _p(params), _p(x1), _p(out), _p(h), _p(mean), _p(rstd), nb, _ADD_LN_MIX_RUN)
        G = torch.mm(h.view(-1, D), wcat.t())
        ctx.has_branch, ctx.has_beta = branch is not None, beta is not None
        ctx.save_for_backward(x1 if branch is not None else x, h, mean, rstd, gamma_c, mask, params, wcat)
        FUSED_MIX_LORA_HITS[0] += 1
        return ((x1 if branch is not None else x), *[out[i] for i in range(nmix)], G.view(B, T, -1))

    @staticmethod
    def backward(ctx, d_x1, *gs):
        xs, h, mean, rstd, gamma, mask, params, wcat = ctx.saved_tensors
        B, T, D = xs.shape
        nmix = params.shape[0]
        rows = B * T
        dG = gs[nmix]
        gs = [torch.zeros_like(xs) if g is None else _c(g) for g in gs[:nmix]]
        nb = max(1, min(-(-rows // _MIX_BWD_ROWS), _MIX_BWD_BLOCKS))
        dh = torch.empty_like(xs)
        part_m = torch.empty(nb, nmix, D, dtype=torch.float32, device=xs.device)
        ptrs = (ctypes.c_void_p * nmix)(*[g.data_ptr() for g in gs])
        _call("mix_bwd", xs, B, T, D, nmix, ptrs, _p(h), _p(None), _p(mask), _p(params), _p(dh), _p(part_m), nb, _MIX_BWD_ROWS)
        dwcat = None
        if dG is not None:
            dG2 = _c(dG).view(-1, dG.shape[-1])
            dh.view(-1, D).addmm_(dG2, wcat)
            dwcat = wgrad_splitk(dG2, h.view(-1, D), slabs=WGRAD_SLABS_WCAT)
        d_x1 = None if d_x1 is None else _c(d_x1)
        nb2 = min(rows, _BWD_BLOCKS)
        dx = torch.empty_like(xs)
        part = torch.empty(nb2, 2, D, dtype=torch.float32, device=xs.device)
        _call("add_ln_bwd", xs, ctypes.c_long(rows), D, _p(dh), _p(d_x1), _p(xs), _p(mean), _p(rstd), _p(gamma), _p(dx), _p(part), nb2)
        dp = _colsum(part, xs.dtype)
        return (dx, (dx if ctx.has_branch else None), dp[0], (dp[1] if ctx.has_beta else None), None, None,
                _colsum(part_m, params.dtype), dwcat)


def add_layer_norm_mix_lora(x, branch, norm, mask, x_r, x_k, x_v, mus, w1s, acts):
    """fused.add_layer_norm + fused.mix_lora with one forward kernel.  Returns (x + branch, x_r, x_k, x_v, [a_i])."""
    D = x.shape[-1]
    params = torch.cat([p.reshape(1, D) for p in (x_r, x_k, x_v)], 0).to(x.dtype)
    wcat = _WcatBuild.apply(len(w1s), *w1s, *mus)
    mr = _mask_rows(mask, x)
    x1, xr, xk, xv, G = _AddLNMixLora.apply(x, branch, norm.weight, norm.bias, norm.eps, mr, params, wcat)
    hs = _CombineAct.apply(G, mr, tuple(w.shape[0] for w in w1s), tuple(_ACT_CODE[a] for a in acts))
    return x1, xr, xk, xv, list(hs)


def add_ln_mix_supported(x, state):
    return x.is_cuda and state is None and x.dim() == 3 and x.dtype in (torch.bfloat16, torch.float32) and torch.is_grad_enabled()


def add_layer_norm_mix(x, branch, norm, mask, mix_params, fwd_only=False):
    """(x + branch, [token-shift lerps of norm(x + branch) * mask]); branch may be None.  mix_params: the lerp coefficient
    vectors (6 for the time-mix block, 1 for the channel-mix block).  fwd_only: one-pass forward, the backward as the two separate
    kernels (_AddLNMixFwd)."""
    D = x.shape[-1]
    params = torch.cat([p.reshape(1, D) for p in mix_params], 0).to(x.dtype) if len(mix_params) > 1 else mix_params[0].reshape(1, D).to(x.dtype)
    node = _AddLNMixFwd if (fwd_only and len(mix_params) > 1) else _AddLNMix
    res = node.apply(x, branch, norm.weight, norm.bias, norm.eps, _mask_rows(mask, x), params)
    return res[0], res[1:]


def layer_norm(x, norm):
    """norm: nn.LayerNorm over the last dim."""
    return _AddLN.apply(x, None, norm.weight, norm.bias, norm.eps)


def add_layer_norm(x, branch, norm):
    """returns (x + branch, norm(x + branch))."""
    return _AddLN.apply(x, branch, norm.weight, norm.bias, norm.eps)
