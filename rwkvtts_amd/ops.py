"""Host-side mirror of the reference's WKV7 operator interface, backed by librwkv7_hip.so.

Same names, argument order and error behaviour as the reference so that its call sites work unchanged:

    torch.ops.wind_backstepping.forward / .backward      model/llm/cuda/wkv7_op.cpp:21-29
    torch.ops.rwkv7_state_fwd_fp16.forward               model/llm/cuda/rwkv7_state_fwd_fp16.cpp:8-14
    torch.ops.wkv7s.forward                              model/llm/cuda/wkv7s_op.cpp:9-15
    WindBackstepping, RUN_CUDA_RWKV7g                    model/llm/rwkv_s2s_single_ffn.py:15-40
    WKV_7 / RWKV7_OP, WKV_7_batch / RWKV7_BATCH_OP       model/llm/rwkv_asr_cuda_whisper.py:50-81

The ops are registered for the CUDA dispatch key (which is HIP on ROCm) only -- exactly like the
reference (wkv7_op.cpp:26-29) -- so CPU tensors raise NotImplementedError instead of silently
running somewhere else.  PyTorch is plumbing here: it owns the device memory and the stream.

Extension over the reference: fp32 tensors are accepted everywhere bf16 is (routed to the *_f32
C entry points); that is what the fp32 logit-parity tests use.
"""
import ctypes

import torch

from . import _lib

HEAD_SIZE = 64
CHUNK_LEN = 16
DTYPE = torch.bfloat16


# bench.py sets this to a dict to time individual launches with HIP events recorded on the launch stream
# (torch's current stream IS the stream handed to the C ABI): name -> [(start_event, end_event), ...]
KERNEL_TIMERS = None


class _timed:
    def __init__(self, name, ref):
        self.name, self.on = name, KERNEL_TIMERS is not None
        if self.on:
            st = torch.cuda.current_stream(ref.device)
            self.s, self.e, self.st = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True), st

    def __enter__(self):
        if self.on:
            self.s.record(self.st)

    def __exit__(self, *a):
        if self.on:
            self.e.record(self.st)
            KERNEL_TIMERS.setdefault(self.name, []).append((self.s, self.e))


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream(t):
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _sfx(ts, what):
    dt = ts[0].dtype
    if dt not in (torch.bfloat16, torch.float32):
        raise TypeError(f"{what}: tensors must be bfloat16 (reference) or float32, got {dt}")
    for t in ts:
        if t.dtype != dt:
            raise TypeError(f"{what}: mixed dtypes {dt} / {t.dtype}")
        if not t.is_contiguous():
            raise ValueError(f"{what}: tensors must be contiguous (rwkv_s2s_single_ffn.py:21)")
        if not t.is_cuda:
            raise NotImplementedError(f"{what}: HIP device tensors only (no CPU path)")
    return "bf16" if dt == torch.bfloat16 else "f32"


# ------------------------------------------------------------------------------------------------
# raw ops (caller allocates everything, ops mutate in place and return None)
# ------------------------------------------------------------------------------------------------
# torch.ops.wind_backstepping.{forward,backward} launch the chunked (MFMA) pair whenever they can -- bf16 tensors, T % 32 == 0 --
# with `s` (a private forward -> backward scratch in the reference, rwkv_s2s_single_ffn.py:22-35) as an opaque arena of the same
# size (include/rwkv7_hip.h, rwkv7_wkv_fwd_fast_bf16); otherwise (fp32, T % 32 == 16) the scalar kernels, whose `s` holds the
# reference's checkpoints.  False = always the scalar kernels (A/B).
REFERENCE_OP_FAST = True


def reference_op_is_fast(dtype, T):
    """Which kernels torch.ops.wind_backstepping.* launch for tensors of this dtype and length."""
    return bool(REFERENCE_OP_FAST and dtype == torch.bfloat16 and T % 32 == 0)


def wkv7_forward_scalar(w, q, k, v, z, a, y, s, sa):
    """The scalar forward kernel (rwkv7_wkv_fwd_*): s = the reference's checkpoints (fp32 [B,H,T/16,64,64], transposed), which
    wkv7_backward_scalar / wkv7_backward_split consume."""
    B, T, H, C = w.shape
    sfx = _sfx([w, q, k, v, z, a, y], "wind_backstepping.forward")
    assert C == HEAD_SIZE and s.dtype == torch.float32 and sa.dtype == torch.float32
    with torch.cuda.device_of(w), _timed("wkv7_fwd", w):
        rc = getattr(_lib.lib(), "rwkv7_wkv_fwd_" + sfx)(B, T, H, _p(w), _p(q), _p(k), _p(v), _p(z), _p(a),
                                                       _p(y), _p(s), _p(sa), _stream(w))
    _lib.check(rc, "wind_backstepping.forward")


def wkv7_backward_scalar(w, q, k, v, z, a, dy, s, sa, dw, dq, dk, dv, dz, da):
    B, T, H, C = w.shape
    sfx = _sfx([w, q, k, v, z, a, dy, dw, dq, dk, dv, dz, da], "wind_backstepping.backward")
    with torch.cuda.device_of(w), _timed("wkv7_bwd", w):
        rc = getattr(_lib.lib(), "rwkv7_wkv_bwd_" + sfx)(B, T, H, _p(w), _p(q), _p(k), _p(v), _p(z), _p(a),
                                                       _p(dy), _p(s), _p(sa), _p(dw), _p(dq), _p(dk), _p(dv),
                                                       _p(dz), _p(da), _stream(w))
    _lib.check(rc, "wind_backstepping.backward")


def _check_arena(w, s, what):
    B, T, H, C = w.shape
    if s.dtype != torch.float32 or not s.is_contiguous() or s.numel() < B * H * (T // CHUNK_LEN) * C * C:
        raise ValueError(f"{what}: s must be the reference's contiguous fp32 [B,H,T/16,64,64] scratch (rwkv_s2s_single_ffn.py:23)")


def _wb_forward(w, q, k, v, z, a, y, s, sa):
    B, T, H, C = w.shape
    if not reference_op_is_fast(w.dtype, T):
        return wkv7_forward_scalar(w, q, k, v, z, a, y, s, sa)
    _sfx([w, q, k, v, z, a, y], "wind_backstepping.forward")
    assert C == HEAD_SIZE and sa.dtype == torch.float32
    _check_arena(w, s, "wind_backstepping.forward")
    with torch.cuda.device_of(w), _timed("wkv7c_op_fwd", w):
        rc = _lib.lib().rwkv7_wkv_fwd_fast_bf16(B, T, H, _p(w), _p(q), _p(k), _p(v), _p(z), _p(a), _p(y), _p(s), _p(sa), _stream(w))
    _lib.check(rc, "wind_backstepping.forward")


def _wb_backward(w, q, k, v, z, a, dy, s, sa, dw, dq, dk, dv, dz, da):
    B, T, H, C = w.shape
    if not reference_op_is_fast(w.dtype, T):
        return wkv7_backward_scalar(w, q, k, v, z, a, dy, s, sa, dw, dq, dk, dv, dz, da)
    _sfx([w, q, k, v, z, a, dy, dw, dq, dk, dv, dz, da], "wind_backstepping.backward")
    _check_arena(w, s, "wind_backstepping.backward")
    with torch.cuda.device_of(w), _timed("wkv7c_op_bwd", w):
        rc = _lib.lib().rwkv7_wkv_bwd_fast_bf16(B, T, H, _p(w), _p(q), _p(k), _p(v), _p(z), _p(a), _p(dy), _p(s), _p(sa),
                                                _p(dw), _p(dq), _p(dk), _p(dv), _p(dz), _p(da), _stream(w))
    _lib.check(rc, "wind_backstepping.backward")


def wkv7_backward_split(w, q, k, v, z, a, dy, s, sa, wide=None):
    """WKV7 backward with each head split over two workgroups (rwkv7_wkv_bwd_split_*: all 256 CUs busy at B*H=128); s, sa as
    written by wkv7_forward_scalar.
    Returns (dw2, dq2, dk2, dv, dz2, da2): the *2 tensors are [2, B,T,H,64] partial column sums whose sum over dim 0
    is the gradient wind_backstepping.backward returns; dv is complete.
    wide (bf16 only, measurements/tests): 0 / 1 selects the 256- / 512-thread shape explicitly (rwkv7_wkv_bwd_split_variant_bf16)."""
    B, T, H, C = w.shape
    sfx = _sfx([w, q, k, v, z, a, dy], "wkv7_backward_split")
    dw2, dq2, dk2, dz2, da2 = [torch.empty((2,) + tuple(w.shape), dtype=w.dtype, device=w.device) for _ in range(5)]
    dv = torch.empty_like(v)
    pair = lambda t: (ctypes.c_void_p * 2)(t[0].data_ptr(), t[1].data_ptr())
    with torch.cuda.device_of(w), _timed("wkv7_bwd", w):
        args = (B, T, H, _p(w), _p(q), _p(k), _p(v), _p(z), _p(a), _p(dy), _p(s), _p(sa), pair(dw2), pair(dq2), pair(dk2),
                _p(dv), pair(dz2), pair(da2))
        if wide is None:
            rc = getattr(_lib.lib(), "rwkv7_wkv_bwd_split_" + sfx)(*args, _stream(w))
        else:
            rc = getattr(_lib.lib(), "rwkv7_wkv_bwd_split_variant_" + sfx)(*args, int(wide), _stream(w))
    _lib.check(rc, "wkv7_backward_split")
    return dw2, dq2, dk2, dv, dz2, da2


def _state_forward(B, T, C, H, state, r, w, k, v, a, b, y):
    sfx = _sfx([r, w, k, v, a, b, y], "rwkv7_state_fwd.forward")
    if state.dtype != torch.float32 or not state.is_contiguous():
        raise TypeError("rwkv7_state_fwd.forward: state must be contiguous float32 [B,H,64,64]")
    with torch.cuda.device_of(r):
        rc = getattr(_lib.lib(), "rwkv7_wkv_state_fwd_" + sfx)(int(B), int(T), int(C), int(H), _p(state), _p(r),
                                                             _p(w), _p(k), _p(v), _p(a), _p(b), _p(y), _stream(r))
    _lib.check(rc, "rwkv7_state_fwd.forward")


def _wkv7s_forward(B, T, C, H, state, r, w, k, v, a, b, y):
    assert B == 1, "wkv7s is the B=1 operator (wkv7s.cu:62)"
    _state_forward(B, T, C, H, state, r, w, k, v, a, b, y)


_registered = []


def _register():
    """Define the three reference op namespaces and bind them to the HIP library."""
    if _registered:
        return
    defs = [
        ("wind_backstepping",
         [("forward(Tensor w, Tensor q, Tensor k, Tensor v, Tensor z, Tensor a, Tensor(a!) y, Tensor(b!) s, "
           "Tensor(c!) sa) -> ()", "forward", _wb_forward),
          ("backward(Tensor w, Tensor q, Tensor k, Tensor v, Tensor z, Tensor a, Tensor dy, Tensor s, Tensor sa, "
           "Tensor(a!) dw, Tensor(b!) dq, Tensor(c!) dk, Tensor(d!) dv, Tensor(e!) dz, Tensor(f!) da) -> ()",
           "backward", _wb_backward)]),
        ("rwkv7_state_fwd_fp16",
         [("forward(int B, int T, int C, int H, Tensor(a!) state, Tensor r, Tensor w, Tensor k, Tensor v, "
           "Tensor a, Tensor b, Tensor(b!) y) -> ()", "forward", _state_forward)]),
        ("wkv7s",
         [("forward(int B, int T, int C, int H, Tensor(a!) state, Tensor r, Tensor w, Tensor k, Tensor v, "
           "Tensor a, Tensor b, Tensor(b!) y) -> ()", "forward", _wkv7s_forward)]),
    ]
    for ns, ops in defs:
        lib = torch.library.Library(ns, "DEF")
        for schema, name, fn in ops:
            lib.define(schema)
            lib.impl(name, fn, "CUDA")
        _registered.append(lib)  # keep alive


_register()


# ------------------------------------------------------------------------------------------------
# autograd wrapper and call-site helpers (reference names)
# ------------------------------------------------------------------------------------------------
class WindBackstepping(torch.autograd.Function):
    """rwkv_s2s_single_ffn.py:15-35.  Argument order (w,q,k,v,z,b) with z == a, b == b."""

    @staticmethod
    def forward(ctx, w, q, k, v, z, b):
        B, T, H, C = w.shape
        assert T % CHUNK_LEN == 0, f"T={T} must be a multiple of {CHUNK_LEN}"
        assert all(i.dtype == w.dtype for i in [w, q, k, v, z, b])
        assert all(i.is_contiguous() for i in [w, q, k, v, z, b])
        y = torch.empty_like(v)
        s = torch.empty(B, H, T // CHUNK_LEN, C, C, dtype=torch.float32, device=w.device)
        sa = torch.empty(B, T, H, C, dtype=torch.float32, device=w.device)
        torch.ops.wind_backstepping.forward(w, q, k, v, z, b, y, s, sa)
        ctx.save_for_backward(w, q, k, v, z, b, s, sa)
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous()
        w, q, k, v, z, b, s, sa = ctx.saved_tensors
        assert dy.dtype == w.dtype
        dw, dq, dk, dv, dz, db = [torch.empty_like(x) for x in [w, q, k, v, z, b]]
        torch.ops.wind_backstepping.backward(w, q, k, v, z, b, dy, s, sa, dw, dq, dk, dv, dz, db)
        return dw, dq, dk, dv, dz, db


def RUN_CUDA_RWKV7g(q, w, k, v, a, b):
    """rwkv_s2s_single_ffn.py:37-40: [B,T,H*64] in, [B,T,H*64] out, differentiable."""
    B, T, HC = q.shape
    q, w, k, v, a, b = [i.view(B, T, HC // HEAD_SIZE, HEAD_SIZE) for i in [q, w, k, v, a, b]]
    return WindBackstepping.apply(w, q, k, v, a, b).view(B, T, HC)


def wkv7_forward_nograd(q, w, k, v, a, b):
    """Inference-only zero-state scan: no checkpoints, no sa (s = sa = NULL in the C ABI)."""
    B, T, HC = q.shape
    H = HC // HEAD_SIZE
    sfx = _sfx([w, q, k, v, a, b], "wkv7_forward_nograd")
    if T % CHUNK_LEN != 0:
        raise ValueError("T must be a multiple of 16; use the state-carrying op for ragged lengths")
    y = torch.empty_like(v)
    with torch.cuda.device_of(w):
        rc = getattr(_lib.lib(), "rwkv7_wkv_fwd_" + sfx)(B, T, H, _p(w), _p(q), _p(k), _p(v), _p(a), _p(b),
                                                       _p(y), None, None, _stream(w))
    _lib.check(rc, "wkv7_forward_nograd")
    return y


def RWKV7_OP(state, r, w, k, v, a, b):
    """rwkv_s2s_single_ffn.py:45-59 (torch.ops.wkv7s): r..b [T,C], state [H,64,64] updated in place."""
    with torch.no_grad():
        T, C = r.shape
        H = C // HEAD_SIZE
        y = torch.empty((T, C), device=k.device, dtype=r.dtype)
        torch.ops.wkv7s.forward(1, T, C, H, state, r, w, k, v, a, b, y)
        return y


def RWKV7_BATCH_OP(state, r, w, k, v, a, b):
    """rwkv_asr_cuda_whisper.py:67-81: r..b [B,T,C], state [B,H,64,64] fp32 updated in place."""
    with torch.no_grad():
        B, T, C = r.shape
        H = C // HEAD_SIZE
        y = torch.empty((B, T, C), device=k.device, dtype=r.dtype)
        torch.ops.rwkv7_state_fwd_fp16.forward(B, T, C, H, state, r, w, k, v, a, b, y)
        return y


# ------------------------------------------------------------------------------------------------
# chunked (MFMA) WKV7: the training fast path (csrc/chunk_common.h, wkv7_chunk_fwd.hip, wkv7_chunk_bwd.hip)
# ------------------------------------------------------------------------------------------------
CHUNK_T = 32
Q15_REC = 64 * 64 + 2 * 256   # int16 units of one 64x64 state checkpoint: 4096 mantissas + 256 fp32 scales
_Q15_IDX = None


def q15_decode(rec):
    """[..., Q15_REC] int16 records (csrc/chunk_common.h: what the chunked forward saves as hs and the adjoint-state kernel as
    e_vk; mantissas in MFMA accumulator order mant[vh][kt][lane][16], one scale per lane) -> fp32 [..., 64 (value), 64 (key)]."""
    global _Q15_IDX
    if _Q15_IDX is None:
        v = torch.arange(64).view(64, 1).expand(64, 64)
        k = torch.arange(64).view(1, 64).expand(64, 64)
        slot = ((v >> 5) * 2 + (k >> 5)) * 64 + (v & 31) + 32 * ((k >> 2) & 1)
        r = (k & 3) + 4 * ((k >> 3) & 3)
        _Q15_IDX = ((slot * 16 + r).reshape(-1), slot.reshape(-1))
    mi, si = (t.to(rec.device) for t in _Q15_IDX)
    q = rec[..., :4096].float().index_select(-1, mi)
    sc = rec[..., 4096:].contiguous().view(torch.float32).index_select(-1, si)
    return (q * sc).reshape(*rec.shape[:-1], 64, 64)


def wkv7_chunk_prep(w, a, b):
    """(I - A_ab)^-1 of every 32-step chunk: fp32 [B,H,T/32,32,32].  w,a,b: [B,T,H,64]."""
    B, T, H, C = w.shape
    sfx = _sfx([w, a, b], "wkv7_chunk_prep")
    tinv = torch.empty(B, H, T // CHUNK_T, CHUNK_T, CHUNK_T, dtype=torch.float32, device=w.device)
    with torch.cuda.device_of(w), _timed("wkv7c_prep", w):
        rc = getattr(_lib.lib(), "rwkv7_wkv_chunk_prep_" + sfx)(B, T, H, _p(w), _p(a), _p(b), _p(tinv), _stream(w))
    _lib.check(rc, "wkv7_chunk_prep")
    return tinv


def wkv7_chunk_forward(w, q, k, v, a, b, save=True, seq_off=None):
    """Chunked forward.  Returns y, and (tinv, sa, hs) when save (what the chunked backward consumes; hs = the state at the
    start of every chunk as q15 records, see q15_decode).
    seq_off: packed rows -- int32 [nseq + 1] device tensor of cumulative 32-step chunk counts over the [B][T/32] chunk
    space; sequence s owns chunks seq_off[s] .. seq_off[s+1] - 1 and starts from the zero state."""
    B, T, H, C = w.shape
    sfx = _sfx([w, q, k, v, a, b], "wkv7_chunk_forward")
    if T % CHUNK_T != 0:
        raise ValueError(f"chunked WKV7 needs T % {CHUNK_T} == 0, got T={T}")
    tinv = wkv7_chunk_prep(w, a, b)
    y = torch.empty_like(v)
    sa = torch.empty(B, T, H, C, dtype=torch.float32, device=w.device) if save else None
    hs = torch.empty(B, H, T // CHUNK_T, Q15_REC, dtype=torch.int16, device=w.device) if save else None
    with torch.cuda.device_of(w), _timed("wkv7c_fwd", w):
        args = (B, T, H, _p(w), _p(q), _p(k), _p(v), _p(a), _p(b), _p(tinv), _p(y),
                None if sa is None else _p(sa), None if hs is None else _p(hs), *_seq_args(seq_off))
        rc = getattr(_lib.lib(), "rwkv7_wkv_chunk_fwd_seq_" + sfx)(*args, _stream(w))
    _lib.check(rc, "wkv7_chunk_forward")
    return (y, tinv, sa, hs) if save else y


def _seq_args(seq_off):
    if seq_off is None:
        return None, 0
    if seq_off.dtype != torch.int32 or seq_off.dim() != 1 or seq_off.numel() < 2 or not seq_off.is_contiguous() or not seq_off.is_cuda:
        raise TypeError("seq_off must be a contiguous int32 [nseq + 1] device tensor")
    return _p(seq_off), seq_off.numel() - 1


def wkv7_chunk_bwd_seq(w, q, a, b, dy, tinv, seq_off=None, want_z=False):
    """The adjoint-state recurrence of the chunked backward as ONE sequential kernel (csrc/wkv7_chunk_bseq.hip): the factored
    form E_c = E' + A~^T Z + Q~^T dY, Z = (T^T B^) E' + (T^T A_qb^T) dY -- M_c^T / N'_c are not materialised.  Returns e_vk
    (e_vk[b,h,c] = E_{c+1} as q15 records); with want_z also Z (fp32 [B,T,H,64],
    Z_t = dL/du_t) as (e_vk, z)."""
    B, T, H, C = w.shape
    if w.dtype != torch.bfloat16:
        raise TypeError("the chunked backward is bf16 only")
    if T % CHUNK_T != 0:
        raise ValueError(f"chunked WKV7 needs T % {CHUNK_T} == 0, got T={T}")
    e_vk = torch.empty(B, H, T // CHUNK_T, Q15_REC, dtype=torch.int16, device=w.device)
    # packed rows leave the positions behind the last sequence untouched: zeros there (the gradient kernel reads every chunk)
    z = (torch.empty if seq_off is None else torch.zeros)(B, T, H, C, dtype=torch.float32, device=w.device) if want_z else None
    with torch.cuda.device_of(w), _timed("wkv7c_bseq", w):
        rc = _lib.lib().rwkv7_wkv_chunk_bseq_bf16(B, T, H, _p(w), _p(q), _p(a), _p(b), _p(dy), _p(tinv), _p(e_vk), _p(z),
                                                  *_seq_args(seq_off), _stream(w))
    _lib.check(rc, "wkv7_chunk_bseq")
    return (e_vk, z) if want_z else e_vk


def wkv7_chunk_backward(w, q, k, v, a, b, dy, hs, sa, tinv, seq_off=None):
    """Chunked (MFMA) WKV7 backward, bf16: same gradients as torch.ops.wind_backstepping.backward, T % 32 == 0, from what
    wkv7_chunk_forward saved (hs, sa, tinv).  Two launches: the adjoint-state recurrence (csrc/wkv7_chunk_bseq.hip, which also
    writes Z = dL/du) and the per-chunk gradients from Z (csrc/wkv7_chunk_bwd10.hip, two matrix phases).
    Returns (dw, dq, dk, dv, da, db)."""
    B, T, H, C = w.shape
    if hs.dtype != torch.int16 or hs.shape[-1] != Q15_REC or sa.dtype != torch.float32 or tinv.dtype != torch.float32:
        raise TypeError("wkv7_chunk_backward takes hs (q15 records), sa and tinv (fp32) as saved by wkv7_chunk_forward")
    e_vk, z = wkv7_chunk_bwd_seq(w, q, a, b, dy, tinv, seq_off, want_z=True)
    grads = [torch.empty_like(w) for _ in range(6)]
    with torch.cuda.device_of(w), _timed("wkv7c_bwd_out", w):
        rc = _lib.lib().rwkv7_wkv_chunk_bwd_out_z_bf16(B, T, H, _p(w), _p(q), _p(k), _p(v), _p(a), _p(b), _p(dy), _p(hs), _p(sa),
                                                       _p(z), _p(e_vk), *[_p(g) for g in grads], _stream(w))
    _lib.check(rc, "wkv7_chunk_bwd_out")
    return tuple(grads)


def debug_mma32(X, Y):
    """GPU unit-test hook: D = X Y^T for X,Y fp32 [32,64] through the bf16-split MFMA primitive; returns (D, DT)."""
    D = torch.empty(32, 32, device=X.device)
    DT = torch.empty(32, 32, device=X.device)
    rc = _lib.lib().rwkv7_debug_mma32(_p(X.contiguous()), _p(Y.contiguous()), _p(D), _p(DT), _stream(X))
    _lib.check(rc, "debug_mma32")
    return D, DT
