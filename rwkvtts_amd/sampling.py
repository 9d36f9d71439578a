"""Host side of csrc/sampling.hip: the token draws of the generation loops as one launch each.

RowSampler   argmax or HF's temperature -> top-k -> top-p -> multinomial chain (spark_llm.sample_next, the reference's generate():
             utils/utilities.py:101-117; the eight channels of an XY frame: model/llm/xy_llm.py:88-101) for every (row, segment)
             of a logits matrix, ids straight from the decode step's fp32 logits.
ras_step     CosyVoice's repetition-aware sampler + the streaming loop's bookkeeping (cosy_llm.ras_sampling_device).
Both draw with a counter-based generator keyed by (seed, step tensor): no torch generator state, graph-capturable, reproducible.
"""
import ctypes
from typing import Optional, Sequence

import torch

from . import _lib

MAX_DOMAIN = 15360
MAX_TOP_K = 64


def fresh_seed() -> int:
    """A key for one generation call: drawn from torch's default (CPU) generator, so that repeated calls draw different ids and
    torch.manual_seed(...) makes a run reproducible -- like the torch chains, which consume the global generator."""
    return int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())


class SampleTail(ctypes.Structure):
    """rwkv7_sample_tail (include/rwkv7_hip.h): what the decode loop does with a drawn id, in the draw's launch."""
    _fields_ = [("unfinished", ctypes.c_void_p), ("eos", ctypes.c_long), ("pad", ctypes.c_long), ("ids", ctypes.c_void_p),
                ("seq", ctypes.c_void_p), ("seq_ld", ctypes.c_long), ("emb", ctypes.c_void_p), ("x", ctypes.c_void_p), ("D", ctypes.c_int)]

    @classmethod
    def make(cls, ids, seq=None, unfinished=None, eos=None, pad=0, emb=None, x=None):
        """ids [rows] int64; seq [rows, n] int64 (column *step receives the id); unfinished [rows] bool with eos / pad; emb bf16
        [V, D] and x bf16 [rows, D] (the next step's input).  The tensors must outlive the launches."""
        assert ids.dtype == torch.int64 and ids.is_contiguous()
        t = cls()
        t.ids = ids.data_ptr()
        t.unfinished = None
        t.eos, t.pad = -1, int(pad)
        if unfinished is not None and eos is not None:
            assert unfinished.dtype == torch.bool and unfinished.is_contiguous()
            t.unfinished, t.eos = unfinished.data_ptr(), int(eos)
        t.seq, t.seq_ld = None, 0
        if seq is not None:
            assert seq.dtype == torch.int64 and seq.is_contiguous() and seq.shape[0] == ids.shape[0]
            t.seq, t.seq_ld = seq.data_ptr(), seq.shape[1]
        t.emb, t.x, t.D = None, None, 0
        if emb is not None:
            assert emb.dtype == torch.bfloat16 and emb.is_contiguous() and x.dtype == torch.bfloat16 and x.is_contiguous()
            assert x.shape == (ids.shape[0], emb.shape[1]) and emb.shape[1] % 8 == 0
            t.emb, t.x, t.D = emb.data_ptr(), x.data_ptr(), emb.shape[1]
        t._keep = (ids, seq, unfinished, emb, x)
        return t


class RowSampler:
    @staticmethod
    def supported(logits_device, seg_len: Sequence[int], allow=None, suppress=None, do_sample=False, top_k=0, top_p=1.0,
                  temperature=1.0) -> Optional[str]:
        """None if rwkv7_sample_rows_f32 covers the request, else the reason (the caller then runs the torch chain)."""
        if logits_device.type != "cuda":
            return "logits not on the HIP device"
        dom = [(hi - lo) for lo, hi in allow] if allow is not None else list(seg_len)
        if max(dom) > MAX_DOMAIN or min(dom) < 1:
            return f"segment of {max(dom)} ids"
        if suppress is not None and len(suppress) > 256:
            return "more than 256 suppressed ids"
        if do_sample:
            top_k, top_p = int(top_k or 0), 1.0 if top_p is None else float(top_p)
            if not (temperature and temperature > 0):
                return "temperature"
            if top_k < 0 or top_k > MAX_TOP_K:
                return f"top_k = {top_k}"
            if top_k == 0 and top_p < 1.0:
                return "top_p without top_k"
        return None

    @staticmethod
    def fold_suppress(n: int, suppress):
        """A long suppress list that leaves one contiguous range of ids (the reference's global-token stage suppresses
        range(num_global_tokens, vocab_size): utils/utilities.py:100) -> (allowed range, the suppressed ids inside it)."""
        if suppress is None or len(suppress) <= 256:
            return None, suppress
        sup = sorted({int(t) for t in suppress if 0 <= int(t) < n})
        blocked = set(sup)
        lo = next((i for i in range(n) if i not in blocked), None)
        if lo is None:
            return None, suppress
        hi = next(i for i in range(n - 1, -1, -1) if i not in blocked) + 1
        return (lo, hi), [t for t in sup if lo <= t < hi]

    def __init__(self, device, seg_len: Sequence[int], allow=None, suppress=None, do_sample=False, top_k=0, top_p=1.0, temperature=1.0,
                 seed: Optional[int] = None, min_eos: Optional[tuple] = None, seg_off: Optional[Sequence[int]] = None):
        """min_eos = (id, n): `id` cannot be drawn while the step counter is below n (min_new_tokens).
        seg_off: where each segment's id 0 sits in a logits row (default: the segments back to back).  An offset may be negative
        when the row holds only the allowed part of a segment: segment s with allow = (lo, hi) stored at column c has
        seg_off = c - lo, so that ids keep their vocabulary values (XY channel 0: the audio range of a 66 661-id head)."""
        self.min_id, self.min_until = (-1, 0) if min_eos is None else (int(min_eos[0]), int(min_eos[1]))
        why = self.supported(device, seg_len, allow, suppress, do_sample, top_k, top_p, temperature)
        if why:
            raise ValueError("rwkv7_sample_rows_f32: " + why)
        self.nseg = len(seg_len)
        if seg_off is None:
            off = [0]
            for n in seg_len[:-1]:
                off.append(off[-1] + int(n))
        else:
            off = [int(o) for o in seg_off]
        ends = [(allow[i][1] if allow is not None else int(seg_len[i])) for i in range(len(seg_len))]
        self.width = max(o + e for o, e in zip(off, ends))   # columns a logits row must have
        i32 = dict(dtype=torch.int32, device=device)
        self.seg_off, self.seg_len = torch.tensor(off, **i32), torch.tensor([int(n) for n in seg_len], **i32)
        self.allow_lo = self.allow_hi = None
        if allow is not None:
            self.allow_lo, self.allow_hi = torch.tensor([a[0] for a in allow], **i32), torch.tensor([a[1] for a in allow], **i32)
        self.max_domain = max((hi - lo) for lo, hi in allow) if allow is not None else max(int(n) for n in seg_len)
        self.suppress = torch.tensor([int(t) for t in suppress], **i32) if suppress is not None and len(suppress) else None
        self.do_sample, self.top_k = int(bool(do_sample)), int(top_k or 0)
        self.top_p, self.temperature = 1.0 if top_p is None else float(top_p), float(temperature or 1.0)
        self.seed = int(fresh_seed() if seed is None else seed) & ((1 << 64) - 1)

    def __call__(self, logits: torch.Tensor, step: torch.Tensor, out: Optional[torch.Tensor] = None,
                 tail: Optional[SampleTail] = None) -> torch.Tensor:
        """logits fp32 [rows, >= width] (row stride arbitrary, unit column stride); step: int64 device tensor (first element is
        read); returns int64 [rows, nseg] (the raw draws).  tail (one segment): the decode loop's handling of the id in the same
        launch (SampleTail.make)."""
        assert logits.dtype == torch.float32 and logits.dim() == 2 and logits.stride(1) == 1 and logits.shape[1] >= self.width
        assert step.dtype == torch.int64 and step.is_cuda
        rows = logits.shape[0]
        if out is None:
            out = torch.empty(rows, self.nseg, dtype=torch.int64, device=logits.device)
        p = lambda t: ctypes.c_void_p(t.data_ptr() if t is not None else None)
        common = (p(self.seg_off), p(self.seg_len), p(self.allow_lo), p(self.allow_hi), p(self.suppress),
                  0 if self.suppress is None else self.suppress.numel(), self.max_domain, self.do_sample, self.top_k,
                  ctypes.c_float(self.top_p), ctypes.c_float(self.temperature), ctypes.c_ulonglong(self.seed), p(step), p(out))
        stream = ctypes.c_void_p(torch.cuda.current_stream(logits.device).cuda_stream)
        with torch.cuda.device_of(logits):
            if tail is None and self.min_id < 0:
                rc = _lib.lib().rwkv7_sample_rows_f32(rows, self.nseg, p(logits), ctypes.c_long(logits.stride(0)), *common, stream)
            else:
                assert self.nseg == 1
                rc = _lib.lib().rwkv7_sample_rows_tail_f32(rows, p(logits), ctypes.c_long(logits.stride(0)), *common,
                                                           ctypes.byref(tail) if tail is not None else None, self.min_id,
                                                           ctypes.c_long(self.min_until), stream)
        _lib.check(rc, "rwkv7_sample_rows_f32")
        return out


def ras_step(logits: torch.Tensor, tok: torch.Tensor, recent: torch.Tensor, ptr: torch.Tensor, step_i: torch.Tensor, n_ignore: int,
             eos: int, top_p=0.8, top_k=25, win_size=10, tau_r=0.1, seed: Optional[int] = None):
    """One token of CosyVoice's streaming loop on device tensors (see rwkv7_ras_step_f32): logits fp32 [V]; tok [1], recent
    [win_size], ptr [1], step_i [] int64, all updated in place."""
    assert logits.dtype == torch.float32 and logits.is_contiguous() and logits.dim() == 1
    for t in (tok, recent, ptr, step_i):
        assert t.dtype == torch.int64 and t.is_cuda and t.is_contiguous()
    assert recent.numel() == win_size
    seed = int(fresh_seed() if seed is None else seed) & ((1 << 64) - 1)   # (callers in a loop pass one seed per generation)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    with torch.cuda.device_of(logits):
        rc = _lib.lib().rwkv7_ras_step_f32(logits.numel(), p(logits), p(tok), p(recent), p(ptr), p(step_i), ctypes.c_long(int(n_ignore)),
                                           int(eos), ctypes.c_float(top_p), int(top_k), int(win_size), ctypes.c_float(tau_r),
                                           ctypes.c_ulonglong(seed), ctypes.c_void_p(torch.cuda.current_stream(logits.device).cuda_stream))
    _lib.check(rc, "rwkv7_ras_step_f32")


def xy_frame_step(nt, out, row, pos, unfinished, needs, all_done, n_rows, text_shift, speech_vocab, pad, eos0, total, eos_list,
                  reference_termination):
    """One frame of the XY loop's bookkeeping on device tensors (rwkv7_xy_frame_step; the torch form: xy_llm._XYFrameState.step).
    nt [B, C] int64 drawn ids; out [B, rows, C], row [B, C], pos [1], unfinished / needs [B], n_rows [] int64; all_done [] bool."""
    B, C = nt.shape
    for t in (nt, out, row, pos, unfinished, needs, n_rows):
        assert t.dtype == torch.int64 and t.is_cuda and t.is_contiguous()
    assert all_done.dtype == torch.bool and out.shape[0] == B and out.shape[2] == C
    p = lambda t: ctypes.c_void_p(t.data_ptr() if t is not None else None)
    L = ctypes.c_long
    with torch.cuda.device_of(nt):
        rc = _lib.lib().rwkv7_xy_frame_step(B, C, out.shape[1], L(text_shift), L(speech_vocab), L(pad), L(-1 if eos0 is None else eos0),
                                            L(-1 if total is None else total), p(eos_list), 0 if eos_list is None else eos_list.numel(),
                                            int(bool(reference_termination)), p(nt), p(out), p(row), p(pos), p(unfinished), p(needs),
                                            p(all_done), p(n_rows), ctypes.c_void_p(torch.cuda.current_stream(nt.device).cuda_stream))
    _lib.check(rc, "rwkv7_xy_frame_step")


class XYEmbed:
    """x[b] = sum over channels of table_c[row[b, c]] as one launch (rwkv7_xy_embed_bf16), into a fixed buffer."""

    @staticmethod
    def supported(tables) -> bool:
        D = tables[0].shape[1]
        return len(tables) <= 16 and D % 8 == 0 and all(t.dtype == torch.bfloat16 and t.is_cuda and t.is_contiguous() and t.shape[1] == D
                                                        for t in tables)

    def __init__(self, tables, B):
        self.tables = [t.detach() for t in tables]
        self.C, self.D = len(tables), tables[0].shape[1]
        self.ptrs = (ctypes.c_void_p * self.C)(*[t.data_ptr() for t in self.tables])
        self.x = torch.empty(B, self.D, dtype=torch.bfloat16, device=tables[0].device)

    def __call__(self, row: torch.Tensor) -> torch.Tensor:
        assert row.dtype == torch.int64 and row.is_contiguous() and row.shape == (self.x.shape[0], self.C)
        with torch.cuda.device_of(row):
            rc = _lib.lib().rwkv7_xy_embed_bf16(row.shape[0], self.C, self.D, self.ptrs, ctypes.c_void_p(row.data_ptr()),
                                                ctypes.c_void_p(self.x.data_ptr()),
                                                ctypes.c_void_p(torch.cuda.current_stream(row.device).cuda_stream))
        _lib.check(rc, "rwkv7_xy_embed_bf16")
        return self.x
