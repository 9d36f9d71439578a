"""Persistent-state greedy decode (BASELINE.json configs[4]; reference loops: model/llm/rwkv_asr_cuda_whisper.py:694-717,
rwkv_s2s_single_ffn.py:417-445, HF generate via inference/rwkv7speech_inference.py:99-107).

Two levels:
  * DecodeStep -- the whole T = 1 step of the stack (all layers on the in-place recurrent state, final norm, head
    projection) through rwkv7_decode_step_bf16 (csrc/decode_step.hip): 7 grid-wide phases per layer instead of ~18
    module-level launches, either as one launch per phase (default) or as ONE persistent kernel with device-scope barriers
    between the phases (persistent=1; slower on MI355X, the barrier costs 7.4 us against ~1.5 us for a kernel boundary).
  * GraphDecoder -- greedy or sampled (temperature / top-k / top-p, device RNG) loop around it: embedding lookup of the
    previous ids, the step, suppress + argmax/sampling and the bookkeeping are recorded once into a hipGraph on static buffers
    and replayed per token; the ids never leave the device until the end.  Models the step kernel does not cover (fp32 weights, B > 32, odd low-rank sizes) run the module-by-module
    step inside the same graph.
"""
from __future__ import annotations

import ctypes
from typing import Optional, Sequence

import torch
import torch.nn.functional as F

from . import _lib
from .backbone import Cache

# order of include/rwkv7_hip.h: RWKV7_DEC_*
_DEC_ORDER = ("ln0_w", "ln0_b", "ln1_w", "ln1_b", "ln2_w", "ln2_b", "x_r", "x_w", "x_k", "x_v", "x_a", "x_g",
              "wr", "wk", "wv", "wo", "w1", "w2", "w0", "a1", "a2", "a0", "v1", "v2", "v0", "g1", "g2",
              "k_k", "k_a", "r_k", "gn_w", "gn_b", "fx_k", "wkey", "wval", "att_x_prev", "att_kv", "ffn_x_prev")


class _Dims(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int) for n in ("B", "D", "H", "L", "F", "V", "Rw", "Ra", "Rv", "Rg")] + \
               [("ln_eps", ctypes.c_float), ("gn_eps", ctypes.c_float)]


class DecodeStep:
    """logits[B,V] (fp32) = step(x_in[B,D] bf16) on the live `cache` of a bf16 RWKV7Model + head; the cache's state tensors
    are updated in place.  Raises ValueError for shapes/dtypes the kernel does not cover (see `supported`)."""

    def __init__(self, backbone, lm_head, cache: Cache, persistent: int = 0, host_table: bool = True):
        why = self.supported(backbone, lm_head, cache)
        if why:
            raise ValueError("rwkv7_decode_step_bf16: " + why)
        # host_table = False: rwkv7_decode_step_bf16 (device table only: every phase kernel fetches its pointer row first)
        self.host_table = bool(host_table)
        cfg = backbone.config
        lib = _lib.lib()
        if lib.rwkv7_decode_layer_ptrs() != len(_DEC_ORDER):
            raise _lib.Rwkv7HipError("librwkv7_hip.so and rwkvtts_amd/decode.py disagree on the layer pointer table")
        self.B = cache[0].att_x_prev.shape[0]
        a0 = backbone.layers[0].attn
        self.dims = _Dims(self.B, cfg.hidden_size, cfg.num_heads, len(backbone.layers), backbone.layers[0].ffn.key.weight.shape[0],
                          lm_head.weight.shape[0], a0.w_lora.rank, a0.a_lora.rank, backbone.layers[1].attn.v_lora.rank
                          if len(backbone.layers) > 1 else 32, a0.g_lora.rank, cfg.norm_eps, a0.g_norm.eps)
        lib.rwkv7_decode_workspace_bytes.restype = ctypes.c_size_t
        nbytes = lib.rwkv7_decode_workspace_bytes(ctypes.byref(self.dims))
        if nbytes == 0:
            raise ValueError("rwkv7_decode_step_bf16: unsupported shape")
        dev = lm_head.weight.device
        self.workspace = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
        rows = []
        self._keep = []  # tensors whose addresses are in the table
        for i, blk in enumerate(backbone.layers):
            at, ff, st = blk.attn, blk.ffn, cache[i]
            t = dict(ln0_w=blk.pre_norm.weight if i == 0 else None, ln0_b=blk.pre_norm.bias if i == 0 else None,
                     ln1_w=blk.attn_norm.weight, ln1_b=blk.attn_norm.bias, ln2_w=blk.ffn_norm.weight, ln2_b=blk.ffn_norm.bias,
                     x_r=at.x_r, x_w=at.x_w, x_k=at.x_k, x_v=at.x_v, x_a=at.x_a, x_g=at.x_g,
                     wr=at.r_proj.weight, wk=at.k_proj.weight, wv=at.v_proj.weight, wo=at.o_proj.weight,
                     w1=at.w_lora.lora[0].weight, w2=at.w_lora.lora[2].weight, w0=at.w_lora.lora[2].bias,
                     a1=at.a_lora.lora[0].weight, a2=at.a_lora.lora[2].weight, a0=at.a_lora.lora[2].bias,
                     v1=at.v_lora.lora[0].weight if i else None, v2=at.v_lora.lora[2].weight if i else None,
                     v0=at.v_lora.lora[2].bias if i else None,
                     g1=at.g_lora.lora[0].weight, g2=at.g_lora.lora[2].weight,
                     k_k=at.k_k, k_a=at.k_a, r_k=at.r_k, gn_w=at.g_norm.weight, gn_b=at.g_norm.bias,
                     fx_k=ff.x_k, wkey=ff.key.weight, wval=ff.value.weight,
                     att_x_prev=st.att_x_prev, att_kv=st.att_kv, ffn_x_prev=st.ffn_x_prev)
            row = []
            for name in _DEC_ORDER:
                ten = t[name]
                if ten is not None:
                    ten = ten.detach()
                    self._keep.append(ten)
                row.append(0 if ten is None else ten.data_ptr())
            rows.append(row)
        self.table_host = torch.tensor(rows, dtype=torch.int64)   # the phase kernels take their pointers as arguments
        self.table = self.table_host.to(dev)
        self.norm, self.head = backbone.norm, lm_head
        self.logits = torch.empty(self.B, self.dims.V, dtype=torch.float32, device=dev)
        self.persistent = int(persistent)   # 2 (debug): the barriers of the persistent kernel without the phases

    @staticmethod
    def supported(backbone, lm_head, cache: Cache) -> Optional[str]:
        """None if the step kernel covers this model/cache, else the reason."""
        cfg = backbone.config
        if cache is None or len(cache) != len(backbone.layers):
            return "no per-layer cache"
        B = cache[0].att_x_prev.shape[0]
        if not 1 <= B <= 32:
            return f"B = {B} (1..32)"
        if cfg.hidden_size > 4096 or cfg.hidden_size % 64:
            return "hidden size"
        if not getattr(cfg, "norm_bias", True):
            return "LayerNorm without bias"
        tensors = [lm_head.weight] + [p for p in backbone.layers.parameters()] + list(backbone.norm.parameters())
        if any(p.dtype != torch.bfloat16 or not p.is_cuda or not p.is_contiguous() for p in tensors):
            return "parameters must be contiguous bf16 tensors on the HIP device"
        a0 = backbone.layers[0].attn
        ranks = [a0.w_lora.rank, a0.a_lora.rank, a0.g_lora.rank] + ([backbone.layers[1].attn.v_lora.rank] if len(backbone.layers) > 1 else [])
        if any(r % 32 or r > 256 for r in ranks) or sum(ranks) > 512:
            return f"low-rank sizes {ranks}"
        if backbone.layers[0].ffn.key.weight.shape[0] % 64:
            return "channel-mix width"
        for st in cache.states:
            if st.att_x_prev.dtype != torch.bfloat16 or st.att_kv.dtype != torch.float32 or st.ffn_x_prev.dtype != torch.bfloat16:
                return "state dtypes"
        return None

    def __call__(self, x_in: torch.Tensor) -> torch.Tensor:
        assert x_in.shape == (self.B, self.dims.D) and x_in.dtype == torch.bfloat16 and x_in.is_contiguous()
        hb = self.head.bias
        with torch.cuda.device_of(x_in):
            tail = (ctypes.c_void_p(x_in.data_ptr()),
                    ctypes.c_void_p(self.norm.weight.data_ptr()), ctypes.c_void_p(self.norm.bias.data_ptr()),
                    ctypes.c_void_p(self.head.weight.data_ptr()), ctypes.c_void_p(hb.data_ptr() if hb is not None else None),
                    ctypes.c_void_p(self.logits.data_ptr()), ctypes.c_void_p(self.workspace.data_ptr()), int(self.persistent),
                    ctypes.c_void_p(torch.cuda.current_stream(x_in.device).cuda_stream))
            if self.host_table:
                rc = _lib.lib().rwkv7_decode_step_tbl_bf16(ctypes.byref(self.dims), ctypes.c_void_p(self.table.data_ptr()),
                                                           ctypes.c_void_p(self.table_host.data_ptr()), *tail)
            else:
                rc = _lib.lib().rwkv7_decode_step_bf16(ctypes.byref(self.dims), ctypes.c_void_p(self.table.data_ptr()), *tail)
        _lib.check(rc, "rwkv7_decode_step_tbl_bf16" if self.host_table else "rwkv7_decode_step_bf16")
        return self.logits

    def barrier_timed_out(self) -> bool:
        """True if a grid barrier of an earlier step was not met in time (the kernel bails out instead of hanging)."""
        return bool(self.workspace[4:8].view(torch.int32).item())


class GraphDecoder:
    def __init__(self, model, batch_size: int, step_kernel: Optional[bool] = None):
        """step_kernel: True = require the persistent step kernel, False = module-by-module step, None = kernel when the
        model is covered (DecodeStep.supported)."""
        self.model = model.eval()
        self.B = batch_size
        self.graph = None
        self.step_kernel = step_kernel
        self.step = None
        self.fused_sampling = True   # False: the torch chain (spark_llm.sample_next) on torch's device generator
        self.seed_offset = 0         # added to the device seed for the fused sampler (MultiGroupDecoder: one stream of draws per group)

    @torch.no_grad()
    def _step(self):
        if self.step is not None and self.tail is not None:
            # step kernel -> draw + EOS / pad handling + output column + the next input's embedding row (one launch) -> position
            self.cache.seen_tokens += 1
            self.sampler(self.step(self.x_next), self.pos, self._raw, self.tail)
            self.pos += 1
            return
        if self.step is not None:
            x = F.embedding(self.ids, self.model.get_input_embeddings().weight)
            logits = self.step(x)           # the step kernel's own buffer: _pick copies it before it edits it
            self.cache.seen_tokens += 1
        else:
            out = self.model(input_ids=self.ids.unsqueeze(1), past_key_values=self.cache, use_cache=True)
            logits = out.logits[:, -1].float()
        nxt = self._pick(logits)
        if self.eos is not None:
            nxt = torch.where(self.unfinished, nxt, self.pad_t)
            self.unfinished &= nxt != self.eos
        self.ids.copy_(nxt)
        self.out.scatter_(1, self.pos.expand(self.B, 1), nxt.unsqueeze(1))
        self.pos += 1

    def _pick(self, logits):
        """argmax, or temperature / top-k / top-p multinomial sampling (spark_llm.sample_next: the HF warper order the reference's
        generate runs, utils/utilities.py:101-117) -- inside the captured step too: torch's device generator is graph-safe
        (philox seed/offset live in device memory and advance per replay), so sampled decode replays like greedy decode."""
        from .sampling import RowSampler, fresh_seed
        from .spark_llm import sample_next
        if self.sampler is None and self._try_fused:
            # suppression, the warper chain and the draw as ONE launch (csrc/sampling.hip) when the request is covered: the torch chain
            # is ~15 launches (0.45 ms of a 1.4 ms step at B = 32); keyed by (seed, self.pos) instead of torch's generator
            self._try_fused = False
            V = logits.shape[-1]
            allow, sup = RowSampler.fold_suppress(V, None if self.suppress is None else self.suppress.tolist())
            allow = None if allow is None else [allow]
            if RowSampler.supported(logits.device, [V], allow, sup, self.do_sample, self.top_k, self.top_p, self.temperature) is None:
                min_eos = (int(self.eos), self.min_new_tokens) if self.eos is not None and self.min_new_tokens else None
                self.sampler = RowSampler(logits.device, [V], allow, sup, self.do_sample, self.top_k, self.top_p, self.temperature,
                                          seed=(fresh_seed() if self._seed is None else self._seed) + self.seed_offset, min_eos=min_eos)
        if self.sampler is not None:
            return self.sampler(logits, self.pos)[:, 0]
        if self.min_new_tokens:
            raise ValueError("min_new_tokens needs the fused sampler (csrc/sampling.hip): request outside its range")
        if self.suppress is not None:
            logits = logits.clone().index_fill_(1, self.suppress, float("-inf"))
        return sample_next(logits, self.do_sample, self.top_k, self.top_p, self.temperature)

    @torch.no_grad()
    def generate(self, inputs_embeds=None, input_ids=None, attention_mask=None, max_new_tokens=256,
                 eos_token_id: Optional[int] = None, pad_token_id: Optional[int] = None,
                 suppress_tokens: Optional[Sequence[int]] = None, do_sample: bool = False, temperature: float = 1.0,
                 top_k: int = 0, top_p: float = 1.0, seed: Optional[int] = None, min_new_tokens: int = 0):
        self.prepare(inputs_embeds=inputs_embeds, input_ids=input_ids, attention_mask=attention_mask, max_new_tokens=max_new_tokens,
                     eos_token_id=eos_token_id, pad_token_id=pad_token_id, suppress_tokens=suppress_tokens, do_sample=do_sample,
                     temperature=temperature, top_k=top_k, top_p=top_p, seed=seed, min_new_tokens=min_new_tokens)
        ran = 0
        while ran < self.steps_left:
            n = min(self.EOS_CHECK_EVERY if self.eos is not None else self.steps_left, self.steps_left - ran)
            for _ in range(n):
                self.graph.replay()
            ran += n
            # the host loop leaves with the step in which the last sequence meets EOS (the reference's scripts ask for 1 024..3 000 new
            # tokens and an utterance ends after a few hundred): one read-back per EOS_CHECK_EVERY replays, as XY generate does per 8 frames
            if self.eos is not None and ran < self.steps_left and not bool(self.unfinished.any()):
                break
        self.steps_run = ran
        return self.finish()

    EOS_CHECK_EVERY = 16

    def finish(self):
        """After the last replay: bookkeeping, the barrier-timeout check of the step kernel, the ids.  After an early exit (every
        sequence met EOS) the columns that were not run carry the pad id, and `cache.seen_tokens` counts the steps actually run (the
        state has been advanced through at most EOS_CHECK_EVERY - 1 pad tokens past the last EOS)."""
        if self.graph is not None:
            ran = self.steps_left if getattr(self, "steps_run", None) is None else self.steps_run
            self.cache.seen_tokens = self._seen0 + ran
            if ran < self.steps_left:
                self.out[:, 1 + ran:self.max_new_tokens] = self.pad_t
        if self.step is not None and self.step.barrier_timed_out():
            raise _lib.Rwkv7HipError("rwkv7_decode_step_bf16: a grid barrier timed out; the generated ids are invalid")
        return self.out[:, :self.max_new_tokens]

    @torch.no_grad()
    def prepare(self, inputs_embeds=None, input_ids=None, attention_mask=None, max_new_tokens=256,
                eos_token_id: Optional[int] = None, pad_token_id: Optional[int] = None,
                suppress_tokens: Optional[Sequence[int]] = None, do_sample: bool = False, temperature: float = 1.0,
                top_k: int = 0, top_p: float = 1.0, seed: Optional[int] = None, min_new_tokens: int = 0):
        """Prefill, first token, and the capture of one decode step on this decoder's static buffers (on the current stream).
        Afterwards `self.graph.replay()` advances every sequence by one token, `self.steps_left` times for max_new_tokens, and
        `finish()` returns the ids."""
        self.do_sample, self.temperature, self.top_k, self.top_p = bool(do_sample), float(temperature), int(top_k or 0), float(top_p)
        self.max_new_tokens, self.steps_left, self.graph = max_new_tokens, 0, None
        self.steps_run = None    # None = "every replay ran" (prepare + manual replays + finish); generate() / MultiGroupDecoder set it
        self.min_new_tokens = int(min_new_tokens or 0)   # the EOS id cannot be drawn before that many tokens (fused sampler only)
        if seed is not None:
            torch.cuda.manual_seed(seed)
        self._seed = seed   # None: one fresh key per generation from torch's default generator
        self.sampler, self._try_fused, self.tail = None, self.fused_sampling, None
        m = self.model
        dev = m.device
        B = self.B
        self.cache = Cache.zeros(m.config, B, dev, m.dtype)
        self.suppress = None if not suppress_tokens else torch.tensor(list(suppress_tokens), device=dev)
        self.eos = None if eos_token_id is None else torch.tensor(eos_token_id, device=dev)
        self.pad_t = torch.tensor(pad_token_id if pad_token_id is not None else (eos_token_id or 0), device=dev)
        self.unfinished = torch.ones(B, dtype=torch.bool, device=dev)
        # >= 3 columns: the two un-captured warm-up steps below scatter into columns 1 and 2 before the state is restored
        self.out = torch.zeros(B, max(max_new_tokens, 3), dtype=torch.long, device=dev)
        self.pos = torch.zeros(1, 1, dtype=torch.long, device=dev)
        # prefill (eager, state-carrying kernel over the whole prompt) + first token
        o = m(input_ids=input_ids if inputs_embeds is None else None, inputs_embeds=inputs_embeds,
              attention_mask=attention_mask, past_key_values=self.cache, use_cache=True, logits_to_keep=1)
        logits = o.logits[:, -1].float()
        first = self._pick(logits)
        self.ids = first.clone()
        self.out[:, 0] = first
        self.pos.fill_(1)
        if self.eos is not None:
            self.unfinished &= first != self.eos
        self.step = None
        self._seen0 = self.cache.seen_tokens
        if max_new_tokens <= 1:
            return self
        if self.step_kernel is not False:
            why = DecodeStep.supported(m.model, m.lm_head, self.cache)
            if why is None:
                self.step = DecodeStep(m.model, m.lm_head, self.cache)
            elif self.step_kernel:
                raise ValueError("persistent decode step unavailable: " + why)
        emb_w = m.get_input_embeddings().weight
        if (self.step is not None and self.sampler is not None and emb_w.dtype == torch.bfloat16 and emb_w.is_contiguous()
                and emb_w.shape[1] % 8 == 0):
            # the loop's handling of the drawn id rides in the draw's launch (csrc/sampling.hip SmpTail)
            from .sampling import SampleTail
            self.x_next = F.embedding(self.ids, emb_w).contiguous()
            self._raw = torch.empty(B, 1, dtype=torch.long, device=dev)
            self.tail = SampleTail.make(self.ids, self.out, self.unfinished if self.eos is not None else None,
                                        None if self.eos is None else int(self.eos), int(self.pad_t), emb_w.detach(), self.x_next)
        # capture one step on the live state tensors.  Warm-up (un-captured) steps would advance the state, so the
        # state/ids are snapshotted and restored around them.
        snap = [(s.att_x_prev.clone(), s.att_kv.clone(), s.ffn_x_prev.clone()) for s in self.cache.states]
        ids0, pos0, out0, unf0, seen0 = self.ids.clone(), self.pos.clone(), self.out.clone(), self.unfinished.clone(), self.cache.seen_tokens
        x0 = self.x_next.clone() if self.tail is not None else None
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                self._step()
        torch.cuda.current_stream().wait_stream(side)

        def restore():
            for s, (a, kv, f) in zip(self.cache.states, snap):
                s.att_x_prev.copy_(a)
                s.att_kv.copy_(kv)
                s.ffn_x_prev.copy_(f)
            self.ids.copy_(ids0)
            if x0 is not None:
                self.x_next.copy_(x0)
            self.pos.copy_(pos0)
            self.out.copy_(out0)
            self.unfinished.copy_(unf0)
            self.cache.seen_tokens = seen0

        restore()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self._step()
        restore()  # capture does not execute, but keep the bookkeeping identical either way
        self.steps_left = max_new_tokens - 1
        return self


class MultiGroupDecoder:
    """More sequences than one step kernel launch covers (B <= 32), and more of the GPU than one group uses: a decode step is a
    chain of ~170 latency-bound launches that leaves most of the chip idle, so k independent groups of up to 32 sequences -- each
    a GraphDecoder with its own recurrent state, static buffers and captured step -- are replayed round-robin on k streams and
    overlap (measured on MI355X, 0.4B, greedy: round 2 kernels: 1 x 32 sequences 27.8 k tokens/s, 2 x 32 41 k, 8 x 32 44.5 k; round 3: 1 x 32 = 32.7 k).  The serving shape
    of SURVEY 8f N3 ("persistent multi-request decode"): the reference runs one engine thread per request on a side stream
    (service/tts_service.py:42-60, cosyvoice/cli/model.py:64,147-169).

    generate() takes the same arguments as GraphDecoder.generate with a leading batch of any size; the batch is cut into groups of
    `group_size` in order, and the ids come back in that order.  Groups are independent, so every sequence gets exactly the ids a
    GraphDecoder run on its group alone gives (tests/test_decode_step_gpu.py)."""

    def __init__(self, model, group_size: int = 32, step_kernel: Optional[bool] = None):
        if not 1 <= group_size <= 32:
            raise ValueError("group_size must be 1..32 (the step kernel's batch)")
        self.model, self.group_size, self.step_kernel = model.eval(), group_size, step_kernel
        self.decoders, self.streams = [], []

    @torch.no_grad()
    def generate(self, inputs_embeds=None, input_ids=None, attention_mask=None, max_new_tokens=256, **kw):
        lead = inputs_embeds if inputs_embeds is not None else input_ids
        B = lead.shape[0]
        cuts = list(range(0, B, self.group_size))
        sl = lambda t, a: None if t is None else t[a:a + self.group_size]
        main = torch.cuda.current_stream()
        self.decoders = [GraphDecoder(self.model, min(self.group_size, B - a), self.step_kernel) for a in cuts]
        for gi, d in enumerate(self.decoders):
            d.seed_offset = gi
        while len(self.streams) < len(cuts):
            self.streams.append(torch.cuda.Stream())
        # model.eval() in GraphDecoder.__init__ reset the lazily built parameter caches (backbone._stacked_mix): rebuild them HERE, on
        # `main`, so that no group's stream reads what another group's stream is still writing (each group waits only on `main`)
        for mod in self.model.modules():
            if hasattr(mod, "_stacked_mix"):
                mod._stacked_mix(self.model.dtype)
        seed = kw.get("seed", None)   # handed to every group: their draws differ by seed_offset
        if seed is not None:
            torch.cuda.manual_seed(seed)
        for d, st, a in zip(self.decoders, self.streams, cuts):
            st.wait_stream(main)
            with torch.cuda.stream(st):   # prefill + capture of each group on its own stream
                d.prepare(inputs_embeds=sl(inputs_embeds, a), input_ids=sl(input_ids, a), attention_mask=sl(attention_mask, a),
                          max_new_tokens=max_new_tokens, **kw)
        live = [d for d in self.decoders if d.graph is not None]
        for d in live:
            d.steps_run = 0
        for it in range(max(0, max_new_tokens - 1)):
            for d, st in zip(self.decoders, self.streams):
                if d in live:
                    with torch.cuda.stream(st):
                        d.graph.replay()
                    d.steps_run += 1
            # groups whose sequences have all met EOS drop out (one read-back per group every EOS_CHECK_EVERY rounds)
            if (it + 1) % GraphDecoder.EOS_CHECK_EVERY == 0 and live and live[0].eos is not None:
                still = []
                for d, st in zip(self.decoders, self.streams):
                    if d in live:
                        # reduction AND read-back on the stream that writes the flag (the copy to the host is ordered behind the
                        # replays queued there so far; on the main stream it would race with them)
                        with torch.cuda.stream(st):
                            if bool(d.unfinished.any()):
                                still.append(d)
                live = still
                if not live:
                    break
        outs = []
        for d, st in zip(self.decoders, self.streams):
            main.wait_stream(st)
            with torch.cuda.stream(st):
                outs.append(d.finish())
        for st in self.streams[:len(cuts)]:
            main.wait_stream(st)
        return torch.cat(outs, 0)
