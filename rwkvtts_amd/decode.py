"""Persistent-state greedy decode with the per-token step captured in a hipGraph (BASELINE.json configs[4];
reference loops: model/llm/rwkv_asr_cuda_whisper.py:694-717, rwkv_s2s_single_ffn.py:417-445, HF generate via
inference/rwkv7speech_inference.py:99-107).

One decode step of the 0.4B model is ~25 small launches per layer; eagerly it is launch-bound (host ~3-4 us per
launch).  The step -- embedding lookup of the previous ids, 24 layers on the in-place recurrent state
(att_x_prev, att_kv, ffn_x_prev per layer), final norm, lm_head, suppress/argmax, bookkeeping -- is recorded once
into a graph on static buffers and replayed per token; the ids never leave the device until the end.
"""
from __future__ import annotations

from typing import Optional, Sequence

import torch

from .backbone import Cache


class GraphDecoder:
    def __init__(self, model, batch_size: int):
        self.model = model.eval()
        self.B = batch_size
        self.graph = None

    @torch.no_grad()
    def _step(self):
        out = self.model(input_ids=self.ids.unsqueeze(1), past_key_values=self.cache, use_cache=True)
        logits = out.logits[:, -1].float()
        if self.suppress is not None:
            logits.index_fill_(1, self.suppress, float("-inf"))
        nxt = torch.argmax(logits, dim=-1)
        if self.eos is not None:
            nxt = torch.where(self.unfinished, nxt, self.pad_t)
            self.unfinished &= nxt != self.eos
        self.ids.copy_(nxt)
        self.out.scatter_(1, self.pos.expand(self.B, 1), nxt.unsqueeze(1))
        self.pos += 1

    @torch.no_grad()
    def generate(self, inputs_embeds=None, input_ids=None, attention_mask=None, max_new_tokens=256,
                 eos_token_id: Optional[int] = None, pad_token_id: Optional[int] = None,
                 suppress_tokens: Optional[Sequence[int]] = None):
        m = self.model
        dev = m.device
        B = self.B
        self.cache = Cache.zeros(m.config, B, dev, m.dtype)
        self.suppress = None if not suppress_tokens else torch.tensor(list(suppress_tokens), device=dev)
        self.eos = None if eos_token_id is None else torch.tensor(eos_token_id, device=dev)
        self.pad_t = torch.tensor(pad_token_id if pad_token_id is not None else (eos_token_id or 0), device=dev)
        self.unfinished = torch.ones(B, dtype=torch.bool, device=dev)
        self.out = torch.zeros(B, max_new_tokens, dtype=torch.long, device=dev)
        self.pos = torch.zeros(1, 1, dtype=torch.long, device=dev)
        # prefill (eager, state-carrying kernel over the whole prompt) + first token
        o = m(input_ids=input_ids if inputs_embeds is None else None, inputs_embeds=inputs_embeds,
              attention_mask=attention_mask, past_key_values=self.cache, use_cache=True, logits_to_keep=1)
        logits = o.logits[:, -1].float()
        if self.suppress is not None:
            logits.index_fill_(1, self.suppress, float("-inf"))
        first = torch.argmax(logits, dim=-1)
        self.ids = first.clone()
        self.out[:, 0] = first
        self.pos.fill_(1)
        if self.eos is not None:
            self.unfinished &= first != self.eos
        if max_new_tokens <= 1:
            return self.out
        # capture one step on the live state tensors.  Warm-up (un-captured) steps would advance the state, so the
        # state/ids are snapshotted and restored around them.
        snap = [(s.att_x_prev.clone(), s.att_kv.clone(), s.ffn_x_prev.clone()) for s in self.cache.states]
        ids0, pos0, out0, unf0, seen0 = self.ids.clone(), self.pos.clone(), self.out.clone(), self.unfinished.clone(), self.cache.seen_tokens
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
            self.pos.copy_(pos0)
            self.out.copy_(out0)
            self.unfinished.copy_(unf0)
            self.cache.seen_tokens = seen0

        restore()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self._step()
        restore()  # capture does not execute, but keep the bookkeeping identical either way
        for _ in range(max_new_tokens - 1):
            self.graph.replay()
        self.cache.seen_tokens = seen0 + max_new_tokens - 1
        return self.out
