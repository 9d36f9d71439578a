"""XY_LM-layout TTS language model on the HIP backbone: drop-in for model/llm/xy_llm.py (RWKV7XYLM).

8 parallel channels per step: channel 0 = text + RVQ-0 (audio ids offset by text_shift_size), channels 1..7 = RVQ-k
delayed by k steps.  Input = SUM of the 8 channel embeddings (padding_idx rows zero, xy_llm.py:162,168,176-187);
8 Linear heads with bias; loss = sum of 8 CrossEntropyLoss(label_smoothing) with ignore_index -100 and NO label shift
inside (labels are pre-shifted by the collator, data/utils/collator.py:75) -- xy_llm.py:189-257.
Generation = CustomGenerationMixin._sample (xy_llm.py:39-146): channel-0 logits masked to the audio range, independent
multinomial per channel, 7-step flush once channel 0 leaves the audio range.

N4: with config.fuse_cross_entropy and model.training, the channel heads go through the chunked
fused linear+CE (the channel-0 logits are [B,T,66661]: 4.4 GB in bf16 at B=4,T=8192) and `logits` is None.
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from .backbone import Cache, ModelOutput, RWKV7Config, RWKV7Model
from .hf_api import HFModelMixin
from .losses import fused_linear_cross_entropy


class RWKV7XYConfig(RWKV7Config):
    """xy_llm.py:17-28."""
    _EXTRA = dict(llm_input_size=None, speech_vocab_size=1024, length_normalized_loss=True, lsm_weight=0.0,
                  num_channels=8, drop_ratio=0.0, speech_pad_token=None, text_shift_size=65536)

    def __init__(self, **kw):
        base = {k: v for k, v in kw.items() if k in RWKV7Config.__dataclass_fields__ and k != "extra"}
        super().__init__(**base)
        for k, dflt in self._EXTRA.items():
            setattr(self, k, kw.get(k, dflt))
        self.llm_input_size = self.llm_input_size or self.hidden_size
        if self.speech_pad_token is None:
            self.speech_pad_token = self.speech_vocab_size - 1
        self.extra = {k: v for k, v in kw.items() if k not in base and k not in self._EXTRA}

    def to_dict(self):
        d = super().to_dict()
        d.update({k: getattr(self, k) for k in self._EXTRA}, architectures=["RWKV7XYLM"])
        return d

    @classmethod
    def from_dict(cls, d):
        return cls(**d)


class RWKV7XYLM(HFModelMixin, nn.Module):
    config_class = RWKV7XYConfig

    def __init__(self, config: RWKV7XYConfig):
        super().__init__()
        self.config = config
        self.model = RWKV7Model(config)
        self.embs = nn.ModuleList()
        self.heads = nn.ModuleList()
        # channel 0: text (+ shifted RVQ-0); channels 1..: speech
        self.embs.append(nn.Embedding(config.vocab_size, config.hidden_size, padding_idx=config.vocab_size - 1))
        self.heads.append(nn.Linear(config.hidden_size, config.vocab_size))
        for _ in range(1, config.num_channels):
            self.embs.append(nn.Embedding(config.speech_vocab_size, config.hidden_size,
                                          padding_idx=config.speech_vocab_size - 1))
            self.heads.append(nn.Linear(config.hidden_size, config.speech_vocab_size))
        self.dropout = nn.Dropout(config.drop_ratio) if config.drop_ratio > 0 else None

    def get_input_embeddings(self):
        return self.embs[0]

    def zero_embs(self):
        """xy_llm.py:176-187."""
        with torch.no_grad():
            self.embs[0].weight[self.config.vocab_size - 1].zero_()
            for i in range(1, self.config.num_channels):
                self.embs[i].weight[self.config.speech_vocab_size - 1].zero_()

    def embed(self, input_ids):
        if input_ids.dim() != 3 or input_ids.shape[2] != self.config.num_channels:
            raise ValueError(f"input_ids must have shape (B, T, num_channels), but got {tuple(input_ids.shape)}")
        x = self.embs[0](input_ids[:, :, 0])
        for i in range(1, self.config.num_channels):
            x = x + self.embs[i](input_ids[:, :, i])
        return x

    def forward(self, input_ids=None, attention_mask=None, inputs_embeds=None, past_key_values: Optional[Cache] = None,
                labels=None, use_cache=None, return_dict=None, **kwargs):
        return_dict = True if return_dict is None else return_dict
        if inputs_embeds is None and input_ids is not None:
            inputs_embeds = self.embed(input_ids)
        if self.dropout is not None:
            inputs_embeds = self.dropout(inputs_embeds)
        outputs = self.model(inputs_embeds=inputs_embeds, attention_mask=attention_mask, past_key_values=past_key_values,
                             use_cache=use_cache)
        hidden = outputs[0]
        fuse = self.config.fuse_cross_entropy and self.training and labels is not None
        total_loss, all_logits = None, None
        if fuse:
            total_loss = 0
            for i in range(self.config.num_channels):
                total_loss = total_loss + fused_linear_cross_entropy(
                    hidden, labels[:, :, i], self.heads[i].weight, self.heads[i].bias, -100,
                    label_smoothing=self.config.lsm_weight)
        else:
            all_logits = [self.heads[i](hidden) for i in range(self.config.num_channels)]
            if labels is not None:
                total_loss = 0
                for i, logits in enumerate(all_logits):
                    total_loss = total_loss + F.cross_entropy(logits.view(-1, logits.shape[-1]).float(),
                                                              labels[:, :, i].reshape(-1),
                                                              label_smoothing=self.config.lsm_weight)
        if not return_dict:
            return ((total_loss,) if total_loss is not None else ()) + (all_logits, outputs.past_key_values)
        return ModelOutput(loss=total_loss, logits=all_logits, past_key_values=outputs.past_key_values,
                           hidden_states=None, attentions=None)

    def is_audio_token(self, token_id):
        c = self.config
        return (token_id >= c.text_shift_size) & (token_id < c.text_shift_size + c.speech_vocab_size)

    @torch.no_grad()
    def generate(self, input_ids, attention_mask=None, max_new_tokens=None, max_length=None, do_sample=True,
                 temperature=1.0, top_k=0, top_p=1.0, eos_token_id=None, generator: Optional[torch.Generator] = None,
                 return_dict_in_generate=False, reference_termination=False, use_graph: Optional[bool] = None, **unused):
        """CustomGenerationMixin._sample (xy_llm.py:39-146) on the persistent-state decode path.
        input_ids [B,T,C].  do_sample=False replaces each multinomial by argmax (greedy, for parity tests).

        Flush (xy_llm.py:104-121): a non-audio id on channel 0 starts a countdown of C-1 further rows; in all C rows channel 0
        carries EOS (if one is configured), channel i keeps its sampled ids for i more rows and is padded afterwards, and the
        sequence finishes with the last row.  Two places where the reference's loop cannot do what its comments say, and
        where this loop follows the comments by default (tests/test_heads_gpu.py hand-steps the rules):
          * `unfinished & ~stopping & ~(needs_additional_steps == -1)` (:140) is false for every sequence that is NOT flushing,
            i.e. from the first step on -- the reference emits ONE frame and stops; here only a flushing sequence whose
            countdown reached -1 finishes;
          * with an EOS id configured, the EOS that the flush writes on channel 0 meets the EOS stopping criterion in the
            very row that starts the countdown (:139); here the criterion is not applied to flushing sequences.
        reference_termination=True reproduces lines :139-140 literally (same ids, same length as the reference on the same
        draws).

        One frame step = embedding sum of the previous row -> the whole stack + the eight heads (one concatenated projection,
        rwkv7_decode_step_bf16) -> channel-0 mask -> eight draws -> flush / pad / stop bookkeeping, all as tensor operations on
        fixed buffers with no host read-back (`_XYFrameState.step`), so the step is captured in a hipGraph and replayed
        (use_graph: None = when the step kernel covers the model, a length bound is given and no private generator is used);
        the host looks at the `all finished` flag only every 8 frames and trims the surplus rows at the end."""
        cfg = self.config
        was_training = self.training
        self.eval()
        B, cur_len, C = input_ids.shape
        total = max_length if max_new_tokens is None else cur_len + max_new_tokens
        eos = None if eos_token_id is None else ([eos_token_id] if isinstance(eos_token_id, int) else list(eos_token_id))
        dev = input_ids.device
        cache = Cache.zeros(cfg, B, dev, self.dtype)
        out = self(input_ids=input_ids, attention_mask=attention_mask, past_key_values=cache, use_cache=True)
        # T = 1 steps: the whole stack + the eight heads (as one concatenated projection) through rwkv7_decode_step_bf16
        # when the model is covered (bf16, B <= 32); else module by module
        step_kernel, sizes = None, [h.weight.shape[0] for h in self.heads]
        st = _XYFrameState(self, input_ids, total, eos, do_sample, top_k, top_p, temperature, generator, reference_termination)
        # Channel 0 may only emit audio ids (xy_llm.py:82-86): with the fused draw (which applies that range itself) the step kernel
        # projects onto the 1 025 audio rows of the 66 661-row head only -- 134 of the 151 MB of head weights are never read
        lo0, hi0 = cfg.text_shift_size, cfg.text_shift_size + cfg.speech_vocab_size
        audio_only = (getattr(self, "use_step_kernel", True) and st.sampler_supported(sizes, getattr(self, "fused_sampling", True))
                      and hi0 <= sizes[0])
        w0, b0 = (self.heads[0].weight[lo0:hi0], self.heads[0].bias[lo0:hi0]) if audio_only else (self.heads[0].weight, self.heads[0].bias)
        if getattr(self, "use_step_kernel", True):
            from types import SimpleNamespace
            from .decode import DecodeStep
            head = SimpleNamespace(weight=torch.cat([w0] + [h.weight for h in self.heads[1:]], 0).contiguous(),
                                   bias=torch.cat([b0] + [h.bias for h in self.heads[1:]], 0).contiguous())
            if DecodeStep.supported(self.model, head, cache) is None:
                step_kernel = DecodeStep(self.model, head, cache)
        audio_only = audio_only and step_kernel is not None
        # the frame's logits as ONE buffer (the step kernel's layout); st.logits are views of its eight segments
        first = [l[:, -1, :].float() for l in out.logits]
        if audio_only:
            first[0] = first[0][:, lo0:hi0]
            st.logits_cat = torch.cat(first, 1).contiguous()
            st.make_sampler(sizes, True, col0=[0] + [hi0 - lo0 + sum(sizes[1:i]) for i in range(1, len(sizes))])
        else:
            st.logits_cat = torch.cat(first, 1).contiguous()
            st.logits = list(torch.split(st.logits_cat, sizes, dim=1))
            st.make_sampler(sizes, getattr(self, "fused_sampling", True))

        embed_k = None   # the embedding sum of the previous row as one launch (bf16 tables)
        if step_kernel is not None and getattr(self, "fused_frame", True) and input_ids.dtype == torch.int64:
            from .sampling import XYEmbed
            tables = [e.weight for e in self.embs]
            if XYEmbed.supported(tables):
                embed_k = XYEmbed(tables, B)

        def advance():   # the model step that follows a frame: logits of the next one
            if step_kernel is not None:
                x = embed_k(st.row) if embed_k is not None else self.embed(st.row.unsqueeze(1))[:, 0].contiguous()
                lg = step_kernel(x)
                if st.sampler is not None:
                    st.logits_cat = lg          # the fused draw reads the step kernel's own buffer
                else:
                    st.logits_cat.copy_(lg)     # st.logits are views of this buffer
            else:
                o = self(input_ids=st.row.unsqueeze(1), past_key_values=cache, use_cache=True)
                for dst, l in zip(st.logits, o.logits):
                    dst.copy_(l[:, -1, :].float())

        graph_ok = step_kernel is not None and total is not None and generator is None
        if use_graph is None:
            use_graph = graph_ok
        elif use_graph and not graph_ok:
            raise ValueError("use_graph=True needs the step kernel (bf16, B <= 32), a length bound and the default generator")
        frames = 0
        if not use_graph:
            while True:
                st.step()
                frames += 1
                cache.seen_tokens += 1
                if bool(st.all_done.item()):
                    break
                if st.need_rows(frames + 1):
                    st.grow()
                advance()
        else:
            # frame 0 eagerly (its logits come from the prefill), then `advance(); step()` as ONE captured unit
            st.step()
            frames = 1
            if not bool(st.all_done.item()):
                snap = [(s.att_x_prev.clone(), s.att_kv.clone(), s.ffn_x_prev.clone()) for s in cache.states]
                keep = st.snapshot()
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):   # warm-up outside the capture (allocations, lazy init), state restored afterwards
                    advance()
                    st.step()
                torch.cuda.current_stream().wait_stream(side)

                def restore():
                    for s_, (a_, kv_, f_) in zip(cache.states, snap):
                        s_.att_x_prev.copy_(a_)
                        s_.att_kv.copy_(kv_)
                        s_.ffn_x_prev.copy_(f_)
                    st.restore(keep)

                restore()
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    advance()
                    st.step()
                restore()
                budget = total - cur_len - 1
                while budget > 0:
                    n = min(8, budget)
                    for _ in range(n):
                        graph.replay()
                    budget -= n
                    frames += n
                    if bool(st.all_done.item()):
                        break
                if step_kernel.barrier_timed_out():
                    from . import _lib
                    raise _lib.Rwkv7HipError("rwkv7_decode_step_bf16: a grid barrier timed out; the generated ids are invalid")
            cache.seen_tokens += frames
        n_rows = int(st.n_rows.item())   # rows the reference's loop would have appended before its `break`
        input_ids = st.out[:, :n_rows].clone()
        if was_training:
            self.train()
        if return_dict_in_generate:
            return ModelOutput(sequences=input_ids)
        return input_ids


class _XYFrameState:
    """Everything CustomGenerationMixin._sample (xy_llm.py:39-146) carries from frame to frame, as device tensors at fixed
    addresses, and one frame of its bookkeeping as tensor operations only (no `.any()`, no boolean-mask assignment, no
    `.item()`): `step()` consumes `logits` (one tensor per channel), writes the new row into `out` and `row`, and updates the
    countdown / unfinished / all_done / n_rows tensors -- the same code runs eagerly and inside a captured hipGraph."""

    def __init__(self, model, input_ids, total, eos, do_sample, top_k, top_p, temperature, generator, reference_termination):
        cfg = model.config
        self.cfg, self.model = cfg, model
        B, cur_len, C = input_ids.shape
        dev = input_ids.device
        self.B, self.C, self.total, self.cur_len = B, C, total, cur_len
        self.sample = dict(do_sample=do_sample, top_k=top_k, top_p=top_p, temperature=temperature, generator=generator)
        self.reference_termination = bool(reference_termination)
        self.eos = None if eos is None else torch.tensor(eos, device=dev)
        self.eos0 = None if eos is None else int(eos[0])
        rows = total if total is not None else cur_len + 256
        self.out = torch.zeros(B, max(rows, cur_len + 1), C, dtype=input_ids.dtype, device=dev)
        self.out[:, :cur_len] = input_ids
        self.row = torch.zeros(B, C, dtype=input_ids.dtype, device=dev)          # the row written last (input of the next model step)
        self.pos = torch.full((1,), cur_len, dtype=torch.long, device=dev)       # where the next row goes
        self.unfinished = torch.ones(B, dtype=torch.long, device=dev)
        self.needs = -torch.ones(B, dtype=torch.long, device=dev)               # needs_additional_steps
        self.all_done = torch.zeros((), dtype=torch.bool, device=dev)            # the reference's `unfinished.max() == 0` -> break
        self.n_rows = torch.full((), cur_len, dtype=torch.long, device=dev)      # length of the sequence tensor at that break
        self.ch0_block = torch.ones(cfg.vocab_size, dtype=torch.bool, device=dev)
        self.ch0_block[cfg.text_shift_size: cfg.text_shift_size + cfg.speech_vocab_size] = False
        self.logits: List[torch.Tensor] = []
        self.logits_cat: Optional[torch.Tensor] = None
        self.sampler = None
        # the frame's bookkeeping as one launch (csrc/sampling.hip xy_frame_kernel) instead of the ~45 tensor operations below
        self.fused_frame = getattr(model, "fused_frame", True) and input_ids.dtype == torch.int64 and input_ids.is_cuda and B <= 64

    def _sampler_args(self, sizes):
        cfg, sm = self.cfg, self.sample
        allow = [(cfg.text_shift_size, cfg.text_shift_size + cfg.speech_vocab_size)] + [(0, n) for n in sizes[1:]]
        return allow, dict(do_sample=sm["do_sample"], top_k=sm["top_k"], top_p=sm["top_p"], temperature=sm["temperature"])

    def sampler_supported(self, sizes, enabled=True) -> bool:
        from . import spark_llm
        from .sampling import RowSampler
        if not enabled or self.sample["generator"] is not None or spark_llm.sample_next is not spark_llm._STOCK_SAMPLE_NEXT:
            return False
        allow, kw = self._sampler_args(sizes)
        return RowSampler.supported(self.out.device, sizes, allow, None, **kw) is None

    def make_sampler(self, sizes, enabled=True, col0=None):
        """The eight draws of a frame as ONE launch (csrc/sampling.hip: channel 0 restricted to the audio ids, xy_llm.py:82-86) when
        the request is covered -- default generator, the stock sample_next, top-k <= 64 ...; as torch operations the eight warper
        chains are ~120 launches, 1.8 ms of a 3.1 ms frame.  col0: the column at which each channel's logits start when the row
        holds only the audio range of channel 0 (ids keep their vocabulary values)."""
        from .sampling import RowSampler
        if not self.sampler_supported(sizes, enabled):
            return
        allow, kw = self._sampler_args(sizes)
        seg_off = None if col0 is None else [c - a[0] for c, a in zip(col0, allow)]
        self.sampler = RowSampler(self.out.device, sizes, allow, None, seg_off=seg_off, **kw)

    def need_rows(self, frames_after):
        return self.cur_len + frames_after > self.out.shape[1]

    def grow(self):   # unbounded generation (no max_length): the row buffer grows in blocks (host-side decision, no read-back)
        self.out = torch.cat([self.out, torch.zeros_like(self.out[:, :256])], 1)

    def snapshot(self):
        return [t.clone() for t in (self.out, self.row, self.pos, self.unfinished, self.needs, self.all_done, self.n_rows)] + \
               [self.logits_cat.clone()]

    def restore(self, keep):
        for t, k in zip([self.out, self.row, self.pos, self.unfinished, self.needs, self.all_done, self.n_rows, self.logits_cat], keep):
            t.copy_(k)

    def step(self):
        from .spark_llm import sample_next
        cfg, C = self.cfg, self.C
        pad = cfg.speech_pad_token
        running = (~self.all_done).long()            # 0 once the reference's loop would have left: later frames change nothing
        if self.sampler is not None:
            nt = self.sampler(self.logits_cat, self.pos).to(self.row.dtype)
        else:
            lg0 = self.logits[0].masked_fill(self.ch0_block, float("-inf"))   # channel 0 may only emit audio ids (:82-86)
            toks = [sample_next(lg0, **self.sample)] + [sample_next(l, **self.sample) for l in self.logits[1:]]
            nt = torch.stack(toks, -1)
        if self.fused_frame:
            from .sampling import xy_frame_step
            xy_frame_step(nt.contiguous(), self.out, self.row, self.pos, self.unfinished, self.needs, self.all_done, self.n_rows,
                          cfg.text_shift_size, cfg.speech_vocab_size, pad, self.eos0, self.total, self.eos, self.reference_termination)
            return
        is_audio = self.model.is_audio_token(nt[:, 0])
        to_flush = (~is_audio) & (self.needs < 0)
        needs = torch.where(to_flush, torch.full_like(self.needs, C - 1), self.needs)
        is_flushing = needs >= 0
        cols = [nt[:, 0] if self.eos0 is None else torch.where(is_flushing, torch.full_like(nt[:, 0], self.eos0), nt[:, 0])]
        for i in range(1, C):
            pad_this = is_flushing & (needs < C - i)
            cols.append(torch.where(pad_this, torch.full_like(nt[:, i], pad), nt[:, i]))
        pddp_text = self.eos0 if self.eos0 is not None else 0
        u = self.unfinished
        cols[0] = cols[0] * u + pddp_text * (1 - u)
        for i in range(1, C):
            cols[i] = cols[i] * u + pad * (1 - u)
        row = torch.stack(cols, -1)
        # append the row (where the loop is still running) and remember it as the next model input
        idx = self.pos.clamp(max=self.out.shape[1] - 1).view(1, 1, 1).expand(self.B, 1, C)
        cur = torch.gather(self.out, 1, idx)
        self.out.scatter_(1, idx, torch.where(running.bool().view(1, 1, 1), row.unsqueeze(1), cur))
        self.row.copy_(row)
        self.pos += running
        self.n_rows += running
        needs = torch.where(is_flushing, needs - 1, needs)
        stop = torch.zeros(self.B, dtype=torch.bool, device=row.device)
        if self.total is not None:
            stop = stop | (self.pos >= self.total)
        if self.eos is not None:
            hit = torch.isin(row[:, 0], self.eos)
            # default: the EOS criterion applies to sequences that are NOT flushing (the flush itself writes EOS on channel 0)
            stop = stop | (hit if self.reference_termination else (hit & ~is_flushing))
        if self.reference_termination:
            u = u & (~stop).long() & (~(needs == -1)).long()                       # xy_llm.py:140, literally
        else:
            u = u & (~stop).long() & (~(needs == -1) | (~is_flushing)).long()
        keep = running.bool()
        self.unfinished.copy_(torch.where(keep, u, self.unfinished))
        self.needs.copy_(torch.where(keep, needs, self.needs))
        self.all_done.copy_(self.all_done | (self.unfinished.max() == 0))
