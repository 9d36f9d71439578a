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
                 return_dict_in_generate=False, **unused):
        """CustomGenerationMixin._sample (xy_llm.py:39-146) on the persistent-state decode path.
        input_ids [B,T,C].  do_sample=False replaces each multinomial by argmax (greedy, for parity tests).

        Flush (xy_llm.py:104-121): a non-audio id on channel 0 starts a countdown of C-1 further rows; in all C rows channel 0
        carries EOS (if one is configured), channel i keeps its sampled ids for i more rows and is padded afterwards, and the
        sequence finishes with the last row.  Two places where the reference's loop cannot do what its comments say, and
        where this loop follows the comments (tests/test_heads_gpu.py hand-steps the rules):
          * `unfinished & ~stopping & ~(needs_additional_steps == -1)` (:140) is false for every sequence that is NOT flushing,
            i.e. from the first step on -- the reference would emit one frame; here only a flushing sequence whose countdown
            reached -1 finishes;
          * with an EOS id configured, the EOS that the flush writes on channel 0 meets the EOS stopping criterion in the
            very row that starts the countdown (:139); here the criterion is not applied to flushing sequences."""
        from .spark_llm import sample_next
        cfg = self.config
        was_training = self.training
        self.eval()
        B, cur_len, C = input_ids.shape
        total = max_length if max_new_tokens is None else cur_len + max_new_tokens
        eos = None if eos_token_id is None else ([eos_token_id] if isinstance(eos_token_id, int) else list(eos_token_id))
        dev = input_ids.device
        unfinished = torch.ones(B, dtype=torch.long, device=dev)
        needs_additional_steps = -torch.ones(B, dtype=torch.long, device=dev)
        cache = Cache.zeros(cfg, B, dev, self.dtype)
        out = self(input_ids=input_ids, attention_mask=attention_mask, past_key_values=cache, use_cache=True)
        # T = 1 steps: the whole stack + the eight heads (as one concatenated projection) through rwkv7_decode_step_bf16
        # when the model is covered (bf16, B <= 32); else module by module
        step_kernel, sizes = None, [h.weight.shape[0] for h in self.heads]
        if getattr(self, "use_step_kernel", True):
            from types import SimpleNamespace
            from .decode import DecodeStep
            head = SimpleNamespace(weight=torch.cat([h.weight for h in self.heads], 0).contiguous(),
                                   bias=torch.cat([h.bias for h in self.heads], 0).contiguous())
            if DecodeStep.supported(self.model, head, cache) is None:
                step_kernel = DecodeStep(self.model, head, cache)
        while True:
            logits = [l[:, -1, :].clone().float() for l in out.logits]
            mask = torch.ones_like(logits[0], dtype=torch.bool)
            mask[:, cfg.text_shift_size: cfg.text_shift_size + cfg.speech_vocab_size] = False
            logits[0].masked_fill_(mask, float("-inf"))  # channel 0 may only emit audio ids (:82-86)
            next_tokens = torch.stack([sample_next(l, do_sample, top_k, top_p, temperature, generator) for l in logits], -1)
            is_audio = self.is_audio_token(next_tokens[:, 0])
            to_flush = (~is_audio) & (needs_additional_steps < 0)
            needs_additional_steps[to_flush] = C - 1
            is_flushing = needs_additional_steps >= 0
            if is_flushing.any():
                if eos is not None:
                    next_tokens[is_flushing, 0] = eos[0]
                for i in range(1, C):
                    pad_this = is_flushing & (needs_additional_steps < C - i)
                    next_tokens[pad_this, i] = cfg.speech_pad_token
            pddp_text = eos[0] if eos is not None else 0
            next_tokens[:, 0] = next_tokens[:, 0] * unfinished + pddp_text * (1 - unfinished)
            next_tokens[:, 1:] = next_tokens[:, 1:] * unfinished.unsqueeze(-1) + cfg.speech_pad_token * (1 - unfinished.unsqueeze(-1))
            input_ids = torch.cat([input_ids, next_tokens[:, None, :]], dim=1)
            needs_additional_steps[is_flushing] -= 1
            stop = torch.zeros(B, dtype=torch.bool, device=dev)
            if total is not None and input_ids.shape[1] >= total:
                stop[:] = True
            if eos is not None:
                # the EOS criterion applies to sequences that are NOT flushing: the flush itself writes EOS on channel 0, and
                # letting that end the sequence would cut the countdown after its first row (see the docstring)
                stop |= torch.isin(input_ids[:, -1, 0], torch.tensor(eos, device=dev)) & ~is_flushing
            unfinished = unfinished & (~stop).long() & (~(needs_additional_steps == -1) | (~is_flushing)).long()
            if unfinished.max() == 0:
                break
            if step_kernel is not None:
                lg = step_kernel(self.embed(next_tokens[:, None, :])[:, 0].contiguous())
                cache.seen_tokens += 1
                out = ModelOutput(logits=[l.unsqueeze(1) for l in torch.split(lg, sizes, dim=1)])
            else:
                out = self(input_ids=next_tokens[:, None, :], past_key_values=cache, use_cache=True)
        if was_training:
            self.train()
        if return_dict_in_generate:
            return ModelOutput(sequences=input_ids)
        return input_ids
