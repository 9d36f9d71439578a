"""Cosy-2.0-layout TTS language model on the HIP backbone: drop-in for model/llm/cosy_llm.py (RWKV7CosyLM)
and the nn.Module wrapper model/llm/llm.py (RWKV7LM).

Layout (cosy_llm.py:64-73,98-130 / llm.py:73-83,101-130): one sample is the embedding sequence
    [sos_eos, text..., task_id, speech...]      (llm_embedding rows 0/1, text_embedding, speech_embedding)
right-padded with the VALUE -1 (pad_sequence(..., padding_value=IGNORE_ID), cosy_llm.py:71) and masked;
target = [-1 x (2 + text_len), speech..., EOS = speech_token_size], shifted by one (lm_target[:, 1:]);
loss = LabelSmoothingLoss (KL form, cosyvoice/transformer/label_smoothing_loss.py:68-96), accuracy = th_accuracy.
"""
from __future__ import annotations

import time
from typing import Callable, List, Optional

import torch
import torch.nn as nn
from torch.nn.utils.rnn import pad_sequence

from .backbone import Cache, ModelOutput, RWKV7Config, RWKV7Model
from .hf_api import HFModelMixin
from .losses import label_smoothing_kl, th_accuracy

IGNORE_ID = -1  # cosyvoice/utils/common.py:24


class RWKV7CosyConfig(RWKV7Config):
    """cosy_llm.py:13-22."""
    _EXTRA = dict(llm_input_size=None, llm_output_size=None, speech_token_size=6561, length_normalized_loss=True,
                  lsm_weight=0.0, mix_ratio=(5, 15), drop_ratio=0.0)

    def __init__(self, **kw):
        base = {k: v for k, v in kw.items() if k in RWKV7Config.__dataclass_fields__ and k != "extra"}
        super().__init__(**base)
        for k, dflt in self._EXTRA.items():
            setattr(self, k, kw.get(k, dflt))
        self.llm_input_size = self.llm_input_size or self.hidden_size
        self.llm_output_size = self.llm_output_size or self.hidden_size
        self.mix_ratio = list(self.mix_ratio)
        self.extra = {k: v for k, v in kw.items() if k not in base and k not in self._EXTRA}

    def to_dict(self):
        d = super().to_dict()
        d.update({k: getattr(self, k) for k in self._EXTRA}, architectures=["RWKV7CosyLM"])
        return d

    @classmethod
    def from_dict(cls, d):
        return cls(**d)


def ras_sampling(weighted_scores, decoded_tokens, sampling, top_p=0.8, top_k=25, win_size=10, tau_r=0.1,
                 generator=None):
    """Repetition-aware sampling (cosyvoice/utils/common.py:109-137): nucleus (top-p AND top-k) sample; if it
    repeats >= win_size*tau_r times in the last win_size tokens, resample from the full distribution."""
    top_ids = nucleus_sampling(weighted_scores, top_p, top_k, generator)
    rep_num = (torch.tensor(decoded_tokens[-win_size:], device=weighted_scores.device) == top_ids).sum().item()
    if rep_num >= win_size * tau_r:
        top_ids = weighted_scores.softmax(dim=0).multinomial(1, replacement=True, generator=generator)
    return top_ids


def ras_sampling_device(logp, recent, ignore_eos, eos: int, top_p=0.8, top_k=25, win_size=10, tau_r=0.1):
    """ras_sampling + the EOS rejection of sampling_ids (cosyvoice/utils/common.py:109-137, llm.py:160-176) as tensor operations
    only -- no `.item()`, no data-dependent slicing -- so that the draw can live inside a captured decode step:
      nucleus  : the sorted prefix with cum_prob (before adding) < top_p and rank < top_k, sampled by multinomial over the masked
                 probabilities (the same distribution as multinomial over the slice);
      repeat   : count of the candidate in the last win_size emitted ids (`recent`, a device ring filled with -1), and if it is
                 >= win_size * tau_r the draw from the full distribution instead (both draws are always made; one is selected);
      EOS      : while `ignore_eos` (device bool) the reference rejects draws that are EOS and draws again -- the distribution of
                 the accepted draw is the candidate set's distribution with EOS removed, which is what is sampled here directly
                 (if the nucleus is EOS alone: the full distribution without EOS, where the reference would give up after 100
                 rejections).
    Different consumption of the random stream (the pinned id-for-id comparison with the reference's functions,
    tests/test_sampling.py, is on the host pair).  Distribution: identical to the host functions when `ignore_eos` is off or EOS has
    no mass; while `ignore_eos` is on it is an APPROXIMATION of the reference's rejection loop -- the reference re-runs the WHOLE
    sampler (nucleus draw, repeat check, full-distribution fallback) after an EOS result, so its accepted law is proportional to
    N(x)[x not repeated] + N(R)F(x) over x != EOS renormalised ONCE (N nucleus law, F full law, R the repeated ids), while removing
    EOS per stage gives N'(x)[x not repeated] + N'(R)F(x)/(1 - F(EOS)).  The two differ only when a repeat fallback is possible AND
    F(EOS) > 0, by a factor (1 - F(EOS))^-1 on the fallback term relative to the nucleus term: total variation
    <= N(R) F(EOS) / (1 - N(EOS)) (tests/test_sampling.py::test_eos_rejection_laws_reference_vs_per_stage_bound_and_host_sampler)."""
    probs = logp.softmax(dim=0)
    sv, si = probs.sort(descending=True, stable=True)
    cum_before = torch.cumsum(sv, 0) - sv
    rank = torch.arange(sv.numel(), device=sv.device)   # (no host-built tensors in here: the function runs under graph capture)
    keep = ((cum_before < top_p) & (rank < top_k)).long().cumprod(0).to(sv.dtype)
    is_eos = si == eos
    pk = sv * keep
    pk_no = pk.masked_fill(is_eos, 0.0)
    pk_no = torch.where(pk_no.sum() > 0, pk_no, sv.masked_fill(is_eos, 0.0))
    cand = si[torch.where(ignore_eos, pk_no, pk).multinomial(1, replacement=True)]
    full_p = torch.where(ignore_eos, probs.masked_fill(rank == eos, 0.0), probs)
    full = full_p.multinomial(1, replacement=True)
    rep = (recent == cand).sum()
    return torch.where(rep >= win_size * tau_r, full, cand)


def nucleus_sampling(weighted_scores, top_p=0.8, top_k=25, generator=None):
    sorted_value, sorted_idx = weighted_scores.softmax(dim=0).sort(descending=True, stable=True)
    # keep while cum_prob (before adding) < top_p and fewer than top_k kept (common.py:121-128), vectorised
    cum_before = torch.cumsum(sorted_value, 0) - sorted_value
    keep = (cum_before < top_p) & (torch.arange(sorted_value.numel(), device=sorted_value.device) < top_k)
    n = int(keep.long().cumprod(0).sum().item())
    prob, indices = sorted_value[:n], sorted_idx[:n]
    return indices[prob.multinomial(1, replacement=True, generator=generator)]


class RWKV7CosyLM(HFModelMixin, nn.Module):
    config_class = RWKV7CosyConfig

    def __init__(self, config: RWKV7CosyConfig):
        super().__init__()
        self.config = config
        self.model = RWKV7Model(config)
        self.sos_eos, self.task_id, self.fill_token = 0, 1, 2
        self.llm_embedding = nn.Embedding(2, config.llm_input_size)
        self.text_embedding = nn.Embedding(config.vocab_size, config.llm_input_size)
        self.speech_embedding = nn.Embedding(config.speech_token_size + 1, config.llm_input_size)
        self.lm_head = nn.Linear(config.hidden_size, config.speech_token_size + 1)
        self.dropout = nn.Dropout(config.drop_ratio) if config.drop_ratio > 0 else None
        self.sampling: Optional[Callable] = ras_sampling
        self.mix_ratio = config.mix_ratio
        self.speech_token_size = config.speech_token_size

    def criterion_ce(self, logits, target):
        return label_smoothing_kl(logits, target, self.speech_token_size + 1, IGNORE_ID, self.config.lsm_weight,
                                  self.config.length_normalized_loss)

    def pad_unpad_sequence(self, sos_eos_emb, text_token, text_token_len, task_id_emb, speech_token, speech_token_len):
        """Per sample [sos, text[:n_t], task_id, speech[:n_s]], right-padded to the longest sample.  Behaviour of
        cosy_llm.py:64-73: the padding VALUE in the embedding tensor is -1 (IGNORE_ID), the mask is int32 ones over the real
        length.  Built into preallocated tensors (one slice write per piece) rather than through unpad/pad_sequence."""
        B, D = text_token.shape[0], text_token.shape[-1]
        n_t = [int(n) for n in text_token_len.tolist()]
        n_s = [int(n) for n in speech_token_len.tolist()]
        total = [2 + a + b for a, b in zip(n_t, n_s)]
        L = max(total)
        lm_input = text_token.new_full((B, L, D), float(IGNORE_ID))
        attention_mask = torch.zeros(B, L, dtype=torch.int32, device=text_token.device)
        sos, task = sos_eos_emb.reshape(D), task_id_emb.reshape(D)
        for i in range(B):
            row = lm_input[i]
            row[0] = sos
            row[1:1 + n_t[i]] = text_token[i, :n_t[i]]
            row[1 + n_t[i]] = task
            row[2 + n_t[i]:total[i]] = speech_token[i, :n_s[i]]
            attention_mask[i, :total[i]] = 1
        return lm_input, attention_mask

    def build_inputs(self, batch):
        """cosy_llm.py:92-120: batch dict -> (inputs_embeds, attention_mask, labels)."""
        text_token, text_token_len = batch["text_token"], batch["text_token_len"]
        speech_token, speech_token_len = batch["speech_token"], batch["speech_token_len"]
        lm_target = [torch.tensor([IGNORE_ID] * (2 + int(text_token_len[i])) +
                                  speech_token[i, :int(speech_token_len[i])].tolist() + [self.speech_token_size])
                     for i in range(text_token.size(0))]
        lm_target = pad_sequence(lm_target, batch_first=True, padding_value=IGNORE_ID).to(text_token.device)
        text_emb = self.text_embedding(text_token)
        sos_eos_emb = self.llm_embedding.weight[self.sos_eos].reshape(1, 1, -1)
        task_id_emb = self.llm_embedding.weight[self.task_id].reshape(1, 1, -1)
        speech_emb = self.speech_embedding(speech_token)
        inputs_embeds, attention_mask = self.pad_unpad_sequence(sos_eos_emb, text_emb, text_token_len, task_id_emb,
                                                                speech_emb, speech_token_len)
        return inputs_embeds, attention_mask, lm_target[:, 1:].contiguous()

    def forward(self, input_ids=None, attention_mask=None, inputs_embeds=None, past_key_values: Optional[Cache] = None,
                labels=None, use_cache=None, return_dict=None, max_tokens_k: Optional[int] = None, **kwargs):
        """cosy_llm.py:75-160, incl. the `batch=` form and the max_tokens_k batch slicing (:122-130)."""
        return_dict = True if return_dict is None else return_dict
        if "batch" in kwargs:
            inputs_embeds, attention_mask, labels = self.build_inputs(kwargs["batch"])
            if self.dropout is not None:
                inputs_embeds = self.dropout(inputs_embeds)
            if max_tokens_k is not None:
                max_tokens = max_tokens_k * 1024
                bsz, seq_len, _ = inputs_embeds.shape
                max_bsz = max_tokens // seq_len
                if max_bsz < bsz:
                    inputs_embeds, labels, attention_mask = inputs_embeds[:max_bsz], labels[:max_bsz], attention_mask[:max_bsz]
        outputs = self.model(input_ids=input_ids, attention_mask=attention_mask, inputs_embeds=inputs_embeds,
                             past_key_values=past_key_values, use_cache=use_cache)
        logits = self.lm_head(outputs[0])
        loss = self.criterion_ce(logits, labels) if labels is not None else None
        if not return_dict:
            return ((loss,) if loss is not None else ()) + (logits, outputs.past_key_values)
        return ModelOutput(loss=loss, logits=logits, past_key_values=outputs.past_key_values, hidden_states=None,
                           attentions=None)

    def forward_one_step(self, xs, masks, cache=None):
        """cosy_llm.py:274-286 / llm.py:146-158."""
        input_masks = masks[:, -1, :]
        outs = self.model(inputs_embeds=xs, attention_mask=input_masks, use_cache=True, past_key_values=cache)
        return self.lm_head(outs.last_hidden_state), outs.past_key_values

    def sampling_ids(self, weighted_scores, decoded_tokens: List, sampling: int, ignore_eos: bool = True):
        """One draw from self.sampling; while EOS must be ignored (min length not reached) a draw that contains EOS is
        rejected and repeated, at most 100 more times (behaviour of cosy_llm.py:162-178, same error text)."""
        eos, max_trials = self.speech_token_size, 100
        for _ in range(max_trials + 1):
            ids = self.sampling(weighted_scores, decoded_tokens, sampling)
            if not ignore_eos or eos not in ids:
                return ids
        raise RuntimeError(f"sampling reaches max_trials {max_trials} and still get eos when ignore_eos is True, "
                           "check your input!")

    @torch.inference_mode()
    def inference(self, text, text_len, prompt_text, prompt_text_len, prompt_speech_token, prompt_speech_token_len,
                  embedding=None, sampling: int = 25, max_token_text_ratio: float = 20, min_token_text_ratio: float = 0.5,
                  cache=None):
        """Streaming generator of speech token ids (cosy_llm.py:180-272): prefill [sos, prompt_text+text, task_id,
        prompt_speech], then one persistent-state step per token; ids >= speech_token_size are EOS / skipped."""
        device = text.device
        text = torch.cat([prompt_text, text], dim=1)
        n_text = int((text_len + prompt_text_len).item())
        # an instruction prefix ends at the <|endofprompt|> id (65531): it does not count towards the length budget
        # (cosy_llm.py:195-207)
        hits = (text[0] == 65531).nonzero()
        n_instr = int(hits[0, 0].item()) + 1 if hits.numel() else 0
        content_length = n_text - n_instr
        original_text_len = n_text - n_instr
        emb_w = self.llm_embedding.weight
        pieces = [emb_w[self.sos_eos].view(1, 1, -1), self.text_embedding(text), emb_w[self.task_id].view(1, 1, -1)]
        if int(prompt_speech_token_len) != 0:
            pieces.append(self.speech_embedding(prompt_speech_token))
        lm_input = torch.cat(pieces, dim=1)
        min_len, max_len = int(content_length * min_token_text_ratio), int(content_length * max_token_text_ratio)
        out_tokens = []
        if cache is None:
            cache = Cache.zeros(self.config, 1, device, lm_input.dtype)
        step_kernel = None   # T = 1 steps through rwkv7_decode_step_bf16 once the prompt is in (bf16 models)
        graph = None         # ... and, with the stock sampler, the whole step incl. the draw replayed from a hipGraph
        self.last_inference_used_graph = False
        eos = self.speech_token_size
        for i in range(max_len):
            if graph is not None:
                # embedding of the previous id -> stack + head -> log-softmax -> repetition-aware draw -> ring / counter update: one
                # replay; the id crosses to the host once, because a streaming generator has to yield it
                graph.replay()
                cache.seen_tokens += 1
                top_ids = int(g_tok.item())
            else:
                if step_kernel is not None and lm_input.shape[1] == 1:
                    logits = step_kernel(lm_input[:, 0].contiguous()).unsqueeze(1)
                    cache.seen_tokens += 1
                else:
                    masks = torch.ones((1, lm_input.shape[1], lm_input.shape[1]), device=device, dtype=torch.bool)
                    logits, cache = self.forward_one_step(lm_input, masks=masks, cache=cache)
                    if getattr(self, "use_step_kernel", True) and step_kernel is None:
                        from .decode import DecodeStep
                        if DecodeStep.supported(self.model, self.lm_head, cache) is None:
                            step_kernel = DecodeStep(self.model, self.lm_head, cache)
                logp = logits[:, -1].float().log_softmax(dim=-1)
                top_ids = int(self.sampling_ids(logp.squeeze(dim=0), out_tokens, sampling,
                                                ignore_eos=(i + original_text_len < min_len)).item())
            if top_ids == eos:
                for st in cache.states:  # cosy_llm.py:247-251: token-shift states are zeroed at end of utterance
                    if graph is not None:   # the captured step holds these addresses
                        st.att_x_prev.zero_()
                        st.ffn_x_prev.zero_()
                    else:
                        st.att_x_prev = torch.zeros_like(st.att_x_prev)
                        st.ffn_x_prev = torch.zeros_like(st.ffn_x_prev)
                break
            if top_ids > eos:
                continue
            yield top_ids
            out_tokens.append(top_ids)
            lm_input = self.speech_embedding.weight[top_ids].reshape(1, 1, -1)
            if (graph is None and step_kernel is not None and getattr(self, "use_graph", True) and self.sampling is ras_sampling
                    and self.lm_head.weight.shape[0] == eos + 1):
                # capture the step on fixed buffers: g_tok (previous / new id), g_recent (last 10 emitted ids), g_i (step index)
                win = 10
                g_tok = torch.tensor([top_ids], device=device)
                g_recent = torch.full((win,), -1, dtype=torch.long, device=device)
                tail = out_tokens[-win:]
                g_recent[:len(tail)] = torch.tensor(tail, device=device)
                g_ptr = torch.tensor([len(tail) % win], device=device)
                g_i = torch.tensor(i + 1, device=device)
                n_ignore = min_len - original_text_len   # ignore_eos while step index < n_ignore

                from .sampling import MAX_DOMAIN, ras_step
                fused = getattr(self, "fused_sampling", True) and eos + 1 <= MAX_DOMAIN and 1 <= sampling <= 128
                from .sampling import fresh_seed
                seed_ = fresh_seed()   # one key per utterance, from torch's default generator (torch.manual_seed reproduces it)

                def captured():
                    x = self.speech_embedding.weight[g_tok]                      # [1, D]
                    if fused:   # draw + ring / counter update as ONE launch (csrc/sampling.hip); keyed by (seed, g_i)
                        ras_step(step_kernel(x.contiguous())[0], g_tok, g_recent, g_ptr, g_i, n_ignore, eos, top_k=sampling, win_size=win,
                                 seed=seed_)
                        return
                    logp_ = step_kernel(x.contiguous())[0].float().log_softmax(dim=-1)
                    new = ras_sampling_device(logp_, g_recent, g_i < n_ignore, eos, top_k=sampling)
                    g_tok.copy_(new)
                    keep_ = new != eos                                           # the reference appends emitted ids only
                    g_recent.scatter_(0, g_ptr, torch.where(keep_, new, g_recent.gather(0, g_ptr)))
                    g_ptr.copy_(torch.where(keep_, (g_ptr + 1) % win, g_ptr))
                    g_i.add_(1)

                snap = [(s_.att_x_prev.clone(), s_.att_kv.clone(), s_.ffn_x_prev.clone()) for s_ in cache.states]
                keep = [t.clone() for t in (g_tok, g_recent, g_ptr, g_i)]
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    captured()   # warm-up outside the capture; the state it advanced is restored below
                torch.cuda.current_stream().wait_stream(side)

                def restore():
                    for s_, (a_, kv_, f_) in zip(cache.states, snap):
                        s_.att_x_prev.copy_(a_)
                        s_.att_kv.copy_(kv_)
                        s_.ffn_x_prev.copy_(f_)
                    for t, k_ in zip((g_tok, g_recent, g_ptr, g_i), keep):
                        t.copy_(k_)

                restore()
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    captured()
                restore()
                self.last_inference_used_graph = True


class RWKV7LM(nn.Module):
    """model/llm/llm.py:17-133: wrapper whose forward(batch) returns {'loss', 'acc'}.  `llm` is a RWKV7CosyLM-like
    causal LM (here: backbone + lm_head); text embeddings come from llm.get_input_embeddings()."""

    def __init__(self, llm_input_size, llm_output_size, speech_token_size, llm: RWKV7CosyLM, sampling: Callable = None,
                 length_normalized_loss=True, lsm_weight=0.0, mix_ratio=(5, 15), drop_ratio=0.0):
        super().__init__()
        self.llm_input_size, self.llm_output_size, self.speech_token_size = llm_input_size, llm_output_size, speech_token_size
        self.sos_eos, self.task_id, self.fill_token = 0, 1, 2
        self.llm_embedding = nn.Embedding(2, llm_input_size)
        self.llm = llm
        self.text_embedding = llm.get_input_embeddings()
        self.speech_embedding = nn.Embedding(speech_token_size + 1, llm_input_size)
        self.length_normalized_loss, self.lsm_weight = length_normalized_loss, lsm_weight
        self.sampling = sampling or ras_sampling
        self.mix_ratio = list(mix_ratio)
        self.dropout = nn.Dropout(drop_ratio) if drop_ratio > 0 else None

    pad_unpad_sequence = RWKV7CosyLM.pad_unpad_sequence

    def forward(self, batch: dict):
        text_token, text_token_len = batch["text_token"], batch["text_token_len"]
        speech_token, speech_token_len = batch["speech_token"], batch["speech_token_len"]
        lm_target = [torch.tensor([IGNORE_ID] * (2 + int(text_token_len[i])) +
                                  speech_token[i, :int(speech_token_len[i])].tolist() + [self.speech_token_size])
                     for i in range(text_token.size(0))]
        lm_target = pad_sequence(lm_target, batch_first=True, padding_value=IGNORE_ID).to(text_token.device)
        text_emb = self.text_embedding(text_token)
        sos_eos_emb = self.llm_embedding.weight[self.sos_eos].reshape(1, 1, -1)
        task_id_emb = self.llm_embedding.weight[self.task_id].reshape(1, 1, -1)
        speech_emb = self.speech_embedding(speech_token)
        lm_input, attention_mask = self.pad_unpad_sequence(sos_eos_emb, text_emb, text_token_len, task_id_emb, speech_emb,
                                                           speech_token_len)
        if self.dropout is not None:
            lm_input = self.dropout(lm_input)
        logits = self.llm(inputs_embeds=lm_input, attention_mask=attention_mask).logits
        lm_target = lm_target[:, 1:].contiguous()
        loss = label_smoothing_kl(logits, lm_target, self.speech_token_size + 1, IGNORE_ID, self.lsm_weight,
                                  self.length_normalized_loss)
        acc = th_accuracy(logits.view(-1, self.speech_token_size + 1), lm_target, ignore_label=IGNORE_ID)
        return {"loss": loss, "acc": acc}
