"""Spark-layout TTS language model on the HIP backbone: drop-in for model/llm/spark_llm.py (RWKV7ForSpeech).

Same constructor/config fields, attribute names (`model`, `lm_head`, `text_embedder`, `global_embedder`,
`tts_tag_embedder`, `dropout`), `forward(...)` kwargs and return fields, `generate(...)` kwargs used by
inference/rwkv7speech_inference.py:99-107 and utils/utilities.py:101-117, `prepare_inputs_for_generation`,
`copy_state_dict` and the rwkvfla state_dict key layout -- so train_scripts/train_spark_rwkv7speech.py:234-246
and data/utils/spark_dataset.py:163-239 can drive it unchanged.

The reference inherits transformers.GenerationMixin (pinned 4.51.3, requirements.txt:245); the image has
transformers 5.x whose Cache/generate internals moved, so `generate` is a small self-contained loop over the
persistent-state decode path (greedy / top-k / top-p / temperature / suppress_tokens / min_new_tokens).
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from .backbone import Cache, Linear, ModelOutput, RWKV7Config, RWKV7Model
from .hf_api import HFModelMixin
from .losses import fused_linear_cross_entropy


class RWKV7SpeechConfig(RWKV7Config):
    """spark_llm.py:13-17: RWKV7Config + text_vocab_size + audio_global_vocab_size."""

    def __init__(self, text_vocab_size=65536, audio_global_vocab_size=4096, **kw):
        super().__init__(**{k: v for k, v in kw.items() if k in RWKV7Config.__dataclass_fields__ and k != "extra"})
        self.extra = {k: v for k, v in kw.items() if k not in RWKV7Config.__dataclass_fields__}
        self.text_vocab_size = text_vocab_size
        self.audio_global_vocab_size = audio_global_vocab_size

    def to_dict(self):
        d = super().to_dict()
        d.update(text_vocab_size=self.text_vocab_size, audio_global_vocab_size=self.audio_global_vocab_size,
                 architectures=["RWKV7ForSpeech"])
        return d

    @classmethod
    def from_dict(cls, d):
        d = {k: v for k, v in d.items() if k != "architectures"}
        return cls(**d)


class _GenerateOutput(ModelOutput):
    pass


class RWKV7ForSpeech(HFModelMixin, nn.Module):
    config_class = RWKV7SpeechConfig

    def __init__(self, config: RWKV7SpeechConfig):
        super().__init__()
        self.config = config
        self.model = RWKV7Model(config)
        self.vocab_size = config.vocab_size
        self.lm_head = Linear(config.hidden_size, config.vocab_size, bias=False)  # 8192 + eos (spark_llm.py:26)
        self.criterion = None
        self.text_embedder = nn.Embedding(config.text_vocab_size, config.hidden_size)
        self.global_embedder = nn.Embedding(config.audio_global_vocab_size, config.hidden_size)
        self.tts_tag_embedder = nn.Embedding(3, config.hidden_size)  # GLOBAL=0, SEMANTIC=1, START_TTS=2
        self.dropout = nn.Dropout(0.02)

    # ---- forward -------------------------------------------------------------------------------------
    def forward(self, input_ids=None, attention_mask=None, inputs_embeds=None, past_key_values: Optional[Cache] = None,
                labels=None, use_cache=None, output_attentions=None, output_hidden_states=None, return_dict=None,
                logits_to_keep: Optional[int] = 0, **kwargs):
        """spark_llm.py:105-172.  labels are shifted by one HERE (spark_llm.py:156), on top of whatever the batch
        builder did (data/utils/spark_dataset.py:223-233 pre-shifts) -- reproduced as is, not "fixed"."""
        return_dict = True if return_dict is None else return_dict
        if self.training and inputs_embeds is not None:
            inputs_embeds = self.dropout(inputs_embeds)
        outputs = self.model(input_ids=input_ids, attention_mask=attention_mask, inputs_embeds=inputs_embeds,
                             past_key_values=past_key_values, use_cache=use_cache, **kwargs)
        hidden_states = outputs[0]
        fuse = self.config.fuse_cross_entropy and self.training
        loss, logits = None, None
        if not fuse or labels is None:
            h = hidden_states if not logits_to_keep else hidden_states[:, -logits_to_keep:]
            logits = self.lm_head(h)
        if labels is not None:
            ignore_index = getattr(self.criterion, "ignore_index", -100)
            labels = labels.to(hidden_states.device)
            labels = torch.cat((labels[..., 1:], torch.full_like(labels[:, :1], ignore_index)), 1)
            cu = kwargs.get("cu_seqlens")
            if cu is not None:
                # packed rows: the reference's builder appends the sample that overflows max_cu_seqlens WITHOUT a cu_seqlens
                # entry (data/utils/spark_dataset.py:150-158, reproduced bit for bit in layouts.process_single_batch_culens);
                # the backbone returns zeros there, so those labels would add log(V)-sized terms with no useful gradient
                # and inflate the valid-token count that scales the whole step -- they are ignored instead
                # (config.strict_reference_loss = True keeps the reference's behaviour: CE on those positions, counted in the
                # normaliser; tests/test_model_gpu.py pins both, INTEGRATION.md lists the divergence)
                if not getattr(self.config, "strict_reference_loss", False):
                    pos = torch.arange(labels.shape[1], device=labels.device)
                    labels = torch.where(pos.unsqueeze(0) >= cu[-1].to(labels.device), torch.full_like(labels, ignore_index), labels)
            if fuse:
                loss = fused_linear_cross_entropy(hidden_states, labels, self.lm_head.weight, self.lm_head.bias,
                                                  ignore_index)
            elif self.criterion is not None:
                loss = self.criterion(logits.view(labels.numel(), -1), labels.view(-1))
            else:
                loss = F.cross_entropy(logits.view(labels.numel(), -1).float(), labels.view(-1),
                                       ignore_index=ignore_index)
        if not return_dict:
            out = (logits, outputs.past_key_values)
            return (loss,) + out if loss is not None else out
        return ModelOutput(loss=loss, logits=logits, past_key_values=outputs.past_key_values, hidden_states=None,
                           attentions=None)

    def prepare_inputs_for_generation(self, input_ids=None, past_key_values=None, attention_mask=None,
                                      inputs_embeds=None, use_cache=True, logits_to_keep=None, **kwargs):
        """spark_llm.py:70-102: embeddings on the first step only, afterwards the last generated id."""
        if past_key_values is not None and len(past_key_values) > 0 and past_key_values.seen_tokens > 0:
            input_ids = input_ids[:, -1:]
        if inputs_embeds is not None and (past_key_values is None or past_key_values.seen_tokens == 0):
            model_inputs = {"inputs_embeds": inputs_embeds}
        else:
            model_inputs = {"input_ids": input_ids.contiguous()}
        model_inputs.update(past_key_values=past_key_values, use_cache=use_cache, attention_mask=attention_mask,
                            logits_to_keep=logits_to_keep)
        return model_inputs

    # ---- generation ------------------------------------------------------------------------------------
    @torch.no_grad()
    def generate(self, input_ids=None, inputs_embeds=None, attention_mask=None, max_new_tokens=None, max_length=None,
                 do_sample=False, top_k=0, top_p=1.0, temperature=1.0, eos_token_id=None, pad_token_id=None,
                 suppress_tokens=None, min_new_tokens=0, return_dict_in_generate=False, use_cache=True,
                 generator: Optional[torch.Generator] = None, use_graph: Optional[bool] = None, **unused):
        """Prefill on the state-carrying kernel, then one persistent-state step per token.
        With inputs_embeds only, the returned sequences hold the NEW tokens only (HF semantics the reference
        relies on, inference/rwkv7speech_inference.py:108-118).

        use_graph: the per-token loop replayed from a hipGraph with the draw on the device (decode.GraphDecoder /
        MultiGroupDecoder: ~33 k tokens/s at B = 32 on MI355X against ~3 k for the host loop below, which reads `unfinished.any()`
        back every token).  None = when the request is covered: bf16 model on the step kernel, default generator, at most one EOS
        id, >= 32 new tokens.  Same ids as the host loop for greedy decoding (tests/test_model_gpu.py); sampled
        draws follow the same distribution through the fused sampler's own random stream."""
        was_training = self.training
        self.eval()
        if inputs_embeds is None:
            assert input_ids is not None
            B, P = input_ids.shape
        else:
            B, P = inputs_embeds.shape[:2]
        if max_new_tokens is None:
            max_new_tokens = (max_length or 20) - (0 if inputs_embeds is not None and input_ids is None else P)
        eos = [] if eos_token_id is None else ([eos_token_id] if isinstance(eos_token_id, int) else list(eos_token_id))
        pad = pad_token_id if pad_token_id is not None else (eos[0] if eos else 0)
        dev = self.device
        graph_ok = generator is None and len(eos) <= 1 and max_new_tokens >= 2 and self.dtype == torch.bfloat16 and dev.type == "cuda"
        if graph_ok and min_new_tokens and eos:   # min_new_tokens lives in the fused sampler only
            from .sampling import RowSampler
            V = self.lm_head.weight.shape[0]
            allow, sup = RowSampler.fold_suppress(V, suppress_tokens)
            graph_ok = RowSampler.supported(dev, [V], None if allow is None else [allow], sup, do_sample, top_k, top_p, temperature) is None
        if use_graph and not graph_ok:
            raise ValueError("use_graph=True needs a bf16 model on the HIP device, the default generator and at most one EOS id")
        if use_graph is None:
            use_graph = graph_ok and max_new_tokens >= 32
        if use_graph:
            from .decode import DecodeStep, GraphDecoder, MultiGroupDecoder
            probe = Cache.zeros(self.config, min(B, 32), dev, self.dtype)
            if DecodeStep.supported(self.model, self.lm_head, probe) is None:
                dec = GraphDecoder(self, B, step_kernel=True) if B <= 32 else MultiGroupDecoder(self, 32, step_kernel=True)
                seq = dec.generate(inputs_embeds=inputs_embeds, input_ids=input_ids if inputs_embeds is None else None,
                                   attention_mask=attention_mask, max_new_tokens=max_new_tokens, eos_token_id=eos[0] if eos else None,
                                   pad_token_id=pad, suppress_tokens=suppress_tokens, do_sample=do_sample, temperature=temperature,
                                   top_k=top_k, top_p=top_p, min_new_tokens=min_new_tokens)
                if eos:   # the host loop leaves with the step in which the last sequence finishes
                    done = (seq == eos[0]).long().cumsum(1) > 0
                    if bool(done[:, -1].all()):
                        seq = seq[:, :int((~done).sum(1).max()) + 1]
                seq = seq.clone()
                if input_ids is not None and inputs_embeds is None:
                    seq = torch.cat([input_ids, seq], 1)
                if was_training:
                    self.train()
                if return_dict_in_generate:
                    cache = dec.cache if B <= 32 else None
                    return _GenerateOutput(sequences=seq, past_key_values=cache)
                return seq
        cache = Cache.zeros(self.config, B, dev, self.dtype)
        out = self(input_ids=input_ids if inputs_embeds is None else None, inputs_embeds=inputs_embeds,
                   attention_mask=attention_mask, past_key_values=cache, use_cache=True, logits_to_keep=1)
        logits = out.logits[:, -1].float()
        # per-token steps: the whole stack through rwkv7_decode_step_bf16 when the model is covered (bf16, B <= 32)
        from .decode import DecodeStep
        step_kernel = DecodeStep(self.model, self.lm_head, cache) if DecodeStep.supported(self.model, self.lm_head, cache) is None else None
        unfinished = torch.ones(B, dtype=torch.bool, device=dev)
        new_tokens = []
        for step in range(max_new_tokens):
            if suppress_tokens:
                logits[:, suppress_tokens] = float("-inf")
            if eos and step < min_new_tokens:
                logits[:, eos] = float("-inf")
            nxt = sample_next(logits, do_sample, top_k, top_p, temperature, generator)
            nxt = torch.where(unfinished, nxt, torch.full_like(nxt, pad))
            new_tokens.append(nxt)
            for e in eos:
                unfinished &= nxt != e
            if eos and not bool(unfinished.any()):
                break
            if step_kernel is not None:
                logits = step_kernel(F.embedding(nxt, self.model.embeddings.weight)).clone()
                cache.seen_tokens += 1
            else:
                out = self(input_ids=nxt.unsqueeze(1), past_key_values=cache, use_cache=True)
                logits = out.logits[:, -1].float()
        seq = torch.stack(new_tokens, 1) if new_tokens else torch.empty(B, 0, dtype=torch.long, device=dev)
        if input_ids is not None and inputs_embeds is None:
            seq = torch.cat([input_ids, seq], 1)
        if was_training:
            self.train()
        if return_dict_in_generate:
            return _GenerateOutput(sequences=seq, past_key_values=cache)
        return seq

    # ---- checkpoints -----------------------------------------------------------------------------------
    def copy_state_dict(self, state_dict: dict):
        """spark_llm.py:174-201: take backbone weights from a base RWKV-7 LM; its token embedding becomes
        text_embedder; embeddings and lm_head of the speech model are left untouched."""
        target = self.state_dict()
        new_sd = {}
        for key, val in state_dict.items():
            if key == "model.embeddings.weight":
                new_sd["text_embedder.weight"] = val
                continue
            if "embeddings" in key or "lm_head" in key:
                continue
            if key in target or key.endswith("attn.x_x"):
                new_sd[key] = val
        info = self.load_state_dict(new_sd, strict=False)
        print(info)
        return self


def sample_next(logits, do_sample=False, top_k=0, top_p=1.0, temperature=1.0, generator=None):
    """Greedy (argmax, first max wins like torch.argmax) or temperature/top-k/top-p multinomial sampling,
    in HF's processor order: temperature -> top-k -> top-p."""
    if not do_sample:
        return torch.argmax(logits, dim=-1)
    if temperature and temperature != 1.0:
        logits = logits / temperature
    if top_k and top_k > 0:
        k = min(top_k, logits.shape[-1])
        kth = torch.topk(logits, k, dim=-1).values[..., -1:]
        logits = logits.masked_fill(logits < kth, float("-inf"))
    if top_p is not None and top_p < 1.0:
        sorted_logits, sorted_idx = torch.sort(logits, descending=False, dim=-1)
        cum = sorted_logits.softmax(dim=-1).cumsum(dim=-1)
        remove = cum <= (1 - top_p)
        remove[..., -1:] = False
        logits = logits.masked_fill(remove.scatter(1, sorted_idx, remove), float("-inf"))
    probs = torch.softmax(logits, dim=-1)
    return torch.multinomial(probs, 1, generator=generator).squeeze(1)


_STOCK_SAMPLE_NEXT = sample_next   # (tests script the draws by replacing spark_llm.sample_next: the fused sampler then stands down)
