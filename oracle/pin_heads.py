"""oracle/pin_heads.py -- AUTHORING-CONTAINER ONLY (needs /root/reference).

Pins the three heads and the XY sampler to the REFERENCE'S OWN CLASSES (round-3 verdict, missing #1):

  model/llm/spark_llm.py:105-172   RWKV7ForSpeech.forward   (dropout, double label shift, fused / plain CE)
  model/llm/cosy_llm.py:75-160     RWKV7CosyLM.forward      (pad_unpad_sequence, IGNORE_ID targets, LabelSmoothingLoss, max_tokens_k)
  model/llm/xy_llm.py:189-257      RWKV7XYLM.forward        (8 embeddings with padding_idx, 8 biased heads, summed CE)
  model/llm/xy_llm.py:39-146       CustomGenerationMixin._sample (channel-0 mask, 7-step flush, termination)

Those modules import `rwkvfla` (un-vendored, absent: requirements.txt:213) and `typing.Unpack` (Python >= 3.11).  The stubs below
stand in for the DEPENDENCY only -- `RWKV7Model` is an adapter over oracle/rwkv7_ref.backbone (itself pinned to the reference's
x070 module classes by pin_against_reference.py), `RWKV7ForCausalLM` a minimal PreTrainedModel + GenerationMixin,
`Fused*CrossEntropyLoss` torch's CE -- every line of the head classes and of `_sample` that runs is the reference's.

Writes tests/golden/heads.npz (inputs, toy weights, logits, losses, label masks, generated [B,T,C] id grids: data only) and checks
oracle/rwkv7_ref.{spark,cosy,xy}_forward against the reference outputs.     Usage: python oracle/pin_heads.py [--write]
"""
import argparse
import importlib.machinery
import os
import sys
import types
import typing

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import rwkv7_ref as R  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden", "heads.npz")
REF = "/root/reference"
SMALL = dict(hidden_size=128, num_hidden_layers=2, decay_low_rank_dim=32, a_low_rank_dim=32, v_low_rank_dim=16,
             gate_low_rank_dim=32)


# ------------------------------------------------------------------------------------------------------------------
# stubs for the absent dependency (rwkv-fla 0.7.202503140658)
# ------------------------------------------------------------------------------------------------------------------
def install_stubs():
    import typing_extensions
    if not hasattr(typing, "Unpack"):
        typing.Unpack = typing_extensions.Unpack
    from transformers import PretrainedConfig, PreTrainedModel
    from transformers.generation import GenerationMixin
    from transformers.modeling_outputs import BaseModelOutputWithPast

    class RWKV7Config(PretrainedConfig):
        model_type = "rwkv7"

        def __init__(self, hidden_size=128, num_hidden_layers=2, vocab_size=100, head_dim=64, decay_low_rank_dim=64,
                     a_low_rank_dim=64, v_low_rank_dim=32, gate_low_rank_dim=128, norm_eps=1e-5, fuse_cross_entropy=True,
                     use_cache=True, **kwargs):
            self.hidden_size, self.num_hidden_layers, self.vocab_size, self.head_dim = hidden_size, num_hidden_layers, vocab_size, head_dim
            self.decay_low_rank_dim, self.a_low_rank_dim = decay_low_rank_dim, a_low_rank_dim
            self.v_low_rank_dim, self.gate_low_rank_dim = v_low_rank_dim, gate_low_rank_dim
            self.norm_eps, self.fuse_cross_entropy, self.use_cache = norm_eps, fuse_cross_entropy, use_cache
            super().__init__(**kwargs)

    class RWKV7PreTrainedModel(PreTrainedModel):
        config_class = RWKV7Config
        base_model_prefix = "model"

        def _init_weights(self, module):
            pass

    def ref_cfg(config):
        return R.RefConfig(hidden_size=config.hidden_size, num_hidden_layers=config.num_hidden_layers, vocab_size=0,
                           decay_low_rank_dim=config.decay_low_rank_dim, a_low_rank_dim=config.a_low_rank_dim,
                           v_low_rank_dim=config.v_low_rank_dim, gate_low_rank_dim=config.gate_low_rank_dim, norm_eps=config.norm_eps)

    class RWKV7Model(RWKV7PreTrainedModel):
        """rwkvfla's backbone, as an adapter over oracle/rwkv7_ref.backbone; parameters registered under fla's key names."""

        def __init__(self, config):
            super().__init__(config)
            self.rcfg = ref_cfg(config)
            for key, val in R.init_params(self.rcfg, seed=0).items():
                if key.startswith("model."):
                    self._put(key[len("model."):], nn.Parameter(val.clone()))
            self.embeddings = nn.Embedding(config.vocab_size, config.hidden_size)

        def _put(self, key, param):
            mod, parts = self, key.split(".")
            for name in parts[:-1]:
                if not hasattr(mod, name):
                    setattr(mod, name, nn.Module())
                mod = getattr(mod, name)
            setattr(mod, parts[-1], param)

        def forward(self, input_ids=None, attention_mask=None, inputs_embeds=None, past_key_values=None, use_cache=None,
                    output_attentions=None, output_hidden_states=None, return_dict=None, **kwargs):
            x = self.embeddings(input_ids) if inputs_embeds is None else inputs_embeds
            p = {"model." + k: v for k, v in self.named_parameters()}
            h, _ = R.backbone(p, self.rcfg, x, attention_mask, None)
            return BaseModelOutputWithPast(last_hidden_state=h, past_key_values=None, hidden_states=None, attentions=None)

    class RWKV7ForCausalLM(RWKV7PreTrainedModel, GenerationMixin):
        def __init__(self, config):
            super().__init__(config)
            self.criterion = None

        # stateless stand-ins for what GenerationMixin/fla do with the recurrent cache: the whole sequence is re-run each step
        def prepare_inputs_for_generation(self, input_ids, **kwargs):
            return {"input_ids": input_ids}

        def _update_model_kwargs_for_generation(self, outputs, model_kwargs, is_encoder_decoder=False, **kw):
            return model_kwargs

    class FusedLinearCrossEntropyLoss(nn.Module):
        ignore_index = -100

        def forward(self, hidden, labels, weight, bias=None):
            logits = F.linear(hidden, weight, bias)
            return F.cross_entropy(logits.view(labels.numel(), -1), labels.view(-1), ignore_index=self.ignore_index)

    class FusedCrossEntropyLoss(nn.CrossEntropyLoss):
        def __init__(self, inplace_backward=False, **kw):
            super().__init__(**kw)

    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__spec__ = importlib.machinery.ModuleSpec(name, None)
        m.__path__ = []
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m

    for n in ("rwkvfla", "rwkvfla.models", "rwkvfla.models.rwkv7"):
        stub(n)
    stub("rwkvfla.models.rwkv7.modeling_rwkv7", RWKV7Model=RWKV7Model, RWKV7PreTrainedModel=RWKV7PreTrainedModel,
         RWKV7ForCausalLM=RWKV7ForCausalLM, Cache=object, FusedLinearCrossEntropyLoss=FusedLinearCrossEntropyLoss,
         FusedCrossEntropyLoss=FusedCrossEntropyLoss)
    stub("rwkvfla.models.rwkv7.configuration_rwkv7", RWKV7Config=RWKV7Config)
    for pth in (REF, os.path.join(REF, "third_party")):
        if pth not in sys.path:
            sys.path.insert(0, pth)


def head_weights(kind, g, D, **n):
    rnd = lambda *s, std=0.5: torch.randn(*s, generator=g) * std
    if kind == "spark":
        return {"lm_head.weight": rnd(n["V"], D, std=0.05), "text_embedder.weight": rnd(n["TV"], D),
                "global_embedder.weight": rnd(n["GV"], D), "tts_tag_embedder.weight": rnd(3, D),
                "model.embeddings.weight": rnd(n["V"], D)}
    if kind == "cosy":
        return {"llm_embedding.weight": rnd(2, D), "text_embedding.weight": rnd(n["V"], D),
                "speech_embedding.weight": rnd(n["S"] + 1, D), "lm_head.weight": rnd(n["S"] + 1, D, std=0.05),
                "lm_head.bias": rnd(n["S"] + 1, std=0.1), "model.embeddings.weight": torch.zeros(n["V"], D)}
    out = {"model.embeddings.weight": torch.zeros(n["V"], D)}
    for i in range(n["C"]):
        v = n["V"] if i == 0 else n["SV"]
        e = rnd(v, D)
        e[v - 1] = 0   # padding_idx row (xy_llm.py:162,168; zero_embs :176-187)
        out[f"embs.{i}.weight"], out[f"heads.{i}.weight"], out[f"heads.{i}.bias"] = e, rnd(v, D, std=0.05), rnd(v, std=0.1)
    return out


def load(model, rcfg, extra, seed):
    p = R.init_params(rcfg, seed=seed)
    p.update(extra)
    missing, unexpected = model.load_state_dict(p, strict=False)
    assert not unexpected, unexpected
    assert all(k.startswith("criterion") for k in missing), missing
    return p


def npd(prefix, d):
    return {f"{prefix}.{k}": (v.detach().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in d.items()}


def pin_spark(gold):
    import model.llm.spark_llm as ref
    g = torch.Generator().manual_seed(11)
    D, V = 128, 97
    cfg = ref.RWKV7SpeechConfig(vocab_size=V, text_vocab_size=60, audio_global_vocab_size=20, **SMALL)
    rcfg = R.RefConfig(vocab_size=V, **SMALL)
    m = ref.RWKV7ForSpeech(cfg)
    p = load(m, rcfg, head_weights("spark", g, D, V=V, TV=60, GV=20), seed=5)
    B, T = 3, 24
    x = torch.randn(B, T, D, generator=g) * 0.5
    mask = torch.ones(B, T, dtype=torch.long)
    mask[1, :5] = 0          # left padding (process_single_batch)
    mask[2, :9] = 0
    labels = torch.randint(0, V, (B, T), generator=g)
    labels[0, :7] = -100     # prompt positions
    labels[1, :11] = -100
    labels[2, :] = -100      # a row with no supervised position
    labels[2, 20:] = torch.tensor([3, 4, 5, V - 1])
    m.eval()
    with torch.no_grad():
        out_eval = m(inputs_embeds=x, attention_mask=mask, labels=labels)
        out_nolab = m(inputs_embeds=x, attention_mask=mask)
    m.train()
    m.dropout.p = 0.0        # the dropout draw is the only non-deterministic part of the training forward
    out_train = m(inputs_embeds=x, attention_mask=mask, labels=labels)      # fuse_linear_and_cross_entropy: logits is None
    assert out_train.logits is None
    out_train.loss.backward()
    dW = m.lm_head.weight.grad.clone()
    # the restatement (oracle/rwkv7_ref.spark_forward) against the reference class
    loss_o, logits_o, _ = R.spark_forward(p, rcfg, x, mask, labels)
    e1 = (logits_o - out_eval.logits).abs().max().item()
    e2 = abs(loss_o.item() - out_eval.loss.item())
    e3 = abs(out_train.loss.item() - out_eval.loss.item())
    assert e1 < 1e-5 and e2 < 1e-6 and e3 < 1e-6, (e1, e2, e3)
    assert torch.equal(out_nolab.logits, out_eval.logits)
    print(f"  OK  spark: RWKV7ForSpeech.forward logits == restatement to {e1:.1e}, loss {out_eval.loss.item():.6f} (d {e2:.1e}; fused-train d {e3:.1e})")
    gold.update(npd("spark.p", p))
    gold.update(npd("spark", dict(x=x, mask=mask, labels=labels, logits=out_eval.logits, loss=out_eval.loss, loss_train=out_train.loss,
                                  d_lm_head=dW, cfg=np.asarray([V, 60, 20]))))


def pin_cosy(gold):
    import model.llm.cosy_llm as ref
    g = torch.Generator().manual_seed(12)
    D, V, S = 128, 200, 50
    hw = head_weights("cosy", g, D, V=V, S=S)       # one set of weights for every scenario
    for tag, lsm, norm in (("cosy", 0.1, True), ("cosy_b", 0.0, False)):
        cfg = ref.RWKV7CosyConfig(vocab_size=V, speech_token_size=S, lsm_weight=lsm, length_normalized_loss=norm, **SMALL)
        rcfg = R.RefConfig(vocab_size=0, **SMALL)
        m = ref.RWKV7CosyLM(cfg).eval()
        p = load(m, rcfg, hw, seed=6)
        tl = torch.tensor([4, 2, 6], dtype=torch.int32)
        sl = torch.tensor([7, 3, 5], dtype=torch.int32)
        tt = torch.randint(0, V, (3, 6), generator=g)
        st = torch.randint(0, S, (3, 7), generator=g)
        for i in range(3):
            tt[i, tl[i]:] = 0
            st[i, sl[i]:] = 0
        batch = dict(text_token=tt, text_token_len=tl, speech_token=st, speech_token_len=sl)
        with torch.no_grad():
            out = m(batch=batch, return_dict=True)
            out_k = m(batch=batch, max_tokens_k=1, return_dict=True)      # 1024 positions budget: nothing is cut at T = 15
            out_tuple = m(batch=batch)     # return_dict=None: the tuple form the trainer indexes (train_cosy_...:276-277)
            T = out.logits.shape[1]
        assert isinstance(out_tuple, tuple) and torch.equal(out_tuple[0], out.loss)
        assert torch.equal(out_k.logits, out.logits)
        loss_o, acc_o, logits_o = R.cosy_forward(p, rcfg, batch, S, lsm, norm)
        # compare on valid positions (the padded tail sees the -1 padding VALUE in both, but is masked out of the loss)
        lens = (2 + tl + sl).tolist()
        valid = torch.zeros(3, T, dtype=torch.bool)
        for i, n in enumerate(lens):
            valid[i, :n] = True
        e1 = (logits_o - out.logits)[valid].abs().max().item()
        e2 = abs(loss_o.item() - out.loss.item())
        assert e1 < 1e-5 and e2 < 1e-5, (e1, e2)
        print(f"  OK  {tag}: RWKV7CosyLM.forward logits == restatement to {e1:.1e}, loss {out.loss.item():.6f} (d {e2:.1e})")
        if tag == "cosy":
            gold.update(npd("cosy.p", p))
        gold.update(npd(tag, dict(text_token=tt, text_token_len=tl, speech_token=st, speech_token_len=sl, logits=out.logits,
                                  valid=valid, loss=out.loss,
                                  cfg=np.asarray([V, S, lsm, float(norm)]))))


    # max_tokens_k (cosy_llm.py:122-130): the budget is max_tokens_k * 1024 positions; a batch of 3 x 600 positions under k = 1 keeps
    # 1024 // 600 = 1 sequence (inputs, labels and mask cut alike); the loss is that sequence's alone
    tl = torch.tensor([9, 5, 7], dtype=torch.int32)
    sl = torch.tensor([589, 300, 411], dtype=torch.int32)
    tt = torch.randint(0, V, (3, 9), generator=g)
    st = torch.randint(0, S, (3, 589), generator=g)
    batch = dict(text_token=tt, text_token_len=tl, speech_token=st, speech_token_len=sl)
    with torch.no_grad():
        out_cut = m(batch=batch, max_tokens_k=1, return_dict=True)
    assert out_cut.logits.shape[:2] == (1, 600), out_cut.logits.shape
    one = {k: v[:1] for k, v in batch.items()}
    loss_o, _, _ = R.cosy_forward(p, rcfg, one, S, lsm, norm)
    assert abs(loss_o.item() - out_cut.loss.item()) < 1e-5
    print(f"  OK  cosy_cut: max_tokens_k=1 on 3 x 600 positions keeps B={out_cut.logits.shape[0]}, loss {out_cut.loss.item():.6f}")
    gold.update(npd("cosy_cut", dict(text_token=tt, text_token_len=tl, speech_token=st, speech_token_len=sl, loss=out_cut.loss,
                                     last_logits=out_cut.logits[0, -1])))


def pin_xy(gold):
    import model.llm.xy_llm as ref
    g = torch.Generator().manual_seed(13)
    D, V, SV, C, SHIFT = 128, 120, 16, 4, 100
    for tag, lsm in (("xy", 0.0), ("xy_ls", 0.1)):
        cfg = ref.RWKV7XYConfig(vocab_size=V, speech_vocab_size=SV, num_channels=C, text_shift_size=SHIFT, lsm_weight=lsm, **SMALL)
        rcfg = R.RefConfig(vocab_size=0, **SMALL)
        m = ref.RWKV7XYLM(cfg).eval()
        if tag == "xy":
            hw = head_weights("xy", g, D, V=V, SV=SV, C=C)
            B, T = 2, 20
            ids = torch.stack([torch.randint(0, V - 1, (B, T), generator=g)] +
                              [torch.randint(0, SV - 1, (B, T), generator=g) for _ in range(1, C)], -1)
            ids[:, :3, 1:] = SV - 1          # text phase: the speech channels carry the pad id (zero embedding rows)
            ids[1, -2:, 0] = V - 1           # channel-0 pad
            labels = ids.roll(-1, 1).clone()
            labels[:, -1] = -100
            labels[0, :4, 0] = -100
            labels[1, 5:9, 2] = -100
            mask = torch.ones(B, T, dtype=torch.long)
            mask[1, -2:] = 0                 # right padding (xy_data_processor)
        p = load(m, rcfg, hw, seed=7)
        with torch.no_grad():
            out = m(input_ids=ids, attention_mask=mask, labels=labels, return_dict=True)
        loss_o, logits_o = R.xy_forward(p, rcfg, ids, mask, labels, C, lsm)
        e1 = max((a - b).abs().max().item() for a, b in zip(logits_o, out.logits))
        e2 = abs(loss_o.item() - out.loss.item())
        assert e1 < 1e-5 and e2 < 1e-5, (e1, e2)
        print(f"  OK  {tag}: RWKV7XYLM.forward {C} heads == restatement to {e1:.1e}, summed CE {out.loss.item():.6f} (d {e2:.1e})")
        if tag == "xy":
            gold.update(npd("xy.p", p))
            gold.update(npd("xy", dict(ids=ids, mask=mask, labels=labels, cfg=np.asarray([V, SV, C, SHIFT]),
                                       **{f"logits{i}": l for i, l in enumerate(out.logits)})))
        gold.update(npd(tag, dict(loss=out.loss)))
    # ---- _sample on the toy model: draws = argmax of the processed distribution (a scripted torch.multinomial), so the ids depend
    # on the model: channel-0 mask + heads + the loop's one-frame termination (xy_llm.py:139-140: `needs_additional_steps == -1`
    # holds for every sequence that is not flushing, so the loop leaves after the first frame)
    prompt = ids[:, :6].clone()
    grid = run_sample(ref, m, prompt, lambda probs, num_samples=1: probs.argmax(-1, keepdim=True), eos=None, max_new=5)
    assert grid.shape == (2, 7, C), grid.shape
    gold.update(npd("xy", dict(sample_prompt=prompt, sample_greedy=grid)))
    print(f"  OK  xy: _sample on the toy model (argmax draws): {tuple(grid.shape)}, new row {grid[:, -1].tolist()}")
    return ref


def run_sample(ref, model, prompt, draw, eos, max_new):
    """CustomGenerationMixin._sample (xy_llm.py:39-146) called as the reference's generate() would: an empty LogitsProcessorList,
    MaxLength (+ EosToken) stopping criteria; torch.multinomial replaced by `draw` for the duration of the call."""
    from transformers.generation import GenerationConfig, LogitsProcessorList, StoppingCriteriaList
    from transformers.generation.stopping_criteria import EosTokenCriteria, MaxLengthCriteria
    crit = [MaxLengthCriteria(max_length=prompt.shape[1] + max_new)]
    if eos is not None:
        crit.append(EosTokenCriteria(eos_token_id=eos))
    gc = GenerationConfig(eos_token_id=eos, return_dict_in_generate=False, output_scores=False)
    real = torch.multinomial
    torch.multinomial = draw
    try:
        with torch.no_grad():
            out = ref.CustomGenerationMixin._sample(model, prompt.clone(), logits_processor=LogitsProcessorList(),
                                                    stopping_criteria=StoppingCriteriaList(crit), generation_config=gc,
                                                    synced_gpus=False, streamer=None)
    finally:
        torch.multinomial = real
    return out


class _ScriptedXY(nn.Module):
    """Duck-typed model for the scripted scenarios at the real channel count and vocabularies (8 channels, V0 = 66 661): `_sample`
    only touches config, prepare_inputs_for_generation, __call__, _update_model_kwargs_for_generation and is_audio_token; the
    logits are constant (the scripted draws do not look at them beyond the channel-0 mask, which IS checked)."""

    def __init__(self, ref, cfg):
        super().__init__()
        self.config, self.ref = cfg, ref

    def prepare_inputs_for_generation(self, input_ids, **kw):
        return {"input_ids": input_ids}

    def _update_model_kwargs_for_generation(self, outputs, model_kwargs, is_encoder_decoder=False):
        return model_kwargs

    def is_audio_token(self, token_id):
        return self.ref.CustomGenerationMixin.is_audio_token(self, token_id)

    def forward(self, input_ids=None, return_dict=True, **kw):
        B, T, C = input_ids.shape
        sizes = [self.config.vocab_size] + [self.config.speech_vocab_size] * (C - 1)
        return types.SimpleNamespace(logits=[torch.zeros(B, T, v) for v in sizes], past_key_values=None)


def pin_xy_scripted(ref, gold):
    C, V0, SV, SHIFT = 8, 66661, 1025, 65536
    cfg = ref.RWKV7XYConfig(vocab_size=V0, speech_vocab_size=SV, num_channels=C, text_shift_size=SHIFT, **SMALL)
    cfg.is_encoder_decoder = False
    model = _ScriptedXY(ref, cfg)
    PAD, EOS = cfg.speech_pad_token, 65535
    B, T0 = 3, 4
    prompt = torch.full((B, T0, C), PAD, dtype=torch.long)
    prompt[:, :, 0] = torch.arange(T0) + 5
    scenarios = [  # name, {sequence: frame at which channel 0 draws a text id}, eos id, max_new_tokens
        ("s1", {0: 2, 1: 5}, EOS, 16),            # nobody flushes at frame 0: ONE frame
        ("s2", {1: 0}, None, 16),                 # sequence 1 flushes from frame 0: C frames, the others stop after frame 0
        ("s3", {1: 0}, EOS, 16),                  # the flush's own EOS meets the EOS criterion in its first row
        ("s4", {0: 0, 1: 0, 2: 0}, None, 16),     # all three flush together
        ("s5", {0: 0, 2: 3}, None, 16),           # a stopped sequence drawing a text id later
        ("s6", {}, SHIFT + 10, 16),               # EOS inside the audio range, drawn at frame 0
        ("s7", {0: 0}, None, 4),                  # the length bound ends a flush
        ("s8", {0: 0, 1: 0}, [EOS, 3], 16),       # a list of EOS ids: eos_token_id[0] is what the flush writes
    ]
    for name, trigger, eos, max_new in scenarios:
        calls = {"n": 0}

        def draw(probs, num_samples=1):
            step, ch = divmod(calls["n"], C)
            calls["n"] += 1
            if ch == 0:   # what the mask must have done: only audio ids have probability
                assert (probs[:, :SHIFT] == 0).all() and (probs[:, SHIFT + SV:] == 0).all() and (probs[:, SHIFT:SHIFT + SV] > 0).all()
                out = torch.full((probs.shape[0],), SHIFT + 10 + step, dtype=torch.long)
                for s_, st_ in trigger.items():
                    if step == st_:
                        out[s_] = 7
                return out[:, None]
            return torch.full((probs.shape[0], 1), 100 * ch + step, dtype=torch.long)

        grid = run_sample(ref, model, prompt, draw, eos, max_new)
        trig = np.full(B, -1, dtype=np.int64)
        for s_, st_ in trigger.items():
            trig[s_] = st_
        eos_arr = np.asarray([] if eos is None else ([eos] if isinstance(eos, int) else eos), dtype=np.int64)
        gold.update(npd(f"xys.{name}", dict(trigger=trig, eos=eos_arr, max_new=np.int64(max_new), grid=grid)))
        print(f"  OK  xy _sample scripted {name}: trigger {trigger} eos {eos} max_new {max_new} -> {grid.shape[1] - T0} frame(s)")
    gold.update(npd("xys", dict(prompt=prompt, cfg=np.asarray([C, V0, SV, SHIFT, PAD]))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--write", action="store_true")
    a = ap.parse_args()
    torch.set_num_threads(4)
    install_stubs()
    gold = {}
    pin_spark(gold)
    pin_cosy(gold)
    ref_xy = pin_xy(gold)
    pin_xy_scripted(ref_xy, gold)
    if a.write:
        np.savez_compressed(GOLD, **gold)
        print("wrote", GOLD, f"({os.path.getsize(GOLD) / 1e6:.2f} MB, {len(gold)} arrays)")


if __name__ == "__main__":
    main()
