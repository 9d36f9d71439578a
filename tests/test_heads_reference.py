"""The three heads and the XY sampler against OUTPUTS OF THE REFERENCE'S OWN CLASSES (tests/golden/heads.npz, written by
oracle/pin_heads.py in the authoring container: model/llm/spark_llm.py:105-172, cosy_llm.py:75-160, xy_llm.py:189-257 and
CustomGenerationMixin._sample xy_llm.py:39-146 executed over a stub of the absent rwkvfla dependency).

CPU part (not gpu): the oracle restatement (oracle/rwkv7_ref.py) reproduces the reference outputs.
GPU part (-m gpu): the HIP-backed product classes reproduce them -- fp32 logits within 1e-3, losses within 1e-4, generated id grids
id for id (`reference_termination=True` = xy_llm.py:139-140 literally)."""
import os

import numpy as np
import pytest
import torch

from oracle import rwkv7_ref as R

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "heads.npz"))
SMALL = dict(hidden_size=128, num_hidden_layers=2, decay_low_rank_dim=32, a_low_rank_dim=32, v_low_rank_dim=16,
             gate_low_rank_dim=32)
DEV = "cuda:0"
gpu = pytest.mark.gpu


def T(name):
    return torch.from_numpy(np.asarray(GOLD[name]))


def params(prefix):
    pre = prefix + ".p."
    return {k[len(pre):]: T(k) for k in GOLD.files if k.startswith(pre)}


def cosy_batch(tag):
    return {k: T(f"{tag}.{k}") for k in ("text_token", "text_token_len", "speech_token", "speech_token_len")}


# ---------------------------------------------------------------- CPU: the restatement against the reference classes' outputs
def test_oracle_spark_head_matches_reference_class():
    p, rcfg = params("spark"), R.RefConfig(vocab_size=int(T("spark.cfg")[0]), **SMALL)
    loss, logits, _ = R.spark_forward(p, rcfg, T("spark.x"), T("spark.mask"), T("spark.labels"))
    assert (logits - T("spark.logits")).abs().max().item() < 1e-5
    assert abs(loss.item() - float(T("spark.loss"))) < 1e-6 and abs(loss.item() - float(T("spark.loss_train"))) < 1e-6


def test_oracle_cosy_head_matches_reference_class():
    p, rcfg = params("cosy"), R.RefConfig(vocab_size=0, **SMALL)
    for tag in ("cosy", "cosy_b"):
        V, S, lsm, norm = T(f"{tag}.cfg").tolist()
        loss, _, logits = R.cosy_forward(p, rcfg, cosy_batch(tag), int(S), lsm, bool(norm))
        valid = T(f"{tag}.valid")
        assert (logits - T(f"{tag}.logits"))[valid].abs().max().item() < 1e-5
        assert abs(loss.item() - float(T(f"{tag}.loss"))) < 1e-5 * max(1.0, abs(loss.item()))


def test_oracle_xy_head_matches_reference_class():
    p, rcfg = params("xy"), R.RefConfig(vocab_size=0, **SMALL)
    V, SV, C, SHIFT = T("xy.cfg").tolist()
    for tag, lsm in (("xy", 0.0), ("xy_ls", 0.1)):
        loss, logits = R.xy_forward(p, rcfg, T("xy.ids"), T("xy.mask"), T("xy.labels"), C, lsm)
        assert abs(loss.item() - float(T(f"{tag}.loss"))) < 1e-5
    for i, l in enumerate(logits):
        assert (l - T(f"xy.logits{i}")).abs().max().item() < 1e-5


def test_reference_sample_grids_obey_the_documented_rules():
    """Sanity of the fixtures themselves (what xy_llm.py:100-140 does, read off the reference's own output): a sequence that draws a
    text id at frame 0 carries EOS (or the drawn id when there is no EOS id) on channel 0 and pads channel i from flush row i on; the
    other sequences stop after frame 0 and carry pad_text / speech pad."""
    C, V0, SV, SHIFT, PAD = T("xys.cfg").tolist()
    T0 = T("xys.prompt").shape[1]
    g = T("xys.s2.grid")[:, T0:]
    assert g.shape[1] == C
    assert g[1, :, 0].tolist() == [7] + [SHIFT + 10 + s for s in range(1, C)]
    for j in range(C):
        for ch in range(1, C):
            assert int(g[1, j, ch]) == (100 * ch + j if j < ch else PAD)
    assert (g[0, 1:, 0] == 0).all() and (g[0, 1:, 1:] == PAD).all()
    assert T("xys.s1.grid").shape[1] == T0 + 1 and T("xys.s3.grid").shape[1] == T0 + 1


# ---------------------------------------------------------------- GPU: the product classes against the reference classes' outputs
@gpu
def test_spark_head_hip_vs_reference_class():
    from rwkvtts_amd.spark_llm import RWKV7ForSpeech, RWKV7SpeechConfig
    V, TV, GV = T("spark.cfg").tolist()
    model = RWKV7ForSpeech(RWKV7SpeechConfig(vocab_size=V, text_vocab_size=TV, audio_global_vocab_size=GV, **SMALL))
    model.load_state_dict(params("spark"), strict=True)
    model = model.to(DEV).eval()
    x, mask, labels = T("spark.x").to(DEV), T("spark.mask").to(DEV), T("spark.labels").to(DEV)
    with torch.no_grad():
        out = model(inputs_embeds=x, attention_mask=mask, labels=labels)
    valid = T("spark.mask").bool()
    assert (out.logits.cpu() - T("spark.logits"))[valid].abs().max().item() < 1e-3
    assert abs(out.loss.item() - float(T("spark.loss"))) < 1e-4
    # training forward (fused linear + CE, logits never materialised), dropout off as in the fixture; head gradient
    model.train()
    model.dropout.p = 0.0
    out = model(inputs_embeds=x, attention_mask=mask, labels=labels)
    assert out.logits is None and abs(out.loss.item() - float(T("spark.loss_train"))) < 1e-4
    out.loss.backward()
    dW = T("spark.d_lm_head")
    assert (model.lm_head.weight.grad.cpu() - dW).abs().max().item() < 1e-3 * dW.abs().max().item() + 1e-6


@gpu
def test_cosy_head_hip_vs_reference_class():
    from rwkvtts_amd.cosy_llm import RWKV7CosyConfig, RWKV7CosyLM
    for tag in ("cosy", "cosy_b"):
        V, S, lsm, norm = T(f"{tag}.cfg").tolist()
        cfg = RWKV7CosyConfig(vocab_size=int(V), speech_token_size=int(S), lsm_weight=lsm, length_normalized_loss=bool(norm), **SMALL)
        model = RWKV7CosyLM(cfg)
        model.load_state_dict(params("cosy"), strict=True)
        model = model.to(DEV).eval()
        batch = {k: v.to(DEV) for k, v in cosy_batch(tag).items()}
        with torch.no_grad():
            out = model(batch=batch)
            tup = model(batch=batch, return_dict=False)
        valid = T(f"{tag}.valid")
        want = float(T(f"{tag}.loss"))
        assert (out.logits.cpu() - T(f"{tag}.logits"))[valid].abs().max().item() < 1e-3
        assert abs(out.loss.item() - want) < 1e-4 * max(1.0, abs(want))
        assert abs(float(tup[0]) - want) < 1e-4 * max(1.0, abs(want))      # the tuple form train_cosy_...:276-277 indexes
    # max_tokens_k = 1 on 3 x 600 positions keeps 1024 // 600 = 1 sequence (cosy_llm.py:122-130); the last model is cosy_b's
    batch = {k: v.to(DEV) for k, v in cosy_batch("cosy_cut").items()}
    with torch.no_grad():
        out = model(batch=batch, max_tokens_k=1)
    want = float(T("cosy_cut.loss"))
    assert out.logits.shape[:2] == (1, 600)
    assert abs(out.loss.item() - want) < 2e-4 * want
    assert (out.logits[0, -1].cpu() - T("cosy_cut.last_logits")).abs().max().item() < 1e-3


def _xy_model(lsm=0.0):
    from rwkvtts_amd.xy_llm import RWKV7XYConfig, RWKV7XYLM
    V, SV, C, SHIFT = T("xy.cfg").tolist()
    cfg = RWKV7XYConfig(vocab_size=V, speech_vocab_size=SV, num_channels=C, text_shift_size=SHIFT, lsm_weight=lsm, **SMALL)
    model = RWKV7XYLM(cfg)
    model.load_state_dict(params("xy"), strict=True)
    return model.to(DEV).eval()


@gpu
def test_xy_head_hip_vs_reference_class():
    ids, mask, labels = T("xy.ids").to(DEV), T("xy.mask").to(DEV), T("xy.labels").to(DEV)
    valid = T("xy.mask").bool()
    for tag, lsm in (("xy", 0.0), ("xy_ls", 0.1)):
        model = _xy_model(lsm)
        with torch.no_grad():
            out = model(input_ids=ids, attention_mask=mask, labels=labels, return_dict=True)
        assert abs(out.loss.item() - float(T(f"{tag}.loss"))) < 1e-4
        for i, l in enumerate(out.logits):
            assert (l.cpu() - T(f"xy.logits{i}"))[valid].abs().max().item() < 1e-3
        model.train()
        out = model(input_ids=ids, attention_mask=mask, labels=labels)
        assert out.logits is None and abs(out.loss.item() - float(T(f"{tag}.loss"))) < 1e-4


@gpu
def test_xy_generate_toy_model_equals_reference_sample():
    """_sample run by the reference on the toy model with argmax draws: our greedy generate with the reference's termination rule
    returns the same [B, T0+1, C] grid (channel-0 mask, eight heads, one-frame termination)."""
    model = _xy_model()
    prompt = T("xy.sample_prompt").to(DEV)
    want = T("xy.sample_greedy")
    for fused in (True, False):
        model.fused_frame = fused
        out = model.generate(prompt, max_new_tokens=5, do_sample=False, reference_termination=True)
        assert torch.equal(out.cpu(), want), (fused, out.cpu(), want)
    # default termination: the same first frame, then the run continues to the length bound
    out = model.generate(prompt, max_new_tokens=5, do_sample=False)
    assert out.shape[1] == prompt.shape[1] + 5 and torch.equal(out[:, :want.shape[1]].cpu(), want)


@gpu
@pytest.mark.parametrize("fused_frame", [True, False])
def test_xy_generate_scripted_scenarios_equal_reference_sample(monkeypatch, fused_frame):
    """Eight scripted scenarios at the real channel count and vocabularies (8 channels, V0 = 66 661): flush from frame 0, several
    sequences flushing together, a stopped sequence drawing a text id later, EOS ids in and outside the audio range, an EOS list,
    the length bound inside a flush.  `reference_termination=True` must return the reference's grid id for id; the default mode
    must agree on every row of a sequence that flushes from frame 0 when no EOS id interferes (the two differ only in WHEN
    non-flushing sequences stop)."""
    from rwkvtts_amd import spark_llm
    from rwkvtts_amd.xy_llm import RWKV7XYConfig, RWKV7XYLM
    C, V0, SV, SHIFT, PAD = T("xys.cfg").tolist()
    cfg = RWKV7XYConfig(vocab_size=V0, speech_vocab_size=SV, num_channels=C, text_shift_size=SHIFT, **SMALL)
    model = RWKV7XYLM(cfg).init_weights(seed=2)
    model.zero_embs()
    model = model.to(DEV).eval()
    model.fused_frame = fused_frame
    prompt = T("xys.prompt").to(DEV)
    T0 = prompt.shape[1]
    calls = {"n": 0}
    trigger = {}

    def scripted(logits, *a, **k):
        step, ch = divmod(calls["n"], C)
        calls["n"] += 1
        if ch == 0:
            out = torch.full((logits.shape[0],), SHIFT + 10 + step, dtype=torch.long, device=logits.device)
            for s_, st_ in trigger.items():
                if step == st_:
                    out[s_] = 7
            return out
        return torch.full((logits.shape[0],), 100 * ch + step, dtype=torch.long, device=logits.device)

    monkeypatch.setattr(spark_llm, "sample_next", scripted)
    for name in ("s1", "s2", "s3", "s4", "s5", "s6", "s7", "s8"):
        trig, eos = T(f"xys.{name}.trigger").tolist(), T(f"xys.{name}.eos").tolist()
        max_new, want = int(T(f"xys.{name}.max_new")), T(f"xys.{name}.grid")
        trigger.clear()
        trigger.update({b: t for b, t in enumerate(trig) if t >= 0})
        eos_arg = None if not eos else (eos[0] if len(eos) == 1 else eos)
        calls["n"] = 0
        out = model.generate(prompt, max_new_tokens=max_new, eos_token_id=eos_arg, reference_termination=True).cpu()
        assert out.shape == want.shape and torch.equal(out, want), (name, out.shape, want.shape)
        if not eos:   # default termination: rows of the sequences that flush from frame 0 are the reference's
            calls["n"] = 0
            out = model.generate(prompt, max_new_tokens=max_new, eos_token_id=eos_arg).cpu()
            for b, t in trigger.items():
                if t == 0:
                    n = want.shape[1]
                    assert torch.equal(out[b, :n], want[b]), (name, b)
