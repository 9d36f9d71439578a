"""CPU: checkpoint interchange (rwkvtts_amd/checkpoint.py) against the golden output of the reference's converter
(tests/golden/convert_keys.npz, made by oracle/pin_checkpoint.py from utils/convert_rwkv.py:15-41) and the
model-construction helpers' contracts (train_functions.py:9-33, convert_rwkv7_to_xy.py:10-103)."""
import torch

from conftest import load_golden
from rwkvtts_amd import checkpoint as C
from rwkvtts_amd.backbone import RWKV7Config, RWKV7ForCausalLM, init_weights

SMALL = dict(hidden_size=128, num_hidden_layers=2, decay_low_rank_dim=32, a_low_rank_dim=32, v_low_rank_dim=16,
             gate_low_rank_dim=32, vocab_size=50)


def test_fla_to_x070_matches_reference_converter_and_inverts():
    g = load_golden("convert_keys.npz")
    sd = {k[3:]: v for k, v in g.items() if k.startswith("in.")}
    ref = {k[4:]: v for k, v in g.items() if k.startswith("ref.")}
    got = C.fla_to_x070(sd)
    assert set(got) == set(ref)
    for k, v in ref.items():
        assert got[k].shape == v.shape and torch.equal(got[k].contiguous(), v), k
    # names of Appendix B
    assert "blocks.0.att.w1" in got and "blocks.0.att.w0" in got and "blocks.1.att.v2" in got
    assert "blocks.0.ln0.weight" in got and "ln_out.bias" in got and "head.weight" in got
    assert got["blocks.0.att.w1"].shape == (8, 3)            # transposed to [in, out]
    assert got["blocks.0.att.x_w"].shape == (1, 8)            # x_x rows in the order r,w,k,v,a,g
    assert torch.equal(got["blocks.0.att.x_w"][0], sd["model.layers.0.attn.x_x"][1])
    back = C.x070_to_fla(got)
    want = C.split_x_x(sd)
    assert set(back) == set(want)
    for k in want:
        assert torch.equal(back[k].contiguous(), want[k].contiguous()), k
    fused = C.fuse_x_x(want)
    assert torch.equal(fused["model.layers.0.attn.x_x"], sd["model.layers.0.attn.x_x"])
    flat = C.spark_flat_vocab_export(got, pad_head=True)
    assert flat["emb.weight"].shape[0] == 11 + 3 + 4 + 5 and flat["head.weight"].shape[0] == 23
    assert torch.equal(flat["emb.weight"][11:14], sd["tts_tag_embedder.weight"])
    assert flat["head.weight"][11:].abs().sum() == 0


def test_x070_state_dict_loads_into_backbone():
    cfg = RWKV7Config(**SMALL)
    m = RWKV7ForCausalLM(cfg)
    init_weights(m, cfg, seed=1)
    x070 = C.fla_to_x070(m.state_dict())
    assert not any(k.startswith("model.") for k in x070)
    m2 = RWKV7ForCausalLM(cfg)
    m2.load_state_dict(C.x070_to_fla(x070), strict=True)
    for (k, a), (_, b) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(a, b), k


def test_alter_emb_and_head():
    cfg = RWKV7Config(**SMALL)
    m = RWKV7ForCausalLM(cfg)
    init_weights(m, cfg, seed=2)
    old = m.model.embeddings.weight.detach().clone()
    C.alter_emb_and_head(m, 70, 40, generator=torch.Generator().manual_seed(0))
    assert m.model.embeddings.weight.shape == (70, 128) and m.config.vocab_size == 70
    assert torch.equal(m.model.embeddings.weight[:50], old)
    assert abs(m.model.embeddings.weight[50:].std().item() - old.std().item()) < 0.3 * old.std().item()
    assert m.lm_head.weight.shape == (41, 128) and m.lm_head.bias is not None


def test_xy_and_cosy_from_base():
    cfg = RWKV7Config(**SMALL)
    base = RWKV7ForCausalLM(cfg)
    init_weights(base, cfg, seed=3)
    sd = C.fuse_x_x(base.state_dict())       # base checkpoints may carry the fused x_x form
    xy = C.xy_from_base(sd, cfg.to_dict(), num_channels=4, speech_vocab_size=9, n_special=6)
    assert xy.config.vocab_size == 50 + 9 + 6 and len(xy.embs) == 4 and xy.heads[1].weight.shape == (9, 128)
    assert torch.equal(xy.embs[0].weight[:50], base.model.embeddings.weight)
    assert torch.equal(xy.heads[0].weight[:50], base.lm_head.weight)
    assert xy.embs[0].weight[-1].abs().sum() == 0 and xy.embs[2].weight[-1].abs().sum() == 0      # zero_embs
    assert torch.equal(xy.model.layers[1].attn.x_k, base.model.layers[1].attn.x_k)
    assert torch.equal(xy.model.layers[0].attn.w_lora.lora[0].weight, base.model.layers[0].attn.w_lora.lora[0].weight)
    cosy = C.cosy_from_base(sd, cfg.to_dict(), vocab_size=64, speech_token_size=20)
    assert cosy.text_embedding.weight.shape == (64, 128) and cosy.lm_head.weight.shape == (21, 128)
    assert torch.equal(cosy.text_embedding.weight[:50], base.model.embeddings.weight)
    assert torch.equal(cosy.model.layers[1].ffn.key.weight, base.model.layers[1].ffn.key.weight)


def test_auto_map_shim_loads_through_transformers_auto_classes(tmp_path):
    """The reference loads its Spark checkpoints with AutoModelForCausalLM.from_pretrained(dir, trust_remote_code=True) through
    config.json's auto_map -> modeling_rwkvspeech.py (model/test/audio_rwkv.config:9-13, data/spark/modeling_rwkvspeech.py:1-6).
    save_pretrained() writes both; the Auto classes of the installed transformers must hand back OUR classes with identical
    weights.  Runs in a subprocess so that transformers' dynamic-module cache lives under tmp_path."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = f'''
import sys, json, os, torch
sys.path.insert(0, {root!r})
from rwkvtts_amd.spark_llm import RWKV7ForSpeech, RWKV7SpeechConfig
cfg = RWKV7SpeechConfig(vocab_size=257, text_vocab_size=300, audio_global_vocab_size=64, hidden_size=128, num_hidden_layers=2,
                        decay_low_rank_dim=32, a_low_rank_dim=32, v_low_rank_dim=32, gate_low_rank_dim=32)
m = RWKV7ForSpeech(cfg).init_weights(1)
d = {str(tmp_path / "ckpt")!r}
m.save_pretrained(d)
c = json.load(open(os.path.join(d, "config.json")))
assert c["auto_map"]["AutoModelForCausalLM"] == "modeling_rwkvspeech.RWKV7ForSpeech" and c["architectures"] == ["RWKV7ForSpeech"]
assert c["text_vocab_size"] == 300 and c["audio_global_vocab_size"] == 64
from transformers import AutoConfig, AutoModelForCausalLM
ac = AutoConfig.from_pretrained(d, trust_remote_code=True)
assert type(ac).__name__ == "RWKV7SpeechConfig" and ac.hidden_size == 128 and ac.text_vocab_size == 300
m2 = AutoModelForCausalLM.from_pretrained(d, trust_remote_code=True)
assert isinstance(m2, RWKV7ForSpeech), type(m2).__mro__   # transformers may wrap the class to add its GenerationMixin
sd, sd2 = m.state_dict(), m2.state_dict()
assert sd.keys() == sd2.keys() and all(torch.equal(sd[k], sd2[k]) for k in sd)
m3 = RWKV7ForSpeech.from_pretrained(d, torch_dtype=torch.bfloat16)
assert m3.lm_head.weight.dtype == torch.bfloat16
print("AUTO_MAP_OK")
'''
    env = dict(os.environ, HF_HOME=str(tmp_path / "hf"), HF_MODULES_CACHE=str(tmp_path / "hf" / "modules"), HF_HUB_OFFLINE="1",
               PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
    out = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, env=env, cwd=str(tmp_path), timeout=600)
    assert out.returncode == 0 and "AUTO_MAP_OK" in out.stdout, out.stderr[-3000:]
