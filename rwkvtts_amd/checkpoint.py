"""Checkpoint interchange: key maps between the rwkvfla naming this package uses and the
BlinkDL "x070" naming, fused-`x_x` migration, and construction of the Spark / Cosy / XY models from a base RWKV-7
language model.  Pure state-dict work on CPU tensors; nothing here touches the HIP kernels.

Behaviour restated from the reference (no code shared):
  utils/convert_rwkv.py:15-41      fla -> x070 renames, LoRA transposes, x_x split in the order r,w,k,v,a,g
  utils/convert_rwkv.py:43-74      flat-vocabulary export  emb = [semantic | tts_tag | global | text], padded head
  model/llm/convert_2_cosy_llm.py:6-47 / cosyvoice/cli/model.py:99-111   x_x [6,D] -> x_r..x_g [1,1,D]
  train_scripts/train_functions.py:9-33   alter_emb_and_head (vocabulary enlargement + fresh speech head)
  model/llm/convert_rwkv7_to_xy.py:10-103 multi-channel XY model from a base model
  model/llm/spark_llm.py:174-201   RWKV7ForSpeech.copy_state_dict (lives on the model class)
"""
from __future__ import annotations

import re
from typing import Dict, Optional

import torch
import torch.nn as nn

MIX_ORDER = ("r", "w", "k", "v", "a", "g")

# ordered (fla fragment, x070 fragment); applied left to right exactly like the reference's chain of str.replace
_FLA_TO_X070 = (("model.", ""), ("layers.", "blocks."), ("lm_head", "head"), ("ffn_norm", "ln2"), ("attn_norm", "ln1"),
                ("pre_norm", "ln0"), ("g_norm", "ln_x"), ("norm", "ln_out"), ("attn", "att"), ("r_proj", "receptance"),
                ("k_proj", "key"), ("v_proj", "value"), ("o_proj", "output"))
_LORA_TO_X070 = (("_lora.lora.2.bias", "0"), ("_lora.lora.2.weight", "2"), ("_lora.lora.0.weight", "1"))


def split_x_x(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """`...attn.x_x` [6,D] (RWKV7Attention "version 1") -> `...attn.x_r` .. `x_g`, each [1,1,D]."""
    out = {}
    for k, v in sd.items():
        if k.endswith(".x_x"):
            for i, n in enumerate(MIX_ORDER):
                out[k[:-3] + f"x_{n}"] = v[i].reshape(1, 1, -1)
        else:
            out[k] = v
    return out


def fuse_x_x(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """inverse of split_x_x: x_r..x_g -> x_x [6,D] (for runtimes that expect the fused form)."""
    out = dict(sd)
    for k in [k for k in sd if k.endswith(".x_r")]:
        base = k[:-3]
        out[base + "x_x"] = torch.cat([sd[base + f"x_{n}"].reshape(1, -1) for n in MIX_ORDER], 0)
        for n in MIX_ORDER:
            del out[base + f"x_{n}"]
    return out


def fla_to_x070(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """rwkvfla keys -> BlinkDL x070 keys (LoRA matrices transposed to [in,out], x_x split)."""
    out = {}
    for k, v in sd.items():
        for a, b in _FLA_TO_X070:
            k = k.replace(a, b)
        if "_lora.lora." in k and "weight" in k:
            v = v.transpose(0, 1)
        for a, b in _LORA_TO_X070:
            k = k.replace(a, b)
        if "att.x_x" in k:
            for i, n in enumerate(MIX_ORDER):
                out[k.replace("x_x", f"x_{n}")] = v[i:i + 1]
        else:
            out[k] = v
    return out


_X070_BLOCK = re.compile(r"^blocks\.(\d+)\.(.*)$")
_X070_SUB = {"ln0": "pre_norm", "ln1": "attn_norm", "ln2": "ffn_norm"}
_X070_ATT = {"receptance": "r_proj", "key": "k_proj", "value": "v_proj", "output": "o_proj", "ln_x": "g_norm"}


def x070_to_fla(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """BlinkDL x070 keys (e.g. an RWKV-7 "world" .pth) -> the rwkvfla keys of this package's backbone."""
    out = {}
    for k, v in sd.items():
        if k in ("emb.weight", "embeddings.weight"):   # BlinkDL name / the reference converter's Spark output
            out["model.embeddings.weight"] = v
        elif k.startswith("head."):
            out["lm_head." + k[5:]] = v
        elif k.startswith("ln_out."):
            out["model.norm." + k[7:]] = v
        else:
            m = _X070_BLOCK.match(k)
            if not m:
                out[k] = v     # layout-specific extras (text_embedder, ...) pass through
                continue
            i, rest = m.group(1), m.group(2)
            pre = f"model.layers.{i}."
            head, _, tail = rest.partition(".")
            if head in _X070_SUB:
                out[pre + _X070_SUB[head] + "." + tail] = v
            elif head == "att":
                name, _, leaf = tail.partition(".")
                if name in _X070_ATT:
                    out[pre + "attn." + _X070_ATT[name] + "." + leaf] = v
                elif len(name) == 2 and name[0] in "wavg" and name[1] in "012":
                    kind = {"1": "lora.0.weight", "2": "lora.2.weight", "0": "lora.2.bias"}[name[1]]
                    out[pre + f"attn.{name[0]}_lora.{kind}"] = v.transpose(0, 1) if name[1] in "12" else v.reshape(-1)
                elif name.startswith("x_"):
                    out[pre + "attn." + name] = v.reshape(1, 1, -1)
                else:                      # k_k, k_a, r_k
                    out[pre + "attn." + name] = v.reshape(-1) if name in ("k_k", "k_a") else v
            elif head == "ffn":
                name, _, leaf = tail.partition(".")
                if name == "x_k":
                    out[pre + "ffn.x_k"] = v.reshape(-1)
                else:
                    out[pre + f"ffn.{name}.{leaf}"] = v
            else:
                out[pre + rest] = v
    return out


def spark_flat_vocab_export(x070_sd: Dict[str, torch.Tensor], pad_head: bool = False) -> Dict[str, torch.Tensor]:
    """Single-table export of a converted Spark model: emb = [semantic | tts_tag | global | text]
    (convert_rwkv.py:52-58); with pad_head the head gets zero rows up to the table size (:70-72)."""
    sd = dict(x070_sd)
    sd["emb.weight"] = torch.cat([sd.pop("embeddings.weight"), sd.pop("tts_tag_embedder.weight"),
                                  sd.pop("global_embedder.weight"), sd.pop("text_embedder.weight")], 0)
    if pad_head:
        h = sd["head.weight"]
        sd["head.weight"] = torch.cat([h, h.new_zeros(sd["emb.weight"].shape[0] - h.shape[0], h.shape[1])], 0)
    return sd


def alter_emb_and_head(model, vocab_size: int, audio_token_size: int, generator: Optional[torch.Generator] = None):
    """Enlarge `model.model.embeddings` to vocab_size rows (new rows ~ N(0, std(old)^2)) and replace `lm_head` by a
    fresh Linear(hidden, audio_token_size + 1) with N(0, 0.02^2) weights (train_functions.py:9-33)."""
    old = model.model.embeddings
    cur, dim = old.weight.shape
    new = nn.Embedding(max(vocab_size, cur), dim).to(old.weight.device, old.weight.dtype)
    with torch.no_grad():
        new.weight[:cur] = old.weight
        if new.weight.shape[0] > cur:
            std = old.weight.float().std().item()
            new.weight[cur:] = (torch.randn(new.weight.shape[0] - cur, dim, generator=generator) * std).to(new.weight)
    model.model.embeddings = new
    model.config.vocab_size = new.weight.shape[0]
    head = nn.Linear(model.config.hidden_size, audio_token_size + 1).to(old.weight.device, old.weight.dtype)
    with torch.no_grad():
        head.weight.copy_((torch.randn(head.weight.shape, generator=generator) * 0.02).to(head.weight))
    model.lm_head = head
    return model


def xy_from_base(base_sd: Dict[str, torch.Tensor], base_config: dict, num_channels: int = 8,
                 speech_vocab_size: int = 1025, n_special: int = 100, seed: int = 0):
    """Multi-channel XY model from a base RWKV-7 LM state dict (convert_rwkv7_to_xy.py:10-103): channel-0 vocabulary =
    base vocabulary + speech_vocab_size `[SPi]` + n_special `[Si]/[CTLi]` tokens; backbone copied; overlapping rows of
    the channel-0 embedding/head copied, the rest and all speech channels ~ N(0, initializer_range^2); pad rows zeroed
    (zero_embs).  Returns the new model (on CPU)."""
    from .xy_llm import RWKV7XYConfig, RWKV7XYLM
    base_sd = split_x_x(base_sd)
    old_vocab = base_sd["model.embeddings.weight"].shape[0]
    cfg = dict(base_config)
    cfg.update(num_channels=num_channels, speech_vocab_size=speech_vocab_size,
               vocab_size=old_vocab + speech_vocab_size + n_special)
    config = RWKV7XYConfig.from_dict(cfg)
    model = RWKV7XYLM(config)
    g = torch.Generator().manual_seed(seed)
    std = getattr(config, "initializer_range", 0.02)
    backbone = {k[len("model."):]: v for k, v in base_sd.items() if k.startswith("model.") and "embeddings" not in k}
    missing, unexpected = model.model.load_state_dict(backbone, strict=False)
    assert not unexpected and all("embeddings" in m for m in missing), (missing, unexpected)
    with torch.no_grad():
        for i in range(num_channels):
            model.embs[i].weight.copy_(torch.randn(model.embs[i].weight.shape, generator=g) * std)
            model.heads[i].weight.copy_(torch.randn(model.heads[i].weight.shape, generator=g) * std)
            if model.heads[i].bias is not None:
                model.heads[i].bias.zero_()
        model.embs[0].weight[:old_vocab] = base_sd["model.embeddings.weight"]
        model.heads[0].weight[:old_vocab] = base_sd["lm_head.weight"]
        if "lm_head.bias" in base_sd and model.heads[0].bias is not None:
            model.heads[0].bias[:old_vocab] = base_sd["lm_head.bias"]
    model.zero_embs()
    return model


def cosy_from_base(base_sd: Dict[str, torch.Tensor], base_config: dict, vocab_size: int = 65548,
                   speech_token_size: int = 6561, seed: int = 0):
    """Cosy-layout model from a base RWKV-7 LM (convert_2_cosy_llm.py + RWKV7LM.__init__ -> alter_emb_and_head):
    backbone copied, text embedding enlarged to vocab_size, fresh speech head / speech + task embeddings."""
    from .cosy_llm import RWKV7CosyConfig, RWKV7CosyLM
    base_sd = split_x_x(base_sd)
    cfg = dict(base_config)
    cfg.update(vocab_size=vocab_size, speech_token_size=speech_token_size)
    model = RWKV7CosyLM(RWKV7CosyConfig.from_dict(cfg))
    model.init_weights(seed)
    backbone = {k[len("model."):]: v for k, v in base_sd.items() if k.startswith("model.") and "embeddings" not in k}
    missing, unexpected = model.model.load_state_dict(backbone, strict=False)
    assert not unexpected and all("embeddings" in m for m in missing), (missing, unexpected)
    old = base_sd["model.embeddings.weight"]
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        emb = model.text_embedding.weight   # RWKV7LM.text_embedding = the (enlarged) input embedding of the base LM, llm.py:62
        n = min(old.shape[0], emb.shape[0])
        emb[:n] = old[:n]
        if emb.shape[0] > n:
            emb[n:] = torch.randn(emb.shape[0] - n, emb.shape[1], generator=g) * old.float().std().item()
    return model
