"""Text + audio-token batch layouts (Spark first; Cosy / XY live next to their heads).

Spark layout (reference: inference/rwkv7speech_inference.py:35-67 `create_inputs`,
data/utils/spark_dataset.py:163-239 `process_single_batch`): one sample is the embedding sequence
    [TAG2(START_TTS), text..., TAG0(GLOBAL), global x32, TAG1(SEMANTIC), semantic...(, EOS)]
built from four embedding tables (text_embedder, global_embedder, tts_tag_embedder, model.embeddings);
EOS id = vocab_size - 1 = 8192.  Inference left-pads and returns (inputs_embeds, attention_mask).
"""
from __future__ import annotations

from typing import List, Sequence

import torch


def spark_embed_sample(llm, text_ids: Sequence[int], global_ids: Sequence[int], semantic_ids: Sequence[int]):
    """[1, 3 + len(text) + len(global) + len(semantic), D] embedding sequence of one sample."""
    dev = llm.device
    t = lambda ids: torch.tensor([list(ids)], dtype=torch.long, device=dev)
    tag = lambda i: llm.tts_tag_embedder(t([i]))
    return torch.cat([tag(2), llm.text_embedder(t(text_ids)), tag(0), llm.global_embedder(t(global_ids)), tag(1),
                      llm.model.embeddings(t(semantic_ids))], dim=1)


def create_inputs(texts_ids: List[Sequence[int]], global_tokens_ids: List[Sequence[int]],
                  semantic_tokens_ids: List[Sequence[int]], llm):
    """inference/rwkv7speech_inference.py:35-67 with pre-tokenised text: left-padded embeddings + mask."""
    assert len(texts_ids) == len(global_tokens_ids) == len(semantic_tokens_ids)
    embs = [spark_embed_sample(llm, a, b, c) for a, b, c in zip(texts_ids, global_tokens_ids, semantic_tokens_ids)]
    L = max(e.shape[1] for e in embs)
    B = len(embs)
    mask = torch.zeros(B, L, dtype=torch.long, device=llm.device)
    out = []
    for i, e in enumerate(embs):
        mask[i, L - e.shape[1]:] = 1
        out.append(torch.cat([e.new_zeros(1, L - e.shape[1], e.shape[2]), e], dim=1))
    return torch.cat(out, dim=0), mask


def synthetic_spark_batch(llm, B: int, T: int = 4096, seed: int = 1234, n_text: int = 255, n_global: int = 32):
    """BASELINE.json configs[1]/[2]: per sample text ids ~U[0,65536) x255, global ids
    ~U[0,4096) x32, semantic ids ~U[0,8192) x(T-3-255-32) -> T positions, no padding.  Labels are aligned
    to positions (the model shifts by one, spark_llm.py:156): semantic positions carry their own id, the rest
    is -100.  Built with batched device lookups; returns dict(inputs_embeds, attention_mask, labels)."""
    cfg = llm.config
    dev = llm.device
    g = torch.Generator().manual_seed(seed)
    n_sem = T - 3 - n_text - n_global
    assert n_sem > 0
    text = torch.randint(0, cfg.text_vocab_size, (B, n_text), generator=g).to(dev)
    glob = torch.randint(0, cfg.audio_global_vocab_size, (B, n_global), generator=g).to(dev)
    sem = torch.randint(0, cfg.vocab_size - 1, (B, n_sem), generator=g).to(dev)
    tag = lambda i: llm.tts_tag_embedder(torch.full((B, 1), i, dtype=torch.long, device=dev))
    embs = torch.cat([tag(2), llm.text_embedder(text), tag(0), llm.global_embedder(glob), tag(1),
                      llm.model.embeddings(sem)], dim=1)
    labels = torch.full((B, T), -100, dtype=torch.long, device=dev)
    labels[:, T - n_sem:] = sem
    mask = torch.ones(B, T, dtype=torch.long, device=dev)
    return dict(inputs_embeds=embs, attention_mask=mask, labels=labels)
