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

from .backbone import mark_all_ones


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
    # ids leave the host from pinned memory without a synchronising copy (what a DataLoader with pin_memory=True + non_blocking .to()
    # does): a pageable .to(dev) makes the host wait for everything queued before it, i.e. for the previous step's optimizer, and the
    # GPU then idles while the host builds the batch (1.75 ms per step in profiles/r05i_step_busy.txt)
    def up(t):
        if torch.device(dev).type != "cuda":
            return t.to(dev)
        return t.pin_memory().to(dev, non_blocking=True)
    text = up(torch.randint(0, cfg.text_vocab_size, (B, n_text), generator=g))
    glob = up(torch.randint(0, cfg.audio_global_vocab_size, (B, n_global), generator=g))
    sem = up(torch.randint(0, cfg.vocab_size - 1, (B, n_sem), generator=g))
    tag = lambda i: llm.tts_tag_embedder(torch.full((B, 1), i, dtype=torch.long, device=dev))
    embs = torch.cat([tag(2), llm.text_embedder(text), tag(0), llm.global_embedder(glob), tag(1),
                      llm.model.embeddings(sem)], dim=1)
    labels = torch.full((B, T), -100, dtype=torch.long, device=dev)
    labels[:, T - n_sem:] = sem
    mask = mark_all_ones(torch.ones(B, T, dtype=torch.long, device=dev), True)
    return dict(inputs_embeds=embs, attention_mask=mask, labels=labels)


# ----------------------------------------------------------------------------------------------------------
# Spark training batches
# ----------------------------------------------------------------------------------------------------------
def _unpad_left(ids, mask):
    n = int(mask.sum().item())
    return ids[-n:] if n > 0 else ids[:0]


def process_single_batch(batch, rwkv7speech_model, eos_token_id=8192):
    """data/utils/spark_dataset.py:163-239.  batch: left-padded id tensors `input_ids`, `global_tokens_ids`,
    `semantic_tokens_ids` with their `attention_mask_input_ids` / `global_tokens_attention_mask` /
    `semantic_tokens_attention_mask`.  Returns left-padded `input_embs`, `attention_mask`, and labels that are
    ALREADY shifted by one (label[p] = token at p+1, EOS on the last position) -- the model shifts once more
    (spark_llm.py:156); reproduced as is."""
    llm = rwkv7speech_model
    device = llm.device
    B = batch["input_ids"].shape[0]
    embs, sems = [], []
    for i in range(B):
        text = _unpad_left(batch["input_ids"][i], batch["attention_mask_input_ids"][i])
        glob = _unpad_left(batch["global_tokens_ids"][i], batch["global_tokens_attention_mask"][i])
        sem = _unpad_left(batch["semantic_tokens_ids"][i], batch["semantic_tokens_attention_mask"][i])
        embs.append(spark_embed_sample(llm, text.tolist(), glob.tolist(), sem.tolist()))
        sems.append(sem)
    L = max(e.shape[1] for e in embs)
    attention_mask = torch.zeros(B, L, dtype=torch.long, device=device)
    labels = torch.full((B, L), -100, dtype=torch.long, device=device)
    out = []
    for i, e in enumerate(embs):
        attention_mask[i, L - e.shape[1]:] = 1
        out.append(torch.cat([e.new_zeros(1, L - e.shape[1], e.shape[2]), e], dim=1))
        n = sems[i].numel()
        labels[i, -n - 1:-1] = sems[i].to(device)
        labels[i, -1] = eos_token_id
    mark_all_ones(attention_mask, all(e.shape[1] == L for e in embs))   # known on the host: no read-back in the model
    return {"input_embs": torch.cat(out, dim=0), "attention_mask": attention_mask, "labels": labels}


def process_single_batch_culens(batch, rwkv7speech_model, eos_token_id=8192, max_cu_seqlens=8192):
    """data/utils/spark_dataset.py:111-162: samples packed back to back into ONE row [1, sum T, D] with
    cu_seqlens; packing stops (after appending the overflowing sample's tensors, as the reference does) once the
    running length would exceed max_cu_seqlens."""
    llm = rwkv7speech_model
    device = llm.device
    B = batch["input_ids"].shape[0]
    embs, labels, cu = [], [], [0]
    for i in range(B):
        text = _unpad_left(batch["input_ids"][i], batch["attention_mask_input_ids"][i])
        glob = _unpad_left(batch["global_tokens_ids"][i], batch["global_tokens_attention_mask"][i])
        sem = _unpad_left(batch["semantic_tokens_ids"][i], batch["semantic_tokens_attention_mask"][i])
        e = spark_embed_sample(llm, text.tolist(), glob.tolist(), sem.tolist())[0]
        embs.append(e)
        n = e.shape[0]
        lab = torch.full((n,), -100, dtype=torch.long, device=device)
        lab[-sem.numel() - 1:-1] = sem.to(device)
        lab[-1] = eos_token_id
        labels.append(lab)
        if cu[-1] + n > max_cu_seqlens:
            break
        cu.append(cu[-1] + n)
    return {"input_embs": torch.cat(embs, 0).unsqueeze(0), "labels": torch.cat(labels, 0).unsqueeze(0),
            "cu_seqlens": torch.tensor(cu, dtype=torch.long, device=device)}


def _jsonl_sample(model, t, g, s, eos_token_id):
    """One utterance of the jsonl trainers (utils/multiple_jsonl.py:13-55): inputs [TAG2, text, TAG0, global, TAG1,
    semantic + EOS] and labels ALIGNED with them (the model's forward shifts): -100 over the prefix, then the
    semantic ids + EOS."""
    dev = model.device
    s_pred = list(s) + [eos_token_id]
    emb = spark_embed_sample(model, t, g, s_pred)[0]
    prefix = 1 + len(t) + 1 + len(g) + 1
    lab = torch.cat([torch.full((prefix,), -100, dtype=torch.long, device=dev),
                     torch.tensor(s_pred, dtype=torch.long, device=dev)])
    return emb, lab


def _jsonl_sample_with_properties(model, t, g, s, props, eos_token_id, emb, semantic_labels=True):
    """The controllable-TTS twin of an utterance (utils/multiple_jsonl.py:183-210): the property-token ids go through
    `text_embedder` in FRONT of the plain sample, and the labels now also cover the global tokens:
        [-100 x (len(props) + 1 + len(text) + 1), global ids, -100 (TAG1), semantic ids + EOS]
    (`semantic_labels=False`: the `_global_tokens` variants, :362-374, ignore the semantic part as well)."""
    dev = model.device
    p_emb = model.text_embedder(torch.tensor(list(props), dtype=torch.long, device=dev))
    ign = lambda n: torch.full((n,), -100, dtype=torch.long, device=dev)
    s_pred = torch.tensor(list(s) + [eos_token_id], dtype=torch.long, device=dev)
    lab = torch.cat([ign(len(props) + 1 + len(t) + 1), torch.tensor(list(g), dtype=torch.long, device=dev), ign(1),
                     s_pred if semantic_labels else ign(s_pred.numel())])
    return torch.cat([p_emb, emb], dim=0), lab


def _right_pad(embs, labs, dev):
    lengths = [e.shape[0] for e in embs]
    mask = torch.zeros(len(embs), max(lengths), dtype=torch.long, device=dev)
    for i, n in enumerate(lengths):
        mask[i, :n] = 1
    mark_all_ones(mask, min(lengths) == max(lengths))
    return {"input_embs": torch.nn.utils.rnn.pad_sequence(embs, batch_first=True, padding_value=0.0),
            "labels": torch.nn.utils.rnn.pad_sequence(labs, batch_first=True, padding_value=-100),
            "attention_mask": mask}


def _pack(embs, labs, dev):
    cu = [0]
    for e in embs:
        cu.append(cu[-1] + e.shape[0])
    return {"input_embs": torch.cat(embs, 0).unsqueeze(0), "labels": torch.cat(labs, 0).unsqueeze(0),
            "cu_seqlens": torch.tensor(cu, dtype=torch.long, device=dev)}


def _jsonl_samples(text_ids, global_tokens, semantic_tokens, model, eos_token_id):
    embs, labs = [], []
    for t, g, s in zip(text_ids, global_tokens, semantic_tokens):
        e, l = _jsonl_sample(model, t, g, s, eos_token_id)
        embs.append(e)
        labs.append(l)
    return embs, labs


def create_inputs_and_labels(text_ids: List[Sequence[int]], global_tokens: List[Sequence[int]],
                             semantic_tokens: List[Sequence[int]], model, eos_token_id):
    """utils/multiple_jsonl.py:4-74 with pre-tokenised text: semantic ids + EOS are INPUTS too, labels are
    aligned with the inputs (the model's forward shifts), right padding, mask of ones over the real length."""
    return _right_pad(*_jsonl_samples(text_ids, global_tokens, semantic_tokens, model, eos_token_id), model.device)


def create_inputs_and_labels_culens(text_ids: List[Sequence[int]], global_tokens: List[Sequence[int]],
                                    semantic_tokens: List[Sequence[int]], model, eos_token_id):
    """utils/multiple_jsonl.py:76-136: the same samples packed back to back into ONE row [1, sum T, D] with
    `cu_seqlens` [n+1] (int64) -- what `model(inputs_embeds=, labels=, cu_seqlens=)` trains on without padding
    (train_spark_rwkv7speech.py:238-239).  No length cap here (unlike spark_dataset's `_culens` twin)."""
    return _pack(*_jsonl_samples(text_ids, global_tokens, semantic_tokens, model, eos_token_id), model.device)


def _properties_samples(text_ids, global_tokens, semantic_tokens, properties_ids, model, eos_token_id, both, semantic_labels):
    assert len(properties_ids) == len(text_ids)
    embs, labs = [], []
    for t, g, s, pr in zip(text_ids, global_tokens, semantic_tokens, properties_ids):
        e, l = _jsonl_sample(model, t, g, s, eos_token_id)
        if both:   # sample 1: plain TTS, sample 2: the same utterance behind its property tokens
            embs.append(e)
            labs.append(l)
        e2, l2 = _jsonl_sample_with_properties(model, t, g, s, pr, eos_token_id, e, semantic_labels)
        embs.append(e2)
        labs.append(l2)
    return embs, labs


def create_inputs_and_labels_with_properties(text_ids, global_tokens, semantic_tokens, properties_ids, model, eos_token_id):
    """utils/multiple_jsonl.py:139-234.  `properties_ids[i]` = tokenizer ids of the property string the reference
    builds with `convert_properties_to_tokens(age, gender, emotion, pitch, speed)` ("SPCT_0SPCT_15..."; string work
    on the host, outside the path).  TWO rows per utterance, interleaved [plain_0, props_0, plain_1, props_1, ...];
    right padding; the property rows also carry labels on the 32 global tokens (:198-210)."""
    return _right_pad(*_properties_samples(text_ids, global_tokens, semantic_tokens, properties_ids, model,
                                           eos_token_id, True, True), model.device)


def create_inputs_and_labels_with_properties_culens(text_ids, global_tokens, semantic_tokens, properties_ids, model,
                                                    eos_token_id):
    """utils/multiple_jsonl.py:236-311: the interleaved plain / property rows packed into [1, sum T, D] + cu_seqlens."""
    return _pack(*_properties_samples(text_ids, global_tokens, semantic_tokens, properties_ids, model,
                                      eos_token_id, True, True), model.device)


def create_inputs_and_labels_with_properties_global_tokens(text_ids, global_tokens, semantic_tokens, properties_ids,
                                                           model, eos_token_id):
    """utils/multiple_jsonl.py:313-400: ONE row per utterance (the property row only), labels on the global tokens
    ONLY (the semantic part is input but ignored by the loss, :362-374); right padding."""
    return _right_pad(*_properties_samples(text_ids, global_tokens, semantic_tokens, properties_ids, model,
                                           eos_token_id, False, False), model.device)


def create_inputs_and_labels_with_properties_global_tokens_culens(text_ids, global_tokens, semantic_tokens,
                                                                  properties_ids, model, eos_token_id):
    """utils/multiple_jsonl.py:403-478: the packed form of the global-token-only property rows."""
    return _pack(*_properties_samples(text_ids, global_tokens, semantic_tokens, properties_ids, model,
                                      eos_token_id, False, False), model.device)


# ----------------------------------------------------------------------------------------------------------
# Cosy batches (data/utils/llm_dataset.py:118-188 after tokenisation)
# ----------------------------------------------------------------------------------------------------------
def cosy_collate(text_tokens: List[Sequence[int]], speech_tokens: List[Sequence[int]], pad_to_max_length=True,
                 max_length=2048):
    """text/speech id lists (prompt already concatenated in front, llm_dataset.py:150-160) -> the dict
    RWKV7CosyLM.forward(batch=...) / RWKV7LM.forward(batch) consume.  int32 tensors, zero right-padding, `skip`
    when the longest sample exceeds max_length; with pad_to_max_length the speech tensor is padded so that the
    longest sample reaches max_length (llm_dataset.py:176-180)."""
    tt = [torch.tensor(list(t), dtype=torch.int32) for t in text_tokens]
    st = [torch.tensor(list(s), dtype=torch.int32) for s in speech_tokens]
    my_max = max(len(a) + len(b) for a, b in zip(tt, st))
    skip = my_max > max_length
    text = torch.nn.utils.rnn.pad_sequence(tt, batch_first=True, padding_value=0)
    speech = torch.nn.utils.rnn.pad_sequence(st, batch_first=True, padding_value=0)
    if pad_to_max_length and not skip and max_length - my_max > 0:
        speech = torch.nn.functional.pad(speech, (0, max_length - my_max), value=0)
    return {"text_token": text, "text_token_len": torch.tensor([len(t) for t in tt], dtype=torch.int32),
            "speech_token": speech, "speech_token_len": torch.tensor([len(s) for s in st], dtype=torch.int32),
            "skip": skip}


def synthetic_cosy_batch(B: int, n_text: int = 126, n_speech: int = 384, text_vocab=65548, speech_vocab=6561,
                         seed: int = 1234):
    """BASELINE.json configs[0]: [sos, 126 text ids, task, 384 speech ids] = 512 positions."""
    g = torch.Generator().manual_seed(seed)
    text = [torch.randint(0, text_vocab, (n_text,), generator=g).tolist() for _ in range(B)]
    speech = [torch.randint(0, speech_vocab, (n_speech,), generator=g).tolist() for _ in range(B)]
    return cosy_collate(text, speech, pad_to_max_length=False)


# ----------------------------------------------------------------------------------------------------------
# XY batches (utils/xy_data_processor.py:30-130 == data/utils/collator.py:8-132 == train_xy_llm.py:90-215)
# ----------------------------------------------------------------------------------------------------------
class XYDataProcessor:
    """utils/xy_data_processor.py:7-130 with pre-tokenised text.  `text_tokens[i]` must already be the ids of
    "[S0]" + text + "[CTL0]" (:43,52); `audio_tokens[i]` is [num_channels][T2]."""

    def __init__(self, text_vocab_size, num_channels, text_shift_size, speech_vocab_size):
        self.num_channels = num_channels
        self.text_shift_size = text_shift_size
        self.speech_vocab_size = speech_vocab_size
        self.audio_token_pad_token_id = speech_vocab_size - 1
        self.text_token_pad_token_id = text_vocab_size - 1
        self.ignore_id = -100

    def process_batch(self, text_tokens: List[Sequence[int]], audio_tokens: List[Sequence[Sequence[int]]]):
        C, apad, tpad, ign = self.num_channels, self.audio_token_pad_token_id, self.text_token_pad_token_id, self.ignore_id
        ids_l, lab_l, msk_l = [], [], []
        for tt, at in zip(text_tokens, audio_tokens):
            text = torch.tensor(list(tt), dtype=torch.long)
            speech = torch.tensor(at, dtype=torch.long).clone()
            speech[0, :] += self.text_shift_size
            T1, T2 = text.numel(), speech.shape[1]
            total = T1 + T2 + C - 1
            input_ids = torch.full((total, C), apad, dtype=torch.long)
            labels = torch.full((total, C), ign, dtype=torch.long)
            input_ids[:T1, 0] = text
            input_ids[T1:, 0] = tpad
            for ch in range(C):  # channel ch is delayed by ch steps
                input_ids[T1 + ch: T1 + ch + T2, ch] = speech[ch]
            labels[:-1, :] = input_ids[1:, :]
            labels[:T1 - 1, :] = ign
            labels[labels == apad] = ign
            labels[labels == tpad] = ign
            for i in range(C):
                labels[T1 + T2 - 1 + i, i] = tpad if i == 0 else apad
            ids_l.append(input_ids)
            lab_l.append(labels)
            msk_l.append(torch.ones(total, dtype=torch.long))
        if not ids_l:
            return {}
        L = max(x.shape[0] for x in ids_l)
        for i in range(len(ids_l)):
            pad = L - ids_l[i].shape[0]
            if pad > 0:
                p = torch.full((pad, C), apad, dtype=torch.long)
                p[:, 0] = tpad
                ids_l[i] = torch.cat([ids_l[i], p], 0)
                lab_l[i] = torch.cat([lab_l[i], torch.full((pad, C), ign, dtype=torch.long)], 0)
                msk_l[i] = torch.cat([msk_l[i], torch.zeros(pad, dtype=torch.long)], 0)
        return {"input_ids": torch.stack(ids_l), "labels": torch.stack(lab_l), "attention_mask": torch.stack(msk_l)}


def xy_data_collator(features, num_channels, text_shift_size, speech_vocab_size, text_vocab_size):
    """data/utils/collator.py:8-132 after tokenisation: `features[i]` = {"text": ids of "[S0]" + text + "[CTL0]",
    "codes": [num_channels][T2] RVQ codes as the XY codec returns them (channel 0 NOT yet shifted)}; a feature with
    no text or no codes is skipped (:20-22) and an empty batch returns {} (:40-41).  The reference wrote this collator
    and `XYDataProcessor.process_batch` independently (and ships `verify_collator_logic.py` to compare them); here both
    names run the one vectorised builder above, and each is pinned against its own reference function."""
    kept = [f for f in features if f.get("text") is not None and len(f["text"]) > 0 and f.get("codes") is not None]
    proc = XYDataProcessor(text_vocab_size, num_channels, text_shift_size, speech_vocab_size)
    return proc.process_batch([f["text"] for f in kept], [f["codes"] for f in kept])


def synthetic_xy_batch(B: int, T1: int = 128, T2: int = 8057, num_channels=8, text_vocab=66661, speech_vocab=1025,
                       text_shift=65536, seed: int = 1234):
    """BASELINE.json configs[3]: 128 text steps + 8057 frames x 8 channels (+7 delay) = 8192 steps."""
    g = torch.Generator().manual_seed(seed)
    text = [torch.randint(0, 65536, (T1,), generator=g).tolist() for _ in range(B)]
    audio = [torch.randint(0, speech_vocab - 1, (num_channels, T2), generator=g).tolist() for _ in range(B)]
    return XYDataProcessor(text_vocab, num_channels, text_shift, speech_vocab).process_batch(text, audio)
