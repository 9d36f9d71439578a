"""oracle/pin_layouts.py -- AUTHORING-CONTAINER ONLY (needs /root/reference).

Runs the reference's own batch builders and loss helpers on toy inputs and writes their outputs to
tests/golden/layouts.npz:
    create_inputs                       inference/rwkv7speech_inference.py:35-67
    process_single_batch / _culens      data/utils/spark_dataset.py:163-239 / 111-162
    create_inputs_and_labels[_culens]   utils/multiple_jsonl.py:4-74 / 76-136
    create_inputs_and_labels_with_properties[_culens]                 utils/multiple_jsonl.py:139-234 / 236-311
    create_inputs_and_labels_with_properties_global_tokens[_culens]   utils/multiple_jsonl.py:313-400 / 403-478
    xy_data_collator                    data/utils/collator.py:8-132
    XYDataProcessor.process_batch       utils/xy_data_processor.py:30-130
    collate_fn (Cosy)                   data/utils/llm_dataset.py:118-188
    LabelSmoothingLoss, th_accuracy     cosyvoice/transformer/label_smoothing_loss.py:68-96, cosyvoice/utils/common.py:76-95
and checks rwkvtts_amd/layouts.py + rwkvtts_amd/losses.py against them (bit-exact for ids/masks/labels and for the
embedding concatenations).   Usage: python oracle/pin_layouts.py [--write]
"""
import argparse
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_import  # noqa: E402
from rwkvtts_amd import layouts as L, losses  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden", "layouts.npz")


class FakeTok:
    """text is a string of space separated ints; encode() returns them (a stand-in for the RWKV world tokenizer,
    which is CPU preprocessing outside the path)."""
    vocab_size = 500

    def encode(self, text, add_special_tokens=False):
        if text.startswith("SPCT_"):   # the property string "SPCT_0SPCT_15SPCT_46..." -> one id (300 + n) per SPCT_n
            return [300 + int(t) for t in text.split("SPCT_")[1:]]
        return [int(t) for t in text.split()]

    def __call__(self, text, return_tensors="pt"):
        # "[S0]1 2 3[CTL0]" -> ids with 400 / 401 as the [S0] / [CTL0] specials
        body = text.replace("[S0]", "").replace("[CTL0]", "")
        return types.SimpleNamespace(input_ids=torch.tensor([[400] + self.encode(body) + [401]]))


class Duck(torch.nn.Module):
    def __init__(self, D=8):
        super().__init__()
        g = torch.Generator().manual_seed(0)
        mk = lambda n: torch.nn.Embedding.from_pretrained(torch.randn(n, D, generator=g), freeze=True)
        self.text_embedder, self.global_embedder, self.tts_tag_embedder = mk(500), mk(64), mk(3)
        self.model = types.SimpleNamespace(embeddings=mk(101))

    @property
    def device(self):
        return torch.device("cpu")


def eq(a, b, what):
    assert a.shape == b.shape and torch.equal(a, b), what
    print(f"  OK  {what}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--write", action="store_true")
    a = ap.parse_args()
    sp = ref_import.import_spark_layout()
    tok, duck = FakeTok(), Duck()
    gold = {"duck." + k: v for k, v in duck.state_dict().items()}
    gold["duck.model.embeddings.weight"] = duck.model.embeddings.weight
    texts = ["5 6 7", "9 10 11 12 13 14", "3"]
    text_ids = [tok.encode(t) for t in texts]
    glob = [[1, 2, 3, 4], [7, 8, 9, 10], [11, 12, 13, 14]]
    sem = [[20, 21, 22, 23, 24], [30, 31], [40, 41, 42]]
    print("[Spark] create_inputs")
    e_ref, m_ref = sp.create_inputs(texts, glob, sem, tok, duck)
    e, m = L.create_inputs(text_ids, glob, sem, duck)
    eq(e, e_ref.to(e.dtype), "create_inputs embeddings"); eq(m, m_ref, "create_inputs mask")
    gold["ci.emb"], gold["ci.mask"] = e_ref, m_ref

    def lpad(seqs):
        n = max(len(s) for s in seqs)
        ids = torch.zeros(len(seqs), n, dtype=torch.long)
        msk = torch.zeros(len(seqs), n, dtype=torch.long)
        for i, s in enumerate(seqs):
            ids[i, n - len(s):] = torch.tensor(s)
            msk[i, n - len(s):] = 1
        return ids, msk

    batch = {}
    batch["input_ids"], batch["attention_mask_input_ids"] = lpad(text_ids)
    batch["global_tokens_ids"], batch["global_tokens_attention_mask"] = lpad(glob)
    batch["semantic_tokens_ids"], batch["semantic_tokens_attention_mask"] = lpad(sem)
    for k, v in batch.items():
        gold["psb.in." + k] = v
    print("[Spark] process_single_batch")
    r = sp.spark_dataset.process_single_batch(batch, duck, eos_token_id=100)
    o = L.process_single_batch(batch, duck, eos_token_id=100)
    for k in ("input_embs", "attention_mask", "labels"):
        eq(o[k], r[k].to(o[k].dtype), "process_single_batch." + k)
        gold["psb." + k] = r[k]
    print("[Spark] process_single_batch_culens")
    r = sp.spark_dataset.process_single_batch_culens(batch, duck, eos_token_id=100, max_cu_seqlens=30)
    o = L.process_single_batch_culens(batch, duck, eos_token_id=100, max_cu_seqlens=30)
    for k in ("input_embs", "labels", "cu_seqlens"):
        eq(o[k], r[k], "process_single_batch_culens." + k)
        gold["psbc." + k] = r[k]
    print("[Spark] create_inputs_and_labels")
    r = sp.multiple_jsonl.create_inputs_and_labels({"text": texts, "global_tokens": glob, "semantic_tokens": sem}, tok,
                                                   duck, 100, torch.device("cpu"))
    o = L.create_inputs_and_labels(text_ids, glob, sem, duck, 100)
    for k in ("input_embs", "labels", "attention_mask"):
        eq(o[k], r[k], "create_inputs_and_labels." + k)
        gold["cil." + k] = r[k]

    print("[Spark] create_inputs_and_labels_culens")
    mj = sp.multiple_jsonl
    dev = torch.device("cpu")
    r = mj.create_inputs_and_labels_culens({"text": texts, "global_tokens": glob, "semantic_tokens": sem}, tok, duck, 100, dev)
    o = L.create_inputs_and_labels_culens(text_ids, glob, sem, duck, 100)
    for k in ("input_embs", "labels", "cu_seqlens"):
        eq(o[k], r[k], "create_inputs_and_labels_culens." + k)
        gold["cilc." + k] = r[k]
    # properties-prefixed layouts: the reference builds the property string and tokenises it; ours takes the ids
    props = dict(age=["child", "youth-adult", "elderly"], gender=["female", "male", "female"],
                 emotion=["HAPPY", "neutral", "WHISPER"], pitch=[260.0, 120.0, 230.0], speed=[3.0, 4.2, 5.5])
    from utils.properties_util import convert_properties_to_tokens
    prop_ids = [tok.encode(convert_properties_to_tokens(props["age"][i], props["gender"][i], props["emotion"][i],
                                                        props["pitch"][i], props["speed"][i])) for i in range(3)]
    for i, pi in enumerate(prop_ids):
        gold[f"props.ids{i}"] = torch.tensor(pi)
    pbatch = {"text": texts, "global_tokens": glob, "semantic_tokens": sem, **props}
    mj.global_debug = False
    for name, short, keys in (("create_inputs_and_labels_with_properties", "cilp", ("input_embs", "labels", "attention_mask")),
                              ("create_inputs_and_labels_with_properties_culens", "cilpc", ("input_embs", "labels", "cu_seqlens")),
                              ("create_inputs_and_labels_with_properties_global_tokens", "cilpg",
                               ("input_embs", "labels", "attention_mask")),
                              ("create_inputs_and_labels_with_properties_global_tokens_culens", "cilpgc",
                               ("input_embs", "labels", "cu_seqlens"))):
        print(f"[Spark] {name}")
        r = getattr(mj, name)(pbatch, tok, duck, 100, dev)
        o = getattr(L, name)(text_ids, glob, sem, prop_ids, duck, 100)
        for k in keys:
            eq(o[k], r[k], f"{name}.{k}")
            gold[f"{short}.{k}"] = r[k]

    print("[XY] XYDataProcessor.process_batch")
    from utils.xy_data_processor import XYDataProcessor as RefXY
    C = 4
    g = torch.Generator().manual_seed(1)
    audio = [torch.randint(0, 15, (C, n), generator=g).tolist() for n in (6, 3, 9)]
    r = RefXY(tok, C, 450, 16).process_batch({"text": texts, "audio_tokens": audio})
    xy_text = [[400] + t + [401] for t in text_ids]
    o = L.XYDataProcessor(tok.vocab_size, C, 450, 16).process_batch(xy_text, audio)
    for k in ("input_ids", "labels", "attention_mask"):
        eq(o[k], r[k], "XY." + k)
        gold["xy." + k] = r[k]
    gold["xy.audio0"], gold["xy.audio1"], gold["xy.audio2"] = [torch.tensor(x) for x in audio]

    print("[XY] xy_data_collator")
    from data.utils.collator import xy_data_collator as ref_collator

    class FakeCodec:   # stands in for the XY audio tokenizer: the "audio" array already holds the codes
        def encode(self, wavs, device=None):
            return {"codes_list": [w.long() for w in wavs]}

    feats = [{"json": {"text": texts[i]}, "audio": {"array": np.asarray(audio[i], dtype=np.int64)}} for i in range(3)]
    feats.insert(1, {"json": {"text": "1 2"}, "audio": {}})   # no audio: skipped by the collator
    r = ref_collator(feats, tok, FakeCodec(), C, 450, 16, dev)
    ours = [{"text": xy_text[i], "codes": audio[i]} for i in range(3)]
    ours.insert(1, {"text": [400, 1, 2, 401], "codes": None})
    o = L.xy_data_collator(ours, C, 450, 16, tok.vocab_size)
    for k in ("input_ids", "labels", "attention_mask"):
        eq(o[k], r[k], "xy_data_collator." + k)
        gold["xyc." + k] = r[k]
    assert ref_collator([], tok, FakeCodec(), C, 450, 16, dev) == {} and L.xy_data_collator([], C, 450, 16, 500) == {}

    print("[Cosy] collate_fn")
    from data.utils import llm_dataset
    samples = [dict(text=texts[i], prompt_text="1 2", tts_speech_tokens=sem[i], llm_prompt_speech_token=[50, 51])
               for i in range(3)]
    r = llm_dataset.collate_fn(samples, tok, pad_to_max_length=True, max_length=24, drop_prompt_audio_rate=-0.1)
    o = L.cosy_collate([[1, 2] + t for t in text_ids], [[50, 51] + s for s in sem], True, 24)
    for k in ("text_token", "text_token_len", "speech_token", "speech_token_len"):
        eq(o[k], r[k], "cosy_collate." + k)
        gold["cosy." + k] = r[k]
    assert o["skip"] == r["skip"]

    print("[Cosy] LabelSmoothingLoss / th_accuracy")
    from cosyvoice.transformer.label_smoothing_loss import LabelSmoothingLoss
    from cosyvoice.utils.common import th_accuracy
    logits = torch.randn(2, 7, 11, generator=g)
    target = torch.randint(0, 11, (2, 7), generator=g)
    target[0, :3] = -1
    gold["ls.logits"], gold["ls.target"] = logits, target
    for sm, nl in ((0.0, True), (0.1, True), (0.1, False)):
        want = LabelSmoothingLoss(11, -1, sm, nl)(logits, target)
        got = losses.label_smoothing_kl(logits, target, 11, -1, sm, nl)
        assert abs(want.item() - got.item()) < 1e-6, (want, got)
        gold[f"ls.loss_{sm}_{int(nl)}"] = want
        print(f"  OK  LabelSmoothingLoss smoothing={sm} normalize_length={nl}: {want.item():.6f}")
    acc = th_accuracy(logits.view(-1, 11), target, -1)
    assert abs(acc.item() - losses.th_accuracy(logits.view(-1, 11), target, -1).item()) < 1e-7
    gold["ls.acc"] = acc
    if a.write:
        np.savez_compressed(GOLD, **{k: v.detach().numpy() for k, v in gold.items()})
        print(f"wrote {GOLD} ({os.path.getsize(GOLD) / 1024:.0f} KiB)")
    print("layouts pinned against the reference's batch builders: all checks passed")


if __name__ == "__main__":
    main()
