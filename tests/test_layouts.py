"""CPU: rwkvtts_amd/layouts.py and the Cosy loss helpers against tests/golden/layouts.npz, which holds the outputs of
the REFERENCE's own batch builders (oracle/pin_layouts.py lists which function produced which entry).  Integer /
index work: bit-exact."""
import types

import torch

from conftest import load_golden
from rwkvtts_amd import layouts as L, losses


class Duck(torch.nn.Module):
    def __init__(self, g):
        super().__init__()
        emb = lambda k: torch.nn.Embedding.from_pretrained(g[k], freeze=True)
        self.text_embedder, self.global_embedder = emb("duck.text_embedder.weight"), emb("duck.global_embedder.weight")
        self.tts_tag_embedder = emb("duck.tts_tag_embedder.weight")
        self.model = types.SimpleNamespace(embeddings=emb("duck.model.embeddings.weight"))

    @property
    def device(self):
        return torch.device("cpu")


TEXT = [[5, 6, 7], [9, 10, 11, 12, 13, 14], [3]]
GLOB = [[1, 2, 3, 4], [7, 8, 9, 10], [11, 12, 13, 14]]
SEM = [[20, 21, 22, 23, 24], [30, 31], [40, 41, 42]]


def test_spark_create_inputs_left_pad():
    g = load_golden("layouts.npz")
    e, m = L.create_inputs(TEXT, GLOB, SEM, Duck(g))
    assert torch.equal(e, g["ci.emb"].to(e.dtype)) and torch.equal(m, g["ci.mask"])
    assert m[2].tolist() == [0] * 4 + [1] * 11  # shortest sample is padded on the LEFT


def test_spark_process_single_batch_and_culens():
    g = load_golden("layouts.npz")
    duck = Duck(g)
    batch = {k[len("psb.in."):]: v for k, v in g.items() if k.startswith("psb.in.")}
    o = L.process_single_batch(batch, duck, eos_token_id=100)
    for k in ("input_embs", "attention_mask", "labels"):
        assert torch.equal(o[k], g["psb." + k].to(o[k].dtype)), k
    # labels are pre-shifted: last position predicts EOS, position before the first semantic id predicts it
    assert o["labels"][0, -1] == 100 and o["labels"][0, -6:-1].tolist() == SEM[0]
    o = L.process_single_batch_culens(batch, duck, eos_token_id=100, max_cu_seqlens=30)
    for k in ("input_embs", "labels", "cu_seqlens"):
        assert torch.equal(o[k], g["psbc." + k]), k
    assert o["cu_seqlens"].tolist()[0] == 0 and o["input_embs"].shape[0] == 1


def test_spark_create_inputs_and_labels_right_pad_aligned():
    g = load_golden("layouts.npz")
    o = L.create_inputs_and_labels(TEXT, GLOB, SEM, Duck(g), 100)
    for k in ("input_embs", "labels", "attention_mask"):
        assert torch.equal(o[k], g["cil." + k]), k


def test_spark_create_inputs_and_labels_culens_packed():
    """utils/multiple_jsonl.py:76-136 -- the row SURVEY section 8 a10 / N1 cite by name."""
    g = load_golden("layouts.npz")
    o = L.create_inputs_and_labels_culens(TEXT, GLOB, SEM, Duck(g), 100)
    for k in ("input_embs", "labels", "cu_seqlens"):
        assert torch.equal(o[k], g["cilc." + k]) and o[k].dtype == g["cilc." + k].dtype, k
    lens = [3 + len(t) + len(gl) + len(s) + 1 for t, gl, s in zip(TEXT, GLOB, SEM)]
    assert o["cu_seqlens"].tolist() == [0, lens[0], lens[0] + lens[1], sum(lens)]
    assert o["input_embs"].shape[:2] == (1, sum(lens)) and o["labels"].shape == (1, sum(lens))
    # packed == the right-padded rows with the padding removed
    p = L.create_inputs_and_labels(TEXT, GLOB, SEM, Duck(g), 100)
    keep = p["attention_mask"].bool()
    assert torch.equal(o["input_embs"][0], p["input_embs"][keep]) and torch.equal(o["labels"][0], p["labels"][keep])


def _prop_ids(g):
    return [g[f"props.ids{i}"].tolist() for i in range(3)]


def test_spark_properties_layouts_two_rows_per_utterance():
    """utils/multiple_jsonl.py:139-311: plain row + property-prefixed row per utterance; the property rows carry
    labels on the global tokens too (:198-210)."""
    g = load_golden("layouts.npz")
    duck, props = Duck(g), _prop_ids(g)
    o = L.create_inputs_and_labels_with_properties(TEXT, GLOB, SEM, props, duck, 100)
    for k in ("input_embs", "labels", "attention_mask"):
        assert torch.equal(o[k], g["cilp." + k]), k
    assert o["input_embs"].shape[0] == 6
    P, T0 = len(props[0]), len(TEXT[0])
    row = o["labels"][1]     # utterance 0 behind its property tokens
    assert row[:P + 1 + T0 + 1].eq(-100).all()
    assert row[P + 2 + T0: P + 2 + T0 + 4].tolist() == GLOB[0] and row[P + 2 + T0 + 4] == -100
    assert row[P + 2 + T0 + 5: P + 2 + T0 + 5 + 6].tolist() == SEM[0] + [100]
    assert o["labels"][0][: 3 + T0 + 4].eq(-100).all()           # the plain row: nothing before the semantic ids
    o = L.create_inputs_and_labels_with_properties_culens(TEXT, GLOB, SEM, props, duck, 100)
    for k in ("input_embs", "labels", "cu_seqlens"):
        assert torch.equal(o[k], g["cilpc." + k]), k
    assert o["cu_seqlens"].numel() == 7


def test_spark_properties_global_tokens_only_layouts():
    """utils/multiple_jsonl.py:313-478: one property row per utterance, loss on the global tokens only."""
    g = load_golden("layouts.npz")
    duck, props = Duck(g), _prop_ids(g)
    o = L.create_inputs_and_labels_with_properties_global_tokens(TEXT, GLOB, SEM, props, duck, 100)
    for k in ("input_embs", "labels", "attention_mask"):
        assert torch.equal(o[k], g["cilpg." + k]), k
    assert o["input_embs"].shape[0] == 3
    assert sorted(o["labels"][0][o["labels"][0] != -100].tolist()) == GLOB[0]
    o = L.create_inputs_and_labels_with_properties_global_tokens_culens(TEXT, GLOB, SEM, props, duck, 100)
    for k in ("input_embs", "labels", "cu_seqlens"):
        assert torch.equal(o[k], g["cilpgc." + k]), k


def test_xy_data_collator_pinned_separately():
    """data/utils/collator.py:8-132 (the reference compares it with XYDataProcessor in verify_collator_logic.py:99-181):
    its own golden entry, a feature without audio skipped, empty batch -> {}."""
    g = load_golden("layouts.npz")
    audio = [g[f"xy.audio{i}"].tolist() for i in range(3)]
    feats = [{"text": [400] + TEXT[i] + [401], "codes": audio[i]} for i in range(3)]
    feats.insert(1, {"text": [400, 1, 2, 401], "codes": None})
    o = L.xy_data_collator(feats, 4, 450, 16, 500)
    for k in ("input_ids", "labels", "attention_mask"):
        assert torch.equal(o[k], g["xyc." + k]), k
        assert torch.equal(g["xyc." + k], g["xy." + k]), k    # the reference's two builders agree on this batch
    assert L.xy_data_collator([], 4, 450, 16, 500) == {}


def test_xy_delay_pattern_and_labels():
    g = load_golden("layouts.npz")
    audio = [g[f"xy.audio{i}"].tolist() for i in range(3)]
    o = L.XYDataProcessor(500, 4, 450, 16).process_batch([[400] + t + [401] for t in TEXT], audio)
    for k in ("input_ids", "labels", "attention_mask"):
        assert torch.equal(o[k], g["xy." + k]), k
    T1 = len(TEXT[0]) + 2
    assert o["input_ids"][0, T1, 0] == audio[0][0][0] + 450       # channel 0 is shifted by text_shift_size
    assert o["input_ids"][0, T1 + 2, 2] == audio[0][2][0]         # channel k delayed by k steps
    assert L.XYDataProcessor(500, 4, 450, 16).process_batch([], []) == {}


def test_cosy_collate():
    g = load_golden("layouts.npz")
    o = L.cosy_collate([[1, 2] + t for t in TEXT], [[50, 51] + s for s in SEM], True, 24)
    for k in ("text_token", "text_token_len", "speech_token", "speech_token_len"):
        assert torch.equal(o[k], g["cosy." + k]) and o[k].dtype == torch.int32, k
    assert o["skip"] is False
    assert L.cosy_collate([[1] * 30], [[2] * 30], True, 24)["skip"] is True


def test_label_smoothing_and_accuracy_values():
    g = load_golden("layouts.npz")
    for sm, nl in ((0.0, True), (0.1, True), (0.1, False)):
        got = losses.label_smoothing_kl(g["ls.logits"], g["ls.target"], 11, -1, sm, nl)
        assert abs(got.item() - g[f"ls.loss_{sm}_{int(nl)}"].item()) < 1e-6
    acc = losses.th_accuracy(g["ls.logits"].view(-1, 11), g["ls.target"], -1)
    assert abs(acc.item() - g["ls.acc"].item()) < 1e-7


def test_synthetic_batches_have_the_baseline_shapes():
    b = L.synthetic_xy_batch(1, T1=128, T2=8057)
    assert tuple(b["input_ids"].shape) == (1, 8192, 8) and int(b["attention_mask"].sum()) == 8192
    c = L.synthetic_cosy_batch(2)
    assert c["text_token"].shape == (2, 126) and c["speech_token"].shape == (2, 384)  # 1+126+1+384 = 512
