"""GPU (-m gpu): the oracle at the headline models' REAL depth and width.

The full-size configuration tests (test_configs_gpu.py) can only check properties at B x L = 8 x 4096; the oracle cannot run that.  What it
can run in a minute is the real MODEL at a short sequence: every layer, every width, every low-rank size of the shipped configurations.

  (a) RWKV7-0.4B Spark (24 layers, D = 1024, H = 16, V = 8193), B = 2, L = 256, one row left-padded: fp32 logits within 1e-3 of
      oracle/rwkv7_ref.spark_forward, argmax ids equal, loss equal; one bf16 training step: every parameter gradient (and the input
      gradient) against the oracle's autograd, relative L2 per tensor.                      spark_llm.py:105-172
  (b) RWKV7-1.5B XY (24 layers, D = 2048, H = 32, ranks 96/96/64/256, 8 channels, V0 = 66 661), B = 1, 128 steps: the eight logit
      tensors within 1e-3, the summed loss, and the gradients of a bf16 step.                 xy_llm.py:189-257
  (d) the packed path: one cu_seqlens row with lengths that are NOT multiples of 32 (and one that is), bf16, native chunk ranges:
      loss and every parameter gradient against the oracle run on each sequence alone.       train_spark_rwkv7speech.py:238-239
"""
import math

import pytest
import torch

from oracle import rwkv7_ref as R
from rwkvtts_amd import backbone
from rwkvtts_amd import layouts as L

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _perturb_norms(model, seed):
    """init_weights leaves every norm at (1, 0) and every bias at 0: a transposed or skipped affine would go unnoticed."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for m in model.modules():
            if isinstance(m, (torch.nn.LayerNorm, torch.nn.GroupNorm)):
                m.weight.add_(torch.randn(m.weight.shape, generator=g) * 0.1)
                m.bias.add_(torch.randn(m.bias.shape, generator=g) * 0.1)


def _rel_errors(named, pr, skip=()):
    rels = {}
    for k, v in pr.items():
        if v.grad is None or k in skip:
            continue
        gh = named[k].grad
        assert gh is not None, f"no gradient reached {k}"
        gh, ref = gh.float().cpu(), v.grad
        if gh.dim() == 2 and ref.abs().sum(-1).eq(0).any():      # embedding tables: compare the rows that were touched
            rows = ref.abs().sum(-1) > 0
            gh, ref = gh[rows], ref[rows]
        rels[k] = ((gh - ref).norm() / ref.norm().clamp(min=1e-12)).item()
    return rels


def _summary(rels):
    vals = sorted(rels.values())
    return vals[len(vals) // 2], sorted(rels.items(), key=lambda kv: -kv[1])[:5]


@pytest.mark.timeout(1500)
def test_spark_0p4b_full_depth_B2_L256_left_padded_logits_argmax_loss_and_gradients_vs_oracle():
    from rwkvtts_amd.spark_llm import RWKV7ForSpeech, RWKV7SpeechConfig
    base = backbone.config_0p4b()
    kw = {k: v for k, v in base.to_dict().items() if k in backbone.RWKV7Config.__dataclass_fields__ and k != "extra"}
    cfg = RWKV7SpeechConfig(**kw)
    assert (cfg.num_hidden_layers, cfg.hidden_size, cfg.num_heads, cfg.vocab_size) == (24, 1024, 16, 8193)
    model = RWKV7ForSpeech(cfg).init_weights(seed=3)
    _perturb_norms(model, 4)
    with torch.no_grad():
        model.lm_head.weight.normal_(0, 0.05, generator=torch.Generator().manual_seed(5))
    p = {k: v.detach().clone() for k, v in model.state_dict().items()}
    rcfg = R.RefConfig(hidden_size=1024, num_hidden_layers=24, vocab_size=8193)
    B, T, PAD = 2, 256, 41
    g = torch.Generator().manual_seed(6)
    x = torch.randn(B, T, 1024, generator=g) * 0.5
    mask = torch.ones(B, T, dtype=torch.long)
    mask[1, :PAD] = 0                                   # left padding, as inference/rwkv7speech_inference.py:35-67 builds it
    labels = torch.randint(0, 8193, (B, T), generator=g)
    labels[1, :PAD] = -100
    # the oracle: fp32 eager, per-token torch scan, autograd
    R.pick_threads()
    skip = ("text_embedder.weight", "global_embedder.weight", "tts_tag_embedder.weight", "model.embeddings.weight")
    pr = {k: v.clone().requires_grad_(k not in skip) for k, v in p.items()}
    xr = x.clone().requires_grad_(True)
    loss_o, logits_o, _ = R.spark_forward(pr, rcfg, xr, mask, labels)
    loss_o.backward()
    logits_o = logits_o.detach()
    # (i) fp32 on the HIP path
    m32 = model.to(DEV).eval()
    with torch.no_grad():
        out = m32(inputs_embeds=x.to(DEV), attention_mask=mask.to(DEV), labels=labels.to(DEV))
    valid = mask.bool()
    err = (out.logits.float().cpu() - logits_o)[valid].abs().max().item()
    assert err < 1e-3, f"fp32 logits differ from the oracle by {err}"
    assert torch.equal(out.logits.argmax(-1).cpu()[valid], logits_o.argmax(-1)[valid]), "greedy ids differ"
    assert abs(out.loss.item() - loss_o.item()) < 1e-4
    # (ii) one bf16 training step: chunked MFMA WKV7 pair, fused stages, fused linear + CE
    m16 = m32.to(torch.bfloat16).train()
    m16.dropout.p = 0.0       # spark_llm.py:123-124 drops 2 % of the input embeddings in training: off, the oracle has no RNG twin
    x16 = x.to(DEV, torch.bfloat16).requires_grad_(True)
    out16 = m16(inputs_embeds=x16, attention_mask=mask.to(DEV), labels=labels.to(DEV))
    out16.loss.backward()
    assert abs(out16.loss.item() - loss_o.item()) < 2e-2 * abs(loss_o.item())
    rels = _rel_errors(dict(m16.named_parameters()), pr, skip)
    # (the last padded position predicts the first valid label -- labels are shifted by one, spark_llm.py:154-156 -- so padded
    # positions do carry gradient: compare all of them)
    dx, dxo = x16.grad.float().cpu(), xr.grad
    rels["inputs_embeds"] = ((dx - dxo).norm() / dxo.norm()).item()
    assert (dx[1, :PAD - 1] == 0).all() and (dxo[1, :PAD - 1] == 0).all(), "gradient on padded positions that predict nothing"
    median, top = _summary(rels)
    print(f"0.4B x 24 layers: fp32 logits max|d| {err:.2e}; bf16 gradient rel. L2 error median {median:.2e}, worst five {top} over {len(rels)} tensors")
    assert len(rels) > 24 * 30
    # measured: median 2.2e-2, worst 8.7e-2 (layer 21 a_lora bias); a wrong kernel gives O(1)
    assert median < 4e-2, f"median relative L2 gradient error {median:.3e}; worst five: {top}"
    assert top[0][1] < 0.15, f"relative L2 gradient errors, worst five: {top}"


@pytest.mark.timeout(1500)
def test_xy_1p5b_full_depth_128_steps_logits_loss_and_gradients_vs_oracle():
    from rwkvtts_amd.xy_llm import RWKV7XYConfig, RWKV7XYLM
    base = backbone.config_1p5b()
    kw = {k: v for k, v in base.to_dict().items() if k in backbone.RWKV7Config.__dataclass_fields__ and k != "extra"}
    kw.update(vocab_size=66661)
    cfg = RWKV7XYConfig(speech_vocab_size=1025, num_channels=8, text_shift_size=65536, **kw)
    assert (cfg.num_hidden_layers, cfg.hidden_size, cfg.num_heads) == (24, 2048, 32)
    model = RWKV7XYLM(cfg).init_weights(seed=7)
    _perturb_norms(model, 8)
    with torch.no_grad():
        g = torch.Generator().manual_seed(9)
        for h in model.heads:
            h.weight.normal_(0, 0.05, generator=g)
            h.bias.normal_(0, 0.1, generator=g)
    model.zero_embs()
    p = {k: v.detach().clone() for k, v in model.state_dict().items()}
    rcfg = R.RefConfig(hidden_size=2048, num_hidden_layers=24, vocab_size=0, decay_low_rank_dim=96, a_low_rank_dim=96,
                       v_low_rank_dim=64, gate_low_rank_dim=256)
    batch = L.synthetic_xy_batch(1, T1=16, T2=105, seed=77)          # 16 text + 105 frames + 7 delay steps = 128
    assert batch["input_ids"].shape == (1, 128, 8)
    R.pick_threads()
    skip = ("model.embeddings.weight",)
    pr = {k: v.clone().requires_grad_(k not in skip) for k, v in p.items()}
    loss_o, logits_o = R.xy_forward(pr, rcfg, batch["input_ids"], batch["attention_mask"], batch["labels"], 8, 0.0)
    loss_o.backward()
    m32 = model.to(DEV).eval()
    bd = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in batch.items()}
    with torch.no_grad():
        out = m32(**bd)
    valid = batch["attention_mask"].bool()
    errs = [(a.float().cpu() - b.detach())[valid].abs().max().item() for a, b in zip(out.logits, logits_o)]
    assert max(errs) < 1e-3, errs
    for a, b in zip(out.logits, logits_o):
        assert torch.equal(a.argmax(-1).cpu()[valid], b.argmax(-1)[valid])
    assert abs(out.loss.item() - loss_o.item()) < 1e-3
    m16 = m32.to(torch.bfloat16).train()
    out16 = m16(**bd, use_cache=False)
    out16.loss.backward()
    assert abs(out16.loss.item() - loss_o.item()) < 2e-2 * abs(loss_o.item())
    rels = _rel_errors(dict(m16.named_parameters()), pr, skip)
    median, top = _summary(rels)
    print(f"1.5B XY x 24 layers: fp32 logits max|d| {max(errs):.2e}; bf16 gradient rel. L2 error median {median:.2e}, worst five {top} over {len(rels)} tensors")
    assert len(rels) > 24 * 30 + 16
    # measured: median 2.35e-2, worst 3.7e-2 (layer 23 x_w)
    assert median < 4e-2, f"median relative L2 gradient error {median:.3e}; worst five: {top}"
    assert top[0][1] < 0.08, f"relative L2 gradient errors, worst five: {top}"


@pytest.mark.timeout(900)
def test_packed_row_non_aligned_lengths_loss_and_gradients_vs_oracle_per_sequence():
    from rwkvtts_amd.spark_llm import RWKV7ForSpeech, RWKV7SpeechConfig
    D, NL, V = 256, 4, 513
    cfg = RWKV7SpeechConfig(hidden_size=D, num_hidden_layers=NL, vocab_size=V, text_vocab_size=300, audio_global_vocab_size=64,
                            decay_low_rank_dim=32, a_low_rank_dim=32, v_low_rank_dim=32, gate_low_rank_dim=64)
    model = RWKV7ForSpeech(cfg).init_weights(seed=11)
    _perturb_norms(model, 12)
    with torch.no_grad():
        model.lm_head.weight.normal_(0, 0.05, generator=torch.Generator().manual_seed(13))
    p = {k: v.detach().clone() for k, v in model.state_dict().items()}
    rcfg = R.RefConfig(hidden_size=D, num_hidden_layers=NL, vocab_size=V, decay_low_rank_dim=32, a_low_rank_dim=32, v_low_rank_dim=32,
                       gate_low_rank_dim=64)
    lens = [77, 33, 160, 5, 131, 64]                      # 64: exactly two chunks; the others straddle chunk boundaries
    total = sum(lens)
    g = torch.Generator().manual_seed(14)
    x = torch.randn(1, total, D, generator=g) * 0.5
    labels = torch.randint(0, V, (1, total), generator=g)
    cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32)
    # the model shifts labels by one over the PACKED row (spark_llm.py:154-156): position p predicts labels[p + 1], also across a
    # boundary.  The oracle does the same on the concatenation of the per-sequence hidden states.
    skip = ("text_embedder.weight", "global_embedder.weight", "tts_tag_embedder.weight", "model.embeddings.weight")
    pr = {k: v.clone().requires_grad_(k not in skip) for k, v in p.items()}
    R.pick_threads()
    hs, o = [], 0
    for n in lens:
        h, _ = R.backbone(pr, rcfg, x[:, o:o + n], None, None)     # each sequence alone: zero state, zero shift
        hs.append(h)
        o += n
    logits_o = torch.cat(hs, 1) @ pr["lm_head.weight"].t()
    lab = torch.cat([labels[:, 1:], torch.full_like(labels[:, :1], -100)], 1)
    loss_o = torch.nn.functional.cross_entropy(logits_o.view(total, -1), lab.view(-1), ignore_index=-100)
    loss_o.backward()
    m16 = model.to(DEV).to(torch.bfloat16).train()
    m16.dropout.p = 0.0       # (the 2 % input dropout of spark_llm.py:123-124 has no twin in the oracle)
    for cu_t in (cu.to(DEV), cu):                         # device cu_seqlens (no host read-back) and host cu_seqlens (exact layout)
        m16.zero_grad(set_to_none=True)
        out = m16(inputs_embeds=x.to(DEV, torch.bfloat16), labels=labels.to(DEV), cu_seqlens=cu_t)
        out.loss.backward()
        assert abs(out.loss.item() - loss_o.item()) < 2e-2 * abs(loss_o.item()), (out.loss.item(), loss_o.item())
        rels = _rel_errors(dict(m16.named_parameters()), pr, skip)
        median, top = _summary(rels)
        print(f"packed {lens} ({'device' if cu_t.is_cuda else 'host'} cu_seqlens): bf16 gradient rel. L2 error median {median:.2e}, worst five {top}")
        assert len(rels) > NL * 30
        assert median < 3e-2 and top[0][1] < 0.06, f"median {median:.3e}; worst five: {top}"     # measured 1.4e-2 / 2.5e-2
