"""GPU (-m gpu): the Spark LM on SHAPES the shipped configurations do not have -- widths that are not multiples of 256 or 1024 (the own
GEMMs and the fused row kernels have tile grids), low-rank sizes that are not multiples of 32, odd vocabularies, sequence lengths that are
not multiples of 16 / 32 (scalar kernels, ragged tails), random left padding -- two layers each, against oracle/rwkv7_ref: fp32 logits
within 1e-3 (north_star's bar), argmax ids, loss; one bf16 training step: every parameter gradient as relative L2.  A shape a kernel
cannot take must either fall back to a path that can or raise; it must not return something else.          spark_llm.py:105-172"""
import pytest
import torch

from oracle import rwkv7_ref as R
from rwkvtts_amd import backbone
from test_fullshape_gpu import _perturb_norms, _rel_errors, _summary

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

CASES = [  # D, ranks (w, a, v, g), V, B, T, pad
    (128, (16, 16, 16, 32), 257, 2, 33, 5),
    (192, (24, 40, 16, 48), 1000, 3, 100, 17),
    (320, (64, 32, 32, 96), 8193, 1, 64, 0),
    (640, (96, 64, 48, 160), 513, 2, 160, 31),
    (2560, (128, 128, 96, 320), 1025, 1, 48, 7),      # the width of RWKV7-2.9B
    (1024, (64, 64, 32, 128), 8193, 3, 7, 2),         # T < 16
    (768, (64, 64, 32, 128), 6562, 2, 272, 0),        # 0.1B width, T = 8.5 chunks of 32
    (256, (32, 32, 32, 64), 70, 4, 16, 15),           # a row that is padding but for one position
]


@pytest.mark.parametrize("D,ranks,V,B,T,pad", CASES)
def test_spark_lm_on_unshipped_shapes_vs_oracle(D, ranks, V, B, T, pad):
    from rwkvtts_amd.spark_llm import RWKV7ForSpeech, RWKV7SpeechConfig
    rw, ra, rv, rg = ranks
    cfg = RWKV7SpeechConfig(hidden_size=D, num_hidden_layers=2, vocab_size=V, decay_low_rank_dim=rw, a_low_rank_dim=ra, v_low_rank_dim=rv,
                            gate_low_rank_dim=rg, text_vocab_size=64, audio_global_vocab_size=64)
    model = RWKV7ForSpeech(cfg).init_weights(seed=D + T)
    _perturb_norms(model, 1)
    with torch.no_grad():
        model.lm_head.weight.normal_(0, 0.05, generator=torch.Generator().manual_seed(2))
    p = {k: v.detach().clone() for k, v in model.state_dict().items()}
    rcfg = R.RefConfig(hidden_size=D, num_hidden_layers=2, vocab_size=V, decay_low_rank_dim=rw, a_low_rank_dim=ra, v_low_rank_dim=rv,
                       gate_low_rank_dim=rg)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(B, T, D, generator=g) * 0.5
    mask = torch.ones(B, T, dtype=torch.long)
    labels = torch.randint(0, V, (B, T), generator=g)
    if pad:
        mask[-1, :pad] = 0
        labels[-1, :pad] = -100
    R.pick_threads()
    skip = ("text_embedder.weight", "global_embedder.weight", "tts_tag_embedder.weight", "model.embeddings.weight")
    pr = {k: v.clone().requires_grad_(k not in skip) for k, v in p.items()}
    xr = x.clone().requires_grad_(True)
    loss_o, logits_o, _ = R.spark_forward(pr, rcfg, xr, mask, labels)
    loss_o.backward()
    logits_o = logits_o.detach()
    m32 = model.to(DEV).eval()
    with torch.no_grad():
        out = m32(inputs_embeds=x.to(DEV), attention_mask=mask.to(DEV), labels=labels.to(DEV))
    valid = mask.bool()
    err = (out.logits.float().cpu() - logits_o)[valid].abs().max().item()
    assert err < 1e-3, f"fp32 logits differ from the oracle by {err}"
    top2 = logits_o[valid].topk(2, -1).values
    clear = (top2[:, 0] - top2[:, 1]) > 1e-4                 # ids where the oracle's own margin is above the fp32 noise
    assert torch.equal(out.logits.argmax(-1).cpu()[valid][clear], logits_o.argmax(-1)[valid][clear]), "greedy ids differ"
    assert abs(out.loss.item() - loss_o.item()) < 1e-4 * max(1.0, abs(loss_o.item()))
    m16 = m32.to(torch.bfloat16).train()
    m16.dropout.p = 0.0
    x16 = x.to(DEV, torch.bfloat16).requires_grad_(True)
    out16 = m16(inputs_embeds=x16, attention_mask=mask.to(DEV), labels=labels.to(DEV))
    out16.loss.backward()
    assert abs(out16.loss.item() - loss_o.item()) < 2e-2 * abs(loss_o.item())
    rels = _rel_errors(dict(m16.named_parameters()), pr, skip)
    rels["inputs_embeds"] = ((x16.grad.float().cpu() - xr.grad).norm() / xr.grad.norm()).item()
    median, top = _summary(rels)
    print(f"D={D} ranks={ranks} V={V} B={B} T={T} pad={pad}: fp32 logits max|d| {err:.2e}; bf16 gradient rel. L2 median {median:.2e}, worst {top[:2]}")
    assert median < 4e-2 and top[0][1] < 0.2, f"relative L2 gradient errors: median {median:.3e}, worst five {top}"
