"""GPU (-m gpu): BASELINE.json's configurations at their FULL sizes (the other GPU tests run reduced models).

  configs[0]  RWKV7-0.1B Cosy layout, B=2, L=512 -- the one configuration the CPU oracle can run whole: logits (fp32, 1e-3),
              loss, th_accuracy and every parameter gradient (bf16 train step) against oracle/rwkv7_ref.cosy_forward + autograd.
  configs[3]  RWKV7-1.5B XY layout, B=4, L=8192: (i) the chunked WKV7 pair at (4,8192,32,64) against the C oracle on head slices;
              (ii) the 66 661-wide fused linear+CE at 32 768 rows against F.cross_entropy on a row subset; (iii) one training
              step of the 8-channel model (2 layers of the 24 -- depth is covered by bench.py --model 1.5b --layout xy).
  configs[4]  RWKV7-0.4B greedy decode, 24 layers, B=32, prompt 128, 2048 new tokens through GraphDecoder, against the fp32
              twin teacher-forced along the generated ids: ids equal wherever the fp32 top-2 margin exceeds the bf16 noise, the
              recurrent state after 2048 steps within a drift bound.
"""
import pytest
import torch

from oracle import rwkv7_ref as R
from rwkvtts_amd import layouts as L
from rwkvtts_amd import ops
from rwkvtts_amd.synthetic import make_wkv_inputs
from test_wkv7_gpu import _assert_bf16_close, NAMES

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _to(batch, dev):
    return {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in batch.items()}


# ---------------------------------------------------------------------------------------------------------------------
# configs[0]
# ---------------------------------------------------------------------------------------------------------------------
def _cosy_0p1b(seed=5):
    from rwkvtts_amd.cosy_llm import RWKV7CosyConfig, RWKV7CosyLM
    D, NL, TV, SV = 768, 12, 65548, 6561
    cfg = RWKV7CosyConfig(vocab_size=TV, speech_token_size=SV, lsm_weight=0.0, hidden_size=D, num_hidden_layers=NL)
    rcfg = R.RefConfig(vocab_size=0, hidden_size=D, num_hidden_layers=NL)
    p = R.init_params(rcfg, seed=seed)
    g = torch.Generator().manual_seed(seed)
    p.update({"llm_embedding.weight": torch.randn(2, D, generator=g) * 0.5,
              "text_embedding.weight": torch.randn(TV, D, generator=g) * 0.5,
              "speech_embedding.weight": torch.randn(SV + 1, D, generator=g) * 0.5,
              "lm_head.weight": torch.randn(SV + 1, D, generator=g) * 0.05, "lm_head.bias": torch.randn(SV + 1, generator=g) * 0.1,
              "model.embeddings.weight": torch.zeros(TV, D)})
    model = RWKV7CosyLM(cfg)
    model.load_state_dict(p, strict=True)
    return model, p, rcfg, SV


@pytest.mark.timeout(900)
def test_config0_cosy_0p1b_B2_L512_logits_loss_acc_and_gradients_vs_oracle():
    from rwkvtts_amd.losses import th_accuracy
    model, p, rcfg, SV = _cosy_0p1b()
    batch = L.synthetic_cosy_batch(2, seed=1234)        # [sos, 126 text, task, 384 speech] = 512 positions, B = 2
    assert batch["text_token"].shape == (2, 126) and batch["speech_token"].shape == (2, 384)
    # oracle: the reference's PyTorch-CPU path (per-token torch scan), with autograd for the gradient check below
    pr = {k: v.clone().requires_grad_(k != "model.embeddings.weight") for k, v in p.items()}
    R.pick_threads()
    loss_o, acc_o, logits_o = R.cosy_forward(pr, rcfg, batch, SV, 0.0, True)
    loss_o.backward()
    assert logits_o.shape == (2, 512, SV + 1)
    # (i) fp32 model: logits within 1e-3, loss, accuracy
    m32 = model.to(DEV).eval()
    with torch.no_grad():
        out = m32(batch=_to(batch, DEV))
    err = (out.logits.cpu() - logits_o.detach()).abs().max().item()
    assert err < 1e-3, f"logits differ from the oracle by {err}"
    assert abs(out.loss.item() - loss_o.item()) < 1e-4
    _, _, labels = m32.build_inputs(_to(batch, DEV))
    acc = th_accuracy(out.logits.view(-1, SV + 1), labels, ignore_label=-1)
    assert abs(acc.item() - acc_o.item()) < 1e-6
    # (ii) bf16 training step (chunked MFMA WKV7 pair, fused stages): every parameter gradient against oracle autograd.
    # Yardstick: relative L2 error per tensor; bf16 activations through 12 layers give ~1e-2, a wrong kernel gives O(1).
    m16 = m32.to(torch.bfloat16).train()
    out16 = m16(batch=_to(batch, DEV))
    out16.loss.backward()
    assert abs(out16.loss.item() - loss_o.item()) < 2e-2 * abs(loss_o.item())
    named = dict(m16.named_parameters())
    rels = {}
    for k, v in pr.items():
        if v.grad is None:
            continue
        gh = named[k].grad.float().cpu()
        if k in ("text_embedding.weight", "speech_embedding.weight"):   # sparse rows: compare the touched ones
            rows = v.grad.abs().sum(-1) > 0
            gh, ref = gh[rows], v.grad[rows]
        elif k == "llm_embedding.weight":
            # row 0 is the sos embedding = the gradient at POSITION 0, which bf16 cannot resolve: the state is zero there, so
            # y_0 = v_0 (k_0 . q_0) is a multiple of v_0 and GroupNorm removes the multiple -- d(k_0 . q_0) is analytically ~0 and
            # numerically the difference of large bf16-rounded terms.  Measured (tools/diag_r02.py): relative error 0.4-1.3 at
            # t = 0, 0.04-0.13 at t = 1, 0.03 from t = 2 on; the SCALAR bf16 kernels show the same (0.4-1.0) and fp32 is exact
            # (3e-4), so it is a property of bf16 storage, not of a kernel.  Row 1 (task id, position 127) gets the normal bar.
            assert ((gh[0] - v.grad[0]).norm() / v.grad[0].norm()).item() < 2.5
            gh, ref = gh[1], v.grad[1]
        else:
            ref = v.grad
        rels[k] = ((gh - ref).norm() / ref.norm().clamp(min=1e-12)).item()
    top = sorted(rels.items(), key=lambda kv: -kv[1])[:5]
    vals = sorted(rels.values())
    median = vals[len(vals) // 2]
    # bf16 activations and bf16 gradients through 12 layers against an fp32 reference: a few per cent on the noisiest tensors
    # (the layer-0 low-rank biases, sums of 1024 bf16-rounded rows), well under one per cent typically; a wrong kernel gives O(1)
    assert len(rels) > 12 * 30, len(rels)
    # measured: median 2.8e-2, worst 6.8e-2 (layer-0 a_lora bias); the same numbers with the scalar WKV7 kernels
    assert top[0][1] < 0.12, f"relative L2 gradient errors, worst five: {top}"
    assert median < 4e-2, f"median relative L2 gradient error {median:.3e}; worst five: {top}"
    print(f"config0: logits max|d| {err:.2e}; gradient rel. L2 error median {median:.2e}, worst {top[0][1]:.2e} ({top[0][0]}) over {len(rels)} tensors")


# ---------------------------------------------------------------------------------------------------------------------
# configs[3]
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.timeout(900)
def test_config3_chunked_wkv7_pair_B4_T8192_H32_vs_oracle_slices(c_oracle):
    B, T, H = 4, 8192, 32
    ins = make_wkv_inputs(B, T, H, 4321, torch.bfloat16)
    d = [t.to(DEV) for t in ins]
    dy = torch.randn(B, T, H, 64, generator=torch.Generator().manual_seed(7)).bfloat16()
    y, tinv, sa, hs = ops.wkv7_chunk_forward(*d)
    grads = ops.wkv7_chunk_backward(*d, dy.to(DEV), hs, sa, tinv)
    torch.cuda.synchronize()
    assert torch.isfinite(y.float()).all()
    for n, ga in zip(NAMES, grads):
        assert torch.isfinite(ga.float()).all(), n
    for (bi, hi) in ((0, 0), (2, 17), (3, 31)):      # first / middle / last head
        sl = [t[bi:bi + 1, :, hi:hi + 1].contiguous() for t in ins]
        y_o, s_o, sa_o = c_oracle.wkv7_fwd(*sl)
        g_o = c_oracle.wkv7_bwd(*sl, dy[bi:bi + 1, :, hi:hi + 1].contiguous(), s_o, sa_o)
        _assert_bf16_close(y[bi:bi + 1, :, hi:hi + 1], y_o, f"y[{bi},{hi}]")
        for n, ga, go in zip(NAMES, grads, g_o):
            _assert_bf16_close(ga[bi:bi + 1, :, hi:hi + 1], go, f"{n}[{bi},{hi}]", ulps=2.0)


def test_config3_fused_linear_ce_V66661_at_32768_rows():
    """The channel-0 head of the XY model: V0 = 66 661, 4 x 8192 rows, label smoothing off and on -- loss against
    F.cross_entropy on the same bf16 logits for a row subset, hidden-state gradient on that subset, ignored rows zero."""
    from rwkvtts_amd.losses import fused_linear_cross_entropy
    g = torch.Generator().manual_seed(6)
    N, D, V = 32768, 2048, 66661
    h = (torch.randn(N, D, generator=g) * 0.5).bfloat16().to(DEV).requires_grad_(True)
    w = (torch.randn(V, D, generator=g) * 0.02).bfloat16().to(DEV).requires_grad_(True)
    b = (torch.randn(V, generator=g) * 0.1).bfloat16().to(DEV).requires_grad_(True)
    lab = torch.randint(0, V, (N,), generator=g).to(DEV)
    lab[::7] = -100
    sub = torch.arange(0, N, 61, device=DEV)      # 538 rows spread over all chunks
    for lsm in (0.0, 0.1):
        for t in (h, w, b):
            t.grad = None
        loss = fused_linear_cross_entropy(h, lab, w, b, -100, label_smoothing=lsm)
        loss.backward()
        n_valid = (lab != -100).sum().item()
        logits = (h.detach()[sub].float() @ w.detach().float().t() + b.detach().float()).requires_grad_(True)
        per = torch.nn.functional.cross_entropy(logits, lab[sub], ignore_index=-100, label_smoothing=lsm, reduction="sum")
        per.backward()
        dh_ref = (logits.grad / n_valid) @ w.detach().float()
        got = h.grad[sub].float()
        assert (got - dh_ref).abs().max().item() <= 3e-2 * dh_ref.abs().max().item(), lsm
        assert h.grad[::7].abs().max().item() == 0
        # the full loss: mean over valid rows, estimated from the subset within its sampling error is not a test -- compare the
        # exact chunked evaluation in fp32 instead
        tot = 0.0
        with torch.no_grad():
            for s in range(0, N, 4096):
                lg = h.detach()[s:s + 4096].float() @ w.detach().float().t() + b.detach().float()
                tot += torch.nn.functional.cross_entropy(lg, lab[s:s + 4096], ignore_index=-100, label_smoothing=lsm,
                                                         reduction="sum").item()
        assert abs(loss.item() - tot / n_valid) < 2e-3 * abs(tot / n_valid), (lsm, loss.item(), tot / n_valid)
        assert torch.isfinite(w.grad.float()).all() and torch.isfinite(b.grad.float()).all()


@pytest.mark.timeout(900)
def test_config3_xy_model_8_channels_V66661_train_step_B4_L8192():
    """RWKV7XYLM at the 1.5B widths (D = 2048, H = 32, ranks 96/96/64/256), 8 channels, V0 = 66 661 + 7 x 1 025 heads, B = 4,
    L = 8192 (128 text + 8057 frames + 7 delay steps), two layers: eval loss equals the sum of the eight cross-entropies of the
    materialised logits on a position subset is covered above -- here: the fused training path runs at size, gives a finite loss
    near sum(log V_i) for random weights, gradients reach all 16 channel tensors, and two trainer steps reduce the loss."""
    from rwkvtts_amd import backbone, trainer
    from rwkvtts_amd.xy_llm import RWKV7XYConfig, RWKV7XYLM
    base = backbone.config_1p5b()
    kw = {k: v for k, v in base.to_dict().items() if k in backbone.RWKV7Config.__dataclass_fields__ and k != "extra"}
    kw.update(num_hidden_layers=2, vocab_size=66661)
    cfg = RWKV7XYConfig(speech_vocab_size=1025, num_channels=8, text_shift_size=65536, **kw)
    model = RWKV7XYLM(cfg).init_weights(seed=0)
    model.zero_embs()
    model = model.to(DEV).to(torch.bfloat16).train()
    batch = L.synthetic_xy_batch(4, T1=128, T2=8057, seed=1234)
    assert batch["input_ids"].shape == (4, 8192, 8)
    batch = _to(batch, DEV)
    import math
    expect = math.log(66661) + 7 * math.log(1025)
    tr = trainer.DataParallelTrainer(model, lr=2e-3, warmup_steps=0, total_steps=10)
    l0 = tr.step(**batch, use_cache=False).item()
    assert abs(l0 - expect) < 0.1 * expect, (l0, expect)     # random heads are not exactly uniform (measured 62.9 vs 59.6)
    for i in range(8):
        assert model.heads[i].weight.grad.float().abs().sum().item() > 0 and model.embs[i].weight.grad.float().abs().sum().item() > 0
    l1 = tr.step(**batch, use_cache=False).item()
    l2 = tr.step(**batch, use_cache=False).item()
    assert math.isfinite(l2) and l2 < l0, (l0, l1, l2)


@pytest.mark.timeout(900)
def test_config3_xy_model_full_depth_24_layers_train_steps_B4_L8192():
    """configs[3] at its full depth (xy_llm.py:189-257 on a 1.5B base, convert_rwkv7_to_xy.py:23-32): 24 layers, D = 2048,
    H = 32, 8 channels, V0 = 66 661, B = 4, L = 8192, bf16, AdamW on fp32 masters -- the step bench.py --model 1.5b --layout xy
    times (profiles/r03a_cfg3_xy_1p5b_line.txt: 494.7 ms, 142.9 GiB).  Finite loss near sum(log V_i), the loss falls over three
    steps on one batch, gradients reach the first and the last layer, and the peak stays well inside the 288 GB of one GPU."""
    import math
    from rwkvtts_amd import backbone, trainer
    from rwkvtts_amd.xy_llm import RWKV7XYConfig, RWKV7XYLM
    torch.cuda.empty_cache()
    torch.cuda.reset_peak_memory_stats()
    base = backbone.config_1p5b()
    kw = {k: v for k, v in base.to_dict().items() if k in backbone.RWKV7Config.__dataclass_fields__ and k != "extra"}
    kw.update(vocab_size=66661)
    assert kw["num_hidden_layers"] == 24 and kw["hidden_size"] == 2048
    cfg = RWKV7XYConfig(speech_vocab_size=1025, num_channels=8, text_shift_size=65536, **kw)
    model = RWKV7XYLM(cfg).init_weights(seed=0)
    model.zero_embs()
    model = model.to(DEV).to(torch.bfloat16).train()
    batch = _to(L.synthetic_xy_batch(4, T1=128, T2=8057, seed=1234), DEV)
    assert batch["input_ids"].shape == (4, 8192, 8)
    expect = math.log(66661) + 7 * math.log(1025)
    tr = trainer.DataParallelTrainer(model, lr=1e-3, warmup_steps=0, total_steps=10)
    l0 = tr.step(**batch, use_cache=False).item()
    assert math.isfinite(l0) and abs(l0 - expect) < 0.1 * expect, (l0, expect)
    for li in (0, 23):
        g = model.model.layers[li].attn.r_proj.weight.grad
        assert g is not None and torch.isfinite(g.float()).all() and g.float().abs().sum().item() > 0, li
    l1 = tr.step(**batch, use_cache=False).item()
    l2 = tr.step(**batch, use_cache=False).item()
    assert math.isfinite(l2) and l2 < l0, (l0, l1, l2)
    peak = torch.cuda.max_memory_allocated()
    assert peak < 200 * 2**30, peak / 2**30   # measured 142.9 GiB; the GPU has 288 GB
    del tr, model
    torch.cuda.empty_cache()


# ---------------------------------------------------------------------------------------------------------------------
# configs[1] -- the headline configuration, whole
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.timeout(900)
def test_config1_spark_0p4b_full_model_24_layers_train_steps_B8_L4096():
    """configs[1] as bench.py times it (spark_llm.py:105-172 on the 0.4B base): 24 layers, D = 1024, H = 16, V = 8193, the Spark
    batch [TAG2, 255 text, TAG0, 32 global, TAG1, 3806 semantic] built WITH autograd through the four embedding tables, B = 8,
    L = 4096, bf16, AdamW on fp32 masters.  The loss starts at log 8193 (random head), falls over three steps on one batch,
    gradients reach the first and the last layer AND the input-side embedding tables, the chunked MFMA WKV7 kernels are what ran
    (T % 32 == 0, bf16), and the peak stays under 80 GiB (bench: 52.9 GiB)."""
    import math
    from rwkvtts_amd import backbone, trainer
    from rwkvtts_amd.spark_llm import RWKV7ForSpeech, RWKV7SpeechConfig
    torch.cuda.empty_cache()
    torch.cuda.reset_peak_memory_stats()
    base = backbone.config_0p4b()
    kw = {k: v for k, v in base.to_dict().items() if k in backbone.RWKV7Config.__dataclass_fields__ and k != "extra"}
    assert kw["num_hidden_layers"] == 24 and kw["hidden_size"] == 1024
    model = RWKV7ForSpeech(RWKV7SpeechConfig(**kw)).init_weights(seed=0)
    assert model.config.vocab_size == 8193 and model.config.num_heads == 16
    model = model.to(DEV).to(torch.bfloat16).train()
    tr = trainer.DataParallelTrainer(model, lr=1e-3, warmup_steps=0, total_steps=10)
    mk = lambda: L.synthetic_spark_batch(model, 8, 4096, seed=1234)
    b0 = mk()
    assert b0["inputs_embeds"].shape == (8, 4096, 1024) and int((b0["labels"] != -100).sum()) == 8 * 3806
    l0 = tr.step(**b0).item()
    assert math.isfinite(l0) and abs(l0 - math.log(8193)) < 0.05 * math.log(8193), (l0, math.log(8193))
    for li in (0, 23):
        g = model.model.layers[li].attn.r_proj.weight.grad
        assert g is not None and torch.isfinite(g.float()).all() and g.float().abs().sum().item() > 0, li
    for tab in (model.text_embedder, model.global_embedder, model.tts_tag_embedder, model.model.embeddings):
        assert tab.weight.grad is not None and tab.weight.grad.float().abs().sum().item() > 0
    l1 = tr.step(**mk()).item()
    l2 = tr.step(**mk()).item()
    assert math.isfinite(l2) and l2 < l1 < l0, (l0, l1, l2)
    peak = torch.cuda.max_memory_allocated()
    assert peak < 80 * 2**30, peak / 2**30
    del tr, model
    torch.cuda.empty_cache()


# ---------------------------------------------------------------------------------------------------------------------
# configs[4]
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.timeout(1200)
def test_config4_greedy_decode_24_layers_B32_P128_2048_tokens_vs_fp32_twin():
    """Long-horizon check of the persistent-state decode path (rwkv_asr_cuda_whisper.py:694-717): the bf16 step kernel replayed
    2048 times from a hipGraph against the fp32 model run module by module on the SAME ids (teacher forcing removes the
    divergence of histories after a near-tie).  North-star: "bit-exact argmax ids for greedy decode" -- exact for fp32
    (test_model_gpu.py); for bf16 the statement that can hold is: the ids equal the fp32 argmax wherever the fp32 top-2 margin
    exceeds the bf16 logit noise.  State drift: the recurrent state after 2048 in-place updates stays within 2 % (relative L2,
    per layer) of the fp32 state."""
    from rwkvtts_amd import backbone
    from rwkvtts_amd.backbone import Cache
    from rwkvtts_amd.decode import GraphDecoder
    from rwkvtts_amd.spark_llm import RWKV7ForSpeech, RWKV7SpeechConfig
    base = backbone.config_0p4b()
    kw = {k: v for k, v in base.to_dict().items() if k in backbone.RWKV7Config.__dataclass_fields__ and k != "extra"}
    cfg = RWKV7SpeechConfig(**kw)
    m32 = RWKV7ForSpeech(cfg).init_weights(seed=0)
    with torch.no_grad():   # random heads give near-uniform logits; sharpen so that margins are informative
        m32.lm_head.weight.mul_(4.0)
        for p_ in m32.parameters():
            p_.copy_(p_.to(torch.bfloat16).float())          # bf16-valued weights on both sides
    m32 = m32.to(DEV).eval()
    import copy
    m16 = copy.deepcopy(m32).to(torch.bfloat16).eval()
    B, P, NEW = 32, 128, 2048
    g = torch.Generator().manual_seed(1234)
    emb = (torch.randn(B, P, cfg.hidden_size, generator=g) * 0.5).to(DEV)
    mask = torch.ones(B, P, dtype=torch.long, device=DEV)
    eos = cfg.vocab_size - 1
    dec = GraphDecoder(m16, B)
    got = dec.generate(inputs_embeds=emb.to(torch.bfloat16), attention_mask=mask, max_new_tokens=NEW, suppress_tokens=[eos])
    assert dec.step is not None and not dec.step.barrier_timed_out(), "the step kernel must cover configs[4]"
    assert got.shape == (B, NEW)
    # fp32 twin, teacher-forced
    c32 = Cache.zeros(cfg, B, DEV, torch.float32)
    with torch.no_grad():
        lg = m32(inputs_embeds=emb, attention_mask=mask, past_key_values=c32, use_cache=True, logits_to_keep=1).logits[:, -1].float()
    sure_n = agree_sure = agree_all = 0
    for t in range(NEW):
        lg[:, eos] = float("-inf")
        top2 = torch.topk(lg, 2, -1)
        rng = lg[:, :eos].abs().amax(-1)
        sure = (top2.values[:, 0] - top2.values[:, 1]) > 3e-2 * rng
        agree_sure += int((got[sure, t] == top2.indices[sure, 0]).sum())
        sure_n += int(sure.sum())
        agree_all += int((got[:, t] == top2.indices[:, 0]).sum())
        if t + 1 < NEW:   # the last generated id is not fed back by generate(): both states have seen P + NEW - 1 tokens
            with torch.no_grad():
                lg = m32(input_ids=got[:, t:t + 1], past_key_values=c32, use_cache=True).logits[:, -1].float()
    assert sure_n > 0.2 * B * NEW, f"only {sure_n} decisive positions: the check would be vacuous"
    assert agree_sure == sure_n, f"{sure_n - agree_sure} of {sure_n} decisive argmax ids differ from the fp32 twin"
    worst = 0.0
    for s16, s32 in zip(dec.cache.states, c32.states):
        rel = ((s16.att_kv - s32.att_kv).norm() / s32.att_kv.norm()).item()
        worst = max(worst, rel)
    # measured (tools/diag_r02.py): 0.4-1.5 % after the first step (bf16 activations, growing with depth), 0.4 % after 2048
    assert worst < 2e-2, f"recurrent state drifted by {worst:.3e} (relative L2) after {NEW} steps"
    print(f"config4: {agree_all}/{B * NEW} ids equal the fp32 argmax, {sure_n} decisive all equal, state drift {worst:.2e}")
