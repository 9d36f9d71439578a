"""GPU (-m gpu): Cosy and XY heads on the HIP backbone against the oracle restatement (oracle/rwkv7_ref.py
cosy_forward / xy_forward).  fp32 models: logits within 1e-3, losses within 1e-4, greedy ids equal."""
import pytest
import torch

from oracle import rwkv7_ref as R
from rwkvtts_amd import layouts as L
from rwkvtts_amd.backbone import Cache
from rwkvtts_amd.cosy_llm import RWKV7CosyConfig, RWKV7CosyLM, RWKV7LM
from rwkvtts_amd.xy_llm import RWKV7XYConfig, RWKV7XYLM

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
SMALL = dict(hidden_size=128, num_hidden_layers=2, decay_low_rank_dim=32, a_low_rank_dim=32, v_low_rank_dim=16,
             gate_low_rank_dim=32)


def _to(batch, dev):
    return {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in batch.items()}


def test_cosy_forward_loss_acc_and_batch_slicing():
    cfg = RWKV7CosyConfig(vocab_size=200, speech_token_size=50, lsm_weight=0.1, **SMALL)
    rcfg = R.RefConfig(vocab_size=0, **SMALL)
    p = R.init_params(rcfg, seed=4)
    g = torch.Generator().manual_seed(4)
    p.update({"llm_embedding.weight": torch.randn(2, 128, generator=g) * 0.5,
              "text_embedding.weight": torch.randn(200, 128, generator=g) * 0.5,
              "speech_embedding.weight": torch.randn(51, 128, generator=g) * 0.5,
              "lm_head.weight": torch.randn(51, 128, generator=g) * 0.05, "lm_head.bias": torch.randn(51, generator=g) * 0.1,
              "model.embeddings.weight": torch.zeros(200, 128)})
    model = RWKV7CosyLM(cfg)
    model.load_state_dict(p, strict=True)
    model = model.to(DEV).eval()
    batch = L.cosy_collate([[3, 4, 5, 6], [7, 8]], [[10, 11, 12, 13, 14, 15, 16], [20, 21, 22]], pad_to_max_length=False)
    with torch.no_grad():
        out = model(batch=_to(batch, DEV))
        loss_o, acc_o, logits_o = R.cosy_forward(p, rcfg, batch, 50, 0.1, True)
    emb, mask, labels = model.build_inputs(_to(batch, DEV))
    valid = mask.bool().cpu()
    assert (out.logits.cpu() - logits_o)[valid].abs().max().item() < 1e-3
    assert abs(out.loss.item() - loss_o.item()) < 1e-4
    assert (emb[1, -1] == -1).all()  # padding VALUE is -1 (cosy_llm.py:71), masked out
    # RWKV7LM wrapper: same numbers through forward(batch) -> {'loss','acc'}
    wrapper = RWKV7LM(128, 128, 50, model, lsm_weight=0.1).to(DEV).eval()
    wrapper.llm_embedding.weight.data.copy_(p["llm_embedding.weight"])
    wrapper.speech_embedding.weight.data.copy_(p["speech_embedding.weight"])
    wrapper.text_embedding = model.text_embedding
    with torch.no_grad():
        r = wrapper(_to(batch, DEV))
    assert abs(r["loss"].item() - loss_o.item()) < 1e-4 and abs(r["acc"].item() - acc_o.item()) < 1e-6
    # max_tokens_k slicing (cosy_llm.py:122-130): 1 "k-token" = 1024 positions; seq_len 13 -> whole batch kept
    with torch.no_grad():
        assert model(batch=_to(batch, DEV), max_tokens_k=1).logits.shape[0] == 2


def test_cosy_streaming_inference_matches_greedy_oracle():
    cfg = RWKV7CosyConfig(vocab_size=200, speech_token_size=50, **SMALL)
    model = RWKV7CosyLM(cfg).init_weights(seed=1).to(DEV).eval()
    model.sampling = lambda scores, decoded, sampling: scores.argmax().reshape(1)  # greedy
    text = torch.tensor([[5, 6, 7, 8]], device=DEV)
    gen = model.inference(text, torch.tensor([4], device=DEV), torch.zeros(1, 0, dtype=torch.long, device=DEV),
                          torch.tensor([0], device=DEV), torch.zeros(1, 0, dtype=torch.long, device=DEV),
                          torch.tensor([0], device=DEV), max_token_text_ratio=5, min_token_text_ratio=0)
    ids = list(gen)
    # oracle: same weights, full re-forward of the growing sequence each step (no state), greedy
    p = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    rcfg = R.RefConfig(vocab_size=0, **SMALL)
    seq = torch.cat([p["llm_embedding.weight"][0:1], p["text_embedding.weight"][text[0].cpu()],
                     p["llm_embedding.weight"][1:2]])[None]
    want = []
    for _ in range(20):
        h, _ = R.backbone(p, rcfg, seq, None, None)
        nxt = int((h[0, -1] @ p["lm_head.weight"].t() + p["lm_head.bias"]).argmax())
        if nxt >= 50:
            break
        want.append(nxt)
        seq = torch.cat([seq, p["speech_embedding.weight"][nxt][None, None]], 1)
    assert ids == want[:len(ids)] and len(ids) >= min(len(want), 1)


def _xy_pair(lsm=0.0):
    cfg = RWKV7XYConfig(vocab_size=120, speech_vocab_size=16, num_channels=4, text_shift_size=100, lsm_weight=lsm, **SMALL)
    model = RWKV7XYLM(cfg).init_weights(seed=2)
    with torch.no_grad():
        for h in model.heads:
            h.bias.normal_(0, 0.1)
    model.zero_embs()
    p = {k: v.detach().clone() for k, v in model.state_dict().items()}
    return model.to(DEV), p, R.RefConfig(vocab_size=0, **SMALL)


@pytest.mark.parametrize("lsm", [0.0, 0.1])
def test_xy_forward_and_fused_training_loss(lsm):
    model, p, rcfg = _xy_pair(lsm)
    g = torch.Generator().manual_seed(0)
    text = [[101 + i for i in range(5)], [90, 91, 92]]
    audio = [torch.randint(0, 15, (4, n), generator=g).tolist() for n in (9, 5)]
    batch = L.XYDataProcessor(120, 4, 100, 16).process_batch(text, audio)
    assert (model.embs[0].weight[119] == 0).all() and (model.embs[1].weight[15] == 0).all()  # zero_embs
    model.eval()
    with torch.no_grad():
        out = model(**_to(batch, DEV))
        loss_o, logits_o = R.xy_forward(p, rcfg, batch["input_ids"], batch["attention_mask"], batch["labels"], 4, lsm)
    valid = batch["attention_mask"].bool()
    for a, b in zip(out.logits, logits_o):
        assert (a.cpu() - b)[valid].abs().max().item() < 1e-3
    assert abs(out.loss.item() - loss_o.item()) < 1e-4
    # training mode: chunked fused linear+CE (N4), logits not materialised, same loss, gradients flow to all heads
    model.train()
    out2 = model(**_to(batch, DEV))
    assert out2.logits is None and abs(out2.loss.item() - loss_o.item()) < 1e-4
    out2.loss.backward()
    assert all(h.weight.grad is not None and torch.isfinite(h.weight.grad).all() for h in model.heads)
    with pytest.raises(ValueError):
        model(input_ids=batch["input_ids"][:, :, :3].to(DEV))


def test_xy_generate_channel0_mask_and_shapes():
    model, p, rcfg = _xy_pair()
    model.eval()
    prompt = L.XYDataProcessor(120, 4, 100, 16).process_batch([[101, 102, 103]], [[[1, 2], [3, 4], [5, 6], [7, 8]]])
    ids = prompt["input_ids"][:, :5].to(DEV)
    out = model.generate(ids, max_new_tokens=6, do_sample=False)
    assert out.shape == (1, 11, 4)
    new = out[0, 5:]
    assert ((new[:, 0] >= 100) & (new[:, 0] < 116)).all()  # channel 0 constrained to [text_shift, +speech_vocab)
    # greedy parity with the oracle on the first generated step
    _, logits_o = R.xy_forward(p, rcfg, ids.cpu(), None, None, 4)
    l0 = logits_o[0][0, -1].clone()
    l0[:100] = float("-inf")
    l0[116:] = float("-inf")
    assert int(new[0, 0]) == int(l0.argmax()) and [int(new[0, i]) for i in (1, 2, 3)] == \
        [int(logits_o[i][0, -1].argmax()) for i in (1, 2, 3)]


def test_xy_generate_step_kernel_vs_module_path():
    """bf16 XY model: the T = 1 steps of generate() through rwkv7_decode_step_bf16 (eight heads as one concatenated projection)
    against the module-by-module steps: same per-channel logits step by step (teacher-forced, bf16 tolerance), and generate()
    itself keeps its contract (shape, channel-0 constraint, first frame -- which comes from the shared prefill -- identical)."""
    from types import SimpleNamespace
    from rwkvtts_amd.backbone import Cache
    from rwkvtts_amd.decode import DecodeStep
    cfg = RWKV7XYConfig(vocab_size=120, speech_vocab_size=16, num_channels=4, text_shift_size=100, hidden_size=128,
                        num_hidden_layers=2, decay_low_rank_dim=32, a_low_rank_dim=32, v_low_rank_dim=32, gate_low_rank_dim=32)
    model = RWKV7XYLM(cfg).init_weights(seed=5)
    with torch.no_grad():
        for h in model.heads:
            h.bias.normal_(0, 0.1)
    model.zero_embs()
    model = model.to(DEV).to(torch.bfloat16).eval()
    g = torch.Generator().manual_seed(1)
    B = 5
    text = [[101 + (i + b) % 10 for i in range(4)] for b in range(B)]
    audio = [torch.randint(0, 15, (4, 3), generator=g).tolist() for _ in range(B)]
    prompt = L.XYDataProcessor(120, 4, 100, 16).process_batch(text, audio)
    ids = prompt["input_ids"][:, :6].to(DEV)
    # step by step, both paths fed the same frames
    caches = [Cache.zeros(cfg, B, DEV, torch.bfloat16) for _ in range(2)]
    with torch.no_grad():
        for c in caches:
            model(input_ids=ids, past_key_values=c, use_cache=True)
        head = SimpleNamespace(weight=torch.cat([h.weight for h in model.heads], 0).contiguous(),
                               bias=torch.cat([h.bias for h in model.heads], 0).contiguous())
        step = DecodeStep(model.model, head, caches[0])
        sizes = [h.weight.shape[0] for h in model.heads]
        for t in range(6):
            frame = torch.stack([torch.randint(100, 116, (B,), generator=g)] + [torch.randint(0, 15, (B,), generator=g) for _ in range(3)], -1).to(DEV)
            lk = torch.split(step(model.embed(frame[:, None, :])[:, 0].contiguous()), sizes, dim=1)
            lm = model(input_ids=frame[:, None, :], past_key_values=caches[1], use_cache=True).logits
            for ch in range(4):
                ref = lm[ch][:, -1].float()
                assert (lk[ch] - ref).abs().max().item() < 3e-2 * ref.abs().max().item(), (t, ch)
    model.use_step_kernel = True
    a = model.generate(ids, max_new_tokens=14, do_sample=False)
    model.use_step_kernel = False
    b = model.generate(ids, max_new_tokens=14, do_sample=False)
    assert a.shape == b.shape == (B, 20, 4)
    assert torch.equal(a[:, :7], b[:, :7])
    assert ((a[:, 6:, 0] >= 100) & (a[:, 6:, 0] < 116)).all()


def test_cosy_inference_step_kernel_vs_fp32_twin():
    """bf16 Cosy model, B = 1 streaming inference (greedy) with the T = 1 steps on rwkv7_decode_step_bf16: every emitted id must
    be the fp32 twin's argmax (same bf16-valued weights, fp32 arithmetic, teacher-forced along the emitted ids) wherever the
    twin's top-2 margin exceeds the bf16 noise; the module-by-module path must emit the same first id (shared prefill)."""
    import copy
    cfg = RWKV7CosyConfig(vocab_size=200, speech_token_size=50, hidden_size=128, num_hidden_layers=2, decay_low_rank_dim=32,
                          a_low_rank_dim=32, v_low_rank_dim=32, gate_low_rank_dim=32)
    model = RWKV7CosyLM(cfg).init_weights(seed=4).to(DEV).to(torch.bfloat16).eval()
    twin = copy.deepcopy(model).float().eval()
    greedy = lambda scores, decoded, sampling: scores.argmax().reshape(1)
    model.sampling = greedy
    text = torch.tensor([[5, 6, 7, 8, 9, 10]], device=DEV)

    def run(flag):
        model.use_step_kernel = flag
        z = torch.zeros(1, 0, dtype=torch.long, device=DEV)
        return list(model.inference(text, torch.tensor([6], device=DEV), z, torch.tensor([0], device=DEV), z,
                                    torch.tensor([0], device=DEV), max_token_text_ratio=4, min_token_text_ratio=1))

    a, b = run(True), run(False)
    assert len(a) >= 6 and a[0] == b[0]
    with torch.no_grad():
        x = torch.cat([twin.llm_embedding.weight[twin.sos_eos].reshape(1, 1, -1), twin.text_embedding(text),
                       twin.llm_embedding.weight[twin.task_id].reshape(1, 1, -1)], 1)
        cache = Cache.zeros(cfg, 1, DEV, torch.float32)
        checked = 0
        for t, tok in enumerate(a):
            masks = torch.ones((1, x.shape[1], x.shape[1]), device=DEV, dtype=torch.bool)
            lg, cache = twin.forward_one_step(x, masks=masks, cache=cache)
            lg = lg[0, -1, :50]   # before min_len EOS is ignored (resampled away): the ids compete among the speech tokens
            top2 = torch.topk(lg, 2)
            if (top2.values[0] - top2.values[1]).item() > 2e-2 * lg.abs().max().item():
                assert tok == int(top2.indices[0]), (t, tok, int(top2.indices[0]))
                checked += 1
            x = twin.speech_embedding.weight[tok].reshape(1, 1, -1)
    assert checked >= len(a) // 2, (checked, len(a))


@pytest.mark.parametrize("fused_frame", [True, False])   # the frame's bookkeeping as one HIP launch / as tensor operations
def test_xy_generate_flush_stagger_eos_and_stop_at_full_channel_count(monkeypatch, fused_frame):
    """CustomGenerationMixin._sample (xy_llm.py:88-146) at the real channel count and vocabularies (8 channels, V0 = 66 661,
    1 025 per speech channel): once channel 0 leaves the audio range a C-1 = 7 step countdown starts; during its 8 rows channel 0
    carries EOS, and channel i keeps its sampled value for i more rows (the delay pattern: RVQ-i lags i steps) before it is
    padded; the sequence finishes when the countdown ends; finished sequences emit EOS / pad while the rest of the batch
    continues; an EOS on channel 0 stops a sequence at once.  The sampler is scripted (channel 0 is masked to the audio range, so
    only a scripted draw can leave it -- exactly the situation the reference's flush logic is written for) and the expectation
    is hand-stepped from the reference's rules, not from our code.  The reference's own termination test
    (`& ~(needs_additional_steps == -1)`, true from the first step on) would stop every sequence after ONE token; that bug is
    not reproduced."""
    from rwkvtts_amd import spark_llm
    C, V0, SV, SHIFT = 8, 66661, 1025, 65536
    cfg = RWKV7XYConfig(vocab_size=V0, speech_vocab_size=SV, num_channels=C, text_shift_size=SHIFT, **SMALL)
    model = RWKV7XYLM(cfg).init_weights(seed=2)
    model.zero_embs()
    model = model.to(DEV).eval()
    model.fused_frame = fused_frame
    PAD, EOS = cfg.speech_pad_token, 65535          # EOS: a text id, outside the audio range
    assert PAD == SV - 1
    B, T0 = 3, 4
    trigger = {0: 2, 1: 5}                           # sequence -> step at which channel 0 draws a non-audio id; sequence 2: never
    calls = {"n": 0}

    def scripted(logits, *a, **k):                   # called once per channel and step, channels in order
        step, ch = divmod(calls["n"], C)
        calls["n"] += 1
        if ch == 0:
            # what the mask must have done before the draw: only audio ids are finite
            assert torch.isinf(logits[:, :SHIFT]).all() and torch.isinf(logits[:, SHIFT + SV:]).all()
            assert torch.isfinite(logits[:, SHIFT:SHIFT + SV]).all()
            out = torch.full((logits.shape[0],), SHIFT, dtype=torch.long, device=logits.device) + 10 + step
            for s_, st_ in trigger.items():
                if step == st_:
                    out[s_] = 7                      # a text id
            return out
        return torch.full((logits.shape[0],), 100 * ch + step, dtype=torch.long, device=logits.device)

    monkeypatch.setattr(spark_llm, "sample_next", scripted)
    prompt = torch.full((B, T0, C), PAD, dtype=torch.long, device=DEV)
    prompt[:, :, 0] = torch.arange(T0, device=DEV) + 5
    out = model.generate(prompt, max_new_tokens=16, eos_token_id=EOS)
    new = out[:, T0:].cpu()
    n_steps = new.shape[1]
    # hand-stepped expectation
    want = torch.zeros(B, n_steps, C, dtype=torch.long)
    for b in range(B):
        done_after = trigger[b] + C - 1 if b in trigger else None      # last row of the flush
        for step in range(n_steps):
            for ch in range(C):
                sampled = (SHIFT + 10 + step) if ch == 0 else 100 * ch + step
                if done_after is not None and step > done_after:         # finished: EOS / pad
                    want[b, step, ch] = EOS if ch == 0 else PAD
                elif b in trigger and step >= trigger[b]:                # flushing rows j = step - trigger = 0..7
                    j = step - trigger[b]
                    want[b, step, ch] = EOS if ch == 0 else (sampled if j < ch else PAD)
                else:
                    want[b, step, ch] = sampled
    assert n_steps == 16                                                 # sequence 2 never flushes: max_new_tokens ends it
    assert torch.equal(new, want), (new - want).abs().sum(-1)
    # without an EOS id (no EOS criterion in the reference either): channel 0 keeps what was drawn, finished rows carry 0 / pad
    calls["n"] = 0
    out0 = model.generate(prompt, max_new_tokens=16)[:, T0:].cpu()
    for b in range(B):
        for step in range(16):
            drawn0 = 7 if trigger.get(b) == step else SHIFT + 10 + step
            fin = b in trigger and step > trigger[b] + C - 1
            assert int(out0[b, step, 0]) == (0 if fin else drawn0), (b, step)
    assert torch.equal(out0[:, :, 1:], want[:, :, 1:])
    # EOS inside the audio range (stopping criteria on channel 0, xy_llm.py:139): the sequence stops with that row
    calls["n"] = 0
    trigger.clear()
    out2 = model.generate(prompt[:1], max_new_tokens=16, eos_token_id=SHIFT + 10 + 3)
    assert out2.shape[1] == T0 + 4 and int(out2[0, -1, 0]) == SHIFT + 13
    # reference_termination=True: xy_llm.py:139-140 literally.  `& ~(needs_additional_steps == -1)` is false for every sequence that
    # is not flushing, so a batch in which nobody flushes at the first frame ends after ONE frame; a sequence that starts its flush at
    # frame 0 carries the loop on alone, the others emit EOS / pad rows, and the EOS the flush writes on channel 0 meets the EOS
    # criterion in that very row (the two defects the default does not reproduce)
    calls["n"] = 0
    trigger.update({0: 2, 1: 5})
    ref1 = model.generate(prompt, max_new_tokens=16, eos_token_id=EOS, reference_termination=True)
    assert ref1.shape[1] == T0 + 1 and torch.equal(ref1[:, T0].cpu(), want[:, 0])
    calls["n"] = 0
    trigger.clear()
    trigger[1] = 0                                   # sequence 1 draws a text id at the first frame
    ref2 = model.generate(prompt, max_new_tokens=16, reference_termination=True)[:, T0:].cpu()   # no EOS id: the countdown runs
    assert ref2.shape[1] == C                        # frames 0 .. C-1 of sequence 1's flush, then needs == -1 ends it
    assert [int(x) for x in ref2[1, :, 0]] == [7] + [SHIFT + 10 + s for s in range(1, C)]        # channel 0 keeps the draws (no EOS id)
    for j in range(C):
        for ch in range(1, C):
            assert int(ref2[1, j, ch]) == (100 * ch + j if j < ch else PAD), (j, ch)
    assert (ref2[0, 1:, 0] == 0).all() and (ref2[0, 1:, 1:] == PAD).all() and (ref2[2, 1:, 1:] == PAD).all()   # 0 and 2 stopped after frame 0
    calls["n"] = 0
    ref3 = model.generate(prompt, max_new_tokens=16, eos_token_id=EOS, reference_termination=True)
    assert ref3.shape[1] == T0 + 1                   # with an EOS id the flush's own EOS stops sequence 1 in its first row


def test_xy_generate_captured_frame_step_equals_eager_loop():
    """RWKV7XYLM.generate replays one frame -- embedding sum, the stack + eight heads (rwkv7_decode_step_bf16), channel-0 mask,
    the draws, flush / pad / stop bookkeeping -- from a hipGraph (no per-frame host read-back; the `all finished` flag is read
    every 8 frames and surplus rows are trimmed).  Greedy: id for id the eager loop's output, for a length-bounded run and for a
    run in which every sequence finishes by EOS before the bound (the graph over-runs the break by up to 7 frames, which must not
    show)."""
    cfg = RWKV7XYConfig(vocab_size=120, speech_vocab_size=16, num_channels=4, text_shift_size=100, hidden_size=128,
                        num_hidden_layers=2, decay_low_rank_dim=32, a_low_rank_dim=32, v_low_rank_dim=32, gate_low_rank_dim=32)
    model = RWKV7XYLM(cfg).init_weights(seed=5)
    with torch.no_grad():
        for h in model.heads:
            h.bias.normal_(0, 0.1)
    model.zero_embs()
    model = model.to(DEV).to(torch.bfloat16).eval()
    g = torch.Generator().manual_seed(1)
    B = 5
    text = [[101 + (i + b) % 10 for i in range(4)] for b in range(B)]
    audio = [torch.randint(0, 15, (4, 3), generator=g).tolist() for _ in range(B)]
    ids = L.XYDataProcessor(120, 4, 100, 16).process_batch(text, audio)["input_ids"][:, :6].to(DEV)
    eager = model.generate(ids, max_new_tokens=21, do_sample=False, use_graph=False)
    graph = model.generate(ids, max_new_tokens=21, do_sample=False, use_graph=True)
    assert eager.shape == graph.shape == (B, 27, 4) and torch.equal(eager, graph)
    # an EOS that greedy decoding reaches: take channel 0's id at frame 3 of sequence 0 as the EOS id -> every sequence that emits it stops
    eos = int(eager[0, 6 + 3, 0])
    e2 = model.generate(ids, max_new_tokens=21, do_sample=False, eos_token_id=eos, use_graph=False)
    g2 = model.generate(ids, max_new_tokens=21, do_sample=False, eos_token_id=eos, use_graph=True)
    assert e2.shape == g2.shape and torch.equal(e2, g2)
    # sampled decode replays from the graph as well (device generator): shape and channel-0 constraint
    s = model.generate(ids, max_new_tokens=9, do_sample=True, top_k=5, use_graph=True)
    assert s.shape == (B, 15, 4) and ((s[:, 6:, 0] >= 100) & (s[:, 6:, 0] < 116)).all()
    # the fused frame (embedding sum, draws and bookkeeping as three launches) against the tensor-operation forms of the same
    # steps on the same draws -- the fused sampler is keyed by (seed, frame), so both runs see identical ids: greedy, sampled,
    # with an EOS that is reached, eager and captured
    for kw in (dict(do_sample=False), dict(do_sample=True, top_k=5, top_p=0.9), dict(do_sample=True, top_k=5, eos_token_id=eos)):
        for graph in (False, True):
            outs = []
            for fused in (True, False):
                model.fused_frame = fused
                torch.manual_seed(11)
                outs.append(model.generate(ids, max_new_tokens=21, use_graph=graph, **kw))
            assert outs[0].shape == outs[1].shape and torch.equal(outs[0], outs[1]), (kw, graph)
    model.fused_frame = True


def test_cosy_streaming_inference_replays_the_step_from_a_graph_with_the_stock_sampler():
    """bf16 Cosy model, stock repetition-aware sampler: after the prompt and the first id, every further token is ONE graph replay
    (embedding -> rwkv7_decode_step_bf16 -> log-softmax -> ras_sampling_device -> ring / counter update) and one 8-byte read-back
    for the generator's yield (the host path: three read-backs per token).  Contract: ids are speech tokens, EOS never before the
    minimum length, the run ends by EOS or at the maximum length, a seed fixes the sequence, and the greedy prefix logic is shared
    with the host path (same first id under the same seed: it is drawn before the capture)."""
    cfg = RWKV7CosyConfig(vocab_size=200, speech_token_size=50, hidden_size=128, num_hidden_layers=2, decay_low_rank_dim=32,
                          a_low_rank_dim=32, v_low_rank_dim=32, gate_low_rank_dim=32)
    model = RWKV7CosyLM(cfg).init_weights(seed=4).to(DEV).to(torch.bfloat16).eval()
    text = torch.tensor([[5, 6, 7, 8, 9, 10]], device=DEV)
    z = torch.zeros(1, 0, dtype=torch.long, device=DEV)

    def run(seed, graph=True):
        model.use_graph = graph
        torch.manual_seed(seed)
        ids = list(model.inference(text, torch.tensor([6], device=DEV), z, torch.tensor([0], device=DEV), z, torch.tensor([0], device=DEV),
                                   max_token_text_ratio=8, min_token_text_ratio=3))
        return ids, model.last_inference_used_graph

    a, used = run(7)
    assert used and 18 <= len(a) <= 48 and all(0 <= t < 50 for t in a)     # min_len = 18: no EOS before it; max_len = 48
    b, _ = run(7)
    c, _ = run(8)
    assert a == b and a != c
    h, used_h = run(7, graph=False)
    assert not used_h and h[0] == a[0] and all(0 <= t < 50 for t in h)
