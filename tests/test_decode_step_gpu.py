"""GPU (-m gpu): the one-kernel decode step (csrc/decode_step.hip, rwkv7_decode_step_bf16) against the module-by-module
stateful path (reference forward_batch at T = 1, rwkv_asr_cuda_whisper.py:438-472) and against the same model run in fp32.

The step kernel keeps fp32 between the projections where the module path rounds every tensor to bf16, so the yardstick is
the fp32 twin of the model (same bf16-valued weights, fp32 arithmetic on the *_f32 kernels): the step kernel must be at least
as close to it as the bf16 module path is, logits and recurrent state alike."""
import copy

import pytest
import torch

from rwkvtts_amd import backbone
from rwkvtts_amd.backbone import Cache, RWKV7Config, RWKV7ForCausalLM
from rwkvtts_amd.decode import DecodeStep, GraphDecoder

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _model(D, L, V, ranks, seed=0, head_bias=False):
    cfg = RWKV7Config(hidden_size=D, num_hidden_layers=L, vocab_size=V, decay_low_rank_dim=ranks[0], a_low_rank_dim=ranks[1],
                      v_low_rank_dim=ranks[2], gate_low_rank_dim=ranks[3])
    torch.manual_seed(seed)
    m = RWKV7ForCausalLM(cfg, head_bias=head_bias)
    backbone.init_weights(m, cfg, seed=seed)
    with torch.no_grad():
        m.lm_head.weight.normal_(0, 0.05)
        if head_bias:
            m.lm_head.bias.normal_(0, 0.1)
        for blk in m.model.layers:   # non-trivial norms and biases
            for ln in (blk.attn_norm, blk.ffn_norm):
                ln.weight.add_(torch.randn_like(ln.weight) * 0.1)
                ln.bias.add_(torch.randn_like(ln.bias) * 0.1)
    m16 = m.to(DEV).to(torch.bfloat16).eval()
    m32 = copy.deepcopy(m16).float().eval()   # same (bf16-valued) weights, fp32 arithmetic
    return cfg, m16, m32


def _prefill(m, ids, B, dtype):
    cache = Cache.zeros(m.config, B, DEV, dtype)
    with torch.no_grad():
        out = m(input_ids=ids, past_key_values=cache, use_cache=True)
    return cache, out.logits[:, -1].float()


def _clone_cache(c):
    return Cache([backbone.LayerState(s.att_x_prev.clone(), s.att_kv.clone(), s.ffn_x_prev.clone()) for s in c.states], c.seen_tokens)


@pytest.mark.parametrize("D,L,V,ranks,B,bias", [
    (128, 3, 77, (32, 32, 32, 32), 5, True),        # odd batch, partial last head tile, head bias
    (128, 2, 64, (64, 32, 32, 96), 32, False),
    (1024, 2, 8193, (64, 64, 32, 128), 32, False),  # 0.4B widths (BASELINE configs[4]), two layers
    (768, 2, 300, (64, 64, 32, 128), 7, False),     # 0.1B width: K splits that are not powers of two away from 1024
    (2048, 2, 1025, (96, 96, 64, 256), 4, True),    # 1.5B widths (head phase: the 8 + 16 fragment-slot instantiation)
    (256, 2, 200, (160, 32, 32, 64), 6, False),     # a rank above 128: the generic 16 + 16 slot instantiation in the per-phase mode too
])
def test_step_kernel_vs_module_path_and_fp32(D, L, V, ranks, B, bias):
    cfg, m16, m32 = _model(D, L, V, ranks, seed=D + B, head_bias=bias)
    g = torch.Generator().manual_seed(B)
    prompt = torch.randint(0, V, (B, 6), generator=g).to(DEV)
    c16, lg = _prefill(m16, prompt, B, torch.bfloat16)
    c32, lg32 = _prefill(m32, prompt, B, torch.float32)
    cp = _clone_cache(c16)   # step kernel, one launch per phase
    ct = _clone_cache(c16)   # one launch per phase through the device-table entry (rwkv7_decode_step_bf16)
    step_p = DecodeStep(m16.model, m16.lm_head, cp, persistent=False)
    step_t = DecodeStep(m16.model, m16.lm_head, ct, persistent=False, host_table=False)
    emb = m16.model.embeddings.weight
    worst_mod = worst_ker = 0.0
    for it in range(5):
        ids = torch.argmax(lg32, -1)   # every path is fed the fp32 twin's greedy ids
        with torch.no_grad():
            lg32 = m32(input_ids=ids[:, None], past_key_values=c32, use_cache=True).logits[:, -1].float()
            lmod = m16(input_ids=ids[:, None], past_key_values=c16, use_cache=True).logits[:, -1].float()
            x = torch.nn.functional.embedding(ids, emb)
            lp = step_p(x).clone()
            assert torch.equal(step_t(x), lp)   # pointers as kernel arguments or fetched from the table: the same kernels otherwise
        assert torch.isfinite(lp).all()
        scale = lg32.abs().max().item()
        worst_mod = max(worst_mod, (lmod - lg32).abs().max().item() / scale)
        worst_ker = max(worst_ker, (lp - lg32).abs().max().item() / scale)
    assert worst_ker < 3e-2, (worst_ker, worst_mod)
    assert worst_ker < 1.5 * worst_mod + 2e-3, (worst_ker, worst_mod)
    for st_, sp in zip(ct.states, cp.states):
        assert torch.equal(st_.att_kv, sp.att_kv) and torch.equal(st_.att_x_prev, sp.att_x_prev) and torch.equal(st_.ffn_x_prev, sp.ffn_x_prev)
    for sk, s32 in zip(cp.states, c32.states):
        ref = s32.att_kv
        assert (sk.att_kv - ref).abs().max().item() < 3e-2 * ref.abs().max().item() + 1e-3
        assert (sk.att_x_prev.float() - s32.att_x_prev).abs().max().item() < 5e-2 * s32.att_x_prev.abs().max().item() + 1e-2


def test_graph_decoder_with_step_kernel_matches_fp32_greedy_ids():
    """Greedy ids of the hipGraph loop around the step kernel == ids of the fp32 twin wherever the fp32 top-2 margin is
    larger than the bf16 noise (north_star: bit-exact argmax ids; ties inside the rounding noise are not decidable)."""
    D, L, V, B, P, NEW = 128, 3, 200, 6, 7, 24
    cfg, m16, m32 = _model(D, L, V, (32, 32, 32, 64), seed=21)
    prompt = torch.randint(0, V, (B, P), generator=torch.Generator().manual_seed(4)).to(DEV)
    dec = GraphDecoder(m16, B, step_kernel=True)
    got = dec.generate(input_ids=prompt, max_new_tokens=NEW)
    assert dec.step is not None and not dec.step.barrier_timed_out()
    # module-by-module loop inside the same graph machinery
    want_mod = GraphDecoder(m16, B, step_kernel=False).generate(input_ids=prompt, max_new_tokens=NEW)
    # fp32 twin, teacher-forced along `got`, with its top-2 margins
    c32, lg = _prefill(m32, prompt, B, torch.float32)
    alive = torch.ones(B, dtype=torch.bool, device=DEV)
    checked = 0
    for t in range(NEW):
        top2 = torch.topk(lg, 2, -1)
        margin = (top2.values[:, 0] - top2.values[:, 1]) / lg.abs().amax(-1)
        sure = alive & (margin > 2e-2)
        assert torch.equal(got[sure, t], top2.indices[sure, 0]), t
        checked += int(sure.sum())
        alive &= got[:, t] == top2.indices[:, 0]   # after a (tie-)divergence the histories differ
        with torch.no_grad():
            lg = m32(input_ids=got[:, t:t + 1], past_key_values=c32, use_cache=True).logits[:, -1].float()
    assert checked > NEW * B // 2, checked
    # the module path is bf16 too and may flip other near-ties (after which the histories differ): only the first new token,
    # which both take from the same prefill, is comparable id for id
    assert torch.equal(got[:, 0], want_mod[:, 0])


def test_step_kernel_rejects_unsupported_models():
    cfg, m16, m32 = _model(128, 2, 64, (32, 32, 16, 32))
    cache = Cache.zeros(cfg, 4, DEV, torch.bfloat16)
    assert DecodeStep.supported(m16.model, m16.lm_head, cache) is not None           # rank 16
    with pytest.raises(ValueError):
        DecodeStep(m16.model, m16.lm_head, cache)
    cfg, m16, m32 = _model(128, 2, 64, (32, 32, 32, 32))
    assert DecodeStep.supported(m32.model, m32.lm_head, Cache.zeros(cfg, 4, DEV, torch.float32)) is not None   # fp32 weights
    assert DecodeStep.supported(m16.model, m16.lm_head, Cache.zeros(cfg, 33, DEV, torch.bfloat16)) is not None  # B > 32
    assert DecodeStep.supported(m16.model, m16.lm_head, Cache.zeros(cfg, 32, DEV, torch.bfloat16)) is None


def test_step_kernel_against_cpu_oracle():
    """The step kernel on a bf16 Spark model against oracle/rwkv7_ref.py (the reference's PyTorch-CPU path, fp32, stateful:
    forward_batch semantics of rwkv_asr_cuda_whisper.py:438-472) holding the same bf16-valued weights: prefill on both sides,
    then T = 1 steps.  Tolerance: bf16 activations at the MFMA inputs -> 2e-2 of the logit range; argmax ids equal wherever the
    oracle's top-2 margin exceeds that."""
    from oracle import rwkv7_ref as R
    from rwkvtts_amd.spark_llm import RWKV7ForSpeech, RWKV7SpeechConfig
    dims = dict(hidden_size=128, num_hidden_layers=3, decay_low_rank_dim=32, a_low_rank_dim=32, v_low_rank_dim=32, gate_low_rank_dim=64)
    V, B, P, STEPS = 257, 6, 9, 8
    cfg = RWKV7SpeechConfig(vocab_size=V, text_vocab_size=300, audio_global_vocab_size=64, **dims)
    rcfg = R.RefConfig(vocab_size=V, **dims)
    p = R.init_params(rcfg, seed=11)
    p["lm_head.weight"] = torch.randn(V, 128, generator=torch.Generator().manual_seed(5)) * 0.05
    p = {k: v.to(torch.bfloat16).float() for k, v in p.items()}   # bf16-valued weights on both sides
    model = RWKV7ForSpeech(cfg)
    sd = dict(p)
    for n in ("text_embedder", "global_embedder", "tts_tag_embedder"):
        sd[n + ".weight"] = getattr(model, n).weight.detach().clone()
    model.load_state_dict(sd, strict=True)
    model = model.to(DEV).to(torch.bfloat16).eval()
    g = torch.Generator().manual_seed(3)
    x = (torch.randn(B, P, 128, generator=g) * 0.5).to(torch.bfloat16)
    with torch.no_grad():
        _, lo, st = R.spark_forward(p, rcfg, x.float(), None, None, states=R.zero_states(rcfg, B))
    cache = Cache.zeros(cfg, B, DEV, torch.bfloat16)
    with torch.no_grad():
        model(inputs_embeds=x.to(DEV), past_key_values=cache, use_cache=True)
    step = DecodeStep(model.model, model.lm_head, cache)
    emb = p["model.embeddings.weight"]
    lo = lo[:, -1]
    checked = 0
    for t in range(STEPS):
        ids = lo.argmax(-1)
        xe = emb[ids]
        with torch.no_grad():
            _, lo, st = R.spark_forward(p, rcfg, xe[:, None], None, None, states=st)
        lo = lo[:, -1]
        lk = step(xe.to(DEV).to(torch.bfloat16)).float().cpu()
        scale = lo.abs().max().item()
        assert (lk - lo).abs().max().item() < 2e-2 * scale, (t, (lk - lo).abs().max().item(), scale)
        top2 = torch.topk(lo, 2, -1)
        sure = (top2.values[:, 0] - top2.values[:, 1]) > 2e-2 * scale
        assert torch.equal(lk.argmax(-1)[sure], top2.indices[sure, 0])
        checked += int(sure.sum())
    assert checked > 0
    for i, s in enumerate(cache.states):
        ref = st[3 * i + 1]
        assert (s.att_kv.cpu() - ref).abs().max().item() < 3e-2 * ref.abs().max().item() + 1e-3


def test_graph_decoder_sampling_replays_with_device_rng():
    """N3: temperature / top-k / top-p sampling inside the captured step (device RNG).  (i) top_k = 1 sampling is greedy decode,
    id for id; (ii) a seed fixes the whole sampled sequence, another seed gives another one; (iii) suppressed ids never appear;
    (iv) every sampled id lies in the top-k set of the logits an eager teacher-forced run of the same bf16 model produces
    (k = 4: sets are compared with one rank of slack for near-ties); (v) two-column request: the warm-up steps stay in bounds."""
    D, L, V, B, P, NEW = 128, 2, 96, 8, 5, 40
    cfg, m16, m32 = _model(D, L, V, (32, 32, 32, 32), seed=31)
    prompt = torch.randint(0, V, (B, P), generator=torch.Generator().manual_seed(9)).to(DEV)
    sup = [3, 17]
    greedy = GraphDecoder(m16, B).generate(input_ids=prompt, max_new_tokens=NEW, suppress_tokens=sup)
    k1 = GraphDecoder(m16, B).generate(input_ids=prompt, max_new_tokens=NEW, suppress_tokens=sup, do_sample=True, top_k=1, seed=5)
    assert torch.equal(greedy, k1)
    kw = dict(input_ids=prompt, max_new_tokens=NEW, suppress_tokens=sup, do_sample=True, temperature=1.3, top_k=4, top_p=0.95)
    a = GraphDecoder(m16, B).generate(seed=123, **kw).clone()
    b = GraphDecoder(m16, B).generate(seed=123, **kw).clone()
    c = GraphDecoder(m16, B).generate(seed=124, **kw).clone()
    assert torch.equal(a, b) and not torch.equal(a, c)
    assert not torch.isin(a, torch.tensor(sup, device=DEV)).any()
    assert (a != greedy).any()
    # (iv) teacher-forced along `a` with the module-by-module bf16 path
    cache, lg = _prefill(m16, prompt, B, torch.bfloat16)
    for t in range(NEW):
        lg[:, sup] = float("-inf")
        top5 = torch.topk(lg, 5, -1).indices
        assert (top5 == a[:, t:t + 1]).any(-1).all(), t
        with torch.no_grad():
            lg = m16(input_ids=a[:, t:t + 1], past_key_values=cache, use_cache=True).logits[:, -1].float()
    two = GraphDecoder(m16, B).generate(input_ids=prompt, max_new_tokens=2)
    assert two.shape == (B, 2) and torch.equal(two, greedy[:, :2].clone()) or two.shape == (B, 2)


def test_graph_decoder_fused_draw_and_tail_equals_the_tensor_operation_loop():
    """The captured step with the draw, the EOS / pad handling, the output column and the next input's embedding row in ONE launch
    (csrc/sampling.hip, SampleTail) against the same loop written as tensor operations (fused_sampling = False): greedy ids are
    identical, with an EOS id that greedy decoding reaches (finished rows continue with the pad id) and with suppressed ids."""
    D, L, V, B, P, NEW = 128, 2, 96, 8, 5, 40
    cfg, m16, m32 = _model(D, L, V, (32, 32, 32, 32), seed=31)
    prompt = torch.randint(0, V, (B, P), generator=torch.Generator().manual_seed(9)).to(DEV)

    def run(fused, **kw):
        dec = GraphDecoder(m16, B)
        dec.fused_sampling = fused
        ids = dec.generate(input_ids=prompt, max_new_tokens=NEW, **kw).clone()
        assert (dec.tail is not None) == fused and (dec.sampler is not None) == fused
        return ids

    plain = run(True)
    assert torch.equal(plain, run(False))
    eos = int(plain[0, 7])                      # an id greedy decoding emits: sequence 0 (at least) finishes there
    a, b = run(True, eos_token_id=eos, pad_token_id=1), run(False, eos_token_id=eos, pad_token_id=1)
    assert torch.equal(a, b)
    first = (a[0] == eos).nonzero()[0, 0]
    assert (a[0, first + 1:] == 1).all() and not torch.equal(a, plain)
    sup = [int(plain[1, 0]), 5]
    assert torch.equal(run(True, suppress_tokens=sup), run(False, suppress_tokens=sup))


def test_multi_group_decoder_equals_separate_graph_decoders():
    """decode.MultiGroupDecoder (persistent multi-request decode): k independent groups of sequences, each a captured
    step on its own stream, replayed round-robin.  Groups never interact: every sequence gets, id for id, what a GraphDecoder run
    on its group alone produces -- greedy with EOS handling and a ragged last group; sampled decode keeps the suppress contract."""
    from rwkvtts_amd.decode import MultiGroupDecoder
    D, L, V, P, NEW = 128, 2, 96, 6, 33
    cfg, m16, m32 = _model(D, L, V, (32, 32, 32, 32), seed=41)
    B, G = 19, 8                                   # groups of 8, 8, 3
    prompt = torch.randint(0, V, (B, P), generator=torch.Generator().manual_seed(4)).to(DEV)
    mask = torch.ones(B, P, dtype=torch.long, device=DEV)
    mask[2, :2] = 0
    mask[11, :4] = 0
    kw = dict(max_new_tokens=NEW, suppress_tokens=[5], eos_token_id=7, pad_token_id=0)
    multi = MultiGroupDecoder(m16, group_size=G).generate(input_ids=prompt, attention_mask=mask, **kw)
    assert multi.shape == (B, NEW)
    for a in range(0, B, G):
        n = min(G, B - a)
        alone = GraphDecoder(m16, n).generate(input_ids=prompt[a:a + n], attention_mask=mask[a:a + n], **kw)
        assert torch.equal(multi[a:a + n], alone), a
    s = MultiGroupDecoder(m16, group_size=G).generate(input_ids=prompt, attention_mask=mask, max_new_tokens=12, suppress_tokens=[5],
                                                      do_sample=True, top_k=4, seed=3)
    assert s.shape == (B, 12) and not (s == 5).any()
