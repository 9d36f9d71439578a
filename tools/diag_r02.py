"""Diagnostics behind the bars of tests/test_configs_gpu.py (run on the GPU box):  python tools/diag_r02.py [cfg0] [cfg4]"""
import copy, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
DEV = "cuda:0"


def cfg0():
    from oracle import rwkv7_ref as R
    from rwkvtts_amd import layouts as L
    import test_configs_gpu as T
    model, p, rcfg, SV = T._cosy_0p1b()
    batch = L.synthetic_cosy_batch(2, seed=1234)
    R.pick_threads()
    # oracle gradient w.r.t. the input embeddings, per position
    pr = {k: v.clone() for k, v in p.items()}
    m32 = model.to(DEV).eval()
    emb, mask, labels = m32.build_inputs(T._to(batch, DEV))
    x_o = emb.detach().cpu().clone().requires_grad_(True)
    h, _ = R.backbone(pr, rcfg, x_o, mask.cpu().float(), None)
    logits = h @ pr["lm_head.weight"].t() + pr["lm_head.bias"]
    loss_o = R.label_smoothing_kl_ref(logits, labels.cpu(), SV + 1, -1, 0.0, True)
    loss_o.backward()
    go = x_o.grad
    from rwkvtts_amd.losses import label_smoothing_kl
    from rwkvtts_amd import fused
    for name, mdl, dt in (("fp32", m32, torch.float32), ("bf16", copy.deepcopy(m32).to(torch.bfloat16), torch.bfloat16),
                          ("bf16 scalar WKV kernels", copy.deepcopy(m32).to(torch.bfloat16), torch.bfloat16)):
        fused.CHUNKED_WKV_FWD = fused.CHUNKED_WKV_BWD = not name.endswith("kernels")
        mdl.train()
        x = emb.detach().to(dt).clone().requires_grad_(True)
        out = mdl(inputs_embeds=x, attention_mask=mask, labels=labels)
        out.loss.backward()
        g = x.grad.float().cpu()
        rel = (g - go).norm(dim=-1) / go.norm(dim=-1).clamp(min=1e-12)     # [B, T]
        print(f"{name}: loss {out.loss.item():.5f} (oracle {loss_o.item():.5f}); d_inputs_embeds rel. L2 error by position:")
        for t in (0, 1, 2, 3, 8, 31, 32, 33, 64, 126, 127, 128, 129, 256, 400, 510, 511):
            print(f"   t={t:4d}: {rel[0, t].item():.3e} {rel[1, t].item():.3e}   |g| {go[0, t].norm().item():.3e}")
        print(f"   overall {((g - go).norm() / go.norm()).item():.3e}; worst position {rel.max().item():.3e} at {rel.argmax().item()}")


def cfg4():
    from rwkvtts_amd import backbone
    from rwkvtts_amd.backbone import Cache
    from rwkvtts_amd.decode import GraphDecoder
    from rwkvtts_amd.spark_llm import RWKV7ForSpeech, RWKV7SpeechConfig
    base = backbone.config_0p4b()
    kw = {k: v for k, v in base.to_dict().items() if k in backbone.RWKV7Config.__dataclass_fields__ and k != "extra"}
    cfg = RWKV7SpeechConfig(**kw)
    m32 = RWKV7ForSpeech(cfg).init_weights(seed=0)
    with torch.no_grad():
        m32.lm_head.weight.mul_(4.0)
        for p_ in m32.parameters():
            p_.copy_(p_.to(torch.bfloat16).float())
    m32 = m32.to(DEV).eval()
    m16 = copy.deepcopy(m32).to(torch.bfloat16).eval()
    B, P = 32, 128
    g = torch.Generator().manual_seed(1234)
    emb = (torch.randn(B, P, cfg.hidden_size, generator=g) * 0.5).to(DEV)
    mask = torch.ones(B, P, dtype=torch.long, device=DEV)
    eos = cfg.vocab_size - 1
    snaps = {}
    for n in (1, 64, 256, 1024, 2048):
        dec = GraphDecoder(m16, B)
        got = dec.generate(inputs_embeds=emb.to(torch.bfloat16), attention_mask=mask, max_new_tokens=n + 1, suppress_tokens=[eos])
        snaps[n] = ([s.att_kv.clone() for s in dec.cache.states], got)
    got = snaps[2048][1]
    c32 = Cache.zeros(cfg, B, DEV, torch.float32)
    with torch.no_grad():
        m32(inputs_embeds=emb, attention_mask=mask, past_key_values=c32, use_cache=True, logits_to_keep=1)
    # module-by-module bf16 path too (same ids)
    c16 = Cache.zeros(cfg, B, DEV, torch.bfloat16)
    from rwkvtts_amd import fused
    with torch.no_grad():
        m16(inputs_embeds=emb.to(torch.bfloat16), attention_mask=mask, past_key_values=c16, use_cache=True, logits_to_keep=1)
    for t in range(2048):
        with torch.no_grad():
            m32(input_ids=got[:, t:t + 1], past_key_values=c32, use_cache=True)
        if t + 1 in snaps:
            st = snaps[t + 1][0]
            same = bool((snaps[t + 1][1][:, :t + 1] == got[:, :t + 1]).all())
            rels = [((a - b.att_kv).norm() / b.att_kv.norm()).item() for a, b in zip(st, c32.states)]
            print(f"after {t + 1:5d} steps (ids prefix equal: {same}): state rel. L2 drift per layer: " + " ".join(f"{r:.3f}" for r in rels))




def wkv0():
    """WKV-level gradients at the first steps: chunked bf16 pair vs the C oracle, inputs in the trained range, T = 512."""
    from oracle import c_oracle
    from rwkvtts_amd import ops
    from rwkvtts_amd.synthetic import make_wkv_inputs
    c_oracle.build()
    B, T, H = 2, 512, 12
    ins = make_wkv_inputs(B, T, H, 2, torch.bfloat16)
    dy = torch.randn(B, T, H, 64, generator=torch.Generator().manual_seed(102)).bfloat16()
    y_o, s_o, sa_o = c_oracle.wkv7_fwd(*ins)
    g_o = c_oracle.wkv7_bwd(*ins, dy, s_o, sa_o)
    d = [t.to(DEV) for t in ins]
    y, tinv, sa, hs = ops.wkv7_chunk_forward(*d)
    grads = ops.wkv7_chunk_backward(*d, dy.to(DEV), hs, sa, tinv)
    for n, g, go in zip(("dw", "dq", "dk", "dv", "da", "db"), grads, g_o):
        g = g.float().cpu(); go = go.float()
        err = (g - go).norm(dim=-1) / go.norm(dim=-1).mean()      # [B,T,H] relative to the mean row norm
        print(f"{n}: mean row norm {go.norm(dim=-1).mean().item():.3e}; rel err by t: " +
              " ".join(f"t{t}:{err[:, t].mean().item():.1e}" for t in (0, 1, 2, 3, 15, 16, 31, 32, 33, 63, 64, 255, 511)) +
              f" | |oracle| at t0 {go[:, 0].norm(dim=-1).mean().item():.2e}")


if __name__ == "__main__":
    which = sys.argv[1:] or ["cfg0", "cfg4"]
    for w in which:
        globals()[w]()
