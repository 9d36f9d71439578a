"""oracle/pin_bf16_greedy.py -- AUTHORING-CONTAINER ONLY (needs /root/reference).

`north_star`: "bit-exact argmax token ids for greedy decode".  The reference decodes with bf16 modules and an fp32 recurrent state
(model/llm/rwkv_asr_cuda_whisper.py:438-472 `RWKV7ModelForCausalLMCuda.forward_batch`, greedy loop :694-717).  This script runs
THAT code on CPU in bf16 -- the reference's own model class, blocks, LayerNorms, head and loop arithmetic, every tensor bf16 where
the reference's is, with `RWKV7_BATCH_OP` (the CUDA op that cannot exist here) routed to the C oracle's bf16 state-carrying scan --
greedy, 256 steps on a toy model, and commits prompt, bf16 weights, the ids and the per-step top-2 margins of the reference's
logits as tests/golden/bf16_greedy.npz.  tests/test_bf16_greedy_gpu.py then reports id-for-id agreement of the HIP decode
(GraphDecoder free-running, and DecodeStep teacher-forced along the reference's ids).

One deviation, the same as in pin_against_reference.py [3]: `RWKV_Tmix_x070.forward_batch` hands back `x[:,-1,:]` AFTER `x` was
re-bound to the block output (:213-215), i.e. the attention OUTPUT as token-shift state.  The TTS path's decode (rwkvfla's
conv_state; RWKV_x070_TMix_one, rwkv_s2s_single_ffn.py:506) carries the last INPUT.  The method is wrapped (not edited) so that it
returns its input's last row; everything else is the reference's.       Usage: python oracle/pin_bf16_greedy.py [--write]
"""
import argparse
import os
import sys
from argparse import Namespace

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import c_oracle, ref_import  # noqa: E402
from oracle import rwkv7_ref as R  # noqa: E402
from oracle.pin_against_reference import fla_to_x070  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden", "bf16_greedy.npz")
CFG = dict(hidden_size=128, num_hidden_layers=3, decay_low_rank_dim=32, a_low_rank_dim=32, v_low_rank_dim=32, gate_low_rank_dim=128)   # the LoRA sizes the reference's formulas give at D = 128
V, B, P, STEPS = 256, 4, 16, 256


def block_state_dict(p, cfg, layer_id):
    tm, _ = fla_to_x070(p, cfg, layer_id)
    pre, D, sd = f"model.layers.{layer_id}.", cfg.hidden_size, {}
    for n in "rwkvag":
        sd[f"att.x_{n}"] = tm[f"x_{n}"].view(1, 1, D)
    for n in ("w", "a", "v"):
        if n == "v" and layer_id == 0:
            continue
        sd[f"att.{n}0"] = tm[f"{n}0"].view(1, 1, D)
        sd[f"att.{n}1"], sd[f"att.{n}2"] = tm[f"{n}1"], tm[f"{n}2"]
    sd["att.g1"], sd["att.g2"] = tm["g1"], tm["g2"]
    sd["att.k_k"], sd["att.k_a"] = tm["k_k"].view(1, 1, D), tm["k_a"].view(1, 1, D)
    sd["att.r_k"] = p[pre + "attn.r_k"]
    for a, b in (("receptance", "r_proj"), ("key", "k_proj"), ("value", "v_proj"), ("output", "o_proj")):
        sd[f"att.{a}.weight"] = p[pre + f"attn.{b}.weight"]
    sd["att.ln_x.weight"], sd["att.ln_x.bias"] = tm["ln_w"], tm["ln_b"]
    sd["ffn.x_k"] = p[pre + "ffn.x_k"].view(1, 1, D)
    sd["ffn.key.weight"], sd["ffn.value.weight"] = p[pre + "ffn.key.weight"], p[pre + "ffn.value.weight"]
    for a, b in (("ln1", "attn_norm"), ("ln2", "ffn_norm")) + ((("ln0", "pre_norm"),) if layer_id == 0 else ()):
        sd[f"{a}.weight"], sd[f"{a}.bias"] = p[pre + b + ".weight"], p[pre + b + ".bias"]
    return sd


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--write", action="store_true")
    a = ap.parse_args()
    torch.set_num_threads(4)
    ref_b = ref_import.import_batch_twin()
    c_oracle.build()
    cfg = R.RefConfig(vocab_size=V, **CFG)
    g = torch.Generator().manual_seed(21)
    p = R.init_params(cfg, seed=41)
    p["model.embeddings.weight"] = torch.randn(V, cfg.hidden_size, generator=g) * 0.5
    p["lm_head.weight"] = torch.randn(V, cfg.hidden_size, generator=g) * 0.08
    p = {k: v.bfloat16() for k, v in p.items()}          # the checkpoint IS bf16: both sides load these exact values
    args = Namespace(n_layer=cfg.num_hidden_layers, n_embd=cfg.hidden_size, head_size_a=64, head_size=64, head_size_divisor=8,
                     dropout=0, need_init_tmix=False, need_init_cmix=False, grad_cp=0, vocab_size=V)
    model = ref_b.RWKV7ModelForCausalLMCuda(args)
    sd = {"emb.weight": p["model.embeddings.weight"], "head.weight": p["lm_head.weight"],
          "ln_out.weight": p["model.norm.weight"], "ln_out.bias": p["model.norm.bias"]}
    pf = {k: v.float() for k, v in p.items()}
    for i in range(cfg.num_hidden_layers):
        sd.update({f"blocks.{i}.{k}": v for k, v in block_state_dict(pf, cfg, i).items()})
    model.load_state_dict(sd, strict=True)
    model = model.to(torch.bfloat16).eval()                # bf16 modules (DTYPE, rwkv_asr_cuda_whisper.py:49)

    def batch_op(state, r, w, k, v, a_, b_):               # RWKV7_BATCH_OP (:80-81): bf16 tensors, fp32 state updated in place
        assert all(t.dtype == torch.bfloat16 for t in (r, w, k, v, a_, b_)) and state.dtype == torch.float32
        return c_oracle.wkv7_state_fwd(state, *[t.contiguous() for t in (r, w, k, v, a_, b_)])

    ref_b.RWKV7_BATCH_OP = batch_op
    for blk in model.blocks:                               # the token-shift state: last INPUT row (docstring), by wrapping
        orig = blk.att.forward_batch

        def wrapped(x, attention_mask=None, v_first=None, x_prev=None, state=None, _orig=orig):
            out, v_first, _, state = _orig(x, attention_mask, v_first, x_prev, state)
            return out, v_first, x.mul(attention_mask)[:, -1, :], state

        blk.att.forward_batch = wrapped

    D, H = cfg.hidden_size, cfg.num_heads
    states = []
    for _ in range(cfg.num_hidden_layers):                 # :443-447 with device="cpu"
        states += [torch.zeros(B, D, dtype=torch.bfloat16), torch.zeros(B, H, 64, 64, dtype=torch.float32),
                   torch.zeros(B, D, dtype=torch.bfloat16)]
    prompt = (torch.randn(B, P, D, generator=g) * 0.5).bfloat16()
    ids, margins, top1 = [], [], []
    with torch.inference_mode():
        x, logits, states = model.forward_batch(prompt, None, states)
        for step in range(STEPS):
            lg = logits.float()
            t2 = torch.topk(lg, 2, dim=-1)
            nxt = t2.indices[:, 0]                          # sample_logits(top_k=1) == argmax (:700,707)
            ids.append(nxt.clone())
            margins.append((t2.values[:, 0] - t2.values[:, 1]) / lg.abs().amax(-1))
            top1.append(t2.values[:, 0].clone())
            x, logits, states = model.forward_batch(model.emb(nxt).unsqueeze(1), None, states)
    ids, margins = torch.stack(ids, 1), torch.stack(margins, 1)
    print(f"reference bf16 greedy: ids {tuple(ids.shape)}, distinct ids per row {[len(set(r.tolist())) for r in ids]}, "
          f"relative top-2 margin: median {margins.median().item():.4f}, min {margins.min().item():.5f}, "
          f"{(margins < 0.01).float().mean().item() * 100:.1f}% of the positions below 1% of the logit range")
    print("  first 16 ids of row 0:", ids[0, :16].tolist())
    if a.write:
        gold = {"p." + k: v.view(torch.int16).numpy() for k, v in p.items()}     # bf16 bit patterns
        gold.update(prompt=prompt.view(torch.int16).numpy(), ids=ids.numpy(), margins=margins.numpy(),
                    cfg=np.asarray([V, B, P, STEPS, cfg.hidden_size, cfg.num_hidden_layers]))
        np.savez_compressed(GOLD, **gold)
        print("wrote", GOLD, f"({os.path.getsize(GOLD) / 1e6:.2f} MB)")


if __name__ == "__main__":
    main()
