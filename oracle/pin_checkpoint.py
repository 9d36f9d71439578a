"""Pins rwkvtts_amd/checkpoint.py:fla_to_x070 against the reference's converter (authoring container only).

utils/convert_rwkv.py is a script (reads sys.argv at import), so its rename loop (source lines 15-41) is executed here
on a toy rwkvfla-style state dict; only the resulting DATA -- output keys and tensors -- is committed as
tests/golden/convert_keys.npz.        python oracle/pin_checkpoint.py [--write]
"""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
REF = "/root/reference/utils/convert_rwkv.py"


def toy_fla_state_dict(D=8, L=2, seed=0):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g)
    sd = {"model.embeddings.weight": r(11, D), "model.norm.weight": r(D), "model.norm.bias": r(D), "lm_head.weight": r(11, D),
          "text_embedder.weight": r(5, D), "global_embedder.weight": r(4, D), "tts_tag_embedder.weight": r(3, D)}
    for i in range(L):
        p = f"model.layers.{i}."
        if i == 0:
            sd[p + "pre_norm.weight"], sd[p + "pre_norm.bias"] = r(D), r(D)
        for n in ("attn_norm", "ffn_norm"):
            sd[p + n + ".weight"], sd[p + n + ".bias"] = r(D), r(D)
        if i == 0:
            sd[p + "attn.x_x"] = r(6, D)
        else:
            for n in "rwkvag":
                sd[p + f"attn.x_{n}"] = r(1, 1, D)
        for n in ("k_k", "k_a"):
            sd[p + "attn." + n] = r(D)
        sd[p + "attn.r_k"] = r(2, D // 2)
        for n in ("r_proj", "k_proj", "v_proj", "o_proj"):
            sd[p + f"attn.{n}.weight"] = r(D, D)
        for n, rank in (("w", 3), ("a", 3), ("g", 4)) + ((("v", 2),) if i else ()):
            sd[p + f"attn.{n}_lora.lora.0.weight"] = r(rank, D)
            sd[p + f"attn.{n}_lora.lora.2.weight"] = r(D, rank)
            if n != "g":
                sd[p + f"attn.{n}_lora.lora.2.bias"] = r(D)
        sd[p + "attn.g_norm.weight"], sd[p + "attn.g_norm.bias"] = r(D), r(D)
        sd[p + "ffn.x_k"] = r(D)
        sd[p + "ffn.key.weight"], sd[p + "ffn.value.weight"] = r(4 * D, D), r(D, 4 * D)
    return sd


def reference_rename(sd):
    lines = open(REF).read().split("\n")
    src = "\n".join(lines[13:41])          # `w_new = {}` ... end of the rename loop (file lines 14-41)
    env = {"w": dict(sd), "torch": torch, "print": lambda *a, **k: None}
    exec(compile(src, REF, "exec"), env)
    return env["w_new"]


def main():
    from rwkvtts_amd import checkpoint as C
    sd = toy_fla_state_dict()
    ref = reference_rename(sd)
    got = C.fla_to_x070(sd)
    assert set(ref) == set(got), (sorted(set(ref) ^ set(got)))
    for k in ref:
        assert ref[k].shape == got[k].shape and torch.equal(ref[k].contiguous(), got[k].contiguous()), k
    back = C.x070_to_fla(got)
    want = C.split_x_x(sd)
    assert set(back) == set(want), sorted(set(back) ^ set(want))
    for k in want:
        assert torch.equal(back[k].contiguous(), want[k].contiguous()), k
    print(f"fla_to_x070 == reference rename loop on {len(sd)} tensors -> {len(ref)}; x070_to_fla inverts it")
    if "--write" in sys.argv:
        out = {"in." + k: v.numpy() for k, v in sd.items()}
        out.update({"ref." + k: v.contiguous().numpy() for k, v in ref.items()})
        np.savez_compressed(os.path.join(REPO, "tests", "golden", "convert_keys.npz"), **out)
        print("wrote tests/golden/convert_keys.npz")


if __name__ == "__main__":
    main()
