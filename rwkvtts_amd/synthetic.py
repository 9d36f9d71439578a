"""Synthetic inputs of the shapes BASELINE.json names (there is no network for datasets/checkpoints).

make_wkv_inputs: operator-level inputs in the trained model's range:
    w = -softplus(-z) - 0.5 (so exp(-exp(w)) in [e^-0.6065, 1)), kk unit-norm per head,
    a = -kk, b = kk * sigmoid(.)  -- exactly how RWKV_Tmix_x070.forward feeds the kernel
    (model/llm/rwkv_s2s_single_ffn.py:172,186-191).
"""
import torch
import torch.nn.functional as F


def make_wkv_inputs(B, T, H, seed=0, dtype=torch.bfloat16, device="cpu", scale=0.5):
    """Returns (w, q, k, v, a, b), each [B,T,H,64] contiguous -- the argument order of
    torch.ops.wind_backstepping.forward."""
    g = torch.Generator().manual_seed(seed)
    N = 64

    def rn(s=1.0):
        return torch.randn(B, T, H, N, generator=g) * s

    q, k, v = rn(scale), rn(scale), rn(scale)
    w = -F.softplus(-(rn(2.0) - 1.0)) - 0.5
    kk = F.normalize(rn(), dim=-1)
    a = -kk
    b = kk * torch.sigmoid(rn())
    return [t.to(dtype).contiguous().to(device) for t in (w, q, k, v, a, b)]
