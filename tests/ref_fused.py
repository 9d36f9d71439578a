"""TEST REFERENCE (not product code): plain PyTorch restatement of the fused elementwise stages, used by
tests/test_fused_gpu.py as the fp32 reference for rwkvtts_amd/fused.py (the HIP kernels) and for their
gradients via torch.autograd.  Formulas: model/llm/rwkv_s2s_single_ffn.py:160-195,224-229."""
import torch
import torch.nn.functional as F



def _shift(x, x_prev):
    """x_{t-1}; zeros (training, rwkv_s2s_single_ffn.py:162 ZeroPad2d) or the carried row at t = 0."""
    if x_prev is None:
        return F.pad(x, (0, 0, 1, -1))
    return torch.cat([x_prev.unsqueeze(1).to(x.dtype), x[:, :-1]], dim=1)


def token_shift_mix6(x, x_prev, x_r, x_w, x_k, x_v, x_a, x_g):
    """xx = shift(x) - x ; x + xx * x_?  for ? in r,w,k,v,a,g   (rwkv_s2s_single_ffn.py:162-169)."""
    xx = _shift(x, x_prev) - x
    return tuple(torch.addcmul(x, xx, p.view(1, 1, -1)) for p in (x_r, x_w, x_k, x_v, x_a, x_g))


def token_shift_mix1(x, x_prev, x_k):
    """channel-mix input: x + (shift(x) - x) * x_k   (rwkv_s2s_single_ffn.py:225-227)."""
    xx = _shift(x, x_prev) - x
    return torch.addcmul(x, xx, x_k.view(1, 1, -1))


def relu_sq(x):
    """relu(x)^2   (rwkv_s2s_single_ffn.py:228)."""
    return torch.relu(x).square()


def tmix_prepare(w_pre, k, v, a_pre, v_pre, v_first, k_k, k_a, mask, H, is_layer0):
    """Everything between the projections and the scan (rwkv_s2s_single_ffn.py:172-190):
        w  = (-softplus(-w_pre) - 0.5) * mask
        k  = k * mask ; v = v * mask
        v  = v + (v_first - v) * sigmoid(v_pre)                    (layers > 0)
        a  = sigmoid(a_pre)
        kk = l2norm_per_head(k * k_k) * mask
        k2 = k * (1 + (a - 1) * k_a) ; v2 = v * mask
    returns w, k2, v2, -kk, kk * a   (the scan's w, k, v, a, b operands)."""
    B, T, D = k.shape
    w = -F.softplus(-w_pre) - 0.5
    if mask is not None:
        w, k, v = w * mask, k * mask, v * mask
    if not is_layer0:
        v = v + (v_first - v) * torch.sigmoid(v_pre)
    a = torch.sigmoid(a_pre)
    kk = F.normalize((k * k_k.view(1, 1, D)).view(B, T, H, -1), dim=-1, p=2.0).view(B, T, D)
    if mask is not None:
        kk = kk * mask
    k2 = k * (1 + (a - 1) * k_a.view(1, 1, D))
    if mask is not None:
        v = v * mask
    return w.contiguous(), k2.contiguous(), v.contiguous(), (-kk).contiguous(), (kk * a).contiguous()


def tmix_post(y, r, k, v, g, gn_weight, gn_bias, r_k, H, eps):
    """After the scan (rwkv_s2s_single_ffn.py:192-195): GroupNorm over each head, the (r.k.r_k) v bonus, gate."""
    B, T, D = y.shape
    N = D // H
    yn = F.group_norm(y.reshape(B * T, D), H, gn_weight, gn_bias, eps).view(B, T, D)
    bonus = (r.view(B, T, H, N) * k.view(B, T, H, N) * r_k.view(1, 1, H, N)).sum(-1, keepdim=True) * v.view(B, T, H, N)
    return (yn + bonus.view(B, T, D)) * g
