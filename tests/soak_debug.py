"""Debug helper for tests/test_chunk_soak_gpu.py: one soak case against an fp64 autograd scan (truth), the C oracle and the HIP pair.
    python tests/soak_debug.py CASE"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rwkvtts_amd import ops
from test_chunk_soak_gpu import soak_inputs
from oracle import c_oracle as co
co.build()
NAMES = ["dw", "dq", "dk", "dv", "da", "db"]


def truth(ins, dy):
    w, q, k, v, a, b = [t.double().requires_grad_(True) for t in ins]
    B, T, H, N = w.shape
    S = torch.zeros(B, H, N, N, dtype=torch.float64)      # S[v][k]
    ys = []
    for t in range(T):
        dec = torch.exp(-torch.exp(w[:, t]))              # [B,H,N] over k
        sa = torch.einsum("bhvk,bhk->bhv", S, a[:, t])
        S = S * dec[:, :, None, :] + sa[..., None] * b[:, t][:, :, None, :] + v[:, t][..., None] * k[:, t][:, :, None, :]
        ys.append(torch.einsum("bhvk,bhk->bhv", S, q[:, t]))
    y = torch.stack(ys, 1)
    y.backward(dy.double())
    return y.detach(), [t.grad for t in (w, q, k, v, a, b)]


case = int(sys.argv[1])
mode, shape, ins, dy = soak_inputs(case)
print(mode, shape)
y_t, g_t = truth(ins, dy)
y_o, s_o, sa_o = co.wkv7_fwd(*ins)
g_o = co.wkv7_bwd(*ins, dy, s_o, sa_o)
d = [t.cuda() for t in ins]
y, tinv, sa, hs = ops.wkv7_chunk_forward(*d)
g_h = ops.wkv7_chunk_backward(*d, dy.cuda(), hs, sa, tinv)
torch.cuda.synchronize()


def report(name, t, o, h):
    t, o, h = t.double(), o.double(), h.double().cpu()
    ulp = 2.0 ** -8 * torch.clamp(t.abs(), min=t.abs().mean() * 0.25 + 1e-6)     # half a bf16 ulp of the truth = what RNE of the truth may cost
    eo, eh = ((o - t).abs() / ulp), ((h - t).abs() / ulp)
    print(f"{name:3s} max|truth| {t.abs().max():.3e} mean {t.abs().mean():.3e} | oracle: max err {(o - t).abs().max():.3e} = {eo.max():.2f} half-ulps "
          f"| HIP: max err {(h - t).abs().max():.3e} = {eh.max():.2f} half-ulps | HIP vs oracle max {(h - o).abs().max():.3e}")


report("y", y_t, y_o, y)
for n, t, o, h in zip(NAMES, g_t, g_o, g_h):
    report(n, t, o, h)
