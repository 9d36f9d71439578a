"""The Spark head (V = 8193, odd) inside fused_linear_cross_entropy: time of the whole node (forward incl. the in-forward gradient
GEMMs) and of its three GEMM shapes with the real V against a V padded to a multiple of 64.   python tools/ce_head_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rwkvtts_amd.losses import fused_linear_cross_entropy
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
N, D = 32768, 1024


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for s, e in ev:
        s.record(); fn(); e.record()
    torch.cuda.synchronize()
    ts = sorted(s.elapsed_time(e) * 1e3 for s, e in ev)
    return ts[len(ts) // 2]


h = (torch.randn(N, D, device=dev, generator=g) * 0.5).bfloat16().requires_grad_(True)
for V in (8193, 8256):
    W = (torch.randn(V, D, device=dev, generator=g) * 0.02).bfloat16().requires_grad_(True)
    lab = torch.randint(0, 8193, (N,), device=dev, generator=g)
    def node():
        h.grad = None; W.grad = None
        loss = fused_linear_cross_entropy(h, lab, W)
        loss.backward()
    t = timed(node)
    print(f"V = {V}: fused_linear_cross_entropy fwd + bwd on [{N}, {D}]: {t / 1e3:.3f} ms", flush=True)
    hc = h.detach()[:4096]
    Wd = W.detach()
    pd = torch.empty(4096, V, device=dev, dtype=torch.bfloat16).normal_(generator=g)
    t1 = timed(lambda: torch.nn.functional.linear(hc, Wd))
    t2 = timed(lambda: pd @ Wd)
    t3 = timed(lambda: pd.t() @ hc)
    fl = 2.0 * 4096 * V * D
    print(f"    per 4096-row chunk: logits {t1:.1f} us ({fl / t1 / 1e6:.0f} TF/s), dh = pd @ W {t2:.1f} us ({fl / t2 / 1e6:.0f}), dW = pd^T @ h {t3:.1f} us ({fl / t3 / 1e6:.0f})", flush=True)
    if V % 64 == 0:
        Wt = Wd.t().contiguous()
        t2b = timed(lambda: pd @ Wt.t())
        print(f"    dh on W^T (NT): {t2b:.1f} us ({fl / t2b / 1e6:.0f} TF/s)")
