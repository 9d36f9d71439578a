"""Decode-shaped products (M = 32 rows): which BLAS formulation is fastest?  F.linear(x, W) vs x @ W^T stored contiguous,
vs a batched call for r,k,v."""
import os, sys, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_wkv import timeit
dev = "cuda:0"
M = 32
for K, N in ((1024, 1024), (1024, 4096), (4096, 1024), (1024, 64), (64, 1024), (1024, 8193)):
    x = torch.randn(M, K, device=dev).bfloat16()
    w = (torch.randn(N, K, device=dev) * 0.02).bfloat16()
    wt = w.t().contiguous()
    a = timeit(lambda: F.linear(x, w), 50)[0]
    b = timeit(lambda: x @ wt, 50)[0]
    c = timeit(lambda: (w @ x.t()), 50)[0]
    byts = N * K * 2
    print(f"K={K:5d} N={N:5d}: F.linear {a*1e3:6.1f} us | x@Wt {b*1e3:6.1f} us | W@x^T {c*1e3:6.1f} us | weight bytes {byts/1e6:5.1f} MB -> {byts/min(a,b,c)/1e6:6.0f} GB/s best")
x3 = torch.randn(3, M, 1024, device=dev).bfloat16()
w3 = (torch.randn(3, 1024, 1024, device=dev) * 0.02).bfloat16()
print("bmm r,k,v (3 x 1024x1024):", timeit(lambda: torch.bmm(x3, w3.transpose(1, 2)), 50)[0] * 1e3, "us;  bmm with pre-transposed weights:",
      timeit(lambda: torch.bmm(x3, w3), 50)[0] * 1e3, "us")
