"""Low-rank weight gradients at the training step's sizes: rwkv7_wgrad_skinny_bf16 + reduction against the batched-GEMM route."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rwkvtts_amd import fused
dev = "cuda:0"
M = 32768


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def blas(dy, x):
    S = 8
    part = torch.bmm(dy.view(S, M // S, -1).transpose(1, 2), x.view(S, M // S, -1), out_dtype=torch.float32)
    return part.sum(0).to(torch.bfloat16)


for N, K in ((64, 1024), (1024, 64), (32, 1024), (1024, 32), (128, 1024), (1024, 128)):
    dy = torch.randn(M, N, device=dev).bfloat16()
    x = torch.randn(M, K, device=dev).bfloat16()
    t_new = timeit(lambda: fused.wgrad_splitk(dy, x))
    t_old = timeit(lambda: blas(dy, x))
    mb = (M * (N + K) * 2) / 1e6
    print(f"dW[{N:4d}][{K:4d}]: kernel + reduction {t_new:6.1f} us ({mb / t_new:5.2f} TB/s of operand bytes)   batched GEMM + torch sum {t_old:6.1f} us", flush=True)
