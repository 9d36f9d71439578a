"""GEMM shapes of the training step, timed hot (same operands every call) and cold (cycling through 24 operand sets, as the
24 layers of the step do).  python tools/bench_gemm_cold.py"""
import os
import sys

sys.path.insert(0, os.getcwd())
import torch

M = 32768
dev = "cuda:0"


def timeit(fn, n):
    for i in range(3):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    torch.manual_seed(0)
    for N, K in ((1024, 1024), (4096, 1024), (1024, 4096)):
        S = 24
        xs = [torch.randn(M, K, device=dev, dtype=torch.bfloat16) for _ in range(S)]
        ws = [torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.02 for _ in range(S)]
        dys = [torch.randn(M, N, device=dev, dtype=torch.bfloat16) for _ in range(S)]
        outs = [torch.empty(M, N, device=dev, dtype=torch.bfloat16) for _ in range(S)]
        dxs = [torch.empty(M, K, device=dev, dtype=torch.bfloat16) for _ in range(S)]
        fl = 2.0 * M * N * K
        for name, hot, cold in (
            ("fwd   x @ W^T", lambda i: torch.mm(xs[0], ws[0].t(), out=outs[0]), lambda i: torch.mm(xs[i % S], ws[i % S].t(), out=outs[i % S])),
            ("dgrad dy @ W ", lambda i: torch.mm(dys[0], ws[0], out=dxs[0]), lambda i: torch.mm(dys[i % S], ws[i % S], out=dxs[i % S])),
        ):
            th, tc = timeit(hot, 48), timeit(cold, 48)
            print(f"N={N:5d} K={K:5d} {name}: hot {th:7.1f} us ({fl / th / 1e9:6.3f} PF/s)   cold {tc:7.1f} us ({fl / tc / 1e9:6.3f} PF/s)", flush=True)
        del xs, ws, dys, outs, dxs
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
