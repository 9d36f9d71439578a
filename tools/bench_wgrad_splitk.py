"""Experiment: weight-gradient products dy^T @ x (reduction over M = 32768 rows) as one BLAS call vs a batched
split over M (bmm of S slabs + sum)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_wkv import timeit  # noqa: E402

M = 8 * 4096
dev = "cuda:0"
for name, K, N in (("proj 1024x1024", 1024, 1024), ("ffn key 4096x1024", 1024, 4096), ("ffn val 1024x4096", 4096, 1024),
                   ("lora 64x1024", 1024, 64), ("lora 1024x64", 64, 1024), ("lora 128x1024", 1024, 128)):
    x = torch.randn(M, K, device=dev).bfloat16()
    dy = torch.randn(M, N, device=dev).bfloat16()
    base = timeit(lambda: dy.t() @ x, 10)[0]
    ref = (dy.t() @ x).float()
    line = f"{name:18s} mm {base*1e3:7.1f} us |"
    for S in (4, 8, 16, 32, 64):
        def f():
            p = torch.bmm(dy.view(S, M // S, N).transpose(1, 2), x.view(S, M // S, K))
            return p.sum(0)
        try:
            t = timeit(f, 10)[0]
            err = (f().float() - ref).abs().max().item() / ref.abs().max().item()
            line += f" S={S}: {t*1e3:6.1f} us (err {err:.1e})"
        except Exception as e:  # noqa
            line += f" S={S}: fail {type(e).__name__}"
    try:
        t = timeit(lambda: torch.mm(dy.t(), x, out_dtype=torch.float32), 10)[0]
        line += f" | mm fp32-out {t*1e3:6.1f}"
    except Exception as e:  # noqa
        line += f" | mm out_dtype: {type(e).__name__}"
    print(line)
