"""Timing of the low-rank branch products at BASELINE configs[1] sizes (M = 8*4096 rows, D = 1024):
hand-written skinny kernels (csrc/lora.hip) against the BLAS calls they replace.
    python tools/bench_lora.py
"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rwkvtts_amd import _lib  # noqa: E402
from bench_wkv import timeit  # noqa: E402

M, D = 8 * 4096, 1024
dev = "cuda:0"
P = lambda t: ctypes.c_void_p(t.data_ptr())
for R, act in ((32, 0), (64, 1), (64, 0), (128, 2)):
    x = torch.randn(M, D, device=dev).bfloat16()
    w1 = (torch.randn(R, D, device=dev) * 0.05).bfloat16()
    w2 = (torch.randn(D, R, device=dev) * 0.05).bfloat16()
    w2t = w2.t().contiguous()
    a = torch.empty(M, R, device=dev, dtype=torch.bfloat16)
    dy = torch.empty_like(a)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    L = _lib.lib()
    t_down = timeit(lambda: L.rwkv7_lora_down_bf16(ctypes.c_long(M), D, R, act, P(x), P(w1), P(a), st), 20)[0]
    t_dg = timeit(lambda: L.rwkv7_lora_dgrad_up_bf16(ctypes.c_long(M), D, R, act, P(x), P(w2t), P(a), P(dy), st), 20)[0]
    t_bl = timeit(lambda: torch.mm(x, w1.t()), 20)[0]
    t_bl2 = timeit(lambda: torch.mm(x, w2), 20)[0]
    t_up = timeit(lambda: torch.mm(a, w2.t()), 20)[0]
    t_dd = timeit(lambda: torch.mm(a, w1), 20)[0]
    t_wg = timeit(lambda: torch.mm(x.t(), a), 20)[0]
    gb = M * D * 2 / 1e9
    print(f"R={R:3d} act={act}: down {t_down*1e3:7.1f} us ({gb/t_down*1e3:6.0f} GB/s)  BLAS x@w1^T {t_bl*1e3:7.1f} us | "
          f"dgrad_up {t_dg*1e3:7.1f} us  BLAS dz@w2 {t_bl2*1e3:7.1f} us | BLAS up a@w2^T {t_up*1e3:6.1f}  a@w1 {t_dd*1e3:6.1f}  "
          f"wgrad x^T@a {t_wg*1e3:6.1f} us")
