"""Per-shape timing of every BLAS product of one RWKV-7 0.4B layer at B*T = 32768 rows (forward, dgrad, wgrad in
the layouts autograd's F.linear backward uses).  Attribution aid for rocprofv3's Cijk_* kernel names."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_wkv import timeit  # noqa: E402

M = 8 * 4096
dev = "cuda:0"
shapes = [("proj D->D (x4)", 1024, 1024, 4), ("ffn key D->4D", 1024, 4096, 1), ("ffn value 4D->D", 4096, 1024, 1),
          ("lora down R=64 (x2)", 1024, 64, 2), ("lora down R=32", 1024, 32, 1), ("lora down R=128", 1024, 128, 1),
          ("lora up R=64 (x2)", 64, 1024, 2), ("lora up R=32", 32, 1024, 1), ("lora up R=128", 128, 1024, 1)]
tot = 0.0
for name, K, N, cnt in shapes:
    x = torch.randn(M, K, device=dev).bfloat16()
    w = (torch.randn(N, K, device=dev) * 0.02).bfloat16()
    dy = torch.randn(M, N, device=dev).bfloat16()
    f = timeit(lambda: F.linear(x, w), 10)[0]
    dg = timeit(lambda: dy @ w, 10)[0]
    wg = timeit(lambda: dy.t() @ x, 10)[0]
    fl = 2.0 * M * K * N
    byt = 2.0 * (M * K + M * N)
    tot += cnt * (f + dg + wg)
    print(f"{name:22s} fwd {f*1e3:7.1f} us  dgrad {dg*1e3:7.1f} us  wgrad {wg*1e3:7.1f} us   "
          f"[{fl/1e9:6.1f} GF, {byt/1e6:6.1f} MB -> fwd {fl/f/1e9:7.0f} TF/s.. {byt/f/1e6:6.0f} GB/s; wgrad {byt/wg/1e6:6.0f} GB/s]")
print(f"sum per layer {tot:.3f} ms -> x24 = {tot*24:.1f} ms/step")
