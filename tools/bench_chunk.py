"""Per-kernel HIP-event timing of the chunked WKV7 kernels (prep, fwd, pre, state, out) at B=8, T=4096, H=16 bf16."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rwkvtts_amd import ops  # noqa: E402
from rwkvtts_amd.synthetic import make_wkv_inputs  # noqa: E402

B, T, H = 8, 4096, 16
dev = "cuda:0"
w, q, k, v, a, b = make_wkv_inputs(B, T, H, 1234, torch.bfloat16, dev)
dy = torch.randn(B, T, H, 64, device=dev).bfloat16()
y = torch.empty_like(v)
s = torch.empty(B, H, T // 16, 64, 64, device=dev)
sa = torch.empty(B, T, H, 64, device=dev)
torch.ops.wind_backstepping.forward(w, q, k, v, a, b, y, s, sa)
for _ in range(3):
    ops.wkv7_chunk_forward(w, q, k, v, a, b)
    ops.wkv7_chunk_backward(w, q, k, v, a, b, dy, s, sa)
torch.cuda.synchronize()
ops.KERNEL_TIMERS = {}
for _ in range(10):
    ops.wkv7_chunk_forward(w, q, k, v, a, b)
    ops.wkv7_chunk_backward(w, q, k, v, a, b, dy, s, sa)
torch.cuda.synchronize()
tot = 0.0
for name, evs in ops.KERNEL_TIMERS.items():
    ts = sorted(x.elapsed_time(e) for x, e in evs)
    tot += ts[len(ts) // 2]
    print(f"{name:16s} median {ts[len(ts) // 2] * 1e3:8.1f} us   best {ts[0] * 1e3:8.1f} us")
print(f"sum of medians {tot * 1e3:.1f} us")
