"""What streaming kernels reach on this box: library copy / add / sum and the package's own elementwise kernels."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rwkvtts_amd import fused

dev = "cuda:0"
n = 1 << 29   # 512 Mi bf16 elements = 1 GiB
a = torch.randn(n, device=dev).bfloat16()
b = torch.randn(n, device=dev).bfloat16()
c = torch.empty_like(a)


def t(f, bytes_, name, it=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(it):
        f()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / it
    print(f"{name:44s} {dt * 1e6:8.1f} us  {bytes_ / dt / 1e12:5.2f} TB/s", flush=True)


g = 2.0 * n
t(lambda: c.copy_(a), 2 * g, "copy_ (1 read + 1 write, bf16)")
t(lambda: torch.add(a, b, out=c), 3 * g, "add (2 reads + 1 write)")
t(lambda: a.float().sum() if False else a.sum(), g, "sum (1 read)")
af = a.view(torch.float32)
cf = c.view(torch.float32)
t(lambda: cf.copy_(af), 2 * g, "copy_ (same bytes as fp32)")
t(lambda: c.zero_(), g, "zero_ (1 write)")
x = a.view(8, 4096, -1)[:, :, :4096].contiguous()   # [8,4096,4096] like the channel-mix hidden
y = torch.empty_like(x)
t(lambda: fused.relu_sq(x), 2 * x.numel() * 2, "rwkv7_relusq_fwd (1 read + 1 write)")
