"""Weight-gradient GEMMs dW = dy^T x (contraction over M = 32768 rows; 16-64 output tiles) as ONE library call under different
hipBLASLt workspace sizes (env HIPBLASLT_WORKSPACE_SIZE / CUBLASLT_WORKSPACE_SIZE in KiB, read at start-up) against the shipped
form (bmm over row slabs with fp32 partials + rwkv7_sum_slabs_bf16).   python tools/wgrad_workspace_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rwkvtts_amd import fused
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)


def timed(fn, n=30):
    for _ in range(4):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for s, e in ev:
        s.record(); fn(); e.record()
    torch.cuda.synchronize()
    ts = sorted(s.elapsed_time(e) * 1e3 for s, e in ev)
    return ts[len(ts) // 2]


print("workspace env:", {k: os.environ.get(k) for k in ("HIPBLASLT_WORKSPACE_SIZE", "CUBLASLT_WORKSPACE_SIZE", "TORCH_BLAS_PREFER_HIPBLASLT")})
M = 32768
for N, K in ((1024, 1024), (4096, 1024), (1024, 4096)):
    dy = (torch.randn(M, N, device=dev, generator=g) * 0.1).bfloat16()
    x = (torch.randn(M, K, device=dev, generator=g) * 0.5).bfloat16()
    out = torch.empty(N, K, device=dev, dtype=torch.bfloat16)
    fl = 2.0 * M * N * K
    t0 = timed(lambda: fused.wgrad_splitk(dy, x, out=out))
    t1 = timed(lambda: torch.mm(dy.t(), x, out=out))
    print(f"dW [{N}, {K}] over {M} rows: slabs + sum {t0:7.1f} us ({fl / t0 / 1e6:5.0f} TF/s)   one call {t1:7.1f} us ({fl / t1 / 1e6:5.0f} TF/s)", flush=True)
