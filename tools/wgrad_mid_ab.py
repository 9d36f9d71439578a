"""Same-process timing of the [W_a ; W_b] weight gradient (dG^T x, M = 32768, N = 576, K = 1024): rwkv7_wgrad_mid_bf16 + reduction against the
32 library slabs + reduction.  Run under rocprofv3 --kernel-trace --stats for the per-kernel split."""
import sys, torch
sys.path.insert(0, "/root/repo")
from rwkvtts_amd import fused
DEV = "cuda"
M, N, K = 32768, 576, 1024
g = torch.Generator().manual_seed(4)
dy = torch.randn(M, N, generator=g).to(DEV, torch.bfloat16)
x = torch.randn(M, K, generator=g).to(DEV, torch.bfloat16)


def run(own, n):
    fused.MID_WGRAD = own
    for _ in range(3):
        fused.wgrad_splitk(dy, x, slabs=32)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fused.wgrad_splitk(dy, x, slabs=32)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for rnd in range(3):
    print("round", rnd, "library slabs + reduction %.1f us   own kernel + reduction %.1f us" % (run(False, 40), run(True, 40)))
