"""GPU time of one low-rank branch (backbone.LoRA: Linear(D, r) -> act -> Linear(r, D) + bias) forward + backward at the training shape,
queue kept full (no sync inside the loop), against the HBM time of its streams.   python tools/bench_lora.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rwkvtts_amd import backbone
dev = 'cuda:0'
torch.manual_seed(0)
M, D = 32768, 1024
for rank, act in ((64, 'tanh'), (64, None), (32, None), (128, 'sigmoid')):
    m = backbone.LoRA(D, D, rank, act, bias=True).to(dev, torch.bfloat16)
    xs = [torch.randn(8, 4096, D, device=dev, dtype=torch.bfloat16, requires_grad=True) for _ in range(6)]
    dys = [torch.randn(8, 4096, D, device=dev, dtype=torch.bfloat16) for _ in range(6)]
    for i in range(6):
        m(xs[i]).backward(dys[i])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 60
    e0.record()
    for j in range(n):
        m(xs[j % 6]).backward(dys[j % 6])
    e1.record(); torch.cuda.synchronize()
    tot = e0.elapsed_time(e1) / n * 1e3
    e0.record()
    with torch.no_grad():
        for j in range(n):
            m(xs[j % 6])
    e1.record(); torch.cuda.synchronize()
    f = e0.elapsed_time(e1) / n * 1e3
    print(f"rank {rank:3d} act {act}: fwd {f:6.1f} us  fwd+bwd {tot:6.1f} us   (one [M,D] bf16 stream = {M*D*2/4.5e12*1e6:.0f} us at 4.5 TB/s: fwd 2 streams, bwd 3)")
