"""The mix_lora projection G = x [W_a ; W_b]^T at N = 2 R = 576 (0.4B) is an awkward width for the library (measured in the step:
71 us forward = 0.54 PF/s, 61 us for the input gradient).  Does another width or a split do better?  Interleaved, rotating operands."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
dev = "cuda:0"
M, K = 32768, 1024
R = 4
xs = [(torch.randn(M, K, device=dev) * 0.5).bfloat16() for _ in range(R)]
dxs = [torch.randn(M, K, device=dev).bfloat16() for _ in range(R)]
def mk(N): return (torch.randn(N, K, device=dev) * 0.03).bfloat16()
W = {n: mk(n) for n in (64, 512, 576, 640, 768, 1024)}
dG = {n: [torch.randn(M, n, device=dev).bfloat16() for _ in range(R)] for n in (64, 512, 576, 640, 768)}
variants = {
    "fwd N=576": lambda i: torch.mm(xs[i], W[576].t()),
    "fwd N=640": lambda i: torch.mm(xs[i], W[640].t()),
    "fwd N=768": lambda i: torch.mm(xs[i], W[768].t()),
    "fwd N=1024": lambda i: torch.mm(xs[i], W[1024].t()),
    "fwd N=512 + N=64": lambda i: (torch.mm(xs[i], W[512].t()), torch.mm(xs[i], W[64].t())),
    "dgrad addmm N=576": lambda i: dxs[i].addmm_(dG[576][i], W[576]),
    "dgrad addmm N=640": lambda i: dxs[i].addmm_(dG[640][i], W[640]),
    "dgrad addmm N=768": lambda i: dxs[i].addmm_(dG[768][i], W[768]),
    "dgrad addmm 512 + 64": lambda i: (dxs[i].addmm_(dG[512][i], W[512]), dxs[i].addmm_(dG[64][i], W[64])),
}
times = {k: [] for k in variants}
for rep in range(6):
    for name, fn in variants.items():
        for i in range(2): fn(i)
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(8)]
        for j, (s, e) in enumerate(ev):
            s.record(); fn(j % R); e.record()
        torch.cuda.synchronize()
        if rep: times[name] += [s.elapsed_time(e) for s, e in ev]
for name in variants:
    ts = sorted(times[name]); print(f"{name:26s} {ts[len(ts)//2]*1e3:8.1f} us")
