"""Kernel time of the fused token draws (csrc/sampling.hip) against the torch chains they replace."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rwkvtts_amd.sampling import RowSampler, ras_step
from rwkvtts_amd.spark_llm import sample_next
from rwkvtts_amd.cosy_llm import ras_sampling_device

DEV = "cuda:0"


def timeit(fn, n=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        for _ in range(10):
            fn()
    g.replay()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n // 10):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / (n // 10 * 10) * 1e6


step = torch.zeros(1, dtype=torch.long, device=DEV)
for name, rows, sizes, allow, kw in [
        ("spark B=32 V=8193 greedy", 32, [8193], None, dict()),
        ("spark B=32 V=8193 k=50 p=.95", 32, [8193], None, dict(do_sample=True, top_k=50, top_p=0.95, temperature=0.8)),
        ("spark B=32 V=8193 k=10 p=.95", 32, [8193], None, dict(do_sample=True, top_k=10, top_p=0.95, temperature=0.8)),
        ("spark B=32 V=8193 multinomial", 32, [8193], None, dict(do_sample=True)),
        ("xy 8 x 8 segments k=50", 8, [66661] + [1025] * 7, [(65536, 66561)] + [(0, 1025)] * 7, dict(do_sample=True, top_k=50, top_p=0.95, temperature=0.8)),
        ("xy 8 x 8 segments greedy", 8, [66661] + [1025] * 7, [(65536, 66561)] + [(0, 1025)] * 7, dict())]:
    lg = torch.randn(rows, sum(sizes), device=DEV)
    smp = RowSampler(lg.device, sizes, allow=allow, **kw)
    out = torch.empty(rows, len(sizes), dtype=torch.long, device=DEV)
    us = timeit(lambda: smp(lg, step, out))
    segs = torch.split(lg, sizes, 1)
    def chain():
        for i, sg in enumerate(segs):
            x = sg
            if allow is not None and i == 0:
                x = sg[:, allow[0][0]:allow[0][1]]
            sample_next(x, kw.get("do_sample", False), kw.get("top_k", 0), kw.get("top_p", 1.0), kw.get("temperature", 1.0))
    us_t = timeit(chain)
    print(f"{name:34s}: fused {us:7.1f} us   torch chain {us_t:8.1f} us (in a replayed graph)", flush=True)
V, eos = 6562, 6561
lg = torch.randn(V, device=DEV)
tok, recent, ptr, si = torch.zeros(1, dtype=torch.long, device=DEV), torch.full((10,), -1, dtype=torch.long, device=DEV), torch.zeros(1, dtype=torch.long, device=DEV), torch.tensor(0, device=DEV)
us = timeit(lambda: ras_step(lg, tok, recent, ptr, si, 5, eos))
ig = torch.tensor(True, device=DEV)
us_t = timeit(lambda: ras_sampling_device(lg.log_softmax(0), recent, ig, eos))
print(f"{'cosy ras V=6562 k=25':34s}: fused {us:7.1f} us   torch chain {us_t:8.1f} us (draw only, without the ring update)")
