"""The shipped weight-gradient form (fused.wgrad_splitk: bmm over S row slabs with fp32 partials + rwkv7_sum_slabs_bf16) for S = 2 ... 32 at the
three big shapes of the 0.4B step, interleaved."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rwkvtts_amd import fused
dev = "cuda:0"
M = 32768
for name, N, K in (("1024 x 1024", 1024, 1024), ("key 4096 x 1024", 4096, 1024), ("value 1024 x 4096", 1024, 4096)):
    xs = [(torch.randn(M, K, device=dev) * 0.5).bfloat16() for _ in range(3)]
    dys = [torch.randn(M, N, device=dev).bfloat16() for _ in range(3)]
    out = torch.empty(N, K, device=dev, dtype=torch.bfloat16)
    res = {}
    for rep in range(4):
        for S in (2, 4, 8, 16, 32):
            for i in range(2): fused.wgrad_splitk(dys[i], xs[i], out=out, slabs=S)
            ev = []
            for j in range(6):
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record(); fused.wgrad_splitk(dys[j % 3], xs[j % 3], out=out, slabs=S); e.record(); ev.append((s, e))
            torch.cuda.synchronize()
            if rep: res.setdefault(S, []).extend(a.elapsed_time(b) for a, b in ev)
    print(f"{name:18s} " + "  ".join(f"S={S}: {sorted(v)[len(v)//2]*1e3:6.1f} us" for S, v in res.items()), flush=True)
