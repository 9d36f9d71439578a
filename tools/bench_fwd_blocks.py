"""Sweep of the workgroup count of the row-walking forward stages (tmix_prepare_fwd, tmix_post_fwd) at B=8,T=4096,D=1024."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rwkvtts_amd import fused
from tools.bench_wkv import timeit
B, T, D, H = 8, 4096, 1024, 16
dev = "cuda:0"
mk = lambda *s: torch.randn(*s, device=dev).bfloat16()
ins = [mk(B, T, D) for _ in range(6)]
kk, ka, gw, gb, rk = mk(D), mk(D), mk(D), mk(D), mk(H, 64)
y, r, k2, v2, g = [mk(B, T, D) for _ in range(5)]
ps = [mk(1, 1, D) for _ in range(6)]
ln = torch.nn.LayerNorm(D).to(dev).bfloat16()
with torch.no_grad():
    for rep in range(2):
        for nb in (2048, 4096, 8192, 16384, 32768):
            fused._FWD_BLOCKS = nb
            tp = timeit(lambda: fused.tmix_prepare(*ins, kk, ka, None, H, False), 50)[0]
            to = timeit(lambda: fused.tmix_post(y, r, k2, v2, g, gw, gb, rk, H, 64e-5), 50)[0]
            tm = timeit(lambda: fused.token_shift_mix6(y, None, *ps), 50)[0]
            tl = timeit(lambda: fused.add_layer_norm(y, r, ln), 50)[0]
            print(f"blocks {nb:6d}: prepare_fwd {tp * 1e3:6.1f} us  post_fwd {to * 1e3:6.1f} us  mix6_fwd {tm * 1e3:6.1f} us  add_ln_fwd {tl * 1e3:6.1f} us", flush=True)
