"""Timing of the fused elementwise stages at B=8,T=4096,D=1024 bf16 (HIP events)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rwkvtts_amd import fused
from tools.bench_wkv import timeit
B, T, D, H = 8, 4096, 1024, 16
dev = "cuda:0"
mk = lambda *s: torch.randn(*s, device=dev).bfloat16()
x = mk(B, T, D).requires_grad_(True)
ps = [mk(1, 1, D).requires_grad_(True) for _ in range(6)]
outs = fused.token_shift_mix6(x, None, *ps)
gs = [mk(B, T, D) for _ in range(6)]
def mixb():
    torch.autograd.grad(outs, [x] + ps, gs, retain_graph=True)
print("mix6 fwd  ms", timeit(lambda: fused.token_shift_mix6(x, None, *ps), 10)[0])
print("mix6 bwd  ms (incl. stack of 6 grads + partial sum)", timeit(mixb, 10)[0])
for nr, nblk in ((1, 1024), (2, 1024), (4, 512), (4, 1024), (4, 2048), (8, 1024), (16, 1024), (16, 2048)):
    fused._MIX_BWD_ROWS, fused._MIX_BWD_BLOCKS = nr, nblk
    print("  MIX_BWD_ROWS", nr, "BLOCKS", nblk, timeit(mixb, 10)[0])
fused._MIX_BWD_ROWS, fused._MIX_BWD_BLOCKS = 4, 1024
ins = [mk(B, T, D).requires_grad_(True) for _ in range(6)]
kk, ka = mk(D).requires_grad_(True), mk(D).requires_grad_(True)
po = fused.tmix_prepare(*ins, kk, ka, None, H, False)
g5 = [mk(B, T, D) for _ in range(5)]
print("prepare fwd ms", timeit(lambda: fused.tmix_prepare(*ins, kk, ka, None, H, False), 10)[0])
print("prepare bwd ms", timeit(lambda: torch.autograd.grad(po, ins + [kk, ka], g5, retain_graph=True), 10)[0])
