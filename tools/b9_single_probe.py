"""Test infrastructure (not collected): the operand-by-operand precision probe of wkv7c_bwd_out9.

For the library named by RWKV7_HIP_SO (a build of csrc/lab/wkv7_chunk_bwd9.hip (lab build) with -DWKV7C_B9_SINGLE=<mask>: the masked operands enter their
products as ONE bf16 plane) it prints, per gradient, the worst error in units of the parity bar of tests/test_chunk_gpu.py (2 bf16 ulp with
the floor of _assert_bf16_close): < 1 passes.  Shapes and seeds: the parametrisations of test_chunked_forward_plus_backward_vs_oracle
plus two more seeds of the largest, and three (batch, head) slices of BASELINE configs[1].

    RWKV7_HIP_SO=tools/ab/lib_b9_<mask>.so python tools/b9_single_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rwkvtts_amd import ops
from rwkvtts_amd.synthetic import make_wkv_inputs
from oracle import c_oracle as co

co.build()
DEV = "cuda:0"
NAMES = ["dw", "dq", "dk", "dv", "da", "db"]


def margin(got, want, ulps=2.0):
    got, want = got.float().cpu(), want.float()
    floor = want.abs().mean().item() * 0.25 + 1e-6
    tol = ulps * 2.0 ** -7 * torch.clamp(want.abs(), min=floor)
    r = (got - want).abs() / tol
    return r.max().item(), (r > 1).float().mean().item()


worst = {n: 0.0 for n in NAMES}
fails = {n: 0.0 for n in NAMES}
for (B, T, H, seed) in [(1, 32, 1, 0), (2, 64, 3, 1), (2, 512, 12, 2), (2, 512, 12, 3), (2, 512, 12, 4), (1, 2048, 4, 5)]:
    ins = make_wkv_inputs(B, T, H, seed, torch.bfloat16)
    dy = (torch.randn(B, T, H, 64, generator=torch.Generator().manual_seed(seed + 100))).bfloat16()
    y_o, s_o, sa_o = co.wkv7_fwd(*ins)
    g_o = co.wkv7_bwd(*ins, dy, s_o, sa_o)
    d = [t.to(DEV) for t in ins]
    y, tinv, sa, hs = ops.wkv7_chunk_forward(*d)
    grads = ops.wkv7_chunk_backward(*d, dy.to(DEV), hs, sa, tinv)
    torch.cuda.synchronize()
    for n, g, go in zip(NAMES, grads, g_o):
        m, f = margin(g, go)
        worst[n] = max(worst[n], m)
        fails[n] = max(fails[n], f)
B, T, H = 8, 4096, 16
ins = make_wkv_inputs(B, T, H, 1234, torch.bfloat16)
d = [t.to(DEV) for t in ins]
dy1 = torch.randn(B, T, H, 64, generator=torch.Generator().manual_seed(99)).bfloat16()
y, tinv, sa, hs = ops.wkv7_chunk_forward(*d)
g1 = ops.wkv7_chunk_backward(*d, dy1.to(DEV), hs, sa, tinv)
torch.cuda.synchronize()
for (bi, hi) in ((0, 0), (5, 11), (7, 15)):
    sl = [t[bi:bi + 1, :, hi:hi + 1].contiguous() for t in ins]
    y_o, s_o, sa_o = co.wkv7_fwd(*sl)
    g_o = co.wkv7_bwd(*sl, dy1[bi:bi + 1, :, hi:hi + 1].contiguous(), s_o, sa_o)
    for n, ga, go in zip(NAMES, g1, g_o):
        m, f = margin(ga[bi:bi + 1, :, hi:hi + 1], go)
        worst[n] = max(worst[n], m)
        fails[n] = max(fails[n], f)
tag = os.path.basename(os.environ.get("RWKV7_HIP_SO", "default"))
print(f"{tag:24s} " + "  ".join(f"{n} {worst[n]:5.2f}" for n in NAMES) + f"   worst {max(worst.values()):5.2f}  max fail frac {max(fails.values()):.1e}", flush=True)
