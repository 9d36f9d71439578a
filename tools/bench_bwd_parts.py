"""Per-kernel times of the chunked WKV7 backward (bseq / bwd_out) at several grid sizes: does the sequential kernel's
step time depend on how many workgroups run beside it (shared HBM / fabric) or not (per-CU latency chain)?"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rwkvtts_amd import ops
from rwkvtts_amd.synthetic import make_wkv_inputs
dev = "cuda:0"
for B, T, H in ((8, 4096, 16), (8, 4096, 8), (4, 4096, 8), (2, 4096, 8), (8, 2048, 16)):
    ins = make_wkv_inputs(B, T, H, 1, torch.bfloat16, dev)
    dy = torch.randn(B, T, H, 64, device=dev).bfloat16()
    y, tinv, sa, hs = ops.wkv7_chunk_forward(*ins)
    run = lambda: ops.wkv7_chunk_backward(*ins, dy, hs, sa, tinv)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    ops.KERNEL_TIMERS = {}
    for _ in range(10):
        run()
    torch.cuda.synchronize()
    med = {k: sorted(s.elapsed_time(e) for s, e in v)[len(v) // 2] * 1e3 for k, v in ops.KERNEL_TIMERS.items()}
    ops.KERNEL_TIMERS = None
    nc = T // 32
    print(f"B={B} T={T} H={H}: " + "  ".join(f"{k} {v:6.1f} us" for k, v in med.items()) + f"  | bseq {med['wkv7c_bseq'] / nc * 1e3:6.0f} ns/chunk, {B * H * 2} workgroups", flush=True)
