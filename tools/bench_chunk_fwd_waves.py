import sys, os
sys.path.insert(0, os.getcwd())
import torch
from rwkvtts_amd import ops, _lib
from rwkvtts_amd.synthetic import make_wkv_inputs
B, T, H = 8, 4096, 16
w, q, k, v, a, b = make_wkv_inputs(B, T, H, 1234, torch.bfloat16, "cuda:0")
lib = _lib.lib()
for waves in (4, 9, 4, 9):
    for _ in range(3):
        ops.wkv7_chunk_forward(w, q, k, v, a, b, waves=waves)
    torch.cuda.synchronize()
    ops.KERNEL_TIMERS = {}
    for _ in range(10):
        ops.wkv7_chunk_forward(w, q, k, v, a, b, waves=waves)
    torch.cuda.synchronize()
    ts = sorted(x.elapsed_time(e) for x, e in ops.KERNEL_TIMERS["wkv7c_fwd"])
    ops.KERNEL_TIMERS = None
    print(f"waves {waves}: wkv7c_fwd median {ts[len(ts)//2]*1e3:.1f} us best {ts[0]*1e3:.1f} us", flush=True)
