"""Per-segment cycle breakdown of one step of wkv7c_state_kernel, workgroup 0, both waves (needs
`python -m rwkvtts_amd.build --timing`; the stamps force the waits they measure)."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rwkvtts_amd import _lib, ops
from rwkvtts_amd.synthetic import make_wkv_inputs
B, T, H = 8, 4096, 16
dev = "cuda:0"
ins = make_wkv_inputs(B, T, H, 1, torch.bfloat16, dev)
dy = torch.randn(B, T, H, 64, device=dev).bfloat16()
y, tinv, sa, hs = ops.wkv7_chunk_forward(*ins)
lib = _lib.lib()
run = lambda: ops.wkv7_chunk_backward(*ins, dy, hs, sa, tinv)
run(); torch.cuda.synchronize()
lib.rwkv7_debug_cstate_timing(None, 1)
N = 5
ops.KERNEL_TIMERS = {}
for _ in range(N):
    run()
torch.cuda.synchronize()
ts = sorted(x.elapsed_time(e) for x, e in ops.KERNEL_TIMERS["wkv7c_state"])
ops.KERNEL_TIMERS = None
buf = (ctypes.c_longlong * 32)()
lib.rwkv7_debug_cstate_timing(buf, 0)
steps = T // 32
print(f"wkv7c_state with stamps: median {ts[len(ts) // 2] * 1e3:.1f} us; cycles per step (workgroup 0):")
names = {0: ["loop", "LDS reads issued", "wait + 12 MFMA + sum", "E -> LDS (fp32 + hi/lo planes)", "barrier"],
         1: ["loop + prefetch", "E record (q15) + stores", "decode M^T, N' -> LDS", "-", "barrier"]}
for wv in range(4):
    vals = [buf[wv * 8 + i] / N / steps for i in range(5)]
    nm = names[min(wv >> 1, 1)]
    print(f"wave {wv} ({'product' if wv < 2 else 'helper'}): total {sum(vals):6.0f} | " + "  ".join(f"{nm[i]}={vals[i]:5.0f}" for i in range(5)))
