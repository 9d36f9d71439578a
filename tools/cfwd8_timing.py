"""Per-interval cycle breakdown of wkv7c_fwd8_kernel, workgroup 0, every wave (needs `python -m rwkvtts_amd.build --timing`).
Each interval: cycles of work, then cycles waiting at the barrier that ends it."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rwkvtts_amd import _lib, ops
from rwkvtts_amd.synthetic import make_wkv_inputs
B, T, H = 8, 4096, 16
ins = make_wkv_inputs(B, T, H, 1, torch.bfloat16, "cuda:0")
lib = _lib.lib()
run = lambda: ops.wkv7_chunk_forward(*ins)
run(); torch.cuda.synchronize()
lib.rwkv7_debug_cfwd8_timing(None, 1)
N = 5
for _ in range(N):
    run()
torch.cuda.synchronize()
buf = (ctypes.c_longlong * 96)()
lib.rwkv7_debug_cfwd8_timing(buf, 0)
chunks = T // 32 + 1
print("cycles per chunk (workgroup 0); w = work, b = barrier wait; intervals 1-4 and the epilogue (consumer) / 1-4 (producer):")
for wv in range(8):
    vals = [buf[wv * 12 + i] / N / chunks for i in range(12)]
    print(f"wave {wv} ({'consumer' if wv < 4 else 'producer'}): total {sum(vals):6.0f} | " + "  ".join(f"i{i // 2 + 1}{'w' if i % 2 == 0 else 'b'}={vals[i]:5.0f}" for i in range(10)))
