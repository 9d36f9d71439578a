"""Per-interval cycle breakdown of wkv7c_fwd9_kernel, workgroup 0, every wave (needs `python -m rwkvtts_amd.build --timing`).
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
lib.rwkv7_debug_cfwd9_timing(None, 1)
N = 5
for _ in range(N):
    run()
torch.cuda.synchronize()
buf = (ctypes.c_longlong * 64)()
lib.rwkv7_debug_cfwd9_timing(buf, 0)
chunks = T // 32 + 1
print("cycles per chunk (workgroup 0): interval a = products | epilogue + LDS drain | barrier wait;  interval b = products | update + state planes | rest (record, next chunk's matrices) | barrier wait")
for wv in range(8):
    v = [buf[wv * 8 + i] / N / chunks for i in range(8)]
    print(f"wave {wv} ({'consumer' if wv < 4 else 'producer'}): total {sum(v):6.0f} | a {v[0]:5.0f} | {v[1]:5.0f} | {v[2]:5.0f}    b {v[3]:5.0f} | {v[4]:5.0f} | {v[5]:5.0f} | {v[6]:5.0f}")
