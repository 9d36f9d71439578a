"""Per-phase cycle breakdown of the chunked forward (needs `python -m rwkvtts_amd.build --timing`)."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rwkvtts_amd import _lib, ops
from rwkvtts_amd.synthetic import make_wkv_inputs
B, T, H = 8, 4096, 16
ins = make_wkv_inputs(B, T, H, 1, torch.bfloat16, "cuda:0")
lib = _lib.lib()
ops.wkv7_chunk_forward(*ins); torch.cuda.synchronize()
lib.rwkv7_debug_chunk_timing(None, 1)
ops.wkv7_chunk_forward(*ins); torch.cuda.synchronize()
buf = (ctypes.c_longlong * 64)()
lib.rwkv7_debug_chunk_timing(buf, 0)
names = ["loop/ph7", "ph1 lw+cumsum", "ph2 scale+planes", "bar", "ph3 A-mats/T", "bar", "ph4 R", "bar", "ph5 U", "bar", "ph6 Y|S", "bar"]
nc = T // 32
for w in range(4):
    tot = sum(buf[w * 16 + i] for i in range(12))
    print(f"wave {w}: total {tot / nc:8.0f} cyc/chunk | " + " ".join(f"{names[i]}={buf[w * 16 + i] / nc:6.0f}" for i in range(12)))
    print("        ph2 detail: G+seg=%d exps+mul+split=%d b128 stores=%d (rest of ph2 = b16 transposed stores + v)" % tuple(buf[w * 16 + i] / nc for i in (12, 13, 14)))
