"""Per-phase cycle breakdown of wkv7c_bwd_out_kernel, workgroup 0 (needs `python -m rwkvtts_amd.build --timing`)."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rwkvtts_amd import _lib, ops
from rwkvtts_amd.synthetic import make_wkv_inputs
B, T, H = 8, 4096, 16
dev = "cuda:0"
w, q, k, v, a, b = make_wkv_inputs(B, T, H, 1, torch.bfloat16, dev)
dy = torch.randn(B, T, H, 64, device=dev).bfloat16()
y = torch.empty_like(v); s = torch.empty(B, H, T // 16, 64, 64, device=dev); sa = torch.empty(B, T, H, 64, device=dev)
torch.ops.wind_backstepping.forward(w, q, k, v, a, b, y, s, sa)
lib = _lib.lib()
ops.wkv7_chunk_backward(w, q, k, v, a, b, dy, s, sa); torch.cuda.synchronize()
lib.rwkv7_debug_cbwd_timing(None, 1)
N = 5
for _ in range(N):
    ops.wkv7_chunk_backward(w, q, k, v, a, b, dy, s, sa)
torch.cuda.synchronize()
buf = (ctypes.c_longlong * 128)()
lib.rwkv7_debug_cbwd_timing(buf, 0)
names = ["loads", "prologue", "bar+EG", "barA", "phA", "phB", "phC", "phD", "phE1", "phF1", "phE2", "phF2+epi"]
names = ["issue loads", "cumsum+planes", "bar, EG planes", "bar | A", "A->B", "B->C", "C->D", "D->E1", "E1->F1", "F1->E2", "E2->F2", "F2+bar -> end"]
for wv in range(4):
    vals = [buf[wv * 32 + i] / N for i in range(12)]
    print(f"wave {wv}: total {sum(vals):8.0f} cyc | " + " ".join(f"[{i}]{names[i]}={vals[i]:6.0f}" for i in range(12)))
