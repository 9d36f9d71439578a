"""Per-phase cycle breakdown of wkv7c_bwd_out10_kernel, workgroup 0, every wave (needs `python -m rwkvtts_amd.build --timing`)."""
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
lib.rwkv7_debug_cbwd10_timing(None, 1)
N = 5
for _ in range(N):
    run()
torch.cuda.synchronize()
buf = (ctypes.c_longlong * 128)()
lib.rwkv7_debug_cbwd10_timing(buf, 0)
names = ["rawread", "I0 prologue+H0", "bar", "I1 first", "I1 second", "bar", "B", "bar", "stage(+mat loads)", "bar", "epilogue", "vmwait+bar+stores", "dterm+dma"]
chunks = 64   # B*H*T/32 = 16384 chunks -> 64 per workgroup
print("cycles per chunk (workgroup 0):")
for wv in range(8):
    vals = [buf[wv * 16 + i] / N / chunks for i in range(13)]
    print(f"wave {wv}: total {sum(vals):7.0f} | " + " ".join(f"{names[i]}={vals[i]:5.0f}" for i in range(13)))
