"""Chunked bf16 forward: the shipped 8-wave producer / consumer kernel against the lab twin (4-wave kernel; tools/lab.py, needs
`python -m rwkvtts_amd.build --lab`), interleaved."""
import sys, os
sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import lab
from rwkvtts_amd import ops
from rwkvtts_amd.synthetic import make_wkv_inputs
B, T, H = 8, 4096, 16
w, q, k, v, a, b = make_wkv_inputs(B, T, H, 1234, torch.bfloat16, "cuda:0")
tinv = ops.wkv7_chunk_prep(w, a, b)
for waves in (4, 9, 4, 9):
    run = (lambda: lab.chunk_forward4(w, q, k, v, a, b, tinv)) if waves == 4 else (lambda: ops.wkv7_chunk_forward(w, q, k, v, a, b))
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    ev = []
    for _ in range(10):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); run(); e.record()
        ev.append((s, e))
    torch.cuda.synchronize()
    ts = sorted(x.elapsed_time(e) for x, e in ev)
    print(f"waves {waves}: forward{' (incl. prep)' if waves == 9 else ''} median {ts[len(ts)//2]*1e3:.1f} us best {ts[0]*1e3:.1f} us", flush=True)
