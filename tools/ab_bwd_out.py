"""Same-process A/B of the two per-chunk gradient kernels (shipped wkv7c_bwd_out10 against the lab twin wkv7c_bwd_out9, tools/lab.py;
needs `python -m rwkvtts_amd.build --lab`): launches interleaved, HIP events,
and the six gradients compared bit for bit (same arithmetic, same MFMA order).   python tools/ab_bwd_out.py [iters]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rwkvtts_amd import _lib, ops
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import lab
from rwkvtts_amd.synthetic import make_wkv_inputs

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
lib = _lib.lib()
dev = "cuda:0"
P = lambda t: ctypes.c_void_p(t.data_ptr())
for (B, T, H) in ((8, 4096, 16), (4, 8192, 32), (2, 96, 3)):
    w, q, k, v, a, b = make_wkv_inputs(B, T, H, 1234, torch.bfloat16, dev)
    dy = torch.randn(B, T, H, 64, device=dev).bfloat16()
    y, tinv, sa, hs = ops.wkv7_chunk_forward(w, q, k, v, a, b)
    e_vk, z = ops.wkv7_chunk_bwd_seq(w, q, a, b, dy, tinv, want_z=True)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    outs = {}
    def run(gen, grads):
        if gen == 9:
            lab.bwd_out9(w, q, k, v, a, b, dy, hs, sa, z, e_vk, grads)
            return
        rc = lib.rwkv7_wkv_chunk_bwd_out_z_bf16(B, T, H, P(w), P(q), P(k), P(v), P(a), P(b), P(dy), P(hs), P(sa), P(z), P(e_vk), *[P(g) for g in grads], st)
        assert rc == 0, rc
    for gen in (9, 10):
        outs[gen] = [torch.full_like(w, float("nan")) for _ in range(6)]
        run(gen, outs[gen])
    torch.cuda.synchronize()
    same = [torch.equal(x, y_) for x, y_ in zip(outs[9], outs[10])]
    worst = max((x.float() - y_.float()).abs().max().item() for x, y_ in zip(outs[9], outs[10]))
    print(f"(B,T,H)=({B},{T},{H}): gradients identical {same}  max|d| {worst:.3e}  finite {all(torch.isfinite(g.float()).all().item() for g in outs[10])}", flush=True)
    ts = {9: [], 10: []}
    for _ in range(iters):
        for gen in (9, 10):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record(); run(gen, outs[gen]); e.record()
            ts[gen].append((s, e))
    torch.cuda.synchronize()
    med = {g: sorted(s.elapsed_time(e) for s, e in t)[len(t) // 2] * 1e3 for g, t in ts.items()}
    print(f"    bwd_out9 {med[9]:7.1f} us   bwd_out10 {med[10]:7.1f} us   ({100 * (med[10] / med[9] - 1):+.1f} %)", flush=True)
