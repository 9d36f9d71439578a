"""Same-box A/B of two builds of librwkv7_hip.so on the chunked WKV7 kernels (boxes of the pool differ by ~5 %, more than most kernel
changes are worth): both libraries are loaded into ONE process and their launches are interleaved, HIP events on the launch stream.

    python tools/ab_kernel.py tools/ab/librwkv7_hip_base.so rwkvtts_amd/lib/librwkv7_hip.so [iters]
Build the baseline with e.g.  git stash; python -m rwkvtts_amd.build --force; cp rwkvtts_amd/lib/librwkv7_hip.so tools/ab/librwkv7_hip_base.so; git stash pop."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rwkvtts_amd import ops
from rwkvtts_amd.synthetic import make_wkv_inputs

paths = sys.argv[1:3]
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 30
libs = [ctypes.CDLL(os.path.abspath(p)) for p in paths]
B, T, H = 8, 4096, 16
dev = "cuda:0"
w, q, k, v, a, b = make_wkv_inputs(B, T, H, 1234, torch.bfloat16, dev)
dy = torch.randn(B, T, H, 64, device=dev).bfloat16()
y, tinv, sa, hs = ops.wkv7_chunk_forward(w, q, k, v, a, b)
e_vk, z = ops.wkv7_chunk_bwd_seq(w, q, a, b, dy, tinv, want_z=True)
grads = [torch.empty_like(w) for _ in range(6)]
P = lambda t: ctypes.c_void_p(t.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
y2, sa2, hs2 = torch.empty_like(y), torch.empty_like(sa), torch.empty_like(hs)
tinv2 = torch.empty_like(tinv)
kernels = {
    "wkv7c_prep": lambda L: L.rwkv7_wkv_chunk_prep_bf16(B, T, H, P(w), P(a), P(b), P(tinv2), st),
    "wkv7c_fwd9": lambda L: L.rwkv7_wkv_chunk_fwd_bf16(B, T, H, P(w), P(q), P(k), P(v), P(a), P(b), P(tinv), P(y2), P(sa2), P(hs2), st),
    "wkv7c_bseq": lambda L: L.rwkv7_wkv_chunk_bseq_bf16(B, T, H, P(w), P(q), P(a), P(b), P(dy), P(tinv), P(e_vk), P(z), None, 0, st),
    "wkv7c_bwd_out9": lambda L: L.rwkv7_wkv_chunk_bwd_out_z_bf16(B, T, H, P(w), P(q), P(k), P(v), P(a), P(b), P(dy), P(hs), P(sa), P(z), P(e_vk),
                                                                 *[P(g) for g in grads], st),
}
for name, fn in kernels.items():
    ts = [[], []]
    for L in libs:
        for _ in range(3):
            assert fn(L) == 0
    torch.cuda.synchronize()
    for _ in range(iters):
        for i, L in enumerate(libs):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record(); fn(L); e.record()
            ts[i].append((s, e))
    torch.cuda.synchronize()
    med = [sorted(s.elapsed_time(e) for s, e in t)[len(t) // 2] * 1e3 for t in ts]
    print(f"{name:16s} A {med[0]:7.1f} us   B {med[1]:7.1f} us   ({med[1] - med[0]:+.1f} us, {100 * (med[1] / med[0] - 1):+.1f} %)", flush=True)
# results of the two libraries agree bit for bit?
res = []
for L in libs:
    y2.zero_(); sa2.zero_(); hs2.zero_(); e_vk.zero_(); z.zero_()
    kernels["wkv7c_fwd9"](L); kernels["wkv7c_bseq"](L)
    torch.cuda.synchronize()
    res.append([t.clone() for t in (y2, sa2, hs2, e_vk, z)])
print("fwd9 (y, sa, hs) and bseq (e_vk, z) identical:", [torch.equal(x, y_) for x, y_ in zip(*res)])
outs = []
for L in libs:
    kernels["wkv7c_bwd_out9"](L)
    torch.cuda.synchronize()
    outs.append([g.clone() for g in grads])
print("bwd_out9 gradients identical:", all(torch.equal(x, y_) for x, y_ in zip(*outs)))
