"""GEMM lab: epilogue 3 (x 2 sqrt(aux)) of several builds (lib*.so with lab_gemm_aux), interleaved, rotating operands."""
import ctypes, glob, os, sys
import torch
M, N, K = (int(x) for x in sys.argv[1:4]) if len(sys.argv) >= 4 else (32768, 4096, 1024)
EPI = int(sys.argv[4]) if len(sys.argv) >= 5 else 3
R = 4
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
As = [(torch.randn(M, K, device=dev, generator=g) * 0.5).bfloat16() for _ in range(R)]
W = (torch.randn(N, K, device=dev, generator=g) * 0.05).bfloat16()
Xs = [(torch.randn(M, N, device=dev, generator=g).abs()).bfloat16() for _ in range(R)]
Cs = [torch.empty(M, N, device=dev, dtype=torch.bfloat16) for _ in range(R)]
P = lambda t: ctypes.c_void_p(t.data_ptr())
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
variants, outs = [], []
for so in sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib*.so"))):
    L = ctypes.CDLL(so)
    if not hasattr(L, "lab_gemm_aux"): continue
    def fn(i, L=L):
        rc = L.lab_gemm_aux(M, N, K, P(As[i]), P(W), P(Cs[i]), P(Xs[i]), EPI, st()); assert rc == 0, rc
    fn(0); torch.cuda.synchronize(); outs.append(Cs[0].clone())
    variants.append((os.path.basename(so), fn))
print("identical outputs:", all(torch.equal(o, outs[0]) for o in outs))
times = {n: [] for n, _ in variants}
for rep in range(8):
    for name, fn in variants:
        for i in range(3): fn(i)
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
        for j, (s, e) in enumerate(ev):
            s.record(); fn(j % R); e.record()
        torch.cuda.synchronize()
        if rep: times[name] += [s.elapsed_time(e) for s, e in ev]
for name, _ in variants:
    ts = sorted(times[name]); t = ts[len(ts) // 2]
    print(f"{name:30s} {t*1e3:8.1f} us")
