"""GEMM lab: several builds of the own GEMM (tools/gemm_lab/lib*.so, entry lab_gemm) and the library, operands rotating through R
buffer sets (cold), the variants INTERLEAVED round-robin (a variant measured alone right after another kernel reads 10 % off)."""
import ctypes, glob, os, sys
import torch
M, N, K = (int(x) for x in sys.argv[1:4]) if len(sys.argv) >= 4 else (32768, 4096, 1024)
R = 6
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
As = [(torch.randn(M, K, device=dev, generator=g) * 0.5).bfloat16() for _ in range(R)]
W = (torch.randn(N, K, device=dev, generator=g) * 0.05).bfloat16()
Cs = [torch.empty(M, N, device=dev, dtype=torch.bfloat16) for _ in range(R)]
P = lambda t: ctypes.c_void_p(t.data_ptr())
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
flops = 2.0 * M * N * K
torch.mm(As[0], W.t(), out=Cs[0]); ref = Cs[0].clone()
variants = [("library", lambda i: torch.mm(As[i], W.t(), out=Cs[i]))]
for so in sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib*.so"))):
    L = ctypes.CDLL(so)
    def fn(i, L=L):
        rc = L.lab_gemm(M, N, K, P(As[i]), P(W), P(Cs[i]), 0, st()); assert rc == 0, rc
    Cs[0].zero_(); fn(0); torch.cuda.synchronize()
    err = (Cs[0].float() - ref.float()).abs().max().item()
    variants.append((f"{os.path.basename(so)} (max|d| {err:.1e})", fn))
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
    print(f"{name:40s} {t*1e3:8.1f} us {flops/t/1e9:8.1f} TFLOP/s")
