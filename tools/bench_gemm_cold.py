"""Library GEMM (hipBLASLt through torch) against the own MFMA GEMM (rwkv7_gemm_nt_bf16: csrc/gemm_nt4.hip since round 4; the first generation is csrc/lab/gemm_relusq.hip), HOT (same operands every
call: they sit in the 256 MB infinity cache) and COLD (operands rotate through R buffer sets larger than the cache together -- what the
training step sees).  HIP events, same process.

    python tools/bench_gemm_cold.py [M N K] [R]      default 32768 1024 1024, R = 8"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rwkvtts_amd import _lib

M, N, K = (int(x) for x in sys.argv[1:4]) if len(sys.argv) >= 4 else (32768, 1024, 1024)
R = int(sys.argv[4]) if len(sys.argv) >= 5 else 8
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
As = [(torch.randn(M, K, device=dev, generator=g) * 0.5).bfloat16() for _ in range(R)]
W = (torch.randn(N, K, device=dev, generator=g) * 0.05).bfloat16()
Cs = [torch.empty(M, N, device=dev, dtype=torch.bfloat16) for _ in range(R)]
lib = _lib.lib()
P = lambda t: ctypes.c_void_p(t.data_ptr())
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def own(i):
    rc = lib.rwkv7_gemm_nt_bf16(M, N, K, P(As[i]), P(W), P(Cs[i]), 0, st())
    assert rc == 0, rc


def library(i):
    torch.nn.functional.linear(As[i], W, out=None) if False else torch.mm(As[i], W.t(), out=Cs[i])


def timeit(fn, rotate, n=48):
    for i in range(R):
        fn(i if rotate else 0)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for j, (s, e) in enumerate(ev):
        s.record(); fn(j % R if rotate else 0); e.record()
    torch.cuda.synchronize()
    ts = sorted(s.elapsed_time(e) for s, e in ev)
    return ts[len(ts) // 2]


library(0)
ref = Cs[0].clone()
Cs[0].zero_()
own(0)
print(f"own vs library: max|d| {(Cs[0].float() - ref.float()).abs().max().item():.3e} (max|ref| {ref.float().abs().max().item():.3e})")
flops = 2.0 * M * N * K
for name, fn in (("library", library), ("own", own)):
    for rot in (False, True):
        t = timeit(fn, rot)
        print(f"{name:8s} {'cold' if rot else 'hot ':4s} {t * 1e3:8.1f} us  {flops / t / 1e9:8.1f} TFLOP/s")
