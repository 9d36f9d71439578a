"""The channel-mix key projection with relu(.)^2 as the GEMM epilogue (first generation, csrc/lab/gemm_relusq.hip: needs `python -m rwkvtts_amd.build --lab`) against what the step runs today: the
library GEMM (hipBLASLt through torch) followed by rwkv7_relusq_fwd.  Same process, same tensors, HIP events.

    python tools/bench_gemm_relusq.py [M N K]      default 32768 4096 1024 (0.4B, B=8, L=4096)"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rwkvtts_amd import _lib, fused
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import lab

M, N, K = (int(x) for x in sys.argv[1:4]) if len(sys.argv) >= 4 else (32768, 4096, 1024)
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
A = (torch.randn(M, K, device=dev, generator=g) * 0.5).bfloat16()
W = (torch.randn(N, K, device=dev, generator=g) * 0.05).bfloat16()
C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
lib = _lib.lib()
P = lambda t: ctypes.c_void_p(t.data_ptr())
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def own(epi, variant=1):
    return lab.gemm_nt_gen1(A, W, epi, variant)


def lib_pair():
    k = torch.nn.functional.linear(A, W)
    return fused.relu_sq(k)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for s, e in ev:
        s.record(); fn(); e.record()
    torch.cuda.synchronize()
    ts = sorted(s.elapsed_time(e) for s, e in ev)
    return ts[len(ts) // 2]


ref = torch.nn.functional.linear(A, W)
got = own(0).clone()
err = (got.float() - ref.float()).abs().max().item()
print(f"plain GEMM vs library: max|d| {err:.3e} (max|ref| {ref.float().abs().max().item():.3e})")
want = torch.relu(ref) ** 2
got2 = own(1).clone()
print(f"relu^2 epilogue vs relu(library)^2: max|d| {(got2.float() - want.float()).abs().max().item():.3e}")
flops = 2.0 * M * N * K
t_lib = timeit(lambda: torch.nn.functional.linear(A, W))
t_pair = timeit(lib_pair)
for v in (0, 1):
    d = (own(1, v).float() - want.float()).abs().max().item()
    assert d == 0.0, (v, d)
rows = [("library GEMM", t_lib), ("library GEMM + relusq kernel", t_pair)]
for v, nm in ((0, "BK 64, 2 buffers"), (1, "BK 32, 4 buffers")):
    rows.append((f"own GEMM ({nm})", timeit(lambda: own(0, v))))
    rows.append((f"own GEMM + relu^2 ({nm})", timeit(lambda: own(1, v))))
for name, t in rows:
    print(f"{name:32s} {t * 1e3:8.1f} us  {flops / t / 1e9:8.1f} TFLOP/s")
