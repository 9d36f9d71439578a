import ctypes, os, sys, glob
import torch
M, N, K = (int(x) for x in sys.argv[1:4]) if len(sys.argv) >= 4 else (32768, 4096, 1024)
dev = "cuda:0"
A = (torch.randn(M, K, device=dev) * 0.5).bfloat16(); W = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
P = lambda t: ctypes.c_void_p(t.data_ptr())
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for so in sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "tim*.so"))):
    L = ctypes.CDLL(so)
    tb = torch.zeros(256 * 4 * 4, device=dev, dtype=torch.int64)
    for _ in range(60):
        rc = L.lab_gemm_t(M, N, K, P(A), P(W), P(C), P(tb), st()); assert rc == 0
    torch.cuda.synchronize()
    t = tb.view(256, 4, 4).double()
    nkt = (M // 256) * (N // 256) // 256 * (K // 64)
    print(os.path.basename(so), "per K tile (cycles), mean over WGs, per wave:")
    m = t.mean((0, 1))
    tot = t[:, :, :3].sum(2).mean(1)   # per WG
    print(f"  per-WG total cycles: min {tot.min():.0f} mean {tot.mean():.0f} max {tot.max():.0f};  by XCD (wg % 8): " + " ".join(f"{tot[x::8].mean():.0f}" for x in range(8)))
    ntile = (M // 256) * (N // 256) // 256
    print(f"  per K tile: wait+barrier (4 phases) {m[0]/nkt:.0f}  phase bodies {m[1]/nkt:.0f}   per tile: epilogue {m[2]/ntile:.0f}   total per K tile {m[:3].sum()/nkt:.0f}   clock {m[3]/2**24*100:.0f} MHz")
