"""Input-gradient GEMMs dx = dy . W (W stored [N_out, K_in] as nn.Linear keeps it: an "NN" product for the library) against the same
product on a pre-transposed copy Wt = W^T ([K_in, N_out], K-contiguous for the contraction: "NT", the layout of the forward GEMM).
    python tools/dgrad_layout_probe.py"""
import torch
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)


def timed(fn, n=60):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for s, e in ev:
        s.record(); fn(); e.record()
    torch.cuda.synchronize()
    ts = sorted(s.elapsed_time(e) * 1e3 for s, e in ev)
    return ts[len(ts) // 2]


for (M, nout, kin, what) in ((32768, 1024, 1024, "r/k/v/o dgrad"), (32768, 4096, 1024, "FFN key dgrad (dy [M,4096] -> dx [M,1024])"),
                             (32768, 1024, 4096, "FFN value dgrad (dy [M,1024] -> dx [M,4096])")):
    dy = (torch.randn(M, nout, device=dev, generator=g) * 0.1).bfloat16()
    W = (torch.randn(nout, kin, device=dev, generator=g) * 0.05).bfloat16()
    Wt = W.t().contiguous()          # [kin, nout]
    out = torch.empty(M, kin, device=dev, dtype=torch.bfloat16)
    t_nn = timed(lambda: torch.mm(dy, W, out=out))
    ref = out.clone()
    t_nt = timed(lambda: torch.mm(dy, Wt.t(), out=out))
    same = torch.equal(ref, out)
    t_tr = timed(lambda: W.t().contiguous())
    fl = 2.0 * M * nout * kin
    print(f"{what:48s} NN {t_nn:7.1f} us ({fl / t_nn / 1e6:5.0f} TF/s)   NT on W^T {t_nt:7.1f} us ({fl / t_nt / 1e6:5.0f} TF/s)   "
          f"transpose of W {t_tr:5.1f} us   identical {same}  max|d| {(ref.float() - out.float()).abs().max().item():.2e}", flush=True)
