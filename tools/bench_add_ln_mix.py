"""add + LayerNorm + token-shift lerps: the one-pass kernels against the two separate stages, forward and backward, at B*T = 32768, D = 1024."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rwkvtts_amd import fused
dev = "cuda:0"
B, T, D = 8, 4096, 1024


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


norm = torch.nn.LayerNorm(D).to(dev).bfloat16()
for nmix in (1, 6):
    for run, blocks, bblocks in ((8, 4096, 1024), (8, 4096, 2048), (8, 4096, 4096), (4, 8192, 2048), (4, 8192, 4096), (4, 8192, 8192), (16, 2048, 2048)):
        fused._ADD_LN_MIX_RUN, fused._ADD_LN_MIX_BLOCKS, fused._ADD_LN_MIX_BWD_BLOCKS = run, blocks, bblocks
        x = torch.randn(B, T, D, device=dev).bfloat16().requires_grad_(True)
        br = torch.randn(B, T, D, device=dev).bfloat16().requires_grad_(True)
        mixp = tuple(torch.rand(D, device=dev).bfloat16().requires_grad_(True) for _ in range(nmix))
        gout = [torch.randn(B, T, D, device=dev).bfloat16() for _ in range(nmix)]
        gx1 = torch.randn(B, T, D, device=dev).bfloat16()

        def fwd_fused():
            return fused.add_layer_norm_mix(x, br, norm, None, mixp)

        def fwd_sep():
            x1, h = fused.add_layer_norm(x, br, norm)
            outs = fused.token_shift_mix6(h, None, *mixp) if nmix == 6 else (fused.token_shift_mix1(h, None, mixp[0]),)
            return x1, outs

        def both(f):
            x1, outs = f()
            torch.autograd.backward([x1] + list(outs), [gx1] + gout)

        tf, ts = timeit(fwd_fused), timeit(fwd_sep)
        tbf, tbs = timeit(lambda: both(fwd_fused)), timeit(lambda: both(fwd_sep))
        print(f"nmix {nmix} run {run:2d} blocks {blocks:5d} / {bblocks:5d}: fwd one-pass {tf:6.1f} us, separate {ts:6.1f} us | fwd+bwd one-pass {tbf:6.1f} us, separate {tbs:6.1f} us", flush=True)
