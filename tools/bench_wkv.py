"""Kernel-only timing of the WKV7 ops at BASELINE.json sizes (HIP events on the launch stream).

    python tools/bench_wkv.py [--B 8 --T 4096 --H 16 --iters 10 --dtype bf16]
Prints ms per launch, algorithmic GB/s (896 B fwd / 1664 B bwd per token-head)
and the fraction of the 8 TB/s HBM roofline.
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rwkvtts_amd import ops  # noqa: E402
from rwkvtts_amd.synthetic import make_wkv_inputs  # noqa: E402

HBM_PEAK = 8.0e12


def timeit(fn, iters, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for s, e in ev:
        s.record()
        fn()
        e.record()
    torch.cuda.synchronize()
    ts = sorted(s.elapsed_time(e) for s, e in ev)
    return ts[len(ts) // 2], ts[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=8)
    ap.add_argument("--T", type=int, default=4096)
    ap.add_argument("--H", type=int, default=16)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--dtype", default="bf16")
    a = ap.parse_args()
    dt = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    esz = 2 if a.dtype == "bf16" else 4
    B, T, H = a.B, a.T, a.H
    dev = "cuda:0"
    w, q, k, v, aa, b = make_wkv_inputs(B, T, H, 1234, dt, dev)
    dy = torch.randn(B, T, H, 64, device=dev).to(dt)
    y = torch.empty_like(v)
    s = torch.empty(B, H, T // 16, 64, 64, device=dev)
    sa = torch.empty(B, T, H, 64, device=dev)
    grads = [torch.empty_like(w) for _ in range(6)]
    th = B * T * H
    fwd = lambda: ops.wkv7_forward_scalar(w, q, k, v, aa, b, y, s, sa)
    bwd = lambda: ops.wkv7_backward_scalar(w, q, k, v, aa, b, dy, s, sa, *grads)
    bwd2 = lambda: ops.wkv7_backward_split(w, q, k, v, aa, b, dy, s, sa)
    st = torch.zeros(B, H, 64, 64, device=dev)
    f3 = lambda t: t.view(B, T, H * 64)
    yy = torch.empty(B, T, H * 64, device=dev, dtype=dt)
    sfw = lambda: torch.ops.rwkv7_state_fwd_fp16.forward(B, T, H * 64, H, st, f3(q), f3(w), f3(k), f3(v), f3(aa), f3(b), yy)
    prep = lambda: ops.wkv7_chunk_prep(w, aa, b)
    tinv = prep()
    sa2 = torch.empty(B, T, H, 64, device=dev)
    hs = torch.empty(B, H, T // 32, ops.Q15_REC, device=dev, dtype=torch.int16)
    import ctypes
    from rwkvtts_amd import _lib
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    st_ = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    sfx = "bf16" if a.dtype == "bf16" else "f32"
    cf = lambda: getattr(_lib.lib(), "rwkv7_wkv_chunk_fwd_" + sfx)(B, T, H, P(w), P(q), P(k), P(v), P(aa), P(b), P(tinv), P(y), P(sa2), P(hs), st_)
    def shaped(cw, fn):
        # explicit shape through the *_variant entry points (bf16 for the state-carrying one)
        if fn is fwd:
            return lambda: getattr(_lib.lib(), "rwkv7_wkv_fwd_variant_" + sfx)(B, T, H, P(w), P(q), P(k), P(v), P(aa), P(b), P(y), P(s), P(sa), cw, st_)
        if a.dtype != "bf16":
            return None
        return lambda: _lib.lib().rwkv7_wkv_state_fwd_variant_bf16(B, T, H * 64, H, P(st), P(q), P(w), P(k), P(v), P(aa), P(b), P(yy), cw, st_)
    hs16 = None
    if a.dtype == "bf16":
        _, _, sa2, hs16 = ops.wkv7_chunk_forward(w, q, k, v, aa, b)
    cb = (lambda: ops.wkv7_chunk_backward(w, q, k, v, aa, b, dy, hs16, sa2, tinv)) if a.dtype == "bf16" else None
    # every bseq launch of this script writes Z, as the training step's does: a PMC pass over this script (tools/pmc_wkv.sh) averages
    # the counters of all launches of a kernel, and a Z-less launch in the mix would understate the kernel's WRITE_SIZE
    cb_seq = (lambda: ops.wkv7_chunk_bwd_seq(w, q, aa, b, dy, tinv, want_z=True)) if a.dtype == "bf16" else None
    for name, fn, bytes_per in (("wkv7_fwd(save s,sa)", fwd, 7 * 64 * esz),
                                ("wkv7_fwd 8 col/lane", shaped(8, fwd), 7 * 64 * esz),
                                ("wkv7_fwd 4 col/lane", shaped(4, fwd), 7 * 64 * esz),
                                ("wkv7_state_fwd 8 col/lane", shaped(8, sfw), 7 * 64 * esz),
                                ("wkv7_state_fwd 4 col/lane", shaped(4, sfw), 7 * 64 * esz), ("wkv7_bwd", bwd, 13 * 64 * esz),
                                ("wkv7_bwd row-split 256 thr", bwd2, 13 * 64 * esz),
                                ("wkv7_bwd row-split 512 thr", (lambda: ops.wkv7_backward_split(w, q, k, v, aa, b, dy, s, sa, wide=1)) if a.dtype == "bf16" else None, 13 * 64 * esz),
                                ("wkv7_state_fwd", sfw, 7 * 64 * esz), ("wkv7c_prep (T inverse)", prep, 3 * 64 * esz),
                                ("wkv7c_fwd (chunked, save)", cf, 7 * 64 * esz),
                                ("wkv7c bseq (adjoint recurrence + Z, 1 launch)", cb_seq, 13 * 64 * esz),
                                ("wkv7c bwd total, bseq+out (default)", cb, 13 * 64 * esz)):
        if fn is None:
            continue
        med, best = timeit(fn, a.iters)
        gbs = th * bytes_per / (med * 1e-3) / 1e9
        print(f"{name:22s} B={B} T={T} H={H} {a.dtype}: median {med:8.3f} ms  best {best:8.3f} ms  "
              f"algorithmic {gbs:8.1f} GB/s  = {gbs * 1e9 / HBM_PEAK * 100:5.2f}% of 8 TB/s")


if __name__ == "__main__":
    main()
