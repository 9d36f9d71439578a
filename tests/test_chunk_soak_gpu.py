"""GPU (-m gpu): randomized soak of the chunked (MFMA) WKV7 training pair against the C oracle -- random shapes and input REGIMES the fixed
seeds of test_chunk_gpu.py do not reach (strongest decay the model can produce in every channel, no decay at all, channel-wise mixes,
silent steps, a fully open removal gate, large and small magnitudes).  Bars as in test_chunk_gpu.py: y within 1 bf16 ulp of the oracle, the
six gradients within 2 -- and where an element is NOT, an fp64 scan of the same bf16 inputs decides which side is inexact: the HIP value
must then be within 1 bf16 ulp of that truth (or 1e-4 of the tensor's maximum, the fp32-accumulation bar).  (It is the oracle that leaves the bar in the strong-decay regimes, and faithfully so: the
reference's backward rebuilds S_{t-1} from S_t by dividing by the decay, wkv7_cuda.cu:97-104, which amplifies fp32 rounding by up to
1.83x per step between its checkpoints; the chunked kernels never divide.  soak_debug.py prints the three-way comparison of a case.)
12 cases by default (seconds); RWKV7_SOAK_CASES=N for a long run (profiles/r06z_chunk_soak.txt: 400 cases)."""
import os

import pytest
import torch
import torch.nn.functional as F

from rwkvtts_amd import ops
from test_wkv7_gpu import NAMES

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
MODES = ("typical", "max_decay", "no_decay", "mixed_decay", "silent_steps", "gate_open", "large", "small")


def soak_inputs(case):
    g = torch.Generator().manual_seed(10_000 + case)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
    mode = MODES[case % len(MODES)]
    B, T, H, N = ri(1, 3), 32 * ri(1, 12), ri(1, 4), 64
    rn = lambda s=1.0: torch.randn(B, T, H, N, generator=g) * s
    scale = {"large": 3.0, "small": 0.05}.get(mode, (0.25, 0.5, 1.0)[ri(0, 2)])
    q, k, v = rn(scale), rn(scale), rn(scale)
    w = -F.softplus(-(rn(2.0) - 1.0)) - 0.5                     # the model's range: exp(-exp(w)) in [0.545, 1)
    if mode == "max_decay":
        w = torch.full_like(w, -0.5)
    elif mode == "no_decay":
        w = torch.full_like(w, -12.0)
    elif mode == "mixed_decay":
        w = torch.where(torch.rand(1, 1, H, N, generator=g) < 0.5, torch.full_like(w, -0.5), torch.full_like(w, -9.0))
    kk = F.normalize(rn(), dim=-1)
    a = -kk
    b = kk * (torch.ones_like(kk) if mode == "gate_open" else torch.sigmoid(rn()))
    if mode == "silent_steps":                                 # steps that carry nothing (padding inside a row looks like this)
        off = (torch.rand(B, T, 1, 1, generator=g) < 0.3).float()
        q, k, v, a, b = (t * (1 - off) for t in (q, k, v, a, b))
    dy = rn((0.1, 1.0, 3.0)[ri(0, 2)])
    return mode, (B, T, H), [t.bfloat16().contiguous() for t in (w, q, k, v, a, b)], dy.bfloat16()


def fp64_truth(ins, dy):
    """The recurrence of wkv7_cuda.cu:17-51 in fp64 on the bf16 inputs, gradients by autograd."""
    w, q, k, v, a, b = [t.double().requires_grad_(True) for t in ins]
    B, T, H, N = w.shape
    S = torch.zeros(B, H, N, N, dtype=torch.float64)      # S[v][k]
    ys = []
    for t in range(T):
        dec = torch.exp(-torch.exp(w[:, t]))
        sa = torch.einsum("bhvk,bhk->bhv", S, a[:, t])
        S = S * dec[:, :, None, :] + sa[..., None] * b[:, t][:, :, None, :] + v[:, t][..., None] * k[:, t][:, :, None, :]
        ys.append(torch.einsum("bhvk,bhk->bhv", S, q[:, t]))
    y = torch.stack(ys, 1)
    y.backward(dy.double())
    return y.detach(), [t.grad for t in (w, q, k, v, a, b)]


def _close_to_oracle_or_truth(got, oracle, truth, what, ulps):
    """-> number of elements where the ORACLE is the side outside the bar (the HIP value within 1 bf16 ulp of the fp64 truth there)."""
    got, oracle, truth = got.float().cpu(), oracle.float(), truth().float()
    floor = oracle.abs().mean().item() * 0.25 + 1e-6
    bad = (got - oracle).abs() > ulps * 2.0 ** -7 * torch.clamp(oracle.abs(), min=floor)
    if bad.any():
        # 1 bf16 ulp of the truth, or the fp32-accumulation bar of test_chunk_gpu.py (1e-4 of the tensor's maximum) where the element is a
        # small remainder of large cancelling terms (dw in the mixed-decay regime: |dw| spans six orders of magnitude)
        worse = bad & ((got - truth).abs() > torch.clamp(2.0 ** -7 * torch.clamp(truth.abs(), min=floor), min=1e-4 * truth.abs().max().item()))
        assert not worse.any(), (f"{what}: {int(worse.sum())}/{bad.numel()} elements beyond {ulps} bf16 ulp of the oracle AND beyond 1 ulp of the fp64 "
                                 f"scan, max|HIP - truth| = {(got - truth)[worse].abs().max().item():.3e}, max|truth| = {truth.abs().max().item():.3e}")
    return int(bad.sum())


@pytest.mark.parametrize("case", range(int(os.environ.get("RWKV7_SOAK_CASES", "12"))))
def test_chunked_pair_soak_vs_oracle(c_oracle, case):
    mode, shape, ins, dy = soak_inputs(case)
    y_o, s_o, sa_o = c_oracle.wkv7_fwd(*ins)
    g_o = c_oracle.wkv7_bwd(*ins, dy, s_o, sa_o)
    d = [t.to(DEV) for t in ins]
    y, tinv, sa, hs = ops.wkv7_chunk_forward(*d)
    grads = ops.wkv7_chunk_backward(*d, dy.to(DEV), hs, sa, tinv)
    torch.cuda.synchronize()
    cache = []

    def truth(i):
        if not cache:
            y_t, g_t = fp64_truth(ins, dy)
            cache.extend([y_t, *g_t])
        return cache[i]

    outl = {"y": _close_to_oracle_or_truth(y, y_o, lambda: truth(0), f"y [{mode} {shape}]", 1.0)}
    for i, (n, gr, go) in enumerate(zip(NAMES, grads, g_o)):
        outl[n] = _close_to_oracle_or_truth(gr, go, lambda i=i: truth(i + 1), f"{n} [{mode} {shape}]", 2.0)
    if any(outl.values()):
        print(f"\n[soak] case {case} {mode} {shape}: elements where the oracle (fp32 backstepping) is outside the bar and the HIP value is within "
              f"1 ulp of the fp64 scan: {({k_: v_ for k_, v_ in outl.items() if v_})}")


@pytest.mark.parametrize("case", range(max(6, int(os.environ.get("RWKV7_SOAK_CASES", "12")) // 4)))
def test_packed_rows_soak_vs_oracle(c_oracle, case):
    """The same regimes on PACKED rows (seq_off: fla chunk_rwkv7's cu_seqlens at chunk granularity): random cut points, every segment
    against the oracle (or the fp64 scan) run on that segment alone from the zero state."""
    mode, (B, T, H), ins, dy = soak_inputs(5000 + case)
    g = torch.Generator().manual_seed(77 + case)
    nc = T // 32
    starts = [[0] + [c for c in range(1, nc) if float(torch.rand(1, generator=g)) < 0.35] for _ in range(B)]
    off = sorted(b * nc + c for b, st in enumerate(starts) for c in st) + [B * nc]
    seq_off = torch.tensor(off, dtype=torch.int32, device=DEV)
    d = [t.to(DEV) for t in ins]
    y, tinv, sa, hs = ops.wkv7_chunk_forward(*d, seq_off=seq_off)
    grads = ops.wkv7_chunk_backward(*d, dy.to(DEV), hs, sa, tinv, seq_off=seq_off)
    torch.cuda.synchronize()
    for b, st in enumerate(starts):
        bounds = [32 * c for c in st] + [T]
        for lo, hi in zip(bounds[:-1], bounds[1:]):
            seg = [t[b:b + 1, lo:hi].contiguous() for t in ins]
            dseg = dy[b:b + 1, lo:hi].contiguous()
            y_o, s_o, sa_o = c_oracle.wkv7_fwd(*seg)
            g_o = c_oracle.wkv7_bwd(*seg, dseg, s_o, sa_o)
            cache = []

            def truth(i):
                if not cache:
                    y_t, g_t = fp64_truth(seg, dseg)
                    cache.extend([y_t, *g_t])
                return cache[i]

            tag = f"[{mode} {(B, T, H)} row {b} steps {lo}:{hi}]"
            _close_to_oracle_or_truth(y[b:b + 1, lo:hi], y_o, lambda: truth(0), "y " + tag, 1.0)
            for i, (n, gr, go) in enumerate(zip(NAMES, grads, g_o)):
                _close_to_oracle_or_truth(gr[b:b + 1, lo:hi], go, lambda i=i: truth(i + 1), f"{n} {tag}", 2.0)
