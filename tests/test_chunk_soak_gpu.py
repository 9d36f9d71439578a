"""GPU (-m gpu): randomized soak of the chunked (MFMA) WKV7 training pair against the C oracle -- random shapes and input REGIMES the fixed
seeds of test_chunk_gpu.py do not reach (strongest decay the model can produce in every channel, no decay at all, channel-wise mixes,
silent steps, a fully open removal gate, large and small magnitudes).  Bars as in test_chunk_gpu.py: y within 1 bf16 ulp of the oracle, the
six gradients within 2 -- and where an element is NOT, an fp64 scan of the same bf16 inputs decides which side is inexact: the HIP value
must then be within 1 bf16 ulp of that truth (or 1e-4 of the tensor's maximum, the fp32-accumulation bar).  (It is the oracle that leaves the bar in the strong-decay regimes, and faithfully so: the
reference's backward rebuilds S_{t-1} from S_t by dividing by the decay, wkv7_cuda.cu:97-104, which amplifies fp32 rounding by up to
1.83x per step between its checkpoints; the chunked kernels never divide.  soak_debug.py prints the three-way comparison of a case.)
8 cases by default (all eight regimes once); RWKV7_SOAK_CASES=N for a long run (profiles/r06z_chunk_soak.txt: 400 cases)."""
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


@pytest.mark.parametrize("case", range(int(os.environ.get("RWKV7_SOAK_CASES", "8"))))
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


@pytest.mark.parametrize("case", range(max(4, int(os.environ.get("RWKV7_SOAK_CASES", "8")) // 4)))
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


@pytest.mark.parametrize("case", range(max(8, int(os.environ.get("RWKV7_SOAK_CASES", "8")) // 4)))
def test_reference_op_soak_vs_oracle(c_oracle, case):
    """The drop-in boundary itself in the same regimes: torch.ops.wind_backstepping.forward / backward through ops.WindBackstepping
    (rwkv_s2s_single_ffn.py:15-35) with T any multiple of 16 -- multiples of 32 in bf16 take the chunked pair, the others and fp32 the scalar
    kernels -- y and the six gradients against the oracle, the fp64 scan deciding where they part."""
    mode, (B, T, H), ins, dy = soak_inputs(9000 + case)
    g = torch.Generator().manual_seed(33 + case)
    T = 16 * int(torch.randint(1, 25, (1,), generator=g))
    T = min(T, ins[0].shape[1]) if ins[0].shape[1] >= 16 else 16
    T -= T % 16
    T = max(T, 16)
    dtype = torch.float32 if case % 2 else torch.bfloat16
    ins = [t[:, :T].contiguous().to(dtype) for t in ins]      # bf16-representable values in both dtypes
    dy = dy[:, :T].contiguous().to(dtype)
    y_o, s_o, sa_o = c_oracle.wkv7_fwd(*ins)
    g_o = c_oracle.wkv7_bwd(*ins, dy, s_o, sa_o)
    d = [t.to(DEV).requires_grad_(True) for t in ins]
    y = ops.WindBackstepping.apply(*d)
    y.backward(dy.to(DEV))
    torch.cuda.synchronize()
    cache = []

    def truth(i):
        if not cache:
            y_t, g_t = fp64_truth(ins, dy)
            cache.extend([y_t, *g_t])
        return cache[i]

    tag = f"[{mode} {(B, T, H)} {str(dtype)[6:]}]"
    outs = [("y", y.detach(), y_o, 1.0, 2e-5)] + [(n, t.grad, go, 2.0, 5e-4) for n, t, go in zip(NAMES, d, g_o)]
    for i, (n, got, want, ulps, ftol) in enumerate(outs):
        if dtype == torch.bfloat16:
            _close_to_oracle_or_truth(got, want, lambda i=i: truth(i), f"{n} {tag}", ulps)
        else:
            got = got.float().cpu()
            if (got - want).abs().max().item() > ftol * max(want.abs().max().item(), 1e-3):
                # fp32 here means the SCALAR kernels, which restate the reference's algorithm including its division by the decay: in the
                # strong-decay regimes both they and the oracle are ~1e-3 of the maximum away from the fp64 scan, in different directions
                # (other summation order under the same amplification).  The kernel must not be worse than the reference's own algorithm.
                t = truth(i).float()
                e_h, e_o = (got - t).abs().max().item(), (want - t).abs().max().item()
                print(f"\n[soak] case {case} {n} {tag}: |HIP - fp64| = {e_h:.3e}, |oracle - fp64| = {e_o:.3e}, max|fp64| = {t.abs().max().item():.3e}")
                assert e_h <= max(ftol * max(t.abs().max().item(), 1e-3), 2.0 * e_o), \
                    f"{n} {tag}: beyond {ftol} of the oracle, and {e_h:.3e} from the fp64 scan where the oracle is {e_o:.3e}"


@pytest.mark.parametrize("case", range(max(8, int(os.environ.get("RWKV7_SOAK_CASES", "8")) // 4)))
def test_state_forward_soak_vs_oracle(c_oracle, case):
    """rwkv7_state_fwd_fp16.forward (rwkv7_state_fwd_fp16.cu:9-57: prefill and decode on an external fp32 state, updated in place) in the
    same regimes: ragged T (1 ... 70, no multiple required), initial states from zero to 10x the typical magnitude, bf16 and fp32; then the
    SAME rows again in two calls (split at a random step): the carried state must give the same result as the single call, bit for bit."""
    mode, (B, _, H), ins, _ = soak_inputs(13000 + case)
    g = torch.Generator().manual_seed(55 + case)
    T = int(torch.randint(1, min(71, ins[0].shape[1] + 1), (1,), generator=g))
    dtype = torch.float32 if case % 2 else torch.bfloat16
    w, q, k, v, a, b = [t[:, :T].contiguous().to(dtype).view(B, T, H * 64) for t in ins]
    st0 = torch.randn(B, H, 64, 64, generator=g) * (0.0, 0.1, 1.0, 10.0)[case % 4]
    st_o = st0.clone()
    y_o = c_oracle.wkv7_state_fwd(st_o, q, w, k, v, a, b)
    d = [t.to(DEV) for t in (q, w, k, v, a, b)]
    st = st0.to(DEV)
    y = ops.RWKV7_BATCH_OP(st, *d)
    torch.cuda.synchronize()
    tag = f"[{mode} {(B, T, H)} {str(dtype)[6:]} state x{(0.0, 0.1, 1.0, 10.0)[case % 4]}]"
    if dtype == torch.bfloat16:
        _close_to_oracle_or_truth(y, y_o, lambda: y_o, "y " + tag, 1.0)     # no backward, no division: the oracle IS the reference here
    else:
        assert (y.cpu() - y_o).abs().max().item() <= 2e-5 * max(y_o.abs().max().item(), 1e-3), "y " + tag
    assert (st.cpu() - st_o).abs().max().item() <= 2e-5 * max(st_o.abs().max().item(), 1e-3), "state " + tag
    if T > 1:
        cut = int(torch.randint(1, T, (1,), generator=g))
        st2 = st0.to(DEV)
        y1 = ops.RWKV7_BATCH_OP(st2, *[t[:, :cut].contiguous() for t in d])
        y2 = ops.RWKV7_BATCH_OP(st2, *[t[:, cut:].contiguous() for t in d])
        assert torch.equal(torch.cat([y1, y2], 1), y) and torch.equal(st2, st), "split at step %d differs from the single call %s" % (cut, tag)
