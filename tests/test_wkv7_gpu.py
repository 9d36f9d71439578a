"""GPU (-m gpu): the HIP WKV7 operators, called through the reference's op names (which go through the
C ABI), against the CPU oracle on the same seeded inputs, the committed golden vectors, and -- at
BASELINE.json's full size -- size-independent properties plus oracle checks on head slices.

Tolerances (floating point; the reference itself rounds every output to bf16, wkv7_cuda.cu:42,89,108-111):
  bf16 I/O : |hip - oracle| <= 2^-7 * max(|oracle|, floor) elementwise  (one bf16 ulp at the value's scale;
             the two sides differ only in fp32 summation order and expf vs v_exp_f32)
  fp32 I/O : forward 2e-5 * max|oracle|; backward 5e-4 * max|oracle| (the state reconstruction
             S_{t-1} = (S_t - ...)/w~ amplifies fp32 rounding by up to (1/w~)^15 inside a 16-step chunk)
"""
import pytest
import torch

from conftest import load_golden
from rwkvtts_amd import ops
from rwkvtts_amd.synthetic import make_wkv_inputs

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
NAMES = ("dw", "dq", "dk", "dv", "da", "db")


def _assert_bf16_close(got, want, what, ulps=1.0):
    got, want = got.float().cpu(), want.float()
    floor = want.abs().mean().item() * 0.25 + 1e-6
    tol = ulps * 2.0 ** -7 * torch.clamp(want.abs(), min=floor)
    bad = ((got - want).abs() > tol)
    assert not bad.any(), f"{what}: {bad.sum().item()}/{bad.numel()} beyond {ulps} bf16 ulp, " \
                          f"max|d|={(got - want).abs().max().item():.3e} max|ref|={want.abs().max().item():.3e}"


def _assert_f32_close(got, want, what, tol):
    got = got.cpu()
    err = (got - want).abs().max().item()
    ref = want.abs().max().item()
    assert err <= tol * max(ref, 1e-3), f"{what}: max|d|={err:.3e} max|ref|={ref:.3e} tol={tol}"


def _hip_fwd_bwd(ins_cpu, dy_cpu):
    ins = [t.to(DEV).requires_grad_(True) for t in ins_cpu]
    y = ops.WindBackstepping.apply(*ins)
    y.backward(dy_cpu.to(DEV))
    torch.cuda.synchronize()
    return y.detach(), [t.grad for t in ins]


@pytest.mark.parametrize("B,T,H,seed", [(1, 16, 1, 0), (2, 64, 3, 1), (2, 512, 12, 2), (1, 1024, 2, 3)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_fwd_bwd_vs_oracle(c_oracle, B, T, H, seed, dtype):
    ins = make_wkv_inputs(B, T, H, seed, dtype)
    dy = (torch.randn(B, T, H, 64, generator=torch.Generator().manual_seed(seed + 100))).to(dtype)
    y_o, s_o, sa_o = c_oracle.wkv7_fwd(*ins)
    g_o = c_oracle.wkv7_bwd(*ins, dy, s_o, sa_o)
    y, grads = _hip_fwd_bwd(ins, dy)
    if dtype == torch.bfloat16:
        _assert_bf16_close(y, y_o, "y")
        for n, g, go in zip(NAMES, grads, g_o):
            _assert_bf16_close(g, go, n, ulps=2.0)
    else:
        _assert_f32_close(y, y_o, "y", 2e-5)
        for n, g, go in zip(NAMES, grads, g_o):
            _assert_f32_close(g, go, n, 5e-4)


def test_saved_tensors_match_reference_layout(c_oracle):
    """s is [B,H,T/16,64,64] stored transposed, sa is [B,T,H,64] fp32 (wkv7_cuda.cu:32,45-48)."""
    B, T, H = 2, 64, 3
    ins = make_wkv_inputs(B, T, H, 5, torch.float32)
    y_o, s_o, sa_o = c_oracle.wkv7_fwd(*ins)
    d = [t.to(DEV) for t in ins]
    y = torch.empty_like(d[0])
    s = torch.empty(B, H, T // 16, 64, 64, device=DEV)
    sa = torch.empty(B, T, H, 64, device=DEV)
    torch.ops.wind_backstepping.forward(*d, y, s, sa)
    _assert_f32_close(s, s_o, "s", 2e-5)
    _assert_f32_close(sa, sa_o, "sa", 2e-5)
    _assert_f32_close(y, y_o, "y", 2e-5)


def test_golden_bf16_vectors():
    g = load_golden("wkv7_scan.npz")
    for tag in ("B1T16H1", "B2T64H3"):
        ins = [g[f"{tag}.{n}"] for n in ("w", "q", "k", "v", "a", "b")]
        y, grads = _hip_fwd_bwd(ins, g[f"{tag}.dy"])
        _assert_bf16_close(y, g[f"{tag}.y"], f"{tag}.y")
        for n, gr in zip(NAMES, grads):
            _assert_bf16_close(gr, g[f"{tag}.{n}"], f"{tag}.{n}", ulps=2.0)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("B,T,H", [(1, 1, 1), (3, 7, 2), (2, 16, 4), (2, 37, 3), (32, 1, 16), (1, 128, 2)])
def test_state_fwd_vs_oracle(c_oracle, B, T, H, dtype):
    """rwkv7_state_fwd_fp16.forward: ragged T, T=1 decode, in-place state update."""
    w, q, k, v, a, b = [t.view(B, T, H * 64) for t in make_wkv_inputs(B, T, H, 7, dtype)]
    st0 = torch.randn(B, H, 64, 64, generator=torch.Generator().manual_seed(1)) * 0.1
    st_o = st0.clone()
    y_o = c_oracle.wkv7_state_fwd(st_o, q, w, k, v, a, b)
    st = st0.to(DEV)
    y = ops.RWKV7_BATCH_OP(st, *[t.to(DEV) for t in (q, w, k, v, a, b)])
    torch.cuda.synchronize()
    if dtype == torch.bfloat16:
        _assert_bf16_close(y, y_o, "y")
    else:
        _assert_f32_close(y, y_o, "y", 2e-5)
    _assert_f32_close(st, st_o, "state", 2e-5)


def test_wkv7s_b1_op(c_oracle):
    T, H = 24, 2
    w, q, k, v, a, b = [t.view(T, H * 64) for t in make_wkv_inputs(1, T, H, 11, torch.bfloat16)]
    st_o = torch.zeros(1, H, 64, 64)
    y_o = c_oracle.wkv7_state_fwd(st_o, *[t.view(1, T, H * 64) for t in (q, w, k, v, a, b)])
    st = torch.zeros(H, 64, 64, device=DEV)
    y = ops.RWKV7_OP(st, *[t.to(DEV) for t in (q, w, k, v, a, b)])
    _assert_bf16_close(y, y_o.view(T, H * 64), "y")
    _assert_f32_close(st, st_o[0], "state", 2e-5)


def test_state_carry_split_is_bit_identical():
    """G2: one call over T == two calls over T1 + (T-T1), bit for bit, and == the zero-state training fwd."""
    B, T, H = 2, 256, 4
    w, q, k, v, a, b = [t.to(DEV) for t in make_wkv_inputs(B, T, H, 13, torch.bfloat16)]
    f = lambda t: t.view(B, T, H * 64)
    st1 = torch.zeros(B, H, 64, 64, device=DEV)
    y1 = ops.RWKV7_BATCH_OP(st1, f(q), f(w), f(k), f(v), f(a), f(b))
    st2 = torch.zeros(B, H, 64, 64, device=DEV)
    T1 = 96  # whole 16-step stages on both sides: same unrolled code path, so bit for bit
    ya = ops.RWKV7_BATCH_OP(st2, *[f(t)[:, :T1].contiguous() for t in (q, w, k, v, a, b)])
    yb = ops.RWKV7_BATCH_OP(st2, *[f(t)[:, T1:].contiguous() for t in (q, w, k, v, a, b)])
    assert torch.equal(torch.cat([ya, yb], 1), y1)
    assert torch.equal(st1, st2)
    # ragged split (T1 = 100): the 4-step tail runs the rolled loop whose FMAs the compiler contracts
    # differently, so fp32-rounding equal, not bit equal
    st3 = torch.zeros(B, H, 64, 64, device=DEV)
    yc = ops.RWKV7_BATCH_OP(st3, *[f(t)[:, :100].contiguous() for t in (q, w, k, v, a, b)])
    yd = ops.RWKV7_BATCH_OP(st3, *[f(t)[:, 100:].contiguous() for t in (q, w, k, v, a, b)])
    _assert_bf16_close(torch.cat([yc, yd], 1), y1.cpu(), "ragged split")
    _assert_f32_close(st3, st1.cpu(), "ragged split state", 2e-5)
    # the zero-state instantiation is a different template instance of the same kernel: the compiler may
    # contract/reorder its fp32 FMAs differently, so equal to one bf16 ulp rather than bit for bit
    y3 = ops.wkv7_forward_nograd(f(q), f(w), f(k), f(v), f(a), f(b))
    _assert_bf16_close(y3, y1.cpu(), "nograd vs state")


def test_error_behaviour_on_device():
    x = torch.zeros(1, 24, 1, 64, dtype=torch.bfloat16, device=DEV)
    with pytest.raises(AssertionError):  # T % 16, rwkv_s2s_single_ffn.py:19
        ops.WindBackstepping.apply(x, x, x, x, x, x)
    y = torch.empty_like(x)
    s = torch.empty(1, 1, 1, 64, 64, device=DEV)
    sa = torch.empty(1, 24, 1, 64, device=DEV)
    with pytest.raises(ValueError):  # C ABI returns RWKV7_ECHUNK instead of aborting (wkv7_cuda.cu:136)
        torch.ops.wind_backstepping.forward(x, x, x, x, x, x, y, s, sa)
    with pytest.raises(ValueError):  # non-contiguous, rwkv_s2s_single_ffn.py:21
        xt = torch.zeros(1, 16, 2, 64, dtype=torch.bfloat16, device=DEV)[:, :, :1]  # strided view
        assert not xt.is_contiguous()
        torch.ops.wind_backstepping.forward(xt, xt, xt, xt, xt, xt, xt.contiguous(), s, sa[:, :16].contiguous())


def test_full_size_config2_properties_and_slices(c_oracle):
    """BASELINE.json configs[1]: B=8, T=4096, H=16 bf16.  (i) the recurrence is linear in v for fixed
    w,q,k,a,b: y(v1+v2) == y(v1)+y(v2) up to bf16 rounding; (ii) zero-state forward == state-carrying
    forward in two halves, bit for bit; (iii) forward+backward of two (batch, head) slices equal the
    oracle run on exactly those slices."""
    B, T, H = 8, 4096, 16
    ins = make_wkv_inputs(B, T, H, 1234, torch.bfloat16)
    w, q, k, v, a, b = [t.to(DEV) for t in ins]
    dy_cpu = torch.randn(B, T, H, 64, generator=torch.Generator().manual_seed(99)).bfloat16()
    leaves = [t.clone().requires_grad_(True) for t in (w, q, k, v, a, b)]
    y = ops.WindBackstepping.apply(*leaves)
    y.backward(dy_cpu.to(DEV))
    torch.cuda.synchronize()
    assert torch.isfinite(y.float()).all()
    for n, lf in zip(NAMES, leaves):
        assert torch.isfinite(lf.grad.float()).all(), n
    # (iii) oracle on slices
    for (bi, hi) in ((0, 0), (5, 11)):
        sl = [t[bi:bi + 1, :, hi:hi + 1].contiguous() for t in ins]
        y_o, s_o, sa_o = c_oracle.wkv7_fwd(*sl)
        g_o = c_oracle.wkv7_bwd(*sl, dy_cpu[bi:bi + 1, :, hi:hi + 1].contiguous(), s_o, sa_o)
        _assert_bf16_close(y[bi:bi + 1, :, hi:hi + 1], y_o, f"y[{bi},{hi}]")
        for n, lf, go in zip(NAMES, leaves, g_o):
            _assert_bf16_close(lf.grad[bi:bi + 1, :, hi:hi + 1], go, f"{n}[{bi},{hi}]", ulps=2.0)
    # (ii) split == whole
    f = lambda t: t.view(B, T, H * 64)
    st = torch.zeros(B, H, 64, 64, device=DEV)
    halves = [ops.RWKV7_BATCH_OP(st, *[f(t)[:, i * 2048:(i + 1) * 2048].contiguous() for t in (q, w, k, v, a, b)])
              for i in range(2)]
    _assert_bf16_close(torch.cat(halves, 1).view(B, T, H, 64), y.detach().cpu(), "split vs whole")
    # (i) linearity in v (fp32 I/O so that the check is not dominated by bf16 output rounding)
    sub = [t[:2, :1024].float().contiguous() for t in (w, q, k, v, a, b)]
    v2 = torch.randn_like(sub[3])
    fq = lambda vv: ops.wkv7_forward_nograd(*[t.view(2, 1024, H * 64) for t in (sub[1], sub[0], sub[2], vv, sub[4], sub[5])])
    lhs = fq(sub[3] + v2)
    rhs = fq(sub[3]) + fq(v2)
    assert (lhs - rhs).abs().max().item() <= 1e-4 * rhs.abs().max().item()


@pytest.fixture(params=[1, 0, None], ids=["512thr", "256thr", "plain-entry"])
def bwd_shape(request):
    """explicit `wide` argument of rwkv7_wkv_bwd_split_variant_* (bf16), None = the plain entry point; no global switch"""
    return request.param


@pytest.mark.parametrize("B,T,H,seed", [(1, 16, 1, 0), (2, 64, 3, 1), (2, 512, 12, 2)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_row_split_backward_vs_oracle(c_oracle, bwd_shape, B, T, H, seed, dtype):
    """rwkv7_wkv_bwd_split_*: two workgroups per head (512 threads x 1 row per lane tile, or 256 threads x 2 rows);
    dv complete, the other five as two partial column sums."""
    ins = make_wkv_inputs(B, T, H, seed, dtype)
    dy = (torch.randn(B, T, H, 64, generator=torch.Generator().manual_seed(seed + 100))).to(dtype)
    y_o, s_o, sa_o = c_oracle.wkv7_fwd(*ins)
    g_o = dict(zip(NAMES, c_oracle.wkv7_bwd(*ins, dy, s_o, sa_o)))
    d = [t.to(DEV) for t in ins]
    y, s, sa = torch.empty_like(d[0]), torch.empty(B, H, T // 16, 64, 64, device=DEV), torch.empty(B, T, H, 64, device=DEV)
    ops.wkv7_forward_scalar(*d, y, s, sa)   # the scalar pair: s = the reference's checkpoints (the op itself may launch the chunked pair)
    if dtype != torch.bfloat16 and bwd_shape is not None:
        pytest.skip("the shape variant entry exists for bf16 only")
    dw2, dq2, dk2, dv, da2, db2 = ops.wkv7_backward_split(*d, dy.to(DEV), s, sa, wide=bwd_shape)
    full = [torch.empty_like(d[0]) for _ in range(6)]
    ops.wkv7_backward_scalar(*d, dy.to(DEV), s, sa, *full)
    # dv is a complete row sum in every shape; the instantiations may associate the fp32 terms differently
    if dtype == torch.bfloat16:
        _assert_bf16_close(dv, full[3].float().cpu(), "dv vs unsplit kernel", ulps=1.0)
    else:
        _assert_f32_close(dv, full[3].cpu(), "dv vs unsplit kernel", 1e-5)
    got = dict(dw=dw2, dq=dq2, dk=dk2, da=da2, db=db2)
    for n, g2 in got.items():
        tot = g2[0].float() + g2[1].float()
        if dtype == torch.bfloat16:
            # each half is rounded to bf16 on its own (half an ulp of |half|, the halves can cancel), on top of the
            # 2 ulp granted to the unsplit kernel
            want = g_o[n].float()
            floor = want.abs().mean().item() * 0.25 + 1e-6
            tol = 2.0 ** -8 * (g2[0].float().abs() + g2[1].float().abs()).cpu() + 2.0 ** -6 * torch.clamp(want.abs(), min=floor)
            bad = (tot.cpu() - want).abs() > tol
            assert not bad.any(), f"{n}: {bad.sum().item()}/{bad.numel()} out of tolerance"
        else:
            _assert_f32_close(tot, g_o[n], n, 5e-4)
    if dtype == torch.bfloat16:
        _assert_bf16_close(dv, g_o["dv"], "dv", ulps=2.0)


@pytest.mark.parametrize("cw", [4, 8])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_forward_both_lane_shapes_vs_oracle(c_oracle, cw, dtype):
    """The launcher picks 4 or 8 state columns per lane from B*H; force each and check forward + saved tensors +
    carried state against the oracle."""
    from rwkvtts_amd import _lib
    B, T, H = 2, 64, 3
    ins = make_wkv_inputs(B, T, H, 11, dtype)
    y_o, s_o, sa_o = c_oracle.wkv7_fwd(*ins)
    d = [t.to(DEV) for t in ins]
    y, s, sa = torch.empty_like(d[0]), torch.empty(B, H, T // 16, 64, 64, device=DEV), torch.empty(B, T, H, 64, device=DEV)
    st0 = torch.randn(B, H, 64, 64, generator=torch.Generator().manual_seed(5)) * 0.1
    w, q, k, v, a, b = ins
    st_o = st0.clone()
    y2_o = c_oracle.wkv7_state_fwd(st_o, *[t.reshape(B, T, H * 64) for t in (q, w, k, v, a, b)])
    st = st0.to(DEV).clone()
    y2 = torch.empty_like(d[0])
    f3 = lambda t: t.view(B, T, H * 64)
    import ctypes
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    sfx = "bf16" if dtype == torch.bfloat16 else "f32"
    lib = _lib.lib()
    # explicit shape argument of the *_variant entry points (each variant has its own entry point; the library has no global switches)
    rc = getattr(lib, "rwkv7_wkv_fwd_variant_" + sfx)(B, T, H, *[P(t) for t in d], P(y), P(s), P(sa), cw, None)
    assert rc == 0
    if dtype == torch.bfloat16:
        rc = lib.rwkv7_wkv_state_fwd_variant_bf16(B, T, H * 64, H, P(st), P(d[1]), P(d[0]), P(d[2]), P(d[3]), P(d[4]), P(d[5]), P(y2),
                                                  cw, None)
        assert rc == 0
    else:
        torch.ops.rwkv7_state_fwd_fp16.forward(B, T, H * 64, H, st, f3(d[1]), f3(d[0]), f3(d[2]), f3(d[3]), f3(d[4]),
                                                f3(d[5]), f3(y2))
    torch.cuda.synchronize()
    assert getattr(lib, "rwkv7_wkv_fwd_variant_" + sfx)(B, T, H, *[P(t) for t in d], P(y), P(s), P(sa), 5, None) == -4
    if dtype == torch.bfloat16:
        _assert_bf16_close(y, y_o, "y")
        _assert_bf16_close(y2, y2_o.view(B, T, H, 64), "y(state)")
    else:
        _assert_f32_close(y, y_o, "y", 2e-5)
        _assert_f32_close(y2, y2_o.view(B, T, H, 64), "y(state)", 2e-5)
    _assert_f32_close(s, s_o, "s", 2e-5)
    _assert_f32_close(sa, sa_o, "sa", 2e-5)
    _assert_f32_close(st, st_o, "state", 2e-5)


@pytest.mark.parametrize("B,T,H,seed", [(1, 32, 1, 0), (2, 64, 3, 1), (2, 512, 12, 2)])
def test_reference_op_launches_the_chunked_pair_for_bf16_T32_and_the_scalar_pair_otherwise(c_oracle, B, T, H, seed):
    """torch.ops.wind_backstepping.forward / .backward (wkv7_op.cpp:21-24, the plug-in point SURVEY 8b names): for bf16 tensors
    with T % 32 == 0 the op runs the chunked MFMA kernels with `s` as their arena (rwkv7_wkv_fwd_fast_bf16), otherwise the scalar
    kernels -- asserted from the launch timers -- and either way y, sa and the six gradients meet the oracle's bars."""
    ins = make_wkv_inputs(B, T, H, seed, torch.bfloat16)
    dy = (torch.randn(B, T, H, 64, generator=torch.Generator().manual_seed(seed + 100))).bfloat16()
    y_o, s_o, sa_o = c_oracle.wkv7_fwd(*ins)
    g_o = c_oracle.wkv7_bwd(*ins, dy, s_o, sa_o)
    d = [t.to(DEV) for t in ins]
    for fast in (True, False):
        ops.REFERENCE_OP_FAST = fast
        ops.KERNEL_TIMERS = {}
        try:
            y = torch.empty_like(d[0])
            s = torch.empty(B, H, T // 16, 64, 64, device=DEV)
            sa = torch.empty(B, T, H, 64, device=DEV)
            grads = [torch.empty_like(d[0]) for _ in range(6)]
            torch.ops.wind_backstepping.forward(*d, y, s, sa)
            torch.ops.wind_backstepping.backward(*d, dy.to(DEV), s, sa, *grads)
            torch.cuda.synchronize()
            ran = set(ops.KERNEL_TIMERS)
        finally:
            ops.KERNEL_TIMERS, ops.REFERENCE_OP_FAST = None, True
        assert ran == ({"wkv7c_op_fwd", "wkv7c_op_bwd"} if fast else {"wkv7_fwd", "wkv7_bwd"}), ran
        assert ops.reference_op_is_fast(torch.bfloat16, T) and not ops.reference_op_is_fast(torch.bfloat16, T + 16) \
            and not ops.reference_op_is_fast(torch.float32, T)
        _assert_bf16_close(y, y_o, "y")
        _assert_f32_close(sa, sa_o, "sa", 2e-3 if fast else 2e-5)
        for n, g, go in zip(NAMES, grads, g_o):
            _assert_bf16_close(g, go, n, ulps=2.0)
    # T % 32 == 16: the op falls back to the scalar pair by itself
    ins = make_wkv_inputs(1, 48, 2, seed, torch.bfloat16)
    d = [t.to(DEV) for t in ins]
    ops.KERNEL_TIMERS = {}
    try:
        yg, _ = _hip_fwd_bwd(ins, torch.randn(1, 48, 2, 64).bfloat16())
        ran = set(ops.KERNEL_TIMERS)
    finally:
        ops.KERNEL_TIMERS = None
    assert ran == {"wkv7_fwd", "wkv7_bwd"}, ran
    _assert_bf16_close(yg, c_oracle.wkv7_fwd(*ins)[0], "y (T = 48)")
