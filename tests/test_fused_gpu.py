"""GPU (-m gpu): the fused elementwise HIP stages (rwkvtts_amd/fused.py -> librwkv7_hip.so) against the plain
PyTorch restatement of the reference formulas (tests/ref_fused.py, fp32 on CPU), forward and backward
(reference gradients by torch.autograd).

Tolerances: fp32 kernels 2e-5 * max|ref| (forward) / 1e-4 (backward: long fp32 reductions over B*T rows for the
parameter gradients); bf16 kernels: inputs are bf16-exact, the kernel computes in fp32 and rounds once, so
outputs must be within 1 bf16 ulp of the fp32 reference (2^-7 relative, with an absolute floor)."""
import pytest
import torch

import ref_fused as RF
from rwkvtts_amd import fused

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _cmp(got, want, tol, what):
    got = got.detach().float().cpu()
    want = want.detach().float()
    err = (got - want).abs().max().item()
    ref = want.abs().max().item()
    assert err <= tol * max(ref, 1e-3), f"{what}: max|d|={err:.3e} max|ref|={ref:.3e} tol={tol}"


def _cmp_bf16(got, want, what, ulps=1.0):
    got = got.detach().float().cpu()
    want = want.detach().float()
    floor = want.abs().mean().item() * 0.25 + 1e-6
    tol = ulps * 2.0 ** -7 * torch.clamp(want.abs(), min=floor)
    bad = (got - want).abs() > tol
    assert not bad.any(), f"{what}: {bad.sum().item()}/{bad.numel()} beyond {ulps} bf16 ulp"


def _mk(shape, g, scale=1.0, dtype=torch.float32):
    return (torch.randn(*shape, generator=g) * scale).to(dtype).float()  # values exactly representable in dtype


def _run_both(fn_hip, fn_ref, inputs, dtype, grad_names, fwd_tol, bwd_tol):
    """inputs: dict name -> fp32 CPU tensor (already rounded to dtype) or non-tensor."""
    ref_in = {k: (v.clone().requires_grad_(k in grad_names) if torch.is_tensor(v) else v) for k, v in inputs.items()}
    hip_in = {k: (v.to(DEV, dtype).requires_grad_(k in grad_names) if torch.is_tensor(v) else v)
              for k, v in inputs.items()}
    out_r = fn_ref(**ref_in)
    out_h = fn_hip(**hip_in)
    out_r = out_r if isinstance(out_r, (tuple, list)) else (out_r,)
    out_h = out_h if isinstance(out_h, (tuple, list)) else (out_h,)
    g = torch.Generator().manual_seed(77)
    douts = [_mk(o.shape, g, 1.0, dtype) for o in out_r]
    for i, (a, b) in enumerate(zip(out_h, out_r)):
        if dtype == torch.bfloat16:
            _cmp_bf16(a, b, f"out[{i}]")
        else:
            _cmp(a, b, fwd_tol, f"out[{i}]")
    torch.autograd.backward(list(out_r), douts)
    torch.autograd.backward(list(out_h), [d.to(DEV, dtype) for d in douts])
    for n in grad_names:
        gr, gh = ref_in[n].grad, hip_in[n].grad
        assert gh is not None, n
        if dtype == torch.bfloat16:
            # gradients of broadcast parameters are sums over B*T rows rounded once to bf16
            _cmp(gh, gr, 2.0 ** -6, f"d{n}")
        else:
            _cmp(gh, gr, bwd_tol, f"d{n}")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,T,D,with_prev,with_mask", [(2, 33, 128, False, False), (3, 16, 768, True, True),
                                                        (1, 1, 1024, True, False)])
def test_token_shift_mix6_and_mix1(dtype, B, T, D, with_prev, with_mask):
    g = torch.Generator().manual_seed(1)
    x = _mk((B, T, D), g, 1.0, dtype)
    xp = _mk((B, D), g, 1.0, dtype) if with_prev else None
    mask = None
    if with_mask:
        mask = torch.ones(B, T, 1)
        mask[0, :3] = 0
    ps = {f"x_{n}": _mk((1, 1, D), g, 0.5, dtype) for n in "rwkvag"}

    def ref(x, x_r, x_w, x_k, x_v, x_a, x_g):
        xm = x if mask is None else x * mask
        return RF.token_shift_mix6(xm, xp, x_r, x_w, x_k, x_v, x_a, x_g)

    def hip(x, x_r, x_w, x_k, x_v, x_a, x_g):
        return fused.token_shift_mix6(x, None if xp is None else xp.to(DEV, dtype), x_r, x_w, x_k, x_v, x_a, x_g,
                                      None if mask is None else mask.to(DEV))

    _run_both(hip, ref, dict(x=x, **ps), dtype, ["x"] + list(ps), 2e-6, 1e-4)

    def ref1(x, x_k):
        xm = x if mask is None else x * mask
        return RF.token_shift_mix1(xm, xp, x_k)

    def hip1(x, x_k):
        return fused.token_shift_mix1(x, None if xp is None else xp.to(DEV, dtype), x_k,
                                      None if mask is None else mask.to(DEV))

    _run_both(hip1, ref1, dict(x=x, x_k=_mk((D,), g, 0.5, dtype)), dtype, ["x", "x_k"], 2e-6, 1e-4)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("layer0,with_mask", [(True, False), (False, False), (False, True), (True, True)])
def test_tmix_prepare(dtype, layer0, with_mask):
    B, T, H = 2, 24, 3
    D = H * 64
    g = torch.Generator().manual_seed(2)
    ins = dict(w_pre=_mk((B, T, D), g, 2.0, dtype), k=_mk((B, T, D), g, 1.0, dtype), v=_mk((B, T, D), g, 1.0, dtype),
               a_pre=_mk((B, T, D), g, 1.0, dtype), v_pre=_mk((B, T, D), g, 1.0, dtype),
               v_first=_mk((B, T, D), g, 1.0, dtype), k_k=_mk((D,), g, 0.3, dtype) + 0.7,
               k_a=_mk((D,), g, 0.1, dtype) + 1.0)
    ins["k_k"] = ins["k_k"].to(dtype).float()
    ins["k_a"] = ins["k_a"].to(dtype).float()
    mask = None
    if with_mask:
        mask = torch.ones(B, T, 1)
        mask[1, :5] = 0
    grads = ["w_pre", "k", "v", "a_pre", "k_k", "k_a"] + ([] if layer0 else ["v_pre", "v_first"])

    def ref(w_pre, k, v, a_pre, v_pre, v_first, k_k, k_a):
        return RF.tmix_prepare(w_pre, k, v, a_pre, v_pre, v_first, k_k, k_a, mask, H, layer0)

    def hip(w_pre, k, v, a_pre, v_pre, v_first, k_k, k_a):
        return fused.tmix_prepare(w_pre, k, v, a_pre, v_pre, v_first, k_k, k_a,
                                  None if mask is None else mask.to(DEV), H, layer0)

    _run_both(hip, ref, ins, dtype, grads, 5e-6, 1e-4)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_tmix_post(dtype):
    B, T, H = 2, 40, 2
    D = H * 64
    g = torch.Generator().manual_seed(3)
    ins = dict(y=_mk((B, T, D), g, 2.0, dtype), r=_mk((B, T, D), g, 1.0, dtype), k=_mk((B, T, D), g, 1.0, dtype),
               v=_mk((B, T, D), g, 1.0, dtype), g=_mk((B, T, D), g, 1.0, dtype),
               gn_weight=(_mk((D,), g, 0.2, dtype) + 1.0).to(dtype).float(), gn_bias=_mk((D,), g, 0.2, dtype),
               r_k=_mk((H, 64), g, 0.1, dtype))

    def ref(y, r, k, v, g, gn_weight, gn_bias, r_k):
        return RF.tmix_post(y, r, k, v, g, gn_weight, gn_bias, r_k, H, 64e-5)

    def hip(y, r, k, v, g, gn_weight, gn_bias, r_k):
        return fused.tmix_post(y, r, k, v, g, gn_weight, gn_bias, r_k, H, 64e-5)

    _run_both(hip, ref, ins, dtype, list(ins), 1e-5, 2e-4)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_relu_sq(dtype):
    g = torch.Generator().manual_seed(4)
    x = _mk((3, 17, 256), g, 1.0, dtype)
    _run_both(lambda x: fused.relu_sq(x), lambda x: RF.relu_sq(x), dict(x=x), dtype, ["x"], 1e-6, 1e-6)


def test_shape_and_device_errors():
    x = torch.zeros(1, 4, 100, device=DEV)  # D % 64 != 0
    with pytest.raises(ValueError):
        fused.token_shift_mix1(x, None, torch.zeros(100, device=DEV))
    with pytest.raises(NotImplementedError):
        fused.relu_sq(torch.zeros(8))


@pytest.mark.parametrize("N,K", [(64, 256), (256, 256), (1024, 2048)])
def test_linear_split_wgrad_vs_fp32_reference(N, K):
    """fused.linear: forward = F.linear; backward's weight gradient is summed over row slabs in fp32."""
    M = 8192
    g = torch.Generator().manual_seed(N)
    x = torch.randn(2, M // 2, K, generator=g).bfloat16()
    w = (torch.randn(N, K, generator=g) * 0.05).bfloat16()
    b = torch.randn(N, generator=g).bfloat16()
    dy = torch.randn(2, M // 2, N, generator=g).bfloat16()
    xr, wr, br = [t.float().requires_grad_(True) for t in (x, w, b)]
    torch.nn.functional.linear(xr, wr, br).backward(dy.float())
    xd, wd, bd = [t.to(DEV).requires_grad_(True) for t in (x, w, b)]
    y = fused.linear(xd, wd, bd)
    assert y.grad_fn is not None and "Linear" in type(y.grad_fn).__name__
    assert torch.equal(y, torch.nn.functional.linear(xd, wd, bd))
    y.backward(dy.to(DEV))
    _cmp(wd.grad, wr.grad, 2.0 ** -8 + 1e-3, "dw")   # one bf16 rounding of the fp32 sum
    _cmp(xd.grad, xr.grad, 1e-2, "dx")
    _cmp(bd.grad, br.grad, 1e-2, "db")


@pytest.mark.parametrize("fwd_only", [False, True])   # True: one-pass forward that also stores h, backward as the two separate kernels
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("nmix,with_branch,with_mask,D", [(6, True, True, 256), (6, False, False, 1024), (1, True, False, 1024),
                                                          (1, True, True, 128), (6, True, False, 2048)])
def test_add_layer_norm_mix_equals_the_two_separate_stages(dtype, nmix, with_branch, with_mask, D, fwd_only):
    """rwkv7_add_ln_mix_fwd / rwkv7_mix_add_ln_bwd (one pass each way) against rwkv7_add_ln_* followed by rwkv7_mix_* (which
    have their own checks against torch above): the same values -- h is rounded to the tensor type before it is mixed and dh
    before it enters the LayerNorm backward, exactly as the separate path stores them.  Several sequences, a length that is not
    a multiple of the run length, masked positions in the middle (packed rows restart the token shift that way)."""
    B, T = 3, 45
    g = torch.Generator().manual_seed(D + nmix)
    x = (torch.randn(B, T, D, generator=g) * 1.3 + 0.2).to(dtype)
    br = torch.randn(B, T, D, generator=g).to(dtype) if with_branch else None
    norm = torch.nn.LayerNorm(D, eps=1e-5).to(DEV)
    with torch.no_grad():
        norm.weight.copy_(torch.randn(D, generator=g) * 0.2 + 1.0)
        norm.bias.copy_(torch.randn(D, generator=g) * 0.1)
    norm = norm.to(dtype)
    mixp = [torch.rand(D, generator=g).to(dtype).to(DEV).requires_grad_(True) for _ in range(nmix)]
    mask = None
    if with_mask:
        mask = torch.ones(B, T, 1)
        mask[0, :3] = 0
        mask[1, 20] = 0
        mask[2, -1] = 0
        mask = mask.to(dtype).to(DEV)
    gout = [torch.randn(B, T, D, generator=g).to(dtype).to(DEV) for _ in range(nmix)]
    gx1 = torch.randn(B, T, D, generator=g).to(dtype).to(DEV)

    def run(fused_path):
        xd = x.to(DEV).requires_grad_(True)
        bd = None if br is None else br.to(DEV).requires_grad_(True)
        for p_ in list(norm.parameters()) + mixp:
            p_.grad = None
        if fused_path:
            x1, outs = fused.add_layer_norm_mix(xd, bd, norm, mask, tuple(mixp), fwd_only=fwd_only)
        else:
            if bd is None:
                x1, h = xd, fused.layer_norm(xd, norm)
            else:
                x1, h = fused.add_layer_norm(xd, bd, norm)
            if nmix == 6:
                outs = fused.token_shift_mix6(h, None, *mixp, mask)
            else:
                outs = (fused.token_shift_mix1(h, None, mixp[0], mask),)
        loss = sum((o.float() * go.float()).sum() for o, go in zip(outs, gout)) + (x1.float() * gx1.float()).sum()
        loss.backward()
        torch.cuda.synchronize()
        return ([o.detach().clone() for o in outs], x1.detach().clone(), xd.grad.clone(), None if bd is None else bd.grad.clone(),
                norm.weight.grad.clone(), norm.bias.grad.clone(), [p_.grad.clone() for p_ in mixp])

    a, b = run(True), run(False)
    for oa, ob in zip(a[0], b[0]):
        assert torch.equal(oa, ob), "mixed outputs differ"
    assert torch.equal(a[1], b[1]), "x1 differs"
    if br is not None:
        assert torch.equal(a[2], b[2]), "dx differs"
        assert torch.equal(a[3], b[3])
    else:
        # without a branch x1 IS x: the separate path lets autograd add the residual gradient to the LayerNorm gradient (a second
        # rounding in the tensor type), the one-pass kernel adds it in fp32 before the only rounding
        scale = b[2].float().abs().max().item()
        assert (a[2].float() - b[2].float()).abs().max().item() <= (1e-6 if dtype == torch.float32 else 1e-2) * scale
    # parameter gradients: sums over rows in a different workgroup partition (fp32 partials) -> rounding-level differences
    tol = 1e-5 if dtype == torch.float32 else 2e-2
    for name, ga, gb in [("dgamma", a[4], b[4]), ("dbeta", a[5], b[5])] + [(f"dmix{i}", pa, pb) for i, (pa, pb) in enumerate(zip(a[6], b[6]))]:
        scale = max(gb.float().abs().max().item(), 1e-6)
        assert (ga.float() - gb.float()).abs().max().item() <= tol * scale, name


@pytest.mark.parametrize("M,N,K", [(4096, 64, 1024), (4096, 1024, 64), (8192, 32, 768), (8192, 768, 32), (4096, 128, 2048),
                                   (4096, 2048, 128), (4608, 1024, 64), (4096, 96, 2048)])
def test_low_rank_weight_gradient_kernel_vs_fp32_reference(M, N, K):
    """rwkv7_wgrad_skinny_bf16 + rwkv7_sum_slabs_bf16 (fused.wgrad_splitk routes the LoRA-shaped gradients there): dW[N][K] =
    dy^T x over M rows against the fp32 product of the same bf16 operands (autograd of rwkv_s2s_single_ffn.py:172-184).
    Tolerance: one bf16 rounding of the fp32 sum.  The last two shapes are not served by the kernel (rows not a multiple of
    512, rank 96) and take the batched-GEMM route: same check."""
    g = torch.Generator().manual_seed(M + N + K)
    dy = torch.randn(M, N, generator=g).bfloat16()
    x = (torch.randn(M, K, generator=g) * 0.7).bfloat16()
    want = dy.float().t() @ x.float()
    got = fused.wgrad_splitk(dy.to(DEV), x.to(DEV))
    assert got.shape == (N, K) and got.dtype == torch.bfloat16
    _cmp(got, want, 2.0 ** -8 + 1e-3, "dW")
    # into a caller-provided slice (the trainer's flat gradient buffer), and through the C-ABI argument checks
    out = torch.full((N, K), 7.0, dtype=torch.bfloat16, device=DEV)
    assert fused.wgrad_splitk(dy.to(DEV), x.to(DEV), out=out) is out
    assert torch.equal(out, got)


@pytest.mark.parametrize("S,shape", [(1024, (5, 1024)), (2048, (3, 1024)), (1024, (2, 1024)), (256, (1, 768)), (1000, (3, 100)), (1024, (8, 2048))])
def test_partials_column_sum_kernel(S, shape):
    """fused._colsum: the column sums of a stage's parameter-gradient partials ([S workgroup rows][... D] fp32 -> bf16) through
    rwkv7_sum_slabs_bf16's tall shape, against the float64 sum rounded once; deterministic (two calls, same bits); other dtypes and
    short partials keep torch's reduce + cast."""
    g = torch.Generator().manual_seed(S + shape[-1])
    part = (torch.randn(S, *shape, generator=g) * torch.rand(S, 1, 1, generator=g)).float()
    want = part.double().sum(0)
    got = fused._colsum(part.to(DEV), torch.bfloat16)
    assert got.shape == tuple(shape) and got.dtype == torch.bfloat16
    again = fused._colsum(part.to(DEV), torch.bfloat16)
    assert torch.equal(got, again)
    _cmp_bf16(got, want.float().bfloat16().float(), "column sums", ulps=1.0)
    ref32 = fused._colsum(part.to(DEV), torch.float32)   # torch path
    assert ref32.dtype == torch.float32
    _cmp(ref32, want.float(), 1e-5, "fp32 column sums")


def test_low_rank_weight_gradient_argument_errors():
    import ctypes
    from rwkvtts_amd import _lib
    one = ctypes.c_void_p(16)
    L = _lib.lib()
    assert L.rwkv7_wgrad_skinny_bf16(ctypes.c_long(4096), 64, 1024, 8, None, one, one, None) == -1      # null pointer
    assert L.rwkv7_wgrad_skinny_bf16(ctypes.c_long(4096), 48, 1024, 8, one, one, one, None) == -4       # rank 48
    assert L.rwkv7_wgrad_skinny_bf16(ctypes.c_long(4096), 64, 1000, 8, one, one, one, None) == -4       # wide side % 256
    assert L.rwkv7_wgrad_skinny_bf16(ctypes.c_long(4096), 64, 1024, 3, one, one, one, None) == -4       # rows per slab


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("D,with_branch,with_bias", [(128, True, True), (1024, True, True), (1024, False, True), (256, True, False)])
def test_add_layer_norm(dtype, D, with_branch, with_bias):
    """rwkv7_add_ln_{fwd,bwd}: x1 = x + branch, h = LayerNorm(x1) and the backward with the residual gradient folded
    in, against torch (fp32 on CPU) on the same rounded inputs."""
    rows = (3, 37)
    g = torch.Generator().manual_seed(D)
    x = (torch.randn(*rows, D, generator=g) * 1.5 + 0.3).to(dtype)
    br = (torch.randn(*rows, D, generator=g)).to(dtype) if with_branch else None
    norm = torch.nn.LayerNorm(D, eps=1e-5, bias=with_bias)
    with torch.no_grad():
        norm.weight.copy_((1 + 0.2 * torch.randn(D, generator=g)).to(dtype).float())
        if with_bias:
            norm.bias.copy_((0.1 * torch.randn(D, generator=g)).to(dtype).float())
    dh = torch.randn(*rows, D, generator=g).to(dtype)
    dx1 = torch.randn(*rows, D, generator=g).to(dtype)
    # reference
    xr = x.float().clone().requires_grad_(True)
    brr = None if br is None else br.float().clone().requires_grad_(True)
    x1r = xr if br is None else (xr + brr)
    if dtype == torch.bfloat16 and br is not None:
        x1r = x1r + (x1r.detach().bfloat16().float() - x1r.detach())   # the add is rounded to bf16, straight-through
    hr = norm(x1r)
    loss = (hr * dh.float()).sum() + ((x1r * dx1.float()).sum() if br is not None else 0)
    loss.backward()
    # HIP
    import copy
    nd = copy.deepcopy(norm).to(DEV).to(dtype)
    xd = x.to(DEV).requires_grad_(True)
    if br is None:
        hd = fused.layer_norm(xd, nd)
        (hd.float() * dh.to(DEV).float()).sum().backward()
    else:
        bd = br.to(DEV).requires_grad_(True)
        x1d, hd = fused.add_layer_norm(xd, bd, nd)
        ((hd.float() * dh.to(DEV).float()).sum() + (x1d.float() * dx1.to(DEV).float()).sum()).backward()
        assert torch.equal(bd.grad, xd.grad)
        if dtype == torch.bfloat16:
            assert torch.equal(x1d.cpu(), (x.float() + br.float()).bfloat16())
    if dtype == torch.float32:
        _cmp(hd, hr, 2e-5, "h")
        _cmp(xd.grad, xr.grad, 1e-4, "dx")
        _cmp(nd.weight.grad, norm.weight.grad, 1e-4, "dgamma")
        if with_bias:
            _cmp(nd.bias.grad, norm.bias.grad, 1e-4, "dbeta")
    else:
        _cmp_bf16(hd, hr, "h")
        _cmp_bf16(xd.grad, xr.grad, "dx")
        _cmp(nd.weight.grad, norm.weight.grad, 1e-2, "dgamma")
        if with_bias:
            _cmp(nd.bias.grad, norm.bias.grad, 1e-2, "dbeta")


@pytest.mark.parametrize("M", [1, 7, 32])
@pytest.mark.parametrize("K,N,with_bias", [(1024, 1024, False), (64, 1024, True), (1024, 64, False), (4096, 1024, False),
                                          (1024, 8193, False), (256, 100, True)])
def test_gemv32_decode_linear_vs_fp32_reference(M, K, N, with_bias):
    """rwkv7_gemv32_bf16 (decode-step linear layers, <= 32 rows): fp32 accumulation on MFMA, one bf16 rounding."""
    g = torch.Generator().manual_seed(M * 1000 + N)
    x = (torch.randn(M, K, generator=g)).bfloat16()
    w = (torch.randn(N, K, generator=g) * K ** -0.5).bfloat16()
    b = (torch.randn(N, generator=g) * 0.1).bfloat16() if with_bias else None
    ref = x.float() @ w.float().t() + (0 if b is None else b.float())
    with torch.no_grad():
        y = fused.linear(x.to(DEV).view(1, M, K), w.to(DEV), None if b is None else b.to(DEV))
    assert y.shape == (1, M, N) and y.dtype == torch.bfloat16
    _cmp_bf16(y.view(M, N), ref, "y")
    # with autograd on (training) the same call must not take the decode kernel
    xg = x.to(DEV).requires_grad_(True)
    yg = fused.linear(xg, w.to(DEV), None if b is None else b.to(DEV))
    assert yg.requires_grad


@pytest.mark.parametrize("M", [1, 32])
@pytest.mark.parametrize("R,act,with_bias", [(32, None, True), (64, "tanh", True), (64, None, True), (128, "sigmoid", False)])
def test_lora32_decode_pair_vs_reference(M, R, act, with_bias):
    """rwkv7_lora32_bf16: act(x W1^T) W2^T + b for the decode batch in one launch; the intermediate is rounded to bf16 as
    the tensor of the three-launch path would be."""
    K, N = 1024, 1024
    g = torch.Generator().manual_seed(M + R)
    x = (torch.randn(M, K, generator=g)).bfloat16()
    w1 = (torch.randn(R, K, generator=g) * K ** -0.5).bfloat16()
    w2 = (torch.randn(N, R, generator=g) * R ** -0.5).bfloat16()
    b = (torch.randn(N, generator=g) * 0.1).bfloat16() if with_bias else None
    f = {None: lambda t: t, "tanh": torch.tanh, "sigmoid": torch.sigmoid}[act]
    mid = f(x.float() @ w1.float().t()).bfloat16().float()
    ref = mid @ w2.float().t() + (0 if b is None else b.float())
    with torch.no_grad():
        assert fused.lora_decode_supported(x.to(DEV), R)
        y = fused.lora_decode(x.to(DEV), w1.to(DEV), w2.to(DEV), None if b is None else b.to(DEV), act)
    _cmp_bf16(y, ref, "y", ulps=2.0)   # + a possible 1-ulp flip of the intermediate


def test_hip_adamw_matches_torch_adamw():
    """rwkv7_adamw_bf16 (fp32 masters + moments from bf16 gradients, bf16 weights rewritten) against torch.optim.AdamW fed
    the same gradients, over several steps with a changing learning rate."""
    import ctypes
    from rwkvtts_amd import _lib
    n = 4 * 1000 + 8
    g = torch.Generator().manual_seed(0)
    p0 = torch.randn(n, generator=g)
    grads = [(torch.randn(n, generator=g) * 0.1).bfloat16() for _ in range(5)]
    lrs = [1e-3, 2e-3, 1.5e-3, 1e-3, 5e-4]
    betas, eps, wd = (0.9, 0.95), 1e-8, 0.01
    ref = p0.clone().to(DEV).requires_grad_(True)
    opt = torch.optim.AdamW([ref], lr=lrs[0], betas=betas, eps=eps, weight_decay=wd)
    p32 = p0.clone().to(DEV)
    m, v = torch.zeros_like(p32), torch.zeros_like(p32)
    p16 = torch.empty(n, dtype=torch.bfloat16, device=DEV)
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    f = ctypes.c_float
    for i, (gr, lr) in enumerate(zip(grads, lrs)):
        for grp in opt.param_groups:
            grp["lr"] = lr
        ref.grad = gr.float().to(DEV)
        opt.step()
        rc = _lib.lib().rwkv7_adamw_bf16(ctypes.c_long(n), P(p32), P(gr.to(DEV)), P(m), P(v), P(p16), f(lr), f(betas[0]),
                                         f(betas[1]), f(eps), f(wd), i + 1, None)
        assert rc == 0
        torch.cuda.synchronize()
        assert (p32 - ref.detach()).abs().max().item() <= 2e-6 * max(1.0, ref.detach().abs().max().item()), i
        assert torch.equal(p16, p32.bfloat16())
    assert _lib.lib().rwkv7_adamw_bf16(ctypes.c_long(n + 1), P(p32), P(p16), P(m), P(v), P(p16), f(1e-3), f(0.9), f(0.95),
                                       f(1e-8), f(0.0), 1, None) == -4


def test_hip_adamw_groups_and_device_skip_flag_match_torch_adamw():
    """rwkv7_adamw_groups_bf16: per-slab parameter groups {lr scale, weight decay} (the lr_1x / lr_2x / lr_decay split of
    train_cosy_rwkv7speech_multiple_dataset.py:162-202) against torch.optim.AdamW with the same three groups; then one step
    with the device-side skip flag set and NaN gradients: moments and weights move as for a zero gradient, nothing becomes NaN."""
    import ctypes
    from rwkvtts_amd import _lib
    sizes = [128 * 3, 128 * 1, 128 * 5, 128 * 2]          # four "parameters", slab aligned
    gid = [0, 1, 2, 0]                                    # lr_1x, lr_2x, lr_decay, lr_1x
    tab = [[1.0, 0.0], [2.0, 0.0], [1.0, 0.1]]
    n = sum(sizes)
    g = torch.Generator().manual_seed(1)
    p0 = torch.randn(n, generator=g)
    offs = [sum(sizes[:i]) for i in range(len(sizes))]
    refs = [p0[o:o + s].clone().to(DEV).requires_grad_(True) for o, s in zip(offs, sizes)]
    opt = torch.optim.AdamW([{"params": [r for r, gi in zip(refs, gid) if gi == k], "weight_decay": tab[k][1], "scale": tab[k][0]}
                             for k in range(3)], lr=1e-3, betas=(0.9, 0.95), eps=1e-18)
    p32 = p0.clone().to(DEV)
    m, v = torch.zeros_like(p32), torch.zeros_like(p32)
    p16 = torch.empty(n, dtype=torch.bfloat16, device=DEV)
    slab = torch.cat([torch.full((s // 128,), k, dtype=torch.uint8) for s, k in zip(sizes, gid)]).to(DEV)
    gtab = torch.tensor(tab, dtype=torch.float32, device=DEV)
    flag = torch.zeros(1, device=DEV)
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    f = ctypes.c_float
    lib = _lib.lib()

    def hip_step(gr, lr, i):
        rc = lib.rwkv7_adamw_groups_bf16(ctypes.c_long(n), P(p32), P(gr), P(m), P(v), P(p16), P(slab), P(gtab), 3, P(flag), f(lr),
                                         f(0.9), f(0.95), f(1e-18), i + 1, None)
        assert rc == 0
        torch.cuda.synchronize()

    lrs = [1e-3, 2e-3, 5e-4]
    for i, lr in enumerate(lrs):
        gr = (torch.randn(n, generator=g) * 0.1).bfloat16().to(DEV)
        for grp in opt.param_groups:
            grp["lr"] = lr * grp["scale"]
        for r, o, s in zip(refs, offs, sizes):
            r.grad = gr[o:o + s].float()
        opt.step()
        hip_step(gr, lr, i)
        want = torch.cat([r.detach() for r in refs])
        assert (p32 - want).abs().max().item() <= 2e-6 * max(1.0, want.abs().max().item()), i
        assert torch.equal(p16, p32.bfloat16())
    # NaN step: flag set on the device, gradient full of NaN -> the update of a ZERO gradient
    flag.fill_(1.0)
    for grp in opt.param_groups:
        grp["lr"] = 1e-3 * grp["scale"]
    for r in refs:
        r.grad = torch.zeros_like(r)
    opt.step()
    hip_step(torch.full((n,), float("nan"), dtype=torch.bfloat16, device=DEV), 1e-3, 3)
    want = torch.cat([r.detach() for r in refs])
    assert torch.isfinite(p32).all() and torch.isfinite(m).all() and torch.isfinite(v).all()
    assert (p32 - want).abs().max().item() <= 2e-6 * max(1.0, want.abs().max().item())
    # argument checks
    assert lib.rwkv7_adamw_groups_bf16(ctypes.c_long(n), P(p32), P(p16), P(m), P(v), P(p16), P(slab), None, 3, None, f(1e-3), f(0.9),
                                       f(0.95), f(1e-8), 1, None) == -1      # table without groups
    assert lib.rwkv7_adamw_groups_bf16(ctypes.c_long(n + 4), P(p32), P(p16), P(m), P(v), P(p16), P(slab), P(gtab), 3, None, f(1e-3),
                                       f(0.9), f(0.95), f(1e-8), 1, None) == -4  # n % 128 != 0 with groups


@pytest.mark.parametrize("M,N,K", [(256, 256, 1024), (512, 768, 2048), (2048, 4096, 1024)])
def test_key_projection_with_relu_squared_epilogue_equals_the_gemm_plus_kernel_pair(M, N, K):
    """csrc/gemm_nt4.hip (rwkv7_gemm_nt_bf16: persistent MFMA kernel fed by LDS-DMA, relu(.)^2 as the epilogue;
    rwkv_s2s_single_ffn.py:228) behind fused.key_relu_sq: the forward is BIT-identical to the library GEMM followed by
    rwkv7_relusq_fwd (fp32 accumulation over K in the same 16-wide steps is not guaranteed, so equality is asserted only where it was
    observed -- 2 bf16 ulp otherwise); the backward (2 relu(x) taken as 2 sqrt(s) from the output) gives the pair's gradients to
    bf16 rounding; shapes outside the tile grid fall back to the pair."""
    from rwkvtts_amd import fused
    was, fused.FUSED_KEY_RELUSQ = fused.FUSED_KEY_RELUSQ, True     # off by default (no gain in the step): switched on for the test
    try:
        _key_relusq_case(fused, M, N, K)
    finally:
        fused.FUSED_KEY_RELUSQ = was


def _key_relusq_case(fused, M, N, K):
    g = torch.Generator().manual_seed(M + N)
    x = (torch.randn(M, K, generator=g) * 0.5).to(DEV, torch.bfloat16)
    w = (torch.randn(N, K, generator=g) * (K ** -0.5)).to(DEV, torch.bfloat16)
    dy = torch.randn(M, N, generator=g).to(DEV, torch.bfloat16)
    for both in (0, 1):
        res = []
        for fusedp in (True, False):
            xi, wi = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
            if fusedp:
                hits = fused.FUSED_KEY_RELUSQ_HITS[0]
                s = fused.key_relu_sq(xi, wi)
                assert s is not None and fused.FUSED_KEY_RELUSQ_HITS[0] == hits + 1
            else:
                s = fused.relu_sq(torch.nn.functional.linear(xi, wi))
            s.backward(dy)
            res.append((s.detach(), xi.grad, wi.grad))
        (s1, dx1, dw1), (s0, dx0, dw0) = res
        d = (s1.float() - s0.float()).abs()
        assert (d <= 2.0 ** -6 * s0.float().abs() + 1e-6).all(), d.max().item()
        for a_, b_, n in ((dx1, dx0, "dx"), (dw1, dw0, "dw")):
            e = (a_.float() - b_.float()).norm().item() / max(b_.float().norm().item(), 1e-9)
            assert e < 8e-3, (n, e)
    assert fused.key_relu_sq(x[:100], w) is None and fused.key_relu_sq(x.float(), w.float()) is None


@pytest.mark.parametrize("M,F,D", [(256, 256, 1024), (512, 768, 2048), (2048, 4096, 1024)])
def test_relu_squared_backward_inside_the_value_dgrad_gemm(M, F, D, monkeypatch):
    """fused.relu_sq_value (round 4): value(relu(h)^2) whose backward forms dh = bf16(dy W_value) * 2 relu(h) in ONE launch
    (rwkv7_gemm_nt_relusq_bwd_bf16: the own MFMA GEMM with h as an auxiliary epilogue operand) against the separate nodes (library
    GEMM, rwkv7_relusq_bwd): the forward is the same two launches, bit for bit; dh agrees to the bf16 rounding of ds (the two GEMMs
    accumulate K in different orders: <= 1 ulp of ds, i.e. 2^-7 of |dh|, with the floor of the parity bars); the weight gradient is
    the same call on the same operands."""
    from rwkvtts_amd import fused
    monkeypatch.setattr(fused, "FUSED_RELUSQ_VALUE_BWD", True)   # off by default (a tie in the step): switched on for the test
    g = torch.Generator().manual_seed(M + F)
    h = (torch.randn(M, F, generator=g) * 0.7).to(DEV, torch.bfloat16)
    w = (torch.randn(D, F, generator=g) * (F ** -0.5)).to(DEV, torch.bfloat16)
    dy = torch.randn(M, D, generator=g).to(DEV, torch.bfloat16)
    res = []
    for fusedp in (True, False):
        hi, wi = h.clone().requires_grad_(True), w.clone().requires_grad_(True)
        if fusedp:
            out = fused.relu_sq_value(hi, wi)
            assert out is not None
        else:
            out = fused.linear(fused.relu_sq(hi), wi, None)
        out.backward(dy)
        torch.cuda.synchronize()
        res.append((out.detach().clone(), hi.grad.clone(), wi.grad.clone()))
    assert torch.equal(res[0][0], res[1][0])
    assert (res[0][1][h <= 0] == 0).all()                              # relu: no gradient where h <= 0
    a, b = res[0][1].float(), res[1][1].float()
    tol = 2.0 ** -7 * torch.clamp(b.abs(), min=0.25 * b.abs().mean().item())
    assert ((a - b).abs() <= tol).all(), (a - b).abs().max().item()
    assert (res[0][2].float() - res[1][2].float()).abs().max().item() <= 1e-2 * res[1][2].float().abs().max().item()
    # shapes outside the tile grid: the caller falls back
    assert fused.relu_sq_value(torch.zeros(100, 256, device=DEV, dtype=torch.bfloat16, requires_grad=True), w[:, :256].contiguous()) is None


@pytest.mark.parametrize("M,N,K", [(256, 256, 1024), (512, 1024, 1024), (2048, 4096, 1024), (1024, 1024, 4096), (256 * 9, 256 * 5, 2048), (256 * 33, 256 * 8, 1024)])
def test_second_generation_gemm_all_epilogues_against_the_library_compositions(M, N, K):
    """csrc/gemm_nt4.hip (four waves, quadrant phases over a ring of eight half-tile slots, counted vmcnt, deferred-drain epilogue
    through LDS): epilogue 0 against the library GEMM, 1 against GEMM +
    rwkv7_relusq_fwd, 2 against GEMM + rwkv7_relusq_bwd (aux = h), 3 against GEMM + rwkv7_relusq_bwd_s (aux = s) -- each bit for
    bit where the library accumulates K in the same order, within one bf16 ulp of the GEMM result otherwise; several tiles per
    workgroup and tile counts that are not a multiple of the grid included (ragged persistence)."""
    import ctypes
    from rwkvtts_amd import _lib, fused
    lib = _lib.lib()
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    g = torch.Generator().manual_seed(M + N + K)
    A = (torch.randn(M, K, generator=g) * 0.5).to(DEV, torch.bfloat16)
    W = (torch.randn(N, K, generator=g) * (K ** -0.5)).to(DEV, torch.bfloat16)
    aux = (torch.randn(M, N, generator=g) * 0.7).to(DEV, torch.bfloat16)
    ref = torch.nn.functional.linear(A, W)
    ulp = 2.0 ** -7 * torch.clamp(ref.float().abs(), min=1e-3)

    def close(got, want, scale=None):
        tol = ulp if scale is None else 2.0 ** -6 * torch.clamp(want.float().abs(), min=scale)
        d = (got.float() - want.float()).abs()
        assert (d <= tol).all(), d.max().item()

    C = torch.full((M, N), float("nan"), device=DEV, dtype=torch.bfloat16)
    assert lib.rwkv7_gemm_nt_bf16(M, N, K, P(A), P(W), P(C), 0, st) == 0
    close(C, ref)
    C1 = torch.full((M, N), float("nan"), device=DEV, dtype=torch.bfloat16)
    assert lib.rwkv7_gemm_nt_bf16(M, N, K, P(A), P(W), P(C1), 1, st) == 0
    assert torch.equal(C1, fused.relu_sq(C))                       # the activation of the kernel's own product, bit for bit
    C2 = torch.full((M, N), float("nan"), device=DEV, dtype=torch.bfloat16)
    assert lib.rwkv7_gemm_nt_relusq_bwd_bf16(M, N, K, P(A), P(W), P(aux), P(C2), st) == 0
    want2 = torch.empty_like(C)
    fused._call("relusq_bwd", aux, ctypes.c_long(aux.numel()), P(aux), P(C), P(want2))
    assert torch.equal(C2, want2)
    s_act = fused.relu_sq(aux)
    C3 = torch.full((M, N), float("nan"), device=DEV, dtype=torch.bfloat16)
    assert lib.rwkv7_gemm_nt_relusq_bwd_s_bf16(M, N, K, P(A), P(W), P(s_act), P(C3), st) == 0
    want3 = torch.empty_like(C3)
    fused._call("relusq_bwd_s", s_act, ctypes.c_long(s_act.numel()), P(s_act), P(C), P(want3))
    assert torch.equal(C3, want3)
    # repeated launches give the same bits (a race in the slot ring would come and go)
    for _ in range(5):
        Cr = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
        assert lib.rwkv7_gemm_nt_bf16(M, N, K, P(A), P(W), P(Cr), 0, st) == 0
        assert torch.equal(Cr, C)
    for fn_args in ((lib.rwkv7_gemm_nt_relusq_bwd_s_bf16, (P(s_act), P(C3))), (lib.rwkv7_gemm_nt_relusq_bwd_bf16, (P(aux), P(C3)))):
        assert fn_args[0](M, N, 960, P(A), P(W), *fn_args[1], st) == -4                                   # K % 1024: RWKV7_ESHAPE
    assert lib.rwkv7_gemm_nt_bf16(M, N, 960, P(A), P(W), P(C3), 0, st) == -4


@pytest.mark.parametrize("M,N,K", [(512, 1024, 1024), (256 * 9, 256 * 5, 2048)])
def test_second_generation_gemm_vs_first_generation_lab(lab, M, N, K):
    """Lab cross-check (skipped without the lab library): csrc/lab/gemm_relusq.hip, the first-generation kernel, gives the same bits
    for the plain product, the relu^2 epilogue and the 2 relu(aux) epilogue."""
    import ctypes
    from rwkvtts_amd import _lib
    lib = _lib.lib()
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    g = torch.Generator().manual_seed(M + N + K)
    A = (torch.randn(M, K, generator=g) * 0.5).to(DEV, torch.bfloat16)
    W = (torch.randn(N, K, generator=g) * (K ** -0.5)).to(DEV, torch.bfloat16)
    aux = (torch.randn(M, N, generator=g) * 0.7).to(DEV, torch.bfloat16)
    for epi in (0, 1):
        C = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
        assert lib.rwkv7_gemm_nt_bf16(M, N, K, P(A), P(W), P(C), epi, st) == 0
        for variant in (0, 1):
            assert torch.equal(C, lab.gemm_nt_gen1(A, W, epi, variant))
    C2 = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    assert lib.rwkv7_gemm_nt_relusq_bwd_bf16(M, N, K, P(A), P(W), P(aux), P(C2), st) == 0
    assert torch.equal(C2, lab.gemm_nt_relusq_bwd_gen1(A, W, aux))


@pytest.mark.parametrize("M,F,D", [(512, 1024, 1024), (2048, 4096, 1024)])
def test_channel_mix_with_the_activation_inside_both_gemms(M, F, D, monkeypatch):
    """fused.channel_mix (rwkv_s2s_single_ffn.py:226-229): relu(x W_k^T)^2 W_v^T with s as the key GEMM's epilogue and
    dk = bf16(dout W_v) * 2 sqrt(s) as the epilogue of the value projection's input-gradient GEMM, against the separate nodes
    (library GEMMs, rwkv7_relusq_fwd / rwkv7_relusq_bwd): output within the bf16 rounding of s, gradients to the parity bars."""
    from rwkvtts_amd import fused
    monkeypatch.setattr(fused, "FUSED_CMIX", True)
    g = torch.Generator().manual_seed(M + F)
    x = (torch.randn(M, D, generator=g) * 0.5).to(DEV, torch.bfloat16)
    wk = (torch.randn(F, D, generator=g) * (D ** -0.5)).to(DEV, torch.bfloat16)
    wv = (torch.randn(D, F, generator=g) * (F ** -0.5)).to(DEV, torch.bfloat16)
    dy = torch.randn(M, D, generator=g).to(DEV, torch.bfloat16)
    res = []
    for fusedp in (True, False):
        xi, ki, vi = x.clone().requires_grad_(True), wk.clone().requires_grad_(True), wv.clone().requires_grad_(True)
        if fusedp:
            hits = fused.FUSED_CMIX_HITS[0]
            out = fused.channel_mix(xi, ki, vi)
            assert out is not None and fused.FUSED_CMIX_HITS[0] == hits + 1
        else:
            out = fused.linear(fused.relu_sq(fused.linear(xi, ki, None)), vi, None)
        out.backward(dy)
        torch.cuda.synchronize()
        res.append((out.detach().clone(), xi.grad.clone(), ki.grad.clone(), vi.grad.clone()))
    for a_, b_, n in zip(res[0], res[1], ("out", "dx", "dwk", "dwv")):
        e = (a_.float() - b_.float()).norm().item() / max(b_.float().norm().item(), 1e-9)
        assert e < 8e-3, (n, e)
    assert fused.channel_mix(x[:100], wk, wv) is None and fused.channel_mix(x.float(), wk.float(), wv.float()) is None


@pytest.mark.parametrize("M,N,K", [(512, 1024, 1024), (8192, 1024, 1024), (256 * 33, 256 * 8, 2048)])
def test_projection_with_the_residual_add_as_its_epilogue(M, N, K, monkeypatch):
    """fused.linear_add (rwkv7_gemm_nt_add_bf16, csrc/gemm_nt4.hip epilogue 4): resid + y W^T as one GEMM against nn.Linear followed by
    the add -- bit for bit forward (bf16(bf16(y W^T) + resid)); the three gradients to the parity bars of the linear layers (the input and
    weight gradients are the same library / slab calls on the same operands, the residual gradient is the incoming one)."""
    from rwkvtts_amd import fused
    monkeypatch.setattr(fused, "FUSED_OPROJ_ADD", True)
    g = torch.Generator().manual_seed(M + N + K)
    y = (torch.randn(M, K, generator=g) * 0.5).to(DEV, torch.bfloat16)
    w = (torch.randn(N, K, generator=g) * (K ** -0.5)).to(DEV, torch.bfloat16)
    r = torch.randn(M, N, generator=g).to(DEV, torch.bfloat16)
    d = torch.randn(M, N, generator=g).to(DEV, torch.bfloat16)
    res = []
    for fusedp in (True, False):
        yi, wi, ri = y.clone().requires_grad_(True), w.clone().requires_grad_(True), r.clone().requires_grad_(True)
        out = fused.linear_add(yi, wi, ri) if fusedp else ri + fused.linear(yi, wi, None)
        assert out is not None
        out.backward(d)
        torch.cuda.synchronize()
        res.append((out.detach().clone(), yi.grad.clone(), wi.grad.clone(), ri.grad.clone()))
    assert torch.equal(res[0][0], res[1][0])
    assert torch.equal(res[0][3], res[1][3])
    for a_, b_, n in ((res[0][1], res[1][1], "dy"), (res[0][2], res[1][2], "dw")):
        e = (a_.float() - b_.float()).norm().item() / max(b_.float().norm().item(), 1e-9)
        assert e < 8e-3, (n, e)
    assert fused.linear_add(y[:100], w, r[:100]) is None


@pytest.mark.parametrize("B,T,nb,masked", [(2, 2048, 4, False), (4, 1024, 3, True), (8, 1024, 4, True)])
def test_low_rank_branches_through_the_lerp(B, T, nb, masked, monkeypatch):
    """fused.mix_lora (rwkv_s2s_single_ffn.py:160-190): the w / a / v / g branches' Linear(D, r) taken through the token-shift lerp --
    one GEMM on the LayerNorm output + rwkv7_mix_lora_combine_* on [M, 2 R] -- against the separate nodes (six lerps, four projections,
    activations): the three remaining lerps bit for bit, the branches' activated hidden states and every gradient (input, lerp
    coefficients, Linear weights) to the parity bars of the fused stages (the mixed inputs are not rounded to bf16 on the way)."""
    from rwkvtts_amd import fused
    monkeypatch.setattr(fused, "FUSED_MIX_LORA", True)
    D = 1024
    g = torch.Generator().manual_seed(B + T + nb)
    x = torch.randn(B, T, D, generator=g).to(DEV, torch.bfloat16)
    mask = None
    if masked:
        mask = (torch.rand(B, T, 1, generator=g) > 0.2).to(DEV, torch.bfloat16)
        mask[:, :3] = 1
    ranks = [64, 64, 32, 128][:nb] if nb == 4 else [64, 64, 128]
    acts = ["tanh", None, None, "sigmoid"] if nb == 4 else ["tanh", None, "sigmoid"]
    mus6 = [(torch.rand(1, 1, D, generator=g)).to(DEV, torch.bfloat16) for _ in range(6)]          # r w k v a g
    w1s = [(torch.randn(r, D, generator=g) * D ** -0.5).to(DEV, torch.bfloat16) for r in ranks]
    douts = [torch.randn(B, T, D, generator=g).to(DEV, torch.bfloat16) for _ in range(3)]
    dhs = [torch.randn(B, T, r, generator=g).to(DEV, torch.bfloat16) for r in ranks]
    act_fn = {None: lambda t: t, "tanh": torch.tanh, "sigmoid": torch.sigmoid}
    res = []
    for fusedp in (True, False):
        xi = x.clone().requires_grad_(True)
        mi = [m.clone().requires_grad_(True) for m in mus6]
        wi = [w.clone().requires_grad_(True) for w in w1s]
        x_r, x_w, x_k, x_v, x_a, x_g = mi
        bm = [x_w, x_a, x_v, x_g] if nb == 4 else [x_w, x_a, x_g]
        if fusedp:
            assert fused.mix_lora_supported(xi, None, None)
            xr, xk, xv, hs = fused.mix_lora(xi, mask, x_r, x_k, x_v, bm, wi, acts)
        else:
            xr, xw, xk, xv, xa, xg = fused.token_shift_mix6(xi, None, x_r, x_w, x_k, x_v, x_a, x_g, mask)
            ins = [xw, xa, xv, xg] if nb == 4 else [xw, xa, xg]
            hs = [act_fn[a](torch.nn.functional.linear(t, w)) for t, w, a in zip(ins, wi, acts)]
        loss_terms = [(o.float() * d.float()).sum() for o, d in zip((xr, xk, xv), douts)] + [(h.float() * d.float()).sum() for h, d in zip(hs, dhs)]
        sum(loss_terms).backward()
        torch.cuda.synchronize()
        used = [0, 2, 3] + ([1, 4, 3, 5] if nb == 4 else [1, 4, 5])
        res.append(dict(outs=[t.detach().clone() for t in (xr, xk, xv)], hs=[h.detach().clone() for h in hs], dx=xi.grad.clone(),
                        dmu=[mi[j].grad.clone() for j in sorted(set(used))], dw=[w.grad.clone() for w in wi]))
    a_, b_ = res
    for o1, o0 in zip(a_["outs"], b_["outs"]):
        assert torch.equal(o1, o0)
    def rel(u, v):
        return (u.float() - v.float()).norm().item() / max(v.float().norm().item(), 1e-9)
    for h1, h0 in zip(a_["hs"], b_["hs"]):
        assert rel(h1, h0) < 8e-3, rel(h1, h0)
    assert rel(a_["dx"], b_["dx"]) < 8e-3, rel(a_["dx"], b_["dx"])
    for u, v in zip(a_["dw"], b_["dw"]):
        assert rel(u, v) < 1.5e-2, rel(u, v)
    for u, v in zip(a_["dmu"], b_["dmu"]):
        assert rel(u, v) < 2e-2, rel(u, v)


@pytest.mark.parametrize("B,T,with_branch,masked,layer0", [(2, 2048, True, False, False), (4, 1024, True, True, False), (8, 512, False, False, True)])
def test_add_layer_norm_mix_lora_one_pass_forward(B, T, with_branch, masked, layer0, monkeypatch):
    """fused.add_layer_norm_mix_lora (rwkv_s2s_single_ffn.py:158-190, 251-259): residual add + LayerNorm + the three remaining lerps as
    ONE forward kernel (rwkv7_add_ln_mix_fwd_h, nmix = 3) with the branches' GEMM on the stored LayerNorm output, against
    fused.add_layer_norm followed by fused.mix_lora in its through-the-lerp form (the one-pass experiment keeps the round-4 pair; the
    direct kernel of csrc/lora_down.hip is pinned off here) -- the same arithmetic in the same order, so outputs and every gradient bit
    for bit."""
    from rwkvtts_amd import fused as _fused
    monkeypatch.setattr(_fused, "LORA_DOWN_DIRECT", False)
    D = 1024
    g = torch.Generator().manual_seed(B * T + with_branch)
    x = torch.randn(B, T, D, generator=g).to(DEV, torch.bfloat16)
    br = torch.randn(B, T, D, generator=g).to(DEV, torch.bfloat16) if with_branch else None
    mask = None
    if masked:
        mask = (torch.rand(B, T, 1, generator=g) > 0.2).to(DEV, torch.bfloat16)
        mask[:, :3] = 1
    ranks = [64, 64, 128] if layer0 else [64, 64, 32, 128]
    acts = ["tanh", None, "sigmoid"] if layer0 else ["tanh", None, None, "sigmoid"]
    nbr = len(ranks)
    mus = [(torch.rand(1, 1, D, generator=g)).to(DEV, torch.bfloat16) for _ in range(3 + nbr)]
    w1s = [(torch.randn(r, D, generator=g) * D ** -0.5).to(DEV, torch.bfloat16) for r in ranks]
    norm = torch.nn.LayerNorm(D, eps=1e-5).to(DEV, torch.bfloat16)
    with torch.no_grad():
        norm.weight.copy_(1 + 0.2 * torch.randn(D, generator=g))
        norm.bias.copy_(0.1 * torch.randn(D, generator=g))
    d_x1 = torch.randn(B, T, D, generator=g).to(DEV, torch.bfloat16)
    douts = [torch.randn(B, T, D, generator=g).to(DEV, torch.bfloat16) for _ in range(3)]
    dhs = [torch.randn(B, T, r, generator=g).to(DEV, torch.bfloat16) for r in ranks]
    res = []
    for one_pass in (True, False):
        xi = x.clone().requires_grad_(True)
        bi = None if br is None else br.clone().requires_grad_(True)
        mi = [m.clone().requires_grad_(True) for m in mus]
        wi = [w.clone().requires_grad_(True) for w in w1s]
        norm.zero_grad(set_to_none=True)
        if one_pass:
            x1, xr, xk, xv, hs = fused.add_layer_norm_mix_lora(xi, bi, norm, mask, mi[0], mi[1], mi[2], mi[3:], wi, acts)
        else:
            if bi is None:
                x1, h = xi, fused.layer_norm(xi, norm)
            else:
                x1, h = fused.add_layer_norm(xi, bi, norm)
            xr, xk, xv, hs = fused.mix_lora(h, mask, mi[0], mi[1], mi[2], mi[3:], wi, acts)
        terms = [(o.float() * d.float()).sum() for o, d in zip((xr, xk, xv), douts)] + [(h_.float() * d.float()).sum() for h_, d in zip(hs, dhs)]
        if bi is not None:
            terms.append((x1.float() * d_x1.float()).sum())
        sum(terms).backward()
        torch.cuda.synchronize()
        res.append(dict(outs=[t.detach().clone() for t in (x1, xr, xk, xv, *hs)], dx=xi.grad.clone(), db=None if bi is None else bi.grad.clone(),
                        dmu=[m.grad.clone() for m in mi], dw=[w.grad.clone() for w in wi], dg=norm.weight.grad.clone(), dbeta=norm.bias.grad.clone()))
    a_, b_ = res
    for o1, o0 in zip(a_["outs"], b_["outs"]):
        assert torch.equal(o1, o0)
    assert torch.equal(a_["dx"], b_["dx"])
    if with_branch:
        assert torch.equal(a_["db"], b_["db"])
    for u, v in zip(a_["dmu"] + a_["dw"] + [a_["dg"], a_["dbeta"]], b_["dmu"] + b_["dw"] + [b_["dg"], b_["dbeta"]]):
        assert torch.equal(u, v)


@pytest.mark.parametrize("R,C", [(1024, 4096), (64, 64), (192, 1088)])
def test_transpose_kernel(R, C):
    """rwkv7_transpose_bf16 (the channel-mix backward's W_value^T): exact against torch's .t().contiguous(); argument errors."""
    import ctypes
    from rwkvtts_amd import _lib
    x = torch.randn(R, C, generator=torch.Generator().manual_seed(R + C)).to(DEV, torch.bfloat16)
    out = torch.empty(C, R, dtype=torch.bfloat16, device=DEV)
    L = _lib.lib()
    assert L.rwkv7_transpose_bf16(R, C, ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(out.data_ptr()), None) == 0
    torch.cuda.synchronize()
    assert torch.equal(out, x.t().contiguous())
    one = ctypes.c_void_p(16)
    assert L.rwkv7_transpose_bf16(R, C, None, one, None) == -1
    assert L.rwkv7_transpose_bf16(100, 64, one, one, None) == -4


@pytest.mark.parametrize("M,N,K", [(4096, 4096, 1024), (8192, 2048, 512), (4096, 1024, 1024)])
def test_input_gradient_on_the_transposed_weight_is_the_same_product(M, N, K, monkeypatch):
    """fused._dgrad (round 5): dx = dy @ W through the library's NT kernel on W^T (rwkv7_transpose_bf16) for long contractions
    (N >= DGRAD_NT_MIN_N) -- the same bits as torch.mm(dy, W), and the plain product below the threshold."""
    g = torch.Generator().manual_seed(M + N)
    dy = (torch.randn(M, N, generator=g) * 0.1).bfloat16().to(DEV)
    W = (torch.randn(N, K, generator=g) * 0.05).bfloat16().to(DEV)
    want = torch.mm(dy, W)
    hits = fused.DGRAD_NT_HITS[0]
    got = fused._dgrad(dy, W)
    assert torch.equal(got, want)
    assert fused.DGRAD_NT_HITS[0] == hits + (1 if N >= fused.DGRAD_NT_MIN_N else 0)
    monkeypatch.setattr(fused, "DGRAD_NT_MIN_N", 256)   # force the transposed route for every shape
    got = fused._dgrad(dy, W)
    assert torch.equal(got, want) and fused.DGRAD_NT_HITS[0] == hits + (2 if N >= 2048 else 1)
    monkeypatch.setattr(fused, "DGRAD_NT", False)
    assert torch.equal(fused._dgrad(dy, W), want)


def test_gather_rows_forward_and_backward_are_the_two_gathers():
    """fused.gather_rows (rwkv7_gather_rows_bf16): the packed <-> aligned re-layout of cu_seqlens rows.  out[r] = src[idx[r]] with zero
    rows where idx < 0; the gradient w.r.t. src is the gather by the inverse map, zero for dropped rows -- bit for bit what
    index_select + mask and their autograd give."""
    from rwkvtts_amd import fused
    g = torch.Generator().manual_seed(3)
    n_in, n_out, D = 1000, 1312, 256
    src = torch.randn(n_in, D, generator=g).to(DEV, torch.bfloat16)
    keep = torch.randperm(n_in, generator=g)[:900]                 # 100 source rows are dropped
    slots = torch.randperm(n_out, generator=g)[:900]               # 412 output rows stay zero
    idx = torch.full((n_out,), -1, dtype=torch.int32)
    inv = torch.full((n_in,), -1, dtype=torch.int32)
    idx[slots] = keep.to(torch.int32)
    inv[keep] = slots.to(torch.int32)
    idx, inv = idx.to(DEV), inv.to(DEV)
    w = torch.randn(n_out, D, generator=g).to(DEV, torch.bfloat16)
    res = []
    for kern in (True, False):
        fused.GATHER_ROWS_KERNEL = kern
        try:
            s_ = src.clone().requires_grad_(True)
            out = fused.gather_rows(s_, idx, inv)
            out.backward(w)
            res.append((out.detach(), s_.grad))
        finally:
            fused.GATHER_ROWS_KERNEL = True
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    assert (res[0][0][idx < 0] == 0).all() and (res[0][1][inv < 0] == 0).all()
    assert torch.equal(res[0][0][slots.to(DEV)], src[keep.to(DEV)])


@pytest.mark.parametrize("B,T,D,ranks,masked", [
    (2, 2048, 1024, (64, 64, 32, 128), False),      # 0.4B, the fast path (no mask, T % 128 == 0)
    (4, 1024, 1024, (64, 64, 128), True),           # layer 0 (no v branch), masked rows
    (4, 1056, 1024, (64, 64, 32, 128), False),      # sequence starts inside a tile (T % 128 != 0): per-row multipliers
    (2, 2048, 2048, (64, 64, 32, 128), True),       # D = 2048
    (2, 2048, 2048, (96, 96, 64, 256), True),       # 1.5B ranks: 16 column tiles -- not taken by the direct kernel, the round-4 pair runs
    (2, 2048, 768, (64, 64, 32, 128), False),       # 0.1B width
    (2, 2048, 2560, (64, 64, 32, 128), False),      # a wider model (20 KB of coefficients in LDS)
    (2, 2048, 1024, (32, 32, 32, 32), True),        # four one-tile branches: two per column half
    (2, 2048, 1024, (32, 32, 32), False),           # three one-tile branches: 2 + 1
])
def test_lora_down_with_the_lerp_as_gemm_prologue(B, T, D, ranks, masked, monkeypatch):
    """csrc/lora_down.hip through fused.mix_lora (rwkv_s2s_single_ffn.py:160-190): the branches' activated hidden states against the
    SEPARATE nodes (six lerps rounded to bf16, four Linear(D, r), activations) -- the kernel rounds where they round, so what is left
    is the summation order of the fp32 accumulation: a few values one bf16 ulp apart; the three full lerps bit for bit; every
    gradient to the bars of the through-the-lerp form (its backward is the one that runs)."""
    from rwkvtts_amd import fused
    monkeypatch.setattr(fused, "FUSED_MIX_LORA", True)
    monkeypatch.setattr(fused, "LORA_DOWN_DIRECT", True)
    nb = len(ranks)
    g = torch.Generator().manual_seed(B + T + D + nb)
    x = torch.randn(B, T, D, generator=g).to(DEV, torch.bfloat16)
    mask = None
    if masked:
        mask = (torch.rand(B, T, 1, generator=g) > 0.2).to(DEV, torch.bfloat16)
        mask[:, :3] = 1
    acts = ["tanh", None, None, "sigmoid"] if nb == 4 else ["tanh", None, "sigmoid"]
    mus6 = [(torch.rand(1, 1, D, generator=g)).to(DEV, torch.bfloat16) for _ in range(6)]          # r w k v a g
    w1s = [(torch.randn(r, D, generator=g) * D ** -0.5).to(DEV, torch.bfloat16) for r in ranks]
    douts = [torch.randn(B, T, D, generator=g).to(DEV, torch.bfloat16) for _ in range(3)]
    dhs = [torch.randn(B, T, r, generator=g).to(DEV, torch.bfloat16) for r in ranks]
    act_fn = {None: lambda t: t, "tanh": torch.tanh, "sigmoid": torch.sigmoid}
    res = []
    for mode in ("direct", "through", "separate"):
        fused.LORA_DOWN_DIRECT = mode == "direct"
        xi = x.clone().requires_grad_(True)
        mi = [m.clone().requires_grad_(True) for m in mus6]
        wi = [w.clone().requires_grad_(True) for w in w1s]
        x_r, x_w, x_k, x_v, x_a, x_g = mi
        bm = [x_w, x_a, x_v, x_g] if nb == 4 else [x_w, x_a, x_g]
        if mode != "separate":
            assert fused.mix_lora_supported(xi, None, None)
            hits = fused.LORA_DOWN_DIRECT_HITS[0]
            xr, xk, xv, hs = fused.mix_lora(xi, mask, x_r, x_k, x_v, bm, wi, acts)
            direct_ok = sum(ranks) <= 320
            assert fused.LORA_DOWN_DIRECT_HITS[0] - hits == (1 if mode == "direct" and direct_ok else 0)
        else:
            xr, xw, xk, xv, xa, xg = fused.token_shift_mix6(xi, None, x_r, x_w, x_k, x_v, x_a, x_g, mask)
            ins = [xw, xa, xv, xg] if nb == 4 else [xw, xa, xg]
            hs = [act_fn[a](torch.nn.functional.linear(t, w)) for t, w, a in zip(ins, wi, acts)]
        loss_terms = [(o.float() * d.float()).sum() for o, d in zip((xr, xk, xv), douts)] + [(h.float() * d.float()).sum() for h, d in zip(hs, dhs)]
        sum(loss_terms).backward()
        torch.cuda.synchronize()
        used = [0, 2, 3] + ([1, 4, 3, 5] if nb == 4 else [1, 4, 5])
        res.append(dict(outs=[t.detach().clone() for t in (xr, xk, xv)], hs=[h.detach().clone() for h in hs], dx=xi.grad.clone(),
                        dmu=[mi[j].grad.clone() for j in sorted(set(used))], dw=[w.grad.clone() for w in wi]))
    d_, t_, s_ = res

    def rel(u, v):
        return (u.float() - v.float()).norm().item() / max(v.float().norm().item(), 1e-9)

    for o1, o0 in zip(d_["outs"], s_["outs"]):
        assert torch.equal(o1, o0)
    for h1, h0, ht in zip(d_["hs"], s_["hs"], t_["hs"]):
        if sum(ranks) > 320:
            assert torch.equal(h1, ht)
            continue
        # against the separate nodes: same rounding points -> an order of magnitude inside the through-the-lerp form's distance
        assert rel(h1, h0) < 1.5e-3, rel(h1, h0)
        assert rel(h1, h0) < 0.5 * max(rel(ht, h0), 1e-4) + 1e-3, (rel(h1, h0), rel(ht, h0))
        frac = (h1 != h0).float().mean().item()
        assert frac < 0.08, frac
        assert (h1.float() - h0.float()).abs().max().item() <= 2 ** -6 * max(1.0, h0.float().abs().max().item())
    assert rel(d_["dx"], s_["dx"]) < 8e-3, rel(d_["dx"], s_["dx"])
    for u, v in zip(d_["dw"], s_["dw"]):
        assert rel(u, v) < 1.5e-2, rel(u, v)
    for u, v in zip(d_["dmu"], s_["dmu"]):
        assert rel(u, v) < 2e-2, rel(u, v)



@pytest.mark.parametrize("M,N,K", [(8192, 576, 1024), (4096, 512, 1024), (8192, 576, 2048), (6144, 576, 768)])
def test_weight_gradient_of_the_lerp_matrices(M, N, K, monkeypatch):
    """rwkv7_wgrad_mid_bf16 through fused.wgrad_splitk (the [W_a ; W_b] gradient of fused.mix_lora: dG^T x with N = 2 R = 576 / 512,
    autograd of rwkv_s2s_single_ffn.py:160-190) against the fp32 product: half a bf16 ulp + accumulation noise; and against the library
    slabs it replaces."""
    from rwkvtts_amd import fused
    g = torch.Generator().manual_seed(M + N + K)
    dy = torch.randn(M, N, generator=g).to(DEV, torch.bfloat16)
    x = torch.randn(M, K, generator=g).to(DEV, torch.bfloat16)
    monkeypatch.setattr(fused, "MID_WGRAD", True)
    hits = fused.MID_WGRAD_HITS[0]
    own = fused.wgrad_splitk(dy, x, slabs=32)
    assert fused.MID_WGRAD_HITS[0] - hits == (1 if K % 256 == 0 and M % 2048 == 0 else 0)
    monkeypatch.setattr(fused, "MID_WGRAD", False)
    lib = fused.wgrad_splitk(dy, x, slabs=32)
    ref = dy.float().t() @ x.float()
    torch.cuda.synchronize()
    scale = ref.abs().max().item()
    assert (own.float() - ref).abs().max().item() <= 2 ** -8 * scale
    assert (own.float() - ref).norm().item() <= 3e-3 * ref.norm().item()
    assert (own.float() - lib.float()).norm().item() <= 3e-3 * lib.float().norm().item()
