"""GPU (-m gpu): the chunked (MFMA) WKV7 kernels against the scalar oracle.

Tolerance: operands are split into two bf16 pieces (about 16 mantissa bits), accumulation is fp32; the CPU
prototype (tests/chunked_proto.py) measures 5e-6..2e-5 relative error for this scheme, so fp32-I/O results must be
within 1e-4 * max|oracle| and bf16-I/O results within 1 bf16 ulp (2 for gradients), like the scalar kernels."""
import pytest
import torch

from rwkvtts_amd import ops
from rwkvtts_amd.synthetic import make_wkv_inputs
from test_wkv7_gpu import _assert_bf16_close, _assert_f32_close, NAMES

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_mfma_fragment_layout_and_transposed_writeback():
    """Asymmetric operands: catches swapped rows/cols in the fragment maps (cdna_hip_programming.md rule 16)."""
    g = torch.Generator().manual_seed(0)
    X = torch.randn(32, 64, generator=g)
    Y = torch.randn(32, 64, generator=g) * torch.linspace(0.5, 2.0, 32)[:, None]
    D, DT = ops.debug_mma32(X.to(DEV), Y.to(DEV))
    want = X.double() @ Y.double().t()
    assert (D.cpu().double() - want).abs().max().item() < 2e-4 * want.abs().max().item()
    assert (DT.cpu().double() - want.t()).abs().max().item() < 2e-4 * want.abs().max().item()


@pytest.mark.parametrize("B,T,H", [(2, 64, 3), (1, 96, 1), (3, 32, 5)])   # one wave per PAIR of chunks: even, odd (3, 15) chunk counts
def test_prep_inverse_matches_torch(B, T, H):
    w, q, k, v, a, b = make_wkv_inputs(B, T, H, 3, torch.float32)
    tinv = ops.wkv7_chunk_prep(w.to(DEV), a.to(DEV), b.to(DEV)).cpu()
    lw = -torch.exp(w.double())
    for bi in range(B):
        for hi in range(H):
            for c in range(T // 32):
                sl = slice(32 * c, 32 * c + 32)
                G = torch.cumsum(lw[bi, sl, hi], 0)
                At = a[bi, sl, hi].double() * torch.exp(G - lw[bi, sl, hi])
                Bh = b[bi, sl, hi].double() * torch.exp(-G)
                A = torch.tril(At @ Bh.t(), -1)
                want = torch.linalg.inv(torch.eye(32, dtype=torch.float64) - A)
                err = (tinv[bi, hi, c].double() - want).abs().max().item()
                assert err < 2e-4 * want.abs().max().item(), (bi, hi, c, err)


@pytest.mark.parametrize("B,T,H,seed", [(1, 32, 1, 0), (2, 64, 3, 1), (2, 512, 12, 2), (1, 1024, 2, 3)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_chunk_forward_vs_oracle(c_oracle, B, T, H, seed, dtype):
    ins = make_wkv_inputs(B, T, H, seed, dtype)
    y_o, s_o, sa_o = c_oracle.wkv7_fwd(*ins)
    y, tinv, sa, hs = ops.wkv7_chunk_forward(*[t.to(DEV) for t in ins])
    torch.cuda.synchronize()
    if dtype == torch.bfloat16:
        _assert_bf16_close(y, y_o, "y")
    else:
        _assert_f32_close(y, y_o, "y", 1e-4)
    _assert_f32_close(sa, sa_o, "sa", 2e-4 if dtype == torch.float32 else 2e-3)
    # hs[c] = state at the start of chunk c, the backward's checkpoint: q15 records (int16 mantissas [value][key] + a scale per
    # (value half, key): 2^-15 of the column maximum); oracle checkpoint after step 32c-1 is [key][value] fp32
    hsf = ops.q15_decode(hs).transpose(-1, -2)
    for c in range(1, T // 32):
        _assert_f32_close(hsf[:, :, c], s_o[:, :, 2 * c - 1], f"hs[{c}]", 3e-4 if dtype == torch.float32 else 2e-3)
    assert hsf[:, :, 0].abs().max().item() == 0.0


def test_chunked_backward_state_recurrence_vs_prototype():
    """wkv7c_bseq (the factored adjoint recurrence E_c = E' + A~^T Z + Q~^T dY, M_c^T / N'_c never formed): the adjoint states E
    against the CPU prototype of the unfactored algebra E_c = M_c^T E_{c+1} + N'_c (tests/chunked_proto2.py, fp32), which itself is
    checked against the scalar oracle; plain rows, and packed rows (a cut inside the row restarts the recurrence from E = 0)."""
    import chunked_proto2 as P2
    B, T, H = 1, 128, 2
    ins = make_wkv_inputs(B, T, H, 21, torch.bfloat16)
    dy = torch.randn(B, T, H, 64, generator=torch.Generator().manual_seed(3)).bfloat16()
    d = [t.to(DEV) for t in ins]
    w, q, k, v, a, b = d
    tinv = ops.wkv7_chunk_prep(w, a, b)
    nc = T // 32
    for cuts in (None, [0, 3, nc]):
        so = None if cuts is None else torch.tensor(cuts, dtype=torch.int32, device=DEV)
        e_f = ops.q15_decode(ops.wkv7_chunk_bwd_seq(w, q, a, b, dy.to(DEV), tinv, so))
        torch.cuda.synchronize()
        segs = [(0, nc)] if cuts is None else list(zip(cuts[:-1], cuts[1:]))
        for h in range(H):
            for (c0, c1) in segs:
                one = [t[0, c0 * 32:c1 * 32, h].float() for t in ins]
                y, U, hs, L, Ms = P2.fwd3(*one, 32, torch.float32, 0)
                dyh = dy[0, c0 * 32:c1 * 32, h].float()
                n = c1 - c0
                Np = [l["Qt"].T @ dyh[c * 32:c * 32 + 32] + l["W"].T @ (l["A_qb"].T @ dyh[c * 32:c * 32 + 32]) for c, l in enumerate(L)]
                E = torch.zeros(64, 64)
                for c in range(n - 1, -1, -1):
                    scale = max(E.abs().max().item(), 1e-3)
                    # e_vk: q15 record of the recurrence's E ([v][k], 2^-15 of each lane's maximum) on top of the fp32-level error
                    assert (e_f[0, h, c0 + c].cpu().t() - E).abs().max() <= 3e-4 * scale, ("E[v][k]", cuts, h, c0 + c)
                    E = Ms[c].T @ E + Np[c]


@pytest.mark.parametrize("B,T,H,seed", [(1, 32, 1, 0), (2, 64, 3, 1), (2, 512, 12, 2)])
def test_chunked_forward_plus_backward_vs_oracle(c_oracle, B, T, H, seed):
    """The all-MFMA training pair: chunked forward (saves tinv, sa, hs) feeding the chunked backward: the six gradients against
    the C oracle, same 2-ulp bf16 bar as the scalar backward kernel."""
    ins = make_wkv_inputs(B, T, H, seed, torch.bfloat16)
    dy = (torch.randn(B, T, H, 64, generator=torch.Generator().manual_seed(seed + 100))).bfloat16()
    y_o, s_o, sa_o = c_oracle.wkv7_fwd(*ins)
    g_o = c_oracle.wkv7_bwd(*ins, dy, s_o, sa_o)
    d = [t.to(DEV) for t in ins]
    y, tinv, sa, hs = ops.wkv7_chunk_forward(*d)
    grads = ops.wkv7_chunk_backward(*d, dy.to(DEV), hs, sa, tinv)
    torch.cuda.synchronize()
    _assert_bf16_close(y, y_o, "y")
    for n, g, go in zip(NAMES, grads, g_o):
        _assert_bf16_close(g, go, n, ulps=2.0)


@pytest.mark.parametrize("B,T,H,seed", [(1, 32, 1, 0), (2, 96, 3, 4), (3, 2080, 2, 5)])   # one chunk; odd chunk counts; 65 chunks x 6 heads = 390:
def test_gradient_kernel_edge_chunk_counts_vs_oracle(c_oracle, B, T, H, seed):                 # not a multiple of the chunks per workgroup
    """wkv7c_bwd_out10 (raw rows by LDS-DMA, swizzled planes, merged prologue / phase A) inside the 2-ulp bar against the C oracle.  Its
    landing area is refilled one chunk ahead: a workgroup's last chunk, a launch with one chunk in all, and a chunk count that does
    not divide by the chunks per workgroup are the edge cases."""
    ins = make_wkv_inputs(B, T, H, seed, torch.bfloat16)
    dy = (torch.randn(B, T, H, 64, generator=torch.Generator().manual_seed(seed + 100))).bfloat16()
    y_o, s_o, sa_o = c_oracle.wkv7_fwd(*ins)
    g_o = c_oracle.wkv7_bwd(*ins, dy, s_o, sa_o)
    d = [t.to(DEV) for t in ins]
    y, tinv, sa, hs = ops.wkv7_chunk_forward(*d)
    out = ops.wkv7_chunk_backward(*d, dy.to(DEV), hs, sa, tinv)
    torch.cuda.synchronize()
    for n, g, go in zip(NAMES, out, g_o):
        _assert_bf16_close(g, go, n, ulps=2.0)


@pytest.mark.parametrize("B,T,H,seed", [(1, 32, 1, 0), (2, 96, 3, 4), (3, 2080, 2, 5)])
def test_gradient_kernel_vs_round4_twin_lab(lab, B, T, H, seed):
    """Lab cross-check (skipped without the lab library): against csrc/lab/wkv7_chunk_bwd9.hip dq, dk, dv, da, db are the SAME BITS (same
    products, same MFMA order) and dw within one bf16 ulp (its epilogue sum is contracted differently under -ffast-math)."""
    d = [t.to(DEV) for t in make_wkv_inputs(B, T, H, seed, torch.bfloat16)]
    dy = (torch.randn(B, T, H, 64, generator=torch.Generator().manual_seed(seed + 100))).bfloat16().to(DEV)
    y, tinv, sa, hs = ops.wkv7_chunk_forward(*d)
    e_vk, z = ops.wkv7_chunk_bwd_seq(d[0], d[1], d[4], d[5], dy, tinv, want_z=True)
    new = ops.wkv7_chunk_backward(*d, dy, hs, sa, tinv)
    old = lab.bwd_out9(*d, dy, hs, sa, z, e_vk)
    torch.cuda.synchronize()
    for n, a, b in zip(NAMES, old, new):
        if n == "dw":
            _assert_bf16_close(a, b.float().cpu(), "dw, bwd_out9 vs bwd_out10", ulps=1.0)
        else:
            assert torch.equal(a, b), n


def test_chunked_pair_against_committed_g1_vectors_2_512_12_64():
    """SURVEY 8(c) G1 (2,512,12,64) from the COMMITTED vectors (tests/golden/wkv7_scan_g1.npz, written by
    oracle/pin_against_reference.py [4] next to the check of the oracle's backward against torch.autograd on every head): y within
    1 bf16 ulp, the six gradients within 2 ulp of the oracle's vectors on three heads, and every gradient tensor within bf16 noise of
    the committed fp32 autograd gradients -- no oracle code runs in this test."""
    from conftest import load_golden
    from test_oracle_golden import _g1_inputs
    g = load_golden("wkv7_scan_g1.npz")
    ins, dy, heads = _g1_inputs(g)
    d = [t.to(DEV) for t in ins]
    y, tinv, sa, hs = ops.wkv7_chunk_forward(*d)
    grads = ops.wkv7_chunk_backward(*d, dy.to(DEV), hs, sa, tinv)
    torch.cuda.synchronize()
    _assert_bf16_close(y[:, :, heads], g["y"], "y")
    for n, gr in zip(NAMES, grads):
        _assert_bf16_close(gr[:, :, heads], g[n], n, ulps=2.0)
        ref = g[f"autograd.{n}"]
        rel = ((gr[:, :, heads].float().cpu() - ref).norm() / ref.norm()).item()
        assert rel < 2e-2, (n, rel)


def test_full_size_config2_chunked_pair_vs_oracle_slices(c_oracle):
    """BASELINE.json configs[1] (B=8, T=4096, H=16, bf16) through the chunked MFMA forward + backward that the training
    step uses: (i) everything finite; (ii) three (batch, head) slices -- first, middle, last workgroups -- equal the
    oracle run on exactly those slices (1 ulp on y, 2 ulp on the gradients); (iii) linearity of the backward in dy:
    grads(dy1 + dy2) == grads(dy1) + grads(dy2) up to bf16 rounding (a size-independent property of the adjoint)."""
    B, T, H = 8, 4096, 16
    ins = make_wkv_inputs(B, T, H, 1234, torch.bfloat16)
    d = [t.to(DEV) for t in ins]
    g = torch.Generator().manual_seed(99)
    dy1 = torch.randn(B, T, H, 64, generator=g).bfloat16()
    dy2 = (torch.randn(B, T, H, 64, generator=g) * 0.5).bfloat16()
    y, tinv, sa, hs = ops.wkv7_chunk_forward(*d)
    g1 = ops.wkv7_chunk_backward(*d, dy1.to(DEV), hs, sa, tinv)
    g2 = ops.wkv7_chunk_backward(*d, dy2.to(DEV), hs, sa, tinv)
    g12 = ops.wkv7_chunk_backward(*d, (dy1.float() + dy2.float()).bfloat16().to(DEV), hs, sa, tinv)
    torch.cuda.synchronize()
    assert torch.isfinite(y.float()).all()
    for n, ga in zip(NAMES, g1):
        assert torch.isfinite(ga.float()).all(), n
    for (bi, hi) in ((0, 0), (5, 11), (7, 15)):
        sl = [t[bi:bi + 1, :, hi:hi + 1].contiguous() for t in ins]
        y_o, s_o, sa_o = c_oracle.wkv7_fwd(*sl)
        g_o = c_oracle.wkv7_bwd(*sl, dy1[bi:bi + 1, :, hi:hi + 1].contiguous(), s_o, sa_o)
        _assert_bf16_close(y[bi:bi + 1, :, hi:hi + 1], y_o, f"y[{bi},{hi}]")
        for n, ga, go in zip(NAMES, g1, g_o):
            _assert_bf16_close(ga[bi:bi + 1, :, hi:hi + 1], go, f"{n}[{bi},{hi}]", ulps=2.0)
    for n, a1, a2, a12 in zip(NAMES, g1, g2, g12):
        want = a1.float() + a2.float()
        err = (a12.float() - want).abs()
        # three bf16 roundings, each relative to its own operand (a1 and a2 may cancel in the sum), + the rounding of
        # dy1 + dy2 propagated through the adjoint
        mag = a1.float().abs() + a2.float().abs()
        tol = 2.0 ** -6 * torch.clamp(mag, min=mag.mean().item())
        frac = (err > tol).float().mean().item()
        assert frac < 1e-3, f"{n}: {frac:.2e} of the elements break linearity in dy"


def test_packed_sequences_equal_separate_sequences_vs_oracle(c_oracle):
    """Packed rows (rwkv7_wkv_chunk_fwd_seq_bf16 / rwkv7_wkv_chunk_bseq_bf16, fla chunk_rwkv7's cu_seqlens): rows whose
    chunks 0, 2, 3 (row 0) and 0, 1 (row 1) start new sequences must give, segment by segment, exactly what the C oracle gives
    for each segment run on its own from the zero state -- outputs and all six gradients, same bars as the plain tests."""
    B, T, H, seed = 2, 160, 3, 7
    nc = T // 32
    ins = make_wkv_inputs(B, T, H, seed, torch.bfloat16)
    dy = (torch.randn(B, T, H, 64, generator=torch.Generator().manual_seed(seed + 100))).bfloat16()
    starts = [[0, 2, 3], [0, 1]]   # chunk indices (32 steps each), 5 chunks per row
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
            _assert_bf16_close(y[b:b + 1, lo:hi], y_o, f"y[{b},{lo}:{hi}]")
            for n, g, go in zip(NAMES, grads, g_o):
                _assert_bf16_close(g[b:b + 1, lo:hi], go, f"{n}[{b},{lo}:{hi}]", ulps=2.0)


def _fwd_cuts(B, T):
    nc = T // 32
    cuts = {0, B * nc}
    for b in range(B):   # every row ends a sequence; rows with more than one chunk are cut once more
        cuts.add(b * nc + nc)
        if nc > 1:
            cuts.add(b * nc + (nc + b) // 2)
    return torch.tensor(sorted(cuts), dtype=torch.int32, device=DEV)


@pytest.mark.parametrize("B,T,H,seed", [(1, 32, 1, 0), (2, 64, 3, 1), (2, 512, 12, 2), (1, 1024, 2, 3)])
def test_eight_wave_forward_kernel_vs_oracle(c_oracle, B, T, H, seed):
    """wkv7_chunk_fwd9.hip (producer / consumer split; W = T A~ and X' = T A_ak made beside the chain, two dependent products per
    chunk; what rwkv7_wkv_chunk_fwd_seq_bf16 launches): y, sa and the chunk states against the C oracle, feeding the chunked backward
    (2-ulp bar on the six gradients)."""
    ins = make_wkv_inputs(B, T, H, seed, torch.bfloat16)
    dy = (torch.randn(B, T, H, 64, generator=torch.Generator().manual_seed(seed + 100))).bfloat16()
    y_o, s_o, sa_o = c_oracle.wkv7_fwd(*ins)
    g_o = c_oracle.wkv7_bwd(*ins, dy, s_o, sa_o)
    d = [t.to(DEV) for t in ins]
    nc = T // 32
    y, tinv, sa, hs = ops.wkv7_chunk_forward(*d)
    grads = ops.wkv7_chunk_backward(*d, dy.to(DEV), hs, sa, tinv)
    torch.cuda.synchronize()
    _assert_bf16_close(y, y_o, "y")
    _assert_f32_close(sa, sa_o, "sa", 2e-3)
    hsf = ops.q15_decode(hs).transpose(-1, -2)
    for c in range(1, nc):
        _assert_f32_close(hsf[:, :, c], s_o[:, :, 2 * c - 1], f"hs[{c}]", 2e-3)
    for n, g, go in zip(NAMES, grads, g_o):
        _assert_bf16_close(g, go, n, ulps=2.0)


@pytest.mark.parametrize("B,T,H,seed", [(1, 32, 1, 0), (2, 64, 3, 1), (2, 512, 12, 2), (1, 1024, 2, 3)])
def test_eight_wave_forward_kernel_vs_four_wave_kernel_lab(lab, B, T, H, seed):
    """Lab cross-check (tools/lab.py, skipped without the lab library): against the bf16 instantiation of the 4-wave kernel the fp32
    outputs agree to rounding (hipcc contracts the split prologue differently: last-ulp differences of the scaled operands; the
    two-product form associates U = T(A~ S + A_ak V) as (T A~) S + (T A_ak) V), also on packed rows."""
    d = [t.to(DEV) for t in make_wkv_inputs(B, T, H, seed, torch.bfloat16)]
    seq_off = _fwd_cuts(B, T)
    for so in (None, seq_off):
        got = ops.wkv7_chunk_forward(*d, seq_off=so)
        ref = lab.chunk_forward4(*d, got[1], seq_off=so)
        torch.cuda.synchronize()
        _assert_bf16_close(got[0], ref[0].cpu(), "y 8 vs 4", ulps=1.0)
        _assert_f32_close(got[2], ref[2].cpu(), "sa 8 vs 4", 1e-4)
        _assert_f32_close(ops.q15_decode(got[3]), ops.q15_decode(ref[3]).cpu(), "hs 8 vs 4", 1e-4)


