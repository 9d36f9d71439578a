"""CPU: the algebra of the round-3 sequential kernels (csrc/wkv7_chunk_fwd9.hip, wkv7_chunk_bseq.hip) in float64, against the
three-kernel prototype (tests/chunked_proto2.py, itself checked against the scalar C oracle) and against the oracle directly.

  forward   U = (T A~) S + (T A_ak) V ,  S' = g_C (S + B^^T U + K^^T V) ,  Y = Q~ S + A_qb U + A_qk V
  backward  E' = g_C E_{c+1} ,  Z = (T^T B^) E' + (T^T A_qb^T) dY ,  E_c = E' + A~^T Z + Q~^T dY
            and Z is the Z = T^T (A_qb^T dY + B^ E') the per-chunk gradient kernel (wkv7_chunk_bwd9.hip) consumes.
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import chunked_proto2 as P2  # noqa: E402
from rwkvtts_amd.synthetic import make_wkv_inputs  # noqa: E402


def _one_head(T, seed):
    ins = make_wkv_inputs(1, T, 1, seed=seed, dtype=torch.bfloat16)
    one = [t[0, :, 0].double() for t in ins]
    dy = torch.randn(T, 64, generator=torch.Generator().manual_seed(seed + 1)).bfloat16().double()
    return ins, one, dy


def test_factored_forward_equals_prototype_and_oracle(c_oracle):
    T, C = 256, 32
    ins, (w, q, k, v, a, b), dy = _one_head(T, 5)
    y_p, U_p, hs_p, L, Ms = P2.fwd3(w, q, k, v, a, b, C, torch.float64, 0)
    S = torch.zeros(64, 64, dtype=torch.float64)          # H = S^T in R^{K x V}
    ys, us = [], []
    for c, l in enumerate(L):
        V = v[c * C:c * C + C]
        W, Xp = l["Tm"] @ l["At"], l["Tm"] @ l["A_ak"]    # made one chunk ahead, no state in them
        assert (S - hs_p[c]).abs().max() < 1e-9 * max(1.0, hs_p[c].abs().max().item())
        U = W @ S + Xp @ V                                # interval a
        ys.append(l["Qt"] @ S + l["A_qb"] @ U + l["A_qk"] @ V)
        us.append(U)
        S = l["gC"][:, None] * (S + l["Bh"].T @ U + l["Kh"].T @ V)   # interval b
    y, U = torch.cat(ys), torch.cat(us)
    assert (y - y_p).abs().max() < 1e-9 * y_p.abs().max() and (U - U_p).abs().max() < 1e-9 * U_p.abs().max()
    y_o, s_o, sa_o = c_oracle.wkv7_fwd(*[t.float() for t in ins])
    assert (y.float() - y_o[0, :, 0]).abs().max() < 2e-5 * y_o.abs().max()        # the oracle rounds y to bf16-exact fp32 inputs only
    assert (U.float() - sa_o[0, :, 0]).abs().max() < 2e-5 * sa_o.abs().max()


def test_factored_adjoint_recurrence_and_z_equal_prototype_and_oracle_gradients(c_oracle):
    T, C = 256, 32
    ins, (w, q, k, v, a, b), dy = _one_head(T, 9)
    y_p, U_p, hs_p, L, Ms = P2.fwd3(w, q, k, v, a, b, C, torch.float64, 0)
    n = T // C
    # prototype: E_c = M_c^T E_{c+1} + N'_c with M, N' materialised
    E = torch.zeros(64, 64, dtype=torch.float64)
    Es_ref = [None] * n
    for c in range(n - 1, -1, -1):
        Es_ref[c] = E
        dY = dy[c * C:c * C + C]
        Np = L[c]["Qt"].T @ dY + L[c]["W"].T @ (L[c]["A_qb"].T @ dY)
        E = Ms[c].T @ E + Np
    # factored: nothing but per-chunk 32-row factors
    E = torch.zeros(64, 64, dtype=torch.float64)
    Zs = [None] * n
    for c in range(n - 1, -1, -1):
        l, dY = L[c], dy[c * C:c * C + C]
        assert (E - Es_ref[c]).abs().max() < 1e-9 * max(1.0, Es_ref[c].abs().max().item())
        Bpp, Xpp = l["Tm"].T @ l["Bh"], l["Tm"].T @ l["A_qb"].T
        Ep = l["gC"][:, None] * E
        Z = Bpp @ Ep + Xpp @ dY                           # interval a
        # what the gradient kernel used to rebuild: Z = T^T (A_qb^T dY + B^ E')
        Z_ref = l["Tm"].T @ (l["A_qb"].T @ dY + (l["Bh"] * l["gC"]) @ E)
        assert (Z - Z_ref).abs().max() < 1e-9 * max(1.0, Z_ref.abs().max().item())
        Zs[c] = Z
        E = Ep + l["At"].T @ Z + l["Qt"].T @ dY           # interval b
    # and the gradients that come out of (hs, E, Z) are the oracle's
    grads = P2.bwd3(w, q, k, v, a, b, dy, U_p, hs_p, L, Ms, C, torch.float64, 0)
    y_o, s_o, sa_o = c_oracle.wkv7_fwd(*[t.float() for t in ins])
    g_o = c_oracle.wkv7_bwd(*[t.float() for t in ins], dy.float().view(1, T, 1, 64), s_o, sa_o)
    for name, g, go in zip(("dw", "dq", "dk", "dv", "da", "db"), grads, g_o):
        assert (g.float() - go[0, :, 0]).abs().max() < 2e-3 * go.abs().max(), name   # the oracle's backward divides by the decay (fp32)
