"""CPU prototype of the THREE-KERNEL chunked WKV7 (parallel pre -> sequential state -> parallel out), forward and
backward, against the scalar oracle.  Per chunk (C steps), per head, H = S^T in R^{K x V}:
    forward   W = T A~            U0 = T (A_ak V)          U = W H0 + U0
              M = diag(gC)(I + B^^T W)     N = diag(gC)(B^^T U0 + K^^T V)      H1 = M H0 + N        (state kernel)
              Y = Q~ H0 + A_qb U + A_qk V                                                            (out kernel)
    backward  E0 = M^T E1 + N'    N' = Q~^T dY + W^T (A_qb^T dY)                                    (state kernel, reversed)
              gradients from (H0, E1, local quantities) exactly as tools/chunked_proto.py:chunk_bwd
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from oracle import c_oracle  # noqa: E402
from rwkvtts_amd.synthetic import make_wkv_inputs  # noqa: E402
from chunked_proto import bf16_split  # noqa: E402


def local(w, q, k, v, a, b, sl, dt, S):
    lw = -torch.exp(w[sl])
    G = torch.cumsum(lw, 0)
    gam, gam_prev = torch.exp(G), torch.exp(G - lw)
    gC = gam[-1]
    Qt, At, Kh, Bh = q[sl] * gam, a[sl] * gam_prev, k[sl] / gam, b[sl] / gam
    C = Qt.shape[0]
    tril_s = torch.tril(torch.ones(C, C, dtype=dt), -1)
    tril = torch.tril(torch.ones(C, C, dtype=dt))
    A_ab = (S(At) @ S(Bh).T) * tril_s
    A_ak = (S(At) @ S(Kh).T) * tril_s
    A_qb = (S(Qt) @ S(Bh).T) * tril
    A_qk = (S(Qt) @ S(Kh).T) * tril
    Tm = torch.linalg.inv(torch.eye(C, dtype=dt) - A_ab)
    W = S(Tm) @ S(At)
    U0 = S(Tm) @ S(S(A_ak) @ v[sl])
    return dict(lw=lw, gam=gam, gam_prev=gam_prev, gC=gC, Qt=Qt, At=At, Kh=Kh, Bh=Bh, A_ak=A_ak, A_qb=A_qb, A_qk=A_qk,
                Tm=Tm, W=W, U0=U0)


def fwd3(w, q, k, v, a, b, C, dt, ns):
    T = w.shape[0]
    w, q, k, v, a, b = [t.to(dt) for t in (w, q, k, v, a, b)]
    S = lambda x: bf16_split(x, ns)
    n = T // C
    L = [local(w, q, k, v, a, b, slice(c * C, c * C + C), dt, S) for c in range(n)]            # pre (parallel)
    Ms, Ns = [], []
    for c, l in enumerate(L):
        V = v[c * C:c * C + C]
        BhC = l["Bh"] * l["gC"]                   # fold g_C into B^ / K^ (bounded: b * gamma_C / gamma_t)
        KhC = l["Kh"] * l["gC"]
        Ms.append(torch.diag(l["gC"]) + S(BhC).T @ S(l["W"]))
        Ns.append(S(BhC).T @ S(l["U0"]) + S(KhC).T @ V)
    H = torch.zeros(64, 64, dtype=dt)
    hs = []
    for c in range(n):                                                                             # state (sequential)
        hs.append(H)
        H = S(Ms[c]) @ S(H) + Ns[c]
    hs.append(H)
    ys, us = [], []
    for c, l in enumerate(L):                                                                      # out (parallel)
        V = v[c * C:c * C + C]
        U = S(l["W"]) @ S(hs[c]) + l["U0"]
        ys.append(S(l["Qt"]) @ S(hs[c]) + S(l["A_qb"]) @ S(U) + S(l["A_qk"]) @ V)
        us.append(U)
    return torch.cat(ys), torch.cat(us), hs, L, Ms


def bwd3(w, q, k, v, a, b, dy, U, hs, L, Ms, C, dt, ns):
    T = w.shape[0]
    w, q, k, v, a, b, dy, U = [t.to(dt) for t in (w, q, k, v, a, b, dy, U)]
    S = lambda x: bf16_split(x, ns)
    n = T // C
    Np = []
    for c, l in enumerate(L):                                                                      # bwd pre (parallel)
        dY = dy[c * C:c * C + C]
        Np.append(S(l["Qt"]).T @ dY + S(l["W"]).T @ S(S(l["A_qb"]).T @ dY))
    E = torch.zeros(64, 64, dtype=dt)
    Es = [None] * n
    for c in range(n - 1, -1, -1):                                                                 # bwd state (sequential)
        Es[c] = E                                  # E1 of chunk c = dL/dH at its end, from the future
        E = S(Ms[c]).T @ S(E) + Np[c]
    outs = [torch.zeros(T, 64, dtype=dt) for _ in range(6)]
    for c, l in enumerate(L):                                                                      # bwd out (parallel)
        sl = slice(c * C, c * C + C)
        H0, HC, E1 = hs[c], hs[c + 1], Es[c]
        gam, gam_prev, gC = l["gam"], l["gam_prev"], l["gC"]
        Qt, At, Kh, Bh, Tm = l["Qt"], l["At"], l["Kh"], l["Bh"], l["Tm"]
        V, Uc, dY = v[sl], U[sl], dy[sl]
        Z = S(Tm).T @ S(S(l["A_qb"]).T @ dY + S(Bh * gC) @ S(E1))
        dV = S(l["A_qk"]).T @ dY + S(l["A_ak"]).T @ S(Z) + S(Kh * gC) @ S(E1)
        P_vy, P_vz = torch.triu(V @ dY.T), torch.triu(V @ S(Z).T, 1)
        P_uy, P_uz = torch.triu(S(Uc) @ dY.T), torch.triu(S(Uc) @ S(Z).T, 1)
        dK = (S(P_vy) @ S(Qt) + S(P_vz) @ S(At) + gC * (V @ S(E1).T)) / gam
        dB = (S(P_uy) @ S(Qt) + S(P_uz) @ S(At) + gC * (S(Uc) @ S(E1).T)) / gam
        dQ = (dY @ S(H0).T + S(P_vy).T @ S(Kh) + S(P_uy).T @ S(Bh)) * gam
        dA = (S(Z) @ S(H0).T + S(P_vz).T @ S(Kh) + S(P_uz).T @ S(Bh)) * gam_prev
        e = q[sl] * dQ - k[sl] * dK - b[sl] * dB
        e[:-1] += (a[sl] * dA)[1:]
        dG = torch.flip(torch.cumsum(torch.flip(e, [0]), 0), [0]) + (E1 * HC).sum(1)[None, :]
        for o, g in zip(outs, (dG * l["lw"], dQ, dK, dV, dA, dB)):
            o[sl] = g
    return outs


if __name__ == "__main__":
    T = 512
    ins = make_wkv_inputs(1, T, 1, seed=5, dtype=torch.bfloat16)
    w, q, k, v, a, b = [t[0, :, 0].float() for t in ins]
    dy = torch.randn(T, 64, generator=torch.Generator().manual_seed(1)).bfloat16().float()
    y_o, s_o, sa_o = c_oracle.wkv7_fwd(*[t.float() for t in ins])
    g_o = c_oracle.wkv7_bwd(*[t.float() for t in ins], dy.view(1, T, 1, 64), s_o, sa_o)
    for C in (16, 32):
        for ns, dt in ((0, torch.float64), (0, torch.float32), (2, torch.float32), (3, torch.float32)):
            y, U, hs, L, Ms = fwd3(w, q, k, v, a, b, C, dt, ns)
            ey = (y.float() - y_o[0, :, 0]).abs().max().item() / y_o.abs().max().item()
            eu = (U.float() - sa_o[0, :, 0]).abs().max().item() / sa_o.abs().max().item()
            grads = bwd3(w, q, k, v, a, b, dy, U, hs, L, Ms, C, dt, ns)
            eg = [(g.float() - go[0, :, 0]).abs().max().item() / go.abs().max().item() for g, go in zip(grads, g_o)]
            print(f"C={C:2d} split={ns} {str(dt)[6:]:8s} rel err y {ey:.2e} sa {eu:.2e} | dw,dq,dk,dv,da,db " +
                  " ".join(f"{e:.1e}" for e in eg))
