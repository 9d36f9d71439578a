"""CPU prototype (torch fp64/fp32) of the chunked WKV7 forward + backward used to validate the math of the
MFMA kernels before writing them.  Notation: per head, H = S^T in R^{K x V}; chunk length C.
Checks against the scalar oracle (oracle/c_oracle.py)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import c_oracle
from rwkvtts_amd.synthetic import make_wkv_inputs


def bf16_split(x, nsplit):
    """emulate hi+lo bf16 operand splitting: returns sum of nsplit bf16 pieces (as float)"""
    if nsplit == 0:
        return x
    out = torch.zeros_like(x)
    r = x.clone()
    for _ in range(nsplit):
        p = r.to(torch.bfloat16).to(x.dtype)
        out = out + p
        r = r - p
    return out


def chunk_fwd(w, q, k, v, a, b, C=32, dt=torch.float64, ns=0):
    """w,q,k,v,a,b [T,64] one head.  Returns y [T,V], U (=sa) [T,V], list of chunk-start states H0 [K,V] (+final)."""
    T = w.shape[0]
    w, q, k, v, a, b = [t.to(dt) for t in (w, q, k, v, a, b)]
    lw = -torch.exp(w)
    H = torch.zeros(64, 64, dtype=dt)
    ys, us, hs = [], [], []
    S = lambda x: bf16_split(x, ns)
    for c0 in range(0, T, C):
        sl = slice(c0, c0 + C)
        G = torch.cumsum(lw[sl], 0)                 # [C,K]
        gam = torch.exp(G)
        gam_prev = torch.exp(G - lw[sl])
        Qt, At = q[sl] * gam, a[sl] * gam_prev
        Kh, Bh = k[sl] / gam, b[sl] / gam
        V = v[sl]
        tril_s = torch.tril(torch.ones(C, C, dtype=dt), -1)
        tril = torch.tril(torch.ones(C, C, dtype=dt))
        A_ab = (S(At) @ S(Bh).T) * tril_s
        A_ak = (S(At) @ S(Kh).T) * tril_s
        A_qb = (S(Qt) @ S(Bh).T) * tril
        A_qk = (S(Qt) @ S(Kh).T) * tril
        Tm = torch.linalg.inv(torch.eye(C, dtype=dt) - A_ab)
        R = S(At) @ S(H) + S(A_ak) @ V
        U = S(Tm) @ S(R)
        Y = S(Qt) @ S(H) + S(A_qb) @ S(U) + S(A_qk) @ V
        hs.append(H)
        H = gam[-1][:, None] * (H + S(Bh).T @ S(U) + S(Kh).T @ V)
        ys.append(Y); us.append(U)
    hs.append(H)
    return torch.cat(ys), torch.cat(us), hs


def chunk_bwd(w, q, k, v, a, b, dy, U, hs, C=32, dt=torch.float64, ns=0):
    T = w.shape[0]
    w, q, k, v, a, b, dy, U = [t.to(dt) for t in (w, q, k, v, a, b, dy, U)]
    lw = -torch.exp(w)
    E = torch.zeros(64, 64, dtype=dt)  # dL/dH_C from the future, before adding q_C dy_C^T
    S = lambda x: bf16_split(x, ns)
    dq, dk, dv, da, db, dlw = [torch.zeros(T, 64, dtype=dt) for _ in range(6)]
    nchunk = T // C
    for ci in range(nchunk - 1, -1, -1):
        sl = slice(ci * C, ci * C + C)
        H0, HC = hs[ci].to(dt), hs[ci + 1].to(dt)
        G = torch.cumsum(lw[sl], 0)
        gam, gam_prev = torch.exp(G), torch.exp(G - lw[sl])
        gC = gam[-1]
        Qt, At, Kh, Bh = q[sl] * gam, a[sl] * gam_prev, k[sl] / gam, b[sl] / gam
        V, Uc, dY = v[sl], U[sl], dy[sl]
        tril_s = torch.tril(torch.ones(C, C, dtype=dt), -1)
        tril = torch.tril(torch.ones(C, C, dtype=dt))
        A_ab = (S(At) @ S(Bh).T) * tril_s
        A_ak = (S(At) @ S(Kh).T) * tril_s
        A_qb = (S(Qt) @ S(Bh).T) * tril
        A_qk = (S(Qt) @ S(Kh).T) * tril
        Tm = torch.linalg.inv(torch.eye(C, dtype=dt) - A_ab)
        # Z_t = D_t^T b_t  (= dL/du_t)
        Z = S(Tm).T @ S(S(A_qb).T @ dY + S(Bh * gC) @ S(E))
        dV = S(A_qk).T @ dY + S(A_ak).T @ S(Z) + S(Kh * gC) @ S(E)
        P_vy = torch.triu(V @ dY.T)           # [t,s] = v_t . dy_s, s >= t
        P_vz = torch.triu(V @ S(Z).T, 1)      # s > t
        P_uy = torch.triu(S(Uc) @ dY.T)
        P_uz = torch.triu(S(Uc) @ S(Z).T, 1)
        dK = (S(P_vy) @ S(Qt) + S(P_vz) @ S(At) + gC * (V @ S(E).T)) / gam
        dB = (S(P_uy) @ S(Qt) + S(P_uz) @ S(At) + gC * (S(Uc) @ S(E).T)) / gam
        dQ = (dY @ S(H0).T + S(P_vy).T @ S(Kh) + S(P_uy).T @ S(Bh)) * gam
        P_zv = torch.tril(S(Z) @ V.T, -1)
        P_zu = torch.tril(S(Z) @ S(Uc).T, -1)
        dA = (S(Z) @ S(H0).T + S(P_zv) @ S(Kh) + S(P_zu) @ S(Bh)) * gam_prev
        # decay gradient: local reverse cumsum + chunk-end term
        e = q[sl] * dQ - k[sl] * dK - b[sl] * dB
        e[:-1] += (a[sl] * dA)[1:]
        dG = torch.flip(torch.cumsum(torch.flip(e, [0]), 0), [0])
        dG = dG + (E * HC).sum(1)[None, :]
        dq[sl], dk[sl], dv[sl], da[sl], db[sl], dlw[sl] = dQ, dK, dV, dA, dB, dG
        E = gC[:, None] * E + S(Qt).T @ dY + S(At).T @ S(Z)
    # a_{t} of the first step of chunk c+1 depends on gamma_{t-1} = 1 of its own chunk: nothing crosses chunks
    dw = dlw * lw  # d lw / d w = -exp(w) = lw
    return dw, dq, dk, dv, da, db


if __name__ == "__main__":
    torch.manual_seed(0)
    T = 256
    ins = make_wkv_inputs(1, T, 1, seed=5, dtype=torch.bfloat16)
    w, q, k, v, a, b = [t[0, :, 0].float() for t in ins]
    dy = torch.randn(T, 64).bfloat16().float()
    y_o, s_o, sa_o = c_oracle.wkv7_fwd(*[t.float() for t in ins])
    g_o = c_oracle.wkv7_bwd(*[t.float() for t in ins], dy.view(1, T, 1, 64), s_o, sa_o)
    for C in (16, 32, 64):
        for ns, dt in ((0, torch.float64), (0, torch.float32), (2, torch.float32), (1, torch.float32)):
            y, U, hs = chunk_fwd(w, q, k, v, a, b, C, dt, ns)
            ey = (y.float() - y_o[0, :, 0]).abs().max().item() / y_o.abs().max().item()
            eu = (U.float() - sa_o[0, :, 0]).abs().max().item() / sa_o.abs().max().item()
            grads = chunk_bwd(w, q, k, v, a, b, dy, U, hs, C, dt, ns)
            eg = [(g.float() - go[0, :, 0]).abs().max().item() / go.abs().max().item() for g, go in zip(grads, g_o)]
            print(f"C={C:2d} split={ns} {str(dt)[6:]:8s} rel err y {ey:.2e} sa {eu:.2e} | dw,dq,dk,dv,da,db " +
                  " ".join(f"{e:.1e}" for e in eg))
