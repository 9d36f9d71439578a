"""CPU (float64): how many bf16 ulp of extra error the per-chunk gradients of the chunked WKV7 backward pick up when the two
state checkpoints they read (H at the chunk start / end, adjoint E) are stored rounded -- bf16, fp16, or the q15 records of
csrc/chunk_common.h (int16 mantissas + a scale per (key, half of the value rows)).  Variant 1: rowsum(E * H_C) from the rounded
records; variant 2: the same term through the identity rowsum(E' * (H0 + B^^T U + K^^T V)); variant 3: U recomputed from the
rounded H0 as well.  Result (T = 512): bf16 up to 5.8 ulp (20 on dw), fp16 0.6 (2.6), q15 0.1 (0.2).
    python tools/diag_checkpoint_precision.py"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import chunked_proto2 as P
from rwkvtts_amd.synthetic import make_wkv_inputs
torch.manual_seed(0)
T, C = 512, 32
dt = torch.float64
ins = make_wkv_inputs(1, T, 1, seed=5, dtype=torch.bfloat16)
w, q, k, v, a, b = [t[0, :, 0].double() for t in ins]
dy = torch.randn(T, 64, generator=torch.Generator().manual_seed(1)).bfloat16().double()
y, U, hs, L, Ms = P.fwd3(w, q, k, v, a, b, C, dt, 0)
S = lambda x: x
bf = lambda x: x.float().bfloat16().double()
n = T // C
Np = [l["Qt"].T @ dy[c*C:c*C+C] + l["W"].T @ (l["A_qb"].T @ dy[c*C:c*C+C]) for c, l in enumerate(L)]
E = torch.zeros(64, 64, dtype=dt); Es = [None]*n
for c in range(n-1, -1, -1):
    Es[c] = E; E = Ms[c].T @ E + Np[c]

def out_stage(variant):
    outs = [torch.zeros(T, 64, dtype=dt) for _ in range(6)]
    for c, l in enumerate(L):
        sl = slice(c*C, c*C+C)
        H0, HC, E1 = hs[c], hs[c+1], Es[c]
        gam, gam_prev, gC = l["gam"], l["gam_prev"], l["gC"]
        Qt, At, Kh, Bh, Tm = l["Qt"], l["At"], l["Kh"], l["Bh"], l["Tm"]
        V, Uc, dY = v[sl], U[sl], dy[sl]
        if variant >= 1:
            H0, HC, E1 = bf(H0), bf(HC), bf(E1)
        if variant == 3:
            Uc = Tm @ (At @ H0 + l["A_ak"] @ V)          # U recomputed from the rounded H0
        Ep = gC[:, None] * E1                              # E' = g_C E
        Z = Tm.T @ (l["A_qb"].T @ dY + Bh @ Ep)
        dV = l["A_qk"].T @ dY + l["A_ak"].T @ Z + Kh @ Ep
        P_vy, P_vz = torch.triu(V @ dY.T), torch.triu(V @ Z.T, 1)
        P_uy, P_uz = torch.triu(Uc @ dY.T), torch.triu(Uc @ Z.T, 1)
        dK3, dB3 = V @ Ep.T, Uc @ Ep.T
        dK = (P_vy @ Qt + P_vz @ At + dK3) / gam
        dB = (P_uy @ Qt + P_uz @ At + dB3) / gam
        dQ = (dY @ H0.T + P_vy.T @ Kh + P_uy.T @ Bh) * gam
        dA = (Z @ H0.T + P_vz.T @ Kh + P_uz.T @ Bh) * gam_prev
        e = q[sl]*dQ - k[sl]*dK - b[sl]*dB
        e[:-1] += (a[sl]*dA)[1:]
        if variant in (0, 1):
            dterm = (E1 * HC).sum(1)
        else:   # identity: rowsum(E' * (H0 + B^^T U + K^^T V))
            dterm = (Ep * H0).sum(1) + (Bh * dB3).sum(0) + (Kh * dK3).sum(0)
        dG = torch.flip(torch.cumsum(torch.flip(e, [0]), 0), [0]) + dterm[None, :]
        for o, g in zip(outs, (dG*l["lw"], dQ, dK, dV, dA, dB)):
            o[sl] = g
    return outs
ref = out_stage(0)
names = ["dw","dq","dk","dv","da","db"]
for var in (1, 2, 3):
    got = out_stage(var)
    msg = []
    for nme, g, r in zip(names, got, ref):
        floor = r.abs().mean()*0.25 + 1e-6
        ulp = ((g - r).abs() / (2.0**-7 * torch.clamp(r.abs(), min=floor)))
        msg.append(f"{nme}: max {ulp.max().item():.2f} ulp, >0.5ulp {(ulp>0.5).double().mean().item():.3f}")
    print(f"variant {var}: " + " | ".join(msg))

def q16(x):   # x [k][v] : int16 with one scale per (k, half of v)
    out = torch.empty_like(x)
    for h in range(2):
        blk = x[:, 32*h:32*h+32]
        m = blk.abs().amax(1, keepdim=True).clamp(min=1e-30)
        out[:, 32*h:32*h+32] = torch.round(blk / m * 32767.0) * (m / 32767.0)
    return out
def f16(x): return x.float().half().double()
for name, fn in (("int16 block (k, v-half)", q16), ("fp16", f16)):
    bf = fn
    for var in (1, 2):
        got = out_stage(var)
        msg = []
        for nme, g, r in zip(names, got, ref):
            floor = r.abs().mean()*0.25 + 1e-6
            ulp = ((g - r).abs() / (2.0**-7 * torch.clamp(r.abs(), min=floor)))
            msg.append(f"{nme}: max {ulp.max().item():.3f}")
        print(f"{name} variant {var}: " + " | ".join(msg))


def q16_lane(x):   # x [k][v]: int16 with one scale per (v, k tile of 32, parity of bit 2 of k): the 16 values an MFMA lane holds
    out = torch.empty_like(x)
    kk = torch.arange(64)
    for kt in range(2):
        for h in range(2):
            sel = ((kk >> 5) == kt) & (((kk >> 2) & 1) == h)
            blk = x[sel, :]                      # [16][64 v]
            m = blk.abs().amax(0, keepdim=True).clamp(min=1e-30)
            out[sel, :] = torch.round(blk / m * 32767.0) * (m / 32767.0)
    return out
def q16_row(x):    # one scale per value column v over all 64 k
    m = x.abs().amax(0, keepdim=True).clamp(min=1e-30)
    return torch.round(x / m * 32767.0) * (m / 32767.0)
for name, fn in (("int16, scale per MFMA lane (v, 16 k)", q16_lane), ("int16, scale per v over all k", q16_row)):
    bf = fn
    for var in (1, 2):
        got = out_stage(var)
        msg = []
        for nme, g, r in zip(names, got, ref):
            floor = r.abs().mean()*0.25 + 1e-6
            ulp = ((g - r).abs() / (2.0**-7 * torch.clamp(r.abs(), min=floor)))
            msg.append(f"{nme}: max {ulp.max().item():.3f}")
        print(f"{name} variant {var}: " + " | ".join(msg))
