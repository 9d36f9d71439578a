"""Same-process A/B of two builds of librwkv7_hip.so on the AdamW kernel at the 0.4B model's size (404.6 M parameters, three groups):
    python tools/probes/adamw_ab.py tools/ab/librwkv7_hip_base.so rwkvtts_amd/lib/librwkv7_hip.so"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
libs = [ctypes.CDLL(os.path.abspath(p)) for p in sys.argv[1:3]]
dev = "cuda:0"
n = 404_635_648 // 128 * 128
P = lambda t: ctypes.c_void_p(t.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
def state(seed):
    g = torch.Generator(device=dev).manual_seed(seed)
    p32 = torch.randn(n, device=dev, generator=g) * 0.02
    return dict(p32=p32, g16=(torch.randn(n, device=dev, generator=g) * 1e-3).bfloat16(), m=torch.zeros(n, device=dev), v=torch.zeros(n, device=dev),
                p16=p32.bfloat16())
grp = torch.randint(0, 3, (n // 128,), device=dev, dtype=torch.uint8)
tab = torch.tensor([[1.0, 0.0], [2.0, 0.0], [1.0, 0.1]], device=dev)
flag = torch.zeros(1, device=dev)
S = [state(0), state(0)]
def run(L, s, step):
    rc = L.rwkv7_adamw_groups_bf16(ctypes.c_long(n), P(s["p32"]), P(s["g16"]), P(s["m"]), P(s["v"]), P(s["p16"]), P(grp), P(tab), 3, P(flag),
                                   ctypes.c_float(1e-4), ctypes.c_float(0.9), ctypes.c_float(0.95), ctypes.c_float(1e-18), step, st)
    assert rc == 0, rc
ts = [[], []]
for it in range(1, 13):
    for i, L in enumerate(libs):
        s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s_.record(); run(L, S[i], it); e_.record()
        ts[i].append((s_, e_))
torch.cuda.synchronize()
med = [sorted(a.elapsed_time(b) for a, b in t[2:])[len(t[2:]) // 2] for t in ts]
print(f"adamw (404.6 M params, 28 B each = {28 * n / 1e9:.2f} GB): A {med[0]:.3f} ms ({28 * n / med[0] / 1e6:.0f} GB/s)   B {med[1]:.3f} ms ({28 * n / med[1] / 1e6:.0f} GB/s)   ({100 * (med[1] / med[0] - 1):+.1f} %)")
print("state after 12 steps identical:", [torch.equal(S[0][k], S[1][k]) for k in ("p32", "m", "v", "p16")])
