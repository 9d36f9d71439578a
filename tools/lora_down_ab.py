# same-process timing: the direct kernel against the library GEMM + combine pair (forward only), B=8 T=4096 D=1024
import torch, time, sys
sys.path.insert(0, "/root/repo")
from rwkvtts_amd import fused
DEV = "cuda"
B, T, D = 8, 4096, 1024
ranks = (64, 64, 32, 128)
acts = ["tanh", None, None, "sigmoid"]
g = torch.Generator().manual_seed(1)
x = torch.randn(B, T, D, generator=g).to(DEV, torch.bfloat16).requires_grad_(True)
mus6 = [torch.rand(1, 1, D, generator=g).to(DEV, torch.bfloat16).requires_grad_(True) for _ in range(6)]
w1s = [(torch.randn(r, D, generator=g) * D ** -0.5).to(DEV, torch.bfloat16).requires_grad_(True) for r in ranks]
x_r, x_w, x_k, x_v, x_a, x_g = mus6
def run(direct, n):
    fused.LORA_DOWN_DIRECT = direct
    for _ in range(3):
        fused.mix_lora(x, None, x_r, x_k, x_v, [x_w, x_a, x_v, x_g], w1s, acts)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fused.mix_lora(x, None, x_r, x_k, x_v, [x_w, x_a, x_v, x_g], w1s, acts)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for rnd in range(3):
    print("round", rnd, "through-the-lerp pair %.1f us   direct %.1f us   (whole fused.mix_lora forward incl. mix_fwd<3>)" % (run(False, 50), run(True, 50)))
