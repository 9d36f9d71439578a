"""Why does the 32768 x 1024 x 1024 bf16 NT GEMM (r/k/v/o projections) take ~93 us inside the training step and ~63 us in
tools/bench_gemm_cold.py?  ("explain the in-step penalty".)  Four measurements on one box, HIP events:
  A  isolated: 48 calls after an idle gap (what bench_gemm_cold reports)
  B  sustained: 4000 back-to-back calls (~0.3 s of continuous MFMA load), medians of windows of 250 calls -> clock droop under load?
  C  right behind training steps (chip warm, power budget in use): 3 steps, then 100 calls immediately, no gap
  D  the same call with the operands a layer actually has: x = a fresh activation written by the previous kernel, W one of 24 layers'
     weights rotating (cold W each call), out = a fresh buffer
and the sclk torch reports before / after each."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

dev = torch.device("cuda:0")
M, N, K = 32768, 1024, 1024
g = torch.Generator(device=dev).manual_seed(0)
A = (torch.randn(M, K, device=dev, generator=g) * 0.5).bfloat16()
W = (torch.randn(N, K, device=dev, generator=g) * 0.05).bfloat16()
C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
flops = 2.0 * M * N * K


def clk():
    try:
        return torch.cuda.clock_rate()
    except Exception as e:  # noqa: BLE001
        return -1


def timed(fn, n):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for s, e in ev:
        s.record(); fn(); e.record()
    torch.cuda.synchronize()
    return [s.elapsed_time(e) * 1e3 for s, e in ev]


med = lambda ts: sorted(ts)[len(ts) // 2]
f = lambda: torch.mm(A, W.t(), out=C)
for _ in range(5):
    f()
torch.cuda.synchronize()
time.sleep(1.0)
print(f"sclk idle {clk()} MHz", flush=True)
ts = timed(f, 48)
print(f"A isolated (48 calls after 1 s idle): median {med(ts):.1f} us, first {ts[0]:.1f}, last {ts[-1]:.1f}  ({flops / med(ts) / 1e6:.0f} TF/s)  sclk {clk()}", flush=True)
time.sleep(1.0)
ts = timed(f, 4000)
print("B sustained 4000 calls, window medians (us):", " ".join(f"{med(ts[i:i + 250]):.1f}" for i in range(0, 4000, 250)), f" sclk {clk()}", flush=True)

from rwkvtts_amd import backbone, trainer  # noqa: E402
from rwkvtts_amd.layouts import synthetic_spark_batch  # noqa: E402
from rwkvtts_amd.spark_llm import RWKV7ForSpeech, RWKV7SpeechConfig  # noqa: E402
base = backbone.config_0p4b()
kw = {k: v for k, v in base.to_dict().items() if k in backbone.RWKV7Config.__dataclass_fields__ and k != "extra"}
model = RWKV7ForSpeech(RWKV7SpeechConfig(**kw)).init_weights(seed=0).to(device=dev, dtype=torch.bfloat16).train()
tr = trainer.DataParallelTrainer(model, lr=1e-4, warmup_steps=10, total_steps=1000)
for i in range(3):
    tr.step(**synthetic_spark_batch(model, 8, 4096, seed=i))
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(3):
    tr.step(**synthetic_spark_batch(model, 8, 4096, seed=10 + i))
ts = timed(f, 100)   # queued right behind the steps
dt = time.perf_counter() - t0
print(f"C behind 3 training steps ({dt * 1e3 / 3:.1f} ms/step incl. the probe): median {med(ts):.1f} us, first 10 {med(ts[:10]):.1f}, last 10 {med(ts[-10:]):.1f}  sclk {clk()}", flush=True)

# E: F.linear on a 3-D activation, as the model calls it
x3 = A.view(8, 4096, K)
ts = timed(lambda: torch.nn.functional.linear(x3, W), 100)
print(f"E F.linear([8,4096,1024], W): median {med(ts):.1f} us", flush=True)
Ws = [l.attn.r_proj.weight.detach() for l in model.model.layers]
xs = [torch.empty(M, K, device=dev, dtype=torch.bfloat16) for _ in range(4)]
outs = [torch.empty(M, N, device=dev, dtype=torch.bfloat16) for _ in range(4)]
it = [0]


def layer_like():
    i = it[0]
    it[0] += 1
    x = xs[i % 4]
    torch.mul(A, 1.0, out=x)     # the activation is freshly written by the previous kernel
    return x, Ws[i % 24], outs[i % 4]


def d_call():
    x, w, o = layer_like()
    torch.mm(x, w.t(), out=o)


ev = []
for _ in range(200):
    x, w, o = layer_like()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); torch.mm(x, w.t(), out=o); e.record()
    ev.append((s, e))
torch.cuda.synchronize()
ts = [s.elapsed_time(e) * 1e3 for s, e in ev]
print(f"D layer-like operands (fresh activation, rotating weights): median {med(ts):.1f} us  sclk {clk()}", flush=True)
