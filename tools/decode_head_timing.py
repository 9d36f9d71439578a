"""Cycle split of the decode step's head phase (build with `python -m rwkvtts_amd.build --timing` first)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rwkvtts_amd import backbone
from rwkvtts_amd.backbone import Cache, RWKV7ForCausalLM
from rwkvtts_amd.decode import DecodeStep

cfg = backbone.config_0p4b(vocab_size=8193)
m = RWKV7ForCausalLM(cfg)
backbone.init_weights(m, cfg, seed=0)
m = m.to("cuda:0", torch.bfloat16).eval()
B = 32
cache = Cache.zeros(cfg, B, "cuda:0", torch.bfloat16)
step = DecodeStep(m.model, m.lm_head, cache)
x = (torch.randn(B, cfg.hidden_size, device="cuda:0") * 0.5).to(torch.bfloat16)
for _ in range(3):
    step(x)
torch.cuda.synchronize()
step.workspace[64:64 + 64].zero_()
N = 10
for _ in range(N):
    step(x)
torch.cuda.synchronize()
t = step.workspace[64:64 + 48].view(torch.int64).tolist()
names = ["prefetch + A loads", "A act -> LDS", "B up-proj", "C elementwise", "D state", "E norm+store"]
L = cfg.num_hidden_layers
for n, v in zip(names, t):
    print(f"{n:20s} {v / N / L:9.0f} cycles")
print(f"{'sum':20s} {sum(t) / N / L:9.0f} cycles")
