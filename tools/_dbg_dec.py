import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rwkvtts_amd import backbone
from rwkvtts_amd.backbone import Cache, RWKV7ForCausalLM, RWKV7Config
from rwkvtts_amd.decode import DecodeStep
D, L, V, B = 1024, int(sys.argv[1]) if len(sys.argv) > 1 else 1, 8193, 32
cfg = RWKV7Config(hidden_size=D, num_hidden_layers=L, vocab_size=V)
m = RWKV7ForCausalLM(cfg); backbone.init_weights(m, cfg, seed=0); m = m.to("cuda:0", torch.bfloat16).eval()
c1 = Cache.zeros(cfg, B, "cuda:0", torch.bfloat16); c2 = Cache.zeros(cfg, B, "cuda:0", torch.bfloat16)
for c in (c1, c2):
    g = torch.Generator().manual_seed(1)
    for s in c.states:
        s.att_kv.copy_(torch.randn(s.att_kv.shape, generator=g) * 0.1); s.att_x_prev.copy_(torch.randn(s.att_x_prev.shape, generator=g)); s.ffn_x_prev.copy_(torch.randn(s.ffn_x_prev.shape, generator=g))
sk = DecodeStep(m.model, m.lm_head, c1, persistent=True); sp = DecodeStep(m.model, m.lm_head, c2, persistent=False)
x = (torch.randn(B, D, device="cuda:0") * 0.5).to(torch.bfloat16)
for it in range(6):
    x = (torch.randn(B, D, device="cuda:0") * 0.5).to(torch.bfloat16)
    lk = sk(x).clone(); lp = sp(x).clone(); torch.cuda.synchronize()
    dd = (lk - lp).abs()
    print(it, "logits diff", dd.max().item(), "n", (dd > 0).sum().item(), "rows", (dd > 0).any(1).nonzero().flatten().tolist()[:8], "cols", (dd > 0).any(0).nonzero().flatten().tolist()[:8], [torch.equal(s1.att_kv, s2.att_kv) for s1, s2 in zip(c1.states, c2.states)])
N2 = 3 * D + 64 + 64 + 32 + 128; F = 4 * D
def al(x): return (x + 255) & ~255
sizes = [("bar", 256), ("xa", 32*D*4), ("xb", 32*D*4), ("vfirst", 32*D*4), ("p_qkv", 2*32*N2*4), ("p_att", 8*32*D*4), ("kact", 32*F*2), ("p_val", 8*32*D*4), ("mixed", 6*32*D*2), ("yg", 32*D*2), ("kx", 32*D*2), ("hfin", 32*D*2)]
o = 0
wk, wp = sk.workspace.view(torch.uint8), sp.workspace.view(torch.uint8)
for n, sz in sizes:
    a, b = wk[o:o+sz], wp[o:o+sz]
    nd = (a != b).sum().item()
    print(f"{n:8s} off {o:9d} size {sz:9d} differing bytes {nd}")
    o += al(sz)
print("total", o, sk.workspace.numel())
for s1, s2 in zip(c1.states, c2.states):
    print("state eq", torch.equal(s1.att_kv, s2.att_kv), torch.equal(s1.att_x_prev, s2.att_x_prev), torch.equal(s1.ffn_x_prev, s2.ffn_x_prev))
