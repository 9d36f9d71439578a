"""Packed variable-length rows (cu_seqlens): the native path (32-aligned re-layout + per-sequence chunk ranges in the chunked WKV7
kernels) against unpacking into a padded masked batch.  RWKV7-0.4B backbone, bf16, forward + backward."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rwkvtts_amd import backbone
from rwkvtts_amd.backbone import RWKV7Model

dev = "cuda:0"
cfg = backbone.config_0p4b(vocab_size=8193)
model = RWKV7Model(cfg)
backbone.init_weights(model, cfg, seed=0)
model = model.to(dev, torch.bfloat16).train()
lens = [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else "4000,2500,1200,900,600,300,150,80".split(","))]
total = sum(lens)
x0 = (torch.randn(1, total, cfg.hidden_size, device=dev) * 0.5).to(torch.bfloat16)
cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32, device=dev)
for native in (True, False, True, False):
    backbone.PACKED_NATIVE = native
    for it in range(4):
        if it == 1:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        model.zero_grad(set_to_none=True)
        x = x0.clone().requires_grad_(True)
        model(inputs_embeds=x, cu_seqlens=cu).last_hidden_state.float().square().mean().backward()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print(f"{'native (sequence ranges)' if native else 'padded batch          '}: {len(lens)} sequences, {total} tokens "
          f"(longest {max(lens)}): {dt * 1e3:7.1f} ms fwd+bwd -> {total / dt:9.0f} tokens/s", flush=True)
