"""Decode throughput when the B sequences are split over N concurrently replayed graphs (N streams): a decode step is a chain of
~170 latency-bound launches that leave most of the GPU idle, so independent groups of sequences can overlap."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rwkvtts_amd import backbone
from rwkvtts_amd.spark_llm import RWKV7ForSpeech, RWKV7SpeechConfig
from rwkvtts_amd.decode import GraphDecoder

dev = torch.device("cuda:0")
base = backbone.config_0p4b()
cfg = RWKV7SpeechConfig(**{k: v for k, v in base.to_dict().items() if k in backbone.RWKV7Config.__dataclass_fields__ and k != "extra"})
model = RWKV7ForSpeech(cfg).init_weights(0).to(dev, torch.bfloat16).eval()
P, STEPS = 128, 400
for total, groups in ((32, 1), (32, 2), (32, 4), (64, 2), (64, 4), (128, 4), (128, 8), (256, 8)):
    b = total // groups
    decs, streams = [], []
    for g in range(groups):
        emb = (torch.randn(b, P, cfg.hidden_size, generator=torch.Generator().manual_seed(g)) * 0.5).to(dev, torch.bfloat16)
        mask = torch.ones(b, P, dtype=torch.long, device=dev)
        d = GraphDecoder(model, b)
        d.generate(inputs_embeds=emb, attention_mask=mask, max_new_tokens=STEPS + 8, suppress_tokens=[8192])
        d.pos.fill_(1)   # replay again from position 1 (the ids written are irrelevant here)
        decs.append(d)
        streams.append(torch.cuda.Stream())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(STEPS):
        for d, s in zip(decs, streams):
            with torch.cuda.stream(s):
                d.graph.replay()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{total:4d} sequences as {groups} x {b:3d}: {dt / STEPS * 1e3:7.3f} ms per step of all groups -> {total * STEPS / dt:9.0f} tokens/s", flush=True)
