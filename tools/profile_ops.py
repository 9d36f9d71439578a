"""torch.profiler view of one training step: which ATen ops are behind the small launches (copies, fills, reductions)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rwkvtts_amd import backbone, trainer
from rwkvtts_amd.layouts import synthetic_spark_batch
from rwkvtts_amd.spark_llm import RWKV7ForSpeech, RWKV7SpeechConfig
dev = torch.device("cuda:0")
base = backbone.config_0p4b()
kw = {k: v for k, v in base.to_dict().items() if k in backbone.RWKV7Config.__dataclass_fields__ and k != "extra"}
model = RWKV7ForSpeech(RWKV7SpeechConfig(**kw)).init_weights(seed=0).to(device=dev, dtype=torch.bfloat16).train()
tr = trainer.DataParallelTrainer(model, lr=1e-4, warmup_steps=10, total_steps=1000)
step = lambda i: tr.step(**synthetic_spark_batch(model, 8, 4096, seed=1234 + i))
for i in range(2):
    step(i)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step(2)
    torch.cuda.synchronize()
ka = prof.key_averages()
rows = sorted(ka, key=lambda e: -e.count)
print(f"{'op':60s} {'count':>6s} {'cuda ms':>9s}")
for e in rows[:60]:
    print(f"{e.key[:60]:60s} {e.count:6d} {getattr(e, 'device_time_total', getattr(e, 'cuda_time_total', 0)) / 1e3:9.2f}")
