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

# second view (round 4): the small ATen launches by input shape and by the first frame of this package on their stack
if os.environ.get("PROFILE_SITES", "1") == "1":
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof2:
        step(3)
        torch.cuda.synchronize()
    want = ("aten::fill_", "aten::zero_", "aten::copy_", "aten::sum", "aten::cat", "aten::add", "aten::add_", "aten::mul", "aten::mul_",
            "aten::sigmoid", "aten::tanh", "aten::_foreach_copy_", "aten::index_select", "aten::embedding_dense_backward", "aten::eq", "aten::ne")
    sites = {}
    for ev in prof2.events():
        if ev.name not in want:
            continue
        dt = getattr(ev, "device_time_total", getattr(ev, "cuda_time_total", 0))
        if dt <= 0:
            continue
        frame = next((s for s in (ev.stack or []) if "rwkvtts_amd" in s or "bench" in s or "profile_ops" in s), "?")
        frame = frame.split("rwkvtts_amd/")[-1][:70]
        key = (ev.name, str(ev.input_shapes)[:60], frame)
        c = sites.setdefault(key, [0, 0.0])
        c[0] += 1
        c[1] += dt
    print(f"\n{'op':28s} {'shapes':60s} {'site':70s} {'count':>6s} {'dev ms':>8s}")
    for (name, shp, frame), (n, t) in sorted(sites.items(), key=lambda kv: -kv[1][1])[:70]:
        print(f"{name:28s} {shp:60s} {frame:70s} {n:6d} {t / 1e3:8.2f}")
