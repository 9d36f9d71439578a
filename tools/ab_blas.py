"""Same-box A/B of torch's BLAS backend preference (rocBLAS vs hipBLASLt) on the 0.4B Spark training step."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rwkvtts_amd import backbone, trainer
from rwkvtts_amd.layouts import synthetic_spark_batch
from rwkvtts_amd.spark_llm import RWKV7ForSpeech, RWKV7SpeechConfig
dev = torch.device("cuda:0")
base = backbone.config_0p4b()
kw = {k: v for k, v in base.to_dict().items() if k in backbone.RWKV7Config.__dataclass_fields__ and k != "extra"}
model = RWKV7ForSpeech(RWKV7SpeechConfig(**kw)).init_weights(seed=0).to(device=dev, dtype=torch.bfloat16).train()
tr = trainer.DataParallelTrainer(model, lr=1e-4, warmup_steps=10, total_steps=1000)
it = [0]


def steps(n):
    ts = []
    for _ in range(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tr.step(**synthetic_spark_batch(model, 8, 4096, seed=1234 + it[0]))
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
        it[0] += 1
    return ts


print("default preference:", torch.backends.cuda.preferred_blas_library(), flush=True)
steps(3)
res = {}
for rep in range(4):
    for lib in ("cublas", "cublaslt"):
        torch.backends.cuda.preferred_blas_library(lib)
        steps(1)
        res.setdefault(lib, []).extend(steps(4))
for lib, ts in res.items():
    print(f"{lib:9s} ({'rocBLAS' if lib == 'cublas' else 'hipBLASLt'}): median {sorted(ts)[len(ts) // 2]:7.2f} ms", flush=True)
