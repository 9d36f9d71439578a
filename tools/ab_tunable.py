"""Same-box A/B: hipBLASLt/rocBLAS solutions picked by PyTorch TunableOp (results file given as argv[1], tuning off) against the
library defaults, interleaved in one process, on the 0.4B Spark training step."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rwkvtts_amd import backbone, trainer
from rwkvtts_amd.layouts import synthetic_spark_batch
from rwkvtts_amd.spark_llm import RWKV7ForSpeech, RWKV7SpeechConfig
import torch.cuda.tunable as tun
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


steps(3)
tun.tuning_enable(False)
ok = tun.read_file(sys.argv[1])
print("results file accepted:", ok, "entries:", len(tun.get_results()), flush=True)
res = {False: [], True: []}
for rep in range(3):
    for v in (False, True):
        tun.enable(v)
        steps(1)
        res[v] += steps(4)
tun.enable(False)
med = {v: sorted(res[v])[len(res[v]) // 2] for v in res}
print(f"TunableOp solutions: off {med[False]:7.2f} ms   on {med[True]:7.2f} ms   ({med[True] - med[False]:+.2f} ms)", flush=True)
