"""Per-step time of the generation loops around the decode step (0.4B widths): Spark GraphDecoder greedy / sampled (B = 32), XY
frames (8 channels, B = 8), Cosy streaming tokens (B = 1).  `python tools/bench_generate.py [trace]`: with `trace`, run few steps
(for rocprofv3 --kernel-trace)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rwkvtts_amd import backbone
from rwkvtts_amd.spark_llm import RWKV7ForSpeech, RWKV7SpeechConfig
from rwkvtts_amd.xy_llm import RWKV7XYLM, RWKV7XYConfig
from rwkvtts_amd.cosy_llm import RWKV7CosyLM, RWKV7CosyConfig
from rwkvtts_amd.decode import GraphDecoder

DEV = torch.device("cuda:0")
TRACE = len(sys.argv) > 1 and sys.argv[1] == "trace"
SIZE = os.environ.get("GEN_MODEL", "0.4b")   # 0.4b | 1.5b
base = {k: v for k, v in (backbone.config_1p5b() if SIZE == "1.5b" else backbone.config_0p4b()).to_dict().items()
        if k in backbone.RWKV7Config.__dataclass_fields__ and k != "extra"}


def timed(fn, n1, n2):
    fn(4)
    torch.cuda.synchronize(); t = time.perf_counter(); fn(n1); torch.cuda.synchronize(); t1 = time.perf_counter() - t
    torch.cuda.synchronize(); t = time.perf_counter(); fn(n2); torch.cuda.synchronize(); t2 = time.perf_counter() - t
    return (t2 - t1) / (n2 - n1) * 1e3


def spark():
    cfg = RWKV7SpeechConfig(**base)
    m = RWKV7ForSpeech(cfg).init_weights(0).to(DEV, torch.bfloat16).eval()
    B, P = 32, 128
    emb = (torch.randn(B, P, cfg.hidden_size) * 0.5).to(DEV, torch.bfloat16)
    mask = torch.ones(B, P, dtype=torch.long, device=DEV)
    for name, kw in (("greedy", {}), ("sampled top_k=50 top_p=0.95 T=0.8", dict(do_sample=True, top_k=50, top_p=0.95, temperature=0.8)),
                     ("sampled top_k=0 top_p=1 T=1", dict(do_sample=True))):
        dec = GraphDecoder(m, B, step_kernel=True)
        ms = timed(lambda n: dec.generate(inputs_embeds=emb, attention_mask=mask, max_new_tokens=n, suppress_tokens=[8192], seed=1, **kw),
                   8 if TRACE else 128, 16 if TRACE else 640)
        print(f"spark B=32 {name:36s}: {ms:.3f} ms per step, {B / ms * 1e3:9.0f} tokens/s", flush=True)


def xy():
    b = dict(base); b.update(vocab_size=66661)
    cfg = RWKV7XYConfig(speech_vocab_size=1025, num_channels=8, text_shift_size=65536, **b)
    m = RWKV7XYLM(cfg).init_weights(seed=1).to(DEV).to(torch.bfloat16).eval()
    B, T0 = 8, 32
    ids = torch.randint(0, 1024, (B, T0, 8), device=DEV)
    ids[:, :, 0] += 65536
    for name, kw in (("greedy", dict(do_sample=False)), ("sampled top_k=50", dict(do_sample=True, top_k=50, top_p=0.95, temperature=0.8))):
        ms = timed(lambda n: m.generate(ids, max_new_tokens=n, use_graph=True, **kw), 8 if TRACE else 64, 16 if TRACE else 320)
        print(f"xy    B=8 x 8 channels {name:24s}: {ms:.3f} ms per frame, {B / ms * 1e3:9.0f} frames/s", flush=True)


def cosy():
    b = dict(base); b.update(vocab_size=5000)
    cfg = RWKV7CosyConfig(speech_token_size=6561, **b)
    m = RWKV7CosyLM(cfg).init_weights(seed=4).to(DEV).to(torch.bfloat16).eval()
    text = torch.randint(5, 4000, (1, 200), device=DEV)
    z = torch.zeros(1, 0, dtype=torch.long, device=DEV)

    def run(n):
        torch.manual_seed(3)
        k = 0
        for _ in m.inference(text, torch.tensor([200], device=DEV), z, torch.tensor([0], device=DEV), z, torch.tensor([0], device=DEV),
                             max_token_text_ratio=20, min_token_text_ratio=20):   # min = max: never stops early
            k += 1
            if k >= n:
                break
    m.use_graph = True
    ms = timed(run, 8 if TRACE else 100, 16 if TRACE else 500)
    print(f"cosy  B=1 streaming, ras sampler           : {ms:.3f} ms per token, {1 / ms * 1e3:9.0f} tokens/s  graph={m.last_inference_used_graph}", flush=True)


if __name__ == "__main__":
    which = sys.argv[2].split(",") if len(sys.argv) > 2 else ["spark", "xy", "cosy"]
    for w in which:
        {"spark": spark, "xy": xy, "cosy": cosy}[w]()
