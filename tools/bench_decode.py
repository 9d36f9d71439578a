"""BASELINE.json configs[4]: RWKV7-0.4B greedy autoregressive decode, B=32, prompt=128, gen=2048 (default shorter),
persistent-state decode path.  Reports prefill time, ms/token-step and tokens/s, eager vs hipGraph replay."""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rwkvtts_amd import backbone
from rwkvtts_amd.spark_llm import RWKV7ForSpeech, RWKV7SpeechConfig
from rwkvtts_amd.decode import GraphDecoder


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--prompt", type=int, default=128)
    ap.add_argument("--gen", type=int, default=256)
    ap.add_argument("--modes", default="graph-modules,graph-phases,graph-persistent")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    base = backbone.config_0p4b()
    cfg = RWKV7SpeechConfig(**{k: v for k, v in base.to_dict().items() if k in backbone.RWKV7Config.__dataclass_fields__ and k != "extra"})
    model = RWKV7ForSpeech(cfg).init_weights(0).to(dev, torch.bfloat16).eval()
    B, P = a.batch, a.prompt
    g = torch.Generator().manual_seed(1234)
    emb = (torch.randn(B, P, cfg.hidden_size, generator=g) * 0.5).to(dev, torch.bfloat16)
    mask = torch.ones(B, P, dtype=torch.long, device=dev)
    def run(mode, gen):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if mode == "eager":
            ids = model.generate(inputs_embeds=emb, attention_mask=mask, max_new_tokens=gen, do_sample=False,
                                 eos_token_id=8192, pad_token_id=8192, suppress_tokens=[8192])
        else:
            dec = GraphDecoder(model, B, step_kernel=(mode != "graph-modules"))
            if mode == "graph-persistent":
                import rwkvtts_amd.decode as D
                orig = D.DecodeStep.__init__
                D.DecodeStep.__init__ = lambda self, *a_, **k_: orig(self, *a_, **{**k_, "persistent": 1})
            try:
                ids = dec.generate(inputs_embeds=emb, attention_mask=mask, max_new_tokens=gen, suppress_tokens=[8192])
            finally:
                if mode == "graph-persistent":
                    D.DecodeStep.__init__ = orig
        torch.cuda.synchronize()
        return ids, time.perf_counter() - t0

    ref = None
    for mode in a.modes.split(","):
        run(mode, 8)   # warm-up (allocator, kernels)
        ids1, t1 = run(mode, a.gen // 4)
        ids, t2 = run(mode, a.gen)
        step = (t2 - t1) / (a.gen - a.gen // 4)
        print(f"{mode:14s}: B={B} prompt={P} gen={a.gen}: total {t2:.3f} s -> {B * a.gen / t2:9.1f} tokens/s incl. prefill+capture; "
              f"{step * 1e3:.3f} ms per decode step -> {B / step:9.1f} tokens/s steady  first ids {ids[0, :6].tolist()}", flush=True)
        if ref is None:
            ref = ids
        else:
            print(f"   ids agree with {a.modes.split(',')[0]}: {(ids == ref).float().mean().item():.4f}")


if __name__ == "__main__":
    main()
