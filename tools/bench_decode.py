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
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    base = backbone.config_0p4b()
    cfg = RWKV7SpeechConfig(**{k: v for k, v in base.to_dict().items() if k in backbone.RWKV7Config.__dataclass_fields__ and k != "extra"})
    model = RWKV7ForSpeech(cfg).init_weights(0).to(dev, torch.bfloat16).eval()
    B, P = a.batch, a.prompt
    g = torch.Generator().manual_seed(1234)
    emb = (torch.randn(B, P, cfg.hidden_size, generator=g) * 0.5).to(dev, torch.bfloat16)
    mask = torch.ones(B, P, dtype=torch.long, device=dev)
    for mode in ("eager", "graph"):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if mode == "eager":
            ids = model.generate(inputs_embeds=emb, attention_mask=mask, max_new_tokens=a.gen, do_sample=False,
                                 eos_token_id=8192, pad_token_id=8192, suppress_tokens=[8192])
        else:
            dec = GraphDecoder(model, B)
            ids = dec.generate(inputs_embeds=emb, attention_mask=mask, max_new_tokens=a.gen, suppress_tokens=[8192])
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"{mode:6s}: B={B} prompt={P} gen={a.gen}: total {dt:.3f} s  -> {B * a.gen / dt:9.1f} tokens/s  "
              f"({dt / a.gen * 1e3:.3f} ms per decode step incl. prefill amortised)  first ids {ids[0, :6].tolist()}")
        if mode == "eager":
            ref = ids
    print("graph ids == eager ids:", bool(torch.equal(ref, ids)))


if __name__ == "__main__":
    main()
