"""Closed-form output laws used as yardsticks by the sampler tests (test infrastructure).  `exact_probs` is the float64 warper chain
the GPU distribution tests compare frequencies against (tests/test_sampling_gpu.py); tests/test_sampling.py ties it to the HF warpers
the reference's generate runs (transformers, requirements.txt:245) on CPU."""
import torch


def exact_probs(logits64, top_k, top_p, temperature):
    """the warper chain (temperature -> top-k -> top-p) in float64 on one row -> probability of every id"""
    x = logits64 / temperature
    if top_k > 0:
        kth = torch.topk(x, min(top_k, x.numel())).values[-1]
        x = x.masked_fill(x < kth, float("-inf"))
    if top_p < 1.0:
        sv, si = torch.sort(x, descending=False)
        cum = sv.softmax(-1).cumsum(-1)
        rem = cum <= (1 - top_p)
        rem[-1] = False
        x = x.masked_fill(torch.zeros_like(rem).scatter(0, si, rem), float("-inf"))
    return x.softmax(-1)
