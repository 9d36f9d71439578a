"""oracle/pin_sampling.py -- AUTHORING-CONTAINER ONLY (needs /root/reference).

Runs the reference's token samplers (third_party/cosyvoice/utils/common.py:106-135: ras_sampling, nucleus_sampling,
random_sampling -- what model/llm/llm.py:160-176 calls per generated speech token) on fixed score vectors under fixed global
seeds and writes inputs + the ids they return to tests/golden/sampling.npz, then checks
rwkvtts_amd.cosy_llm.{ras_sampling, nucleus_sampling} against them: both draw once from the same renormalised candidate
vector, so under the same seed the ids must be identical.   Usage: python oracle/pin_sampling.py [--write]
"""
import argparse
import importlib.util
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden", "sampling.npz")
REF = "/root/reference/third_party/cosyvoice/utils/common.py"


def cases():
    g = torch.Generator().manual_seed(7)
    out = []
    for i, (n, scale) in enumerate([(50, 1.0), (50, 4.0), (6562, 2.0), (30, 0.1), (100, 8.0)]):
        scores = torch.randn(n, generator=g) * scale
        hist = torch.randint(0, n, (14,), generator=g).tolist()
        if i % 2 == 1:   # a history that makes the repetition branch fire: the arg-max id repeated
            hist = hist[:4] + [int(scores.argmax())] * 10
        out.append((scores, hist))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--write", action="store_true")
    a = ap.parse_args()
    spec = importlib.util.spec_from_file_location("ref_common", REF)
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    from rwkvtts_amd import cosy_llm as C
    gold = {}
    for ci, (scores, hist) in enumerate(cases()):
        gold[f"scores{ci}"] = scores.numpy()
        gold[f"hist{ci}"] = np.asarray(hist, dtype=np.int64)
        ids_ras, ids_nuc = [], []
        for seed in range(12):
            torch.manual_seed(seed)
            want = int(ref.ras_sampling(scores, hist, 25))
            torch.manual_seed(seed)
            got = int(C.ras_sampling(scores, hist, 25))
            assert got == want, ("ras", ci, seed, got, want)
            ids_ras.append(want)
            torch.manual_seed(seed)
            want = int(ref.nucleus_sampling(scores, top_p=0.7, top_k=10))
            torch.manual_seed(seed)
            got = int(C.nucleus_sampling(scores, top_p=0.7, top_k=10))
            assert got == want, ("nucleus", ci, seed, got, want)
            ids_nuc.append(want)
        gold[f"ras{ci}"] = np.asarray(ids_ras, dtype=np.int64)
        gold[f"nuc{ci}"] = np.asarray(ids_nuc, dtype=np.int64)
        print(f"  OK  case {ci}: n={scores.numel()} ras {ids_ras[:6]}... nucleus {ids_nuc[:6]}...")
    if a.write:
        np.savez_compressed(GOLD, **gold)
        print("wrote", GOLD)


if __name__ == "__main__":
    main()
