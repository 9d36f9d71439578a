"""CPU: the Cosy token samplers against golden ids produced by the reference's own functions
(third_party/cosyvoice/utils/common.py:106-135, via oracle/pin_sampling.py): same seed -> same id, including the
repetition-aware resampling branch."""
import torch

from conftest import load_golden
from rwkvtts_amd import cosy_llm as C


def test_ras_and_nucleus_sampling_match_reference_ids():
    g = load_golden("sampling.npz")
    ncase = sum(1 for k in g if k.startswith("scores"))
    assert ncase == 5
    for ci in range(ncase):
        scores, hist = g[f"scores{ci}"], g[f"hist{ci}"].tolist()
        for seed in range(12):
            torch.manual_seed(seed)
            assert int(C.ras_sampling(scores, hist, 25)) == int(g[f"ras{ci}"][seed]), (ci, seed)
            torch.manual_seed(seed)
            assert int(C.nucleus_sampling(scores, top_p=0.7, top_k=10)) == int(g[f"nuc{ci}"][seed]), (ci, seed)


def test_repetition_branch_fires_on_a_repeated_history():
    g = load_golden("sampling.npz")
    scores, hist = g["scores1"], g["hist1"].tolist()
    top = int(scores.argmax())
    assert hist.count(top) >= 10
    # with the arg-max id filling the window, a nucleus draw of that id must be replaced by a full-distribution draw:
    # over many seeds the ids therefore spread beyond the nucleus candidates
    nucleus_ids = {int(C.nucleus_sampling(scores, 0.8, 25, torch.Generator().manual_seed(s))) for s in range(50)}
    ras_ids = {int(C.ras_sampling(scores, hist, 25, generator=torch.Generator().manual_seed(s))) for s in range(200)}
    assert not ras_ids <= nucleus_ids
