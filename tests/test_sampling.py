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


def test_spark_sampling_filter_matches_hf_warpers():
    """inference/rwkv7speech_inference.py:99-107 samples through HF generate (transformers pinned in the reference's
    requirements.txt:245): temperature -> top-k -> top-p warpers, then one multinomial draw.  sample_next must keep exactly
    the candidates the HF warpers keep and, under the same seed, draw the same ids."""
    import pytest
    tf = pytest.importorskip("transformers")
    from transformers.generation.logits_process import TemperatureLogitsWarper, TopKLogitsWarper, TopPLogitsWarper
    from rwkvtts_amd.spark_llm import sample_next
    g = torch.Generator().manual_seed(5)
    for V, temp, k, p in [(8193, 1.0, 50, 0.95), (8193, 0.8, 0, 0.9), (300, 1.3, 20, 1.0), (64, 0.7, 5, 0.5)]:
        logits = torch.randn(4, V, generator=g) * 3
        ids = torch.zeros(4, 1, dtype=torch.long)
        x = logits.clone()
        if temp != 1.0:
            x = TemperatureLogitsWarper(temp)(ids, x)
        if k:
            x = TopKLogitsWarper(k)(ids, x)
        if p < 1.0:
            x = TopPLogitsWarper(p)(ids, x)
        torch.manual_seed(11)
        want = torch.multinomial(torch.softmax(x, -1), 1).squeeze(1)
        torch.manual_seed(11)
        got = sample_next(logits.clone(), True, k, p, temp)
        assert torch.equal(got, want), (V, temp, k, p)
    assert torch.equal(sample_next(torch.tensor([[1.0, 3.0, 3.0, 2.0]])), torch.tensor([1]))   # greedy: first max


def test_device_ras_sampling_has_the_distribution_of_the_host_functions():
    """cosy_llm.ras_sampling_device (tensor-only, lives inside the captured Cosy decode step) against ras_sampling + the EOS rejection
    loop of sampling_ids (the pair pinned id for id against cosyvoice/utils/common.py above): same distribution over 4000 draws --
    nucleus branch, repetition branch (full-distribution resample), and EOS ignored / allowed."""
    from rwkvtts_amd.cosy_llm import ras_sampling, ras_sampling_device
    g = torch.Generator().manual_seed(0)
    V, EOS, N = 24, 23, 4000
    logp = (torch.randn(V, generator=g) * 1.5).log_softmax(0)
    logp[EOS] = logp.max() + 0.3          # EOS is the most likely id: the rejection matters
    logp = logp.log_softmax(0)

    def host(decoded, ignore_eos):
        while True:
            t = int(ras_sampling(logp, decoded, 25, top_p=0.8, top_k=6))
            if not ignore_eos or t != EOS:
                return t

    for decoded, ignore in (([], False), ([], True), ([3, 3, 5], True), ([int(logp[:EOS].argmax())] * 4, True)):
        recent = torch.full((10,), -1, dtype=torch.long)
        recent[:len(decoded[-10:])] = torch.tensor(decoded[-10:], dtype=torch.long) if decoded else recent[:0]
        torch.manual_seed(1)
        a = torch.bincount(torch.tensor([host(decoded, ignore) for _ in range(N)]), minlength=V).float() / N
        b = torch.bincount(torch.cat([ras_sampling_device(logp, recent, torch.tensor(ignore), EOS, top_p=0.8, top_k=6) for _ in range(N)]),
                           minlength=V).float() / N
        assert (a - b).abs().max().item() < 0.035, (decoded, ignore, (a - b).abs().max().item())
        if ignore:
            assert b[EOS] == 0


def _ras_laws(logp, recent, eos, top_p, top_k, win, tau_r):
    """Exact output laws while EOS is being ignored.  `ref`: the reference's rejection loop (sampling_ids, llm.py:160-176, around
    ras_sampling, common.py:109-137): one trial draws x ~ N (nucleus); if x repeats in the window the trial's result is a fresh draw
    from F (full); an EOS result restarts the WHOLE trial.  `dev`: EOS removed per stage (cosy_llm.ras_sampling_device and
    csrc/sampling.hip ras_step_kernel -- tests/test_sampling_gpu.py::_ras_exact is this law).  Also returns N(R), F(EOS), N(EOS)."""
    F = logp.double().softmax(0)
    sv, si = F.sort(descending=True, stable=True)
    cum_before = sv.cumsum(0) - sv
    keep = ((cum_before < top_p) & (torch.arange(sv.numel()) < top_k)).long().cumprod(0).double()
    N = torch.zeros_like(F).scatter(0, si, sv * keep)
    N = N / N.sum()
    rep = torch.tensor([(recent == i).sum().item() for i in range(F.numel())])
    inR = (rep >= win * tau_r).double()
    rho = (N * inR).sum()
    one = N * (1 - inR) + rho * F                       # law of ONE trial, EOS included
    ref = one.clone()
    ref[eos] = 0
    ref = ref / ref.sum()
    Nn = N.clone()
    Nn[eos] = 0
    Nn = Nn / Nn.sum() if Nn.sum() > 0 else None
    Fn = F.clone()
    Fn[eos] = 0
    Fn = Fn / Fn.sum()
    if Nn is None:
        Nn = Fn
    dev = Nn * (1 - inR) + (Nn * inR).sum() * Fn
    return ref, dev, float(rho), float(F[eos]), float(N[eos])


def test_eos_rejection_laws_reference_vs_per_stage_bound_and_host_sampler():
    """(1) The pinned host pair (ras_sampling inside the rejection loop) follows `ref` empirically (5 sigma over 20 000 draws): ties the
    closed form to the functions that are pinned id for id against the reference.  (2) The per-stage law the device kernels implement
    differs from it only when a repeat fallback is possible AND EOS has mass, by total variation <= N(R) F(EOS) / (1 - N(EOS)) (both
    laws are mixtures of the same two components with weights A/(A + N(R)(1 - F(EOS))) and A/(A + N(R)), A = 1 - N(EOS) - N(R)); in
    the other cases they are identical."""
    from rwkvtts_amd.cosy_llm import ras_sampling
    g = torch.Generator().manual_seed(3)
    V, EOS, win = 24, 23, 10
    for case in ("no_repeat", "repeat", "repeat_eos_heavy"):
        logp = (torch.randn(V, generator=g) * 1.5)
        if case == "repeat_eos_heavy":
            logp[EOS] = logp.max() + 0.5
        logp = logp.log_softmax(0)
        top = int(logp[:EOS].argmax())
        decoded = [] if case == "no_repeat" else [top, 2, top]
        recent = torch.full((win,), -1, dtype=torch.long)
        recent[:len(decoded)] = torch.tensor(decoded, dtype=torch.long) if decoded else recent[:0]
        ref, dev, rho, f, n = _ras_laws(logp, recent, EOS, 0.8, 6, win, 0.1)
        tv = 0.5 * (ref - dev).abs().sum().item()
        assert tv <= rho * f / (1 - n) + 1e-12, (case, tv, rho, f, n)
        if case == "no_repeat":
            assert tv < 1e-12
        else:
            assert rho > 0.2
        torch.manual_seed(5)
        Nd = 20000

        def host():
            while True:
                t = int(ras_sampling(logp, decoded, 25, top_p=0.8, top_k=6))
                if t != EOS:
                    return t

        freq = torch.bincount(torch.tensor([host() for _ in range(Nd)]), minlength=V).double() / Nd
        sigma = (ref * (1 - ref) / Nd).sqrt()
        assert ((freq - ref).abs() <= 5 * sigma + 1e-9).all(), (case, (freq - ref).abs().max().item())
        print(f"[ras EOS rejection, {case}] N(R)={rho:.3f} F(EOS)={f:.3f} N(EOS)={n:.3f}: TV(reference law, per-stage law) = {tv:.4f} "
              f"(bound {rho * f / (1 - n):.4f})")


def test_float64_warper_chain_used_by_the_gpu_tests_equals_the_hf_warpers():
    """tests/sampling_laws.exact_probs -- the yardstick of the fused draw's distribution tests on the GPU -- against the HF warpers
    themselves in float64: the same support and the same probabilities (ties at the k-th value and at the nucleus boundary included)."""
    import pytest
    pytest.importorskip("transformers")
    from transformers.generation.logits_process import TemperatureLogitsWarper, TopKLogitsWarper, TopPLogitsWarper
    from sampling_laws import exact_probs
    g = torch.Generator().manual_seed(9)
    cases = [(8193, 1.0, 50, 0.95), (8193, 0.8, 50, 1.0), (300, 1.3, 20, 0.7), (64, 0.7, 5, 0.5), (1025, 1.0, 64, 0.9)]
    for V, temp, k, p in cases:
        logits = (torch.randn(V, generator=g) * 3).double()
        logits[7] = logits[3]                      # an exact tie somewhere
        if k:
            kth = torch.topk(logits, k).values[-1]
            logits[(logits < kth).nonzero()[0]] = kth   # ... and one AT the k-th value (both warpers keep ties of the k-th)
        ids = torch.zeros(1, 1, dtype=torch.long)
        x = logits.clone()[None]
        if temp != 1.0:
            x = TemperatureLogitsWarper(temp)(ids, x)
        if k:
            x = TopKLogitsWarper(k)(ids, x)
        if p < 1.0:
            x = TopPLogitsWarper(p)(ids, x)
        want = torch.softmax(x[0], -1)
        got = exact_probs(logits, k, p, temp)
        assert torch.equal(got > 0, want > 0), (V, temp, k, p)
        assert (got - want).abs().max().item() < 1e-12
