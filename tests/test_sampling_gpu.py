"""GPU (-m gpu): csrc/sampling.hip -- the fused token draws (rwkv7_sample_rows_f32, rwkv7_ras_step_f32) against the torch chains
they replace: spark_llm.sample_next (HF temperature -> top-k -> top-p -> multinomial; reference utils/utilities.py:101-117,
model/llm/xy_llm.py:88-101) and cosy_llm.ras_sampling_device (third_party/cosyvoice/utils/common.py:109-137 + llm.py:160-176).
Argmax is compared id for id; sampled draws are compared as DISTRIBUTIONS (exact probabilities from the float64 chain against
frequencies over thousands of independent draws, 5-sigma bands) and by their support (an id the chain gives probability 0 must
never come out)."""
import pytest
import torch

from rwkvtts_amd.sampling import RowSampler, ras_step
from sampling_laws import exact_probs as _exact_probs

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _check_freq(ids, p, what):
    n = ids.numel()
    freq = torch.bincount(ids.flatten().cpu(), minlength=p.numel()).double() / n
    assert (freq[p == 0] == 0).all(), f"{what}: drew an id outside the support"
    band = 5.0 * torch.sqrt(p * (1 - p) / n) + 2e-4
    bad = (freq - p).abs() > band
    assert not bad.any(), (what, bad.nonzero().flatten().tolist()[:5], freq[bad][:5].tolist(), p[bad][:5].tolist())


def test_argmax_segments_allowed_ranges_and_suppressed_ids():
    g = torch.Generator().manual_seed(0)
    sizes = [700, 33, 1025, 64]
    B = 7
    lg = torch.randn(B, sum(sizes) + 5, generator=g).to(DEV)            # a wider buffer than the segments: row stride != width
    allow = [(100, 400), (0, 33), (0, 1025), (10, 11)]
    sup = [120, 5]                                                         # segment-relative ids, applied to every segment
    lg[:, 120] = 50.0                                                      # the suppressed id would win segment 0
    smp = RowSampler(lg.device, sizes, allow=allow, suppress=sup)
    step = torch.zeros(1, dtype=torch.long, device=DEV)
    got = smp(lg, step)
    off = 0
    for s, (n, (lo, hi)) in enumerate(zip(sizes, allow)):
        x = lg[:, off:off + n].clone()
        x[:, :lo] = float("-inf")
        x[:, hi:] = float("-inf")
        for t in sup:
            if t < n:
                x[:, t] = float("-inf")
        assert torch.equal(got[:, s], x.argmax(-1)), s
        off += n
    # first maximum wins (torch.argmax on the host: sample_next's contract)
    tie = torch.tensor([[1.0, 3.0, 3.0, 2.0]], device=DEV)
    assert int(RowSampler(tie.device, [4])(tie, step)) == 1


@pytest.mark.parametrize("V,top_k,top_p,temp", [(200, 50, 0.95, 0.8), (8193, 20, 0.7, 1.3), (300, 5, 1.0, 1.0), (64, 64, 0.5, 2.0),
                                                (1025, 1, 0.9, 0.7)])
def test_topk_topp_draws_follow_the_warper_chain(V, top_k, top_p, temp):
    g = torch.Generator().manual_seed(V + top_k)
    row = torch.randn(V, generator=g) * 2.0
    row[3] = row[7]                                   # a tie inside the row
    N = 16384
    lg = row.to(DEV).unsqueeze(0).expand(N, V).contiguous()
    smp = RowSampler(lg.device, [V], do_sample=True, top_k=top_k, top_p=top_p, temperature=temp, seed=1234)
    step = torch.tensor([5], dtype=torch.long, device=DEV)
    ids = smp(lg, step)[:, 0]
    p = _exact_probs(row.double(), top_k, top_p, temp)
    _check_freq(ids, p, f"V={V} k={top_k} p={top_p} T={temp}")
    if top_k == 1:
        assert (ids == int(torch.argmax(row))).all()
    # (seed, step) fixes the draws; another step or seed changes them
    assert torch.equal(ids, smp(lg, step)[:, 0])
    if top_k > 1:
        assert not torch.equal(ids, smp(lg, step + 1)[:, 0])
        other = RowSampler(lg.device, [V], do_sample=True, top_k=top_k, top_p=top_p, temperature=temp, seed=1235)
        assert not torch.equal(ids, other(lg, step)[:, 0])


def test_ties_at_the_kth_value_stay_in_the_candidate_set():
    # TopKLogitsWarper removes `scores < kth value`: with three equal values at rank k all of them can be drawn
    row = torch.tensor([5.0, 1.0, 4.0, 4.0, 4.0, 0.0, -1.0])
    N = 8192
    lg = row.to(DEV).unsqueeze(0).expand(N, -1).contiguous()
    ids = RowSampler(lg.device, [7], do_sample=True, top_k=2, seed=3)(lg, torch.zeros(1, dtype=torch.long, device=DEV))[:, 0]
    _check_freq(ids, _exact_probs(row.double(), 2, 1.0, 1.0), "ties")
    assert set(ids.unique().tolist()) == {0, 2, 3, 4}


def test_clustered_and_constant_rows_take_the_exact_fallback_paths():
    """select_bins bins the values linearly between the row's minimum and maximum; a row whose values sit in one bin (one far
    outlier below, everything else within 1e-2) has more than 256 elements at or above the k-th value's bin and goes through the
    radix select on the order-preserving bit image instead -- same exact candidate set.  A constant row (every id tied at the k-th
    value) goes one level further, to the round-per-candidate form, which caps the candidate list at 128 ids (smallest first)."""
    g = torch.Generator().manual_seed(5)
    V, N = 8193, 16384
    row = torch.randn(V, generator=g) * 1e-2 + 10.0
    row[100] = -1000.0
    lg = row.to(DEV).unsqueeze(0).expand(N, V).contiguous()
    step = torch.zeros(1, dtype=torch.long, device=DEV)
    ids = RowSampler(lg.device, [V], do_sample=True, top_k=20, top_p=0.9, temperature=0.01, seed=8)(lg, step)[:, 0]
    _check_freq(ids, _exact_probs(row.double(), 20, 0.9, 0.01), "clustered")
    flat = torch.full((64, 1000), 3.0, device=DEV)
    ids = RowSampler(flat.device, [1000], do_sample=True, top_k=5, seed=1)(flat, step)[:, 0]
    assert ((ids >= 0) & (ids < 128)).all() and ids.unique().numel() > 8
    assert (RowSampler(flat.device, [1000])(flat, step) == 0).all()          # argmax: the first of the equal maxima


def test_plain_multinomial_over_a_wide_row_and_eight_channel_frame():
    g = torch.Generator().manual_seed(4)
    V = 5000
    row = torch.randn(V, generator=g) * 3.0
    N = 32768
    lg = row.to(DEV).unsqueeze(0).expand(N, V).contiguous()
    ids = RowSampler(lg.device, [V], do_sample=True, seed=9)(lg, torch.ones(1, dtype=torch.long, device=DEV))[:, 0]
    _check_freq(ids, row.double().softmax(-1), "multinomial")
    # XY frame shape: channel 0 = 66 661 ids of which [65536, 66561) may be drawn, seven channels of 1025; different rows and
    # channels must draw independently (no shared random number)
    sizes = [66661] + [1025] * 7
    B = 16
    lg = torch.randn(B, sum(sizes), generator=g).to(DEV)
    smp = RowSampler(lg.device, sizes, allow=[(65536, 66561)] + [(0, 1025)] * 7, do_sample=True, top_k=50, top_p=0.95, temperature=0.8, seed=2)
    step = torch.zeros(1, dtype=torch.long, device=DEV)
    fr = smp(lg, step)
    assert fr.shape == (B, 8) and ((fr[:, 0] >= 65536) & (fr[:, 0] < 66561)).all() and (fr[:, 1:] < 1025).all()
    off = 0
    for s, n in enumerate(sizes):
        x = lg[:, off:off + n]
        if s == 0:
            x = x.clone()
            x[:, :65536] = float("-inf")
            x[:, 66561:] = float("-inf")
        top = torch.topk(x / 0.8, 50, -1).indices
        assert (top == fr[:, s:s + 1]).any(-1).all(), s
        off += n
    same = lg[:1].expand(B, -1).contiguous()         # identical rows: the draws must still differ between rows
    fr2 = smp(same, step)
    assert len({tuple(r) for r in fr2.tolist()}) > B // 2


def test_small_segments_k_larger_than_the_segment_and_single_rows():
    step = torch.zeros(1, dtype=torch.long, device=DEV)
    row = torch.tensor([0.3, -1.0, 2.0, 0.1, 1.5])
    N = 8192
    lg = row.to(DEV).unsqueeze(0).expand(N, -1).contiguous()
    ids = RowSampler(lg.device, [5], do_sample=True, top_k=50, top_p=0.9, seed=4)(lg, step)[:, 0]      # k > number of ids
    _check_freq(ids, _exact_probs(row.double(), 50, 0.9, 1.0), "k > n")
    one = torch.tensor([[7.0]], device=DEV)
    assert int(RowSampler(one.device, [1], do_sample=True, top_k=3, seed=1)(one, step)) == 0
    assert int(RowSampler(one.device, [1], do_sample=True, seed=1)(one, step)) == 0
    # one row, three segments of different widths, one of them with -inf entries (masked ids must never be drawn)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(1, 300 + 7 + 40, generator=g).to(DEV)
    x[0, 300:305] = float("-inf")
    smp = RowSampler(x.device, [300, 7, 40], do_sample=True, top_k=4, seed=2)
    seen = set()
    for t in range(200):
        step.fill_(t)
        r = smp(x, step)[0].tolist()
        assert 0 <= r[0] < 300 and r[1] in (5, 6) and 0 <= r[2] < 40
        seen.add(tuple(r))
    assert len(seen) > 20


def test_unsupported_requests_are_refused():
    assert RowSampler.supported(torch.device(DEV), [20000]) is not None                                   # segment too wide for LDS
    assert RowSampler.supported(torch.device(DEV), [100], do_sample=True, top_k=0, top_p=0.9) is not None   # top-p needs the full sort
    assert RowSampler.supported(torch.device(DEV), [100], do_sample=True, top_k=65) is not None
    assert RowSampler.supported(torch.device(DEV), [66661], allow=[(65536, 66561)], do_sample=True, top_k=50) is None
    with pytest.raises(ValueError):
        RowSampler(torch.device(DEV), [100], do_sample=True, top_k=0, top_p=0.9)


def _ras_exact(logp64, recent, ignore_eos, eos, top_p, top_k, win, tau_r):
    """probability of every output id of ras_sampling + EOS rejection (the semantics of cosy_llm.ras_sampling_device)"""
    probs = logp64.softmax(0)
    sv, si = probs.sort(descending=True, stable=True)
    cum_before = sv.cumsum(0) - sv
    keep = ((cum_before < top_p) & (torch.arange(sv.numel()) < top_k)).long().cumprod(0).double()
    pk = sv * keep
    if ignore_eos:
        pk = pk.masked_fill(si == eos, 0.0)
        if pk.sum() == 0:
            pk = sv.masked_fill(si == eos, 0.0)
    pc = torch.zeros_like(probs).scatter(0, si, pk / pk.sum())          # P(candidate = id)
    full = probs.clone()
    if ignore_eos:
        full[eos] = 0
    full = full / full.sum()
    rep = torch.tensor([(recent == i).sum().item() for i in range(probs.numel())])
    redo = (rep >= win * tau_r).double()
    return pc * (1 - redo) + (pc * redo).sum() * full


@pytest.mark.parametrize("case", ["plain", "ignore_eos", "eos_alone", "repeat"])
def test_ras_step_distribution_and_bookkeeping(case):
    V, eos, win = 51, 50, 10
    g = torch.Generator().manual_seed(11)
    lg = torch.randn(V, generator=g) * 1.5
    recent0 = torch.full((win,), -1, dtype=torch.long)
    n_ignore = 0
    if case == "ignore_eos":
        lg[eos] = lg.max() + 0.5          # EOS is in the nucleus but must be rejected
        n_ignore = 10 ** 9
    if case == "eos_alone":
        lg[eos] = lg.max() + 12.0         # the nucleus is EOS alone: the draw comes from the full distribution without EOS
        n_ignore = 10 ** 9
    if case == "repeat":
        recent0[:3] = torch.tensor([int(lg.argmax()), 4, int(lg.argmax())])   # the likeliest candidate already occurred: random_sampling
    p = _ras_exact(lg.double(), recent0, n_ignore > 0, eos, 0.8, 25, win, 0.1)
    lgd = lg.to(DEV)
    tok = torch.zeros(1, dtype=torch.long, device=DEV)
    recent, ptr, step_i = recent0.to(DEV), torch.tensor([3], device=DEV), torch.tensor(0, device=DEV)
    r0 = recent.clone()
    N = 12000
    ids = torch.empty(N, dtype=torch.long, device=DEV)
    for i in range(N):
        recent.copy_(r0)
        ptr.fill_(3)
        ras_step(lgd, tok, recent, ptr, step_i, n_ignore, eos, seed=77)
        ids[i] = tok[0]
    assert int(step_i) == N                                   # the step index advanced once per call (and keyed the draws)
    _check_freq(ids, p, case)
    # bookkeeping of the last call: an emitted id goes into the ring at ptr, ptr moves on; EOS is not appended
    last = int(tok)
    if last != eos:
        assert int(recent[3]) == last and int(ptr) == 4
    else:
        assert torch.equal(recent, r0) and int(ptr) == 3
    if case == "plain":
        a = []
        for seed in (5, 5, 6):
            step_i.fill_(0)
            recent.copy_(r0)
            ptr.fill_(3)
            out = []
            for _ in range(40):
                ras_step(lgd, tok, recent, ptr, step_i, 0, eos, seed=seed)
                out.append(int(tok))
            a.append(out)
        assert a[0] == a[1] and a[0] != a[2]
