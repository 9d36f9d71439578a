"""CPU, world sizes 2 and 8 over gloo: the data-parallel machinery of rwkvtts_amd/trainer.py (flat buffers, bucketed
all-reduce from backward hooks, NaN flag, fp32 master AdamW) on a model-agnostic toy network.  The HIP model
itself cannot run on CPU (by design), the exchange logic can."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from rwkvtts_amd import trainer


class Toy(torch.nn.Module):
    def __init__(self):
        super().__init__()
        torch.manual_seed(0)
        self.a = torch.nn.Linear(16, 64)
        self.b = torch.nn.Linear(64, 64)
        self.c = torch.nn.Linear(64, 4)
        self.unused = torch.nn.Parameter(torch.zeros(7))  # never touched by forward: its bucket must still reduce
        # a parameter with the NAME the reference's optimizer groups key on ('attn.w_lora.lora.2.bias' -> lr x 2,
        # train_cosy_rwkv7speech_multiple_dataset.py:169) and LoRA weights (never weight-decayed, :171)
        self.attn = torch.nn.Module()
        self.attn.w_lora = torch.nn.Module()
        self.attn.w_lora.lora = torch.nn.Sequential(torch.nn.Linear(64, 8, bias=False), torch.nn.Tanh(), torch.nn.Linear(8, 64))

    def forward(self, x, y, poison=False, lora_first=False):
        h = torch.tanh(self.a(x))
        if lora_first:   # same value, but the autograd nodes are created (hence run backwards) in the other order
            lo = self.attn.w_lora.lora(h)
            h = torch.tanh(self.b(h) + lo)
        else:
            h = torch.tanh(self.b(h) + self.attn.w_lora.lora(h))
        loss = torch.nn.functional.mse_loss(self.c(h), y)
        if poison:
            loss = loss * float("nan")
        return type("O", (), {"loss": loss})()


def _data(rank, step):
    g = torch.Generator().manual_seed(100 * step + rank)
    return torch.randn(8, 16, generator=g), torch.randn(8, 4, generator=g)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q, shard=False, bucket_opt=True):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    trainer.init_distributed("gloo")
    torch.set_num_threads(1)
    model = Toy()
    tr = trainer.DataParallelTrainer(model, lr=1e-2, warmup_steps=0, total_steps=100, bucket_bytes=4096, shard_optimizer=shard,
                                     bucket_optimizer=bucket_opt)
    if shard:   # the slabs tile the flat buffer, and a bucket really is split between owners
        slabs = [tr.reducer.slab(r) for r in range(world)]
        assert slabs[0][0] == 0 and slabs[-1][1] == tr.flat.numel and all(a[1] == b[0] for a, b in zip(slabs, slabs[1:]))
        assert all(a[1] % 128 == 0 for a in slabs[:-1])
        assert any(s < slabs[0][1] < e for s, e, _ in tr.reducer.buckets)
    assert len(tr.reducer.buckets) > 2  # several buckets -> hooks fire in backward order
    losses = []
    names = [n for n, p_ in model.named_parameters() if p_.requires_grad]
    first_cut = [list(r) for r in tr.reducer.runs]
    for step in range(3):
        x, y = _data(rank, step)
        # step 0: rank 1 builds its graph in another order (what differently shaped batches do to the fused paths of the real
        # model), so ITS gradient-ready order differs from rank 0's -- the buckets must still be cut identically (rank 0's order
        # is broadcast, trainer.BucketedAllReduce.rebuild_from_ready_order)
        own_order = []
        if step == 0:
            orig = tr.reducer._ready
            tr.flat.on_ready = lambda i, o=orig: (own_order.append(i), o(i))
        losses.append(float(tr.step(x=x, y=y, lora_first=(rank == 1 and step == 0))))
        if step == 0:
            tr.flat.on_ready = tr.reducer._ready
            assert tr.reducer.order_differs_from_rank0 == (rank == 1), (rank, own_order, tr.reducer.ready_order)
            # after the first backward pass the buckets are re-cut along the order the gradient hooks fired in (trainer.BucketedAllReduce.
            # rebuild_from_ready_order): the LoRA is registered AFTER `c` but used BEFORE it, so registration order put its gradients
            # (ready late) into the bucket that should open the exchange.  Now: `c` first, `a` last among the used ones, the never-used
            # parameter at the very end; runs tile the flat buffer exactly once.
            r = tr.reducer
            assert r.rebuilt and [list(x_) for x_ in r.runs] != first_cut
            order = [names[i] for i in r.ready_order]
            assert order[0].startswith("c.") and order[-1].startswith("a.") and "unused" not in order
            assert names[[i for i in range(len(names)) if r.param_bucket[i] == 0][0]].split(".")[0] in ("c", "b", "attn")
            assert r.param_bucket[names.index("c.bias")] == 0 and r.param_bucket[names.index("unused")] == len(r.buckets) - 1
            pieces = sorted(x_ for runs in r.runs for x_ in runs)
            assert pieces[0][0] == 0 and pieces[-1][1] == tr.flat.numel and all(a_[1] == b_[0] for a_, b_ in zip(pieces, pieces[1:]))
            assert any(len(runs) > 1 for runs in r.runs)   # a bucket of several runs (non-adjacent slices) is exercised
    cut = [list(x_) for runs in tr.reducer.runs for x_ in runs]
    before = tr.flat.flat_param.clone()
    x, y = _data(rank, 3)
    tr.step(x=x, y=y, poison=(rank == 1))  # NaN on ONE rank: flag all_reduce(MAX) -> both ranks take the zero-grad step
    q.put((rank, tr.flat.flat_param.numpy().copy(), before.numpy().copy(), losses, cut))  # by value, not shared-memory handles
    dist.barrier()
    dist.destroy_process_group()


def _run_world(world, shard, bucket_opt):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, shard, bucket_opt)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    reps = [(torch.from_numpy(a), torch.from_numpy(b)) for _, a, b, _, _ in res]
    for pr, br in reps[1:]:
        assert torch.equal(reps[0][0], pr) and torch.equal(reps[0][1], br), "replicas diverged"
    assert all(r[4] == res[0][4] for r in res), "the ranks re-cut their buckets differently"
    p0, b0 = reps[0]
    # single-process reference: torch.optim.AdamW on an untouched copy of the model, fed the gradient of the mean of the ranks'
    # losses (= the average of the ranks' gradients)
    model = Toy()
    opt = torch.optim.AdamW(model.parameters(), lr=1e-2, betas=(0.9, 0.95), eps=1e-18, weight_decay=0.0)
    for step in range(3):
        opt.zero_grad()
        loss = sum(model(*_data(r, step)).loss for r in range(world)) / world
        loss.backward()
        model.unused.grad = torch.zeros_like(model.unused)
        for g_ in opt.param_groups:
            g_["lr"] = trainer.linear_warmup_decay(step, 100, 0, 1e-2, 1e-5)
        opt.step()
    ref = torch.cat([torch.nn.functional.pad(p.detach().reshape(-1), (0, (-p.numel()) % 128)) for p in model.parameters()])
    assert torch.allclose(ref, b0, atol=1e-6), (ref - b0).abs().max()
    assert torch.isfinite(p0).all()  # the poisoned step did not write NaNs into the weights


@pytest.mark.timeout(300)
@pytest.mark.parametrize("shard,bucket_opt", [(False, True), (False, False), (True, True)])
def test_two_rank_gloo_matches_single_process_average(shard, bucket_opt):
    """bucket_opt: AdamW bucket by bucket as each bucket's all-reduce completes (the default) vs one pass after the last one --
    the same parameters either way.  shard=False: bucketed all-reduce + replicated AdamW.  shard=True (DataParallelTrainer(shard_optimizer=True), SURVEY H6's
    fallback): gradient pieces reduced to the slab owners, AdamW on the own slab only, parameter slabs broadcast -- the replicas
    must come out identical to each other and to the single-process reference in both modes, NaN step included."""
    _run_world(2, shard, bucket_opt)


@pytest.mark.timeout(600)
@pytest.mark.parametrize("shard", [False, True])
def test_eight_rank_gloo_matches_single_process_average(shard):
    """BASELINE configs[2]'s world size (8 ranks) over gloo: eight replicas identical to each other and to single-process AdamW on the
    mean of the eight gradients; rank 1 records another gradient-ready order on step 0, one rank poisons a step with NaN; with
    shard=True eight slab owners tile the flat buffer."""
    _run_world(8, shard, True)


def test_reference_param_groups_and_cosine_schedule_match_torch_adamw():
    """Grouped optimizer + cosine schedule of the Cosy trainer (train_cosy_rwkv7speech_multiple_dataset.py:162-202,224-244):
    lr_2x for 'attn.w_lora.lora.2.bias', weight decay on >= 2-D non-LoRA '.weight', everything else plain -- our flat-buffer
    update against torch.optim.AdamW built with the same three groups, five steps, lr driven by the cosine schedule."""
    wd = 0.1
    m1, m2 = Toy(), Toy()
    tr = trainer.DataParallelTrainer(m1, lr=1e-2, lr_final=1e-3, warmup_steps=2, total_steps=5, weight_decay=wd,
                                     param_groups="reference", schedule="cosine", nan_guard=True)
    names = [n for n, _ in m2.named_parameters()]
    groups = dict(zip(names, trainer.reference_param_groups(m2, wd)))
    assert groups["attn.w_lora.lora.2.bias"] == ("lr_2x", 2.0, 0.0)
    assert groups["a.weight"] == ("lr_decay", 1.0, wd) and groups["a.bias"] == ("lr_1x", 1.0, 0.0)
    assert groups["attn.w_lora.lora.0.weight"] == ("lr_1x", 1.0, 0.0) and groups["attn.w_lora.lora.2.weight"][0] == "lr_1x"
    assert groups["unused"][0] == "lr_1x"
    by = {}
    for n, p in m2.named_parameters():
        by.setdefault(groups[n], []).append(p)
    opt = torch.optim.AdamW([{"params": ps, "weight_decay": g[2], "my_lr_scale": g[1]} for g, ps in by.items()],
                            lr=1e-2, betas=(0.9, 0.95), eps=1e-18)
    for step in range(5):
        x, y = _data(0, step)
        base = trainer.cosine_warmup_decay(step, 5, 2, 1e-2, 1e-3)
        assert tr.group_lrs() == {"lr_1x": base, "lr_2x": 2 * base, "lr_decay": base}
        tr.step(x=x, y=y)
        opt.zero_grad()
        m2(x, y).loss.backward()
        m2.unused.grad = torch.zeros_like(m2.unused)
        for g_ in opt.param_groups:
            g_["lr"] = base * g_["my_lr_scale"]
        opt.step()
    for (n, a), b in zip(m1.named_parameters(), m2.parameters()):
        assert torch.allclose(a, b, atol=2e-6, rtol=1e-5), (n, (a - b).abs().max())


def test_cosine_schedule_matches_reference_function_values():
    """Golden values: tests/golden/lr_schedule.npz holds what the reference's own update_learning_rate
    (train_cosy_rwkv7speech_multiple_dataset.py:224-244, executed by oracle/pin_optimizer.py) wrote into the three param groups."""
    from conftest import load_golden
    g = load_golden("lr_schedule.npz")
    total, warm, lr, lr_final = int(g["total_steps"]), int(g["warmup_steps"]), float(g["lr"]), float(g["lr_final"])
    for i, step in enumerate(g["steps"].tolist()):
        base = trainer.cosine_warmup_decay(step, total, warm, lr, lr_final)
        assert abs(base - float(g["lr_1x"][i])) <= 1e-12 * max(1.0, abs(base))
        assert abs(2 * base - float(g["lr_2x"][i])) <= 1e-12
        assert abs(base - float(g["lr_decay"][i])) <= 1e-12


def test_flat_buffers_alias_parameters_and_grads():
    m = Toy()
    fb = trainer.FlatBuffers(m)
    x, y = _data(0, 0)
    m(x, y).loss.backward()
    for p, o in zip(fb.params, fb.offsets):
        assert p.data_ptr() == fb.flat_param.data_ptr() + o * 4
        if p is not m.unused:
            assert p.grad.data_ptr() == fb.flat_grad.data_ptr() + o * 4
    assert fb.flat_grad.abs().sum() > 0
    fb.zero_grad()
    assert fb.flat_grad.abs().sum() == 0


def test_lr_schedule_matches_reference_lambda():
    # train_spark_rwkv7speech.py:219-232
    assert trainer.linear_warmup_decay(0, 100, 10, 1.0, 0.1) == 0.0
    assert trainer.linear_warmup_decay(5, 100, 10, 1.0, 0.1) == 0.5
    assert abs(trainer.linear_warmup_decay(55, 100, 10, 1.0, 0.1) - (1 - 0.5 * 0.9)) < 1e-12
    assert trainer.linear_warmup_decay(1000, 100, 10, 1.0, 0.1) == 0.1


def test_param_groups_match_reference_configure_optimizer_golden():
    """tests/golden/optimizer_groups.npz: the groups the reference's own configure_optimizer assigned to every parameter of a
    small RWKV7CosyLM (names are rwkvfla's, matched by substring as the reference does), for weight_decay 0 and 0.1."""
    import numpy as np
    import os
    from conftest import GOLDEN
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(GOLDEN), "..", "oracle"))
    from oracle.pin_optimizer import small_cosy_model
    z = np.load(os.path.join(GOLDEN, "optimizer_groups.npz"))
    model = small_cosy_model()
    names = [n for n, p in model.named_parameters() if p.requires_grad]
    for tag, wd in (("wd0", 0.0), ("wd", 0.1)):
        assert names == z[f"names_{tag}"].tolist()
        ours = trainer.reference_param_groups(model, wd)
        assert [g[0] for g in ours] == z[f"group_{tag}"].tolist()
        assert [g[1] for g in ours] == z[f"scale_{tag}"].tolist() and [g[2] for g in ours] == z[f"decay_{tag}"].tolist()
    assert "lr_2x" in z["group_wd0"].tolist() and "lr_decay" in z["group_wd"].tolist()
