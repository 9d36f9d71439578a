"""CPU, world size 2 over gloo: the data-parallel machinery of rwkvtts_amd/trainer.py (flat buffers, bucketed
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

    def forward(self, x, y, poison=False):
        h = torch.tanh(self.b(torch.tanh(self.a(x))))
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


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    trainer.init_distributed("gloo")
    torch.set_num_threads(1)
    model = Toy()
    tr = trainer.DataParallelTrainer(model, lr=1e-2, warmup_steps=0, total_steps=100, bucket_bytes=4096)
    assert len(tr.reducer.buckets) > 2  # several buckets -> hooks fire in backward order
    losses = []
    for step in range(3):
        x, y = _data(rank, step)
        losses.append(float(tr.step(x=x, y=y)))
    before = tr.flat.flat_param.clone()
    x, y = _data(rank, 3)
    tr.step(x=x, y=y, poison=(rank == 1))  # NaN on ONE rank: flag all_reduce(MAX) -> both ranks take the zero-grad step
    q.put((rank, tr.flat.flat_param.numpy().copy(), before.numpy().copy(), losses))  # by value, not shared-memory handles
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_two_rank_gloo_matches_single_process_average():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=150) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    (_, p0, b0, _), (_, p1, b1, _) = [(r, torch.from_numpy(a), torch.from_numpy(b), l) for r, a, b, l in res]
    assert torch.equal(p0, p1) and torch.equal(b0, b1), "replicas diverged"
    # single-process reference: average of the two ranks' gradients == gradient of the mean of the two losses
    model = Toy()
    tr = trainer.DataParallelTrainer(model, lr=1e-2, warmup_steps=0, total_steps=100)
    for step in range(3):
        tr.flat.zero_grad()
        loss = sum(model(*_data(r, step)).loss for r in range(2)) / 2
        loss.backward()
        for g_ in tr.opt.param_groups:
            g_["lr"] = trainer.linear_warmup_decay(tr.step_idx, 100, 0, 1e-2, 1e-5)
        tr.opt.step()
        tr.step_idx += 1
    assert torch.allclose(tr.flat.flat_param, b0, atol=1e-6), (tr.flat.flat_param - b0).abs().max()
    assert torch.isfinite(p0).all()  # the poisoned step did not write NaNs into the weights


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
