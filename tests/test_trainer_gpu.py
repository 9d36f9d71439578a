"""GPU (-m gpu): the data-parallel trainer around the REAL bf16 Spark model (reference DP: train_spark_rwkv7speech.py:461-467,
566-572,664-691).

  * one GPU, RCCL group of one rank, collectives forced on: the whole hook -> flush -> bucket all-reduce -> wait path runs with
    the in-place split weight gradients (`_grad_slot`, rwkv7_sum_slabs_bf16) and must leave exactly the parameters the
    collective-free trainer leaves;
  * two ranks, three steps, different data per rank -> replicas bit-identical, and equal (bf16 bar) to one process stepping on
    the mean loss.  Over RCCL on two GPUs (skipped on a one-GPU box; the driver's 8-GPU scaling run is the other user of that
    path), and -- so that the multi-process trainer with the real model and its in-place gradients runs on every GPU box --
    as two processes sharing GPU 0 with the bucket exchange over gloo (RCCL refuses two ranks on one device).
"""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _model(dev):
    from rwkvtts_amd.spark_llm import RWKV7ForSpeech, RWKV7SpeechConfig
    cfg = RWKV7SpeechConfig(vocab_size=257, text_vocab_size=300, audio_global_vocab_size=64, hidden_size=128, num_hidden_layers=2,
                            decay_low_rank_dim=32, a_low_rank_dim=32, v_low_rank_dim=32, gate_low_rank_dim=32)
    m = RWKV7ForSpeech(cfg).init_weights(seed=3).to(dev).to(torch.bfloat16).train()
    m.dropout.p = 0.0
    return m


def _batch(model, rank, step, B=2, T=2048):
    from rwkvtts_amd.layouts import synthetic_spark_batch
    return synthetic_spark_batch(model, B, T, seed=100 * step + rank, n_text=31, n_global=8)   # 4096 rows: split weight gradients


@pytest.fixture(params=[False, True], ids=["wgrad-inline", "wgrad-side-stream"])
def side_stream(request):
    """fused.WGRAD_SIDE_STREAM off (default) and on: the weight gradients written into the flat buffer from a second stream must
    leave exactly the same parameters (the reducer and finish_backward wait for that stream)."""
    from rwkvtts_amd import fused
    old = fused.WGRAD_SIDE_STREAM
    fused.WGRAD_SIDE_STREAM = request.param
    yield request.param
    fused.WGRAD_SIDE_STREAM = old


@pytest.mark.parametrize("shard", [False, True])
def test_forced_allreduce_on_one_rank_equals_plain_trainer(shard, side_stream):
    """shard=True: the sharded-optimizer exchange (RCCL reduce of every bucket piece to its slab owner, AdamW kernel on the own
    slab through offset pointers, broadcast of the parameter slabs) in a group of one rank, where it must be the identity."""
    from rwkvtts_amd import trainer
    dev = torch.device("cuda:0")
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        m1, m2 = _model(dev), _model(dev)
        t1 = trainer.DataParallelTrainer(m1, lr=1e-3, warmup_steps=0, total_steps=10, bucket_bytes=64 << 10, force_allreduce=True,
                                         shard_optimizer=shard)
        assert t1.shard_optimizer == shard
        t2 = trainer.DataParallelTrainer(m2, lr=1e-3, warmup_steps=0, total_steps=10)
        t2.reducer.enabled = False
        assert t1.reducer.enabled and len(t1.reducer.buckets) > 4 and t1.reducer.backend == "nccl"
        from rwkvtts_amd import fused
        for step in range(3):
            l1 = t1.step(**_batch(m1, 0, step))
            fused.WGRAD_SIDE_STREAM = False          # the comparison trainer always computes its weight gradients in line
            l2 = t2.step(**_batch(m2, 0, step))
            fused.WGRAD_SIDE_STREAM = side_stream
            assert torch.equal(l1, l2)
        torch.cuda.synchronize()
        assert torch.equal(t1.flat.flat_param, t2.flat.flat_param), "bucketed RCCL path changed the update"
        assert torch.equal(t1.flat.flat_grad, t2.flat.flat_grad)
    finally:
        dist.destroy_process_group()


def _worker(rank, world, port, q, backend, one_device, shard=False):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(0 if one_device else rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    from rwkvtts_amd import trainer
    trainer.init_distributed(backend)
    dev = torch.device("cuda", 0 if one_device else rank)
    torch.cuda.set_device(dev)
    model = _model(dev)
    tr = trainer.DataParallelTrainer(model, lr=1e-3, warmup_steps=0, total_steps=10, bucket_bytes=64 << 10, shard_optimizer=shard)
    losses = [float(tr.step(**_batch(model, rank, step))) for step in range(3)]
    torch.cuda.synchronize()
    q.put((rank, tr.flat.flat_param.float().cpu().numpy().copy(), losses))
    dist.barrier()
    dist.destroy_process_group()


def _two_ranks(backend, one_device, shard=False):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, backend, one_device, shard)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=500) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    p0, p1 = torch.from_numpy(res[0][1]), torch.from_numpy(res[1][1])
    assert torch.equal(p0, p1), "replicas diverged"
    # one process, mean of the two ranks' losses per step
    from rwkvtts_amd import trainer
    dev = torch.device("cuda:0")
    model = _model(dev)
    tr = trainer.DataParallelTrainer(model, lr=1e-3, warmup_steps=0, total_steps=10)

    class Both(torch.nn.Module):
        def forward(self, b0, b1):
            l = (model(**b0).loss + model(**b1).loss) / 2
            return type("O", (), {"loss": l})()

    tr.model = Both()
    for step in range(3):
        tr.step(b0=_batch(model, 0, step), b1=_batch(model, 1, step))
    ref = tr.flat.flat_param.float().cpu()
    # the two-rank run averages bf16-rounded per-rank gradients, the single process rounds the summed gradient once; Adam's
    # first steps move every weight by ~lr * sign(g), so a gradient whose sign is within the bf16 noise lands 2 lr apart
    d = (p0 - ref).abs()
    assert d.max().item() <= 3 * 2 * 1e-3 + 2e-2 * ref.abs().max().item(), d.max().item()
    assert (d > 1e-3).float().mean().item() < 0.2


@pytest.mark.timeout(600)
@pytest.mark.parametrize("shard", [False, True])
def test_two_ranks_rccl_real_model_replicas_identical_and_equal_single_process_mean(shard):
    """Runs whenever the box has two GPUs (bf16 AVG buckets from the backward hooks, MAX NaN flag; shard=True: the --shard-optimizer
    fall-back, reduce to slab owners + sharded AdamW + broadcast)."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (the GPU test box has one; the same run over gloo on one GPU is the next test)")
    _two_ranks("nccl", False, shard)


@pytest.mark.timeout(600)
@pytest.mark.parametrize("shard", [False, True])
def test_two_ranks_sharing_one_gpu_over_gloo_real_model(shard):
    """shard=True: each of the two processes steps the AdamW kernel on its half of the flat buffers only and receives the other
    half by broadcast."""
    _two_ranks("gloo", True, shard)


DEV = "cuda:0"


def test_nan_loss_step_on_the_hip_model_leaves_parameters_unchanged():
    """train_spark_rwkv7speech.py:664-687: a NaN loss makes every rank step on a zero gradient.  Here backward runs on the NaN
    activations (nothing waits for the flag on the host) and rwkv7_adamw_groups_bf16 reads the flag on the device: the HIP kernels
    see NaN inputs, the parameters and the fp32 masters must come out exactly as after a zero-gradient step, and the next clean
    step must train."""
    from rwkvtts_amd import trainer
    from rwkvtts_amd.spark_llm import RWKV7ForSpeech, RWKV7SpeechConfig
    cfg = RWKV7SpeechConfig(hidden_size=128, num_hidden_layers=2, vocab_size=257, text_vocab_size=300, audio_global_vocab_size=64,
                            decay_low_rank_dim=32, a_low_rank_dim=32, v_low_rank_dim=16, gate_low_rank_dim=32)
    model = RWKV7ForSpeech(cfg).init_weights(seed=0).to(DEV).to(torch.bfloat16).train()
    tr = trainer.DataParallelTrainer(model, lr=1e-3, warmup_steps=0, total_steps=10)
    g = torch.Generator().manual_seed(0)
    x = (torch.randn(2, 64, 128, generator=g) * 0.5).to(DEV, torch.bfloat16)
    labels = torch.randint(0, 256, (2, 64), generator=g).to(DEV)
    l0 = tr.step(inputs_embeds=x, labels=labels)
    assert torch.isfinite(l0)
    before, m_before = tr.flat.flat_param.clone(), tr.master.clone()
    ea, eq = tr.exp_avg.clone(), tr.exp_avg_sq.clone()
    xn = x.clone()
    xn[0, 5, 7] = float("nan")
    ln = tr.step(inputs_embeds=xn, labels=labels)
    assert not torch.isfinite(ln)
    assert torch.isfinite(tr.flat.flat_param.float()).all() and torch.isfinite(tr.master).all()
    assert torch.isfinite(tr.exp_avg).all() and torch.isfinite(tr.exp_avg_sq).all()
    # a zero-gradient AdamW step: moments decay, parameters move only by the (bias-corrected) momentum of the first step
    b1, b2 = tr.betas
    assert torch.allclose(tr.exp_avg, ea * b1, rtol=1e-6, atol=0) and torch.allclose(tr.exp_avg_sq, eq * b2, rtol=1e-6, atol=0)
    l2 = tr.step(inputs_embeds=x, labels=labels)
    assert torch.isfinite(l2) and torch.isfinite(tr.flat.flat_param.float()).all()
