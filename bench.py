#!/usr/bin/env python
"""bench.py -- the headline metric of BASELINE.json on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...)

A "step" is one full training step of the hot path over one synthetic batch: RWKV7-0.4B, Spark layout,
B=8 sequences of L=4096 text+speech-token positions per GPU (BASELINE.json configs[1]; configs[2] for N>1):
embedding lookups -> 24 RWKV-7 layers (HIP WKV7 fwd/bwd + fused stages, library GEMMs) -> fused linear+CE ->
backward -> bucketed RCCL gradient all-reduce (N>1) -> AdamW on fp32 master weights.  Nothing is skipped.

Prints ONE JSON line (rank 0): metric/value/unit per BASELINE.json, plus
  roofline     : the dominant kernel (WKV7 backward): algorithmic bytes per launch (13*64*2 B per token-head,
                 SURVEY.md section 8d) / its average launch duration measured live with HIP events on the launch
                 stream, against the 8 TB/s HBM peak (MI355X_MICROARCH.md).  `traffic` = HBM bytes per launch
                 from the committed PMC pass (profiles/pmc_wkv7.json: 2*FETCH_SIZE + WRITE_SIZE, KiB), else null.
  cpu_baseline : the oracle's eager-PyTorch fp32 CPU restatement of the same training step (oracle/rwkv7_ref.py,
                 the reference's PyTorch-CPU path) timed on this box's host cores on a bounded sample.
  decode       : (N = 1 only, after the timed region, not part of `value`) greedy decode of the same model at
                 BASELINE.json configs[4] -- B=32, prompt 128 -- tokens/s and ms per step (decode.GraphDecoder).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12  # B/s, MI355X_MICROARCH.md
WKV_BWD_BYTES_PER_TOKEN_HEAD = 13 * 64 * 2  # read w,q,k,v,a,b,dy ; write 6 grads ; bf16
WKV_FWD_BYTES_PER_TOKEN_HEAD = 7 * 64 * 2


def cpu_baseline(seconds_budget=25.0):
    """Reference PyTorch-CPU path (the oracle's restatement), bounded sample of the same workload:
    0.4B Spark model, fp32, fwd+bwd on B=1 sequences of T=128."""
    from oracle import rwkv7_ref as R
    torch.manual_seed(0)
    # the per-token scan is thousands of tiny ops: beyond ~16 threads the fork/join cost of each op dominates
    cores = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(cores)
    cfg = R.RefConfig(hidden_size=1024, num_hidden_layers=24, vocab_size=8193)
    p = R.init_params(cfg, seed=0)
    p["lm_head.weight"] = torch.randn(8193, 1024) * 0.02
    for v in p.values():
        v.requires_grad_(True)
    B, T = 1, 32
    x = torch.randn(B, T, 1024) * 0.5
    labels = torch.randint(0, 8192, (B, T))
    t0 = time.time()
    n = 0
    while True:
        loss, _, _ = R.spark_forward(p, cfg, x, None, labels)
        loss.backward()
        n += 1
        if time.time() - t0 > seconds_budget * 0.6:
            break
    dt = time.time() - t0
    return {"value": round(n * B * T / dt, 2), "unit": "tokens/s", "cores": cores, "kind": "port",
            "sample": f"RWKV7-0.4B Spark fwd+bwd, fp32 eager PyTorch on CPU (oracle/rwkv7_ref.py, per-token torch scan), "
                      f"B={B} T={T}, {n} steps in {dt:.1f}s"}


def decode_rate(model, dev, B=32, P=128, n1=64, n2=448):
    """BASELINE.json configs[4] on the model that was just trained: greedy decode, B = 32, prompt 128, through
    decode.GraphDecoder (rwkv7_decode_step_bf16 phases replayed from a hipGraph).  Two generate() calls of different length
    separate the per-step time from prefill + capture.  Outside the timed region; rank 0, one GPU only."""
    from rwkvtts_amd.decode import GraphDecoder
    was_training = model.training
    model.eval()
    g = torch.Generator().manual_seed(1234)
    emb = (torch.randn(B, P, model.config.hidden_size, generator=g) * 0.5).to(dev, torch.bfloat16)
    mask = torch.ones(B, P, dtype=torch.long, device=dev)
    eos = model.config.vocab_size - 1   # suppressed, as in the synthetic workload

    def run(n):
        dec = GraphDecoder(model, B)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        dec.generate(inputs_embeds=emb, attention_mask=mask, max_new_tokens=n, suppress_tokens=[eos])
        torch.cuda.synchronize()
        return time.perf_counter() - t0, dec.step is not None

    run(8)
    t1, _ = run(n1)
    t2, kernel = run(n2)
    step = (t2 - t1) / (n2 - n1)
    if was_training:
        model.train()
    return {"metric": "greedy decode tokens/s, RWKV7-0.4B B=32 prompt=128 (BASELINE.json configs[4])", "value": round(B / step, 1),
            "unit": "tokens/s", "ms_per_step": round(step * 1e3, 4), "batch": B, "prompt": P, "new_tokens": n2,
            "path": "rwkv7_decode_step_bf16, one launch per phase, hipGraph replay" if kernel else "module by module, hipGraph replay"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=8, help="sequences per GPU")
    ap.add_argument("--seq-len", type=int, default=4096)
    ap.add_argument("--model", default="0.4b", choices=["0.1b", "0.4b", "1.5b"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--grad-checkpoint", action="store_true")
    ap.add_argument("--no-decode", action="store_true", help="skip the greedy-decode measurement (BASELINE configs[4]) after the timed region")
    ap.add_argument("--scalar-wkv-fwd", action="store_true", help="A/B: scalar WKV7 forward instead of the chunked MFMA one")
    ap.add_argument("--scalar-wkv-bwd", action="store_true", help="A/B: row-split scalar WKV7 backward instead of the chunked MFMA one")
    a = ap.parse_args()
    if a.scalar_wkv_bwd:
        from rwkvtts_amd import fused as _fused
        _fused.CHUNKED_WKV_BWD = False
    if a.scalar_wkv_fwd:
        from rwkvtts_amd import fused as _fused
        _fused.CHUNKED_WKV_FWD = False

    from rwkvtts_amd import build
    if int(os.environ.get("LOCAL_RANK", "0")) == 0:
        build.build()  # no-op when the prebuilt .so is current; one rank per node, the others wait at the barrier below
    from rwkvtts_amd import backbone, ops, trainer
    from rwkvtts_amd.layouts import synthetic_spark_batch
    from rwkvtts_amd.spark_llm import RWKV7ForSpeech, RWKV7SpeechConfig

    rank, local_rank, world = trainer.init_distributed()
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        torch.distributed.barrier()

    base = {"0.1b": backbone.config_0p1b, "0.4b": backbone.config_0p4b, "1.5b": backbone.config_1p5b}[a.model]()
    cfg = RWKV7SpeechConfig(**{k: v for k, v in base.to_dict().items() if k in backbone.RWKV7Config.__dataclass_fields__
                               and k != "extra"})
    model = RWKV7ForSpeech(cfg).init_weights(seed=0).to(device=dev, dtype=torch.bfloat16).train()
    if a.grad_checkpoint:
        model.gradient_checkpointing_enable()
    tr = trainer.DataParallelTrainer(model, lr=1e-4, warmup_steps=10, total_steps=1000)
    B, T = a.batch, a.seq_len
    H = cfg.num_heads

    def one_step(i):
        with torch.no_grad():
            batch = synthetic_spark_batch(model, B, T, seed=1234 + rank + 1000 * i)
        batch["inputs_embeds"] = batch["inputs_embeds"].detach()
        return tr.step(**batch)

    def sync():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    def log(msg):
        if rank == 0:
            print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)

    log(f"model built: {sum(p.numel() for p in model.parameters()) / 1e6:.1f} M params, B={B} T={T}")
    for i in range(a.warmup):
        loss = one_step(i)
        torch.cuda.synchronize()
        log(f"warmup step {i} done, loss {float(loss):.4f}, mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
    sync()
    ops.KERNEL_TIMERS = {}
    t0 = time.perf_counter()
    for i in range(a.steps):
        loss = one_step(a.warmup + i)
    sync()
    dt = time.perf_counter() - t0
    timers, ops.KERNEL_TIMERS = ops.KERNEL_TIMERS, None
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        dt = float(tt.item())
    loss_val = float(loss.item())

    if rank == 0:
        ms = dt / a.steps * 1e3
        value = world * B * T * a.steps / dt
        kern = {k: sum(s.elapsed_time(e) for s, e in v) / len(v) for k, v in timers.items() if v}
        th = B * T * H
        bwd_ms = kern.get("wkv7_bwd")
        bwd_name = "wkv7_bwd_kernel<bf16,2,4> (row-split scalar WKV7 backward, 24 launches/step)"
        pmc_key = "wkv7_bwd"
        chunk_parts = ["wkv7c_bwd_pre", "wkv7c_state", "wkv7c_bwd_out"]
        fwd_ms, fwd_name = kern.get("wkv7_fwd"), "wkv7_fwd_kernel (scalar)"
        if "wkv7c_fwd" in kern:   # chunked forward: T^-1 (wkv7c_prep, reused by the backward) + the chunk kernel
            fwd_ms, fwd_name = kern["wkv7c_fwd"] + kern.get("wkv7c_prep", 0.0), "wkv7c_prep + wkv7c_fwd (chunked MFMA)"
        if bwd_ms is None and all(k in kern for k in chunk_parts):
            # chunked MFMA backward: three launches per layer (four when the forward was scalar and T^-1 is computed here)
            bwd_ms = sum(kern[k] for k in chunk_parts) + (0.0 if "wkv7c_fwd" in kern else kern.get("wkv7c_prep", 0.0))
            bwd_name = "WKV7 backward, chunked MFMA (wkv7c_bwd_pre + wkv7c_state + wkv7c_bwd_out, 24x per step)"
            pmc_key = "wkv7c_bwd"
        achieved = th * WKV_BWD_BYTES_PER_TOKEN_HEAD / (bwd_ms * 1e-3) / 1e9 if bwd_ms else None
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_wkv7.json")
        if os.path.exists(pmc):
            try:
                d = json.load(open(pmc))[pmc_key]
                if d.get("B") == B and d.get("T") == T and d.get("H") == H:
                    traffic = int((2 * d["FETCH_SIZE_KiB"] + d["WRITE_SIZE_KiB"]) * 1024)
            except Exception:
                traffic = None
        out = {
            "metric": "audio-tokens/sec/GPU (train fwd+bwd) RWKV7-0.4B L=4096; 1->8 GPU scaling",
            "value": round(value, 1), "unit": "tokens/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
            "per_gpu": round(value / world, 1),
            "config": {"workload": f"RWKV7-{a.model.upper()} Spark layout train step (fwd+bwd+AdamW), B={B}/GPU L={T}, "
                                   f"synthetic text+speech tokens, random init (BASELINE.json configs[{1 if world == 1 else 2}])",
                       "global_batch": world * B, "seq_len": T, "parallelism": f"dp{world}",
                       "grad_allreduce": "bucketed RCCL AVG, bf16, overlapped with backward" if world > 1 else "none"},
            "loss": round(loss_val, 4),
            "kernel_ms": {k: round(v, 4) for k, v in kern.items()},
            "roofline": {"kernel": bwd_name, "bound": "hbm",
                         "achieved": round(achieved, 1) if achieved else None, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                         "frac": round(achieved * 1e9 / HBM_PEAK, 4) if achieved else None, "traffic": traffic,
                         "algorithmic_bytes_per_launch": th * WKV_BWD_BYTES_PER_TOKEN_HEAD,
                         "launch_ms": round(bwd_ms, 4) if bwd_ms else None,
                         "fwd": {"kernel": fwd_name, "launch_ms": round(fwd_ms, 4),
                                 "achieved": round(th * WKV_FWD_BYTES_PER_TOKEN_HEAD / (fwd_ms * 1e-3) / 1e9, 1)}
                         if fwd_ms else None},
        }
        if world == 1 and not a.no_decode:
            try:
                out["decode"] = decode_rate(model, dev)
            except Exception as e:  # the headline number must not depend on the secondary measurement
                out["decode"] = {"error": repr(e)}
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
