#!/usr/bin/env python
"""bench.py -- the headline metric of BASELINE.json on MI355X.

    python bench.py --gpus N --steps K --warmup W

With N > 1 and no WORLD_SIZE in the environment the script re-executes itself under
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free>`
(one rank per GPU over RCCL); launched under torchrun by somebody else it just joins the group it is given.

A "step" is one full training step of the hot path over one synthetic batch: RWKV7-0.4B, Spark layout,
B=8 sequences of L=4096 text+speech-token positions per GPU (BASELINE.json configs[1]; configs[2] for N>1):
embedding lookups (with their backward) -> 24 RWKV-7 layers (HIP WKV7 fwd/bwd + fused stages, library GEMMs) ->
fused linear+CE -> backward -> bucketed RCCL gradient all-reduce (N>1) -> AdamW on fp32 master weights.
Nothing is skipped and nothing in the step waits for the device (the NaN flag stays on the device, trainer.py).

Prints ONE JSON line (rank 0): metric/value/unit per BASELINE.json, plus
  roofline     : the dominant kernel group (WKV7 backward): algorithmic bytes per launch (13*64*2 B per token-head,
                 SURVEY.md section 8d) / its average launch duration measured live with HIP events on the launch
                 stream, against the 8 TB/s HBM peak (MI355X_MICROARCH.md).  `traffic` = HBM bytes per launch,
                 `mfma_util` / `valu_frac` = MFMA-busy and VALU-issue fractions of the SIMD time, all from the
                 committed PMC pass (profiles/pmc_wkv7.json, tools/pmc_wkv.sh), null when it does not cover
                 this shape.  `fwd` = the forward pair, `fwd_bwd` = forward + backward together (the unit the
                 north-star target of 0.40 is stated on).
  cpu_baseline : the oracle's eager-PyTorch fp32 CPU restatement of the reference's training step on
                 BASELINE.json configs[0] -- RWKV7-0.1B, Cosy layout, B=2, L=512, fwd+bwd, per-token torch scan,
                 all host cores -- plus the WKV scan alone at (B,T,H,N) = (2,512,12,64) through the C oracle
                 (SURVEY.md section 8d); bounded to ~25 s.
  decode       : (N = 1 only, after the timed region, not part of `value`) greedy decode of the same model at
                 BASELINE.json configs[4] -- B=32, prompt 128, 2048 new tokens -- tokens/s, ms per step and the
                 fraction of the weight+state streaming bound (decode.GraphDecoder).
  comm         : (N > 1) bytes all-reduced per step, bucket count, exposed (non-overlapped) wait per step.
"""
import argparse
import json
import os
import socket
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12  # B/s, MI355X_MICROARCH.md
WKV_BWD_BYTES_PER_TOKEN_HEAD = 13 * 64 * 2  # read w,q,k,v,a,b,dy ; write 6 grads ; bf16
WKV_FWD_BYTES_PER_TOKEN_HEAD = 7 * 64 * 2


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _device_info(torch, dev):
    """Name / CU count of the GPU the line was measured on, plus a FINGERPRINT of the box (boxes of one pool differ by several per
    cent in kernel time; lines from different boxes can be normalised by these): the shader clock torch reports right after the
    timed steps, a ~40 ms device-to-device copy probe (1 GiB read + 1 GiB written, GB/s) and a ~40 ms bf16 GEMM probe
    (8192 x 8192 x 8192 through the library, TFLOP/s).  Run AFTER the timed region (called when the JSON line is assembled)."""
    try:
        p = torch.cuda.get_device_properties(dev)
        out = {"name": p.name, "arch": getattr(p, "gcnArchName", ""), "compute_units": p.multi_processor_count,
               "hbm_gib": round(p.total_memory / 2**30)}
    except Exception as e:
        return {"error": repr(e)}
    try:
        out["sclk_mhz_after_steps"] = int(torch.cuda.clock_rate())
    except Exception:   # amdsmi / rocm-smi bindings absent: the probes below still fingerprint the box
        out["sclk_mhz_after_steps"] = None
    try:
        def timed(fn, n):
            fn()
            torch.cuda.synchronize(dev)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(n):
                fn()
            e.record()
            torch.cuda.synchronize(dev)
            return s.elapsed_time(e) / n
        src = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
        dst = torch.empty_like(src)
        ms = timed(lambda: dst.copy_(src), 8)
        out["copy_probe_gbs"] = round(2 * (1 << 30) / ms / 1e6, 0)
        del src, dst
        a = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
        b = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
        ms = timed(lambda: torch.mm(a, b.t()), 40)
        out["gemm8192_probe_tflops"] = round(2 * 8192**3 / ms / 1e9, 0)
        del a, b
    except Exception as e:   # never let the fingerprint take the line down
        out["probe_error"] = repr(e)
    return out



class Watchdog:
    """`with Watchdog(seconds, what, state)`: if the block has not finished after `seconds`, print WHICH rank is stuck WHERE (and
    `state()`, e.g. the gradient buckets on the wire) to stderr and end the process with exit code 3 -- so that a hung first
    multi-GPU run leaves a diagnosis instead of a bare timeout.  One daemon timer thread per block; no cost when it does not fire."""

    def __init__(self, seconds, what, state=None):
        self.seconds, self.what, self.state = seconds, what, state

    def _fire(self):
        rank = os.environ.get("RANK", "0")
        lines = [f"[bench watchdog] rank {rank} (pid {os.getpid()}, LOCAL_RANK {os.environ.get('LOCAL_RANK', '0')}) still in "
                 f"'{self.what}' after {self.seconds:.0f} s"]
        try:
            if self.state is not None:
                lines.append(f"[bench watchdog] rank {rank} state: {self.state()}")
        except Exception as e:
            lines.append(f"[bench watchdog] rank {rank} state unavailable: {e!r}")
        print("\n".join(lines), file=sys.stderr, flush=True)
        try:
            import faulthandler
            faulthandler.dump_traceback(file=sys.stderr, all_threads=True)
        except Exception:
            pass
        os._exit(3)

    def __enter__(self):
        import threading
        self.t = threading.Timer(self.seconds, self._fire)
        self.t.daemon = True
        self.t.start()
        return self

    def __exit__(self, *exc):
        self.t.cancel()
        return False


def _checkin(store, stage, rank):
    if store is not None:
        try:
            store.set(f"bench/{stage}/{rank}", "1")
        except Exception:
            pass


def _missing(store, stage, world):
    """Ranks that have NOT passed `stage` yet (read from the rendezvous TCP store, which does not need the collective backend)."""
    if store is None:
        return "unknown (no store)"
    out = []
    for r in range(world):
        try:
            if not store.check([f"bench/{stage}/{r}"]):
                out.append(r)
        except Exception:
            out.append(r)
    return f"ranks not past '{stage}': {out or 'none'}"


def preflight(torch, rank, local_rank, world, dev, one_device, timeout_s=60.0, log=print):
    """First contact with the collective backend, BEFORE the model is built: who sits where, which RCCL, and one all-reduce of each
    kind the step uses (64 MiB bf16 AVG -- two 32 MiB buckets' worth -- and the 1-element MAX of the NaN flag;
    train_spark_rwkv7speech.py:566-572, 664-670), each under a watchdog that names the ranks that did not arrive.  Returns a dict for
    the JSON line's `comm.preflight`."""
    import torch.distributed as dist
    info = {}
    try:
        store = dist.distributed_c10d._get_default_store()
    except Exception:
        store = None
    backend = dist.get_backend()
    try:
        ver = ".".join(str(x) for x in torch.cuda.nccl.version()) if backend == "nccl" else None
    except Exception as e:
        ver = repr(e)
    try:
        props = torch.cuda.get_device_properties(dev)
        mine = {"rank": rank, "local_rank": local_rank, "device": int(dev.index), "name": props.name,
                "pci": getattr(props, "pci_bus_id", None), "pid": os.getpid()}
    except Exception as e:
        mine = {"rank": rank, "local_rank": local_rank, "error": repr(e)}
    _checkin(store, "start", rank)
    with Watchdog(timeout_s, "preflight: all_gather_object of the rank <-> device map", lambda: _missing(store, "start", world)):
        table = [None] * world
        dist.all_gather_object(table, mine)
    _checkin(store, "map", rank)
    info.update({"backend": backend, "rccl_version": ver, "ranks": table,
                 "env": {k: os.environ.get(k) for k in ("NCCL_DEBUG", "HSA_ENABLE_IPC_MODE_LEGACY", "NCCL_P2P_DISABLE", "NCCL_SOCKET_IFNAME",
                                                        "HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES") if os.environ.get(k) is not None}})
    devs = [t.get("device") for t in table if isinstance(t, dict)]
    if not one_device and len(set(devs)) != world:
        raise SystemExit(f"preflight: {world} ranks on devices {devs}: every rank needs its own GPU (LOCAL_RANK not set by the launcher?)")
    if rank == 0:
        log(f"preflight: backend {backend}, RCCL {ver}, ranks -> devices {[(t.get('rank'), t.get('device'), t.get('pci')) for t in table]}")
    # the two collectives of the step
    n = (64 << 20) // 2
    buf = torch.full((n,), float(rank + 1), dtype=torch.bfloat16, device=dev)
    flag = torch.tensor([float(rank)], device=dev)
    avg = dist.ReduceOp.AVG if backend == "nccl" else dist.ReduceOp.SUM    # gloo (the one-device rehearsal) has no AVG
    with Watchdog(timeout_s, "preflight: first 64 MiB bf16 all-reduce (communicator set-up)", lambda: _missing(store, "map", world)):
        dist.all_reduce(buf, op=avg)
        torch.cuda.synchronize(dev)
    _checkin(store, "allreduce1", rank)
    want = (world + 1) / 2.0 if backend == "nccl" else world * (world + 1) / 2.0
    got = float(buf[:8].float().mean().item())
    if abs(got - want) > 1e-2 * want:
        raise SystemExit(f"preflight: all-reduce gave {got}, expected {want} (rank {rank})")
    reps = 5
    with Watchdog(timeout_s, f"preflight: {reps} timed 64 MiB all-reduces + MAX", lambda: _missing(store, "allreduce1", world)):
        buf.fill_(1.0)
        torch.cuda.synchronize(dev)
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(reps):
            dist.all_reduce(buf, op=avg)
        torch.cuda.synchronize(dev)
        dt = (time.perf_counter() - t0) / reps
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        torch.cuda.synchronize(dev)
    _checkin(store, "done", rank)
    if float(flag.item()) != float(world - 1):
        raise SystemExit(f"preflight: MAX all-reduce gave {float(flag.item())}, expected {world - 1}")
    nbytes = n * 2
    info["allreduce_64mib"] = {"ms": round(dt * 1e3, 3), "algbw_gbs": round(nbytes / dt / 1e9, 1),
                               "busbw_gbs": round(nbytes / dt / 1e9 * 2 * (world - 1) / world, 1), "op": "AVG" if backend == "nccl" else "SUM"}
    if rank == 0:
        log(f"preflight: 64 MiB bf16 all-reduce {dt * 1e3:.3f} ms = {nbytes / dt / 1e9:.1f} GB/s algorithmic "
            f"({info['allreduce_64mib']['busbw_gbs']} GB/s bus), MAX ok")
    del buf
    return info


def wkv7_probe(torch, dev, reps=12):
    """Box fingerprint that DOES separate the box classes of the pool: the WKV7 chunked group itself, isolated --
    prep + fwd9 + bseq + bwd_out10 at configs[1]'s shape (8, 4096, 16) on fixed synthetic inputs, median of `reps` back-to-back rounds
    timed with HIP events, after the timed region.  `ms_per_step / wkv7_group_probe_ms` is the box-normalised step."""
    from rwkvtts_amd import ops
    from rwkvtts_amd.synthetic import make_wkv_inputs
    ins = make_wkv_inputs(8, 4096, 16, 1234, torch.bfloat16, dev)
    dy = torch.randn(8, 4096, 16, 64, device=dev, generator=torch.Generator(device=dev).manual_seed(5)).bfloat16()

    def once():
        y, tinv, sa, hs = ops.wkv7_chunk_forward(*ins)
        ops.wkv7_chunk_backward(*ins, dy, hs, sa, tinv)

    for _ in range(3):
        once()
    torch.cuda.synchronize(dev)
    ts = []
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        once()
        e.record()
        ts.append((s, e))
    torch.cuda.synchronize(dev)
    ms = sorted(s.elapsed_time(e) for s, e in ts)
    return round(ms[len(ms) // 2], 4)


PACKED_FRACTIONS = (0.125, 0.11, 0.1, 0.095, 0.09, 0.085, 0.08, 0.07, 0.06, 0.05, 0.04, 0.035, 0.03, 0.02, 0.01)


def packed_lengths(budget):
    """A fixed spread of sequence lengths (1 % ... 12.5 % of the row, none a multiple of 32) whose 32-ALIGNED layout fills exactly
    `budget` = B x L positions (every sequence is followed by >= 1 masked position up to the next chunk boundary: backbone
    ._forward_packed): the packed variable-length workload of --packed (data/utils/spark_dataset.py:111-162 packs real
    utterances into a token budget, max_cu_seqlens, the same way).  The row then has the shape the GEMMs of the [B, L] step have."""
    n = len(PACKED_FRACTIONS)
    tokens = budget - 32 * n
    lens = [max(65, int(tokens * f) // 2 * 2 + 1) for f in PACKED_FRACTIONS]
    lens[-1] += tokens - sum(lens)
    if lens[-1] % 32 == 0:
        lens[-1] -= 1
    # device-side cu_seqlens: the aligned row is sized from shapes alone, round-up-256(tokens + 32 n) = budget
    assert lens[-1] > 32 and (sum(lens) + 32 * n + 255) // 256 * 256 == budget, (lens, budget)
    return lens


def self_launch(n):
    """`python bench.py --gpus N` (the form the driver uses) -> N ranks under torch.distributed.run on this node."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on this pool (RCCL needs it)
    env.setdefault("NCCL_DEBUG", "WARN")                # RCCL's own warnings (transport fall-backs, failed IPC) reach stderr
    env.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    os.execvpe(sys.executable, cmd, env)


def cpu_baseline(seconds_budget=25.0):
    """BASELINE.json configs[0] on the host cores: the reference's PyTorch-CPU training step (oracle/rwkv7_ref.py: fp32
    eager, per-token torch scan) on the 0.1B Cosy model, B=2, L=512, and the WKV scan alone at (2,512,12,64) through
    the C oracle.  Whole steps are repeated until ~60 % of the budget is used (at least one: a step is 20-60 s of CPU work)."""
    import torch
    from oracle import c_oracle
    from oracle import rwkv7_ref as R
    from rwkvtts_amd.layouts import synthetic_cosy_batch
    from rwkvtts_amd.synthetic import make_wkv_inputs
    cores = R.pick_threads()   # fastest thread count for the per-token scan on this host (more is not faster), of os.cpu_count()
    cpu_model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu_model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    D, L, V = 768, 12, 6562
    cfg = R.RefConfig(hidden_size=D, num_hidden_layers=L, vocab_size=0)
    p = R.init_params(cfg, seed=0)
    g = torch.Generator().manual_seed(0)
    p.update({"llm_embedding.weight": torch.randn(2, D, generator=g) * 0.02,
              "text_embedding.weight": torch.randn(65548, D, generator=g) * 0.02,
              "speech_embedding.weight": torch.randn(V, D, generator=g) * 0.02,
              "lm_head.weight": torch.randn(V, D, generator=g) * 0.02, "lm_head.bias": torch.zeros(V)})
    for v in p.values():
        v.requires_grad_(True)
    batch = synthetic_cosy_batch(2, seed=1234)
    B, T = 2, 512
    t0 = time.time()
    n = 0
    while True:
        loss, _, _ = R.cosy_forward(p, cfg, batch, V - 1, 0.0, True)
        loss.backward()
        n += 1
        if time.time() - t0 > seconds_budget * 0.6:
            break
    dt = time.time() - t0
    # the scan alone (C oracle = scalar restatement of wkv7_cuda.cu, one core)
    ins = make_wkv_inputs(2, 512, 12, 1234, torch.float32)
    c_oracle.wkv7_fwd(*ins)
    t1 = time.time()
    reps = 3
    for _ in range(reps):
        c_oracle.wkv7_fwd(*ins)
    scan_ms = (time.time() - t1) / reps * 1e3
    return {"value": round(n * B * T / dt, 2), "unit": "tokens/s", "cores": cores, "kind": "port", "cpu": cpu_model,
            "sample": f"BASELINE configs[0]: RWKV7-0.1B Cosy layout train step fwd+bwd (LabelSmoothing KL), fp32 eager PyTorch "
                      f"(oracle/rwkv7_ref.py, per-token torch scan), B={B} L={T}, {n} step(s) in {dt:.1f} s on {cores} threads (fastest of 4/8/16/32 on this host, {os.cpu_count()} cores present)",
            "wkv_scan_fwd_ms": round(scan_ms, 2),
            "wkv_scan_sample": "C oracle (oracle/wkv7_oracle.c, 1 thread) forward scan at (B,T,H,N)=(2,512,12,64) fp32"}


def decode_rate(model, dev, B=32, P=128, n1=256, n2=2048):
    """BASELINE.json configs[4] on the model that was just trained: greedy decode, B = 32, prompt 128, 2048 new tokens through
    decode.GraphDecoder (rwkv7_decode_step_bf16 phases replayed from a hipGraph).  Two generate() calls of different length
    separate the per-step time from prefill + capture.  Outside the timed region; rank 0, one GPU only."""
    import torch
    from rwkvtts_amd.decode import GraphDecoder
    was_training = model.training
    model.eval()
    g = torch.Generator().manual_seed(1234)
    emb = (torch.randn(B, P, model.config.hidden_size, generator=g) * 0.5).to(dev, torch.bfloat16)
    mask = torch.ones(B, P, dtype=torch.long, device=dev)
    eos = model.config.vocab_size - 1   # suppressed, as in the synthetic workload

    def run(n, **kw):
        dec = GraphDecoder(model, B)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        dec.generate(inputs_embeds=emb, attention_mask=mask, max_new_tokens=n, suppress_tokens=[eos], **kw)
        torch.cuda.synchronize()
        return time.perf_counter() - t0, dec.step is not None

    run(8)
    t1, _ = run(n1)
    t2, kernel = run(n2)
    step = (t2 - t1) / (n2 - n1)
    # the same loop with the sampling parameters the reference's inference scripts use (inference/rwkv7speech_inference.py:99-107:
    # do_sample, top_k 50, top_p 0.95): the draw is one launch of csrc/sampling.hip inside the captured step
    sampled = None
    try:
        skw = dict(do_sample=True, top_k=50, top_p=0.95, temperature=1.0, seed=1)
        run(8, **skw)
        s1, _ = run(128, **skw)
        s2, _ = run(640, **skw)
        sstep = (s2 - s1) / (640 - 128)
        sampled = {"do_sample": True, "top_k": 50, "top_p": 0.95, "ms_per_step": round(sstep * 1e3, 4), "value": round(B / sstep, 1),
                   "unit": "tokens/s"}
    except Exception as e:
        sampled = {"error": repr(e)}
    # more requests than one 32-sequence group: k groups, each its own captured step, replayed round-robin on k streams
    # (decode.MultiGroupDecoder) -- the launch-latency-bound step of one group leaves most of the chip idle
    multi = None
    try:
        from rwkvtts_amd.decode import MultiGroupDecoder
        G = 4
        embG, maskG = emb.repeat(G, 1, 1), mask.repeat(G, 1)

        def run_multi(n):
            dec = MultiGroupDecoder(model, group_size=B)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            dec.generate(inputs_embeds=embG, attention_mask=maskG, max_new_tokens=n, suppress_tokens=[eos])
            torch.cuda.synchronize()
            return time.perf_counter() - t0

        run_multi(8)
        m1, m2 = run_multi(64), run_multi(448)
        mstep = (m2 - m1) / (448 - 64)
        multi = {"groups": G, "sequences": G * B, "ms_per_step_all_groups": round(mstep * 1e3, 4), "value": round(G * B / mstep, 1),
                 "unit": "tokens/s", "note": "decode.MultiGroupDecoder: independent 32-sequence groups on separate streams"}
    except Exception as e:
        multi = {"error": repr(e)}
    if was_training:
        model.train()
    # streaming bound of one step: every bf16 weight of the stack + head once, the fp32 state read and written
    cfg = model.config
    wbytes = 2 * (sum(p.numel() for p in model.model.layers.parameters()) + model.lm_head.weight.numel())
    sbytes = 2 * B * cfg.num_heads * 64 * 64 * 4 * cfg.num_hidden_layers
    floor = (wbytes + sbytes) / HBM_PEAK
    return {"metric": "greedy decode tokens/s, RWKV7-0.4B B=32 prompt=128 gen=2048 (BASELINE.json configs[4])",
            "value": round(B / step, 1), "unit": "tokens/s", "ms_per_step": round(step * 1e3, 4), "batch": B, "prompt": P,
            "new_tokens": n2, "total_s_incl_prefill_capture": round(t2, 3),
            "bytes_per_step": wbytes + sbytes, "hbm_floor_ms": round(floor * 1e3, 4), "frac_of_hbm_bound": round(floor / step, 4),
            "path": "rwkv7_decode_step_tbl_bf16, one kernel launch per phase, hipGraph replay" if kernel else "module by module, hipGraph replay",
            "sampled": sampled, "multi_group": multi}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=None, help="sequences per GPU (default 8; 4 for --layout xy)")
    ap.add_argument("--seq-len", type=int, default=None, help="positions per sequence (default 4096; 8192 for --layout xy)")
    ap.add_argument("--model", default="0.4b", choices=["0.1b", "0.4b", "1.5b"])
    ap.add_argument("--layout", default="spark", choices=["spark", "xy"],
                    help="xy: BASELINE.json configs[3] (8 channels, V0 = 66661, 8 fused linear+CE heads; use with --model 1.5b)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--grad-checkpoint", action="store_true")
    ap.add_argument("--no-decode", action="store_true", help="skip the greedy-decode measurement (BASELINE configs[4]) after the timed region")
    ap.add_argument("--one-device", action="store_true",
                    help="rehearsal of the N > 1 launch path on a one-GPU box: all ranks share GPU 0 and exchange over gloo "
                         "(RCCL refuses two ranks on one device); marked in the JSON line, not a measurement")
    ap.add_argument("--shard-optimizer", action="store_true",
                    help="N > 1: gradient pieces reduced to slab owners, AdamW on 1/N of the flat buffers per rank, parameter slabs "
                         "broadcast (trainer.DataParallelTrainer(shard_optimizer=True), SURVEY H6) instead of all-reduce + replicated AdamW")
    ap.add_argument("--wgrad-side-stream", action="store_true",
                    help="A/B: weight gradients on a second stream (fused.WGRAD_SIDE_STREAM): -2 ms per step, kernels that share the GPU measure longer")
    ap.add_argument("--via-reference-op", action="store_true",
                    help="A/B: the scan through torch.ops.wind_backstepping.forward/backward (the reference's plug-in point, wkv7_op.cpp:21-29; "
                         "for bf16 and T % 32 == 0 it launches the same chunked MFMA kernels, with `s` as their arena) instead of the direct calls")
    ap.add_argument("--no-packed", action="store_true", help="skip the secondary packed-row measurement after the timed region (N = 1)")
    ap.add_argument("--packed", action="store_true",
                    help="SURVEY 8f N1: the same B x L positions as ONE packed variable-length row [1, B L, D] + cu_seqlens on the device "
                         "(15 sequences of 300 ... 4096 positions; spark layout only): the packed training path of train_spark_rwkv7speech.py:238-239")
    ap.add_argument("--preflight-timeout", type=float, default=60.0, help="N > 1: seconds each preflight collective may take before the watchdog reports")
    ap.add_argument("--step-timeout", type=float, default=300.0, help="N > 1: seconds a training step may take before the watchdog reports")
    ap.add_argument("--scalar-wkv", action="store_true", help="A/B: scalar WKV7 kernels (reference schema fwd, row-split bwd) instead of the chunked MFMA pair")
    a = ap.parse_args()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(a.gpus)   # does not return

    import torch
    if a.scalar_wkv:
        from rwkvtts_amd import fused as _fused
        _fused.CHUNKED_WKV_BWD = False
        _fused.CHUNKED_WKV_FWD = False
    if a.via_reference_op:
        from rwkvtts_amd import fused as _fused
        _fused.VIA_REFERENCE_OP = True
    if a.wgrad_side_stream:
        from rwkvtts_amd import fused as _fused
        _fused.WGRAD_SIDE_STREAM = True

    from rwkvtts_amd import build
    if int(os.environ.get("LOCAL_RANK", "0")) == 0:
        build.build()  # no-op when the prebuilt .so is current; one rank per node, the others wait at the barrier below
    from rwkvtts_amd import backbone, ops, trainer
    from rwkvtts_amd.layouts import synthetic_spark_batch, synthetic_xy_batch

    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        # ranks started by torchrun directly (the driver's form) get the same defaults self_launch() gives its children
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("NCCL_DEBUG", "WARN")
        os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "1")
        # rendezvous + (with device_id) eager creation of the RCCL communicator: the first place a multi-GPU run can hang
        with Watchdog(max(a.preflight_timeout, 120.0), "init_process_group: TCP rendezvous at MASTER_ADDR:MASTER_PORT, then RCCL communicator creation "
                      "(ncclCommInitRank)", lambda: f"MASTER_ADDR={os.environ.get('MASTER_ADDR')} MASTER_PORT={os.environ.get('MASTER_PORT')} "
                      f"WORLD_SIZE={os.environ.get('WORLD_SIZE')} HSA_ENABLE_IPC_MODE_LEGACY={os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY')}"):
            rank, local_rank, world = trainer.init_distributed("gloo" if a.one_device else None)
    else:
        rank, local_rank, world = trainer.init_distributed("gloo" if a.one_device else None)
    if a.one_device:
        local_rank = 0
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {a.gpus} (or let bench.py launch itself)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    pre = None
    if world > 1:
        pre = preflight(torch, rank, local_rank, world, dev, a.one_device, a.preflight_timeout,
                        log=lambda m: print(f"[bench {time.strftime('%H:%M:%S')}] {m}", file=sys.stderr, flush=True))
        torch.distributed.barrier()

    base = {"0.1b": backbone.config_0p1b, "0.4b": backbone.config_0p4b, "1.5b": backbone.config_1p5b}[a.model]()
    base_kw = {k: v for k, v in base.to_dict().items() if k in backbone.RWKV7Config.__dataclass_fields__ and k != "extra"}
    if a.layout == "spark":
        from rwkvtts_amd.spark_llm import RWKV7ForSpeech, RWKV7SpeechConfig
        B, T = a.batch or 8, a.seq_len or 4096
        model = RWKV7ForSpeech(RWKV7SpeechConfig(**base_kw)).init_weights(seed=0)
    else:
        from rwkvtts_amd.xy_llm import RWKV7XYConfig, RWKV7XYLM
        B, T = a.batch or 4, a.seq_len or 8192
        base_kw["vocab_size"] = 66661
        model = RWKV7XYLM(RWKV7XYConfig(speech_vocab_size=1025, num_channels=8, text_shift_size=65536, **base_kw)).init_weights(seed=0)
        model.zero_embs()
    cfg = model.config
    model = model.to(device=dev, dtype=torch.bfloat16).train()
    if a.grad_checkpoint:
        model.gradient_checkpointing_enable()
    tr = trainer.DataParallelTrainer(model, lr=1e-4, warmup_steps=10, total_steps=1000, shard_optimizer=a.shard_optimizer)
    H = cfg.num_heads

    def packed_builder():
        """(lens, make_packed_batch): the packed variable-length workload of --packed for this model and B x T."""
        lens = packed_lengths(B * T)
        ntok = sum(lens)
        cu_dev = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32, device=dev)
        # the row's layout is fixed across steps (the ids are not): per table, where its lookups go in the row -- one lookup per table
        # and one index_copy each per step, as a collator that batches its embedding launches would do (the reference launches them per
        # sample, spark_dataset.py:111-162)
        pos = {"tag": [], "text": [], "glob": [], "sem": []}
        tag_ids, o = [], 0
        for n in lens:
            n_text = min(255, n // 4)
            n_sem = n - 3 - n_text - 32
            assert n_sem > 0, f"--packed: a sequence of {n} positions has no room for the Spark layout (3 tags + text + 32 global + semantic); use B x L >= 8192"
            pos["tag"] += [o, o + 1 + n_text, o + 2 + n_text + 32]
            tag_ids += [2, 0, 1]
            pos["text"] += range(o + 1, o + 1 + n_text)
            pos["glob"] += range(o + 2 + n_text, o + 2 + n_text + 32)
            pos["sem"] += range(o + 3 + n_text + 32, o + n)
            assert o + 3 + n_text + 32 + n_sem == o + n
            o += n
        pos = {k_: torch.tensor(v_, dtype=torch.long, device=dev) for k_, v_ in pos.items()}
        tag_ids = torch.tensor(tag_ids, dtype=torch.long, device=dev)

        def make_packed(i):
            # one packed row [1, sum T, D], embedded WITH autograd (the four tables get gradients), cu_seqlens a DEVICE tensor: no host
            # read-back in the step (backbone._forward_packed_device)
            g = torch.Generator().manual_seed(1234 + rank + 1000 * i)
            up = lambda t: t.pin_memory().to(dev, non_blocking=True)
            text = up(torch.randint(0, cfg.text_vocab_size, (pos["text"].numel(),), generator=g))
            glob = up(torch.randint(0, cfg.audio_global_vocab_size, (pos["glob"].numel(),), generator=g))
            sem = up(torch.randint(0, cfg.vocab_size - 1, (pos["sem"].numel(),), generator=g))
            row = torch.zeros(ntok, cfg.hidden_size, dtype=torch.bfloat16, device=dev)
            row = row.index_copy(0, pos["tag"], model.tts_tag_embedder(tag_ids))
            row = row.index_copy(0, pos["text"], model.text_embedder(text))
            row = row.index_copy(0, pos["glob"], model.global_embedder(glob))
            row = row.index_copy(0, pos["sem"], model.model.embeddings(sem))
            labels = torch.full((ntok,), -100, dtype=torch.long, device=dev)
            labels[pos["sem"]] = sem
            return dict(inputs_embeds=row.unsqueeze(0), labels=labels.unsqueeze(0), cu_seqlens=cu_dev)
        return lens, make_packed

    if a.layout == "spark" and a.packed:
        lens, make_batch = packed_builder()
    elif a.layout == "spark":
        def make_batch(i):
            # built inside the step WITH autograd, as data/utils/spark_dataset.py:163-239 does under the reference's training
            # loop (train_spark_rwkv7speech.py:630-634): the four embedding tables get gradients, their backward is timed
            return synthetic_spark_batch(model, B, T, seed=1234 + rank + 1000 * i)
    else:
        xy_cache = {}

        def make_batch(i):
            # ids are built on the host once per distinct seed (the reference's collator does it in the DataLoader workers)
            k = i % 4
            if k not in xy_cache:
                b = synthetic_xy_batch(B, T1=128, T2=T - 128 - 7, seed=1234 + rank + 1000 * k)
                b = {n: v.to(dev) for n, v in b.items()}
                backbone.mark_all_ones(b["attention_mask"], True)
                xy_cache[k] = b
            return dict(xy_cache[k], use_cache=False)

    def reducer_state():
        r = tr.reducer
        return (f"{len(r.buckets)} gradient buckets; on the wire (launch order) {list(r.launched)}; next bucket to launch {r.next_bucket}; "
                f"gradients still missing per bucket {list(getattr(r, 'pending', []))}; collectives not yet waited for {len(r.works)}")

    def one_step(i):
        if world == 1:
            return tr.step(**make_batch(i))
        with Watchdog(a.step_timeout, f"training step {i} (forward / backward / bucketed gradient exchange / AdamW)", reducer_state):
            out_ = tr.step(**make_batch(i))
            if i < a.warmup:
                torch.cuda.synchronize()    # warm-up steps complete inside their watchdog; timed steps stay asynchronous
            return out_

    def sync():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    def log(msg):
        if rank == 0:
            print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)

    log(f"model built: {sum(p.numel() for p in model.parameters()) / 1e6:.1f} M params, layout {a.layout}, B={B} T={T}, world {world}")
    for i in range(a.warmup):
        loss = one_step(i)
        torch.cuda.synchronize()
        log(f"warmup step {i} done, loss {float(loss):.4f}, mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
    sync()
    ops.KERNEL_TIMERS = {}
    tr.reducer.measure, tr.reducer.wait_events = world > 1, []
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
        tokens_per_rank = sum(lens) if (a.packed and a.layout == "spark") else B * T    # packed: the real tokens, not the aligned buffer
        value = world * tokens_per_rank * a.steps / dt
        kern = {k: sum(s.elapsed_time(e) for s, e in v) / len(v) for k, v in timers.items() if v}
        th = B * T * H
        chunk_parts = [k for k in ("wkv7c_bseq", "wkv7c_bwd_out") if k in kern]
        if "wkv7c_op_bwd" in kern:   # --via-reference-op: the op's launches are timed as one unit each way
            bwd_ms, bwd_name = kern["wkv7c_op_bwd"], f"torch.ops.wind_backstepping.backward -> wkv7c_bseq + wkv7c_bwd_out10 ({cfg.num_hidden_layers}x per step)"
            fwd_ms, fwd_name = kern["wkv7c_op_fwd"], "torch.ops.wind_backstepping.forward -> wkv7c_prep + wkv7c_fwd9"
            pmc_bwd, pmc_fwd = "wkv7c_bwd", "wkv7c_fwd"
        elif "wkv7c_bwd_out" in kern:
            bwd_ms = sum(kern[k] for k in chunk_parts)
            bwd_name = "WKV7 backward, chunked MFMA (" + " + ".join(chunk_parts) + f", {cfg.num_hidden_layers}x per step)"
            fwd_ms = kern["wkv7c_fwd"] + kern.get("wkv7c_prep", 0.0)
            fwd_name = "wkv7c_prep + wkv7c_fwd (chunked MFMA)"
            pmc_bwd, pmc_fwd = "wkv7c_bwd", "wkv7c_fwd"
        else:
            bwd_ms, bwd_name = kern.get("wkv7_bwd"), "wkv7_bwd_kernel<bf16,2,4> (row-split scalar WKV7 backward)"
            fwd_ms, fwd_name = kern.get("wkv7_fwd"), "wkv7_fwd_kernel (scalar)"
            pmc_bwd, pmc_fwd = "wkv7_bwd", "wkv7_fwd"
        pmc = {}
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_wkv7.json")))
        except Exception:
            pass

        stale = {}

        def pmc_of(key):
            """The committed PMC record of a kernel group -- only when it covers this shape AND was collected on the kernel sources
            that are shipped now (sha256 of the files recorded by tools/pmc_distill.py): otherwise the fields are null and
            `pmc_stale` says why."""
            d = pmc.get(key) or {}
            if not (d.get("B") == B and d.get("T") == T and d.get("H") == H):
                return {}
            rec = d.get("sources")
            if not rec or build.source_hashes(list(rec)) != rec:
                stale[key] = "no source hashes in the record" if not rec else \
                    "kernel sources changed since the PMC pass: " + ", ".join(n for n, h in build.source_hashes(list(rec)).items() if h != rec[n])
                return {}
            return d

        def traffic(d):
            return int((2 * d["FETCH_SIZE_KiB"] + d["WRITE_SIZE_KiB"]) * 1024) if "FETCH_SIZE_KiB" in d else None

        gbs = lambda nbytes, t_ms: nbytes / (t_ms * 1e-3) / 1e9 if t_ms else None
        b_bytes, f_bytes = th * WKV_BWD_BYTES_PER_TOKEN_HEAD, th * WKV_FWD_BYTES_PER_TOKEN_HEAD
        ach = gbs(b_bytes, bwd_ms)
        pb, pf = pmc_of(pmc_bwd), pmc_of(pmc_fwd)
        roof = {"kernel": bwd_name, "bound": "hbm", "achieved": round(ach, 1) if ach else None, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                "frac": round(ach * 1e9 / HBM_PEAK, 4) if ach else None, "traffic": traffic(pb),
                "mfma_util": pb.get("mfma_util"), "valu_frac": pb.get("valu_frac"), "pmc_source": pb.get("source"),
                "algorithmic_bytes_per_launch": b_bytes, "launch_ms": round(bwd_ms, 4) if bwd_ms else None}
        if fwd_ms:
            roof["fwd"] = {"kernel": fwd_name, "launch_ms": round(fwd_ms, 4), "achieved": round(gbs(f_bytes, fwd_ms), 1),
                           "frac": round(gbs(f_bytes, fwd_ms) * 1e9 / HBM_PEAK, 4), "traffic": traffic(pf),
                           "mfma_util": pf.get("mfma_util"), "valu_frac": pf.get("valu_frac"), "algorithmic_bytes_per_launch": f_bytes}
        from rwkvtts_amd import fused as _fused
        if getattr(_fused, "WGRAD_SIDE_STREAM", False):
            # the durations above are measured in the step, where the weight-gradient GEMMs of fused._wgrad run on a second stream
            # beside these kernels (same-box A/B: the step is 2 ms shorter, these kernels ~5 % longer than when they run alone)
            roof["concurrent"] = ("weight-gradient GEMMs on a side stream share the GPU with these kernels (without it: WKV7 group "
                                  "1.05 ms / frac 0.16, step +2 ms)")
        roof["pmc_stale"] = bool(stale)
        if stale:
            roof["pmc_stale_why"] = stale
        if fwd_ms and bwd_ms:
            both = gbs(f_bytes + b_bytes, fwd_ms + bwd_ms)
            roof["fwd_bwd"] = {"launch_ms": round(fwd_ms + bwd_ms, 4), "achieved": round(both, 1), "frac": round(both * 1e9 / HBM_PEAK, 4),
                               "target_frac": 0.40}
        which = {("spark", 1): 1, ("spark", 0): 2, ("xy", 1): 3, ("xy", 0): 3}[(a.layout, 1 if world == 1 else 0)]
        out = {
            "metric": "audio-tokens/sec/GPU (train fwd+bwd) RWKV7-0.4B L=4096; 1->8 GPU scaling",
            "value": round(value, 1), "unit": "tokens/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
            "per_gpu": round(value / world, 1),
            "config": {"workload": f"RWKV7-{a.model.upper()} {'Spark' if a.layout == 'spark' else 'XY_LM (8 channels)'} layout train step "
                                   f"(fwd+bwd+AdamW), B={B}/GPU L={T}, synthetic text+speech tokens, random init "
                                   f"(BASELINE.json configs[{which}])",
                       "global_batch": world * B, "seq_len": T, "parallelism": f"dp{world}",
                       "grad_allreduce": (("bucketed gloo SUM in fp32 (rehearsal)" if a.one_device else
                                           "bucketed RCCL AVG, bf16, overlapped with backward") if not tr.shard_optimizer else
                                          "bucket pieces reduced to slab owners + sharded AdamW + parameter broadcast") if world > 1 else "none"},
            "loss": round(loss_val, 4),
            "peak_mem_gib": round(torch.cuda.max_memory_allocated() / 2**30, 1),
            "kernel_ms": {k: round(v, 4) for k, v in kern.items()},
            "device": _device_info(torch, dev),
            "roofline": roof,
        }
        try:   # the box-class indicator: the isolated WKV7 group, and the step expressed in units of it
            out["device"]["wkv7_group_probe_ms"] = wkv7_probe(torch, dev)
            out["device"]["step_over_wkv7_probe"] = round(ms / out["device"]["wkv7_group_probe_ms"], 2)
        except Exception as e:
            out["device"]["wkv7_probe_error"] = repr(e)
        if a.packed and a.layout == "spark":
            out["config"]["workload"] += (f"; PACKED: one row [1, {sum(lens)}, D] of {len(lens)} sequences ({min(lens)}..{max(lens)} positions; "
                                          f"32-aligned layout = {B * T} positions), cu_seqlens on the device (SURVEY 8f N1); value counts the {sum(lens)} real tokens")
            out["config"]["packed_lengths"] = lens
        if a.wgrad_side_stream:
            out["config"]["wgrad"] = "weight gradients on a side stream (fused.WGRAD_SIDE_STREAM)"
        if a.via_reference_op:
            out["config"]["wkv_entry"] = "torch.ops.wind_backstepping.forward/backward (reference schema)"
        if a.one_device:
            out["rehearsal"] = f"{world} ranks sharing GPU 0, exchange over gloo: exercises the launch path, not a measurement"
        if world > 1:
            r = tr.reducer
            ev_ms = sum(s_.elapsed_time(e_) for s_, e_ in r.wait_events) / a.steps
            out["comm"] = {"preflight": pre, "rccl_ranks": world, "backend": r.backend, "bytes_allreduced_per_step": tr.flat.numel * tr.flat.flat_grad.element_size(),
                           "buckets": len(r.buckets), "bucket_order": "rank 0's gradient-ready order, broadcast (re-cut after the first backward pass); launched in index order" if r.rebuilt else "reverse registration order",
                           "optimizer": "AdamW bucket by bucket as each all-reduce completes" if tr.bucket_optimizer and not tr.shard_optimizer else "one pass behind the last bucket",
                           "bucket_mib": [round(sum(e - s for s, e in runs) * tr.flat.flat_grad.element_size() / 2**20, 1) for runs in r.runs],
                           "bucket_runs": [len(runs) for runs in r.runs],
                           "exposed_wait_ms_per_step": round(sum(s_.elapsed_time(e_) for s_, e_ in r.wait_events) / a.steps, 3),
                           "note": "exposed = time the compute stream stalls in the buckets' wait() calls (HIP event pairs around each bucket's waits, rank 0; the optimizer launches between them are not counted)"}
        if world == 1 and not a.no_packed and not a.packed and a.layout == "spark" and B * T >= 8192:
            # SURVEY 8f N1, after the timed region and not part of `value`: the same model and trainer on the PACKED variable-length
            # workload of --packed (one row of 15 sequences, cu_seqlens on the device), 1 warm-up + 3 timed steps
            try:
                plens, make_packed = packed_builder()
                tr.step(**make_packed(10_000))
                torch.cuda.synchronize()
                tp = time.perf_counter()
                for i_ in range(3):
                    tr.step(**make_packed(10_001 + i_))
                torch.cuda.synchronize()
                pdt = (time.perf_counter() - tp) / 3
                out["packed"] = {"metric": "packed variable-length training step (cu_seqlens), same model, tokens/s counted on real tokens",
                                 "ms_per_step": round(pdt * 1e3, 3), "value": round(sum(plens) / pdt, 1), "unit": "tokens/s", "sequences": len(plens),
                                 "tokens": sum(plens), "aligned_positions": B * T, "steps": 3,
                                 "vs_plain_step": round(pdt * 1e3 / ms, 4)}
            except Exception as e:
                out["packed"] = {"error": repr(e)}
        if world == 1 and not a.no_decode and a.layout == "spark":
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
