"""Batch-sharded data-parallel training step: one process per GPU, RCCL over xGMI.

What it replaces in the reference: the DeepSpeed engine of train_scripts/train_spark_rwkv7speech.py:483-516,
566-572 (ZeRO-2, bf16 grads, reduce bucket 5e6 elements, overlap_comm False) and its step semantics
(:621-691): forward -> NaN flag all_reduce(MAX) (:664-670) -> backward -> gradient reduction -> AdamW
(betas .9/.95, eps 1e-18, weight_decay 0, :178-197) with the linear warmup/decay schedule (:219-232).

MI355X-first choices:
  * the model is replicated (0.4B: 0.8 GB bf16 + 4.8 GB fp32 master/Adam; 1.5B: ~27 GB) -- 288 GB of HBM
    per GPU makes ZeRO sharding/offload pointless at these sizes;
  * parameters and gradients live in two flat bf16 buffers; gradients are all-reduced (AVG) in buckets of
    ~32 MiB fired from post-accumulate-grad hooks in backward order, asynchronously on RCCL's stream, so
    the exchange overlaps the rest of backward; xGMI is point-to-point (7 links/GPU), so few large
    buckets beat DeepSpeed's 10 MB ones;
  * fp32 master weights + fused AdamW on one flat tensor, then one bf16 copy back.
"""
from __future__ import annotations

import math
import os
from typing import List, Optional

import torch
import torch.distributed as dist


FLUSH_EVERY = int(os.environ.get("RWKV7_FLUSH_EVERY", "48"))   # adopted gradients per in-backward flush (0: one flush at the end)


def init_distributed(backend: Optional[str] = None):
    """Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torch.distributed.run contract).  Returns
    (rank, local_rank, world).  No-op for world size 1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"  # "nccl" IS RCCL on ROCm
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, local_rank, world


class FlatBuffers:
    """Re-homes every parameter (and its .grad) of `model` into two contiguous buffers.

    Gradients reach the flat buffer without an accumulate pass: before backward every .grad is None, so autograd ADOPTS the
    tensor a backward node returns instead of adding it to a zeroed buffer; a post-accumulate hook then copies it into the
    parameter's slice -- unless the node already wrote there (`param._grad_slot`, used by fused._Linear's split weight
    gradient: 14 matrices per layer arrive with no extra launch at all) -- and re-attaches the slice as .grad.  Slices of
    parameters that received no gradient in a pass are zeroed by `finish_backward()`."""

    def __init__(self, model: torch.nn.Module):
        params = [p for p in model.parameters() if p.requires_grad]
        assert params, "no trainable parameters"
        dt, dev = params[0].dtype, params[0].device
        assert all(p.dtype == dt for p in params), "mixed parameter dtypes"
        self.params = params
        self.offsets, n = [], 0
        for p in params:
            self.offsets.append(n)
            n += (p.numel() + 127) // 128 * 128  # 256-B aligned slices
        self.numel = n
        self.flat_param = torch.zeros(n, dtype=dt, device=dev)
        self.flat_grad = torch.zeros(n, dtype=dt, device=dev)
        self.views = []
        self.fired = [False] * len(params)
        self.to_copy = []
        self.on_ready = None   # callable(i): parameter i's slice is final (BucketedAllReduce)
        for i, (p, o) in enumerate(zip(params, self.offsets)):
            self.flat_param[o:o + p.numel()].copy_(p.data.reshape(-1))
            p.data = self.flat_param[o:o + p.numel()].view_as(p)
            v = self.flat_grad[o:o + p.numel()].view_as(p)
            self.views.append(v)
            p.grad = v
            p._grad_slot = v
            p._grad_slot_used = True   # armed by zero_grad()
            p.register_post_accumulate_grad_hook(self._make_hook(i))

    def _make_hook(self, i):
        def hook(param):
            g = param.grad
            if g is not None and g.data_ptr() != self.views[i].data_ptr():
                self.to_copy.append(i)   # moved into its slice by flush(): one multi-tensor copy for many parameters
                # ... in batches WHILE backward runs: left to the end, the ~500 small gradients of a 24-layer model are one burst of
                # host work (list building + multi-tensor launches, 1.4 ms) behind the last backward kernel, with the GPU idle
                # (profiles/r05k_step_busy.txt); a batch flushed from a hook is host work under the layers still queued
                if len(self.to_copy) >= FLUSH_EVERY > 0:
                    self.flush()
            self.fired[i] = True
            if self.on_ready is not None:
                self.on_ready(i)
        return hook

    def flush(self):
        """Copy the adopted gradients collected so far into their slices (one multi-tensor launch) and attach the slices."""
        if self.to_copy:
            # INVARIANT: every gradient in to_copy was produced on the CURRENT stream.  Weight gradients computed on the side stream
            # (fused.WGRAD_SIDE_STREAM) are written straight into their slice (param._grad_slot) and never come through here; a future
            # fused node that hands a side-stream tensor back as .grad must call fused.wgrad_side_sync() before this copy.
            src = [self.params[i].grad for i in self.to_copy]
            dst = [self.views[i] for i in self.to_copy]
            if src[0].is_cuda and all(s.dtype == d.dtype for s, d in zip(src, dst)):
                torch._foreach_copy_(dst, src)
            else:
                for d, s_ in zip(dst, src):
                    d.copy_(s_)
            for i in self.to_copy:
                self.params[i].grad = self.views[i]
            self.to_copy = []

    def zero_grad(self):
        """Plain mode: zero the buffer and attach the slices as .grad; backward then accumulates into them."""
        self.flat_grad.zero_()
        for i, p in enumerate(self.params):
            p.grad = self.views[i]
            p._grad_slot_used = True
            self.fired[i] = False

    def arm(self):
        """Fast mode (DataParallelTrainer.step): nothing is zeroed, the next backward pass overwrites the slices;
        finish_backward() must follow it."""
        for i, p in enumerate(self.params):
            p.grad = None
            p._grad_slot_used = False
            self.fired[i] = False

    def finish_backward(self, reattach=True):
        """After backward: zero the slices that received nothing, re-attach every .grad.
        reattach=False (DataParallelTrainer.step): only what the GPU needs -- the flush and the zeroing -- so that the optimizer can be
        launched at once; the ~800 attribute writes of `reattach()` (about 1 ms of host time during which the GPU sat idle between
        the last backward kernel and AdamW, profiles/r05i_step_busy.txt) run behind that launch."""
        from . import fused
        fused.wgrad_side_sync()   # weight gradients queued on the side stream land before anything reads the buffer
        self.flush()
        for i in range(len(self.params)):
            if not self.fired[i]:
                self.views[i].zero_()
        if reattach:
            self.reattach()

    def reattach(self):
        for i, p in enumerate(self.params):
            p.grad = self.views[i]
            p._grad_slot_used = True


class BucketedAllReduce:
    """Gradient all-reduce in buckets, launched from backward hooks as soon as a bucket is complete."""

    def __init__(self, flat: FlatBuffers, bucket_bytes: int = 32 << 20, group=None, force: bool = False, shard: bool = False):
        """force: run the collectives even in a group of one rank (tests: the whole hook -> flush -> RCCL -> wait path on a
        single GPU, where the exchange is the identity).
        shard: every rank owns one contiguous 1/N slab of the flat buffers (`slab(r)`); a bucket's pieces are REDUCED to their
        owners instead of all-reduced (half the bytes on the wire per step; the other half is the parameter broadcast after the
        owners' optimizer step, DataParallelTrainer) -- SURVEY H6's fallback for an exposed all-reduce tail."""
        self.flat, self.group = flat, group
        self.shard = bool(shard)
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.backend = dist.get_backend(group) if dist.is_initialized() else None
        self.bucket_bytes = bucket_bytes
        # First pass: buckets follow REGISTRATION order backwards (last parameters first), the usual proxy for gradient-ready order.
        # It is wrong where registration and use differ -- the Spark model registers its three input-side embedding tables AFTER
        # lm_head, so they sat in bucket 0 (136 MiB) although their gradients arrive at the very end of backward, and the bucket
        # that should open the exchange was the last to fire.  The order the hooks actually fire in is recorded during the first
        # backward pass and the buckets are re-cut along it once (`rebuild_from_ready_order`, what DDP does after its first
        # iteration).  The flat buffers are NOT re-laid-out: a bucket is a list of contiguous runs of the flat gradient buffer
        # (normally one; the embedding-side bucket has one run per end of the buffer), each run one collective.
        self._cut(list(range(len(flat.params) - 1, -1, -1)))
        self.ready_order: List[int] = []
        self.rebuilt = False
        self.order_differs_from_rank0 = False
        self.works = []          # (bucket, work) in launch order
        self.launched = []       # buckets in launch order (the order finish() hands them to `on_bucket`)
        self.use_avg, self.summed = True, []
        self.measure, self.wait_events = False, []
        self.defer_reattach = False   # DataParallelTrainer sets it: it calls flat.reattach() itself, behind the optimizer launch
        self.enabled = self.world > 1 or (force and dist.is_initialized())
        if self.enabled:
            flat.on_ready = self._ready

    def _cut(self, order):
        """Buckets of >= bucket_bytes along `order` (parameter indices, first-ready first).  self.buckets[b] = [start, end, n_params]
        with [start, end) the span of the bucket's runs (what bench.py prints); self.runs[b] = the contiguous [s, e) pieces."""
        flat, esz = self.flat, self.flat.flat_grad.element_size()
        size = lambda i: ((flat.params[i].numel() + 127) // 128 * 128)
        self.buckets, self.runs = [], []
        self.param_bucket = [0] * len(flat.params)
        cur, nbytes = [], 0
        for n, i in enumerate(order):
            cur.append(i)
            nbytes += size(i) * esz
            self.param_bucket[i] = len(self.buckets)
            if nbytes >= self.bucket_bytes or n == len(order) - 1:
                pieces = sorted((flat.offsets[j], flat.offsets[j] + size(j)) for j in cur)
                runs = [list(pieces[0])]
                for s_, e_ in pieces[1:]:
                    if s_ == runs[-1][1]:
                        runs[-1][1] = e_
                    else:
                        runs.append([s_, e_])
                self.buckets.append([runs[0][0], runs[-1][1], len(cur)])
                self.runs.append([tuple(r) for r in runs])
                cur, nbytes = [], 0
        self.pending = [b[2] for b in self.buckets]
        self.next_bucket = 0

    def rebuild_from_ready_order(self):
        """Re-cut the buckets along the order in which the gradient hooks fired in the pass just finished (once; parameters whose
        hook did not fire go last).  STEP 0 IS NOT REPRESENTATIVE: until this re-cut the buckets follow reverse registration order and
        go on the wire in index order, and bucket 0 then holds the input-side embedding tables whose gradients arrive LAST -- no
        all-reduce starts before backward ends (no overlap), and the re-cut itself is a host-synchronous broadcast + tolist().  Warm-up
        steps absorb it: bench.py times from step W on (W >= 1) and reports `comm.bucket_order` so a line measured on the initial cut
        is recognisable.  The order is RANK 0's, broadcast to everybody (what DDP does): the order a rank records
        depends on the shapes of ITS batch -- fused.mix_lora_supported needs B*T >= 4096, linear_add_eligible / cmix_eligible
        need M % 256 == 0, and the fused nodes hand over the LoRA / x_* gradients in a different order than the plain ones --
        so two ranks fed differently shaped batches on step 0 would otherwise cut their buckets at different byte boundaries
        and the collectives would mismatch (hang or silently wrong sums).  One host-synchronous broadcast of n_params int32,
        once per run.  tests/test_trainer_dist.py drives the two ranks through different orders and asserts the common cut."""
        n = len(self.flat.params)
        order = list(self.ready_order)
        if self.world > 1:
            dev = self.flat.flat_grad.device if self.backend == "nccl" else torch.device("cpu")
            t = torch.full((n,), -1, dtype=torch.int32, device=dev)
            if self.rank == 0:
                t[:len(order)] = torch.tensor(order, dtype=torch.int32)
            src = dist.get_global_rank(self.group, 0) if self.group is not None else 0
            dist.broadcast(t, src=src, group=self.group)
            theirs = [int(i) for i in t.tolist() if i >= 0]
            self.order_differs_from_rank0 = theirs != order
            order = theirs
        self.ready_order = order
        seen = set(order)
        order = order + [i for i in range(n - 1, -1, -1) if i not in seen]
        self._cut(order)
        self.rebuilt = True

    def slab(self, r):
        """[start, end) of rank r's slab of the flat buffers (128-element aligned, the last one may be shorter or empty)."""
        n = self.flat.numel
        size = -(-n // (self.world * 128)) * 128
        return min(r * size, n), min((r + 1) * size, n)

    def _ready(self, i):
        if not self.rebuilt:
            self.ready_order.append(i)
        b = self.param_bucket[i]
        self.pending[b] -= 1
        # buckets go on the wire strictly in INDEX order (as DDP's reducer does): a bucket that completes early waits for its
        # predecessors.  The sequence of collectives is then the same on every rank at every step even when the ranks' hooks
        # fire in different orders (differently shaped batches take different fused paths) -- a mismatched sequence is a hang.
        while self.next_bucket < len(self.pending) and self.pending[self.next_bucket] == 0:
            self._launch(self.next_bucket)
            self.next_bucket += 1

    def _launch(self, b):
        from . import fused
        fused.wgrad_side_sync()   # ... including the weight gradients still running on fused's side stream
        self.flat.flush()   # the bucket's slices must hold the final gradients
        self.launched.append(b)
        for s, e in self.runs[b]:
            self._exchange(s, e, b)

    def _exchange(self, s, e, b=-1):
        if self.shard:
            for r in range(self.world):   # the run's intersection with every rank's slab goes to that rank only
                lo, hi = self.slab(r)
                lo, hi = max(lo, s), min(hi, e)
                if lo >= hi:
                    continue
                piece = self.flat.flat_grad[lo:hi]
                if self.backend == "nccl":
                    self.works.append((b, dist.reduce(piece, dst=dist.get_global_rank(self.group, r) if self.group is not None else r,
                                                      op=dist.ReduceOp.SUM, group=self.group, async_op=True)))
                    if r == self.rank:
                        self.summed.append((b, lo, hi))
                else:   # gloo (CPU tests): fp32, synchronous
                    tmp = piece.float()
                    dist.reduce(tmp, dst=r, op=dist.ReduceOp.SUM, group=self.group)
                    if r == self.rank:
                        piece.copy_((tmp / self.world).to(piece.dtype))
            return
        view = self.flat.flat_grad[s:e]
        if self.backend == "nccl":
            if self.use_avg:
                try:
                    self.works.append((b, dist.all_reduce(view, op=dist.ReduceOp.AVG, group=self.group, async_op=True)))
                    return
                except RuntimeError:   # a collective library without AVG for this dtype: SUM now, one scale in finish()
                    self.use_avg = False
            self.works.append((b, dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True)))
            self.summed.append((b, s, e))
        else:  # gloo (CPU tests): no AVG, and bf16 support varies -> reduce in fp32
            tmp = view.float()
            dist.all_reduce(tmp, op=dist.ReduceOp.SUM, group=self.group)
            view.copy_((tmp / self.world).to(view.dtype))

    def finish(self, on_bucket=None):
        """Wait for every bucket (also launches buckets whose hooks never fired, e.g. unused params).  Backward always runs: a
        NaN step backpropagates like any other and the optimizer kernel substitutes a zero gradient (DataParallelTrainer.step).

        on_bucket(runs): called once per bucket, in launch order, as soon as THAT bucket's collectives have completed on the
        compute stream's timeline (work.wait() = the compute stream waits for RCCL's stream up to that collective) -- the
        optimizer steps bucket b while buckets b+1.. are still on the wire, instead of one replicated pass behind the last
        all-reduce.  Returns True if it was called for every bucket (False: exchange disabled, caller does one whole pass)."""
        self.flat.finish_backward(reattach=self.defer_reattach is False)
        if not self.enabled:
            return False
        for b in range(self.next_bucket, len(self.pending)):   # incomplete buckets (unused parameters) and their successors
            self._launch(b)
        # wait() makes the compute stream wait for RCCL's stream; what the compute stream then stalls is the exposed (non-overlapped)
        # part of the exchange -- bracketed by event pairs when `measure` is on (bench.py): one pair around ALL waits in the one-pass
        # mode, one pair around each bucket's waits in the per-bucket mode (the optimizer launches between them are work, not stall)
        def bracket():
            if self.measure and self.works:
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                return e
            return None
        per_bucket = on_bucket is not None and not self.shard
        if per_bucket:
            by_b, scale_b = {}, {}
            for b, w in self.works:
                by_b.setdefault(b, []).append(w)
            for b, s, e in self.summed:
                scale_b.setdefault(b, []).append((s, e))
            for b in self.launched:
                e0 = bracket()
                for w in by_b.get(b, ()):
                    w.wait()
                e1 = bracket()
                if e0 is not None:
                    self.wait_events.append((e0, e1))
                for s, e in scale_b.get(b, ()):
                    self.flat.flat_grad[s:e].mul_(1.0 / self.world)
                on_bucket(self.runs[b])
        else:
            e0 = bracket()
            for _, w in self.works:
                w.wait()
            e1 = bracket()
            if e0 is not None:
                self.wait_events.append((e0, e1))
            for _, s, e in self.summed:
                self.flat.flat_grad[s:e].mul_(1.0 / self.world)
        self.works, self.summed, self.launched = [], [], []
        if not self.rebuilt and (self.ready_order or self.world > 1):
            self.rebuild_from_ready_order()   # once, after the first backward pass (also resets `pending`)
        else:
            self.pending = [b[2] for b in self.buckets]
            self.next_bucket = 0
        return per_bucket


def linear_warmup_decay(step, total_steps, warmup_steps, lr, lr_final):
    """train_spark_rwkv7speech.py:219-232."""
    if step < warmup_steps:
        return lr * float(step) / float(max(1, warmup_steps))
    progress = float(step - warmup_steps) / float(max(1, total_steps - warmup_steps))
    return lr * max(lr_final / lr, 1.0 - progress * (1.0 - lr_final / lr))


def cosine_warmup_decay(step, total_steps, warmup_steps, lr, lr_final):
    """train_cosy_rwkv7speech_multiple_dataset.py:224-234: warmup from 1 % of lr, then half a cosine down to lr_final."""
    if step < warmup_steps:
        return lr * (0.01 + 0.99 * step / warmup_steps)
    progress = float(step - warmup_steps) / float(max(1, total_steps - warmup_steps))
    progress = max(0.0, min(1.0, progress))
    f = lr_final / lr
    return lr * ((0.5 + f / 2) + (0.5 - f / 2) * math.cos(math.pi * progress))


SCHEDULES = {"linear": linear_warmup_decay, "cosine": cosine_warmup_decay}


def reference_param_groups(model: torch.nn.Module, weight_decay: float):
    """The parameter groups of configure_optimizer (train_cosy_rwkv7speech_multiple_dataset.py:162-190), one entry per
    trainable parameter in model.parameters() order: (group name, lr scale, weight decay).
      lr_2x    : names containing 'attn.w_lora.lora.2.bias' (the decay bias w0)            -> lr x 2, no decay
      lr_decay : >= 2-D (after squeeze) '.weight' tensors outside the LoRAs, if weight_decay > 0 -> lr x 1, decay
      lr_1x    : everything else                                                            -> lr x 1, no decay"""
    out = []
    for n, p in model.named_parameters():
        if not p.requires_grad:
            continue
        if "attn.w_lora.lora.2.bias" in n:
            out.append(("lr_2x", 2.0, 0.0))
        elif len(p.squeeze().shape) >= 2 and weight_decay > 0 and ".weight" in n and "lora" not in n:
            out.append(("lr_decay", 1.0, float(weight_decay)))
        else:
            out.append(("lr_1x", 1.0, 0.0))
    return out


class DataParallelTrainer:
    """param_groups: None = the Spark trainer's single group (train_spark_rwkv7speech.py:178-197: every parameter lr x 1 and an
    explicit "weight_decay": 0.0 at :188, which overrides the optimizer default -- the reference decays NOTHING there, whatever
    args.weight_decay says; so does this mode: `weight_decay` is ignored); "all" = one group with `weight_decay` on every tensor
    (explicit opt-in; not what either reference trainer does); "reference" = reference_param_groups(model, weight_decay) (the
    Cosy trainer's lr_2x / lr_decay split); or a list of (name, lr scale, weight decay), one per trainable parameter.
    schedule: "linear" (Spark trainer) or "cosine" (Cosy)."""

    def __init__(self, model: torch.nn.Module, lr=1e-4, lr_final=1e-5, warmup_steps=100, total_steps=100000,
                 weight_decay=0.0, betas=(0.9, 0.95), eps=1e-18, bucket_bytes=32 << 20, nan_guard=True,
                 master_fp32=True, param_groups=None, schedule="linear", force_allreduce=False, shard_optimizer=False,
                 bucket_optimizer=True):
        """shard_optimizer: every rank reduces the gradient pieces of ITS 1/N slab only, steps AdamW on that slab and broadcasts
        the slab's new bf16 parameters (reduce-scatter + sharded optimizer + all-gather, ZeRO-1 style: the reference's DeepSpeed
        ZeRO-2 engine does the same exchange, train_spark_rwkv7speech.py:483-516).  Same parameters after every step as the
        default (all-reduce + replicated AdamW) up to the rounding of the reduction; the optimizer pass is 1/N as long.  One flag:
        for the case that the 8-GPU scaling run shows an exposed all-reduce tail.
        NOTE for code that reads gradients after step(): in shard mode only the slices of a rank's OWN slab hold the reduced (mean)
        gradient; `p.grad` of parameters in foreign slabs holds this rank's local, unreduced gradient (gradient-norm logging or
        clipping must all-reduce its own partial over the own slab, `reducer.slab(rank)`).
        bucket_optimizer (default, all-reduce mode with > 1 rank): AdamW runs bucket by bucket, each as soon as its own
        all-reduce has completed, so the optimizer pass (2.1-2.4 ms for the 0.4B model) overlaps the buckets still on the wire
        instead of sitting behind the last one.  Element-wise the same update: identical parameters to the one-pass mode."""
        self.model = model
        self.bucket_optimizer = bool(bucket_optimizer)
        self.flat = FlatBuffers(model)
        self.reducer = BucketedAllReduce(self.flat, bucket_bytes, force=force_allreduce, shard=shard_optimizer)
        self.shard_optimizer = bool(shard_optimizer) and self.reducer.enabled
        self.reducer.defer_reattach = True
        self.world = self.reducer.world
        self.master = self.flat.flat_param.float() if master_fp32 and self.flat.flat_param.dtype != torch.float32 \
            else self.flat.flat_param
        self.betas, self.eps, self.weight_decay = betas, eps, weight_decay
        if param_groups == "reference":
            param_groups = reference_param_groups(model, weight_decay)
        if param_groups is None:
            if weight_decay:
                import warnings
                warnings.warn("DataParallelTrainer(param_groups=None) decays nothing, like the Spark trainer's single group "
                              "(train_spark_rwkv7speech.py:188); weight_decay=%g is ignored -- pass param_groups='all' (decay every "
                              "parameter) or 'reference' (the Cosy trainer's lr_decay group)" % weight_decay, stacklevel=2)
            param_groups = [("all", 1.0, 0.0)] * len(self.flat.params)
        elif param_groups == "all":
            param_groups = [("all", 1.0, float(weight_decay))] * len(self.flat.params)
        assert len(param_groups) == len(self.flat.params), "one (name, lr scale, weight decay) per trainable parameter"
        self.group_defs = []          # distinct (name, lr scale, weight decay)
        self.param_group_idx = []
        for g in param_groups:
            g = (g[0], float(g[1]), float(g[2]))
            if g not in self.group_defs:
                self.group_defs.append(g)
            self.param_group_idx.append(self.group_defs.index(g))
        assert len(self.group_defs) <= 256
        dev = self.master.device
        self.schedule = SCHEDULES[schedule] if isinstance(schedule, str) else schedule
        # bf16 parameters on the HIP device: one kernel reads the bf16 gradients, updates the fp32 master weights and
        # moments and rewrites the bf16 parameters (rwkv7_adamw_groups_bf16) -- no fp32 gradient copy, no separate cast
        # back; the group of every 128-element slab of the flat buffer comes from a uint8 table (FlatBuffers aligns the
        # parameters to 128 elements), the NaN flag stays on the device
        self.hip_adamw = (self.master is not self.flat.flat_param and self.master.is_cuda
                          and self.flat.flat_param.dtype == torch.bfloat16 and self.flat.numel % 128 == 0)
        self.exp_avg = torch.zeros_like(self.master)
        self.exp_avg_sq = torch.zeros_like(self.master)
        if self.hip_adamw:
            slab = torch.zeros(self.flat.numel // 128, dtype=torch.uint8)
            ends = self.flat.offsets[1:] + [self.flat.numel]
            for o, e, gi in zip(self.flat.offsets, ends, self.param_group_idx):
                slab[o // 128:e // 128] = gi
            self.slab_group = slab.to(dev)
            self.group_tab = torch.tensor([[g[1], g[2]] for g in self.group_defs], dtype=torch.float32).to(dev)
        else:
            # CPU (gloo tests) / fp32 models: the same update rule in torch, group by group on runs of the flat buffers
            self.runs = []   # (start, end, group index): consecutive parameters of one group merged
            ends = self.flat.offsets[1:] + [self.flat.numel]
            for o, e, gi in zip(self.flat.offsets, ends, self.param_group_idx):
                if self.runs and self.runs[-1][2] == gi and self.runs[-1][1] == o:
                    self.runs[-1] = (self.runs[-1][0], e, gi)
                else:
                    self.runs.append((o, e, gi))
        self.lr, self.lr_final, self.warmup_steps, self.total_steps = lr, lr_final, warmup_steps, total_steps
        self.nan_guard = nan_guard
        self.nan_flag = torch.zeros(1, dtype=torch.float32, device=dev)
        self.step_idx = 0
        self.last_lr = None
        # modules that cache tensors derived from parameters (RWKV7Attention._stacked_mix): the optimizer kernel rewrites
        # parameter memory through raw pointers without touching autograd's version counters, so they are told explicitly
        self._param_caches = [m for m in model.modules() if hasattr(m, "_mix_key") or hasattr(m, "_stacked_mix")]

    def current_lr(self):
        return self.schedule(self.step_idx, self.total_steps, self.warmup_steps, self.lr, self.lr_final)

    def group_lrs(self):
        """{group name: lr of the next step} -- what update_learning_rate writes into optimizer.param_groups (:236-241)."""
        base = self.current_lr()
        return {g[0]: base * g[1] for g in self.group_defs}

    def _torch_adamw(self, lr, skip, lo_s=None, hi_s=None):
        g_all = self.flat.flat_grad
        b1, b2 = self.betas
        t = self.step_idx + 1
        bc1, bc2 = 1.0 - b1 ** t, 1.0 - b2 ** t
        if lo_s is None:
            lo_s, hi_s = self.reducer.slab(self.reducer.rank) if self.shard_optimizer else (0, self.flat.numel)
        for s, e, gi in self.runs:
            s, e = max(s, lo_s), min(e, hi_s)   # sharded: this rank's slab only
            if s >= e:
                continue
            _, scale, wd = self.group_defs[gi]
            p, m, v = self.master[s:e], self.exp_avg[s:e], self.exp_avg_sq[s:e]
            g = g_all[s:e].to(p.dtype)
            g = torch.where(skip.to(torch.bool), torch.zeros_like(g), g)
            lr_g = lr * scale
            p.mul_(1.0 - lr_g * wd)
            m.mul_(b1).add_(g, alpha=1.0 - b1)
            v.mul_(b2).addcmul_(g, g, value=1.0 - b2)
            p.addcdiv_(m, (v.sqrt() / math.sqrt(bc2)).add_(self.eps), value=-lr_g / bc1)
        if self.master is not self.flat.flat_param:
            self.flat.flat_param[lo_s:hi_s].copy_(self.master[lo_s:hi_s])

    def _broadcast_slabs(self):
        """Sharded mode: every rank's freshly stepped parameter slab to everybody (an all-gather written as N broadcasts: the
        slabs need not be equally long), and the fp32 masters of the foreign slabs re-derived from it are not needed -- a rank
        only ever steps its own slab."""
        r_ = self.reducer
        works = []
        for r in range(r_.world):
            lo, hi = r_.slab(r)
            if lo >= hi:
                continue
            piece = self.flat.flat_param[lo:hi]
            src = dist.get_global_rank(r_.group, r) if r_.group is not None else r
            if r_.backend == "nccl":
                works.append(dist.broadcast(piece, src=src, group=r_.group, async_op=True))
            else:
                tmp = piece.float()
                dist.broadcast(tmp, src=src, group=r_.group)
                piece.copy_(tmp.to(piece.dtype))
        for w in works:
            w.wait()

    def step(self, **batch):
        """One optimisation step on this rank's shard of the batch.  Returns the (detached) loss tensor.

        No host synchronisation anywhere in the step: the reference's NaN guard (forward -> 1-element all_reduce(MAX) of
        `isnan(loss)` -> every rank backpropagates loss * 0 and steps on a zero gradient, train_spark_rwkv7speech.py:
        664-687) keeps its flag on the device -- the all-reduce is enqueued before backward, backward runs regardless,
        and the optimizer kernel reads the flag and substitutes a zero gradient.  Nothing waits for `flag.item()`, so
        the gradient buckets start as soon as backward reaches them."""
        self.flat.arm()
        out = self.model(**batch)
        loss = out.loss
        flag_work = None
        if self.nan_guard:
            self.nan_flag.copy_((~torch.isfinite(loss.detach())).reshape(1))
            if self.world > 1:
                flag_work = dist.all_reduce(self.nan_flag, op=dist.ReduceOp.MAX, async_op=True)  # :664-670
        else:
            self.nan_flag.zero_()
        loss.backward()
        lr = self.current_lr()
        self.last_lr = lr
        if flag_work is not None:
            flag_work.wait()   # enqueued before backward: long done; the optimizer launches below read the flag

        def adamw(lo, hi):
            if hi <= lo:
                return
            if self.hip_adamw:
                import ctypes
                from . import _lib
                P = lambda t: ctypes.c_void_p(t.data_ptr())
                f = ctypes.c_float
                with torch.cuda.device_of(self.master):
                    rc = _lib.lib().rwkv7_adamw_groups_bf16(
                        ctypes.c_long(hi - lo), P(self.master[lo:hi]), P(self.flat.flat_grad[lo:hi]), P(self.exp_avg[lo:hi]),
                        P(self.exp_avg_sq[lo:hi]), P(self.flat.flat_param[lo:hi]), P(self.slab_group[lo // 128:hi // 128]), P(self.group_tab),
                        len(self.group_defs), P(self.nan_flag),
                        f(lr), f(self.betas[0]), f(self.betas[1]), f(self.eps), self.step_idx + 1,
                        ctypes.c_void_p(torch.cuda.current_stream(self.master.device).cuda_stream))
                _lib.check(rc, "adamw")
            else:
                self._torch_adamw(lr, self.nan_flag, lo, hi)

        def on_bucket(runs):   # the runs of one bucket tile 128-aligned pieces of the flat buffers
            for s_, e_ in runs:
                adamw(s_, e_)

        stepped = self.reducer.finish(on_bucket if self.bucket_optimizer and not self.shard_optimizer else None)
        if not stepped:
            adamw(*(self.reducer.slab(self.reducer.rank) if self.shard_optimizer else (0, self.flat.numel)))
        if self.shard_optimizer:
            self._broadcast_slabs()
        self.flat.reattach()   # host-only bookkeeping, behind the optimizer launch (see FlatBuffers.finish_backward)
        for m in self._param_caches:
            m._mix_key = None
        self.step_idx += 1
        return loss.detach()
