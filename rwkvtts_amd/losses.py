"""Loss heads of the three layouts.

fused_linear_cross_entropy : stands in for rwkvfla's FusedLinearCrossEntropyLoss (model/llm/spark_llm.py:8,146-160):
    lm_head GEMM + softmax-CE computed chunk by chunk over the tokens so the [B*T, V] logits (512 MiB at
    B=8,T=4096,V=8193 bf16) are never materialised; gradients w.r.t. hidden and weight are produced in the
    same pass and scaled by the incoming grad in backward.
label_smoothing_kl         : third_party/cosyvoice/transformer/label_smoothing_loss.py:21-96 (Cosy layout).
th_accuracy                : third_party/cosyvoice/utils/common.py:76-95.
"""
import os

import torch
import torch.nn.functional as F

HIP_CE = os.environ.get("RWKV7_HIP_CE", "1") == "1"   # A/B switch: 0 = the torch chain for every head
CHECK_LABELS = os.environ.get("RWKV7_CHECK_LABELS", "0") == "1"
PADDED_HEAD = os.environ.get("RWKV7_PADDED_HEAD", "1") == "1"   # A/B switch (see _FusedLinearCE.forward)
PADDED_HEAD_HITS = [0]


class _FusedLinearCE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, hidden, weight, bias, labels, ignore_index, chunk, label_smoothing):
        N, D = hidden.shape
        valid = labels != ignore_index
        n_valid = valid.sum().clamp(min=1)
        inv = 1.0 / n_valid.float()
        loss = torch.zeros((), dtype=torch.float32, device=hidden.device)
        need = ctx.needs_input_grad
        dh = torch.empty_like(hidden) if need[0] else None
        dw = None   # allocated below (the padded head accumulates into its own [Vp, D] buffer)
        db = torch.zeros_like(bias, dtype=torch.float32) if (bias is not None and need[2]) else None
        # bias and label smoothing (the XY heads, xy_llm.py:233-240) ride on the same kernel since round 4: at configs[3] the torch
        # chain behind them (fp32 logits, logsumexp, softmax, scatter, casts) was ~40 ms of a 417 ms step
        # NUMERICS of the HIP path (also for the XY heads since round 4): the logits are the bf16 output of the head GEMM -- logsumexp,
        # softmax and the smoothing mean term sum(x)/V are computed in fp32 FROM those bf16 logits, where the torch chain below keeps
        # fp32 logits.  The formula is torch's CrossEntropyLoss(label_smoothing, ignore_index); tests/test_heads_reference.py holds
        # bias + smoothing to the training tolerance against the reference's own forward.  Labels must be < V or == ignore_index
        # (the kernel indexes x[label]); RWKV7_CHECK_LABELS=1 asserts it (one host sync per call, debugging only).
        if CHECK_LABELS:
            bad = (labels != ignore_index) & ((labels < 0) | (labels >= weight.shape[0]))
            assert not bool(bad.any()), f"labels outside [0, {weight.shape[0]}) that are not ignore_index={ignore_index}"
        hip_ce = (HIP_CE and hidden.is_cuda and hidden.dtype == torch.bfloat16 and weight.dtype == torch.bfloat16
                  and (bias is None or bias.dtype == torch.bfloat16) and 0 <= label_smoothing < 1 and labels.dtype == torch.int64)
        # Round 6 (late): a vocabulary that is not a multiple of 256 (the Spark head: 8 193) makes every row of the logits start on a 2-byte
        # boundary and gives the three head GEMMs an odd leading dimension (0.75 PF/s in the library).  With PADDED_HEAD the logits live in
        # a [rows, Vp] buffer, Vp = V rounded up to 256, the weight in a zero-padded [Vp, D] copy: the logits GEMM runs on the own kernel
        # (rwkv7_gemm_nt_bf16: Vp % 256 == 0), the padding columns are exactly 0 before and after the loss kernel (which never touches them),
        # so the two gradient GEMMs run over Vp with aligned operands and give the same sums.
        V = weight.shape[0]
        Vp = -(-V // 256) * 256
        padded = (hip_ce and PADDED_HEAD and bias is None and Vp != V and D % 1024 == 0 and N % 256 == 0 and chunk % 256 == 0
                  and hidden.is_contiguous())
        if need[1] and not padded:
            dw = torch.zeros_like(weight, dtype=torch.float32)
        if padded:
            w_pad = torch.zeros(Vp, D, dtype=weight.dtype, device=weight.device)
            w_pad[:V].copy_(weight)
            dw_pad = torch.zeros(Vp, D, dtype=torch.float32, device=weight.device) if need[1] else None
            PADDED_HEAD_HITS[0] += 1
        for s in range(0, N, chunk):
            h = hidden[s:s + chunk]
            lab = labels[s:s + chunk]
            if hip_ce:
                # bf16 logits straight from the GEMM; one kernel turns them into per-row losses and d loss / d logits
                import ctypes
                from . import _lib
                rows = h.shape[0]
                stream = ctypes.c_void_p(torch.cuda.current_stream(h.device).cuda_stream)
                if padded:
                    pd = torch.empty(rows, Vp, dtype=h.dtype, device=h.device)
                    with torch.cuda.device_of(h):
                        rc = _lib.lib().rwkv7_gemm_nt_bf16(rows, Vp, D, ctypes.c_void_p(h.data_ptr()), ctypes.c_void_p(w_pad.data_ptr()),
                                                           ctypes.c_void_p(pd.data_ptr()), 0, stream)
                    _lib.check(rc, "gemm_nt (head logits)")
                else:
                    pd = F.linear(h, weight, bias)   # bf16 logits as nn.Linear gives them (the bias inside the GEMM's fp32 epilogue)
                loss_rows = torch.empty(rows, dtype=torch.float32, device=pd.device)
                lab_c = lab.contiguous()
                with torch.cuda.device_of(pd):
                    rc = _lib.lib().rwkv7_ce_fwd_bwd_ld_bf16(
                        ctypes.c_long(rows), V, ctypes.c_long(pd.shape[1]), ctypes.c_void_p(pd.data_ptr()), ctypes.c_void_p(lab_c.data_ptr()),
                        ctypes.c_long(ignore_index), ctypes.c_float(1.0), ctypes.c_void_p(loss_rows.data_ptr()),
                        ctypes.c_float(float(label_smoothing)), stream)
                _lib.check(rc, "ce_fwd_bwd")
                loss += loss_rows.sum()
                if need[0]:
                    dh[s:s + chunk] = pd @ (w_pad if padded else weight)
                if need[1]:
                    if padded:
                        dw_pad += (pd.t() @ h).float()
                    else:
                        dw += (pd.t() @ h).float()
                if db is not None:
                    db += pd.sum(0, dtype=torch.float32)
                continue
            logits = (h @ weight.t()).float()
            if bias is not None:
                logits = logits + bias.float()
            lse = torch.logsumexp(logits, dim=-1)
            vmask = lab != ignore_index
            safe = lab.clamp(min=0)
            tgt = logits.gather(1, safe.unsqueeze(1)).squeeze(1)
            nll = lse - tgt
            if label_smoothing > 0:
                smooth = lse - logits.mean(dim=-1)
                per = (1 - label_smoothing) * nll + label_smoothing * smooth
            else:
                per = nll
            loss += (per * vmask).sum()
            if need[0] or need[1]:
                p = torch.softmax(logits, dim=-1)
                if label_smoothing > 0:
                    p = p - label_smoothing / logits.shape[-1]
                    p.scatter_add_(1, safe.unsqueeze(1), torch.full_like(tgt, -(1 - label_smoothing)).unsqueeze(1))
                else:
                    p.scatter_add_(1, safe.unsqueeze(1), torch.full_like(tgt, -1.0).unsqueeze(1))
                p = p * (vmask.unsqueeze(1) * inv)
                pd = p.to(hidden.dtype)
                if need[0]:
                    dh[s:s + chunk] = pd @ weight
                if need[1]:
                    dw += (pd.t() @ h).float()
                if db is not None:
                    db += p.sum(0)
        # the HIP path leaves dh / dw unscaled (scale = 1 in the kernel: 1/n_valid lives on the device); backward folds
        # 1/n_valid into the incoming gradient
        if padded and need[1]:
            dw = dw_pad[:V]
        ctx.save_for_backward(dh, dw, db, inv if hip_ce else None)
        ctx.wdtype = weight.dtype
        return loss * inv

    @staticmethod
    def backward(ctx, g):
        dh, dw, db, post = ctx.saved_tensors
        if post is not None:
            g = g * post
        # scale in fp32 and round once: rounding the scalar to bf16 first would put a systematic error of up to 2^-9 on
        # every hidden-state gradient relative to the (fp32-scaled) head-weight gradient
        return ((dh.float() * g).to(dh.dtype) if dh is not None else None,
                (dw * g).to(ctx.wdtype) if dw is not None else None,
                (db * g).to(ctx.wdtype) if db is not None else None, None, None, None, None)


LOGITS_CHUNK_BYTES = int(os.environ.get("RWKV7_CE_CHUNK_BYTES", str(1 << 30)))


def auto_chunk(rows, V, esize=2):
    """Rows per chunk of the head GEMMs: as many as keep the chunk's logits under LOGITS_CHUNK_BYTES (1 GiB of the GPU's 288), in
    multiples of 4096.  The first cut used 4096 rows whatever the vocabulary: eight rounds of three 69-GFLOP GEMMs for the Spark head
    (V = 8193), each too small to fill the chip (0.69-0.80 PF/s, 2.96 ms per step for the node, profiles/r05q_ce_head_probe.txt); the
    whole batch as ONE chunk is 0.54 GB of logits.  The 66 661-wide XY head keeps 4096-row chunks (0.55 GB each)."""
    c = max(4096, (LOGITS_CHUNK_BYTES // (esize * V)) // 4096 * 4096)
    return rows if c >= rows else c


def fused_linear_cross_entropy(hidden, labels, weight, bias=None, ignore_index=-100, chunk=None,
                               label_smoothing=0.0):
    """hidden [..., D], labels [...] (already shifted by the caller) -> mean CE over labels != ignore_index.
    chunk: rows per round of head GEMMs (None: auto_chunk)."""
    D = hidden.shape[-1]
    if chunk is None:
        # the HIP kernel path keeps ONE bf16 logits tensor per chunk; the torch chain (RWKV7_HIP_CE=0, fp32 models, int32 labels, a
        # non-bf16 bias) keeps fp32 logits, an fp32 softmax and a copy in the hidden dtype: budget 3 x 4 bytes per logit there
        hip = (HIP_CE and hidden.is_cuda and hidden.dtype == torch.bfloat16 and weight.dtype == torch.bfloat16
               and (bias is None or bias.dtype == torch.bfloat16) and labels.dtype == torch.int64)
        chunk = auto_chunk(hidden.numel() // D, weight.shape[0], hidden.element_size() if hip else 12)
    return _FusedLinearCE.apply(hidden.reshape(-1, D), weight, bias, labels.reshape(-1), ignore_index, chunk,
                                label_smoothing)


def label_smoothing_kl(logits, target, size, padding_idx, smoothing, normalize_length=False):
    """LabelSmoothingLoss.forward (cosyvoice/transformer/label_smoothing_loss.py:68-96): KL(true_dist || softmax)
    with true_dist = smoothing/(size-1) off-target, 1-smoothing on target; ignored rows zeroed; divided by the
    number of valid tokens (normalize_length) or by the batch size."""
    B = logits.shape[0]
    x = logits.reshape(-1, size)
    t = target.reshape(-1)
    ignore = t == padding_idx
    total = t.numel() - ignore.sum()
    t = t.masked_fill(ignore, 0)
    true_dist = torch.full_like(x, smoothing / (size - 1), dtype=torch.float32)
    true_dist.scatter_(1, t.unsqueeze(1), 1.0 - smoothing)
    kl = F.kl_div(torch.log_softmax(x.float(), dim=1), true_dist, reduction="none")
    denom = total if normalize_length else B
    return kl.masked_fill(ignore.unsqueeze(1), 0).sum() / denom


def th_accuracy(pad_outputs, pad_targets, ignore_label):
    """cosyvoice/utils/common.py:76-95."""
    pred = pad_outputs.view(pad_targets.size(0), pad_targets.size(1), pad_outputs.size(1)).argmax(2)
    mask = pad_targets != ignore_label
    num = torch.sum(pred.masked_select(mask) == pad_targets.masked_select(mask))
    return (num / torch.sum(mask)).detach()
