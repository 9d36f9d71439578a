"""RWKV-7 backbone for MI355X: the model the reference trains/serves through rwkvfla
(RWKV7Model / RWKV7Block / RWKV7Attention / RWKV7FeedForward / LoRA / Cache), rebuilt on the HIP ops.

Call sites in the reference: model/llm/spark_llm.py:24,126-136 ; cosy_llm.py:31,132-141 ; xy_llm.py:155,219-227.
Arithmetic spec (in-tree twin): model/llm/rwkv_s2s_single_ffn.py:158-259 (training), :417-445,482-556 (stateful),
model/llm/rwkv_asr_cuda_whisper.py:181-326 (batched stateful).

state_dict keys are rwkvfla's (left-hand side of utils/convert_rwkv.py:17-30): both the fused `attn.x_x [6,D]`
(order r,w,k,v,a,g -- third_party/cosyvoice/cli/model.py:99-111) and the split `attn.x_r .. x_g [1,1,D]`
checkpoints load.  LoRA weights are [out,in] (utils/convert_rwkv.py:26-27).

Host code is PyTorch-ROCm plumbing (device memory, streams, autograd tape, library GEMMs for the dense
projections); the recurrence and the fused elementwise stages run in librwkv7_hip.so (rwkvtts_amd/ops.py,
rwkvtts_amd/fused.py).  There is no CPU path: tensors must live on the HIP device.
"""
from __future__ import annotations

import json
import math
import os
from dataclasses import asdict, dataclass, field
from typing import List, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import fused, ops

HEAD_SIZE = 64


@dataclass
class RWKV7Config:
    """Subset of rwkvfla's RWKV7Config that the reference's checkpoints use (model/test/audio_rwkv.config)."""
    hidden_size: int = 1024
    num_hidden_layers: int = 24
    head_dim: int = 64
    vocab_size: int = 8193
    decay_low_rank_dim: int = 64
    a_low_rank_dim: int = 64
    v_low_rank_dim: int = 32
    gate_low_rank_dim: int = 128
    hidden_ratio: float = 4.0
    intermediate_size: Optional[int] = None
    norm_eps: float = 1e-5
    norm_bias: bool = True
    fuse_cross_entropy: bool = True
    initializer_range: float = 0.006
    model_type: str = "rwkv7"
    extra: dict = field(default_factory=dict)

    def __post_init__(self):
        if self.intermediate_size is None:
            self.intermediate_size = int(self.hidden_size * self.hidden_ratio)
        assert self.head_dim == HEAD_SIZE, "the HIP kernels are built for head size 64 (reference: -D_N_=64)"
        assert self.hidden_size % self.head_dim == 0

    @property
    def num_heads(self):
        return self.hidden_size // self.head_dim

    @classmethod
    def from_dict(cls, d: dict):
        names = set(cls.__dataclass_fields__) - {"extra"}
        known = {k: v for k, v in d.items() if k in names}
        cfg = cls(**known)
        cfg.extra = {k: v for k, v in d.items() if k not in names}
        return cfg

    @property
    def auto_map(self):
        """config.json's `auto_map` (model/test/audio_rwkv.config:9-13): transformers' Auto* loaders read it from the config
        object to find the remote-code classes."""
        return self.extra.get("auto_map") or {}

    def to_dict(self):
        d = asdict(self)
        extra = d.pop("extra")
        d.update(extra)
        return d

    @classmethod
    def from_pretrained(cls, path, return_unused_kwargs=False, **kwargs):
        """config.json of a checkpoint directory.  Signature as transformers' AutoConfig calls it for `auto_map` classes."""
        with open(os.path.join(path, "config.json")) as f:
            cfg = cls.from_dict(json.load(f))
        drop = {"trust_remote_code", "code_revision", "cache_dir", "force_download", "local_files_only", "token", "revision",
                "proxies", "subfolder", "name_or_path", "_from_auto", "_commit_hash"}
        rest = {k: v for k, v in kwargs.items() if k not in drop}
        return (cfg, rest) if return_unused_kwargs else cfg

    @classmethod
    def register_for_auto_class(cls, auto_class="AutoConfig"):
        """called by transformers' AutoConfig on `auto_map` classes; nothing to record"""


# canonical sizes (read config.json for real checkpoints)
def config_0p1b(**kw):
    return RWKV7Config(hidden_size=768, num_hidden_layers=12, **kw)


def config_0p4b(**kw):
    return RWKV7Config(hidden_size=1024, num_hidden_layers=24, **kw)


def config_1p5b(**kw):
    return RWKV7Config(hidden_size=2048, num_hidden_layers=24, decay_low_rank_dim=96, a_low_rank_dim=96,
                       v_low_rank_dim=64, gate_low_rank_dim=256, **kw)


class Linear(nn.Linear):
    """nn.Linear (same parameters / state_dict keys) whose weight gradient is reduced over B*T in slabs
    (fused.wgrad_splitk) when training in bf16."""

    def forward(self, x):
        return fused.linear(x, self.weight, self.bias)


class LoRA(nn.Module):
    """rwkvfla LoRA: Linear(in,r,bias=False) -> act -> Linear(r,out,bias) ; keys lora.0.weight / lora.2.{weight,bias}."""

    def __init__(self, in_dim, out_dim, rank, activation: Optional[str], bias: bool):
        super().__init__()
        act = {None: nn.Identity(), "tanh": nn.Tanh(), "sigmoid": nn.Sigmoid()}[activation]
        self.lora = nn.Sequential(Linear(in_dim, rank, bias=False), act, Linear(rank, out_dim, bias=bias))
        self.activation, self.rank = activation, rank

    def forward(self, x):
        if fused.lora_decode_supported(x, self.rank) and self.lora[0].weight.dtype == torch.bfloat16:
            return fused.lora_decode(x, self.lora[0].weight, self.lora[2].weight, self.lora[2].bias, self.activation)
        return self.lora(x)  # fp32 models and decode-sized inputs: BLAS

    def forward_from_hidden(self, h):
        """The branch after its down projection (h = x @ lora.0.weight^T computed elsewhere, fused.dual_linear)."""
        return self.lora[2](self.lora[1](h))


class LayerState:
    """Per-layer recurrent state (reference: rwkv_asr_cuda_whisper.py:443-447; fla Cache entries
    conv_state / recurrent_state / ffn_state, model/llm/llm.py:250-252).
      att_x_prev [B,D]  last input of the time-mix block      (fla: conv_state)
      att_kv     [B,H,64,64] fp32, row = value idx, col = key idx (fla: recurrent_state, transposed)
      ffn_x_prev [B,D]  last input of the channel-mix block   (fla: ffn_state)"""
    __slots__ = ("att_x_prev", "att_kv", "ffn_x_prev")

    def __init__(self, att_x_prev, att_kv, ffn_x_prev):
        self.att_x_prev, self.att_kv, self.ffn_x_prev = att_x_prev, att_kv, ffn_x_prev

    # dict-style access with the fla names, as the reference pokes at them (llm.py:250-252)
    def __getitem__(self, k):
        return {"conv_state": self.att_x_prev, "recurrent_state": self.att_kv, "ffn_state": self.ffn_x_prev}[k]


class Cache:
    """Minimal stand-in for rwkvfla's Cache: a list of LayerState plus seen_tokens (llm.py:254)."""

    def __init__(self, states: Optional[List[LayerState]] = None, seen_tokens: int = 0):
        self.states = states or []
        self.seen_tokens = seen_tokens

    def __len__(self):
        return len(self.states)

    def __getitem__(self, i):
        return self.states[i]

    @classmethod
    def zeros(cls, cfg: RWKV7Config, B, device, dtype):
        H = cfg.num_heads
        return cls([LayerState(torch.zeros(B, cfg.hidden_size, device=device, dtype=dtype),
                               torch.zeros(B, H, HEAD_SIZE, HEAD_SIZE, device=device, dtype=torch.float32),
                               torch.zeros(B, cfg.hidden_size, device=device, dtype=dtype))
                    for _ in range(cfg.num_hidden_layers)])


def mark_all_ones(attention_mask: torch.Tensor, all_ones: bool):
    """Host-side knowledge about a mask (True: no padding anywhere; False: padded): RWKV7Model.forward then decides
    whether to apply it without reading it back from the device -- `bool(mask.all())` is a host synchronisation in the
    middle of the training step, which delays the first gradient bucket of the previous step's overlap window.
    The hint rides on the tensor object: it is lost by .to() / slicing (safe: the model then reads the mask) but SURVIVES in-place
    edits -- mark the mask after the last edit, or pass RWKV7Model.forward(attention_mask_all_ones=...) explicitly, which takes
    precedence.  With RWKV7_CHECK_MASK_HINT=1 in the environment (tests/conftest.py sets it) every hint is checked against the
    mask itself."""
    attention_mask._rwkv7_all_ones = bool(all_ones)
    return attention_mask


# Training runs prepare -> scan -> post as one autograd node (fused._TmixCore: row-split scan backward, gradient sums
# folded into the prepare backward).  False selects the three separate nodes (same forward kernels).
FUSED_TMIX_CORE = True
# training: residual add + LayerNorm + token-shift lerp(s) as one kernel each way (rwkv7_add_ln_mix_fwd / rwkv7_mix_add_ln_bwd).
# Measured on MI355X (tools/bench_add_ln_mix.py, B*T = 32768, D = 1024): channel-mix side (1 lerp) forward 56 us against 79 us
# for the two separate stages, backward equal -> on.  Time-mix side (6 lerps): forward 140 against 170 us, but the one-pass
# backward carries 6 x 3 x 8 values per thread next to the LayerNorm backward, spills, runs at two waves per SIMD with a
# barrier per row: 560 against 320 us -> off (a 4-channels-per-thread variant of that backward: 176 registers, 365 us -- still
# behind the two separate kernels, not kept).
FUSED_ADD_LN_MIX1 = True
DUAL_LINEAR_XV = True   # training: value projection + value-residual down projection as one autograd node (fused._DualLinear)
FUSED_ADD_LN_MIX6 = False
# round 4: the six-lerp side's one-pass FORWARD alone (rwkv7_add_ln_mix_fwd_h: it also stores h, the backward runs as the two separate
# kernels).  Measured in the same-box A/B (tools/ab_step.py): +0.15 ms per step -- the extra 64 MiB store eats the 30 us the one-pass
# forward is ahead; off.
FUSED_ADD_LN_MIX6_FWD = False
def _row_align(n):
    """Rows of the re-laid-out packed row: a multiple of the 32-step chunk, and of 256 once the row is long enough for the own GEMMs'
    256-row tiles to matter (fused.cmix_eligible / linear_add need M % 256 == 0: a 33 248-position row fell off those paths and ran
    50 % slower than the same tokens as [8, 4096])."""
    return (n + 255) // 256 * 256 if n >= 2048 else (n + 31) // 32 * 32


# cu_seqlens batches run on the chunked kernels' sequence flags (bf16); False: always unpack into a padded masked batch
PACKED_NATIVE = True


class RWKV7Attention(nn.Module):
    """Time-mix block (rwkv_s2s_single_ffn.py:158-196; Appendix A of SURVEY.md)."""

    def __init__(self, cfg: RWKV7Config, layer_idx: int):
        super().__init__()
        D, H, N = cfg.hidden_size, cfg.num_heads, cfg.head_dim
        self.layer_idx, self.hidden_size, self.num_heads, self.head_dim = layer_idx, D, H, N
        for n in "rwkvag":
            setattr(self, f"x_{n}", nn.Parameter(torch.zeros(1, 1, D)))
        self.k_k = nn.Parameter(torch.zeros(D))
        self.k_a = nn.Parameter(torch.zeros(D))
        self.r_k = nn.Parameter(torch.zeros(H, N))
        self.r_proj = Linear(D, D, bias=False)
        self.k_proj = Linear(D, D, bias=False)
        self.v_proj = Linear(D, D, bias=False)
        self.o_proj = Linear(D, D, bias=False)
        self.w_lora = LoRA(D, D, cfg.decay_low_rank_dim, "tanh", True)
        if layer_idx != 0:
            self.v_lora = LoRA(D, D, cfg.v_low_rank_dim, None, True)
        self.a_lora = LoRA(D, D, cfg.a_low_rank_dim, None, True)
        self.g_lora = LoRA(D, D, cfg.gate_low_rank_dim, "sigmoid", False)
        self.g_norm = nn.GroupNorm(H, D, eps=N * cfg.norm_eps)  # 64e-5, rwkv_s2s_single_ffn.py:504

    def _load_from_state_dict(self, state_dict, prefix, *args, **kw):
        # fused x_x [6,D] checkpoints (RWKV7Attention "version 1", cosyvoice/cli/model.py:99-111)
        key = prefix + "x_x"
        if key in state_dict:
            x_x = state_dict.pop(key)
            for i, n in enumerate("rwkvag"):
                state_dict[prefix + f"x_{n}"] = x_x[i].reshape(1, 1, -1)
        super()._load_from_state_dict(state_dict, prefix, *args, **kw)

    def train(self, mode: bool = True):
        self._mix_key = None   # optimizers may rewrite parameter memory without touching the version counters
        return super().train(mode)

    def _stacked_mix(self, dtype):
        """[6,D] stack of x_r..x_g for inference (no autograd through it), rebuilt when a parameter changes or the module
        switches between train() and eval()."""
        if torch.is_grad_enabled() or self.training:
            return None
        ps = (self.x_r, self.x_w, self.x_k, self.x_v, self.x_a, self.x_g)
        key = tuple(p._version for p in ps) + tuple(p.data_ptr() for p in ps) + (dtype,)
        if getattr(self, "_mix_key", None) != key:
            self._mix_cache = torch.cat([p.detach().reshape(1, -1) for p in ps], 0).to(dtype)
            self._mix_key = key
        return self._mix_cache

    def forward(self, x, mask, v_first, state: Optional[LayerState] = None, seq_start=None, resid=None):
        """x [B,T,D] (LayerNorm'ed), mask [B,T,1] or None.  Returns (out, v_first); resid: see forward_mixed.
        With `state`, token shift and the WKV state are carried (and updated in place).
        seq_start (int32 [nseq+1] chunk offsets): packed rows, see RWKV7Model._forward_packed."""
        x_prev = None if state is None else state.att_x_prev
        if FUSED_TMIX_CORE and fused.mix_lora_supported(x, state, seq_start, lambda: self.lora_branches()[1], mask):
            # training: the four low-rank branches' down projections taken THROUGH the lerp (fused.mix_lora): x_w, x_a, x_g and the
            # branch copy of x_v are never formed
            mus, w1s, acts, names = self.lora_branches()
            xr, xk, xv, hs = fused.mix_lora(x, mask, self.x_r, self.x_k, self.x_v, mus, w1s, acts)
            return self.forward_mixed((xr, None, xk, xv, None, None), x, mask, v_first, state, seq_start, resid, dict(zip(names, hs)))
        mixed = fused.token_shift_mix6(x, x_prev, self.x_r, self.x_w, self.x_k, self.x_v,
                                       self.x_a, self.x_g, mask, self._stacked_mix(x.dtype))
        return self.forward_mixed(mixed, x, mask, v_first, state, seq_start, resid)

    def lora_branches(self):
        """(lerp coefficients, Linear(D, r) weights, activation names, keys) of the low-rank branches, in fused.mix_lora's order."""
        loras = [self.w_lora, self.a_lora] + ([self.v_lora] if self.layer_idx != 0 else []) + [self.g_lora]
        mus = [self.x_w, self.x_a] + ([self.x_v] if self.layer_idx != 0 else []) + [self.x_g]
        names = ("w", "a") + (("v",) if self.layer_idx != 0 else ()) + ("g",)
        return mus, [l.lora[0].weight for l in loras], [l.activation for l in loras], names

    def mix_params(self):
        return (self.x_r, self.x_w, self.x_k, self.x_v, self.x_a, self.x_g)

    def forward_mixed(self, mixed, x, mask, v_first, state: Optional[LayerState] = None, seq_start=None, resid=None, hid=None):
        """The block after the token-shift lerps (`mixed` = xr, xw, xk, xv, xa, xg; with `hid` -- the low-rank branches' hidden
        pre-activations from fused.mix_lora -- only xr, xk, xv are given); x (the LayerNorm'ed input) is only read for
        the carried state and may be None without one.  resid (training one-pass path): the residual stream; if the output
        projection can take the add as its epilogue the first return value is a pair (resid + out, True)."""
        xr, xw, xk, xv, xa, xg = mixed
        B, T, D = xr.shape
        H, N = self.num_heads, self.head_dim
        r = self.r_proj(xr)
        k = self.k_proj(xk)
        v_lo = None
        if hid is not None:   # the branches' hidden pre-activations came through the lerp (fused.mix_lora)
            v = self.v_proj(xv)
            w_pre = self.w_lora.lora[2](hid["w"])     # hid: the branches' hidden states, activation applied
            a_pre = self.a_lora.lora[2](hid["a"])
            g = self.g_lora.lora[2](hid["g"])
            v_pre = None if self.layer_idx == 0 else self.v_lora.lora[2](hid["v"])
            if self.layer_idx == 0:
                if mask is not None:
                    v = v * mask
                v_first = v
            if mask is not None:
                r = r * mask
            y, vf_next = fused.tmix_core(r, w_pre, k, v, a_pre, g, v_pre, v_first, self.k_k, self.k_a, self.g_norm.weight,
                                         self.g_norm.bias, self.r_k, mask, H, self.g_norm.eps, self.layer_idx == 0, seq_start)
            if resid is not None and self.o_proj.bias is None:
                x1 = fused.linear_add(y, self.o_proj.weight, resid)
                if x1 is not None:
                    return (x1, True), (v_first if vf_next is None else vf_next)
            return self.o_proj(y), (v_first if vf_next is None else vf_next)
        if (self.layer_idx != 0 and DUAL_LINEAR_XV and self.v_proj.bias is None
                and fused.dual_linear_supported(xv, self.v_proj.weight, self.v_lora.lora[0].weight)):
            # xv feeds the value projection AND the value-residual branch: one autograd node, the input gradient without an add pass
            v, v_lo = fused.dual_linear(xv, self.v_proj.weight, self.v_lora.lora[0].weight)
        else:
            v = self.v_proj(xv)
        w_pre = self.w_lora(xw)
        a_pre = self.a_lora(xa)
        g = self.g_lora(xg)
        if self.layer_idx == 0:
            v_pre = None
        else:
            v_pre = self.v_lora.forward_from_hidden(v_lo) if v_lo is not None else self.v_lora(xv)
        if self.layer_idx == 0:
            if mask is not None:
                v = v * mask  # rwkv_s2s_single_ffn.py:178: v is masked before it becomes v_first
            v_first = v
        if mask is not None:
            r = r * mask
        if seq_start is not None or (state is None and torch.is_grad_enabled() and (r.requires_grad or w_pre.requires_grad)
                                     and FUSED_TMIX_CORE):
            y, vf_next = fused.tmix_core(r, w_pre, k, v, a_pre, g, v_pre, v_first, self.k_k, self.k_a, self.g_norm.weight,
                                         self.g_norm.bias, self.r_k, mask, H, self.g_norm.eps, self.layer_idx == 0, seq_start)
            if resid is not None and self.o_proj.bias is None:
                x1 = fused.linear_add(y, self.o_proj.weight, resid)
                if x1 is not None:
                    return (x1, True), (v_first if vf_next is None else vf_next)
            return self.o_proj(y), (v_first if vf_next is None else vf_next)
        w, k2, v2, a_in, b_in = fused.tmix_prepare(w_pre, k, v, a_pre, v_pre, v_first, self.k_k, self.k_a, mask,
                                                   H, self.layer_idx == 0)
        if state is None:
            if torch.is_grad_enabled() and (r.requires_grad or w.requires_grad):
                y = ops.RUN_CUDA_RWKV7g(r, w, k2, v2, a_in, b_in)
            else:
                y = ops.wkv7_forward_nograd(r, w, k2, v2, a_in, b_in)
        else:
            y = ops.RWKV7_BATCH_OP(state.att_kv, r.contiguous(), w, k2, v2, a_in, b_in)
            last = x[:, -1].detach()
            # in place: the state tensors keep their addresses (hipGraph-captured decode steps replay on them)
            state.att_x_prev.copy_(last * mask[:, -1] if mask is not None else last)
        y = fused.tmix_post(y, r, k2, v2, g, self.g_norm.weight, self.g_norm.bias, self.r_k, H, self.g_norm.eps)
        return self.o_proj(y), v_first


class RWKV7FeedForward(nn.Module):
    """Channel-mix block (rwkv_s2s_single_ffn.py:223-230): relu(key(x + (shift(x)-x) x_k))^2 -> value."""

    def __init__(self, cfg: RWKV7Config, layer_idx: int):
        super().__init__()
        self.x_k = nn.Parameter(torch.zeros(cfg.hidden_size))
        self.key = Linear(cfg.hidden_size, cfg.intermediate_size, bias=False)
        self.value = Linear(cfg.intermediate_size, cfg.hidden_size, bias=False)

    def forward(self, x, mask, state: Optional[LayerState] = None):
        x_prev = None if state is None else state.ffn_x_prev
        kx = fused.token_shift_mix1(x, x_prev, self.x_k, mask)
        if state is not None:
            last = x[:, -1].detach()
            state.ffn_x_prev.copy_(last * mask[:, -1] if mask is not None else last)
        return self.forward_mixed(kx)

    def forward_mixed(self, kx):
        if self.key.bias is None and self.value.bias is None:
            out = fused.channel_mix(kx, self.key.weight, self.value.weight)   # the activation inside both GEMMs (own MFMA kernel)
            if out is not None:
                return out
        s = fused.key_relu_sq(kx, self.key.weight) if self.key.bias is None else None   # GEMM with the activation as epilogue
        if s is None:
            h = self.key(kx)
            if self.value.bias is None:
                out = fused.relu_sq_value(h, self.value.weight)   # training: the activation's backward rides in the value dgrad GEMM
                if out is not None:
                    return out
            s = fused.relu_sq(h)
        return self.value(s)


class RWKV7Block(nn.Module):
    def __init__(self, cfg: RWKV7Config, layer_idx: int):
        super().__init__()
        self.layer_idx = layer_idx
        D = cfg.hidden_size
        if layer_idx == 0:
            self.pre_norm = nn.LayerNorm(D, eps=cfg.norm_eps, bias=cfg.norm_bias)
        self.attn_norm = nn.LayerNorm(D, eps=cfg.norm_eps, bias=cfg.norm_bias)
        self.attn = RWKV7Attention(cfg, layer_idx)
        self.ffn_norm = nn.LayerNorm(D, eps=cfg.norm_eps, bias=cfg.norm_bias)
        self.ffn = RWKV7FeedForward(cfg, layer_idx)

    def forward(self, x, delta, mask, v_first, state: Optional[LayerState] = None, seq_start=None):
        """The block input is x + delta (delta = the previous block's channel-mix output, None for the first
        block): every residual add is fused with the LayerNorm that follows it (fused.add_layer_norm), so this
        block's own last add is left to the next block / the model's final norm.  Returns (x, delta, v_first)."""
        if self.layer_idx == 0:
            if delta is not None:
                x = x + delta
                delta = None
            x = fused.layer_norm(x, self.pre_norm)
        one_pass = fused.add_ln_mix_supported(x, state)
        if one_pass and (FUSED_ADD_LN_MIX6 or FUSED_ADD_LN_MIX6_FWD):
            x, mixed = fused.add_layer_norm_mix(x, delta, self.attn_norm, mask, self.attn.mix_params(), fwd_only=not FUSED_ADD_LN_MIX6)
            att, v_first = self.attn.forward_mixed(mixed, None, mask, v_first, None, seq_start,
                                                   resid=x if FUSED_ADD_LN_MIX1 else None)
        elif (one_pass and FUSED_TMIX_CORE and fused.FUSED_ADD_LN_MIX_LORA_FWD and x.requires_grad
              and fused.mix_lora_supported(x, state, seq_start, self.attn.lora_branches()[1])):
            # residual add + LayerNorm + the three lerps of the full projections in one forward kernel; the low-rank branches take
            # their inputs through the lerp from the stored LayerNorm output (fused.add_layer_norm_mix_lora)
            a = self.attn
            mus, w1s, acts, names = a.lora_branches()
            x, xr, xk, xv, hs = fused.add_layer_norm_mix_lora(x, delta, self.attn_norm, mask, a.x_r, a.x_k, a.x_v, mus, w1s, acts)
            att, v_first = a.forward_mixed((xr, None, xk, xv, None, None), None, mask, v_first, None, seq_start,
                                           x if FUSED_ADD_LN_MIX1 else None, dict(zip(names, hs)))
        else:
            if delta is None:
                h = fused.layer_norm(x, self.attn_norm)
            else:
                x, h = fused.add_layer_norm(x, delta, self.attn_norm)
            att, v_first = self.attn(h, mask, v_first, state, seq_start, resid=x if (one_pass and FUSED_ADD_LN_MIX1) else None)
        if one_pass and FUSED_ADD_LN_MIX1:
            # training path: add + LayerNorm + token-shift lerp in one pass each way (h / dh never reach HBM)
            if isinstance(att, tuple):   # the add already happened in the output projection's epilogue
                x, (kx,) = fused.add_layer_norm_mix(att[0], None, self.ffn_norm, mask, (self.ffn.x_k,))
            else:
                x, (kx,) = fused.add_layer_norm_mix(x, att, self.ffn_norm, mask, (self.ffn.x_k,))
            return x, self.ffn.forward_mixed(kx), v_first
        x, h = fused.add_layer_norm(x, att, self.ffn_norm)
        return x, self.ffn(h, mask, state), v_first


class ModelOutput(dict):
    """Attribute + index access like transformers' ModelOutput (the reference reads .loss/.logits/
    .past_key_values, train_spark_rwkv7speech.py:238-242, and indexes outputs[0], spark_llm.py:138)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __getitem__(self, k):
        if isinstance(k, int):
            return [v for v in self.values() if v is not None][k]
        return super().__getitem__(k)


class RWKV7Model(nn.Module):
    """Backbone: embeddings -> L blocks -> final LayerNorm.  forward() mirrors
    RWKV7Model(input_ids|inputs_embeds, attention_mask, past_key_values, use_cache) (llm.py:148-155)."""

    def __init__(self, cfg: RWKV7Config):
        super().__init__()
        self.config = cfg
        self.embeddings = nn.Embedding(cfg.vocab_size, cfg.hidden_size)
        self.layers = nn.ModuleList([RWKV7Block(cfg, i) for i in range(cfg.num_hidden_layers)])
        self.norm = nn.LayerNorm(cfg.hidden_size, eps=cfg.norm_eps, bias=cfg.norm_bias)
        self.gradient_checkpointing = False

    def forward(self, input_ids=None, attention_mask=None, inputs_embeds=None, past_key_values: Optional[Cache] = None,
                use_cache: Optional[bool] = None, cu_seqlens=None, attention_mask_all_ones: Optional[bool] = None, **kwargs):
        if (input_ids is None) == (inputs_embeds is None):
            raise ValueError("You must specify exactly one of input_ids or inputs_embeds")
        x = self.embeddings(input_ids) if inputs_embeds is None else inputs_embeds
        if cu_seqlens is not None:
            return self._forward_packed(x, cu_seqlens)
        B, T, D = x.shape
        if not x.is_cuda:
            raise RuntimeError("RWKV7Model runs on the HIP device only (no CPU path); move the model and inputs to cuda")
        mask = None
        if attention_mask is not None:
            # an all-ones mask (unpadded batches, the common training case) is dropped: multiplying by it is the
            # identity and only costs HBM traffic.  Batch builders that know the answer on the host say so
            # (mark_all_ones); only an unmarked mask costs a device round trip here.
            known = attention_mask_all_ones if attention_mask_all_ones is not None else getattr(attention_mask, "_rwkv7_all_ones", None)
            if known is not None and os.environ.get("RWKV7_CHECK_MASK_HINT") == "1":
                assert bool(attention_mask[:, -T:].all()) == bool(known), \
                    "attention mask hint (mark_all_ones / attention_mask_all_ones) contradicts the mask: marked before an in-place edit?"
            if not (known if known is not None else bool(attention_mask[:, -T:].all())):
                mask = attention_mask[:, -T:].to(x.dtype).unsqueeze(-1)
        if use_cache and past_key_values is None:
            past_key_values = Cache.zeros(self.config, B, x.device, x.dtype)
        stateful = past_key_values is not None and len(past_key_values) > 0
        pad = 0
        # the training kernels need T % 16 == 0 (reference), the chunked MFMA pair that bf16 training runs T % 32 == 0: a bf16
        # batch is padded to 32 so that none falls back to the scalar kernels (2.7x slower scan)
        gran = ops.CHUNK_T if x.dtype == torch.bfloat16 else ops.CHUNK_LEN
        if not stateful and T % gran != 0:
            # left-pad with masked zeros (rwkv_asr_cuda_whisper.py:482-486)
            pad = gran - T % gran
            x = torch.cat([x.new_zeros(B, pad, D), x], 1)
            m = torch.ones(B, T, 1, dtype=x.dtype, device=x.device) if mask is None else mask
            mask = torch.cat([m.new_zeros(B, pad, 1), m], 1)
        x = self._run_layers(x, mask, past_key_values if stateful else None)
        if pad:
            x = x[:, pad:]
        if stateful:
            past_key_values.seen_tokens += T
        return ModelOutput(last_hidden_state=x, past_key_values=past_key_values if stateful else None)


    def _run_layers(self, x, mask, cache: Optional[Cache], seq_start=None):
        v_first = delta = None
        for i, layer in enumerate(self.layers):
            st = cache[i] if cache is not None else None
            if self.gradient_checkpointing and self.training and cache is None:
                x, delta, v_first = torch.utils.checkpoint.checkpoint(layer, x, delta, mask, v_first, None, seq_start,
                                                                      use_reentrant=False)
            else:
                x, delta, v_first = layer(x, delta, mask, v_first, st, seq_start)
        return fused.add_layer_norm(x, delta, self.norm)[1] if delta is not None else fused.layer_norm(x, self.norm)

    def _forward_packed_device(self, x, cu_seqlens):   # noqa: D401
        """The packed path for a `cu_seqlens` that lives on the DEVICE (as fla consumes it; train_spark_rwkv7speech.py:238-239): no
        host read-back.  The 32-aligned layout of `_forward_packed` is computed with tensor ops from the cumulative lengths; only its
        SIZE must be known on the host, and that is bounded by shapes alone: every non-empty sequence grows by at most 32 positions, so
        the aligned row has at most total + 32 nseq positions (rounded up to a chunk).  Chunks between the last sequence and that bound
        form one extra all-masked pseudo-sequence, so no chunk is left to uninitialised memory.  Empty sequences own empty chunk ranges
        (the kernels return at once); positions outside [cu[0], cu[-1]) come back as zeros."""
        C = ops.CHUNK_T
        total, D = x.shape[1], x.shape[-1]
        nseq = cu_seqlens.numel() - 1
        t_max = _row_align(total + C * nseq)
        cu = cu_seqlens.to(torch.int64)
        lens = cu[1:] - cu[:-1]
        alen = torch.where(lens > 0, (torch.div(lens, C, rounding_mode="floor") + 1) * C, torch.zeros_like(lens))
        ends = torch.cumsum(alen, 0)
        starts = ends - alen
        j = torch.arange(total, device=x.device)
        sq = torch.searchsorted(cu[1:].contiguous(), j, right=True).clamp_(max=nseq - 1)
        valid = (j >= cu[0]) & (j < cu[-1])
        dest = torch.where(valid, starts[sq] + (j - cu[sq]), torch.full_like(j, -1))        # aligned row of every packed position; -1: unowned
        seq_off = torch.cat([starts.new_zeros(1), ends, ends.new_full((1,), t_max)])
        seq_off = torch.div(seq_off, C, rounding_mode="floor").to(torch.int32)                 # [nseq + 2]: the sequences + the masked tail
        return self._run_packed_aligned(x, dest.to(torch.int32), t_max, seq_off)

    def _run_packed_aligned(self, x, dest, t_al, seq_off):
        """x [1, total, D] -> the 32-aligned row of t_al positions (dest int32 [total]: the aligned row of every packed position, -1 for
        positions that belong to no sequence), the layers, and back.  Both re-layouts and both of their gradients are row gathers
        (fused.gather_rows: the maps are injective)."""
        total = x.shape[1]
        j = torch.arange(total, device=x.device, dtype=torch.int32)
        src_of = torch.full((t_al + 1,), -1, dtype=torch.int32, device=x.device)       # packed position held by every aligned row (-1: masked)
        src_of = src_of.scatter(0, torch.where(dest >= 0, dest, torch.full_like(dest, t_al)).long(), j)[:t_al].contiguous()
        x_al = fused.gather_rows(x[0], src_of, dest)
        mask = (src_of >= 0).to(x.dtype).unsqueeze(-1)
        out = self._run_layers(x_al.unsqueeze(0), mask.unsqueeze(0), None, seq_off)
        packed = fused.gather_rows(out[0], dest, src_of)
        return ModelOutput(last_hidden_state=packed.unsqueeze(0), past_key_values=None)

    def _forward_packed(self, x, cu_seqlens):
        """Packed variable-length batch (data/utils/spark_dataset.py:111-162,
        train_spark_rwkv7speech.py:238-239; fla's `cu_seqlens`): x is ONE row [1, sum T, D], sequence i occupies
        [cu_seqlens[i], cu_seqlens[i+1]); WKV state and token shift restart at every boundary.  Positions past
        cu_seqlens[-1] (the reference's builder may append one overflowing sample) come back as zeros.

        bf16 on the chunked kernels: the row is re-laid out with every sequence starting on a 32-step chunk boundary and
        followed by at least one masked position (<= 32 wasted positions per sequence instead of padding every sequence to
        the longest).  The masked position makes the token shift of the next sequence start from zero with the ordinary
        mask handling; the chunked WKV7 kernels take the sequences' chunk ranges and give every (sequence, head) its own
        workgroups, each starting from the zero state -- forward and adjoint recurrences of different sequences run in
        parallel.  Everything else is position-wise.  Other dtypes: unpack into a right-padded masked
        batch, run, pack again."""
        assert x.shape[0] == 1, "cu_seqlens expects a packed [1, total, D] row"
        native = (PACKED_NATIVE and x.is_cuda and x.dtype == torch.bfloat16 and fused.CHUNKED_WKV_FWD and fused.CHUNKED_WKV_BWD)
        if native and cu_seqlens.is_cuda:
            return self._forward_packed_device(x, cu_seqlens)
        cu = cu_seqlens.tolist()      # a HOST tensor (what the reference's collators build: spark_dataset.py:150-160): no device sync
        lens = [b - a for a, b in zip(cu[:-1], cu[1:])]
        if native and sum(lens) > 0:
            C = ops.CHUNK_T
            D = x.shape[-1]
            starts, t_al = [], 0
            for n in lens:
                starts.append(t_al)
                if n > 0:
                    t_al += (n // C + 1) * C      # >= n + 1, multiple of 32
            dest = torch.full((x.shape[1],), -1, dtype=torch.int32)
            for s_, n, lo in zip(starts, lens, cu[:-1]):
                if n > 0:
                    dest[lo:lo + n] = torch.arange(s_, s_ + n, dtype=torch.int32)
            seq_chunks = [s_ // C for s_, n in zip(starts, lens) if n > 0] + [t_al // C]
            if _row_align(t_al) > t_al:      # an all-masked pseudo-sequence up to the next multiple of 256 rows (the fused GEMM paths' tile grid)
                t_al = _row_align(t_al)
                seq_chunks.append(t_al // C)
            seq_off = torch.tensor(seq_chunks, dtype=torch.int32)
            return self._run_packed_aligned(x, dest.to(x.device, non_blocking=True), t_al, seq_off.to(x.device, non_blocking=True))
        seqs = list(x[0, :cu[-1]].split(lens))
        xb = torch.nn.utils.rnn.pad_sequence(seqs, batch_first=True)
        mask = torch.zeros(len(lens), xb.shape[1], dtype=torch.long, device=x.device)
        for i, n in enumerate(lens):
            mask[i, :n] = 1
        out = self.forward(inputs_embeds=xb, attention_mask=mask).last_hidden_state
        packed = torch.cat([out[i, :n] for i, n in enumerate(lens)], 0)
        if packed.shape[0] < x.shape[1]:
            packed = torch.cat([packed, packed.new_zeros(x.shape[1] - packed.shape[0], packed.shape[1])], 0)
        return ModelOutput(last_hidden_state=packed.unsqueeze(0), past_key_values=None)


# ------------------------------------------------------------------------------------------------
# initialisation: the reference's own time-mix init (rwkv_s2s_single_ffn.py:74-156) so that random-init
# models have decays/gates in the trained range (used by the synthetic benchmarks)
# ------------------------------------------------------------------------------------------------
@torch.no_grad()
def init_weights(model: nn.Module, cfg: RWKV7Config, seed: int = 0):
    g = torch.Generator().manual_seed(seed)
    D, L, N = cfg.hidden_size, cfg.num_hidden_layers, cfg.head_dim

    def normal_(t, std):
        t.copy_(torch.randn(t.shape, generator=g) * std)

    for mod in model.modules():
        if isinstance(mod, nn.Embedding):
            normal_(mod.weight, 0.02)
        elif isinstance(mod, nn.Linear):
            normal_(mod.weight, 0.02)
            if mod.bias is not None:
                mod.bias.zero_()
        elif isinstance(mod, (nn.LayerNorm, nn.GroupNorm)):
            mod.weight.fill_(1.0)
            if mod.bias is not None:
                mod.bias.zero_()
    blocks = [m for m in model.modules() if isinstance(m, RWKV7Block)]
    for blk in blocks:
        i = blk.layer_idx
        r01 = i / max(L - 1, 1)
        r10 = 1.0 - i / L
        ddd = torch.arange(D, dtype=torch.float32) / D
        lin = torch.arange(D, dtype=torch.float32) / max(D - 1, 1) - 0.5
        n = torch.arange(D) % N
        zig = (n.float() - (N - 1) / 2) / ((N - 1) / 2)
        zig = zig * zig.abs()
        www = -6 + 6 * (torch.arange(D, dtype=torch.float32) / max(D - 1, 1)) ** (1 + r01 ** 0.3)
        at = blk.attn
        for nm, e in (("r", 0.2), ("w", 0.9), ("k", 0.7), ("v", 0.7), ("a", 0.9), ("g", 0.2)):
            getattr(at, f"x_{nm}").copy_((1.0 - torch.pow(ddd, e * r10)).view(1, 1, D))
        at.k_k.copy_(0.71 - lin * 0.1)
        at.k_a.fill_(1.02)
        at.r_k.fill_(-0.04)
        at.w_lora.lora[2].bias.copy_(www + 0.5 + zig * 2.5)
        at.a_lora.lora[2].bias.copy_(-0.19 + zig * 0.3 + lin * 0.4)
        if i != 0:
            at.v_lora.lora[2].bias.copy_(0.73 - lin * 0.4)
        for lo in (at.w_lora, at.a_lora, at.g_lora) + ((at.v_lora,) if i != 0 else ()):
            normal_(lo.lora[0].weight, 0.02)
            normal_(lo.lora[2].weight, 0.1 / math.sqrt(lo.lora[2].weight.shape[1]))
        s = 1.0 / math.sqrt(D)
        normal_(at.r_proj.weight, 0.5 * s)
        normal_(at.k_proj.weight, 0.5 * s)
        normal_(at.v_proj.weight, 0.5 * s)
        normal_(at.o_proj.weight, 0.5 * s)
        blk.ffn.x_k.copy_(1.0 - torch.pow(ddd, r10 ** 4))
        normal_(blk.ffn.key.weight, 0.5 * s)
        normal_(blk.ffn.value.weight, 0.5 / math.sqrt(cfg.intermediate_size))
    return model


class RWKV7ForCausalLM(nn.Module):
    """rwkvfla's RWKV7ForCausalLM as the reference uses it (model/llm/llm.py:44-50, after
    train_functions.alter_emb_and_head): backbone + lm_head, logits for every position, optional shifted-label CE."""

    def __init__(self, cfg: RWKV7Config, head_size: Optional[int] = None, head_bias: bool = False):
        super().__init__()
        self.config = cfg
        self.model = RWKV7Model(cfg)
        self.lm_head = nn.Linear(cfg.hidden_size, head_size or cfg.vocab_size, bias=head_bias)

    def get_input_embeddings(self):
        return self.model.embeddings

    @property
    def device(self):
        return self.lm_head.weight.device

    @property
    def dtype(self):
        return self.lm_head.weight.dtype

    def forward(self, input_ids=None, attention_mask=None, inputs_embeds=None, past_key_values=None, labels=None,
                use_cache=None, **kwargs):
        out = self.model(input_ids=input_ids, attention_mask=attention_mask, inputs_embeds=inputs_embeds,
                         past_key_values=past_key_values, use_cache=use_cache)
        logits = self.lm_head(out[0])
        loss = None
        if labels is not None:
            lab = torch.cat((labels[..., 1:], torch.full_like(labels[:, :1], -100)), 1)
            loss = F.cross_entropy(logits.view(lab.numel(), -1).float(), lab.view(-1), ignore_index=-100)
        return ModelOutput(loss=loss, logits=logits, past_key_values=out.past_key_values)
