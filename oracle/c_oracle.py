"""ctypes binding of oracle/libwkv7_oracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The C file restates model/llm/cuda/wkv7_cuda.cu:10-130 and rwkv7_state_fwd_fp16.cu:9-57
(see oracle/wkv7_oracle.c for line-by-line citations).
"""
import ctypes
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libwkv7_oracle.so")
_lib = None

CHUNK_LEN = 16
HEAD_SIZE = 64


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "wkv7_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "libwkv7_oracle.so"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
        for name in ("wkv7_fwd_bf16", "wkv7_fwd_f32", "wkv7_bwd_bf16", "wkv7_bwd_f32",
                     "wkv7_state_fwd_bf16", "wkv7_state_fwd_f32"):
            getattr(_lib, name).restype = ctypes.c_int
    return _lib


def _p(t):
    if t is None:
        return ctypes.c_void_p(0)
    assert t.device.type == "cpu" and t.is_contiguous()
    return ctypes.c_void_p(t.data_ptr())


def _sfx(t):
    if t.dtype == torch.bfloat16:
        return "bf16"
    if t.dtype == torch.float32:
        return "f32"
    raise TypeError(t.dtype)


def wkv7_fwd(w, q, k, v, a, b, save=True):
    """wind_backstepping.forward (wkv7_op.cpp:21-22).  All inputs [B,T,H,64], bf16 or fp32.
    Returns y, s [B,H,T/16,64,64] fp32 (transposed per wkv7_cuda.cu:45-48), sa [B,T,H,64] fp32."""
    B, T, H, C = w.shape
    assert C == HEAD_SIZE
    y = torch.empty_like(v)
    s = torch.empty(B, H, T // CHUNK_LEN, C, C, dtype=torch.float32) if save else None
    sa = torch.empty(B, T, H, C, dtype=torch.float32) if save else None
    rc = getattr(lib(), "wkv7_fwd_" + _sfx(w))(B, T, H, _p(w), _p(q), _p(k), _p(v), _p(a), _p(b),
                                                _p(y), _p(s), _p(sa))
    if rc != 0:
        raise ValueError(f"wkv7_fwd oracle rc={rc} (T must be a multiple of {CHUNK_LEN})")
    return y, s, sa


def wkv7_bwd(w, q, k, v, a, b, dy, s, sa):
    """wind_backstepping.backward (wkv7_op.cpp:23-24).  Returns dw,dq,dk,dv,da,db like the inputs."""
    B, T, H, C = w.shape
    outs = [torch.empty_like(w) for _ in range(6)]
    rc = getattr(lib(), "wkv7_bwd_" + _sfx(w))(B, T, H, _p(w), _p(q), _p(k), _p(v), _p(a), _p(b),
                                                _p(dy), _p(s), _p(sa), *[_p(o) for o in outs])
    if rc != 0:
        raise ValueError(f"wkv7_bwd oracle rc={rc}")
    return outs


def wkv7_state_fwd(state, r, w, k, v, a, b):
    """rwkv7_state_fwd_fp16.forward (rwkv7_state_fwd_fp16.cpp:8-14): state fp32 [B,H,64,64]
    updated IN PLACE, r..b are [B,T,C]; returns y [B,T,C]."""
    B, T, C = r.shape
    H = C // HEAD_SIZE
    assert state.dtype == torch.float32 and tuple(state.shape) == (B, H, HEAD_SIZE, HEAD_SIZE)
    y = torch.empty_like(r)
    rc = getattr(lib(), "wkv7_state_fwd_" + _sfx(r))(B, T, C, H, _p(state), _p(r), _p(w), _p(k),
                                                      _p(v), _p(a), _p(b), _p(y))
    if rc != 0:
        raise ValueError(f"wkv7_state_fwd oracle rc={rc}")
    return y
