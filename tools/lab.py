"""Loader for the LAB library (rwkvtts_amd/lib/librwkv7_hip_lab.so, `python -m rwkvtts_amd.build --lab`, include/rwkv7_hip_lab.h):
the superseded kernels kept as A/B twins of the shipped ones, each under its own entry point.  Used by tools/ab_*.py and by the
lab cases of tests/ (skipped when the lab library has not been built).  Nothing under rwkvtts_amd/ imports this."""
import ctypes
import os

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LAB_SO = os.path.join(ROOT, "rwkvtts_amd", "lib", "librwkv7_hip_lab.so")
_lab = None


def available() -> bool:
    return os.path.exists(LAB_SO)


def lib():
    global _lab
    if _lab is None:
        if not available():
            raise FileNotFoundError(f"{LAB_SO} is missing: python -m rwkvtts_amd.build --lab")
        _lab = ctypes.CDLL(LAB_SO)
    return _lab


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _st(t):
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def chunk_forward4(w, q, k, v, a, b, tinv, seq_off=None, save=True):
    """The 4-wave chunked forward on bf16 tensors (the kernel fp32 tensors run in the shipped library)."""
    from rwkvtts_amd import ops
    B, T, H, C = w.shape
    y = torch.empty_like(v)
    sa = torch.empty(B, T, H, C, dtype=torch.float32, device=w.device) if save else None
    hs = torch.empty(B, H, T // ops.CHUNK_T, ops.Q15_REC, dtype=torch.int16, device=w.device) if save else None
    so, ns = (None, 0) if seq_off is None else (_p(seq_off), seq_off.numel() - 1)
    rc = lib().rwkv7_lab_wkv_chunk_fwd4_seq_bf16(B, T, H, _p(w), _p(q), _p(k), _p(v), _p(a), _p(b), _p(tinv), _p(y), _p(sa), _p(hs),
                                                 so, ns, _st(w))
    assert rc == 0, rc
    return (y, tinv, sa, hs) if save else y


def bwd_out9(w, q, k, v, a, b, dy, hs, sa, z, e_vk, grads=None):
    """The round-3/4 per-chunk gradient kernel; same arguments / results as rwkv7_wkv_chunk_bwd_out_z_bf16."""
    B, T, H, C = w.shape
    grads = grads or [torch.empty_like(w) for _ in range(6)]
    rc = lib().rwkv7_lab_wkv_chunk_bwd_out9_z_bf16(B, T, H, _p(w), _p(q), _p(k), _p(v), _p(a), _p(b), _p(dy), _p(hs), _p(sa), _p(z),
                                                   _p(e_vk), *[_p(g) for g in grads], _st(w))
    assert rc == 0, rc
    return tuple(grads)


def gemm_nt_gen1(A, W, epilogue=0, variant=0):
    M, K = A.shape
    N = W.shape[0]
    C = torch.full((M, N), float("nan"), device=A.device, dtype=torch.bfloat16)
    rc = lib().rwkv7_lab_gemm_nt_gen1_bf16(M, N, K, _p(A), _p(W), _p(C), int(epilogue), int(variant), _st(A))
    assert rc == 0, rc
    return C


def gemm_nt_relusq_bwd_gen1(A, W, aux):
    M, K = A.shape
    N = W.shape[0]
    C = torch.full((M, N), float("nan"), device=A.device, dtype=torch.bfloat16)
    rc = lib().rwkv7_lab_gemm_nt_relusq_bwd_gen1_bf16(M, N, K, _p(A), _p(W), _p(aux), _p(C), _st(A))
    assert rc == 0, rc
    return C
