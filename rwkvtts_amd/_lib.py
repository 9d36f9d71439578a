"""ctypes loader for librwkv7_hip.so (C ABI: include/rwkv7_hip.h).

There is deliberately NO fallback: if the library is missing or a call fails, the caller gets an
exception.  Nothing in this package routes through oracle/ or through a CPU/eager re-implementation.
"""
import ctypes
import os

# torch must be imported BEFORE librwkv7_hip.so is dlopen'ed: the .so needs libamdhip64.so.7, and the process
# must end up with exactly one HIP runtime -- the one PyTorch-ROCm bundles (torch/lib/libamdhip64.so, same
# SONAME).  Loaded the other way round, /opt/rocm's runtime gets in first, torch's libraries bind to it, and
# launches fail with hipErrorNoDevice (seen on the GPU box).
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("RWKV7_HIP_SO") or os.path.join(_HERE, "lib", "librwkv7_hip.so")   # override: A/B builds

_ERR = {-1: "RWKV7_EINVAL: null pointer or non-positive size",
        -2: "RWKV7_ECHUNK: T must be a multiple of 16 (reference assert, wkv7_cuda.cu:136)",
        -3: "RWKV7_EHEAD: H*64 != C (reference assert, rwkv7_state_fwd_fp16.cu:61)",
        -4: "RWKV7_ESHAPE: unsupported size for a fused elementwise op"}

_lib = None


class Rwkv7HipError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise Rwkv7HipError(
                f"{SO_PATH} is missing: build it with `python -m rwkvtts_amd.build` "
                "(hipcc --offload-arch=gfx950).  There is no CPU/eager fallback for the HIP ops.")
        _lib = ctypes.CDLL(SO_PATH)
        _lib.rwkv7_version.restype = ctypes.c_char_p
    return _lib


def version() -> str:
    return lib().rwkv7_version().decode()


def check(rc: int, what: str):
    if rc == 0:
        return
    if rc < 0:
        raise ValueError(f"{what}: {_ERR.get(rc, f'error {rc}')}")
    raise Rwkv7HipError(f"{what}: HIP error {rc} at launch")


def exported_symbols():
    """Names declared in include/rwkv7_hip.h (parsed, so the header stays the single source of truth)."""
    import re
    hdr = os.path.join(_HERE, "..", "include", "rwkv7_hip.h")
    txt = open(hdr).read()
    return sorted(set(re.findall(r"\b(rwkv7_[a-z0-9_]+)\s*\(", txt)))
