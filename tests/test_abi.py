"""CPU: the C-ABI library builds for gfx950, loads without a GPU, exports every symbol that
include/rwkv7_hip.h declares, and rejects bad arguments before launching anything."""
import ctypes
import os
import subprocess

import pytest
import torch

from rwkvtts_amd import _lib, build


def test_library_builds_and_loads(hip_lib):
    assert os.path.exists(build.SO)
    v = _lib.version()
    assert v.startswith("rwkv7_hip") and "gfx950" in v


def test_every_declared_symbol_is_exported(hip_lib):
    names = _lib.exported_symbols()
    assert "rwkv7_wkv_fwd_bf16" in names and "rwkv7_wkv_bwd_bf16" in names and "rwkv7_wkv_state_fwd_bf16" in names
    for n in names:
        assert hasattr(hip_lib, n), f"{n} declared in include/rwkv7_hip.h but not exported"


def test_code_object_is_gfx950(hip_lib):
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", build.SO], capture_output=True, text=True)
    if out.returncode != 0:
        pytest.skip("llvm-readelf unavailable")
    blob = open(build.SO, "rb").read()
    assert b"gfx950" in blob


def test_argument_errors_do_not_launch(hip_lib):
    one = ctypes.c_void_p(16)  # never dereferenced: the checks fire first
    # T % 16 != 0 -> RWKV7_ECHUNK (reference: assert at wkv7_cuda.cu:136)
    assert hip_lib.rwkv7_wkv_fwd_bf16(1, 15, 1, one, one, one, one, one, one, one, one, one, None) == -2
    assert hip_lib.rwkv7_wkv_bwd_bf16(1, 17, 1, *([one] * 15), None) == -2
    # null pointer -> RWKV7_EINVAL
    assert hip_lib.rwkv7_wkv_fwd_f32(1, 16, 1, None, one, one, one, one, one, one, None, None, None) == -1
    # s without sa -> RWKV7_EINVAL
    assert hip_lib.rwkv7_wkv_fwd_f32(1, 16, 1, one, one, one, one, one, one, one, one, None, None) == -1
    # H*64 != C -> RWKV7_EHEAD (reference: assert at rwkv7_state_fwd_fp16.cu:61)
    assert hip_lib.rwkv7_wkv_state_fwd_bf16(1, 1, 100, 2, one, one, one, one, one, one, one, one, None) == -3
    assert hip_lib.rwkv7_wkv_state_fwd_bf16(0, 1, 128, 2, one, one, one, one, one, one, one, one, None) == -1
    # the row-split backward and the fused stages validate the same way
    two = (ctypes.c_void_p * 2)(16, 16)
    assert hip_lib.rwkv7_wkv_bwd_split_bf16(1, 24, 1, *([one] * 9), two, two, two, one, two, two, None) == -2
    bad = (ctypes.c_void_p * 2)(16, None)
    assert hip_lib.rwkv7_wkv_bwd_split_bf16(1, 16, 1, *([one] * 9), two, bad, two, one, two, two, None) == -1
    assert hip_lib.rwkv7_add_ln_fwd_bf16(ctypes.c_long(4), 100, one, None, one, None, ctypes.c_float(1e-5), None, one, one,
                                         one, 4, None) == -4          # D % 64 != 0 -> RWKV7_ESHAPE


def test_workspace_query(hip_lib):
    s, sa = ctypes.c_size_t(), ctypes.c_size_t()
    assert hip_lib.rwkv7_wkv_workspace_bytes(8, 4096, 16, ctypes.byref(s), ctypes.byref(sa)) == 0
    assert s.value == 8 * 16 * 256 * 64 * 64 * 4 and sa.value == 8 * 4096 * 16 * 64 * 4   # 512 MiB / 128 MiB, SURVEY 8(a) a1
    assert hip_lib.rwkv7_wkv_workspace_bytes(8, 4097, 16, ctypes.byref(s), ctypes.byref(sa)) == -2


def test_reference_op_namespaces_exist_and_refuse_cpu():
    from rwkvtts_amd import ops
    for ns, name in (("wind_backstepping", "forward"), ("wind_backstepping", "backward"),
                     ("rwkv7_state_fwd_fp16", "forward"), ("wkv7s", "forward")):
        assert hasattr(getattr(torch.ops, ns), name)
    x = torch.zeros(1, 16, 1, 64, dtype=torch.bfloat16)
    with pytest.raises(NotImplementedError):  # CUDA(HIP)-only dispatch, as in wkv7_op.cpp:26-29
        ops.WindBackstepping.apply(x, x, x, x, x, x)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "SO_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.Rwkv7HipError):
        _lib.lib()
