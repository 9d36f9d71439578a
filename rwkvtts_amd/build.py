"""Ahead-of-time build of librwkv7_hip.so for gfx950 (no import-time JIT, unlike the reference's
torch.utils.cpp_extension.load at model/llm/rwkv_s2s_single_ffn.py:13).

hipcc cross-compiles without a GPU; the .so is written in-tree (rwkvtts_amd/lib/) so that it travels
with the repo snapshot to the GPU box and is visible to the driver's "which .so was loaded" audit.

    python -m rwkvtts_amd.build [--force] [--verbose]
"""
import argparse
import glob
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
SO = os.path.join(LIBDIR, "librwkv7_hip.so")
LAB_SO = os.path.join(LIBDIR, "librwkv7_hip_lab.so")   # --lab: the shipped entries + the superseded A/B twins (include/rwkv7_hip_lab.h)
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-ffast-math", "-fno-finite-math-only", "-fgpu-flush-denormals-to-zero",
         "-Wno-unused-result", "-Wno-pass-failed"]


def source_hashes(names):
    """sha256 (first 16 hex digits) of kernel source files under csrc/ -- how profiles/pmc_wkv7.json records what its counters were
    collected on and how bench.py detects that they no longer describe the shipped kernels (no .git on the GPU box)."""
    import hashlib
    out = {}
    for n in names:
        try:
            out[n] = hashlib.sha256(open(os.path.join(CSRC, n), "rb").read()).hexdigest()[:16]
        except OSError:
            out[n] = None
    return out


def sources(lab: bool = False):
    out = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    if lab:
        out += sorted(glob.glob(os.path.join(CSRC, "lab", "*.hip")))
    return out


def _stale():
    if not os.path.exists(SO):
        return True
    mt = os.path.getmtime(SO)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + \
        glob.glob(os.path.join(HERE, "..", "include", "*.h"))
    return any(os.path.getmtime(p) > mt for p in deps)


def build_lab(verbose: bool = False) -> str:
    """The lab library: every source compiled with -DRWKV7_LAB into its own object directory, linked into LAB_SO.  Never touches SO."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    objdir = os.path.join(LIBDIR, "lab")
    os.makedirs(objdir, exist_ok=True)
    hdrs = glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(HERE, "..", "include", "*.h"))
    objs = []
    for src in sources(lab=True):
        obj = os.path.join(objdir, os.path.basename(src).replace(".hip", ".o"))
        if not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(p) for p in [src] + hdrs):
            cmd = [hipcc, f"--offload-arch={ARCH}", *FLAGS, "-DRWKV7_LAB", "-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
        objs.append(obj)
    subprocess.check_call([hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LAB_SO, *objs])
    return LAB_SO


def build(force: bool = False, verbose: bool = False, extra=()) -> str:
    # objects compiled with other flags (e.g. a --timing build that was interrupted) must not be linked into a normal build
    stamp, flags = os.path.join(LIBDIR, ".flags"), " ".join(FLAGS + list(extra))
    if os.path.exists(stamp) and open(stamp).read() != flags:
        force = True
    elif not os.path.exists(stamp) and extra:
        force = True
    if not force and not _stale():
        return SO
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: librwkv7_hip.so cannot be built (and there is no fallback)")
    os.makedirs(LIBDIR, exist_ok=True)
    with open(stamp, "w") as f:
        f.write("INCOMPLETE")   # replaced by the flags after a complete build: an interrupted build forces the next one
    objs = []
    for src in sources():
        obj = os.path.join(LIBDIR, os.path.basename(src).replace(".hip", ".o"))
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(
                os.path.getmtime(p) for p in [src] + glob.glob(os.path.join(CSRC, "*.h"))):
            cmd = [hipcc, f"--offload-arch={ARCH}", *FLAGS, *extra, "-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
        objs.append(obj)
    cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", SO, *objs]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    with open(stamp, "w") as f:
        f.write(flags)
    return SO


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--verbose", action="store_true")
    ap.add_argument("--resource-usage", action="store_true", help="print VGPR/LDS/occupancy per kernel")
    ap.add_argument("--timing", action="store_true", help="profiling build: per-phase s_memtime counters in the chunked kernels")
    ap.add_argument("--lab", action="store_true", help="also build lib/librwkv7_hip_lab.so (superseded A/B twins, rwkv7_hip_lab.h)")
    a = ap.parse_args()
    extra = ["-Rpass-analysis=kernel-resource-usage"] if a.resource_usage else []
    if a.timing:
        extra.append("-DWKV7C_TIMING")
    print(build(force=a.force or a.resource_usage or a.timing, verbose=a.verbose, extra=extra))
    if a.lab:
        print(build_lab(verbose=a.verbose))
    sys.exit(0)
