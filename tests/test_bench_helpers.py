"""CPU: the pieces of bench.py that only matter when something goes wrong on a multi-GPU node, or that define a workload."""
import importlib.util
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_packed_lengths_fill_exactly_the_budget_after_alignment():
    b = _bench()
    for budget in (32768, 16384, 65536):
        lens = b.packed_lengths(budget)
        n = len(lens)
        assert all(x > 32 and x % 32 != 0 for x in lens)
        assert sum(lens) == budget - 32 * n                               # what `value` counts: the real tokens
        assert sum((x // 32 + 1) * 32 for x in lens) <= budget            # every sequence + >= 1 masked position, 32-aligned, fits
        assert (sum(lens) + 32 * n + 255) // 256 * 256 == budget          # the device path's shape-only bound is the budget itself


def test_watchdog_reports_the_stuck_stage_and_exits_3():
    code = textwrap.dedent(f"""
        import sys, time
        sys.path.insert(0, {ROOT!r})
        import importlib.util
        spec = importlib.util.spec_from_file_location("bench_mod", {os.path.join(ROOT, 'bench.py')!r})
        b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
        with b.Watchdog(0.3, "preflight: first all-reduce", lambda: "ranks not past 'map': [1]"):
            time.sleep(5)
        print("NOT REACHED")
    """)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60, env=dict(os.environ, RANK="1", LOCAL_RANK="1"))
    assert r.returncode == 3, (r.returncode, r.stderr[-400:])
    assert "rank 1" in r.stderr and "preflight: first all-reduce" in r.stderr and "ranks not past 'map': [1]" in r.stderr
    assert "NOT REACHED" not in r.stdout


def test_watchdog_is_silent_when_the_stage_finishes():
    b = _bench()
    with b.Watchdog(5.0, "quick stage"):
        pass


def test_committed_pmc_record_matches_the_shipped_kernel_sources():
    """profiles/pmc_wkv7.json (the `traffic` / `mfma_util` of bench.py's roofline object) is pinned to the kernel sources it was
    collected on; a source edit without new PMC passes (tools/pmc_wkv.sh + tools/pmc_distill.py) would make the driver's line report
    `pmc_stale` and a null `traffic` -- caught here instead."""
    import hashlib
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rec = json.load(open(os.path.join(root, "profiles", "pmc_wkv7.json")))
    for key in ("wkv7c_fwd", "wkv7c_bwd"):
        for name, h in rec[key]["sources"].items():
            got = hashlib.sha256(open(os.path.join(root, "rwkvtts_amd", "csrc", name), "rb").read()).hexdigest()[:len(h)]
            assert got == h, f"{name} changed since the PMC passes of {rec[key]['source']}: re-run tools/pmc_wkv.sh and tools/pmc_distill.py"


def test_lora_down_direct_predicate_mirrors_the_kernels_limits():
    """fused.lora_down_direct_supported must refuse exactly what rwkv7_lora_down_fwd_bf16 refuses (csrc/lora_down.hip: lora_down_cut and
    the LDS budget of launch_lora_down) -- a shape it lets through that the C entry rejects would raise RWKV7_ESHAPE mid-training."""
    import torch
    from rwkvtts_amd import fused

    def sup(B, T, D, ranks):
        x = torch.empty(B, T, D, dtype=torch.bfloat16, device="meta")
        return fused.lora_down_direct_supported(x, [torch.empty(r, D, dtype=torch.bfloat16, device="meta") for r in ranks])

    assert fused.LORA_DOWN_DIRECT
    assert sup(8, 4096, 1024, (64, 64, 32, 128))          # 0.4B
    assert sup(8, 4096, 1024, (64, 64, 128))              # layer 0
    assert sup(2, 2048, 768, (64, 64, 32, 128))           # 0.1B
    assert sup(2, 2048, 2560, (64, 64, 32, 128))
    assert not sup(4, 8192, 2048, (96, 96, 64, 256))      # 1.5B: 16 column tiles
    assert not sup(2, 1000, 1024, (64, 64, 32, 128))      # rows not a multiple of 128
    assert not sup(2, 2048, 1024, (64, 64, 40, 128))      # a rank that is not a multiple of 32
    assert not sup(2, 2048, 4096, (64, 64, 64, 128))      # 10 tiles at D = 4096: 170 KB of LDS
    assert sup(2, 2048, 1024, (32, 32, 32, 32))           # four one-tile branches: two per half
    assert sup(2, 2048, 1024, (32, 32, 32))               # three one-tile branches: 2 + 1
    assert not sup(2, 2048, 1024, (32,))                  # a single tile cannot be cut in two
