"""Per-phase durations of the decode step (csrc/decode_step.hip) in its one-launch-per-phase mode.

    cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d $OUT -o dec -- python tools/decode_phase_profile.py run
    python tools/decode_phase_profile.py report $OUT
"""
import csv, glob, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

STEPS, WARM = 6, 3
NAMES = ["P0 row ln1+mix6", "P1 gemv rkv+lora-down", "P2 head", "P3 gemv o_proj", "P4 row ln2+mix1", "P5 gemv key", "P6 gemv value"]


def run():
    import torch
    from rwkvtts_amd import backbone
    from rwkvtts_amd.backbone import Cache, RWKV7ForCausalLM
    from rwkvtts_amd.decode import DecodeStep
    cfg = (backbone.config_1p5b if os.environ.get("GEN_MODEL") == "1.5b" else backbone.config_0p4b)(vocab_size=8193)
    m = RWKV7ForCausalLM(cfg)
    backbone.init_weights(m, cfg, seed=0)
    m = m.to("cuda:0", torch.bfloat16).eval()
    B = 32
    cache = Cache.zeros(cfg, B, "cuda:0", torch.bfloat16)
    step = DecodeStep(m.model, m.lm_head, cache, persistent=False)
    x = (torch.randn(B, cfg.hidden_size, device="cuda:0") * 0.5).to(torch.bfloat16)
    for _ in range(WARM + STEPS):
        step(x)
    torch.cuda.synchronize()
    pstep = DecodeStep(m.model, m.lm_head, cache, persistent=True)
    for _ in range(WARM + STEPS):
        pstep(x)
    torch.cuda.synchronize()
    print("flag", pstep.barrier_timed_out())
    for mode in (2, 3, 4, 5):
        bstep = DecodeStep(m.model, m.lm_head, cache, persistent=mode)
        for _ in range(WARM + STEPS):
            bstep(x)
        torch.cuda.synchronize()


def report(d):
    f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    ph = sorted([r for r in rows if "decode_phase_kernel" in r["Kernel_Name"]], key=lambda r: int(r["Start_Timestamp"]))
    per = len(ph) // (WARM + STEPS)
    L = (per - 2) // 7
    ph = ph[WARM * per:]
    dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in ph]
    gap = [int(ph[i + 1]["Start_Timestamp"]) - int(ph[i]["End_Timestamp"]) for i in range(len(ph) - 1)]
    print(f"{per} launches per step, {L} layers; mean gap between launches {sum(gap) / len(gap) / 1e3:.2f} us")
    tot = 0.0
    for p in range(7):
        v = [dur[s * per + l * 7 + p] for s in range(STEPS) for l in range(L)]
        tot += sum(v) / len(v) * L
        print(f"  {NAMES[p]:24s} {sum(v) / len(v) / 1e3:7.2f} us  (min {min(v) / 1e3:.2f})")
    for p, n in ((0, "final row"), (1, "head gemv")):
        v = [dur[s * per + L * 7 + p] for s in range(STEPS)]
        tot += sum(v) / len(v)
        print(f"  {n:24s} {sum(v) / len(v) / 1e3:7.2f} us")
    print(f"  sum of kernel durations per step: {tot / 1e3:.1f} us")
    pk = sorted([r for r in rows if "decode_persistent_kernel" in r["Kernel_Name"]], key=lambda r: int(r["Start_Timestamp"]))
    n = WARM + STEPS
    for name, part in (("persistent kernel", pk[WARM:n]), ("barriers alone", pk[n + WARM:2 * n]), ("barriers, no fences", pk[2 * n + WARM:3 * n]),
                       ("barriers, release only", pk[3 * n + WARM:4 * n]), ("barriers, acquire only", pk[4 * n + WARM:5 * n])):
        if part:
            v = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in part]
            print(f"  {name}: {sum(v) / len(v) / 1e3:.1f} us per step ({sum(v) / len(v) / 1e3 / (per - 1):.2f} us per barrier interval)")


if __name__ == "__main__":
    run() if sys.argv[1] == "run" else report(sys.argv[2])
