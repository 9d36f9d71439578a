"""Why does every decode phase run 25-40 % above its own best case?    The replayed hipGraph of the greedy
decode step (GraphDecoder, 0.4B, B = 32), N steps under `rocprofv3 --kernel-trace`; per phase: min / p10 / median / p90 / mean of the
kernel durations, the same by layer index and by step index, and against the idle gap in front of the launch.

    cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d $OUT -o dec -- python tools/decode_gap_trace.py run
    python tools/decode_gap_trace.py report $OUT"""
import csv, glob, os, re, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
GEN = 260
NAMES = {0: "P0 row ln1+mix6", 1: "P1 gemv rkv+lora", 2: "P2 head", 3: "P3 gemv o_proj", 4: "P4 row ln2+mix1", 5: "P5 gemv key", 6: "P6 gemv value",
         7: "final row", 8: "head gemv"}


def run():
    import torch
    from rwkvtts_amd import backbone
    from rwkvtts_amd.spark_llm import RWKV7ForSpeech, RWKV7SpeechConfig
    from rwkvtts_amd.decode import GraphDecoder
    dev = torch.device("cuda:0")
    base = backbone.config_0p4b()
    cfg = RWKV7SpeechConfig(**{k: v for k, v in base.to_dict().items() if k in backbone.RWKV7Config.__dataclass_fields__ and k != "extra"})
    model = RWKV7ForSpeech(cfg).init_weights(0).to(dev, torch.bfloat16).eval()
    B, P = 32, 128
    g = torch.Generator().manual_seed(1234)
    emb = (torch.randn(B, P, cfg.hidden_size, generator=g) * 0.5).to(dev, torch.bfloat16)
    mask = torch.ones(B, P, dtype=torch.long, device=dev)
    dec = GraphDecoder(model, B, step_kernel=True)
    dec.generate(inputs_embeds=emb, attention_mask=mask, max_new_tokens=8, suppress_tokens=[8192])
    torch.cuda.synchronize()
    dec.generate(inputs_embeds=emb, attention_mask=mask, max_new_tokens=GEN, suppress_tokens=[8192])
    torch.cuda.synchronize()


def pct(v, q):
    v = sorted(v)
    return v[min(len(v) - 1, int(q * len(v)))]


def report(d):
    f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows]
    # the last GEN - 1 replays: walk back from the end; a step = the launches between two "head gemv" kernels (phase id 8)
    ph = lambda n: int(re.search(r"decode_phase_kernel<(\d+)", n).group(1)) if "decode_phase_kernel<" in n else None
    idx = [i for i, k in enumerate(ks) if ph(k[2]) == 8]
    idx = idx[-(GEN - 2):]
    per = collections.defaultdict(list)   # phase -> [(dur_us, gap_before_us, layer, step)]
    for si in range(len(idx) - 1):
        seg = ks[idx[si] + 1: idx[si + 1] + 1]
        layer, seen0 = -1, 0
        for j, (s, e, n) in enumerate(seg):
            p = ph(n)
            if p is None:
                continue
            if p == 0:
                layer += 1
            prev_end = ks[idx[si] + j][1]
            per[p].append(((e - s) / 1e3, (s - prev_end) / 1e3, layer, si))
    tot_mean = tot_min = tot_med = 0.0
    print(f"{len(idx) - 1} replayed steps; per phase (us): min / p10 / median / p90 / mean | mean idle gap in front | per step count")
    for p in sorted(per):
        v = [x[0] for x in per[p]]
        gaps = [x[1] for x in per[p]]
        cnt = len(v) / (len(idx) - 1)
        tot_mean += sum(v) / len(v) * cnt; tot_min += min(v) * cnt; tot_med += pct(v, 0.5) * cnt
        print(f"  {NAMES.get(p, p):20s} {min(v):6.2f} {pct(v, 0.1):6.2f} {pct(v, 0.5):6.2f} {pct(v, 0.9):6.2f} {sum(v) / len(v):6.2f} | gap {sum(gaps) / len(gaps):5.2f} | x{cnt:.0f}")
    span = (ks[idx[-1]][1] - ks[idx[0]][1]) / 1e3 / (len(idx) - 1)
    print(f"  per step: wall {span:.1f} us; sum of means {tot_mean:.1f}, of medians {tot_med:.1f}, of minima {tot_min:.1f}")
    # by layer, by step decile, by gap
    for p in (1, 2, 5):
        v = per[p]
        by_layer = collections.defaultdict(list)
        for d_, g_, l_, s_ in v:
            by_layer[l_].append(d_)
        print(f"  {NAMES[p]} median by layer:", " ".join(f"{pct(by_layer[l], 0.5):.1f}" for l in sorted(by_layer)))
        ns = len(idx) - 1
        print(f"  {NAMES[p]} median by step decile:", " ".join(f"{pct([d_ for d_, g_, l_, s_ in v if s_ * 10 // ns == q], 0.5):.2f}" for q in range(10)))
        lo = [d_ for d_, g_, l_, s_ in v if g_ < 0.5]
        hi = [d_ for d_, g_, l_, s_ in v if g_ >= 0.5]
        if lo and hi:
            print(f"  {NAMES[p]} median with gap < 0.5 us in front: {pct(lo, 0.5):.2f} ({len(lo)}), gap >= 0.5 us: {pct(hi, 0.5):.2f} ({len(hi)})")
    # does the min come from a particular place?
    for p in (2, 5):
        best = sorted(per[p])[:8]
        print(f"  {NAMES[p]} eight fastest: " + ", ".join(f"{d_:.2f}us@L{l_}/s{s_}" for d_, g_, l_, s_ in best))


if __name__ == "__main__":
    run() if sys.argv[1] == "run" else report(sys.argv[2])
