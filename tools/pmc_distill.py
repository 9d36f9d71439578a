"""summary.txt of tools/pmc_wkv.sh (mean counter values per kernel, separate rocprofv3 --pmc passes) -> profiles/pmc_wkv7.json,
the file bench.py reads `roofline.traffic`, `roofline.mfma_util` and `roofline.valu_frac` from.

    python tools/pmc_distill.py <summary.txt> <label of the source file under profiles/> [B T H]

Units (MI355X_MICROARCH.md): FETCH_SIZE / WRITE_SIZE in KiB, FETCH_SIZE counts half of a wide coalesced read stream on gfx950
(HBM bytes = (2 * FETCH + WRITE) * 1024); GRBM_GUI_ACTIVE is summed over the 8 XCDs; SQ_VALU_MFMA_BUSY_CYCLES counts cycles
summed over the 1024 SIMDs; SQ_ACTIVE_INST_VALU counts quad-cycles.
    mfma_util = MFMA_BUSY / (1024 * GUI_ACTIVE / 8)         valu_frac = 4 * ACTIVE_INST_VALU / (1024 * GUI_ACTIVE / 8)
"""
import collections
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rwkvtts_amd.build import source_hashes  # noqa: E402

COMMON = ["chunk_common.h", "wkv7_common.h"]


def parse(path):
    out, cur = collections.OrderedDict(), None
    for line in open(path):
        if not line.startswith(" "):
            cur = line.strip()
            out[cur] = {}
        else:
            m = re.match(r"\s+(\S+)\s+n=\s*\d+\s+mean=(\S+)", line)
            if m and cur:
                out[cur][m.group(1)] = float(m.group(2))
    return out


def find(tab, needle):
    hits = [k for k in tab if needle in k]
    if not hits:
        raise SystemExit(f"{needle} not in the summary")
    return tab[hits[0]], hits[0]


def group(tab, needles, B, T, H, source, files):
    parts, fetch, write, busy, valu, cyc = {}, 0.0, 0.0, 0.0, 0.0, 0.0
    for n in needles:
        c, name = find(tab, n)
        parts[name.replace("void ", "").replace("rwkv7::", "")] = [c["FETCH_SIZE"], c["WRITE_SIZE"]]
        fetch += c["FETCH_SIZE"]
        write += c["WRITE_SIZE"]
        busy += c["SQ_VALU_MFMA_BUSY_CYCLES"]
        valu += 4.0 * c["SQ_ACTIVE_INST_VALU"]
        cyc += 1024.0 * c["GRBM_GUI_ACTIVE"] / 8.0
    return {"kernel": " + ".join(parts), "B": B, "T": T, "H": H, "dtype": "bf16", "FETCH_SIZE_KiB": fetch, "WRITE_SIZE_KiB": write,
            "mfma_util": round(busy / cyc, 4), "valu_frac": round(valu / cyc, 4), "parts": parts, "source": source,
            # sha256 of the kernel sources the counters were collected on: bench.py reports pmc_stale (and nulls traffic / mfma_util /
            # valu_frac) when the shipped sources differ
            "sources": source_hashes(files)}


def main():
    summary, label = sys.argv[1], sys.argv[2]
    B, T, H = (int(x) for x in sys.argv[3:6]) if len(sys.argv) >= 6 else (8, 4096, 16)
    tab = parse(summary)
    src = f"profiles/{label} (tools/pmc_wkv.sh, separate --pmc passes; distilled by tools/pmc_distill.py)"
    out = {"_comment": "HBM traffic and SIMD utilisation of the WKV7 kernels from rocprofv3 --pmc passes (tools/pmc_wkv.sh: separate "
                       "runs per counter group, no tracing). KiB per launch; HBM bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950: "
                       "FETCH_SIZE counts half of a wide coalesced read stream, MI355X_MICROARCH.md). wkv7c_* = what bf16 training "
                       "launches; wkv7_* = the scalar reference-schema kernels.",
           "wkv7c_bwd": group(tab, ["wkv7c_bseq_kernel", "wkv7c_bwd_out10_kernel"], B, T, H, src,
                              ["wkv7_chunk_bseq.hip", "wkv7_chunk_bwd10.hip", "chunk_bwd_common.h"] + COMMON),
           "wkv7c_fwd": group(tab, ["wkv7c_prep_kernel", "wkv7c_fwd9_kernel"], B, T, H, src,
                              ["wkv7_chunk_fwd.hip", "wkv7_chunk_fwd9.hip"] + COMMON),
           "wkv7_bwd": group(tab, ["wkv7_bwd_kernel<rwkv7::bf16_t, 2, 4>"], B, T, H, src, ["wkv7_bwd.hip", "wkv7_common.h"]),
           "wkv7_fwd": group(tab, ["wkv7_fwd_kernel<rwkv7::bf16_t, true, false, 4>"], B, T, H, src, ["wkv7_fwd.hip", "wkv7_common.h"])}
    with open(os.path.join(ROOT, "profiles", "pmc_wkv7.json"), "w") as f:
        json.dump(out, f, indent=1)
    for k in ("wkv7c_bwd", "wkv7c_fwd"):
        d = out[k]
        print(f"{k}: HBM {(2 * d['FETCH_SIZE_KiB'] + d['WRITE_SIZE_KiB']) * 1024 / 1e9:.3f} GB per launch, MFMA busy {d['mfma_util']:.3f}, "
              f"VALU issue {d['valu_frac']:.3f}")


if __name__ == "__main__":
    main()
