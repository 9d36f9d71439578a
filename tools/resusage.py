"""Compile one .hip file with -Rpass-analysis=kernel-resource-usage and print one line per kernel: VGPR / AGPR / SGPR / spills / scratch / LDS.
    python tools/resusage.py rwkvtts_amd/csrc/decode_step.hip [filter]"""
import re, subprocess, sys
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffast-math", "-fno-finite-math-only", "-fgpu-flush-denormals-to-zero",
       "-Wno-unused-result", "-Wno-pass-failed", "-c", src, "-o", "/tmp/_res.o", "-Rpass-analysis=kernel-resource-usage"] + sys.argv[3:]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    if "error" in line:
        print(line)
    m = re.search(r"remark:\s+(.*?): (.*?) \[-Rpass", line)
    if not m:
        continue
    k, v = m.group(1).strip(), m.group(2).strip()
    if k == "Function Name":
        cur = subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip()
        rows[cur] = {}
    elif cur:
        rows[cur][k] = v
for name, r in rows.items():
    if flt and flt not in name:
        continue
    short = re.sub(r"rwkv7::\(anonymous namespace\)::|rwkv7::", "", name)[:90]
    print(f"{short:90s} V {str(r.get('VGPRs')):>4} A {str(r.get('AGPRs')):>4} S {str(r.get('SGPRs')):>4} spillV {str(r.get('VGPRs Spill')):>4} spillS {str(r.get('SGPRs Spill')):>4} scratch {str(r.get('ScratchSize [bytes/lane]')):>5} occ {r.get('Occupancy [waves/SIMD]')} lds {r.get('LDS Size [bytes/block]')}")
