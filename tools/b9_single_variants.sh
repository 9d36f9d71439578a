#!/bin/bash
# Builds tools/ab/lib_b9_<mask>.so for the masks given (default: the fourteen single-operand masks): csrc/wkv7_chunk_bwd9.hip compiled with
# -DWKV7C_B9_SINGLE=<mask>, linked with the objects of the current library build (python -m rwkvtts_amd.build first).  Then, on the GPU box:
#   for f in tools/ab/lib_b9_*.so; do RWKV7_HIP_SO=$f python tools/b9_single_probe.py; done
cd "$(dirname "$0")/.."
FLAGS="-O3 -std=c++17 -fPIC -ffast-math -fno-finite-math-only -fgpu-flush-denormals-to-zero -Wno-unused-result -Wno-pass-failed"
OTHERS=$(ls rwkvtts_amd/lib/*.o | grep -v wkv7_chunk_bwd9.o)
MASKS=${@:-1 2 4 8 16 32 64 128 256 512 1024 2048 4096 8192}
mkdir -p tools/ab
for m in $MASKS; do
  ( hipcc --offload-arch=gfx950 $FLAGS -DWKV7C_B9_SINGLE=$m -c rwkvtts_amd/csrc/wkv7_chunk_bwd9.hip -o /tmp/b9_$m.o && hipcc --offload-arch=gfx950 -shared -fPIC -o tools/ab/lib_b9_$m.so /tmp/b9_$m.o $OTHERS ) &
  while [ $(jobs -r | wc -l) -ge 7 ]; do sleep 1; done
done
wait
