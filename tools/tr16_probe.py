"""What does ds_read_b64_tr_b16 return?  Lane l points at row l of a [64][stride] u16 matrix whose element value is its
index; prints, for the first lanes, which (row, col) each of the 4 returned values came from."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rwkvtts_amd import _lib
dev = "cuda:0"
P = lambda t: ctypes.c_void_p(t.data_ptr())
for stride, desc in ((4, "row l = elements [4l, 4l+4)"), (40, "row stride 40"), (64, "row stride 64 (lanes 0..15 rows 0..15, +4 cols per 16 lanes)")):
    src = torch.arange(4096, dtype=torch.int16, device=dev)
    if stride == 64:
        addr = torch.tensor([(l % 16) * 64 + (l // 16) * 4 for l in range(64)], dtype=torch.int32, device=dev)
    else:
        addr = torch.tensor([l * stride for l in range(64)], dtype=torch.int32, device=dev)
    out = torch.zeros(64, 4, dtype=torch.int16, device=dev)
    rc = _lib.lib().rwkv7_debug_tr16(P(src), P(addr), P(out), None)
    torch.cuda.synchronize()
    print(f"--- {desc}; rc={rc}")
    o = out.cpu().tolist()
    a = addr.cpu().tolist()
    for l in list(range(0, 20)) + [31, 32, 33, 47, 48, 63]:
        srcs = []
        for v in o[l]:
            # which lane's row and which column did value v come from?
            owner = [(ll, v - a[ll]) for ll in range(64) if 0 <= v - a[ll] < 4]
            srcs.append(owner[0] if owner else ("?", v))
        print(f"lane {l:2d} addr {a[l]:4d}: values {o[l]} <- (lane,col) {srcs}")
