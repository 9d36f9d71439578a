import sys, torch
sys.path.insert(0, '/root/repo')
from rwkvtts_amd import _lib, ops
from rwkvtts_amd.synthetic import make_wkv_inputs
mode = sys.argv[1]
if mode == 'early':
    _lib.lib()
w,q,k,v,a,b = [t.view(1,16,64).to('cuda:0') for t in make_wkv_inputs(1,16,1,0,torch.float32)]
try:
    y = ops.wkv7_forward_nograd(q,w,k,v,a,b); torch.cuda.synchronize(); print(mode, 'ok', y.abs().sum().item())
except Exception as e: print(mode, 'ERR', e)
