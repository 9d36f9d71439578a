import torch, sys
sys.path.insert(0,'/root/repo')
from rwkvtts_amd import fused
dev='cuda:0'
M,N,K=32768,576,1024
dys=[torch.randn(M,N,device=dev).bfloat16() for _ in range(4)]
xs=[torch.randn(M,K,device=dev).bfloat16() for _ in range(4)]
def t(fn,n=20):
    for i in range(4): fn(i)
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for j in range(n): fn(j%4)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/n*1e3
print("mm(dy.t, x)", t(lambda i: torch.mm(dys[i].t(), xs[i])))
for S in (2,4,8,16,32,64):
    def f(i,S=S):
        return torch.bmm(dys[i].view(S,M//S,N).transpose(1,2), xs[i].view(S,M//S,K), out_dtype=torch.float32)
    print("bmm slabs fp32", S, t(f))
    def g(i,S=S):
        return torch.bmm(dys[i].view(S,M//S,N).transpose(1,2), xs[i].view(S,M//S,K))
    print("bmm slabs bf16", S, t(g))
print("wgrad_splitk now", t(lambda i: fused.wgrad_splitk(dys[i], xs[i])))
# forward / dgrad shapes
w=torch.randn(N,K,device=dev).bfloat16()
print("fwd mm [M,1024]x[1024,576]", t(lambda i: torch.mm(xs[i], w.t())))
dx=[torch.randn(M,K,device=dev).bfloat16() for _ in range(4)]
print("addmm_ [M,576]x[576,1024]", t(lambda i: dx[i].addmm_(dys[i], w)))
print("mm   [M,576]x[576,1024]", t(lambda i: torch.mm(dys[i], w)))
