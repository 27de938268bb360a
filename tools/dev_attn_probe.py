import sys, torch
sys.path.insert(0, '.')
from tcow_amd import ops
dev='cuda'
B,T,S,heads=3,30,301,12; D=heads*64; M=B*T*S
qkv=(torch.randn(M,3*D,device=dev)).bfloat16(); out=torch.empty(M,D,device=dev,dtype=torch.bfloat16); lse=torch.empty(M,heads,device=dev)
shape=ops.attn_shape(ops.BF16,B,T,S,D,heads,1)
def bench(f,n=20,w=5):
    for _ in range(w): f()
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/n*1e3
print('spatial fwd us:', round(bench(lambda: ops.attn_fwd(shape, True, qkv, out, lse)),1))
