import sys, torch
sys.path.insert(0, '.')
from tcow_amd import ops
dev='cuda'; M=27090
def bench(f, n=30, w=5):
    for _ in range(w): f()
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/n*1e3
for N in (768, 2304, 3072):
    for K in (64, 128, 256, 768, 1536, 3072):
        A=torch.randn(M,K,device=dev).bfloat16(); W=(torch.randn(N,K,device=dev)*0.05).bfloat16(); C=torch.empty(M,N,device=dev,dtype=torch.bfloat16); b=torch.randn(N,device=dev)
        Cf=torch.empty(M,N,device=dev); R=torch.randn(M,N,device=dev); rs=torch.rand(M,device=dev)
        t=bench(lambda: ops.gemm_nt(ops.BF16,A,W,C,bias=b)); t2=bench(lambda: ops.gemm_nt(ops.BF16,A,W,Cf,bias=b,row_scale=rs,resid=R))
        print(f'N={N} K={K}: plain bf16 {t:7.1f} us   resid f32 {t2:7.1f} us', flush=True)
