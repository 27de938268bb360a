"""Dev: weight-gradient GEMM (incl. fold) as a function of the token count M: slope = stage loop, intercept = prologue + slab write + fold."""
import sys, torch
sys.path.insert(0, '.')
from tcow_amd import ops
dev = 'cuda'
def bench(f, n=30, w=5):
    for _ in range(w): f()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
for (N, K) in [(3072, 768), (768, 768)]:
    for M in (3584, 7168, 14336, 27090, 54180):
        dY = torch.randn(M, N, device=dev).bfloat16(); X = torch.randn(M, K, device=dev).bfloat16(); dW = torch.empty(N, K, device=dev); db = torch.empty(N, device=dev)
        t = bench(lambda: ops.gemm_tn(ops.BF16, dY, X, dW, bias_grad=db)); t2 = bench(lambda: ops.gemm_tn(ops.BF16, dY, X, dW))
        print(f'TN N={N} K={K} M={M}: with bias grad {t:7.1f} us  ({2.0*M*N*K/t/1e6:5.0f} TF)   without {t2:7.1f} us', flush=True)
