"""Dev: time the weight-gradient GEMM (kernel + fold) on the path's shapes."""
import sys, os, torch
sys.path.insert(0, '.')
from tcow_amd import ops
dev = 'cuda'
def bench(f, n=30, w=5):
    for _ in range(w): f()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e-3
for (M, N, K) in [(27090, 768, 768), (27090, 3072, 768), (27090, 768, 3072), (27090, 2304, 768)]:
    dY = torch.randn(M, N, device=dev, dtype=torch.bfloat16); X = torch.randn(M, K, device=dev, dtype=torch.bfloat16); dW = torch.empty(N, K, device=dev); db = torch.empty(N, device=dev)
    t = bench(lambda: ops.gemm_tn(ops.BF16, dY, X, dW, bias_grad=db))
    print(f'TN {M}x{N}x{K}: {t*1e6:.1f} us {2*M*N*K/t/1e12:.0f} TF', flush=True)
