import sys, os, torch, ctypes
sys.path.insert(0, '.')
from tcow_amd import ops
dev='cuda'
def bench(f, n=20, w=5):
    for _ in range(w): f()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e-3
for (M, K, N) in [(27090, 768, 768), (27090, 768, 3072), (27090, 3072, 768), (8192, 8192, 8192)]:
    A = torch.randn(M, K, device=dev, dtype=torch.bfloat16); W = torch.randn(N, K, device=dev, dtype=torch.bfloat16); C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    t = bench(lambda: ops.gemm_nt(ops.BF16, A, W, C)); print(f'dbg={os.environ.get("TCOW_GEMM_DBG","0")} {M}x{K}x{N}: {t*1e6:.1f} us {2*M*K*N/t/1e12:.0f} TF', flush=True)
