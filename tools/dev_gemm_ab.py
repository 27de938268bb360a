import sys, os, torch
sys.path.insert(0, '.')
from tcow_amd import ops
dev='cuda'
def bench(f, n=30, w=5):
    for _ in range(w): f()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e-3
torch.manual_seed(0)
for (M, K, N) in [(1000, 64, 192), (555, 128, 64), (27090, 768, 768), (27090, 768, 2304), (27090, 768, 3072), (27090, 3072, 768), (9030, 768, 768), (8192, 8192, 8192)]:
    A = torch.randn(M, K, device=dev, dtype=torch.bfloat16); W = torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.05; C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    bias = torch.randn(N, device=dev)
    ops.gemm_nt(ops.BF16, A, W, C, bias=bias); ref = A.float() @ W.float().t() + bias
    err = ((C.float() - ref).abs().max() / ref.abs().max()).item()
    t = bench(lambda: ops.gemm_nt(ops.BF16, A, W, C, bias=bias)); print(f'v={os.environ.get("TCOW_GEMM_NT","2")} {M}x{K}x{N}: rel err {err:.2e}  {t*1e6:.1f} us {2*M*K*N/t/1e12:.0f} TF', flush=True)
