"""Dev: cost of the row-operand epilogues (residual / row scale / aux) on the path's GEMM shapes."""
import sys, torch
sys.path.insert(0, '.')
from tcow_amd import ops
dev = 'cuda'
def bench(f, n=30, w=5):
    for _ in range(w): f()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e-3
torch.manual_seed(0)
for (M, K, N) in [(27090, 768, 768), (27090, 3072, 768)]:
    A = torch.randn(M, K, device=dev, dtype=torch.bfloat16); W = torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.05
    Cb = torch.empty(M, N, device=dev, dtype=torch.bfloat16); Cf = torch.empty(M, N, device=dev); R = torch.randn(M, N, device=dev); bias = torch.randn(N, device=dev); rs = torch.rand(M, device=dev)
    for name, out, kw in (('bias -> bf16', Cb, dict(bias=bias)), ('bias -> f32', Cf, dict(bias=bias)), ('bias+rowscale -> bf16', Cb, dict(bias=bias, row_scale=rs)),
                          ('bias+rowscale+resid -> f32', Cf, dict(bias=bias, row_scale=rs, resid=R))):
        t = bench(lambda: ops.gemm_nt(ops.BF16, A, W, out, **kw)); print(f'{M}x{K}x{N} {name}: {t*1e6:.1f} us {2*M*K*N/t/1e12:.0f} TF', flush=True)
