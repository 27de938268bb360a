"""Dev: exact-f32 GEMM kernel vs the bf16 x 3 split kernel on the benchmark's shapes (time + error against f64)."""
import sys, torch
sys.path.insert(0, '.')
from tcow_amd import ops
dev = 'cuda'; M = 27090
def bench(f, n=5, w=2):
    for _ in range(w): f()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
torch.manual_seed(0)
for N, K in ((768, 768), (2304, 768), (3072, 768), (768, 3072)):
    A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) * 0.05; out = torch.empty(M, N, device=dev); dW = torch.empty(K, N, device=dev)
    ref = A[:2048].double() @ W.double().t()
    fl = 2.0 * M * N * K
    for name, mode in (('f32', ops.F32), ('x3', ops.F32X3)):
        t = bench(lambda: ops.gemm_nt(mode, A, W, out)); err = float((out[:2048].double() - ref).abs().max() / ref.abs().max())
        dY = out
        t2 = bench(lambda: ops.gemm_tn(mode, A, dY, dW))
        print(f'{name:4s} NT {M}x{N}x{K}: {t:8.1f} us {fl/t/1e6:6.0f} TF  rel err {err:.1e} | TN {t2:8.1f} us {fl/t2/1e6:6.0f} TF', flush=True)
