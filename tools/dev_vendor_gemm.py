"""Dev: hipBLASLt (through torch.matmul, bf16) on the benchmark's GEMM shapes next to libtcow_hip's kernels -- a yardstick for how much of
the 2.5 PFLOP/s the vendor library reaches at K = 768 / M = 27 090, not a product path."""
import sys, torch
sys.path.insert(0, '.')
from tcow_amd import ops
dev = 'cuda'; M = 27090
def bench(f, n=30, w=8):
    for _ in range(w): f()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
torch.manual_seed(0)
for N, K in ((768, 768), (2304, 768), (3072, 768), (768, 3072), (768, 2304)):
    A = torch.randn(M, K, device=dev).bfloat16(); W = torch.randn(N, K, device=dev).bfloat16(); out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    Wt = W.t().contiguous()
    t0 = bench(lambda: ops.gemm_nt(ops.BF16, A, W, out))
    t1 = bench(lambda: torch.matmul(A, W.t(), out=out)); t2 = bench(lambda: torch.matmul(A, Wt, out=out))
    fl = 2.0 * M * N * K
    print(f'NT {M}x{N}x{K}: tcow {t0:7.1f} us {fl/t0/1e6:6.0f} TF | hipBLASLt A@W.t() {t1:7.1f} us {fl/t1/1e6:6.0f} TF | A@Wt {t2:7.1f} us {fl/t2/1e6:6.0f} TF', flush=True)
for N1, N2 in ((768, 768), (2304, 768), (3072, 768), (768, 3072)):
    dY = torch.randn(M, N1, device=dev).bfloat16(); X = torch.randn(M, N2, device=dev).bfloat16(); o32 = torch.empty(N1, N2, device=dev); o16 = torch.empty(N1, N2, device=dev, dtype=torch.bfloat16)
    t0 = bench(lambda: ops.gemm_tn(ops.BF16, dY, X, o32))
    t1 = bench(lambda: torch.matmul(dY.t(), X, out=o16))
    fl = 2.0 * M * N1 * N2
    print(f'TN {M}x{N1}x{N2}: tcow {t0:7.1f} us {fl/t0/1e6:6.0f} TF | hipBLASLt dY.t()@X (bf16 out) {t1:7.1f} us {fl/t1/1e6:6.0f} TF', flush=True)
