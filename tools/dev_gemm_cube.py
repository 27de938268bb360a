"""Dev: the NT tile kernels on square problems (uniform random [-1, 1) operands, plain bf16 output), for comparison with published same-hardware
numbers (MI355X guide: 256-square 8-phase HIP template 1320-1340 TFLOP/s at 4096^3, ~1470 at 8192^3)."""
import sys, torch
sys.path.insert(0, '.')
from tcow_amd import ops
dev = 'cuda'
def bench(f, n=20, w=5):
    for _ in range(w): f()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e-3
torch.manual_seed(0)
for n in (4096, 8192):
    A = (torch.rand(n, n, device=dev) * 2 - 1).bfloat16(); W = (torch.rand(n, n, device=dev) * 2 - 1).bfloat16(); C = torch.empty(n, n, device=dev, dtype=torch.bfloat16)
    for tile in (320, 256, 128, 0):
        t = bench(lambda: ops.gemm_nt(ops.BF16, A, W, C, tile=tile))
        print(f'{n}^3 tile {tile}: {t*1e6:8.1f} us  {2.0*n**3/t/1e12:7.0f} TFLOP/s', flush=True)
# path shapes with a plain epilogue (the launches the phase-structured kernel can take)
for (M, K, N) in [(27090, 768, 2304), (27090, 2304, 768), (27090, 3072, 768), (27090, 768, 768)]:
    A = (torch.rand(M, K, device=dev) * 2 - 1).bfloat16(); W = ((torch.rand(N, K, device=dev) * 2 - 1) * 0.05).bfloat16(); C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    bias = torch.randn(N, device=dev)
    t = bench(lambda: ops.gemm_nt(ops.BF16, A, W, C, bias=bias))
    print(f'{M}x{K}x{N} default dispatch: {t*1e6:8.1f} us  {2.0*M*K*N/t/1e12:7.0f} TFLOP/s', flush=True)
