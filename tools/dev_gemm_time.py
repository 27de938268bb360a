"""Dev: time the bf16 NT / TN GEMMs on the benchmark shapes (M = 27090) incl. the epilogues the engine issues; one line per case."""
import os, sys, torch
sys.path.insert(0, '.')
from tcow_amd import ops
dev = 'cuda'
M = 27090
def bench(f, n=30, w=5):
    for _ in range(w): f()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
tile = int(os.environ.get('TILE', '0'))
tot = 0.0
for (K, N, kind) in [(768, 768, 'bf16'), (768, 768, 'rowscale'), (768, 768, 'resid'), (768, 2304, 'bf16'), (2304, 768, 'bf16'), (768, 3072, 'dsave'), (3072, 768, 'resid'), (768, 3072, 'mulaux'), (3072, 768, 'bf16')]:
    A = torch.randn(M, K, device=dev).bfloat16(); W = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    bias = torch.randn(N, device=dev); rs = torch.rand(M, device=dev); res = torch.randn(M, N, device=dev); aux = torch.randn(M, N, device=dev).bfloat16()
    Cb = torch.empty(M, N, device=dev, dtype=torch.bfloat16); Cf = torch.empty(M, N, device=dev)
    f = {'bf16': lambda: ops.gemm_nt(ops.BF16, A, W, Cb, bias=bias, tile=tile), 'rowscale': lambda: ops.gemm_nt(ops.BF16, A, W, Cb, bias=bias, row_scale=rs, tile=tile),
         'resid': lambda: ops.gemm_nt(ops.BF16, A, W, Cf, bias=bias, row_scale=rs, resid=res, tile=tile), 'dsave': lambda: ops.gemm_nt(ops.BF16, A, W, Cb, bias=bias, act=ops.ACT_GELU_DSAVE, aux=aux, tile=tile),
         'mulaux': lambda: ops.gemm_nt(ops.BF16, A, W, Cb, act=ops.ACT_MUL_AUX, aux=aux, tile=tile)}[kind]
    t = bench(f); tot += t
    print(f'NT {M}x{K}x{N} {kind:8s}: {t:7.1f} us  {2.0 * M * K * N / t / 1e6:7.0f} TF', flush=True)
print(f'NT sum {tot:.1f} us')
tot = 0.0
for (N, K) in [(768, 768), (2304, 768), (3072, 768), (768, 3072)]:
    dY = torch.randn(M, N, device=dev).bfloat16(); X = torch.randn(M, K, device=dev).bfloat16(); dW = torch.empty(N, K, device=dev); db = torch.empty(N, device=dev)
    t = bench(lambda: ops.gemm_tn(ops.BF16, dY, X, dW, bias_grad=db)); tot += t
    print(f'TN {M}x{N}x{K}: {t:7.1f} us  {2.0 * M * K * N / t / 1e6:7.0f} TF', flush=True)
print(f'TN sum {tot:.1f} us')
