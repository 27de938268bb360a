"""Dev: the 320 x 256 NT kernel's main-loop variants (TCOW_GEMM_ML, read once per process) on the path's shapes at M = 27090."""
import os, sys, torch
sys.path.insert(0, '.')
from tcow_amd import ops
dev = 'cuda'; M = 27090
def bench(f, n=40, w=8):
    for _ in range(w): f()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
out = []; tot = 0.0
for (K, N, kind) in [(768, 768, 'bf16'), (768, 768, 'resid'), (768, 2304, 'bf16'), (2304, 768, 'bf16'), (768, 3072, 'dsave'), (3072, 768, 'resid'), (768, 3072, 'mulaux'), (3072, 768, 'bf16'), (3072, 3072, 'bf16')]:
    A = torch.randn(M, K, device=dev).bfloat16(); W = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    bias = torch.randn(N, device=dev); rs = torch.rand(M, device=dev); res = torch.randn(M, N, device=dev); aux = torch.randn(M, N, device=dev).bfloat16()
    Cb = torch.empty(M, N, device=dev, dtype=torch.bfloat16); Cf = torch.empty(M, N, device=dev)
    f = {'bf16': lambda: ops.gemm_nt(ops.BF16, A, W, Cb, bias=bias, tile=320), 'resid': lambda: ops.gemm_nt(ops.BF16, A, W, Cf, bias=bias, row_scale=rs, resid=res, tile=320),
         'dsave': lambda: ops.gemm_nt(ops.BF16, A, W, Cb, bias=bias, act=ops.ACT_GELU_DSAVE, aux=aux, tile=320), 'mulaux': lambda: ops.gemm_nt(ops.BF16, A, W, Cb, act=ops.ACT_MUL_AUX, aux=aux, tile=320)}[kind]
    t = bench(f); tot += t
    ref = (A[:512].double() @ W.double().t() + (bias.double() if kind != 'mulaux' else 0))
    got = (Cf if kind == 'resid' else Cb)[:512].double()
    if kind == 'resid': ref = ref * rs[:512, None].double() + res[:512].double()
    err = float((got - ref).abs().max()) if kind in ('bf16', 'resid') else -1.0
    out.append(f'{K}->{N} {kind} {t:6.1f} ({err:.2g})')
print(f"ML {os.environ.get('TCOW_GEMM_ML', '1')}: " + ' | '.join(out) + f' | sum {tot:.1f}', flush=True)
