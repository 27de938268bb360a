"""Dev: the 160 x 256 two-per-CU NT kernel (tile=160) against the 320 x 256 kernel (tile=320) on the path's shapes at M = 27090: time + max difference (the
two produce bit-identical outputs: same K order per element).  TILES=320,321,160 ran the persistent form of round 5 (tools/gemm_nt_320p.inc) as well."""
import os, sys, torch
sys.path.insert(0, '.')
from tcow_amd import ops
dev = 'cuda'; M = 27090
def bench(f, n=30, w=5):
    for _ in range(w): f()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
TILES = tuple(int(x) for x in os.environ.get('TILES', '320,160').split(','))
tot = {t: 0.0 for t in TILES}
check = os.environ.get('CHECK', '1') == '1'
for (K, N, kind) in [(768, 768, 'bf16'), (768, 768, 'rowscale'), (768, 768, 'resid'), (768, 2304, 'bf16'), (2304, 768, 'bf16'), (768, 3072, 'dsave'), (3072, 768, 'resid'), (768, 3072, 'mulaux'), (3072, 768, 'bf16')]:
    A = torch.randn(M, K, device=dev).bfloat16(); W = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    bias = torch.randn(N, device=dev); rs = torch.rand(M, device=dev); res = torch.randn(M, N, device=dev); aux = torch.randn(M, N, device=dev).bfloat16()
    line = f'NT {M}x{K}x{N} {kind:8s}:'
    outs = {}
    for tile in TILES:
        Cb = torch.empty(M, N, device=dev, dtype=torch.bfloat16); Cf = torch.empty(M, N, device=dev); ax = aux.clone()
        f = {'bf16': lambda: ops.gemm_nt(ops.BF16, A, W, Cb, bias=bias, tile=tile), 'rowscale': lambda: ops.gemm_nt(ops.BF16, A, W, Cb, bias=bias, row_scale=rs, tile=tile),
             'resid': lambda: ops.gemm_nt(ops.BF16, A, W, Cf, bias=bias, row_scale=rs, resid=res, tile=tile), 'dsave': lambda: ops.gemm_nt(ops.BF16, A, W, Cb, bias=bias, act=ops.ACT_GELU_DSAVE, aux=ax, tile=tile),
             'mulaux': lambda: ops.gemm_nt(ops.BF16, A, W, Cb, act=ops.ACT_MUL_AUX, aux=ax, tile=tile)}[kind]
        t = bench(f); tot[tile] += t
        outs[tile] = (Cf if kind == 'resid' else Cb).float().clone()
        line += f'  tile {tile}: {t:7.1f} us {2.0 * M * K * N / t / 1e6:6.0f} TF'
    if check:
        d = float((outs[320] - outs[160]).abs().max()); line += f'   max|320-160| {d:.3g} (max {float(outs[320].abs().max()):.3g})'
    print(line, flush=True)
print('NT sum ' + '   '.join(f'{t}: {tot[t]:.1f} us' for t in TILES), flush=True)
