"""Dev: A/B of the epilogue-operand L2 prefetch (TCOW_GEMM_EPF=0|1) on the path's row-operand GEMMs; operands rotate over 4 buffer sets so that
nothing is served from the 256 MiB Infinity Cache."""
import os, subprocess, sys
if len(sys.argv) > 1:
    import torch
    sys.path.insert(0, '.')
    from tcow_amd import ops
    dev = 'cuda'; M = 27090; NB = 4
    def bench(fs, n=40, w=8):
        for i in range(w): fs[i % NB]()
        torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(n): fs[i % NB]()
        e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e-3
    torch.manual_seed(0)
    for (K, N, kind) in [(768, 768, 'resid'), (3072, 768, 'resid'), (768, 3072, 'mul_aux'), (768, 3072, 'gelu_dsave'), (768, 2304, 'plain'), (768, 768, 'fold')]:
        fs = []
        for b in range(NB):
            A = torch.randn(M, K, device=dev, dtype=torch.bfloat16); W = torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.05
            bias = torch.randn(N, device=dev); rs = torch.rand(M, device=dev)
            if kind in ('resid', 'fold'):
                R = torch.randn(M, N, device=dev); C = torch.empty(M, N, device=dev)
                kw = dict(bias=bias, row_scale=rs, resid=R)
                if kind == 'fold': kw.update(bias2=bias.clone(), row_scale2=rs.clone())
            elif kind == 'mul_aux':
                C = torch.empty(M, N, device=dev, dtype=torch.bfloat16); aux = torch.randn(M, N, device=dev, dtype=torch.bfloat16)
                kw = dict(act=ops.ACT_MUL_AUX, aux=aux)
            elif kind == 'gelu_dsave':
                C = torch.empty(M, N, device=dev, dtype=torch.bfloat16); aux = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
                kw = dict(bias=bias, act=ops.ACT_GELU_DSAVE, aux=aux)
            else:
                C = torch.empty(M, N, device=dev, dtype=torch.bfloat16); kw = dict(bias=bias)
            fs.append(lambda A=A, W=W, C=C, kw=kw: ops.gemm_nt(ops.BF16, A, W, C, **kw))
        t = bench(fs); print(f'EPF={os.environ.get("TCOW_GEMM_EPF", "1")} {M}x{K}x{N} {kind:10s}: {t*1e6:7.1f} us {2*M*K*N/t/1e12:5.0f} TF', flush=True)
else:
    for v in ('0', '1', '2', '3'):
        subprocess.run([sys.executable, __file__, 'child'], env=dict(os.environ, TCOW_GEMM_EPF=v), check=False)
