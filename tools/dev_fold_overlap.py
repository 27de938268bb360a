"""Dev: can a light HBM-bound kernel (the slab fold: 198 MB, no LDS, few registers) run UNDER the compute-bound kernels of the backward chain when it
is launched on a second stream?  Serial vs two-stream time of [NT GEMM chain | attention backward] + [a streaming kernel of the fold's size]."""
import sys, torch
sys.path.insert(0, '.')
from tcow_amd import ops
dev = 'cuda'; M = 27090; D = 768
torch.manual_seed(0)
A = torch.randn(M, D, device=dev).bfloat16(); A4 = torch.randn(M, 4 * D, device=dev).bfloat16()
W4 = torch.randn(D, 4 * D, device=dev).bfloat16(); W4b = torch.randn(4 * D, D, device=dev).bfloat16(); aux = torch.randn(M, 4 * D, device=dev).bfloat16()
Ob = torch.empty(M, D, device=dev, dtype=torch.bfloat16); O4 = torch.empty(M, 4 * D, device=dev, dtype=torch.bfloat16)
slab = torch.randn(5, 7 * 1024 * 1024, device=dev); out = torch.empty(7 * 1024 * 1024, device=dev)      # 5 x 28 MB in, 28 MB out
B, T, S, heads = 3, 30, 301, 12
qkv = torch.randn(M, 3 * D, device=dev).bfloat16(); o = torch.empty(M, D, device=dev, dtype=torch.bfloat16); lse = torch.empty(M, heads, device=dev)
dout = torch.randn(M, D, device=dev).bfloat16(); dqkv = torch.empty(M, 3 * D, device=dev, dtype=torch.bfloat16)
shape = ops.attn_shape(ops.BF16, B, T, S, D, heads, 1)
ops.attn_fwd(shape, True, qkv, o, lse)
def fold(): torch.sum(slab, dim=0, out=out)
def gemms():
    ops.gemm_nt(ops.BF16, A, W4b, O4, act='mul_aux', aux=aux) if False else ops.gemm_nt(ops.BF16, A, W4b, O4)     # N = 3072
    ops.gemm_nt(ops.BF16, A4, W4, Ob)                                                                             # K = 3072
def attn(): ops.attn_bwd(shape, True, qkv, o, dout, lse, dqkv)
s2 = torch.cuda.Stream()
def bench(f, n=20, w=3):
    for _ in range(w): f()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
for name, main in (('two NT GEMMs', gemms), ('spatial attention backward', attn)):
    def serial(): fold(); main()
    def dual():
        s2.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s2): fold()
        main()
        torch.cuda.current_stream().wait_stream(s2)
    print(f'{name}: alone {bench(main):.0f} us, fold-sized stream kernel alone {bench(fold):.0f} us, serial {bench(serial):.0f} us, two streams {bench(dual):.0f} us', flush=True)
