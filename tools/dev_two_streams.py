"""Dev: does running the weight-gradient (TN) GEMMs on a second stream next to the data-gradient chain (NT GEMMs with f32-residual
epilogues, LayerNorm backward) shorten the sum?  Serial vs two-stream wall time of the same launches."""
import sys, torch
sys.path.insert(0, '.')
from tcow_amd import ops
dev = 'cuda'; M = 27090; D = 768
torch.manual_seed(0)
A = torch.randn(M, D, device=dev).bfloat16(); A4 = torch.randn(M, 4 * D, device=dev).bfloat16()
W = torch.randn(D, D, device=dev).bfloat16(); W4 = torch.randn(D, 4 * D, device=dev).bfloat16(); W4b = torch.randn(4 * D, D, device=dev).bfloat16()
R = torch.randn(M, D, device=dev); O = torch.empty(M, D, device=dev); Ob = torch.empty(M, D, device=dev, dtype=torch.bfloat16); O4 = torch.empty(M, 4 * D, device=dev, dtype=torch.bfloat16)
dW = torch.empty(D, D, device=dev); dW4 = torch.empty(4 * D, D, device=dev); dW4b = torch.empty(D, 4 * D, device=dev)
def nt_chain():
    for _ in range(4):
        ops.gemm_nt(ops.BF16, A, W, O, resid=R)          # f32 residual epilogue (HBM-bound)
        ops.gemm_nt(ops.BF16, A4, W4, Ob)                # K = 3072
        ops.gemm_nt(ops.BF16, A, W4b, O4)                # N = 3072
def tn_chain():
    for _ in range(4):
        ops.gemm_tn(ops.BF16, A, A, dW); ops.gemm_tn(ops.BF16, A4, A, dW4); ops.gemm_tn(ops.BF16, A, A4, dW4b)
s2 = torch.cuda.Stream()
def serial(): nt_chain(); tn_chain()
def dual():
    s2.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s2): tn_chain()
    nt_chain()
    torch.cuda.current_stream().wait_stream(s2)
def bench(f, n=10, w=3):
    for _ in range(w): f()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
print(f'nt chain {bench(nt_chain):.0f} us, tn chain {bench(tn_chain):.0f} us, serial {bench(serial):.0f} us, two streams {bench(dual):.0f} us')
