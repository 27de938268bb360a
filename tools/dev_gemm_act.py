"""Dev: cost of the fused GELU / GELU' epilogues on the fc1-shaped GEMMs."""
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
M, K, N = 27090, 768, 3072
A = torch.randn(M, K, device=dev, dtype=torch.bfloat16); W = torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.05
C = torch.empty(M, N, device=dev, dtype=torch.bfloat16); aux = torch.empty(M, N, device=dev, dtype=torch.bfloat16); bias = torch.randn(N, device=dev)
ref = A.float() @ W.float().t() + bias
for name, kw in (('bias', dict(bias=bias)), ('bias+gelu+aux', dict(bias=bias, act=ops.ACT_GELU, aux=aux)), ('dgelu(aux)', dict(act=ops.ACT_DGELU, aux=aux)),
                 ('bias+gelu+dsave', dict(bias=bias, act=ops.ACT_GELU_DSAVE, aux=aux)), ('mul_aux', dict(act=ops.ACT_MUL_AUX, aux=aux))):
    ops.gemm_nt(ops.BF16, A, W, C, **kw)
    if name == 'bias+gelu+aux':
        want = torch.nn.functional.gelu(ref); err = ((C.float() - want).abs().max() / want.abs().max()).item(); e2 = ((aux.float() - ref).abs().max() / ref.abs().max()).item()
        print(f'  gelu rel err {err:.2e}, aux rel err {e2:.2e}')
    if name.startswith('dgelu'):
        x = aux.float().requires_grad_(True); g = torch.autograd.grad(torch.nn.functional.gelu(x).sum(), x)[0]
        want = (A.float() @ W.float().t()) * g; err = ((C.float() - want).abs().max() / want.abs().max()).item()
        print(f'  dgelu rel err {err:.2e}')
    t = bench(lambda: ops.gemm_nt(ops.BF16, A, W, C, **kw)); print(f'{name}: {t*1e6:.1f} us {2*M*K*N/t/1e12:.0f} TF', flush=True)
