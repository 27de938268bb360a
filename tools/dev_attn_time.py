"""Dev: time the spatial attention kernels one by one (HIP events around single launches) at the benchmark shape + check vs torch."""
import os, sys, torch
sys.path.insert(0, '.')
from tcow_amd import ops
sys.path.insert(0, 'tools')
dev = 'cuda'
B, T, S, heads = int(os.environ.get('B', '3')), 30, int(os.environ.get('S', '301')), 12; D = heads * 64; M = B * T * S
torch.manual_seed(0)
qkv = torch.randn(M, 3 * D, device=dev).bfloat16(); out = torch.empty(M, D, device=dev, dtype=torch.bfloat16); lse = torch.empty(M, heads, device=dev)
dout = torch.randn(M, D, device=dev).bfloat16(); dqkv = torch.empty(M, 3 * D, device=dev, dtype=torch.bfloat16)
shape = ops.attn_shape(ops.BF16, B, T, S, D, heads, 1)
def bench(f, n=20, w=5):
    for _ in range(w): f()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
tf = bench(lambda: ops.attn_fwd(shape, True, qkv, out, lse)); tb = bench(lambda: ops.attn_bwd(shape, True, qkv, out, dout, lse, dqkv))
# correctness on one frame / two heads against torch f32
x = qkv.float().reshape(B, T, S, 3, heads, 64)[0, 0].requires_grad_(True)       # S,3,h,64
q, k, v = [x[:, i].permute(1, 0, 2) for i in range(3)]
o = ((q @ k.transpose(-1, -2)) * 0.125).softmax(-1) @ v                         # h,S,64
ref = o.permute(1, 0, 2).reshape(S, D)
ef = (out[:S].float() - ref).abs().max().item()
(ref * dout[:S].float()).sum().backward()
eb = (dqkv[:S].float().reshape(S, 3, heads, 64) - x.grad).abs().max().item()
print(f'spatial fwd {tf:.1f} us, bwd {tb:.1f} us | max|d| fwd {ef:.2e} bwd {eb:.2e} (ref max {ref.abs().max().item():.2f} / {x.grad.abs().max().item():.2f})', flush=True)
