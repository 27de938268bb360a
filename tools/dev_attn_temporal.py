"""Dev: temporal attention fwd / bwd at the benchmark shape (HIP events around single launches)."""
import os, sys, torch
sys.path.insert(0, '.')
from tcow_amd import ops
dev = 'cuda'
B, T, S, heads = 3, 30, 301, 12; D = heads * 64; M = B * T * S
torch.manual_seed(0)
qkv = torch.randn(M, 3 * D, device=dev).bfloat16(); dout = torch.randn(M, D, device=dev).bfloat16()
def bench(f, n=20, w=5):
    for _ in range(w): f()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
shape = ops.attn_shape(ops.BF16, B, T, S, D, heads, 1)
out = torch.empty(M, D, device=dev, dtype=torch.bfloat16); lse = torch.empty(M, heads, device=dev); dqkv = torch.empty(M, 3 * D, device=dev, dtype=torch.bfloat16)
tf = bench(lambda: ops.attn_fwd(shape, False, qkv, out, lse)); tb = bench(lambda: ops.attn_bwd(shape, False, qkv, out, dout, lse, dqkv))
print(f'temporal fwd {tf:.1f} us ({4 * M * D * 2 / tf / 1e6:.2f} TB/s), bwd {tb:.1f} us ({(6 + 2) * M * D * 2 / tb / 1e6:.2f} TB/s of the 8 M D operands it has to move)', flush=True)
