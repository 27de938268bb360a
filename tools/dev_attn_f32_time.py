"""Dev: the f32 MFMA attention kernels (attention_f32.hip) at the benchmark shape, temporal and spatial, forward and backward."""
import sys, torch
sys.path.insert(0, '.')
from tcow_amd import ops
dev = 'cuda'; B, T, S, heads = 3, 30, 301, 12; D = heads * 64; M = B * T * S
torch.manual_seed(0)
qkv = torch.randn(M, 3 * D, device=dev); out = torch.empty(M, D, device=dev); lse = torch.empty(M, heads, device=dev)
dout = torch.randn(M, D, device=dev); dqkv = torch.empty(M, 3 * D, device=dev)
shape = ops.attn_shape(ops.F32, B, T, S, D, heads, 1)
def bench(f, n=10, w=3):
    for _ in range(w): f()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
for spatial in (True, False):
    tf = bench(lambda: ops.attn_fwd(shape, spatial, qkv, out, lse)); tb = bench(lambda: ops.attn_bwd(shape, spatial, qkv, out, dout, lse, dqkv))
    print(f'{"spatial" if spatial else "temporal"}: fwd {tf:.1f} us, bwd {tb:.1f} us', flush=True)
