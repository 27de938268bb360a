"""Dev: spatial attention forward + backward at the benchmark shape (B*Qs = 3, T = 30, S = 301, 12 heads), a few launches -- the
program rocprofv3 wraps for the per-kernel PMC passes of tools/pmc_attn.sh."""
import os, sys, torch
sys.path.insert(0, '.')
from tcow_amd import ops
dev = 'cuda'
B, T, S, heads = 3, 30, int(os.environ.get('S', '301')), 12; D = heads * 64; M = B * T * S
torch.manual_seed(0)
qkv = torch.randn(M, 3 * D, device=dev).bfloat16(); out = torch.empty(M, D, device=dev, dtype=torch.bfloat16); lse = torch.empty(M, heads, device=dev)
dout = torch.randn(M, D, device=dev).bfloat16(); dqkv = torch.empty(M, 3 * D, device=dev, dtype=torch.bfloat16)
shape = ops.attn_shape(ops.BF16, B, T, S, D, heads, 1)
spatial = os.environ.get('TEMPORAL', '0') != '1'
for _ in range(int(os.environ.get('N', '6'))):
    ops.attn_fwd(shape, spatial, qkv, out, lse)
    ops.attn_bwd(shape, spatial, qkv, out, dout, lse, dqkv)
torch.cuda.synchronize()
