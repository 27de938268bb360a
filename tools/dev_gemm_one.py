import sys, os, torch
sys.path.insert(0, '.')
from tcow_amd import ops
dev='cuda'
M, K, N = [int(x) for x in os.environ.get('MKN', '27090,768,3072').split(',')]
A = torch.randn(M, K, device=dev, dtype=torch.bfloat16); W = torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.05; C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
bias = torch.randn(N, device=dev)
for _ in range(5): ops.gemm_nt(ops.BF16, A, W, C, bias=bias, tile=int(os.environ.get('TILE', '0')))
torch.cuda.synchronize()
