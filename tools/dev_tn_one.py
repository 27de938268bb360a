import sys, os, torch
sys.path.insert(0, '.')
from tcow_amd import ops
dev='cuda'
M, N, K = [int(x) for x in os.environ.get('MNK', '27090,768,768').split(',')]
dY = torch.randn(M, N, device=dev, dtype=torch.bfloat16); X = torch.randn(M, K, device=dev, dtype=torch.bfloat16); dW = torch.empty(N, K, device=dev); db = torch.empty(N, device=dev)
for _ in range(5): ops.gemm_tn(ops.BF16, dY, X, dW, bias_grad=db)
torch.cuda.synchronize()
