"""Dev: the grouped weight-gradient launch of one ViT-B block (7 problems) + its fold, repeated -- traced with rocprofv3 to see both kernels
in isolation (usage through gpurun: rocprofv3 --kernel-trace --stats -d /tmp/x -- python3 tools/dev_tn_group.py)."""
import sys, torch
sys.path.insert(0, '.')
from tcow_amd import ops
dev = 'cuda'; M = 27090; D = 768
torch.manual_seed(0)
def t(n): return torch.randn(M, n, device=dev, dtype=torch.bfloat16)
shapes = [(3 * D, D), (D, D), (3 * D, D), (D, D), (4 * D, D), (D, 4 * D)]
probs = []
for (N, K) in shapes:
    probs.append((t(N), t(K), torch.empty(N, K, device=dev), torch.empty(N, device=dev)))
for _ in range(10):
    ops.gemm_tn_grouped(ops.BF16, probs)
torch.cuda.synchronize()
