"""Dev: producer -> consumer kernel pairs of one transformer block at M = 27 090, timed as pairs (HIP events around 20 repetitions of the pair) -- the
measurement behind the cache-policy choices of the GEMM epilogues (profiles/r05_nontemporal.txt): a non-temporal store helps the producer and may cost
the consumer its Infinity-Cache hits.  Run once per library build (TCOW_LIB=...)."""
import os, sys, torch
sys.path.insert(0, '.')
from tcow_amd import ops
dev = 'cuda'; B, T, S, heads = 3, 30, 301, 12; D = 768; M = B * T * S
torch.manual_seed(0)
FC1_TILE = int(os.environ.get('FC1_TILE', '0')); FC2G_TILE = int(os.environ.get('FC2G_TILE', '0'))      # forced tile of the two GELU GEMMs (0 = the library's routing)
def bench(f, n=20, w=4):
    for _ in range(w): f()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
bf = lambda *s: torch.randn(*s, device=dev).bfloat16()
x = bf(M, D); W1 = (torch.randn(4 * D, D, device=dev) * 0.05).bfloat16(); W2 = (torch.randn(D, 4 * D, device=dev) * 0.05).bfloat16()
Wq = (torch.randn(3 * D, D, device=dev) * 0.05).bfloat16(); Wp = (torch.randn(D, D, device=dev) * 0.05).bfloat16()
b1 = torch.randn(4 * D, device=dev); b2 = torch.randn(D, device=dev); bq = torch.randn(3 * D, device=dev)
g = torch.empty(M, 4 * D, device=dev, dtype=torch.bfloat16); d = torch.empty_like(g); res = torch.randn(M, D, device=dev); y = torch.empty(M, D, device=dev)
rs = torch.rand(M, device=dev)
qkv = torch.empty(M, 3 * D, device=dev, dtype=torch.bfloat16); ao = torch.empty(M, D, device=dev, dtype=torch.bfloat16); lse = torch.empty(M, heads, device=dev)
shape = ops.attn_shape(ops.BF16, B, T, S, D, heads, 1)
dy = bf(M, D); dh = torch.empty(M, 4 * D, device=dev, dtype=torch.bfloat16); dx = torch.empty(M, D, device=dev, dtype=torch.bfloat16)
dqkv = bf(M, 3 * D); dxq = torch.empty(M, D, device=dev, dtype=torch.bfloat16)
pairs = {
    'fc1 (GELU, GELU\' saved) -> fc2 (+ residual)': lambda: (ops.gemm_nt(ops.BF16, x, W1, g, bias=b1, act=ops.ACT_GELU_DSAVE, aux=d, tile=FC1_TILE), ops.gemm_nt(ops.BF16, g, W2, y, bias=b2, row_scale=rs, resid=res)),
    'qkv -> spatial attention forward': lambda: (ops.gemm_nt(ops.BF16, x, Wq, qkv, bias=bq), ops.attn_fwd(shape, True, qkv, ao, lse)),
    'fc2 input gradient x GELU\' -> fc1 input gradient': lambda: (ops.gemm_nt(ops.BF16, dy, W2.t().contiguous(), dh, act=ops.ACT_MUL_AUX, aux=d, tile=FC2G_TILE), ops.gemm_nt(ops.BF16, dh, W1.t().contiguous(), dx)),
    'qkv input gradient (K = 2304) -> projection input gradient (K = 768)': lambda: (ops.gemm_nt(ops.BF16, dqkv, Wq.t().contiguous(), dxq), ops.gemm_nt(ops.BF16, dxq, Wp, dx)),
    'temporal projection (row scale) -> qkv': lambda: (ops.gemm_nt(ops.BF16, x, Wp, dx, bias=b2, row_scale=rs), ops.gemm_nt(ops.BF16, dx, Wq, qkv, bias=bq)),
}
W2t = W2.t().contiguous(); W1t = W1.t().contiguous(); Wqt = Wq.t().contiguous()
pairs['fc2 input gradient x GELU\' -> fc1 input gradient'] = lambda: (ops.gemm_nt(ops.BF16, dy, W2t, dh, act=ops.ACT_MUL_AUX, aux=d, tile=FC2G_TILE), ops.gemm_nt(ops.BF16, dh, W1t, dx))
pairs['qkv input gradient (K = 2304) -> projection input gradient (K = 768)'] = lambda: (ops.gemm_nt(ops.BF16, dqkv, Wqt, dxq), ops.gemm_nt(ops.BF16, dxq, Wp, dx))
tot = 0.0
for name, f in pairs.items():
    t = bench(f); tot += t
    print(f'{name:70s} {t:7.1f} us', flush=True)
print(f'sum {tot:.1f} us   lib {os.environ.get("TCOW_LIB", "(shipped)")} fc1 tile {FC1_TILE} fc2-grad tile {FC2G_TILE}', flush=True)
