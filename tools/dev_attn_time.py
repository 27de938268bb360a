"""Dev: time the spatial attention kernels one by one (HIP events around single launches) at the benchmark shape + check EVERY (frame, head) vs torch.
Environment: B (3), S (301), T (30), FWD_ONLY=1.  (TCOW_LIB=<another build> compares builds on one box.)"""
import os, sys, torch
sys.path.insert(0, '.')
from tcow_amd import ops
dev = 'cuda'
B, T, S, heads = int(os.environ.get('B', '3')), int(os.environ.get('T', '30')), int(os.environ.get('S', '301')), 12; D = heads * 64; M = B * T * S
torch.manual_seed(0)
qkv = torch.randn(M, 3 * D, device=dev).bfloat16(); out = torch.empty(M, D, device=dev, dtype=torch.bfloat16); lse = torch.empty(M, heads, device=dev)
dout = torch.randn(M, D, device=dev).bfloat16(); dqkv = torch.empty(M, 3 * D, device=dev, dtype=torch.bfloat16)
shape = ops.attn_shape(ops.BF16, B, T, S, D, heads, 1)
def bench(f, n=20, w=5):
    for _ in range(w): f()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
out.fill_(float('nan')); lse.fill_(float('nan'))
ops.attn_fwd(shape, True, qkv, out, lse); torch.cuda.synchronize()
# correctness of the forward on every frame / head against torch f32 (+ log-sum-exp)
x = qkv.float().reshape(B * T, S, 3, heads, 64)
q, k, v = [x[:, :, i].permute(0, 2, 1, 3) for i in range(3)]                   # (BT, h, S, 64)
sc = (q @ k.transpose(-1, -2)) * 0.125
ref = (sc.softmax(-1) @ v).permute(0, 2, 1, 3).reshape(M, D)
ref_lse = torch.logsumexp(sc, -1).permute(0, 2, 1).reshape(M, heads)
ef = (out.float() - ref).abs().max().item(); el = (lse - ref_lse).abs().max().item()
bad = (~torch.isfinite(out.float())).sum().item()
tf = bench(lambda: ops.attn_fwd(shape, True, qkv, out, lse))
msg = f'S={S} B={B} T={T}: spatial fwd {tf:.1f} us | max|d| out {ef:.2e} lse {el:.2e} non-finite {bad} (ref max {ref.abs().max().item():.2f})'
if not os.environ.get('FWD_ONLY'):
    tb = bench(lambda: ops.attn_bwd(shape, True, qkv, out, dout, lse, dqkv))
    xg = qkv.float().reshape(B, T, S, 3, heads, 64)[0, 0].requires_grad_(True)       # S,3,h,64
    q1, k1, v1 = [xg[:, i].permute(1, 0, 2) for i in range(3)]
    o1 = ((q1 @ k1.transpose(-1, -2)) * 0.125).softmax(-1) @ v1
    (o1.permute(1, 0, 2).reshape(S, D) * dout[:S].float()).sum().backward()
    eb = (dqkv[:S].float().reshape(S, 3, heads, 64) - xg.grad).abs().max().item()
    msg += f' | bwd {tb:.1f} us max|d| {eb:.2e} (ref max {xg.grad.abs().max().item():.2f})'
print(msg, flush=True)
