"""Dev: LayerNorm forward / backward at the benchmark shape (27 090 x 768), effective HBM bandwidth."""
import sys, torch
sys.path.insert(0, '.')
from tcow_amd import ops
dev = 'cuda'; M, D = 27090, 768
torch.manual_seed(0)
x = torch.randn(M, D, device=dev); w = torch.rand(D, device=dev) + 0.5; b = torch.randn(D, device=dev)
y = torch.empty(M, D, device=dev, dtype=torch.bfloat16); mu = torch.empty(M, device=dev); rs = torch.empty(M, device=dev)
dy = torch.randn(M, D, device=dev).bfloat16(); dres = torch.randn(M, D, device=dev); dx = torch.empty(M, D, device=dev); dxc = torch.empty(M, D, device=dev, dtype=torch.bfloat16)
dg = torch.empty(D, device=dev); db = torch.empty(D, device=dev); cs = torch.rand(M, device=dev)
def bench(f, n=30, w_=8):
    for _ in range(w_): f()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
ops.layernorm_fwd(ops.BF16, x, w, b, y, mu, rs)
tf = bench(lambda: ops.layernorm_fwd(ops.BF16, x, w, b, y, mu, rs))
tb = bench(lambda: ops.layernorm_bwd(ops.BF16, dy, x, mu, rs, w, dres, dx, dg, db, dx_cast=dxc, cast_scale=cs))
cso = torch.empty(D, device=dev); ss = torch.rand(M, device=dev)
tc = bench(lambda: ops.layernorm_bwd(ops.BF16, dy, x, mu, rs, w, dres, dx, dg, db, dx_cast=dxc, cast_scale=cs, colsum_out=cso, colsum_scale=ss))
tn = bench(lambda: ops.layernorm_bwd(ops.BF16, dy, x, mu, rs, w, dres, dx, dg, db))
print(f'ln bwd csum {tc:.1f} us, ln bwd no cast {tn:.1f} us ({M*D*14/tn/1e6:.2f} TB/s)')
print(f'ln fwd {tf:.1f} us ({M*D*6/tf/1e6:.2f} TB/s)   ln bwd + cast + param-grad fold {tb:.1f} us ({M*D*16/tb/1e6:.2f} TB/s)')
