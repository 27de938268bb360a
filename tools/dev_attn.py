"""Dev harness (GPU box): bf16 MFMA attention kernels vs a plain torch fp32 reference, forward + backward, plus timing."""
import sys, torch, ctypes
sys.path.insert(0, '.')
from tcow_amd import ops
dev = 'cuda'
torch.manual_seed(0)

def ref_attn(qkv, B, T, S, D, heads, ca, spatial):
    x = qkv.float().reshape(B, T, S, 3, heads, 64)
    if spatial:
        s0 = 0 if ca in (0, 1) else 1
        q, k, v = [x[:, :, s0:, i].permute(0, 1, 3, 2, 4) for i in range(3)]        # B,T,h,L,d
        a = (q @ k.transpose(-1, -2)) * 0.125
        o = a.softmax(-1) @ v                                                           # B,T,h,L,d
        out = torch.zeros(B, T, S, heads, 64, device=qkv.device)
        out[:, :, s0:] = o.permute(0, 1, 3, 2, 4)
    else:
        q, k, v = [x[:, :, 1:, i].permute(0, 2, 3, 1, 4) for i in range(3)]         # B,N,h,T,d
        a = (q @ k.transpose(-1, -2)) * 0.125
        if ca > 0:
            keep = torch.ones(T, T, dtype=torch.bool, device=qkv.device).tril(0 if ca <= 2 else ca - 2)
            a = a.masked_fill(~keep, -1e10)
        o = a.softmax(-1) @ v                                                           # B,N,h,T,d
        out = torch.zeros(B, T, S, heads, 64, device=qkv.device)
        out[:, :, 1:] = o.permute(0, 3, 1, 2, 4)
    return out.reshape(B * T * S, D)

def bench(f, n=10, w=3):
    for _ in range(w): f()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e-3

def run(B, T, S, heads, ca, spatial, time_it=False):
    D = heads * 64; M = B * T * S
    qkv = (torch.randn(M, 3 * D, device=dev) * 1.0).bfloat16()
    shape = ops.attn_shape(ops.BF16, B, T, S, D, heads, ca)
    out = torch.empty(M, D, device=dev, dtype=torch.bfloat16); lse = torch.empty(M, heads, device=dev)
    ops.attn_fwd(shape, spatial, qkv, out, lse)
    q32 = qkv.float().requires_grad_(True)
    ref = ref_attn(q32, B, T, S, D, heads, ca, spatial)
    e_f = (out.float() - ref).abs().max().item()
    dout = torch.randn(M, D, device=dev).bfloat16()
    # rows the kernels never touch must not carry gradient in the reference either
    (ref * dout.float()).sum().backward()
    dqkv = torch.empty(M, 3 * D, device=dev, dtype=torch.bfloat16)
    ops.attn_bwd(shape, spatial, qkv, out, dout, lse, dqkv)
    gr = q32.grad
    e_b = (dqkv.float() - gr).abs().max().item(); sc = gr.abs().max().item()
    parts = [(dqkv.float()[:, i * D:(i + 1) * D] - gr[:, i * D:(i + 1) * D]).abs().max().item() for i in range(3)]
    msg = f'{"spatial" if spatial else "temporal"} B={B} T={T} S={S} h={heads} ca={ca}: fwd max|d|={e_f:.3e} (ref max {ref.abs().max().item():.2f}) bwd max|d|={e_b:.3e} dq/dk/dv={parts[0]:.2e}/{parts[1]:.2e}/{parts[2]:.2e} (ref max {sc:.2f})'
    if time_it:
        tf = bench(lambda: ops.attn_fwd(shape, spatial, qkv, out, lse)); tb = bench(lambda: ops.attn_bwd(shape, spatial, qkv, out, dout, lse, dqkv))
        L = S if spatial else T; n_seq = B * T * heads if spatial else B * (S - 1) * heads
        fl = 4.0 * n_seq * L * L * 64
        byts = 4.0 * M * D * 2
        msg += f' | fwd {tf*1e6:.1f} us ({fl/tf/1e12:.1f} TF, {byts/tf/1e9:.0f} GB/s) bwd {tb*1e6:.1f} us ({2.5*fl/tb/1e12:.1f} TF)'
    print(msg, flush=True)

for ca in (1, 0, 3):
    run(1, 4, 17, 4, ca, False)
    run(2, 30, 21, 2, ca, False)
run(1, 40, 9, 2, 1, False)      # T > 32: two tiles
run(1, 60, 9, 2, 2, False)
for ca in (1, 2):
    run(1, 2, 17, 4, ca, True)
    run(2, 3, 301, 2, ca, True)
    run(1, 2, 77, 2, ca, True)
run(1, 30, 301, 12, 1, False, True)
run(1, 30, 301, 12, 1, True, True)
run(3, 30, 301, 12, 1, False, True)
run(3, 30, 301, 12, 1, True, True)
run(1, 6, 1201, 12, 1, True, True)
run(1, 2, 333, 2, 2, True)
