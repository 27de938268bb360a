"""Reference-ceiling probe (not part of the product): times vendor-library GEMM / attention through
torch on the GPU box at the Seeker shapes, so the hand-written kernels have a same-hardware yardstick."""
import torch, time, json
dev = 'cuda'
def bench(f, n=20, w=5):
    for _ in range(w): f()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e-3
out = {}
for dt in (torch.bfloat16, torch.float32):
    for (M, K, N) in [(9030, 768, 2304), (9030, 768, 768), (9030, 768, 3072), (9030, 3072, 768), (27090, 768, 2304), (27090, 768, 768), (27090, 768, 3072), (27090, 3072, 768), (8192, 8192, 8192)]:
        if dt == torch.float32 and M > 10000: continue
        a = torch.randn(M, K, device=dev, dtype=dt); b = torch.randn(N, K, device=dev, dtype=dt)
        t = bench(lambda: a @ b.t())
        out[f'gemm_nt_{str(dt)[6:]}_{M}x{K}x{N}'] = dict(us=t * 1e6, tflops=2 * M * K * N / t / 1e12)
        print(f'gemm NT {dt} {M}x{K}x{N}: {t*1e6:.1f} us {2*M*K*N/t/1e12:.1f} TF', flush=True)
    # TN (weight grad): dW[N,K] = dY[M,N]^T X[M,K]
    for (M, K, N) in [(27090, 768, 768), (27090, 768, 3072)]:
        if dt == torch.float32: continue
        x = torch.randn(M, K, device=dev, dtype=dt); dy = torch.randn(M, N, device=dev, dtype=dt)
        t = bench(lambda: dy.t() @ x)
        print(f'gemm TN {dt} {M}x{K}x{N}: {t*1e6:.1f} us {2*M*K*N/t/1e12:.1f} TF', flush=True)
import torch.nn.functional as F
for (Bh, S, d) in [(30 * 12, 301, 64), (90 * 12, 301, 64), (300 * 12, 30, 64), (60 * 12, 1201, 64)]:
    q = torch.randn(1, Bh, S, d, device=dev, dtype=torch.bfloat16); k = torch.randn_like(q); v = torch.randn_like(q)
    try:
        t = bench(lambda: F.scaled_dot_product_attention(q, k, v))
        print(f'sdpa bf16 Bh={Bh} S={S}: {t*1e6:.1f} us {4*Bh*S*S*d/t/1e12:.1f} TF  {4*Bh*S*d*2/t/1e9:.0f} GB/s', flush=True)
    except Exception as e:
        print('sdpa failed', e)
x = torch.empty(1 << 28, device=dev, dtype=torch.float32); y = torch.empty_like(x)
t = bench(lambda: y.copy_(x)); print(f'torch copy 1GiB: {2*x.numel()*4/t/1e12:.2f} TB/s')
json.dump(out, open('gpurun_out/torch_ref.json', 'w'), indent=1)
