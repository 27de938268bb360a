"""Dev: how fast can the temporal access pattern be streamed at all?  A (b, t, s) -> (b, s, t) gather of the qkv rows (what the reference's
physical transposes do, vit.py:170) with torch's copy kernel, next to a plain contiguous copy of the same bytes."""
import sys, torch
dev = 'cuda'; B, T, S, D3 = 3, 30, 301, 2304
x = torch.randn(B, T, S, D3, device=dev).bfloat16()
def bench(f, n=20, w=5):
    for _ in range(w): f()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
y = torch.empty(B, S, T, D3, device=dev, dtype=torch.bfloat16); z = torch.empty_like(x)
t1 = bench(lambda: y.copy_(x.transpose(1, 2))); t0 = bench(lambda: z.copy_(x))
nb = x.numel() * 2 * 2
print(f'contiguous copy {t0:.1f} us ({nb / t0 / 1e6:.2f} TB/s)   (b,t,s)->(b,s,t) row gather {t1:.1f} us ({nb / t1 / 1e6:.2f} TB/s)')
# per-head pieces: 128-byte segments at stride S rows, like one (site, head) sequence reads them
xh = x.view(B, T, S, 36, 64); yh = torch.empty(B, S, 36, T, 64, device=dev, dtype=torch.bfloat16)
t2 = bench(lambda: yh.copy_(xh.permute(0, 2, 3, 1, 4)))
print(f'(b,t,s,h,64) -> (b,s,h,t,64) 128-byte pieces {t2:.1f} us ({nb / t2 / 1e6:.2f} TB/s)')
