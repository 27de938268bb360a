"""Dev: what do held CUs cost the step's GEMMs?  k single-wave spin kernels (torch.cuda._sleep on k side streams) hold k CU slots the way a
collective's workgroups would while a chain of the path's NT GEMMs runs on the main stream.  The 320 x 256 kernel needs a whole CU per
workgroup (144 KiB LDS, the full register file), so any held CU pushes a 255-tile launch into a second round; TILE=128 / 256 (environment of THIS script) forces
the square kernels through tcow_gemm_args.tile, TILE=0 (default) is the library's routing."""
import os, sys, torch
sys.path.insert(0, '.')
from tcow_amd import ops
dev = 'cuda'; M = 27090; D = 768; TILE = int(os.environ.get('TILE', '0'))
torch.manual_seed(0)
A = torch.randn(M, D, device=dev).bfloat16(); A4 = torch.randn(M, 4 * D, device=dev).bfloat16()
W = torch.randn(D, D, device=dev).bfloat16(); W4 = torch.randn(D, 4 * D, device=dev).bfloat16(); W4b = torch.randn(4 * D, D, device=dev).bfloat16()
R = torch.randn(M, D, device=dev); O = torch.empty(M, D, device=dev); Ob = torch.empty(M, D, device=dev, dtype=torch.bfloat16); O4 = torch.empty(M, 4 * D, device=dev, dtype=torch.bfloat16)
def chain():
    for _ in range(4):
        ops.gemm_nt(ops.BF16, A, W, O, resid=R, tile=TILE); ops.gemm_nt(ops.BF16, A4, W4, Ob, tile=TILE); ops.gemm_nt(ops.BF16, A, W4b, O4, tile=TILE)
streams = [torch.cuda.Stream() for _ in range(64)]
def run(k, spin_cycles=6_000_000):
    torch.cuda.synchronize()
    for s in streams[:k]:
        with torch.cuda.stream(s): torch.cuda._sleep(spin_cycles)          # ~3 ms each: longer than the chain
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); chain(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3
for _ in range(3): chain()
base = min(run(0) for _ in range(3))
print(f'tile {TILE}: 12 NT GEMMs, nothing held: {base:.0f} us')
for k in (1, 8, 16, 32, 64):
    t = min(run(k) for _ in range(3))
    print(f'   {k:2d} CU slots held: {t:.0f} us  ({t / base:.2f}x)')
