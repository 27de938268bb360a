"""Dev harness (GPU box): correctness + timing of libtcow_hip GEMMs against torch. Not a test; see tests/."""
import ctypes, sys, torch, json
sys.path.insert(0, '.')
from tcow_amd import _lib as L
lib = L.lib()
dev = 'cuda'
import os
DBG = os.environ.get('DBG', '0') == '1'
def SYNC():
    if DBG: torch.cuda.synchronize()
def stream(): return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
def p(t): return ctypes.c_void_p(t.data_ptr()) if t is not None else None
def gemm_nt(A, W, bias=None, row_scale=None, resid=None, act=0, aux=None, out_f32=False):
    M, K = A.shape; N = W.shape[0]
    dt = L.TCOW_BF16 if A.dtype == torch.bfloat16 else L.TCOW_F32
    C = torch.empty(M, N, device=dev, dtype=torch.float32 if (out_f32 or dt == L.TCOW_F32) else torch.bfloat16)
    a = L.GemmArgs(M, N, K, dt, p(A), A.stride(0), p(W), W.stride(0), p(C), C.stride(0), int(out_f32 or dt == L.TCOW_F32), p(bias), p(row_scale), p(resid),
                   resid.stride(0) if resid is not None else 0, act, p(aux), aux.stride(0) if aux is not None else 0)
    L.check(lib.tcow_gemm_nt(stream(), ctypes.byref(a)), 'gemm_nt'); SYNC(); return C
def gemm_tn(dY, X, bias_grad=False, accumulate=False, dW=None, db=None):
    M, N = dY.shape; K = X.shape[1]
    dt = L.TCOW_BF16 if dY.dtype == torch.bfloat16 else L.TCOW_F32
    wsb = lib.tcow_gemm_tn_workspace_bytes(M, N, K); ws = torch.empty(wsb, device=dev, dtype=torch.uint8)
    if dW is None: dW = torch.empty(N, K, device=dev, dtype=torch.float32)
    if bias_grad and db is None: db = torch.empty(N, device=dev, dtype=torch.float32)
    L.check(lib.tcow_gemm_tn(stream(), dt, M, N, K, p(dY), dY.stride(0), p(X), X.stride(0), p(dW), dW.stride(0), p(db) if bias_grad else None, int(accumulate), p(ws), wsb), 'gemm_tn'); SYNC()
    return dW, db
def bench(f, n=20, w=5):
    for _ in range(w): f()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e-3
torch.manual_seed(0)
ok = True
def rel(a, b): return ((a.float() - b.float()).abs().max() / (b.float().abs().max() + 1e-9)).item()
for dt in (torch.bfloat16, torch.float32):
    for (M, K, N) in [(300, 64, 64), (1000, 256, 192), (130, 128, 48), (9030, 768, 768), (2057, 1024, 256)]:
        A = torch.randn(M, K, device=dev, dtype=dt); W = torch.randn(N, K, device=dev, dtype=dt) * 0.05
        bias = torch.randn(N, device=dev); rs = torch.rand(M, device=dev) + 0.5; resid = torch.randn(M, N, device=dev)
        ref0 = (A.double() @ W.double().t())
        C = gemm_nt(A, W); e = rel(C, ref0); print(f'nt plain {dt} {M}x{K}x{N} rel={e:.2e}'); ok &= e < (2e-2 if dt == torch.bfloat16 else 1e-5)
        ref = ((ref0 + bias.double()) * rs.double()[:, None])
        C = gemm_nt(A, W, bias=bias, row_scale=rs, resid=resid, out_f32=True); e = rel(C, ref + resid.double()); print(f'   bias+scale+resid rel={e:.2e}'); ok &= e < (1e-2 if dt == torch.bfloat16 else 1e-5)
        aux = torch.empty(M, N, device=dev, dtype=dt)
        C = gemm_nt(A, W, bias=bias, act=1, aux=aux); r2 = torch.nn.functional.gelu(ref0 + bias.double()); e = rel(C, r2); e2 = rel(aux, ref0 + bias.double()); print(f'   gelu rel={e:.2e} aux rel={e2:.2e}'); ok &= e < (2e-2 if dt == torch.bfloat16 else 1e-5)
        pre = torch.randn(M, N, device=dev, dtype=dt)
        C = gemm_nt(A, W, act=2, aux=pre); x = pre.double().requires_grad_(True); g = torch.autograd.grad(torch.nn.functional.gelu(x).sum(), x)[0]; e = rel(C, ref0 * g); print(f'   dgelu rel={e:.2e}'); ok &= e < (2e-2 if dt == torch.bfloat16 else 1e-5)
    for (M, N, K) in [(300, 64, 64), (1000, 192, 256), (9030, 768, 768), (2057, 48, 256), (9030, 3072, 768)]:
        dY = torch.randn(M, N, device=dev, dtype=dt); X = torch.randn(M, K, device=dev, dtype=dt)
        dW, db = gemm_tn(dY, X, bias_grad=True); ref = dY.double().t() @ X.double(); e = rel(dW, ref); e2 = rel(db, dY.double().sum(0)); print(f'tn {dt} {M}x{N}x{K} rel={e:.2e} bias rel={e2:.2e}'); ok &= e < (1e-2 if dt == torch.bfloat16 else 1e-5) and e2 < 1e-3
        dW2, db2 = gemm_tn(dY, X, bias_grad=True, accumulate=True, dW=dW.clone(), db=db.clone()); e = rel(dW2, 2 * ref); print(f'   accumulate rel={e:.2e}'); ok &= e < 1e-2
print('ALL OK' if ok else 'SOME FAILED')
res = {}
for (M, K, N) in [(9030, 768, 2304), (9030, 768, 768), (9030, 768, 3072), (9030, 3072, 768), (27090, 768, 2304), (27090, 768, 768), (27090, 768, 3072), (27090, 3072, 768), (8192, 8192, 8192)]:
    A = torch.randn(M, K, device=dev, dtype=torch.bfloat16); W = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
    t = bench(lambda: gemm_nt(A, W)); t2 = bench(lambda: A @ W.t())
    print(f'NT bf16 {M}x{K}x{N}: ours {t*1e6:.1f} us {2*M*K*N/t/1e12:.0f} TF | torch {t2*1e6:.1f} us {2*M*K*N/t2/1e12:.0f} TF', flush=True)
for (M, N, K) in [(27090, 768, 768), (27090, 3072, 768), (27090, 768, 3072), (27090, 2304, 768), (9030, 768, 768)]:
    dY = torch.randn(M, N, device=dev, dtype=torch.bfloat16); X = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    wsb = lib.tcow_gemm_tn_workspace_bytes(M, N, K); ws = torch.empty(wsb, device=dev, dtype=torch.uint8); dW = torch.empty(N, K, device=dev)
    f = lambda: L.check(lib.tcow_gemm_tn(stream(), 1, M, N, K, p(dY), dY.stride(0), p(X), X.stride(0), p(dW), dW.stride(0), None, 0, p(ws), wsb))
    t = bench(f); t2 = bench(lambda: dY.t() @ X)
    print(f'TN bf16 {M}x{N}x{K}: ours {t*1e6:.1f} us {2*M*K*N/t/1e12:.0f} TF | torch {t2*1e6:.1f} us {2*M*K*N/t2/1e12:.0f} TF', flush=True)
for (M, K, N) in [(9030, 768, 2304), (9030, 3072, 768)]:
    A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev)
    t = bench(lambda: gemm_nt(A, W), n=5, w=2); print(f'NT f32 {M}x{K}x{N}: ours {t*1e6:.1f} us {2*M*K*N/t/1e12:.1f} TF')
