"""GPU: every libtcow_hip entry point, called through the C ABI, against a plain PyTorch fp32 reference of the same op.
Tolerances: f32 mode ~1e-5 relative (exact-f32 MFMA/FMA, different summation order); bf16 mode ~1e-2 relative of the
tensor's max (bf16 operands, f32 accumulation); fp16 = the same kernels built for binary16 storage (libtcow_hip_fp16.so): 1/8 of the
bf16 bounds (11 instead of 8 significand bits)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import load_golden

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = a.detach(), b.detach()
    return float((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-12))


@pytest.fixture(scope='module')
def ops(cuda):
    from tcow_amd import ops as o
    return o


MODES = [('f32', 2e-5), ('bf16', 2e-2), ('fp16', 2.5e-3)]
GEMM_MODES = MODES + [('f32x3', 4e-5)]      # f32 tensors, bf16 x 3 split products (csrc/gemm_x3.hip): ~2^-17 per product


def _mode(ops, name):
    return {'f32': (ops.F32, torch.float32), 'f32x3': (ops.F32X3, torch.float32), 'bf16': (ops.BF16, torch.bfloat16), 'fp16': (ops.FP16, torch.float16)}[name]


@pytest.mark.parametrize('mname,tol', GEMM_MODES)
@pytest.mark.parametrize('M,K,N', [(1, 64, 64), (300, 64, 192), (130, 128, 48), (1000, 1024, 256), (4097, 768, 768)])
def test_gemm_nt_epilogues(ops, cuda, mname, tol, M, K, N):
    mode, dt = _mode(ops, mname)
    g = torch.Generator(device='cuda').manual_seed(M + N)
    A = torch.randn(M, K, device=cuda, generator=g).to(dt); W = (torch.randn(N, K, device=cuda, generator=g) * 0.05).to(dt)
    bias = torch.randn(N, device=cuda, generator=g); rs = torch.rand(M, device=cuda, generator=g) + 0.5; resid = torch.randn(M, N, device=cuda, generator=g)
    ref0 = A.double() @ W.double().t()
    C = ops.gemm_nt(mode, A, W, torch.empty(M, N, device=cuda, dtype=dt))
    assert rel(C, ref0) < tol
    C = ops.gemm_nt(mode, A, W, torch.empty(M, N, device=cuda), bias=bias, row_scale=rs, resid=resid)
    assert rel(C, (ref0 + bias.double()) * rs.double()[:, None] + resid.double()) < tol
    inplace = resid.clone()                                                   # residual may alias the f32 output
    ops.gemm_nt(mode, A, W, inplace, bias=bias, resid=inplace)
    assert rel(inplace, ref0 + bias.double() + resid.double()) < tol
    # second bias with its own row scale (the folded temporal projection: mask0 * (s (O W'^T + b') + b_fc))
    b2 = torch.randn(N, device=cuda, generator=g); rs2 = (torch.rand(M, device=cuda, generator=g) > 0.3).float()
    want2 = (ref0 + bias.double()) * rs.double()[:, None] + rs2.double()[:, None] * b2.double() + resid.double()
    C = ops.gemm_nt(mode, A, W, torch.empty(M, N, device=cuda), bias=bias, row_scale=rs, resid=resid, bias2=b2, row_scale2=rs2)
    assert rel(C, want2) < tol
    C = ops.gemm_nt(mode, A, W, torch.empty(M, N, device=cuda, dtype=dt), bias=bias, bias2=b2)
    assert rel(C, ref0 + bias.double() + b2.double()) < tol
    aux = torch.empty(M, N, device=cuda, dtype=dt)
    C = ops.gemm_nt(mode, A, W, torch.empty(M, N, device=cuda, dtype=dt), bias=bias, act=ops.ACT_GELU, aux=aux)
    assert rel(C, F.gelu(ref0 + bias.double())) < tol and rel(aux, ref0 + bias.double()) < tol
    pre = torch.randn(M, N, device=cuda, generator=g).to(dt)
    x = pre.double().requires_grad_(True)
    dg = torch.autograd.grad(F.gelu(x).sum(), x)[0]
    C = ops.gemm_nt(mode, A, W, torch.empty(M, N, device=cuda, dtype=dt), act=ops.ACT_DGELU, aux=pre)
    assert rel(C, ref0 * dg) < tol
    # the training pair: forward saves GELU'(v) next to GELU(v), backward multiplies by it
    C = ops.gemm_nt(mode, A, W, torch.empty(M, N, device=cuda, dtype=dt), bias=bias, act=ops.ACT_GELU_DSAVE, aux=aux)
    xv = (ref0 + bias.double()).requires_grad_(True)
    dgv = torch.autograd.grad(F.gelu(xv).sum(), xv)[0]
    assert rel(C, F.gelu(xv.detach())) < tol and rel(aux, dgv) < tol
    C = ops.gemm_nt(mode, A, W, torch.empty(M, N, device=cuda, dtype=dt), act=ops.ACT_MUL_AUX, aux=pre)
    assert rel(C, ref0 * pre.double()) < tol


@pytest.mark.parametrize('M,K,N', [(4099, 3072, 768), (4200, 1024, 1288), (300, 128, 260), (2570, 64 * 5, 516)])
def test_gemm_nt_phase_kernel_ragged_shapes(ops, cuda, M, K, N):
    """The 160 x 256 two-per-CU kernel (gemm_nt_c2.hip) and the 320 x 256 kernel on shapes that do not fill their tiles (row / column guards, partial last
    tiles, short K) against an f64 product."""
    H16, h16 = _mode(ops, 'bf16')
    g = torch.Generator(device='cuda').manual_seed(M + K + N)
    A = torch.randn(M, K, device=cuda, generator=g).to(h16); W = (torch.randn(N, K, device=cuda, generator=g) * 0.05).to(h16)
    bias = torch.randn(N, device=cuda, generator=g)
    ref = A.double() @ W.double().t() + bias.double()
    for tile in (320, 160):
        if tile == 160 and K % 128:                                        # the two-per-CU kernel walks K two 64-wide tiles at a time
            continue
        out = torch.full((M + 1, N), 7.0, device=cuda)                     # a guard row behind the output: nothing may be written past M
        ops.gemm_nt(H16, A, W, out[:M], bias=bias, tile=tile)
        assert rel(out[:M], ref) < 2e-5 and bool((out[M] == 7.0).all())
        assert rel(ops.gemm_nt(H16, A, W, torch.empty(M, N, device=cuda, dtype=h16), bias=bias, tile=tile), ref) < 4e-3


@pytest.mark.parametrize('fmt', ['bf16', 'fp16'])
@pytest.mark.parametrize('N,K', [(768, 768), (2304, 768), (3072, 768), (768, 3072)])
def test_gemm_nt_every_tile_kernel_at_bench_size(ops, cuda, N, K, fmt):
    """The benchmarked instantiations against an f64 product: M = 27 090 token rows (3 queries x 30 frames x 301 slots, BASELINE
    configs[1]) with the four weight shapes of a block, on each bf16 tile kernel (forced through tcow_gemm_args.tile: 320 x 256
    incl. its compile-time epilogues, 256 x 256, 128 x 128) and every epilogue combination the engine issues.  Operands are bf16
    values, so the only error is f32 accumulation order + the output rounding: bf16 outputs within 2^-8 of the tile maximum, f32
    outputs within 2e-5."""
    M = 27090
    H16, h16 = _mode(ops, fmt)
    g = torch.Generator(device='cuda').manual_seed(N + K)
    A = torch.randn(M, K, device=cuda, generator=g).to(h16); W = (torch.randn(N, K, device=cuda, generator=g) * 0.05).to(h16)
    bias = torch.randn(N, device=cuda, generator=g); rs = torch.rand(M, device=cuda, generator=g) + 0.5; rs[::7] = 0.0
    resid = torch.randn(M, N, device=cuda, generator=g); pre = torch.randn(M, N, device=cuda, generator=g).to(h16)
    b2 = torch.randn(N, device=cuda, generator=g); rs2 = (torch.rand(M, device=cuda, generator=g) > 0.3).float()
    ref0 = A.double() @ W.double().t()
    refb = ref0 + bias.double()
    xv = refb.clone().requires_grad_(True)
    gel = F.gelu(xv); dgel = torch.autograd.grad(gel.sum(), xv)[0]; gel = gel.detach()
    BF, F32 = (4e-3 if fmt == 'bf16' else 5e-4), 2e-5
    bf = lambda: torch.empty(M, N, device=cuda, dtype=h16)
    f32 = lambda: torch.empty(M, N, device=cuda)
    for tile in (320, 160, 256, 128, 0):                                                                               # 160: the two-workgroups-per-CU kernel (gemm_nt_c2.hip)
        t = dict(tile=tile)
        assert rel(ops.gemm_nt(H16, A, W, bf(), bias=bias, **t), refb) < BF                                              # EpiCfg<NONE, 0>: qkv, proj-input grads
        assert rel(ops.gemm_nt(H16, A, W, f32(), **t), ref0) < F32                                                     # f32 output (dFeat)
        assert rel(ops.gemm_nt(H16, A, W, bf(), bias=bias, row_scale=rs, **t), refb * rs.double()[:, None]) < BF         # <NONE, 1>: temporal proj + DropPath row scale
        assert rel(ops.gemm_nt(H16, A, W, f32(), bias=bias, resid=resid, **t), refb + resid.double()) < F32             # <NONE, 2>: residual
        assert rel(ops.gemm_nt(H16, A, W, f32(), bias=bias, row_scale=rs, resid=resid, **t), refb * rs.double()[:, None] + resid.double()) < F32   # <NONE, 3>
        inplace = resid.clone()
        ops.gemm_nt(H16, A, W, inplace, bias=bias, row_scale=rs, resid=inplace, **t)                                    # eval: residual aliases the output
        assert rel(inplace, refb * rs.double()[:, None] + resid.double()) < F32
        if N == 768 and K == 768:                                                                                       # <NONE, 7>: the folded temporal projection
            want7 = refb * rs.double()[:, None] + rs2.double()[:, None] * b2.double() + resid.double()
            assert rel(ops.gemm_nt(H16, A, W, f32(), bias=bias, row_scale=rs, resid=resid, bias2=b2, row_scale2=rs2, **t), want7) < F32
            inplace = resid.clone()
            ops.gemm_nt(H16, A, W, inplace, bias=bias, row_scale=rs, resid=inplace, bias2=b2, row_scale2=rs2, **t)
            assert rel(inplace, want7) < F32
        aux = bf()
        assert rel(ops.gemm_nt(H16, A, W, bf(), bias=bias, act=ops.ACT_GELU_DSAVE, aux=aux, **t), gel) < BF and rel(aux, dgel) < BF   # <GELU_DSAVE, 0>: fc1 (training)
        assert rel(ops.gemm_nt(H16, A, W, bf(), bias=bias, act=ops.ACT_GELU, **t), gel) < BF                            # <GELU, 0>: fc1 (inference)
        assert rel(ops.gemm_nt(H16, A, W, bf(), act=ops.ACT_MUL_AUX, aux=pre, **t), ref0 * pre.double()) < BF           # <MUL_AUX, 0>: fc2 input gradient x GELU'


@pytest.mark.parametrize('mname,tol', GEMM_MODES)
@pytest.mark.parametrize('M,N,K', [(5, 64, 64), (300, 192, 256), (2057, 48, 1024), (9030, 768, 768),
                                   (9030, 2304, 768), (4200, 1032, 1288), (4099, 3072, 768)])     # the last three take the 256-tile kernel (incl. ragged tiles / last slice)
def test_gemm_tn_weight_and_bias_grad(ops, cuda, mname, tol, M, N, K):
    mode, dt = _mode(ops, mname)
    g = torch.Generator(device='cuda').manual_seed(M)
    dY = torch.randn(M, N, device=cuda, generator=g).to(dt); X = torch.randn(M, K, device=cuda, generator=g).to(dt)
    dW = torch.empty(N, K, device=cuda); db = torch.empty(N, device=cuda)
    ops.gemm_tn(mode, dY, X, dW, bias_grad=db)
    ref = dY.double().t() @ X.double()
    assert rel(dW, ref) < max(tol / 4, 1e-5) and rel(db, dY.double().sum(0)) < 1e-4
    ops.gemm_tn(mode, dY, X, dW, bias_grad=db, accumulate=True)
    assert rel(dW, 2 * ref) < max(tol / 4, 1e-5) and rel(db, 2 * dY.double().sum(0)) < 1e-4


@pytest.mark.parametrize('M,K,N', [(27090, 768, 768), (27090, 3072, 768), (333, 100, 70), (257, 50, 129), (5, 8, 3), (1, 4, 1)])
def test_gemm_x3_split_products(ops, cuda, M, K, N):
    """TCOW_F32X3 at the benchmark's row count and on ragged / unaligned shapes (K not a multiple of the 32-wide k-slice, K % 4 != 0:
    scalar loader; row views with an odd pitch) against the f64 product, next to the exact-f32 kernel on the same operands:
    the split products are within 4e-5 of the result's maximum (f32 kernel: 2e-5), both forms, every epilogue operand in play."""
    g = torch.Generator(device='cuda').manual_seed(M + K)
    Ab = torch.randn(M, K + 3, device=cuda, generator=g); A = Ab[:, 1:K + 1] if K % 4 else Ab[:, :K]       # odd pitch / unaligned start
    W = torch.randn(N, K, device=cuda, generator=g) * 0.05
    bias = torch.randn(N, device=cuda, generator=g); rs = torch.rand(M, device=cuda, generator=g) + 0.5; resid = torch.randn(M, N, device=cuda, generator=g)
    ref = (A.double() @ W.double().t() + bias.double()) * rs.double()[:, None] + resid.double()
    c3 = ops.gemm_nt(ops.F32X3, A, W, torch.empty(M, N, device=cuda), bias=bias, row_scale=rs, resid=resid)
    c1 = ops.gemm_nt(ops.F32, A, W, torch.empty(M, N, device=cuda), bias=bias, row_scale=rs, resid=resid)
    assert rel(c3, ref) < 4e-5 and rel(c1, ref) < 2e-5
    dY = torch.randn(M, N, device=cuda, generator=g)
    dW = ops.gemm_tn(ops.F32X3, dY, A, torch.empty(N, K, device=cuda))
    assert rel(dW, dY.double().t() @ A.double()) < 2e-5
    # structure of the error: a plain bf16 product of the same operands is two orders of magnitude further away
    cb = ops.gemm_nt(ops.BF16, A.contiguous().bfloat16(), W.bfloat16(), torch.empty(M, N, device=cuda)) if K % 64 == 0 else None
    if cb is not None:
        r0 = A.double() @ W.double().t()
        assert rel(cb, r0) > 30 * rel(ops.gemm_nt(ops.F32X3, A, W, torch.empty(M, N, device=cuda)), r0)


@pytest.mark.parametrize('fmt', ['bf16', 'fp16'])
@pytest.mark.parametrize('M', [27090, 9030, 300])
def test_gemm_tn_grouped_block_weights(ops, cuda, M, fmt):
    """tcow_gemm_tn_grouped on the seven Linear layers of a divided space-time block (vit.py:50-61,74-76,146) at the benchmark's row count
    (one grid, common slice count), at a smaller M and at one too small for the 256-tile kernel (the library loops there): every dW / db
    against the f64 product and against the one-by-one entry point."""
    D = 768
    H16, h16 = _mode(ops, fmt)
    shapes = [(D, 4 * D), (4 * D, D), (D, D), (3 * D, D), (D, D), (D, D), (3 * D, D)]      # (N, K) of fc2, fc1, proj, qkv, tfc, tproj, tqkv
    g = torch.Generator(device='cuda').manual_seed(M)
    probs, refs = [], []
    for i, (N, K) in enumerate(shapes):
        dY = torch.randn(M, N, device=cuda, generator=g).to(h16); X = torch.randn(M, K, device=cuda, generator=g).to(h16)
        dW = torch.empty(N, K, device=cuda); db = torch.empty(N, device=cuda) if i != 2 else None
        probs.append((dY, X, dW, db)); refs.append((dY.double().t() @ X.double(), dY.double().sum(0)))
    ops.gemm_tn_grouped(H16, probs)
    for (dY, X, dW, db), (rw, rb) in zip(probs, refs):
        assert rel(dW, rw) < 1e-5
        if db is not None:
            assert rel(db, rb) < 1e-4
        one = ops.gemm_tn(H16, dY, X, torch.empty_like(dW))
        assert rel(dW, one) < 1e-5
    # f32-storage modes go through the same entry point (library-side loop)
    small = [(p[0][:257].float(), p[1][:257].float(), torch.empty_like(p[2]), None) for p in probs[2:4]]
    ops.gemm_tn_grouped(ops.F32X3, small)
    for dY, X, dW, _ in small:
        assert rel(dW, dY.double().t() @ X.double()) < 2e-5


@pytest.mark.parametrize('D', [768, 256, 100])
def test_sgemm_batched_small_products(ops, cuda, D):
    """The small f32 products of the folded temporal projection, every operand form the engine issues (plain, B transposed, A transposed,
    matrix x vector both ways), several problems per launch; split-bf16 arithmetic: ~5e-6 relative."""
    g = torch.Generator(device='cuda').manual_seed(D)
    mats = [torch.randn(D, D, device=cuda, generator=g) * 0.05 for _ in range(6)]
    vecs = [torch.randn(D, device=cuda, generator=g) for _ in range(3)]
    out = lambda *sh: torch.full(sh, float('nan'), device=cuda)
    # W' = Wfc Wproj (three problems in one launch)
    C = [out(D, D) for _ in range(3)]
    ops.sgemm_batched([(mats[i], mats[i + 3], C[i]) for i in range(3)])
    for i in range(3):
        assert rel(C[i], mats[i].double() @ mats[i + 3].double()) < 2e-5
    # dWfc = dW' Wproj^T, dWproj = Wfc^T dW'
    C = [out(D, D) for _ in range(2)]
    ops.sgemm_batched([(mats[0], mats[1].t(), C[0]), (mats[2], mats[3].t(), C[1])])
    assert rel(C[0], mats[0].double() @ mats[1].double().t()) < 2e-5 and rel(C[1], mats[2].double() @ mats[3].double().t()) < 2e-5
    C = [out(D, D) for _ in range(2)]
    ops.sgemm_batched([(mats[0].t(), mats[1], C[0]), (mats[2].t(), mats[3], C[1])])
    assert rel(C[0], mats[0].double().t() @ mats[1].double()) < 2e-5 and rel(C[1], mats[2].double().t() @ mats[3].double()) < 2e-5
    # b' = Wfc b_proj, db_proj = Wfc^T db'
    c = [out(D) for _ in range(2)]
    ops.sgemm_batched([(mats[0], vecs[0].view(1, -1).t(), c[0].view(-1, 1)), (mats[1], vecs[1].view(1, -1).t(), c[1].view(-1, 1))])
    assert rel(c[0], mats[0].double() @ vecs[0].double()) < 2e-5 and rel(c[1], mats[1].double() @ vecs[1].double()) < 2e-5
    c = [out(D) for _ in range(2)]
    ops.sgemm_batched([(mats[0].t(), vecs[0].view(-1, 1), c[0].view(-1, 1)), (mats[1].t(), vecs[2].view(-1, 1), c[1].view(-1, 1))])
    assert rel(c[0], mats[0].double().t() @ vecs[0].double()) < 2e-5 and rel(c[1], mats[1].double().t() @ vecs[2].double()) < 2e-5
    # rank-1 update C += u v^T (dWfc's bias term)
    C = [mats[4].clone(), mats[5].clone()]
    ops.sgemm_batched([(vecs[0].view(-1, 1), vecs[1].view(1, -1), C[0]), (vecs[1].view(-1, 1), vecs[2].view(1, -1), C[1])], accumulate=True)
    assert rel(C[0], mats[4].double() + torch.outer(vecs[0].double(), vecs[1].double())) < 2e-5
    assert rel(C[1], mats[5].double() + torch.outer(vecs[1].double(), vecs[2].double())) < 2e-5


def test_gemm_rejects_bad_arguments(ops, cuda):
    from tcow_amd._lib import TcowError
    A = torch.zeros(8, 72, device=cuda, dtype=torch.bfloat16); W = torch.zeros(8, 72, device=cuda, dtype=torch.bfloat16)
    with pytest.raises(TcowError):                                            # K not a multiple of 64 in bf16 mode
        ops.gemm_nt(ops.BF16, A, W, torch.empty(8, 8, device=cuda, dtype=torch.bfloat16))
    with pytest.raises(TcowError):                                            # no silent CPU fallback
        ops.gemm_nt(ops.F32, torch.zeros(4, 4), torch.zeros(4, 4), torch.zeros(4, 4))


@pytest.mark.parametrize('mname,tol', MODES)
@pytest.mark.parametrize('rows,D', [(7, 64), (1000, 256), (9030, 768), (33, 1024)])
def test_layernorm_fwd_bwd(ops, cuda, mname, tol, rows, D):
    mode, dt = _mode(ops, mname)
    g = torch.Generator(device='cuda').manual_seed(rows)
    x = torch.randn(rows, D, device=cuda, generator=g) * 2 + 0.5
    w = torch.rand(D, device=cuda, generator=g) + 0.5; b = torch.randn(D, device=cuda, generator=g)
    y = torch.empty(rows, D, device=cuda, dtype=dt); mu = torch.empty(rows, device=cuda); rs = torch.empty(rows, device=cuda)
    ops.layernorm_fwd(mode, x, w, b, y, mu, rs)
    xd = x.double().requires_grad_(True); wd = w.double().requires_grad_(True); bd = b.double().requires_grad_(True)
    ref = F.layer_norm(xd, (D,), wd, bd, 1e-6)
    assert rel(y, ref) < tol and rel(mu, x.double().mean(1)) < 1e-5
    dy = torch.randn(rows, D, device=cuda, generator=g).to(dt); dres = torch.randn(rows, D, device=cuda, generator=g)
    gx, gw, gb = torch.autograd.grad((ref * dy.double()).sum(), [xd, wd, bd])
    dx = torch.empty(rows, D, device=cuda); dg = torch.empty(D, device=cuda); db = torch.empty(D, device=cuda)
    ops.layernorm_bwd(mode, dy, x, mu, rs, w, dres, dx, dg, db)
    assert rel(dx, gx + dres.double()) < 1e-4 and rel(dg, gw) < 1e-4 and rel(db, gb) < 1e-4
    # fused operand cast: the same gradient, row-scaled, in the mode's dtype == a separate tcow_scale_cast of dx
    sc = torch.rand(rows, device=cuda, generator=g) + 0.5
    for scale in (sc, None):
        dx2 = torch.empty(rows, D, device=cuda); dxc = torch.empty(rows, D, device=cuda, dtype=dt); want = torch.empty(rows, D, device=cuda, dtype=dt)
        ops.layernorm_bwd(mode, dy, x, mu, rs, w, dres, dx2, dx_cast=dxc, cast_scale=scale)
        ops.scale_cast(mode, dx2, scale, want)
        assert torch.equal(dx2, dx) and torch.equal(dxc, want)
    # fused row-weighted column sum of dx (the bias gradient of the Linear below the norm), with and without weights
    for scale in (sc, None):
        dx3 = torch.empty(rows, D, device=cuda); dg3 = torch.empty(D, device=cuda); db3 = torch.empty(D, device=cuda); cs = torch.full((D,), float('nan'), device=cuda)
        ops.layernorm_bwd(mode, dy, x, mu, rs, w, dres, dx3, dg3, db3, colsum_out=cs, colsum_scale=scale)
        want_cs = (dx.double() * (scale.double()[:, None] if scale is not None else 1.0)).sum(0)
        assert torch.equal(dx3, dx) and rel(dg3, dg) < 1e-5 and rel(db3, db) < 1e-5 and rel(cs, want_cs) < 1e-5


def _ref_attn(qkv, B, T, S, D, heads, ca, spatial):
    x = qkv.float().reshape(B, T, S, 3, heads, 64)
    out = torch.zeros(B, T, S, heads, 64, device=qkv.device)
    if spatial:
        s0 = 0 if ca in (0, 1) else 1
        q, k, v = [x[:, :, s0:, i].permute(0, 1, 3, 2, 4) for i in range(3)]
        out[:, :, s0:] = (((q @ k.transpose(-1, -2)) * 0.125).softmax(-1) @ v).permute(0, 1, 3, 2, 4)
    else:
        q, k, v = [x[:, :, 1:, i].permute(0, 2, 3, 1, 4) for i in range(3)]
        a = (q @ k.transpose(-1, -2)) * 0.125
        if ca > 0:                                                            # vit.py:93-99
            keep = torch.ones(T, T, dtype=torch.bool, device=qkv.device).tril(0 if ca <= 2 else ca - 2)
            a = a.masked_fill(~keep, -1e10)
        out[:, :, 1:] = (a.softmax(-1) @ v).permute(0, 3, 1, 2, 4)
    return out.reshape(B * T * S, D)


@pytest.mark.parametrize('mname,tol', [('f32', 2e-5), ('bf16', 1.5e-2), ('fp16', 2e-3), ('f32x3', 6e-5)])   # f32x3: f32 tensors, split-bf16 products (csrc/attention_x3.hip)
@pytest.mark.parametrize('spatial,B,T,S,heads,ca', [
    (False, 1, 4, 17, 4, 1), (False, 2, 30, 21, 2, 1), (False, 1, 30, 9, 2, 0), (False, 1, 7, 9, 2, 3), (False, 1, 40, 9, 2, 1), (False, 1, 70, 5, 1, 2),
    (True, 1, 2, 17, 4, 1), (True, 2, 3, 301, 2, 1), (True, 1, 2, 77, 2, 2), (True, 1, 1, 2, 1, 0), (True, 1, 1, 333, 1, 1),
    (True, 1, 2, 100, 2, 1), (True, 1, 1, 161, 1, 1), (True, 2, 1, 256, 2, 1), (True, 1, 1, 289, 3, 1),      # the one-kernel backward's range (4..10 key tiles): ragged / exact / one valid key in the last tile
    # the forward's packed placement (query tiles 4 + 4 + 2: the remainders of two sequences share a workgroup) with ragged lists per XCD -- a mixed workgroup
    # whose second sequence does not exist, an odd sequence out, mixed workgroups only (two tiles) -- and its half last key tile (<= 16 keys: 13 / exactly 16 / 17)
    (True, 1, 3, 301, 5, 1), (True, 1, 3, 40, 3, 1), (True, 1, 5, 173, 5, 1), (True, 1, 2, 304, 2, 1), (True, 1, 2, 305, 3, 0), (True, 3, 9, 301, 1, 1),
    (True, 1, 2, 1201, 12, 1)])          # last: the spatial sequence of BASELINE configs[3] (480x640: 1200 patches + cls)
def test_attention_fwd_bwd(ops, cuda, mname, tol, spatial, B, T, S, heads, ca):
    """Empty / ragged cases included: S=2 (one patch), T not a multiple of 32, sequences longer than the MFMA limits
    (T=70, S=333 take the f32-arithmetic kernels), causal windows 0 / 1 / look-ahead."""
    mode, dt = _mode(ops, mname)
    D = heads * 64; M = B * T * S
    g = torch.Generator(device='cuda').manual_seed(S * T)
    qkv = torch.randn(M, 3 * D, device=cuda, generator=g).to(dt)
    shape = ops.attn_shape(mode, B, T, S, D, heads, ca)
    out = torch.full((M, D), float('nan'), device=cuda, dtype=dt); lse = torch.empty(M, heads, device=cuda)
    ops.attn_fwd(shape, spatial, qkv, out, lse)
    q32 = qkv.float().requires_grad_(True)
    ref = _ref_attn(q32, B, T, S, D, heads, ca, spatial)
    assert torch.isfinite(out.float()).all()
    assert rel(out, ref) < tol
    dout = torch.randn(M, D, device=cuda, generator=g).to(dt)
    (ref * dout.float()).sum().backward()
    dqkv = torch.full((M, 3 * D), float('nan'), device=cuda, dtype=dt)
    ops.attn_bwd(shape, spatial, qkv, out, dout, lse, dqkv)
    assert torch.isfinite(dqkv.float()).all()
    assert rel(dqkv, q32.grad) < tol * 1.5


def test_im2col_matches_reference_patch_order(ops, cuda):
    """Bit-exact: pixel order of the patch gather against the map pushed through a one-hot Conv2d (G6)."""
    _, g6 = load_golden('g6_index_maps')
    pat = g6['patchify_4_4_2_3']                                              # [n, k] -> flat index in a (4, 8, 12) image
    img = torch.arange(4 * 8 * 12, dtype=torch.float32, device=cuda).reshape(1, 4, 1, 8, 12)
    out = torch.empty(1 * 1 * 7, 64, device=cuda)
    ops.im2col(ops.F32, img[:, :3].contiguous(), img[:, 3:].contiguous(), 4, False, out)
    assert torch.equal(out[0], torch.zeros(64, device=cuda))                  # slot 0 = cls placeholder
    assert np.array_equal(out[1:].cpu().numpy().astype(np.int32), pat)


@pytest.mark.parametrize('mname', ['f32', 'bf16', 'fp16'])
def test_im2col_normalisation_and_embeddings(ops, cuda, mname):
    mode, dt = _mode(ops, mname)
    B, T, H, W, P, D = 2, 3, 32, 48, 16, 64
    g = torch.Generator(device='cuda').manual_seed(0)
    rgb = torch.rand(B, 3, T, H, W, device=cuda, generator=g); qm = (torch.rand(B, 1, T, H, W, device=cuda, generator=g) > 0.5).float()
    N = (H // P) * (W // P); S = N + 1
    out = torch.empty(B * T * S, 4 * P * P, device=cuda, dtype=dt)
    ops.im2col(mode, rgb, qm, P, True, out)
    x = torch.cat([(rgb - 0.45) / 0.225, qm], 1)                              # vision_tf.py:81-89: query channel untouched
    ref = x.reshape(B, 4, T, H // P, P, W // P, P).permute(0, 2, 3, 5, 1, 4, 6).reshape(B, T, N, -1)
    got = out.float().reshape(B, T, S, -1)
    assert float(got[:, :, 0].abs().max()) == 0.0 and rel(got[:, :, 1:], ref) < (1e-6 if mname == 'f32' else 8e-3)
    tok = torch.randn(B * T * S, D, device=cuda, generator=g); cls = torch.randn(D, device=cuda, generator=g)
    pos = torch.randn(S, D, device=cuda, generator=g); te = torch.randn(T, D, device=cuda, generator=g)
    want = tok.reshape(B, T, S, D) + pos[None, None] + te[None, :, None]
    want[:, :, 0] = cls + pos[0]
    ops.embed_fwd(tok, B, T, S, cls, pos, te)
    assert rel(tok.reshape(B, T, S, D), want) < 1e-6
    gsrc = torch.randn(B * T * S, D, device=cuda, generator=g)
    dpos = torch.empty(S, D, device=cuda); dtime = torch.empty(T, D, device=cuda)
    ops.embed_bwd(gsrc, B, T, S, dpos, dtime)
    g4 = gsrc.reshape(B, T, S, D)
    assert rel(dpos, g4.sum((0, 1))) < 1e-5 and rel(dtime, g4[:, :, 1:].sum((0, 2))) < 1e-5
    # more frames than one unrolled round of the position-gradient kernel (24), a last channel block that is not full, accumulate mode
    B2, T2, S2, D2 = 3, 10, 5, 320
    g2 = torch.randn(B2 * T2 * S2, D2, device=cuda, generator=g)
    dpos2 = torch.ones(S2, D2, device=cuda); dtime2 = torch.ones(T2, D2, device=cuda)
    ops.embed_bwd(g2, B2, T2, S2, dpos2, dtime2, accumulate=True)
    g24 = g2.double().reshape(B2, T2, S2, D2)
    assert rel(dpos2, 1 + g24.sum((0, 1))) < 1e-6 and rel(dtime2, 1 + g24[:, :, 1:].sum((0, 2))) < 1e-6


def test_cls_merge_and_adjoint(ops, cuda):
    B, T, S, D = 2, 5, 7, 64
    x = torch.randn(B * T * S, D, device=cuda)
    for mode in (1, 0):
        y = x.clone(); ops.cls_merge(y, B, T, S, mode)
        y4, x4 = y.reshape(B, T, S, D), x.reshape(B, T, S, D)
        want = x4[:, 0:1, 0] if mode == 1 else x4[:, :, 0].mean(1, keepdim=True)
        assert rel(y4[:, :, 0], want.expand(B, T, D)) < 1e-6 and torch.equal(y4[:, :, 1:], x4[:, :, 1:])
        gy = torch.randn_like(x); gx = gy.clone(); ops.cls_merge(gx, B, T, S, mode, backward=True)
        # adjoint identity <merge(x), gy> == <x, merge^T(gy)>
        assert abs(float((y.double() * gy.double()).sum() - (x.double() * gx.double()).sum())) < 1e-3
        # the adjoint that also refreshes the operand copy of the rows it rewrites == adjoint followed by a full scale + cast
        sc = torch.rand(B * T * S, device=cuda) + 0.5
        for cmode, dt in ((ops.BF16, torch.bfloat16), (ops.FP16, torch.float16), (ops.F32, torch.float32)):
            for scale in (sc, None):
                g2 = gy.clone(); cast = torch.empty(B * T * S, D, device=cuda, dtype=dt); want_c = torch.empty_like(cast)
                ops.scale_cast(cmode, g2, scale, cast)                               # what the LayerNorm backward leaves: the copy of the un-merged gradient
                ops.cls_merge(g2, B, T, S, mode, backward=True, cast_mode=cmode, cast_out=cast, cast_scale=scale)
                ops.scale_cast(cmode, gx, scale, want_c)
                assert torch.equal(g2, gx) and torch.equal(cast, want_c)


@pytest.mark.parametrize('st,bilinear', [(4, True), (4, False), (2, True), (1, False)])
def test_mask_head_pool_upsample(ops, cuda, st, bilinear):
    """mask_tracker.py:113-132 against F.avg_pool2d + F.interpolate(align_corners=True)."""
    B, T, Hp, Wp, P, C = 2, 3, 3, 4, 16, 3
    S = Hp * Wp + 1
    pm = torch.randn(B * T * S, C * P * P, device=cuda)
    x = pm.reshape(B, T, S, C, P, P)[:, :, 1:].reshape(B, T, Hp, Wp, C, P, P).permute(0, 4, 1, 2, 5, 3, 6).reshape(B, C, T, Hp * P, Wp * P)
    xf = x.permute(0, 2, 1, 3, 4).reshape(B * T, C, Hp * P, Wp * P).double().requires_grad_(True)
    ref = F.avg_pool2d(xf, st, st) if st > 1 else xf
    pooled = torch.empty(B * T, C, Hp * P // st, Wp * P // st, device=cuda)
    ops.unpatchify_pool_fwd(ops.F32, pm, B * T, Hp, Wp, P, C, st, pooled)
    assert rel(pooled, ref) < 1e-6
    pb = torch.empty_like(pooled)                                                       # the binary-16 operand image of the same values (the path's head at bf16)
    ops.unpatchify_pool_fwd(ops.BF16, pm.bfloat16(), B * T, Hp, Wp, P, C, st, pb)
    xb = pm.bfloat16().double().reshape(B, T, S, C, P, P)[:, :, 1:].reshape(B, T, Hp, Wp, C, P, P).permute(0, 1, 4, 2, 5, 3, 6).reshape(B * T, C, Hp * P, Wp * P)
    assert rel(pb, F.avg_pool2d(xb, st, st) if st > 1 else xb) < 1e-6
    if st > 1:
        up = F.interpolate(ref, scale_factor=st, mode='bilinear', align_corners=True) if bilinear else F.interpolate(ref, scale_factor=st, mode='nearest')
    else:
        up = ref
    out = torch.empty(B, C, T, Hp * P, Wp * P, device=cuda)
    ops.upsample_fwd(pooled, B, T, C, Hp * P // st, Wp * P // st, st, bilinear and st > 1, out)
    want = up.reshape(B, T, C, Hp * P, Wp * P).permute(0, 2, 1, 3, 4)
    assert rel(out, want) < 1e-5
    gout = torch.randn_like(out)
    (up * gout.permute(0, 2, 1, 3, 4).reshape_as(up).double()).sum().backward()
    dpooled = torch.empty_like(pooled)
    ops.upsample_bwd(gout, B, T, C, Hp * P // st, Wp * P // st, st, bilinear and st > 1, dpooled)
    if st == 4 and bilinear:          # the variant that also returns max |dout| (the binary16 mode's loss-scale statistic): same gradient, exact maximum
        dp2, amax = ops.upsample_bwd_amax(gout, B, T, C, Hp * P // st, Wp * P // st, st, torch.empty_like(pooled))
        assert torch.equal(dp2, dpooled) and float(amax) == float(gout.abs().max())
    dpm = torch.empty_like(pm)
    ops.unpatchify_pool_bwd(ops.F32, dpooled, B * T, Hp, Wp, P, C, st, dpm)
    gx = xf.grad.reshape(B, T, C, Hp, P, Wp, P).permute(0, 1, 3, 5, 2, 4, 6).reshape(B, T, Hp * Wp, C * P * P)
    got = dpm.reshape(B, T, S, -1)
    assert float(got[:, :, 0].abs().max()) == 0.0 and rel(got[:, :, 1:], gx) < 1e-5


def test_flags_casts(ops, cuda):
    BT, S, D, Fc = 6, 13, 128, 3
    x = torch.randn(BT * S, D, device=cuda); Wf = torch.randn(Fc, D, device=cuda); bf = torch.randn(Fc, device=cuda)
    fl = torch.empty(BT, Fc, device=cuda)
    ops.flags_fwd(x, BT, S, Wf, bf, fl)
    assert rel(fl, x.reshape(BT, S, D)[:, 1:].mean(1) @ Wf.t() + bf) < 1e-5
    W = torch.randn(100, 72, device=cuda)
    Wc = torch.empty(100, 72, device=cuda, dtype=torch.bfloat16); Wt = torch.empty(72, 100, device=cuda, dtype=torch.bfloat16)
    ops.cast_transpose(ops.BF16, W, Wc, Wt)
    assert torch.equal(Wc, W.bfloat16()) and torch.equal(Wt, W.bfloat16().t().contiguous())    # round-to-nearest-even like torch
    rs = torch.rand(50, device=cuda); src = torch.randn(50, 64, device=cuda); dst = torch.empty(50, 64, device=cuda, dtype=torch.bfloat16)
    ops.scale_cast(ops.BF16, src, rs, dst)
    assert torch.equal(dst, (src * rs[:, None]).bfloat16())
    # the binary16 build rounds to nearest even as well
    Wh = torch.empty(100, 72, device=cuda, dtype=torch.float16); Wth = torch.empty(72, 100, device=cuda, dtype=torch.float16)
    ops.cast_transpose(ops.FP16, W, Wh, Wth)
    assert torch.equal(Wh, W.half()) and torch.equal(Wth, W.half().t().contiguous())
    dh = torch.empty(50, 64, device=cuda, dtype=torch.float16)
    ops.scale_cast(ops.FP16, src, rs, dh)
    assert torch.equal(dh, (src * rs[:, None]).half())


def test_layernorm_deferred_fold_is_bit_identical(ops, cuda):
    """layernorm_bwd(..., defer=jobs) + layernorm_fold(jobs) (one launch for several LayerNorm calls: tcow_layernorm_fold) vs the immediate fold:
    the same partial tables through the same reduction -- dgamma, dbeta and the fused column sum bit for bit; with and without accumulation."""
    torch.manual_seed(3)
    rows, D = 1000, 768
    jobs, want = [], []
    for k in range(3):
        x = torch.randn(rows, D, device=cuda); dy = torch.randn(rows, D, device=cuda).bfloat16(); g = torch.randn(D, device=cuda); dres = torch.randn(rows, D, device=cuda)
        mu = x.mean(1); rs = (x.var(1, unbiased=False) + 1e-6).rsqrt(); sc = torch.rand(rows, device=cuda)
        acc = k == 2
        outs = []
        for defer in (None, jobs):
            dg = torch.full((D,), 0.5, device=cuda); db = torch.full((D,), -0.25, device=cuda); cs = torch.zeros(D, device=cuda) if k != 1 else None
            dx = torch.empty(rows, D, device=cuda)
            ops.layernorm_bwd(ops.BF16, dy, x, mu, rs, g, dres, dx, dg, db, accumulate=acc, colsum_out=cs, colsum_scale=sc if cs is not None else None, defer=defer)
            outs.append((dg, db, cs, dx))
        want.append(outs)
    assert len(jobs) == 3
    ops.layernorm_fold(jobs)
    assert not jobs
    for (a, b) in want:
        for u, v in zip(a, b):
            assert (u is None and v is None) or torch.equal(u, v)
