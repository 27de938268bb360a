"""Forward / backward schedule of the Seeker hot path on libtcow_hip, as one autograd.Function.

Layout: residual stream R [M, D] f32 with M = B*T*S rows, row(b,t,s) = (b*T+t)*S + s, slot 0 = cls replica
(see include/tcow_hip.h).  Per block (vit.py:155-217), with the reference lines each launch replaces:

    U   = LN_t(R0)                                  temporal_norm1            vit.py:172
    QKV = U Wqkv_t^T + b                            temporal_attn.qkv         vit.py:81
    O   = attn_temporal(QKV)          causal mask   vit.py:88-109
    Pj  = dp_t * (O Wproj_t^T + b)                  temporal_attn.proj + DropPath   vit.py:111,172
    R1  = R0 + [s>=1] * (Pj Wfc^T + b)              temporal_fc + residual    vit.py:174-176
    V   = LN_1(R1);  QKV = V Wqkv^T + b;  O = attn_spatial(QKV)               vit.py:184-186 / 206-208
    R2  = R1 + dp_s * (O Wproj^T + b);  cls merge                             vit.py:189-215
    H   = GELU(LN_2(R2) W1^T + b1);  R3 = R2 + dp_m * (H W2^T + b2)           vit.py:216, 55-61

The backward below is the hand-derived adjoint of exactly this schedule; nothing goes through torch autograd.

16-bit modes FOLD the temporal projection: temporal_attn.proj -> DropPath -> temporal_fc (vit.py:111,172-176) are two Linear layers
with only a per-row scale dp_t between them, so

    R1 = R0 + [s>=1] * (dp_t * (O W'^T + b') + b_fc),      W' = Wfc Wproj,  b' = Wfc b_proj

is ONE [M,768] x [768,768] GEMM instead of two (forward, input gradient and weight gradient: 36 of them per step at depth 12).  The
state dict keeps both Linear layers: W', b' are recomputed from the f32 masters after every optimizer step (one batched launch of small
f32 products, ops.sgemm_batched), and the backward turns dW' = dY'^T O, db' into dWfc = dW' Wproj^T, dWproj = Wfc^T dW',
db_proj = Wfc^T db' the same way; b_fc's gradient is the row-masked column sum of dR1, taken in f32 inside the LayerNorm backward
that produces dR1.  (The f32 parity modes keep the reference's op order.)
"""
import os
import torch

from . import ops
from .ops import ACT_GELU, ACT_GELU_DSAVE, ACT_MUL_AUX

def _layout(module):
    """(parameters per block, offset of each sub-module's weight inside a block's slice of QueryMaskTracker.param_list(); bias = +1)."""
    if module.attention_type == 'divided_space_time':
        return 20, dict(tn=0, tqkv=2, tproj=4, tfc=6, n1=8, qkv=10, proj=12, n2=14, fc1=16, fc2=18)
    return 12, dict(n1=0, qkv=2, proj=4, n2=6, fc1=8, fc2=10)          # joint_space_time block: norm1, attn, norm2, mlp (vit.py:135-153)


def _joint_rows(module, g, dev):
    """Row indices [B * (1 + T*N)] of the joint-attention sequences inside the [B*T*S] token matrix: per clip its cls row (frame 0's
    slot 0) followed by every patch row.  The reference's sequence is (cls, patches in n-major order) (vision_tf.py:137); attention
    without a mask is invariant to the order of its keys / queries, so the frame-major order of our layout is used as is."""
    key = ('jrows', g['B'], str(dev))
    idx = module._wcache.get(key)
    if idx is None:
        B, T, S = g['B'], g['T'], g['S']
        r = torch.arange(B * T * S, device=dev).reshape(B, T * S)
        keep = torch.ones(T * S, dtype=torch.bool, device=dev)
        keep[torch.arange(1, T, device=dev) * S] = False                      # cls replicas of frames 1..T-1 take no part
        idx = r[:, keep].reshape(-1).contiguous()
        module._wcache[key] = idx
    return idx


def _w2d(p):
    return p.reshape(p.shape[0], -1)


def _get_weight(module, p, train):
    """Operand copies of a weight: (Wc [N,K] in the mode's dtype, Wt [K,N] or None), cached per parameter version."""
    mode = module.mode
    key = id(p)
    ent = module._wcache.get(key)
    ver = (p._version, getattr(module, '_wepoch', 0))
    if ent is not None and ent[0] == ver and ent[1] == mode and (ent[3] is not None or not train):
        return ent[2], ent[3]
    w = _w2d(p.detach())
    N, K = w.shape
    dt = ops.tdtype(mode)
    if ops.is16(mode):
        Wc = torch.empty(N, K, dtype=dt, device=w.device)
        Wt = torch.empty(K, N, dtype=dt, device=w.device) if train else None
        ops.cast_transpose(mode, w.contiguous(), Wc, Wt)
    else:
        Wc = w.contiguous()
        Wt = None
        if train:
            Wt = torch.empty(K, N, dtype=dt, device=w.device)
            ops.cast_transpose(mode, Wc, None, Wt)
    module._wcache[key] = (ver, mode, Wc, Wt)
    if train:
        # remember the buffers: after an optimizer step refresh_weights() re-casts ALL registered weights in one launch
        reg = module.__dict__.setdefault('_wreg', {})
        reg[key] = (p, Wc if ops.is16(mode) else None, Wt, N, K)
        module.__dict__['_wreg_gen'] = module.__dict__.get('_wreg_gen', 0) + 1      # (FusedAdamWClip's tile table points into these buffers: it rebuilds when this moves)
    return Wc, Wt


def group_sizes(depth, spec=None):
    """Blocks per gradient group, TOP group first (the order the backward finishes them): TCOW_DDP_GROUP = one number (every group that
    size, the bottom one takes the remainder) or a comma list 'a,b,c' (top group a blocks, the next b, ...; the last entry repeats until the
    blocks are used up, the bottom group is cut to what is left).  Default '5,5,2' at depth 12: the bottom group's gradients only exist
    when the backward is over, so its all-reduce cannot overlap anything -- it is kept small (2 blocks + embeddings = 76 MB instead of the
    190 MB of an even 4/4/4 split), and the weight-gradient launches of the 5-block groups fill three whole rounds of the chip with ONE
    token slice (765 tiles of 256 x 256 on 256 CUs).  Other depths default to groups of four."""
    if spec is None:
        spec = os.environ.get('TCOW_DDP_GROUP') or ('5,5,2' if depth == 12 else '4')
    want = [max(1, int(x)) for x in str(spec).split(',') if x.strip()] or [4]
    sizes, left, k = [], depth, 0
    while left > 0:
        gs = min(want[min(k, len(want) - 1)], left)
        sizes.append(gs); left -= gs; k += 1
    return sizes


def _fold_on(module):
    return ops.is16(module.mode) and module.attention_type == 'divided_space_time'


def _fold_products(ents):
    """W' = Wfc Wproj and b' = Wfc b_proj for a list of fold entries: two batched launches (per 24 entries)."""
    for c0 in range(0, len(ents), 24):
        ops.sgemm_batched([(e['fc'].detach(), e['proj'].detach(), e['W32']) for e in ents[c0:c0 + 24]])
        ops.sgemm_batched([(e['fc'].detach(), e['bproj'].detach().view(1, -1).t(), e['b32'].view(-1, 1)) for e in ents[c0:c0 + 24]])


def _folded_weight(module, i, q, ix, train):
    """Operands of block i's folded temporal projection: (Wc' [D,D], Wt' [D,D] or None, b' [D] f32), cached per parameter version."""
    pfc, pproj, bproj = q[ix['tfc']], q[ix['tproj']], q[ix['tproj'] + 1]
    key = ('fold', i)
    ver = (pfc._version, pproj._version, bproj._version, getattr(module, '_wepoch', 0))
    ent = module._wcache.get(key)
    if ent is not None and ent['ver'] == ver and ent['mode'] == module.mode and (ent['Wt'] is not None or not train):
        return ent['Wc'], ent['Wt'], ent['b32']
    D = pfc.shape[0]
    dev = pfc.device
    dt = ops.tdtype(module.mode)
    if ent is None or ent['mode'] != module.mode:
        ent = dict(fc=pfc, proj=pproj, bproj=bproj, W32=torch.empty(D, D, dtype=torch.float32, device=dev), b32=torch.empty(D, dtype=torch.float32, device=dev),
                   Wc=torch.empty(D, D, dtype=dt, device=dev), Wt=None, mode=module.mode)
    if train and ent['Wt'] is None:
        ent['Wt'] = torch.empty(D, D, dtype=dt, device=dev)
    _fold_products([ent])
    ops.cast_transpose(module.mode, ent['W32'], ent['Wc'], ent['Wt'])
    ent['ver'] = ver
    module._wcache[key] = ent
    if train:
        # registered like any GEMM weight: refresh_weights() recomputes W' (all blocks, one launch) and re-casts it with the others
        module.__dict__.setdefault('_wreg', {})[key] = (ent['W32'], ent['Wc'], ent['Wt'], D, D)
        module.__dict__.setdefault('_foldreg', {})[key] = ent
    return ent['Wc'], ent['Wt'], ent['b32']


def refresh_weights(module):
    """Re-cast every registered GEMM weight (bf16 copy + transposed copy) with ONE kernel launch instead of one per weight;
    called by QueryMaskTracker.invalidate_weight_cache() right after the optimizer has written the f32 master weights."""
    import struct
    reg = getattr(module, '_wreg', None)
    if not reg:
        return False
    mode = module.mode
    folds = module.__dict__.get('_foldreg')
    if folds:
        live = [ent for k, ent in folds.items() if k in reg]         # (a fold whose operand copies are no longer registered is stale: skip it)
        if live:
            _fold_products(live)
    # weights whose copies the optimizer has just written itself (FusedAdamWClip's tile kernel, 16-bit modes): only their version stamps are refreshed below
    done = module.__dict__.pop('_opt_cast_keys', None) or ()
    all_ents = list(reg.items())
    ents = [(k, v) for k, v in all_ents if k not in done]
    if not ents:
        _stamp_versions(module, all_ents, folds, mode)
        return True
    sig = tuple((k, p.data_ptr(), 0 if Wc is None else Wc.data_ptr(), 0 if Wt is None else Wt.data_ptr()) for k, (p, Wc, Wt, N, K) in ents) + (mode,)
    tab = module.__dict__.get('_wtab')
    if tab is None or tab[0] != sig:
        rec, tiles = [], 0
        edge = 64 if all(N % 64 == 0 and K % 64 == 0 for _, (_, _, _, N, K) in ents) else 32     # tile edge of tcow_cast_transpose_batched (all records alike)
        for k, (p, Wc, Wt, N, K) in ents:
            rec.append(struct.pack('<QQQiiii', p.data_ptr(), 0 if Wc is None else Wc.data_ptr(), 0 if Wt is None else Wt.data_ptr(), N, K, tiles, edge))
            tiles += ((N + edge - 1) // edge) * ((K + edge - 1) // edge)
        assert len(rec[0]) == L_cast_desc_bytes()
        dev = ents[0][1][0].device
        buf = torch.frombuffer(bytearray(b''.join(rec)), dtype=torch.uint8).to(dev)
        tab = (sig, buf, len(rec), tiles)
        module.__dict__['_wtab'] = tab
    ops.cast_transpose_batched(mode, tab[1], tab[2], tab[3])
    _stamp_versions(module, all_ents, folds, mode)
    return True


def _stamp_versions(module, ents, folds, mode):
    """The operand copies of `ents` are current for the parameters' present versions (and this weight epoch)."""
    epoch = getattr(module, '_wepoch', 0)
    for k, (p, Wc, Wt, N, K) in ents:
        if folds and k in folds:
            e = folds[k]
            e['ver'] = (e['fc']._version, e['proj']._version, e['bproj']._version, epoch)
            continue
        w = _w2d(p.detach())
        module._wcache[k] = ((p._version, epoch), mode, Wc if Wc is not None else w.contiguous(), Wt)


def L_cast_desc_bytes():
    from . import _lib
    return int(_lib.lib().tcow_cast_desc_bytes())


def _row_vectors(module, g, train):
    """mask0 [M] (0 on slot 0) and the DropPath row scales of every block (None in eval)."""
    B, T, S, N = g['B'], g['T'], g['S'], g['N']
    dev = module.vit.pos_embed.device
    key = ('mask0', B, str(dev))
    mask0 = module._wcache.get(key)
    if mask0 is None:
        mask0 = torch.ones(B, T, S, dtype=torch.float32, device=dev)
        mask0[:, :, 0] = 0
        mask0 = mask0.reshape(-1).contiguous()
        module._wcache[key] = mask0
    depth = module.network_depth
    rates = torch.linspace(0, module.drop_path_rate, depth).tolist() if depth > 1 else [0.0]   # vit.py:272
    scales = []
    forced = module.forced_drop_masks
    joint = module.attention_type != 'divided_space_time'
    if joint:
        # joint_space_time block (vit.py:161-162): both DropPath calls see x of shape (B, 1+N*T, D) -> one draw per sample each
        for i in range(depth):
            ent = {'t': None, 's': None, 'm': None}
            for kind, name in (('s', 'spatial'), ('m', 'mlp')):
                if forced is not None:
                    if (i, name) in forced:
                        keep, rate = forced[(i, name)]
                        ent[kind] = (keep.to(dev, torch.float32).reshape(B) / (1.0 - rate))[:, None].expand(B, T * S).reshape(-1).contiguous()
                elif train and rates[i] > 0.:
                    keep = (torch.rand(B, device=dev) + (1.0 - rates[i])).floor_()
                    ent[kind] = (keep / (1.0 - rates[i]))[:, None].expand(B, T * S).reshape(-1).contiguous()
            scales.append(ent)
        return mask0, scales
    if forced is None and train and max(rates) > 0.:
        # all DropPath draws of the step in one batch (vit_utils.py:150-152 per call: keep = floor(rand + 1 - r), x / (1 - r) * keep):
        # one rand and three broadcasts instead of ~15 tiny launches per block
        kkey = ('keep_p', tuple(rates), str(dev))
        keep_p = module._wcache.get(kkey)
        if keep_p is None:
            keep_p = (1.0 - torch.tensor(rates, dtype=torch.float32)).to(dev)
            module._wcache[kkey] = keep_p
        u = torch.rand(depth, B * N + B * T + B, device=dev)
        if u.is_cuda and N == S - 1:
            from . import ops
            rt, rsp, rml, rt0 = ops.droppath_rows(u, keep_p, mask0, B, T, S)       # the broadcasts below as one launch
        else:
            kp = keep_p[:, None]
            sc = (u + kp).floor_() / kp
            kt = sc[:, :B * N].reshape(depth, B, 1, N); ks = sc[:, B * N:B * N + B * T].reshape(depth, B, T, 1); km = sc[:, B * N + B * T:].reshape(depth, B, 1)
            rt = torch.ones(depth, B, T, S, dtype=torch.float32, device=dev)
            rt[:, :, :, 1:] = kt
            rt = rt.reshape(depth, -1)
            rsp = ks.expand(depth, B, T, S).reshape(depth, -1)
            rml = km.expand(depth, B, T * S).reshape(depth, -1)
            rt0 = rt * mask0[None, :]               # mask0 * dp_t: row scale of the folded temporal projection
        for i in range(depth):
            live = rates[i] > 0.
            scales.append({'t': rt[i] if live else None, 's': rsp[i] if live else None, 'm': rml[i] if live else None, 't0': rt0[i] if live else mask0})
        return mask0, scales
    for i in range(depth):
        r = rates[i]
        ent = {'t': None, 's': None, 'm': None}
        if forced is not None or (train and r > 0.):
            def draw(kind, shape):
                if forced is not None:
                    if (i, kind) not in forced:
                        return None
                    keep, rate = forced[(i, kind)]
                    return keep.to(dev, torch.float32).reshape(shape) / (1.0 - rate)
                keep = (torch.rand(shape, device=dev) + (1.0 - r)).floor_()           # vit_utils.py:150-152
                return keep / (1.0 - r)
            kt = draw('temporal', (B, N))
            if kt is not None:
                rs = torch.ones(B, T, S, dtype=torch.float32, device=dev)
                rs[:, :, 1:] = kt[:, None, :]
                ent['t'] = rs.reshape(-1).contiguous()
            ks = draw('spatial', (B, T))
            if ks is not None:
                ent['s'] = ks[:, :, None].expand(B, T, S).reshape(-1).contiguous()
            km = draw('mlp', (B,))
            if km is not None:
                ent['m'] = km[:, None].expand(B, T * S).reshape(-1).contiguous()
        ent['t0'] = mask0 if ent['t'] is None else ent['t'] * mask0
        scales.append(ent)
    return mask0, scales


def _effective_embeddings(module, g):
    """pos [S,D] / time [T,D] as used by the forward, with the nearest-neighbour resize of vision_tf.py:103-115,
    127-132 when the stored tables do not match the clip; returns index maps for the backward."""
    v = module.vit
    pos = v.pos_embed.detach()[0]
    S, T = g['S'], g['T']
    pos_idx = None
    if pos.shape[0] != S:
        n_old = pos.shape[0] - 1
        Pold = int(n_old ** 0.5)
        Hn = S // g['Wp']                       # vision_tf.py:108: H = x.size(1) // W
        ys = (torch.arange(Hn, device=pos.device).float() * (Pold / Hn)).floor().long().clamp_(max=Pold - 1)
        xs = (torch.arange(g['Wp'], device=pos.device).float() * (Pold / g['Wp'])).floor().long().clamp_(max=Pold - 1)
        grid = (ys[:, None] * Pold + xs[None, :]).reshape(-1) + 1
        pos_idx = torch.cat([torch.zeros(1, dtype=torch.long, device=pos.device), grid])
        pos = pos[pos_idx].contiguous()
    te = v.time_embed.detach()[0]
    time_idx = None
    if te.shape[0] != T:
        time_idx = (torch.arange(T, device=te.device).float() * (te.shape[0] / T)).floor().long().clamp_(max=te.shape[0] - 1)
        te = te[time_idx].contiguous()
    return pos.contiguous(), te.contiguous(), pos_idx, time_idx


def run_forward(module, rgb, qm, params, save):
    mode = module.mode
    gmode = module.gemm_mode           # the GEMM entry points' arithmetic: `mode`, or TCOW_F32X3 (f32 tensors, bf16 x 3 split products) for precision='bf16x3'
    dt = ops.tdtype(mode)
    dev = rgb.device
    B = qm.shape[0]                          # query rows (== clips unless the rgb frames are shared between a clip's queries)
    g = module.geometry(B)
    T, S, D, M, P, heads = g['T'], g['S'], g['D'], g['M'], g['P'], g['heads']
    ca = module.causal_attention
    use_cls = ca in (0, 1)
    train = save
    f32 = torch.float32

    def E(*shape, dtype=dt):
        return torch.empty(*shape, dtype=dtype, device=dev)

    W = lambda p: _get_weight(module, p, train)[0]
    mask0, dps = _row_vectors(module, g, module.training)
    sv = {'g': g, 'blocks': [], 'mask0': mask0, 'dps': dps} if save else None

    # ---- patch embed (vit.py:233-241) + embeddings (vision_tf.py:99-138)
    Kpe = module.input_channels * P * P
    X = E(M, D, dtype=f32)
    Bc = rgb.shape[0]                        # clips; B = Bc * Qs query rows
    Qs = B // Bc
    if Qs > 1:
        # shared rgb: the Qs queries of a clip see the same frames (pipeline.py:134-158), only the mask channel differs:
        # conv(cat[rgb, mask]) = W[:, :3] * rgb + W[:, 3] * mask -> one K = 3 P^2 GEMM per clip + one K = P^2 GEMM per query, the
        # latter taking the clip's rgb part as its residual operand
        Krgb = 3 * P * P
        Wc = W(params[3])                                            # [D, 4 P^2], channel-major columns (vit.py:233)
        A_rgb = E(Bc * T * S, Krgb); A_m = E(M, Kpe - Krgb)
        ops.im2col_channels(mode, rgb, P, module.tracker_pretrained, A_rgb)
        ops.im2col_channels(mode, qm, P, False, A_m)
        Xrgb = E(Bc * T * S, D, dtype=f32)
        ops.gemm_nt(gmode, A_rgb, Wc[:, :Krgb], Xrgb, bias=params[4].detach())
        TS = T * S
        for bq in range(B):
            b = bq // Qs
            ops.gemm_nt(gmode, A_m[bq * TS:(bq + 1) * TS], Wc[:, Krgb:], X[bq * TS:(bq + 1) * TS], resid=Xrgb[b * TS:(b + 1) * TS])
        A_pe = (A_rgb, A_m)
    else:
        A_pe = E(M, Kpe)
        ops.im2col(mode, rgb, qm, P, module.tracker_pretrained, A_pe)
        ops.gemm_nt(gmode, A_pe, W(params[3]), X, bias=params[4].detach())
    pos, te, pos_idx, time_idx = _effective_embeddings(module, g)
    ops.embed_fwd(X, B, T, S, params[0].detach().reshape(-1), pos, te)
    if save:
        sv.update(A_pe=A_pe, pos_idx=pos_idx, time_idx=time_idx)

    BP, ix = _layout(module)
    joint = module.attention_type != 'divided_space_time'
    fold = _fold_on(module)
    amode = gmode if gmode == ops.F32X3 else mode      # precision='bf16x3': the attention products on split-bf16 MFMAs too (csrc/attention_x3.hip)
    shape_attn = ops.attn_shape(amode, B, T, S, D, heads, ca)
    if joint:
        jrows = _joint_rows(module, g, dev)
        Lj = 1 + T * (S - 1)
        shape_joint = ops.attn_shape(amode, B, 1, Lj, D, heads, 0)          # one sequence of 1 + N*T tokens per clip, cls included, no mask (vit.py:159-161)
    for i in range(module.network_depth):
        q = params[5 + i * BP: 5 + (i + 1) * BP]
        P_ = lambda name, k=0: q[ix[name] + k].detach()                     # weight (k = 0) / bias (k = 1) of a sub-module of this block
        n1_w, n1_b, qkv_b, proj_b, n2_w, n2_b, fc1_w, fc1_b, fc2_b = P_('n1'), P_('n1', 1), P_('qkv', 1), P_('proj', 1), P_('n2'), P_('n2', 1), P_('fc1'), P_('fc1', 1), P_('fc2', 1)
        dp = dps[i]
        st = {}
        R0 = X
        if joint:
            # ---- joint space-time attention (vit.py:159-162): x = x + drop_path(attn(norm1(x))) over all 1 + N*T tokens of a clip
            V = E(M, D); mu1 = E(M, dtype=f32) if save else None; rs1 = E(M, dtype=f32) if save else None
            ops.layernorm_fwd(mode, R0, n1_w, n1_b, V, mu1, rs1)
            QKV2 = E(M, 3 * D)
            ops.gemm_nt(gmode, V, W(q[ix['qkv']]), QKV2, bias=qkv_b)
            QJ = QKV2.index_select(0, jrows)                                # compact (cls, patches) sequences: the kernels take contiguous ones
            OJ = E(B * Lj, D); lse_s = E(B * Lj, heads, dtype=f32) if save else None
            ops.attn_fwd(shape_joint, True, QJ, OJ, lse_s)
            O2 = torch.zeros(M, D, dtype=dt, device=dev)
            O2.index_copy_(0, jrows, OJ)                                    # (the unused cls replicas of frames 1.. get a zero attention output)
            rs_s = dp['s']
            R2 = E(M, D, dtype=f32) if save else R0
            ops.gemm_nt(gmode, O2, W(q[ix['proj']]), R2, bias=proj_b, row_scale=rs_s, resid=R0)
            if save:
                st.update(R1=R0, mu1=mu1, rs1=rs1, V=V, QKV_s=QJ, O_s=O2, OJ=OJ, lse_s=lse_s, rs_s=rs_s)
        else:
            tn_w, tn_b, tqkv_b, tproj_b, tfc_b = P_('tn'), P_('tn', 1), P_('tqkv', 1), P_('tproj', 1), P_('tfc', 1)
            # temporal
            U = E(M, D); mu0 = E(M, dtype=f32) if save else None; rs0 = E(M, dtype=f32) if save else None
            ops.layernorm_fwd(mode, R0, tn_w, tn_b, U, mu0, rs0)
            QKV = E(M, 3 * D)
            ops.gemm_nt(gmode, U, W(q[ix['tqkv']]), QKV, bias=tqkv_b)
            O = E(M, D); lse_t = E(M, heads, dtype=f32) if save else None
            ops.attn_fwd(shape_attn, False, QKV, O, lse_t)
            R1 = E(M, D, dtype=f32) if save else R0
            if fold:
                Wf, _, bprime = _folded_weight(module, i, q, ix, train)
                ops.gemm_nt(gmode, O, Wf, R1, bias=bprime, row_scale=dp['t0'], resid=R0, bias2=tfc_b, row_scale2=mask0)
                Pj = None
            else:
                Pj = E(M, D)
                ops.gemm_nt(gmode, O, W(q[ix['tproj']]), Pj, bias=tproj_b, row_scale=dp['t'])
                ops.gemm_nt(gmode, Pj, W(q[ix['tfc']]), R1, bias=tfc_b, row_scale=mask0, resid=R0)
            if save:
                st.update(R0=R0, mu0=mu0, rs0=rs0, U=U, QKV_t=QKV, O_t=O, lse_t=lse_t, Pj=Pj)
            # spatial
            V = E(M, D); mu1 = E(M, dtype=f32) if save else None; rs1 = E(M, dtype=f32) if save else None
            ops.layernorm_fwd(mode, R1, n1_w, n1_b, V, mu1, rs1)
            QKV2 = E(M, 3 * D)
            ops.gemm_nt(gmode, V, W(q[ix['qkv']]), QKV2, bias=qkv_b)
            O2 = E(M, D); lse_s = E(M, heads, dtype=f32) if save else None
            ops.attn_fwd(shape_attn, True, QKV2, O2, lse_s)
            rs_s = dp['s']
            if not use_cls:
                rs_s = mask0 if rs_s is None else rs_s * mask0
            R2 = E(M, D, dtype=f32) if save else R1
            ops.gemm_nt(gmode, O2, W(q[ix['proj']]), R2, bias=proj_b, row_scale=rs_s, resid=R1)
            if use_cls:
                ops.cls_merge(R2, B, T, S, 1 if ca == 1 else 0)
            if save:
                st.update(R1=R1, mu1=mu1, rs1=rs1, V=V, QKV_s=QKV2, O_s=O2, lse_s=lse_s, rs_s=rs_s)
        # mlp
        Wn = E(M, D); mu2 = E(M, dtype=f32) if save else None; rs2 = E(M, dtype=f32) if save else None
        ops.layernorm_fwd(mode, R2, n2_w, n2_b, Wn, mu2, rs2)
        Hd = fc1_w.shape[0]
        pre = E(M, Hd) if save else None
        H = E(M, Hd)
        # training saves GELU'(pre-activation) instead of the pre-activation: the backward is then one multiply per element
        ops.gemm_nt(gmode, Wn, W(q[ix['fc1']]), H, bias=fc1_b, act=ACT_GELU_DSAVE if save else ACT_GELU, aux=pre)
        R3 = E(M, D, dtype=f32) if save else R2
        ops.gemm_nt(gmode, H, W(q[ix['fc2']]), R3, bias=fc2_b, row_scale=dp['m'], resid=R2)
        if save:
            st.update(R2=R2, mu2=mu2, rs2=rs2, Wn=Wn, pre=pre, H=H)
            sv['blocks'].append(st)
        X = R3

    # ---- output heads
    nb = 5 + module.network_depth * BP
    norm_w, norm_b, head_w, head_b = params[nb], params[nb + 1], params[nb + 2], params[nb + 3]
    feat32 = X
    if module.norm_embeddings:                                            # vision_tf.py:152-153
        Fm = E(M, D); muf = E(M, dtype=f32) if save else None; rsf = E(M, dtype=f32) if save else None
        ops.layernorm_fwd(mode, X, norm_w.detach(), norm_b.detach(), Fm, muf, rsf)
        if module.flag_channels > 0 and mode != ops.F32:
            feat32 = E(M, D, dtype=f32)
            ops.layernorm_fwd(ops.F32, X, norm_w.detach(), norm_b.detach(), feat32)
        elif mode == ops.F32:
            feat32 = Fm
        if save:
            sv.update(muf=muf, rsf=rsf)
    elif mode == ops.F32:
        Fm = X
    else:
        Fm = E(M, D)
        ops.scale_cast(mode, X, None, Fm)
    Co = module.output_channels
    Pm = E(M, Co * P * P)
    ops.gemm_nt(gmode, Fm, W(head_w), Pm, bias=head_b.detach())           # mask_tracker.py:113
    stp = module.track_map_stride if module.track_map_stride > 1 else 1
    h, w = module.frame_height // stp, module.frame_width // stp
    pooled = E(B * T, Co, h, w, dtype=f32)
    ops.unpatchify_pool_fwd(mode, Pm, B * T, g['Hp'], g['Wp'], P, Co, stp, pooled)
    out_mask = E(B, Co, T, module.frame_height, module.frame_width, dtype=f32)
    bilinear = (module.track_map_resize == 'bilinear') and stp > 1
    ops.upsample_fwd(pooled, B, T, Co, h, w, stp, bilinear, out_mask)
    if module.flag_channels > 0:
        flags = E(B, T, module.flag_channels, dtype=f32)
        ops.flags_fwd(feat32, B * T, S, params[nb + 4].detach(), params[nb + 5].detach(), flags)   # mask_tracker.py:135-137
    else:
        flags = torch.zeros(0, device=dev)
    if save:
        sv.update(X_final=X, Fm=Fm, feat32=feat32, stp=stp, bilinear=bilinear, h=h, w=w)
    return out_mask, flags, sv


def run_backward(module, sv, params, d_mask, d_flags):
    mode = module.mode
    gmode = module.gemm_mode           # the GEMM entry points' arithmetic: `mode`, or TCOW_F32X3 (f32 tensors, bf16 x 3 split products) for precision='bf16x3'
    dt = ops.tdtype(mode)
    g = sv['g']
    B, T, S, D, M, P, heads = g['B'], g['T'], g['S'], g['D'], g['M'], g['P'], g['heads']
    dev = sv['X_final'].device
    ca = module.causal_attention
    use_cls = ca in (0, 1)
    f32 = torch.float32
    mask0 = sv['mask0']
    # binary16 activations: gradients of a mean-reduced loss (~1e-9 per element at BASELINE configs[1]) sit far below fp16's normal range
    # (6.1e-5), so the whole backward runs on gradients multiplied by a power of two and every finished gradient bucket is multiplied back
    # (exact) before anyone sees it -- everything in between is linear in the seed.  module.loss_scale = 'dynamic' (default): the scale is
    # chosen on the device, per backward, so that the largest seed element lands in [2^t / 2, 2^t) with t = module.ls_log2 (a device
    # scalar, -2 to start with).  Measured over 300 training steps (tools/dev_fp16_amp.py): the 16-bit gradient operands peak at 3 ... 12 000
    # times the seed maximum (median 240), i.e. at <= 6 000 of binary16's 65 504, with their median near 1e-3 (normal range).  An overflow
    # all the same shows up as a non-finite gradient norm: FusedAdamWClip then skips the update and lowers t by 4 (optim.py).
    # A number instead of 'dynamic' = static scale.
    # binary16: the backward runs on gradients multiplied by a power of two (exact), chosen from max |seed gradient| so that the seed's largest
    # entry lands on 2^ls_log2.  The mask head's first two steps (bilinear x4 adjoint, un-patchify) are f32 and linear, so the scale is applied to
    # the POOLED gradient (5 MB) behind them, not to d_mask (83 MB) -- bit-identical, a power of two commutes with the sums -- and max |d_mask| comes
    # out of the adjoint's own pass over d_mask (tcow_upsample_bwd_amax) instead of an abs + amax pair of passes.
    gscale = inv_gscale = None
    ls_mode = getattr(module, 'loss_scale', 'dynamic') if mode == ops.FP16 else None
    if ls_mode == 'dynamic' and _live_optim(module) is None and not module.__dict__.get('_warned_ls'):
        import warnings
        module.__dict__['_warned_ls'] = True
        warnings.warn("precision='fp16' with the dynamic loss scale but no FusedAdamWClip(..., module=net) attached: nothing lowers the scale "
                      "after an overflow or skips the poisoned step -- pass module= to the optimizer or set net.seeker.loss_scale to a number")

    def choose_scale(mask_amax):
        """-> (gscale, 1 / gscale) device scalars or (None, None); mask_amax = max |d_mask| (device scalar or None)."""
        if ls_mode is None:
            return None, None
        if ls_mode == 'dynamic':
            amax = mask_amax if mask_amax is not None else torch.zeros((), dtype=f32, device=dev)
            if d_flags is not None and d_flags.numel():
                amax = torch.maximum(amax, d_flags.detach().abs().amax().to(f32))
            gs = torch.exp2(torch.floor(module.ls_log2 - torch.log2(amax.clamp_min(1e-37)))).clamp(2.0 ** -20, 2.0 ** 60)     # no host sync
        elif float(ls_mode) != 1.0:
            gs = torch.tensor(float(ls_mode), dtype=f32, device=dev)
        else:
            return None, None
        return gs, 1.0 / gs

    def E(*shape, dtype=dt):
        return torch.empty(*shape, dtype=dtype, device=dev)

    Wt = lambda p: _get_weight(module, p, True)[1]
    grads = [None] * len(params)

    # Who undoes the loss scale.  With a data-parallel hook every finished bucket is multiplied back before the hook sees it (ranks choose their own
    # scales).  Without one (or with one that says it is not `active`: a GradSync of one rank), and with a live FusedAdamWClip(module=net) attached, the buckets stay scaled and
    # the optimizer's clip-coefficient kernel folds the inverse scale into the update (tcow_adamw_clip_step_scaled): 488 MB less read and written per
    # step.  param.grad then holds SCALED gradients (as under torch.cuda.amp.GradScaler before unscale_); module.pending_inv_scale (device scalar,
    # valid until the next backward) is the factor the optimizer applies on the fly.  Without an attached optimizer the gradients are unscaled here, as before.
    # The deferral needs ONE backward per optimizer step -- several backwards into the same param.grad (the reference's per-query model loop, gradient
    # accumulation, DataParallel replicas) would make autograd add buckets that carry different scales -- so it is tied to `persistent_grads` (whose
    # contract is exactly that: each backward OVERWRITES the gradients), never taken by a DataParallel replica, and only while the attached optimizer is
    # still alive (a weak reference: a discarded or replaced FusedAdamWClip must not leave param.grad scaled for torch.optim.AdamW or clip_grad_norm_).
    defer_unscale = ((module.grad_hook is None or getattr(module.grad_hook, 'active', True) is False) and _live_optim(module) is not None
                     and bool(getattr(module, 'persistent_grads', False)) and not getattr(module, '_is_replica', False)
                     and module.__dict__.get('_defer_unscale', True))                 # (tests switch the deferral off to compare the two paths)

    def publish(tag, flat):
        """A finished gradient bucket: undo the loss scale (unless the optimizer will), then hand it to the data-parallel hook."""
        if inv_gscale is not None and not defer_unscale:
            flat.mul_(inv_gscale)
        if module.grad_hook is not None:
            module.grad_hook(tag, flat)

    def bucket(indices):
        """One flat f32 buffer per gradient bucket (a transformer block, the heads, the embeddings): the views
        become the parameters' gradients and the flat buffer is what the data-parallel all-reduce moves."""
        indices = list(indices)
        total = sum(params[j].numel() for j in indices)
        flat = None
        if getattr(module, 'persistent_grads', False):      # stable gradient storage across steps (see QueryMaskTracker.persistent_grads)
            key = ('gbuf', indices[0], total, str(dev))
            flat = module._gbufs.get(key)
            if flat is None:
                flat = module._gbufs[key] = torch.empty(total, dtype=f32, device=dev)
        if flat is None:
            flat = torch.empty(total, dtype=f32, device=dev)
        off = 0
        for j in indices:
            n = params[j].numel()
            grads[j] = flat[off:off + n].view(params[j].shape)
            off += n
        return flat

    def galloc(idx):
        return grads[idx]

    pending = []          # weight-gradient GEMMs of the current block, issued together at its end (one grouped launch, ops.gemm_tn_grouped)

    def linear_bwd(idx_w, dY, Xin, defer=False):
        """dW, db of a Linear whose output-gradient operand is dY [M,N] and input operand Xin [M,K]."""
        dW = galloc(idx_w); db = galloc(idx_w + 1)
        if defer:
            pending.append((dY, Xin, dW.reshape(dW.shape[0], -1), db))
        else:
            ops.gemm_tn(gmode, dY, Xin, dW.reshape(dW.shape[0], -1), bias_grad=db)

    tn_group_max = ops.tn_group_max()

    ln_jobs = []          # LayerNorm parameter-gradient folds of the current group of blocks, one launch at its end (ops.layernorm_fold)

    def flush_pending():
        if pending:
            ops.gemm_tn_grouped(gmode, pending)
            pending.clear()
        if ln_jobs:
            ops.layernorm_fold(ln_jobs)

    BP, ix = _layout(module)
    joint = module.attention_type != 'divided_space_time'
    fold = _fold_on(module)
    depth = module.network_depth
    nb = 5 + depth * BP
    Co = module.output_channels
    have_flags = module.flag_channels > 0 and d_flags is not None and d_flags.numel() > 0
    head_idx = ([nb, nb + 1] if module.norm_embeddings else []) + [nb + 2, nb + 3] + ([nb + 4, nb + 5] if have_flags else [])
    # Gradient buckets = GROUPS of transformer blocks (group_sizes(): TCOW_DDP_GROUP, default 5 / 5 / 2 from the top at depth 12; the top group
    # also carries the output heads, the bottom one the embeddings): 3 all-reduces of ~75-190 MB at depth 12 instead of 14 of ~38 MB -- each collective is a window in which
    # resident RCCL workgroups push the one-workgroup-per-CU GEMMs into an extra round (profiles/r02_cu_contention.txt), so fewer, larger
    # ones.  The WEIGHT-GRADIENT GEMMs of a group are issued together at its end as well (one grouped launch over 28 problems = 612 tiles of
    # 256 x 256 at ViT-B): so many tiles fill whole rounds of the chip with TWO token slices instead of five, i.e. 60 % less f32 partial-sum
    # traffic in the GEMM and in the fold (tcow_tn_group_slices); nothing in the backward chain waits for a weight gradient, and the bucket
    # is not published before the group's end anyway.  Their operands (a block's activations and output gradients) stay alive that long.
    group_lo = {}                                   # block index -> first block of its group
    hi_ = depth
    for gs in group_sizes(depth):
        for j in range(hi_ - gs, hi_):
            group_lo[j] = hi_ - gs
        hi_ -= gs
    top_lo = group_lo[depth - 1]
    # The folded projection's three gradients per block (tproj.weight, tproj.bias, tfc.weight) come out of small products that are worth
    # batching over ALL blocks (a launch of three 768^3 problems is pure latency: 35 us whether it carries 3 problems or 12), so they form one
    # LATE bucket of their own (57 MB at ViT-B), finished and published once after the last block.
    late = set()
    if fold:
        for j in range(depth):
            late.update((5 + j * BP + ix['tproj'], 5 + j * BP + ix['tproj'] + 1, 5 + j * BP + ix['tfc']))
    late_flat = bucket(sorted(late)) if late else None
    blk_idx = lambda lo_, hi_: [j for j in range(5 + lo_ * BP, 5 + hi_ * BP) if j not in late]
    flat_cur = bucket((list(range(0, 5)) if top_lo == 0 else []) + blk_idx(top_lo, depth) + head_idx)
    fold_jobs = []          # (block, dW' [D,D], db' [D])

    def fold_tmp(i):
        t = module._gbufs.get(('foldtmp', i, str(dev)))
        if t is None:
            t = module._gbufs[('foldtmp', i, str(dev))] = (torch.empty(D, D, dtype=f32, device=dev), torch.empty(D, dtype=f32, device=dev))
        return t

    def finish_fold_group():
        """Z = P Wfc^T + b_fc with P = dp_t (O Wproj^T + b_proj) and dY' = dp_t dZ:  dWfc = dZ^T P = dW' Wproj^T + db' b_proj^T,
        dWproj = Wfc^T dW', db_proj = Wfc^T db' for all blocks: four batched launches (<= 24 problems each)."""
        if not fold_jobs:
            return
        nt_, r1_, tn_, gv_ = [], [], [], []
        for (i, tW, tb) in fold_jobs:
            o_ = 5 + i * BP
            wfc, wp, bp = params[o_ + ix['tfc']].detach(), params[o_ + ix['tproj']].detach(), params[o_ + ix['tproj'] + 1].detach()
            nt_.append((tW, wp.t(), grads[o_ + ix['tfc']]))
            r1_.append((tb.view(-1, 1), bp.view(1, -1), grads[o_ + ix['tfc']]))
            tn_.append((wfc.t(), tW, grads[o_ + ix['tproj']]))
            gv_.append((wfc.t(), tb.view(-1, 1), grads[o_ + ix['tproj'] + 1].view(-1, 1)))
        for c0 in range(0, len(nt_), 24):
            ops.sgemm_batched(nt_[c0:c0 + 24]); ops.sgemm_batched(r1_[c0:c0 + 24], accumulate=True)
            ops.sgemm_batched(tn_[c0:c0 + 24]); ops.sgemm_batched(gv_[c0:c0 + 24])
        fold_jobs.clear()

    # ---- mask head backward (mask_tracker.py:113-132)
    if d_mask is None:
        d_mask = torch.zeros(B, Co, T, module.frame_height, module.frame_width, dtype=f32, device=dev)
    d_mask = d_mask.to(f32).contiguous()
    dpooled = E(B * T, Co, sv['h'], sv['w'], dtype=f32)
    mask_amax = None
    if ls_mode == 'dynamic' and sv['bilinear'] and sv['stp'] == 4 and sv['h'] > 4 and sv['w'] > 4:
        _, mask_amax = ops.upsample_bwd_amax(d_mask, B, T, Co, sv['h'], sv['w'], sv['stp'], dpooled)
    else:
        if ls_mode == 'dynamic' and d_mask.numel():
            mask_amax = d_mask.abs().amax()
        ops.upsample_bwd(d_mask, B, T, Co, sv['h'], sv['w'], sv['stp'], sv['bilinear'], dpooled)
    gscale, inv_gscale = choose_scale(mask_amax)
    if gscale is not None:
        dpooled.mul_(gscale)
        d_flags = None if d_flags is None else d_flags * gscale
    dPm = E(M, Co * P * P)
    ops.unpatchify_pool_bwd(mode, dpooled, B * T, g['Hp'], g['Wp'], P, Co, sv['stp'], dPm)
    linear_bwd(nb + 2, dPm, sv['Fm'])
    # gradient w.r.t. the features that feed both heads, always f32: dFeat = dPm . Whead (+ flags adjoint)
    dFeat = E(M, D, dtype=f32)
    ops.gemm_nt(gmode, dPm, Wt(params[nb + 2]), dFeat)
    if module.flag_channels > 0 and have_flags:
        # flags head adjoint (F x D, once per step; only the plugin path ever asks for it, pipeline.py:238)
        df = d_flags.to(f32).reshape(B * T, -1).contiguous()
        meanf = sv['feat32'].reshape(B * T, S, D)[:, 1:, :].float().mean(dim=1)
        dmean = torch.empty(B * T, D, dtype=f32, device=dev)
        ops.sgemm_batched([(df.t(), meanf, grads[nb + 4])])                       # dWf = df^T mean(features)   (library kernel: no vendor BLAS on the path)
        ops.sgemm_batched([(df, params[nb + 4].detach(), dmean)])
        grads[nb + 5].copy_(df.sum(0))
        dFeat.reshape(B * T, S, D)[:, 1:, :] += (dmean / float(S - 1))[:, None, :]
    # (without a flags gradient flag_post_linear.* keep grad None, like the reference where pipeline.py:157 drops them)
    if module.norm_embeddings:
        dX = E(M, D, dtype=f32)
        ops.layernorm_bwd(ops.F32, dFeat, sv['X_final'], sv['muf'], sv['rsf'], params[nb].detach(), None, dX, galloc(nb), galloc(nb + 1))
    else:
        dX = dFeat     # model.norm takes no part when norm_embeddings is False (vision_tf.py:152): its grads stay None
    amode = gmode if gmode == ops.F32X3 else mode      # precision='bf16x3': the attention products on split-bf16 MFMAs too (csrc/attention_x3.hip)
    shape_attn = ops.attn_shape(amode, B, T, S, D, heads, ca)
    if joint:
        jrows = _joint_rows(module, g, dev)
        Lj = 1 + T * (S - 1)
        shape_joint = ops.attn_shape(amode, B, 1, Lj, D, heads, 0)
    dR3 = dX
    G3_next = None
    for i in reversed(range(module.network_depth)):
        o = 5 + i * BP
        q = params[o: o + BP]
        st = sv['blocks'][i]
        dp = sv['dps'][i]
        Hd = q[ix['fc1']].shape[0]
        if i != depth - 1 and group_lo[i] != group_lo[i + 1]:          # first (top) block of the next group: its bucket
            lo_ = group_lo[i]
            flat_cur = bucket((list(range(0, 5)) if lo_ == 0 else []) + blk_idx(lo_, i + 1))
        next_scale = (sv['dps'][i - 1]['m'] if i > 0 else mask0)      # row scale of the operand the block below (or the patch embedding) consumes
        # ---- mlp
        if G3_next is not None:
            G3 = G3_next                       # produced by the LayerNorm backward of the block above (fused cast)
        else:
            G3 = E(M, D)
            ops.scale_cast(mode, dR3, dp['m'], G3)
        dpre = E(M, Hd)
        ops.gemm_nt(gmode, G3, Wt(q[ix['fc2']]), dpre, act=ACT_MUL_AUX, aux=st['pre'])
        linear_bwd(o + ix['fc2'], G3, st['H'], defer=True)
        dWn = E(M, D)
        ops.gemm_nt(gmode, dpre, Wt(q[ix['fc1']]), dWn)
        linear_bwd(o + ix['fc1'], dpre, st['Wn'], defer=True)
        dR2 = E(M, D, dtype=f32)
        G2 = E(M, D)                           # operand copy of dR2 * row scale: written by the LayerNorm backward, its slot-0 rows redone by the cls adjoint
        ops.layernorm_bwd(mode, dWn, st['R2'], st['mu2'], st['rs2'], q[ix['n2']].detach(), dR3, dR2, galloc(o + ix['n2']), galloc(o + ix['n2'] + 1),
                          dx_cast=G2, cast_scale=st['rs_s'], defer=ln_jobs)
        del G3, dpre, dWn
        # ---- spatial / joint attention
        if use_cls and not joint:
            ops.cls_merge(dR2, B, T, S, 1 if ca == 1 else 0, backward=True, cast_mode=mode, cast_out=G2, cast_scale=st['rs_s'])
        dO2 = E(M, D)
        ops.gemm_nt(gmode, G2, Wt(q[ix['proj']]), dO2)
        linear_bwd(o + ix['proj'], G2, st['O_s'], defer=True)
        if joint:
            dQJ = E(B * Lj, 3 * D)
            ops.attn_bwd(shape_joint, True, st['QKV_s'], st['OJ'], dO2.index_select(0, jrows), st['lse_s'], dQJ)
            dQKV2 = torch.zeros(M, 3 * D, dtype=dt, device=dev)       # the unused cls replicas receive no gradient
            dQKV2.index_copy_(0, jrows, dQJ)
        else:
            dQKV2 = E(M, 3 * D)
            ops.attn_bwd(shape_attn, True, st['QKV_s'], st['O_s'], dO2, st['lse_s'], dQKV2)
        dV = E(M, D)
        ops.gemm_nt(gmode, dQKV2, Wt(q[ix['qkv']]), dV)
        linear_bwd(o + ix['qkv'], dQKV2, st['V'], defer=True)
        dR1 = E(M, D, dtype=f32)
        G1 = E(M, D)                           # bf16(dR1 * row scale), written by the same LayerNorm backward pass
        if fold:
            # G1 = bf16(dR1 * mask0 * dp_t) = dY' of the folded projection; its bias b_fc sees dR1 * mask0: summed here, in f32
            ops.layernorm_bwd(mode, dV, st['R1'], st['mu1'], st['rs1'], q[ix['n1']].detach(), dR2, dR1, galloc(o + ix['n1']), galloc(o + ix['n1'] + 1), dx_cast=G1,
                              cast_scale=dp['t0'], colsum_out=galloc(o + ix['tfc'] + 1), colsum_scale=mask0, defer=ln_jobs)
        else:
            ops.layernorm_bwd(mode, dV, st['R1'], st['mu1'], st['rs1'], q[ix['n1']].detach(), dR2, dR1, galloc(o + ix['n1']), galloc(o + ix['n1'] + 1), dx_cast=G1,
                              cast_scale=(next_scale if joint else mask0), defer=ln_jobs)
        del G2, dO2, dQKV2, dV
        if joint:
            G3_next = G1                       # a joint block has no temporal half: its input gradient is complete here
            dR3 = dR1
        else:
            # ---- temporal
            dO = E(M, D)
            if fold:
                ops.gemm_nt(gmode, G1, _folded_weight(module, i, q, ix, True)[1], dO)
                tW, tb = fold_tmp(i)
                pending.append((G1, st['O_t'], tW, tb))
                fold_jobs.append((i, tW, tb))
            else:
                dPj = E(M, D)
                ops.gemm_nt(gmode, G1, Wt(q[ix['tfc']]), dPj, row_scale=dp['t'])
                linear_bwd(o + ix['tfc'], G1, st['Pj'], defer=True)
                ops.gemm_nt(gmode, dPj, Wt(q[ix['tproj']]), dO)
                linear_bwd(o + ix['tproj'], dPj, st['O_t'], defer=True)
            dQKV = E(M, 3 * D)
            ops.attn_bwd(shape_attn, False, st['QKV_t'], st['O_t'], dO, st['lse_t'], dQKV)
            dU = E(M, D)
            ops.gemm_nt(gmode, dQKV, Wt(q[ix['tqkv']]), dU)
            linear_bwd(o + ix['tqkv'], dQKV, st['U'], defer=True)
            dR0 = E(M, D, dtype=f32)
            G3_next = E(M, D)                  # operand of the next (lower) block's MLP backward, or of the patch-embed weight gradient
            ops.layernorm_bwd(mode, dU, st['R0'], st['mu0'], st['rs0'], q[ix['tn']].detach(), dR1, dR0, galloc(o + ix['tn']), galloc(o + ix['tn'] + 1),
                              dx_cast=G3_next, cast_scale=next_scale, defer=ln_jobs)
            dR3 = dR0
        if group_lo[i] == i or len(pending) + 8 > tn_group_max:
            flush_pending()      # the group's weight-gradient GEMMs (six or seven per block; joint: four) as one grouped launch
        sv['blocks'][i] = None   # free this block's activations (the queued weight-gradient operands keep theirs)
        if i > 0 and group_lo[i] == i:                                 # last (bottom) block of a group that is not the bottom group: done
            publish('g%d' % i, flat_cur)

    # ---- embeddings + patch embed backward
    gX = dR3
    finish_fold_group()
    if late_flat is not None:
        publish('fold', late_flat)
    resized = sv['pos_idx'] is not None or sv['time_idx'] is not None
    dpos_eff = grads[1][0] if sv['pos_idx'] is None else E(S, D, dtype=f32)
    dtime_eff = grads[2][0] if sv['time_idx'] is None else E(T, D, dtype=f32)
    ops.embed_bwd(gX, B, T, S, dpos_eff, dtime_eff)
    grads[0].copy_(dpos_eff[0].reshape(1, 1, D))
    if sv['pos_idx'] is not None:        # nearest-resized tables (vision_tf.py:103-115): scatter back to the stored rows
        grads[1].zero_()
        grads[1][0].index_add_(0, sv['pos_idx'], dpos_eff)
    if sv['time_idx'] is not None:
        grads[2].zero_()
        grads[2][0].index_add_(0, sv['time_idx'], dtime_eff)
    if G3_next is not None:
        Gpe = G3_next                          # bf16(gX * mask0) from block 0's LayerNorm backward
    else:
        Gpe = E(M, D)
        ops.scale_cast(mode, gX, mask0, Gpe)
    dWpe = galloc(3).reshape(D, -1)
    if isinstance(sv['A_pe'], tuple):                                   # shared rgb: dW[:, :3 P^2] = (sum over the clip's queries of G)^T A_rgb
        A_rgb, A_m = sv['A_pe']
        Krgb = A_rgb.shape[1]
        Bc = A_rgb.shape[0] // (T * S)
        Gsum = E(Bc * T * S, D)
        ops.scale_cast(mode, gX.reshape(Bc, B // Bc, T * S, D).sum(dim=1).reshape(Bc * T * S, D), mask0[:Bc * T * S], Gsum)
        ops.gemm_tn(gmode, Gsum, A_rgb, dWpe[:, :Krgb])
        ops.gemm_tn(gmode, Gpe, A_m, dWpe[:, Krgb:])
    else:
        ops.gemm_tn(gmode, Gpe, sv['A_pe'], dWpe)
    grads[4].copy_(dtime_eff.sum(0))       # bias gradient = sum over all patch rows
    publish('g0', flat_cur)
    module.__dict__['pending_inv_scale'] = inv_gscale if defer_unscale else None
    if module.grad_hook is not None:
        # The collectives launched above were overlapped with the remaining backward compute; they must be complete (in
        # stream order) before autograd copies the bucket views into param.grad, so the hook is drained here.
        if hasattr(module.grad_hook, 'finish'):
            module.grad_hook.finish()
    return grads


def _live_optim(module):
    """The FusedAdamWClip attached to this module (FusedAdamWClip(..., module=net)), or None when there is none or it has been garbage-collected."""
    ref = module.__dict__.get('_optim_ref')
    return ref() if ref is not None else None


class SeekerFunction(torch.autograd.Function):
    """(module, rgb, query_mask, *params) -> (output_mask, output_flags)."""

    @staticmethod
    def forward(ctx, module, rgb, qm, *params):
        need = any(ctx.needs_input_grad[3:])
        ctx.set_materialize_grads(False)      # an unused output (e.g. output_flags in Kubric training) arrives as None, not zeros
        out_mask, flags, sv = run_forward(module, rgb, qm, params, save=need)
        ctx.module = module
        ctx.sv = sv
        ctx.params = params
        return out_mask, flags

    @staticmethod
    def backward(ctx, d_mask, d_flags):
        if ctx.sv is None:
            raise ops.L.TcowError('backward called on a forward that did not save activations')
        grads = run_backward(ctx.module, ctx.sv, ctx.params, d_mask, d_flags)
        ctx.sv = None
        out = []
        persistent = getattr(ctx.module, 'persistent_grads', False)
        for p, gr, need in zip(ctx.params, grads, ctx.needs_input_grad[3:]):
            if not need or gr is None:
                out.append(None)
                continue
            gr = gr.reshape(p.shape)
            if persistent and (p.grad is None or p.grad.data_ptr() == gr.data_ptr()):
                p.grad = gr          # delivered out of band: the same storage every step, no autograd copy
                out.append(None)
            else:
                out.append(gr)
        return (None, None, None, *out)
