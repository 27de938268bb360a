"""bf16 error budget of the Seeker forward: which of the bf16 roundings of precision='bf16' produce the mask-logit
deviation from the fp32 reference?  Emulated on the CPU oracle: every tensor the HIP engine keeps in bf16 is rounded to bf16 (and back)
at the same point of the computation -- GEMM operands (activations and weight copies), im2col pixels, LayerNorm outputs, qkv, the
attention probabilities fed to P.V, attention / projection / GELU outputs, the head input and output -- with f32 accumulation everywhere,
as the kernels do.  Then one class at a time is left in f32.  Prints max|d| / mean|d| of the pooled mask logits against the all-f32 run.

    python tools/bf16_budget.py [--small]          (full BASELINE configs[1] geometry: ~20 s per variant on 8 cores)
"""
import argparse, sys, time
import torch
sys.path.insert(0, '.')
from oracle import seeker_oracle as so
from tcow_amd import synth

CLASSES = ['weights', 'pixels', 'ln_out', 'qkv', 'probs', 'attn_out', 'proj_out', 'gelu_out', 'head_in', 'head_out']


def bf(x):
    return x.to(torch.bfloat16).to(torch.float32)


def forward(sd, cfg, rgb, qm, rnd):
    """so.seeker_forward with rounding hooks: a restatement of the same op sequence (divided space-time, eval) with `rnd` = set of classes."""
    R = lambda name, x: bf(x) if name in rnd else x
    P = so.PREFIX
    B, _, T, Hf, Wf = rgb.shape
    Pp = cfg['patch_size']; D = cfg['embed_dim']; heads = cfg['num_heads']; ca = cfg['causal_attention']
    Hp, Wp = Hf // Pp, Wf // Pp; N = Hp * Wp; d = D // heads
    W = lambda k: R('weights', sd[k])
    x_in = torch.cat([rgb, qm], 1)
    patches = R('pixels', x_in.reshape(B, 4, T, Hp, Pp, Wp, Pp).permute(0, 2, 3, 5, 1, 4, 6).reshape(B, T, N, 4 * Pp * Pp))
    X = patches @ W(P + 'patch_embed.proj.weight').reshape(D, -1).t() + sd[P + 'patch_embed.proj.bias']
    pos = sd[P + 'pos_embed']; te = sd[P + 'time_embed']
    CLS = (sd[P + 'cls_token'][0, 0] + pos[0, 0]).unsqueeze(0).expand(B, D).clone()
    X = X + pos[0, 1:][None, None] + te[0][None, :, None]

    def attn(x, pre, mask):
        qkv = R('qkv', x @ W(pre + 'qkv.weight').t() + sd[pre + 'qkv.bias'])
        Bq, L, _ = x.shape
        q, k, v = qkv.reshape(Bq, L, 3, heads, d).permute(2, 0, 3, 1, 4)
        a = (q @ k.transpose(-2, -1)) * d ** -0.5
        if mask is not None:
            a = a.masked_fill(~mask, -1e10)
        a = a.softmax(-1)
        # the kernels normalise AFTER the P.V product with the f32 row sum; P itself (unnormalised, <= 1) is what gets rounded
        mx = a.max(-1, keepdim=True)[0]
        p = R('probs', a / mx)
        o = (p @ v) * (mx / 1.0)
        o = R('attn_out', o.transpose(1, 2).reshape(Bq, L, D))
        return o @ W(pre + 'proj.weight').t() + sd[pre + 'proj.bias']

    keep = so.causal_keep_mask(T, ca) if ca > 0 else None
    for i in range(cfg['depth']):
        b = P + f'blocks.{i}.'
        U = R('ln_out', so.layer_norm(X, sd[b + 'temporal_norm1.weight'], sd[b + 'temporal_norm1.bias']))
        Rr = R('proj_out', attn(U.permute(0, 2, 1, 3).reshape(B * N, T, D), b + 'temporal_attn.', keep))
        Rr = Rr @ W(b + 'temporal_fc.weight').t() + sd[b + 'temporal_fc.bias']
        Xt = X + Rr.reshape(B, N, T, D).permute(0, 2, 1, 3)
        Vin = torch.cat([CLS[:, None, None, :].expand(B, T, 1, D), Xt], 2)
        V = R('ln_out', so.layer_norm(Vin, sd[b + 'norm1.weight'], sd[b + 'norm1.bias']))
        Y = attn(V.reshape(B * T, N + 1, D), b + 'attn.', None).reshape(B, T, N + 1, D)
        cls_out = Y[:, 0, 0] if ca == 1 else Y[:, :, 0].mean(1)
        X = Xt + Y[:, :, 1:]; CLS = CLS + cls_out
        allt = torch.cat([CLS[:, None], X.reshape(B, T * N, D)], 1)
        h = R('ln_out', so.layer_norm(allt, sd[b + 'norm2.weight'], sd[b + 'norm2.bias']))
        h = R('gelu_out', so.gelu_erf(h @ W(b + 'mlp.fc1.weight').t() + sd[b + 'mlp.fc1.bias']))
        allt = allt + (h @ W(b + 'mlp.fc2.weight').t() + sd[b + 'mlp.fc2.bias'])
        CLS = allt[:, 0]; X = allt[:, 1:].reshape(B, T, N, D)
    Co = cfg['output_channels']
    Pm = R('head_out', R('head_in', X) @ W('seeker.tracker_post_linear.weight').t() + sd['seeker.tracker_post_linear.bias'])
    M = Pm.reshape(B, T, Hp, Wp, Co, Pp, Pp).permute(0, 4, 1, 2, 5, 3, 6).reshape(B, Co, T, Hf, Wf)
    st = cfg['track_map_stride']
    return torch.nn.functional.avg_pool2d(M.permute(0, 2, 1, 3, 4).reshape(B * T, Co, Hf, Wf), st, st)   # pooled logits: the bilinear x4 is a convex combination of these


def main():
    ap = argparse.ArgumentParser(); ap.add_argument('--small', action='store_true'); args = ap.parse_args()
    cfg = synth.seeker_config(num_total_frames=8, frame_height=96, frame_width=128, causal_attention=1) if args.small else synth.seeker_config(causal_attention=1)
    sd = so.to_torch_state_dict(synth.make_state_dict(cfg, 900))
    T, H, W = cfg['num_total_frames'], cfg['frame_height'], cfg['frame_width']
    clip = synth.make_clip(1, T, H, W, seed=900)
    rgb = torch.from_numpy(clip['rgb']); qm = torch.from_numpy(synth.make_query_mask(clip, 0, 0))
    with torch.no_grad():
        t0 = time.time(); ref = forward(sd, cfg, rgb, qm, set()); print(f'# f32 run {time.time() - t0:.1f} s; pooled logit std {ref.std():.4f}', flush=True)
        taps = {}
        so.seeker_forward(sd, cfg, rgb, qm, taps=taps)
        chk = taps['pooled'].reshape(ref.shape)
        print(f'# restatement-with-hooks vs oracle (no rounding): {float((ref - chk).abs().max()):.2e}')
        rows = [('all bf16 classes (= precision bf16)', set(CLASSES))] + [(f'all but {c} (kept f32)', set(CLASSES) - {c}) for c in CLASSES] + [(f'only {c}', {c}) for c in CLASSES]
        for name, rnd in rows:
            out = forward(sd, cfg, rgb, qm, rnd)
            dlt = (out - ref).abs()
            print(f'{name:42s} max|d| {float(dlt.max()):.3e}   mean|d| {float(dlt.mean()):.3e}', flush=True)


if __name__ == '__main__':
    main()
