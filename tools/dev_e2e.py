"""Dev harness (GPU box): end-to-end Seeker forward/backward on libtcow_hip vs the CPU oracle."""
import sys, time, torch, numpy as np
sys.path.insert(0, '.')
from tcow_amd import synth
from tcow_amd.seeker import Seeker
from oracle import seeker_oracle as so

def build(cfg, precision, seed=900):
    net = Seeker(None, num_total_frames=cfg['num_total_frames'], frame_height=cfg['frame_height'], frame_width=cfg['frame_width'],
                 tracker_pretrained=False, patch_size=cfg['patch_size'], causal_attention=cfg['causal_attention'],
                 norm_embeddings=cfg['norm_embeddings'], drop_path_rate=0.0, network_depth=cfg['depth'],
                 track_map_stride=cfg['track_map_stride'], track_map_resize=cfg['track_map_resize'], embed_dim=cfg['embed_dim'],
                 num_heads=cfg['num_heads'], precision=precision)
    sd = synth.make_state_dict(cfg, seed)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    net.seeker.tracker_pretrained = cfg['pretrained_norm']
    return net.cuda(), sd

def run_case(name, cfg, B, precision, check_grad=True, drop=None):
    torch.manual_seed(0)
    net, sd = build(cfg, precision)
    T, H, W = cfg['num_total_frames'], cfg['frame_height'], cfg['frame_width']
    clip = synth.make_clip(B, T, H, W, seed=900)
    rgb = torch.from_numpy(clip['rgb']); qm = torch.from_numpy(synth.make_query_mask(clip, 0, 0))
    osd = {k: torch.from_numpy(v).double().requires_grad_(check_grad) for k, v in sd.items()}
    dm = None
    if drop:
        N = (H // 16) * (W // 16)
        gen = torch.Generator().manual_seed(5)
        dm = {}
        for i in range(cfg['depth']):
            dm[(i, 'temporal')] = ((torch.rand(B, N, generator=gen) > 0.3).float(), 0.3)
            dm[(i, 'spatial')] = ((torch.rand(B, T, generator=gen) > 0.3).float(), 0.3)
            dm[(i, 'mlp')] = (torch.ones(B) if i == 0 else (torch.rand(B, generator=gen) > 0.3).float(), 0.3)
        net.seeker.forced_drop_masks = dm
    om_ref, fl_ref = so.seeker_forward(osd, cfg, rgb.double(), qm.double(), drop_masks=dm)
    net.train(check_grad)
    t0 = time.time()
    om, fl = net(rgb.cuda(), qm.cuda())
    torch.cuda.synchronize(); t1 = time.time()
    e_m = (om.cpu().double() - om_ref).abs().max().item(); e_f = (fl.cpu().double() - fl_ref).abs().max().item()
    line = f'{name:34s} {precision} B={B} logits max|d|={e_m:.2e} (std {om_ref.std().item():.3f}) flags max|d|={e_f:.2e}'
    if check_grad:
        gen = torch.Generator().manual_seed(3)
        Gm = torch.randn(om_ref.shape, generator=gen).double(); Gf = torch.randn(fl_ref.shape, generator=gen).double()
        loss_ref = (om_ref * Gm).sum() + (fl_ref * Gf).sum()
        keys = list(osd.keys())
        gref = torch.autograd.grad(loss_ref, [osd[k] for k in keys], allow_unused=True)
        loss = (om * Gm.float().cuda()).sum() + (fl * Gf.float().cuda()).sum()
        loss.backward()
        torch.cuda.synchronize()
        worst = (0, None); allrel = []
        named = dict(net.named_parameters())
        for k, gr in zip(keys, gref):
            gp = named[k].grad
            if gr is None:
                assert gp is None or gp.abs().max() == 0, k
                continue
            assert gp is not None, f'missing grad {k}'
            rel = ((gp.cpu().double() - gr).abs().max() / (gr.abs().max() + 1e-12)).item()
            allrel.append(rel)
            if rel > worst[0]: worst = (rel, k)
        line += f' | grads: worst rel={worst[0]:.2e} ({worst[1].split("model.")[-1] if worst[1] else None}) median={np.median(allrel):.2e}'
    print(line, flush=True)
    return e_m

base = dict(num_total_frames=4, frame_height=64, frame_width=64, embed_dim=256, depth=2, num_heads=4)
for prec in ('fp32', 'bf16'):
    run_case('c1 ca=1', synth.seeker_config(**base, causal_attention=1), 2, prec)
    run_case('c1 ca=0', synth.seeker_config(**base, causal_attention=0), 2, prec)
    run_case('c1 ca=2', synth.seeker_config(**base, causal_attention=2), 1, prec)
    run_case('c1 ca=3', synth.seeker_config(**base, causal_attention=3), 1, prec)
    run_case('c1 ca=-1', synth.seeker_config(**base, causal_attention=-1), 1, prec)
    run_case('c1 norm_emb nearest', synth.seeker_config(**base, causal_attention=1, norm_embeddings=True, track_map_resize='nearest'), 1, prec)
    run_case('c1 stride1 prenorm', synth.seeker_config(**base, causal_attention=1, track_map_stride=1, pretrained_norm=True), 1, prec)
    run_case('c1 droppath', synth.seeker_config(**base, causal_attention=1), 2, prec, drop=True)
    run_case('mid T=8 96x128 D=768 d=2', synth.seeker_config(num_total_frames=8, frame_height=96, frame_width=128, embed_dim=768, depth=2, num_heads=12, causal_attention=1), 1, prec)
print('done')
