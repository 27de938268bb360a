"""CPU: host-side behaviour of the drop-in module (constructor contract, state dict, error surface, checkpoint surgery)."""
import numpy as np
import pytest
import torch

from tcow_amd import synth
from tcow_amd._lib import TcowError
from tcow_amd.checkpoint import pretrained_surgery
from tcow_amd.seeker import QueryMaskTracker, Seeker


class _Log:
    def __init__(self): self.lines = []
    def info(self, m): self.lines.append(m)


def test_state_dict_keys_and_shapes_match_reference_layout():
    net = Seeker(_Log(), num_total_frames=30, frame_height=240, frame_width=320, causal_attention=1)
    sd = net.state_dict()
    shapes = synth.state_dict_shapes(synth.seeker_config())
    assert list(sd.keys()) == list(shapes.keys()) and len(sd) == 251
    assert all(tuple(sd[k].shape) == tuple(v) for k, v in shapes.items())
    assert sum(p.numel() for p in net.parameters()) == 122145027
    net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict(synth.seeker_config(), 1).items()}, strict=True)


def test_constructor_contract():
    log = _Log()
    net = Seeker(log, num_total_frames=4, frame_height=32, frame_width=32, network_depth=12, tracker_pretrained='0')
    assert net.seeker.tracker_pretrained is False and 'tracker_pretrained: False' in log.lines[0]      # mask_tracker.py:55-69
    with pytest.raises(ValueError):
        Seeker(None, network_depth=7)                                                                    # vit.py:449
    with pytest.raises(ValueError):
        Seeker(None, tracker_pretrained=3)                                                               # mask_tracker.py:67
    with pytest.raises(AssertionError):
        Seeker(None, frame_height=100, frame_width=64)                                                   # mask_tracker.py:89
    with pytest.raises(TcowError):
        Seeker(None, attention_type='space_only')                                                        # vision_tf.py:127 needs time_embed, vit.py:263-265 does not create it
    with pytest.raises(AssertionError):
        Seeker(None, attention_type='nope')                                                              # vit.py:133
    joint = Seeker(None, num_total_frames=4, frame_height=32, frame_width=32, attention_type='joint_space_time', causal_attention=0)
    assert not any('temporal' in k for k in joint.state_dict()) and len(joint.state_dict()) == 155           # vit.py:140-146: no temporal parameters
    with pytest.raises(TcowError):
        Seeker(None, tracker_pretrained='1')          # ImageNet weights need the network (vit.py:35)


def test_stock_init_matches_reference_rules():
    net = QueryMaskTracker(None, num_total_frames=4, frame_height=32, frame_width=32)
    v = net.vit
    assert float(v.time_embed.detach().abs().max()) == 0.0                                   # vit.py:268
    assert all(float(b.temporal_fc.weight.detach().abs().max()) == 0.0 for b in v.blocks)    # vit.py:289-297 (all blocks)
    assert float(v.blocks[0].attn.qkv.bias.detach().abs().max()) == 0.0 and float(v.blocks[0].norm1.weight.detach().min()) == 1.0
    assert abs(float(v.pos_embed.detach().std()) - 0.02) < 2e-3                      # trunc_normal_(std=.02) with ABSOLUTE cut-offs +-2 (vit_utils.py:25-76)


def test_forward_fails_loudly_without_gpu():
    net = Seeker(None, num_total_frames=4, frame_height=32, frame_width=32)
    with pytest.raises(TcowError):
        net(torch.zeros(1, 3, 4, 32, 32), torch.zeros(1, 1, 4, 32, 32))
    with pytest.raises(AssertionError):
        net(torch.zeros(1, 3, 4, 32, 32), torch.zeros(1, 2, 4, 32, 32))            # mask_tracker.py:105


def test_pretrained_surgery_rules():
    """helpers.py:117-199 on a toy ViT state dict."""
    D, P = 8, 4
    sd = {'cls_token': torch.zeros(1, 1, D), 'pos_embed': torch.arange(5 * D, dtype=torch.float32).reshape(1, 5, D),
          'patch_embed.proj.weight': torch.ones(D, 3, P, P), 'patch_embed.proj.bias': torch.zeros(D),
          'blocks.0.attn.qkv.weight': torch.full((3 * D, D), 2.0), 'blocks.0.norm1.weight': torch.full((D,), 3.0),
          'head.weight': torch.zeros(10, D), 'head.bias': torch.zeros(10)}
    out = pretrained_surgery(sd, in_chans=4, num_patches=6, num_frames=5)
    assert out['patch_embed.proj.weight'].shape == (D, 4, P, P) and torch.allclose(out['patch_embed.proj.weight'], torch.full((D, 4, P, P), 0.75))
    assert 'head.weight' not in out
    assert out['pos_embed'].shape == (1, 7, D)
    src = (np.floor(np.arange(6) * (4 / 6))).astype(int) + 1                       # F.interpolate(mode='nearest') source index
    assert torch.equal(out['pos_embed'][0, 1:], sd['pos_embed'][0, src])
    assert torch.equal(out['blocks.0.temporal_attn.qkv.weight'], sd['blocks.0.attn.qkv.weight'])
    assert torch.equal(out['blocks.0.temporal_norm1.weight'], sd['blocks.0.norm1.weight'])


def test_reference_checkpoint_round_trip(tmp_path):
    """A checkpoint written the way train.py:269-304 writes it loads through eval/inference.py:38-54's recipe."""
    from tcow_amd.checkpoint import load_tcow_checkpoint
    args = dict(num_total_frames=4, num_visible_frames=4, frame_height=32, frame_width=48, tracker_pretrained='1', attention_type='divided_space_time',
                patch_size=16, causal_attention=1, norm_embeddings=False, drop_path_rate=0.1, network_depth=12, track_map_stride=4,
                track_map_resize='bilinear', query_channels=1, output_channels=3, flag_channels=3)
    cfg = synth.seeker_config(num_total_frames=4, frame_height=32, frame_width=48, causal_attention=1)
    sd = {k: torch.from_numpy(v) for k, v in synth.make_state_dict(cfg, 3).items()}
    path = tmp_path / 'checkpoint.pth'
    import argparse
    train_args = argparse.Namespace(name='v1', num_frames=4, learn_rate=1e-4)         # train.py:269-275 pickles the argparse Namespace
    torch.save({'epoch': 7, 'train_args': train_args, 'dset_args': {'x': 1}, 'seeker_args': args, 'net_seeker': sd, 'optim_seeker': {}, 'lr_sched_seeker': {}}, path)
    net = load_tcow_checkpoint(str(path), device='cpu')
    got = net.state_dict()
    assert list(got.keys()) == list(sd.keys()) and all(torch.equal(got[k], sd[k]) for k in sd)
    # a model trained from the ImageNet ViT ('1') keeps its rgb normalisation at eval time (eval/inference.py:44-52 passes seeker_args through)
    assert net.seeker.causal_attention == 1 and net.seeker.tracker_pretrained is True
    args['tracker_pretrained'] = 'false'
    torch.save({'epoch': 7, 'train_args': train_args, 'dset_args': {}, 'seeker_args': args, 'net_seeker': sd, 'optim_seeker': {}, 'lr_sched_seeker': {}}, path)
    assert load_tcow_checkpoint(str(path), device='cpu').seeker.tracker_pretrained is False


def test_pretrained_surgery_matches_reference_load_pretrained(tmp_path):
    """tests/golden/g10_pretrained.npz holds the state dict the reference's own helpers.load_pretrained (helpers.py:100-205) produced
    from a toy image-ViT checkpoint file; ours must give the same tensors key for key, through the same file formats."""
    import sys
    from conftest import ROOT, load_golden
    sys.path.insert(0, ROOT)
    from oracle.make_golden_r2 import toy_vit_checkpoint
    meta, g = load_golden('g10_pretrained')
    toy = toy_vit_checkpoint(**meta['toy'])
    cfg = meta['cfg']
    for wrap in ('plain', 'state_dict', 'model'):
        path = tmp_path / f'vit_{wrap}.pth'
        torch.save(toy if wrap == 'plain' else {wrap: toy}, path)
        torch.manual_seed(0)
        net = Seeker(None, num_total_frames=cfg['num_total_frames'], frame_height=cfg['frame_height'], frame_width=cfg['frame_width'], tracker_pretrained=str(path),
                     causal_attention=1, network_depth=cfg['depth'], embed_dim=cfg['embed_dim'], num_heads=cfg['num_heads'])
        assert net.seeker.tracker_pretrained is True and net.seeker.pretrained_path == str(path)
        got = net.seeker.vit.state_dict()
        changed = 0
        for k, ref in ((k[4:], v) for k, v in g.items() if k.startswith('sd::')):
            if bool(g['changed::' + k]):                                   # tensors the reference's load overwrote
                assert np.array_equal(got[k].numpy(), ref), (wrap, k)
                changed += 1
            else:                                                           # untouched by the load: stock init on both sides (time_embed, temporal_fc)
                assert k.endswith('time_embed') or 'temporal_fc' in k, k
                assert float(got[k].abs().max()) == 0.0
        assert changed > 20


def test_plugin_usage_modes_and_items():
    """data_utils.py:301-342 / data_plugin.py:141-199: usage modes equal the reference's own (golden g9), item assembly by known answers."""
    from conftest import load_golden
    from tcow_amd import plugin_data as pd
    _, g = load_golden('g9_cfg4_eval')
    i = 0
    while f'modes_{i}' in g:
        a = g[f'modes_{i}_args'].tolist()
        n_in, nf, qt, mtc = a[:4]
        rest = a[5:]; cut = rest.index(-1)
        got = pd.get_usage_modes(range(n_in), rest[:cut], rest[cut + 1:], nf, qt, min_target_frames_covered=mtc)
        assert np.allclose(np.asarray(got, dtype=np.float64).reshape(-1, 3), g[f'modes_{i}']), i
        i += 1
    assert i == 5
    rgb = np.arange(3 * 20 * 2 * 2, dtype=np.float32).reshape(3, 20, 2, 2)
    q = {4: np.ones((2, 2), np.uint8)}
    sn = {4: np.ones((2, 2), np.uint8), 9: np.zeros((2, 2), np.uint8), 11: np.ones((2, 2), np.uint8)}
    oc = {11: np.ones((2, 2), np.uint8)}
    it = pd.build_plugin_item(rgb, q, sn, oc, {}, frame_start=2, frame_stride=2, num_frames=6, query_time_idx=1)
    assert it['frame_inds'] == [2, 4, 6, 8, 10, 12] and np.array_equal(it['pv_rgb_tf'], rgb[:, [2, 4, 6, 8, 10, 12]])
    assert it['pv_query_tf'].sum() == 4 and it['pv_query_tf'][0, 1].all()
    tg = it['pv_target_tf']
    # snitch: round((t - start) / stride) -> t=4 -> 1, t=9 -> round(3.5) = 4 (banker's), t=11 -> round(4.5) = 4 (last write wins); occl: (11 - 2) // 2 = 4
    assert (tg[0, 1] == 1).all() and (tg[0, 4] == 1).all() and (tg[0, [0, 2, 3, 5]] == -1).all()
    assert (tg[1, 4] == 1).all() and (tg[1, [0, 1, 2, 3, 5]] == -1).all() and (tg[2] == -1).all()


def test_checkpoint_written_here_resumes_the_reference_way(tmp_path):
    """train.py:269-304 / :246-257: save_tcow_checkpoint writes the reference's dictionary (+ its side files); the reference's resume recipe
    -- load_state_dict of net / torch.optim.AdamW / MultiStepLR from 'net_seeker' / 'optim_seeker' / 'lr_sched_seeker' -- accepts it, and
    eval/inference.py's recipe (load_tcow_checkpoint) rebuilds the module from 'seeker_args'."""
    import argparse
    from tcow_amd.checkpoint import load_tcow_checkpoint, resume_tcow_checkpoint, save_tcow_checkpoint
    from tcow_amd.optim import FusedAdamWClip
    args = dict(num_total_frames=4, num_visible_frames=4, frame_height=32, frame_width=48, tracker_pretrained='0', attention_type='divided_space_time',
                patch_size=16, causal_attention=1, norm_embeddings=False, drop_path_rate=0.1, network_depth=2, track_map_stride=4,
                track_map_resize='bilinear', query_channels=1, output_channels=3, flag_channels=3, embed_dim=64, num_heads=1)      # (explicit small geometry: a host-logic test)
    net = Seeker(None, **args)
    opt = FusedAdamWClip(list(net.parameters()), lr=1e-4, max_norm=0.3)
    sched = torch.optim.lr_scheduler.MultiStepLR(opt, [3, 6], gamma=0.3)
    for p_ in list(net.parameters())[:4]:                                  # give the optimizer some state in torch.optim.AdamW's layout
        opt.state[p_] = {'step': torch.tensor(5.0), 'exp_avg': torch.full_like(p_, 0.25), 'exp_avg_sq': torch.full_like(p_, 0.5)}
    sched.last_epoch = 4
    path = save_tcow_checkpoint(str(tmp_path), 4, net, opt, sched, seeker_args=args, train_args=argparse.Namespace(name='run'), dset_args={'n': 1}, name='run')
    assert (tmp_path / 'checkpoint_epoch.txt').read_text().strip() == '4' and (tmp_path / 'checkpoint_name.txt').read_text().strip() == 'run'
    ck = torch.load(path, map_location='cpu', weights_only=False)
    assert sorted(ck) == ['dset_args', 'epoch', 'lr_sched_seeker', 'net_seeker', 'optim_seeker', 'seeker_args', 'train_args']
    # the reference's resume (train.py:246-257) with stock torch objects
    net2 = Seeker(None, **args)
    ref_opt = torch.optim.AdamW(net2.parameters(), lr=1e-4); ref_sched = torch.optim.lr_scheduler.MultiStepLR(ref_opt, [3, 6], gamma=0.3)
    net2.load_state_dict(ck['net_seeker']); ref_opt.load_state_dict(ck['optim_seeker']); ref_sched.load_state_dict(ck['lr_sched_seeker'])
    assert all(torch.equal(a, b) for a, b in zip(net.state_dict().values(), net2.state_dict().values()))
    first = next(iter(net2.parameters()))
    assert float(ref_opt.state[first]['exp_avg'].flatten()[0]) == 0.25 and ref_sched.last_epoch == 4
    # ... and ours
    net3 = Seeker(None, **args); opt3 = FusedAdamWClip(list(net3.parameters()), lr=1e-4); sched3 = torch.optim.lr_scheduler.MultiStepLR(opt3, [3, 6], gamma=0.3)
    assert resume_tcow_checkpoint(path, net3, opt3, sched3) == 5 and opt3.step_count == 5
    net4 = load_tcow_checkpoint(path, device='cpu')
    assert all(torch.equal(a, b) for a, b in zip(net.state_dict().values(), net4.state_dict().values()))
