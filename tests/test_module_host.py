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
        Seeker(None, attention_type='joint_space_time')
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
    torch.save({'epoch': 7, 'train_args': None, 'dset_args': {}, 'seeker_args': args, 'net_seeker': sd, 'optim_seeker': {}, 'lr_sched_seeker': {}}, path)
    net = load_tcow_checkpoint(str(path), device='cpu')
    got = net.state_dict()
    assert list(got.keys()) == list(sd.keys()) and all(torch.equal(got[k], sd[k]) for k in sd)
    assert net.seeker.causal_attention == 1 and net.seeker.tracker_pretrained is False
