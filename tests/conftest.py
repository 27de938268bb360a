import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def load_golden(name):
    """Returns (meta dict, arrays dict) of tests/golden/<name>.npz (written by oracle/make_golden.py from the reference)."""
    z = np.load(os.path.join(GOLDEN, name + '.npz'))
    meta = json.loads(bytes(z['meta']).decode()) if 'meta' in z.files else {}
    return meta, {k: z[k] for k in z.files if k != 'meta'}


def golden_inputs(meta, inst=0):
    """Regenerate the weights / clip of a fixture from its seeds (tcow_amd.synth is deterministic across machines)."""
    import torch
    from tcow_amd import synth
    cfg = meta['cfg']
    sd = synth.make_state_dict(cfg, meta['seed'])
    clip = synth.make_clip(meta['B'], cfg['num_total_frames'], cfg['frame_height'], cfg['frame_width'], seed=meta['seed'])
    rgb = torch.from_numpy(clip['rgb'])
    qm = torch.from_numpy(synth.make_query_mask(clip, inst, 0))
    return cfg, sd, rgb, qm


def build_hip_seeker(cfg, sd, precision, drop_path_rate=0.0):
    import torch
    from tcow_amd.seeker import Seeker
    net = Seeker(None, num_total_frames=cfg['num_total_frames'], frame_height=cfg['frame_height'], frame_width=cfg['frame_width'],
                 tracker_pretrained=False, patch_size=cfg['patch_size'], causal_attention=cfg['causal_attention'],
                 norm_embeddings=cfg['norm_embeddings'], drop_path_rate=drop_path_rate, network_depth=cfg['depth'],
                 track_map_stride=cfg['track_map_stride'], track_map_resize=cfg['track_map_resize'], embed_dim=cfg['embed_dim'],
                 num_heads=cfg['num_heads'], precision=precision)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    net.seeker.tracker_pretrained = cfg.get('pretrained_norm', False)   # enables only the rgb normalisation (vision_tf.py:81-89)
    return net


@pytest.fixture(scope='session')
def cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    return torch.device('cuda:0')
