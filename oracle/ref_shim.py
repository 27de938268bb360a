"""ORACLE tooling: import the real TCOW reference from /root/reference in the build container.

Runs only where /root/reference exists (never on the GPU box, never from the product). Nothing is
copied: the reference modules are imported in place, with stub modules standing in for the optional
third-party packages its star-import hub pulls in (/root/reference/__init__.py:11-68) and which are not
installed here (cv2, imageio, wandb, torchvision, timm, fvcore, ...). None of those is on the Seeker
forward path.
"""
import os
import sys
import types
from unittest import mock

REF = '/root/reference'


class _Stub(types.ModuleType):
    def __getattr__(self, k):
        if k.startswith('__'):
            raise AttributeError(k)
        m = mock.MagicMock(name=f'{self.__name__}.{k}')
        setattr(self, k, m)
        return m


def _stub(name):
    parts = name.split('.')
    for i in range(1, len(parts) + 1):
        n = '.'.join(parts[:i])
        if n not in sys.modules:
            s = _Stub(n)
            s.__path__ = []
            sys.modules[n] = s
            if i > 1:
                setattr(sys.modules['.'.join(parts[:i - 1])], parts[i - 1], s)


class _Registry:   # stands in for fvcore.common.registry.Registry (timesformer/models/build.py:6)
    def __init__(self, name):
        pass

    def register(self, obj=None):
        return (lambda f: f) if obj is None else obj


class NullLogger:
    def __getattr__(self, k):
        return lambda *a, **k2: None


_loaded = {}


def available():
    return os.path.isdir(os.path.join(REF, 'model'))


def load():
    """Returns dict of reference modules: seeker, pipeline, loss, metrics, vit, my_utils, data_utils."""
    if _loaded:
        return _loaded
    if not available():
        raise RuntimeError('reference tree not present (this tooling only runs in the build container)')
    sys.dont_write_bytecode = True
    for n in ['cv2', 'imageio', 'lovely_numpy', 'lovely_tensors', 'seaborn', 'timm', 'wandb',
              'torch_optimizer', 'torchvision', 'torchvision.datasets', 'torchvision.io',
              'torchvision.models', 'torchvision.transforms', 'torchvision.transforms.functional',
              'torchvision.utils', 'torchvision.ops', 'torchvision.ops.focal_loss', 'fvcore',
              'fvcore.common', 'fvcore.common.registry', 'matplotlib', 'matplotlib.pyplot',
              'skimage', 'skimage.metrics', 'sklearn', 'sklearn.metrics', 'PIL', 'PIL.Image']:
        try:
            __import__(n)
        except Exception:
            _stub(n)
    if isinstance(sys.modules.get('fvcore.common.registry'), _Stub):
        sys.modules['fvcore.common.registry'].Registry = _Registry
    R = os.path.join(REF, 'third_party/TimeSformer/timesformer')
    for name, path in [('timesformer', R), ('timesformer.models', R + '/models')]:
        m = types.ModuleType(name)
        m.__path__ = [path]
        sys.modules[name] = m
    for d in ['', 'data', 'eval', 'model', 'utils', 'third_party']:
        sys.path.insert(0, os.path.join(REF, d))
    cwd = os.getcwd()
    os.chdir(REF)
    try:
        import seeker, pipeline, loss, metrics, my_utils, data_utils   # noqa: E401
        import timesformer.models.vit as vit
    finally:
        os.chdir(cwd)
    _loaded.update(seeker=seeker, pipeline=pipeline, loss=loss, metrics=metrics, vit=vit,
                   my_utils=my_utils, data_utils=data_utils)
    return _loaded


def build_reference_seeker(cfg, np_state_dict, drop_path_rate=0.0):
    """Instantiate the reference Seeker for a tcow_amd.synth config and load our synthetic weights.
    Non-{12,18,24} depths are not constructible through Seeker(network_depth=...) (vit.py:424-449), so
    for those a VisionTransformer of the requested geometry is grafted in."""
    import torch
    from functools import partial
    mods = load()
    D, depth, heads = cfg['embed_dim'], cfg['depth'], cfg['num_heads']
    std = {12: (768, 12), 18: (896, 14), 24: (1024, 16)}
    native = depth in std and std[depth] == (D, heads)
    kw = dict(num_total_frames=cfg['num_total_frames'], frame_height=cfg['frame_height'],
              frame_width=cfg['frame_width'], tracker_pretrained=False, attention_type=cfg.get('attention_type', 'divided_space_time'),
              patch_size=cfg['patch_size'], causal_attention=cfg['causal_attention'],
              norm_embeddings=cfg['norm_embeddings'], drop_path_rate=drop_path_rate,
              network_depth=depth if native else 12, track_map_stride=cfg['track_map_stride'],
              track_map_resize=cfg['track_map_resize'], query_channels=cfg['query_channels'],
              output_channels=cfg['output_channels'], flag_channels=cfg['flag_channels'])
    if not native:
        # build the smallest legal Seeker cheaply, then graft: use tiny frames for the throwaway ViT-B
        kw_small = dict(kw); kw_small.update(frame_height=cfg['patch_size'], frame_width=cfg['patch_size'], num_total_frames=1)
        net = mods['seeker'].Seeker(NullLogger(), **kw_small)
        qt = net.seeker
        for k in ('num_total_frames', 'frame_height', 'frame_width'):
            setattr(qt, k, cfg[k])
        vt = mods['vit'].VisionTransformer(
            img_size=(cfg['frame_height'], cfg['frame_width']), patch_size=cfg['patch_size'],
            in_chans=3 + cfg['query_channels'], num_classes=0, embed_dim=D, depth=depth, num_heads=heads,
            mlp_ratio=cfg['mlp_ratio'], qkv_bias=True, norm_layer=partial(torch.nn.LayerNorm, eps=1e-6),
            drop_path_rate=drop_path_rate, num_frames=cfg['num_total_frames'], attention_type=cfg.get('attention_type', 'divided_space_time'),
            causal_attention=cfg['causal_attention'])
        bb = qt.tracker_backbone
        bb.timesformer.model = vt
        bb.output_feature_dim = D
        bb.T = cfg['num_total_frames']; bb.Hf = cfg['frame_height']; bb.Wf = cfg['frame_width']
        bb.Ho = cfg['frame_height'] // cfg['patch_size']; bb.Wo = cfg['frame_width'] // cfg['patch_size']
        qt.use_feature_dim = D
        qt.tracker_post_linear = torch.nn.Linear(D, cfg['output_channels'] * cfg['patch_size'] ** 2)
        if cfg['flag_channels'] > 0:
            qt.flag_post_linear = torch.nn.Linear(D, cfg['flag_channels'])
    else:
        net = mods['seeker'].Seeker(NullLogger(), **kw)
    if cfg.get('pretrained_norm', False):
        net.seeker.tracker_backbone.pretrained = True     # enables vision_tf.py:81-89 only
    sd = {k: torch.from_numpy(v) for k, v in np_state_dict.items()}
    net.load_state_dict(sd, strict=True)
    net.eval()
    return net
