"""ORACLE tooling: golden values for the caller rows P / L / M (pipeline, loss, metrics) from the REAL reference.

Runs the reference's MyTrainPipeline.forward + MyLosses.per_example / entire_batch (pipeline.py:50-258, loss.py:238-421,
eval/metrics.py:9-113) on a synthetic Kubric-shaped batch and stores what they produce.  Build container only:
    python -m oracle.make_golden_pipeline
"""
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_shim                      # noqa: E402
from oracle.make_golden import OUT, SEED         # noqa: E402
from tcow_amd import synth                       # noqa: E402
from tcow_amd.tcow_loss import DEFAULT_ARGS, hard_negative_band   # noqa: E402


def gaussian_blur_reference(x, k, sigma):
    """torchvision.transforms.functional.gaussian_blur semantics (separable, reflect padding), written out here because
    torchvision is not installed: used only to confirm that `blur > 0` equals a k x k box dilation (loss.py:136-146)."""
    half = (k - 1) * 0.5
    pts = torch.linspace(-half, half, k)
    pdf = torch.exp(-0.5 * (pts / sigma) ** 2)
    k1 = (pdf / pdf.sum()).to(x.dtype)
    x4 = torch.nn.functional.pad(x[:, None], (k // 2, k // 2, k // 2, k // 2), mode='reflect')
    x4 = torch.nn.functional.conv2d(x4, k1.view(1, 1, 1, k))
    x4 = torch.nn.functional.conv2d(x4, k1.view(1, 1, k, 1))
    return x4[:, 0]


def main():
    mods = ref_shim.load()
    cfg = synth.seeker_config(num_total_frames=4, frame_height=64, frame_width=64, embed_dim=256, depth=2, num_heads=4, causal_attention=1)
    B, Qs = 2, 3
    net = ref_shim.build_reference_seeker(cfg, synth.make_state_dict(cfg, SEED))
    args = SimpleNamespace(num_frames=4, num_queries=Qs, **{**DEFAULT_ARGS, 'hard_negative_factor': 1.0})   # >1 needs torchvision (loss.py:140)
    arrays = {}
    for phase in ('test', 'train'):
        data = synth.to_torch_tree(synth.make_kubric_batch(B, 4, 64, 64, seed=SEED))
        data['within_batch_idx'] = torch.arange(B)
        pipe = mods['pipeline'].MyTrainPipeline(args, ref_shim.NullLogger(), {'seeker': net}, 'cpu')
        np.random.seed(SEED)
        pipe.set_phase(phase)                                   # NB: flips the global grad mode (pipeline.py:38-47)
        for progress in (0.0, 0.5):
            np.random.seed(SEED)
            model_retval, loss_retval = pipe(data, 0, 0, 0, progress, True, False)
            per_ex = {k: (v.detach().clone() if torch.is_tensor(v) else v) for k, v in loss_retval.items() if k != 'metrics'}
            metrics = {k: v.clone() for k, v in loss_retval['metrics'].items()}
            final = pipe.process_entire_batch(data, model_retval, loss_retval, 0, 0, 0, progress)
            tag = f'{phase}_p{int(progress * 10)}'
            for k in ('track', 'occl_mask', 'cont_mask'):
                arrays[f'{tag}::{k}'] = np.float64(per_ex[k].item())
            arrays[f'{tag}::total_seeker'] = np.float64(final['total_seeker'].item())
            for k, v in metrics.items():
                arrays[f'{tag}::metric::{k}'] = v.numpy()
            if progress == 0.0:
                for k in ('sel_query_inds', 'sel_occl_fracs', 'sel_desirability', 'seeker_query_mask', 'snitch_occl_by_ptr', 'full_occl_cont_id',
                          'target_mask', 'output_mask', 'snitch_weights'):
                    arrays[f'{phase}::{k}'] = model_retval[k].detach().numpy()
                if phase == 'train':
                    final['total_seeker'].backward()
                    named = dict(net.named_parameters())
                    arrays['train::grad_norm_total'] = np.float64(sum(float(p.grad.norm()) ** 2 for p in named.values() if p.grad is not None) ** 0.5)
                    arrays['train::grad::seeker.tracker_post_linear.bias'] = named['seeker.tracker_post_linear.bias'].grad.numpy().copy()
                    net.zero_grad()
        torch.set_grad_enabled(True)
    # hard-negative band: the reference's gaussian_blur(...) > 0 equals our box dilation
    tm = torch.from_numpy(arrays['test::target_mask'])[:, :, 0]             # (B,Q,T,H,W)
    k = int(np.sqrt(64 * 64) / 12.0); k += (k % 2 == 0)
    blur = gaussian_blur_reference(tm.reshape(-1, 64, 64), k, float(k)).reshape(tm.shape) > 0.0
    blur[tm >= 0.5] = False
    band = hard_negative_band(tm, 64, 64)
    assert torch.equal(blur, band), 'box dilation != gaussian_blur > 0'
    arrays['hard_negative_band_sum'] = np.int64(band.sum().item())
    np.savez_compressed(os.path.join(OUT, 'g5_pipeline_cfg1.npz'), **arrays)
    print('wrote g5_pipeline_cfg1.npz', {k: float(v) for k, v in arrays.items() if '::' in k and np.ndim(v) == 0 and 'metric' not in k})


if __name__ == '__main__':
    main()
