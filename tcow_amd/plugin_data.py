"""Plugin-video clip sampling: the index logic that sits between a raw video and the Seeker in the reference's eval path
("batched multi-query / multi-stride inference").

* `get_usage_modes`  -- data/data_utils.py:301-342: every (frame_start, frame_stride in 1..10, target_coverage) under which a
  video can be sub-sampled into a `num_frames` clip with the query annotation at clip position `query_time`.
* `build_plugin_item` -- data/data_plugin.py:141-199: rgb frames at the strided indices, the query mask placed at the query
  frame only, sparse int8 targets with -1 on unlabeled frames.  (Image decoding / resizing of the reference loader is out of
  scope; frames arrive as arrays.)
* `eval_items`       -- all (query, stride) items of one video; `SeekerPipeline.forward_plugin_items` runs them as ONE batch.

Pure numpy host code, no GPU involved.  Pinned against the reference's own `get_usage_modes` in tests/golden/g9_cfg4_eval.npz.
"""
import numpy as np


def get_usage_modes(available_input_inds, available_query_inds, available_target_inds, num_frames, query_time, min_target_frames_covered=2):
    inputs = sorted(set(available_input_inds)); queries = sorted(set(available_query_inds)); targets = set(available_target_inds)
    in_set = set(inputs)
    last_input = max(inputs)
    modes = []
    for query_idx in queries:
        for stride in range(1, 11):
            first = query_idx - query_time * stride
            last = first + (num_frames - 1) * stride
            if first < 0 or last > last_input:
                continue
            covered = sum(1 for f in range(first, last + 1, stride) if f in in_set and f in targets)
            if covered >= min_target_frames_covered:
                modes.append((first, stride, covered / num_frames))
    return modes


def build_plugin_item(rgb, query_frames, snitch_frames, occl_frames, cont_frames, frame_start, frame_stride, num_frames, query_time_idx):
    """rgb (3, Tv, H, W) float32 in [0, 1]; *_frames: {video frame index: (H, W) 0/1 mask}.  Returns the tensors of one dataset
    item (data_plugin.py:216-233): pv_rgb_tf (3,T,H,W) f32, pv_query_tf (1,T,H,W) u8, pv_target_tf (3,T,H,W) i8."""
    T = num_frames
    inds = list(range(frame_start, frame_start + T * frame_stride, frame_stride))
    H, W = rgb.shape[-2:]
    pv_rgb = np.ascontiguousarray(rgb[:, inds])
    pv_query = np.zeros((1, T, H, W), np.uint8)
    pv_query[0, query_time_idx] = query_frames[inds[query_time_idx]]
    pv_target = -np.ones((3, T, H, W), np.int8)
    for t, v in snitch_frames.items():                                    # data_plugin.py:181-184: round((t - start) / stride)
        f = int(round((t - frame_start) / frame_stride))
        if 0 <= f < T:
            pv_target[0, f] = v
    for ch, frames in ((1, occl_frames), (2, cont_frames)):               # data_plugin.py:185-192: floor division for these two
        for t, v in frames.items():
            f = int(round((t - frame_start) // frame_stride))
            if 0 <= f < T:
                pv_target[ch, f] = v
    return dict(pv_rgb_tf=pv_rgb, pv_query_tf=pv_query, pv_target_tf=pv_target, frame_start=frame_start, frame_stride=frame_stride, frame_inds=inds)


def eval_items(video, num_frames=30, query_time_idx=0, queries=(0,), strides=None, min_target_frames_covered=0):
    """All (query instance, usage mode) items of a synthetic plugin video (tcow_amd.synth.make_plugin_video): for every query
    instance its visible mask at the annotated query frame(s) is the query, its amodal mask on the annotated frames the snitch
    target.  `strides` restricts the usage modes to those strides (BASELINE configs[4]: 1..4)."""
    Tv = video['rgb'].shape[1]
    items = []
    for q in queries:
        query_frames = {t: (video['segm'][t] == q + 1).astype(np.uint8) for t in video['query_frames']}
        snitch = {t: video['div'][q, t] for t in video['annot_frames']}
        modes = get_usage_modes(range(Tv), query_frames.keys(), snitch.keys(), num_frames, query_time_idx, min_target_frames_covered)
        for (start, stride, cov) in modes:
            if strides is not None and stride not in strides:
                continue
            it = build_plugin_item(video['rgb'], query_frames, snitch, {}, {}, start, stride, num_frames, query_time_idx)
            it.update(query=q, target_coverage=cov)
            items.append(it)
    return items
