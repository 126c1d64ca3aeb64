"""Online keyframe / source-view selection in front of the depth hot path -- the streaming counterpart of the
precomputed test tuples (SURVEY.md §8f "next" #4).  API-compatible with the reference's tools/keyframe_buffer.py
(`KeyframeBuffer`, `DVMVS_Config`, `pose_distance`; DeepVideoMVS heuristics) and with the source ordering of
datasets/generic_mvs_dataset.py:643-659.  Host-side control logic on numpy, like the reference's: a handful of 4x4
operations per frame, nothing here belongs on the GPU.

    buf = KeyframeBuffer(DVMVS_Config.test_keyframe_buffer_size, DVMVS_Config.test_keyframe_pose_distance,
                         DVMVS_Config.test_optimal_t_measure, DVMVS_Config.test_optimal_R_measure, store_return_indices=True)
    if buf.try_new_keyframe(world_T_cam, image, index=i) == KeyframeBuffer.ADDED:
        sources = buf.get_best_measurement_frames(7)      # [(pose, image, index), ...]
"""
import numpy as np


class DVMVS_Config:
    """Tuple settings of the reference (tools/keyframe_buffer.py:12-22)."""
    train_minimum_pose_distance = 0.125
    train_maximum_pose_distance = 0.325
    train_crawl_step = 3
    test_keyframe_buffer_size = 30
    test_keyframe_pose_distance = 0.1
    test_optimal_t_measure = 0.15
    test_optimal_R_measure = 0.0


def is_pose_available(pose):
    """False when tracking delivered NaN / inf (keyframe_buffer.py:24-31)."""
    return bool(np.isfinite(pose).all())


def pose_distance(reference_pose, measurement_pose):
    """DVMVS pose measures between two camera-to-world poses (keyframe_buffer.py:53-69):
    returns (combined, R_measure, t_measure) with R = sqrt(2 (1 - min(3, tr R) / 3)), t = |t|."""
    rel = np.dot(np.linalg.inv(reference_pose), measurement_pose)
    R_measure = np.sqrt(2 * (1 - min(3.0, np.trace(rel[:3, :3])) / 3))
    t_measure = np.linalg.norm(rel[:3, 3])
    return np.sqrt(t_measure ** 2 + R_measure ** 2), R_measure, t_measure


def is_valid_pair(reference_pose, measurement_pose, pose_dist_min, pose_dist_max, t_norm_threshold=0.05,
                  return_measure=False):
    """Training-tuple validity test (keyframe_buffer.py:33-51)."""
    combined, _, t_measure = pose_distance(reference_pose, measurement_pose)
    ok = bool(pose_dist_min <= combined <= pose_dist_max and t_measure >= t_norm_threshold)
    return (ok, combined) if return_measure else ok


class KeyframeBuffer:
    """Bounded FIFO of keyframes with the DVMVS admission rule and source-view choice (keyframe_buffer.py:71-186).

    try_new_keyframe() return codes (the reference's): 0 first frame stored, 1 new keyframe stored (predict depth),
    2 not enough motion, 3 buffer reset (tracking lost / gap in valid frames), 4 still lost, 5 pose missing (waiting)."""
    FIRST, ADDED, TOO_CLOSE, RESET, LOST, WAITING = 0, 1, 2, 3, 4, 5
    LOST_AFTER = 30  # frames without a pose (about a second) before the buffer is dropped

    def __init__(self, buffer_size, keyframe_pose_distance, optimal_t_score, optimal_R_score, store_return_indices):
        self.buffer_size = buffer_size
        self.keyframe_pose_distance = keyframe_pose_distance
        self.optimal_t_score = optimal_t_score
        self.optimal_R_score = optimal_R_score
        self._with_index = store_return_indices
        self._frames = []          # oldest first; entries (pose, image[, index])
        self._missing = 0

    # the reference exposes its deque as `.buffer`
    @property
    def buffer(self):
        return self._frames

    def _push(self, pose, image, index):
        self._frames.append((pose, image, index) if self._with_index else (pose, image))
        if len(self._frames) > self.buffer_size:
            del self._frames[0]

    def calculate_penalty(self, t_score, R_score):
        """Squared distance from the optimal baseline; too-short baselines cost 5x (keyframe_buffer.py:90-98)."""
        t_diff = t_score - self.optimal_t_score
        t_penalty = (5.0 if t_diff < 0.0 else 1.0) * np.abs(t_diff) ** 2.0
        return np.abs(R_score - self.optimal_R_score) ** 2.0 + t_penalty

    def try_new_keyframe(self, pose, image, dist_to_last_valid=None, index=None):
        if self._with_index and index is None:
            raise ValueError("Storing and returning the frame indices is requested in the constructor, but "
                             "index=None is passed to the function")
        if dist_to_last_valid is not None and dist_to_last_valid > 30:   # gap in the valid-frame list
            self._frames.clear()
            self._missing = 0
            self._push(pose, image, index)
            return self.RESET
        if not is_pose_available(pose):
            self._missing += 1
            if self._missing <= self.LOST_AFTER:
                return self.WAITING
            if self._frames:
                self._frames.clear()
                return self.RESET
            return self.LOST
        self._missing = 0
        if not self._frames:
            self._push(pose, image, index)
            return self.FIRST
        combined, _, _ = pose_distance(pose, self._frames[-1][0])
        if combined >= self.keyframe_pose_distance:
            self._push(pose, image, index)
            return self.ADDED
        return self.TOO_CLOSE

    def get_best_measurement_frames(self, n_requested_measurement_frames):
        """The n buffered frames whose baseline to the newest keyframe is closest to the optimum (unordered, as
        np.argpartition returns them -- the dataset orders sources afterwards, see sort_sources_by_pose_penalty)."""
        reference_pose = self._frames[-1][0]
        candidates = self._frames[:-1]
        n = min(n_requested_measurement_frames, len(candidates))
        penalties = []
        for frame in candidates:
            _, R_measure, t_measure = pose_distance(reference_pose, frame[0])
            penalties.append(self.calculate_penalty(t_measure, R_measure))
        chosen = np.argpartition(penalties, n - 1)[:n]
        return [candidates[i] for i in chosen]


def default_dvmvs_tuples(scan, poses, dists_to_last_valid, n_measurement_frames):
    """The reference's default ("online") test tuples for one scan -- the list its evaluation runs on
    (data_scripts/generate_test_tuples.py:159-212): every pose is offered to a KeyframeBuffer with the DVMVS test
    settings; each accepted keyframe yields {"scan": scan, "indices": [keyframe, source, ...]} with up to
    n_measurement_frames earlier keyframes chosen by baseline penalty.  `dists_to_last_valid[i]` = frames since the
    last valid pose (None to let the buffer handle invalid poses itself)."""
    cfg = DVMVS_Config
    buf = KeyframeBuffer(cfg.test_keyframe_buffer_size, cfg.test_keyframe_pose_distance, cfg.test_optimal_t_measure,
                         cfg.test_optimal_R_measure, store_return_indices=True)
    samples = []
    for i, pose in enumerate(poses):
        if buf.try_new_keyframe(np.array(pose, copy=True), None, dists_to_last_valid[i], index=i) != buf.ADDED:
            continue
        sources = [frame[2] for frame in buf.get_best_measurement_frames(n_measurement_frames)]
        samples.append({"scan": scan, "indices": [i] + sources})
    return samples


def write_tuple_file(path, samples):
    """One line per tuple, `scan frame_id_0 frame_id_1 ...` with the reference frame first -- the file format the
    reference's datasets read (generate_test_tuples.py:1-8)."""
    with open(path, "w") as f:
        for s in samples:
            f.write(" ".join([str(s["scan"])] + [str(i) for i in s["indices"]]) + "\n")


def sort_sources_by_pose_penalty(cur_cam_T_world, src_world_T_cam):
    """Order of the source views as the reference's dataset feeds them to the model: ascending combined pose
    distance of cur_cam_T_src_cam (generic_mvs_dataset.py:643-659 with utils/geometry_utils.pose_distance :178-191).
    cur_cam_T_world [4,4], src_world_T_cam [K,4,4] (numpy or torch, fp32) -> list of K indices."""
    cur = np.asarray(cur_cam_T_world, dtype=np.float32)
    src = np.asarray(src_world_T_cam, dtype=np.float32)
    rel = cur[None] @ src                                        # cur_cam_T_src_cam, fp32 like the dataset's tensors
    tr = np.trace(rel[:, :3, :3], axis1=1, axis2=2)
    r_m = np.sqrt(2 * (1 - np.minimum(np.float32(3.0), tr) / 3))
    t_m = np.linalg.norm(rel[:, :3, 3], axis=1)
    return np.argsort(np.sqrt(t_m ** 2 + r_m ** 2), kind="stable").tolist()
