"""OpenPose JSON <-> keypoint arrays, the wire format on both sides of the path.

Host-side mirror of the reference's helpers:
  format_keypoints / load_keypoints   body2hand/src/dataloaders/text_pose_dataset.py:14-50
  PoseDataset.pad / clip / to_tensor  text_pose_dataset.py:145-178   (pad REPEATS frame 0)
  array2open_pose                     body2hand/src/steps/utils.py:355-364
  per-frame JSON rewrite              body2hand/src/steps/traintest.py:267-300
An OpenPose frame file holds people[0].{pose,hand_left,hand_right}_keypoints_2d as flat
[x, y, c] * N lists; the model uses 12 of the 25 BODY_25 joints.
"""
import json
import os

import numpy as np

# joints kept from BODY_25: head + arms, legs filtered out (text_pose_dataset.py:14)
BODY_HEAD_KEYPOINTS = [0, 1, 2, 3, 4, 5, 6, 7, 15, 16, 17, 18]


def format_keypoints(keypoints, n_dim=2):
    """[x1, y1, c1, x2, y2, c2, ...] -> [[x1, y1, c1], [x2, y2, c2], ...] (text_pose_dataset.py:16-26)."""
    n_dim += 1
    return [keypoints[n_dim * i:n_dim * i + n_dim] for i in range(len(keypoints) // n_dim)]


def load_keypoints(input_json):
    """One frame -> (r_hand_kp, r_hand_conf, l_hand_kp, l_hand_conf, body_kp, body_conf), the
    reference's return order (text_pose_dataset.py:29-50).  `input_json`: path or parsed dict."""
    if isinstance(input_json, str):
        with open(input_json) as f:
            data = json.load(f)
    elif isinstance(input_json, dict):
        data = input_json
    else:
        raise Exception("Input type not supported")
    person = data["people"][0]
    body = np.asarray(person["pose_keypoints_2d"], dtype=np.float64).reshape(-1, 3)[BODY_HEAD_KEYPOINTS]
    lh = np.asarray(person["hand_left_keypoints_2d"], dtype=np.float64).reshape(-1, 3)
    rh = np.asarray(person["hand_right_keypoints_2d"], dtype=np.float64).reshape(-1, 3)
    return (rh[:, :2].tolist(), rh[:, 2].tolist(), lh[:, :2].tolist(), lh[:, 2].tolist(),
            body[:, :2].tolist(), body[:, 2].tolist())


def load_utterance(frames, max_frames):
    """All frames of one utterance -> float32 arrays padded/clipped to `max_frames`.

    `frames`: list of frame JSON paths or parsed dicts, in time order.  Like the reference's
    PoseDataset (text_pose_dataset.py:100-178): keep the first `max_frames` frames, pad short
    utterances by REPEATING frame 0, n_frames = min(len, max_frames).
    Returns dict(body_kp (T,12,2), body_conf (T,12), right_hand_kp (T,21,2), right_hand_conf,
    left_hand_kp, left_hand_conf, n_frames, json_paths)."""
    if not frames:
        raise ValueError("utterance without frames")
    frames = list(frames)[:max_frames]
    cols = {k: [] for k in ("right_hand_kp", "right_hand_conf", "left_hand_kp", "left_hand_conf",
                            "body_kp", "body_conf")}
    for fr in frames:
        for k, v in zip(cols, load_keypoints(fr)):
            cols[k].append(v)
    n = len(frames)
    out = {}
    for k, v in cols.items():
        a = np.asarray(v, dtype=np.float32)
        if n < max_frames:
            a = np.concatenate([a, np.repeat(a[:1], max_frames - n, axis=0)], axis=0)
        out[k] = a
    out["n_frames"] = n
    out["json_paths"] = [f if isinstance(f, str) else None for f in frames]
    return out


def array2open_pose(array, confidence=None):
    """(21, 2) keypoints -> flat [x, y, 1.0] * 21 list of Python floats (steps/utils.py:355-364)."""
    array = np.asarray(array)
    if confidence is None:
        confidence = np.zeros((array.shape[0], 1)) + 1.0
    flat = np.reshape(np.concatenate((array, confidence), axis=1), (-1))
    return [float(x) for x in flat]


def replace_right_hand(frame_json, hand_xy):
    """The frame dict with people[0].hand_right_keypoints_2d replaced by the prediction
    (traintest.py:292-293); returns the same (mutated) dict."""
    frame_json["people"][0]["hand_right_keypoints_2d"] = array2open_pose(hand_xy)
    return frame_json


def write_predictions(frames, prediction, output_folder):
    """Re-dump every frame file with its predicted right hand (traintest.py:274-300).
    `frames`: the utterance's frame paths; `prediction`: (>= len(frames), 21, 2) in pixels."""
    os.makedirs(output_folder, exist_ok=True)
    written = []
    for i, path in enumerate(frames):
        with open(path) as f:
            data = json.load(f)
        replace_right_hand(data, prediction[i])
        out = os.path.join(output_folder, os.path.basename(path))
        with open(out, "w") as f:
            json.dump(data, f)
        written.append(out)
    return written
