"""OpenPose JSON / merged JSON / HDF5 row <-> keypoint arrays: the wire formats on both sides
of the path.

Host-side mirror of the reference's helpers:
  format_keypoints / load_keypoints   body2hand/src/dataloaders/text_pose_dataset.py:14-50
  select_jsons                        text_pose_dataset.py:52-68
  PoseDataset.pad / clip / to_tensor  text_pose_dataset.py:145-178   (pad REPEATS frame 0)
  merged-JSON frame entries           How2Sign/util_scripts/build_dataset.py:66-72 (writer),
                                      FastTextPoseDataset.load_jsons, text_pose_dataset.py:478-505
  per-utterance merged file           How2Sign/util_scripts/merge_utt_jsons.py:24-45
  HDF5 row                            TextPoseH5Dataset.array2item / pad / clip, text_pose_dataset.py:587-632;
                                      order_and_reshape_toh5, body2hand/src/steps/traintest.py:302-317
  array2open_pose                     body2hand/src/steps/utils.py:355-364
  per-frame JSON rewrite              body2hand/src/steps/traintest.py:267-300
An OpenPose frame file holds people[0].{pose,hand_left,hand_right}_keypoints_2d as flat
[x, y, c] * N lists; the model uses 12 of the 25 BODY_25 joints.
"""
import json
import os
import random

import numpy as np

# joints kept from BODY_25: head + arms, legs filtered out (text_pose_dataset.py:14)
BODY_HEAD_KEYPOINTS = [0, 1, 2, 3, 4, 5, 6, 7, 15, 16, 17, 18]


def format_keypoints(keypoints, n_dim=2):
    """[x1, y1, c1, x2, y2, c2, ...] -> [[x1, y1, c1], [x2, y2, c2], ...] (text_pose_dataset.py:16-26)."""
    n_dim += 1
    return [keypoints[n_dim * i:n_dim * i + n_dim] for i in range(len(keypoints) // n_dim)]


def frame_data(entry):
    """The parsed OpenPose frame behind one frame entry: a path, a parsed frame dict, a merged-JSON
    entry {"json_path", "json_data"} (build_dataset.py:66-72) or {"id", "data"} (merge_utt_jsons.py:38-41)."""
    if isinstance(entry, str):
        with open(entry) as f:
            return json.load(f)
    if isinstance(entry, dict):
        if "json_data" in entry:
            return entry["json_data"]
        if "data" in entry and "people" not in entry:
            return entry["data"]
        return entry
    raise Exception("Input type not supported")


def frame_path(entry):
    """The frame's file path if the entry carries one (text_pose_dataset.py:497-503), else None."""
    if isinstance(entry, str):
        return entry
    if isinstance(entry, dict):
        return entry.get("json_path")
    return None


def load_keypoints(input_json):
    """One frame -> (r_hand_kp, r_hand_conf, l_hand_kp, l_hand_conf, body_kp, body_conf), the
    reference's return order (text_pose_dataset.py:29-50).  `input_json`: path, parsed frame dict
    or merged-JSON entry (see frame_data)."""
    data = frame_data(input_json)
    person = data["people"][0]
    body = np.asarray(person["pose_keypoints_2d"], dtype=np.float64).reshape(-1, 3)[BODY_HEAD_KEYPOINTS]
    lh = np.asarray(person["hand_left_keypoints_2d"], dtype=np.float64).reshape(-1, 3)
    rh = np.asarray(person["hand_right_keypoints_2d"], dtype=np.float64).reshape(-1, 3)
    return (rh[:, :2].tolist(), rh[:, 2].tolist(), lh[:, :2].tolist(), lh[:, 2].tolist(),
            body[:, :2].tolist(), body[:, 2].tolist())


def select_frames(frames, n=100, selection_type=None, rng=random):
    """The reference's select_jsons (text_pose_dataset.py:52-68): (selected frames, index of the
    first one).  At most `n` frames: all of them when the utterance is short, else the first `n`
    ("first") or a random window ("randomcrop", `rng.randint` like the reference).  The reference
    falls off the end (returns None) for a long utterance with any other selection_type; that is
    a ValueError here."""
    frames = list(frames)
    if len(frames) <= n:
        return frames, 0
    if selection_type == "first":
        return frames[:n], 0
    if selection_type == "randomcrop":
        start = rng.randint(0, len(frames) - n)
        return frames[start:start + n], start
    raise ValueError(f"utterance of {len(frames)} frames needs selection_type 'first' or 'randomcrop', got {selection_type!r}")


def load_merged_utterance(path):
    """A per-utterance merged file written by merge_utt_jsons.py: a list of {"id", "data"} in
    frame order -> the list of frame entries load_utterance accepts."""
    with open(path) as f:
        entries = json.load(f)
    if not isinstance(entries, list) or any("data" not in e for e in entries):
        raise ValueError(f"{path}: not a merged utterance file (list of {{'id', 'data'}})")
    return entries


def load_utterance(frames, max_frames):
    """All frames of one utterance -> float32 arrays padded/clipped to `max_frames`.

    `frames`: list of frame JSON paths, parsed dicts or merged-JSON entries, in time order.  Like the reference's
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
    out["json_paths"] = [frame_path(f) for f in frames]
    return out


def write_merged_predictions(entries, prediction, out_path):
    """The merged-file counterpart of write_predictions: `entries` as read by load_merged_utterance
    (or any frame entries), `prediction` (>= len(entries), 21, 2) in pixels -> one merged file of
    {"id", "data"} records with every frame's right hand replaced."""
    out = []
    for i, e in enumerate(entries):
        data = json.loads(json.dumps(frame_data(e)))        # the caller's frames stay untouched
        replace_right_hand(data, prediction[i])
        fid = e.get("id") if isinstance(e, dict) else None
        if fid is None:
            p = frame_path(e)
            fid = os.path.basename(p).replace(".json", "") if p else f"frame_{i:012d}"
        out.append({"id": fid, "data": data})
    os.makedirs(os.path.dirname(os.path.abspath(out_path)), exist_ok=True)
    with open(out_path, "w") as f:
        json.dump(out, f)
    return out_path


# ---- HDF5 row: one dataset per utterance, (n_frames, 150) float32 ---------------------------
H5_BODY, H5_HAND = 8, 21          # joints per part: 8 body | 21 left hand | 21 right hand
H5_JOINTS = H5_BODY + 2 * H5_HAND  # 50 -> row = [x * 50 | y * 50 | c * 50]


def h5_row_to_item(row, max_frames=None):
    """TextPoseH5Dataset.array2item (+ pad + clip when `max_frames` is given; text_pose_dataset.py:587-632):
    (n_frames, 150) -> dict of body_kp (T,8,2), body_conf (T,8), left_hand_kp (T,21,2), left_hand_conf,
    right_hand_kp, right_hand_conf and n_frames = min(n_frames, max_frames).  Unlike the JSON path,
    short utterances are padded with ZEROS here (numpy.pad, :614-624).  The body part has 8 joints,
    not the 12 ConvModel is built for (HandPoseModels.py:28) -- the reference never feeds it one."""
    row = np.asarray(row)
    if row.ndim != 2 or row.shape[1] != 3 * H5_JOINTS:
        raise ValueError(f"expected an (n_frames, {3 * H5_JOINTS}) row array, got {row.shape}")
    n = row.shape[0]
    a = row.reshape((n, 3, -1)).transpose(0, 2, 1)
    kp, conf = a[:, :, :2], a[:, :, 2]
    item = {"body_kp": kp[:, :H5_BODY], "body_conf": conf[:, :H5_BODY],
            "left_hand_kp": kp[:, H5_BODY:H5_BODY + H5_HAND], "left_hand_conf": conf[:, H5_BODY:H5_BODY + H5_HAND],
            "right_hand_kp": kp[:, H5_BODY + H5_HAND:], "right_hand_conf": conf[:, H5_BODY + H5_HAND:]}
    if max_frames is not None:
        pad = max_frames - n
        for k in list(item):
            v = item[k]
            if pad > 0:
                v = np.pad(v, ((0, pad),) + ((0, 0),) * (v.ndim - 1))
            item[k] = v[:max_frames]
    item["n_frames"] = n if max_frames is None else min(n, max_frames)
    return item


def item_to_h5_row(body_kp, left_hand_kp, right_hand_kp, body_conf=None, left_hand_conf=None, right_hand_conf=None):
    """Inverse of h5_row_to_item: parts (T,8,2), (T,21,2), (T,21,2) (+ confidences, default 0) ->
    (T, 150) float32 in the layout the reference's reader expects."""
    parts = [np.asarray(p, dtype=np.float32) for p in (body_kp, left_hand_kp, right_hand_kp)]
    confs = [np.zeros(p.shape[:2], np.float32) if c is None else np.asarray(c, dtype=np.float32)
             for p, c in zip(parts, (body_conf, left_hand_conf, right_hand_conf))]
    kp = np.concatenate(parts, axis=1)            # (T, 50, 2)
    conf = np.concatenate(confs, axis=1)[..., None]
    if kp.shape[1] != H5_JOINTS:
        raise ValueError(f"expected {H5_BODY} + {H5_HAND} + {H5_HAND} joints, got {kp.shape[1]}")
    return np.concatenate([kp, conf], axis=2).transpose(0, 2, 1).reshape(kp.shape[0], -1)


def order_and_reshape_toh5(frame_prediction):
    """The reference's writer-side helper (traintest.py:302-317): (n, J, 2) -> (n, 3J) as
    [x * J | y * J | 0 * J] (confidence written as zeros)."""
    a = np.asarray(frame_prediction, dtype=np.float32)
    n = a.shape[0]
    return np.pad(a, ((0, 0), (0, 0), (0, 1))).transpose(0, 2, 1).reshape(n, -1)


def read_h5_utterance(keypoints_file, utt_id, max_frames=None):
    """One utterance of a keypoints HDF5 file (TextPoseH5Dataset.__getitem__, :645-650).  Needs
    h5py, which this image does not have: without it this raises instead of guessing."""
    try:
        import h5py
    except ImportError as exc:
        raise RuntimeError("reading HDF5 keypoint files needs h5py (not installed); the row codec "
                           "h5_row_to_item / item_to_h5_row works on arrays without it") from exc
    with h5py.File(keypoints_file, "r") as f:
        return h5_row_to_item(np.array(f.get(utt_id)), max_frames)


def array2open_pose(array, confidence=None):
    """(21, 2) keypoints -> flat [x, y, 1.0] * 21 list of Python floats (steps/utils.py:355-364)."""
    array = np.asarray(array)
    if confidence is None:
        confidence = np.zeros((array.shape[0], 1)) + 1.0
    flat = np.reshape(np.concatenate((array, confidence), axis=1), (-1))
    return [float(x) for x in flat]


# ---- the other two `--predict` choices of the reference's CLIs (run.py:56-60) -----------------------
# right_index   : input = body (12) + right hand without its index finger (17 joints), target = joints 5..8
# right_3fingers: input = right-hand joints 13..20 + joint 0 (9) + body (12), target = joints 1..12
# (steps/utils.py:215-259).  The reference's ConvModel / TransformerEnc are built for 12 input and 21
# output joints only (HandPoseModels.py:28,32; infer_utterance.py:99-101), so no model of this path
# can consume these items, there as here; the item builders and the output writers are mirrored for the
# wire format's sake.
PREDICT_TARGET_JOINTS = {"right_hand": slice(0, 21), "right_index": slice(5, 9), "right_3fingers": slice(1, 13)}


def build_item(body_kp, right_hand_kp, predict="right_hand"):
    """(input_kp, target_kp) of BuildRightHandItem / BuildIndexItem / Build3fingerItem
    (steps/utils.py:261-277, 215-236, 238-259) for (T, 12, 2) body and (T, 21, 2) right-hand arrays."""
    body_kp, right_hand_kp = np.asarray(body_kp), np.asarray(right_hand_kp)
    if predict == "right_hand":
        return body_kp, right_hand_kp
    if predict == "right_index":
        noindex = np.concatenate([right_hand_kp[:, 0:5], right_hand_kp[:, 9:]], axis=1)
        return np.concatenate([body_kp, noindex], axis=1), right_hand_kp[:, 5:9]
    if predict == "right_3fingers":
        no3 = np.concatenate([right_hand_kp[:, 13:], right_hand_kp[:, 0:1]], axis=1)
        return np.concatenate([no3, body_kp], axis=1), right_hand_kp[:, 1:13]
    raise ValueError(f"predict must be one of {sorted(PREDICT_TARGET_JOINTS)}")


def array2open_pose_part(openpose_right_hand, predicted, predict):
    """array2open_pose_index / array2open_pose_3finger (steps/utils.py:341-353, 366-381): the flat
    63-float right-hand list with the predicted joints (4 x 2 or 12 x 2, confidence 1.0) written over
    joints 5..8 or 1..12; mutates and returns the list like the reference."""
    sl = PREDICT_TARGET_JOINTS[predict]
    n = sl.stop - sl.start
    predicted = np.asarray(predicted)
    flat = np.reshape(np.concatenate((predicted, np.zeros((n, 1)) + 1.0), axis=1), (-1))
    flat = [float(x) for x in flat]
    assert len(flat) == n * 3
    openpose_right_hand[sl.start * 3:sl.stop * 3] = flat
    return openpose_right_hand


def replace_right_hand(frame_json, hand_xy, predict="right_hand"):
    """The frame dict with people[0].hand_right_keypoints_2d replaced by the prediction -- the whole
    hand, or only the predicted finger joints for `right_index` / `right_3fingers`
    (traintest.py:281-293); returns the same (mutated) dict."""
    person = frame_json["people"][0]
    if predict == "right_hand":
        person["hand_right_keypoints_2d"] = array2open_pose(hand_xy)
    else:
        person["hand_right_keypoints_2d"] = array2open_pose_part(person["hand_right_keypoints_2d"], hand_xy, predict)
    return frame_json


def write_predictions(frames, prediction, output_folder, predict="right_hand"):
    """Re-dump every frame file with its predicted right hand (traintest.py:274-300).
    `frames`: the utterance's frame paths; `prediction`: (>= len(frames), J, 2) in pixels with J = 21,
    4 (`right_index`) or 12 (`right_3fingers`)."""
    os.makedirs(output_folder, exist_ok=True)
    written = []
    for i, path in enumerate(frames):
        with open(path) as f:
            data = json.load(f)
        replace_right_hand(data, prediction[i], predict)
        out = os.path.join(output_folder, os.path.basename(path))
        with open(out, "w") as f:
            json.dump(data, f)
        written.append(out)
    return written
