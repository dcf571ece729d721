"""OpenPose JSON wire format and utterance staging against vectors
produced by the reference's own helpers (text_pose_dataset.py load_keypoints / PoseDataset
pad-clip-to_tensor, steps/utils.py array2open_pose; tests/golden/make_golden.py)."""
import json
import os

import numpy as np
import pytest
import torch

from hand_pose_sl_amd import openpose
from conftest import load_golden

CASES = ["openpose_short_n7_m12", "openpose_long_n30_m20"]


def _frames(rec):
    return json.loads(str(rec["frames_json"]))


def test_format_keypoints():
    assert openpose.format_keypoints([1, 2, .5, 3, 4, .6]) == [[1, 2, .5], [3, 4, .6]]
    assert openpose.format_keypoints([]) == []
    assert openpose.BODY_HEAD_KEYPOINTS == [0, 1, 2, 3, 4, 5, 6, 7, 15, 16, 17, 18]


@pytest.mark.parametrize("name", CASES)
def test_staging_equals_reference_dataset(name):
    rec = load_golden(name)
    frames = _frames(rec)
    item = openpose.load_utterance(frames, rec["T"])
    assert item["n_frames"] == min(int(rec["n_frames"]), rec["T"])
    for k in ("body_kp", "body_conf", "right_hand_kp", "right_hand_conf", "left_hand_kp", "left_hand_conf"):
        assert item[k].dtype == np.float32
        assert np.array_equal(item[k], rec["staged_" + k]), k
    if int(rec["n_frames"]) < rec["T"]:      # padding repeats frame 0 (text_pose_dataset.py:145-153)
        assert np.array_equal(item["body_kp"][-1], item["body_kp"][0])
    r, rc, l, lc, b, bc = openpose.load_keypoints(frames[0])
    assert len(r) == 21 and len(l) == 21 and len(b) == 12 and len(bc) == 12
    assert b[8] == frames[0]["people"][0]["pose_keypoints_2d"][15 * 3:15 * 3 + 2]   # legs filtered out
    with pytest.raises(Exception):
        openpose.load_keypoints(3)
    with pytest.raises(ValueError):
        openpose.load_utterance([], 5)


@pytest.mark.parametrize("name", CASES)
def test_output_writer_equals_reference(name):
    rec = load_golden(name)
    expect = json.loads(str(rec["out_hands_json"]))
    for i, hand in enumerate(expect):
        got = openpose.array2open_pose(rec["pred_px"][i])
        assert got == hand and len(got) == 63 and all(isinstance(v, float) for v in got)
        assert got[2::3] == [1.0] * 21


def test_write_predictions_round_trip(tmp_path):
    rec = load_golden(CASES[0])
    frames = _frames(rec)
    src = tmp_path / "utt"
    src.mkdir()
    paths = []
    for i, fr in enumerate(frames):
        p = src / f"utt_{i:012d}_keypoints.json"
        p.write_text(json.dumps(fr))
        paths.append(str(p))
    out = openpose.write_predictions(paths, rec["pred_px"], str(tmp_path / "out"))
    assert [os.path.basename(o) for o in out] == [os.path.basename(p) for p in paths]
    expect = json.loads(str(rec["out_hands_json"]))
    for o, fr, hand in zip(out, frames, expect):
        d = json.load(open(o))
        assert d["people"][0]["hand_right_keypoints_2d"] == hand
        assert d["people"][0]["pose_keypoints_2d"] == fr["people"][0]["pose_keypoints_2d"]      # untouched
        assert d["people"][0]["hand_left_keypoints_2d"] == fr["people"][0]["hand_left_keypoints_2d"]


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_utterance_inference_end_to_end(name, tmp_path, cuda_device):
    """infer_utterance.py:52-111 + traintest.py:214-300 flow through the CLI entry point."""
    from hand_pose_sl_amd import infer
    rec = load_golden(name)
    frames = _frames(rec)
    src = tmp_path / "utt"
    src.mkdir()
    for i, fr in enumerate(frames):
        (src / f"utt_{i:012d}_keypoints.json").write_text(json.dumps(fr))
    ckpt = tmp_path / "best_model.pth"
    torch.save({k: torch.from_numpy(v) for k, v in rec["state"].items()}, ckpt)
    out = tmp_path / "out"
    infer.main(["--data", str(src), "--model-checkpoint", str(ckpt), "--output-folder", str(out),
                "--max-frames", str(rec["T"])])
    files = sorted(os.listdir(out))
    n = min(int(rec["n_frames"]), rec["T"])
    assert len(files) == n
    expect = json.loads(str(rec["out_hands_json"]))
    for f, hand in zip(files, expect):
        got = json.load(open(out / f))["people"][0]["hand_right_keypoints_2d"]
        assert np.abs(np.array(got) - np.array(hand)).max() <= 2e-5 * 1280      # fp32 kernel, pixels
    with pytest.raises(Exception, match="already exists"):                      # infer_utterance.py:55-56
        infer.main(["--data", str(src), "--model-checkpoint", str(ckpt), "--output-folder", str(out)])


@pytest.mark.gpu
def test_cli_with_transformer_enc(tmp_path, cuda_device):
    """--model TransformerEnc (infer_utterance.py:99-101): frames in, frames out, values equal the
    oracle's transformer on the same staged, normalised input."""
    import oracle
    from hand_pose_sl_amd import infer
    rec = load_golden(CASES[0])
    frames = _frames(rec)
    src = tmp_path / "utt"
    src.mkdir()
    for i, fr in enumerate(frames):
        (src / f"utt_{i:012d}_keypoints.json").write_text(json.dumps(fr))
    w = np.load(os.path.join(os.path.dirname(__file__), "golden", "tenc_weights.npz"))
    state = {k[4:]: w[k] for k in w.files}
    ckpt = tmp_path / "tenc.pth"
    torch.save({k: torch.from_numpy(v) for k, v in state.items()}, ckpt)
    out = tmp_path / "out"
    infer.main(["--data", str(src), "--model", "TransformerEnc", "--model-checkpoint", str(ckpt),
                "--output-folder", str(out), "--max-frames", str(rec["T"])])
    item = openpose.load_utterance(frames, rec["T"])
    ref = oracle.transformer_forward(item["body_kp"][None] / np.float32(1280), state)[0] * np.float32(1280)
    files = sorted(os.listdir(out))
    assert len(files) == item["n_frames"]
    for i, f in enumerate(files):
        got = np.array(json.load(open(out / f))["people"][0]["hand_right_keypoints_2d"]).reshape(21, 3)
        assert np.abs(got[:, :2] - ref[i]).max() <= 2e-5 * 1280 and (got[:, 2] == 1.0).all()
