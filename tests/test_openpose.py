"""OpenPose JSON wire format and utterance staging against vectors
produced by the reference's own helpers (text_pose_dataset.py load_keypoints / PoseDataset
pad-clip-to_tensor, steps/utils.py array2open_pose; tests/golden/make_golden.py)."""
import json
import os

import numpy as np
import pytest
import torch

from hand_pose_sl_amd import openpose
from conftest import GOLDEN, load_golden

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
@pytest.mark.parametrize("name", CASES)
def test_merged_file_inference_end_to_end(name, tmp_path, cuda_device):
    """The same flow with the utterance given as ONE merged file (merge_utt_jsons.py layout), with the
    fp32-grade f16x3 kernel: same predictions, written back as one merged file."""
    from hand_pose_sl_amd import infer
    rec = load_golden(name)
    frames = _frames(rec)
    merged = tmp_path / "utt_7.json"
    merged.write_text(json.dumps([{"id": f"utt_7_{i:012d}_keypoints", "data": fr} for i, fr in enumerate(frames)]))
    ckpt = tmp_path / "best_model.pth"
    torch.save({k: torch.from_numpy(v) for k, v in rec["state"].items()}, ckpt)
    out = tmp_path / "out"
    infer.main(["--data", str(merged), "--model-checkpoint", str(ckpt), "--output-folder", str(out),
                "--max-frames", str(rec["T"]), "--precision", "f16x3"])
    assert os.listdir(out) == ["utt_7.json"]
    back = openpose.load_merged_utterance(str(out / "utt_7.json"))
    n = min(int(rec["n_frames"]), rec["T"])
    assert len(back) == n and [e["id"] for e in back] == [f"utt_7_{i:012d}_keypoints" for i in range(n)]
    expect = json.loads(str(rec["out_hands_json"]))
    for e, hand, fr in zip(back, expect, frames):
        got = e["data"]["people"][0]["hand_right_keypoints_2d"]
        assert np.abs(np.array(got) - np.array(hand)).max() <= 2e-5 * 1280
        assert e["data"]["people"][0]["pose_keypoints_2d"] == fr["people"][0]["pose_keypoints_2d"]


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


# ---- merged JSON and HDF5 row (vectors from the reference's FastTextPoseDataset.load_jsons,
# select_jsons, TextPoseH5Dataset.array2item/pad/clip and order_and_reshape_toh5) -------------
KP_KEYS = ("body_kp", "body_conf", "right_hand_kp", "right_hand_conf", "left_hand_kp", "left_hand_conf")


def test_merged_json_entries_equal_reference():
    rec = np.load(os.path.join(os.path.dirname(__file__), "golden", "wire_formats.npz"))
    merged = json.loads(str(rec["merged_json"]))
    for n, sel in ((20, None), (9, "first"), (5, "first")):
        chosen, start = openpose.select_frames(merged, n, sel)
        tag = f"sel{n}"
        assert start == int(rec[tag + "_start"]) and len(chosen) == min(n, len(merged))
        item = openpose.load_utterance(chosen, len(chosen))
        assert item["json_paths"] == json.loads(str(rec[tag + "_paths"]))
        for k in KP_KEYS:
            assert np.array_equal(item[k], rec[tag + "_" + k].astype(np.float32)), (tag, k)
    with pytest.raises(ValueError, match="selection_type"):
        openpose.select_frames(merged, 5, None)            # the reference returns None here and crashes later

    class FixedRng:
        def randint(self, a, b):
            assert (a, b) == (0, len(merged) - 4)
            return 3
    crop, start = openpose.select_frames(merged, 4, "randomcrop", rng=FixedRng())
    assert start == 3 and crop == merged[3:7]


def test_merged_utterance_file_round_trip(tmp_path):
    rec = np.load(os.path.join(os.path.dirname(__file__), "golden", "wire_formats.npz"))
    merged = json.loads(str(rec["merged_json"]))
    path = tmp_path / "utt_0.json"                          # merge_utt_jsons.py layout
    path.write_text(json.dumps([{"id": f"frame_{i}", "data": e["json_data"]} for i, e in enumerate(merged)]))
    entries = openpose.load_merged_utterance(str(path))
    a = openpose.load_utterance(entries, 12)
    b = openpose.load_utterance([e["json_data"] for e in merged], 12)
    for k in KP_KEYS:
        assert np.array_equal(a[k], b[k])
    assert a["n_frames"] == 9 and a["json_paths"] == [None] * 9
    pred = np.arange(9 * 21 * 2, dtype=np.float32).reshape(9, 21, 2)
    out = openpose.write_merged_predictions(entries, pred, str(tmp_path / "out" / "utt_0.json"))
    back = openpose.load_merged_utterance(out)
    assert [e["id"] for e in back] == [f"frame_{i}" for i in range(9)]
    assert back[4]["data"]["people"][0]["hand_right_keypoints_2d"] == openpose.array2open_pose(pred[4])
    assert back[4]["data"]["people"][0]["hand_left_keypoints_2d"] == merged[4]["json_data"]["people"][0]["hand_left_keypoints_2d"]
    assert entries[4]["data"]["people"][0]["hand_right_keypoints_2d"] == merged[4]["json_data"]["people"][0]["hand_right_keypoints_2d"]
    bad = tmp_path / "bad.json"
    bad.write_text(json.dumps({"people": []}))
    with pytest.raises(ValueError, match="merged utterance"):
        openpose.load_merged_utterance(str(bad))


def test_h5_row_codec_equals_reference():
    rec = np.load(os.path.join(os.path.dirname(__file__), "golden", "wire_formats.npz"))
    row = rec["h5_row"]
    for mf in (12, 5):
        item = openpose.h5_row_to_item(row, mf)
        assert item["n_frames"] == min(7, mf)
        for k in KP_KEYS:
            assert np.array_equal(item[k], rec[f"h5_m{mf}_{k}"]), (mf, k)
        if mf > 7:                                           # zero padding, unlike the JSON path
            assert not item["body_kp"][7:].any() and not item["right_hand_conf"][7:].any()
    raw = openpose.h5_row_to_item(row)
    assert raw["body_kp"].shape == (7, 8, 2) and raw["left_hand_kp"].shape == (7, 21, 2) and raw["n_frames"] == 7
    back = openpose.item_to_h5_row(raw["body_kp"], raw["left_hand_kp"], raw["right_hand_kp"],
                                   raw["body_conf"], raw["left_hand_conf"], raw["right_hand_conf"])
    assert back.dtype == np.float32 and np.array_equal(back, row)
    assert np.array_equal(openpose.order_and_reshape_toh5(rec["h5w_in"]), rec["h5w_out"])
    with pytest.raises(ValueError):
        openpose.h5_row_to_item(np.zeros((3, 149), np.float32))
    try:
        import h5py  # noqa: F401
    except ImportError:
        with pytest.raises(RuntimeError, match="h5py"):
            openpose.read_h5_utterance("nope.h5", "utt")


def test_predict_variants_match_reference(tmp_path):
    """`--predict right_index | right_3fingers` (run.py:56-60): item builders and output writers against
    vectors from the reference's BuildIndexItem / Build3fingerItem / array2open_pose_index /
    array2open_pose_3finger (tests/golden/make_golden.py:predict_variants_case)."""
    import json
    d = np.load(os.path.join(GOLDEN, "predict_variants.npz"))
    for tag in ("right_index", "right_3fingers"):
        inp, tgt = openpose.build_item(d["body"], d["right_hand"], tag)
        assert np.array_equal(inp, d[tag + "_input_kp"]) and np.array_equal(tgt, d[tag + "_target_kp"])
    inp, tgt = openpose.build_item(d["body"], d["right_hand"])
    assert np.array_equal(inp, d["body"]) and np.array_equal(tgt, d["right_hand"])
    hand = [float(v) for v in d["hand_list"]]
    assert openpose.array2open_pose_part(list(hand), d["pred4"], "right_index") == [float(v) for v in d["out_index"]]
    assert openpose.array2open_pose_part(list(hand), d["pred12"], "right_3fingers") == [float(v) for v in d["out_3finger"]]
    with pytest.raises(ValueError):
        openpose.build_item(d["body"], d["right_hand"], "left_hand")
    # through the frame writer: only joints 5..8 change, the rest of the frame is kept
    frame = {"version": 1.3, "people": [{"pose_keypoints_2d": [0.0] * 75, "hand_left_keypoints_2d": [1.0] * 63,
                                         "hand_right_keypoints_2d": list(hand)}]}
    src = tmp_path / "in"
    src.mkdir()
    path = src / "f_000000000000_keypoints.json"
    path.write_text(json.dumps(frame))
    out = openpose.write_predictions([str(path)], d["pred4"][None], str(tmp_path / "out"), predict="right_index")
    got = json.load(open(out[0]))["people"][0]
    assert got["hand_right_keypoints_2d"] == [float(v) for v in d["out_index"]]
    assert got["hand_left_keypoints_2d"] == [1.0] * 63
