"""The oracle (CPU restatement) against the golden vectors produced by the
reference's own classes (tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

import oracle
from conftest import CONV_CASES, load_golden

TOL_F32 = 2e-6   # fp32 summation-order noise; measured <= 1.3e-7


def _pick(y, rec):
    return y[rec["y_idx"]] if "y_idx" in rec else y


@pytest.mark.parametrize("name", CONV_CASES)
def test_c_oracle_matches_reference(name):
    rec = load_golden(name)
    y = oracle.forward_from_state(rec["x"], rec["state"], pos_emb=rec["pos_emb"])
    assert y.shape == (rec["B"], rec["T"], 21, 2) and y.dtype == np.float32
    assert np.abs(_pick(y, rec) - rec["y"]).max() <= TOL_F32
    if "y_row_sum" in rec:  # every element of the big batches, through float64 checksums
        np.testing.assert_allclose(y.astype(np.float64).sum(axis=(1, 2, 3)), rec["y_row_sum"], atol=1e-4)
        assert abs(np.abs(y.astype(np.float64)).sum() - float(rec["y_abs_sum"])) < 1e-2


@pytest.mark.parametrize("name", CONV_CASES)
def test_torch_port_is_bit_exact(name):
    rec = load_golden(name)
    st = {k: torch.from_numpy(v) for k, v in rec["state"].items()}
    y = oracle.torch_forward(torch.from_numpy(rec["x"]), st, rec["pos_emb"]).numpy()
    assert np.array_equal(_pick(y, rec), rec["y"])


def test_acc64_agrees():
    rec = load_golden("cfg1_b1_t200")
    y32 = oracle.forward_from_state(rec["x"], rec["state"])
    y64 = oracle.forward_from_state(rec["x"], rec["state"], acc64=True)
    assert np.abs(y32 - y64).max() < 1e-6


@pytest.mark.parametrize("mode,bound", [("bf16", 2e-3), ("f16", 3e-4)])
def test_reduced_precision_models_are_close(mode, bound):
    rec = load_golden("cfg2_b64_t200_u55")
    y = oracle.forward_from_state(rec["x"][:4], rec["state"], mode=mode)
    ref = oracle.forward_from_state(rec["x"][:4], rec["state"])
    err = np.abs(y - ref).max()
    assert 0 < err < bound


def test_per_layer_zero_padding_not_input_padding():
    """Padding the raw input by 8 and running 'valid' convs is NOT the model
    (SURVEY.md section 7): the first/last 8 frames differ.  Guard the oracle against it."""
    rec = load_golden("cfg1_b1_t200")
    x = rec["x"]
    big = np.zeros((1, 216, 12, 2), np.float32)
    big[:, 8:208] = x
    y_big = oracle.forward_from_state(big, rec["state"])[:, 8:208]
    y = oracle.forward_from_state(x, rec["state"])
    assert np.abs(y_big[:, 8:-8] - y[:, 8:-8]).max() < 1e-6   # interior identical
    assert np.abs(y_big[:, :8] - y[:, :8]).max() > 1e-3       # borders are not


def test_pos_emb_requires_t100():
    rec = load_golden("posemb_b2_t100")
    with pytest.raises(RuntimeError):
        oracle.forward_from_state(np.zeros((1, 200, 12, 2), np.float32), rec["state"], pos_emb=True)


def test_transforms_match_reference():
    rec = load_golden("transforms_b6_t40")
    inp, tgt = oracle.preprocess(rec["body"], rec["right_hand"], dif_encoding=True, normalize=True)
    assert np.array_equal(inp, rec["input_kp"])
    assert np.array_equal(tgt, rec["target_kp"])
    pred = oracle.forward_from_state(inp, rec["state"])
    assert np.abs(pred - rec["pred"]).max() <= TOL_F32
    px = oracle.postprocess(rec["pred"], 1280.0)
    assert np.array_equal(px, rec["pred_px"])
    masked = oracle.postprocess(rec["pred"], 1280.0, rec["n_frames"])
    assert np.array_equal(masked, rec["pred_px_masked"])
    for b, n in enumerate(rec["n_frames"]):
        assert not masked[b, n:].any()


def test_batch_independence_and_empty_batch():
    rec = load_golden("edge_b3_t33")
    y = oracle.forward_from_state(rec["x"], rec["state"])
    y1 = oracle.forward_from_state(rec["x"][1:2], rec["state"])
    assert np.array_equal(y[1:2], y1)
    y0 = oracle.forward_from_state(np.zeros((0, 5, 12, 2), np.float32), rec["state"])
    assert y0.shape == (0, 5, 21, 2)
