"""maskedPoseL1 / poderatedPoseL1 / L12Pixels and the evaluation loop `validate`:
oracle vs the reference's classes (golden), HIP kernels vs both."""
import numpy as np
import pytest
import torch

import oracle
from conftest import GOLDEN
import os


def _rec():
    d = np.load(os.path.join(GOLDEN, "metric_b5_t60.npz"))
    return {k: d[k] for k in d.files}


def test_oracle_matches_reference_metric():
    r = _rec()
    loss, per = oracle.masked_l1(r["pred"], r["target"], r["lengths"])
    assert abs(float(loss) - float(r["loss"])) <= 1e-6
    np.testing.assert_allclose(per, r["per_seq"], rtol=2e-6)
    assert abs(float(loss) / 21 * 1280 - float(r["pixels"])) <= 1e-3


def test_oracle_empty_sequence_is_nan():
    r = _rec()
    loss, per = oracle.masked_l1(r["pred"], r["target"], [60, 0, 33, 59, 17])
    assert np.isnan(per[1]) and np.isnan(loss)


@pytest.mark.gpu
def test_hip_metric_matches_reference(cuda_device):
    import hand_pose_sl_amd as hps
    r = _rec()
    p, t = torch.from_numpy(r["pred"]).to(cuda_device), torch.from_numpy(r["target"]).to(cuda_device)
    loss, per = hps.masked_pose_l1(p, t, r["lengths"], return_per_sequence=True)
    assert loss.dim() == 0 and loss.device.type == "cuda"
    assert abs(float(loss) - float(r["loss"])) <= 2e-6
    np.testing.assert_allclose(per.cpu().numpy(), r["per_seq"], rtol=3e-6)
    assert abs(float(hps.l1_to_pixels(loss)) - float(r["pixels"])) <= 2e-3
    full = hps.masked_pose_l1(p, t)                      # lengths None == all T frames
    assert abs(float(full) - float((p - t).abs().mean())) <= 2e-6
    with pytest.raises(RuntimeError):
        hps.masked_pose_l1(p, t[:, :10])


@pytest.mark.gpu
def test_hip_metric_large_batch_vs_oracle(cuda_device):
    import hand_pose_sl_amd as hps
    g = torch.Generator().manual_seed(9)
    B, T = 513, 200
    p = torch.rand((B, T, 21, 2), generator=g)
    t = torch.rand((B, T, 21, 2), generator=g)
    n = torch.randint(1, T + 1, (B,), generator=g)
    loss, per = hps.masked_pose_l1(p.to(cuda_device), t.to(cuda_device), n, return_per_sequence=True)
    ref_loss, ref_per = oracle.masked_l1(p.numpy(), t.numpy(), n.numpy())
    np.testing.assert_allclose(per.cpu().numpy(), ref_per, rtol=5e-6)
    assert abs(float(loss) - float(ref_loss)) <= 2e-6


# ---- poderatedPoseL1 (`--loss confL1`, steps/utils.py:431-452) --------------------------------
def _conf():
    d = np.load(os.path.join(GOLDEN, "metric_conf_b5_t60.npz"))
    return {k: d[k] for k in d.files}


def test_oracle_matches_reference_weighted_metric():
    r = _conf()
    loss, per = oracle.weighted_l1(r["pred"], r["target"], r["scores"], r["lengths"])
    assert abs(float(loss) - float(r["loss"])) <= 2e-6
    np.testing.assert_allclose(per, r["per_seq"], rtol=2e-6)
    # a SUM over the batch, not a mean (the class never divides)
    assert abs(float(loss) - float(np.sum(r["per_seq"], dtype=np.float64))) <= 2e-6
    with pytest.raises(ValueError):
        oracle.weighted_l1(r["pred"], r["target"], r["scores"][:, :, :20], r["lengths"])
    loss0, per0 = oracle.weighted_l1(r["pred"], r["target"], r["scores"], [60, 0, 33, 59, 17])
    assert np.isnan(per0[1]) and np.isnan(loss0)            # mean of an empty slice, like torch


@pytest.mark.gpu
def test_hip_weighted_metric_matches_reference(cuda_device):
    import hand_pose_sl_amd as hps
    r = _conf()
    p, t = torch.from_numpy(r["pred"]).to(cuda_device), torch.from_numpy(r["target"]).to(cuda_device)
    sc = torch.from_numpy(r["scores"])                       # host tensor: moved like utils.py:439 does
    loss, per = hps.weighted_pose_l1(p, t, r["lengths"], sc, return_per_sequence=True)
    assert loss.dim() == 0 and loss.device.type == "cuda"
    assert abs(float(loss) - float(r["loss"])) <= 3e-6
    np.testing.assert_allclose(per.cpu().numpy(), r["per_seq"], rtol=3e-6)
    _, per0 = hps.weighted_pose_l1(p, t, [60, 0, 33, 59, 17], sc, return_per_sequence=True)
    assert torch.isnan(per0[1]) and not torch.isnan(per0[[0, 2, 3, 4]]).any()   # empty utterance -> NaN, like torch
    ones = torch.ones_like(sc)                              # unit confidences = B x maskedPoseL1
    assert abs(float(hps.weighted_pose_l1(p, t, r["lengths"], ones)) - 5 * float(hps.masked_pose_l1(p, t, r["lengths"]))) <= 5e-6
    with pytest.raises(RuntimeError):
        hps.weighted_pose_l1(p, t, r["lengths"], sc[:, :, :20])
    with pytest.raises(RuntimeError):
        hps.weighted_pose_l1(p, t, r["lengths"], None)
    g = torch.Generator().manual_seed(10)                   # ragged large batch vs the oracle
    B, T = 257, 150
    P, Tg, S = torch.rand((B, T, 21, 2), generator=g), torch.rand((B, T, 21, 2), generator=g), torch.rand((B, T, 21), generator=g)
    n = torch.randint(1, T + 1, (B,), generator=g)
    loss, per = hps.weighted_pose_l1(P.to(cuda_device), Tg.to(cuda_device), n, S.to(cuda_device), return_per_sequence=True)
    ref_loss, ref_per = oracle.weighted_l1(P.numpy(), Tg.numpy(), S.numpy(), n.numpy())
    np.testing.assert_allclose(per.cpu().numpy(), ref_per, rtol=5e-6)
    assert abs(float(loss) - float(ref_loss)) <= 2e-5 * max(1.0, abs(float(ref_loss)))


# ---- the evaluation loop, validate() of steps/traintest.py:168-213 ------------------------------
def _val():
    d = np.load(os.path.join(GOLDEN, "validate_loop.npz"))
    return {k: d[k] for k in d.files}


def _batches(r):
    n = int(r["meta"][0])
    return [{"body_kp": torch.from_numpy(r[f"b{i}_body_kp"]), "target_kp": torch.from_numpy(r[f"b{i}_target_kp"]),
             "n_frames": r[f"b{i}_n_frames"].tolist(), "target_conf": torch.from_numpy(r[f"b{i}_target_conf"])}
            for i in range(n)]


def test_oracle_reproduces_reference_validate_loop():
    """forward (oracle) -> criterion (oracle) -> mean over batches == the reference's validate()."""
    r = _val()
    conv = {k: r[k.replace(".", "_")] for k in ("conv1.weight", "conv1.bias", "conv2.weight", "conv2.bias",
                                                "conv3.weight", "conv3.bias", "conv4.weight", "conv4.bias")}
    w = np.load(os.path.join(GOLDEN, "tenc_weights.npz"))
    tenc = {k[4:]: w[k] for k in w.files}
    for mname, fwd in (("Conv", lambda x: oracle.forward_from_state(x, conv)), ("TransformerEnc", lambda x: oracle.transformer_forward(x, tenc))):
        for lname in ("L1", "confL1"):
            vals = []
            for b in _batches(r):
                y = fwd(b["body_kp"].numpy())
                if lname == "L1":
                    vals.append(float(oracle.masked_l1(y, b["target_kp"].numpy(), b["n_frames"])[0]))
                else:
                    vals.append(float(oracle.weighted_l1(y, b["target_kp"].numpy(), b["target_conf"].numpy(), b["n_frames"])[0]))
            assert abs(np.mean(vals) - float(r[f"loss_{mname}_{lname}"])) <= 3e-6, (mname, lname)


def test_validate_accepts_the_reference_call_forms():
    """Argument handling of validate() (no GPU needed): args.loss wins, then the keyword, then the criterion's class."""
    from types import SimpleNamespace

    import hand_pose_sl_amd as hps
    from hand_pose_sl_amd.evaluate import _loss_name
    assert _loss_name(hps.maskedPoseL1(), None, None) == "L1" and _loss_name(hps.poderatedPoseL1(), None, None) == "confL1"
    assert _loss_name(hps.maskedPoseL1(), SimpleNamespace(loss="confL1"), None) == "confL1"
    assert _loss_name("confL1", None, None) == "confL1" and _loss_name(None, None, None) == "L1"
    assert _loss_name(None, None, "confL1") == "confL1"
    assert _loss_name(torch.nn.MSELoss(), None, None) == "MSELoss"      # refused by validate(): not an evaluation loss


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["fp32", "f16x3"])
def test_hip_validate_loop_matches_reference(precision, cuda_device):
    import hand_pose_sl_amd as hps
    r = _val()
    conv = hps.ConvModel(30, "ReLU", False, precision=precision)
    conv.load_state_dict({k: torch.from_numpy(r[k.replace(".", "_")]) for k in conv.state_dict()})
    w = np.load(os.path.join(GOLDEN, "tenc_weights.npz"))
    tenc = hps.TransformerEnc(24, 4, 128, 42, 4, precision=precision)
    tenc.load_state_dict({k[4:]: torch.from_numpy(w[k]) for k in w.files})
    for mname, model in (("Conv", conv.to(cuda_device)), ("TransformerEnc", tenc.to(cuda_device))):
        for lname in ("L1", "confL1"):
            got = hps.validate(model, _batches(r), loss=lname)          # host batches, like the reference's loader
            assert abs(got - float(r[f"loss_{mname}_{lname}"])) <= 5e-6, (mname, lname, got)
    # ... and called the way steps/traintest.py:136 calls it: validate(model, val_loader, criterion, device, args)
    from types import SimpleNamespace
    for mname, model in (("Conv", conv), ("TransformerEnc", tenc)):
        for lname, crit in (("L1", hps.maskedPoseL1()), ("confL1", hps.poderatedPoseL1())):
            got = hps.validate(model, _batches(r), crit, torch.device("cuda"), SimpleNamespace(model=mname, loss=lname))
            assert abs(got - float(r[f"loss_{mname}_{lname}"])) <= 5e-6, (mname, lname, got)
    assert hps.validate(conv, _batches(r), hps.poderatedPoseL1()) == hps.validate(conv, _batches(r), loss="confL1")
    with pytest.raises(ValueError):      # traintest.py:199-200
        hps.validate(conv, _batches(r), hps.maskedPoseL1(), None, SimpleNamespace(model="TextPoseTransformer", loss="L1"))
    val, pix = hps.validate(conv, _batches(r), return_pixels=True)
    assert abs(pix - val / 21 * 1280) <= 1e-9
    with pytest.raises(ValueError):
        hps.validate(conv, _batches(r), loss="MSE")
    with pytest.raises(ZeroDivisionError):
        hps.validate(conv, [])
