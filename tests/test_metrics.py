"""maskedPoseL1 / L12Pixels: oracle vs the reference's class (golden),
HIP kernel vs both."""
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
