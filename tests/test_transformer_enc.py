"""TransformerEnc: numpy oracle and the Python mirror against vectors from
the reference's class (CPU); the HIP path against both (GPU).  Tolerance: |y| up to 2.4 after
four LayerNorm-ed layers -> 2e-5 max-abs for BOTH kernels (measured ~2e-6): "fp32" computes in
fp32 throughout, "f16x3" splits every Linear operand into f16 hi + lo (22 significant bits, three
f16 MFMAs per product) and is held to the same bar -- it is a cheaper fp32, not a lower precision."""
import os

import numpy as np
import pytest
import torch

import hand_pose_sl_amd as hps
import oracle
from conftest import GOLDEN

CASES = ["b2_t100", "b3_t37", "b1_t1", "b5_t16", "b2_t17"]
PRECISIONS = ["fp32", "f16x3"]
TOL = 2e-5


def _load():
    w = np.load(os.path.join(GOLDEN, "tenc_weights.npz"))
    state = {k[4:]: w[k] for k in w.files}
    c = np.load(os.path.join(GOLDEN, "tenc_cases.npz"))
    return state, {n: (c["x_" + n], c["y_" + n]) for n in CASES}


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_reference(name):
    state, cases = _load()
    x, y = cases[name]
    assert np.abs(oracle.transformer_forward(x, state) - y).max() <= 5e-6
    assert np.abs(oracle.transformer_forward(x, state, dtype=np.float64) - y).max() <= 5e-6


def test_oracle_rejects_long_sequences():
    state, _ = _load()
    with pytest.raises(RuntimeError):
        oracle.transformer_forward(np.zeros((1, 101, 12, 2), np.float32), state)


def test_mirror_state_dict_and_seeded_init_equal_reference():
    state, _ = _load()
    torch.manual_seed(41)
    m = hps.TransformerEnc(ninp=24, nhead=4, nhid=128, nout=42, nlayers=4, dropout=0.5)
    sd = m.state_dict()
    assert sorted(sd) == sorted(state)
    for k, v in sd.items():
        assert np.array_equal(v.numpy(), state[k]), k          # pe table and seeded weights bit-identical
    assert sum(p.numel() for p in m.parameters()) == 406954
    m.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()})
    if not torch.cuda.is_available():
        with torch.no_grad(), pytest.raises(RuntimeError, match="MI355X"):
            m.eval()(torch.zeros(1, 4, 12, 2))


def _gpu_model(dev, precision="fp32"):
    state, cases = _load()
    m = hps.TransformerEnc(24, 4, 128, 42, 4, precision=precision)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()})
    return m.to(dev).eval(), state, cases


@pytest.mark.gpu
@pytest.mark.parametrize("precision", PRECISIONS)
@pytest.mark.parametrize("name", CASES)
def test_hip_matches_reference(name, precision, cuda_device):
    m, state, cases = _gpu_model(cuda_device, precision)
    x, y = cases[name]
    with torch.no_grad():
        out = m(torch.from_numpy(x))                       # host tensor accepted, like ConvModel
    assert out.shape == y.shape and out.is_contiguous() and out.device.type == "cuda"
    assert np.abs(out.cpu().numpy() - y).max() <= TOL


@pytest.mark.gpu
@pytest.mark.parametrize("precision", PRECISIONS)
def test_hip_batch_independence_lengths_and_errors(precision, cuda_device):
    m, state, _ = _gpu_model(cuda_device, precision)
    g = torch.Generator().manual_seed(7)
    with torch.no_grad():
        for T in (1, 2, 15, 16, 17, 31, 33, 64, 99, 100):
            x = torch.rand((3, T, 12, 2), generator=g) - 0.5
            y = m(x.to(cuda_device))
            ref = oracle.transformer_forward(x.numpy(), state)
            assert np.abs(y.cpu().numpy() - ref).max() <= TOL, T
        x = (torch.rand((300, 100, 12, 2), generator=g) - 0.5).to(cuda_device)
        y = m(x)
        idx = [0, 7, 150, 299]
        assert torch.equal(y[idx], m(x[idx].contiguous()))           # a sequence never sees its batch neighbours
        ref = oracle.transformer_forward(x[idx].cpu().numpy(), state)
        assert np.abs(y[idx].cpu().numpy() - ref).max() <= TOL
        assert m(torch.zeros((0, 5, 12, 2))).shape == (0, 5, 21, 2)
        with pytest.raises(RuntimeError, match="max_len"):           # src + pe[:T] raises in the reference
            m(torch.zeros((1, 101, 12, 2)))
        with pytest.raises(RuntimeError):
            m(torch.zeros((1, 5, 11, 2)))
    with pytest.raises(ValueError, match="precision"):
        hps.TransformerEnc(24, 4, 128, 42, 4, precision="bf16")
    with pytest.raises(RuntimeError, match="nhid|ninp|implemented"):
        bad = hps.TransformerEnc(24, 4, 64, 42, 2).to(cuda_device).eval()
        with torch.no_grad():
            bad(torch.zeros((1, 5, 12, 2)))


@pytest.mark.gpu
def test_f16x3_refuses_parameters_outside_f16_range(cuda_device):
    state, cases = _load()
    x, y = cases["b3_t37"]
    bad = {k: torch.from_numpy(v.copy()) for k, v in state.items()}
    bad["transformer_encoder.layers.1.linear1.weight"][5, 7] = 7.0e4
    m = hps.TransformerEnc(24, 4, 128, 42, 4, precision="f16x3")
    m.load_state_dict(bad)
    m = m.to(cuda_device).eval()
    with torch.no_grad(), pytest.raises(RuntimeError, match="f16 range"):
        m(torch.from_numpy(x))
    m.precision = "fp32"                                # the exact kernel takes the same weights
    with torch.no_grad():
        assert torch.isfinite(m(torch.from_numpy(x))).all()
