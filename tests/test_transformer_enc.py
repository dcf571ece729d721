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
def test_hip_lengths_up_to_128_with_a_longer_positional_table(precision, cuda_device):
    """The C ABI takes max_len up to 128 (b2h_tenc_create); the reference's class hard-codes 100
    (HandPoseModels.py:129) but its PositionalEncoding is a plain module a user can swap.  Lengths 101 ... 128
    run the eight-tile attention instantiations (b2h_attn_qkv_h3<8> / b2h_attn_mfma_*<8>), which no other test
    reaches: seven full tiles + a partial one, exactly eight, and the tile edges around them."""
    m, state, _ = _gpu_model(cuda_device, precision)
    m.pos_encoder = hps.PositionalEncoding(24, 0.5, max_len=128).to(cuda_device)
    state = dict(state)
    state["pos_encoder.pe"] = m.pos_encoder.pe.cpu().numpy()
    g = torch.Generator().manual_seed(11)
    with torch.no_grad():
        for T in (100, 101, 111, 112, 113, 127, 128):
            x = torch.rand((5, T, 12, 2), generator=g) - 0.5
            y = m(x.to(cuda_device))
            ref = oracle.transformer_forward(x.numpy(), state)
            assert np.abs(y.cpu().numpy() - ref).max() <= TOL, T
        with pytest.raises(RuntimeError, match="max_len"):
            m(torch.zeros((1, 129, 12, 2)))


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


def _tfix():
    d = np.load(os.path.join(GOLDEN, "tenc_transforms_b6_t40.npz"))
    return {k: d[k] for k in d.files}


@pytest.mark.parametrize("tag,dif", [("dif", True), ("nodif", False)])
def test_oracle_transform_pipeline_matches_reference(tag, dif):
    """The item transforms + TransformerEnc + x1280 + mask_output of the reference (its own classes,
    tests/golden/make_golden.py:tenc_transform_case) against the oracle's restatement."""
    state, _ = _load()
    f = _tfix()
    inp, _t = oracle.preprocess(f["body"], None, dif_encoding=dif, normalize=True)
    assert np.array_equal(inp, f[tag + "_input_kp"])
    pred = oracle.transformer_forward(inp, state)
    assert np.abs(pred - f[tag + "_pred"]).max() <= 5e-6
    px = oracle.postprocess(pred, 1280.0, f["n_frames"])
    assert np.abs(px - f[tag + "_pred_px_masked"]).max() <= 5e-6 * 1280


@pytest.mark.gpu
@pytest.mark.parametrize("precision", PRECISIONS)
def test_hip_fused_transforms_match_reference(precision, cuda_device):
    """TransformerEnc.forward_fused (transforms inside the chain kernel's front and store stages)
    against the reference's transform classes + model (golden): raw pixels in, masked pixel
    predictions out; and bit-identity with the unfused kernels on pre-transformed input."""
    m, state, _ = _gpu_model(cuda_device, precision)
    f = _tfix()
    body = torch.from_numpy(f["body"]).to(cuda_device)
    nf = f["n_frames"]
    with torch.no_grad():
        for tag, dif in (("dif", True), ("nodif", False)):
            px = m.forward_fused(body, n_frames=nf, dif_encoding=dif, mask_tail=True).cpu().numpy()
            px_nomask = m.forward_fused(body, dif_encoding=dif).cpu().numpy()
            plain = m(torch.from_numpy(f[tag + "_input_kp"]).to(cuda_device))
            assert np.abs(px - f[tag + "_pred_px_masked"]).max() <= TOL * 1280
            assert np.abs(px_nomask - f[tag + "_pred_px"]).max() <= TOL * 1280
            assert np.abs(plain.cpu().numpy() - f[tag + "_pred"]).max() <= TOL
            for b, n in enumerate(nf):
                assert not px[b, n:].any()
            # the fused path computes the same bits as transform -> model -> x1280 done separately
            assert np.array_equal(px_nomask, (plain * 1280).cpu().numpy())
        # numerically neutral flags are bit-identical to the plain forward, at every length 1..100
        g = torch.Generator().manual_seed(3)
        for T in list(range(1, 34)) + [47, 48, 49, 63, 64, 65, 99, 100]:
            x = (torch.rand((3, T, 12, 2), generator=g) - 0.5).to(cuda_device)
            plain = m(x)
            a = m.forward_fused(x, dif_encoding=False, normalize=False, denormalize=True, factor=1.0)
            b = m.forward_fused(x, dif_encoding=False, normalize=False, denormalize=False, mask_tail=True, n_frames=[T] * 3)
            assert torch.equal(a, plain) and torch.equal(b, plain), T
        # ragged masks on a larger batch (frames span several workgroups), against the oracle
        rng = np.random.default_rng(5)
        bodyb = rng.random((37, 100, 12, 2), dtype=np.float32) * np.array([1280.0, 720.0], np.float32)
        nfb = rng.integers(0, 101, 37)
        out = m.forward_fused(torch.from_numpy(bodyb).to(cuda_device), n_frames=nfb, mask_tail=True).cpu().numpy()
        inp, _t = oracle.preprocess(bodyb, None)
        ref = oracle.postprocess(oracle.transformer_forward(inp, state), 1280.0, nfb)
        assert np.abs(out - ref).max() <= TOL * 1280
        with pytest.raises(ValueError):
            m.forward_fused(body, mask_tail=True)                    # needs n_frames
