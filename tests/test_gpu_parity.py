"""Parity of the HIP path (through the C ABI) with the reference, on a real
MI355X.  Golden vectors come from the reference's own classes; the oracle is
the checker for inputs the fixtures do not hold.

Tolerances (north_star: <= 1e-3 max-abs in fp32 vs the reference CPU forward):
  fp32 kernels  (VALU, exact-fp32 MFMA)  : 2e-5   (measured ~1e-7; summation order only)
  f16x3 (f16 hi/lo split)                : the same 2e-5 bar as fp32
  bf16 MFMA                              : **1e-3** (north_star's bar) vs the fp32 reference on every
      normalised-keypoint input -- U[0,1] ("u01", pixel/1280-like) and U[-.5,.5] ("u55",
      dif-encoded-like), which is what the path is fed (measured 3.7-5.0e-4) -- AND 6e-4 vs the
      oracle's bf16-operand model (fp32 summation order flips an occasional bf16 rounding of an
      intermediate activation by one ulp = 2^-9 relative; measured <= 2e-4).
      DOCUMENTED EXCEEDANCE: on unnormalised N(0,1) inputs ("randn" fixtures: |x| up to 4.5, which
      keypoints/1280 never reach) bf16 operand rounding itself gives 1.047e-3 (the oracle's
      bf16-operand model measures the same against the reference), i.e. 5 % over north_star's bar.
      Those fixtures are held to the named bound BF16_RANDN_BOUND = 1.1e-3, not to a blanket
      looser tolerance; callers that need <= 1e-3 on such inputs use f16 (1.2e-4) or f16x3 / fp32.
  f16 MFMA                               : 2.5e-4 vs fp32 reference, 8e-5 vs the f16 model
"""
import numpy as np
import pytest
import torch

import hand_pose_sl_amd as hps
import oracle
from conftest import CONV_CASES, load_golden

pytestmark = pytest.mark.gpu

TOL = {"f32_valu": 2e-5, "f32_mfma": 2e-5, "f16x3": 2e-5, "bf16": 1e-3, "f16": 2.5e-4}   # f16x3: the fp32 bar
BF16_RANDN_BOUND = 1.1e-3   # bf16 on N(0,1) inputs only (see the docstring): measured 1.047e-3
TOL_MODEL = {"bf16": 6e-4, "f16": 8e-5}   # vs the oracle's operand-rounding model
ORACLE_MODE = {"bf16": "bf16", "f16": "f16"}
ALL_PREC = ["f32_valu", "f32_mfma", "f16x3", "bf16", "f16"]


def _model(rec, prec, dev):
    m = hps.ConvModel(rec["C"], "ReLU", rec["pos_emb"], precision=prec)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in rec["state"].items()})
    return m.to(dev).eval()


def _supported(rec, prec):
    """Every kernel runs every width 1..64 (33..64: the wide variants -- bf16 / f16 kernel_mfma16w.h,
    f16x3 kernel_mfma3w.h, exact fp32 Geo32<true> in kernel_mfma.h); 65..128 is the VALU kernel's alone."""
    return rec["C"] <= 64 or prec == "f32_valu"


def _tol(rec, prec):
    """north_star's 1e-3 for bf16 on normalised inputs; the named exceedance bound on randn fixtures."""
    if prec == "bf16" and str(rec.get("kind")) == "randn":
        return BF16_RANDN_BOUND
    return TOL[prec]


@pytest.mark.parametrize("prec", ALL_PREC)
@pytest.mark.parametrize("name", CONV_CASES)
def test_golden(name, prec, cuda_device):
    rec = load_golden(name)
    if not _supported(rec, prec):
        m = _model(rec, prec, cuda_device)
        with torch.no_grad(), pytest.raises(RuntimeError, match="conv_channels"):
            m(torch.from_numpy(rec["x"]).to(cuda_device))
        return
    m = _model(rec, prec, cuda_device)
    x = torch.from_numpy(rec["x"]).to(cuda_device)
    with torch.no_grad():
        y = m(x)
    assert y.shape == (rec["B"], rec["T"], 21, 2) and y.dtype == torch.float32 and y.is_contiguous()
    y = y.cpu().numpy()
    ys = y[rec["y_idx"]] if "y_idx" in rec else y
    err = np.abs(ys - rec["y"]).max()
    tol = _tol(rec, prec)
    assert err <= tol, f"{name}/{prec}: max-abs {err:.3e} (bar {tol:.1e}, input kind {rec.get('kind')})"
    # first / last 8 frames carry the per-layer zero padding (SURVEY.md section 7 hard part)
    if rec["T"] >= 16:
        assert np.abs(ys[:, :8] - rec["y"][:, :8]).max() <= tol
        assert np.abs(ys[:, -8:] - rec["y"][:, -8:]).max() <= tol
    if prec in ORACLE_MODE:
        ym = oracle.forward_from_state(rec["x"], rec["state"], pos_emb=rec["pos_emb"], mode=ORACLE_MODE[prec])
        errm = np.abs(y - ym).max()
        assert errm <= TOL_MODEL[prec], f"{name}/{prec}: vs operand-rounding model {errm:.3e}"
    if "y_row_sum" in rec and prec.startswith("f32"):
        np.testing.assert_allclose(y.astype(np.float64).sum(axis=(1, 2, 3)), rec["y_row_sum"], atol=2e-3)


@pytest.mark.parametrize("prec", ALL_PREC)
@pytest.mark.parametrize("T", [1, 4, 15, 16, 31, 32, 47, 48, 100, 111, 112, 113, 127, 128, 191, 192, 193, 207, 208, 209,
                               223, 224, 225, 336, 337, 384, 385, 416, 417, 1000])
def test_lengths_vs_oracle(T, prec, cuda_device):
    """Sequence lengths around the 16-frame tile and the chunk edges (112 frames in the fp32
    MFMA kernel, 208 whole / 192 split in the persistent 16-bit kernel)."""
    rec = load_golden("cfg1_b1_t200")
    g = torch.Generator().manual_seed(T)
    x = torch.rand((5, T, 12, 2), generator=g) - 0.5
    m = _model(rec, prec, cuda_device)
    with torch.no_grad():
        y = m(x.to(cuda_device)).cpu().numpy()
    ref = oracle.forward_from_state(x.numpy(), rec["state"])
    assert np.abs(y - ref).max() <= TOL[prec]
    if prec in ORACLE_MODE:
        ym = oracle.forward_from_state(x.numpy(), rec["state"], mode=ORACLE_MODE[prec])
        assert np.abs(y - ym).max() <= TOL_MODEL[prec]


@pytest.mark.parametrize("prec", ALL_PREC)
def test_kernels_agree_and_batch_independent(prec, cuda_device):
    """BASELINE config 3 shape (256 x 200): each sequence's result must not depend on
    its neighbours in the batch (bit-exact), and equal the oracle on a sample."""
    rec = load_golden("cfg1_b1_t200")
    g = torch.Generator().manual_seed(3)
    x = (torch.rand((256, 200, 12, 2), generator=g) - 0.5).to(cuda_device)
    m = _model(rec, prec, cuda_device)
    with torch.no_grad():
        y = m(x)
        idx = [0, 1, 77, 128, 255]
        y_small = m(x[idx].contiguous())
    assert torch.equal(y[idx], y_small)
    ref = oracle.forward_from_state(x[idx].cpu().numpy(), rec["state"])
    assert np.abs(y_small.cpu().numpy() - ref).max() <= TOL[prec]


@pytest.mark.parametrize("prec", ["bf16", "f16", "f16x3", "f32_mfma"])
def test_config4_stream_sharded_equals_one_launch(prec, cuda_device):
    """BASELINE config 4: 2 000 sequences x 200 frames, U[-.5,.5], seed 1234.
    The stream in ONE launch must equal, bit for bit, the eight `shard_bounds(2000, r, 8)` shards
    (250 sequences each) run separately and concatenated -- what the 8 ranks compute -- and the
    2- and 4-rank partitions; a sample of sequences equals the oracle; and the single-rank
    `ShardedStream` paths (run, run_pipelined) return the same tensor."""
    from hand_pose_sl_amd.stream import ShardedStream, shard_bounds
    rec = load_golden("cfg1_b1_t200")
    g = torch.Generator().manual_seed(1234)
    x = (torch.rand((2000, 200, 12, 2), generator=g) - 0.5).to(cuda_device)
    m = _model(rec, prec, cuda_device)
    with torch.no_grad():
        y = m(x)
        for world in (8, 4, 2):
            parts = []
            for r in range(world):
                lo, hi = shard_bounds(2000, r, world)
                assert world != 8 or hi - lo == 250
                parts.append(m(x[lo:hi].contiguous()))
            assert torch.equal(torch.cat(parts, dim=0), y), (prec, world)
        stream = ShardedStream(m, max_batch=600)        # 2000 sequences = 4 launches of <= 600
        assert torch.equal(stream.run(x, 2000, gather=True), y)
        assert torch.equal(stream.run_pipelined(x, 2000, chunk=250), y)
    idx = [0, 249, 250, 999, 1750, 1999]                  # shard edges of the 8-way partition
    ref = oracle.forward_from_state(x[idx].cpu().numpy(), rec["state"])
    assert np.abs(y[idx].cpu().numpy() - ref).max() <= TOL[prec]


@pytest.mark.parametrize("prec", ["bf16", "f16", "f16x3", "f32_mfma", "f32_valu"])
@pytest.mark.parametrize("C", [33, 40, 47, 48, 49, 56, 63, 64])
def test_wide_models_vs_oracle(C, prec, cuda_device):
    """conv_channels is a free integer in the reference (run.py:37, HandPoseModels.py:24-32).  Widths
    33..64 run on the wide matrix-core kernels (64-channel rows, four M-tiles; bf16 / f16 at two waves
    per SIMD, the fp32-grade f16 hi/lo split and exact fp32 at one wave per SIMD);
    checked against the oracle (reference bars) and the oracle's operand-rounding model on lengths
    around the tile and chunk edges, with pos_emb (T = 100), small and large batches (different
    chunkings must agree bit for bit), the fused transforms and batch independence."""
    torch.manual_seed(1000 + C)
    for pos_emb, lengths in ((False, [1, 5, 16, 17, 50, 111, 112, 113, 200, 225, 337]), (True, [100])):
        m = hps.ConvModel(C, "ReLU", pos_emb, precision=prec).to(cuda_device).eval()
        state = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}
        assert m.kernel_name() == {"bf16": "b2h_fwd_mfma16w<1, false>", "f16": "b2h_fwd_mfma16w<2, false>",
                                   "f16x3": "b2h_fwd_mfma_f16x3w<false>", "f32_mfma": "b2h_fwd_mfma_f32<false, true>",
                                   "f32_valu": "b2h_fwd_f32_valu"}[prec]
        # AUTO = the faster exact-fp32 kernel: VALU just above the 32-channel step, the matrix cores from 40
        assert m.kernel_name("fp32") == ("b2h_fwd_f32_valu" if C < 40 else "b2h_fwd_mfma_f32<false, true>")
        g = torch.Generator().manual_seed(C)
        for T in lengths:
            x = torch.rand((4, T, 12, 2), generator=g) - 0.5
            with torch.no_grad():
                y = m(x.to(cuda_device))
                big = m(torch.cat([x, torch.rand((1200, T, 12, 2), generator=g) - 0.5]).to(cuda_device))
            assert torch.equal(big[:4], y), (C, prec, T)                     # other chunking, other neighbours
            y = y.cpu().numpy()
            ref = oracle.forward_from_state(x.numpy(), state, pos_emb=pos_emb)
            assert np.abs(y - ref).max() <= TOL[prec], (C, prec, T, np.abs(y - ref).max())
            if prec in ORACLE_MODE:
                ym = oracle.forward_from_state(x.numpy(), state, pos_emb=pos_emb, mode=ORACLE_MODE[prec])
                assert np.abs(y - ym).max() <= TOL_MODEL[prec], (C, prec, T)
        if not pos_emb:   # fused pixel pipeline with a ragged tail mask
            rng = np.random.default_rng(C)
            body = (rng.random((5, 130, 12, 2), dtype=np.float32)) * np.array([1280.0, 720.0], np.float32)
            nf = np.array([130, 1, 64, 129, 77])
            with torch.no_grad():
                yf = m.forward_fused(torch.from_numpy(body).to(cuda_device), n_frames=nf, mask_tail=True).cpu().numpy()
            inp, _ = oracle.preprocess(body, None)
            ref = oracle.postprocess(oracle.forward_from_state(inp, state), 1280.0, nf)
            assert np.abs(yf - ref).max() <= 2 * TOL[prec] * 1280
            for b, n in enumerate(nf):
                assert not yf[b, n:].any()


@pytest.mark.parametrize("C", [65, 96, 104, 105, 128])
def test_widths_above_64_run_on_the_valu_kernel(C, cuda_device):
    """`--conv-channels` is a free integer (run.py:37).  65..128 channels: exact fp32 on the VALU kernel
    (three work items per thread above 104), the matrix-core kernels refuse; checked against the oracle
    on lengths around the 64-frame tile, with pos_emb, the fused transforms and batch independence."""
    torch.manual_seed(2000 + C)
    for pos_emb, lengths in ((False, [1, 17, 63, 64, 65, 130, 200]), (True, [100])):
        m = hps.ConvModel(C, "ReLU", pos_emb).to(cuda_device).eval()          # precision "fp32" = AUTO
        assert m.kernel_name() == "b2h_fwd_f32_valu"
        state = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}
        g = torch.Generator().manual_seed(C)
        for T in lengths:
            x = torch.rand((3, T, 12, 2), generator=g) - 0.5
            with torch.no_grad():
                y = m(x.to(cuda_device))
                big = m(torch.cat([x, torch.rand((200, T, 12, 2), generator=g) - 0.5]).to(cuda_device))
            assert torch.equal(big[:3], y), (C, T)
            ref = oracle.forward_from_state(x.numpy(), state, pos_emb=pos_emb)
            assert np.abs(y.cpu().numpy() - ref).max() <= TOL["f32_valu"], (C, T)
        if not pos_emb:
            rng = np.random.default_rng(C)
            body = (rng.random((4, 90, 12, 2), dtype=np.float32)) * np.array([1280.0, 720.0], np.float32)
            nf = np.array([90, 1, 64, 33])
            with torch.no_grad():
                yf = m.forward_fused(torch.from_numpy(body).to(cuda_device), n_frames=nf, mask_tail=True).cpu().numpy()
            inp, _ = oracle.preprocess(body, None)
            ref = oracle.postprocess(oracle.forward_from_state(inp, state), 1280.0, nf)
            assert np.abs(yf - ref).max() <= 2 * TOL["f32_valu"] * 1280
    for prec in ("bf16", "f16", "f16x3", "f32_mfma"):
        mm = hps.ConvModel(C, "ReLU", False, precision=prec).to(cuda_device).eval()
        with torch.no_grad(), pytest.raises(RuntimeError, match="conv_channels"):
            mm(torch.zeros((1, 8, 12, 2), device=cuda_device))
    with pytest.raises(ValueError):
        hps.ConvModel(129, "ReLU", False).to(cuda_device)(torch.zeros((1, 8, 12, 2)))


def test_module_surface(cuda_device):
    """The call surface steps/traintest.py uses: host tensor in (body_kp is never moved,
    :354-358), result on the model's device, caller mutates it in place (:387-388)."""
    rec = load_golden("cfg1_b1_t200")
    m = _model(rec, "fp32", cuda_device)
    with torch.no_grad():
        pred = m(torch.from_numpy(rec["x"]))          # CPU tensor accepted
    assert pred.device.type == "cuda"
    pred *= 1280
    assert np.abs(pred.cpu().numpy() - rec["y"] * 1280).max() < 2e-2
    with torch.no_grad():
        assert m(torch.zeros((0, 7, 12, 2))).shape == (0, 7, 21, 2)   # empty batch
        with pytest.raises(RuntimeError):
            m(torch.zeros((2, 7, 11, 2)))
    # weights replaced in place are picked up (load_state_dict after .to(device))
    rec2 = load_golden("edge_b3_t33")
    m.load_state_dict({k: torch.from_numpy(v) for k, v in rec2["state"].items()})
    with torch.no_grad():
        y2 = m(torch.from_numpy(rec2["x"]).to(cuda_device)).cpu().numpy()
    assert np.abs(y2 - rec2["y"]).max() <= TOL["f32_mfma"]


def test_pos_emb_shape_error(cuda_device):
    rec = load_golden("posemb_b2_t100")
    m = _model(rec, "fp32", cuda_device)
    with torch.no_grad(), pytest.raises(RuntimeError, match="T == 100"):
        m(torch.zeros((1, 200, 12, 2), device=cuda_device))


@pytest.mark.parametrize("prec", ALL_PREC)
def test_fused_transforms(prec, cuda_device):
    """Pre/post-processing fused into the kernel vs the reference's
    own transform classes (golden) -- raw pixels in, masked pixel predictions out."""
    rec = load_golden("transforms_b6_t40")
    m = _model(rec, prec, cuda_device)
    body = torch.from_numpy(rec["body"]).to(cuda_device)
    with torch.no_grad():
        px = m.forward_fused(body, n_frames=rec["n_frames"], mask_tail=True).cpu().numpy()
        px_nomask = m.forward_fused(body).cpu().numpy()
        plain = m(torch.from_numpy(rec["input_kp"]).to(cuda_device)).cpu().numpy()
    tol_px = TOL[prec] * 1280
    assert np.abs(px - rec["pred_px_masked"]).max() <= tol_px
    assert np.abs(px_nomask - rec["pred_px"]).max() <= tol_px
    assert np.abs(plain - rec["pred"]).max() <= TOL[prec]
    for b, n in enumerate(rec["n_frames"]):
        assert not px[b, n:].any()
    tgt = hps.target_transform(body, torch.from_numpy(rec["right_hand"]).to(cuda_device)).cpu().numpy()
    assert np.array_equal(tgt, rec["target_kp"])


def test_large_stream_properties(cuda_device):
    """Full bench size (262 144 x 200 frames): too big for the oracle, so check
    size-independent properties: every 4096-sequence block of a stream made of one
    repeated 4096-block is bit-identical, and a sample equals the oracle."""
    rec = load_golden("cfg1_b1_t200")
    g = torch.Generator().manual_seed(11)
    blk = (torch.rand((4096, 200, 12, 2), generator=g) - 0.5).to(cuda_device)
    for prec, reps in (("bf16", 64), ("f32_mfma", 16), ("f16x3", 32)):
        x = blk.repeat(reps, 1, 1, 1)
        n = reps * 4096
        m = _model(rec, prec, cuda_device)
        with torch.no_grad():
            y = m(x)
        yb = y.view(reps, 4096, 200, 21, 2)
        for i in range(1, reps):
            assert torch.equal(yb[0], yb[i])
        idx = [0, 4095, n - 4096, n - 1]
        ref = oracle.forward_from_state(x[idx].cpu().numpy(), rec["state"])
        assert np.abs(y[idx].cpu().numpy() - ref).max() <= TOL[prec]
        assert torch.isfinite(y).all()
        del y, x


@pytest.mark.parametrize("prec", ["bf16", "f32_mfma", "f16x3"])
def test_hip_graph_capture_and_replay(prec, cuda_device):
    """The launch path does no allocation or synchronisation, so a caller can capture it
    into a HIP graph (torch.cuda.CUDAGraph on ROCm) and replay it on fresh data."""
    rec = load_golden("cfg1_b1_t200")
    m = _model(rec, prec, cuda_device)
    g = torch.Generator().manual_seed(21)
    x_static = (torch.rand((32, 200, 12, 2), generator=g) - 0.5).to(cuda_device)
    with torch.no_grad():
        m._ensure_handle()                            # weights packed (and LDS caps raised) at load time:
        torch.cuda.synchronize()                      # the FIRST forward is already capture-safe
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            y_static = m(x_static)
        for seed in (1, 2):
            x_new = (torch.rand((32, 200, 12, 2), generator=torch.Generator().manual_seed(seed)) - 0.5).to(cuda_device)
            x_static.copy_(x_new)
            graph.replay()
            torch.cuda.synchronize()
            assert torch.equal(y_static, m(x_new))


def test_c_abi_argument_checks(cuda_device):
    """Misaligned or overlapping buffers are refused, not silently mis-computed."""
    import ctypes
    from hand_pose_sl_amd import _lib
    rec = load_golden("cfg1_b1_t200")
    m = _model(rec, "bf16", cuda_device)
    lib = m._ensure_handle()
    buf = torch.zeros(4 + 200 * 24 + 200 * 42 + 64, dtype=torch.float32, device=cuda_device)
    x, y = buf[:200 * 24], buf[200 * 24 + 8:200 * 24 + 8 + 200 * 42]
    vp = ctypes.c_void_p
    assert lib.b2h_forward(m._handle, vp(x.data_ptr()), vp(y.data_ptr()), 1, 200, 3, None) == _lib.OK
    assert lib.b2h_forward(m._handle, vp(x.data_ptr() + 4), vp(y.data_ptr()), 1, 200, 3, None) == _lib.ERR_INVALID
    assert b"aligned" in lib.b2h_last_error()
    assert lib.b2h_forward(m._handle, vp(x.data_ptr()), vp(x.data_ptr() + 1024), 1, 200, 3, None) == _lib.ERR_INVALID
    assert b"overlap" in lib.b2h_last_error()
    assert lib.b2h_forward(m._handle, vp(x.data_ptr()), vp(y.data_ptr()), 1, 0, 3, None) == _lib.ERR_SHAPE
    assert lib.b2h_forward(m._handle, vp(x.data_ptr()), vp(y.data_ptr()), 1, 200, 9, None) == _lib.ERR_UNSUPPORTED
    # the model-free entry points check their pointers too: NULL and host memory are refused
    d = torch.zeros(2 * 5 * 42, dtype=torch.float32, device=cuda_device)
    b = torch.zeros(2 * 5 * 24, dtype=torch.float32, device=cuda_device)
    host = torch.zeros(2 * 5 * 42, dtype=torch.float32)
    out2 = torch.zeros(3, dtype=torch.float32, device=cuda_device)
    assert lib.b2h_masked_l1(vp(d.data_ptr()), vp(d.data_ptr()), None, 2, 5, vp(out2.data_ptr()), vp(out2.data_ptr() + 8), None) == _lib.OK
    assert lib.b2h_masked_l1(vp(host.data_ptr()), vp(d.data_ptr()), None, 2, 5, vp(out2.data_ptr()), vp(out2.data_ptr() + 8), None) == _lib.ERR_INVALID
    assert b"pred" in lib.b2h_last_error()
    assert lib.b2h_masked_l1(vp(d.data_ptr()), None, None, 2, 5, vp(out2.data_ptr()), vp(out2.data_ptr() + 8), None) == _lib.ERR_INVALID
    assert b"NULL" in lib.b2h_last_error()
    sc = torch.ones(2 * 5 * 21, dtype=torch.float32, device=cuda_device)
    assert lib.b2h_weighted_l1(vp(d.data_ptr()), vp(d.data_ptr()), vp(sc.data_ptr()), None, 2, 5, vp(out2.data_ptr()), vp(out2.data_ptr() + 8), None) == _lib.OK
    assert lib.b2h_weighted_l1(vp(d.data_ptr()), vp(d.data_ptr()), None, None, 2, 5, vp(out2.data_ptr()), vp(out2.data_ptr() + 8), None) == _lib.ERR_INVALID
    assert b"scores" in lib.b2h_last_error()
    assert lib.b2h_weighted_l1(vp(d.data_ptr()), vp(d.data_ptr()), vp(host.data_ptr()), None, 2, 5, vp(out2.data_ptr()), vp(out2.data_ptr() + 8), None) == _lib.ERR_INVALID
    assert lib.b2h_weighted_l1(vp(d.data_ptr()), vp(d.data_ptr()), vp(sc.data_ptr()), None, 0, 5, vp(out2.data_ptr()), vp(out2.data_ptr() + 8), None) == _lib.ERR_SHAPE
    assert lib.b2h_target_transform(vp(b.data_ptr()), vp(d.data_ptr()), vp(d.data_ptr()), 2, 5, 3, 1280.0, None) == _lib.OK
    assert lib.b2h_target_transform(vp(b.data_ptr()), vp(host.data_ptr()), vp(d.data_ptr()), 2, 5, 3, 1280.0, None) == _lib.ERR_INVALID
    assert lib.b2h_target_transform(vp(b.data_ptr()), vp(d.data_ptr()), vp(d.data_ptr()), 2, 5, 8, 1280.0, None) == _lib.ERR_INVALID
    torch.cuda.synchronize()


def test_two_streams_concurrently(cuda_device):
    """Independent launches on two streams (each with its own output) give the same
    results as serial launches: the library keeps no per-call device state."""
    rec = load_golden("cfg1_b1_t200")
    m = _model(rec, "bf16", cuda_device)
    g = torch.Generator().manual_seed(5)
    xa = (torch.rand((512, 200, 12, 2), generator=g) - 0.5).to(cuda_device)
    xb = (torch.rand((512, 200, 12, 2), generator=g) - 0.5).to(cuda_device)
    with torch.no_grad():
        ra, rb = m(xa), m(xb)
        torch.cuda.synchronize()
        s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
        outs = []
        for _ in range(5):
            with torch.cuda.stream(s1):
                ya = m(xa)
            with torch.cuda.stream(s2):
                yb = m(xb)
            outs.append((ya, yb))
        torch.cuda.synchronize()
    for ya, yb in outs:
        assert torch.equal(ya, ra) and torch.equal(yb, rb)


@pytest.mark.parametrize("prec", ["bf16", "f32_mfma", "f32_valu"])
def test_long_sequence(prec, cuda_device):
    """One 5 000-frame sequence = 27 chunks (16-bit) / 45 chunks (fp32 MFMA) with halos."""
    rec = load_golden("cfg1_b1_t200")
    g = torch.Generator().manual_seed(77)
    x = torch.rand((2, 5000, 12, 2), generator=g) - 0.5
    m = _model(rec, prec, cuda_device)
    with torch.no_grad():
        y = m(x.to(cuda_device)).cpu().numpy()
    ref = oracle.forward_from_state(x.numpy(), rec["state"])
    assert np.abs(y - ref).max() <= TOL[prec]


@pytest.mark.parametrize("prec", ALL_PREC)
@pytest.mark.parametrize("shift", [1, 7, 16, 37])
def test_time_shift_equivariance(prec, shift, cuda_device):
    """Away from the sequence ends (+-8 frames) the model is a pure convolution: shifting the
    input by k frames shifts the output by k frames, bit for bit -- whatever tile or chunk a
    frame lands in (T = 500 spans three 192-frame chunks / five 112-frame chunks)."""
    rec = load_golden("cfg1_b1_t200")
    T = 500
    g = torch.Generator().manual_seed(shift)
    long = (torch.rand((2, T + shift, 12, 2), generator=g) - 0.5).to(cuda_device)
    m = _model(rec, prec, cuda_device)
    with torch.no_grad():
        ya = m(long[:, :T].contiguous())            # frames 0 .. T-1
        yb = m(long[:, shift:].contiguous())        # frames shift .. T+shift-1
    # global frame f: ya index f, yb index f - shift; both interior for f in [shift+8, T-8)
    assert torch.equal(ya[:, shift + 8:T - 8], yb[:, 8:T - 8 - shift])


@pytest.mark.parametrize("n,chunk", [(1000, 256), (777, 100), (64, 256), (513, 512)])
def test_host_pipeline_equals_direct(n, chunk, cuda_device):
    """H2D / kernel / D2H on three streams over double buffers: same bits as the direct call,
    for piece counts that do and do not divide the stream, pinned and pageable input."""
    from hand_pose_sl_amd.stream import HostPipeline
    rec = load_golden("cfg1_b1_t200")
    m = _model(rec, "bf16", cuda_device)
    g = torch.Generator().manual_seed(n)
    x = torch.rand((n, 200, 12, 2), generator=g) - 0.5
    with torch.no_grad():
        ref = m(x.to(cuda_device)).cpu()
    pipe = HostPipeline(m, chunk=chunk)
    y = pipe.run(x.pin_memory())
    assert y.is_pinned() and torch.equal(y, ref)
    assert torch.equal(pipe.run(x), ref)                    # pageable input
    out = torch.empty_like(ref)
    assert pipe.run(x, out=out) is out and torch.equal(out, ref)
    y50 = pipe.run(x[:, :50].contiguous())                  # buffers follow a new T
    with torch.no_grad():
        assert torch.equal(y50, m(x[:, :50].contiguous().to(cuda_device)).cpu())


def test_f16x3_refuses_weights_outside_f16_range(cuda_device):
    """The split kernel represents every operand as f16 hi + lo: a weight of magnitude >= 65504
    (or a non-finite one) cannot be represented, so the kernel must refuse the model loudly while
    the exact fp32 kernels keep working."""
    rec = load_golden("cfg1_b1_t200")
    x = torch.from_numpy(rec["x"]).to(cuda_device)
    for bad in (1.0e5, float("inf")):
        state = {k: torch.from_numpy(v.copy()) for k, v in rec["state"].items()}
        state["conv2.weight"][3, 2, 1] = bad
        m = hps.ConvModel(rec["C"], "ReLU", rec["pos_emb"], precision="f16x3")
        m.load_state_dict(state)
        m = m.to(cuda_device).eval()
        with torch.no_grad(), pytest.raises(RuntimeError, match="f16 range"):
            m(x)
        with torch.no_grad():
            assert m.forward_into(x, torch.empty((rec["B"], rec["T"], 21, 2), device=cuda_device),
                                  precision="f32_mfma") is not None
    m = _model(rec, "f16x3", cuda_device)              # in range: runs
    with torch.no_grad():
        assert np.abs(m(x).cpu().numpy() - rec["y"]).max() <= TOL["f16x3"]


@pytest.mark.parametrize("prec", ["f32_mfma", "f16x3"])
def test_adaptive_chunk_lengths_are_bit_identical(prec, cuda_device):
    """The wave-per-chunk kernels cut sequences into 112-frame chunks, or 64 / 32 when the batch
    would leave most wave slots idle (b2h_api.hip launch()): the same sequences inside batches of
    3, 400 and 1500 (-> 32, 64, 112 frames per chunk at T = 300) must come out bit-identical, and
    equal the oracle."""
    rec = load_golden("cfg1_b1_t200")
    T = 300
    g = torch.Generator().manual_seed(17)
    x = (torch.rand((1500, T, 12, 2), generator=g) - 0.5).to(cuda_device)
    m = _model(rec, prec, cuda_device)
    with torch.no_grad():
        y_big = m(x)[:3].clone()
        y_mid = m(x[:400].contiguous())[:3].clone()
        y_small = m(x[:3].contiguous())
    assert torch.equal(y_big, y_mid) and torch.equal(y_big, y_small)
    ref = oracle.forward_from_state(x[:3].cpu().numpy(), rec["state"])
    assert np.abs(y_small.cpu().numpy() - ref).max() <= TOL[prec]


@pytest.mark.parametrize("prec", ["bf16", "f16"])
def test_dynamic_launch_with_chunked_sequences(prec, cuda_device):
    """A DYNAMIC launch (>= 256 chunks per workgroup) of sequences longer than one chunk (T = 400: three chunks
    of <= 192 frames with halos): a claimed run of two consecutive chunks then straddles sequences.  Must equal
    the same sequences run in small static launches, bit for bit, and the oracle on a sample."""
    rec = load_golden("cfg2_b64_t200_u55")
    m = _model(rec, prec, cuda_device)
    ncu = torch.cuda.get_device_properties(cuda_device).multi_processor_count
    T = 400
    S = (256 * ncu + 2) // 3 + 11                      # 3 chunks per sequence -> a little over 256 per workgroup
    g = torch.Generator().manual_seed(5)
    x = (torch.rand((S, T, 12, 2), generator=g) - 0.5).to(cuda_device)
    with torch.no_grad():
        y = m(x)
        y_small = torch.cat([m(x[a:a + 1000]) for a in range(0, S, 1000)])
        assert torch.equal(y, y_small)
        assert torch.equal(m(x), y)                                      # the counter is back at zero
        idx = [0, 999, 1000, S - 1]
        ref = oracle.forward_from_state(x[idx].cpu().numpy(), rec["state"])
        assert np.abs(y[idx].cpu().numpy() - ref).max() <= TOL[prec]


@pytest.mark.parametrize("prec", ALL_PREC)
def test_fused_every_length(prec, cuda_device):
    """The fused instantiation against the plain kernel of the same precision at EVERY length 1..80
    and around the chunk sizes, small and large batch (different chunkings), with numerically
    neutral flags (x 1.0, a tail mask that masks nothing): the results must be bit-identical.
    Regression test for a store-data write-after-read hazard in the fused 16-bit kernel that only
    showed when the head layer had an odd number of tiles and the last tile was full
    (T mod 32 in 13..16; tools/stress_conv.py found it)."""
    rec = load_golden("cfg1_b1_t200")
    m = _model(rec, prec, cuda_device)
    g = torch.Generator().manual_seed(23)
    lengths = list(range(1, 81)) + [95, 96, 111, 112, 113, 143, 144, 176, 191, 192, 193, 207, 208, 209, 240, 400, 432, 500]
    for T in lengths:
        for B in ((2, 600) if T <= 208 else (2, 40)):
            x = (torch.rand((B, T, 12, 2), generator=g) - 0.5).to(cuda_device)
            with torch.no_grad():
                plain = m(x)
                a = m.forward_fused(x, dif_encoding=False, normalize=False, denormalize=True, factor=1.0)
                b = m.forward_fused(x, dif_encoding=False, normalize=False, denormalize=False, mask_tail=True,
                                    n_frames=[T] * B)
            assert torch.equal(a, plain), (prec, B, T, "x1.0")
            assert torch.equal(b, plain), (prec, B, T, "mask nothing")


@pytest.mark.parametrize("prec", ALL_PREC)
def test_very_long_sequence_plain_and_fused(prec, cuda_device):
    """T = 70 001 frames (hundreds of chunks per sequence, byte offsets past 2^23): plain forward and
    the fused pixel pipeline with a ragged tail mask, against the oracle."""
    rec = load_golden("cfg1_b1_t200")
    rng = np.random.default_rng(9)
    B, T = 2, 70001
    x = rng.random((B, T, 12, 2), dtype=np.float32) - 0.5
    body = (x + 0.5) * np.array([1280.0, 720.0], np.float32)
    nf = np.array([T, 12345])
    m = _model(rec, prec, cuda_device)
    with torch.no_grad():
        y = m(torch.from_numpy(x).to(cuda_device)).cpu().numpy()
        yf = m.forward_fused(torch.from_numpy(body).to(cuda_device), n_frames=nf, mask_tail=True).cpu().numpy()
    assert np.abs(y - oracle.forward_from_state(x, rec["state"])).max() <= TOL[prec]
    inp, _ = oracle.preprocess(body, None)
    ref = oracle.postprocess(oracle.forward_from_state(inp, rec["state"]), 1280.0, nf)
    assert np.abs(yf - ref).max() <= 2 * TOL[prec] * 1280
    assert not yf[1, 12345:].any()


SCALE_SIGMAS = [0.25, 0.5, 1.0, 2.0, 4.0]


def test_error_vs_input_scale(cuda_device):
    """north_star's gate is <= 1e-3 max-abs vs the reference's fp32 forward.  bf16 meets it on inputs scaled
    like normalised keypoints and exceeds it from sigma ~ 1 on (the documented BF16_RANDN_BOUND); this test
    makes that a CURVE with a stated crossover instead of one fixture: N(0, sigma^2) inputs, sigma in
    {0.25 ... 4}, every 16-bit precision and the fp32-grade ones against the fp32 oracle, seeded default
    weights (the reference ships no checkpoint).  The curve is written to gpurun_out/ for DESIGN.md."""
    import json
    import os
    rec = load_golden("cfg2_b64_t200_randn")
    g = np.random.default_rng(7)
    base = g.standard_normal((64, 200, 12, 2)).astype(np.float32)
    curve = {p: [] for p in ("bf16", "f16", "f16x3", "f32_mfma")}
    for sigma in SCALE_SIGMAS:
        xs = base * np.float32(sigma)
        ref = oracle.forward_from_state(xs, rec["state"])
        xd = torch.from_numpy(xs).to(cuda_device)
        for prec in curve:
            with torch.no_grad():
                y = _model(rec, prec, cuda_device)(xd).cpu().numpy()
            curve[prec].append(float(np.abs(y - ref).max()))
    print("max-abs error vs input sigma", SCALE_SIGMAS, json.dumps(curve))
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "error_vs_input_scale.json"), "w") as f:
            json.dump({"sigmas": SCALE_SIGMAS, "max_abs_err": curve, "inputs": "N(0, sigma^2), (64,200,12,2), seed 7",
                       "weights": "golden cfg2 (seeded default init, C = 30)"}, f)
    for i, sigma in enumerate(SCALE_SIGMAS):
        # the 16-bit kernels' error is operand rounding, linear in the input scale once |x| dominates the bias
        assert curve["f16"][i] <= 1e-3, (sigma, curve["f16"][i])                 # f16 holds the gate over the whole sweep
        assert curve["f16x3"][i] <= 2e-5 * max(1.0, sigma) and curve["f32_mfma"][i] <= 2e-5 * max(1.0, sigma)
        if sigma <= 0.5:
            assert curve["bf16"][i] <= 1e-3, (sigma, curve["bf16"][i])           # normalised-keypoint scale: inside the gate
        else:
            # operand rounding scales with the input: measured 1.08e-3 / 2.32e-3 / 4.31e-3 at sigma 1 / 2 / 4
            assert curve["bf16"][i] <= 1.2e-3 * sigma, (sigma, curve["bf16"][i])
    assert curve["bf16"][SCALE_SIGMAS.index(1.0)] > curve["f16"][SCALE_SIGMAS.index(4.0)]   # f16 at 4 sigma beats bf16 at 1


@pytest.mark.parametrize("prec", ["bf16", "f16"])
def test_claimed_chunks_pool_launches(prec, cuda_device):
    """The persistent 16-bit kernel's work distribution (kernel_mfma16.h, Sched16): from 256 chunks per workgroup
    on a launch is DYNAMIC -- every wave claims runs of two consecutive chunks from a device-wide counter in a
    per-stream slot that the kernel itself leaves at zero.  Which wave computes a chunk must never show in the
    result: a dynamic launch == the same sequences run in small launches (static: LDS queue only) == a launch
    captured in a graph (static: a graph may replay on any stream), bit for bit; back-to-back dynamic launches on
    one stream and concurrent ones on two streams (two slots) repeat it exactly."""
    rec = load_golden("cfg2_b64_t200_u55")
    m = _model(rec, prec, cuda_device)
    ncu = torch.cuda.get_device_properties(cuda_device).multi_processor_count
    S = 256 * ncu + 37                                 # >= 256 chunks per workgroup, not a multiple of anything
    g = torch.Generator().manual_seed(33)
    x = (torch.rand((S, 200, 12, 2), generator=g) - 0.5).to(cuda_device)
    with torch.no_grad():
        y_pool = m(x)
        y_small = torch.cat([m(x[a:a + 4000]) for a in range(0, S, 4000)])          # 15 chunks per workgroup: static
        assert torch.equal(y_pool, y_small)
        idx = [0, 1, 4000, S - 1]
        ref = oracle.forward_from_state(x[idx].cpu().numpy(), rec["state"])
        assert np.abs(y_pool[idx].cpu().numpy() - ref).max() <= TOL[prec]
        for _ in range(4):                                                           # the pool words are back at zero
            assert torch.equal(m(x), y_pool)
        x2 = x.flip(0).contiguous()
        y2 = m(x2)
        assert torch.equal(y2, y_pool.flip(0))
        torch.cuda.synchronize()
        s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
        outs = []
        for _ in range(3):
            with torch.cuda.stream(s1):
                ya = m(x)
            with torch.cuda.stream(s2):
                yb = m(x2)
            outs.append((ya, yb))
        torch.cuda.synchronize()
        for ya, yb in outs:
            assert torch.equal(ya, y_pool) and torch.equal(yb, y2)
        xs = x.clone()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            ys = m(xs)
        for src, want in ((x2, y2), (x, y_pool)):
            xs.copy_(src)
            graph.replay()
            torch.cuda.synchronize()
            assert torch.equal(ys, want)
        assert torch.equal(m(x), y_pool)                                             # and a pool launch after the replays
