"""ctypes front end of oracle/convmodel_oracle.c (ORACLE: test infrastructure)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None

_fp = ctypes.POINTER(ctypes.c_float)


def build_oracle(force=False):
    """gcc-compile the C restatement into oracle/_build/liboracle.so."""
    src = os.path.join(_HERE, "convmodel_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        os.makedirs(os.path.dirname(_SO), exist_ok=True)
        subprocess.check_call(["gcc", "-O2", "-fPIC", "-Wall", "-Wextra", "-std=c11", "-shared",
                               "-o", _SO, src, "-lm"])
    return _SO


def _load():
    global _lib
    if _lib is None:
        build_oracle()
        lib = ctypes.CDLL(_SO)
        lib.b2h_oracle_forward.restype = ctypes.c_int
        lib.b2h_oracle_forward.argtypes = [_fp, _fp] + [ctypes.c_int] * 4 + [_fp] * 8 + [ctypes.c_int] * 2
        lib.b2h_oracle_preprocess.restype = ctypes.c_int
        lib.b2h_oracle_preprocess.argtypes = [_fp, _fp, _fp, _fp, ctypes.c_int, ctypes.c_int,
                                              ctypes.c_int, ctypes.c_float]
        lib.b2h_oracle_postprocess.restype = ctypes.c_int
        lib.b2h_oracle_postprocess.argtypes = [_fp, ctypes.c_int, ctypes.c_int, ctypes.c_float,
                                               ctypes.POINTER(ctypes.c_int64)]
        lib.b2h_oracle_masked_l1.restype = ctypes.c_int
        lib.b2h_oracle_masked_l1.argtypes = [_fp, _fp, ctypes.POINTER(ctypes.c_int64), ctypes.c_int, ctypes.c_int, _fp, _fp]
        lib.b2h_oracle_weighted_l1.restype = ctypes.c_int
        lib.b2h_oracle_weighted_l1.argtypes = [_fp, _fp, _fp, ctypes.POINTER(ctypes.c_int64), ctypes.c_int, ctypes.c_int, _fp, _fp]
        _lib = lib
    return _lib


def _f32(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32))


def _p(a):
    return a.ctypes.data_as(_fp)


MODES = {"fp32": 0, "f32": 0, "bf16": 1, "f16": 2, "fp16": 2}


def forward(x, w1, b1, w2, b2, w3, b3, w4, b4, pos_emb=False, mode="fp32", acc64=False):
    """ConvModel.forward restated (HandPoseModels.py:40-64).

    x (B,T,12,2) float32 -> (B,T,21,2) float32.  Weights in the reference's
    state_dict layout (C_out, C_in, 5)."""
    lib = _load()
    x = _f32(x)
    if x.ndim != 4 or x.shape[2:] != (12, 2):
        raise ValueError(f"expected (B,T,12,2), got {x.shape}")
    B, T = x.shape[:2]
    ws = [_f32(a) for a in (w1, b1, w2, b2, w3, b3, w4, b4)]
    C = ws[0].shape[0]
    cin1 = 25 if pos_emb else 24
    if ws[0].shape != (C, cin1, 5) or ws[2].shape != (C, C, 5) or ws[4].shape != (C, C, 5) \
            or ws[6].shape != (42, C, 5):
        raise ValueError("weight shapes do not match ConvModel(C, 'ReLU', pos_emb)")
    y = np.empty((B, T, 21, 2), dtype=np.float32)
    rc = lib.b2h_oracle_forward(_p(x), _p(y), B, T, C, int(bool(pos_emb)), *[_p(a) for a in ws],
                                MODES[mode], int(bool(acc64)))
    if rc == -2:
        raise RuntimeError("pos_emb requires T == 100 (HandPoseModels.py:23,80-82)")
    if rc != 0:
        raise RuntimeError(f"b2h_oracle_forward failed: {rc}")
    return y


def forward_from_state(x, state, pos_emb=False, mode="fp32", acc64=False):
    """`state`: mapping with keys conv{1..4}.{weight,bias} (or '_' separated)."""
    def g(k):
        for key in (k, k.replace(".", "_")):
            if key in state:
                v = state[key]
                return v.detach().cpu().numpy() if hasattr(v, "detach") else np.asarray(v)
        raise KeyError(k)
    return forward(x, g("conv1.weight"), g("conv1.bias"), g("conv2.weight"), g("conv2.bias"),
                   g("conv3.weight"), g("conv3.bias"), g("conv4.weight"), g("conv4.bias"),
                   pos_emb=pos_emb, mode=mode, acc64=acc64)


def preprocess(body, hand=None, dif_encoding=True, normalize=True, factor=1280.0):
    """WristDifference, ChestDifference, NormalizeFixedFactor, BuildRightHandItem
    (steps/utils.py:180-210,261-277 in run.py:85-90,102 order).
    Returns (input_kp, target_kp or None)."""
    lib = _load()
    body = _f32(body)
    B, T = body.shape[:2]
    inp = np.empty_like(body)
    tgt = None
    hp = None
    tp = None
    if hand is not None:
        hand = _f32(hand)
        tgt = np.empty_like(hand)
        hp, tp = _p(hand), _p(tgt)
    flags = (3 if dif_encoding else 0) | (4 if normalize else 0)
    rc = lib.b2h_oracle_preprocess(_p(body), hp, _p(inp), tp, B, T, flags, factor)
    if rc != 0:
        raise RuntimeError(f"b2h_oracle_preprocess failed: {rc}")
    return inp, tgt


def postprocess(pred, factor=1280.0, n_frames=None):
    """pred * factor, then rows t >= n_frames[b] zeroed (traintest.py:387-388,
    steps/utils.py:309-312).  Returns a new array."""
    lib = _load()
    out = _f32(pred).copy()
    B, T = out.shape[:2]
    nf = None
    if n_frames is not None:
        nfa = np.ascontiguousarray(np.asarray(n_frames, dtype=np.int64))
        nf = nfa.ctypes.data_as(ctypes.POINTER(ctypes.c_int64))
    rc = lib.b2h_oracle_postprocess(_p(out), B, T, factor, nf)
    if rc != 0:
        raise RuntimeError(f"b2h_oracle_postprocess failed: {rc}")
    return out


def weighted_l1(pred, target, scores, lengths=None):
    """poderatedPoseL1 (steps/utils.py:431-452): SUM over the batch of the per-utterance means of
    |pred * s - target * s|.  Returns (loss float32 scalar, per_seq (B,))."""
    lib = _load()
    pred, target, scores = _f32(pred), _f32(target), _f32(scores)
    B, T = pred.shape[:2]
    if scores.shape != (B, T, 21):
        raise ValueError(f"scores must be (B, T, 21), got {scores.shape}")
    nf = None
    if lengths is not None:
        nfa = np.ascontiguousarray(np.asarray(lengths, dtype=np.int64))
        nf = nfa.ctypes.data_as(ctypes.POINTER(ctypes.c_int64))
    per = np.empty((B,), dtype=np.float32)
    loss = np.empty((1,), dtype=np.float32)
    rc = lib.b2h_oracle_weighted_l1(_p(pred), _p(target), _p(scores), nf, B, T, _p(per), _p(loss))
    if rc != 0:
        raise RuntimeError(f"b2h_oracle_weighted_l1 failed: {rc}")
    return loss[0], per


def masked_l1(pred, target, lengths=None):
    """maskedPoseL1 (steps/utils.py:413-428).  Returns (loss float32 scalar, per_seq (B,))."""
    lib = _load()
    pred, target = _f32(pred), _f32(target)
    B, T = pred.shape[:2]
    nf = None
    if lengths is not None:
        nfa = np.ascontiguousarray(np.asarray(lengths, dtype=np.int64))
        nf = nfa.ctypes.data_as(ctypes.POINTER(ctypes.c_int64))
    per = np.empty((B,), dtype=np.float32)
    loss = np.empty((1,), dtype=np.float32)
    rc = lib.b2h_oracle_masked_l1(_p(pred), _p(target), nf, B, T, _p(per), _p(loss))
    if rc != 0:
        raise RuntimeError(f"b2h_oracle_masked_l1 failed: {rc}")
    return loss[0], per

