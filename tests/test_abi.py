"""The C-ABI boundary and the Python mirror, without a GPU: the library builds,
loads and exports exactly what include/b2h.h declares; host-side argument and
error behaviour matches the reference's."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

import hand_pose_sl_amd as hps
from hand_pose_sl_amd import _lib
from conftest import ROOT, load_golden

HEADER = os.path.join(ROOT, "include", "b2h.h")


def _declared():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(b2h_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    names = _declared()
    assert len(names) >= 14
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/b2h.h but not exported"
    assert sorted(_lib.SYMBOLS) == names, "ctypes table and header drifted"
    assert lib.b2h_version() == 100


def test_header_cites_reference_interfaces():
    src = open(HEADER).read()
    for cite in ("HandPoseModels.py:18-37", "HandPoseModels.py:40-64", "infer_utterance.py:109",
                 "steps/utils.py:309-312"):
        assert cite in src


def test_create_error_paths_without_gpu():
    lib = _lib.load()
    h = ctypes.c_void_p()
    assert lib.b2h_create(30, b"Tanh", 0, ctypes.byref(h)) == _lib.ERR_INVALID
    assert b"ReLU" in lib.b2h_last_error()
    assert lib.b2h_create(0, b"ReLU", 0, ctypes.byref(h)) == _lib.ERR_INVALID
    assert lib.b2h_create(129, b"ReLU", 0, ctypes.byref(h)) == _lib.ERR_INVALID
    if not torch.cuda.is_available():
        assert lib.b2h_device_count() == 0
        assert lib.b2h_create(30, b"ReLU", 0, ctypes.byref(h)) == _lib.ERR_NO_DEVICE
        assert h.value is None
    assert lib.b2h_forward(None, None, None, 1, 1, 0, None) == _lib.ERR_INVALID


def test_constructor_mirrors_reference():
    with pytest.raises(ValueError):
        hps.ConvModel(30, "Tanh", False)          # HandPoseModels.py:34-37
    m = hps.ConvModel(30, "ReLU", False)
    sd = m.state_dict()
    assert list(sd) == [f"conv{i}.{p}" for i in (1, 2, 3, 4) for p in ("weight", "bias")]
    assert sd["conv1.weight"].shape == (30, 24, 5) and sd["conv4.weight"].shape == (42, 30, 5)
    assert sum(v.numel() for v in sd.values()) == 19032
    mp = hps.ConvModel(30, "ReLU", True)
    assert mp.state_dict()["conv1.weight"].shape == (30, 25, 5)
    assert sum(v.numel() for v in mp.state_dict().values()) == 19182
    assert "pos_emb" not in "".join(mp.state_dict())   # not a registered buffer in the reference either


@pytest.mark.parametrize("name", ["cfg1_b1_t200", "posemb_b2_t100", "width_c64_b2_t50"])
def test_seeded_init_equals_reference(name):
    """Same constructor order => same default init under the same seed, so a user
    switching frameworks gets the reference's weights bit for bit."""
    rec = load_golden(name)
    torch.manual_seed(rec["seed"])
    m = hps.ConvModel(rec["C"], "ReLU", rec["pos_emb"])
    for k, v in m.state_dict().items():
        assert np.array_equal(v.numpy(), rec["state"][k]), k


def test_state_dict_round_trip():
    rec = load_golden("cfg1_b1_t200")
    m = hps.ConvModel(30, "ReLU", False)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in rec["state"].items()})
    for k, v in m.state_dict().items():
        assert np.array_equal(v.numpy(), rec["state"][k])
    m.eval(); m.train(); list(m.parameters())    # module protocol used by traintest.py


def test_no_cpu_fallback():
    m = hps.ConvModel(30, "ReLU", False).eval()
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with torch.no_grad(), pytest.raises(RuntimeError, match="MI355X"):
        m(torch.zeros(1, 4, 12, 2))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "hand_pose_sl_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt and "liboracle" not in txt, f


def test_concurrent_builds_do_not_race(tmp_path):
    """Eight ranks importing the package at once (torchrun) must not corrupt libb2h.so:
    the build is serialised by a file lock and the library is renamed into place."""
    import subprocess, sys
    code = ("import sys; sys.path.insert(0, %r); from hand_pose_sl_amd import build, _lib; "
            "build.build(); print(_lib.load().b2h_version())" % ROOT)
    procs = [subprocess.Popen([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
             for _ in range(4)]
    for p in procs:
        out, err = p.communicate(timeout=300)
        assert p.returncode == 0, err.decode()[-500:]
        assert out.decode().strip().endswith("100")
