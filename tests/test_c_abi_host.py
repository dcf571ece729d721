"""The C ABI driven by a compiled C++ host program (examples/c_abi_host.cpp): no Python or
torch inside the process that calls libb2h.  Compiles everywhere (hipcc cross-compiles
without a GPU); the run itself needs the MI355X."""
import os
import shutil
import subprocess

import numpy as np
import pytest

import oracle
from hand_pose_sl_amd import build as b2h_build
from conftest import ROOT, load_golden

SRC = os.path.join(ROOT, "examples", "c_abi_host.cpp")
EXE = os.path.join(ROOT, "examples", "c_abi_host")


def _compile():
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    lib = b2h_build.build()
    libdir = os.path.dirname(lib)
    if not os.path.exists(EXE) or os.path.getmtime(EXE) < max(os.path.getmtime(SRC), os.path.getmtime(lib)):
        subprocess.check_call([hipcc, "-O2", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "include"),
                               "-o", EXE, SRC, "-L" + libdir, "-lb2h", "-Wl,-rpath," + libdir])
    return EXE


def test_host_program_compiles_and_links():
    exe = _compile()
    assert os.access(exe, os.X_OK)
    needed = subprocess.check_output(["readelf", "-d", exe]).decode()
    assert "libb2h.so" in needed and "torch" not in needed and "python" not in needed.lower()


@pytest.mark.gpu
@pytest.mark.parametrize("kernel", [1, 2, 3, 4])      # enum b2h_kernel: VALU, f32 MFMA, bf16, f16
@pytest.mark.parametrize("name", ["edge_b3_t33", "posemb_b2_t100", "cfg1_b1_t200"])
def test_host_program_matches_reference(name, kernel, tmp_path, cuda_device):
    exe = _compile()
    rec = load_golden(name)
    inp, out = tmp_path / "in.bin", tmp_path / "out.bin"
    with open(inp, "wb") as f:
        np.array([rec["C"], int(rec["pos_emb"]), rec["B"], rec["T"]], dtype=np.int32).tofile(f)
        for k in ("conv1.weight", "conv1.bias", "conv2.weight", "conv2.bias", "conv3.weight", "conv3.bias",
                  "conv4.weight", "conv4.bias"):
            np.ascontiguousarray(rec["state"][k], dtype=np.float32).tofile(f)
        np.ascontiguousarray(rec["x"], dtype=np.float32).tofile(f)
    r = subprocess.run([exe, str(inp), str(out), str(kernel)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    y = np.fromfile(out, dtype=np.float32).reshape(rec["B"], rec["T"], 21, 2)
    # bf16: north_star's 1e-3 on normalised inputs; cfg1_b1_t200 is the N(0,1) exceedance case
    # (test_gpu_parity.BF16_RANDN_BOUND, measured 1.047e-3)
    tol = {1: 2e-5, 2: 2e-5, 3: 1.1e-3 if str(rec["kind"]) == "randn" else 1e-3, 4: 2.5e-4}[kernel]
    assert np.abs(y - rec["y"]).max() <= tol
    assert np.abs(y - oracle.forward_from_state(rec["x"], rec["state"], pos_emb=rec["pos_emb"])).max() <= tol
