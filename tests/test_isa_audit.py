"""Static check of the compiled gfx950 kernels (no GPU needed): no buffer/global store may have
its data registers rewritten within two wait states (tools/store_war_audit.py) -- the
store-data write-after-read hazard that produced wrong frames in the fused 16-bit kernel
(DESIGN.md section 4).  Compiles b2h_api.hip once with --save-temps (about a minute)."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_no_close_store_data_overwrite(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    src = os.path.join(ROOT, "hand_pose_sl_amd", "csrc", "b2h_api.hip")
    r = subprocess.run([hipcc, "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-function",
                        "--save-temps", "-o", "x.so", src], cwd=tmp_path, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    listing = [f for f in os.listdir(tmp_path) if f.endswith("gfx950.s")]
    assert listing, os.listdir(tmp_path)
    a = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "store_war_audit.py"),
                        os.path.join(tmp_path, listing[0]), "2"], capture_output=True, text=True, timeout=300)
    assert a.returncode == 0, a.stderr
    assert a.stdout.strip().splitlines()[-1] == "total 0", a.stdout
