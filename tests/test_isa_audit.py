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


def test_audit_detects_the_pattern(tmp_path):
    """The audit on synthetic listings: the exact sequence that broke the fused kernel is reported,
    padded / branch-separated / MFMA-late-writer variants are handled as documented."""
    audit = os.path.join(ROOT, "tools", "store_war_audit.py")

    def total(body, window="2"):
        f = tmp_path / "k.s"
        f.write_text("_Z1kv:\n" + body + "\n\ts_endpgm\n")
        r = subprocess.run([sys.executable, audit, str(f), window], capture_output=True, text=True, timeout=60)
        assert r.returncode == 0, r.stderr
        return int(r.stdout.strip().splitlines()[-1].split()[1])

    bad = "\tbuffer_store_dwordx4 v[106:109], v226, s[36:39], s93 offen\n\tv_pk_mul_f32 v[106:107], v[186:187], v[100:101]"
    assert total(bad) == 1
    assert total(bad.replace("v_pk_mul_f32 v[106:107]", "v_pk_mul_f32 v[110:111]")) == 0          # other registers
    assert total(bad.replace("offen\n", "offen\n\ts_nop 3\n")) == 0                             # padded
    assert total(bad.replace("offen\n", "offen\n.LBB0_1:\n")) == 0                              # new basic block
    assert total("\tglobal_store_dwordx4 v[0:1], v[142:145], off\n\tv_mov_b32_e32 v143, v7") == 1
    assert total("\tbuffer_store_dwordx2 v[114:115], v116, s[8:11], 0 offen\n\tv_cndmask_b32_e32 v114, v74, v118, vcc") == 1
