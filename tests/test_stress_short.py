"""Time-boxed runs of the randomized sweeps in tools/ (each a subprocess, fixed seed): random
shapes, widths, kernels and fused flags against the oracle.  The long versions of these sweeps
found the fused-kernel store hazard; the short ones keep guarding it."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tool,args", [("stress_conv.py", ["20", "11"]), ("stress_tenc.py", ["15", "3"]),
                                       ("stress_hostpipe.py", ["10"])])
def test_randomized_sweep(tool, args, cuda_device):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool)] + args, capture_output=True, text=True,
                       timeout=600, cwd=ROOT)
    tail = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else ""
    assert r.returncode == 0 and tail.startswith("PASS"), (r.stdout[-1500:], r.stderr[-1500:])
