"""bench.py's output contract: rank 0 prints exactly ONE JSON line on stdout, with the keys the
driver reads plus the `roofline` and `cpu_baseline` objects."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
            "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"]


@pytest.mark.gpu
def test_one_json_line_with_the_contract_keys(cuda_device):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "1",
                        "--seqs", "2048", "--cpu-seconds", "1.0"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = r.stdout.splitlines()
    assert len(lines) == 1, r.stdout[:2000]
    d = json.loads(lines[0])
    for k in REQUIRED:
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 1
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["unit"] == "frames/s" and d["data"] == "synthetic" and d["dtype"] == "bf16"
    assert d["value"] > 1e8 and abs(d["value"] - 2048 * 200 * 4 / (d["ms_per_step"] * 4e-3)) / d["value"] < 1e-6
    assert "workload" in d["config"] and "model" not in d["config"]
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12 and 0 < rf["frac"] < 1
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["unit"] == "frames/s" and cb["cores"] >= 1 and cb["value"] > 0 and cb["sample"]
    assert cb["gpu_max_abs_err_on_sample"] < 1.5e-3


def test_bench_refuses_without_a_gpu():
    """No CPU path in the product: on a box without an MI355X bench.py exits with a message, not a number."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode != 0 and r.stdout.strip() == "" and "MI355X" in r.stderr
