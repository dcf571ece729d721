"""bench.py's output contract: rank 0 prints exactly ONE JSON line on stdout, with the keys the
driver reads plus the `roofline`, `kernels` and `cpu_baseline` objects; `--gpus N` from a plain
shell starts the N ranks itself; no GPU / no RCCL means a non-zero exit and no line."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")
REQUIRED = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
            "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"]
# north_star: <= 1e-3 max-abs vs the reference CPU forward.  The bench sample is U[-0.5, 0.5]
# (normalised keypoints), where the bf16 headline kernel measures 3.7-5.0e-4.
SAMPLE_TOL = {"bf16": 1e-3, "f16": 2.5e-4, "f16x3": 2e-5, "f32_mfma": 2e-5}


def _check_roofline(rf):
    assert rf["bound"] in ("hbm", "mfma") and rf["unit"] in ("GB/s", "TFLOP/s")
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12 and 0 < rf["frac"] < 1
    assert rf["launch_ms"] > 0 and rf["kernel"]


@pytest.mark.gpu
def test_one_json_line_with_the_contract_keys(cuda_device):
    r = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--steps", "4", "--warmup", "1",
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
    # the driver's protocol exactly: W warmup steps are the only launches before the K timed ones
    assert d["config"]["preconditioning_steps"] == 0 and d["untimed_launches"] == 1
    vs = d["value_sustained"]                               # the settled-stream rate, beside `value`, never as it
    assert vs["value"] > 1e8 and vs["untimed_launches_before"] == 1 + 4 + 200
    anyin = d["value_le_1e-3_any_input"]                    # the precision that meets north_star's gate on any input
    assert anyin["precision"] == "f16" and anyin["value"] == d["kernels"]["f16"]["frames_per_s"]
    assert len(d["config"]["gpus"]) == 1 and d["config"]["gpus"][0]["pci"]
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0 and "traffic" in rf
    _check_roofline(rf)
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["unit"] == "frames/s" and cb["cores"] >= 1 and cb["value"] > 0 and cb["sample"]
    # fixed protocol: every thread count's rate is in the line, `value` is the best of them, load average beside it
    assert cb["value"] == max(cb["rates_by_threads"].values()) and str(cb["cores"]) in cb["rates_by_threads"]
    assert "1" in cb["rates_by_threads"] and cb["loadavg_1m_before"] >= 0 and cb["host_cores_available"] >= 1
    # every precision of the path is graded on the same shard and checked on the CPU sample
    assert set(d["kernels"]) == {"bf16", "f16", "f16x3", "f32_mfma"}
    for prec, rec in d["kernels"].items():
        _check_roofline(rec)
        assert rec["gpu_max_abs_err_on_sample"] <= SAMPLE_TOL[prec], (prec, rec["gpu_max_abs_err_on_sample"])
    assert d["kernels"]["f32_mfma"]["bound"] == "mfma" and d["kernels"]["bf16"]["bound"] == "hbm"
    assert cb["gpu_max_abs_err_on_sample"] == d["kernels"]["bf16"]["gpu_max_abs_err_on_sample"]


@pytest.mark.gpu
def test_self_launch_two_ranks_rehearsal(cuda_device):
    """`python bench.py --gpus 2` from a plain shell spawns the two ranks itself.  On the one-GPU box
    RCCL cannot serve two ranks on one device, so the default backend must FAIL loudly (no line);
    `--backend gloo` rehearses the plumbing: one line, two ranks, the gather object, the GPU census."""
    base = [sys.executable, BENCH, "--gpus", "2", "--steps", "3", "--warmup", "1", "--seqs", "4096"]
    import torch
    if torch.cuda.device_count() < 2:
        r = subprocess.run(base, capture_output=True, text=True, timeout=900, cwd=ROOT)
        assert r.returncode != 0 and r.stdout.strip() == "", (r.returncode, r.stdout[:500])
        assert "RCCL" in r.stderr or "GPU" in r.stderr
    r = subprocess.run(base + ["--backend", "gloo"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = r.stdout.splitlines()
    assert len(lines) == 1, r.stdout[:2000]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and "gloo" in d["config"]["backend"]
    assert len(d["config"]["gpus"]) == 2
    assert abs(d["value"] - 2 * 4096 * 200 * 3 / (d["ms_per_step"] * 3e-3)) / d["value"] < 1e-6
    g = d["gather"]
    for part in ("pipelined", "config4"):
        assert g[part]["frames_per_s_incl"] > 0 and g[part]["frames_per_s_excl"] >= g[part]["frames_per_s_incl"] * 0.5
        assert g[part]["bytes_into_rank0"] > 0
    assert g["config4"]["bytes_into_rank0"] == 1000 * 200 * 168
    assert "cpu_baseline" not in d            # rank 0 at N = 1 only
    # the hand-back is a secondary figure: if it outlives its budget the line is still printed, with
    # `value` / `roofline` intact and the failure named in the `gather` object
    # ... and the exit code says so too (3 = `value` stands, the hand-back measurement failed)
    r = subprocess.run(base + ["--backend", "gloo", "--gather-budget", "0.2"], capture_output=True, text=True,
                       timeout=900, cwd=ROOT)
    assert r.returncode == 3, (r.returncode, r.stderr[-3000:])
    lines = r.stdout.splitlines()
    assert len(lines) == 1, r.stdout[:2000]
    d2 = json.loads(lines[0])
    assert d2["n_gpus"] == 2 and d2["value"] > 1e8 and d2["roofline"]["frac"] > 0 and "error" in d2["gather"]


def test_bench_refuses_without_a_gpu():
    """No CPU path in the product: on a box without an MI355X bench.py exits with a message, not a number."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    r = subprocess.run([sys.executable, BENCH, "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode != 0 and r.stdout.strip() == "" and "MI355X" in r.stderr


def test_self_launch_propagates_rank_failure():
    """`--gpus 2` without WORLD_SIZE starts `python -m torch.distributed.run` itself (before any
    GPU call); here, without a GPU, both ranks refuse and the parent must relay the failure:
    non-zero exit code, nothing on stdout."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0", "--backend", "gloo"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode != 0 and r.stdout.strip() == ""
    assert "self-launch" in r.stderr and "torch.distributed.run" in r.stderr and "MI355X" in r.stderr


def _guard_worker(rank, world, port, mode, out_dir):
    """One rank of a SecondaryGuard scenario.  The store is a plain TCPStore (what torch.distributed's
    rendezvous gives bench.py); no collective backend is involved -- in the scenarios below the
    'collective' is a rank that waits for a peer that never comes."""
    import time
    from datetime import timedelta

    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    import bench
    store = dist.TCPStore("127.0.0.1", port, world, is_master=(rank == 0), timeout=timedelta(seconds=60))

    def emit(reason):
        with open(os.path.join(out_dir, "line.json"), "w") as f:
            json.dump({"value": 1.0, "gather": {"error": reason}}, f)

    def phase():
        if mode == "peer_raises" and rank == 1:
            time.sleep(0.3)
            raise RuntimeError("NCCL error: unhandled system error (injected)")
        if mode == "root_raises" and rank == 0:
            raise RuntimeError("injected failure on root")
        if mode in ("peer_raises", "root_raises", "hang"):
            time.sleep(120)           # parked in a 'collective' the failing rank never joins
        return {"ok": rank}

    g = bench.SecondaryGuard(store, rank, world, budget=(1.0 if mode == "hang" else 60.0), emit=emit, poll=0.05, grace=10.0)
    res = g.run(phase)
    with open(os.path.join(out_dir, f"ok{rank}.json"), "w") as f:
        json.dump(res, f)
    if rank == 0:
        time.sleep(0.5)               # the store's host leaves last


@pytest.mark.parametrize("mode", ["ok", "peer_raises", "root_raises", "hang"])
def test_secondary_guard_exit_codes(tmp_path, mode):
    """The N > 1 hand-back measurement is secondary: a failure in ANY rank (exception on a peer, on root, or
    a phase that outlives its budget) leaves the line -- written by rank 0 BEFORE any rank exits -- and every
    rank's exit code is 3; a clean phase returns its result on every rank (bench.py: SecondaryGuard)."""
    import multiprocessing as mp
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_guard_worker, args=(r, 2, port, mode, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(90)
        assert p.exitcode is not None, "a rank is still alive"
    codes = [p.exitcode for p in procs]
    if mode == "ok":
        assert codes == [0, 0]
        assert json.load(open(tmp_path / "ok0.json")) == {"ok": 0} and json.load(open(tmp_path / "ok1.json")) == {"ok": 1}
        assert not (tmp_path / "line.json").exists()
    else:
        assert codes == [3, 3], codes
        err = json.load(open(tmp_path / "line.json"))["gather"]["error"]
        assert {"peer_raises": "rank 1: RuntimeError: NCCL error", "root_raises": "rank 0: RuntimeError: injected",
                "hang": "did not finish within 1 s"}[mode] in err
        assert not (tmp_path / "ok0.json").exists() and not (tmp_path / "ok1.json").exists()


def test_self_launch_exit_code_when_the_handback_failed(tmp_path, monkeypatch):
    """`python bench.py --gpus N` relays rank 0's line and turns a failed hand-back into exit code 3 (any other
    rank failure keeps the launcher's code)."""
    sys.path.insert(0, ROOT)
    import bench

    class R:
        def __init__(self, rc, out):
            self.returncode, self.stdout = rc, out.encode()

    for rc, out, want in ((1, '{"value": 2.0, "gather": {"error": "rank 1: boom"}}\n', 3), (1, "", 1), (0, '{"value": 2.0}\n', 0)):
        monkeypatch.setattr(bench.subprocess, "run", lambda *a, _r=R(rc, out), **k: _r)
        with pytest.raises(SystemExit) as ei:
            bench.self_launch(bench.parse(["--gpus", "2"]), ["--gpus", "2"])
        assert ei.value.code == want


def test_traffic_is_nulled_when_the_kernel_sources_changed(tmp_path, monkeypatch):
    """roofline.traffic is a committed PMC measurement: it is reported only while the kernel sources
    still hash to what it was measured on, and only for the exact workload it was measured for."""
    sys.path.insert(0, ROOT)
    import bench
    rec = {"seqs_per_gpu": 262144, "frames_per_seq": 200, "kernel": "b2h_fwd_mfma16", "precision": "bf16",
           "traffic_bytes": 13.86e9, "sources_sha256": bench.sources_sha256()}
    p = tmp_path / "traffic.json"
    p.write_text(json.dumps(rec))
    name = "b2h_fwd_mfma16<1, false>"
    assert bench.committed_traffic(262144, 200, name, "bf16", str(p))["traffic"] == 13.86e9
    assert bench.committed_traffic(65536, 200, name, "bf16", str(p))["traffic"] is None      # other workload
    assert bench.committed_traffic(262144, 200, name, "f16", str(p))["traffic"] is None       # other precision
    monkeypatch.setattr(bench, "sources_sha256", lambda: "0" * 64)                              # sources edited
    r = bench.committed_traffic(262144, 200, name, "bf16", str(p))
    assert r["traffic"] is None and "mismatch" in r["traffic_note"]
    assert bench.committed_traffic(262144, 200, name, "bf16", str(tmp_path / "absent.json"))["traffic"] is None
