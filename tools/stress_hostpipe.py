#!/usr/bin/env python3
"""GPU box: HostPipeline (pinned staging, three streams) with random sizes, chunk lengths, depths
and kernels must return exactly what one direct forward returns.   python tools/stress_hostpipe.py [seconds=60]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hand_pose_sl_amd as hps
from hand_pose_sl_amd.stream import HostPipeline

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(0)
dev = torch.device("cuda:0")
models = {p: hps.ConvModel(30, "ReLU", False, precision=p).to(dev).eval() for p in ("bf16", "fp32", "f16x3")}
t_end, n = time.time() + budget, 0
while time.time() < t_end:
    prec = str(rng.choice(list(models)))
    m = models[prec]
    N, T = int(rng.integers(1, 2500)), int(rng.choice([1, 7, 48, 100, 200, rng.integers(1, 400)]))
    if N * T > 300000:
        N = max(1, 300000 // T)
    chunk, depth = int(rng.integers(1, 900)), int(rng.integers(2, 5))
    x = torch.from_numpy(rng.random((N, T, 12, 2), dtype=np.float32) - 0.5)
    with torch.no_grad():
        want = m(x.to(dev)).cpu()
        got = HostPipeline(m, chunk=chunk, depth=depth).run(x)
    n += 1
    if not torch.equal(torch.as_tensor(got).cpu(), want):
        print(f"FAIL case {n}: prec={prec} N={N} T={T} chunk={chunk} depth={depth}")
        sys.exit(1)
print(f"PASS: {n} random HostPipeline cases in {budget:.0f} s")
