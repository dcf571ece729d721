#!/usr/bin/env python3
"""GPU box: host-to-host throughput of HostPipeline (PCIe-inclusive), next to the serial
copy-compute-copy and the kernel alone."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hand_pose_sl_amd as hps
from hand_pose_sl_amd.stream import HostPipeline

dev = torch.device("cuda:0")
torch.manual_seed(0)
m = hps.ConvModel(30, "ReLU", False, precision="bf16").to(dev).eval()
N, T = 131072, 200
x = (torch.rand((N, T, 12, 2)) - 0.5).pin_memory()
y = torch.empty((N, T, 21, 2), pin_memory=True)
out = {"N": N, "T": T, "bytes_in": x.numel() * 4, "bytes_out": y.numel() * 4}
for chunk in (4096, 16384, 32768):
    pipe = HostPipeline(m, chunk=chunk)
    pipe.run(x, out=y)
    t0 = time.perf_counter(); reps = 3
    for _ in range(reps): pipe.run(x, out=y)
    el = (time.perf_counter() - t0) / reps
    out[f"pipeline_chunk{chunk}"] = {"s": el, "frames_per_s": N * T / el, "h2d_GBs": x.numel() * 4 / el / 1e9,
                                     "d2h_GBs": y.numel() * 4 / el / 1e9}
with torch.no_grad():
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(2):
        xd = x.to(dev, non_blocking=True); yd = m(xd); y.copy_(yd, non_blocking=True); torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / 2
out["serial_copy_compute_copy"] = {"s": el, "frames_per_s": N * T / el}
print(json.dumps(out, indent=1))
