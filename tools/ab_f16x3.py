#!/usr/bin/env python3
"""GPU box: A/B of the persistent f16x3 kernel against the wave-per-chunk form it replaces
(B2H_F16X3_CHUNK_KERNEL=1 selects the old dispatch; read once per process), interleaved rounds.
    python tools/ab_f16x3.py            # prints both; run under each env setting for a cross-process pair"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hand_pose_sl_amd as hps

dev = torch.device("cuda:0")
torch.manual_seed(0)
m = hps.ConvModel(30, "ReLU", False, precision="f16x3").to(dev).eval()
tag = "chunk-kernel" if os.environ.get("B2H_F16X3_CHUNK_KERNEL") else "persistent"
for B, T in ((65536, 200), (8192, 200), (256, 200), (1, 200), (4096, 100), (1024, 1000)):
    x = torch.rand((B, T, 12, 2), device=dev) - 0.5
    y = torch.empty((B, T, 21, 2), device=dev)
    m.time_forward(x, y, 5)
    ms = min(m.time_forward(x, y, 20 if B * T > 1e6 else 200) for _ in range(3))
    print(f"{tag:13s} {m.kernel_name():28s} ({B},{T}): {ms*1e3:9.1f} us  {B*T/ms/1e6:7.2f} G frames/s", flush=True)
