#!/usr/bin/env python3
"""GPU box: throughput of every kernel that supports a width, for conv_channels across the whole
range the reference's --conv-channels allows here (run.py:37), at (B, T) = (8192, 200) by default.
    python tools/bench_widths.py [B=8192] [T=200]"""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hand_pose_sl_amd as hps

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
T = int(sys.argv[2]) if len(sys.argv) > 2 else 200
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
x = torch.rand((B, T, 12, 2), device=dev, generator=g) - 0.5
y = torch.empty((B, T, 21, 2), device=dev)
out = {"B": B, "T": T, "rows": []}
for C in (8, 10, 12, 16, 30, 32, 33, 36, 40, 48, 56, 64):
    torch.manual_seed(C)
    m = hps.ConvModel(C, "ReLU", False).to(dev).eval()
    for prec in ("bf16", "f16", "f16x3", "f32_mfma", "f32_valu"):
        try:
            m.time_forward(x, y, 3, precision=prec)
            ms = min(m.time_forward(x, y, 20, precision=prec) for _ in range(3))
        except RuntimeError:
            continue
        row = {"C": C, "precision": prec, "kernel": m.kernel_name(prec), "ms": ms, "G_frames_per_s": B * T / ms / 1e6}
        out["rows"].append(row)
        print(f"C={C:2d} {prec:9s} {row['kernel']:28s} {ms:8.3f} ms  {row['G_frames_per_s']:7.2f} G frames/s", flush=True)
print(json.dumps(out))
