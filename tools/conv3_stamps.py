#!/usr/bin/env python3
"""Development (GPU box): s_memtime stamps of one workgroup (4 waves) of b2h_fwd_mfma_f16x3.
    B2H_ABLATE=32768 python -m hand_pose_sl_amd.build --force && python tools/conv3_stamps.py
Segments per wave: input staging | per layer: weight-fragment request, tile loop."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hand_pose_sl_amd as hps
from hand_pose_sl_amd import _lib

dev = torch.device("cuda:0")
torch.manual_seed(0)
m = hps.ConvModel(30, "ReLU", False, precision="f16x3").to(dev).eval()
x = (torch.rand((65536, 200, 12, 2), device=dev) - 0.5)
with torch.no_grad():
    for _ in range(3):
        m(x)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 64)()
assert _lib.load().b2h_debug_conv3_stamps(buf) == 0
a = np.array(buf[:], dtype=np.uint64).reshape(4, 16).astype(np.int64)
names = ["stage"] + [f"L{l}:{p}" for l in range(4) for p in ("w", "tiles")]
for w in range(4):
    d = np.diff(a[w][:10])
    print("wave", w, " ".join(f"{n}={v}" for n, v in zip(names, d.tolist())), " total", int(d.sum()))
