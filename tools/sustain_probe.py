#!/usr/bin/env python3
"""GPU box: does a kernel's time per launch depend on how long the back-to-back burst is?
(power / clock management reacts on a millisecond scale: a short burst can run faster than a
sustained one).    python tools/sustain_probe.py <lib.so> [<lib2.so>] [precision=bf16]"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hand_pose_sl_amd import _lib
paths = [a for a in sys.argv[1:] if a.endswith(".so")]
prec = [a for a in sys.argv[1:] if not a.endswith(".so")] or ["bf16"]
k = _lib.KERNELS[prec[0]]
dev = torch.device("cuda:0")
torch.manual_seed(0)
import torch.nn as nn
convs = [nn.Conv1d(24, 30, 5, padding=2), nn.Conv1d(30, 30, 5, padding=2), nn.Conv1d(30, 30, 5, padding=2), nn.Conv1d(30, 42, 5, padding=2)]
ps = [p.detach().to(dev).contiguous() for c in convs for p in (c.weight, c.bias)]
libs = []
for p in paths:
    lib = ctypes.CDLL(os.path.abspath(p))
    for name, (res, args) in _lib.SYMBOLS.items():
        if hasattr(lib, name):
            getattr(lib, name).restype = res
            getattr(lib, name).argtypes = args
    h = ctypes.c_void_p()
    assert lib.b2h_create(30, b"ReLU", 0, ctypes.byref(h)) == 0
    assert lib.b2h_load_weights(h, *[ctypes.c_void_p(t.data_ptr()) for t in ps], 1) == 0
    libs.append((p, lib, h))
for S in (65536, 262144):
    x = torch.rand((S, 200, 12, 2), device=dev) - 0.5
    y = torch.empty((S, 200, 21, 2), device=dev)
    for p, lib, h in libs:
        row = []
        for iters in (5, 20, 80, 320):
            torch.cuda.synchronize()
            import time; time.sleep(0.3)   # idle gap before each burst
            ms = ctypes.c_float()
            rc = lib.b2h_time_forward(h, ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(y.data_ptr()), S, 200, k, iters, None, ctypes.byref(ms))
            assert rc == 0
            row.append(f"{iters:4d} launches: {ms.value * 1e3:8.1f} us")
        print(f"S={S:6d} {os.path.basename(p):18s} " + " | ".join(row), flush=True)
    del x, y
