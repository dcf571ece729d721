#!/usr/bin/env python3
"""GPU box: per-wave cycle accounts of one workgroup of the layer-pipeline kernel (B2H_ABLATE=131072 build):
total cycles and the cycles each stage spent waiting for input tiles, for room in its output ring and for
sequence announcements.  The stage that never waits is the pipeline's pace-setter.
    python tools/pipe_accounts.py tools/_build/lib_pipe_acc.so [seqs=65536] [T=200]"""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hand_pose_sl_amd import _lib
import torch.nn as nn
path = sys.argv[1]
S = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
T = int(sys.argv[3]) if len(sys.argv) > 3 else 200
dev = torch.device("cuda:0")
torch.manual_seed(0)
convs = [nn.Conv1d(24, 30, 5, padding=2), nn.Conv1d(30, 30, 5, padding=2), nn.Conv1d(30, 30, 5, padding=2), nn.Conv1d(30, 42, 5, padding=2)]
ps = [p.detach().to(dev).contiguous() for c in convs for p in (c.weight, c.bias)]
x = torch.rand((S, T, 12, 2), device=dev) - 0.5
y = torch.empty((S, T, 21, 2), device=dev)
lib = ctypes.CDLL(os.path.abspath(path))
for name, (res, args) in _lib.SYMBOLS.items():
    if hasattr(lib, name):
        getattr(lib, name).restype = res
        getattr(lib, name).argtypes = args
h = ctypes.c_void_p()
assert lib.b2h_create(30, b"ReLU", 0, ctypes.byref(h)) == 0
assert lib.b2h_load_weights(h, *[ctypes.c_void_p(t.data_ptr()) for t in ps], 1) == 0
ms = ctypes.c_float()
for _ in range(2):
    assert lib.b2h_time_forward(h, ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(y.data_ptr()), S, T, _lib.KERNELS["f16x3"], 10, None, ctypes.byref(ms)) == 0
print(f"launch {ms.value*1e3:.1f} us (instrumented build), {S*T/ms.value/1e6:.2f} G frames/s")
buf = (ctypes.c_ulonglong * 32)()
lib.b2h_debug_pipe_accounts.restype = ctypes.c_int
assert lib.b2h_debug_pipe_accounts(buf) == 0
a = np.array(buf[:], dtype=np.uint64).reshape(8, 4).astype(np.int64)
stage_of = lambda w: (w & 3) if w < 4 else ((w + 2) & 3)
names = ["front", "L1", "L2", "head"]
tiles = (S // 512) * ((T + 15) // 16)
for w in range(8):
    tot, win, wout, wann = a[w]
    print(f"wave {w} ({names[stage_of(w)]:5s} pipeline {w >> 2}): total {tot} cycles ({tot/max(tiles,1):.0f} per tile), waiting: input {100*win/tot:.1f} %  room {100*wout/tot:.1f} %  announcements {100*wann/tot:.1f} %  -> busy {100*(tot-win-wout-wann)/tot:.1f} % = {(tot-win-wout-wann)/max(tiles,1):.0f} cycles per tile")
