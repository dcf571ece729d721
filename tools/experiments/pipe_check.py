#!/usr/bin/env python3
"""GPU box: the layer-pipeline f16x3 kernel (kernel_pipe3.h) against the wave-per-chunk kernel it must equal
bit for bit (B2H_NO_PIPE=1 selects the latter in the same library), over sequence lengths and batch sizes.
    python tools/pipe_check.py [lib.so]"""
import ctypes, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hand_pose_sl_amd import _lib
import torch.nn as nn

path = sys.argv[1] if len(sys.argv) > 1 else _lib.lib_path()
dev = torch.device("cuda:0")
torch.manual_seed(0)
convs = [nn.Conv1d(24, 30, 5, padding=2), nn.Conv1d(30, 30, 5, padding=2), nn.Conv1d(30, 30, 5, padding=2), nn.Conv1d(30, 42, 5, padding=2)]
ps = [p.detach().to(dev).contiguous() for c in convs for p in (c.weight, c.bias)]
lib = ctypes.CDLL(os.path.abspath(path))
for name, (res, args) in _lib.SYMBOLS.items():
    if hasattr(lib, name):
        getattr(lib, name).restype = res
        getattr(lib, name).argtypes = args
h = ctypes.c_void_p()
assert lib.b2h_create(30, b"ReLU", 0, ctypes.byref(h)) == 0
assert lib.b2h_load_weights(h, *[ctypes.c_void_p(t.data_ptr()) for t in ps], 1) == 0
K = _lib.KERNELS["f16x3"]

def run(x, nopipe):
    if nopipe:
        os.environ["B2H_NO_PIPE"] = "1"
    else:
        os.environ.pop("B2H_NO_PIPE", None)
    y = torch.full((x.shape[0], x.shape[1], 21, 2), float("nan"), device=dev)
    rc = lib.b2h_forward(h, ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(y.data_ptr()), x.shape[0], x.shape[1], K, None)
    assert rc == 0, lib.b2h_last_error()
    torch.cuda.synchronize()
    return y

bad = 0
for T in (200, 1, 2, 15, 16, 17, 31, 32, 33, 47, 48, 49, 63, 64, 65, 100, 127, 128, 129, 199, 201, 1000, 3001):
    for B in (1024, 1025, 2000, 4099):
        if B * T > 40_000_000:
            continue
        x = torch.rand((B, T, 12, 2), device=dev) - 0.5
        t0 = time.time()
        yp = run(x, False)
        dt = time.time() - t0
        yc = run(x, True)
        same = torch.equal(yp, yc)
        bad += 0 if same else 1
        if not same:
            d = (yp - yc).abs()
            nanp = int(torch.isnan(yp).sum())
            print(f"T={T} B={B}: MISMATCH max {float(torch.nan_to_num(d, nan=9e9).max()):.3e} nan_in_pipe={nanp} first bad seq {int((d.flatten(1).amax(1) > 0).nonzero()[0]) if (d.flatten(1).amax(1) > 0).any() else -1} ({dt*1e3:.1f} ms)")
        else:
            print(f"T={T} B={B}: identical ({dt*1e3:.1f} ms)")
print("mismatching cases:", bad)
sys.exit(1 if bad else 0)
