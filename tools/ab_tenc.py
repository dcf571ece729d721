#!/usr/bin/env python3
"""GPU box: same-process, interleaved A/B of two or more builds of libb2h.so on the TransformerEnc path.
    python tools/ab_tenc.py <libA.so> <libB.so> [...] [precision=f16x3] [B=32768] [T=100]
Each library gets its own ctypes handle and its own model (same seed, same weights); the Python binding looks
the library up per call, so the handle is swapped in before each timed burst.  Prints median / min ms per
forward and whether the outputs equal the first library's bit for bit."""
import ctypes, os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hand_pose_sl_amd as hps
from hand_pose_sl_amd import _lib

paths = [a for a in sys.argv[1:] if a.endswith(".so")]
rest = [a for a in sys.argv[1:] if not a.endswith(".so")]
prec = rest[0] if len(rest) > 0 else "f16x3"
B = int(rest[1]) if len(rest) > 1 else 32768
T = int(rest[2]) if len(rest) > 2 else 100
dev = torch.device("cuda:0")


def typed(path):
    lib = ctypes.CDLL(os.path.abspath(path))
    for name, (res, args) in _lib.SYMBOLS.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    return lib


libs = [typed(p) for p in paths]
models = []
for lib in libs:
    _lib._lib = lib
    torch.manual_seed(0)
    models.append(hps.TransformerEnc(24, 4, 128, 42, 4, precision=prec).to(dev).eval())
x = (torch.rand((B, T, 12, 2)) - 0.5).to(dev)
outs = []
with torch.no_grad():
    for lib, m in zip(libs, models):
        _lib._lib = lib
        for _ in range(3):
            y = m(x)
        torch.cuda.synchronize()
        outs.append(y.clone())
    print("outputs identical to the first:", [bool(torch.equal(outs[0], o)) for o in outs[1:]],
          " max |diff|:", [float((outs[0] - o).abs().max()) for o in outs[1:]])
    times = [[] for _ in libs]
    for rnd in range(7):
        for i, (lib, m) in enumerate(zip(libs, models)):
            _lib._lib = lib
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                m(x)
            e1.record()
            torch.cuda.synchronize()
            times[i].append(e0.elapsed_time(e1) / 10)
for p, t in zip(paths, times):
    print(f"{p}: median {statistics.median(t):.3f} ms  min {min(t):.3f} ms  ({B * T / statistics.median(t) / 1e3:.1f} M frames/s)")
