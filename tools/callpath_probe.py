#!/usr/bin/env python3
"""GPU box: where the host time of one small-batch module call goes (allocation, stream lookup,
ctypes, the weight-replacement check, the launch) -- behind DESIGN.md's config-1 latencies."""
import ctypes, time, torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hand_pose_sl_amd as hps
from hand_pose_sl_amd import _lib
dev = torch.device("cuda:0")
m = hps.ConvModel(30, "ReLU", False, precision="bf16").to(dev).eval()
x = torch.rand((1, 200, 12, 2), device=dev); y = torch.empty((1, 200, 21, 2), device=dev)
lib = _lib.load()
def t(fn, n=3000):
    for _ in range(200): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6
xp, yp = ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(y.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
m(x)  # creates the native handle
h = m._handle
print("raw ctypes b2h_forward            %.2f us" % t(lambda: lib.b2h_forward(h, xp, yp, 1, 200, 3, st)))
print("+ c_void_p construction           %.2f us" % t(lambda: lib.b2h_forward(h, ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(y.data_ptr()), 1, 200, 3, st)))
print("torch.empty alone                 %.2f us" % t(lambda: torch.empty((1, 200, 21, 2), dtype=torch.float32, device=dev)))
print("current_stream().cuda_stream      %.2f us" % t(lambda: torch.cuda.current_stream(dev).cuda_stream))
print("forward_into                      %.2f us" % t(lambda: m.forward_into(x, y)))
with torch.no_grad():
    print("module call (no_grad)             %.2f us" % t(lambda: m(x)))
    print("m.forward (no __call__)           %.2f us" % t(lambda: m.forward(x)))
k = torch.zeros(1, device=dev)
print("torch elementwise add_ (reference point) %.2f us" % t(lambda: k.add_(1)))
rc = lib.b2h_forward(h, xp, yp, 1, 200, 3, st); print("rc", rc, lib.b2h_last_error())
import cProfile, pstats
with torch.no_grad():
    pr = cProfile.Profile(); pr.enable()
    for _ in range(2000): m.forward_into(x, y)
    pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(12)

te = hps.TransformerEnc(24, 4, 128, 42, 4).to(dev).eval()
xt = torch.rand((1, 100, 12, 2), device=dev)
with torch.no_grad():
    print("TransformerEnc module call (1,100)  %.2f us" % t(lambda: te(xt), 1000))
