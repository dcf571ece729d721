#!/usr/bin/env python3
"""Development (GPU box): s_memtime stamps of ONE workgroup (8 waves) of the persistent 16-bit conv kernel at the
phase boundaries of chunks 40..71 of each wave, on the bench shape.
    hipcc ... -DB2H_ABLATE=65536 -o tools/_build/libb2h_stamps.so hand_pose_sl_amd/csrc/b2h_api.hip
    python tools/conv16_stamps.py tools/_build/libb2h_stamps.so [seqs=262144]
Phases per chunk: commit | issue (20 loads) | L0 | L1 | wait vmcnt(0) | cast + L2 | head; `gap` = end of one chunk's
head to the next stamp 0 (loop overhead).  Prints per-wave medians in cycles and the waves' timelines relative to
the workgroup's first stamp, so that one can see which phases of the SIMD partners (waves w and w+4 share a SIMD
in the 0->2->1->3 placement order only by chance: tools/simd_map.hip) overlap."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hand_pose_sl_amd import _lib
import torch.nn as nn

path = sys.argv[1]
S = int(sys.argv[2]) if len(sys.argv) > 2 else 262144
T = 200
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
for _ in range(3):
    assert lib.b2h_time_forward(h, ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(y.data_ptr()), S, T, _lib.KERNELS["bf16"], 20, None, ctypes.byref(ms)) == 0
print(f"launch {ms.value*1e3:.1f} us (stamped build)")
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (8 * 32 * 8))()
lib.b2h_debug_conv16_stamps.restype = ctypes.c_int
assert lib.b2h_debug_conv16_stamps(buf) == 0
a = np.array(buf[:], dtype=np.uint64).reshape(8, 32, 8).astype(np.int64)
names = ["commit", "issue", "L0", "L1", "wait", "cast+L2", "head"]
print("per-wave medians over 32 chunks, cycles (s_memtime ticks):")
for w in range(8):
    d = np.diff(a[w], axis=1)                       # (32, 7)
    gap = a[w, 1:, 0] - a[w, :-1, 7]
    per = a[w, 1:, 0] - a[w, :-1, 0]
    print(f"wave {w}: " + " ".join(f"{n}={int(np.median(d[:, i]))}" for i, n in enumerate(names)) +
          f" gap={int(np.median(gap))} period={int(np.median(per))} (min {per.min()} max {per.max()})")
d = np.diff(a, axis=2).reshape(-1, 7)
print("all waves: " + " ".join(f"{n}: med {int(np.median(d[:, i]))} p10 {int(np.percentile(d[:, i], 10))} p90 {int(np.percentile(d[:, i], 90))}" for i, n in enumerate(names)))
t0 = a[:, 0, 0].min()
print("timeline of chunks 0..3 (cycles since the workgroup's first stamp): stamp 0..7 per chunk")
for w in range(8):
    print(f"wave {w}: " + " | ".join(" ".join(str(int(v - t0)) for v in a[w, c]) for c in range(4)))

# spans: when every wave of every workgroup started and ended (s_memrealtime, 10-ns ticks), and how many chunks it did
sp = (ctypes.c_ulonglong * (256 * 8 * 3))()
lib.b2h_debug_conv16_spans.restype = ctypes.c_int
assert lib.b2h_debug_conv16_spans(sp) == 0
sp = np.array(sp[:], dtype=np.uint64).reshape(256, 8, 3).astype(np.int64)
t0 = sp[:, :, 0].min()
start, end, cnt = (sp[:, :, 0] - t0) / 100.0, (sp[:, :, 1] - t0) / 100.0, sp[:, :, 2]
print(f"wave start: max {start.max():.1f} us after the first; wave end: min {end.min():.1f} median {np.median(end):.1f} max {end.max():.1f} us")
wg_end = end.max(axis=1)
print(f"workgroup end (its last wave): min {wg_end.min():.1f} p10 {np.percentile(wg_end,10):.1f} median {np.median(wg_end):.1f} p90 {np.percentile(wg_end,90):.1f} max {wg_end.max():.1f} us")
print(f"idle wave-time before the kernel's end: {100*(end.max()-end).mean()/end.max():.2f} % of the kernel")
print("chunks per wave: waves 0-3 mean %.1f, waves 4-7 mean %.1f (of %d per workgroup)" % (cnt[:, :4].mean(), cnt[:, 4:].mean(), cnt.sum(axis=1).mean()))
xcd = np.arange(256) % 8
print("workgroup end by blockIdx % 8 (XCD group), mean us: " + " ".join(f"{wg_end[xcd==i].mean():.1f}" for i in range(8)))
