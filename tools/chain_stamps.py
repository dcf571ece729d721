#!/usr/bin/env python3
"""Development (GPU box): s_memtime stamps of one workgroup of the TransformerEnc chain kernel
(a middle launch of the f16x3 path: 3 stages since round 3).  Build with B2H_ABLATE=16384:
    B2H_ABLATE=16384 python -m hand_pose_sl_amd.build --force && python tools/chain_stamps.py [--precision=f16x3]
Per stage, cycles between: acc-init | GEMM | blob->LDS | epilogue(+split) | stores | barrier."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hand_pose_sl_amd as hps
from hand_pose_sl_amd import _lib

PREC = next((a.split("=")[1] for a in sys.argv if a.startswith("--precision=")), "fp32")
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = hps.TransformerEnc(24, 4, 128, 42, 4, precision=PREC).to(dev).eval()
x = (torch.rand((32768, 100, 12, 2)) - 0.5).to(dev)
with torch.no_grad():
    for _ in range(3):
        m(x)
torch.cuda.synchronize()
lib = _lib.load()
buf = (ctypes.c_ulonglong * 512)()
assert lib.b2h_debug_chain_stamps(buf) == 0
a = np.array(buf[:], dtype=np.uint64).reshape(8, 64).astype(np.int64)
names = ["init", "gemm", "blob", "epi", "store", "barrier"]
NS = 3
print("precision", PREC, "(cycles; stage types: LN, ReLU, LN + store of the residual stream)")
for w in range(8):
    row = a[w][:3 + 6 * NS]
    d = np.diff(row)
    print(f"wave {w}: entry->prologue done {d[0]}  first barrier {d[1]}  whole workgroup {row[-1] - row[0]}")
    for s in range(NS):
        seg = d[2 + 6 * s: 8 + 6 * s]
        print("   stage", s, " ".join(f"{n}={v}" for n, v in zip(names, seg.tolist())), " total", int(seg.sum()))
