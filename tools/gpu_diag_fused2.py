#!/usr/bin/env python3
"""Developer diagnostic: fused kernel with an identity transform (x1.0) must be
bit-identical to the plain kernel; print where it is not."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hand_pose_sl_amd as hps
from hand_pose_sl_amd import _lib
dev = torch.device("cuda:0")
torch.manual_seed(0)
for prec in ("bf16",):
    m = hps.ConvModel(30, "ReLU", False, precision=prec).to(dev).eval()
    lib = m._ensure_handle()
    for (B, T) in ((1, 40), (1, 200), (300, 200)):
        x = (torch.rand((B, T, 12, 2)) - 0.5).to(dev)
        with torch.no_grad():
            y0 = m(x)
        y = torch.full((B, T, 21, 2), 7.0, device=dev)
        rc = lib.b2h_forward_fused(m._handle, ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(y.data_ptr()), B, T,
                                   4, 1.0, None, _lib.KERNELS[prec], None)
        torch.cuda.synchronize()
        d = (y != y0).view(B, T, 42).cpu().numpy()
        print(f"{prec} B={B} T={T}: mismatching elements {int(d.sum())} of {d.size}")
        if d.any():
            bs = np.nonzero(d.any(axis=(1, 2)))[0]
            b = bs[0]
            ts = np.nonzero(d[b].any(axis=1))[0]
            print("  first bad seq", b, "of", len(bs), "bad seqs; bad t:", ts[:40].tolist())
            for t in ts[:6]:
                print("   t", t, "bad channels", np.nonzero(d[b, t])[0].tolist())
                print("      fused", y.view(B, T, 42)[b, t, :6].cpu().numpy(), " plain", y0.view(B, T, 42)[b, t, :6].cpu().numpy())
