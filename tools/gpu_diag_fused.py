#!/usr/bin/env python3
"""Developer diagnostic (GPU box): fused pre/post-processing flags one by one vs the oracle."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hand_pose_sl_amd as hps
from hand_pose_sl_amd import _lib
import oracle
d = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "transforms_b6_t40.npz"))
state = {k.replace("_", ".", 1): d[k] for k in d.files if k.startswith("conv")}
dev = torch.device("cuda:0")
body = torch.from_numpy(d["body"]).to(dev)
nf = torch.from_numpy(d["n_frames"]).to(dev)
for prec in ("f32_mfma", "bf16"):
    m = hps.ConvModel(30, "ReLU", False, precision=prec)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()}); m = m.to(dev).eval()
    lib = m._ensure_handle()
    for flags in (1, 2, 3, 4, 8, 12, 7, 15):
        y = torch.full((6, 40, 21, 2), 7.0, device=dev)
        rc = lib.b2h_forward_fused(m._handle, ctypes.c_void_p(body.data_ptr()), ctypes.c_void_p(y.data_ptr()), 6, 40,
                                   flags, 1280.0, ctypes.c_void_p(nf.data_ptr()), _lib.KERNELS[prec], None)
        torch.cuda.synchronize()
        b = d["body"].copy()
        if flags & 1: b = b - b[:, :, 1:2]
        if flags & 2: b = b / np.float32(1280.0)
        ref = oracle.forward_from_state(b, state)
        if flags & 4: ref = ref * np.float32(1280.0)
        if flags & 8:
            for i, n in enumerate(d["n_frames"]): ref[i, n:] = 0
        e = np.abs(y.cpu().numpy() - ref)
        scale = np.abs(ref).max()
        w = np.unravel_index(np.argmax(e), e.shape)
        bad = sorted(set(zip(*[a.tolist() for a in np.nonzero(e > 0.02 * scale)[:2]])))
        print(f"{prec:9s} flags={flags:2d} rc={rc} max-abs {e.max():.3e} (scale {scale:.3g}) at {tuple(int(v) for v in w)}  bad (b,t): {bad[:10]}{'...' if len(bad)>10 else ''}")
