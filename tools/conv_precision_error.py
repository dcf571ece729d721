#!/usr/bin/env python3
"""GPU box: max-abs error of every ConvModel kernel against the oracle with fp64 accumulation, on
unit-scale and on pixel-scale inputs (where f16 operands would overflow their 11 bits but the
f16x3 split does not)."""
import os
import sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import hand_pose_sl_amd as hps, oracle
from conftest import load_golden
dev = torch.device("cuda:0")
rec = load_golden("cfg1_b1_t200")
g = torch.Generator().manual_seed(5)
x = torch.rand((64, 200, 12, 2), generator=g) - 0.5
ref64 = oracle.forward_from_state(x.numpy(), rec["state"], acc64=True)
for prec in ("f32_valu", "f32_mfma", "f16x3", "f16", "bf16"):
    m = hps.ConvModel(rec["C"], "ReLU", rec["pos_emb"], precision=prec)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in rec["state"].items()})
    m = m.to(dev).eval()
    with torch.no_grad():
        y = m(x.to(dev)).cpu().numpy()
    print(prec, "max|y-ref| = %.3e" % np.abs(y - ref64).max(), " max|y| = %.3f" % np.abs(y).max())
# raw pixel scale inputs (config 5 style, un-normalised): |x| ~ 1000
xp = (torch.rand((16, 200, 12, 2), generator=g) * 1280)
refp = oracle.forward_from_state(xp.numpy(), rec["state"], acc64=True)
for prec in ("f32_mfma", "f16x3"):
    m = hps.ConvModel(rec["C"], "ReLU", rec["pos_emb"], precision=prec)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in rec["state"].items()})
    m = m.to(dev).eval()
    with torch.no_grad():
        y = m(xp.to(dev)).cpu().numpy()
    print("pixel-scale input", prec, "max|y-ref| = %.3e" % np.abs(y - refp).max(), " max|y| = %.1f" % np.abs(y).max(), " rel = %.2e" % (np.abs(y - refp).max() / np.abs(refp).max()))
