#!/usr/bin/env python3
"""Developer diagnostic (GPU box): per-kernel max-abs error vs the oracle with the
location of the worst element.  Not part of the product or the test suite."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hand_pose_sl_amd as hps  # noqa: E402
import oracle  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    print(torch.cuda.get_device_name(0), torch.version.hip)
    shapes = [(1, 1), (2, 16), (3, 33), (2, 200), (2, 208), (2, 209), (1, 600), (64, 200)]
    for pos_emb in (False, True):
        for prec in ("f32_valu", "f32_mfma", "bf16", "f16"):
            torch.manual_seed(0)
            m = hps.ConvModel(30, "ReLU", pos_emb, precision=prec).to(dev).eval()
            state = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}
            for B, T in (shapes if not pos_emb else [(2, 100)]):
                x = torch.rand((B, T, 12, 2), generator=torch.Generator().manual_seed(B * 1000 + T)) - 0.5
                try:
                    with torch.no_grad():
                        y = m(x.to(dev)).cpu().numpy()
                except Exception as e:  # noqa: BLE001
                    print(f"{prec:9s} pe={int(pos_emb)} B={B:3d} T={T:4d}  EXC {e}")
                    continue
                ref = oracle.forward_from_state(x.numpy(), state, pos_emb=pos_emb)
                d = np.abs(y - ref)
                w = np.unravel_index(np.argmax(np.nan_to_num(d, nan=1e9)), d.shape)
                extra = ""
                if prec in ("bf16", "f16"):
                    ym = oracle.forward_from_state(x.numpy(), state, pos_emb=pos_emb, mode=prec)
                    extra = f" vs-model {np.abs(y - ym).max():.2e}"
                tb = d.reshape(B, T, 42).max(axis=(0, 2))
                bad_t = np.nonzero(tb > 2e-3)[0]
                print(f"{prec:9s} pe={int(pos_emb)} B={B:3d} T={T:4d}  max-abs {d.max():.2e} at {w} nan={int(np.isnan(y).sum())}"
                      f"{extra}  bad_t={bad_t[:12].tolist()}{'...' if len(bad_t) > 12 else ''}")
            del m


if __name__ == "__main__":
    main()
