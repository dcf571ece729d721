#!/usr/bin/env python3
"""GPU box: randomized sweep of the TransformerEnc path against the numpy oracle -- random batch,
length (1..100), layer count, weight scale, kernel (fp32 / f16x3) and, for a third of the cases, the
fused item transforms with random flags and ragged tail masks; plus the masked-L1 metric
and the target transform (bit-exact) on random shapes.    python tools/stress_tenc.py [seconds=120] [seed=0]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hand_pose_sl_amd as hps
import oracle

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = torch.device("cuda:0")
t_end = time.time() + budget
n, worst = 0, {"fp32": 0.0, "f16x3": 0.0, "l1": 0.0}
models = {}
while time.time() < t_end:
    prec = str(rng.choice(["fp32", "f16x3"]))
    L = int(rng.choice([1, 2, 4, 4, 4, 6]))
    key = (prec, L, n // 25)                                  # fresh weights every 25 cases
    if key not in models:
        torch.manual_seed(int(rng.integers(1 << 30)))
        m = hps.TransformerEnc(24, 4, 128, 42, L, precision=prec)
        with torch.no_grad():
            for name, p in m.named_parameters():
                if "norm" not in name:
                    p.mul_(float(rng.uniform(0.6, 1.8)))
        models = {key: (m.to(dev).eval(), {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()})}
    m, state = models[key]
    T = int(rng.integers(1, 101))
    B = int(rng.choice([1, 2, 3, rng.integers(4, 24)]))
    fused = rng.random() < 0.35
    if fused:   # raw pixels in, (masked) pixels out: transforms inside the chain kernel
        body = (rng.random((B, T, 12, 2), dtype=np.float32) * np.array([1280.0, 720.0], np.float32))
        nf = rng.integers(0, T + 1, B)
        dif, norm, den, mask = (bool(rng.random() < 0.7) for _ in range(4))
        with torch.no_grad():
            y = m.forward_fused(torch.from_numpy(body).to(dev), n_frames=nf if mask else None, dif_encoding=dif,
                                normalize=norm, denormalize=den, mask_tail=mask).cpu().numpy()
        x, _ = oracle.preprocess(body, None, dif_encoding=dif, normalize=norm)
        # (1) bit-identity with the unfused kernels on the pre-transformed rows, x factor and mask applied after
        with torch.no_grad():
            y_sep = m(torch.from_numpy(x).to(dev))
            if den:
                y_sep = y_sep * 1280.0
            y_sep = y_sep.cpu().numpy()
        y_sep = oracle.postprocess(y_sep, 1.0, nf if mask else None)
        if not np.array_equal(y, y_sep):
            print(f"FAIL case {n + 1}: fused != transform -> model -> x factor; prec={prec} L={L} B={B} T={T} "
                  f"dif={dif} norm={norm} den={den} mask={mask} max diff {np.abs(y - y_sep).max():.3e}")
            sys.exit(1)
        # (2) against the oracle.  Without the normalisation the rows are raw pixels (|x| up to 1280): a
        # random-weight transformer on such inputs has softmax logits in the thousands, near-ties flip
        # on fp32 summation order and single frames move by 1e-3 relative in ANY implementation, so the
        # oracle comparison is kept for normalised inputs only (the bit-identity above covers the rest).
        ref0 = oracle.transformer_forward(x, state)
        ref = oracle.postprocess(ref0, 1280.0 if den else 1.0, nf if mask else None) if norm else y
        scale = (1280.0 if den else 1.0) * max(1.0, float(np.abs(ref0).max()))
    else:
        x = ((rng.random((B, T, 12, 2), dtype=np.float32) - 0.5) * float(rng.choice([1.0, 1.0, 3.0])))
        with torch.no_grad():
            y = m(torch.from_numpy(x).to(dev)).cpu().numpy()
        ref = oracle.transformer_forward(x, state)
        scale = max(1.0, float(np.abs(ref).max()))
    err = float(np.abs(y - ref).max())
    # fp32: the tests' 2e-5 bar.  f16x3: the same bar holds on reference-scale weights (fixed tests);
    # with weights scaled up to 1.8x per tensor here its error reached 0.95 of it in 12 485 cases, so
    # the sweep gives it 2x headroom -- it is looking for structural errors, which are O(0.1)
    tol = (2e-5 if prec == "fp32" else 4e-5) * scale
    worst[prec] = max(worst[prec], err / tol)
    n += 1
    if not (err <= tol) or not np.isfinite(y).all():
        print(f"FAIL case {n}: prec={prec} L={L} B={B} T={T} fused={fused} err={err:.3e} tol={tol:.3e}")
        if fused:
            print(f"  flags: dif={dif} norm={norm} den={den} mask={mask}  n_frames={nf.tolist()}  |ref0|max={np.abs(ref0).max():.3f} |x|max={np.abs(x).max():.1f}")
        bad = np.argwhere(np.abs(y - ref) > tol)
        print(f"  {len(bad)} elements off; first {bad[:6].tolist()}; last {bad[-3:].tolist()}")
        b0, t0 = bad[0][0], bad[0][1]
        print(f"  y[{b0},{t0},0]={y[b0, t0, 0]} ref={ref[b0, t0, 0]}; frames off in seq {b0}: {sorted(set(bad[bad[:, 0] == b0][:, 1].tolist()))[:40]}")
        sys.exit(1)
    if n % 10 == 0:                                           # metric + target transform on the side
        Bm, Tm = int(rng.integers(1, 40)), int(rng.integers(1, 300))
        pred = rng.standard_normal((Bm, Tm, 21, 2)).astype(np.float32)
        tgt = rng.standard_normal((Bm, Tm, 21, 2)).astype(np.float32)
        nf = rng.integers(1, Tm + 1, Bm)
        got = float(hps.masked_pose_l1(torch.from_numpy(pred).to(dev), torch.from_numpy(tgt).to(dev), torch.from_numpy(nf).to(dev)))
        want = float(oracle.masked_l1(pred, tgt, nf)[0])
        e = abs(got - want) / max(1e-6, abs(want))
        worst["l1"] = max(worst["l1"], e / 2e-6)
        if e > 2e-6:
            print(f"FAIL masked_l1 B={Bm} T={Tm}: {got} vs {want}")
            sys.exit(1)
        body = (rng.random((Bm, Tm, 12, 2), dtype=np.float32) * np.array([1280.0, 720.0], np.float32))
        hand = (rng.random((Bm, Tm, 21, 2), dtype=np.float32) * np.array([1280.0, 720.0], np.float32))
        dif, norm = bool(rng.random() < 0.7), bool(rng.random() < 0.7)
        tt = hps.target_transform(torch.from_numpy(body).to(dev), torch.from_numpy(hand).to(dev), dif_encoding=dif,
                                  normalize=norm).cpu().numpy()
        _, want_t = oracle.preprocess(body, hand, dif_encoding=dif, normalize=norm)
        if not np.array_equal(tt, want_t):
            print(f"FAIL target_transform B={Bm} T={Tm} dif={dif} norm={norm}: max diff {np.abs(tt - want_t).max()}")
            sys.exit(1)
    if n % 100 == 0:
        print(f"{n} cases ok; worst err/tol {{{', '.join(f'{k}: {v:.2f}' for k, v in worst.items())}}}", flush=True)
print(f"PASS: {n} random cases in {budget:.0f} s; worst err/tol {{{', '.join(f'{k}: {v:.2f}' for k, v in worst.items())}}}")
