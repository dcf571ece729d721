#!/usr/bin/env python3
"""GPU box: throughput of the TransformerEnc path with HIP events, the CPU
port on the host cores beside it, and max-abs error of the same sample."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hand_pose_sl_amd as hps
import oracle

dev = torch.device("cuda:0")
torch.manual_seed(0)
PREC = next((a.split("=")[1] for a in sys.argv if a.startswith("--precision=")), "fp32")
m = hps.TransformerEnc(24, 4, 128, 42, 4, precision=PREC).to(dev).eval()
state = {k: v.detach().cpu() for k, v in m.state_dict().items()}
FLOP = lambda T: 2 * (24 * 128 + 4 * (128 * 384 + 3 * 128 * 128) + 128 * 42) + 4 * 4 * T * 32 * 2 * 2
QUICK = "--quick" in sys.argv   # one large batch, no CPU leg (tools/ablate_tenc.sh)
out = {"device": torch.cuda.get_device_name(0), "precision": PREC, "runs": []}
with torch.no_grad():
    for (B, T) in (((32768, 100),) if QUICK else ((64, 100), (4096, 100), (32768, 100), (32768, 50))):
        x = (torch.rand((B, T, 12, 2)) - 0.5).to(dev)
        for _ in range(3): y = m(x)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 20
        e0.record()
        for _ in range(n): y = m(x)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        fps = B * T / (ms * 1e-3)
        out["runs"].append({"B": B, "T": T, "ms": ms, "frames_per_s": fps, "tflops_fp32": fps * FLOP(T) / 1e12,
                            "frac_of_fp32_mfma_peak_157": fps * FLOP(T) / 157.3e12})
    if QUICK:
        print(json.dumps(out)); sys.exit(0)
    # CPU port + parity on a bounded sample
    port = oracle.TencTorchPort({k: v.numpy() for k, v in state.items()})
    xs = torch.rand((256, 100, 12, 2)) - 0.5
    best = (0, 1)
    for nt in (1, 8, 16):
        torch.set_num_threads(nt); port(xs); k, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < 1.0: port(xs); k += 1
        r = k / (time.perf_counter() - t0)
        if r > best[0]: best = (r, nt)
    torch.set_num_threads(best[1]); k, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < 8.0: yc = port(xs); k += 1
    el = time.perf_counter() - t0
    yg = m(xs.to(dev)).cpu()
    out["cpu_baseline"] = {"value": k * 256 * 100 / el, "unit": "frames/s", "cores": best[1], "kind": "port",
                           "sample": f"(256,100,12,2) x {k} passes in {el:.1f} s (oracle/tenc_torch_port.py)",
                           "gpu_max_abs_err_on_sample": float((yg - yc).abs().max())}
print(json.dumps(out, indent=1))
