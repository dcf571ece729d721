#!/usr/bin/env python3
"""GPU box: the BASELINE.json configurations that are not the bench line -- call latency of the
small cases (configs 1-3 literal shapes), config 4's stream on one GPU, and config 5's
per-stage timings (item transforms -> ConvModel -> x1280 + mask) unfused vs fused."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hand_pose_sl_amd as hps

dev = torch.device("cuda:0")
out = {"device": torch.cuda.get_device_name(0)}


def ev_time(fn, n=200, warm=20):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3   # us


def wall_time(fn, n=200, warm=20):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


torch.manual_seed(0)
models = {p: hps.ConvModel(30, "ReLU", False, precision=p).to(dev).eval() for p in ("fp32", "f16x3", "bf16", "f16")}
lat = {}
with torch.no_grad():
    for (B, T) in ((1, 1), (1, 200), (64, 200), (256, 200), (2000, 200)):
        x = (torch.rand((B, T, 12, 2)) - 0.5).to(dev)
        y = torch.empty((B, T, 21, 2), device=dev)
        for p, m in models.items():
            us_mod = wall_time(lambda: m(x))                       # nn.Module call incl. output allocation
            ms_raw = m.time_forward(x, y, 200) * 1e3                # back-to-back C-ABI launches, HIP events
            lat[f"B{B}_T{T}_{p}"] = {"module_call_us": round(us_mod, 2), "kernel_us": round(ms_raw, 2),
                                     "frames_per_s_kernel": B * T / (ms_raw * 1e-6)}
out["call_latency"] = lat

# config 5: stages, raw pixels (B,T) with ragged n_frames
B, T = 4096, 200
g = torch.Generator().manual_seed(5)
body = (torch.rand((B, T, 12, 2), generator=g) * torch.tensor([1280.0, 720.0])).to(dev)
nf = torch.randint(50, 201, (B,), generator=g).to(dev)
m = models["bf16"]
stages = {}
with torch.no_grad():
    def s1():
        b = body - body[:, :, 1:2]          # ChestDifference
        return b / 1280.0                   # NormalizeFixedFactor
    xin = s1()
    yb = torch.empty((B, T, 21, 2), device=dev)
    def s2(): return m(xin)
    pred = s2()
    ar = torch.arange(T, device=dev)[None, :, None, None]
    def s3():
        p = pred * 1280.0
        return torch.where(ar < nf[:, None, None, None], p, torch.zeros((), device=dev))
    stages["S1_transforms_torch_elementwise_us"] = round(ev_time(s1, 100, 10), 1)
    stages["S2_convmodel_kernel_us"] = round(m.time_forward(xin, yb, 100) * 1e3, 1)
    stages["S3_denorm_mask_torch_elementwise_us"] = round(ev_time(s3, 100, 10), 1)
    stages["unfused_total_us"] = round(ev_time(lambda: (s1(), s2(), s3()), 100, 10), 1)
    stages["fused_one_kernel_us"] = round(ev_time(lambda: m.forward_fused(body, n_frames=nf, mask_tail=True), 100, 10), 1)
    a = m.forward_fused(body, n_frames=nf, mask_tail=True)
    ref = torch.where(ar < nf[:, None, None, None], m(xin) * 1280.0, torch.zeros((), device=dev))
    stages["fused_vs_unfused_max_abs_px"] = float((a - ref).abs().max())
stages["shape"] = [B, T]
out["config5_stages"] = stages
print(json.dumps(out, indent=1))
