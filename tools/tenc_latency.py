#!/usr/bin/env python3
"""GPU box: small-batch latency of the TransformerEnc path (9 kernel launches per forward), as
back-to-back stream launches and as one HIP-graph replay (the C ABI neither allocates nor
synchronises, so the whole forward can be captured)."""
import ctypes, json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hand_pose_sl_amd as hps
from hand_pose_sl_amd import _lib
from hand_pose_sl_amd.transformer_enc import TENC_KERNELS

dev = torch.device("cuda:0")
out = []
for prec in ("fp32", "f16x3"):
    torch.manual_seed(0)
    m = hps.TransformerEnc(24, 4, 128, 42, 4, precision=prec).to(dev).eval()
    for B in (1, 8, 64):
        T = 100
        x = (torch.rand((B, T, 12, 2), device=dev) - 0.5)
        with torch.no_grad():
            y_ref = m(x)                                   # packs weights, sizes the workspace
        lib = m._ensure_handle()
        y = torch.empty_like(y_ref)
        ws = torch.empty(lib.b2h_tenc_workspace_bytes(m._handle, B, T), dtype=torch.uint8, device=dev)
        lib.b2h_tenc_set_kernel(m._handle, TENC_KERNELS[prec])

        def launch(stream):
            _lib.check(lib.b2h_tenc_forward(m._handle, ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(y.data_ptr()), B, T,
                                            ctypes.c_void_p(ws.data_ptr()), ws.numel(), ctypes.c_void_p(stream.cuda_stream)))
        s = torch.cuda.Stream(dev)
        with torch.cuda.stream(s):
            for _ in range(20):
                launch(s)
            s.synchronize()
            n = 2000
            t0 = time.perf_counter()
            for _ in range(n):
                launch(s)
            s.synchronize()
            us_stream = (time.perf_counter() - t0) / n * 1e6
            assert torch.equal(y, y_ref)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                launch(s)
            for _ in range(20):
                g.replay()
            s.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                g.replay()
            s.synchronize()
            us_graph = (time.perf_counter() - t0) / n * 1e6
            y.zero_(); g.replay(); s.synchronize()
            assert torch.equal(y, y_ref)
        out.append({"precision": prec, "B": B, "T": T, "us_per_forward_stream": us_stream, "us_per_forward_graph": us_graph})
print(json.dumps(out, indent=1))
