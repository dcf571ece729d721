#!/usr/bin/env python3
"""Development: torch-port CPU throughput vs thread count on this host."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
torch.manual_seed(0)
import torch.nn as nn
state = {}
for i, (ci, co) in enumerate([(24, 30), (30, 30), (30, 30), (30, 42)], 1):
    c = nn.Conv1d(ci, co, 5, padding=2); state[f"conv{i}.weight"] = c.weight.detach(); state[f"conv{i}.bias"] = c.bias.detach()
port = oracle.TorchPort(state)
for B in (64, 256, 2048):
    x = torch.rand((B, 200, 12, 2)) - 0.5
    for nt in (1, 4, 8, 16, 32, 64, 128):
        torch.set_num_threads(nt)
        port(x); t0 = time.perf_counter(); n = 0
        while time.perf_counter() - t0 < 1.0:
            port(x); n += 1
        el = time.perf_counter() - t0
        print(f"B={B:5d} threads={nt:3d}: {n*B*200/el/1e6:8.2f} M frames/s")
