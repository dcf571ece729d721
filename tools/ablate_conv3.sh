#!/bin/bash
# GPU box: timing-only builds of the f16x3 conv kernel (csrc/dev/b2h_dev.h): what the per-chunk
# weight reloads and the input loads cost.  Leaves the shipped build in place at the end.
#   bash tools/ablate_conv3.sh > gpurun_out/ablate_conv3.txt
set -e
cd "$(dirname "$0")/.."
# whatever happens below, the shipped (B2H_ABLATE = 0) build is what is left in place
trap 'B2H_ABLATE= python -m hand_pose_sl_amd.build --force > /dev/null 2>&1' EXIT
export B2H_ALLOW_ABLATE=1   # hand_pose_sl_amd._lib refuses ablation builds without it
for a in ${ABLATE_BITS:-0 64 128 192}; do
  B2H_ABLATE=$a python -m hand_pose_sl_amd.build --force > /dev/null 2>&1
  python - <<PY
import torch, hand_pose_sl_amd as hps
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = hps.ConvModel(30, "ReLU", False, precision="f16x3").to(dev).eval()
x = torch.rand((65536, 200, 12, 2), device=dev) - 0.5
y = torch.empty((65536, 200, 21, 2), device=dev)
m.time_forward(x, y, 5)
ms = min(m.time_forward(x, y, 20) for _ in range(3))
print(f"B2H_ABLATE=$a  {m.kernel_name()} 65536x200: {ms:.4f} ms  {65536*200/ms/1e6:.2f} G frames/s", flush=True)
PY
done
