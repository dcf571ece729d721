#!/usr/bin/env python3
"""Register / scratch / occupancy table of every kernel in libb2h.so, from hipcc's own
-Rpass-analysis=kernel-resource-usage remarks (no GPU needed).    python tools/kernel_resources.py [filter]"""
import os, re, subprocess, sys, tempfile
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "hand_pose_sl_amd", "csrc", "b2h_api.hip")
with tempfile.TemporaryDirectory() as td:
    r = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-I" + os.path.join(root, "include"),
                        "-I" + os.path.dirname(src), "-c", src, "-o", os.path.join(td, "b.o"), "-Rpass-analysis=kernel-resource-usage"],
                       capture_output=True, text=True, cwd=td)
rows, cur = [], None
for line in r.stderr.splitlines():
    m = re.search(r"remark: [^:]*:\d+:\d+:\s+(.*?) \[-Rpass", line) or re.search(r"remark:\s+(.*?) \[-Rpass", line)
    if not m:
        m = re.search(r":\d+:\d+:\s+(.*?) \[-Rpass", line)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = {"name": t.split(":", 1)[1].strip()}
        rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
flt = sys.argv[1] if len(sys.argv) > 1 else ""
def demangle(n):
    try:
        return subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip().split("(")[0]
    except OSError:
        return n
print(f"{'kernel':58s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'scratch':>8s} {'vspill':>6s} {'sspill':>6s} {'occ':>4s} {'LDS':>7s}")
for r_ in rows:
    n = demangle(r_["name"])
    if flt and flt not in n:
        continue
    print(f"{n[-58:]:58s} {r_.get('VGPRs','?'):>5s} {r_.get('AGPRs','?'):>5s} {r_.get('TotalSGPRs','?'):>5s} {r_.get('ScratchSize [bytes/lane]','?'):>8s} "
          f"{r_.get('VGPRs Spill','?'):>6s} {r_.get('SGPRs Spill','?'):>6s} {r_.get('Occupancy [waves/SIMD]','?'):>4s} {r_.get('LDS Size [bytes/block]','?'):>7s}")
