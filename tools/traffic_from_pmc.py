#!/usr/bin/env python3
"""Turn the FETCH_SIZE / WRITE_SIZE passes of tools/profile.sh into HBM bytes per launch.

gfx950 corrections per MI355X_MICROARCH.md (HBM section): both counters are in KiB;
FETCH_SIZE reports exactly half the bytes of a wide coalesced streaming read (16 B/lane),
so it is doubled; WRITE_SIZE is exact for 16-B-per-lane streaming stores.

    python tools/traffic_from_pmc.py gpurun_out/<tag> [kernel-substring] [seqs frames precision] > profiles/<tag>/traffic.json

With the workload given, the record is stamped with it and with the sha256 of the kernel sources the
passes ran on (bench.sources_sha256): bench.py reports the figure as `roofline.traffic` only while
both still match, so an edit of the kernel leaves it null instead of silently stale.
"""
import csv, glob, json, os, sys

root = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else "b2h_fwd"


def mean_counter(sub, name):
    f = glob.glob(os.path.join(root, sub, "*counter_collection.csv"))
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f[0]))
            if want in r["Kernel_Name"] and r["Counter_Name"] == name]
    return sum(vals) / len(vals), len(vals)


fetch_kib, n1 = mean_counter("pmc_fetch", "FETCH_SIZE")
write_kib, n2 = mean_counter("pmc_write", "WRITE_SIZE")
out = {"kernel": want, "launches": min(n1, n2),
       "FETCH_SIZE_KiB_raw": fetch_kib, "WRITE_SIZE_KiB_raw": write_kib,
       "read_bytes": 2 * fetch_kib * 1024, "write_bytes": write_kib * 1024,
       "traffic_bytes": 2 * fetch_kib * 1024 + write_kib * 1024,
       "correction": "FETCH_SIZE x2 (gfx950 wide coalesced reads count 128-B requests as 64 B); WRITE_SIZE exact"}
if len(sys.argv) > 5:
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    seqs, frames, prec = int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
    out.update(seqs_per_gpu=seqs, frames_per_seq=frames, precision=prec,
               algorithmic_bytes=seqs * frames * bench.BYTES_PER_FRAME,
               sources_sha256=bench.sources_sha256(), sources=bench.TRAFFIC_SOURCES,
               command="tools/profile.sh: rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python3 bench.py")
print(json.dumps(out, indent=1))
