#!/usr/bin/env python3
"""Per-kernel effective clock and MFMA-pipe occupancy from a rocprofv3 pass with
`--kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES` (counter_collection.csv).
clock = GRBM_GUI_ACTIVE / 8 XCDs / duration; busy = MFMA_BUSY_CYCLES / (1024 SIMDs x cycles)."""
import csv, collections, sys
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    if "b2h" not in r["Kernel_Name"]: continue
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    dur = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    if dur < 200000: continue  # only launches long enough for the clock estimate to mean something
    acc[(k, r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
    acc[(k, r["Dispatch_Id"])]["dur_ns"] = dur
per = collections.defaultdict(list)
for (k, _), c in acc.items():
    if "GRBM_GUI_ACTIVE" in c and "SQ_VALU_MFMA_BUSY_CYCLES" in c:
        cyc = c["GRBM_GUI_ACTIVE"] / 8
        per[k].append((cyc / c["dur_ns"], c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * cyc), c["dur_ns"] / 1e6))
for k, v in per.items():
    n = len(v)
    print(f"{k:40s} n={n:4d} clock {sum(x[0] for x in v)/n:.2f} GHz  MFMA pipe busy {100*sum(x[1] for x in v)/n:.0f} %  avg {sum(x[2] for x in v)/n:.3f} ms")
