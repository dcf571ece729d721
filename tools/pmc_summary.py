import csv, sys, collections, glob, os
root = sys.argv[1]
for d in sorted(glob.glob(os.path.join(root, "pmc_*"))):
    if not os.path.isdir(d): continue
    f = glob.glob(os.path.join(d, "*counter_collection.csv"))
    if not f: continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(f[0])):
        k = row["Kernel_Name"]
        if "b2h" not in k: continue
        acc[k.split("(")[0][:40]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, cs in acc.items():
        for c, v in cs.items():
            print(f"{os.path.basename(d):10s} {k:40s} {c:28s} n={len(v):3d} mean={sum(v)/len(v):.6g}")
