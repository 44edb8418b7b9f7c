"""Aggregate a rocprofv3 --pmc counter_collection.csv: mean counter value per kernel (substring filter) per counter.
usage: python scripts/pmc_report.py <dir> [kernel-substring]"""
import sys, csv, glob, collections
d = sys.argv[1]; filt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if filt and filt not in k: continue
        acc[k[:90]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        print(f"   {c:36s} n={len(v):4d} mean={sum(v)/len(v):.4g} last={v[-1]:.4g}")
