"""Aggregate rocprofv3 --pmc counter_collection CSVs per kernel: python scripts/pmc_summary.py DIR [kernel-substring]"""
import csv, glob, sys, collections
root = sys.argv[1]; want = sys.argv[2] if len(sys.argv) > 2 else ""
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set)
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if want and want not in k:
            continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])].add(r["Dispatch_Id"])
for k in sorted(agg):
    print(k)
    for c in sorted(agg[k]):
        n = len(cnt[(k, c)])
        print(f"   {c:38s} total {agg[k][c]:.6g}  dispatches {n}  per-dispatch {agg[k][c]/max(1,n):.6g}")
