"""Sum rocprofv3 --pmc counter CSV per kernel name: python tools/pmc_sum.py <dir> [kernel substring]"""
import csv
import glob
import sys
from collections import defaultdict

d = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else ""
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    acc = defaultdict(lambda: defaultdict(float))
    nd = defaultdict(set)
    for r in csv.DictReader(open(f)):
        if sub in r["Kernel_Name"]:
            acc[r["Kernel_Name"][:50]][r["Counter_Name"]] += float(r["Counter_Value"])
            nd[r["Kernel_Name"][:50]].add(r["Dispatch_Id"])
    for kname, c in acc.items():
        print(kname, "dispatches", len(nd[kname]))
        for cn, v in sorted(c.items()):
            print(f"   {cn:32s} {v:.4g}")
