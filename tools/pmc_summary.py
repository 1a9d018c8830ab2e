"""Average the counters of one kernel from a rocprofv3 counter_collection.csv: pmc_summary.py <csv> <kernel substring>"""
import collections, csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if sys.argv[2] in r["Kernel_Name"]]
agg = collections.defaultdict(list)
for r in rows:
    agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()):
    print(f"{k:28s} n={len(v):3d} mean={sum(v) / len(v):.4g}")
if rows:
    r = rows[0]
    print("VGPR", r["VGPR_Count"], "AGPR", r["Accum_VGPR_Count"], "SGPR", r["SGPR_Count"], "LDS", r["LDS_Block_Size"], "scratch", r["Scratch_Size"])
