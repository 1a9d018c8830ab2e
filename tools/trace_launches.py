"""Prints the launches of the scan kernels from a rocprofv3 kernel-trace CSV (start-ordered): name, grid, duration."""
import csv
import glob
import sys

for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    for r in rows:
        if "scan" in r["Kernel_Name"]:
            d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            print(f'{r["Kernel_Name"][:60]:60s} grid {r.get("Grid_Size_X", r.get("Grid_Size", "?")):>8s} wg {r.get("Workgroup_Size_X", "?"):>4s} {d:9.1f} us')
