"""Per-launch table of one kernel from a rocprofv3 --kernel-trace CSV, plus the average of its last N launches
(the bench's timed region): python tools/kernel_trace_table.py <kernel_trace.csv> <kernel substring> <N> <out.csv>"""
import csv, sys

src, pat, n, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
rows = [r for r in csv.DictReader(open(src)) if pat in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
with open(out, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["launch", "start_us", "duration_us", "timed_region", "vgpr", "lds_bytes", "grid", "kernel"])
    for i, r in enumerate(rows):
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        w.writerow([i, round((s - t0) / 1e3, 1), round((e - s) / 1e3, 2), int(i >= len(rows) - n), r.get("VGPR_Count", ""),
                    r.get("LDS_Block_Size", ""), r.get("Grid_Size", r.get("Grid_Size_X", "")), r["Kernel_Name"][:60]])
dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows]
print(f"{len(rows)} launches of '{pat}'; last {n}: avg {sum(dur[-n:]) / n:.4f} ms, min {min(dur[-n:]):.4f}, max {max(dur[-n:]):.4f}")
