"""Per-launch table of one kernel from a rocprofv3 --kernel-trace CSV, with the launches attributed to the phases of a
bench.py run, and the average launch duration per phase:

    python tools/kernel_trace_table.py <kernel_trace.csv> <kernel substring> <phases> <out.csv>

<phases> = name:count,... in launch order, '*' = the rest, e.g. for `bench.py --legs none --no-pcie` (8 batches, 3 warm-up,
20 steps): reference:8,warmup:3,timed:20,solo:16,resident:20,sustained:*"""
import csv, sys

src, pat, spec, out = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4]
rows = [r for r in csv.DictReader(open(src)) if pat in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
phase, left = [], len(rows)
for item in spec.split(","):
    name, cnt = item.split(":")
    n = left if cnt == "*" else min(int(cnt), left)
    phase += [name] * n
    left -= n
phase += ["after"] * left
t0 = int(rows[0]["Start_Timestamp"])
acc = {}
with open(out, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["launch", "phase", "start_us", "duration_us", "vgpr", "lds_bytes", "grid", "kernel"])
    for i, r in enumerate(rows):
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        acc.setdefault(phase[i], []).append((e - s) / 1e6)
        w.writerow([i, phase[i], round((s - t0) / 1e3, 1), round((e - s) / 1e3, 2), r.get("VGPR_Count", ""),
                    r.get("LDS_Block_Size", ""), r.get("Grid_Size", r.get("Grid_Size_X", "")), r["Kernel_Name"][:60]])
print(f"{len(rows)} launches of '{pat}'")
for name, d in acc.items():
    print(f"  {name:10s} {len(d):4d} launches: avg {sum(d) / len(d):.4f} ms, min {min(d):.4f}, max {max(d):.4f}")
