"""Per-kernel statistics from a rocprofv3 results database (the --stats table, as CSV):
   python tools/rocprof_db_stats.py <run_results.db> <out.csv> [kernel substring for the last-N average] [N]"""
import csv, sqlite3, statistics, sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end from kernels").fetchall()
by = {}
for n, s, e in rows:
    by.setdefault(n, []).append(e - s)
tot = sum(sum(v) for v in by.values())
out = sorted(((n, len(v), sum(v), sum(v) / len(v), 100 * sum(v) / tot, min(v), max(v), statistics.pstdev(v)) for n, v in by.items()),
             key=lambda r: -r[2])
with open(sys.argv[2], "w", newline="") as f:
    w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
    for r in out:
        w.writerow([r[0], r[1], r[2], round(r[3], 2), round(r[4], 2), r[5], r[6], round(r[7], 2)])
for r in out[:5]:
    print(f"{r[0][:70]:70s} calls {r[1]:4d} avg {r[3] / 1e6:8.3f} ms  min {r[5] / 1e6:.3f}  max {r[6] / 1e6:.3f}")
if len(sys.argv) > 3:
    n = int(sys.argv[4]) if len(sys.argv) > 4 else 20
    tv = [e - s for nm, s, e in sorted(rows, key=lambda r: r[1]) if sys.argv[3] in nm]
    print(f"last {n} launches of '{sys.argv[3]}': avg {sum(tv[-n:]) / n / 1e6:.4f} ms")
