"""Turn a rocprofv3 --pmc FETCH_SIZE pass of tools/pmc_run.py into the committed summary.

    python tools/pmc_parse.py <counter_collection.csv> <pmc_run stdout> <out_dir>

Writes <out_dir>/pmc_fetch_size_1m_balanced.csv (the dist_pairs + traversal rows of the pass) and
<out_dir>/pmc_traffic_1m_balanced.json (per-launch traffic after the calibration described in
tools/pmc_run.py)."""
import csv, json, sys

src, runlog, out = sys.argv[1:4]
info = None
for line in open(runlog):
    if line.startswith("{"):
        info = json.loads(line)
rows = [r for r in csv.DictReader(open(src)) if r["Counter_Name"] == "FETCH_SIZE"]
cal = [r for r in rows if "dist_pairs_kernel" in r["Kernel_Name"]]
trv = [r for r in rows if "traverse" in r["Kernel_Name"]]
assert cal and trv and info
with open(f"{out}/pmc_fetch_size_1m_balanced.csv", "w", newline="") as f:
    w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
    w.writeheader()
    for r in cal + trv:
        w.writerow(r)
cal_bytes = sum(float(r["Counter_Value"]) for r in cal) * 1024.0  # FETCH_SIZE is in KB
corr = info["calib_bytes"] / cal_bytes
# the last three traversal launches are the searchMany batches (earlier ones belong to the build's warmup, if any)
per = [float(r["Counter_Value"]) * 1024.0 for r in trv[-3:]]
fetch = sum(per) / len(per)
kname = trv[-1]["Kernel_Name"]
kernel = "traverse_duo_kernel" if "traverse_duo_kernel" in kname else "traverse_kernel"
alg = info["traverse_algorithmic_bytes"]
doc = {
    "workload": f"{info['n']}x384 balanced, 1000 queries, ef=annK=1200",
    "counter": "FETCH_SIZE (KB)",
    "calibration": {"kernel": "dist_pairs_kernel", "known_bytes": info["calib_bytes"], "fetch_size_bytes": cal_bytes,
                    "correction": corr,
                    "note": "same 8-lanes-x-16-B row gather pattern as the traversal; gfx950 FETCH_SIZE reads ~1/2 of "
                            "such streams (MI355X_MICROARCH.md HBM section)"},
    "traverse": {"kernel": kernel, "fetch_size_bytes_per_launch": fetch, "corrected_bytes_per_launch": fetch * corr,
                 "algorithmic_bytes_per_launch": alg, "traffic_over_algorithmic": fetch * corr / alg,
                 "launches": len(per)},
    "not_included": "visited-bitmap atomics (executed in L2, not in FETCH_SIZE) and the 9.6 MB of result writes",
}
json.dump(doc, open(f"{out}/pmc_traffic_1m_balanced.json", "w"), indent=1)
print(json.dumps(doc["traverse"]))
