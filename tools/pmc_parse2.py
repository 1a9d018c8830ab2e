"""Turn the passes of tools/pmc_passes.sh into the committed traffic summary.

    python tools/pmc_parse2.py <gpurun_out dir> <out_dir> <tag e.g. 1m_balanced>

Per pass: the counter rows of the calibration launch (dist_pairs_kernel, known byte count, the traversal's
8-lanes-x-16-B row gather) and of the LAST THREE traversal launches (the searchMany batches).  Writes
<out_dir>/pmc_counters_<tag>.csv (those rows) and <out_dir>/pmc_traffic_<tag>.json:
  reads   FETCH_SIZE scaled by the calibration ratio (gfx950 counts such streams at ~1/2, MI355X_MICROARCH.md HBM
          section), cross-checked by the request-size breakdown TCC_EA0_RDREQ{,_32B,_128B} scaled the same way
  writes  WRITE_SIZE (exact for 16-B-per-lane stores; the guide calls other widths uncalibrated) beside
          TCC_EA0_WRREQ x 64 B
  atomics TCC_ATOMIC (requests at L2: the visited test-and-set) and TCC_EA0_ATOMIC (forwarded to memory)
"""
import csv, glob, json, os, sys

src, out, tag = sys.argv[1:4]
info = None
for line in open(os.path.join(src, "pmc_fetch.log")):
    if line.startswith("{"):
        info = json.loads(line)
assert info, "pmc_run.py summary line missing"
keep, cal, trv = [], {}, {}
for name in ("fetch", "write", "rdreq", "hit"):
    files = glob.glob(os.path.join(src, f"pmc_{name}", "**", "*counter_collection.csv"), recursive=True)
    assert files, f"no counter csv for pass {name}"
    rows = list(csv.DictReader(open(files[0])))
    for cname in sorted({r["Counter_Name"] for r in rows}):
        c = [r for r in rows if r["Counter_Name"] == cname and "dist_pairs_kernel" in r["Kernel_Name"]]
        t = [r for r in rows if r["Counter_Name"] == cname and "traverse" in r["Kernel_Name"]][-3:]
        keep += c + t
        cal[cname] = sum(float(r["Counter_Value"]) for r in c)
        trv[cname] = sum(float(r["Counter_Value"]) for r in t) / max(len(t), 1)
        kname = t[-1]["Kernel_Name"] if t else ""
with open(os.path.join(out, f"pmc_counters_{tag}.csv"), "w", newline="") as f:
    w = csv.DictWriter(f, fieldnames=list(keep[0].keys()))
    w.writeheader()
    w.writerows(keep)
alg = info["traverse_algorithmic_bytes"]
known = info["calib_bytes"]
corr = known / (cal["FETCH_SIZE"] * 1024.0)
fetch = trv["FETCH_SIZE"] * 1024.0


def rd_bytes(d):  # request-size breakdown as the counters report it
    n, n32, n128 = d.get("TCC_EA0_RDREQ_sum", 0.0), d.get("TCC_EA0_RDREQ_32B_sum", 0.0), d.get("TCC_EA0_RDREQ_128B_sum", 0.0)
    return 32 * n32 + 128 * n128 + 64 * (n - n32 - n128)


rd_corr = known / rd_bytes(cal) if rd_bytes(cal) else None
reads = fetch * corr
writes = trv.get("WRITE_SIZE", 0.0) * 1024.0
doc = {
    "workload": f"{info['n']}x384 balanced, 1000 queries, ef=annK=1200 (tools/pmc_run.py under tools/pmc_passes.sh)",
    "calibration": {"kernel": "dist_pairs_kernel", "known_bytes": known, "fetch_size_bytes": cal["FETCH_SIZE"] * 1024.0,
                    "correction": corr, "rdreq_breakdown_bytes": rd_bytes(cal), "rdreq_correction": rd_corr,
                    "note": "same 8-lanes-x-16-B row gather as the traversal; gfx950 FETCH_SIZE reads ~1/2 of such streams"},
    "traverse": {
        "kernel": "traverse_duo_kernel" if "traverse_duo_kernel" in kname else "traverse_kernel",
        "launches": 3,
        "algorithmic_bytes_per_launch": alg,
        "fetch_size_bytes_per_launch": fetch,
        "read_bytes_per_launch": reads,
        "read_bytes_per_launch_from_rdreq": rd_bytes(trv) * rd_corr if rd_corr else None,
        "write_size_bytes_per_launch": writes,
        "ea_wrreq_per_launch": trv.get("TCC_EA0_WRREQ_sum"), "ea_wrreq_64B_per_launch": trv.get("TCC_EA0_WRREQ_64B_sum"),
        "l2_atomic_requests_per_launch": trv.get("TCC_ATOMIC_sum"),
        "ea_atomic_requests_per_launch": trv.get("TCC_EA0_ATOMIC_sum"),
        "l2_hit": trv.get("TCC_HIT_sum"), "l2_miss": trv.get("TCC_MISS_sum"),
        "l2_writebacks": trv.get("TCC_NORMAL_WRITEBACK_sum"),
        "corrected_bytes_per_launch": reads + writes,
        "traffic_over_algorithmic": (reads + writes) / alg,
        "reads_over_algorithmic": reads / alg,
    },
    "what_is_counted": "reads = every L2 fill from the memory side (rows, neighbour lists AND the visited-bitmap lines the "
                       "atomicOr touches); writes = dirty write-backs (visited-bitmap lines, result lists)",
}
json.dump(doc, open(os.path.join(out, f"pmc_traffic_{tag}.json"), "w"), indent=1)
print(json.dumps(doc["traverse"], indent=1))
