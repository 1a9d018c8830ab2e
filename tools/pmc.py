"""rocprofv3 --pmc passes of tools/pmc_run.py (traversal kernel) -> numbers.  One tool, three uses:

    python tools/pmc.py traffic <gpurun_out dir> <out_dir> <tag e.g. 1m_balanced>
        the four TCC passes of `tools/pmc_passes.sh traffic` -> <out_dir>/pmc_counters_<tag>.csv and
        <out_dir>/pmc_traffic_<tag>.json:
          reads   FETCH_SIZE scaled by the calibration ratio (gfx950 counts 8-lanes-x-16-B gathers at ~1/2,
                  MI355X_MICROARCH.md HBM section), cross-checked by TCC_EA0_RDREQ{,_32B,_128B} scaled the same way
          writes  WRITE_SIZE beside TCC_EA0_WRREQ x 64 B
          atomics TCC_ATOMIC (requests at L2) and TCC_EA0_ATOMIC (executed at the memory side)
    python tools/pmc.py sum <dir> [kernel substring]
        every counter of every pass under <dir>, averaged per launch, per kernel (instruction mix, wait cycles ...),
        with the kernel's VGPR/LDS/scratch figures
"""
import collections
import csv
import glob
import json
import os
import sys


def csvs(d):
    return sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True))


def kernel_of(name):
    for k in ("traverse_lean_kernel", "traverse_duo_kernel", "traverse_kernel"):
        if k in name:
            return k
    return name.split("(")[0][:60]


def cmd_sum(d, sub=""):
    for f in csvs(d):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        res = {}
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][:70]
            if sub in k:
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
                res[k] = (r.get("VGPR_Count"), r.get("Accum_VGPR_Count"), r.get("SGPR_Count"), r.get("LDS_Block_Size"), r.get("Scratch_Size"))
        print(os.path.relpath(f, d))
        for k, c in acc.items():
            print(" ", k, "VGPR/AGPR/SGPR/LDS/scratch", *res[k])
            for cn, v in sorted(c.items()):
                print(f"     {cn:28s} launches={len(v):3d} per_launch={sum(v) / len(v):.6g}")


def cmd_traffic(src, out, tag):
    info = None
    for line in open(os.path.join(src, "pmc_fetch.log")):
        if line.startswith("{"):
            info = json.loads(line)
    assert info, "pmc_run.py summary line missing"
    keep, cal, trv, kname, cal_ns = [], {}, {}, "", []
    for name in ("fetch", "write", "rdreq", "hit"):
        files = csvs(os.path.join(src, f"pmc_{name}"))
        assert files, f"no counter csv for pass {name}"
        rows = list(csv.DictReader(open(files[0])))
        # the calibration launch (known byte count) and the LAST THREE lean/duo/one-wave traversal launches that did
        # work (the fall-back launch of the lean path is skipped: it exits at once)
        main = [r for r in rows if "traverse" in r["Kernel_Name"]]
        if any("traverse_lean_kernel" in r["Kernel_Name"] for r in main):
            main = [r for r in main if "traverse_lean_kernel" in r["Kernel_Name"]]
        for cname in sorted({r["Counter_Name"] for r in rows}):
            c = [r for r in rows if r["Counter_Name"] == cname and "dist_pairs_kernel" in r["Kernel_Name"]]
            t = [r for r in main if r["Counter_Name"] == cname][-3:]
            keep += c + t
            cal_ns += [float(r["End_Timestamp"]) - float(r["Start_Timestamp"]) for r in c]
            cal[cname] = sum(float(r["Counter_Value"]) for r in c)
            trv[cname] = sum(float(r["Counter_Value"]) for r in t) / max(len(t), 1)
            kname = t[-1]["Kernel_Name"] if t else kname
    with open(os.path.join(out, f"pmc_counters_{tag}.csv"), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(keep[0].keys()))
        w.writeheader()
        w.writerows(keep)
    alg, known = info["traverse_algorithmic_bytes"], info["calib_bytes"]
    corr = known / (cal["FETCH_SIZE"] * 1024.0)
    fetch = trv["FETCH_SIZE"] * 1024.0

    def rd_bytes(d):  # request-size breakdown as the counters report it
        n, n32, n128 = d.get("TCC_EA0_RDREQ_sum", 0.0), d.get("TCC_EA0_RDREQ_32B_sum", 0.0), d.get("TCC_EA0_RDREQ_128B_sum", 0.0)
        return 32 * n32 + 128 * n128 + 64 * (n - n32 - n128)

    rd_corr = known / rd_bytes(cal) if rd_bytes(cal) else None
    reads, writes = fetch * corr, trv.get("WRITE_SIZE", 0.0) * 1024.0
    doc = {
        "workload": f"{info['n']}x384 balanced{' clustered' if info.get('clustered') else ''}, 1000 queries, ef=annK=1200 "
                    "(tools/pmc_run.py under tools/pmc_passes.sh)",
        "calibration": {"kernel": "dist_pairs_kernel", "known_bytes": known, "fetch_size_bytes": cal["FETCH_SIZE"] * 1024.0,
                        "correction": corr, "rdreq_breakdown_bytes": rd_bytes(cal), "rdreq_correction": rd_corr,
                        "note": "same 8-lanes-x-16-B row gather as the traversal; gfx950 FETCH_SIZE reads ~1/2 of such streams",
                        # launch durations from the same csv rows (the pure access pattern: tools/micro/probe_rate.hip)
                        "duration_ms": sum(cal_ns) / len(cal_ns) * 1e-6,
                        "rate_gbs": known / (sum(cal_ns) / len(cal_ns) * 1e-9) / 1e9},
        "traverse": {
            "kernel": kernel_of(kname), "launches": 3, "algorithmic_bytes_per_launch": alg,
            "fetch_size_bytes_per_launch": fetch, "read_bytes_per_launch": reads,
            "read_bytes_per_launch_from_rdreq": rd_bytes(trv) * rd_corr if rd_corr else None,
            "write_size_bytes_per_launch": writes,
            "ea_wrreq_per_launch": trv.get("TCC_EA0_WRREQ_sum"), "ea_wrreq_64B_per_launch": trv.get("TCC_EA0_WRREQ_64B_sum"),
            "l2_atomic_requests_per_launch": trv.get("TCC_ATOMIC_sum"),
            "ea_atomic_requests_per_launch": trv.get("TCC_EA0_ATOMIC_sum"),
            "l2_hit": trv.get("TCC_HIT_sum"), "l2_miss": trv.get("TCC_MISS_sum"),
            "l2_writebacks": trv.get("TCC_NORMAL_WRITEBACK_sum"),
            "corrected_bytes_per_launch": reads + writes,
            "traffic_over_algorithmic": (reads + writes) / alg, "reads_over_algorithmic": reads / alg,
        },
        "what_is_counted": "reads = every L2 fill from the memory side (rows, neighbour lists AND the visited-bitmap lines); "
                           "writes = dirty write-backs (visited-bitmap lines, admission logs, result lists)",
    }
    json.dump(doc, open(os.path.join(out, f"pmc_traffic_{tag}.json"), "w"), indent=1)
    print(json.dumps(doc["traverse"], indent=1))


if __name__ == "__main__":
    if len(sys.argv) >= 5 and sys.argv[1] == "traffic":
        cmd_traffic(*sys.argv[2:5])
    elif len(sys.argv) >= 3 and sys.argv[1] == "sum":
        cmd_sum(sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "")
    else:
        sys.exit(__doc__)
