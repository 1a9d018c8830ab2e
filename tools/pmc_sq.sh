#!/bin/bash
# SQ counter passes of the traversal kernel (instruction mix and wait cycles per launch).
# Usage on the GPU box: bash tools/pmc_sq.sh [N] ; results under gpurun_out/pmc_sq*/ , summary on stdout
set -u
REPO=${GRAFT_REPO_ROOT:-$PWD}
export PMC_N=${1:-1000000}
cd /tmp && export TMPDIR=/tmp
run() {
  local name=$1; shift
  timeout -s KILL 240 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$REPO/gpurun_out/pmc_$name" -- \
      python3 "$REPO/tools/pmc_run.py" > "$REPO/gpurun_out/pmc_$name.log" 2>&1 || { echo "pass $name failed"; tail -5 "$REPO/gpurun_out/pmc_$name.log"; return 1; }
}
run sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES &&
run sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_BUSY_CYCLES &&
python3 - <<PY
import csv, glob, collections
for name in ("sq1", "sq2"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for f in glob.glob("$REPO/gpurun_out/pmc_%s/**/*counter_collection.csv" % name, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][:60]
            if "traverse" not in k and "replay" not in k: continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
    for k, d in acc.items():
        print(name, k)
        for c, v in sorted(d.items()):
            print("   %-24s %16.0f per launch" % (c, v / max(cnt[(k, c)], 1)))
PY
