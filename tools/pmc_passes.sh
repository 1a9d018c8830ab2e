#!/bin/bash
# HBM-traffic counter passes of the traversal kernel (one rocprofv3 run per pass: the TCC block has 4 slots,
# FETCH_SIZE takes 3, WRITE_SIZE 2).  Usage on the GPU box: bash tools/pmc_passes.sh [N]; results under gpurun_out/pmc_*/
set -u
REPO=${GRAFT_REPO_ROOT:-$PWD}
export PMC_N=${1:-1000000}
cd /tmp && export TMPDIR=/tmp
run() {  # name, counters...
  local name=$1; shift
  echo "== pass $name: $*"
  timeout -s KILL 240 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$REPO/gpurun_out/pmc_$name" -- \
      python3 "$REPO/tools/pmc_run.py" > "$REPO/gpurun_out/pmc_$name.log" 2>&1 || { echo "pass $name failed"; tail -5 "$REPO/gpurun_out/pmc_$name.log"; return 1; }
  tail -1 "$REPO/gpurun_out/pmc_$name.log"
}
run fetch FETCH_SIZE &&
run write WRITE_SIZE TCC_ATOMIC_sum TCC_EA0_ATOMIC_sum &&
run rdreq TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_WRREQ_sum &&
run hit TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_64B_sum TCC_NORMAL_WRITEBACK_sum
